"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):

    python tools/pmc_mfma_summary.py <dir with p1/p1_counter_collection.csv> "<command description>" [steps]

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs) (the gfx94x derived-metric formula; ROCm 7.2 ships none for gfx950).
A 16x16x32 bf16 MFMA keeps its SIMD's matrix pipe busy for 16 cycles at the 2.5 PFLOP/s peak, so MfmaUtil x (shader clock / 2.4 GHz) is the
fraction of that peak the executed MFMAs amount to -- independent of the algorithmic FLOP count bench.py uses."""
import collections
import csv
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("HIP_vector_type<unsigned int, 4u>", "uint4")[:88]


def main():
    out, command = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    calls, dur = collections.Counter(), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(os.path.join(out, "p1", "p1_counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key)
            calls[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    XCD, SIMDS, PEAK_HZ = 8, 1024, 2.4e9
    tot_busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] for v in rows.values())
    tot_act = sum(v["GRBM_GUI_ACTIVE"] for v in rows.values())
    tot_dur = sum(dur.values())
    lines = [f"`{command}`", "",
             "Kernels one at a time (--serialize).  `MfmaUtil` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs",
             "(check: GUI_ACTIVE / 8 / duration = the shader clock column).  `of 2.5 PF` = busy cycles / (1024 SIMDs x duration x 2.4 GHz): a 16x16x32 bf16 MFMA",
             "holds its SIMD's matrix pipe for 16 cycles at the 2.5 PFLOP/s peak, so this is the fraction of that peak the EXECUTED MFMAs amount to --",
             "measured by the hardware, independent of the algorithmic FLOP count bench.py uses.", "",
             "| kernel | launches / step | ms / step | MFMA busy (Gcycles / step) | shader clock (GHz) | MfmaUtil | of 2.5 PF |", "|---|---|---|---|---|---|---|"]
    for k, v in sorted(rows.items(), key=lambda kv: -dur[kv[0]]):
        act, busy, d = v["GRBM_GUI_ACTIVE"], v["SQ_VALU_MFMA_BUSY_CYCLES"], dur[k]
        if d / steps < 3e-4:
            continue
        lines.append(f"| `{k}` | {calls[k] / steps:g} | {1e3 * d / steps:.2f} | {busy / steps / 1e9:.2f} | {act / XCD / d / 1e9:.2f} | "
                     f"{busy / (act / XCD * SIMDS):.3f} | {busy / (SIMDS * d * PEAK_HZ):.3f} |")
    lines += ["", f"whole step (sum over all kernels, serialized): {1e3 * tot_dur / steps:.1f} ms of kernel time, MFMA busy {tot_busy / steps / 1e9:.2f} Gcycles -> "
                  f"**MfmaUtil {tot_busy / (tot_act / XCD * SIMDS):.3f}, {tot_busy / (SIMDS * tot_dur * PEAK_HZ):.3f} of the 2.5 PFLOP/s peak** "
                  f"(executed MFMA work: {tot_busy / steps / 16 * 16384 / 1e12:.1f} TFLOP per step)"]
    text = "\n".join(lines) + "\n"
    open(os.path.join(out, "summary.md"), "w").write(text)
    import json
    doc = {"command": command, "steps_profiled": steps, "kernel_ms_per_step": 1e3 * tot_dur / steps, "mfma_busy_cycles_per_step": tot_busy / steps,
           "executed_mfma_tflop_per_step": tot_busy / steps / 16 * 16384 / 1e12, "mfma_util": tot_busy / (tot_act / XCD * SIMDS),
           "frac_of_2p5_pflops_over_kernel_time": tot_busy / (SIMDS * tot_dur * PEAK_HZ),
           "kernels": {k: {"launches": calls[k] / steps, "ms_per_step": 1e3 * dur[k] / steps, "mfma_util": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / XCD * SIMDS),
                           "frac_of_2p5_pflops": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * dur[k] * PEAK_HZ)}
                       for k, v in rows.items() if v["SQ_VALU_MFMA_BUSY_CYCLES"] > 0 and dur[k] / steps >= 3e-4}}
    json.dump(doc, open(os.path.join(out, "mfma_util.json"), "w"), indent=1)       # copy to profiles/mfma_util.json: bench.py quotes it
    print(text)


if __name__ == "__main__":
    main()
