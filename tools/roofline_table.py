"""Per-kernel roofline table from a kernel-trace profile and the PMC traffic passes of the same code:
    python tools/roofline_table.py gpurun_out/prof_<tag> gpurun_out/pmcbench_<tag> <n_steps_in_profile> > profiles/<name>.md
Time per launch comes from the (serialized) kernel trace, HBM bytes per launch from FETCH_SIZE x2 + WRITE_SIZE (one counter per
pass, MI355X_MICROARCH.md), FLOP per launch of the convolution kernels from the layer shapes (ResNet-18, 32x32, chunk group 98)."""
import collections
import csv
import sys

PEAK_TF, PEAK_TBS = 2500.0, 6.3          # dense bf16 MFMA; achievable HBM (float4 copy, MI355X_MICROARCH.md)


def main():
    prof, pmc, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    t = {}
    for r in csv.DictReader(open(f"{prof}/bench_kernel_stats.csv")):
        t[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    by = collections.defaultdict(lambda: [0.0, 0.0, 0, 0])
    for p, col in (("p1", 0), ("p2", 1)):
        for r in csv.DictReader(open(f"{pmc}/{p}/{p}_counter_collection.csv")):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                by[r["Kernel_Name"]][col] += float(r["Counter_Value"]) * 1e3
                by[r["Kernel_Name"]][2 + col] += 1
    # conv FLOP per image by kernel family (2 * px * Cout * taps * Cin): fwd + dgrad launches share kernels
    img = 98 * 128
    conv = {"conv3x3s1_c64_halo5": 2 * 1024 * 64 * 9 * 64, "conv3x3s1_halo4_kernel<bf16_tag, 16,": 2 * 256 * 128 * 9 * 128,
            "conv3x3s1_halo4_kernel<bf16_tag, 8,": 2 * 64 * 256 * 9 * 256, "conv3x3s1_halo4_kernel<bf16_tag, 4,": 2 * 16 * 512 * 9 * 512,
            "conv_wgrad3x3_v2_kernel<32, 1,": 2 * 1024 * 64 * 9 * 64,
            "conv_wgrad3x3_v2_kernel<16, 1,": 2 * 256 * 128 * 9 * 128, "conv_wgrad3x3_v2_kernel<8, 1,": 2 * 64 * 256 * 9 * 256,
            "conv_wgrad3x3_v2_kernel<4, 1,": 2 * 16 * 512 * 9 * 512}
    print("| kernel | launches / step | us / launch | ms / step | HBM MB / launch | TB/s | of 6.3 | TFLOP/s | of 2500 | nearer wall |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    rows = []
    for name, (calls, ns) in t.items():
        us = ns / calls / 1e3
        ms_step = ns / steps / 1e6
        if ms_step < 0.5:
            continue
        f, w, nf, nw = by.get(name, [0, 0, 0, 0])
        mb = (2 * f / max(nf, 1) + w / max(nw, 1)) / 1e6 if nf else float("nan")
        tbs = mb / us if nf else float("nan")                          # MB / us = TB/s
        tf = next((v * img / (us * 1e-6) / 1e12 for k, v in conv.items() if k in name), None)
        wall = "-"
        if nf:
            fh, fm = tbs / PEAK_TBS, (tf / PEAK_TF if tf else 0.0)
            wall = "HBM" if fh >= fm else "MFMA"
        rows.append((ms_step, f"| `{name[:60]}` | {calls / steps:.0f} | {us:.0f} | {ms_step:.1f} | {mb:.0f} | {tbs:.2f} | {tbs / PEAK_TBS:.2f} | "
                              f"{'%.0f' % tf if tf else '-'} | {'%.2f' % (tf / PEAK_TF) if tf else '-'} | {wall} |"))
    for _, line in sorted(rows, reverse=True):
        print(line)


if __name__ == "__main__":
    main()
