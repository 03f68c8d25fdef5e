"""Registers, LDS and resident workgroups per CU of every kernel in csrc/build/*.o (from the code objects' metadata notes).
Usage: python tools/kernel_resources.py [substring ...]      (needs the in-tree build; no GPU)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LDS_PER_CU, VGPR_PER_SIMD = 160 * 1024, 512


def kernels_of(obj, tmp):
    fat, out = os.path.join(tmp, os.path.basename(obj) + ".fatbin"), os.path.join(tmp, os.path.basename(obj) + ".co")
    if subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj], capture_output=True).returncode != 0:
        return []                                             # host-only translation unit
    res = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}",
                          f"--output={out}"], capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        return []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
    ks = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda key: (re.search(rf"\.{key}:\s+(\S+)", blk) or [None, "0"])[1]
        name = get("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        ks.append(dict(name=dem, vgpr=int(get("vgpr_count")), agpr=int(get("agpr_count")), sgpr=int(get("sgpr_count")), lds=int(get("group_segment_fixed_size")),
                       scratch=int(get("private_segment_fixed_size")), wg=int(get("max_flat_workgroup_size"))))
    return ks


def main():
    pats = sys.argv[1:]
    bdir = os.path.join(ROOT, "fullbatchtraining_amd", "csrc", "build")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(bdir)):
            if f.endswith(".o"):
                rows += kernels_of(os.path.join(bdir, f), tmp)
    print("| kernel | VGPR (+AGPR) | LDS KiB | scratch | waves/SIMD by regs | workgroups/CU by LDS | resident workgroups/CU (256 threads) |")
    print("|---|---|---|---|---|---|---|")
    for k in rows:
        if pats and not any(p in k["name"] for p in pats):
            continue
        regs = k["vgpr"]                                  # unified register file: vgpr_count already includes the AGPRs on gfx950
        alloc = (regs + 7) // 8 * 8
        by_regs = min(8, VGPR_PER_SIMD // max(alloc, 1))
        by_lds = LDS_PER_CU // k["lds"] if k["lds"] else 99
        waves_per_wg_per_simd = max(1, k["wg"] // 256)
        res = min(by_regs // waves_per_wg_per_simd, by_lds)
        print(f"| `{k['name'][:90]}` | {k['vgpr']} ({k['agpr']}) | {k['lds'] / 1024:.1f} | {k['scratch']} | {by_regs} | {by_lds} | {res} |")


if __name__ == "__main__":
    main()
