"""A/B builds: compiles csrc with extra hipcc flags into csrc/variants/libfbengine_<tag>.so (selected at run time with FB_LIB_PATH).

    python tools/build_variant.py nont -DFB_NO_NT
    python tools/build_variant.py late0 --only conv1x1_pipe.hip -DP1_LATE=0      (only that source with the flags; the other objects from the main build)
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fullbatchtraining_amd import build as B  # noqa: E402


def main():
    tag, extra = sys.argv[1], sys.argv[2:]
    only = None
    if extra and extra[0] == "--only":
        only, extra = set(extra[1].split(",")), extra[2:]
        B.build(force=False, verbose=False)                  # the main build's objects are current
    objdir = os.path.join(B.CSRC, "build_" + tag)
    outdir = os.path.join(B.CSRC, "variants")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(outdir, exist_ok=True)
    lib = os.path.join(outdir, f"libfbengine_{tag}.so")

    def compile_one(src):
        if only is not None and src not in only:
            return os.path.join(B.CSRC, "build", src.rsplit(".", 1)[0] + ".o")
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        res = subprocess.run([B.HIPCC, *B.FLAGS, *extra, "-x", "hip", "-c", os.path.join(B.CSRC, src), "-o", obj], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{res.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=6) as pool:
        objs = list(pool.map(compile_one, B.sources()))
    res = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(res.stderr)
    print("built", lib)


if __name__ == "__main__":
    main()
