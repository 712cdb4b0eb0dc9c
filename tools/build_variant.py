"""Builds a variant of libpar_hip.so with extra flags for sinc.hip (phase-timing experiments, A/B candidates):
    python tools/build_variant.py tools/ab/libpar_exp1.so -DPAR_SINC_EXP=1
The other objects come from the regular in-tree build (run `python -m pyaudiorestoration_amd.build` first)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyaudiorestoration_amd import build as B


def main():
    out, flags = sys.argv[1], sys.argv[2:]
    src = "sinc.hip"
    for f in flags:
        if f.startswith("--src="):
            src = f[6:]
    flags = [f for f in flags if not f.startswith("--src=")]
    B.build()
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    obj = os.path.abspath(out) + "." + src[:-4] + ".o"
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + B.FLAGS + B.PER_FILE.get(src, []) + flags +
                          ["-c", os.path.join(B.CSRC, src), "-o", obj])
    objs = [o for o in glob.glob(os.path.join(B.OBJ, "*.o")) if os.path.basename(o) != src[:-4] + ".o"] + [obj]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    os.remove(obj)
    print("built", out)


if __name__ == "__main__":
    main()
