"""Build ``csrc/libmispec.so`` in-tree with hipcc for gfx950 (``python -m nnaudio_amd.build``)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "mispec.hip")
OUT = os.path.join(HERE, "csrc", "libmispec.so")
INC = os.path.join(ROOT, "include")


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def build(force=False, verbose=True):
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(INC, "mispec.h")] + [
        os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".inl", ".h"))
    ]
    if (not force and os.path.exists(OUT)
            and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)):
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-I", INC, SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
