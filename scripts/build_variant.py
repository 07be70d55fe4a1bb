"""Build a variant of the library for A/B timing on the GPU box:
    python scripts/build_variant.py NAME DEFINE[=VALUE] ...
compiles csrc/mispec.hip with the defines into csrc/libmispec_NAME.so (linked with the shipped octave_stream
object; git-ignored like every .so, travels with gpurun) and prints registers / scratch of the FFT kernels."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnaudio_amd import build  # noqa: E402

name, defines = sys.argv[1], sys.argv[2:]
csrc = os.path.dirname(build.SRC)
obj = os.path.join(csrc, "_obj", "mispec_%s.o" % name)
out = os.path.join(csrc, "libmispec_%s.so" % name)
cmd = [build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-c",
       "-Rpass-analysis=kernel-resource-usage", "-I", build.INC, build.SRC, "-o", obj] + ["-D" + d for d in defines]
res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
if res.returncode:
    sys.stderr.write(res.stderr[-4000:])
    sys.exit(1)
kern = None
for line in res.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        kern = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1))
    m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and kern and kern.startswith("stft_fft_kernel") and m.group(1) != "LDS Size [bytes/block]":
        print("%s %s %s: %s" % (name, kern[:36], m.group(1), m.group(2)))
subprocess.run([build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", obj,
                os.path.join(csrc, "_obj", "octave_stream.o"), "-o", out], check=True)
print("built", out)
