"""Phase clock of the n_fft = 2048 FFT kernel (a variant built with -DMISPEC_FFT_STAMPS=1: scripts/build_variant.py st
MISPEC_FFT_STAMPS=1 [...]): s_memtime of waves 0 and 4 of workgroup 8 at the phase boundaries of tile steps 4 .. 11, cfg2.
    VARIANT=st python scripts/fft_stamps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import _abi, features  # noqa: E402

name = os.environ.get("VARIANT", "st")
lib = _abi._load(os.path.join(os.path.dirname(_abi.LIB_PATH), "libmispec_%s.so" % name), "scripts/build_variant.py")
_abi._lib = lib
dev = "cuda:0"
what = os.environ.get("WHAT", "stft2048")
if what == "mel":      # cfg3
    m = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to(dev)
    x = torch.randn(256, 110250, device=dev)
elif what == "stft1024":
    m = features.STFT(n_fft=1024, hop_length=256, output_format="Magnitude", verbose=False).to(dev)
    x = torch.randn(64, 441000, device=dev)
else:
    m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(dev)
    x = torch.randn(64, 441000, device=dev)
print("workload:", what)
stamps = torch.zeros(2 * 8 * 16, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(30):
        m(x)
    torch.cuda.synchronize()
    os.environ["MISPEC_FFT_STAMPS"] = str(stamps.data_ptr())
    m(x)
    torch.cuda.synchronize()
    del os.environ["MISPEC_FFT_STAMPS"]
s = stamps.cpu().numpy().reshape(2, 8, 16)
names = ["top", "flush/request", "wait vm", "landing+window", "pass 0", "exchange", "pass 1", "row swaps", "pass 2 + sync",
         "mirror reads + request", "post + tile writes", "late flush", "barrier"]
print("cycles (s_memtime) per phase; columns = tile steps 4 .. 11 of workgroup 8")
for w in range(2):
    print("wave %d" % (0 if w == 0 else 4))
    for k in range(1, 13):
        d = s[w, :, k] - s[w, :, k - 1]
        print("  %-24s %s   mean %6.0f" % (names[k], " ".join("%6d" % v for v in d), d.mean()))
    tot = s[w, 1:, 0] - s[w, :-1, 0]
    print("  %-24s %s   mean %6.0f" % ("step (top to top)", " ".join("%6d" % v for v in tot), tot.mean()))
print("wave 4's top minus wave 0's top:", " ".join("%6d" % v for v in (s[1, :, 0] - s[0, :, 0])))
