"""Gammatonegram's dense filterbank contraction (64 x 1056 over 55 168 frame-major frames): tile shapes of the fp32 kernel.
    python scripts/gammatone_tiles.py"""
import sys

import torch

sys.path.insert(0, ".")
from nnaudio_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
B, T, Fp, M = 64, 862, 1056, 64
spec = torch.rand(B, T, Fp, device=dev)
fb = torch.rand(M, Fp, device=dev)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
from nnaudio_amd import _abi  # noqa: E402

for name in ("TILE_AUTO", "TILE_32x256", "TILE_64x256", "TILE_128x128", "TILE_128x128_TALL", "TILE_128x256_SQ"):
    tile = getattr(_abi, name)
    try:
        f = lambda: engine.framed_gemm(spec.view(B, T * Fp), fb, None, hop=Fp, pad=0, pad_mode=engine.PAD_NONE,
                                       epilogue=engine.EPI_REAL, precision="fp32", tile=tile)
        y = f()
        if ref is None:
            ref = y
        print("%-20s %.4f ms   max diff vs auto %.2e" % (name, timed(f), float((y - ref).abs().max())))
    except Exception as e:  # noqa: BLE001
        print(name, "refused:", str(e)[:100])
