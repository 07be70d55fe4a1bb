"""GPU check of the chain kernel (cqt_chain.hip): bit identity with the sequential reference kernel on random nested
banks, then CQT1992v2 at the bench shape: equality with the tile kernels and the time of both.
    python scripts/chain_check.py [--quick]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from nnaudio_amd import engine  # noqa: E402
from nnaudio_amd.features import CQT1992v2  # noqa: E402


def bank(F, K, rng, ratio=12.0, min_len=8):
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    sup = np.zeros((F, 2), np.int32)
    for f in range(F):
        ln = max(min_len, int(K * 2.0 ** (-f / ratio)))
        lo = (K - ln) // 2
        hi = lo + ln
        sup[f] = (lo, hi)
        wr[f, :lo] = wr[f, hi:] = 0.0
        wi[f, :lo] = wi[f, hi:] = 0.0
    return wr, wi, sup


def check_random(dev):
    cases = [  # F, K, hop, B, L, pad_mode, epilogue
        (84, 4096, 256, 2, 20000, engine.PAD_REFLECT, engine.EPI_COMPLEX),
        (84, 4096, 512, 3, 30000, engine.PAD_REFLECT, engine.EPI_COMPLEX),
        (130, 2048, 512, 1, 30000, engine.PAD_REFLECT, engine.EPI_MAGNITUDE),
        (7, 1024, 64, 5, 3000, engine.PAD_ZERO, engine.EPI_COMPLEX),
        (24, 1000, 128, 2, 9000, engine.PAD_ZERO, engine.EPI_COMPLEX),
        (40, 2048, 192, 2, 12345, engine.PAD_REFLECT, engine.EPI_COMPLEX),
        (33, 512, 320, 4, 7001, engine.PAD_NONE, engine.EPI_COMPLEX),
    ]
    ok = True
    for F, K, hop, B, L, pm, epi in cases:
        rng = np.random.default_rng(F * K + hop)
        wr, wi, sup = bank(F, K, rng)
        x = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).to(dev)
        wr_d, wi_d, sup_d = (torch.as_tensor(t).to(dev) for t in (wr, wi, sup))
        chain = engine.chain_basis_f32(wr_d, wi_d, sup)
        assert chain is not None, "chain plan refused a nested bank"
        pad = 0 if pm == engine.PAD_NONE else K // 2
        scale = torch.as_tensor(rng.uniform(0.5, 2.0, F).astype(np.float32)).to(dev)
        kw = dict(hop=hop, pad=pad, pad_mode=pm, epilogue=epi, im_sign=-1.0, precision="fp32", row_scale=scale)
        y = engine.framed_gemm(x, wr_d, wi_d, row_support=sup_d, row_support_host=sup, basis_chain=chain, **kw)
        ref = engine.framed_gemm(x, wr_d, wi_d, reference_kernel=True, **kw)
        torch.cuda.synchronize()
        same = torch.equal(y, ref)
        err = float((y - ref).abs().max())
        print("F=%d K=%d hop=%d B=%d L=%d pad_mode=%d epi=%d: %s (max diff %.3e, ref peak %.3e)"
              % (F, K, hop, B, L, pm, epi, "SAME BITS" if same else "DIFFERENT", err, float(ref.abs().max())), flush=True)
        ok = ok and same
    return ok


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n


def check_module(dev, B=64, seconds=10.0, sr=44100):
    mod = CQT1992v2(sr=sr, hop_length=512, fmin=32.7, n_bins=84, bins_per_octave=12, output_format="Magnitude", verbose=False).to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, int(seconds * sr), generator=g).to(dev)
    with torch.no_grad():
        y = mod(x)
        mod.chain = False
        y0 = mod(x)
        mod.chain = True
        torch.cuda.synchronize()
        same = torch.equal(y, y0)
        print("CQT1992v2 B=%d %.0f s: chain vs tile kernels: %s (max diff %.3e)" % (B, seconds, "SAME BITS" if same else "DIFFERENT", float((y - y0).abs().max())), flush=True)
        t1 = timed(lambda: mod(x))
        mod.chain = False
        t0 = timed(lambda: mod(x))
        mod.chain = True
    print("time per forward: chain %.4f ms, tile kernels %.4f ms" % (t1, t0), flush=True)
    return same


def ablate(dev, B=64, seconds=10.0, sr=44100):
    import os

    mod = CQT1992v2(sr=sr, hop_length=512, fmin=32.7, n_bins=84, bins_per_octave=12, output_format="Magnitude", verbose=False).to(dev)
    x = torch.randn(B, int(seconds * sr), generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        for bits in (0, 1, 2, 4, 6, 7, 3, 5):
            os.environ["MISPEC_CHAIN_DEBUG"] = str(bits)
            print("debug %d (1 no MFMA, 2 no ring DMA, 4 no brick DMA): %.4f ms" % (bits, timed(lambda: mod(x))), flush=True)
    os.environ["MISPEC_CHAIN_DEBUG"] = "0"


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    if "--one" in sys.argv:  # (for rocprofv3: a few forwards of the bench shape)
        mod = CQT1992v2(sr=44100, hop_length=512, fmin=32.7, n_bins=84, bins_per_octave=12, output_format="Magnitude", verbose=False).to(dev)
        x = torch.randn(64, 441000, generator=torch.Generator().manual_seed(0)).to(dev)
        with torch.no_grad():
            print("%.4f ms" % timed(lambda: mod(x), n=5))
        sys.exit(0)
    if "--small" in sys.argv:  # one work unit per CU at most: the time of the longest unit
        mod = CQT1992v2(sr=44100, hop_length=512, fmin=32.7, n_bins=84, bins_per_octave=12, output_format="Magnitude", verbose=False).to(dev)
        x = torch.randn(2, 3 * 44100, generator=torch.Generator().manual_seed(0)).to(dev)
        with torch.no_grad():
            print("%.4f ms" % timed(lambda: mod(x), n=5))
        sys.exit(0)
    if "--ablate" in sys.argv:
        ablate(dev)
        sys.exit(0)
    t = time.time()
    ok = check_random(dev)
    if "--quick" not in sys.argv:
        ok = check_module(dev, B=2, seconds=3.0) and ok
        ok = check_module(dev) and ok
    print("chain_check:", "OK" if ok else "FAILED", "(%.1f s)" % (time.time() - t))
    sys.exit(0 if ok else 1)
