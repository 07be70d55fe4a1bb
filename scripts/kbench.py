#!/usr/bin/env python3
"""Kernel micro-benchmarks on the GPU box: per-workload, per-tile timings of the C-ABI calls
(events on the launch stream).  Usage: python scripts/kbench.py [stft|cqt|mel|cqt2010|fir|all]
The rows that pass `_debug` bits (ablations, A/B kernel selectors) run on the benchmarking
build of the library: python -m nnaudio_amd.build --ablate"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnaudio_amd import engine, features  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def stft():
    B, L = 64, 441000
    x = torch.randn(B, L, device=DEV)
    m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(DEV)
    flops = 2.0 * 2050 * 2048 * B * 862
    for tile in (1, 4, 7, 8, 9):
        ms = timeit(lambda: engine.framed_gemm(x, m.wcos, m.wsin, hop=512, pad=1024, pad_mode=2,
                                               epilogue=engine.EPI_MAGNITUDE, tile=tile))
        print("stft cfg2 tile %d: %.3f ms  %.1f TF (%.1f%% of 157.3)" % (tile, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573))
    sup = torch.tensor([[0, 2048]] * 1025, dtype=torch.int32, device=DEV)
    ms = timeit(lambda: engine.framed_gemm(x, m.wcos, m.wsin, hop=512, pad=1024, pad_mode=2,
                                           epilogue=engine.EPI_MAGNITUDE, tile=1, row_support=sup))
    print("stft cfg2 tile 1 MASKED single launch (17 row blocks): %.3f ms" % ms)
    x8 = x[:, : 441000 - 512 * 6]  # 856 frames -> 54784 = 428*128 columns
    ms = timeit(lambda: engine.framed_gemm(x, m.wcos[:1024], m.wsin[:1024], hop=512, pad=1024, pad_mode=2,
                                           epilogue=engine.EPI_MAGNITUDE, tile=1))
    print("stft 1024 bins only (16 full row blocks, unmasked): %.3f ms -> %.1f TF" % (ms, 2.0*2048*2048*64*862/ms/1e9))
    for dbg, what in ((0, "auto (LDS-direct)"), (0x800, "register-staged"), (0x100, "frame-tile-fastest order"), (0x200, "L2-blocked order"), (1, "no global loads"), (16, "no MFMA (load path only)"), (16 + 8, "no MFMA, no frag reads")):
        ms = timeit(lambda: engine.framed_gemm(x, m.wcos[:1024], m.wsin[:1024], hop=512, pad=1024, pad_mode=2,
                                               epilogue=engine.EPI_MAGNITUDE, tile=1, _debug=dbg))
        print("ablate[%-28s] tile 1, 1024 bins: %.3f ms -> %.1f TF (%.1f%%)" % (what, ms, 2.0*2048*2048*64*862/ms/1e9, 2.0*2048*2048*64*862/ms/1e9/1.573))
    for fmt in ("Complex", "Phase"):
        ms = timeit(lambda: m(x, output_format=fmt))
        print("stft cfg2 %s: %.3f ms" % (fmt, ms))


def fold():
    """Symmetric-fold STFT path (framed_fold.inl): pre-pass + contraction, and ablations of the
    contraction's K loop (benchmarking build)."""
    B, L = 64, 441000
    x = torch.randn(B, L, device=DEV)
    m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(DEV)
    prep = engine.prepare_basis(m.wcos, m.wsin, "bf16x3", hop=512)
    for dbg, what in ((0, "full step (pre-pass + contraction)"), (1, "no LDS-DMA in the loop"),
                      (8, "no fragment reads"), (1 + 8, "no DMA, no fragment reads (MFMA + barrier)"),
                      (1 + 8 + 4, "MFMAs only"), (0x40000, "no epilogue"), (0x40000 + 1, "no epilogue, no DMA"),
                      (0x100000, "dense (unfolded) kernel")):
        ms = timeit(lambda: engine.framed_gemm(x, m.wcos, m.wsin, hop=512, pad=1024, pad_mode=2,
                                               epilogue=engine.EPI_MAGNITUDE, precision="bf16x3",
                                               _debug=dbg, **prep), n=30, w=10)
        print("fold[%-44s] %.3f ms" % (what, ms))


def pyramid():
    """Phase clock of one persistent workgroup of the fused octave kernel, first launch of the cfg5
    chain (octaves 0-2 from x): 100 MHz ticks between the kernel's stamps."""
    B, L = 64, 1323000
    x = torch.randn(B, L, device=DEV)
    m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False).to(DEV)
    m.precision = "bf16x3"
    m(x)
    sp = engine.split_basis(m.cqt_kernels_real, m.cqt_kernels_imag)
    out = torch.empty(B, 96, 2584, device=DEV)
    x2 = torch.empty(B, 330750, device=DEV)
    lv = [dict(split=sp, n_bins=12, kernel=256, row_offset=84 - 12 * i, pad_mode=2, row_scale=None)
          for i in range(3)]
    st = torch.zeros(64, dtype=torch.int64, device=DEV)
    for _ in range(3):
        engine.octave_pyramid(x, lv, hop=512, n_frames=2584, taps=m.lowpass_filter, epilogue=1,
                              im_sign=-1.0, eps=0.0, out=out, x_last=x2, _stamps=st)
    torch.cuda.synchronize()
    t = st.cpu().numpy()
    d = (t[1:] - t[:-1]) * 10  # ns
    print("stamps: set-up | per item: [start, span in LDS, FIR 0->1, FIR 1->2, fix-ups] ...")
    print("set-up -> first item start: %d ns" % d[0])
    for k in range(1, min(len(d), 41), 5):
        print("  item: load+split %5d  FIR1 %5d  FIR2 %5d  fixup %5d  phaseC+sync %5d ns" % tuple(d[k:k + 5]))


def bf16():
    """precision="bf16x3": error against the fp32 kernel and timings of both wave layouts."""
    B, L = 64, 441000
    x = torch.randn(B, L, device=DEV)
    m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(DEV)
    flops = 2.0 * 2050 * 2048 * B * 862
    ref = m(x)
    split = engine.split_basis(m.wcos, m.wsin)
    kw = dict(hop=512, pad=1024, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, precision="bf16x3",
              basis_split=split)
    split1k = engine.split_basis(m.wcos[:1024], m.wsin[:1024])
    kw1k = dict(hop=512, pad=1024, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, precision="bf16x3")
    for tile, what in ((0, "8 waves 64x128"),):
        y = engine.framed_gemm(x, m.wcos, m.wsin, tile=tile, **kw)
        err = float((y - ref).abs().max() / ref.abs().max())
        ms = timeit(lambda: engine.framed_gemm(x, m.wcos, m.wsin, tile=tile, **kw))
        print("stft cfg2 bf16x3 [%s]: %.3f ms  %.1f TF-equivalent (%.1f%% of 833), max err %.2e of peak"
              % (what, ms, flops / ms / 1e9, flops / ms / 1e9 / 8.33, err))
        ms = timeit(lambda: engine.framed_gemm(x, m.wcos[:1024], m.wsin[:1024], tile=tile, hop=512, pad=1024,
                                               pad_mode=2, epilogue=engine.EPI_MAGNITUDE, precision="bf16x3"))
        print("   1024 bins incl. uncached basis split: %.3f ms" % ms)
    for dbg, what in ((0, "full"), (0x40000, "no epilogue"), (0x80000, "2 K stages only"),
                      (0xC0000, "2 K stages, no epilogue")):
        ms = timeit(lambda: engine.framed_gemm(x, m.wcos[:1024], m.wsin[:1024], tile=0, _debug=dbg,
                                               basis_split=split1k, **kw1k))
        print("   ablate[%-36s] 1024 bins: %.3f ms" % (what, ms))
    m.precision = "bf16x3"
    for fmt in ("Magnitude", "Complex", "Phase"):
        ms = timeit(lambda: m(x, output_format=fmt))
        print("stft cfg2 bf16x3 module %s: %.3f ms" % (fmt, ms))
    c = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to(DEV)
    ref = c(x)
    ms0 = timeit(lambda: c(x), n=5, w=2)
    c.precision = "bf16x3"
    y = c(x)
    ms = timeit(lambda: c(x), n=5, w=2)
    print("cqt1992v2 84 bins: fp32 %.3f ms, bf16x3 %.3f ms, max err %.2e of peak"
          % (ms0, ms, float((y - ref).abs().max() / ref.abs().max())))


def mel():
    B, L = 256, 110250
    x = torch.randn(B, L, device=DEV)
    m = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, verbose=False).to(DEV)
    ms = timeit(lambda: m(x))
    print("mel cfg3 total: %.3f ms" % ms)
    ms1 = timeit(lambda: m.stft._spectrum(x[:, None, :], engine.EPI_POWER))
    sp = m.stft._spectrum(x[:, None, :], engine.EPI_POWER)
    ms2 = timeit(lambda: engine.filterbank(m.mel_basis, sp))
    print("   stft-power %.3f ms, filterbank %.3f ms" % (ms1, ms2))
    m.stft.precision = "bf16x3"
    ms = timeit(lambda: m(x))
    ms1 = timeit(lambda: m.stft._spectrum(x[:, None, :], engine.EPI_POWER))
    print("mel cfg3 bf16x3 (filterbank fused into the epilogue): %.3f ms; unfused: stft-power %.3f + filterbank %.3f ms"
          % (ms, ms1, ms2))


def cqt():
    B, L = 64, 441000
    x = torch.randn(B, L, device=DEV)
    m = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to(DEV)
    useful = 2.0 * 2 * float(m.lenghts.sum()) * B * 862
    dense = 2.0 * 168 * 32768 * B * 862
    sup = m._support.get(m.cqt_kernels_real, m.cqt_kernels_imag)
    sc = torch.sqrt(m.lenghts)
    for dbg in (16, 16 + 8):
        ms = timeit(lambda: engine.framed_gemm(x, m.cqt_kernels_real, m.cqt_kernels_imag, hop=512,
                                               pad=16384, pad_mode=2, epilogue=engine.EPI_MAGNITUDE,
                                               row_scale=sc, row_support=sup, tile=4, _debug=dbg), n=5, w=2)
        print("cqt1992v2 tile 4 ablation dbg=%d: %.3f ms" % (dbg, ms))
    for tile, s in ((1, sup), (4, sup)):
        ms = timeit(lambda: engine.framed_gemm(x, m.cqt_kernels_real, m.cqt_kernels_imag, hop=512,
                                               pad=16384, pad_mode=2, epilogue=engine.EPI_MAGNITUDE,
                                               row_scale=sc, row_support=s, tile=tile), n=5, w=2)
        print("cqt1992v2 B=64 tile %d support=%s: %.3f ms  useful %.1f TF dense-equiv %.1f TF"
              % (tile, s is not None, ms, useful / ms / 1e9, dense / ms / 1e9))


def cqt2010():
    B, L = 64, 1323000
    x = torch.randn(B, L, device=DEV)
    for cls, kw in ((features.CQT2010v2, {}), (features.VQT, dict(gamma=0))):
        m = cls(sr=44100, hop_length=512, n_bins=96, verbose=False, **kw).to(DEV)
        ms = timeit(lambda: m(x), n=5, w=2)
        print("%s cfg5 shard B=64: %.3f ms" % (cls.__name__, ms))
    lp = m.lowpass_filter
    xs = x
    for o in range(3):
        ms = timeit(lambda: engine.fir_decimate(xs, lp, 2), n=5, w=2)
        print("   fir_decimate L=%d: %.3f ms (%.1f GB/s in+out, %.1f TF)" % (
            xs.shape[-1], ms, (xs.numel() * 6) / ms / 1e6, 2.0 * 256 * xs.numel() / 2 / ms / 1e9))
        xs = engine.fir_decimate(xs, lp, 2)
    kr, ki = m.cqt_kernels_real_0, m.cqt_kernels_imag_0
    ms = timeit(lambda: engine.framed_gemm(x, kr, ki, hop=512, pad=128, pad_mode=2,
                                           epilogue=engine.EPI_MAGNITUDE), n=5, w=2)
    print("   top-octave framed gemm (12 bins, K=256, hop 512): %.3f ms" % ms)


if __name__ == "__main__":
    which = sys.argv[1:] or ["all"]
    torch.manual_seed(0)
    for name, fn in (("stft", stft), ("fold", fold), ("pyramid", pyramid), ("bf16", bf16), ("mel", mel), ("cqt", cqt), ("cqt2010", cqt2010)):
        if "all" in which or name in which:
            t0 = time.time()
            fn()
            torch.cuda.synchronize()
            print("  [%s done in %.1f s]" % (name, time.time() - t0), flush=True)


def cqtsplit():
    """Where the CQT84 time goes by row tile: the narrow bf16x3 kernel on prefixes / suffixes of the bank."""
    B, L = 64, 441000
    x = torch.randn(B, L, device=DEV)
    c = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to(DEV)
    sup = c._support.get(c.cqt_kernels_real, c.cqt_kernels_imag)
    sc = torch.sqrt(c.lenghts)
    kr, ki = c.cqt_kernels_real.reshape(84, -1), c.cqt_kernels_imag.reshape(84, -1)
    for lo, hi in ((0, 84), (0, 16), (0, 32), (0, 48), (48, 84), (16, 84), (32, 84), (64, 84)):
        r, i = kr[lo:hi].contiguous(), ki[lo:hi].contiguous()
        spl = engine.split_basis(r, i)
        for tile in (0,):
            ms = timeit(lambda: engine.framed_gemm(x, r, i, hop=512, pad=16384, pad_mode=2,
                                                   epilogue=engine.EPI_MAGNITUDE, row_scale=sc[lo:hi].contiguous(),
                                                   row_support=sup[lo:hi].contiguous(), precision="bf16x3",
                                                   basis_split=spl, tile=tile), n=5, w=2)
            print("cqt bins [%2d, %2d): %.3f ms" % (lo, hi, ms))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "cqtsplit":
    cqtsplit()
