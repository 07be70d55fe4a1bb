"""The streaming octave kernel (csrc/octave_stream.hip, ``mispec_octave_stream_f32``) on the GPU:
kernel-level parity with the float64 recursion -- zero-padded stride-2 FIR decimation (utils.py:73-124)
+ reflect- / zero-padded frames of every level (utils.py:498-521) -- over the shapes its fast and slow
paths split on (segments, clip ends, hops that are not multiples of 8, 8-step banks, a loud burst that
forces the fp16 rescale), and the modules on it against the pyramid kernel and fp32."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import octave_stream_model as M  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _taps():
    return (np.hanning(256) * np.sinc((np.arange(256) - 127.5) / 2) / 2).astype(np.float32)


def _run_kernel(x, K, hop0, n_seg, precision, reflect, n_bins=12):
    from nnaudio_amd import engine

    rng = np.random.default_rng(1)
    B, L0 = x.shape
    n_frames = L0 // hop0 + 1
    banks = [None if not k else ((rng.standard_normal((n_bins, k)) + 1j * rng.standard_normal((n_bins, k))) / k).astype(np.complex64)
             for k in K]
    levels, row0 = [], 0
    for b in banks:
        if b is None:
            levels.append(None)
            continue
        re = torch.as_tensor(np.ascontiguousarray(b.real)).to(DEV)
        im = torch.as_tensor(np.ascontiguousarray(b.imag)).to(DEV)
        split = engine.split_basis_f16(re, im) if precision == "f16x3" else engine.split_basis(re, im)
        levels.append(dict(split=split, n_bins=n_bins, kernel=b.shape[1], row_offset=row0,
                           pad_mode=engine.PAD_REFLECT if reflect else engine.PAD_ZERO, row_scale=None))
        row0 += n_bins
    out = torch.full((B, row0, n_frames, 2), float("nan"), device=DEV)
    Ls = [L0]
    for _ in range(len(K) - 1):
        Ls.append(M.decimated_length(Ls[-1]))
    x_last = torch.full((B, (Ls[-1] + 3) // 4 * 4), float("nan"), device=DEV)[:, :Ls[-1]]
    taps = _taps()
    ok = engine.octave_stream(torch.as_tensor(x).to(DEV), levels, hop=hop0, n_frames=n_frames,
                              taps=torch.as_tensor(taps).to(DEV), epilogue=engine.EPI_COMPLEX, im_sign=1.0, eps=0.0,
                              out=out, x_last=x_last, precision=precision, fir_headroom_bits=len(K) - 1,
                              n_segments=n_seg)
    torch.cuda.synchronize()
    assert ok, "the library refused the shape"
    y, xl = out.cpu().numpy(), x_last.cpu().numpy()
    worst = 0.0
    for b in range(B):
        ref, xs = M.reference(x[b].astype(np.float64), taps.astype(np.float64),
                              [None if k is None else k.astype(np.complex128) for k in banks], hop0, n_frames, reflect)
        r0 = 0
        for r in ref:
            if r is None:
                continue
            got = y[b, r0:r0 + n_bins, :, 0] + 1j * y[b, r0:r0 + n_bins, :, 1]
            assert np.isfinite(got).all(), "rows %d: unwritten / non-finite outputs" % r0
            worst = max(worst, np.abs(got - r).max() / np.abs(r).max())
            r0 += n_bins
        assert np.isfinite(xl[b]).all(), "x_last: unwritten samples"
        worst = max(worst, np.abs(xl[b] - xs[-1]).max() / np.abs(xs[-1]).max())
    return worst


CASES = [
    dict(),                                                           # cfg5's first launch in small
    dict(n_seg=1),
    dict(n_seg=3, L0=70000),                                          # segments that start inside the clip
    dict(precision="bf16x3", tol=3e-5),
    dict(reflect=False),                                              # zero padding: no edge tiles
    dict(hop0=64, K=(0, 192, 192, 192, 192), L0=30000, n_seg=3),      # the follow-up launch: hops 32 .. 4
    dict(K=(256, 256, 128, 64), L0=44100),                            # 8-step banks (VQT-like widths)
    dict(K=(192, 192), L0=20000, n_seg=2),                            # two levels
    dict(scale=1e-3),
    dict(scale=300.0),
    dict(n_bins=7),                                                   # a narrow last group of bins
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s=%s" % kv for kv in c.items()) or "default")
def test_stream_kernel_matches_the_float64_recursion(case):
    L0, hop0 = case.get("L0", 40000), case.get("hop0", 512)
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, L0)) * case.get("scale", 1.0)).astype(np.float32)
    err = _run_kernel(x, case.get("K", (192, 192, 192, 192)), hop0, case.get("n_seg", 2),
                      case.get("precision", "f16x3"), case.get("reflect", True), case.get("n_bins", 12))
    assert err < case.get("tol", 2e-6), err


def test_stream_kernel_rescales_for_a_loud_burst():
    """fp16 operands: a quiet clip with a burst 10^4 x louder in the middle (the scale is found from the
    chunks as they stream in: everything resident is rescaled when a louder chunk arrives) -- the result is
    as accurate relative to the clip's peak as without the burst."""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((2, 60000)) * 1e-2).astype(np.float32)
    x[0, 30000:30200] *= 1e4
    x[1, 100:300] *= 1e4  # (in the first chunks: before the first scale is settled)
    err = _run_kernel(x, (192, 192, 192, 192), 512, 2, "f16x3", True)
    assert err < 2e-6, err


@pytest.mark.parametrize("cls,kw,L", [
    ("CQT2010v2", dict(sr=44100, hop_length=512, n_bins=96), 132300),
    ("VQT", dict(sr=44100, hop_length=512, n_bins=96, gamma=10), 132300),
    ("CQT2010v2", dict(sr=22050, hop_length=256, n_bins=84, output_format="Complex"), 66000),
    ("CQT2010v2", dict(sr=44100, hop_length=512, n_bins=96, output_format="Phase"), 88200),
])
def test_modules_on_the_stream_kernel(cls, kw, L):
    """The same module on the streaming kernel, on the pyramid kernel and in fp32."""
    from nnaudio_amd import engine, features

    m = getattr(features, cls)(verbose=False, **kw).to(DEV)
    torch.manual_seed(0)
    x = torch.randn(3, L, device=DEV)
    x[1, 50000:50100] *= 200.0
    old = engine.octave_stream_enabled()
    try:
        with torch.no_grad():
            engine.set_octave_stream(True)
            a = m(x)
            engine.set_octave_stream(False)
            b = m(x)
            m.precision = "fp32"
            c = m(x)
    finally:
        engine.set_octave_stream(old)
    if kw.get("output_format") == "Phase":  # (cos, sin) of a near-silent bin is noise in every arithmetic
        mc = getattr(features, cls)(verbose=False, **dict(kw, output_format="Complex")).to(DEV)
        mc.precision = "fp32"
        with torch.no_grad():
            z = mc(x)
        loud = z.norm(dim=-1) > 1e-3 * z.norm(dim=-1).max()
        assert (a - c)[loud].abs().max() < 1e-3 and (a - b)[loud].abs().max() < 1e-3
        return
    peak = c.abs().max()
    assert (a - c).abs().max() <= 2e-6 * peak and (b - c).abs().max() <= 2e-6 * peak
    assert (a - b).abs().max() <= 1e-6 * peak


def test_stream_launches_are_bit_identical():
    from nnaudio_amd import features

    m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False).to(DEV)
    torch.manual_seed(1)
    x = torch.randn(8, 200000, device=DEV)
    with torch.no_grad():
        y0 = m(x).clone()
        for _ in range(10):
            assert torch.equal(m(x), y0)


@pytest.mark.parametrize("cls,kw,precision", [
    ("CQT2010v2", dict(sr=44100, hop_length=512, n_bins=96), None),             # streaming kernel: scale per chunk
    ("CQT1992v2", dict(sr=22050, hop_length=512, n_bins=60), "f16x3"),          # strip kernel: scale per clip
    ("STFT", dict(n_fft=1024, hop_length=256, freq_scale="linear", fmin=50, fmax=8000, sr=22050,
                  output_format="Magnitude"), "f16x3"),                         # dense f16x3 kernel: scale per clip
])
def test_an_infinite_sample_poisons_only_its_frames(cls, kw, precision):
    """fp16 operands are scaled by a power of two taken from the clip's largest FINITE |sample|: an Inf
    in the clip spoils the frames that contain it (as in the reference) and leaves the others -- and the
    other clips -- as they were."""
    from nnaudio_amd import features

    m = getattr(features, cls)(verbose=False, **kw).to(DEV)
    m.precision = precision
    torch.manual_seed(2)
    L = 220500
    x = torch.randn(2, L, device=DEV)
    x2 = x.clone()
    x2[0, 3000] = float("inf")
    with torch.no_grad():
        y0, y = m(x), m(x2)
    # the frames over the bad sample are lost (non-finite, or saturated where the split saturates)
    t_bad = 3000 // kw["hop_length"]
    lost = y[0, :, t_bad - 1:t_bad + 2]
    assert (not torch.isfinite(lost).all()) or float(lost.abs().max()) > 100 * float(y0.abs().max())
    T = y.shape[2]
    far = slice(T // 2, T)                                             # far beyond every kernel (and FIR chain)
    peak = y0.abs().max()
    assert torch.isfinite(y[0, :, far]).all() and (y[0, :, far] - y0[0, :, far]).abs().max() <= 2e-6 * peak
    assert (y[1] - y0[1]).abs().max() <= 2e-6 * peak
