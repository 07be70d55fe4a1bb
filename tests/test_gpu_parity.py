"""Parity of the HIP path (through the C ABI) with the reference / the pinned oracle.
Everything here needs the MI355X: run with ``pytest -m gpu``.

Tolerance (north_star "within 1e-4 rel fp32"): max|y-ref| <= 1e-4*max|ref| and
allclose(rtol=1e-4, atol=1e-4*max|ref|); phase compared where the bin carries energy."""
import ctypes
import os
import warnings

import numpy as np
import pytest
import torch
from scipy.signal import chirp

from tests import _golden
from tests._golden import (assert_parity, assert_phase_parity, build_module, check_ground_truth,
                           is_phase, oracle_forward)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    from nnaudio_amd import _abi

    _abi.load()  # fail loudly if the extension is not built


def run(mod, x, _method="forward", **fwd):
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fn = mod if _method == "forward" else getattr(mod, _method)
        y = fn(torch.as_tensor(x).to(DEV), **fwd)
    torch.cuda.synchronize()
    return y.cpu().numpy()


# ---------------------------------------------------------------------------------------
# every manifest case: HIP vs the reference's own output AND vs the oracle
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", _golden.case_names(forward_only=True))
def test_case_matches_reference_and_oracle(golden, name):
    case = golden.cases[name]
    x = golden.inputs[case["input"]]
    ref = golden.forward[name]
    mod = build_module(case, DEV)
    y = run(mod, x, _method=case.get("method", "forward"), **case["fwd"])
    assert y.dtype == ref.dtype and list(y.shape) == case["out_shape"]
    orc = oracle_forward(build_module(case), case, x)
    if _golden.is_inverse(case):
        # compare where the overlap-add is conditioned (see _golden.well_conditioned)
        y_r, _ = _golden.well_conditioned(case, mod, x, y, ref)
        y_o, _ = _golden.well_conditioned(case, mod, x, y, orc)
        assert_parity(y_r, ref, rel=1e-4, what=name + " vs reference")
        assert_parity(y_o, orc, rel=1e-4, what=name + " vs oracle")
        return
    if is_phase(case):
        mcase = dict(case, ctor=dict(case["ctor"], output_format="Magnitude"), fwd={})
        mag = oracle_forward(build_module(mcase), mcase, x)
        assert_phase_parity(y, ref, mag, what=name + " vs reference")
        assert_phase_parity(y, orc, mag, what=name + " vs oracle")
    else:
        assert_parity(y, ref, rel=1e-4, what=name + " vs reference")
        assert_parity(y, orc, rel=1e-4, what=name + " vs oracle")


# ---------------------------------------------------------------------------------------
# the reference's own ground-truth arrays (its tests/test_cqt.py:94-262)
# ---------------------------------------------------------------------------------------
def _chirp(method):
    s = np.linspace(0, 1, 44100)
    return chirp(s, 55, 1, 22050, method=method).astype(np.float32)[None, :]


GT_CASES = [
    (sweep, method, cls, tagc, fmt, tag)
    for sweep, method in (("log", "logarithmic"), ("linear", "linear"))
    for cls, tagc in (("CQT1992v2", "1992"), ("CQT2010v2", "2010"))
    for fmt, tag in (("Magnitude", "mag"), ("Complex", "complex"), ("Phase", "phase"))
    if not (tagc == "2010" and fmt == "Phase")
]


# What each arithmetic may miss of the reference's verbatim log(X + eps) assertion (fraction of
# elements; the conditioned ones -- above 1 % of the peak -- must all pass in every arithmetic),
# and from which energy its phase is compared.  The near-silent bins of a sweep carry 1e-9 of the
# peak; the fixture there records the rounding noise of the reference's own summation ORDER:
#   fp32   CQT1992v2 (round 5): the support-aware tile kernel accumulates ONE float32 FMA chain over the taps in
#          ascending order -- the reference's conv1d arithmetic, bit-identical to MIOpen's on this GPU -- and misses
#          NOTHING, like the reference (tests/test_reference_order.py runs the six assertions verbatim); the bar is the
#          reference's own, 1e-4 allowed.  CQT2010v2 (octave path, eps = 1e-2): no miss in fp32 and f16x3 either -- the
#          reference's four assertions pass verbatim on the default module (tests/test_reference_order.py)
#   f16x3  fp32-class operands (2e-6 of the peak at most, see test_cfg4_*), but CQT1992v2 runs on the
#          strip kernel, whose hop-periodic tap order leaves partial sums of ~0.3 x peak in silent
#          bins (aliases of the sweep): 2.7 % / 0.1 % miss on the MI355X (log / linear sweep; exact
#          operands in that order with fp32 accumulation: 1.3 %, scripts/split_emulation.py).  That
#          is why CQT1992v2's default precision stays fp32.  CQT2010v2 (octave path) meets the bar.
#   bf16x3 5e-6 of the peak is the size of those bins: 57 % / 74 %
# The measured fractions are pinned (with margin) so that they cannot grow unnoticed.
GT_MAX_MISS = {"fp32": {"1992": 1e-4, "2010": 1e-4}, "f16x3": {"1992": 0.04, "2010": 1e-4},
               "bf16x3": {"1992": 0.80, "2010": 0.80}}
GT_PHASE_FLOOR = {"fp32": 1e-3, "f16x3": 1e-3, "bf16x3": 1e-2}


@pytest.mark.parametrize("precision", ["fp32", "f16x3", "bf16x3"])
@pytest.mark.parametrize("sweep,method,cls,tagc,fmt,tag", GT_CASES)
def test_reference_ground_truths(golden, sweep, method, cls, tagc, fmt, tag, precision):
    """reference tests/test_cqt.py:94-262 (rtol = atol = 1e-3 on log(X + eps) / Complex / Phase) in
    every arithmetic the modules offer."""
    case = dict(cls=cls, ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24,
                                   output_format=fmt), fwd={})
    mod = build_module(case, DEV)
    mod.precision = precision
    y = run(mod, _chirp(method))
    gt = golden.ground_truth("%s-sweep-cqt-%s-%s-ground-truth.npy" % (sweep, tagc, tag))
    gtc = golden.ground_truth("%s-sweep-cqt-%s-complex-ground-truth.npy" % (sweep, tagc))
    miss = check_ground_truth(y, gt, fmt, 1e-5 if tagc == "1992" else 1e-2, gtc,
                              what="%s %s %s %s" % (sweep, cls, fmt, precision),
                              max_miss=GT_MAX_MISS[precision][tagc], phase_floor=GT_PHASE_FLOOR[precision])
    if miss is not None:
        print("ground truth %s %s %s: %.4f of the elements miss log(X + eps)" % (sweep, cls, precision, miss))


@pytest.mark.parametrize("sweep,method", [("log", "logarithmic"), ("linear", "linear")])
def test_reference_ground_truths_f16x3_natural_order(golden, sweep, method):
    """CQT1992v2 f16x3 with ``hop_periodic = False`` (staged dense kernel, taps in their natural order):
    0.58 % / 0.21 % of the log-magnitude elements miss the reference's tolerance on the MI355X (strip kernel:
    2.7 %; fp32 tile kernels 0) -- pinned with margin."""
    case = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24,
                                           output_format="Magnitude"), fwd={})
    mod = build_module(case, DEV)
    mod.precision, mod.hop_periodic = "f16x3", False
    y = run(mod, _chirp(method))
    gt = golden.ground_truth("%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep)
    gtc = golden.ground_truth("%s-sweep-cqt-1992-complex-ground-truth.npy" % sweep)
    miss = check_ground_truth(y, gt, "Magnitude", 1e-5, gtc, what="%s CQT1992v2 f16x3 natural order" % sweep,
                              max_miss=0.009, phase_floor=1e-3)
    print("ground truth %s CQT1992v2 f16x3 natural order: %.4f miss" % (sweep, miss))


def test_vqt_gamma0_is_bit_identical_to_cqt2010v2(golden):
    """reference tests/test_vqt.py:30-41 asserts exact equality."""
    from nnaudio_amd import features

    x = golden.inputs["x_1s22k"]
    a = run(features.CQT2010v2(sr=22050, verbose=False).to(DEV), x)
    b = run(features.VQT(sr=22050, gamma=0, verbose=False).to(DEV), x)
    assert (a == b).all()


# ---------------------------------------------------------------------------------------
# kernel-level: MFMA kernel vs the one-thread-per-output device kernel vs numpy, over every
# tile shape, epilogue, padding mode and ragged size
# ---------------------------------------------------------------------------------------
def _np_framed(x, wr, wi, hop, pad, mode, scale=None):
    from oracle import spectral_oracle as O

    xp = O.pad_signal(x, pad, {1: "constant", 2: "reflect"}[mode]) if mode else x
    re = O.correlate_strided(xp, wr, hop)
    im = -O.correlate_strided(xp, wi, hop) if wi is not None else None
    if scale is not None:
        re = re * scale[None, :, None]
        if im is not None:
            im = im * scale[None, :, None]
    return re, im


SHAPES = [
    # B, L, F, K, hop, pad, mode
    (3, 1000, 5, 64, 16, 32, 2),
    (2, 777, 16, 96, 17, 48, 1),
    (1, 4096, 70, 200, 50, 0, 0),
    (5, 300, 33, 130, 7, 65, 2),
    (2, 2500, 129, 256, 64, 128, 2),
    (1, 9000, 1, 33, 1, 16, 1),
]


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape", SHAPES)
def test_mfma_kernel_all_tiles(tile, shape):
    from nnaudio_amd import engine

    B, L, F, K, hop, pad, mode = shape
    rng = np.random.default_rng(B * 1000 + F)
    x = rng.standard_normal((B, L)).astype(np.float32)
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    sc = rng.uniform(0.5, 2.0, F).astype(np.float32)
    xd, wrd, wid, scd = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi, sc))
    kw = dict(hop=hop, pad=pad, pad_mode=mode, epilogue=engine.EPI_COMPLEX, row_scale=scd)
    y = engine.framed_gemm(xd, wrd, wid, tile=tile, **kw).cpu().numpy()
    yr = engine.framed_gemm(xd, wrd, wid, reference_kernel=True, **kw).cpu().numpy()
    re, im = _np_framed(x, wr, wi, hop, pad, mode, sc.astype(np.float64))
    want = np.stack((re, im), -1)
    assert_parity(yr, want, rel=2e-5, what="device reference kernel")
    assert_parity(y, want, rel=2e-5, what="mfma tile %d" % tile)
    # fp32 fma chains in different association: tiny
    assert np.abs(y - yr).max() <= 2e-5 * np.abs(want).max()


@pytest.mark.parametrize("epi", ["magnitude", "magnitude_eps", "power2", "power1", "power3.5",
                                 "phase", "cossin", "real"])
def test_epilogues(epi):
    from nnaudio_amd import engine

    rng = np.random.default_rng(7)
    B, L, F, K, hop, pad = 2, 3000, 40, 128, 32, 64
    x = rng.standard_normal((B, L)).astype(np.float32)
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    xd, wrd, wid = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi))
    re, im = _np_framed(x, wr, wi, hop, pad, 2)
    mag = np.sqrt(re * re + im * im)
    kw = dict(hop=hop, pad=pad, pad_mode=2)
    if epi == "magnitude":
        y = engine.framed_gemm(xd, wrd, wid, epilogue=engine.EPI_MAGNITUDE, **kw)
        assert_parity(y.cpu().numpy(), mag, what=epi)
    elif epi == "magnitude_eps":
        y = engine.framed_gemm(xd, wrd, wid, epilogue=engine.EPI_MAGNITUDE, eps=1e-8, **kw)
        assert_parity(y.cpu().numpy(), np.sqrt(mag ** 2 + 1e-8), what=epi)
    elif epi.startswith("power"):
        p = float(epi[5:])
        y = engine.framed_gemm(xd, wrd, wid, epilogue=engine.EPI_POWER, power=p, **kw)
        assert_parity(y.cpu().numpy(), mag ** p, what=epi)
    elif epi == "phase":
        y = engine.framed_gemm(xd, wrd, wid, epilogue=engine.EPI_PHASE_ATAN2, **kw)
        assert_phase_parity(y.cpu().numpy(), np.arctan2(im, re), mag, what=epi, tol=1e-4)
    elif epi == "cossin":
        y = engine.framed_gemm(xd, wrd, wid, epilogue=engine.EPI_PHASE_COSSIN, **kw)
        a = np.arctan2(im, re)
        assert_phase_parity(y.cpu().numpy(), np.stack((np.cos(a), np.sin(a)), -1), mag, what=epi,
                            tol=1e-4)
    else:
        y = engine.framed_gemm(xd, wrd, None, epilogue=engine.EPI_REAL, im_sign=1.0, **kw)
        assert_parity(y.cpu().numpy(), re, what=epi)


def test_phase_zero_input_matches_reference_convention():
    """atan2(-0 + 0.0, 0) = 0 and (cos, sin)(atan2(0, 0)) = (1, 0), as torch computes."""
    from nnaudio_amd import features

    x = torch.zeros(1, 2048, device=DEV)
    y = run(features.STFT(n_fft=256, hop_length=64, output_format="Phase", verbose=False).to(DEV), x)
    assert (y == 0).all()
    y = run(features.CQT1992v2(fmin=440, n_bins=12, output_format="Phase", verbose=False).to(DEV), x)
    assert (y[..., 0] == 1).all() and (y[..., 1] == 0).all()


def test_row_support_skipping_and_row_offset():
    from nnaudio_amd import engine

    rng = np.random.default_rng(11)
    B, L, F, K, hop = 2, 6000, 50, 1024, 64
    x = rng.standard_normal((B, L)).astype(np.float32)
    wr = np.zeros((F, K), np.float32)
    wi = np.zeros((F, K), np.float32)
    sup = np.zeros((F, 2), np.int32)
    for f in range(F):
        ln = max(3, int(900 / (1 + f)))
        s = (K - ln) // 2
        wr[f, s:s + ln] = rng.standard_normal(ln)
        wi[f, s:s + ln] = rng.standard_normal(ln)
        sup[f] = (s, s + ln)
    xd, wrd, wid, supd = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi, sup))
    re, im = _np_framed(x, wr, wi, hop, K // 2, 2)
    want = np.sqrt(re * re + im * im)
    T = want.shape[-1]
    for tile in (0, 1, 4, 5, 6):
        out = torch.full((B, F + 7, T), -1.0, device=DEV)
        engine.framed_gemm(xd, wrd, wid, hop=hop, pad=K // 2, pad_mode=2,
                           epilogue=engine.EPI_MAGNITUDE, row_support=supd, out=out,
                           out_rows_total=F + 7, out_row_offset=3, tile=tile)
        o = out.cpu().numpy()
        assert (o[:, :3] == -1).all() and (o[:, F + 3:] == -1).all()
        assert_parity(o[:, 3:F + 3], want, what="support tile %d" % tile)


@pytest.mark.parametrize("L,stride", [(1000, 2), (1001, 2), (4097, 2), (300, 2), (5000, 4), (20000, 8),
                                       (255, 2)])
def test_fir_decimate(L, stride):
    from nnaudio_amd import engine
    from nnaudio_amd.basis import lowpass_taps
    from oracle import spectral_oracle as O

    rng = np.random.default_rng(L)
    x = rng.standard_normal((3, L)).astype(np.float32)
    taps = lowpass_taps(1.0 / stride, 256, 0.03)
    y = engine.fir_decimate(torch.as_tensor(x).to(DEV), torch.as_tensor(taps).to(DEV), stride)
    want = O.fir_decimate(x, taps, stride)
    assert_parity(y.cpu().numpy(), want, rel=2e-5, what="fir L=%d s=%d" % (L, stride))


@pytest.mark.parametrize("M,F,T,B", [(128, 513, 216, 3), (40, 257, 33, 2), (64, 1025, 100, 1),
                                     (7, 30, 5, 4), (229, 1025, 77, 2)])
def test_filterbank(M, F, T, B):
    from nnaudio_amd import engine

    rng = np.random.default_rng(M)
    fb = rng.uniform(0, 1, (M, F)).astype(np.float32)
    sp = rng.uniform(0, 4, (B, F, T)).astype(np.float32)
    y = engine.filterbank(torch.as_tensor(fb).to(DEV), torch.as_tensor(sp).to(DEV))
    want = np.matmul(fb.astype(np.float64), sp.astype(np.float64))
    assert_parity(y.cpu().numpy(), want, rel=2e-5, what="filterbank")


# ---------------------------------------------------------------------------------------
# plumbing: input layouts, streams, errors
# ---------------------------------------------------------------------------------------
def test_input_layouts_and_streams(golden):
    from nnaudio_amd import features

    m = features.STFT(n_fft=512, hop_length=128, output_format="Magnitude", verbose=False).to(DEV)
    x = torch.as_tensor(golden.inputs["x_short"]).to(DEV)
    base = m(x)
    assert torch.equal(m(x[:, None, :]), base)
    assert torch.equal(m(x[1]), base[1:2])
    wide = torch.zeros(3, 8000, device=DEV)
    wide[:, ::2] = x
    assert torch.equal(m(wide[:, ::2]), base)  # non-unit inner stride
    padded = torch.zeros(3, 5000, device=DEV)
    padded[:, :4000] = x
    assert torch.equal(m(padded[:, :4000]), base)  # batch stride != length
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = m(x)
    s.synchronize()
    assert torch.equal(y, base)
    assert m.num_samples == 4000


def test_error_behaviour_on_device():
    from nnaudio_amd import features

    m = features.STFT(n_fft=512, hop_length=128, verbose=False).to(DEV)
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 100, device=DEV))
    with pytest.raises(RuntimeError):  # reflect needs pad < L (nn.ReflectionPad1d)
        m(torch.zeros(1, 256, device=DEV))
    with pytest.raises(RuntimeError):  # kernel longer than the signal without centring
        features.STFT(n_fft=512, hop_length=128, center=False, verbose=False).to(DEV)(
            torch.zeros(1, 300, device=DEV))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4000, device=DEV, dtype=torch.float64))
    c = features.CQT1992v2(fmin=220, n_bins=24, verbose=False).to(DEV)
    with pytest.raises(ValueError):
        c(torch.zeros(1, 22050, device=DEV), normalization_type="nope")
    with pytest.raises(RuntimeError):  # octave frame counts diverge (torch.cat fails upstream)
        features.CQT2010v2(hop_length=100, earlydownsample=False, verbose=False).to(DEV)(
            torch.zeros(1, 22050, device=DEV))
    assert m(torch.zeros(1, 4000, device=DEV), output_format="nonsense") is None


def test_reflect_fallback_warns_like_reference():
    """utils.py:505-517: when mirroring is impossible an octave is zero-padded with a warning."""
    from nnaudio_amd import features

    m = features.CQT2010v2(sr=22050, fmin=55, n_bins=72, verbose=False).to(DEV)
    x = torch.randn(1, 3000, device=DEV)
    with pytest.warns(UserWarning):
        y = m(x)
    assert torch.isfinite(y).all()


# ---------------------------------------------------------------------------------------
# BASELINE.json full-size configurations: sampled exact check + size-independent properties
# ---------------------------------------------------------------------------------------
def _sample_cols(rng, B, T, n):
    return rng.integers(0, B, n), rng.integers(0, T, n)


def test_cfg2_stft_full_size_sampled_and_linear():
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    B, L = 64, 441000
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, L, generator=g)
    m = features.STFT(n_fft=2048, hop_length=512, window="hann", output_format="Complex",
                      verbose=False).to(DEV)
    xd = x.to(DEV)
    y = m(xd)
    assert tuple(y.shape) == (64, 1025, 862, 2)
    rng = np.random.default_rng(0)
    cb, ct = _sample_cols(rng, B, 862, 96)
    ct[:8] = [0, 1, 2, 3, 858, 859, 860, 861]  # frames that touch the reflected edges
    re, im = O.sampled_complex(x.numpy(), m.wcos.cpu().numpy(), m.wsin.cpu().numpy(), cb, ct,
                               512, 1024, "reflect")
    got = y[torch.as_tensor(cb), :, torch.as_tensor(ct)].cpu().numpy()  # (n, F, 2)
    peak = max(np.abs(re).max(), np.abs(im).max())
    assert np.abs(got[..., 0] - re).max() <= 1e-4 * peak
    assert np.abs(got[..., 1] - im).max() <= 1e-4 * peak
    # magnitude path agrees with |complex|
    mag = m(xd, output_format="Magnitude")
    assert torch.allclose(mag, torch.sqrt(y[..., 0] ** 2 + y[..., 1] ** 2), rtol=1e-5, atol=1e-4)
    # linearity in the waveform
    x2 = torch.randn(B, L, generator=g).to(DEV)
    lhs = m(0.5 * xd - 2.0 * x2)
    rhs = 0.5 * y - 2.0 * m(x2)
    assert (lhs - rhs).abs().max().item() <= 1e-4 * rhs.abs().max().item()


def test_cfg3_mel_full_size_sampled():
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    B, L = 256, 110250
    x = torch.randn(B, L, generator=torch.Generator().manual_seed(1))
    m = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, verbose=False).to(DEV)
    y = m(x.to(DEV))
    assert tuple(y.shape) == (256, 128, 216)
    rng = np.random.default_rng(1)
    cb, ct = _sample_cols(rng, B, 216, 64)
    ct[:4] = [0, 1, 214, 215]
    re, im = O.sampled_complex(x.numpy(), m.stft.wcos.cpu().numpy(), m.stft.wsin.cpu().numpy(),
                               cb, ct, 512, 512, "reflect")
    want = (re * re + im * im) @ m.mel_basis.cpu().numpy().astype(np.float64).T  # (n, M)
    got = y[torch.as_tensor(cb), :, torch.as_tensor(ct)].cpu().numpy()
    assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()


@pytest.mark.parametrize("precision,B", [("fp32", 16), ("bf16x3", 16), ("bf16x3", 64), ("f16x3", 16), ("f16x3", 64)])
def test_cfg4_cqt1992v2_full_size_sampled(precision, B):
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    # B = 16: one rank's shard of cfg4 (128 clips over 8 GPUs); B = 64: the bench's CQT84 batch,
    # where the frame tiles alone fill the chip and every workgroup walks all its row tiles
    L = 441000
    x = torch.randn(B, L, generator=torch.Generator().manual_seed(2))
    m = features.CQT1992v2(sr=44100, hop_length=512, fmin=32.70, n_bins=84, bins_per_octave=12,
                           output_format="Complex", verbose=False).to(DEV)
    m.precision = precision  # bf16x3: the hop-periodic narrow-tile kernel, 64 super-stages
    y = m(x.to(DEV))
    assert tuple(y.shape) == (B, 84, 862, 2)
    rng = np.random.default_rng(2)
    cb, ct = _sample_cols(rng, B, 862, 24)
    ct[:6] = [0, 1, 30, 840, 860, 861]
    re, im = O.sampled_complex(x.numpy(), m.cqt_kernels_real.cpu().numpy(),
                               m.cqt_kernels_imag.cpu().numpy(), cb, ct, 512, 16384, "reflect")
    s = np.sqrt(m.lenghts.cpu().numpy().astype(np.float64))[None, :]
    got = y[torch.as_tensor(cb), :, torch.as_tensor(ct)].cpu().numpy()
    peak = max(np.abs(re * s).max(), np.abs(im * s).max())
    tol = 2e-6 if precision == "f16x3" else 1e-4  # f16x3: fp32 class
    assert np.abs(got[..., 0] - re * s).max() <= tol * peak
    assert np.abs(got[..., 1] - im * s).max() <= tol * peak


def _octave_banks(mod):
    sd = {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}
    if type(mod).__name__ == "VQT":
        banks = [(sd["cqt_kernels_real_%d" % i], sd["cqt_kernels_imag_%d" % i])
                 for i in range(mod.n_octaves)]
    else:
        banks = [(sd["cqt_kernels_real"], sd["cqt_kernels_imag"])] * mod.n_octaves
    return banks, sd


def _cfg5_sampled_check(mod, x, y, rng, what, tol=1e-4):
    """Sampled frames of a (B, 96, T, 2) CQT2010v2 / VQT output against the float64 evaluation of
    the octave recursion on just the samples they depend on (oracle.sampled_octave_complex,
    pinned to the whole-signal oracle by tests/test_oracle_golden.py): the first / last 4 frames
    (reflected edges of every octave) + 24 random ones, every bin = every octave row block."""
    from oracle import spectral_oracle as O

    B, _, T, _ = y.shape
    cb, ct = _sample_cols(rng, B, T, 32)
    ct[:8] = [0, 1, 2, 3, T - 4, T - 3, T - 2, T - 1]
    cb[:8] = [0, B - 1, 1, B - 2, 0, B - 1, 2, B - 3]
    banks, sd = _octave_banks(mod)
    re, im = O.sampled_octave_complex(x.numpy(), banks, sd["lenghts"], mod.hop_length, mod.n_bins,
                                      sd["lowpass_filter"], cb, ct, pad_mode="reflect")
    got = y[torch.as_tensor(cb), :, torch.as_tensor(ct)].cpu().numpy()  # (n, bins, 2)
    peak = max(np.abs(re).max(), np.abs(im).max())
    err = max(np.abs(got[..., 0] - re).max(), np.abs(got[..., 1] - im).max())
    assert err <= tol * peak, "%s: %.3e of peak" % (what, err / peak)
    # every octave's row block individually (a wrong octave must not hide behind the loudest one)
    nf = banks[0][0].shape[0]
    for o in range(0, mod.n_bins, nf):
        pk = max(np.abs(re[:, o:o + nf]).max(), np.abs(im[:, o:o + nf]).max())
        e = max(np.abs(got[:, o:o + nf, 0] - re[:, o:o + nf]).max(),
                np.abs(got[:, o:o + nf, 1] - im[:, o:o + nf]).max())
        assert e <= 2 * tol * pk, "%s rows %d..: %.3e of the block's peak" % (what, o, e / pk)


def test_cfg5_cqt2010v2_vqt_full_length():
    """One rank's shard of cfg5 (B = 64 x 30 s, 96 bins, 8 octaves): CQT2010v2, VQT(gamma=0) and
    VQT(gamma=10) against sampled float64 values of the 7-deep 256-tap recursion; VQT(gamma=0)
    == CQT2010v2 bit-exactly; the recursion is linear in the waveform."""
    from nnaudio_amd import features

    B, L = 64, 1323000
    x = torch.randn(B, L, generator=torch.Generator().manual_seed(3))
    xd = x.to(DEV)
    kw = dict(sr=44100, hop_length=512, n_bins=96, output_format="Complex", verbose=False)
    c = features.CQT2010v2(**kw).to(DEV)
    v0 = features.VQT(gamma=0, **kw).to(DEV)
    v10 = features.VQT(gamma=10, **kw).to(DEV)
    with torch.no_grad():
        yc = c(xd)
        assert tuple(yc.shape) == (64, 96, 2584, 2)
        _cfg5_sampled_check(c, x, yc, np.random.default_rng(3), "CQT2010v2 cfg5")
        yv = v0(xd)
        assert torch.equal(yc, yv)
        del yv
        y10 = v10(xd)
        _cfg5_sampled_check(v10, x, y10, np.random.default_rng(4), "VQT gamma=10 cfg5")
        del y10
        x2 = torch.randn(4, L, generator=torch.Generator().manual_seed(4)).to(DEV)
        lhs = c(xd[:4] - 3.0 * x2)
        rhs = yc[:4] - 3.0 * c(x2)
        assert (lhs - rhs).abs().max().item() <= 1e-4 * rhs.abs().max().item()


@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_cfg5_fused_octave_kernel(precision):
    """cfg5 shard on the fused octave kernel (decimated signals resident in LDS, split operands on the
    matrix pipe, all eight octaves on the chain of launches), in both split arithmetics, against the
    sampled float64 recursion, for CQT2010v2 and VQT(gamma=10); VQT(gamma=0) equals CQT2010v2 bit
    for bit here too; Magnitude agrees with |Complex|.  f16x3 carries 22 operand bits through the
    seven decimations (bf16x3: 16): fp32-class against the float64 values."""
    from nnaudio_amd import features

    B, L = 64, 1323000
    x = torch.randn(B, L, generator=torch.Generator().manual_seed(6))
    xd = x.to(DEV)
    kw = dict(sr=44100, hop_length=512, n_bins=96, output_format="Complex", verbose=False)
    c = features.CQT2010v2(**kw).to(DEV)
    v0 = features.VQT(gamma=0, **kw).to(DEV)
    v10 = features.VQT(gamma=10, **kw).to(DEV)
    for m in (c, v0, v10):
        m.precision = precision
    tol = 1e-4 if precision == "bf16x3" else 5e-6
    with torch.no_grad():
        yc = c(xd)
        assert tuple(yc.shape) == (64, 96, 2584, 2)
        _cfg5_sampled_check(c, x, yc, np.random.default_rng(6), "CQT2010v2 cfg5 " + precision, tol=tol)
        c.precision = "fp32"
        y32 = c(xd)
        assert not torch.equal(yc, y32)  # the fused kernel really ran
        assert (yc - y32).abs().max().item() <= (5e-5 if precision == "bf16x3" else 5e-6) * y32.abs().max().item()
        del y32
        c.precision = precision
        mag = c(xd, output_format="Magnitude")
        assert torch.allclose(mag, torch.sqrt(yc[..., 0] ** 2 + yc[..., 1] ** 2), rtol=1e-5, atol=1e-4)
        del mag
        yv = v0(xd)
        assert torch.equal(yc, yv)
        del yv, yc
        y10 = v10(xd)
        _cfg5_sampled_check(v10, x, y10, np.random.default_rng(7), "VQT gamma=10 cfg5 " + precision, tol=tol)
        # the scale follows the clip: any signal level must do (f16x3: fp16 has 5 exponent bits)
        if precision == "f16x3":
            g = torch.tensor([1e-6, 1.0, 3e4, 1e-3], device=DEV)
            ys = c(xd[:4] * g[:, None])
            want = c(xd[:4]) * g[:, None, None, None]
            assert (ys - want).abs().max(-1)[0].max(-1)[0].max(-1)[0].div(want.abs().amax((1, 2, 3))).max().item() <= 1e-5


def test_cfg3_mel_full_size_sampled_bf16x3():
    """cfg3 in the benched arithmetic: split-bf16 contraction with the mel reduction fused into
    its epilogue, sampled against float64."""
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    B, L = 256, 110250
    x = torch.randn(B, L, generator=torch.Generator().manual_seed(1))
    m = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, verbose=False).to(DEV)
    m.stft.precision = "bf16x3"
    with torch.no_grad():
        y = m(x.to(DEV))
    assert tuple(y.shape) == (256, 128, 216)
    rng = np.random.default_rng(11)
    cb, ct = _sample_cols(rng, B, 216, 64)
    ct[:4] = [0, 1, 214, 215]
    re, im = O.sampled_complex(x.numpy(), m.stft.wcos.cpu().numpy(), m.stft.wsin.cpu().numpy(),
                               cb, ct, 512, 512, "reflect")
    want = (re * re + im * im) @ m.mel_basis.cpu().numpy().astype(np.float64).T  # (n, M)
    got = y[torch.as_tensor(cb), :, torch.as_tensor(ct)].cpu().numpy()
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err <= 1e-4, err
    m.stft.precision = "fp32"
    with torch.no_grad():
        y32 = m(x.to(DEV))
    assert not torch.equal(y, y32)  # the bf16 pipe really ran
    assert (y - y32).abs().max().item() <= 5e-5 * y32.abs().max().item()


def test_cfg2_stft_magnitude_bf16x3_sampled():
    """The exact step bench.py times (cfg2, Magnitude epilogue, bf16x3) against
    sqrt(re^2 + im^2) of the sampled float64 values."""
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    B, L = 64, 441000
    x = torch.randn(B, L, generator=torch.Generator().manual_seed(5))
    m = features.STFT(n_fft=2048, hop_length=512, window="hann", output_format="Magnitude",
                      verbose=False).to(DEV)
    m.precision = "bf16x3"
    with torch.no_grad():
        y = m(x.to(DEV))
    assert tuple(y.shape) == (64, 1025, 862)
    rng = np.random.default_rng(5)
    cb, ct = _sample_cols(rng, B, 862, 96)
    ct[:8] = [0, 1, 2, 3, 858, 859, 860, 861]
    re, im = O.sampled_complex(x.numpy(), m.wcos.cpu().numpy(), m.wsin.cpu().numpy(), cb, ct,
                               512, 1024, "reflect")
    want = np.sqrt(re * re + im * im)
    got = y[torch.as_tensor(cb), :, torch.as_tensor(ct)].cpu().numpy()
    err = np.abs(got - want).max() / want.max()
    assert err <= 1e-4, err
    assert err <= 2e-5, err  # the split's own budget


# ---------------------------------------------------------------------------------------
# precision="bf16x3": split-bf16 operands on the bf16 MFMA (include/mispec.h MISPEC_PREC_BF16X3)
# Same parity bar as the fp32 path; measured error is ~5e-6 of the peak.
# ---------------------------------------------------------------------------------------
@pytest.fixture
def bf16x3():
    import nnaudio_amd

    old = nnaudio_amd.get_precision()
    nnaudio_amd.set_precision("bf16x3")
    yield
    nnaudio_amd.set_precision(old)


def _np_bf16_split(v):
    """numpy restatement of bf16_split (framed_bf16x3.inl): round-to-nearest-even hi, then lo."""
    def rne(a):
        u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
        return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)

    hi = rne(v)
    hf = (hi.astype(np.uint32) << 16).view(np.float32)
    lo = rne(v.astype(np.float32) - hf)
    return hi, lo


@pytest.mark.parametrize("F,K,has_im", [(5, 70, True), (33, 256, True), (12, 31, False)])
def test_split_basis_is_bit_exact(F, K, has_im):
    from nnaudio_amd import engine

    rng = np.random.default_rng(F * 1000 + K)
    re = (rng.standard_normal((F, K)) * 10.0 ** rng.integers(-6, 3, (F, 1))).astype(np.float32)
    im = rng.standard_normal((F, K)).astype(np.float32) if has_im else None
    re[0, :3] = [0.0, -0.0, 1.0]
    got = engine.split_basis(torch.as_tensor(re).to(DEV),
                             torch.as_tensor(im).to(DEV) if has_im else None)
    torch.cuda.synchronize()
    Ks = (K + 31) // 32 * 32
    raw = got.cpu().numpy().view(np.uint16)
    n_planes = 4 if has_im else 2
    planes = raw[:n_planes * F * Ks].reshape(n_planes, F, Ks)
    if has_im:
        # complex banks carry a second copy in the strip kernel's fragment order
        # [16-bin tile][16-tap step][hi | lo][lane = row + 32 * (tap / 8 % 2)][8 taps], row = 2 * bin +
        # component, zero rows past the last bin, and a 4 KB block of zeros behind the last tile
        M = (F + 15) // 16
        frag = raw[n_planes * F * Ks:]
        assert frag.size == M * (Ks // 16) * 1024 + 2048 and not frag[M * (Ks // 16) * 1024:].any()
        frag = frag[:M * (Ks // 16) * 1024].reshape(M, Ks // 16, 2, 2, 32, 8)  # tile, step, hi/lo, lh, row, tap
        want = np.zeros_like(frag)
        for z in range(2):
            for hl in range(2):
                src = np.zeros((M * 16, Ks), np.uint16)
                src[:F] = planes[2 * z + hl]
                # (bin, tap) -> (tile, bin in tile), (step, half, tap in lane)
                v = src.reshape(M, 16, Ks // 16, 2, 8).transpose(0, 2, 3, 1, 4)  # tile, step, lh, bin, tap
                want[:, :, hl, :, z::2, :] = v
        assert np.array_equal(frag, want)
    for z, src in enumerate([re, im] if has_im else [re]):
        hi, lo = _np_bf16_split(src)
        assert np.array_equal(planes[2 * z][:, :K], hi)
        assert np.array_equal(planes[2 * z + 1][:, :K], lo)
        assert not planes[2 * z][:, K:].any() and not planes[2 * z + 1][:, K:].any()
        # hi + lo carries >= 16 significant bits of the fp32 value
        back = ((hi.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
                + (lo.astype(np.uint32) << 16).view(np.float32).astype(np.float64))
        assert np.abs(back - src).max() <= 2.0 ** -16 * np.abs(src).max()


@pytest.mark.parametrize("name", [n for n in _golden.case_names(forward_only=True)
                                  if n.split("_")[0] in ("stft", "mel", "gamma", "cqt1992v2", "mfcc",
                                                         "cqt2010v2", "vqt")])
def test_case_bf16x3_matches_reference_and_oracle(golden, bf16x3, name):
    _case_in_the_process_wide_precision(golden, name)


@pytest.fixture
def f16x3():
    import nnaudio_amd

    old = nnaudio_amd.get_precision()
    nnaudio_amd.set_precision("f16x3")
    yield
    nnaudio_amd.set_precision(old)


@pytest.mark.parametrize("name", [n for n in _golden.case_names(forward_only=True)
                                  if n.split("_")[0] in ("stft", "mel", "gamma", "cqt1992v2", "mfcc",
                                                         "cqt2010v2", "vqt")])
def test_case_f16x3_matches_reference_and_oracle(golden, f16x3, name):
    _case_in_the_process_wide_precision(golden, name)


def _case_in_the_process_wide_precision(golden, name):
    case = golden.cases[name]
    x = golden.inputs[case["input"]]
    ref = golden.forward[name]
    y = run(build_module(case, DEV), x, **case["fwd"])
    assert y.dtype == np.float32 and list(y.shape) == case["out_shape"]
    orc = oracle_forward(build_module(case), case, x)
    if is_phase(case):
        mcase = dict(case, ctor=dict(case["ctor"], output_format="Magnitude"), fwd={})
        mag = oracle_forward(build_module(mcase), mcase, x)
        assert_phase_parity(y, ref, mag, what=name + " vs reference")
        assert_phase_parity(y, orc, mag, what=name + " vs oracle")
    else:
        assert_parity(y, ref, rel=1e-4, what=name + " vs reference")
        assert_parity(y, orc, rel=1e-4, what=name + " vs oracle")


@pytest.mark.parametrize("shape", [  # (B, L, bins, K, hop, pad, mode): 256-row blocks, edges, tails
    (3, 9000, 128, 512, 128, 256, 2),    # exactly one full row block
    (2, 7001, 257, 512, 64, 256, 1),     # 2 blocks + Nyquist bin on the fp32 kernel
    (1, 5000, 200, 300, 50, 150, 2),     # K % 32 != 0, partial second block on bf16
    (2, 4096, 160, 1024, 256, 0, 0),     # center=False
    (5, 1500, 130, 1024, 32, 512, 2),    # short clips: every frame touches the padding
    (2, 6000, 140, 512, 63, 256, 2),     # odd hop: not covered, must run (exactly) in fp32
    # n_frames >= 256 and hop % 32 == 0: with supports these take the hop-periodic (slab) kernel
    (3, 40000, 90, 1024, 128, 512, 2),   # 313 frames per clip: tiles straddle clip boundaries
    (2, 33000, 100, 2048, 64, 1024, 1),  # C = 32 super-stages, zero padding
    (1, 70000, 70, 300, 96, 150, 2),     # K % 32 != 0, hop = 3 sub-stages, C = 4
    (2, 20000, 84, 512, 64, 0, 0),       # center=False
])
@pytest.mark.parametrize("support", [False, True, "narrow", "single-buffer", "strip", "strip-narrow"])
def test_bf16x3_kernel_shapes(shape, support):
    from nnaudio_amd import engine

    B, L, F, K, hop, pad, mode = shape
    rng = np.random.default_rng(B * 100000 + L)
    x = rng.standard_normal((B, L)).astype(np.float32)
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    sup = None
    # one slab buffer: an A/B knob of the benchmarking build (the product library always has
    # room for two), kept under test because kbench measures with it
    dbg = 0x8000 if support == "single-buffer" else 0
    if dbg:
        from nnaudio_amd import _abi

        if not os.path.exists(_abi.ABLATE_LIB_PATH):
            pytest.skip("libmispec_ablate.so not built (python -m nnaudio_amd.build --ablate)")
    if support:  # centred supports that shrink with the row index, like a CQT bank
        widest = min(K // 2, 40) if str(support).endswith("narrow") else K // 2  # narrow: < one hop of taps
        half = np.linspace(widest, 8, F).astype(np.int64)
        lo, hi = K // 2 - half, K // 2 + half
        keep = (np.arange(K)[None, :] >= lo[:, None]) & (np.arange(K)[None, :] < hi[:, None])
        wr, wi = (wr * keep).astype(np.float32), (wi * keep).astype(np.float32)
        sup = torch.as_tensor(np.stack([lo, hi], 1).astype(np.int32)).to(DEV)
        if str(support).startswith("strip"):
            # with a host copy of the supports the library may plan the strip kernel (whenever the
            # shape allows: hop % 32 == 0, >= 64 frames per clip, kernels of >= 2 hops)
            sup.host_copy = np.ascontiguousarray(np.stack([lo, hi], 1).astype(np.int32))
    re, im = _np_framed(x, wr, wi, hop, pad, mode)
    ref = np.stack((re, im), -1)
    kw = dict(hop=hop, pad=pad, pad_mode=mode, epilogue=engine.EPI_COMPLEX, row_support=sup)
    xd, wrd, wid = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi))
    y = engine.framed_gemm(xd, wrd, wid, precision="bf16x3", _debug=dbg, **kw)
    y32 = engine.framed_gemm(xd, wrd, wid, precision="fp32", **kw)
    torch.cuda.synchronize()
    assert_parity(y.cpu().numpy(), ref, rel=1e-4, what="bf16x3 %s" % (shape,))
    if str(support).startswith("strip"):
        a, _out, _dev, _keep = engine._framed_args(xd, wrd, wid, precision="bf16x3", **kw)
        from nnaudio_amd import _abi

        n_pass = _abi.load().mispec_strip_plan(ctypes.byref(a), 256, None, 0)
        T = (L + 2 * pad - K) // hop + 1
        assert (n_pass > 0) == (hop % 32 == 0 and T >= 64 and -(-K // 32) * 32 >= 2 * hop)
    if hop % 2:
        assert torch.equal(y, y32)  # fell back to the fp32 kernel
    else:
        assert not torch.equal(y, y32)  # really took the bf16 pipe


@pytest.mark.parametrize("shape", [  # (B, L, bins, K, hop, pad, mode)
    (3, 50000, 84, 4096, 256, 2048, 2),   # 196 frames per clip: frame tiles straddle clips, 16 hops
    (2, 70001, 70, 2048, 128, 1024, 1),   # last row tile holds 6 bins; 547 frames
    (7, 17000, 100, 1024, 96, 0, 0),      # hop of 3 sub-stages, center=False, 167 frames
])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_strip_kernel_epilogues(shape, precision):
    """Every epilogue through the strip kernel (framed_bf16x3_strip.inl: banks with supports and a
    host copy of them) against the one-thread-per-output device kernel: partial sums of the waves
    that share a row tile, the register and the LDS path of the shared epilogue, row scales, row
    offsets into a taller output."""
    from nnaudio_amd import _abi, engine

    B, L, F, K, hop, pad, mode = shape
    rng = np.random.default_rng(L)
    xd = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).to(DEV)
    half = np.geomspace(K // 2 - 3, 24, F).astype(np.int64)  # a CQT-like bank: lengths fall 2x per ~20 rows
    lo, hi = K // 2 - half, K // 2 + half + (np.arange(F) % 2)
    keep = (np.arange(K)[None, :] >= lo[:, None]) & (np.arange(K)[None, :] < hi[:, None])
    wr = torch.as_tensor((rng.standard_normal((F, K)) * keep).astype(np.float32)).to(DEV)
    wi = torch.as_tensor((rng.standard_normal((F, K)) * keep).astype(np.float32)).to(DEV)
    sup = torch.as_tensor(np.stack([lo, hi], 1).astype(np.int32)).to(DEV)
    sup.host_copy = np.ascontiguousarray(np.stack([lo, hi], 1).astype(np.int32))
    sc = torch.as_tensor(rng.uniform(0.5, 2.0, F).astype(np.float32)).to(DEV)
    base = dict(hop=hop, pad=pad, pad_mode=mode)
    split = {"basis_split": engine.frag_basis_f16(wr, wi)} if precision == "f16x3" else {}
    a, _o, _d, _k = engine._framed_args(xd, wr, wi, precision=precision, row_support=sup,
                                        epilogue=engine.EPI_MAGNITUDE, **base, **split)
    assert _abi.load().mispec_strip_plan(ctypes.byref(a), 256, None, 0) > 0
    z = engine.framed_gemm(xd, wr, wi, reference_kernel=True, epilogue=engine.EPI_COMPLEX, row_scale=sc,
                           **base)
    mag = torch.sqrt(z[..., 0] ** 2 + z[..., 1] ** 2)
    strong = mag > 0.05 * mag.max()
    cases = [(engine.EPI_COMPLEX, {}), (engine.EPI_MAGNITUDE, {}), (engine.EPI_MAGNITUDE, {"eps": 1e-8}),
             (engine.EPI_POWER, {"power": 2.0}), (engine.EPI_POWER, {"power": 0.7, "eps": 1e-8}),
             (engine.EPI_PHASE_ATAN2, {}), (engine.EPI_PHASE_COSSIN, {"im_sign": 1.0})]
    for epi, extra in cases:
        kw = dict(base, epilogue=epi, row_scale=sc, **extra)
        ref = engine.framed_gemm(xd, wr, wi, reference_kernel=True, **kw)
        y = engine.framed_gemm(xd, wr, wi, precision=precision, row_support=sup, **kw, **split)
        assert y.shape == ref.shape
        what = "strip epilogue %d %s %s" % (epi, extra, shape)
        if epi == engine.EPI_PHASE_ATAN2:
            d = torch.remainder(y - ref + np.pi, 2 * np.pi) - np.pi
            assert d[strong].abs().max().item() < 2e-3, what
        elif epi == engine.EPI_PHASE_COSSIN:
            assert (y - ref)[strong].abs().max().item() < 2e-3, what
        else:
            assert (y - ref).abs().max().item() <= (1e-4 if precision == "bf16x3" else 1e-5) * ref.abs().max().item(), what
    # a row block of a taller output (octave assembly): rows outside it stay untouched
    T = z.shape[2]
    out = torch.full((B, F + 9, T), -7.0, device=DEV)
    engine.framed_gemm(xd, wr, wi, precision=precision, row_support=sup, epilogue=engine.EPI_MAGNITUDE,
                       row_scale=sc, out=out, out_rows_total=F + 9, out_row_offset=5, **base, **split)
    ref = engine.framed_gemm(xd, wr, wi, reference_kernel=True, epilogue=engine.EPI_MAGNITUDE, row_scale=sc,
                             **base)
    assert (out[:, 5:5 + F] - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert bool((out[:, :5] == -7.0).all()) and bool((out[:, 5 + F:] == -7.0).all())


@pytest.mark.parametrize("shape", [  # (B, L, bins, K, hop, pad, mode)
    (37, 20, 140, 256, 64, 128, 1),     # one frame per clip: every store quad spans four clips
    (19, 150, 129, 256, 64, 128, 2),    # three frames per clip + a Nyquist-like leftover bin
    (11, 300, 160, 256, 64, 128, 2),    # five frames per clip
    (3, 16448, 130, 256, 64, 128, 2),   # 258 frames per clip: partial second frame tile
])
def test_bf16x3_epilogues_small_clips(shape):
    """Every epilogue of the dense bf16x3 kernel (16-byte row-segment stores through the LDS
    transpose, their clip-straddling slow path, the phase patches, real bases, row scales)
    against the one-thread-per-output device kernel, on clips of very few frames."""
    from nnaudio_amd import engine

    B, L, F, K, hop, pad, mode = shape
    rng = np.random.default_rng(L)
    xd = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).to(DEV)
    wr = torch.as_tensor(rng.standard_normal((F, K)).astype(np.float32)).to(DEV)
    wi = torch.as_tensor(rng.standard_normal((F, K)).astype(np.float32)).to(DEV)
    sc = torch.as_tensor(rng.uniform(0.5, 2.0, F).astype(np.float32)).to(DEV)
    base = dict(hop=hop, pad=pad, pad_mode=mode)
    z = engine.framed_gemm(xd, wr, wi, reference_kernel=True, epilogue=engine.EPI_COMPLEX, row_scale=sc,
                           **base)
    mag = torch.sqrt(z[..., 0] ** 2 + z[..., 1] ** 2)
    strong = mag > 0.05 * mag.max()
    cases = [(engine.EPI_COMPLEX, {}), (engine.EPI_MAGNITUDE, {}), (engine.EPI_MAGNITUDE, {"eps": 1e-8}),
             (engine.EPI_POWER, {"power": 2.0}), (engine.EPI_POWER, {"power": 1.0}),
             (engine.EPI_POWER, {"power": 0.7, "eps": 1e-8}), (engine.EPI_PHASE_ATAN2, {}),
             (engine.EPI_PHASE_COSSIN, {"im_sign": 1.0})]
    for epi, extra in cases:
        kw = dict(base, epilogue=epi, row_scale=sc, **extra)
        ref = engine.framed_gemm(xd, wr, wi, reference_kernel=True, **kw)
        y = engine.framed_gemm(xd, wr, wi, precision="bf16x3", **kw)
        assert y.shape == ref.shape
        what = "epilogue %d %s %s" % (epi, extra, shape)
        if epi == engine.EPI_PHASE_ATAN2:
            d = torch.remainder(y - ref + np.pi, 2 * np.pi) - np.pi
            assert d[strong].abs().max().item() < 2e-3, what
        elif epi == engine.EPI_PHASE_COSSIN:
            assert (y - ref)[strong].abs().max().item() < 2e-3, what
        else:
            assert (y - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), what
    # real basis (two independent row tiles per wave in the planar layout)
    wreal = torch.as_tensor(rng.standard_normal((2 * F, K)).astype(np.float32)).to(DEV)
    kw = dict(base, epilogue=engine.EPI_REAL, im_sign=1.0)
    ref = engine.framed_gemm(xd, wreal, None, reference_kernel=True, **kw)
    y = engine.framed_gemm(xd, wreal, None, precision="bf16x3", **kw)
    assert (y - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    assert not torch.equal(y, engine.framed_gemm(xd, wreal, None, precision="fp32", **kw))


def _fourier_like_basis(rng, F, K, tap0):
    """Random basis with the symmetry of the reference's Fourier kernels (utils.py:379-389 times a
    centred window): re even / im odd about tap K/2; ``tap0``: non-zero coefficients at tap 0."""
    H = K // 2
    wr = np.zeros((F, K), np.float32)
    wi = np.zeros((F, K), np.float32)
    a = rng.standard_normal((F, H - 1)).astype(np.float32)
    b = rng.standard_normal((F, H - 1)).astype(np.float32)
    wr[:, 1:H], wr[:, K - 1:H:-1] = a, a
    wi[:, 1:H], wi[:, K - 1:H:-1] = b, -b
    wr[:, H] = rng.standard_normal(F)
    wi[:, H] = rng.standard_normal(F) if tap0 else 0.0
    if tap0:
        wr[:, 0] = rng.standard_normal(F)
        wi[:, 0] = rng.standard_normal(F)
    return wr, wi


@pytest.mark.parametrize("shape", [  # (B, L, bins, K, hop, pad, mode, tap0)
    (3, 9000, 128, 256, 64, 128, 2, False),    # exactly one 128-bin block, reflect
    (2, 9000, 129, 256, 64, 128, 1, False),    # + one leftover bin: evaluated by the pre-pass
    (2, 7001, 200, 512, 77, 256, 2, True),     # partial second block, ODD hop, tap 0 carried
    (1, 30000, 257, 2048, 512, 1024, 2, False),  # the cfg2 kernel shape, two blocks + Nyquist
    (2, 4000, 64, 64, 16, 0, 0, True),         # smallest kernel, center=False
    (5, 700, 130, 128, 32, 64, 2, False),      # many short clips: tiles span several clips
    (1, 40000, 129, 4096, 1024, 2048, 2, False),  # n_fft = 4096: 64 KB of LDS per pre-pass block
    (1, 70000, 65, 8192, 2048, 4096, 1, True),    # the longest kernel the fold takes
    (2, 6000, 130, 402, 100, 201, 2, False),   # K/2 odd: the quad that straddles tap K/2 (element-wise path)
    (3, 9000, 129, 1000, 250, 500, 2, False),  # 512 folded taps: two thread groups in the pre-pass + last bin
    (1, 3000, 129, 500, 125, 250, 1, True),    # 256 folded taps incl. tap 0: four thread groups in the pre-pass
])
@pytest.mark.parametrize("epi", ["complex", "magnitude", "power2", "phase"])
@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "f16x3"])
def test_symmetric_fold_kernel(shape, epi, precision):
    """``basis_fold`` in either arithmetic: the contraction over K/2 folded taps of x[n] +- x[K-n]
    (framed_fold.inl; split-bf16 or fp32 taps) against the float64 evaluation of the dense
    contraction."""
    from nnaudio_amd import engine

    B, L, F, K, hop, pad, mode, tap0 = shape
    rng = np.random.default_rng(F * 1000 + K + hop)
    x = rng.standard_normal((B, L)).astype(np.float32)
    wr, wi = _fourier_like_basis(rng, F, K, tap0)
    if precision == "f16x3":  # (the fp16 pairs hold coefficient x 2^14: |coefficient| <= 2; a window is <= 1)
        wr, wi = 0.3 * wr, 0.3 * wi
        x = x * np.float32(10.0 ** rng.uniform(-4, 4))  # any signal level must do
    scale = rng.uniform(0.5, 2.0, F).astype(np.float32)
    re, im = _np_framed(x, wr, wi, hop, pad, mode, scale)
    xd, wrd, wid, sd = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi, scale))
    prep = engine.prepare_basis(wrd, wid, precision, hop=hop)
    assert "basis_fold" in prep and prep["basis_fold"][1] == (K // 2 + (1 if tap0 else 0) + 15) // 16 * 16
    assert ("basis_split" in prep) == (precision == "bf16x3") and "basis_fold2" not in prep
    kw = dict(hop=hop, pad=pad, pad_mode=mode, row_scale=sd)
    e = {"complex": engine.EPI_COMPLEX, "magnitude": engine.EPI_MAGNITUDE, "power2": engine.EPI_POWER,
         "phase": engine.EPI_PHASE_ATAN2}[epi]
    y = engine.framed_gemm(xd, wrd, wid, precision=precision, epilogue=e, **kw, **prep)
    ydense = engine.framed_gemm(xd, wrd, wid, precision=precision, epilogue=e, **kw,
                                **{k: v for k, v in prep.items() if k == "basis_split"})
    torch.cuda.synchronize()
    y, ydense = y.cpu().numpy(), ydense.cpu().numpy()
    if F > 128:
        assert not np.array_equal(y, ydense)  # really took the folded kernel
    if epi == "complex":
        ref = np.stack((re, im), -1)
        assert_parity(y, ref, rel=1e-4, what="fold %s" % (shape,))
        # the split's own budget; the fp32 taps: the dense fp32 kernel's
        assert np.abs(y - ref).max() <= (2e-5 if precision == "bf16x3" else 3e-6) * np.abs(ref).max()
    elif epi == "magnitude":
        assert_parity(y, np.sqrt(re * re + im * im), rel=1e-4, what="fold %s" % (shape,))
    elif epi == "power2":
        assert_parity(y, re * re + im * im, rel=1e-4, what="fold %s" % (shape,))
    else:
        mag = np.sqrt(re * re + im * im)
        assert_phase_parity(y, np.arctan2(im, re), mag, what="fold %s" % (shape,))


def test_symmetric_fold_is_refused_for_other_bases():
    """A basis without the symmetry (CQT kernels, freq_scale != 'no', random) must not fold."""
    from nnaudio_amd import engine, features

    rng = np.random.default_rng(0)
    wr = torch.as_tensor(rng.standard_normal((140, 256)).astype(np.float32)).to(DEV)
    wi = torch.as_tensor(rng.standard_normal((140, 256)).astype(np.float32)).to(DEV)
    assert engine.fold_basis(wr, wi) is None
    m = features.STFT(n_fft=512, freq_scale="log", fmin=50, fmax=6000, sr=22050, verbose=False).to(DEV)
    assert engine.fold_basis(m.wcos, m.wsin) is None
    for win in ("hann", "hamming", ("kaiser", 8.0)):
        m = features.STFT(n_fft=512, window=win, verbose=False).to(DEV)
        assert engine.fold_basis(m.wcos, m.wsin) is not None, win
    m = features.STFT(n_fft=512, win_length=400, verbose=False).to(DEV)  # centred shorter window
    assert engine.fold_basis(m.wcos, m.wsin) is not None


def test_symmetric_fold_fused_filterbank():
    """Mel reduction fused into the folded contraction's epilogue == the two-kernel path, and both
    == float64 (incl. a filterbank with weight on the Nyquist bin, which the pre-pass adds)."""
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    rng = np.random.default_rng(7)
    x = rng.standard_normal((9, 12000)).astype(np.float32)
    xd = torch.as_tensor(x).to(DEV)
    for fmax in (None, 7000.0):
        m = features.MelSpectrogram(sr=16000, n_fft=512, n_mels=64, hop_length=128, fmax=fmax,
                                    verbose=False).to(DEV)
        m.stft.precision = "bf16x3"
        if fmax is None:  # put weight on the Nyquist bin
            with torch.no_grad():
                m.mel_basis[-1, 256] = 0.01
        with torch.no_grad():
            y = m(xd)
        ref = O.filterbank_spectrogram(x, m.stft.wsin.cpu().numpy(), m.stft.wcos.cpu().numpy(), 128,
                                       m.mel_basis.cpu().numpy())
        assert_parity(y.cpu().numpy(), ref, rel=1e-4, what="fold + fused mel fmax=%s" % fmax)


def test_bf16x3_split_cache_follows_the_basis(bf16x3):
    """The cached split planes must be rebuilt when the basis changes in place or is replaced."""
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8000, generator=g).to(DEV)
    m = features.STFT(n_fft=512, hop_length=128, output_format="Magnitude", verbose=False).to(DEV)
    y0 = m(x)
    assert m(x).data_ptr() != y0.data_ptr() and torch.equal(m(x), y0)
    with torch.no_grad():
        m.wcos.mul_(2.0)
        m.wsin.mul_(2.0)
    y1 = m(x)
    assert torch.allclose(y1, 2.0 * y0, rtol=1e-6, atol=0.0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["wcos"] *= 0.25
    sd["wsin"] *= 0.25
    m.load_state_dict(sd)
    assert torch.allclose(m(x), 0.5 * y0, rtol=1e-6, atol=0.0)
    m2 = m.to("cpu").to(DEV)  # fresh tensors at possibly recycled addresses
    assert torch.allclose(m2(x), 0.5 * y0, rtol=1e-6, atol=0.0)


def test_bf16x3_cfg2_full_size_sampled():
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    B, L = 64, 441000
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, L, generator=g)
    m = features.STFT(n_fft=2048, hop_length=512, window="hann", output_format="Complex",
                      verbose=False).to(DEV)
    m.precision = "bf16x3"
    xd = x.to(DEV)
    y = m(xd)
    assert tuple(y.shape) == (64, 1025, 862, 2)
    rng = np.random.default_rng(1)
    cb, ct = _sample_cols(rng, B, 862, 96)
    ct[:8] = [0, 1, 2, 3, 858, 859, 860, 861]
    re, im = O.sampled_complex(x.numpy(), m.wcos.cpu().numpy(), m.wsin.cpu().numpy(), cb, ct,
                               512, 1024, "reflect")
    got = y[torch.as_tensor(cb), :, torch.as_tensor(ct)].cpu().numpy()
    peak = max(np.abs(re).max(), np.abs(im).max())
    err = max(np.abs(got[..., 0] - re).max(), np.abs(got[..., 1] - im).max())
    assert err <= 1e-4 * peak
    assert err <= 2e-5 * peak  # the split's own budget (measured 4e-6): catches a lost term
    m.precision = "fp32"
    y32 = m(xd)
    assert (y - y32).abs().max().item() <= 2e-5 * y32.abs().max().item()


def test_mfcc_power_to_db_and_errors(both_stft_routes):
    """power_to_db against the numpy restatement incl. the per-clip top_db floor, in place, and
    the reference's parameter errors (mel.py:255, 273)."""
    from nnaudio_amd import engine, features
    from oracle import spectral_oracle as O

    rng = np.random.default_rng(5)
    S = (rng.random((3, 40, 57)) ** 8 * np.array([1.0, 1e3, 1e-6])[:, None, None]).astype(np.float32)
    S[0, 0, 0] = 0.0
    Sd = torch.as_tensor(S).to(DEV)
    for amin, ref, top_db in ((1e-10, 1.0, 80.0), (1e-6, 2.0, None), (1e-10, 1.0, 0.0)):
        got = engine.power_to_db(Sd, amin, ref, top_db).cpu().numpy()
        want = O.power_to_db(S, amin, ref, top_db)
        assert np.abs(got - want).max() <= 1e-4, (amin, ref, top_db)
    with pytest.raises(NameError):
        features.MFCC(amin=0.0, verbose=False)
    m = features.MFCC(sr=16000, n_fft=512, n_mels=40, top_db=-1.0, verbose=False).to(DEV)
    with pytest.raises(NameError):
        m(torch.zeros(1, 4000, device=DEV))


def test_stft_istft_round_trip(both_stft_routes):
    """reference tests/test_stft.py:28-54: inverse(STFT(x)) == x on randn(4, 16000), for the
    hop lengths the reference's parameter grid uses, through STFT.inverse and the iSTFT class."""
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(11)
    x = torch.randn(4, 16000, generator=g).to(DEV)
    for n_fft, hop, window in ((2048, 512, "hann"), (1024, 128, "hann"), (512, 100, "hamming")):
        fwd = features.STFT(n_fft=n_fft, hop_length=hop, window=window, iSTFT=True, verbose=False).to(DEV)
        X = fwd(x, output_format="Complex")
        y = fwd.inverse(X, length=x.shape[1])
        assert y.dtype == torch.float32 and tuple(y.shape) == (4, 16000)
        assert torch.allclose(y, x, rtol=1e-5, atol=1e-3)  # the reference's own tolerance
        assert (y - x).abs().max().item() <= 2e-5
        inv = features.iSTFT(n_fft=n_fft, hop_length=hop, window=window, verbose=False).to(DEV)
        y2 = inv(X, onesided=True, length=x.shape[1])
        assert y2.dtype == torch.float64  # float64 window buffer, as in the reference
        assert (y2.float() - x).abs().max().item() <= 2e-5
    with pytest.raises(NameError):
        features.STFT(n_fft=512, verbose=False).to(DEV).inverse(X)


# ---------------------------------------------------------------------------------------
# backward: trainable bases and differentiable inputs, against torch autograd
# on a plain conv1d restatement of the reference's forward (fp32, same device)
# ---------------------------------------------------------------------------------------
def _torch_framed(x, w_re, w_im, hop, pad, mode):
    import torch.nn.functional as F

    xp = x[:, None, :]
    if pad:
        xp = F.pad(xp, (pad, pad), mode="reflect" if mode == "reflect" else "constant")
    return F.conv1d(xp, w_re, stride=hop), F.conv1d(xp, w_im, stride=hop)


def _grad_close(got, want, what, rel=2e-4):
    assert got is not None, what
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= rel * ref, "%s: max|d|=%.3e vs %.3e" % (what, err, ref)


@pytest.mark.parametrize("fmt", ["Magnitude", "Complex", "Phase"])
@pytest.mark.parametrize("pad_mode,center", [("reflect", True), ("constant", True), ("reflect", False)])
def test_backward_stft(fmt, pad_mode, center, both_stft_routes):
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 6000, generator=g).to(DEV).requires_grad_(True)
    m = features.STFT(n_fft=512, hop_length=160, trainable=True, output_format=fmt, pad_mode=pad_mode,
                      center=center, verbose=False).to(DEV)
    y = m(x)
    w = torch.randn(y.shape, generator=g).to(DEV)
    if fmt == "Phase":
        # d atan2 ~ 1/|z|^2: weight only bins that carry energy, or fp32 noise in either
        # implementation dominates the comparison
        mag = m(x.detach(), output_format="Magnitude").detach()
        w = w * (mag > 0.25 * mag.mean())
    (y * w).sum().backward()
    got = x.grad.clone(), m.wcos.grad.clone(), m.wsin.grad.clone()

    x2 = x.detach().clone().requires_grad_(True)
    wc, ws = m.wcos.detach().clone().requires_grad_(True), m.wsin.detach().clone().requires_grad_(True)
    re, im = _torch_framed(x2, wc, ws, 160, 256 if center else 0, pad_mode)
    if fmt == "Magnitude":
        y2 = torch.sqrt(re ** 2 + im ** 2 + 1e-8)
    elif fmt == "Complex":
        y2 = torch.stack((re, -im), -1)
    else:
        y2 = torch.atan2(-im + 0.0, re)
    if fmt != "Phase":  # (phase values: +pi / -pi at the real-valued DC / Nyquist bins)
        assert (y - y2).abs().max().item() <= 1e-4 * max(1.0, y2.abs().max().item())
    (y2 * w).sum().backward()
    _grad_close(got[0], x2.grad, "d x")
    _grad_close(got[1], wc.grad, "d wcos")
    _grad_close(got[2], ws.grad, "d wsin")


def test_backward_in_the_forwards_arithmetic_and_split_over_clips():
    """Round 5: under ``precision = "f16x3"`` (the STFT module's default) the backward recomputes the Complex values in the
    forward's own arithmetic and runs the d-basis contraction on the staged dense f16x3 kernel, its sum over the clips
    split into four chunks on side streams (B >= 8).  Both arithmetics against a float64 evaluation of the same loss: the
    gradient of a Magnitude is ill-conditioned where a bin's |z| crosses zero (d wcos at the DC bin), so the bar is what
    fp32 itself reaches there, and f16x3 must be no worse than twice that."""
    from nnaudio_amd import features

    x = torch.randn(8, 20224, generator=torch.Generator().manual_seed(31)).to(DEV)  # (80 frames per clip: even chunks)
    err = {}
    for prec in ("fp32", None):
        m = features.STFT(n_fft=1024, hop_length=256, trainable=True, output_format="Magnitude", verbose=False).to(DEV)
        m.precision = prec
        y = m(x)
        w = torch.randn(y.shape, generator=torch.Generator().manual_seed(32)).to(DEV)
        (y * w).sum().backward()
        torch.cuda.synchronize()
        wc = m.wcos.detach().double().reshape(513, 1024).requires_grad_(True)
        ws = m.wsin.detach().double().reshape(513, 1024).requires_grad_(True)
        xp = torch.nn.functional.pad(x.double()[:, None, :], (512, 512), mode="reflect")[:, 0]
        fr = xp.unfold(1, 1024, 256)
        re, im = torch.einsum("btk,fk->bft", fr, wc), torch.einsum("btk,fk->bft", fr, ws)
        (torch.sqrt(re ** 2 + im ** 2 + 1e-8) * w.double()).sum().backward()
        err[prec] = [float((a.double().reshape(513, 1024) - b).abs().max() / b.abs().max())
                     for a, b in ((m.wcos.grad, wc.grad), (m.wsin.grad, ws.grad))]
        print("STFT(trainable) backward, precision %s: d wcos %.2e, d wsin %.2e of the maximum (float64 reference)" % (prec, *err[prec]))
    assert max(err["fp32"]) <= 3e-3 and max(err[None]) <= 3e-3
    assert err[None][0] <= 2 * err["fp32"][0] + 1e-5 and err[None][1] <= 2 * err["fp32"][1] + 1e-5


@pytest.mark.parametrize("fmt", ["Magnitude", "Complex", "Phase"])
def test_backward_cqt1992v2(fmt):
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 9000, generator=g).to(DEV).requires_grad_(True)
    m = features.CQT1992v2(sr=8000, fmin=110, n_bins=36, hop_length=128, trainable=True,
                           output_format=fmt, verbose=False).to(DEV)
    y = m(x)
    w = torch.randn(y.shape, generator=g).to(DEV)
    (y * w).sum().backward()
    got = x.grad.clone(), m.cqt_kernels_real.grad.clone(), m.cqt_kernels_imag.grad.clone()

    x2 = x.detach().clone().requires_grad_(True)
    kr = m.cqt_kernels_real.detach().clone().requires_grad_(True)
    ki = m.cqt_kernels_imag.detach().clone().requires_grad_(True)
    re, im = _torch_framed(x2, kr, ki, 128, m.kernel_width // 2, "reflect")
    s = torch.sqrt(m.lenghts.view(-1, 1))
    re, im = re * s, -im * s
    if fmt == "Magnitude":
        y2 = torch.sqrt(re ** 2 + im ** 2 + 1e-8)
    elif fmt == "Complex":
        y2 = torch.stack((re, im), -1)
    else:
        a = torch.atan2(im, re)
        y2 = torch.stack((torch.cos(a), torch.sin(a)), -1)
    (y2 * w).sum().backward()
    _grad_close(got[0], x2.grad, "d x")
    _grad_close(got[1], kr.grad, "d cqt_kernels_real")
    _grad_close(got[2], ki.grad, "d cqt_kernels_imag")


def test_backward_mel_and_frozen_front_end(both_stft_routes):
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(23)
    x = torch.randn(3, 5000, generator=g).to(DEV)
    m = features.MelSpectrogram(sr=16000, n_fft=512, n_mels=40, hop_length=128, power=2.0,
                                trainable_mel=True, trainable_STFT=True, verbose=False).to(DEV)
    y = m(x)
    w = torch.randn(y.shape, generator=g).to(DEV)
    (y * w).sum().backward()
    wc = m.stft.wcos.detach().clone().requires_grad_(True)
    ws = m.stft.wsin.detach().clone().requires_grad_(True)
    mb = m.mel_basis.detach().clone().requires_grad_(True)
    re, im = _torch_framed(x, wc, ws, 128, 256, "reflect")
    y2 = torch.matmul(mb, torch.sqrt(re ** 2 + im ** 2 + 1e-8) ** 2.0)
    (y2 * w).sum().backward()
    _grad_close(m.mel_basis.grad, mb.grad, "d mel_basis")
    _grad_close(m.stft.wcos.grad, wc.grad, "d wcos")
    _grad_close(m.stft.wsin.grad, ws.grad, "d wsin")
    # a frozen front-end inside a training graph: gradient w.r.t. the waveform only
    f = features.STFT(n_fft=256, hop_length=64, output_format="Magnitude", verbose=False).to(DEV)
    xg = x.clone().requires_grad_(True)
    f(xg).sum().backward()
    x3 = x.clone().requires_grad_(True)
    re, im = _torch_framed(x3, f.wcos, f.wsin, 64, 128, "reflect")
    torch.sqrt(re ** 2 + im ** 2).sum().backward()
    _grad_close(xg.grad, x3.grad, "d x (frozen STFT)")


@pytest.mark.parametrize("onesided,length", [(True, None), (False, None), (True, 3000)])
def test_backward_istft(onesided, length, both_stft_routes):
    """d spectrogram and d synthesis kernels of the inverse STFT against torch autograd on a
    fold-based restatement."""
    from nnaudio_amd import engine, features

    n_fft, hop, T, B = 256, 64, 50, 3
    F = n_fft // 2 + 1 if onesided else n_fft
    g = torch.Generator().manual_seed(27)
    X = torch.randn(B, F, T, 2, generator=g).to(DEV).requires_grad_(True)
    m = features.iSTFT(n_fft=n_fft, hop_length=hop, trainable_kernels=True, verbose=False).to(DEV)
    y = m(X, onesided=onesided, length=length)
    w = torch.randn(y.shape, generator=g, dtype=torch.float64).to(DEV)
    (y * w).sum().backward()

    X2 = X.detach().clone().requires_grad_(True)
    kc = m.kernel_cos.detach().clone().requires_grad_(True)
    ks = m.kernel_sin.detach().clone().requires_grad_(True)
    basis = engine.istft_basis(kc, ks, F, onesided)
    win = m.window_mask.reshape(-1).float()
    Xp = torch.cat((X2[..., 0], X2[..., 1]), 1)
    frames = torch.einsum("nc,bct->bnt", basis, Xp) * win[None, :, None] / n_fft
    full = (T - 1) * hop + n_fft
    fold = lambda f: torch.nn.functional.fold(f, (1, full), (1, n_fft), stride=(1, hop)).reshape(-1, full)
    wss = fold((win ** 2)[None, :, None].expand(1, n_fft, T))
    yf = fold(frames) / torch.where(wss > 1e-10, wss, torch.ones_like(wss))
    pad = n_fft // 2
    y2 = yf[:, pad:full - pad] if length is None else yf[:, pad:pad + length]
    assert y2.shape == y.shape
    assert (y.float() - y2).abs().max().item() <= 1e-4 * y2.abs().max().item()
    (y2 * w.float()).sum().backward()
    _grad_close(X.grad, X2.grad, "d spectrogram", rel=5e-4)
    _grad_close(m.kernel_cos.grad, kc.grad, "d kernel_cos", rel=5e-4)
    _grad_close(m.kernel_sin.grad, ks.grad, "d kernel_sin", rel=5e-4)


@pytest.mark.parametrize("onesided,length,center", [(True, None, True), (False, 2500, True), (True, None, False)])
def test_backward_istft_trainable_window(onesided, length, center):
    """d window of the inverse STFT (iSTFT(trainable_window=True), reference stft.py:511-512): through
    the windowed frames AND through the window-sum-square normalisation, against torch autograd on
    the fold-based restatement (float64)."""
    from nnaudio_amd import engine, features

    n_fft, hop, T, B = 256, 64, 40, 2
    F = n_fft // 2 + 1 if onesided else n_fft
    g = torch.Generator().manual_seed(29)
    X = torch.randn(B, F, T, 2, generator=g).to(DEV).requires_grad_(True)
    m = features.iSTFT(n_fft=n_fft, hop_length=hop, trainable_window=True, center=center, verbose=False).to(DEV)
    y = m(X, onesided=onesided, length=length)
    w = torch.randn(y.shape, generator=g, dtype=torch.float64).to(DEV)
    (y * w).sum().backward()
    assert m.window_mask.grad is not None and m.window_mask.grad.shape == m.window_mask.shape

    X2 = X.detach().double().clone().requires_grad_(True)
    win = m.window_mask.detach().reshape(-1).double().clone().requires_grad_(True)
    basis = engine.istft_basis(m.kernel_cos, m.kernel_sin, F, onesided).double()
    Xp = torch.cat((X2[..., 0], X2[..., 1]), 1)
    frames = torch.einsum("nc,bct->bnt", basis, Xp) * win[None, :, None] / n_fft
    full = (T - 1) * hop + n_fft
    fold = lambda f: torch.nn.functional.fold(f, (1, full), (1, n_fft), stride=(1, hop)).reshape(-1, full)
    wss = fold((win ** 2)[None, :, None].expand(1, n_fft, T))
    yf = fold(frames) / torch.where(wss > 1e-10, wss, torch.ones_like(wss))
    pad = n_fft // 2 if center else 0
    y2 = yf[:, pad:full - pad] if length is None else yf[:, pad:pad + length]
    assert y2.shape == y.shape
    # (without centring the first / last samples divide by a window sum near zero: compare the gradient
    # through a weight that ignores them, as the forward parity tests do)
    cond = (wss[0, pad:pad + y.shape[1]] > 1e-2 * wss.max()).double()
    m.window_mask.grad = None
    X.grad = None
    y = m(X, onesided=onesided, length=length)
    (y * w * cond).sum().backward()
    (y2 * w * cond).sum().backward()
    _grad_close(m.window_mask.grad.reshape(-1), win.grad, "d window", rel=1e-3)
    _grad_close(X.grad.double(), X2.grad, "d spectrogram", rel=1e-3)


@pytest.mark.parametrize("top_db", [80.0, 20.0, None])
def test_backward_mfcc(top_db, both_stft_routes):
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(25)
    x = torch.randn(3, 6000, generator=g).to(DEV)
    m = features.MFCC(sr=16000, n_mfcc=13, n_fft=512, n_mels=40, hop_length=160, top_db=top_db,
                      trainable_mel=True, trainable_STFT=True, verbose=False).to(DEV)
    y = m(x)
    w = torch.randn(y.shape, generator=g).to(DEV)
    (y * w).sum().backward()
    ml = m.melspec_layer
    wc = ml.stft.wcos.detach().clone().requires_grad_(True)
    ws = ml.stft.wsin.detach().clone().requires_grad_(True)
    mb = ml.mel_basis.detach().clone().requires_grad_(True)
    re, im = _torch_framed(x, wc, ws, 160, 256, "reflect")
    S = torch.matmul(mb, torch.sqrt(re ** 2 + im ** 2 + 1e-8) ** 2.0)
    L = 10.0 * torch.log10(torch.max(S, m.amin)) - 10.0 * torch.log10(torch.max(m.amin, m.ref))
    if top_db is not None:
        L = torch.max(L, L.flatten(1).max(1)[0][:, None, None] - top_db)
    y2 = torch.matmul(m._dct_basis, L)
    assert (y - y2).abs().max().item() <= 1e-4 * y2.abs().max().item()
    (y2 * w).sum().backward()
    _grad_close(ml.mel_basis.grad, mb.grad, "d mel_basis", rel=5e-4)
    _grad_close(ml.stft.wcos.grad, wc.grad, "d wcos", rel=5e-4)
    _grad_close(ml.stft.wsin.grad, ws.grad, "d wsin", rel=5e-4)


@pytest.mark.parametrize("cls,fmt", [("CQT2010v2", "Magnitude"), ("CQT2010v2", "Complex"),
                                     ("VQT", "Magnitude")])
def test_backward_octave_recursion(cls, fmt):
    """CQT2010v2 (trainable top-octave kernels) / VQT (differentiable input) against torch autograd
    on the reference's own structure: early down-sampling, per-octave conv1d pair, decimation by
    the anti-alias FIR, cat, bottom cut, scaling (cqt.py:1070-1139, vqt.py:143-215)."""
    import torch.nn.functional as F

    from nnaudio_amd import features

    g = torch.Generator().manual_seed(24)
    x = torch.randn(2, 12000, generator=g).to(DEV).requires_grad_(True)
    kw = dict(sr=16000, fmin=110, n_bins=40, hop_length=256, output_format=fmt, verbose=False)
    if cls == "CQT2010v2":
        m = features.CQT2010v2(trainable=True, **kw).to(DEV)
        params = [m.cqt_kernels_real, m.cqt_kernels_imag]
    else:
        m = features.VQT(gamma=5, **kw).to(DEV)
        params = []
    y = m(x)
    w = torch.randn(y.shape, generator=g).to(DEV)
    (y * w).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in params]

    x2 = x.detach().clone().requires_grad_(True)
    p2 = [p.detach().clone().requires_grad_(True) for p in params]

    def fir(sig, taps, stride):
        return F.conv1d(sig, taps, stride=stride, padding=(taps.shape[-1] - 1) // 2)

    def octave(sig, kr, ki, hop):
        pad = kr.shape[-1] // 2
        sp = F.pad(sig, (pad, pad), mode="reflect")
        return torch.stack((F.conv1d(sp, kr, stride=hop), -F.conv1d(sp, ki, stride=hop)), -1)

    sig = x2[:, None, :]
    if m.earlydownsample and m.early_downsample_filter is not None:
        sig = fir(sig, m.early_downsample_filter, int(m.downsample_factor))
    hop = m.hop_length
    banks = ([(p2[0], p2[1])] * m.n_octaves if cls == "CQT2010v2" else
             [(getattr(m, "cqt_kernels_real_%d" % i), getattr(m, "cqt_kernels_imag_%d" % i))
              for i in range(m.n_octaves)])
    C = octave(sig, banks[0][0], banks[0][1], hop)
    for i in range(1, m.n_octaves):
        hop //= 2
        sig = fir(sig, m.lowpass_filter, 2)
        C = torch.cat((octave(sig, banks[i][0], banks[i][1], hop), C), 1)
    C = C[:, -m.n_bins:] * m.downsample_factor * torch.sqrt(m.lenghts.view(-1, 1, 1))
    if fmt == "Magnitude":
        y2 = torch.sqrt(C.pow(2).sum(-1) + (1e-8 if cls == "CQT2010v2" else 0.0))
    else:
        y2 = C
    assert (y - y2).abs().max().item() <= 1e-4 * y2.abs().max().item()
    (y2 * w).sum().backward()
    _grad_close(got[0], x2.grad, "d x")
    for a, b, name in zip(got[1:], p2, ("d cqt_kernels_real", "d cqt_kernels_imag")):
        _grad_close(a, b.grad, name)


def test_fused_filterbank_matches_unfused():
    """The mel reduction fused into the contraction's epilogue (both precisions; stores, and
    float atomics into the zeroed output for bands crossing a tile boundary) against the
    two-kernel path: filter bands crossing bin-block boundaries, the Nyquist bin's own tile,
    clip-straddling frame tiles, odd hop (bf16x3 request served by the fp32 kernels), power 1
    and 2; dense filterbanks and graphs fall back to the separate filterbank kernel."""
    from nnaudio_amd import engine, features

    g = torch.Generator().manual_seed(31)
    for n_fft, hop, n_mels, L, B, power in ((1024, 512, 128, 30000, 5, 2.0), (2048, 256, 64, 9000, 3, 1.0),
                                            (512, 128, 40, 4100, 9, 2.0), (512, 127, 40, 4100, 2, 2.0),
                                            (128, 32, 20, 3000, 4, 2.0)):
        x = torch.randn(B, L, generator=g).to(DEV)
        m = features.MelSpectrogram(sr=22050, n_fft=n_fft, n_mels=n_mels, hop_length=hop, power=power,
                                    verbose=False).to(DEV)
        for prec in ("fp32", "bf16x3"):
            m.stft.precision = prec
            assert engine.fused_filterbank_plan(m, m.mel_basis, x, m.stft, m.power) is not None
            y = m(x)
            spec = m.stft._spectrum(x[:, None, :], engine.EPI_POWER, power=power)
            y2 = engine.filterbank(m.mel_basis, spec)  # two kernels, same arithmetic
            peak = y2.abs().max().item()
            assert y.shape == y2.shape
            assert (y - y2).abs().max().item() <= 2e-5 * peak, (n_fft, hop, prec)
            if n_mels >= 128:
                assert torch.equal(y, m(x))  # at most two addends per output: order independent
    # a dense filterbank (gammatone) and a graph go through the separate kernel
    gt = features.Gammatonegram(sr=22050, n_fft=1024, n_bins=64, hop_length=512, verbose=False).to(DEV)
    x = torch.randn(2, 20000, generator=g).to(DEV)
    assert engine.fused_filterbank_plan(gt, gt.gammatone_basis, x, gt.stft, gt.power) is None
    mt = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=64, trainable_mel=True, verbose=False).to(DEV)
    assert engine.fused_filterbank_plan(mt, mt.mel_basis, x, mt.stft, mt.power) is None
    with torch.no_grad():
        assert engine.fused_filterbank_plan(mt, mt.mel_basis, x, mt.stft, mt.power) is not None
    # the C entry refuses the fusion where it is not served
    sup, _ = engine.filterbank_support(m.mel_basis)
    with pytest.raises(RuntimeError):
        engine.framed_gemm(x, m.stft.wcos, m.stft.wsin, hop=32, pad=64, pad_mode=engine.PAD_REFLECT,
                           epilogue=engine.EPI_POWER, power=1.5, fb=m.mel_basis, fb_support=sup)
    with pytest.raises(RuntimeError):
        engine.framed_gemm(x, m.stft.wcos, m.stft.wsin, hop=32, pad=64, pad_mode=engine.PAD_REFLECT,
                           epilogue=engine.EPI_POWER, tile=engine.TILE_128x128 if hasattr(engine, "TILE_128x128") else 1,
                           fb=m.mel_basis, fb_support=sup)


def test_rccl_single_rank_sharded_forward(both_stft_routes):
    """The one-process-per-GPU path on the real backend: `nccl` (= RCCL) process group of one
    rank, sharded forward + all-gather reassembly (the world-size-2 logic is covered on gloo in
    the CPU suite).  Runs in a subprocess: process groups are process-global state."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "nccl_one_rank.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "nccl 1-rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_repeated_launches_are_bit_identical():
    """A race between LDS-direct loads, fragment reads and stores would show up as run-to-run
    differences: 30 launches each of the full-size STFT (both precisions), CQT84 and fused-mel
    workloads must reproduce the first result bit for bit."""
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(41)
    x = torch.randn(64, 441000, generator=g).to(DEV)
    stft = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(DEV)
    cqt = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to(DEV)
    xm = torch.randn(256, 110250, generator=g).to(DEV)
    mel = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to(DEV)
    for mod, inp, precs in ((stft, x, ("fp32", "bf16x3")), (cqt, x, ("bf16x3",)), (mel, xm, ("fp32", "bf16x3"))):
        for prec in precs:
            if hasattr(mod, "stft"):
                mod.stft.precision = prec
            else:
                mod.precision = prec
            first = mod(inp).clone()
            for _ in range(30):
                assert torch.equal(mod(inp), first), (type(mod).__name__, prec)
            del first


@pytest.mark.parametrize("cls,ctor", [
    ("STFT", dict(n_fft=512, hop_length=128, output_format="Magnitude")),
    ("MelSpectrogram", dict(sr=16000, n_fft=512, n_mels=40, hop_length=128)),
    ("CQT1992v2", dict(sr=16000, hop_length=128, fmin=110, n_bins=60, output_format="Complex")),
    ("CQT2010v2", dict(sr=16000, hop_length=128, fmin=110, n_bins=60, earlydownsample=False)),
    ("VQT", dict(sr=16000, hop_length=128, fmin=110, n_bins=60, gamma=3, earlydownsample=False)),
    # frequency-domain CQT2010: own scale and imaginary sign -> the per-octave ops, not the single op
    ("CQT2010", dict(sr=16000, hop_length=128, fmin=110, n_bins=60, earlydownsample=False)),
])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "f16x3"])
def test_torch_compile_runs_the_custom_ops(cls, ctor, precision):
    """torch.compile(fullgraph=True): no graph break at the C boundary (nnaudio_amd.ops), same
    numbers as the eager module (bit for bit where both take the same kernels)."""
    import nnaudio_amd

    old = nnaudio_amd.get_precision()
    nnaudio_amd.set_precision(precision)
    try:
        mod = build_module(dict(cls=cls, ctor=ctor, fwd={}), DEV)
        x = torch.randn(4, 12000, generator=torch.Generator().manual_seed(9)).to(DEV)
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = mod(x)
            got = torch.compile(mod, fullgraph=True, backend="aot_eager")(x)
        torch.cuda.synchronize()
        assert got.shape == want.shape
        assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
        if cls in ("MelSpectrogram", "CQT2010v2", "VQT"):
            # the compiled module runs the SAME fused kernels as the eager one (the filterbank in the
            # contraction's epilogue; the octave recursion on the pyramid kernel in bf16x3): bit for
            # bit -- the unfused launch structure differs in the last bits
            assert torch.equal(got, want), "compiled %s did not take the fused path" % cls
        if cls == "CQT1992v2" and precision == "fp32":
            # one float32 FMA chain per output whichever kernel serves it (the op prepares the chain kernel's copy as well)
            assert torch.equal(got, want)
    finally:
        nnaudio_amd.set_precision(old)
        torch._dynamo.reset()


def test_integration_stub_computes_an_stft():
    """The ctypes stub printed in INTEGRATION.md, executed as is against the in-tree library:
    framed() in fp32 (dense and folded), bf16x3 and bf16x3 + fold_basis reproduces the module."""
    import re

    from nnaudio_amd import _abi, features

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", text, flags=re.S).group(1)
    code = code.replace('ctypes.CDLL("libmispec.so")', "ctypes.CDLL(%r)" % _abi.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    m = features.STFT(n_fft=512, hop_length=128, output_format="Magnitude", verbose=False).to(DEV)
    m.precision = "fp32"
    x = torch.randn(5, 1, 20000, generator=torch.Generator().manual_seed(2)).to(DEV)
    with torch.no_grad():
        want = m(x)
        y32 = ns["framed"](x, m.wcos, m.wsin, 128, 256, 2, 1)
        fold32 = ns["fold_basis"](m.wcos, m.wsin, bf16x3=False)
        y32f = ns["framed"](x, m.wcos, m.wsin, 128, 256, 2, 1, fold=fold32)
        split = ns["split_basis"](m.wcos, m.wsin)
        fold = ns["fold_basis"](m.wcos, m.wsin)
        yb = ns["framed"](x, m.wcos, m.wsin, 128, 256, 2, 1, split=split)
        yf = ns["framed"](x, m.wcos, m.wsin, 128, 256, 2, 1, split=split, fold=fold)
        fold2_32 = ns["fold2_basis"](m.wcos, m.wsin, 0)
        y32q = ns["framed"](x, m.wcos, m.wsin, 128, 256, 2, 1, fold=fold32, fold2=fold2_32, fft=False)
        fold2_16 = ns["fold2_basis"](m.wcos, m.wsin, 2)
        y16 = ns["framed"](x, m.wcos, m.wsin, 128, 256, 2, 1, fold2=fold2_16, f16x3=True, fft=False)
        yfft = ns["framed"](x, m.wcos, m.wsin, 128, 256, 2, 1, fold2=fold2_16, f16x3=True)  # the FFT path
        from nnaudio_amd import engine

        engine.set_fft(True)
        try:
            want_fft = m(x)
        finally:
            engine.set_fft(False)
    torch.cuda.synchronize()
    assert torch.equal(yfft, want_fft) and not torch.equal(yfft, want)
    assert (yfft - want).abs().max().item() <= 3e-6 * want.abs().max().item()
    # the module's fp32 forward IS the twice-folded fp32 contraction; the others differ by rounding
    assert fold is not None and fold32 is not None and fold2_32 is not None and fold2_16 is not None
    assert torch.equal(y32q, want)
    for y in (y32, y32f):
        assert not torch.equal(y, want) and (y - want).abs().max().item() <= 5e-6 * want.abs().max().item()
    assert (y16 - want).abs().max().item() <= 3e-6 * want.abs().max().item()
    m.precision = None  # the module's own default is that f16x3 call
    with torch.no_grad():
        assert torch.equal(m(x), y16)
    for y in (yb, yf):
        assert (y - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    assert not torch.equal(yb, yf)


@pytest.mark.parametrize("name", ["stft", "mel", "mfcc", "cqt1992v2", "cqt2010v2"])
def test_forward_is_hip_graph_capturable(bf16x3, name):
    """A forward (after one eager warm-up call, which builds the cached operands) records into a
    HIP graph -- no host synchronisation, no allocation outside the capture pool, the strip
    kernel's job counter re-armed by the captured pre-pass -- and the replay on new input values
    reproduces the eager result bit for bit."""
    from nnaudio_amd import features

    torch.manual_seed(3)
    if name == "stft":
        m = features.STFT(n_fft=1024, hop_length=256, output_format="Magnitude", verbose=False)
        L = 40000
    elif name == "mel":
        m = features.MelSpectrogram(sr=22050, n_fft=1024, hop_length=256, n_mels=64, verbose=False)
        L = 40000
    elif name == "mfcc":
        m = features.MFCC(sr=22050, n_mfcc=20, n_fft=1024, hop_length=256, n_mels=64, verbose=False)
        L = 40000
    elif name == "gammatone":  # (FFT route: frame-major power spectrogram + the framed contraction over its bins)
        m = features.Gammatonegram(sr=22050, n_fft=1024, hop_length=256, n_bins=48, verbose=False)
        L = 40000
    elif name == "cqt1992v2":
        m = features.CQT1992v2(sr=22050, hop_length=256, n_bins=72, bins_per_octave=12, fmin=65.4, verbose=False)
        L = 60000  # 235 frames per clip: the strip kernel
    else:
        m = features.CQT2010v2(sr=22050, hop_length=256, n_bins=60, bins_per_octave=12, fmin=65.4, verbose=False)
        L = 60000
    m = m.to(DEV)
    x = torch.randn(3, L, device=DEV)
    with torch.no_grad():
        m(x)  # warm-up: split / folded planes, supports, LDS attributes
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            with torch.cuda.graph(g, stream=stream):
                y = m(x)
        torch.cuda.current_stream().wait_stream(stream)
        for seed in (4, 5):
            x.copy_(torch.randn(3, L, device=DEV, generator=torch.Generator(DEV).manual_seed(seed)))
            g.replay()
            torch.cuda.synchronize()
            want = m(x)
            assert torch.equal(y, want), name


@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "f16x3"])
def test_strip_kernel_random_shapes(precision):
    """40 random problems (hop, kernel length, bins, clips, frames, padding, epilogue; supports of
    random length AND position, some empty) through the strip kernel's planner and kernel -- the
    split-bf16 one and the fp32 one (1e-5 of the peak) -- against the one-thread-per-output device
    kernel (scripts/strip_fuzz.py; exits non-zero on a miss)."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "scripts", "strip_fuzz.py"), "40", "7", precision],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    taken = int(re.search(r"all ok; strip kernel taken in (\d+) cases", res.stdout).group(1))
    assert taken >= 30
