"""GPU parity of the second symmetric fold (framed_fold2.inl: the contraction over K/4 taps for
window x DFT bases, even / odd bins) in the three arithmetics, and of the scaled-fp16 split
("f16x3") against float64 -- through the C ABI (engine.framed_gemm)."""
import numpy as np
import pytest
import torch

from tests._golden import assert_parity, assert_phase_parity
from tests.test_gpu_parity import DEV, _np_framed

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    from nnaudio_amd import _abi

    _abi.load()  # fail loudly if the extension is not built

# |y - float64| / peak the arithmetics must stay under on white noise (Complex output)
BUDGET = {"fp32": 3e-6, "bf16x3": 2e-5, "f16x3": 2e-6}


def _dft_basis(F, K, window, rng):
    """wcos / wsin as stft.py:230-232 builds them: float32(cos) * float32(window) in float32."""
    n = np.arange(K)
    if window == "hann":
        w = 0.5 - 0.5 * np.cos(2 * np.pi * n / K)
    elif window == "hamming":
        w = 0.54 - 0.46 * np.cos(2 * np.pi * n / K)
    elif window == "short":  # win_length < n_fft, centred (pad_center)
        L = K // 2
        w = np.zeros(K)
        w[(K - L) // 2:(K - L) // 2 + L] = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(L) / L)
    else:  # any window will do: no symmetry needed
        w = rng.uniform(0.1, 1.5, K)
    w = w.astype(np.float32)
    k = np.arange(F)[:, None]
    wr = np.cos(2 * np.pi * k * n / K).astype(np.float32) * w
    wi = np.sin(2 * np.pi * k * n / K).astype(np.float32) * w
    return wr, wi


@pytest.mark.parametrize("shape", [  # (B, L, bins, K, hop, pad, mode, window)
    (1, 30000, 1025, 2048, 512, 1024, 2, "hann"),     # the cfg2 basis: 4 + 4 row tiles + Nyquist in the pre-pass
    (3, 9000, 513, 1024, 256, 512, 2, "hann"),        # cfg3's
    (2, 9000, 257, 512, 128, 256, 1, "hamming"),      # w[0] != 0, zero padding
    (5, 1500, 129, 256, 64, 128, 2, "random"),        # asymmetric window; many short clips
    (2, 4000, 128, 128, 32, 0, 0, "hann"),            # smallest kernel, center=False, freq_bins > K/2+1 is not needed
    (1, 40000, 300, 4096, 1024, 2048, 2, "short"),    # partial tiles in both parities, win_length < n_fft
    (1, 70000, 259, 8192, 2048, 4096, 1, "hann"),     # the longest kernel; odd parity with one leftover bin
    (2, 7001, 200, 640, 91, 320, 2, "hann"),          # K = 640 (Q = 160), ODD hop
])
@pytest.mark.parametrize("epi", ["complex", "magnitude", "power2", "phase"])
@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "f16x3"])
def test_second_fold_kernel(shape, epi, precision):
    from nnaudio_amd import engine

    B, L, F, K, hop, pad, mode, window = shape
    rng = np.random.default_rng(F * 1000 + K + hop)
    x = rng.standard_normal((B, L)).astype(np.float32)
    wr, wi = _dft_basis(F, K, window, rng)
    re, im = _np_framed(x, wr, wi, hop, pad, mode)
    xd, wrd, wid = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi))
    prep = engine.prepare_basis(wrd, wid, precision, hop=hop)
    assert "basis_fold2" in prep, "window x DFT basis refused"
    kw = dict(hop=hop, pad=pad, pad_mode=mode)
    e = {"complex": engine.EPI_COMPLEX, "magnitude": engine.EPI_MAGNITUDE, "power2": engine.EPI_POWER,
         "phase": engine.EPI_PHASE_ATAN2}[epi]
    y = engine.framed_gemm(xd, wrd, wid, precision=precision, epilogue=e, **kw, **prep)
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    what = "fold2 %s %s" % (shape, precision)
    if epi == "complex":
        ref = np.stack((re, im), -1)
        assert_parity(y, ref, rel=1e-4, what=what)
        err = np.abs(y - ref).max() / np.abs(ref).max()
        assert err <= BUDGET[precision], "%s: %.2e of the peak" % (what, err)
    elif epi == "magnitude":
        assert_parity(y, np.sqrt(re * re + im * im), rel=1e-4, what=what)
    elif epi == "power2":
        assert_parity(y, re * re + im * im, rel=1e-4, what=what)
    else:
        mag = np.sqrt(re * re + im * im)
        assert_phase_parity(y, np.arctan2(im, re), mag, what=what)


def test_second_fold_is_what_runs():
    """The quarter-folded planes are used (result differs in the last bits from the single fold's) and
    bases that are not window x DFT are refused."""
    from nnaudio_amd import engine, features

    rng = np.random.default_rng(1)
    wr, wi = _dft_basis(257, 512, "hann", rng)
    wrd, wid = torch.as_tensor(wr).to(DEV), torch.as_tensor(wi).to(DEV)
    x = torch.as_tensor(rng.standard_normal((2, 8000)).astype(np.float32)).to(DEV)
    prep = engine.prepare_basis(wrd, wid, "fp32", hop=128)
    assert "basis_fold2" in prep and "basis_fold" in prep
    kw = dict(hop=128, pad=256, pad_mode=2, epilogue=engine.EPI_COMPLEX, precision="fp32")
    y2 = engine.framed_gemm(x, wrd, wid, **kw, **prep)
    y1 = engine.framed_gemm(x, wrd, wid, **kw, basis_fold=prep["basis_fold"])
    assert not torch.equal(y1, y2)
    assert (y1 - y2).abs().max() <= 3e-6 * y1.abs().max()
    # a scaled copy of one row breaks window x DFT: refused
    bad = wrd.clone()
    bad[7] *= 1.001
    assert engine.fold2_basis(bad, wid, "fp32") is None
    # the reference's other frequency scales are not DFT rows
    m = features.STFT(n_fft=512, freq_scale="log", sr=22050, fmin=50, fmax=6000, verbose=False).to(DEV)
    assert engine.fold2_basis(m.wcos, m.wsin, "bf16x3") is None
    m = features.STFT(n_fft=512, verbose=False).to(DEV)
    assert engine.fold2_basis(m.wcos, m.wsin, "bf16x3") is not None


@pytest.mark.parametrize("gain", [1.0, 1e-4, 3e4, 1e-15])
def test_f16x3_dynamic_range(gain):
    """"f16x3": a sine exactly on a bin leaves the other bins ~silent; what the arithmetic leaves there,
    relative to the peak, is its dynamic range.  Measured on the MI355X (experiments/split_arith):
    fp32 -173 dB, f16x3 -163 dB, bf16x3 -128 dB.  The operand scaling must make this independent
    of the input's level (fp16 has 5 exponent bits)."""
    from nnaudio_amd import engine, features

    N, hop = 2048, 512
    m = features.STFT(n_fft=N, hop_length=hop, output_format="Complex", verbose=False).to(DEV)
    n = np.arange(40 * hop)
    x = (gain * np.cos(2 * np.pi * 400 * n / N + 0.3)).astype(np.float32)[None, :]
    xd = torch.as_tensor(x).to(DEV)
    floors = {}
    for precision in ("fp32", "f16x3", "bf16x3"):
        m.precision = precision
        y = m(xd).cpu().numpy().astype(np.float64)[0, :, 4:-4]  # interior frames: a pure tone per frame
        mag = np.hypot(y[..., 0], y[..., 1])
        peak = mag.max()
        silent = np.r_[mag[:380], mag[420:]]  # hann side lobes are far below every arithmetic's floor there
        floors[precision] = 20 * np.log10(silent.max() / peak + 1e-300)
        assert np.isfinite(y).all()
    print("dynamic range, gain %g: %s" % (gain, {k: round(v, 1) for k, v in floors.items()}))
    assert floors["f16x3"] <= -145.0, floors
    assert floors["bf16x3"] <= -110.0, floors
    assert floors["f16x3"] <= floors["bf16x3"] - 25.0, floors


def test_f16x3_levels_within_a_clip():
    """A clip whose second half is 80 dB below its first: the scale is chosen per workgroup of frames,
    so the quiet frames keep their relative accuracy."""
    from nnaudio_amd import features

    rng = np.random.default_rng(5)
    N, hop = 1024, 256
    x = rng.standard_normal((1, 60 * hop)).astype(np.float32)
    x[:, 30 * hop:] *= 1e-4
    m = features.STFT(n_fft=N, hop_length=hop, output_format="Complex", verbose=False).to(DEV)
    xd = torch.as_tensor(x).to(DEV)
    m.precision = "fp32"
    ref = m(xd).cpu().numpy().astype(np.float64)
    m.precision = "f16x3"
    y = m(xd).cpu().numpy().astype(np.float64)
    quiet = slice(36, None)  # frames that only see the quiet half
    err = np.abs(y[:, :, quiet] - ref[:, :, quiet]).max() / np.abs(ref[:, :, quiet]).max()
    assert err <= 5e-6, err
    err = np.abs(y - ref).max() / np.abs(ref).max()
    assert err <= 5e-6, err



@pytest.mark.parametrize("cls,ctor", [
    ("STFT", dict(n_fft=1024, hop_length=256, output_format="Complex")),
    ("MelSpectrogram", dict(sr=22050, n_fft=1024, n_mels=64, hop_length=256)),
    ("CQT1992v2", dict(sr=22050, hop_length=256, n_bins=72, output_format="Complex")),
    ("CQT2010v2", dict(sr=22050, hop_length=256, n_bins=72, output_format="Complex", earlydownsample=False)),
])
def test_f16x3_extreme_levels(cls, ctor):
    """The operand scaling at the ends of float32: an all-zero clip, a clip at 1e-30, one at 1e+33 (1e-15 /
    1e+15 where the output is a power) and one with a single full-scale click in digital silence, in one batch -- f16x3 must stay finite and
    agree with the fp32 path relative to each clip's own peak (fp16's 5 exponent bits never show)."""
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(11)
    x = torch.randn(5, 30000, generator=g)
    lo, hi = (1e-15, 1e15) if cls == "MelSpectrogram" else (1e-30, 1e33)  # (a power spectrum squares them)
    x[0] = 0.0
    x[1] *= lo
    x[2] *= hi
    x[3] = 0.0
    x[3, 15000] = 1.0
    m = getattr(features, cls)(verbose=False, **ctor).to(DEV)
    target = m.stft if hasattr(m, "stft") else m
    xd = x.to(DEV)
    target.precision = "fp32"
    ref = m(xd).double()
    target.precision = "f16x3"
    y = m(xd).double()
    assert torch.isfinite(y).all()
    assert float(y[0].abs().max()) == 0.0 and float(ref[0].abs().max()) == 0.0
    for b in (1, 2, 3, 4):
        peak = ref[b].abs().max()
        assert peak > 0
        err = float((y[b] - ref[b]).abs().max() / peak)
        # (power spectra square the scale: 1e-30 underflows to 0 in fp32 too -- compare what is there)
        assert err <= 2e-5, (cls, b, err)
