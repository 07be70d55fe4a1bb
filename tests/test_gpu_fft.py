"""GPU parity of the STFT's FFT path (csrc/stft_fft.inl: window x DFT bases of 256 / 512 / 1024 / 2048 taps evaluated
as an fp32 FFT instead of being contracted with the kernels) against the float64 oracle and against the
contraction kernels -- through the modules and through the C ABI (engine.framed_gemm)."""
import numpy as np
import pytest
import torch

from tests import _golden
from tests import test_gpu_parity as _contraction_suite
from tests._golden import assert_parity, assert_phase_parity
from tests.test_gpu_parity import DEV, _np_framed
from tests.test_gpu_fold2 import _dft_basis

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; torch.cuda.is_available() is False")
    from nnaudio_amd import _abi

    _abi.load()  # fail loudly if the extension is not built


# (tests/conftest.py switches the FFT path ON for this module and off for the others)
@pytest.mark.parametrize("shape", [  # (B, L, bins, K, hop, pad, mode, window)
    (2, 30000, 1025, 2048, 512, 1024, 2, "hann"),     # the cfg2 basis: tiles of 16 frames, edge frames at both ends
    (3, 9000, 513, 1024, 256, 512, 2, "hann"),        # cfg3's: 32 frames per tile
    (5, 4000, 257, 512, 128, 256, 1, "hamming"),      # 64 frames per tile, zero padding, w[0] != 0
    (1, 70001, 700, 2048, 333, 1024, 2, "random"),    # odd hop (frames at odd addresses), freq_bins < n_fft/2, asymmetric window
    (2, 5000, 129, 512, 64, 0, 0, "hann"),            # center=False, a quarter of the bins
    (1, 2100, 1025, 2048, 2048, 1024, 2, "short"),    # two frames, both edge frames; win_length < n_fft
    (7, 1500, 513, 1024, 1024, 512, 1, "hann"),       # many short clips, zero padding
    (3, 6000, 129, 256, 64, 128, 2, "hann"),          # n_fft = 256 (the reference's grid): zero-extended frames on the 512-point instance
    (2, 5000, 100, 256, 100, 0, 0, "random"),         # ... center=False, fewer bins, asymmetric window
    (4, 300, 129, 256, 200, 128, 2, "hann"),          # clips shorter than the 512 samples a zero-extended frame reads
    (2, 40000, 2049, 4096, 1024, 2048, 2, "hann"),    # n_fft = 4096 (round 5): the composite instance -- two 2048-point halves + a butterfly
    (1, 30011, 1500, 4096, 600, 2048, 1, "random"),   # ... odd clip length, zero padding, hop not a divisor, freq_bins, asymmetric window
    (3, 9000, 2049, 4096, 4096, 0, 0, "hamming"),     # ... center=False, two frames per clip
    (2, 30000, 2049, 4096, 1024, 2048, 2, "short"),   # ... even clip length under reflect padding (the de-interleave does not commute with it)
])
@pytest.mark.parametrize("epi", ["complex", "magnitude", "power2", "power1", "phase", "cossin"])
def test_fft_path_against_float64(shape, epi):
    from nnaudio_amd import engine

    B, L, F, K, hop, pad, mode, window = shape
    rng = np.random.default_rng(F * 1000 + K + hop)
    x = rng.standard_normal((B, L)).astype(np.float32)
    wr, wi = _dft_basis(F, K, window, rng)
    re, im = _np_framed(x, wr, wi, hop, pad, mode)
    xd, wrd, wid = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi))
    prep = engine.prepare_basis(wrd, wid, "fp32", hop=hop)
    assert "basis_fold2" in prep, "window x DFT basis refused"
    kw = dict(hop=hop, pad=pad, pad_mode=mode, precision="fp32")
    e, extra = {"complex": (engine.EPI_COMPLEX, {}), "magnitude": (engine.EPI_MAGNITUDE, {}),
                "power2": (engine.EPI_POWER, dict(power=2.0)), "power1": (engine.EPI_POWER, dict(power=1.0, eps=1e-8)),
                "phase": (engine.EPI_PHASE_ATAN2, {}), "cossin": (engine.EPI_PHASE_COSSIN, {})}[epi]
    y = engine.framed_gemm(xd, wrd, wid, epilogue=e, **kw, **extra, **prep)
    y_gemm = engine.framed_gemm(xd, wrd, wid, epilogue=e, fft=False, **kw, **extra, **prep)
    if K == 4096 and epi == "cossin":   # the one epilogue the composite instance refuses (mispec.hip plan_fft4096): contraction kernels
        assert torch.equal(y, y_gemm)
    else:
        assert y.shape == y_gemm.shape and not torch.equal(y, y_gemm), "the contraction kernels ran"
    y = y.cpu().numpy()
    what = "fft %s %s" % (shape, epi)
    mag = np.sqrt(re * re + im * im)
    if epi == "complex":
        ref = np.stack((re, im), -1)
        assert_parity(y, ref, rel=1e-4, what=what)
        err = np.abs(y - ref).max() / np.abs(ref).max()
        assert err <= 2e-6, "%s: %.2e of the peak" % (what, err)
    elif epi == "magnitude":
        assert_parity(y, mag, rel=1e-4, what=what)
        assert np.abs(y - mag).max() <= 2e-6 * mag.max()
    elif epi == "power2":
        assert_parity(y, mag * mag, rel=1e-4, what=what)
    elif epi == "power1":
        assert_parity(y, np.sqrt(mag * mag + 1e-8), rel=1e-4, what=what)
    elif epi == "phase":
        assert_phase_parity(y, np.arctan2(im, re), mag, what=what)
    else:
        ang = np.arctan2(im, re)
        assert_phase_parity(y, np.stack((np.cos(ang), np.sin(ang)), -1), mag, what=what)


def test_fft_path_is_what_the_modules_run_and_can_be_switched_off():
    """STFT's default route for freq_scale='no': the FFT (every precision: it is fp32 arithmetic); set_fft(False)
    returns to the contraction kernels; bases that are not window x DFT never take it."""
    import nnaudio_amd
    from nnaudio_amd import features

    x = torch.as_tensor(np.random.default_rng(0).standard_normal((2, 20000)).astype(np.float32)).to(DEV)
    m = features.STFT(n_fft=1024, hop_length=256, output_format="Complex", verbose=False).to(DEV)
    ys = {}
    for prec in ("fp32", "f16x3", "bf16x3"):
        m.precision = prec
        ys[prec] = m(x)
    assert torch.equal(ys["fp32"], ys["f16x3"]) and torch.equal(ys["fp32"], ys["bf16x3"])
    old = nnaudio_amd.set_fft(False)
    try:
        assert old is True
        m.precision = "fp32"
        g = m(x)
        m.precision = "bf16x3"
        gb = m(x)
    finally:
        nnaudio_amd.set_fft(True)
    assert not torch.equal(g, ys["fp32"]) and not torch.equal(g, gb)
    assert float((g - ys["fp32"]).abs().max() / g.abs().max()) <= 2e-6
    # a log-frequency basis is not window x DFT: same result with the switch on or off
    lg = features.STFT(n_fft=1024, hop_length=256, freq_scale="log", fmin=50, fmax=6000, sr=22050,
                       output_format="Complex", verbose=False).to(DEV)
    a = lg(x)
    nnaudio_amd.set_fft(False)
    try:
        b = lg(x)
    finally:
        nnaudio_amd.set_fft(True)
    assert torch.equal(a, b)
    # n_fft = 4096 (round 5): the composite instance of the FFT route -- fp32-class agreement with the contraction; an odd hop
    # (the de-interleave needs an even one) and n_fft = 8192 stay on the contraction kernels, switch or not
    big = features.STFT(n_fft=4096, hop_length=1024, output_format="Magnitude", verbose=False).to(DEV)
    a = big(x)
    nnaudio_amd.set_fft(False)
    try:
        b = big(x)
    finally:
        nnaudio_amd.set_fft(True)
    assert not torch.equal(a, b) and float((a - b).abs().max() / b.abs().max()) <= 3e-6
    for kw in (dict(n_fft=4096, hop_length=1023), dict(n_fft=8192, hop_length=2048)):
        other = features.STFT(output_format="Magnitude", verbose=False, **kw).to(DEV)
        a = other(x)
        nnaudio_amd.set_fft(False)
        try:
            b = other(x)
        finally:
            nnaudio_amd.set_fft(True)
        assert torch.equal(a, b), kw


def test_n_fft_256_on_the_fft_route_ignores_what_lies_behind_the_frame():
    """n_fft = 256 (tests/parameters.py:25 of the reference) runs on the 512-point instance with zero-extended
    frames: the 256 samples that follow a frame are read but are not part of it -- an Inf there must not reach it."""
    from nnaudio_amd import engine, features

    m = features.STFT(n_fft=256, hop_length=64, center=False, output_format="Magnitude", verbose=False).to(DEV)
    x = torch.as_tensor(np.random.default_rng(4).standard_normal((2, 4000)).astype(np.float32)).to(DEV)
    y = m(x)
    engine.set_fft(False)
    try:
        g = m(x)
    finally:
        engine.set_fft(True)
    assert y.shape == g.shape == (2, 129, (4000 - 256) // 64 + 1)
    assert not torch.equal(y, g), "the contraction kernels ran"
    assert float((y - g).abs().max()) <= 5e-6 * float(g.abs().max())
    x2 = x.clone()
    x2[0, 1000] = float("inf")
    y2 = m(x2)
    clean = (1000 - 256) // 64 + 1  # frames that end at or before sample 1000
    assert torch.equal(y2[0, :, :clean], y[0, :, :clean]) and torch.equal(y2[1], y[1])
    assert not torch.isfinite(y2[0, :, clean]).all()


def test_fft_path_pure_tone_dynamic_range():
    """A sine exactly on a bin: what the other bins hold, relative to the peak (the FFT's rounding error grows
    with log n_fft, not with n_fft)."""
    from nnaudio_amd import features

    n = np.arange(44100)
    x = np.sin(2 * np.pi * 64 * n / 2048.0).astype(np.float32)[None]
    m = features.STFT(n_fft=2048, hop_length=512, window="hann", center=False, output_format="Magnitude",
                      verbose=False).to(DEV)
    y = m(torch.as_tensor(x).to(DEV)).cpu().numpy()[0]
    peak = y.max()
    silent = np.delete(y, [62, 63, 64, 65, 66], axis=0).max()  # (hann: the two neighbours carry half the peak)
    floor_db = 20 * np.log10(max(silent, 1e-30) / peak)
    print("fft path: silent bins %.1f dB below the peak" % floor_db)
    assert floor_db <= -140.0, floor_db


def test_cfg2_stft_fft_sampled():
    """BASELINE cfg2 (64 clips of 10 s at 44.1 kHz, n_fft 2048, hop 512) through the FFT path: sampled clips
    and frames against float64."""
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    rng = np.random.default_rng(11)
    x = rng.standard_normal((64, 441000)).astype(np.float32)
    m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(DEV)
    y = m(torch.as_tensor(x).to(DEV))
    assert y.shape == (64, 1025, 862)
    wsin, wcos = m.wsin.cpu().numpy(), m.wcos.cpu().numpy()
    for c in (0, 17, 63):
        ref = O.stft(x[c:c + 1], wsin, wcos, 512, output_format="Magnitude")[0]
        got = y[c].cpu().numpy()
        for t in (0, 1, 2, 15, 16, 431, 859, 860, 861):
            assert np.abs(got[:, t] - ref[:, t]).max() <= 2e-6 * ref.max(), (c, t)
    assert bool(torch.isfinite(y).all())


# ---------------------------------------------------------------------------------------
# the module-level checks of tests/test_gpu_parity.py once more, on the FFT route
# ---------------------------------------------------------------------------------------
_STFT_FAMILY = ("STFT", "iSTFT", "MelSpectrogram", "MFCC", "Gammatonegram")


def _family_cases():
    import json
    import os

    with open(os.path.join(_golden.GOLDEN, "cases.json")) as f:
        cases = json.load(f)["cases"]
    return [c["name"] for c in cases if c["input"] is not None and c["cls"] in _STFT_FAMILY]


@pytest.mark.parametrize("name", _family_cases())
def test_golden_case_on_the_fft_route(golden, name):
    """every reference-generated fixture of the STFT family (and the oracle) with the FFT path on"""
    from nnaudio_amd import engine

    assert engine.fft_enabled()
    _contraction_suite.test_case_matches_reference_and_oracle(golden, name)


def test_cfg3_mel_on_the_fft_route():
    """BASELINE cfg3 (256 clips of 5 s, n_fft 1024, 128 mels): the filterbank reduced in the FFT kernel's tile"""
    _contraction_suite.test_cfg3_mel_full_size_sampled()


def test_fused_filterbank_on_the_fft_route_matches_two_kernels():
    """MelSpectrogram / Gammatonegram: fused (tile reduced over the bands inside stft_fft_kernel) == power
    spectrogram through the FFT path + the filterbank kernel, and the contraction route agrees."""
    from nnaudio_amd import engine, features

    x = torch.as_tensor(np.random.default_rng(2).standard_normal((5, 30000)).astype(np.float32)).to(DEV)
    for cls, kw in ((features.MelSpectrogram, dict(sr=22050, n_fft=1024, n_mels=128, hop_length=512)),
                    (features.MelSpectrogram, dict(sr=16000, n_fft=512, n_mels=40, hop_length=160, power=1.0)),
                    (features.MelSpectrogram, dict(sr=44100, n_fft=2048, n_mels=229, hop_length=512)),
                    (features.MelSpectrogram, dict(sr=8000, n_fft=256, n_mels=40, hop_length=80)),  # zero-extended frames
                    (features.Gammatonegram, dict(sr=22050, n_fft=1024, n_bins=64, hop_length=512))):
        m = cls(verbose=False, **kw).to(DEV)
        y = m(x)
        basis = getattr(m, m._basis_name)
        spec = m.stft._spectrum(x, engine.EPI_POWER, power=m.power)
        two = engine.filterbank(basis, spec)
        assert float((y - two).abs().max() / two.abs().max()) <= 2e-6, cls.__name__
        engine.set_fft(False)
        try:
            g = m(x)
        finally:
            engine.set_fft(True)
        assert not torch.equal(g, y)
        assert float((y - g).abs().max() / g.abs().max()) <= 1e-5, cls.__name__


@pytest.mark.parametrize("kw,shape,fused", [
    (dict(sr=22050, n_mfcc=20, n_fft=1024, n_mels=128, hop_length=512), (5, 30000), True),
    (dict(sr=16000, n_mfcc=13, n_fft=512, n_mels=40, hop_length=160, top_db=None), (3, 16000), True),
    (dict(sr=44100, n_mfcc=40, n_fft=2048, n_mels=229, hop_length=512, top_db=30.0), (2, 50000), True),
    (dict(sr=22050, n_mfcc=128, n_fft=1024, n_mels=128, hop_length=256), (2, 20000), True),   # as many coefficients as bands
    (dict(sr=44100, n_mfcc=30, n_fft=2048, n_mels=300, hop_length=512), (2, 30000), False),    # > 256 bands: the two calls
])
def test_mfcc_tail_in_one_launch(kw, shape, fused):
    """MFCC without a gradient to record: power_to_db + DCT as one launch (mispec_mfcc_tail_f32) == the two calls."""
    from nnaudio_amd import engine, features

    m = features.MFCC(verbose=False, **kw).to(DEV)
    x = torch.as_tensor(np.random.default_rng(7).standard_normal(shape).astype(np.float32)).to(DEV)
    x[0, : shape[1] // 3] *= 1e-4  # (a quiet stretch: the top_db floor is active)
    with torch.no_grad():
        y = m(x)
        old = engine.set_mfcc_fused(False)
        try:
            r = m(x)
        finally:
            engine.set_mfcc_fused(old)
    assert y.shape == r.shape
    assert torch.equal(y, r) != fused, "fused launch %staken" % ("not " if fused else "")
    assert float((y - r).abs().max()) <= 2e-6 * float(r.abs().max())
    # with a gradient to record the two calls run (and agree)
    xg = x.clone().requires_grad_(True)
    yg = m(xg)
    assert float((yg.detach() - r).abs().max()) <= 2e-6 * float(r.abs().max())


@pytest.mark.parametrize("fmt", ["Magnitude", "Complex"])
def test_backward_through_the_fft_route(fmt):
    """d loss / d waveform of a frozen STFT whose forward ran on the FFT path (the backward contracts with the
    kernels): against torch autograd on the conv1d restatement."""
    from nnaudio_amd import features

    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 6000, generator=g).to(DEV).requires_grad_(True)
    m = features.STFT(n_fft=512, hop_length=160, output_format=fmt, verbose=False).to(DEV)
    y = m(x)
    with torch.no_grad():
        assert torch.equal(y.detach(), m(x.detach()))  # the FFT forward, with or without a graph
    w = torch.randn(y.shape, generator=g).to(DEV)
    (y * w).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    re, im = _contraction_suite._torch_framed(x2, m.wcos.detach(), m.wsin.detach(), 160, 256, "reflect")
    y2 = torch.sqrt(re ** 2 + im ** 2) if fmt == "Magnitude" else torch.stack((re, -im), -1)
    assert (y - y2).abs().max().item() <= 1e-4 * max(1.0, y2.abs().max().item())
    (y2 * w).sum().backward()
    _contraction_suite._grad_close(x.grad, x2.grad, "d x")


def test_mel_backward_on_the_fft_route():
    _contraction_suite.test_backward_mel_and_frozen_front_end(True)


@pytest.mark.parametrize("cls,ctor", [("STFT", dict(n_fft=512, hop_length=128, output_format="Magnitude")),
                                      ("MelSpectrogram", dict(sr=16000, n_fft=512, n_mels=40, hop_length=160))])
def test_torch_compile_on_the_fft_route(cls, ctor):
    """compiled == eager, bit for bit, with the custom ops routing to the FFT kernel"""
    import warnings

    from nnaudio_amd import features

    m = getattr(features, cls)(verbose=False, **ctor).to(DEV)
    x = torch.as_tensor(np.random.default_rng(4).standard_normal((3, 16000)).astype(np.float32)).to(DEV)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eager = m(x)
        compiled = torch.compile(m, fullgraph=True)(x)
    torch._dynamo.reset()
    assert torch.equal(eager, compiled)


# ---------------------------------------------------------------------------------------
# inverse direction: frame synthesis of the inverse STFT as an inverse real FFT
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_fft,hop,B,L,center", [(2048, 512, 3, 40000, True), (1024, 256, 2, 9000, True),
                                                  (512, 128, 4, 5001, True), (2048, 1024, 1, 2100, True),
                                                  (1024, 512, 2, 8192, False), (512, 37, 1, 3000, True)])
def test_inverse_fft_matches_the_contraction_and_inverts(n_fft, hop, B, L, center):
    from nnaudio_amd import engine, features

    m = features.STFT(n_fft=n_fft, hop_length=hop, iSTFT=True, center=center, output_format="Complex",
                      verbose=False).to(DEV)
    x = torch.as_tensor(np.random.default_rng(n_fft + hop).standard_normal((B, L)).astype(np.float32)).to(DEV)
    X = m(x)
    y = m.inverse(X, length=L if center else None)
    engine.set_fft(False)
    try:
        r = m.inverse(X, length=L if center else None)
    finally:
        engine.set_fft(True)
    assert y.shape == r.shape and not torch.equal(y, r), "the contraction ran"
    if not center:  # (untrimmed ends: the division by a vanishing window sum amplifies either route's rounding)
        y, r = y[:, n_fft // 2:-n_fft // 2], r[:, n_fft // 2:-n_fft // 2]
    assert float((y - r).abs().max() / r.abs().max()) <= 3e-6
    if center and 4 * hop <= n_fft:  # (COLA: the round trip is the identity)
        assert float((y - x[:, :y.shape[1]]).abs().max()) <= 5e-6 * float(x.abs().max())
    # the iSTFT class on a one-sided spectrogram takes the same route
    im = features.iSTFT(n_fft=n_fft, hop_length=hop, center=center, verbose=False).to(DEV)
    y2 = im(X, onesided=True, length=L if center else None)
    if not center:
        y2 = y2[:, n_fft // 2:-n_fft // 2]
    assert float((y2.float() - r).abs().max() / r.abs().max()) <= 3e-6


def test_inverse_fft_is_not_taken_for_other_kernels():
    """two-sided spectrograms, trainable or non-DFT synthesis kernels and n_fft = 4096 stay on the contraction"""
    from nnaudio_amd import engine, features

    m = features.STFT(n_fft=512, hop_length=128, iSTFT=True, output_format="Complex", verbose=False).to(DEV)
    x = torch.randn(2, 6000, generator=torch.Generator().manual_seed(3)).to(DEV)
    X = m(x)
    full = torch.cat((X, torch.stack((X[:, 1:-1, :, 0].flip(1), -X[:, 1:-1, :, 1].flip(1)), -1)), 1)  # extend_fbins
    a = m.inverse(full, onesided=False, length=6000)
    engine.set_fft(False)
    try:
        b = m.inverse(full, onesided=False, length=6000)
    finally:
        engine.set_fft(True)
    assert torch.equal(a, b)
    basis = engine.istft_basis(m.kernel_cos_inv, m.kernel_sin_inv, 257, True)
    assert engine.istft_basis_is_dft(basis, 257)
    bad = basis.clone()
    bad[5, 7] *= 1.001
    assert not engine.istft_basis_is_dft(bad, 257)
    big = features.STFT(n_fft=4096, hop_length=1024, iSTFT=True, output_format="Complex", verbose=False).to(DEV)
    bb = engine.istft_basis(big.kernel_cos_inv, big.kernel_sin_inv, 2049, True)
    assert not engine.istft_basis_is_dft(bb, 2049)


@pytest.mark.parametrize("name", ["stft", "mel", "mfcc", "gammatone"])
def test_fft_route_is_hip_graph_capturable(name):
    """the contraction suite's capture test with the FFT path on: one kernel per forward, no workspace, no memset"""
    _contraction_suite.test_forward_is_hip_graph_capturable(None, name)


def test_repeated_launches_on_the_fft_route_are_bit_identical():
    """hand-stated waits, wave-level LDS hand-offs and row swaps: 20 launches on fresh inputs of one shape must
    reproduce each other exactly (a missing wait shows up as run-to-run differences)"""
    from nnaudio_amd import features

    m = features.STFT(n_fft=2048, hop_length=512, output_format="Complex", verbose=False).to(DEV)
    mm = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(DEV)
    x = torch.randn(16, 120000, device=DEV, generator=torch.Generator(DEV).manual_seed(9))
    ref, refm = m(x).clone(), mm(x).clone()
    for _ in range(20):
        torch.randn(16, 120000, device=DEV)  # (disturb the allocator / caches)
        assert torch.equal(m(x), ref) and torch.equal(mm(x), refm)


def test_fft_route_random_shapes_against_the_contraction():
    """60 random STFT problems (n_fft, hop, clip length, batch, padding, freq_bins, strided clips, epilogue incl. a
    general power) through both routes: the FFT path must agree with the fp32 contraction to fp32 rounding"""
    from nnaudio_amd import engine

    rng = np.random.default_rng(2024)
    taken = 0
    for case in range(60):
        K = int(rng.choice([256, 512, 1024, 2048]))
        hop = int(rng.choice([1, 2, 3]) * rng.integers(20, 1200)) if case % 5 else int(rng.integers(K, 2 * K))
        center = bool(rng.integers(0, 4))
        pad = K // 2 if center else 0
        mode = int(rng.choice([1, 2])) if center else 0
        B = int(rng.integers(1, 6))
        T = int(rng.integers(1, 70))
        L = max((T - 1) * hop + K - 2 * pad + int(rng.integers(0, hop)), pad + 2, K - 2 * pad)
        F = int(rng.choice([K // 2 + 1, K // 2 + 1, int(rng.integers(1, K // 2 + 1))]))
        stride = L + int(rng.choice([0, 0, 3, 64]))
        xs = torch.as_tensor(rng.standard_normal((B, stride)).astype(np.float32)).to(DEV)
        x = xs[:, :L]
        window = rng.choice(["hann", "hamming", "random"])
        wr, wi = _dft_basis(K // 2 + 1, K, window, rng)
        wrd, wid = torch.as_tensor(wr).to(DEV), torch.as_tensor(wi).to(DEV)
        prep = engine.prepare_basis(wrd, wid, "fp32", hop=None)
        if "basis_fold2" not in prep:
            continue
        wrd, wid = wrd[:F], wid[:F]
        prep = {"basis_fold2": (prep["basis_fold2"][0], prep["basis_fold2"][1])}
        epi, extra = [(engine.EPI_COMPLEX, {}), (engine.EPI_MAGNITUDE, {}), (engine.EPI_MAGNITUDE, dict(eps=1e-8)),
                      (engine.EPI_POWER, dict(power=2.0)), (engine.EPI_POWER, dict(power=3.5, eps=1e-8)),
                      (engine.EPI_PHASE_COSSIN, {})][int(rng.integers(0, 6))]
        kw = dict(hop=hop, pad=pad, pad_mode=mode, precision="fp32", epilogue=epi, **extra)
        try:
            y = engine.framed_gemm(x, wrd, wid, fft=True, **kw, **prep)
        except RuntimeError:
            continue  # (the quarter-folded planes belong to the full basis: a sliced basis may be refused)
        r = engine.framed_gemm(x, wrd, wid, fft=False, **kw)
        assert y.shape == r.shape
        what = "case %d: K %d hop %d B %d L %d F %d pad %d/%d epi %d %s" % (case, K, hop, B, L, F, pad, mode, epi, extra)
        if epi == engine.EPI_PHASE_COSSIN:
            z = engine.framed_gemm(x, wrd, wid, fft=False, **dict(kw, epilogue=engine.EPI_MAGNITUDE))
            strong = z > 0.05 * z.max()
            if bool(strong.any()):
                assert float((y - r)[strong].abs().max()) <= 2e-4, what
        else:
            tol = 5e-6 * max(1.0, float(extra.get("power", 1.0)))  # (|X|^p carries p times the relative error of |X|)
            assert float((y - r).abs().max()) <= tol * float(r.abs().max()), what
        taken += not torch.equal(y, r)
    assert taken >= 30, taken


def test_layouts_streams_and_errors_on_the_fft_route(golden):
    """the contraction suite's input-layout / side-stream / error-behaviour checks with the FFT path on"""
    _contraction_suite.test_input_layouts_and_streams(golden)
    _contraction_suite.test_error_behaviour_on_device()
    _contraction_suite.test_phase_zero_input_matches_reference_convention()


@pytest.mark.parametrize("shape", [  # (n_fft, hop, B, T, start, out_len or None = to the end)
    (2048, 512, 3, 79, 1024, 40000),     # centred, T not a multiple of 8: runs of several tiles
    (2048, 512, 300, 13, 1024, 6000),    # many short clips: one run per clip
    (1024, 256, 2, 36, 512, 9000),
    (512, 128, 4, 40, 256, 5001),
    (2048, 64, 1, 700, 1024, 44000),     # hop 64: 31 earlier frames reach into a sample, four warm-up tiles per run
    (2048, 2048, 2, 21, 0, None),        # no overlap
    (1024, 512, 5, 7, 0, None),          # fewer frames than a tile; untrimmed ends (window sums near zero)
    (2048, 1024, 1, 300, 1024, 300000),  # one clip over many CUs
    (512, 192, 3, 50, 256, 9000),        # hop a multiple of 64 that does not divide n_fft
])
def test_fused_inverse_is_bit_identical_to_the_two_launches(shape):
    """mispec_istft_fft_f32 (inverse FFT + overlap-add in LDS, one launch) against mispec_istft_frames_fft_f32 +
    mispec_overlap_add_f32: the same summation order, so the same bits (stft.py:15-63)."""
    from nnaudio_amd import engine

    n_fft, hop, B, T, start, out_len = shape
    F = n_fft // 2 + 1
    full = (T - 1) * hop + n_fft
    out_len = full - start if out_len is None else min(out_len, full - start)
    g = torch.Generator().manual_seed(n_fft + hop + T)
    spec = torch.randn(B, F, T, 2, generator=g).to(DEV)
    win = torch.hann_window(n_fft, periodic=True).to(DEV) if n_fft != 512 else torch.rand(n_fft, generator=g).to(DEV)
    n = torch.arange(n_fft, dtype=torch.float64)[:, None]
    k = torch.arange(F, dtype=torch.float64)[None, :]
    ang = 2.0 * np.pi * ((n * k) % n_fft) / n_fft
    c = torch.full((1, F), 2.0, dtype=torch.float64)
    c[0, 0] = c[0, -1] = 1.0
    basis = torch.cat((c * torch.cos(ang), -c * torch.sin(ang)), 1).float().to(DEV)
    assert engine.istft_basis_is_dft(basis, F)
    old = engine.set_istft_fused(True)
    try:
        a = engine.istft(spec, basis, win, hop, start, out_len, dft=True)
        engine.set_istft_fused(False)
        b = engine.istft(spec, basis, win, hop, start, out_len, dft=True)
    finally:
        engine.set_istft_fused(old)
    assert a.shape == b.shape == (B, out_len)
    assert torch.equal(a, b), "max |d| = %.3e" % float((a - b).abs().max())


def test_fused_inverse_refuses_what_it_does_not_serve():
    """hop not a multiple of 64: the two launches run (same result either way)."""
    from nnaudio_amd import features

    m = features.STFT(n_fft=1024, hop_length=200, iSTFT=True, output_format="Complex", verbose=False).to(DEV)
    x = torch.randn(2, 9000, generator=torch.Generator().manual_seed(1)).to(DEV)
    y = m.inverse(m(x), length=9000)
    assert y.shape == x.shape
    # the reconstruction where the window sum is well conditioned (inside the first / last frame's taper it is not: hop = 200
    # does not divide n_fft) must be the signal: the two-launch route is checked here, not merely taken
    inner = slice(1024, 9000 - 1024)
    err = float((y[:, inner] - x[:, inner]).abs().max())
    assert err < 1e-4 * float(x.abs().max()), err


@pytest.mark.parametrize("shape", [  # (B, L, K, hop, pad, mode, window, power)
    (3, 40000, 2048, 512, 1024, 2, "hann", 2.0),      # Gammatonegram's default STFT
    (2, 30011, 2048, 333, 1024, 1, "random", 1.0),    # odd hop and clip length, zero padding, asymmetric window, power 1
    (5, 9000, 1024, 256, 512, 2, "hann", 2.0),        # the 512-point instance (two frames per wave and tile)
    (1, 3000, 1024, 1024, 0, 0, "hamming", 3.5),      # center=False, two frames
])
def test_frame_major_output_is_the_spectrogram_transposed(shape):
    """mispec.h out_frame_major: the same power spectrogram, bit for bit, as (B, T, Fp) rows with zeros behind the bins."""
    from nnaudio_amd import engine

    B, L, K, hop, pad, mode, window, power = shape
    F = K // 2 + 1
    rng = np.random.default_rng(K + hop)
    x = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).to(DEV)
    wr, wi = (torch.as_tensor(a).to(DEV) for a in _dft_basis(F, K, window, rng))
    prep = engine.prepare_basis(wr, wi, "fp32", hop=hop)
    kw = dict(hop=hop, pad=pad, pad_mode=mode, precision="fp32", epilogue=engine.EPI_POWER, power=power, eps=1e-8 if power != 2.0 else 0.0)
    ref = engine.framed_gemm(x, wr, wi, **kw, **prep)
    for Fp in ((F + 31) // 32 * 32, F, F + 63):
        torch.full((B * ref.shape[2] * Fp,), float("nan"), device=DEV)  # (dirty the allocator's blocks: the zeros must be written)
        y = engine.framed_gemm(x, wr, wi, out_frame_major=Fp, **kw, **prep)
        assert y.shape == (B, ref.shape[2], Fp)
        assert torch.equal(y[:, :, :F].transpose(1, 2), ref), "Fp = %d" % Fp
        assert not y[:, :, F:].any()
    # what the mode does not serve is refused, loudly
    for bad in (dict(epilogue=engine.EPI_MAGNITUDE), dict(fft=False)):
        with pytest.raises(RuntimeError, match="out_frame_major"):
            engine.framed_gemm(x, wr, wi, out_frame_major=F, **dict(kw, **bad), **prep)
    with pytest.raises(RuntimeError, match="out_frame_major"):
        engine.framed_gemm(x, wr, wi, out_frame_major=F + 64, **kw, **prep)


def test_gammatonegram_runs_frame_major_and_matches_the_two_kernels():
    """Gammatonegram's dense filterbank (gammatone.py:184-189): FFT-route power spectrogram written frame-major + one framed
    contraction over the bins, against the (bins x frames) spectrogram + planar filterbank kernel it replaces."""
    from nnaudio_amd import engine, features

    x = torch.randn(3, 50000, device=DEV)
    for kw in (dict(sr=44100, n_fft=2048, n_bins=64, hop_length=512), dict(sr=22050, n_fft=1024, n_bins=96, hop_length=300, power=1.0)):
        m = features.Gammatonegram(verbose=False, **kw).to(DEV)
        with torch.no_grad():
            assert engine.frame_major_filterbank_plan(m, m.gammatone_basis, x, m.stft) is not None
            y = m(x)
            spec = m.stft._spectrum(x, engine.EPI_POWER, power=m.power)
            two = engine.filterbank_autograd(m.gammatone_basis, spec)
        assert y.shape == two.shape
        err = float((y - two).abs().max() / two.abs().max())
        assert err <= 2e-6, err
        ref = torch.matmul(m.gammatone_basis.double(), spec.double())
        assert float((y - ref).abs().max() / ref.abs().max()) <= 2e-6
    # a graph is needed, or the filterbank is trainable: the differentiable route
    m = features.Gammatonegram(sr=44100, n_fft=2048, n_bins=64, hop_length=512, trainable_bins=True, verbose=False).to(DEV)
    assert engine.frame_major_filterbank_plan(m, m.gammatone_basis, x, m.stft) is None
    m(x).sum().backward()
    assert m.gammatone_basis.grad is not None
