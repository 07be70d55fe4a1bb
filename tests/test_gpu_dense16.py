"""GPU parity of "f16x3" on the staged dense kernel (framed_bf16x3.inl, F16 instances): complex bases the
symmetric folds refuse -- random / non-linear STFT bases and CQT banks with supports -- against float64,
through the C ABI (engine.framed_gemm with ``basis_split = engine.split_basis_f16``)."""
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


@pytest.mark.parametrize("shape", [  # (B, L, bins, K, hop, pad, mode, supports)
    (2, 9000, 257, 512, 128, 256, 2, False),    # two 256-row tiles + 2 leftover rows in a third
    (3, 5000, 65, 200, 50, 100, 1, False),      # the smallest problem served (130 rows), K not a multiple of 32
    (1, 40000, 84, 6000, 256, 3000, 2, True),   # a CQT-like bank: supports shrinking by octaves
    (4, 3000, 150, 999, 2, 0, 0, True),         # odd K, hop 2, center=False, many frames
])
@pytest.mark.parametrize("epi", ["complex", "magnitude", "phase"])
def test_dense_f16x3_kernel(shape, epi):
    from nnaudio_amd import engine

    B, L, F, K, hop, pad, mode, masked = shape
    rng = np.random.default_rng(F * 1000 + K + hop)
    # (levels far from 1: the operand scaling must take care of them)
    x = (rng.standard_normal((B, L)) * 10.0 ** rng.uniform(-4, 3, (B, 1))).astype(np.float32)
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    g = (10.0 ** rng.uniform(-5, 2, (F, 1))).astype(np.float32)
    wr, wi = wr * g, wi * g
    sup = None
    if masked:
        ln = np.maximum(8, (K * 2.0 ** (-np.arange(F) / 12.0)).astype(np.int64))
        lo = (K - ln) // 2
        sup = np.ascontiguousarray(np.stack([lo, lo + ln], 1).astype(np.int32))
        keep = (np.arange(K)[None, :] >= sup[:, :1]) & (np.arange(K)[None, :] < sup[:, 1:])
        wr, wi = wr * keep, wi * keep
    xd, wrd, wid = (torch.as_tensor(a).to(DEV) for a in (x, wr, wi))
    kw = dict(hop=hop, pad=pad, pad_mode=mode, precision="f16x3", basis_split=engine.split_basis_f16(wrd, wid))
    if masked:
        supd = torch.as_tensor(sup).to(DEV)
        kw.update(row_support=supd, row_support_host=sup)
    e = {"complex": engine.EPI_COMPLEX, "magnitude": engine.EPI_MAGNITUDE, "phase": engine.EPI_PHASE_ATAN2}[epi]
    y = engine.framed_gemm(xd, wrd, wid, epilogue=e, **kw)
    y32 = engine.framed_gemm(xd, wrd, wid, epilogue=e, **dict(kw, precision="fp32", basis_split=None))
    assert not torch.equal(y, y32), "the fp32 kernels ran"
    y = y.cpu().numpy()
    what = "dense f16x3 %s %s" % (shape, epi)
    # per clip and bin the levels differ by orders of magnitude: compare clip by clip, bin by bin
    for b in range(B):
        re, im = _np_framed(x[b:b + 1], wr, wi, hop, pad, mode)
        re, im = re[0], im[0]
        sc = np.sqrt((re * re + im * im).max(-1, keepdims=True))  # (F, 1): the row's peak
        if epi == "complex":
            err = np.abs(y[b] - np.stack((re, im), -1)).max(-1) / sc
        elif epi == "magnitude":
            err = np.abs(y[b] - np.sqrt(re * re + im * im)) / sc
        else:
            mag = np.sqrt(re * re + im * im)
            assert_phase_parity(y[b:b + 1], np.arctan2(im, re)[None], (mag / sc)[None], what=what)
            continue
        assert err.max() <= 2e-6, "%s clip %d: %.2e of the row's peak" % (what, b, err.max())


def test_unfolded_stft_bases_run_f16x3():
    """freq_scale != 'no' (no Fourier symmetry): the default arithmetic is served by the staged dense kernel,
    not by the fp32 fallback -- and agrees with float64 to the fp32 class."""
    from nnaudio_amd import engine, features
    from oracle import spectral_oracle as O

    m = features.STFT(n_fft=512, hop_length=128, freq_scale="log", fmin=50, fmax=6000, sr=22050,
                      output_format="Complex", verbose=False).to(DEV)
    x = np.random.default_rng(3).standard_normal((3, 12000)).astype(np.float32)
    xd = torch.as_tensor(x).to(DEV)
    prep = engine.prepare_basis(m.wcos, m.wsin, "f16x3", hop=128)
    assert "basis_split" in prep and "basis_fold2" not in prep and "basis_fold" not in prep
    y = m(xd)
    m.precision = "fp32"
    y32 = m(xd)
    assert not torch.equal(y, y32)
    ref = O.stft(x, m.wsin.cpu().numpy(), m.wcos.cpu().numpy(), 128, output_format="Complex")
    assert_parity(y.cpu().numpy(), ref, rel=1e-4, what="log-frequency STFT f16x3")
    assert np.abs(y.cpu().numpy() - ref).max() <= 2e-6 * np.abs(ref).max()


def test_cqt1992v2_natural_tap_order():
    """``hop_periodic = False``: CQT1992v2 in f16x3 on the staged dense kernel; same class of accuracy as the
    strip kernel relative to the peak (the difference shows in near-silent bins: test_reference_ground_truths)."""
    from nnaudio_amd import features

    q = features.CQT1992v2(sr=16000, hop_length=64, fmin=65.4, n_bins=72, output_format="Complex", verbose=False).to(DEV)
    x = torch.as_tensor(np.random.default_rng(5).standard_normal((2, 16000)).astype(np.float32)).to(DEV)
    q.precision = "fp32"
    ref = q(x)
    q.precision = "f16x3"
    strip = q(x)
    q.hop_periodic = False
    dense = q(x)
    assert not torch.equal(strip, dense)
    for y in (strip, dense):
        assert float((y - ref).abs().max() / ref.abs().max()) <= 5e-6
