"""Pin the CPU oracle (oracle/spectral_oracle.py):
  (a) against the reference's own asserted ground-truth arrays (tests/test_cqt.py:94-262 of
      the reference: CQT1992v2 / CQT2010v2 on log and linear chirps), same tolerances;
  (b) against outputs of the reference itself on seeded inputs (tests/golden/forward.npz)."""
import numpy as np
import pytest
from scipy.signal import chirp

from tests import _golden
from tests._golden import (assert_parity, assert_phase_parity, build_module, check_ground_truth,
                            is_phase, oracle_forward)


def _chirp(method):
    fs = 44100
    s = np.linspace(0, 1, fs)
    return chirp(s, 55, 1, 22050, method=method).astype(np.float32)[None, :]


def _case(cls, fmt):
    return dict(cls=cls, ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24,
                                   output_format=fmt), fwd={})


GT_CASES = [
    (sweep, method, cls, tagc, fmt, tag)
    for sweep, method in (("log", "logarithmic"), ("linear", "linear"))
    for cls, tagc in (("CQT1992v2", "1992"), ("CQT2010v2", "2010"))
    for fmt, tag in (("Magnitude", "mag"), ("Complex", "complex"), ("Phase", "phase"))
    if not (tagc == "2010" and fmt == "Phase")  # never asserted by the reference
]


@pytest.mark.parametrize("sweep,method,cls,tagc,fmt,tag", GT_CASES)
def test_reference_ground_truths(golden, sweep, method, cls, tagc, fmt, tag):
    x = _chirp(method)
    case = _case(cls, fmt)
    y = oracle_forward(build_module(case), case, x)
    gt = golden.ground_truth("%s-sweep-cqt-%s-%s-ground-truth.npy" % (sweep, tagc, tag))
    gtc = golden.ground_truth("%s-sweep-cqt-%s-complex-ground-truth.npy" % (sweep, tagc))
    check_ground_truth(y, gt, fmt, 1e-5 if tagc == "1992" else 1e-2, gtc,
                       what="%s %s %s" % (sweep, cls, fmt))


@pytest.mark.parametrize("name", _golden.case_names(forward_only=True))
def test_oracle_matches_reference_forward(golden, name):
    case = golden.cases[name]
    x = golden.inputs[case["input"]]
    ref = golden.forward[name]
    y = oracle_forward(build_module(case), case, x)
    if _golden.is_inverse(case):
        y, ref = _golden.well_conditioned(case, build_module(case), x, y, ref)
    if is_phase(case):
        # recompute the magnitude with the oracle to mask ill-conditioned bins
        mcase = dict(case, ctor=dict(case["ctor"], output_format="Magnitude"), fwd={})
        mag = oracle_forward(build_module(mcase), mcase, x)
        assert_phase_parity(y, ref, mag, what=name)
    else:
        # reference = float32 conv1d; oracle = float64 accumulate: well inside 1e-4
        assert_parity(y, ref, rel=2e-5, what=name)


def test_vqt_gamma0_equals_cqt2010v2(golden):
    """reference tests/test_vqt.py:30-41 (bit-exact there; the oracle shares one code path)."""
    x = golden.inputs["x_1s22k"]
    c1 = dict(cls="CQT2010v2", ctor={}, fwd={})
    c2 = dict(cls="VQT", ctor=dict(gamma=0), fwd={})
    a = oracle_forward(build_module(c1), c1, x)
    b = oracle_forward(build_module(c2), c2, x)
    assert np.array_equal(a, b)


def test_oracle_error_behaviour():
    from oracle import spectral_oracle as O

    with pytest.raises(ValueError):
        O.broadcast_dim(np.zeros((1, 1, 1, 4), np.float32))
    w = np.zeros((3, 16), np.float32)
    with pytest.raises(AssertionError):
        O.stft(np.zeros((1, 4), np.float32), w, w, 4)
    with pytest.raises(ValueError):
        O.cqt1992v2(np.zeros((1, 64), np.float32), w, w, np.ones(3), 4, normalization_type="x")


@pytest.mark.parametrize("cls,ctor,L", [
    ("CQT2010v2", dict(sr=22050, fmin=55, n_bins=60), 30000),
    ("VQT", dict(sr=22050, fmin=55, n_bins=52, gamma=10), 30000),     # bottom octave cut, own widths
    ("CQT2010v2", dict(sr=16000, fmin=55, n_bins=72, pad_mode="constant"), 9000),
    ("VQT", dict(sr=16000, fmin=110, n_bins=60, gamma=0), 5000),      # reflect -> zero fallback low down
])
def test_sampled_octave_evaluation_matches_the_full_recursion(cls, ctor, L):
    """oracle.sampled_octave_complex (what the full-size GPU checks of CQT2010v2 / VQT compare
    with) against the whole-signal recursion, which is pinned to the reference above."""
    from oracle import spectral_oracle as O

    case = dict(cls=cls, ctor=dict(ctor, output_format="Complex", earlydownsample=False), fwd={})
    mod = build_module(case)
    rng = np.random.default_rng(L)
    x = rng.standard_normal((3, L)).astype(np.float32)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = oracle_forward(mod, case, x)  # (B, bins, T, 2)
    sd = {k: v.numpy() for k, v in mod.state_dict().items()}
    if cls == "VQT":
        banks = [(sd["cqt_kernels_real_%d" % i], sd["cqt_kernels_imag_%d" % i])
                 for i in range(mod.n_octaves)]
    else:
        banks = [(sd["cqt_kernels_real"], sd["cqt_kernels_imag"])] * mod.n_octaves
    T = full.shape[2]
    cb = np.array([0, 1, 2, 0, 1, 2, 0, 1])
    ct = np.array([0, 1, T - 1, T - 2, T // 2, T // 3, 2, T - 3])
    re, im = O.sampled_octave_complex(x, banks, sd["lenghts"], mod.hop_length, mod.n_bins,
                                      sd["lowpass_filter"], cb, ct,
                                      pad_mode=ctor.get("pad_mode", "reflect"))
    want = full[cb, :, ct]  # (n, bins, 2)
    peak = np.abs(want).max()
    # the full recursion keeps float32 signals between octaves, the sampled one float64
    assert np.abs(re - want[..., 0]).max() <= 2e-6 * peak
    assert np.abs(im - want[..., 1]).max() <= 2e-6 * peak
