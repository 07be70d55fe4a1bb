"""Host-side basis builders vs the reference's buffers (golden fixtures).

For every manifest case the PRODUCT module is constructed on the CPU and its state_dict is
compared with the reference module's: same keys, shapes, dtypes; values bit-identical
(sha256) or, failing that, within 1e-6 abs / 1e-6 rel of peak (different libm paths)."""
import numpy as np
import pytest

from tests import _golden
from tests._golden import build_module, sha


@pytest.mark.parametrize("name", _golden.case_names())
def test_state_dict_matches_reference(golden, name):
    case = golden.cases[name]
    mod = build_module(case)
    sd = mod.state_dict()
    assert list(sd.keys()) == list(case["state"].keys())
    assert [k for k, _ in mod.named_parameters()] == case["param_names"]
    for k, rec in case["state"].items():
        a = sd[k].detach().cpu().numpy()
        assert list(a.shape) == rec["shape"], k
        assert str(a.dtype) == rec["dtype"], k
        if sha(a) == rec["sha256"]:
            continue
        ref = golden.buffers["%s/%s" % (name, k)]
        mine = a if rec["stored"] == "full" else a.reshape(-1)[:: golden.stride]
        peak = max(np.abs(ref).max(), 1e-30)
        err = np.abs(mine.astype(np.float64) - ref.astype(np.float64)).max()
        assert err <= 1e-6 * peak + 1e-9, "%s/%s differs: %.3e (peak %.3e)" % (name, k, err, peak)


@pytest.mark.parametrize("name", _golden.case_names())
def test_attributes_match_reference(golden, name):
    case = golden.cases[name]
    mod = build_module(case)
    for a, want in case["attrs"].items():
        got = getattr(mod, a)
        if isinstance(want, list):
            assert np.allclose(np.asarray(got, dtype=np.float64), np.asarray(want), rtol=1e-12, atol=0), a
        elif isinstance(want, float):
            assert got == pytest.approx(want, rel=1e-12), a
        else:
            assert got == want, (a, got, want)


def test_bit_identical_fraction(golden):
    """Informational guard: the large majority of reference buffers must be reproduced
    bit-for-bit (the builders follow the same float64 formulas)."""
    same = total = 0
    for name, case in golden.cases.items():
        sd = build_module(case).state_dict()
        for k, rec in case["state"].items():
            total += 1
            same += sha(sd[k].detach().cpu().numpy()) == rec["sha256"]
    assert same / total > 0.9, (same, total)


def test_nyquist_errors():
    from nnaudio_amd import features

    with pytest.raises(ValueError):
        features.CQT1992v2(sr=4000, n_bins=84, verbose=False)
    with pytest.raises(ValueError):
        features.CQT2010v2(sr=4000, n_bins=84, verbose=False)
    with pytest.raises(ValueError):
        features.VQT(sr=4000, n_bins=84, verbose=False)


def test_cqt_alias():
    from nnaudio_amd import features

    assert issubclass(features.CQT, features.CQT1992v2)


def test_octave_bank_zero_margin():
    """The margin ``OctaveCache.bank`` cuts off both ends of an octave's kernels: zeros only, a
    multiple of 16, the kernel stays centred and a multiple of 32 wide; the reference's own
    CQT2010v2 bank (cqt.py:1023-1040: kernels centred in a power-of-two width) loses a quarter."""
    import torch
    from nnaudio_amd import features
    from nnaudio_amd.features._cqt_common import zero_margin

    re = torch.zeros(3, 256)
    im = torch.zeros(3, 256)
    assert zero_margin(re, im) == 0                      # nothing to centre on
    re[1, 40:217] = 1.0
    assert zero_margin(re, im) == 32                     # 40 / 39 zeros -> 32 each side, 192 left
    im[2, 17] = 1.0
    assert zero_margin(re, im) == 16
    im[0, 3] = 1.0
    assert zero_margin(re, im) == 0
    assert zero_margin(torch.zeros(2, 48), torch.zeros(2, 48)) == 0   # width not a multiple of 32
    one = torch.zeros(1, 64)
    one[0, 32] = 1.0
    assert zero_margin(one, one) == 16                   # never below 32 taps
    m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False)
    kr = m.cqt_kernels_real.reshape(m.cqt_kernels_real.shape[0], -1)
    ki = m.cqt_kernels_imag.reshape(m.cqt_kernels_imag.shape[0], -1)
    K, mg = kr.shape[1], zero_margin(kr, ki)
    assert K == 256 and mg == 32
    assert not kr[:, :mg].any() and not kr[:, K - mg:].any() and not ki[:, :mg].any() and not ki[:, K - mg:].any()
