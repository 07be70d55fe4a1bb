"""What the REFERENCE'S OWN arithmetic -- torch's conv1d, CQT1992v2.forward restated on the product module's buffers
(reference cqt.py:740-772) -- misses of the reference's fixture assertion (reference tests/test_cqt.py:94-186:
allclose(log(X + 1e-5), ground truth, rtol = atol = 1e-3)), measured where the product runs.

VERDICT r4 asked for the CQT1992v2 fixture bar to be SET FROM THIS MEASUREMENT ("no worse than the reference on this
hardware") instead of a hand-picked 0.5 %.  Measured (scripts/ref_gpu_path_miss.py, profiles/r05/ref_gpu_path_miss.log):

    arithmetic                                              log sweep   linear sweep
    reference order, torch conv1d, CPU (1 .. 8 threads)        0 %          0 %
    reference order, torch conv1d on the MI355X (MIOpen)       0 %          0 %
    exact float64 evaluation                                   0.13 %       0 %
    product fp32 (tile kernels: the reference's tap order)     0.033 %      0 %
    product f16x3, natural tap order (staged dense kernel)     0.58 %       0.21 %
    product f16x3, strip kernel (hop-periodic tap order)       2.7 %        0.87 %
    product bf16x3                                             57 %         74 %

The reference reproduces its own fixture VERBATIM on both devices: the near-silent bins of the fixture (1e-9 of the
peak) record the rounding of one sequential float32 summation, which oneDNN and MIOpen both perform and which an exact
evaluation misses.  A bar derived from the reference is therefore 0 misses, which no arithmetic of this library
reaches; fp32 is the nearest (0.033 %) and stays CQT1992v2's default, pinned here at 0.1 %; f16x3 (4.7e-7 of the peak
against float64 -- 200 x inside north_star's 1e-4, but 2.7 % of this fixture's silent bins) stays the opt-in
`module.precision = "f16x3"` and is named as such on the bench line (roofline_cqt84_f16x3, default_module: false)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from scipy.signal import chirp

from tests._golden import Golden, build_module

CASE = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24, output_format="Magnitude"), fwd={})
SWEEPS = [("log", "logarithmic"), ("linear", "linear")]
# the bars the product's arithmetics are held to, from the measurements above (x3 margin on fp32; f16x3 / bf16x3 are pinned
# so that they cannot grow unnoticed -- they do NOT meet a reference-derived bar and do not ship as the default)
FP32_BAR = 1e-3


def _chirp(method):
    s = np.linspace(0, 1, 44100)
    return torch.from_numpy(chirp(s, 55, 1, 22050, method=method).astype(np.float32)[None, :])


def miss_fraction(y, gt, eps=1e-5):
    ok = np.isclose(np.log(np.asarray(y, dtype=np.float32) + eps), gt.reshape(y.shape), rtol=1e-3, atol=1e-3)
    return float((~ok).mean())


def reference_order(mod, x, device):
    """reference cqt.py:740-772 with torch's own conv1d on `device`"""
    kr, ki, ln = (t.to(device) for t in (mod.cqt_kernels_real, mod.cqt_kernels_imag, mod.lenghts))
    xp = F.pad(x.to(device)[:, None, :], (mod.kernel_width // 2, mod.kernel_width // 2), mode="reflect")
    re = F.conv1d(xp, kr, stride=mod.hop_length) * torch.sqrt(ln.view(-1, 1))
    im = -F.conv1d(xp, ki, stride=mod.hop_length) * torch.sqrt(ln.view(-1, 1))
    return torch.sqrt(re.pow(2) + im.pow(2)).float().cpu().numpy()


@pytest.mark.parametrize("sweep,method", SWEEPS)
def test_reference_conv1d_cpu_reproduces_its_fixture(sweep, method):
    g = Golden()
    mod = build_module(CASE)
    gt = g.ground_truth("%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep)
    miss = miss_fraction(reference_order(mod, _chirp(method), "cpu"), gt)
    print("reference order, torch conv1d on the CPU, %s sweep: %.5f miss" % (sweep, miss))
    assert miss <= 1e-4  # measured 0 at 1, 2, 4 and 8 threads


@pytest.mark.gpu
@pytest.mark.parametrize("sweep,method", SWEEPS)
def test_reference_conv1d_on_the_gpu_and_the_products_arithmetics(sweep, method):
    """The measurement the default precision of CQT1992v2 rests on, repeated on whatever GPU runs the suite."""
    g = Golden()
    dev = torch.device("cuda:0")
    x = _chirp(method)
    gt = g.ground_truth("%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep)
    ref = miss_fraction(reference_order(build_module(CASE), x, dev), gt)
    mod = build_module(CASE, dev)
    got = {}
    for name, prec, hp in (("fp32", "fp32", None), ("f16x3 strip", "f16x3", True), ("f16x3 natural order", "f16x3", False)):
        mod.precision = prec
        if hp is not None:
            mod.hop_periodic = hp
        with torch.no_grad():
            got[name] = miss_fraction(mod(x.to(dev)).float().cpu().numpy(), gt)
    mod.precision = None
    mod.hop_periodic = True
    with torch.no_grad():
        default = miss_fraction(mod(x.to(dev)).float().cpu().numpy(), gt)
    print("%s sweep: reference conv1d on this GPU %.5f | product %s | default module %.5f"
          % (sweep, ref, ", ".join("%s %.5f" % kv for kv in got.items()), default))
    assert ref <= 1e-4                       # the reference's GPU path passes its own assertion (measured 0)
    assert default == got["fp32"]             # the module ships the arithmetic nearest to it ...
    assert got["fp32"] <= FP32_BAR            # ... pinned from the measurement (0.033 % / 0)
    assert got["fp32"] <= got["f16x3 natural order"] <= 0.009 and got["f16x3 strip"] <= 0.04
