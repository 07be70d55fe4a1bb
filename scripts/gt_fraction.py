"""Fraction of elements missing the reference's verbatim log(X + 1e-5) assertion on its CQT1992v2
log-sweep ground truth, per arithmetic path (see tests/_golden.py:check_ground_truth)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.signal import chirp
from tests._golden import build_module
from nnaudio_amd import engine
eps = 1e-5
for sweep, method in (("log", "logarithmic"), ("linear", "linear")):
    s = np.linspace(0, 1, 44100)
    x = torch.as_tensor(chirp(s, 55, 1, 22050, method=method).astype(np.float32)[None, :]).cuda()
    case = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24, output_format="Magnitude"), fwd={})
    m = build_module(case, "cuda")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gt = np.load(os.path.join(root, "tests/golden/ref_ground_truths/%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep))
    for prec in ("fp32", "bf16x3"):
        m.precision = prec
        y = m(x).cpu().numpy().astype(np.float64).reshape(gt.shape)
        ok = np.isclose(np.log(y + eps), gt, rtol=1e-3, atol=1e-3)
        lin = np.exp(gt.astype(np.float64)) - eps
        print(sweep, prec, "missing %.4f; conditioned bins all ok: %s; linear err %.2e of peak"
              % ((~ok).mean(), bool(ok[lin > 1e-2 * lin.max()].all()), np.abs(y - lin).max() / lin.max()))
    kr, ki = m.cqt_kernels_real, m.cqt_kernels_imag
    sup = m._support.get(kr, ki)
    y = engine.framed_gemm(x, kr, ki, hop=512, pad=m.kernel_width // 2, pad_mode=2, epilogue=engine.EPI_MAGNITUDE,
                           row_scale=torch.sqrt(m.lenghts), row_support=sup, precision="fp32").cpu().numpy().astype(np.float64).reshape(gt.shape)
    ok = np.isclose(np.log(y + eps), gt, rtol=1e-3, atol=1e-3)
    print(sweep, "fp32 tile kernel: missing %.4f" % (~ok).mean())
