"""The reference's own CQT1992v2 fixture assertions (reference tests/test_cqt.py:94-186), VERBATIM, on the default module;
also: is the module bit-identical to torch's conv1d (CPU / this GPU) on these inputs?"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import chirp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests._golden import Golden, build_module  # noqa: E402

g = Golden()
dev = torch.device("cuda:0")
s = np.linspace(0, 1, 44100)
for sweep, method in (("log", "logarithmic"), ("linear", "linear")):
    x = torch.from_numpy(chirp(s, 55, 1, 22050, method=method).astype(np.float32)[None, :])
    for fmt, tag in (("Magnitude", "mag"), ("Complex", "complex"), ("Phase", "phase")):
        case = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24, output_format=fmt), fwd={})
        mod = build_module(case, dev)
        with torch.no_grad():
            X = mod(x.to(dev))
        gt = g.ground_truth("%s-sweep-cqt-1992-%s-ground-truth.npy" % (sweep, tag))
        if fmt == "Magnitude":
            X = torch.log(X + 1e-5)
        Xc = X.cpu().numpy()
        ok = np.isclose(Xc, gt.reshape(Xc.shape), rtol=1e-3, atol=1e-3)
        print("%s sweep %-9s verbatim allclose: %s (%.6f of the elements miss)" % (sweep, fmt, bool(ok.all()), float((~ok).mean())), flush=True)
    # bit identity of the complex output with torch's conv1d
    case = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24, output_format="Complex"), fwd={})
    mod = build_module(case, dev)
    with torch.no_grad():
        y = mod(x.to(dev)).cpu()
    cpu = build_module(case)
    for name, d in (("CPU", "cpu"), ("MIOpen", dev)):
        kr, ki, ln = (t.to(d) for t in (cpu.cqt_kernels_real, cpu.cqt_kernels_imag, cpu.lenghts))
        xp = F.pad(x.to(d)[:, None, :], (cpu.kernel_width // 2,) * 2, mode="reflect")
        re = (F.conv1d(xp, kr, stride=cpu.hop_length) * torch.sqrt(ln.view(-1, 1))).cpu()
        im = (-F.conv1d(xp, ki, stride=cpu.hop_length) * torch.sqrt(ln.view(-1, 1))).cpu()
        print("%s sweep: module == torch conv1d (%s) bit for bit: re %.4f, im %.4f of the elements; max |d| %.2e of the peak"
              % (sweep, name, float((y[..., 0] == re).float().mean()), float((y[..., 1] == im).float().mean()),
                 float(max((y[..., 0] - re).abs().max(), (y[..., 1] - im).abs().max()) / y.abs().max())), flush=True)
