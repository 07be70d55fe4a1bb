"""What the REFERENCE'S OWN GPU path -- F.conv1d on this MI355X (MIOpen), a test-side restatement of
CQT1992v2.forward (cqt.py:740-772) on the product module's buffers -- misses of the reference's fixture assertion
(tests/test_cqt.py:94-186: allclose(log(X + 1e-5), gt, rtol = atol = 1e-3)), next to the product's arithmetics.
Run on the GPU box:  python scripts/ref_gpu_path_miss.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import chirp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _golden import Golden, build_module  # noqa: E402


def miss_of(y, gt, eps=1e-5):
    ok = np.isclose(np.log(y + eps), gt.reshape(y.shape), rtol=1e-3, atol=1e-3)
    return float((~ok).mean())


def reference_order(mod, x, device):
    """cqt.py:740-772 on `device` with torch's own conv1d"""
    kr, ki, ln = (t.to(device) for t in (mod.cqt_kernels_real, mod.cqt_kernels_imag, mod.lenghts))
    xp = F.pad(x.to(device)[:, None, :], (mod.kernel_width // 2, mod.kernel_width // 2), mode="reflect")
    re = F.conv1d(xp, kr, stride=mod.hop_length) * torch.sqrt(ln.view(-1, 1))
    im = -F.conv1d(xp, ki, stride=mod.hop_length) * torch.sqrt(ln.view(-1, 1))
    return torch.sqrt(re.pow(2) + im.pow(2)).float().cpu().numpy()


def main():
    g = Golden()
    dev = torch.device("cuda:0")
    rows = []
    for sweep, method in (("log", "logarithmic"), ("linear", "linear")):
        s = np.linspace(0, 1, 44100)
        x = torch.from_numpy(chirp(s, 55, 1, 22050, method=method).astype(np.float32)[None, :])
        case = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24, output_format="Magnitude"), fwd={})
        gt = g.ground_truth("%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep)
        mod = build_module(case)
        rows.append((sweep, "reference order, torch conv1d CPU", miss_of(reference_order(mod, x, "cpu"), gt)))
        for bench in (False, True):
            torch.backends.cudnn.benchmark = bench
            rows.append((sweep, "reference order, torch conv1d on the MI355X (MIOpen, benchmark=%s)" % bench,
                         miss_of(reference_order(mod, x, dev), gt)))
        torch.backends.cudnn.benchmark = False
        # batch of 8 copies: MIOpen may pick another solver
        y8 = reference_order(mod, x.repeat(8, 1), dev)
        rows.append((sweep, "reference order, MIOpen, batch 8 (clip 0)", miss_of(y8[:1], gt)))
        modg = build_module(case, dev)
        for prec, hp in (("fp32", None), ("f16x3", True), ("f16x3", False), ("bf16x3", True)):
            modg.precision = prec
            if hp is not None:
                modg.hop_periodic = hp
            with torch.no_grad():
                y = modg(x.to(dev)).float().cpu().numpy()
            rows.append((sweep, "product %s%s" % (prec, "" if hp is None else (" strip" if hp else " natural order")), miss_of(y, gt)))
    for r in rows:
        print("%-7s %-75s miss %.5f" % r)


if __name__ == "__main__":
    main()
