"""CQT1992v2 84 bins on the bench's full batch (64 x 10 s @ 44.1 kHz): fraction of the module's complex output with the same
bits as reflect pad + F.conv1d with the module's kernels on this GPU (the reference's operator sequence, cqt.py:740-772), and
what that sequence costs there."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import features  # noqa: E402

m = features.CQT1992v2(sr=44100, hop_length=512, fmin=32.70, n_bins=84, bins_per_octave=12, output_format="Complex", verbose=False).cuda()
for B in (1, 8, 64):
    x = torch.randn(B, 441000, device="cuda")
    with torch.no_grad():
        y = m(x)
        s = torch.sqrt(m.lenghts.view(-1, 1))

        def ref():
            xp = F.pad(x[:, None, :], (m.kernel_width // 2,) * 2, mode="reflect")
            return F.conv1d(xp, m.cqt_kernels_real, stride=m.hop_length) * s, -F.conv1d(xp, m.cqt_kernels_imag, stride=m.hop_length) * s

        re, im = ref()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ref()
        torch.cuda.synchronize()
        t_ref = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        for _ in range(10):
            m(x)
        torch.cuda.synchronize()
        t_mod = (time.perf_counter() - t0) / 10 * 1e3
    same = float(((y[..., 0] == re) & (y[..., 1] == im)).float().mean())
    err = float(max((y[..., 0] - re).abs().max(), (y[..., 1] - im).abs().max()) / y.abs().max())
    print("B=%2d: same bits as torch conv1d (MIOpen) on %.6f of the elements (max |d| %.1e of the peak); torch's operator sequence %.2f ms, "
          "this module %.3f ms" % (B, same, err, t_ref, t_mod), flush=True)
