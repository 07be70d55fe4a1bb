"""CPU emulation of the fp32 MFMA summation orders on the reference's CQT1992v2 log-sweep fixture: which assignment of
taps to MFMA steps leaves the near-silent bins closest to the reference's own float32 noise (miss fraction of
allclose(log(X + 1e-5), fixture, 1e-3, 1e-3))?  Model: one MFMA = exact sum of its K products + accumulator, rounded once
(model "fused") or a float32 FMA chain through its K taps (model "chain")."""
import os
import sys

import numpy as np
import torch
from scipy.signal import chirp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests._golden import Golden, build_module  # noqa: E402

CASE = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24, output_format="Magnitude"), fwd={})
method = sys.argv[1] if len(sys.argv) > 1 else "logarithmic"
name = {"logarithmic": "log", "linear": "linear"}[method]
mod = build_module(CASE)
gt = Golden().ground_truth("%s-sweep-cqt-1992-mag-ground-truth.npy" % name)
s = np.linspace(0, 1, 44100)
x = chirp(s, 55, 1, 22050, method=method).astype(np.float32)
K = mod.kernel_width
xp = torch.nn.functional.pad(torch.from_numpy(x)[None, None], (K // 2, K // 2), mode="reflect")[0, 0].numpy()
T = (len(xp) - K) // mod.hop_length + 1
X = np.stack([xp[t * mod.hop_length: t * mod.hop_length + K] for t in range(T)]).astype(np.float32)  # (T, K)
Wr = mod.cqt_kernels_real.numpy()[:, 0, :] if mod.cqt_kernels_real.dim() == 3 else mod.cqt_kernels_real.numpy()
Wi = mod.cqt_kernels_imag.numpy()[:, 0, :] if mod.cqt_kernels_imag.dim() == 3 else mod.cqt_kernels_imag.numpy()
sc = np.sqrt(mod.lenghts.numpy()).astype(np.float32)[:, None]


def miss(re, im):
    re = re.astype(np.float32) * sc
    im = im.astype(np.float32) * sc
    y = np.sqrt(re * re + im * im).astype(np.float32)
    ok = np.isclose(np.log(y + np.float32(1e-5)), gt.reshape(y.shape), rtol=1e-3, atol=1e-3)
    return float((~ok).mean())


def run(groups, model):
    """groups: list of index arrays (the taps of consecutive MFMA steps)"""
    out = []
    for W in (Wr, Wi):
        acc = np.zeros((W.shape[0], T), np.float32)
        W64, X64 = W.astype(np.float64), X.astype(np.float64)
        for g in groups:
            if model == "fused":
                acc = (acc.astype(np.float64) + W64[:, g] @ X64[:, g].T).astype(np.float32)
            else:
                for k in g:  # float32 FMA chain: one rounding per tap
                    acc = (acc.astype(np.float64) + np.outer(W64[:, k], X64[:, k])).astype(np.float32)
        out.append(acc)
    return miss(*out)


def order_32x32x2():   # group q of 8 taps: step s multiplies taps 8q + s and 8q + 4 + s
    return [np.array([8 * q + s, 8 * q + 4 + s]) for q in range(K // 8) for s in range(4)]


def order_16x16x4_strided():  # group q of 16 taps: step s multiplies taps 16q + 4 lq + s
    return [np.array([16 * q + 4 * lq + s for lq in range(4)]) for q in range(K // 16) for s in range(4)]


def order_16x16x4_contiguous():
    return [np.arange(4 * i, 4 * i + 4) for i in range(K // 4)]


def order_seq(n):
    return [np.arange(n * i, n * i + n) for i in range(K // n)]


print("exact float64:", miss(Wr.astype(np.float64) @ X.astype(np.float64).T, Wi.astype(np.float64) @ X.astype(np.float64).T), flush=True)
for nm, groups in (("32x32x2 (8q+s, 8q+4+s)", order_32x32x2()), ("16x16x4 strided (16q+4lq+s)", order_16x16x4_strided()),
                   ("16x16x4 contiguous (4i..4i+3)", order_16x16x4_contiguous()), ("pairs contiguous (2i, 2i+1)", order_seq(2)),
                   ("sequential, one tap per step", order_seq(1))):
    print("%-36s fused %.5f  chain %.5f" % (nm, run(groups, "fused"), run(groups, "chain")), flush=True)
