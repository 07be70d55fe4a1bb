"""CPU emulation (numpy) of split-operand arithmetics on the reference's CQT1992v2 log-magnitude
ground truths: which of them would pass the reference's verbatim log(X + 1e-5) assertion, and
with what dynamic range?  Products are formed exactly (float64) from the split operands; the
accumulator is float64 ("exact") or float32 with one rounding per 16-tap MFMA step ("acc32").
No GPU needed.  Usage: python scripts/split_emulation.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scipy.signal import chirp
import torch
from tests._golden import build_module

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
eps = 1e-5


def bf16(v):
    v = np.asarray(v, np.float32)
    u = v.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(v, kind, n):
    """n parts of v (float32 array) in the 16-bit format `kind`."""
    parts, r = [], np.asarray(v, np.float32).copy()
    for _ in range(n):
        h = bf16(r) if kind == "bf16" else r.astype(np.float16).astype(np.float32)
        parts.append(h.astype(np.float64))
        r = (r - h).astype(np.float32)
    return parts


def pow2_scale(m, top=15):
    e = np.frexp(np.maximum(m, 1e-30))[1]
    return np.ldexp(1.0, top - e)


def contract(A, X, mode, acc32):
    """A (F, K) float32 rows, X (T, K) float32 frames -> (F, T) float64."""
    if mode == "exact":
        terms = [(A.astype(np.float64), X.astype(np.float64))]
        sa = sx = None
    elif mode in ("bf16x3", "bf16x6"):
        n = 2 if mode == "bf16x3" else 3
        a, x = split(A, "bf16", n), split(X, "bf16", n)
        if n == 2:
            terms = [(a[1], x[0]), (a[0], x[1]), (a[0], x[0])]
        else:
            terms = [(a[2], x[0]), (a[0], x[2]), (a[1], x[1]), (a[1], x[0]), (a[0], x[1]), (a[0], x[0])]
        sa = sx = None
    elif mode in ("f16x3", "f16x3u"):
        if mode == "f16x3":
            sa = pow2_scale(np.abs(A).max(1, keepdims=True))
            sx = pow2_scale(np.abs(X).max())  # one scale per clip
        else:
            sa, sx = np.ones((A.shape[0], 1)), 1.0
        a = split((A * sa).astype(np.float32), "f16", 2)
        x = split((X * sx).astype(np.float32), "f16", 2)
        terms = [(a[1], x[0]), (a[0], x[1]), (a[0], x[0])]
    else:
        raise KeyError(mode)
    if not acc32:
        out = sum(a @ x.T for a, x in terms)
    else:
        K = A.shape[1]
        acc = np.zeros((A.shape[0], X.shape[0]), np.float32)
        nz = np.flatnonzero(np.abs(A).max(0) > 0)
        k0, k1 = (nz.min() // 16) * 16, nz.max() + 1
        for k in range(k0, k1, 16):
            for a, x in terms:
                acc = (acc.astype(np.float64) + a[:, k:k + 16] @ x[:, k:k + 16].T).astype(np.float32)
        out = acc.astype(np.float64)
    if sa is not None:
        out = out / (sa * sx)
    return out


for sweep, method in (("log", "logarithmic"), ("linear", "linear")):
    s = np.linspace(0, 1, 44100)
    x = chirp(s, 55, 1, 22050, method=method).astype(np.float32)
    case = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24,
                                           output_format="Magnitude"), fwd={})
    m = build_module(case, "cpu")
    kr = m.cqt_kernels_real.numpy()[:, 0, :]
    ki = m.cqt_kernels_imag.numpy()[:, 0, :]
    Kw = kr.shape[1]
    xp = np.pad(x, Kw // 2, mode="reflect")
    T = (len(xp) - Kw) // 512 + 1
    X = np.stack([xp[t * 512:t * 512 + Kw] for t in range(T)])
    sc = np.sqrt(m.lenghts.numpy().astype(np.float64))[:, None]
    gt = np.load(os.path.join(root, "tests/golden/ref_ground_truths/%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep))
    lin = np.exp(gt.astype(np.float64)) - eps
    ex = None
    for mode in ("exact", "bf16x3", "bf16x6", "f16x3", "f16x3u"):
        for acc32 in (False, True):
            re = contract(kr, X, mode, acc32) * sc
            im = contract(ki, X, mode, acc32) * sc
            y = np.sqrt(re * re + im * im).astype(np.float32).astype(np.float64).reshape(gt.shape)
            if ex is None:
                ex = y
            ok = np.isclose(np.log(y + eps), gt, rtol=1e-3, atol=1e-3)
            print("%-6s %-7s %-6s missing %.4f  conditioned ok %s  err vs gt %.2e of peak, vs exact %.2e"
                  % (sweep, mode, "acc32" if acc32 else "acc64", (~ok).mean(),
                     bool(ok[lin > 1e-2 * lin.max()].all()), np.abs(y - lin).max() / lin.max(),
                     np.abs(y - ex).max() / lin.max()), flush=True)
