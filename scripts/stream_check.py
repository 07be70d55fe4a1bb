"""Bring-up check of the streaming octave kernel on the GPU (run through gpurun): kernel-level parity
against the float64 recursion, stream vs pyramid on the modules, timing of the cfg5 shard."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import octave_stream_model as M
from nnaudio_amd import engine, features

dev = "cuda:0"
torch.manual_seed(0)


def kernel_level(L0=40000, hop0=512, K=(192, 192, 192, 192), n_seg=2, precision="f16x3", B=2, reflect=True, scale=1.0):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((B, L0)) * scale).astype(np.float32)
    taps = (np.hanning(256) * np.sinc((np.arange(256) - 127.5) / 2) / 2).astype(np.float32)
    n_frames = L0 // hop0 + 1
    banks = [None if not k else ((rng.standard_normal((12, k)) + 1j * rng.standard_normal((12, k))) / k).astype(np.complex64) for k in K]
    xd = torch.as_tensor(x).to(dev)
    levels, row0 = [], 0
    for b in banks:
        if b is None:
            levels.append(None)
            continue
        re = torch.as_tensor(np.ascontiguousarray(b.real)).to(dev)
        im = torch.as_tensor(np.ascontiguousarray(b.imag)).to(dev)
        split = engine.split_basis_f16(re, im) if precision == "f16x3" else engine.split_basis(re, im)
        levels.append(dict(split=split, n_bins=12, kernel=b.shape[1], row_offset=row0,
                           pad_mode=engine.PAD_REFLECT if reflect else engine.PAD_ZERO, row_scale=None))
        row0 += 12
    out = torch.full((B, row0, n_frames, 2), float("nan"), device=dev)
    Ls = [L0]
    for _ in range(len(K) - 1):
        Ls.append(M.decimated_length(Ls[-1]))
    x_last = torch.full((B, Ls[-1]), float("nan"), device=dev)
    ok = engine.octave_stream(xd, levels, hop=hop0, n_frames=n_frames, taps=torch.as_tensor(taps).to(dev),
                              epilogue=engine.EPI_COMPLEX, im_sign=1.0, eps=0.0, out=out, x_last=x_last,
                              precision=precision, fir_headroom_bits=1 * (len(K) - 1), n_segments=n_seg)
    torch.cuda.synchronize()
    if not ok:
        return "unsupported"
    y = out.cpu().numpy()
    xl = x_last.cpu().numpy()
    worst = 0.0
    for b in range(B):
        ref, xs = M.reference(x[b].astype(np.float64), taps.astype(np.float64), [None if k is None else k.astype(np.complex128) for k in banks], hop0, n_frames, reflect)
        r0 = 0
        for r in ref:
            if r is None:
                continue
            got = y[b, r0:r0 + 12, :, 0] + 1j * y[b, r0:r0 + 12, :, 1]
            e = np.abs(got - r).max() / np.abs(r).max()
            if not np.isfinite(e):
                bad = np.argwhere(~np.isfinite(got))
                return "NaN at level rows %d: %d elements, first %s" % (r0, len(bad), bad[:3].tolist())
            worst = max(worst, e)
            r0 += 12
        e = np.abs(xl[b] - xs[-1]).max() / np.abs(xs[-1]).max()
        if not np.isfinite(e):
            return "x_last NaN: %s" % np.flatnonzero(~np.isfinite(xl[b]))[:5]
        worst = max(worst, e)
    return worst


def main():
    for kw in (dict(), dict(n_seg=1), dict(n_seg=3, L0=70001 // 4 * 4), dict(precision="bf16x3"), dict(reflect=False),
               dict(hop0=64, K=(0, 192, 192, 192, 192), L0=30000, n_seg=3), dict(K=(256, 256, 128, 64), L0=44100 // 4 * 4),
               dict(scale=1e-3), dict(scale=300.0)):
        t0 = time.time()
        print("kernel", kw, "->", kernel_level(**kw), "(%.1fs)" % (time.time() - t0), flush=True)
    # a loud burst in a quiet clip: the rescale path
    rng = np.random.default_rng(2)
    # module level: stream vs pyramid
    for cls, kw, L in ((features.CQT2010v2, dict(sr=44100, hop_length=512, n_bins=96, verbose=False), 132300),
                       (features.VQT, dict(sr=44100, hop_length=512, n_bins=96, gamma=10, verbose=False), 132300),
                       (features.CQT2010v2, dict(sr=22050, hop_length=256, n_bins=84, output_format="Complex", verbose=False), 66000)):
        m = cls(**kw).to(dev)
        x = torch.randn(3, L, device=dev)
        x[1, 50000:50100] *= 200.0
        with torch.no_grad():
            engine.set_octave_stream(True)
            a = m(x)
            engine.set_octave_stream(False)
            b = m(x)
            m.precision = "fp32"
            c = m(x)
            m.precision = None
        torch.cuda.synchronize()
        print(cls.__name__, kw.get("gamma"), "stream vs pyramid %.2e  stream vs fp32 %.2e  pyramid vs fp32 %.2e  equal=%s" % (
            ((a - b).abs().max() / b.abs().max()).item(), ((a - c).abs().max() / c.abs().max()).item(),
            ((b - c).abs().max() / c.abs().max()).item(), torch.equal(a, b)), flush=True)
    # cfg5 shard timing
    m = features.CQT2010v2(sr=44100, hop_length=512, n_bins=96, verbose=False).to(dev)
    x = torch.randn(64, 1323000, device=dev)
    for mode in (True, False, True):
        engine.set_octave_stream(mode)
        with torch.no_grad():
            for _ in range(5):
                y = m(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = m(x)
            e1.record()
            torch.cuda.synchronize()
        print("cfg5 shard stream=%s: %.4f ms" % (mode, e0.elapsed_time(e1) / 20), flush=True)


if __name__ == "__main__":
    main()
