"""Random shapes through the strip kernel against the one-thread-per-output device kernel:
python scripts/strip_fuzz.py [cases] [seed] [bf16x3|fp32|f16x3]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import _abi, engine

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
PREC = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
DEV = "cuda"
taken = 0
for case in range(n_cases):
    hop = int(rng.choice([32, 64, 96, 128, 160, 256, 512]))
    K = int(rng.integers(2 * hop, min(80 * hop, 12000)))
    F = int(rng.integers(65, 220))
    B = int(rng.integers(1, 5))
    T = int(rng.integers(64, 400))
    center = bool(rng.integers(0, 2))
    pad = K // 2 if center else 0
    mode = int(rng.choice([1, 2])) if center else 0
    L = (T - 1) * hop + K - 2 * pad + int(rng.integers(0, hop))
    if mode == 2 and pad >= L:
        continue
    # supports: random, not necessarily centred, some empty, lengths spread over two decades
    ln = np.maximum(1, (K * 10.0 ** rng.uniform(-2, 0, F)).astype(np.int64))
    ln = np.sort(ln)[::-1] if rng.integers(0, 2) else ln
    lo = (rng.uniform(0, 1, F) * (K - ln)).astype(np.int64) if rng.integers(0, 2) else (K - ln) // 2
    hi = lo + ln
    if rng.integers(0, 3) == 0:
        hi[rng.integers(0, F)] = lo[rng.integers(0, F)]  # an empty row somewhere (maybe)
        hi = np.maximum(hi, lo)
    keep = (np.arange(K)[None, :] >= lo[:, None]) & (np.arange(K)[None, :] < hi[:, None])
    wr = torch.as_tensor((rng.standard_normal((F, K)) * keep).astype(np.float32)).to(DEV)
    wi = torch.as_tensor((rng.standard_normal((F, K)) * keep).astype(np.float32)).to(DEV)
    x = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).to(DEV)
    sup_np = np.ascontiguousarray(np.stack([lo, hi], 1).astype(np.int32))
    sup = torch.as_tensor(sup_np).to(DEV)
    sup.host_copy = sup_np
    sc = torch.as_tensor(rng.uniform(0.5, 2.0, F).astype(np.float32)).to(DEV)
    epi = int(rng.choice([engine.EPI_COMPLEX, engine.EPI_MAGNITUDE, engine.EPI_POWER, engine.EPI_PHASE_COSSIN]))
    kw = dict(hop=hop, pad=pad, pad_mode=mode, epilogue=epi, row_scale=sc)
    if PREC == "fp32":
        kw["basis_split"] = engine.frag_basis_f32(wr, wi)
    if PREC == "f16x3":  # the operand scaling: any level of signal and bank must do
        x = x * float(10.0 ** rng.uniform(-6, 4))
        g = float(10.0 ** rng.uniform(-5, 3))
        wr, wi = wr * g, wi * g
        kw["basis_split"] = engine.frag_basis_f16(wr, wi)
    a, _o, _d, _k = engine._framed_args(x, wr, wi, precision=PREC, row_support=sup, **kw)
    n_pass = _abi.load().mispec_strip_plan(ctypes.byref(a), 256, None, 0)
    taken += n_pass > 0
    rkw = {k: v for k, v in kw.items() if k != "basis_split"}
    ref = engine.framed_gemm(x, wr, wi, reference_kernel=True, **rkw)
    y = engine.framed_gemm(x, wr, wi, precision=PREC, row_support=sup, **kw)
    torch.cuda.synchronize()
    if epi == engine.EPI_PHASE_COSSIN:
        z = engine.framed_gemm(x, wr, wi, reference_kernel=True, **dict(rkw, epilogue=engine.EPI_COMPLEX))
        mag = torch.sqrt(z[..., 0] ** 2 + z[..., 1] ** 2)
        strong = mag > 0.05 * mag.max()
        err = float((y - ref)[strong].abs().max()) if bool(strong.any()) else 0.0
        ok = err < 2e-3
    else:
        scale = float(ref.abs().max())
        err = float((y - ref).abs().max()) / max(scale, 1e-30)
        ok = err <= (1e-4 if PREC == "bf16x3" else 1e-5) and bool(torch.isfinite(y).all())
    print("case %2d hop %3d K %5d F %3d B %d T %3d pad %d epi %d passes %d: err %.2e %s"
          % (case, hop, K, F, B, T, mode, epi, n_pass, err, "ok" if ok else "FAIL"))
    if not ok:
        sys.exit(1)
print("all ok; strip kernel taken in %d cases" % taken)
