import sys, torch
sys.path.insert(0, "/root/repo")
from nnaudio_amd import engine, features
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = torch.randn(B, 441000, device="cuda")
m = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to("cuda")
sup = m._support.get(m.cqt_kernels_real, m.cqt_kernels_imag)
sc = torch.sqrt(m.lenghts)
def run(dbg):
    return engine.framed_gemm(x, m.cqt_kernels_real, m.cqt_kernels_imag, hop=512, pad=16384, pad_mode=2,
                              epilogue=engine.EPI_MAGNITUDE, row_scale=sc, row_support=sup, precision="bf16x3", _debug=dbg)
y = run(0x1000000)
ref = run(0x800000)
torch.cuda.synchronize()
bad = ~torch.isfinite(y)
print("nan count", int(bad.sum()), "of", y.numel())
d = (y - ref).abs()
d[bad] = 1e9
big = d > 1e-3 * ref.abs().max()
print("mismatch count", int(big.sum()))
if big.any():
    idx = big.nonzero()
    print("clips", idx[:, 0].unique().tolist()[:20])
    print("bins", idx[:, 1].unique().tolist())
    fr = idx[:, 2].unique()
    print("frames", fr.tolist()[:40], "...", len(fr))
for i in range(3):
    y2 = run(0)
    print("rerun max diff vs first", float((y2 - y).abs().nan_to_num(1e9).max()))
