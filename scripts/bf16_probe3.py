import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
DEV = "cuda:0"
torch.manual_seed(0)
x = torch.randn(64, 441000, device=DEV)
c = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to(DEV)
sup = c._support.get(c.cqt_kernels_real, c.cqt_kernels_imag)
split = engine.split_basis(c.cqt_kernels_real, c.cqt_kernels_imag)
sc = torch.sqrt(c.lenghts)
def run(dbg):
    return engine.framed_gemm(x, c.cqt_kernels_real, c.cqt_kernels_imag, hop=512, pad=16384, pad_mode=2,
                              epilogue=engine.EPI_MAGNITUDE, row_scale=sc, row_support=sup,
                              precision="bf16x3", basis_split=split, _debug=dbg)
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dbg in [int(a, 0) for a in sys.argv[1:]]:
    print("dbg %#x: %.3f ms" % (dbg, t(lambda: run(dbg))))
