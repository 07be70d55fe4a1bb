import sys, torch
sys.path.insert(0, "/root/repo")
from nnaudio_amd import engine, features
import os
B = int(os.environ.get("B", "64"))
x = torch.randn(B, 441000, device="cuda")
m = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to("cuda")
m.precision = "bf16x3"
sup = m._support.get(m.cqt_kernels_real, m.cqt_kernels_imag)
sc = torch.sqrt(m.lenghts)
def run(dbg):
    return engine.framed_gemm(x, m.cqt_kernels_real, m.cqt_kernels_imag, hop=512, pad=16384, pad_mode=2,
                              epilogue=engine.EPI_MAGNITUDE, row_scale=sc, row_support=sup, precision="bf16x3", _debug=dbg)
run(0x1000000)
torch.cuda.synchronize()
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dbg in [int(a, 0) for a in sys.argv[1:]] or [0, 0x800000]:
    print("debug %#x: %.4f ms" % (dbg, timeit(lambda: run(dbg))))
