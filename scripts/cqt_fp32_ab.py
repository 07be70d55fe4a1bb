import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
m = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to("cuda")
kr, ki = m.cqt_kernels_real, m.cqt_kernels_imag
sup = m._support.get(kr, ki)
frag = engine.frag_basis_f32(kr, ki)
sc = torch.sqrt(m.lenghts)
for B in (1, 4, 16, 32, 48, 64, 128):
    x = torch.randn(B, 441000, device="cuda")
    kw = dict(hop=512, pad=16384, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, row_scale=sc, row_support=sup, precision="fp32")
    a = timeit(lambda: engine.framed_gemm(x, kr, ki, **kw))
    b = timeit(lambda: engine.framed_gemm(x, kr, ki, basis_split=frag, **kw))
    print("B=%3d: fp32 tile kernel %.3f ms, fp32 strip kernel %.3f ms" % (B, a, b))
