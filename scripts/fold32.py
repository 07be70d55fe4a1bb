import sys, torch
sys.path.insert(0, "/root/repo")
import nnaudio_amd
from nnaudio_amd import engine, features
m = features.STFT(n_fft=2048, hop_length=512, sr=44100, output_format="Magnitude", verbose=False).to("cuda")
x = torch.randn(64, 441000, device="cuda")
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
wc, ws = m.wcos, m.wsin
kw = dict(hop=512, pad=1024, pad_mode=engine.PAD_REFLECT, epilogue=engine.EPI_MAGNITUDE)
dense = engine.framed_gemm(x, wc, ws, precision="fp32", **kw)
prep = engine.prepare_basis(wc, ws, "fp32", hop=512)
print("prepared:", list(prep))
fold = engine.framed_gemm(x, wc, ws, precision="fp32", **kw, **prep)
torch.cuda.synchronize()
print("max |fold - dense| / peak = %.3e" % (float((fold - dense).abs().max()) / float(dense.abs().max())))
print("dense fp32 %.3f ms, folded fp32 %.3f ms" % (timeit(lambda: engine.framed_gemm(x, wc, ws, precision="fp32", **kw)),
                                                   timeit(lambda: engine.framed_gemm(x, wc, ws, precision="fp32", **kw, **prep))))
y = m(x)
print("module fp32: %.3f ms" % timeit(lambda: m(x)))
