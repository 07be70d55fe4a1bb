import sys, os, torch
sys.path.insert(0, "/root/repo")
from nnaudio_amd import engine, features
def timeit(fn, n=40, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
m = features.STFT(n_fft=2048, hop_length=512, sr=44100, output_format="Magnitude", verbose=False).to("cuda")
x = torch.randn(64, 441000, device="cuda")
wc, ws = m.wcos, m.wsin
prep = engine.prepare_basis(wc, ws, "bf16x3", hop=512)
def run(dbg):
    return engine.framed_gemm(x, wc, ws, hop=512, pad=1024, pad_mode=engine.PAD_REFLECT, epilogue=engine.EPI_MAGNITUDE, precision="bf16x3", _debug=dbg, **prep)
for _ in range(30): run(0x10000000)
for rep in range(2):
    print("single launch %.4f ms, chunked (folded frames in the Infinity Cache) %.4f ms" % (timeit(lambda: run(0x10000000)), timeit(lambda: run(0x10000000 | 0x200000))))
