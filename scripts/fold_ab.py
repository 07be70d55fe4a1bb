"""A/B of benchmarking-build bits on the folded STFT step (cfg2, bf16x3): python scripts/fold_ab.py 0 0x400 ..."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
bits = [int(a, 0) for a in sys.argv[1:]] or [0]
x = torch.randn(64, 441000, device="cuda")
st = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to("cuda")
prep = engine.prepare_basis(st.wcos, st.wsin, "bf16x3", hop=512)
kw = dict(hop=512, pad=1024, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, precision="bf16x3")
A = 0x10000000  # routes the call to the benchmarking build
def timeit(fn, n=100, w=30):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(3):
    for b in bits:
        print("rep %d bits %#x: %.4f ms" % (rep, b, timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, _debug=A | b, **kw, **prep))))
