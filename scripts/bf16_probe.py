"""bring-up probe for the bf16x3 kernel (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
DEV = "cuda:0"
torch.manual_seed(0)
B = int(os.environ.get("PB", "64"))
x = torch.randn(B, 441000, device=DEV)
m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to(DEV)
wc, ws = m.wcos[:1024].contiguous(), m.wsin[:1024].contiguous()
split = engine.split_basis(wc, ws)
kw = dict(hop=512, pad=1024, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, precision="bf16x3", basis_split=split)
ref = engine.framed_gemm(x, wc, ws, **kw)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for dbg in [int(a, 0) for a in (sys.argv[1:] or ["0", "16", "1", "17"])]:
    y = engine.framed_gemm(x, wc, ws, _debug=dbg, **kw)
    ms = t(lambda: engine.framed_gemm(x, wc, ws, _debug=dbg, **kw))
    print("dbg %#x: %.3f ms, max |y - ref| / peak = %.3e" % (dbg, ms, float((y - ref).abs().max() / ref.abs().max())))
