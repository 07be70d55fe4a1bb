"""Mel cfg3 in bf16x3: where the step goes (benchmarking build): fused step, without the epilogue,
plain power spectrum, pre-pass alone."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features

def timeit(fn, n=50, w=20):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B, L = 256, 110250
x = torch.randn(B, L, device="cuda")
m = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to("cuda")
st = m.stft
prep = engine.prepare_basis(st.wcos, st.wsin, "bf16x3", hop=512)
sup, cov = engine.filterbank_support(m.mel_basis)
kw = dict(hop=512, pad=512, pad_mode=2, epilogue=engine.EPI_POWER, power=2.0, precision="bf16x3")
A = 0x10000000  # (a bit nothing reads: routes the call to the benchmarking build)
print("fused mel step            %.3f ms" % timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, fb=m.mel_basis, fb_support=sup, _debug=A, **kw, **prep)))
print("  ... without epilogue    %.3f ms" % timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, fb=m.mel_basis, fb_support=sup, _debug=A | 0x40000, **kw, **prep)))
print("  ... without the walk   %.3f ms" % timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, fb=m.mel_basis, fb_support=sup, _debug=A | 0x400, **kw, **prep)))
print("  ... walk, no stores    %.3f ms" % timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, fb=m.mel_basis, fb_support=sup, _debug=A | 0x10000, **kw, **prep)))
print("  ... stores, no atomics %.3f ms" % timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, fb=m.mel_basis, fb_support=sup, _debug=A | 0x400000, **kw, **prep)))
print("power spectrum (no mel)   %.3f ms" % timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, _debug=A, **kw, **prep)))
print("  ... without epilogue    %.3f ms" % timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, _debug=A | 0x40000, **kw, **prep)))
print("module forward            %.3f ms" % timeit(lambda: m(x)))
