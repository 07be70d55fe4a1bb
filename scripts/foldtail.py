"""A/B of the folded contraction with and without 128-frame tail tiles (benchmarking build):
python scripts/foldtail.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features

def timeit(fn, n=40, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for tag, n_fft, hop, sr, B, L in (("stft cfg2", 2048, 512, 44100, 64, 441000), ("mel-stft cfg3", 1024, 512, 22050, 256, 110250),
                                   ("stft B=16", 2048, 512, 44100, 16, 441000), ("stft B=4", 2048, 512, 44100, 4, 441000),
                                   ("stft B=100", 2048, 512, 44100, 100, 441000)):
    m = features.STFT(n_fft=n_fft, hop_length=hop, sr=sr, output_format="Magnitude", verbose=False).to("cuda")
    x = torch.randn(B, L, device="cuda")
    F = n_fft // 2 + 1
    wc, ws = m.wcos[:F], m.wsin[:F]
    prep = engine.prepare_basis(wc, ws, "bf16x3", hop=hop)
    def run(dbg):
        return engine.framed_gemm(x, wc, ws, hop=hop, pad=n_fft // 2, pad_mode=engine.PAD_REFLECT,
                                  epilogue=engine.EPI_MAGNITUDE, precision="bf16x3", _debug=dbg, **prep)
    for _ in range(20): run(0x10000000)
    a = timeit(lambda: run(0x10000000))
    b = timeit(lambda: run(0x10000000 | 0x8000000))
    a2 = timeit(lambda: run(0x10000000))
    print("%-14s with tail tiles %.4f / %.4f ms, 256-frame tiles only %.4f ms" % (tag, a, a2, b))
