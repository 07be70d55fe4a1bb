"""STFT cfg2 bf16x3 steps with benchmarking-build bits (run under rocprofv3 --kernel-trace --stats):
0x40 pre-pass without global stores, 0x80 without global loads."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
bits = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
n_fft = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
B, L = (64, 441000) if n_fft == 2048 else (256, 110250)
x = torch.randn(B, L, device="cuda")
st = features.STFT(n_fft=n_fft, hop_length=512, output_format="Magnitude", verbose=False).to("cuda")
prep = engine.prepare_basis(st.wcos, st.wsin, "bf16x3", hop=512)
kw = dict(hop=512, pad=n_fft // 2, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, precision="bf16x3")
A = 0x10000000  # routes the call to the benchmarking build
for _ in range(60):
    engine.framed_gemm(x, st.wcos, st.wsin, _debug=A | bits, **kw, **prep)
torch.cuda.synchronize()
