import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
dbg = int(sys.argv[1]); tile = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x = torch.randn(64, 441000, device="cuda:0")
m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to("cuda:0")
for _ in range(3):
    engine.framed_gemm(x, m.wcos[:1024], m.wsin[:1024], hop=512, pad=1024, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, tile=tile, _debug=dbg)
torch.cuda.synchronize()
