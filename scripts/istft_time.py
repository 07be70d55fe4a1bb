"""cfg2-sized inverse STFT, fused and in two launches (the command rocprofv3 wraps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
m = features.STFT(n_fft=2048, hop_length=512, iSTFT=True, output_format="Complex", verbose=False).cuda()
x = torch.randn(64, 441000, device="cuda")
X = m(x)
for fused in (True, False):
    engine.set_istft_fused(fused)
    for _ in range(30):
        y = m.inverse(X, length=441000)
torch.cuda.synchronize()
