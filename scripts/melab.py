import sys, torch
sys.path.insert(0, ".")
import nnaudio_amd
from nnaudio_amd import engine, features
sys.path.insert(0, "scripts")
from kbench import timeit
DEV="cuda:0"
B, L = 256, 110250
x = torch.randn(B, L, device=DEV)
m = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to(DEV)
st = m.stft
prep = engine.prepare_basis(st.wcos, st.wsin, "bf16x3", hop=512)
sup, cov = engine.filterbank_support(m.mel_basis)
for _ in range(30): m(x)
for dbg, what in ((0, "split"), (0x400000, "unsplit")):
    ms = timeit(lambda: engine.framed_gemm(x, st.wcos, st.wsin, hop=512, pad=512, pad_mode=2, epilogue=engine.EPI_POWER,
                                           power=2.0, precision="bf16x3", fb=m.mel_basis, fb_support=sup, _debug=dbg, **prep), n=50, w=20)
    print("mel cfg3 fused, %s: %.3f ms" % (what, ms))
