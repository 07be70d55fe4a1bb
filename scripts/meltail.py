import sys, torch
sys.path.insert(0, "/root/repo")
import nnaudio_amd
from nnaudio_amd import features
nnaudio_amd.set_precision("bf16x3")
m = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to("cuda")
def timeit(fn, n=50, w=10):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in (256, 227, 190, 152):
    x = torch.randn(B, 110250, device="cuda")
    for _ in range(30): m(x)
    t = timeit(lambda: m(x))
    print("B=%d frames=%d tiles=%d: %.4f ms  (%.3f us per 1000 frames)" % (B, B * 216, -(-B * 216 // 256) * 4, t, t * 1e6 / (B * 216)))
