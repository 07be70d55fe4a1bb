import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features
x = torch.randn(64, 441000, device="cuda:0")
m = features.STFT(n_fft=2048, hop_length=512, output_format="Magnitude", verbose=False).to("cuda:0")
nblk = 16 * 431
for dbg in (0x800 | 256, 0x800, 0x800 | 256 | 1):
    buf = torch.zeros(max(1024, nblk * 16), device="cuda:0")
    for _ in range(2):
        engine.framed_gemm(x, m.wcos[:1024], m.wsin[:1024], hop=512, pad=1024, pad_mode=2, epilogue=engine.EPI_MAGNITUDE, tile=1, _debug=dbg, row_scale=buf[:1024])
    torch.cuda.synchronize()
    d = buf[: nblk * 16].view(nblk, 4, 4).cpu()
    tot, vm, bar, plain = d[..., 0], d[..., 1], d[..., 2], d[..., 3]
    sel = plain[:, 0] > 0
    print("debug=%#x: plain blocks %.1f%%" % (dbg, 100 * sel.float().mean()))
    for name, mask in (("plain", sel), ("general", ~sel)):
        if mask.sum() == 0: continue
        print("   %-8s loop cycles/iter %.0f  vmcnt wait/iter %.0f  barrier wait/iter %.0f  (n=%d)" % (
            name, tot[mask].mean() / 64, vm[mask].mean() / 64, bar[mask].mean() / 64, int(mask.sum())))
