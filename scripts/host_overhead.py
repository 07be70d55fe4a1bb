"""Host-side cost of a forward (launch-bound regime): small inputs, many calls, wall clock."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nnaudio_amd
from nnaudio_amd import features
for prec in (sys.argv[1:] or ["fp32", "bf16x3"]):
    nnaudio_amd.set_precision(prec)
    for name, m, L in (("STFT 512/128", features.STFT(n_fft=512, hop_length=128, output_format="Magnitude", verbose=False), 16000),
                       ("Mel 1024/512", features.MelSpectrogram(sr=22050, n_fft=1024, hop_length=512, verbose=False), 22050),
                       ("CQT1992v2 84", features.CQT1992v2(sr=22050, hop_length=512, n_bins=84, verbose=False), 110250),
                       ("CQT2010v2 84", features.CQT2010v2(sr=22050, hop_length=512, n_bins=84, verbose=False), 110250)):
        m = m.to("cuda")
        x = torch.randn(1, L, device="cuda")
        for _ in range(20): m(x)
        torch.cuda.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n): m(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%-7s %-14s host %.1f us per forward (enqueue), %.1f us incl. drain" % (prec, name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
