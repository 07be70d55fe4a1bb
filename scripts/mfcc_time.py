"""MFCC cfg3 (256 x 5 s, n_fft 1024, 128 mels, 20 coefficients): the tail in one launch vs the two calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nnaudio_amd import engine, features

m = features.MFCC(sr=22050, n_mfcc=20, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to("cuda:0")
mel = features.MelSpectrogram(sr=22050, n_fft=1024, n_mels=128, hop_length=512, verbose=False).to("cuda:0")
x = torch.randn(256, 110250, device="cuda:0")


def timeit(fn, n=300):
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def graphed(mod):
    """the forward recorded into a HIP graph: replay time = device time without the host's launch path"""
    mod(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            y = mod(x)
    torch.cuda.current_stream().wait_stream(st)
    return g, y


with torch.no_grad():
    print("Mel          eager %.4f ms" % timeit(lambda: mel(x)))
    g, _ = graphed(mel)
    print("Mel          graph %.4f ms" % timeit(g.replay))
    print("MFCC fused   eager %.4f ms" % timeit(lambda: m(x)))
    g, _ = graphed(m)
    print("MFCC fused   graph %.4f ms" % timeit(g.replay))
    engine.set_mfcc_fused(False)
    print("MFCC 2 calls eager %.4f ms" % timeit(lambda: m(x)))
    g, _ = graphed(m)
    print("MFCC 2 calls graph %.4f ms" % timeit(g.replay))
