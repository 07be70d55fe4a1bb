import sys, torch
sys.path.insert(0, "/root/repo")
from nnaudio_amd import features
DEV = "cuda:0"
g = torch.Generator().manual_seed(31)
x = torch.randn(9, 20000, generator=g).to(DEV)
grads = {}
for prec in ("fp32", None, "bf16x3"):
    m = features.STFT(n_fft=1024, hop_length=256, trainable=True, output_format="Magnitude", verbose=False).to(DEV)
    m.precision = prec
    y = m(x)
    w = torch.randn(y.shape, generator=torch.Generator().manual_seed(32)).to(DEV)
    (y * w).sum().backward()
    grads[prec] = (m.wcos.grad.clone().double(), m.wsin.grad.clone().double())
# float64 truth
wc = m.wcos.detach().double().reshape(513, 1024).requires_grad_(True)
ws = m.wsin.detach().double().reshape(513, 1024).requires_grad_(True)
xp = torch.nn.functional.pad(x.double()[:, None, :], (512, 512), mode="reflect")[:, 0]
fr = xp.unfold(1, 1024, 256)  # (B, T, K)
re = torch.einsum("btk,fk->bft", fr, wc); im = torch.einsum("btk,fk->bft", fr, ws)
y2 = torch.sqrt(re ** 2 + im ** 2 + 1e-8)
(y2 * w.double()).sum().backward()
for prec in grads:
    for a, b, n in zip(grads[prec], (wc.grad, ws.grad), ("wcos", "wsin")):
        print(prec, n, "max|d| / max = %.3e" % float((a.reshape(513, 1024) - b).abs().max() / b.abs().max()))
