"""Phase clock of one job of the strip kernel (benchmarking build): python scripts/strip_stamps.py"""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from nnaudio_amd import engine, features
x = torch.randn(64, 441000, device="cuda")
m = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False).to("cuda")
sup = m._support.get(m.cqt_kernels_real, m.cqt_kernels_imag)
sc = torch.sqrt(m.lenghts)
def run(dbg):
    return engine.framed_gemm(x, m.cqt_kernels_real, m.cqt_kernels_imag, hop=512, pad=16384, pad_mode=2,
                              epilogue=engine.EPI_MAGNITUDE, row_scale=sc, row_support=sup, precision="bf16x3", _debug=dbg)
extra = int(sys.argv[1], 0) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 0
for _ in range(3):
    run(0x2000000 | extra)
torch.cuda.synchronize()
ws, need = engine._last_workspace
tail = ws.view(torch.uint8)[need - 256 + 8:need].cpu().numpy().view("<u8")
t = [int(v) for v in tail[:22]]
names = ["job start", "tables", "prologue"] + ["sub-stage %d" % i for i in range(16)] + ["loop end", "slabs free", "epilogue done"]
if "--brief" in sys.argv:
    print("job %.2f us: prologue %.2f, loop %.2f, epilogue %.2f" % ((t[21] - t[0]) / 100.0, (t[2] - t[0]) / 100.0,
                                                                   (t[19] - t[2]) / 100.0, (t[21] - t[19]) / 100.0))
    sys.exit(0)
for i in range(1, 22):
    print("%-14s +%7.2f us  (total %7.2f)" % (names[i], (t[i] - t[i - 1]) / 100.0, (t[i] - t[0]) / 100.0))
