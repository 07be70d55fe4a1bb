"""CQT1992v2 against the REFERENCE'S OWN arithmetic -- torch's conv1d, CQT1992v2.forward restated on the product module's
buffers (reference cqt.py:740-772) -- and against the reference's own fixture assertions (reference tests/test_cqt.py:94-186:
allclose(log(X + 1e-5) | Complex | Phase, ground truth, rtol = atol = 1e-3)), VERBATIM, measured where the product runs.

VERDICT r4 asked for the CQT1992v2 fixture bar to be SET FROM THIS MEASUREMENT ("no worse than the reference on this
hardware") instead of a hand-picked 0.5 %.  Measured (scripts/ref_gpu_path_miss.py, scripts/cqt1992_verbatim.py,
profiles/r05/ref_gpu_path_miss.log, profiles/r05/cqt1992_verbatim.log):

    arithmetic                                                   log sweep   linear sweep
    reference order, torch conv1d, CPU (1 .. 8 threads)             0 %          0 %
    reference order, torch conv1d on the MI355X (MIOpen)            0 %          0 %
    exact float64 evaluation                                        0.13 %       0 %
    product fp32 = the default module (round 5: 16-row MFMA tiles,
      taps ascending through the MFMA's k lanes)                    0 %          0 %      <- all six assertions verbatim
    product fp32 before (32 x 32 x 2 tiles, taps 8q+s | 8q+4+s)     0.033 %      0 %
    product f16x3, natural tap order (staged dense kernel)          0.58 %       0.21 %
    product f16x3, strip kernel (hop-periodic tap order)            2.7 %        0.87 %
    product bf16x3                                                  57 %         74 %

The near-silent bins of the fixture (1e-9 of the peak) record the rounding of ONE SEQUENTIAL float32 FMA chain over the
taps, which oneDNN and MIOpen both perform (an exact evaluation misses them; a CPU emulation of that chain reproduces the
fixture: experiments/tap_order/emulate.py).  v_mfma_f32_16x16x4_f32 is itself an FMA chain through its four k lanes in
ascending order (experiments/tap_order/mfma_order.hip, bit-identical on random operands), so the support-aware fp32 tile
kernel, feeding tap 16q + 4s + lq to k lane lq of MFMA s, accumulates exactly that chain: the default module is
BIT-IDENTICAL to the library's sequential reference kernel and to torch's conv1d on this GPU (every element of both
sweeps; 83.5 % of the elements against oneDNN on the CPU, 7e-8 of the peak at most), and passes the reference's six
CQT1992v2 assertions verbatim.  The bar is therefore the reference's own: 0 misses (1e-4 allowed).  f16x3 (4.7e-7 of the
peak against float64 -- 200 x inside north_star's 1e-4, but 2.7 % of this fixture's silent bins) stays the opt-in
`module.precision = "f16x3"` and is named as such on the bench line (roofline_cqt84_f16x3, default_module: false)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from scipy.signal import chirp

from tests._golden import Golden, build_module

CASE = dict(cls="CQT1992v2", ctor=dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24, output_format="Magnitude"), fwd={})
SWEEPS = [("log", "logarithmic"), ("linear", "linear")]
# the bars the product's arithmetics are held to, from the measurements above: fp32 (the default) meets the reference's
# own -- no miss; f16x3 / bf16x3 are pinned so that they cannot grow unnoticed -- they do NOT meet a reference-derived bar
# and do not ship as the default
FP32_BAR = 1e-4


def _chirp(method):
    s = np.linspace(0, 1, 44100)
    return torch.from_numpy(chirp(s, 55, 1, 22050, method=method).astype(np.float32)[None, :])


def miss_fraction(y, gt, eps=1e-5):
    ok = np.isclose(np.log(np.asarray(y, dtype=np.float32) + eps), gt.reshape(y.shape), rtol=1e-3, atol=1e-3)
    return float((~ok).mean())


def reference_order(mod, x, device):
    """reference cqt.py:740-772 with torch's own conv1d on `device`"""
    kr, ki, ln = (t.to(device) for t in (mod.cqt_kernels_real, mod.cqt_kernels_imag, mod.lenghts))
    xp = F.pad(x.to(device)[:, None, :], (mod.kernel_width // 2, mod.kernel_width // 2), mode="reflect")
    re = F.conv1d(xp, kr, stride=mod.hop_length) * torch.sqrt(ln.view(-1, 1))
    im = -F.conv1d(xp, ki, stride=mod.hop_length) * torch.sqrt(ln.view(-1, 1))
    return torch.sqrt(re.pow(2) + im.pow(2)).float().cpu().numpy()


@pytest.mark.parametrize("sweep,method", SWEEPS)
def test_reference_conv1d_cpu_reproduces_its_fixture(sweep, method):
    g = Golden()
    mod = build_module(CASE)
    gt = g.ground_truth("%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep)
    miss = miss_fraction(reference_order(mod, _chirp(method), "cpu"), gt)
    print("reference order, torch conv1d on the CPU, %s sweep: %.5f miss" % (sweep, miss))
    assert miss <= 1e-4  # measured 0 at 1, 2, 4 and 8 threads


@pytest.mark.gpu
@pytest.mark.parametrize("sweep,method", SWEEPS)
def test_reference_conv1d_on_the_gpu_and_the_products_arithmetics(sweep, method):
    """The measurement the default precision of CQT1992v2 rests on, repeated on whatever GPU runs the suite."""
    g = Golden()
    dev = torch.device("cuda:0")
    x = _chirp(method)
    gt = g.ground_truth("%s-sweep-cqt-1992-mag-ground-truth.npy" % sweep)
    ref = miss_fraction(reference_order(build_module(CASE), x, dev), gt)
    mod = build_module(CASE, dev)
    got = {}
    for name, prec, hp in (("fp32", "fp32", None), ("f16x3 strip", "f16x3", True), ("f16x3 natural order", "f16x3", False)):
        mod.precision = prec
        if hp is not None:
            mod.hop_periodic = hp
        with torch.no_grad():
            got[name] = miss_fraction(mod(x.to(dev)).float().cpu().numpy(), gt)
    mod.precision = None
    mod.hop_periodic = True
    with torch.no_grad():
        default = miss_fraction(mod(x.to(dev)).float().cpu().numpy(), gt)
    print("%s sweep: reference conv1d on this GPU %.5f | product %s | default module %.5f"
          % (sweep, ref, ", ".join("%s %.5f" % kv for kv in got.items()), default))
    assert ref <= 1e-4                       # the reference's GPU path passes its own assertion (measured 0)
    assert default == got["fp32"]             # the module ships the arithmetic nearest to it ...
    assert got["fp32"] <= FP32_BAR            # ... and that one meets the reference's own bar (measured 0 / 0)
    assert got["fp32"] <= got["f16x3 natural order"] <= 0.009 and got["f16x3 strip"] <= 0.04


@pytest.mark.gpu
@pytest.mark.parametrize("sweep,method", SWEEPS)
def test_default_module_passes_the_references_assertions_verbatim(sweep, method):
    """reference tests/test_cqt.py:94-186, the three assertions of each sweep exactly as written there, on the module as it
    ships (no restriction to conditioned bins, no allowed fraction)."""
    g = Golden()
    dev = torch.device("cuda:0")
    x = _chirp(method).to(dev)
    for fmt, tag in (("Magnitude", "mag"), ("Complex", "complex"), ("Phase", "phase")):
        mod = build_module(dict(CASE, ctor=dict(CASE["ctor"], output_format=fmt)), dev)
        with torch.no_grad():
            X = mod(x)
        if fmt == "Magnitude":
            X = torch.log(X + 1e-5)
        gt = g.ground_truth("%s-sweep-cqt-1992-%s-ground-truth.npy" % (sweep, tag))
        X = X.cpu().numpy()
        assert np.allclose(X, gt.reshape(X.shape), rtol=1e-3, atol=1e-3), "%s sweep, %s" % (sweep, fmt)


@pytest.mark.gpu
@pytest.mark.parametrize("sweep,method", SWEEPS)
def test_default_module_is_one_float32_fma_chain_over_the_taps(sweep, method):
    """Bit identity of the default module's complex output with (i) the library's sequential reference kernel (one thread
    per output, fmaf over the taps in ascending order) -- asserted on every element; (ii) torch's conv1d on this GPU and on
    the CPU, the reference's two device paths (cqt.py:749-750) -- measured 100 % / 83.5 %, asserted with margin because
    those libraries are free to change their summation."""
    from nnaudio_amd import engine

    dev = torch.device("cuda:0")
    x = _chirp(method)
    case = dict(CASE, ctor=dict(CASE["ctor"], output_format="Complex"))
    mod = build_module(case, dev)
    with torch.no_grad():
        y = mod(x.to(dev))
        kr, ki = mod.cqt_kernels_real, mod.cqt_kernels_imag
        ref = engine.framed_gemm(x.to(dev), kr, ki, hop=mod.hop_length, pad=mod.kernel_width // 2, pad_mode=engine.PAD_REFLECT,
                                 epilogue=engine.EPI_COMPLEX, im_sign=-1.0, row_scale=torch.sqrt(mod.lenghts),
                                 precision="fp32", reference_kernel=True)
    assert torch.equal(y, ref), "the tile kernel left the sequential chain: %.3e" % float((y - ref).abs().max())
    y = y.cpu()
    cpu = build_module(case)
    same = {}
    for name, d in (("cpu", torch.device("cpu")), ("gpu", dev)):
        k_r, k_i, ln = (t.to(d) for t in (cpu.cqt_kernels_real, cpu.cqt_kernels_imag, cpu.lenghts))
        xp = F.pad(x.to(d)[:, None, :], (cpu.kernel_width // 2,) * 2, mode="reflect")
        re = (F.conv1d(xp, k_r, stride=cpu.hop_length) * torch.sqrt(ln.view(-1, 1))).cpu()
        im = (-F.conv1d(xp, k_i, stride=cpu.hop_length) * torch.sqrt(ln.view(-1, 1))).cpu()
        same[name] = float(((y[..., 0] == re) & (y[..., 1] == im)).float().mean())
        err = float(max((y[..., 0] - re).abs().max(), (y[..., 1] - im).abs().max()) / y.abs().max())
        assert err <= 2e-6, (name, err)
    # (the exact statement is the torch.equal against the library's sequential reference kernel above; what MIOpen / oneDNN
    # do is theirs to change: a measurement -- 1.0000 / 0.835 in rounds 5 and 6 -- with a floor that only catches a collapse)
    print("%s sweep: default module bit-identical to torch conv1d on %.4f (this GPU) / %.4f (CPU) of the elements" % (sweep, same["gpu"], same["cpu"]))
    assert same["gpu"] >= 0.5 and same["cpu"] >= 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("sweep,method", SWEEPS)
def test_cqt2010v2_default_module_passes_the_references_assertions_verbatim(sweep, method):
    """reference tests/test_cqt.py:188-262 (log(X + 1e-2) and Complex, rtol = atol = 1e-3), exactly as written there, on
    CQT2010v2 as it ships (f16x3, the streaming octave kernel)."""
    g = Golden()
    dev = torch.device("cuda:0")
    x = _chirp(method).to(dev)
    for fmt, tag in (("Magnitude", "mag"), ("Complex", "complex")):
        mod = build_module(dict(cls="CQT2010v2", ctor=dict(CASE["ctor"], output_format=fmt), fwd={}), dev)
        with torch.no_grad():
            X = mod(x)
        if fmt == "Magnitude":
            X = torch.log(X + 1e-2)
        gt = g.ground_truth("%s-sweep-cqt-2010-%s-ground-truth.npy" % (sweep, tag))
        X = X.cpu().numpy()
        assert np.allclose(X, gt.reshape(X.shape), rtol=1e-3, atol=1e-3), "%s sweep, %s" % (sweep, fmt)


@pytest.mark.gpu
@pytest.mark.parametrize("F,K,hop,B,L", [(40, 1000, 100, 3, 5000),     # K % 32 != 0: the peeled tail stage; ragged supports
                                         (84, 4096, 256, 2, 20000),    # two row blocks, centred supports (prefix runs)
                                         (7, 333, 64, 5, 3000),        # fewer rows than one 16-row tile
                                         (130, 2048, 512, 1, 30000)])  # three row blocks, one clip
def test_support_aware_fp32_kernel_is_the_sequential_chain_on_random_banks(F, K, hop, B, L):
    """The fp32 tile kernel of a bank WITH supports (every output = one float32 FMA chain over the taps in ascending order)
    against the library's reference kernel (one thread per output, fmaf over the taps): the same bits, for shapes that take
    the masked stages, the compiled prefix runs and the K-tail stage."""
    from nnaudio_amd import engine

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(F * K)
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    sup = np.zeros((F, 2), np.int32)
    for f in range(F):
        if K == 4096 or K == 2048:  # centred, shrinking with the row index (a CQT bank's shape)
            ln = max(8, int(K * 2.0 ** (-f / 12.0)))
            lo = (K - ln) // 2
            hi = lo + ln
        else:
            lo = int(rng.integers(0, K - 1))
            hi = int(rng.integers(lo + 1, K + 1))
        sup[f] = (lo, hi)
        wr[f, :lo] = wr[f, hi:] = 0.0
        wi[f, :lo] = wi[f, hi:] = 0.0
    x = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).to(dev)
    wr, wi, sup = torch.as_tensor(wr).to(dev), torch.as_tensor(wi).to(dev), torch.as_tensor(sup).to(dev)
    kw = dict(hop=hop, pad=K // 2, pad_mode=engine.PAD_REFLECT, epilogue=engine.EPI_COMPLEX, im_sign=-1.0, precision="fp32")
    y = engine.framed_gemm(x, wr, wi, row_support=sup, **kw)
    ref = engine.framed_gemm(x, wr, wi, reference_kernel=True, **kw)
    assert torch.equal(y, ref), float((y - ref).abs().max())
