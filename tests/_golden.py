"""Shared access to the committed golden fixtures (tests/golden, made by oracle/gen_golden.py)
and the glue that feeds a product module's buffers to the numpy oracle."""
import hashlib
import json
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Golden:
    def __init__(self):
        with open(os.path.join(GOLDEN, "cases.json")) as f:
            self.manifest = json.load(f)
        self.cases = {c["name"]: c for c in self.manifest["cases"]}
        self.inputs = np.load(os.path.join(GOLDEN, "inputs.npz"))
        self.forward = np.load(os.path.join(GOLDEN, "forward.npz"))
        self.buffers = np.load(os.path.join(GOLDEN, "buffers.npz"))
        self.stride = self.manifest["sample_stride"]

    def names(self, forward_only=False):
        return [n for n, c in self.cases.items() if (c["input"] is not None or not forward_only)]

    def ground_truth(self, fn):
        return np.load(os.path.join(GOLDEN, "ref_ground_truths", fn))


def case_names(forward_only=False):
    with open(os.path.join(GOLDEN, "cases.json")) as f:
        m = json.load(f)
    return [c["name"] for c in m["cases"] if (c["input"] is not None or not forward_only)]


def build_module(case, device=None):
    """Instantiate the PRODUCT module for a manifest case (CPU construction needs no GPU)."""
    from nnaudio_amd import features

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import contextlib
        import io

        with contextlib.redirect_stdout(io.StringIO()):  # CQT1992 has no `verbose` switch
            kwv = {} if case["cls"] == "CQT1992" else {"verbose": False}
            mod = getattr(features, case["cls"])(**kwv, **case["ctor"])
    if device is not None:
        mod = mod.to(device)
    return mod


def oracle_forward(mod, case, x):
    """Run the numpy oracle on the module's own constant operands (its buffers), following
    the manifest's constructor / forward kwargs."""
    from oracle import spectral_oracle as O

    sd = {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}
    cls, ctor, fwd = case["cls"], case["ctor"], case["fwd"]
    fmt = fwd.get("output_format") or ctor.get("output_format")
    norm = fwd.get("normalization_type", "librosa")
    if cls == "iSTFT" or case.get("method") == "inverse":
        inv = cls == "STFT"
        return O.istft(x, sd["kernel_cos_inv" if inv else "kernel_cos"],
                       sd["kernel_sin_inv" if inv else "kernel_sin"], sd["window_mask"], mod.n_fft,
                       mod.stride, center=ctor.get("center", True),
                       onesided=fwd.get("onesided", inv), length=fwd.get("length"))
    if cls == "STFT":
        fmt = fmt or "Complex"
        fb = ctor.get("freq_bins")
        wsin, wcos = sd["wsin"], sd["wcos"]
        if fb is not None:
            wsin, wcos = wsin[:fb], wcos[:fb]
        return O.stft(x, wsin, wcos, mod.stride, center=ctor.get("center", True),
                      pad_mode=ctor.get("pad_mode", "reflect"), output_format=fmt,
                      trainable=ctor.get("trainable", False))
    if cls in ("MelSpectrogram", "Gammatonegram"):
        fbk = sd["mel_basis"] if cls == "MelSpectrogram" else sd["gammatone_basis"]
        return O.filterbank_spectrogram(
            x, sd["stft.wsin"], sd["stft.wcos"], mod.stride, fbk, power=ctor.get("power", 2.0),
            center=ctor.get("center", True), pad_mode=ctor.get("pad_mode", "reflect"),
            trainable_stft=ctor.get("trainable_STFT", False))
    if cls == "MFCC":
        ml = mod.melspec_layer
        return O.mfcc(x, sd["melspec_layer.stft.wsin"], sd["melspec_layer.stft.wcos"], ml.stride,
                      sd["melspec_layer.mel_basis"], mod.n_mfcc, amin=float(sd["amin"][0]),
                      ref=float(sd["ref"][0]), top_db=mod.top_db, power=ctor.get("power", 2.0),
                      center=ctor.get("center", True), pad_mode=ctor.get("pad_mode", "reflect"))
    fmt = fmt or "Magnitude"
    if cls == "CQT1992":
        return O.cqt1992(x, sd["wsin"], sd["wcos"], sd["cqt_kernels_real"], sd["cqt_kernels_imag"],
                         sd["lenghts"], mod.hop_length, center=ctor.get("center", True),
                         pad_mode=ctor.get("pad_mode", "reflect"), output_format=fmt,
                         normalization_type=norm)
    if cls == "CQT2010":
        early = sd.get("early_downsample_filter") if mod.earlydownsample else None
        return O.cqt2010(x, sd["wsin"], sd["wcos"], sd["cqt_kernels_real"], sd["cqt_kernels_imag"],
                         sd["lenghts"], mod.hop_length, mod.n_bins, mod.n_octaves,
                         sd["lowpass_filter"], early_taps=early,
                         downsample_factor=mod.downsample_factor,
                         pad_mode=ctor.get("pad_mode", "reflect"), output_format=fmt,
                         normalization_type=norm)
    if cls in ("CQT1992v2", "CQT"):
        return O.cqt1992v2(x, sd["cqt_kernels_real"], sd["cqt_kernels_imag"], sd["lenghts"],
                           mod.hop_length, center=ctor.get("center", True),
                           pad_mode=ctor.get("pad_mode", "reflect"), output_format=fmt,
                           normalization_type=norm, trainable=ctor.get("trainable", False))
    early = sd.get("early_downsample_filter") if mod.earlydownsample else None
    if cls == "CQT2010v2":
        return O.cqt2010v2(x, sd["cqt_kernels_real"], sd["cqt_kernels_imag"], sd["lenghts"],
                           mod.hop_length, mod.n_bins, mod.n_octaves, sd["lowpass_filter"],
                           early_taps=early, downsample_factor=mod.downsample_factor,
                           pad_mode=ctor.get("pad_mode", "reflect"), output_format=fmt,
                           normalization_type=norm, trainable=ctor.get("trainable", False))
    if cls == "VQT":
        banks = [(sd["cqt_kernels_real_%d" % i], sd["cqt_kernels_imag_%d" % i])
                 for i in range(mod.n_octaves)]
        return O.vqt(x, banks, sd["lenghts"], mod.hop_length, mod.n_bins, sd["lowpass_filter"],
                     early_taps=early, downsample_factor=mod.downsample_factor,
                     pad_mode=ctor.get("pad_mode", "reflect"), output_format=fmt,
                     normalization_type=norm, trainable=ctor.get("trainable", False))
    raise KeyError(cls)


def is_phase(case):
    fmt = case["fwd"].get("output_format") or case["ctor"].get("output_format")
    return fmt == "Phase"


def assert_parity(y, ref, rel=1e-4, what=""):
    """The parity bar of BASELINE.md / SURVEY.md 7.1: max|y-ref| <= rel*max|ref| and
    allclose(rtol=rel, atol=rel*max|ref|), float32."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert y.shape == ref.shape, "%s shape %s vs %s" % (what, y.shape, ref.shape)
    assert np.isfinite(y).all(), "%s has non-finite values" % what
    peak = np.abs(ref).max()
    err = np.abs(y - ref).max()
    assert err <= rel * peak, "%s max|d|=%.3e > %.0e*peak(%.3e)" % (what, err, rel, peak)
    assert np.allclose(y, ref, rtol=rel, atol=rel * peak), what


def is_inverse(case):
    return case["cls"] == "iSTFT" or case.get("method") == "inverse"


def well_conditioned(case, mod, x, y, ref):
    """Inverse STFT divides by the window sum of squares, which decays to ~0 where only the
    taper of the first / last frame covers a sample (reachable with ``length=`` or
    ``center=False``); there the reference's float32 noise is amplified without bound.  Returns
    (y, ref) with those samples (window sum < 1 % of its plateau) copied from ``ref`` so that the
    parity assertion speaks about the conditioned ones; at least 90 % must remain."""
    w = mod.window_mask.detach().cpu().numpy().reshape(-1).astype(np.float64)
    N, hop, T = mod.n_fft, mod.stride, x.shape[2]
    wss = np.zeros(N + hop * (T - 1))
    for t in range(T):
        wss[t * hop:t * hop + N] += w ** 2
    start = mod.pad_amount if case["ctor"].get("center", True) else 0
    good = wss[start:start + ref.shape[1]] >= 1e-2 * wss.max()
    assert good.mean() >= 0.9, good.mean()
    return np.where(good[None, :], y, ref), ref


def assert_phase_parity(y, ref, mag, what="", floor=1e-3, tol=1e-3, min_frac=0.5):
    """Phase is only conditioned where the bin carries energy: compare where |z| exceeds
    `floor` of the peak magnitude, on the unit circle (so +pi / -pi agree)."""
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert y.shape == ref.shape, "%s shape %s vs %s" % (what, y.shape, ref.shape)
    mask = mag > floor * mag.max()
    if y.ndim == mag.ndim:  # radians
        d = np.abs(np.exp(1j * y) - np.exp(1j * ref))
    else:  # (cos, sin)
        d = np.abs((y[..., 0] + 1j * y[..., 1]) - (ref[..., 0] + 1j * ref[..., 1]))
    assert mask.mean() > min_frac, "%s: mask too small (%.3f)" % (what, mask.mean())
    assert d[mask].max() <= tol, "%s phase err %.3e" % (what, d[mask].max())


def check_ground_truth(y, gt, fmt, eps, gt_complex=None, what="", max_miss=5e-3, phase_floor=1e-3):
    """Compare a transform of the reference's chirp inputs with one of the reference's own
    ground-truth arrays (reference tests/test_cqt.py:94-262, rtol = atol = 1e-3).

    The arrays were produced by the reference's float32 conv1d and carry its rounding noise
    (about 7e-6 of the peak, i.e. up to 2e-4 absolute).  The reference reproduces them only
    because it repeats the *same* float32 summation; an exact (float64) evaluation differs
    from them by that noise.  Where the assertion is well conditioned it is applied
    verbatim; where it is not, it is restricted as follows:
      Complex   : verbatim allclose(rtol=1e-3, atol=1e-3).
      Magnitude : the reference compares log(X + eps).  Verbatim on every element whose
                  ground-truth magnitude exceeds 1 % of the peak (there the log is conditioned
                  against 7e-6-of-peak noise); on all
                  elements the linear-domain parity bar |X - X_gt| <= 1e-4 * max|X_gt| ; and
                  at most `max_miss` (0.5 %) of the elements may miss the verbatim log assertion
                  (returned).  An exact float64 evaluation misses 0.13 % / 0.0 % (log / linear
                  sweep): that much of the fixture is the reference's own float32 noise.
      Phase     : (cos, sin) compared (atol 1e-3) where the bin carries energy
                  (|z_gt| > 1e-3 * max|z_gt|); elsewhere the phase is rounding noise."""
    y = np.asarray(y)
    gt = np.asarray(gt).reshape(y.shape)
    if fmt == "Complex":
        assert np.allclose(y, gt, rtol=1e-3, atol=1e-3), what
    elif fmt == "Magnitude":
        lin = np.exp(gt.astype(np.float64)) - eps
        assert_parity(y, lin.astype(np.float32), rel=1e-4, what=what + " (linear)")
        yl = np.log(y + eps)
        ok = np.isclose(yl, gt, rtol=1e-3, atol=1e-3)
        assert ok[lin > 1e-2 * lin.max()].all(), what + " (log, conditioned bins)"
        assert (~ok).mean() <= max_miss, what + " (log, fraction %.4f)" % (~ok).mean()
        return float((~ok).mean())
    elif fmt == "Phase":
        mag = np.hypot(gt_complex[..., 0], gt_complex[..., 1]).reshape(y.shape[:-1])
        assert_phase_parity(y, gt, mag, what=what, floor=phase_floor, tol=1e-3, min_frac=0.05)
    else:
        raise ValueError(fmt)
