"""CPU oracle for the spectrogram hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy restatement of what the reference (KinWaiCheuk/nnAudio v0.3.3,
``/root/reference/Installation/nnAudio``) computes in ``forward`` for
STFT / MelSpectrogram / Gammatonegram / CQT1992v2 / CQT2010v2 / VQT, written as
"pad -> frame -> contract with the basis -> pointwise epilogue" on explicit arrays.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the shipped package (``nnaudio_amd``) never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here
against (a) the reference's own asserted ground-truth arrays
(``tests/ground-truths/*-cqt-{1992,2010}-{mag,complex,phase}*.npy``, the ten files its
``tests/test_cqt.py:94-262`` asserts, copied to ``tests/golden/ref_ground_truths``) and
(b) outputs of the reference itself run in the authoring container on seeded inputs
(``tests/golden/fwd_*.npz``, produced by ``oracle/gen_golden.py``).

All functions take the *constant operands* (bases, taps, lengths) as arguments, so the
oracle is independent of how the product builds them; accumulation is float64 unless
``acc=np.float32`` is requested (the cpu_baseline leg times the float32/BLAS form).

Each function cites the reference lines it follows.
"""
import warnings

import numpy as np


# --------------------------------------------------------------------------- #
# input plumbing
# --------------------------------------------------------------------------- #
def broadcast_dim(x):
    """(L,) / (B,L) / (B,1,L) -> (B,L)  [utils.py:206-222; the singleton channel the
    reference keeps for conv1d is dropped here]."""
    x = np.asarray(x)
    if x.ndim == 1:
        return x[None, :]
    if x.ndim == 2:
        return x
    if x.ndim == 3:
        if x.shape[1] != 1:
            raise ValueError("expected a single channel")
        return x[:, 0, :]
    raise ValueError("Only support input with shape = (batch, len) or shape = (len)")


def pad_signal(x, pad, mode):
    """Zero ('constant') or mirror-without-edge-repeat ('reflect') padding of the last
    axis [nn.ConstantPad1d / nn.ReflectionPad1d as used at stft.py:278-289,
    cqt.py:740-746, 1065-1068, vqt.py:175-179]."""
    if pad == 0:
        return x
    if mode == "constant":
        return np.pad(x, ((0, 0), (pad, pad)), mode="constant")
    if mode == "reflect":
        if pad >= x.shape[-1]:
            raise RuntimeError(
                "Padding size should be less than the corresponding input dimension"
            )
        return np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    raise ValueError("unknown pad mode %r" % (mode,))


def frames(xp, K, hop):
    """(B, Lp) -> strided view (B, T, K), frame t = xp[:, t*hop : t*hop+K]."""
    B, Lp = xp.shape
    if Lp < K:
        raise RuntimeError("Kernel size can't be greater than actual input size")
    T = (Lp - K) // hop + 1
    s0, s1 = xp.strides
    return np.lib.stride_tricks.as_strided(
        xp, shape=(B, T, K), strides=(s0, s1 * hop, s1), writeable=False
    )


def correlate_strided(xp, basis, hop, acc=np.float64):
    """out[b,f,t] = sum_n xp[b, t*hop+n] * basis[f,n]   (= F.conv1d(x, basis, stride=hop),
    stft.py:290-293, cqt.py:749-750, utils.py:518-519).  Returns ``acc`` dtype."""
    fr = frames(np.ascontiguousarray(xp, dtype=acc), basis.shape[-1], hop)
    w = np.asarray(basis, dtype=acc)
    B, T, K = fr.shape
    out = np.empty((B, w.shape[0], T), dtype=acc)
    # chunk frames so the materialised (T,K) block stays small
    step = max(1, (1 << 24) // max(K, 1))
    for b in range(B):
        for t0 in range(0, T, step):
            blk = np.ascontiguousarray(fr[b, t0 : t0 + step])  # (t, K)
            out[b, :, t0 : t0 + step] = w @ blk.T
    return out


# --------------------------------------------------------------------------- #
# STFT family
# --------------------------------------------------------------------------- #
def stft(x, wsin, wcos, hop, center=True, pad_mode="reflect", output_format="Complex",
         trainable=False, acc=np.float64):
    """STFT.forward [stft.py:256-316].  ``wsin``/``wcos`` are the *windowed* bases
    (F,1,K) or (F,K).  Returns float32."""
    x = broadcast_dim(x)
    wsin = np.asarray(wsin).reshape(wsin.shape[0], -1)
    wcos = np.asarray(wcos).reshape(wcos.shape[0], -1)
    K = wsin.shape[-1]
    pad = K // 2
    if center:
        if pad_mode == "constant":
            xp = pad_signal(x, pad, "constant")
        elif pad_mode == "reflect":
            if x.shape[-1] < pad:
                raise AssertionError(
                    "Signal length shorter than reflect padding length (n_fft // 2)."
                )
            xp = pad_signal(x, pad, "reflect")
        else:
            raise UnboundLocalError("padding")
    else:
        xp = x
    im = correlate_strided(xp, wsin, hop, acc)
    re = correlate_strided(xp, wcos, hop, acc)
    if output_format == "Magnitude":
        p = re * re + im * im
        if trainable:
            p = p + 1e-8
        return np.sqrt(p).astype(np.float32)
    if output_format == "Complex":
        return np.stack((re, -im), -1).astype(np.float32)
    if output_format == "Phase":
        return np.arctan2(-im + 0.0, re).astype(np.float32)
    raise ValueError(output_format)


def filterbank_spectrogram(x, wsin, wcos, hop, fb, power=2.0, center=True,
                           pad_mode="reflect", trainable_stft=False, acc=np.float64):
    """MelSpectrogram.forward / Gammatonegram.forward [mel.py:171-189,
    gammatone.py:171-189]: matmul(fb, STFT_magnitude ** power)."""
    x = broadcast_dim(x)
    wsin = np.asarray(wsin).reshape(wsin.shape[0], -1)
    wcos = np.asarray(wcos).reshape(wcos.shape[0], -1)
    K = wsin.shape[-1]
    pad = K // 2
    if center:
        if pad_mode == "reflect" and x.shape[-1] < pad:
            raise AssertionError(
                "Signal length shorter than reflect padding length (n_fft // 2)."
            )
        xp = pad_signal(x, pad, pad_mode)
    else:
        xp = x
    im = correlate_strided(xp, wsin, hop, acc)
    re = correlate_strided(xp, wcos, hop, acc)
    p = re * re + im * im
    if trainable_stft:
        p = p + 1e-8
    spec = np.sqrt(p) ** power
    return np.matmul(np.asarray(fb, dtype=acc), spec).astype(np.float32)


# --------------------------------------------------------------------------- #
# CQT family
# --------------------------------------------------------------------------- #
def _normalisation_rows(lengths, normalization_type, extra=1.0):
    lengths = np.asarray(lengths, dtype=np.float32)
    if normalization_type == "librosa":
        return np.sqrt(lengths).astype(np.float64) * extra
    if normalization_type == "convolutional":
        return np.full(lengths.shape, extra, dtype=np.float64)
    if normalization_type == "wrap":
        return np.full(lengths.shape, 2.0 * extra, dtype=np.float64)
    raise ValueError(
        "The normalization_type %r is not part of our current options."
        % normalization_type
    )


def _cqt_epilogue(re, im, output_format, trainable):
    if output_format == "Magnitude":
        p = re * re + im * im
        if trainable:
            p = p + 1e-8
        return np.sqrt(p).astype(np.float32)
    if output_format == "Complex":
        return np.stack((re, im), -1).astype(np.float32)
    if output_format == "Phase":
        ang = np.arctan2(im, re)
        return np.stack((np.cos(ang), np.sin(ang)), -1).astype(np.float32)
    raise ValueError(output_format)


def cqt1992v2(x, kern_real, kern_imag, lengths, hop, center=True, pad_mode="reflect",
              output_format="Magnitude", normalization_type="librosa", trainable=False,
              acc=np.float64):
    """CQT1992v2.forward [cqt.py:712-780]."""
    x = broadcast_dim(x)
    kr = np.asarray(kern_real).reshape(kern_real.shape[0], -1)
    ki = np.asarray(kern_imag).reshape(kern_imag.shape[0], -1)
    K = kr.shape[-1]
    if center:
        if pad_mode not in ("constant", "reflect"):
            raise UnboundLocalError("padding")
        xp = pad_signal(x, K // 2, pad_mode)
    else:
        xp = x
    re = correlate_strided(xp, kr, hop, acc)
    im = -correlate_strided(xp, ki, hop, acc)
    scale = _normalisation_rows(lengths, normalization_type)[None, :, None]
    re = re * scale
    im = im * scale
    return _cqt_epilogue(re, im, output_format, trainable)


def fir_decimate(x, taps, stride):
    """conv1d(x, taps, stride=stride, padding=(len(taps)-1)//2) [utils.py:73-124]."""
    x = broadcast_dim(x).astype(np.float64)
    taps = np.asarray(taps, dtype=np.float64).reshape(-1)
    pad = (taps.shape[0] - 1) // 2
    xp = np.pad(x, ((0, 0), (pad, pad)))
    return correlate_strided(xp, taps[None, :], stride)[:, 0, :].astype(np.float32)


def cqt_octave(x, kr, ki, hop, pad_mode, acc=np.float64):
    """get_cqt_complex [utils.py:498-521]: pad by K//2 (mirror; zero-pad with a warning
    when mirroring is impossible), then (re, im) = (x*kr, -(x*ki)) at stride hop."""
    K = kr.shape[-1]
    try:
        xp = pad_signal(x, K // 2, pad_mode)
    except RuntimeError:
        warnings.warn(
            "padding with reflection mode might not be the best choice, try using constant padding",
            UserWarning,
        )
        xp = pad_signal(x, K // 2, "constant")
    return correlate_strided(xp, kr, hop, acc), -correlate_strided(xp, ki, hop, acc)


def cqt2010v2(x, kern_real, kern_imag, lengths, hop, n_bins, n_octaves, lowpass,
              early_taps=None, downsample_factor=1, pad_mode="reflect",
              output_format="Magnitude", normalization_type="librosa", trainable=False,
              acc=np.float64):
    """CQT2010v2.forward [cqt.py:1070-1139].  ``hop`` is the module's *post early
    down-sampling* hop (``self.hop_length`` after ``__init__``); one top-octave kernel bank
    is re-used for every octave, the signal is halved between octaves."""
    x = broadcast_dim(x).astype(np.float32)
    kr = np.asarray(kern_real).reshape(kern_real.shape[0], -1)
    ki = np.asarray(kern_imag).reshape(kern_imag.shape[0], -1)
    if early_taps is not None:
        x = fir_decimate(x, early_taps, int(downsample_factor))
    banks = [(kr, ki)] * n_octaves
    return _octave_recursion(x, banks, lengths, hop, n_bins, lowpass, downsample_factor,
                             pad_mode, output_format, normalization_type, trainable, acc)


def vqt(x, banks, lengths, hop, n_bins, lowpass, early_taps=None, downsample_factor=1,
        pad_mode="reflect", output_format="Magnitude", normalization_type="librosa",
        trainable=False, acc=np.float64):
    """VQT.forward [vqt.py:143-215]; ``banks[i] = (real_i, imag_i)`` is octave i's own
    kernel bank (i = 0 is the top octave)."""
    x = broadcast_dim(x).astype(np.float32)
    if early_taps is not None:
        x = fir_decimate(x, early_taps, int(downsample_factor))
    banks = [
        (np.asarray(r).reshape(r.shape[0], -1), np.asarray(i).reshape(i.shape[0], -1))
        for r, i in banks
    ]
    return _octave_recursion(x, banks, lengths, hop, n_bins, lowpass, downsample_factor,
                             pad_mode, output_format, normalization_type, trainable, acc)


def _octave_recursion(x, banks, lengths, hop, n_bins, lowpass, downsample_factor,
                      pad_mode, output_format, normalization_type, trainable, acc):
    xd = x
    blocks_re, blocks_im = [], []
    for i, (kr, ki) in enumerate(banks):
        if i > 0:
            hop = hop // 2
            xd = fir_decimate(xd, lowpass, 2)
        re, im = cqt_octave(xd, kr, ki, hop, pad_mode, acc)
        blocks_re.insert(0, re)
        blocks_im.insert(0, im)
    T = {b.shape[-1] for b in blocks_re}
    if len(T) != 1:
        raise RuntimeError(
            "Sizes of tensors must match except in dimension 1 (octave frame counts %s)"
            % sorted(T)
        )
    re = np.concatenate(blocks_re, 1)[:, -n_bins:]
    im = np.concatenate(blocks_im, 1)[:, -n_bins:]
    scale = _normalisation_rows(lengths, normalization_type, float(downsample_factor))
    re = re * scale[None, :, None]
    im = im * scale[None, :, None]
    return _cqt_epilogue(re, im, output_format, trainable)


# --------------------------------------------------------------------------- #
# sampled evaluation for full-size checks (cheap: only the requested frames)
# --------------------------------------------------------------------------- #
def gather_frames(x, clips, frames_idx, K, hop, pad, mode):
    """float64 (n, K) matrix of the frames (clip, t) of the virtually padded signal."""
    x = broadcast_dim(x)
    L = x.shape[-1]
    n = np.arange(K)[None, :]
    p = np.asarray(frames_idx)[:, None] * hop - pad + n
    if mode == "reflect":
        p = np.where(p < 0, -p, p)
        p = np.where(p >= L, 2 * (L - 1) - p, p)
        ok = np.ones_like(p, dtype=bool)
    else:
        ok = (p >= 0) & (p < L)
        p = np.clip(p, 0, L - 1)
    v = x[np.asarray(clips)[:, None], p].astype(np.float64)
    return np.where(ok, v, 0.0)


def sampled_complex(x, basis_re, basis_im, clips, frames_idx, hop, pad, mode):
    """(re, im) float64 arrays (n, F) for the sampled frames: re = fr @ basis_re.T,
    im = -(fr @ basis_im.T) (the sign convention of stft.py:308-311 / cqt.py:749-750)."""
    wr = np.asarray(basis_re, dtype=np.float64).reshape(basis_re.shape[0], -1)
    wi = np.asarray(basis_im, dtype=np.float64).reshape(basis_im.shape[0], -1)
    fr = gather_frames(x, clips, frames_idx, wr.shape[1], hop, pad, mode)
    return fr @ wr.T, -(fr @ wi.T)


def decimated_lengths(L, n_taps, levels):
    """Lengths of x_0 .. x_levels under repeated conv1d(stride=2, padding=(n_taps-1)//2)
    [utils.py:98-99]."""
    pad = (n_taps - 1) // 2
    out = [int(L)]
    for _ in range(levels):
        out.append((out[-1] + 2 * pad - n_taps) // 2 + 1)
    return out


def decimated_window(x1d, taps, level, lo, hi):
    """float64 ``x_level[lo:hi]`` of ONE clip, where x_0 = x and x_l = downsampling_by_2(x_{l-1})
    [utils.py:102-124], evaluated only on the span of x it depends on (the full-size checks
    cannot afford the whole recursion in float64).  Positions outside [0, len(x_level)) come
    back as 0 (that is what conv1d's zero padding feeds the next level)."""
    x1d = np.asarray(x1d)
    taps = np.asarray(taps, dtype=np.float64).reshape(-1)
    nt = taps.shape[0]
    pad = (nt - 1) // 2
    lens = decimated_lengths(x1d.shape[0], nt, level)
    n = hi - lo
    out = np.zeros(max(n, 0), dtype=np.float64)
    if n <= 0:
        return out
    a, b = max(lo, 0), min(hi, lens[level])
    if a >= b:
        return out
    if level == 0:
        out[a - lo:b - lo] = x1d[a:b].astype(np.float64)
        return out
    # x_l[i] = sum_n taps[n] * x_{l-1}[2 i + n - pad], zeros outside x_{l-1}
    plo = 2 * a - pad
    phi = 2 * (b - 1) - pad + nt
    prev = decimated_window(x1d, taps, level - 1, plo, phi)
    win = np.lib.stride_tricks.sliding_window_view(prev, nt)[::2][: b - a]
    out[a - lo:b - lo] = win @ taps
    return out


def sampled_octave_complex(x, banks, lengths, hop, n_bins, lowpass, clips, frames_idx,
                           pad_mode="reflect", normalization_type="librosa",
                           downsample_factor=1.0):
    """Float64 (re, im) of CQT2010v2 / VQT [cqt.py:1085-1115, vqt.py:160-201] for the sampled
    frames only: arrays (n, n_bins), rows ordered like the module output (lowest octave first),
    scaled by sqrt(lenghts) * downsample_factor.  ``banks[i] = (real_i, imag_i)`` with i = 0 the
    top octave (pass the same pair n_octaves times for CQT2010v2); ``hop`` is the post
    early-down-sampling hop and ``x`` the signal the octave loop starts from."""
    x = broadcast_dim(x)
    clips = np.asarray(clips)
    frames_idx = np.asarray(frames_idx)
    n_oct = len(banks)
    n_filters = banks[0][0].shape[0]
    drop = n_oct * n_filters - n_bins
    lens = decimated_lengths(x.shape[-1], np.asarray(lowpass).reshape(-1).shape[0], n_oct - 1)
    re = np.zeros((clips.shape[0], n_bins), dtype=np.float64)
    im = np.zeros_like(re)
    for i, (kr, ki) in enumerate(banks):
        kr = np.asarray(kr, dtype=np.float64).reshape(n_filters, -1)
        ki = np.asarray(ki, dtype=np.float64).reshape(n_filters, -1)
        K = kr.shape[1]
        h = hop // (2 ** i)
        Lo = lens[i]
        pad = K // 2
        mode = pad_mode
        if mode == "reflect" and pad >= Lo:  # get_cqt_complex falls back to zero padding
            mode = "constant"
        row0 = (n_oct - 1 - i) * n_filters - drop
        first = max(0, -row0)
        if first >= n_filters:
            continue
        for s, (c, t) in enumerate(zip(clips, frames_idx)):
            p = t * h - pad + np.arange(K)
            ok = np.ones(K, dtype=bool)
            if mode == "reflect":
                p = np.where(p < 0, -p, p)
                p = np.where(p >= Lo, 2 * (Lo - 1) - p, p)
            else:
                ok = (p >= 0) & (p < Lo)
                p = np.clip(p, 0, Lo - 1)
            lo_i, hi_i = int(p.min()), int(p.max()) + 1
            w = decimated_window(x[c], lowpass, i, lo_i, hi_i)
            fr = np.where(ok, w[p - lo_i], 0.0)
            re[s, row0 + first:row0 + n_filters] = kr[first:] @ fr
            im[s, row0 + first:row0 + n_filters] = -(ki[first:] @ fr)
    scale = _normalisation_rows(lengths, normalization_type, float(downsample_factor))[None, :]
    return re * scale, im * scale


def power_to_db(S, amin=1e-10, ref=1.0, top_db=80.0):
    """MFCC._power_to_db [mel.py:263-279]: 10*log10(max(S, amin)) - 10*log10(max(amin, |ref|)),
    floored at (per-clip maximum - top_db)."""
    S = np.asarray(S, dtype=np.float64)
    log_spec = 10.0 * np.log10(np.maximum(S, amin)) - 10.0 * np.log10(max(amin, abs(ref)))
    if top_db is not None:
        if top_db < 0:
            raise ValueError("top_db must be non-negative")
        bmax = log_spec.reshape(log_spec.shape[0], -1).max(axis=1)[:, None, None]
        log_spec = np.maximum(log_spec, bmax - top_db)
    return log_spec


def dct_ortho(x):
    """MFCC._dct(x, norm="ortho") [mel.py:281-307] over axis 1 of (B, N, T): the orthonormal
    DCT-II, written as the explicit cosine sum the reference evaluates through an FFT."""
    x = np.asarray(x, dtype=np.float64)
    N = x.shape[1]
    n = np.arange(N)[None, :]
    k = np.arange(N)[:, None]
    D = np.cos(np.pi * (2 * n + 1) * k / (2.0 * N)) * np.sqrt(2.0 / N)
    D[0] *= np.sqrt(0.5)
    return np.einsum("kn,bnt->bkt", D, x)


def mfcc(x, wsin, wcos, hop, mel_basis, n_mfcc, amin=1e-10, ref=1.0, top_db=80.0, **mel_kw):
    """MFCC.forward [mel.py:309-326]."""
    mel = filterbank_spectrogram(x, wsin, wcos, hop, mel_basis, **mel_kw).astype(np.float64)
    return dct_ortho(power_to_db(mel, amin, ref, top_db))[:, :n_mfcc, :].astype(np.float32)


def extend_fbins(X):
    """utils.py:63-70: append the conjugates of bins F-2 .. 1 above a one-sided spectrum."""
    upper = X[:, 1:-1][:, ::-1].copy()
    upper[..., 1] = -upper[..., 1]
    return np.concatenate((X, upper), axis=1)


def istft(X, kernel_cos, kernel_sin, window, n_fft, hop, center=True, onesided=True, length=None):
    """STFTBase.inverse_stft [stft.py:15-63]: frame synthesis with the (n_fft, 1, n_fft, 1)
    inverse kernels, window / n_fft, overlap-add, division by the window sum of squares where it
    exceeds 1e-10, trimming."""
    X = np.asarray(X, dtype=np.float64)
    if onesided:
        X = extend_fbins(X)
    C = np.asarray(kernel_cos, dtype=np.float64).reshape(kernel_cos.shape[0], -1)
    S = np.asarray(kernel_sin, dtype=np.float64).reshape(kernel_sin.shape[0], -1)
    w = np.asarray(window, dtype=np.float64).reshape(-1)
    real = np.einsum("nk,bkt->bnt", C, X[..., 0]) - np.einsum("nk,bkt->bnt", S, X[..., 1])
    real = real * w[None, :, None] / n_fft
    B, N, T = real.shape
    full = N + hop * (T - 1)
    y = np.zeros((B, full))
    wss = np.zeros(full)
    for t in range(T):
        y[:, t * hop:t * hop + N] += real[:, :, t]
        wss[t * hop:t * hop + N] += w ** 2
    nz = wss > 1e-10
    y[:, nz] /= wss[nz]
    pad = n_fft // 2
    if length is None:
        return y[:, pad:full - pad] if center else y
    return y[:, pad:pad + length] if center else y[:, :length]


def _complex_kernel_product(kr, ki, fr, fi):
    """complex_mul (utils.py:175-203): (kr + j ki) @ (fr + j fi) over the frequency axis."""
    kr = np.asarray(kr, dtype=np.float64)
    ki = np.asarray(ki, dtype=np.float64)
    return (np.einsum("kf,bft->bkt", kr, fr) - np.einsum("kf,bft->bkt", ki, fi),
            np.einsum("kf,bft->bkt", kr, fi) + np.einsum("kf,bft->bkt", ki, fr))


def _freq_domain_cqt(x, wsin, wcos, kr, ki, hop, pad_mode, center=True, warn_fallback=False):
    """get_cqt_complex2 with STFT kernels (utils.py:524-559) / the body of CQT1992.forward
    (cqt.py:204-221): un-windowed STFT of size K, then the complex kernel product."""
    wsin = np.asarray(wsin).reshape(wsin.shape[0], -1)
    wcos = np.asarray(wcos).reshape(wcos.shape[0], -1)
    K = wsin.shape[-1]
    if center:
        if warn_fallback and pad_mode == "reflect" and K // 2 >= x.shape[-1]:
            xp = pad_signal(x, K // 2, "constant")
        else:
            xp = pad_signal(x, K // 2, pad_mode)
    else:
        xp = x
    fr = correlate_strided(xp, wcos, hop, np.float64)
    fi = correlate_strided(xp, wsin, hop, np.float64)
    return _complex_kernel_product(kr, ki, fr, fi)


def _cqt_freq_norm(lenghts, width, normalization_type):
    lenghts = np.asarray(lenghts, dtype=np.float64)
    if normalization_type == "librosa":
        return np.sqrt(lenghts) / width
    if normalization_type == "convolutional":
        return np.ones_like(lenghts)
    if normalization_type == "wrap":
        return np.full_like(lenghts, 2.0 / width)
    raise ValueError("The normalization_type %r is not part of our current options." % normalization_type)


def cqt1992(x, wsin, wcos, kr, ki, lenghts, hop, center=True, pad_mode="reflect",
            output_format="Magnitude", normalization_type="librosa"):
    """CQT1992.forward [cqt.py:189-254]."""
    x = broadcast_dim(x)
    re, im = _freq_domain_cqt(x, wsin, wcos, kr, ki, hop, pad_mode, center)
    s = _cqt_freq_norm(lenghts, np.asarray(wsin).shape[-1], normalization_type)[None, :, None]
    if output_format == "Magnitude":
        return np.sqrt((re * s) ** 2 + (im * s) ** 2).astype(np.float32)
    if output_format == "Complex":
        return np.stack((re * s, -im * s), -1).astype(np.float32)
    a = np.arctan2(im, re)  # the reference takes the phase of the un-negated, un-normalised pair
    return np.stack((np.cos(a), np.sin(a)), -1).astype(np.float32)


def cqt2010(x, wsin, wcos, kr, ki, lenghts, hop, n_bins, n_octaves, lowpass, early_taps=None,
            downsample_factor=1, pad_mode="reflect", output_format="Magnitude",
            normalization_type="librosa"):
    """CQT2010.forward [cqt.py:481-555]."""
    x = broadcast_dim(x)
    if early_taps is not None:
        x = fir_decimate(x, early_taps, int(downsample_factor))
    blocks = []
    xd = x
    for i in range(n_octaves):
        if i > 0:
            hop = hop // 2
            xd = fir_decimate(xd, lowpass, 2)
        re, im = _freq_domain_cqt(xd, wsin, wcos, kr, ki, hop, pad_mode, True, warn_fallback=True)
        blocks.insert(0, np.stack((re, im), -1))
    C = np.concatenate(blocks, 1)[:, -n_bins:]
    C = C * _cqt_freq_norm(lenghts, np.asarray(wsin).shape[-1], normalization_type)[None, :, None, None]
    if output_format == "Magnitude":
        return np.sqrt((C ** 2).sum(-1)).astype(np.float32)
    if output_format == "Complex":
        return C.astype(np.float32)
    a = np.arctan2(C[..., 1], C[..., 0])
    return np.stack((np.cos(a), np.sin(a)), -1).astype(np.float32)
