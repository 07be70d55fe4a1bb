"""Shared host logic of the constant-Q family: per-call row scales, kernel supports and
the octave recursion of CQT2010v2 / VQT (reference: cqt.py:1070-1139, vqt.py:143-215)."""
import warnings

import torch

from .. import engine
from ..basis import decimated_length


def output_epilogue(output_format):
    if output_format == "Magnitude":
        return engine.EPI_MAGNITUDE
    if output_format == "Complex":
        return engine.EPI_COMPLEX
    if output_format == "Phase":
        return engine.EPI_PHASE_COSSIN
    return None


def normalisation_scale(lenghts, normalization_type, extra=1.0):
    """Per-bin multiplier applied to (re, im) before the epilogue: sqrt(lenghts) ('librosa'),
    1 ('convolutional') or 2 ('wrap'), times ``extra`` (the early down-sampling factor)."""
    if normalization_type == "librosa":
        s = torch.sqrt(lenghts)
    elif normalization_type == "convolutional":
        s = torch.ones_like(lenghts)
    elif normalization_type == "wrap":
        s = torch.full_like(lenghts, 2.0)
    else:
        raise ValueError(
            "The normalization_type %r is not part of our current options." % normalization_type
        )
    if extra != 1.0:
        # same association order as the reference: (CQT * factor) * sqrt(lenghts)
        s = s * float(extra)
    return s.to(torch.float32).contiguous()


class SupportCache:
    """[start, stop) of the non-zero taps of every kernel row, recomputed whenever the
    kernel tensors are replaced or modified in place (see ``engine.DerivedCache``)."""

    def __init__(self):
        self._cache = engine.DerivedCache()

    @staticmethod
    def _build(real, imag):
        nz = (real.reshape(real.shape[0], -1) != 0) | (imag.reshape(imag.shape[0], -1) != 0)
        K = nz.shape[1]
        idx = torch.arange(K, device=nz.device)
        big = torch.where(nz, idx, torch.full_like(idx, K))
        small = torch.where(nz, idx + 1, torch.zeros_like(idx))
        start = big.min(dim=1).values
        stop = small.max(dim=1).values
        start = torch.minimum(start, stop)
        return torch.stack((start, stop), 1).to(torch.int32).contiguous()

    def get(self, real, imag):
        return self._cache.get((real, imag), lambda: self._build(real, imag))


def octave_recursion(x, banks, lenghts, hop, n_bins, lowpass, downsample_factor, pad_mode,
                     output_format, normalization_type, trainable, supports=None, graph=False,
                     scale=None, im_sign=-1.0):
    """Top-down octave loop.  ``banks[i] = (real_i, imag_i)`` (i = 0: top octave); each octave
    halves the signal with the anti-alias FIR kernel and halves the hop, and its framed
    contraction writes straight into its row block of the final ``(B, n_bins, T[, 2])``
    tensor (the reference's growing ``torch.cat`` is gone)."""
    epi = output_epilogue(output_format)
    if scale is None:
        scale = normalisation_scale(lenghts, normalization_type, downsample_factor)
    if epi is None:
        return None
    n_oct = len(banks)
    n_filters = banks[0][0].shape[0]
    total_rows = n_oct * n_filters
    drop = total_rows - n_bins  # rows cut from the bottom octave by CQT[:, -n_bins:]
    out = None
    xd = x
    T_ref = None
    launches = []  # the per-octave contractions are independent: one grouped launch at the end
    blocks = []    # graph=True (training): per-octave outputs through autograd, concatenated
    for i, (kr, ki) in enumerate(banks):
        if i > 0:
            hop = hop // 2
            xd = engine.fir_decimate_autograd(xd, lowpass, 2) if graph else engine.fir_decimate(xd, lowpass, 2)
        K = kr.shape[-1]
        L = xd.shape[-1]
        pad = K // 2
        mode = engine.pad_mode_id(pad_mode)
        if mode == engine.PAD_REFLECT and pad >= L:
            warnings.warn(
                f"\ninput size = {tuple(xd.shape)}\tkernel size = {K}\n"
                "padding with reflection mode might not be the best choice, try using constant padding",
                UserWarning,
            )
            mode = engine.PAD_ZERO
        T = engine.n_frames(L, K, hop, pad)
        if T_ref is None:
            T_ref = T
            two = epi in (engine.EPI_COMPLEX, engine.EPI_PHASE_COSSIN)
            shape = (x.shape[0], n_bins, T, 2) if two else (x.shape[0], n_bins, T)
            out = engine.alloc_out(shape, x.device)
        elif T != T_ref:
            raise RuntimeError(
                "Sizes of tensors must match except in dimension 1. Expected size %d but got "
                "size %d for octave %d (hop_length must stay divisible while it is halved)"
                % (T_ref, T, i)
            )
        # rows of this octave inside the concatenated bank, after the bottom cut
        row0 = (n_oct - 1 - i) * n_filters - drop
        first = 0
        if row0 < 0:
            first = -row0
            row0 = 0
        if first >= n_filters:
            continue
        kr_i = kr.reshape(n_filters, -1)[first:]
        ki_i = ki.reshape(n_filters, -1)[first:]
        rows = n_filters - first
        sup = None
        if supports is not None and not trainable:
            sup = supports[i].get(kr, ki)[first:].contiguous()
        if graph:
            # the reference's own structure (cqt.py:1091-1105): one contraction per octave, rows
            # concatenated with the lowest octave first
            blocks.append((row0, engine.framed_gemm_autograd(
                xd, kr_i, ki_i, hop=hop, pad=pad, pad_mode=mode, epilogue=epi, im_sign=im_sign,
                eps=1e-8 if trainable else 0.0, row_scale=scale[row0:row0 + rows].contiguous(),
                precision="fp32")))
            continue
        launches.append((xd, kr_i, ki_i, dict(
            hop=hop, pad=pad, pad_mode=mode, epilogue=epi, im_sign=im_sign,
            eps=1e-8 if trainable else 0.0, row_scale=scale[row0:row0 + rows].contiguous(),
            row_support=sup, out=out, out_rows_total=n_bins, out_row_offset=row0,
            precision="fp32")))
    if graph:
        return torch.cat([b for _, b in sorted(blocks, key=lambda rb: rb[0])], 1)
    engine.framed_gemm_group(launches)
    return out


def early_decimate(x, taps, factor):
    return engine.fir_decimate_autograd(x, taps, int(factor))


__all__ = ["output_epilogue", "normalisation_scale", "SupportCache", "octave_recursion",
           "early_decimate", "decimated_length"]
