"""Frequency-domain constant-Q variants ``CQT1992`` and ``CQT2010`` (reference: cqt.py:9-263,
266-560): an un-windowed STFT of size ``kernel_width`` followed by a complex matmul with the FFT
of the CQT kernels (utils.py:175-203, 524-559).

Same constructors, buffers (``wsin``, ``wcos`` of shape ``(F, 1, K)``; ``cqt_kernels_real``,
``cqt_kernels_imag`` of shape ``(n_bins, F)`` -- the *frequency-domain* kernels; ``lenghts``;
``lowpass_filter`` / ``early_downsample_filter`` for CQT2010) and outputs as the reference.  The
two stages are linear in the waveform, so they are one framed contraction with the effective
time-domain kernels

    E_re = K_re @ wcos - K_im @ wsin,        E_im = K_re @ wsin + K_im @ wcos

built from the buffers by the planar contraction kernel (cached while the buffers are unchanged;
rebuilt through autograd when they are trainable) -- the STFT is never materialised."""
from time import time

import numpy as np
import torch
import torch.nn as nn
from scipy.fftpack import fft

from .. import engine
from ..basis import (cqt_bin_frequencies, cqt_kernel_bank, early_downsample_plan, filter_q,
                     fourier_basis, lowpass_taps, top_octave_band)
from ..utils import broadcast_dim
from ._cqt_common import early_decimate, octave_recursion, output_epilogue


def _register(mod, name, tensor, trainable):
    if trainable:
        mod.register_parameter(name, nn.Parameter(tensor, requires_grad=True))
    else:
        mod.register_buffer(name, tensor)


def _effective_kernels(mod):
    """(E_re, E_im), each (n_bins, K): see the module docstring."""
    kr, ki, wc, ws = mod.cqt_kernels_real, mod.cqt_kernels_imag, mod.wcos, mod.wsin

    def build():
        fb = torch.cat((torch.cat((kr, -ki), 1), torch.cat((ki, kr), 1)), 0)  # (2 nb, 2F)
        spec = torch.cat((wc.reshape(wc.shape[0], -1), ws.reshape(ws.shape[0], -1)), 0)[None]
        e = engine.filterbank_autograd(fb.contiguous(), spec.contiguous())[0]  # (2 nb, K)
        nb = kr.shape[0]
        return e[:nb], e[nb:]

    if torch.is_grad_enabled() and any(t.requires_grad for t in (kr, ki, wc, ws)):
        return build()
    if engine.compiling():  # traced tensors: no cache look-up (data pointers), the product is one op
        return build()
    if not hasattr(mod, "_eff"):
        mod._eff = engine.DerivedCache()
    return mod._eff.get((kr, ki, wc, ws), build)


def _scale(lenghts, width, normalization_type):
    if normalization_type == "librosa":
        return (torch.sqrt(lenghts) / width).to(torch.float32).contiguous()
    if normalization_type == "convolutional":
        return torch.ones_like(lenghts)
    if normalization_type == "wrap":
        return torch.full_like(lenghts, 2.0 / width)
    raise ValueError(
        "The normalization_type %r is not part of our current options." % normalization_type
    )


class CQT1992(nn.Module):
    """Brown & Puckette (1992) CQT through the frequency domain (cqt.py:9-263)."""

    def __init__(self, sr=22050, hop_length=512, fmin=220, fmax=None, n_bins=84,
                 trainable_STFT=False, trainable_CQT=False, bins_per_octave=12, filter_scale=1,
                 output_format="Magnitude", norm=1, window="hann", center=True, pad_mode="reflect"):
        super().__init__()
        self.hop_length = hop_length
        self.center = center
        self.pad_mode = pad_mode
        self.norm = norm
        self.output_format = output_format
        Q = filter_q(filter_scale, bins_per_octave)
        print("Creating CQT kernels ...", end="\r")
        start = time()
        bank, self.kernel_width, lengths, freqs = cqt_kernel_bank(
            Q, sr, fmin, n_bins, bins_per_octave, norm, window, fmax)
        self.register_buffer("lenghts", torch.tensor(lengths).float())
        self.frequencies = freqs
        bank = fft(bank)[:, : self.kernel_width // 2 + 1]
        print("CQT kernels created, time used = {:.4f} seconds".format(time() - start))
        print("Creating STFT kernels ...", end="\r")
        start = time()
        ksin, kcos, self.bins2freq, _, win = fourier_basis(
            self.kernel_width, window="ones", freq_scale="no", verbose=False)
        _register(self, "wsin", torch.tensor(ksin * win), trainable_STFT)
        _register(self, "wcos", torch.tensor(kcos * win), trainable_STFT)
        _register(self, "cqt_kernels_real", torch.tensor(bank.real), trainable_CQT)
        _register(self, "cqt_kernels_imag", torch.tensor(bank.imag), trainable_CQT)
        print("STFT kernels created, time used = {:.4f} seconds".format(time() - start))

    def forward(self, x, output_format=None, normalization_type="librosa"):
        output_format = output_format or self.output_format
        x = broadcast_dim(x)
        if self.center:
            if self.pad_mode == "constant":
                mode = engine.PAD_ZERO
            elif self.pad_mode == "reflect":
                mode = engine.PAD_REFLECT
            else:
                raise UnboundLocalError("local variable 'padding' referenced before assignment")
            pad = self.kernel_width // 2
        else:
            pad, mode = 0, engine.PAD_NONE
        scale = _scale(self.lenghts, self.kernel_width, normalization_type)
        epi = output_epilogue(output_format)
        if epi is None:
            return None
        e_re, e_im = _effective_kernels(self)
        # Complex / Magnitude use (real, -imag); Phase uses the un-negated pair (cqt.py:224, 252-254)
        sign = 1.0 if epi == engine.EPI_PHASE_COSSIN else -1.0
        return engine.framed_gemm_autograd(
            x, e_re, e_im, hop=self.hop_length, pad=pad, pad_mode=mode, epilogue=epi, im_sign=sign,
            row_scale=scale, precision="fp32")

    def extra_repr(self) -> str:
        return "STFT kernel size = {}, CQT kernel size = {}".format(
            (*self.wcos.shape,), (*self.cqt_kernels_real.shape,))


class CQT2010(nn.Module):
    """Schoerkhuber & Klapuri (2010) multi-resolution CQT through the frequency domain
    (cqt.py:266-560): the octave recursion of ``CQT2010v2`` with ``(real, +imag)`` output,
    ``sqrt(lenghts) / n_fft`` normalisation and no down-sampling gain."""

    def __init__(self, sr=22050, hop_length=512, fmin=32.70, fmax=None, n_bins=84,
                 bins_per_octave=12, norm=True, basis_norm=1, window="hann", pad_mode="reflect",
                 trainable_STFT=False, filter_scale=1, trainable_CQT=False,
                 output_format="Magnitude", earlydownsample=True, verbose=True):
        super().__init__()
        self.norm = norm
        self.hop_length = hop_length
        self.pad_mode = pad_mode
        self.n_bins = n_bins
        self.output_format = output_format
        self.earlydownsample = earlydownsample
        self.trainable = trainable_STFT or trainable_CQT
        Q = filter_q(filter_scale, bins_per_octave)
        if verbose:
            print("Creating low pass filter ...", end="\r")
        start = time()
        lowpass = torch.tensor(lowpass_taps(band_center=0.5, kernelLength=256, transitionBandwidth=0.001))
        self.register_buffer("lowpass_filter", lowpass[None, None, :])
        if verbose:
            print("Low pass filter created, time used = {:.4f} seconds".format(time() - start))
        n_filters = min(bins_per_octave, n_bins)
        self.n_octaves, self.fmin_t, fmax_t = top_octave_band(fmin, n_bins, bins_per_octave)
        if fmax_t > sr / 2:
            raise ValueError(
                "The top bin {}Hz has exceeded the Nyquist frequency, \
                              please reduce the n_bins".format(fmax_t))
        if self.earlydownsample:
            if verbose:
                print("Creating early downsampling filter ...", end="\r")
            start = time()
            sr, self.hop_length, self.downsample_factor, taps, self.earlydownsample = (
                early_downsample_plan(sr, hop_length, fmax_t, Q, self.n_octaves, verbose))
            early = None if taps is None else torch.from_numpy(taps)[None, None, :]
            self.register_buffer("early_downsample_filter", early)
            if verbose:
                print("Early downsampling filter created, \
                            time used = {:.4f} seconds".format(time() - start))
        else:
            self.downsample_factor = 1.0
        if verbose:
            print("Creating CQT kernels ...", end="\r")
        start = time()
        basis, self.n_fft, _, _ = cqt_kernel_bank(
            Q, sr, self.fmin_t, n_filters, bins_per_octave, norm=basis_norm, topbin_check=False)
        freqs = cqt_bin_frequencies(fmin, n_bins, bins_per_octave)
        self.frequencies = freqs
        self.register_buffer("lenghts", torch.tensor(np.ceil(Q * sr / freqs)).float())
        self.basis = basis
        fft_basis = fft(basis)[:, : self.n_fft // 2 + 1]
        if verbose:
            print("CQT kernels created, time used = {:.4f} seconds".format(time() - start))
            print("Creating STFT kernels ...", end="\r")
        start = time()
        ksin, kcos, self.bins2freq, _, win = fourier_basis(
            self.n_fft, window="ones", freq_scale="no", verbose=False)
        if verbose:
            print("STFT kernels created, time used = {:.4f} seconds".format(time() - start))
        _register(self, "wsin", torch.tensor(ksin * win), trainable_STFT)
        _register(self, "wcos", torch.tensor(kcos * win), trainable_STFT)
        _register(self, "cqt_kernels_real", torch.tensor(fft_basis.real), trainable_CQT)
        _register(self, "cqt_kernels_imag", torch.tensor(fft_basis.imag), trainable_CQT)

    def forward(self, x, output_format=None, normalization_type="librosa"):
        output_format = output_format or self.output_format
        x = broadcast_dim(x)
        graph = engine.needs_grad(self, x)
        if self.pad_mode not in ("constant", "reflect"):
            raise AttributeError("'CQT2010' object has no attribute 'padding'")
        if self.earlydownsample:
            x = early_decimate(x, self.early_downsample_filter, self.downsample_factor)
        scale = _scale(self.lenghts, self.n_fft, normalization_type)
        e_re, e_im = _effective_kernels(self)
        banks = [(e_re, e_im)] * self.n_octaves
        return octave_recursion(
            x, banks, self.lenghts, self.hop_length, self.n_bins, self.lowpass_filter, 1.0,
            self.pad_mode, output_format, None, False, graph=graph, scale=scale, im_sign=1.0)

    def extra_repr(self) -> str:
        return "STFT kernel size = {}, CQT kernel size = {}".format(
            (*self.wcos.shape,), (*self.cqt_kernels_real.shape,))
