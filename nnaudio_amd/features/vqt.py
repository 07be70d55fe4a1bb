"""Variable-Q transform module (drop-in for ``nnAudio.features.VQT``,
reference: Installation/nnAudio/features/vqt.py:9-215): the CQT2010v2 octave recursion with
one kernel bank per octave whose bandwidths are widened by ``gamma``."""
from time import time

import numpy as np
import torch
import torch.nn as nn

from .. import engine
from ..basis import (cqt_bin_frequencies, cqt_kernel_bank, early_downsample_plan, filter_q,
                     lowpass_taps, top_octave_band)
from ..utils import broadcast_dim
from ._cqt_common import OctaveCache, SupportCache, early_decimate, octave_recursion


class VQT(nn.Module):
    """Constructor / ``forward(x, output_format=None, normalization_type='librosa')`` /
    buffers (``lowpass_filter``, ``early_downsample_filter``, ``lenghts``,
    ``cqt_kernels_real_{i}``, ``cqt_kernels_imag_{i}``) / attributes as the reference.

    Reference quirks reproduced on purpose: the per-octave banks are built from the
    constructor's *original* ``sr`` even when early down-sampling has halved the signal
    rate (vqt.py:120-134), ``lenghts`` uses the down-sampled rate (vqt.py:112), and
    ``n_fft`` ends up as the width of the *last* octave's bank."""

    def __init__(
        self,
        sr=22050,
        hop_length=512,
        fmin=32.70,
        fmax=None,
        n_bins=84,
        filter_scale=1,
        bins_per_octave=12,
        norm=True,
        basis_norm=1,
        gamma=0,
        window="hann",
        pad_mode="reflect",
        earlydownsample=True,
        trainable=False,
        output_format="Magnitude",
        verbose=True,
    ):
        super().__init__()
        self.norm = norm
        self.hop_length = hop_length
        self.pad_mode = pad_mode
        self.n_bins = n_bins
        self.earlydownsample = earlydownsample
        self.trainable = trainable
        self.output_format = output_format
        self.filter_scale = filter_scale
        self.bins_per_octave = bins_per_octave
        self.sr = sr
        self.gamma = gamma
        self.basis_norm = basis_norm

        Q = filter_q(filter_scale, bins_per_octave)

        if verbose:
            print("Creating low pass filter ...", end="\r")
        start = time()
        lowpass = torch.from_numpy(
            lowpass_taps(band_center=0.50, kernelLength=256, transitionBandwidth=0.001)
        )
        self.register_buffer("lowpass_filter", lowpass[None, None, :])
        if verbose:
            print("Low pass filter created, time used = {:.4f} seconds".format(time() - start))

        self.n_filters = min(bins_per_octave, n_bins)
        self.n_octaves, self.fmin_t, fmax_t = top_octave_band(fmin, n_bins, bins_per_octave)
        if verbose:
            print("num_octave = ", self.n_octaves)
        if fmax_t > sr / 2:
            raise ValueError(
                "The top bin {}Hz has exceeded the Nyquist frequency, \
                            please reduce the n_bins".format(fmax_t)
            )

        if self.earlydownsample:
            if verbose:
                print("Creating early downsampling filter ...", end="\r")
            start = time()
            sr, self.hop_length, self.downsample_factor, taps, self.earlydownsample = (
                early_downsample_plan(sr, hop_length, fmax_t, Q, self.n_octaves, verbose)
            )
            early = None if taps is None else torch.from_numpy(taps)[None, None, :]
            self.register_buffer("early_downsample_filter", early)
            if verbose:
                print("Early downsampling filter created, \
                        time used = {:.4f} seconds".format(time() - start))
        else:
            self.downsample_factor = 1.0

        alpha = 2.0 ** (1.0 / bins_per_octave) - 1.0
        freqs = cqt_bin_frequencies(fmin, n_bins, bins_per_octave)
        self.frequencies = freqs
        lenghts = np.ceil(Q * sr / (freqs + gamma / alpha))
        self.n_fft = int(2 ** (np.ceil(np.log2(int(max(lenghts))))))
        self.register_buffer("lenghts", torch.tensor(lenghts).float())

        octave_sr = self.sr
        self._supports = []
        self.precision = None  # None: "f16x3" on the streaming octave kernel (see CQT2010v2); "bf16x3" / "fp32"
        self._octaves = OctaveCache()
        for i in range(self.n_octaves):
            if i > 0:
                octave_sr /= 2
            basis, self.n_fft, _, _ = cqt_kernel_bank(
                filter_q(self.filter_scale, self.bins_per_octave),
                octave_sr,
                self.fmin_t * 2 ** -i,
                self.n_filters,
                self.bins_per_octave,
                norm=self.basis_norm,
                topbin_check=False,
                gamma=self.gamma,
            )
            real = torch.tensor(basis.real.astype(np.float32)).unsqueeze(1)
            imag = torch.tensor(basis.imag.astype(np.float32)).unsqueeze(1)
            self.register_buffer("cqt_kernels_real_{}".format(i), real)
            self.register_buffer("cqt_kernels_imag_{}".format(i), imag)
            self._supports.append(SupportCache())

    def forward(self, x, output_format=None, normalization_type="librosa"):
        output_format = output_format or self.output_format
        x = broadcast_dim(x)
        graph = engine.needs_grad(self, x)
        if self.pad_mode not in ("constant", "reflect"):
            raise UnboundLocalError("local variable 'my_padding' referenced before assignment")
        if self.earlydownsample:
            x = early_decimate(x, self.early_downsample_filter, self.downsample_factor)
        banks = [
            (getattr(self, "cqt_kernels_real_{}".format(i)),
             getattr(self, "cqt_kernels_imag_{}".format(i)))
            for i in range(self.n_octaves)
        ]
        return octave_recursion(
            x, banks, self.lenghts, self.hop_length, self.n_bins, self.lowpass_filter,
            self.downsample_factor, self.pad_mode, output_format, normalization_type,
            self.trainable, supports=self._supports, graph=graph,
            precision=engine.resolve_precision(self.precision, "f16x3"), cache=self._octaves,
        )
