"""Constant-Q transform modules (drop-ins for ``nnAudio.features.CQT1992v2`` /
``CQT2010v2`` / ``CQT``; reference: Installation/nnAudio/features/cqt.py:561-802,
805-1139, 1142-1145).

CQT1992v2: one bank of long time-domain kernels, a single support-aware framed MFMA
contraction.  CQT2010v2: one short top-octave bank re-used over ``n_octaves`` octaves with
FIR decimation by 2 in between, each octave writing its row block of the result in place.
"""
from time import time

import numpy as np
import torch
import torch.nn as nn

from .. import engine
from ..basis import (cqt_bin_frequencies, cqt_kernel_bank, early_downsample_plan, filter_q,
                     lowpass_taps, top_octave_band)
from ..utils import broadcast_dim
from ._cqt_common import (OctaveCache, SupportCache, early_decimate, normalisation_scale, octave_recursion,
                          output_epilogue)


def _register_kernels(module, real, imag, trainable, suffix=""):
    if trainable:
        module.register_parameter("cqt_kernels_real" + suffix, nn.Parameter(real, requires_grad=True))
        module.register_parameter("cqt_kernels_imag" + suffix, nn.Parameter(imag, requires_grad=True))
    else:
        module.register_buffer("cqt_kernels_real" + suffix, real)
        module.register_buffer("cqt_kernels_imag" + suffix, imag)


class CQT1992v2(nn.Module):
    """CQT with time-domain kernels (Brown & Puckette 1992, applied as a strided
    correlation).  Constructor / ``forward(x, output_format=None,
    normalization_type='librosa')`` / buffers (``cqt_kernels_real``, ``cqt_kernels_imag``,
    ``lenghts``) / attributes (``kernel_width``, ``frequencies``) as the reference.
    Output ``(batch, n_bins, frames)`` (Magnitude) or ``(..., 2)`` (Complex; Phase = cos, sin)."""

    def __init__(
        self,
        sr=22050,
        hop_length=512,
        fmin=32.70,
        fmax=None,
        n_bins=84,
        bins_per_octave=12,
        filter_scale=1,
        norm=1,
        window="hann",
        center=True,
        pad_mode="reflect",
        trainable=False,
        output_format="Magnitude",
        verbose=True,
    ):
        super().__init__()
        self.trainable = trainable
        self.hop_length = hop_length
        self.center = center
        self.pad_mode = pad_mode
        self.output_format = output_format

        Q = filter_q(filter_scale, bins_per_octave)
        if verbose:
            print("Creating CQT kernels ...", end="\r")
        start = time()
        bank, self.kernel_width, lengths, freqs = cqt_kernel_bank(
            Q, sr, fmin, n_bins, bins_per_octave, norm, window, fmax
        )
        self.register_buffer("lenghts", torch.tensor(lengths).float())
        self.frequencies = freqs
        real = torch.tensor(bank.real).unsqueeze(1)
        imag = torch.tensor(bank.imag).unsqueeze(1)
        _register_kernels(self, real, imag, trainable)
        self._support = SupportCache()
        # None: the module's default, "fp32" -- one float32 FMA chain over the taps per output, the reference's conv1d
        # arithmetic (bit-identical to torch's conv1d on the MI355X); or "f16x3" / "bf16x3" (3.4 x / 3.8 x faster, 4.7e-7 /
        # 4e-6 of the peak from float64, NOT the fixture's silent-bin noise), unless nnaudio_amd.set_precision(...) overrides
        self.precision = None
        self._split = engine.DerivedCache()
        if verbose:
            print("CQT kernels created, time used = {:.4f} seconds".format(time() - start))

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
        epi = output_epilogue(output_format)
        precision = engine.resolve_precision(self.precision)
        if engine.compiling():
            # torch.compile: the custom op looks the supports / split planes up at run time
            if epi is None:
                return None
            return engine.framed_gemm_autograd(
                x, self.cqt_kernels_real, self.cqt_kernels_imag, hop=self.hop_length, pad=pad,
                pad_mode=mode, epilogue=epi, im_sign=-1.0, eps=1e-8 if self.trainable else 0.0,
                row_scale=normalisation_scale(self.lenghts, normalization_type),
                support=not self.trainable, precision=precision)
        if not hasattr(self, "_scale"):
            self._scale = engine.DerivedCache()
        scale = self._scale.get((self.lenghts,),
                                lambda: normalisation_scale(self.lenghts, normalization_type),
                                extra=normalization_type)
        if epi is None:
            return None
        sup = None if self.trainable else self._support.get(self.cqt_kernels_real,
                                                            self.cqt_kernels_imag)
        split = None
        kr, ki = self.cqt_kernels_real, self.cqt_kernels_imag
        if not x.is_cuda:
            pass  # (host path: the buffers as they are)
        elif precision == "bf16x3":
            split = self._split.get((kr, ki), lambda: engine.split_basis(kr, ki), extra=precision)
        elif precision == "f16x3":
            # banks with supports: the strip kernel (fragment-order copy), unless ``hop_periodic = False``
            # asks for the staged dense kernel on row-major planes -- taps in their natural order: 0.6 % of
            # the reference's log-magnitude fixture elements miss its tolerance instead of 2.7 % (fp32 tile
            # kernels: none; the comb-ordered partial sums of the hop-periodic kernels are orders of
            # magnitude larger than a silent bin), at 2.5 x the time.  Trainable banks: the dense kernel.
            if getattr(self, "hop_periodic", True) and sup is not None:
                split = self._split.get((kr, ki), lambda: engine.frag_basis_f16(kr, ki), extra=precision + "-strip")
            else:
                split = self._split.get((kr, ki), lambda: engine.split_basis_f16(kr, ki), extra=precision)
        # fp32 accumulates ONE float32 FMA chain over the taps in ascending order -- the reference's conv1d
        # arithmetic: bit-identical to torch's conv1d on the MI355X, the reference's six fixture assertions
        # pass verbatim (tests/test_reference_order.py).  With the bank's chain copy (hops of 64 .. 512) the
        # chain kernel computes those bits from LDS delay lines; without it the tile kernels do.  (The
        # hop-periodic strip kernel also exists in fp32 -- engine.frag_basis_f32 -- but its order leaves
        # different rounding noise in the near-silent bins: 4.7 % of them miss the reference's tolerance.)
        chain = None
        if x.is_cuda and precision == "fp32" and sup is not None and getattr(self, "chain", True):
            host = self._support.host(kr, ki)
            chain = self._split.get((kr, ki), lambda: engine.chain_basis_f32(kr, ki, host), extra="fp32-chain")
        return engine.framed_gemm_autograd(
            x, self.cqt_kernels_real, self.cqt_kernels_imag, hop=self.hop_length, pad=pad,
            pad_mode=mode, epilogue=epi, im_sign=-1.0, eps=1e-8 if self.trainable else 0.0,
            row_scale=scale, row_support=sup, precision=precision, basis_split=split, basis_chain=chain,
        )


class CQT2010v2(nn.Module):
    """Multi-resolution CQT (Schoerkhuber & Klapuri 2010) with time-domain kernels.
    Constructor / forward signature, buffers (``lowpass_filter``, ``early_downsample_filter``,
    ``cqt_kernels_real``, ``cqt_kernels_imag``, ``lenghts``) and attributes (``hop_length``
    after early down-sampling, ``n_fft``, ``n_octaves``, ``downsample_factor``,
    ``earlydownsample``, ``fmin_t``, ``frequencies``, ``basis``) as the reference."""

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

        n_filters = min(bins_per_octave, n_bins)
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

        if verbose:
            print("Creating CQT kernels ...", end="\r")
        start = time()
        basis, self.n_fft, _, _ = cqt_kernel_bank(
            Q, sr, self.fmin_t, n_filters, bins_per_octave, norm=basis_norm, topbin_check=False
        )
        # per-bin lengths of ALL bins at the (possibly early-down-sampled) rate
        freqs = cqt_bin_frequencies(fmin, n_bins, bins_per_octave)
        self.frequencies = freqs
        self.register_buffer("lenghts", torch.tensor(np.ceil(Q * sr / freqs)).float())

        self.basis = basis
        real = torch.tensor(basis.real).unsqueeze(1)
        imag = torch.tensor(basis.imag).unsqueeze(1)
        _register_kernels(self, real, imag, trainable)
        self._support = SupportCache()
        # not a constructor argument: None = the module's default, "f16x3" on the streaming octave kernel
        # (octave_stream.hip: the whole recursion as a stream, rings of every octave's signal in LDS, scaled fp16 pairs on
        # the matrix pipe; the reference's four CQT2010v2 fixture assertions pass verbatim); "bf16x3" the same kernel on
        # split bf16, "fp32" one launch per octave stage on the fp32 tile kernels (attribute or nnaudio_amd.set_precision)
        self.precision = None
        self._octaves = OctaveCache()
        if verbose:
            print("CQT kernels created, time used = {:.4f} seconds".format(time() - start))

    def forward(self, x, output_format=None, normalization_type="librosa"):
        output_format = output_format or self.output_format
        x = broadcast_dim(x)
        graph = engine.needs_grad(self, x)
        if self.pad_mode not in ("constant", "reflect"):
            raise AttributeError("'CQT2010v2' object has no attribute 'padding'")
        if self.earlydownsample:
            x = early_decimate(x, self.early_downsample_filter, self.downsample_factor)
        banks = [(self.cqt_kernels_real, self.cqt_kernels_imag)] * self.n_octaves
        return octave_recursion(
            x, banks, self.lenghts, self.hop_length, self.n_bins, self.lowpass_filter,
            self.downsample_factor, self.pad_mode, output_format, normalization_type,
            self.trainable, supports=[self._support] * self.n_octaves, graph=graph,
            precision=engine.resolve_precision(self.precision, "f16x3"), cache=self._octaves,
        )


class CQT(CQT1992v2):
    """Alias of :class:`CQT1992v2` (reference: cqt.py:1142-1145)."""

    pass
