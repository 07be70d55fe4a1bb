"""STFT front-end module (drop-in for ``nnAudio.features.STFT``,
reference: Installation/nnAudio/features/stft.py:68-361).

Construction builds the windowed Fourier bases on the host (``basis.fourier_basis``) and
registers them under the reference's buffer names; ``forward`` is a single launch of the
MFMA framed contraction with the Magnitude / Complex / Phase epilogue fused.
"""
from time import time

import torch
import torch.nn as nn

from .. import engine
from ..basis import fourier_basis
from ..utils import broadcast_dim


def _inverse_stft(mod, X, kernel_cos, kernel_sin, onesided, length):
    """STFTBase.inverse_stft (stft.py:15-63) for ``STFT.inverse`` and ``iSTFT.forward``.
    (``refresh_win`` needs no state here: the window sum is evaluated inside the overlap-add
    kernel for whatever number of frames the input has.)"""
    F = X.shape[1]
    graph = torch.is_grad_enabled()
    dft = False
    if graph and (kernel_cos.requires_grad or kernel_sin.requires_grad):
        basis = engine.istft_basis(kernel_cos, kernel_sin, F, onesided)  # differentiable torch indexing
    else:
        if not hasattr(mod, "_inv_basis"):
            mod._inv_basis = engine.DerivedCache()

        def build():
            b = engine.istft_basis(kernel_cos, kernel_sin, F, onesided)
            # plain inverse-DFT kernels + one-sided spectrum: the frames come from an inverse FFT
            return b, bool(onesided) and engine.istft_basis_is_dft(b, F)

        basis, dft = mod._inv_basis.get((kernel_cos, kernel_sin), build, extra=(F, bool(onesided)))
    wdtype = mod.window_mask.dtype
    window = mod.window_mask.reshape(-1).to(torch.float32)
    if window.numel() != mod.n_fft:
        raise RuntimeError(
            "The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton "
            "dimension 1" % (mod.n_fft, window.numel()))
    T = X.shape[2]
    full = mod.n_fft + mod.stride * (T - 1)
    pad = mod.pad_amount if mod.center else 0
    if length is None:
        start, out_len = pad, full - 2 * pad
    else:
        start, out_len = pad, min(int(length), full - pad)
    y = engine.istft_autograd(X, basis, window, mod.stride, start, out_len, dft=dft)
    # the reference multiplies the float32 frames by its window buffer: the iSTFT class keeps a
    # float64 window (stft.py:489-493) and therefore returns float64 waveforms
    return y if wdtype == torch.float32 else y.to(wdtype)


class STFT(nn.Module):
    """Short-time Fourier transform of ``(len,)``, ``(batch, len)`` or ``(batch, 1, len)``
    float32 waveforms.

    Same constructor arguments, defaults, attributes (``stride``, ``n_fft``, ``pad_amount``,
    ``bins2freq``, ``bin_list`` ...) and ``state_dict`` keys (``wsin``, ``wcos``,
    ``window_mask``, optionally ``kernel_sin_inv`` / ``kernel_cos_inv``) as the reference.
    Output: ``(batch, freq_bins, frames)`` for Magnitude / Phase (radians),
    ``(batch, freq_bins, frames, 2)`` for Complex."""

    def __init__(
        self,
        n_fft=2048,
        win_length=None,
        freq_bins=None,
        hop_length=None,
        window="hann",
        freq_scale="no",
        center=True,
        pad_mode="reflect",
        iSTFT=False,
        fmin=50,
        fmax=6000,
        sr=22050,
        trainable=False,
        output_format="Complex",
        verbose=True,
    ):
        super().__init__()
        if win_length is None:
            win_length = n_fft
        if hop_length is None:
            hop_length = int(win_length // 4)

        self.output_format = output_format
        self.trainable = trainable
        self.stride = hop_length
        self.center = center
        self.pad_mode = pad_mode
        self.n_fft = n_fft
        self.freq_bins = freq_bins
        self.pad_amount = self.n_fft // 2
        self.window = window
        self.win_length = win_length
        self.iSTFT = iSTFT
        start = time()

        ksin, kcos, self.bins2freq, self.bin_list, win = fourier_basis(
            n_fft,
            win_length=win_length,
            freq_bins=freq_bins,
            window=window,
            freq_scale=freq_scale,
            fmin=fmin,
            fmax=fmax,
            sr=sr,
            verbose=verbose,
        )
        ksin = torch.from_numpy(ksin)
        kcos = torch.from_numpy(kcos)

        if iSTFT:
            # full-spectrum inverse bases, only kept for state_dict compatibility this round
            self.register_buffer(
                "kernel_sin_inv", torch.cat((ksin, -ksin[1:-1].flip(0)), 0).unsqueeze(-1)
            )
            self.register_buffer(
                "kernel_cos_inv", torch.cat((kcos, kcos[1:-1].flip(0)), 0).unsqueeze(-1)
            )

        # the window is folded into the basis in float32 (stft.py:230-232)
        win = torch.from_numpy(win)
        wsin = ksin * win
        wcos = kcos * win
        if trainable:
            self.register_parameter("wsin", nn.Parameter(wsin, requires_grad=True))
            self.register_parameter("wcos", nn.Parameter(wcos, requires_grad=True))
        else:
            self.register_buffer("wsin", wsin)
            self.register_buffer("wcos", wcos)
        self.register_buffer("window_mask", win.unsqueeze(0).unsqueeze(-1))
        # not a constructor argument (the signature stays the reference's): None = the module's
        # default, "f16x3" (scaled fp16 pairs on the f16 matrix pipe, fp32-class accuracy), unless
        # nnaudio_amd.set_precision(...) overrides it; or "fp32" / "bf16x3" / "f16x3"
        self.precision = None
        self._split = engine.DerivedCache()

        if verbose:
            print("STFT kernels created, time used = {:.4f} seconds".format(time() - start))

    # ------------------------------------------------------------------ #
    def _framing(self, num_samples):
        """(pad, PAD_* id) for this module, with the reference's error behaviour."""
        if not self.center:
            return 0, engine.PAD_NONE
        if self.pad_mode == "constant":
            return self.pad_amount, engine.PAD_ZERO
        if self.pad_mode == "reflect":
            if num_samples < self.pad_amount:
                raise AssertionError(
                    "Signal length shorter than reflect padding length (n_fft // 2)."
                )
            return self.pad_amount, engine.PAD_REFLECT
        # the reference leaves `padding` unbound for any other mode (stft.py:279-289)
        raise UnboundLocalError("local variable 'padding' referenced before assignment")

    def _spectrum(self, x, epilogue, power=2.0, fb=None, fb_support=None, out_frame_major=0):
        """The framed contraction of this layer; with ``fb`` the filterbank reduction of
        MelSpectrogram / Gammatonegram is fused into it (no autograd graph, see
        ``engine.fused_filterbank_ok``); with ``out_frame_major = Fp`` the spectrogram comes back
        ``(B, T, Fp)`` (FFT route only, no graph; None when this layer's basis does not open it)."""
        pad, mode = self._framing(x.shape[-1])
        wsin, wcos = self.wsin, self.wcos
        if self.freq_bins is not None:
            wsin, wcos = wsin[: self.freq_bins], wcos[: self.freq_bins]
        precision = engine.resolve_precision(self.precision, "f16x3")
        prep = {}
        if not engine.compiling():
            # bf16x3: split planes; either arithmetic, the window being symmetric: the folded planes
            # (half the MFMAs) -- cached per basis and precision
            frozen = not (self.trainable or self.wcos.requires_grad or self.wsin.requires_grad)
            prep = self._split.get((self.wcos, self.wsin),
                                   lambda: engine.prepare_basis(wcos, wsin, precision, hop=self.stride, fold=frozen),
                                   extra=(self.freq_bins, self.stride, precision, frozen))
        if out_frame_major:
            if "basis_fold2" not in prep:
                return None
            return engine.framed_gemm(
                x, wcos, wsin, hop=self.stride, pad=pad, pad_mode=mode, epilogue=epilogue,
                im_sign=-1.0, eps=0.0, power=power, precision=precision, out_frame_major=out_frame_major, **prep)
        if fb is not None:
            return engine.framed_gemm(
                x, wcos, wsin, hop=self.stride, pad=pad, pad_mode=mode, epilogue=epilogue,
                im_sign=-1.0, eps=1e-8 if self.trainable else 0.0, power=power,
                precision=precision, fb=fb, fb_support=fb_support, **prep)
        return engine.framed_gemm_autograd(
            x, wcos, wsin, hop=self.stride, pad=pad, pad_mode=mode, epilogue=epilogue,
            im_sign=-1.0, eps=1e-8 if self.trainable else 0.0, power=power,
            precision=precision, **prep,
        )

    def forward(self, x, output_format=None):
        """Waveform batch -> spectrogram; ``output_format`` overrides the constructor's."""
        output_format = output_format or self.output_format
        self.num_samples = x.shape[-1]
        x = broadcast_dim(x)
        if output_format == "Magnitude":
            return self._spectrum(x, engine.EPI_MAGNITUDE)
        if output_format == "Complex":
            return self._spectrum(x, engine.EPI_COMPLEX)
        if output_format == "Phase":
            return self._spectrum(x, engine.EPI_PHASE_ATAN2)
        return None  # the reference falls through its if/elif chain (stft.py:299-316)

    def inverse(self, X, onesided=True, length=None, refresh_win=True):
        if not (hasattr(self, "kernel_sin_inv") and hasattr(self, "kernel_cos_inv")):
            raise NameError(
                "Please activate the iSTFT module by setting `iSTFT=True` if you want to use `inverse`"
            )
        assert X.dim() == 4, (
            "Inverse iSTFT only works for complex number,"
            "make sure our tensor is in the shape of (batch, freq_bins, timesteps, 2)."
            "\nIf you have a magnitude spectrogram, please consider using Griffin-Lim."
        )
        return _inverse_stft(self, X, self.kernel_cos_inv, self.kernel_sin_inv, onesided, length)

    def extra_repr(self) -> str:
        return "n_fft={}, Fourier Kernel size={}, iSTFT={}, trainable={}".format(
            self.n_fft, (*self.wsin.shape,), self.iSTFT, self.trainable
        )


class iSTFT(nn.Module):
    """Spectrogram ``(batch, freq_bins, frames, 2)`` -> waveform; same constructor, buffers
    (``kernel_sin``, ``kernel_cos`` of shape ``(n_fft, 1, n_fft, 1)``, ``window_mask``) and
    ``forward(X, onesided=False, length=None, refresh_win=None)`` as the reference
    (stft.py:364-546)."""

    def __init__(self, n_fft=2048, win_length=None, freq_bins=None, hop_length=None, window="hann",
                 freq_scale="no", center=True, fmin=50, fmax=6000, sr=22050,
                 trainable_kernels=False, trainable_window=False, verbose=True, refresh_win=True):
        super().__init__()
        if win_length is None:
            win_length = n_fft
        if hop_length is None:
            hop_length = int(win_length // 4)
        self.n_fft = n_fft
        self.win_length = win_length
        self.stride = hop_length
        self.center = center
        self.pad_amount = self.n_fft // 2
        self.refresh_win = refresh_win
        self.trainable = trainable_kernels or trainable_window
        start = time()
        ksin, kcos, _, _, _ = fourier_basis(
            n_fft, win_length=win_length, freq_bins=n_fft, window=window, freq_scale=freq_scale,
            fmin=fmin, fmax=fmax, sr=sr, verbose=False)
        # the inverse kernels are not windowed; the window is applied to the synthesised frames
        from scipy.signal import get_window

        window_mask = torch.tensor(get_window(window, int(win_length), fftbins=True)).unsqueeze(0).unsqueeze(-1)
        ksin = torch.tensor(ksin, dtype=torch.float).unsqueeze(-1)
        kcos = torch.tensor(kcos, dtype=torch.float).unsqueeze(-1)
        if trainable_kernels:
            self.register_parameter("kernel_sin", nn.Parameter(ksin, requires_grad=True))
            self.register_parameter("kernel_cos", nn.Parameter(kcos, requires_grad=True))
        else:
            self.register_buffer("kernel_sin", ksin)
            self.register_buffer("kernel_cos", kcos)
        if trainable_window:
            self.register_parameter("window_mask", nn.Parameter(window_mask, requires_grad=True))
        else:
            self.register_buffer("window_mask", window_mask)
        if verbose:
            print("iSTFT kernels created, time used = {:.4f} seconds".format(time() - start))

    def forward(self, X, onesided=False, length=None, refresh_win=None):
        assert X.dim() == 4, (
            "Inverse iSTFT only works for complex number,"
            "make sure our tensor is in the shape of (batch, freq_bins, timesteps, 2)"
        )
        return _inverse_stft(self, X, self.kernel_cos, self.kernel_sin, onesided, length)
