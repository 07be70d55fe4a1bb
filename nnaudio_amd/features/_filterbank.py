"""What MelSpectrogram and Gammatonegram share (reference: features/mel.py:93-189 and its copy
features/gammatone.py:93-189): an STFT power spectrum followed by a filterbank contraction,
``matmul(basis, STFT_mag(x) ** power)``.  The subclasses keep the reference's constructor
signatures and buffer names; this base holds the construction of the inner STFT, the registration
of the filterbank and the forward pass."""
from time import time

import torch.nn as nn

from .. import engine
from ..utils import broadcast_dim
from .stft import STFT


class FilterbankSpectrogram(nn.Module):
    _basis_name = None   # name of the filterbank buffer / parameter ("mel_basis", "gammatone_basis")
    _label = None        # as printed by the reference ("Mel", "Gammatone")

    def _build(self, make_basis, *, sr, n_fft, win_length, hop_length, window, center, pad_mode, power,
               trainable_basis, trainable_STFT, verbose, stft_kwargs):
        self.stride = hop_length
        self.center = center
        self.pad_mode = pad_mode
        self.n_fft = n_fft
        self.power = power
        self.trainable_STFT = trainable_STFT
        self.stft = STFT(n_fft=n_fft, win_length=win_length, freq_bins=None, hop_length=hop_length,
                         window=window, freq_scale="no", center=center, pad_mode=pad_mode, sr=sr,
                         trainable=trainable_STFT, output_format="Magnitude", verbose=verbose,
                         **stft_kwargs)
        start = time()
        basis = make_basis()
        if verbose:
            print("STFT filter created, time used = {:.4f} seconds".format(time() - start))
            print("{} filter created, time used = {:.4f} seconds".format(self._label, time() - start))
        if trainable_basis:
            self.register_parameter(self._basis_name, nn.Parameter(basis, requires_grad=True))
        else:
            self.register_buffer(self._basis_name, basis)

    def forward(self, x):
        """``framed`` MFMA contraction with the ``|.|**power`` epilogue fused (the reference's
        sqrt -> pow round trip disappears for power = 2); the filterbank reduction runs in the same
        epilogue when it is banded (mel), as its own contraction kernel otherwise."""
        x = broadcast_dim(x)
        self.stft.num_samples = x.shape[-1]
        basis = getattr(self, self._basis_name)
        if engine.compiling() and not engine.needs_grad(self, x) and self.stft.freq_bins is None:
            # torch.compile: one op that decides at run time, on the real filterbank, as below
            from .. import ops

            pad, mode = self.stft._framing(x.shape[-1])
            return ops.stft_filterbank(x, self.stft.wcos, self.stft.wsin, basis, int(self.stft.stride), int(pad),
                                       int(mode), float(self.power), 1e-8 if self.stft.trainable else 0.0,
                                       engine.resolve_precision(self.stft.precision, "f16x3"))
        fused = engine.fused_filterbank_plan(self, basis, x, self.stft, self.power)
        if fused is not None:  # reduction fused into the contraction's epilogue
            return self.stft._spectrum(x, engine.EPI_POWER, power=self.power, fb=basis, fb_support=fused)
        padded = engine.frame_major_filterbank_plan(self, basis, x, self.stft)
        if padded is not None:  # dense filterbank (gammatone): frame-major power spectrogram + a contraction over its bins
            spec = self.stft._spectrum(x, engine.EPI_POWER, power=self.power, out_frame_major=padded.shape[1])
            if spec is not None:
                return engine.filterbank_frame_major(padded, spec)
        spec = self.stft._spectrum(x, engine.EPI_POWER, power=self.power)
        return engine.filterbank_autograd(basis, spec)
