"""Gammatonegram module (drop-in for ``nnAudio.features.Gammatonegram``,
reference: Installation/nnAudio/features/gammatone.py:9-194): the mel pipeline with a
(dense) gammatone filterbank."""
from time import time

import torch
import torch.nn as nn

from .. import engine
from ..basis import gammatone_filterbank
from ..utils import broadcast_dim
from .stft import STFT


class Gammatonegram(nn.Module):
    """``(batch, n_bins, frames)`` gammatone spectrogram; constructor, attributes and
    ``state_dict`` keys (``gammatone_basis``, ``stft.wsin`` ...) as the reference."""

    def __init__(
        self,
        sr=22050,
        n_fft=2048,
        win_length=None,
        n_bins=64,
        hop_length=512,
        window="hann",
        center=True,
        pad_mode="reflect",
        power=2.0,
        htk=False,
        fmin=0.0,
        fmax=None,
        norm=1,
        trainable_bins=False,
        trainable_STFT=False,
        verbose=True,
        **kwargs
    ):
        super().__init__()
        self.stride = hop_length
        self.center = center
        self.pad_mode = pad_mode
        self.n_fft = n_fft
        self.power = power
        self.trainable_bins = trainable_bins
        self.trainable_STFT = trainable_STFT

        self.stft = STFT(
            n_fft=n_fft,
            win_length=win_length,
            freq_bins=None,
            hop_length=hop_length,
            window=window,
            freq_scale="no",
            center=center,
            pad_mode=pad_mode,
            sr=sr,
            trainable=trainable_STFT,
            output_format="Magnitude",
            verbose=verbose,
            **kwargs
        )

        start = time()
        basis = torch.from_numpy(gammatone_filterbank(sr, n_fft, n_bins, fmin, fmax))
        if verbose:
            print("STFT filter created, time used = {:.4f} seconds".format(time() - start))
            print("Gammatone filter created, time used = {:.4f} seconds".format(time() - start))

        if trainable_bins:
            self.register_parameter("gammatone_basis", nn.Parameter(basis, requires_grad=True))
        else:
            self.register_buffer("gammatone_basis", basis)

    def forward(self, x):
        x = broadcast_dim(x)
        self.stft.num_samples = x.shape[-1]
        fused = engine.fused_filterbank_plan(self, self.gammatone_basis, x, self.stft, self.power)
        if fused is not None:  # reduction fused into the contraction's epilogue
            return self.stft._spectrum(x, engine.EPI_POWER, power=self.power, fb=self.gammatone_basis,
                                       fb_support=fused)
        spec = self.stft._spectrum(x, engine.EPI_POWER, power=self.power)
        return engine.filterbank_autograd(self.gammatone_basis, spec)

    def extra_repr(self) -> str:
        return "Gammatone filter banks size = {}, trainable_bins={}".format(
            (*self.gammatone_basis.shape,), self.trainable_bins, self.trainable_STFT
        )
