"""Mel spectrogram module (drop-in for ``nnAudio.features.MelSpectrogram``,
reference: Installation/nnAudio/features/mel.py:9-194).

``forward`` = framed MFMA contraction with the ``|.|**power`` epilogue fused (the
reference's sqrt -> pow round trip disappears for power = 2) followed by the filterbank
contraction kernel."""
from time import time

import torch
import torch.nn as nn

from .. import engine
from ..basis import mel_filterbank
from ..utils import broadcast_dim
from .stft import STFT


class MelSpectrogram(nn.Module):
    """``(batch, n_mels, frames)`` mel spectrogram; constructor, attributes and
    ``state_dict`` keys (``mel_basis``, ``stft.wsin`` ...) as the reference."""

    def __init__(
        self,
        sr=22050,
        n_fft=2048,
        win_length=None,
        n_mels=128,
        hop_length=512,
        window="hann",
        center=True,
        pad_mode="reflect",
        power=2.0,
        htk=False,
        fmin=0.0,
        fmax=None,
        norm=1,
        trainable_mel=False,
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
        self.trainable_mel = trainable_mel
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
        mel_basis = torch.from_numpy(
            mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk=htk, norm=norm)
        )
        if verbose:
            print("STFT filter created, time used = {:.4f} seconds".format(time() - start))
            print("Mel filter created, time used = {:.4f} seconds".format(time() - start))

        if trainable_mel:
            self.register_parameter("mel_basis", nn.Parameter(mel_basis, requires_grad=True))
        else:
            self.register_buffer("mel_basis", mel_basis)

    def forward(self, x):
        x = broadcast_dim(x)
        engine.grad_guard(self, x)
        self.stft.num_samples = x.shape[-1]
        spec = self.stft._spectrum(x, engine.EPI_POWER, power=self.power)
        return engine.filterbank(self.mel_basis, spec)

    def extra_repr(self) -> str:
        return "Mel filter banks size = {}, trainable_mel={}".format(
            (*self.mel_basis.shape,), self.trainable_mel, self.trainable_STFT
        )
