"""Gammatonegram module (drop-in for ``nnAudio.features.Gammatonegram``,
reference: Installation/nnAudio/features/gammatone.py:9-194): the mel pipeline
(``features/_filterbank.py``) with a (dense) gammatone filterbank."""
import torch

from ..basis import gammatone_filterbank
from ._filterbank import FilterbankSpectrogram


class Gammatonegram(FilterbankSpectrogram):
    """``(batch, n_bins, frames)`` gammatone spectrogram; constructor, attributes and
    ``state_dict`` keys (``gammatone_basis``, ``stft.wsin`` ...) as the reference."""

    _basis_name, _label = "gammatone_basis", "Gammatone"

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
        self.trainable_bins = trainable_bins
        self._build(
            lambda: torch.from_numpy(gammatone_filterbank(sr, n_fft, n_bins, fmin, fmax)),
            sr=sr, n_fft=n_fft, win_length=win_length, hop_length=hop_length, window=window,
            center=center, pad_mode=pad_mode, power=power, trainable_basis=trainable_bins,
            trainable_STFT=trainable_STFT, verbose=verbose, stft_kwargs=kwargs)

    def extra_repr(self) -> str:
        return "Gammatone filter banks size = {}, trainable_bins={}".format(
            (*self.gammatone_basis.shape,), self.trainable_bins, self.trainable_STFT
        )
