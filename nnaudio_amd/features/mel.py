"""Mel spectrogram module (drop-in for ``nnAudio.features.MelSpectrogram``,
reference: Installation/nnAudio/features/mel.py:9-194).

The pipeline itself (STFT power spectrum -> filterbank) is shared with Gammatonegram:
``features/_filterbank.py``."""

import torch
import torch.nn as nn

from .. import engine
from ..basis import dct_ortho_matrix, mel_filterbank
from ..utils import ParameterError
from ._filterbank import FilterbankSpectrogram


class MelSpectrogram(FilterbankSpectrogram):
    """``(batch, n_mels, frames)`` mel spectrogram; constructor, attributes and
    ``state_dict`` keys (``mel_basis``, ``stft.wsin`` ...) as the reference."""

    _basis_name, _label = "mel_basis", "Mel"

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
        self.trainable_mel = trainable_mel
        self._build(
            lambda: torch.from_numpy(mel_filterbank(sr, n_fft, n_mels, fmin, fmax, htk=htk, norm=norm)),
            sr=sr, n_fft=n_fft, win_length=win_length, hop_length=hop_length, window=window,
            center=center, pad_mode=pad_mode, power=power, trainable_basis=trainable_mel,
            trainable_STFT=trainable_STFT, verbose=verbose, stft_kwargs=kwargs)

    def extra_repr(self) -> str:
        return "Mel filter banks size = {}, trainable_mel={}".format(
            (*self.mel_basis.shape,), self.trainable_mel, self.trainable_STFT
        )


class MFCC(nn.Module):
    """Mel-frequency cepstral coefficients: ``MelSpectrogram`` -> ``power_to_db`` (per-clip
    ``top_db`` floor) -> orthonormal DCT-II over the mel axis -> first ``n_mfcc`` rows.
    Same constructor, buffers (``amin``, ``ref``, ``melspec_layer.*``) and output
    ``(batch, n_mfcc, frames)`` as the reference (mel.py:197-329).  The DCT runs as the same planar
    contraction kernel as the mel filterbank, with the (n_mfcc, n_mels) cosine matrix the
    reference evaluates through an FFT."""

    def __init__(self, sr=22050, n_mfcc=20, norm="ortho", verbose=True, ref=1.0, amin=1e-10,
                 top_db=80.0, **kwargs):
        super().__init__()
        self.melspec_layer = MelSpectrogram(sr=sr, verbose=verbose, **kwargs)
        self.m_mfcc = n_mfcc
        if amin <= 0:
            raise ParameterError("amin must be strictly positive")
        self.register_buffer("amin", torch.tensor([amin]))
        self.register_buffer("ref", torch.abs(torch.tensor([ref])))
        # host copies of the two scalars: forward must not read device buffers back (a
        # device-to-host sync per call, and it would break stream capture); refreshed by
        # load_state_dict
        self._amin_f, self._ref_f = float(amin), abs(float(ref))
        self.top_db = top_db
        self.n_mfcc = n_mfcc
        n_mels = self.melspec_layer.mel_basis.shape[0]
        # derived constant, not part of the reference's state_dict
        self.register_buffer("_dct_basis", torch.from_numpy(dct_ortho_matrix(min(n_mfcc, n_mels), n_mels)),
                             persistent=False)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._amin_f = float(self.amin.detach().cpu())
        self._ref_f = float(self.ref.detach().cpu())

    def forward(self, x):
        spec = self.melspec_layer(x)
        if self.top_db is not None and self.top_db < 0:
            raise ParameterError("top_db must be non-negative")
        if spec.is_cuda and not engine.compiling() and not (torch.is_grad_enabled() and spec.requires_grad):
            # nothing to differentiate: decibels + DCT in one launch (mispec_mfcc_tail_f32)
            y = engine.mfcc_tail(spec, self._amin_f, self._ref_f, self.top_db, self._dct_basis)
            if y is not None:
                return y
        db = engine.power_to_db_autograd(spec, self._amin_f, self._ref_f, self.top_db)
        return engine.filterbank_autograd(self._dct_basis, db)

    def extra_repr(self) -> str:
        return "n_mfcc = {}".format((self.n_mfcc))
