"""Mel spectrogram module (drop-in for ``nnAudio.features.MelSpectrogram``,
reference: Installation/nnAudio/features/mel.py:9-194).

``forward`` = framed MFMA contraction with the ``|.|**power`` epilogue fused (the
reference's sqrt -> pow round trip disappears for power = 2) followed by the filterbank
contraction kernel."""
from time import time

import torch
import torch.nn as nn

from .. import engine
from ..basis import dct_ortho_matrix, mel_filterbank
from ..utils import ParameterError, broadcast_dim
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
        self.stft.num_samples = x.shape[-1]
        fused = engine.fused_filterbank_plan(self, self.mel_basis, x, self.stft, self.power)
        if fused is not None:  # reduction fused into the contraction's epilogue
            return self.stft._spectrum(x, engine.EPI_POWER, power=self.power, fb=self.mel_basis,
                                       fb_support=fused)
        spec = self.stft._spectrum(x, engine.EPI_POWER, power=self.power)
        return engine.filterbank_autograd(self.mel_basis, spec)

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
        db = engine.power_to_db_autograd(spec, self._amin_f, self._ref_f, self.top_db)
        return engine.filterbank_autograd(self._dct_basis, db)

    def extra_repr(self) -> str:
        return "n_mfcc = {}".format((self.n_mfcc))
