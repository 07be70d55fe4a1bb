"""Feature modules with the reference's public names
(reference: nnAudio/features/__init__.py:6-14)."""
from .stft import STFT, iSTFT
from .mel import MFCC, MelSpectrogram
from .gammatone import Gammatonegram
from .cqt import CQT1992v2, CQT2010v2, CQT
from .cqt_freq import CQT1992, CQT2010
from .vqt import VQT

__all__ = ["STFT", "iSTFT", "MelSpectrogram", "MFCC", "Gammatonegram", "CQT1992v2", "CQT2010v2", "CQT", "CQT1992", "CQT2010", "VQT"]
