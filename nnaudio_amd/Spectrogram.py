"""Legacy import location kept for drop-in compatibility
(reference: nnAudio/Spectrogram.py:1-8 re-exports the feature classes and warns)."""
from .features import *  # noqa: F401,F403
import warnings

warnings.warn(
    "importing from `Spectrogram` is deprecated, import from `nnaudio_amd.features` instead",
    Warning,
)
