"""nnaudio_amd: MI355X-native spectrogram front-end, drop-in for
``nnAudio.features.{STFT, MelSpectrogram, Gammatonegram, CQT1992v2 (CQT), CQT2010v2, VQT}``.

``forward`` routes through the C ABI of ``csrc/libmispec.so`` (hand-written HIP for gfx950).
"""
__version__ = "0.1.0"


def set_precision(name):
    """Process-wide arithmetic of the framed contraction: "fp32" (default; fp32 MFMA) or
    "bf16x3" (split-bf16 operands on the bf16 MFMA, fp32 accumulate).  A module's own
    ``precision`` attribute, when not None, takes precedence."""
    from . import engine

    engine.set_precision(name)


def get_precision():
    from . import engine

    return engine.get_precision()


def invalidate_caches(module):
    """Forget the operands derived from a module's bases (split / folded planes, kernel supports);
    only needed after edits through ``tensor.data`` (see ``engine.DerivedCache``)."""
    from . import engine

    engine.invalidate_caches(module)
