"""nnaudio_amd: MI355X-native spectrogram front-end, drop-in for
``nnAudio.features.{STFT, MelSpectrogram, Gammatonegram, CQT1992v2 (CQT), CQT2010v2, VQT}``.

``forward`` routes through the C ABI of ``csrc/libmispec.so`` (hand-written HIP for gfx950).
"""
__version__ = "0.1.0"


def set_precision(name):
    """Process-wide arithmetic of the framed contraction: "fp32" (fp32 MFMA), "bf16x3" (split-bf16
    operands on the bf16 MFMA, ~5e-6 of the spectrum peak), "f16x3" (scaled split-fp16 operands on
    the f16 MFMA, ~1e-7 of the peak: fp32 class), or None: every module uses its own default -- the
    fastest arithmetic that meets the reference's own fixtures (STFT family, CQT2010v2, VQT:
    "f16x3"; CQT1992v2: "fp32").  A module's own ``precision`` attribute, when not None, takes
    precedence; the MISPEC_PRECISION environment variable sets the initial override."""
    from . import engine

    engine.set_precision(name)


def get_precision():
    from . import engine

    return engine.get_precision()


def set_fft(enabled):
    """STFT with window x DFT kernels (freq_scale='no', not trainable, n_fft 256 / 512 / 1024 / 2048) evaluates the
    frames' DFT as an fp32 FFT (default); ``set_fft(False)`` or MISPEC_FFT=0 keeps it on the contraction
    kernels in the arithmetic ``precision`` names.  Returns the previous setting."""
    from . import engine

    return engine.set_fft(enabled)


def invalidate_caches(module):
    """Forget the operands derived from a module's bases (split / folded planes, kernel supports);
    only needed after edits through ``tensor.data`` (see ``engine.DerivedCache``)."""
    from . import engine

    engine.invalidate_caches(module)
