"""nnaudio_amd: MI355X-native spectrogram front-end, drop-in for
``nnAudio.features.{STFT, MelSpectrogram, Gammatonegram, CQT1992v2 (CQT), CQT2010v2, VQT}``.

``forward`` routes through the C ABI of ``csrc/libmispec.so`` (hand-written HIP for gfx950).
"""
__version__ = "0.1.0"
