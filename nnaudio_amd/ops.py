"""``torch.library`` registration of the forward entry points: the ctypes calls of
``engine`` wrapped as custom ops with fake (meta) implementations, so that ``torch.compile``
traces a feature module as opaque device ops instead of graph-breaking at the C boundary.

    mispec::framed_gemm    pad + 2 x conv1d + epilogue (stft.py:278-316, cqt.py:740-780,
                           utils.py:498-521)
    mispec::filterbank     torch.matmul(mel_basis, spec)            (mel.py:188)
    mispec::fir_decimate   conv1d(x, taps, stride, padding)         (utils.py:98-99)
    mispec::power_to_db    MFCC's dB stage                          (mel.py:263-279)

The modules route through these whenever they run under ``torch.compile`` and no autograd graph
is needed (``engine.*_autograd``).  Operands derived from a basis (split / folded planes, kernel
supports) are looked up inside the op -- at run time, on the real tensors -- in a process-wide cache
keyed on what the tensor IS in memory (``_ViewCache``: a compiled graph hands the ops fresh view
objects every call), so nothing about them is baked into a traced graph.
"""
from typing import List, Optional

import torch

from . import engine

class _ViewCache:
    """Derived operands of the tensors an op receives at run time.  Under torch.compile a module's
    ``wsin[:freq_bins]`` or ``kernels.reshape(..)[first:]`` arrives as a FRESH view object on every
    call, so the key is the view's identity in memory -- (storage address, offset, shape, strides) --
    and an entry is valid while the tensors' version counter (shared by all views of a storage) is
    unchanged.  Every entry holds its storages alive, so a cached address cannot be recycled for
    another tensor; at most ``cap`` entries (least recently used dropped)."""

    def __init__(self, cap=32):
        self.cap = cap
        self.entries = {}

    @staticmethod
    def _key(t):
        return (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype)

    def get(self, tensors, extra, build):
        tensors = tuple(t for t in tensors if t is not None)
        key = (tuple(self._key(t) for t in tensors), extra)
        vers = tuple(t._version for t in tensors)
        hit = self.entries.get(key)
        if hit is not None and hit[1] == vers:
            self.entries[key] = self.entries.pop(key)  # most recently used last
            return hit[2]
        val = build()
        self.entries.pop(key, None)
        self.entries[key] = (tuple(t.untyped_storage() for t in tensors), vers, val)
        while len(self.entries) > self.cap:
            self.entries.pop(next(iter(self.entries)))
        return val


_prep_cache = _ViewCache()
_support_cache = _ViewCache()


def _prepare(basis_re, basis_im, precision, hop, support):
    if support and precision == "f16x3":  # banks with supports: the strip kernel's scaled fp16 fragments
        frag = engine.frag_basis_f16(basis_re, basis_im)
        if frag is not None:
            return {"basis_split": frag}
    prep = engine.prepare_basis(basis_re, basis_im, precision, hop=hop)
    if support and precision == "fp32" and basis_re.is_cuda:
        # banks with supports in fp32: the chain kernel's copy (LDS delay lines, the tile kernels' bits; None = not served)
        chain = engine.chain_basis_f32(basis_re, basis_im, _supports(basis_re, basis_im).host_copy)
        if chain is not None:
            prep = dict(prep, basis_chain=chain)
    return prep


def _prepared(basis_re, basis_im, precision, hop, support=False):
    """Split / folded planes of a basis for the op's run-time tensors."""
    if basis_im is None:
        return {}
    precision = engine.resolve_precision(precision)
    return _prep_cache.get((basis_re, basis_im), (int(hop), precision, bool(support)),
                           lambda: _prepare(basis_re, basis_im, precision, hop, support))


def _supports(basis_re, basis_im):
    """[start, stop) of the non-zero taps of every kernel row (CQT banks) for the op's run-time tensors."""
    from .features._cqt_common import SupportCache

    return _support_cache.get((basis_re, basis_im), None, lambda: SupportCache().get(basis_re, basis_im))


@torch.library.custom_op("mispec::framed_gemm", mutates_args=())
def framed_gemm(x: torch.Tensor, basis_re: torch.Tensor, basis_im: Optional[torch.Tensor], hop: int,
                pad: int, pad_mode: int, epilogue: int, im_sign: float, eps: float, power: float,
                row_scale: Optional[torch.Tensor], support: bool, precision: str) -> torch.Tensor:
    """``support``: skip the zero taps outside every row's [start, stop) (CQT banks)."""
    prep = _prepared(basis_re, basis_im, precision, hop, support and basis_im is not None)
    sup = _supports(basis_re, basis_im) if support and basis_im is not None else None
    return engine.framed_gemm(x, basis_re, basis_im, hop=hop, pad=pad, pad_mode=pad_mode,
                              epilogue=epilogue, im_sign=im_sign, eps=eps, power=power,
                              row_scale=row_scale, row_support=sup, precision=precision, **prep)


@framed_gemm.register_fake
def _(x, basis_re, basis_im, hop, pad, pad_mode, epilogue, im_sign, eps, power, row_scale, support,
      precision):
    B, L = x.shape[0], x.shape[-1]
    F, K = basis_re.shape[0], basis_re.shape[-1]
    T = (L + 2 * pad - K) // hop + 1
    if epilogue in (engine.EPI_COMPLEX, engine.EPI_PHASE_COSSIN):
        return x.new_empty((B, F, T, 2))
    return x.new_empty((B, F, T))


@torch.library.custom_op("mispec::filterbank", mutates_args=())
def filterbank(fb: torch.Tensor, spec: torch.Tensor) -> torch.Tensor:
    return engine.filterbank(fb, spec)


@filterbank.register_fake
def _(fb, spec):
    return spec.new_empty((spec.shape[0], fb.shape[0], spec.shape[2]))


@torch.library.custom_op("mispec::fir_decimate", mutates_args=())
def fir_decimate(x: torch.Tensor, taps: torch.Tensor, stride: int) -> torch.Tensor:
    return engine.fir_decimate(x, taps, stride)


@fir_decimate.register_fake
def _(x, taps, stride):
    nt = taps.numel()
    L = x.shape[-1]
    return x.new_empty((x.shape[0], (L + 2 * ((nt - 1) // 2) - nt) // stride + 1))


@torch.library.custom_op("mispec::power_to_db", mutates_args=())
def power_to_db(spec: torch.Tensor, amin: float, ref: float, top_db: float) -> torch.Tensor:
    return engine.power_to_db(spec, amin, ref, None if top_db < 0 else top_db)


@power_to_db.register_fake
def _(spec, amin, ref, top_db):
    return torch.empty_like(spec)


# ---------------------------------------------------------------------------------------
# The fused paths as single ops, so that a compiled module launches the same kernels as the eager
# one: STFT power spectrum + filterbank (the fused epilogue when the filterbank is banded), and the
# whole octave recursion of CQT2010v2 / VQT (the fused pyramid kernel in bf16x3).
# ---------------------------------------------------------------------------------------
_fb_cache = _ViewCache()  # (keyed on what the filterbank IS in memory: compiled graphs hand over fresh views)


def _fb_support(fb):
    return _fb_cache.get((fb,), None, lambda: engine.filterbank_support(fb))


@torch.library.custom_op("mispec::stft_filterbank", mutates_args=())
def stft_filterbank(x: torch.Tensor, basis_re: torch.Tensor, basis_im: torch.Tensor, fb: torch.Tensor,
                    hop: int, pad: int, pad_mode: int, power: float, eps: float,
                    precision: str) -> torch.Tensor:
    """``matmul(fb, |STFT(x)| ** power)`` (mel.py:184-189) -> (B, n_filters, T): the reduction rides
    in the contraction's epilogue when the filterbank is banded (``engine.fused_filterbank_ok``),
    else it is the separate filterbank kernel."""
    prep = _prepared(basis_re, basis_im, precision, hop)
    kw = dict(hop=hop, pad=pad, pad_mode=pad_mode, epilogue=engine.EPI_POWER, im_sign=-1.0, eps=eps,
              power=power, precision=precision)
    sup, coverage = _fb_support(fb)
    if x.is_cuda and engine.fused_filterbank_ok(power, coverage, fb.shape[0]):
        return engine.framed_gemm(x, basis_re, basis_im, fb=fb, fb_support=sup, **kw, **prep)
    return engine.filterbank(fb, engine.framed_gemm(x, basis_re, basis_im, **kw, **prep))


@stft_filterbank.register_fake
def _(x, basis_re, basis_im, fb, hop, pad, pad_mode, power, eps, precision):
    L, K = x.shape[-1], basis_re.shape[-1]
    return x.new_empty((x.shape[0], fb.shape[0], (L + 2 * pad - K) // hop + 1))


_octave_caches = _ViewCache()  # first bank (as it is in memory) -> (OctaveCache, [SupportCache])


def _octave_state(first_bank, n_oct):
    """The derived-operand caches of one module's octave recursion.  Keyed on the first bank's place
    in memory (a compiled graph may hand the op a fresh view object every call; an id() key would
    then rebuild the split planes -- kernels plus host syncs -- inside every compiled forward).  The
    caches themselves track in-place changes of every bank (engine.DerivedCache), so the entry does
    not depend on the bank's version counter."""
    from .features._cqt_common import OctaveCache, SupportCache

    key = (_ViewCache._key(first_bank), int(n_oct))
    hit = _octave_caches.entries.get(key)
    if hit is not None:
        _octave_caches.entries[key] = _octave_caches.entries.pop(key)
        return hit[2]
    state = (OctaveCache(by_memory=True), [SupportCache(by_memory=True) for _ in range(n_oct)])
    _octave_caches.entries[key] = ((first_bank.untyped_storage(),), None, state)
    while len(_octave_caches.entries) > _octave_caches.cap:
        _octave_caches.entries.pop(next(iter(_octave_caches.entries)))
    return state


@torch.library.custom_op("mispec::octave_recursion", mutates_args=())
def octave_recursion(x: torch.Tensor, kernels_real: List[torch.Tensor], kernels_imag: List[torch.Tensor],
                     lenghts: torch.Tensor, lowpass: torch.Tensor, hop: int, n_bins: int,
                     downsample_factor: float, pad_mode: str, output_format: str,
                     normalization_type: str, precision: str) -> torch.Tensor:
    """The octave loop of CQT2010v2 / VQT (cqt.py:1085-1131, vqt.py:160-215) for frozen kernels, as ONE
    op: at run time it is the eager path -- the fused pyramid kernel in bf16x3, grouped contractions
    in fp32 -- with its derived operands cached per bank."""
    from .features._cqt_common import octave_recursion as run

    cache, supports = _octave_state(kernels_real[0], len(kernels_real))
    return run(x, list(zip(kernels_real, kernels_imag)), lenghts, hop, n_bins, lowpass, downsample_factor,
               pad_mode, output_format, normalization_type, False, supports=supports, graph=False,
               precision=precision, cache=cache)


@octave_recursion.register_fake
def _(x, kernels_real, kernels_imag, lenghts, lowpass, hop, n_bins, downsample_factor, pad_mode,
      output_format, normalization_type, precision):
    L, K = x.shape[-1], kernels_real[0].shape[-1]
    T = (L + 2 * (K // 2) - K) // hop + 1
    if output_format in ("Complex", "Phase"):
        return x.new_empty((x.shape[0], n_bins, T, 2))
    return x.new_empty((x.shape[0], n_bins, T))
