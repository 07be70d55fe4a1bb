"""Host-side dispatch from torch CUDA tensors into the C ABI of ``libmispec.so``.

torch is plumbing here (device memory from the caching allocator, the caller's current
stream and device); every FLOP of the hot path is issued by the HIP kernels.  There is no
eager / torch fallback and a missing extension raises.  CPU tensors are served by the library's
host entries (``mispec_*_host_f32``: the same arithmetic on the calling thread) for every forward
path -- STFT / filterbank / CQT / MFCC forward and the inverse STFT; backward on CPU tensors raises.
"""
import ctypes
import os
import threading
import weakref

import numpy as np
import torch

from . import _abi
from ._abi import (  # noqa: F401  (re-exported for the feature modules)
    EPI_COMPLEX, EPI_MAGNITUDE, EPI_PHASE_ATAN2, EPI_PHASE_COSSIN, EPI_POWER, EPI_REAL,
    PAD_NONE, PAD_REFLECT, PAD_ZERO, PREC_BF16X3, PREC_F16X3, PREC_F32, TILE_AUTO,
)
from .basis import decimated_length

# Arithmetic of the framed CONTRACTION kernels (include/mispec.h, MISPEC_PREC_*).  It does not apply to the
# FFT route: STFT-family modules whose kernels are window x DFT (freq_scale='no', frozen, n_fft 256 / 512 /
# 1024 / 2048) evaluate every frame's DFT as an fp32 FFT whatever `precision` says, an explicit "fp32" included
# (2e-7 of the peak; `set_fft(False)` puts them on the contraction kernels, where "fp32" is the reference's
# summation order).
#   "fp32"   fp32 MFMA; banks with supports: one float32 FMA chain over the taps per output = the reference's conv1d (CQT1992v2's default)
#   "bf16x3" split-bf16 operands on the 16x faster bf16 MFMA, fp32 accumulate: ~5e-6 of the
#            spectrum peak, inside the 1e-4 bar; problems it does not cover run in fp32
#   "f16x3"  the same three MFMAs per product on (hi, lo) fp16 pairs of power-of-two scaled
#            operands: 22 operand bits, ~1e-7 of the peak (fp32 class, ~160 dB of dynamic range
#            where bf16x3 has ~105); problems it does not cover run in fp32
_PRECISIONS = {"fp32": PREC_F32, "bf16x3": PREC_BF16X3, "f16x3": PREC_F16X3}
# Process-wide override (nnaudio_amd.set_precision, or the MISPEC_PRECISION environment variable at
# import); None: every module uses its own default -- the fastest arithmetic that meets the
# reference's own fixtures for that module (tests/test_gpu_parity.py::test_reference_ground_truths):
#   STFT (and MelSpectrogram / Gammatonegram / MFCC through it)   "f16x3"
#   CQT2010v2 / VQT                                                "f16x3" (the fused octave kernel)
#   CQT1992v2                                                      "fp32" (the reference's summation order)
_default_precision = os.environ.get("MISPEC_PRECISION") or None
if _default_precision is not None and _default_precision not in _PRECISIONS:
    raise ValueError("MISPEC_PRECISION must be one of %s" % sorted(_PRECISIONS))


def set_precision(name):
    """Process-wide arithmetic for modules whose ``precision`` attribute is None; ``None`` (or
    "auto") returns to the per-module defaults."""
    global _default_precision
    if name == "auto":
        name = None
    if name is not None and name not in _PRECISIONS:
        raise ValueError("precision must be None or one of %s" % sorted(_PRECISIONS))
    _default_precision = name


def get_precision():
    """The process-wide override, or None when the modules use their own defaults."""
    return _default_precision


_fft = os.environ.get("MISPEC_FFT", "1") not in ("0", "false", "off")


def set_fft(enabled):
    """STFT-family modules whose kernels are window x DFT (``freq_scale='no'``, not trainable, n_fft 256 ...
    2048, and 4096 as a composite of two 2048-point halves) evaluate the frames' DFT as an FFT instead of contracting them with the kernels
    (csrc/stft_fft.inl); ``set_fft(False)`` (or ``MISPEC_FFT=0``) keeps every module on the contraction
    kernels -- the arithmetic ``precision`` selects.  Returns the previous setting."""
    global _fft
    old, _fft = _fft, bool(enabled)
    return old


def fft_enabled():
    return _fft


_istft_fused = os.environ.get("MISPEC_ISTFT_FUSED", "1") not in ("0", "false", "off")


def set_istft_fused(enabled):
    """Inverse STFT on the FFT route: synthesis + overlap-add in one launch (``mispec_istft_fft_f32``) where it
    serves the shape; ``False`` (or ``MISPEC_ISTFT_FUSED=0``) keeps the two launches.  Returns the previous setting."""
    global _istft_fused
    old, _istft_fused = _istft_fused, bool(enabled)
    return old


_octave_stream = os.environ.get("MISPEC_OCTAVE_STREAM", "1") not in ("0", "false", "off")


def set_octave_stream(enabled):
    """CQT2010v2 / VQT on the fused kernels: the streaming octave kernel (csrc/octave_stream.hip) where it
    serves the shape, else -- and with ``set_octave_stream(False)`` / ``MISPEC_OCTAVE_STREAM=0`` always -- the
    pyramid kernel (csrc/octave_pyramid.inl).  Returns the previous setting."""
    global _octave_stream
    old, _octave_stream = _octave_stream, bool(enabled)
    return old


def octave_stream_enabled():
    return _octave_stream


def resolve_precision(name=None, default="fp32"):
    """``name`` (a module's attribute / a call's argument), else the process-wide override, else
    ``default`` (the caller's own default)."""
    name = name or _default_precision or default
    if name not in _PRECISIONS:
        raise ValueError("precision must be one of %s, got %r" % (sorted(_PRECISIONS), name))
    return name


_side_streams = {}


def _side_stream(dev, i):
    """A few persistent side streams per device (the backward's split-K chunks run on them side by side)."""
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), i)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=dev)
    return _side_streams[key]


class DerivedCache:
    """A tensor derived from a module's basis buffers (kernel supports, split-bf16 planes),
    rebuilt whenever the buffers are replaced or modified in place (``load_state_dict``,
    ``.to``, an optimiser step).  An entry is valid only while the SAME tensor objects
    (weak references, so a new tensor that reuses a freed address can never alias) still have
    the ``_version`` they were built from.

    One entry per device: ``nn.DataParallel`` replicas share this object and call from one
    host thread per GPU, so entries are only ever replaced whole (no torn state).

    Limit: an edit made through ``.data`` (``p.data.mul_()``, weight clipping, EMA swaps of older
    optimisers) bumps no version counter and moves no storage, so it cannot be seen here: call
    ``nnaudio_amd.invalidate_caches(module)`` after such an edit (``p.data = other`` IS seen: the
    storage address is part of the key)."""

    def __init__(self, by_memory=False):
        # by_memory: sources are recognised by what they ARE in memory (storage address, offset,
        # shape, strides) instead of by object identity, and their storages are held alive -- for the
        # custom ops, which may receive a fresh view object of the same buffer on every call
        self._entries = {}
        self._by_memory = by_memory

    def clear(self):
        self._entries = {}

    @staticmethod
    def _mem_key(t):
        return (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()), t.dtype)

    def get(self, sources, build, extra=None):
        sources = tuple(sources)
        dev = sources[0].device
        vers = tuple((s._version, s.data_ptr()) for s in sources)
        hit = self._entries.get(dev)
        if hit is not None:
            refs, hvers, hextra, val = hit
            if hextra == extra and hvers == vers and len(refs) == len(sources):
                if self._by_memory:
                    if all(r[0] == self._mem_key(s) for r, s in zip(refs, sources)):
                        return val
                elif all(r() is s for r, s in zip(refs, sources)):
                    return val
        val = build()
        refs = (tuple((self._mem_key(s), s.untyped_storage()) for s in sources) if self._by_memory
                else tuple(weakref.ref(s) for s in sources))
        self._entries[dev] = (refs, vers, extra, val)
        return val

_PAD_MODES = {"constant": PAD_ZERO, "reflect": PAD_REFLECT, None: PAD_NONE}


# ---------------------------------------------------------------------------------------
# Output placement: lets a caller that owns a larger buffer (the all-gather buffer of
# nnaudio_amd.dist, SURVEY 8e "each rank computes directly into its slice") have a module's
# output written there, although forward(x) has no `out=` argument (the reference's signature).
# ---------------------------------------------------------------------------------------
_placement = threading.local()


class _Slot:
    def __init__(self, tensor):
        self.tensor = tensor
        self.taken = False


class output_into:
    """``with output_into(t) as slot: y = module(x)``: the first output-sized allocation of the
    forward (same shape / dtype / device as ``t``, which must be contiguous) uses ``t`` instead of
    fresh memory; ``slot.taken`` tells whether that happened (then ``y`` normally IS ``t``: check
    ``y.data_ptr()``).  One-shot, per host thread, no effect on other allocations."""

    def __init__(self, tensor):
        if not tensor.is_contiguous():
            raise RuntimeError("output_into needs a contiguous tensor")
        self.slot = _Slot(tensor)

    def __enter__(self):
        self.prev = getattr(_placement, "slot", None)
        _placement.slot = self.slot
        return self.slot

    def __exit__(self, *exc):
        _placement.slot = self.prev
        return False


def alloc_out(shape, device, zero=False):
    """Fresh float32 output tensor -- or the caller's ``output_into`` tensor when it fits."""
    slot = None if compiling() else getattr(_placement, "slot", None)
    if slot is not None and not slot.taken:
        t = slot.tensor
        if tuple(t.shape) == tuple(shape) and t.dtype == torch.float32 and t.device == device:
            slot.taken = True
            return t.zero_() if zero else t
    return (torch.zeros if zero else torch.empty)(tuple(shape), dtype=torch.float32, device=device)


# Host path: the forward of STFT / MelSpectrogram / Gammatonegram / CQT1992v2 / CQT2010v2 / VQT on CPU
# tensors runs on libmispec's own host loops (mispec_*_host_f32: plain C++, plumbing-sized inputs --
# the reference's forward computes wherever its input lives, stft.py:290-293).  Everything else
# (backward, MFCC, the inverse STFT, the frequency-domain CQTs) needs the GPU; set_host_path(False)
# makes every CPU tensor raise again.
_host_path = os.environ.get("MISPEC_HOST_PATH", "1") != "0"


def set_host_path(enabled):
    global _host_path
    _host_path = bool(enabled)


def _gpu_only(t):
    return RuntimeError(
        "nnaudio_amd computes on the GPU only (libmispec HIP kernels); got a %s tensor. "
        "Move the module and its input with .to('cuda') -- there is no CPU fallback for this operation."
        % t.device)


def _require_device(*tensors, host_ok=False):
    """The common device of the tensors.  CPU tensors raise unless the caller has a host
    implementation (``host_ok``) and the host path is enabled -- then the CPU device is returned."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda and not (host_ok and _host_path and t.device.type == "cpu"):
            raise _gpu_only(t)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(
                "Expected all tensors to be on the same device, but found at least two devices, "
                "%s and %s!" % (dev, t.device)
            )
    return dev


def _f32(t, what):
    if t.dtype != torch.float32:
        raise RuntimeError(
            "%s must be float32 (the reference's conv1d weights are float32), got %s"
            % (what, t.dtype)
        )
    return t


def _rows(t, what):
    """(F,1,K) / (F,K) -> 2-D view with unit inner stride."""
    t = _f32(t, what)
    if t.dim() == 3:
        t = t.reshape(t.shape[0], t.shape[-1])
    if t.dim() != 2:
        raise RuntimeError("%s must be (bins, 1, kernel) or (bins, kernel)" % what)
    if t.stride(1) != 1:
        t = t.contiguous()
    return t


def _signal(x):
    """(B,1,L) / (B,L) -> 2-D float32 view with unit inner stride."""
    x = _f32(x, "input")
    if x.dim() == 3:
        if x.shape[1] != 1:
            raise RuntimeError(
                "Given groups=1, expected input to have 1 channel, got %d" % x.shape[1]
            )
        x = x[:, 0, :]
    if x.dim() != 2:
        raise RuntimeError("signal must be (batch, length)")
    if x.stride(1) != 1:
        x = x.contiguous()
    return x


def n_frames(length, kernel, hop, pad):
    span = length + 2 * pad - kernel
    if span < 0:
        raise RuntimeError(
            "Calculated padded input size per channel: (%d). Kernel size: (%d). "
            "Kernel size can't be greater than actual input size" % (length + 2 * pad, kernel)
        )
    if hop <= 0:
        raise RuntimeError("non-positive stride is not supported")
    return span // hop + 1


def _framed_args(x, basis_re, basis_im, *, hop, pad, pad_mode, epilogue, im_sign=-1.0,
                 eps=0.0, power=2.0, row_scale=None, row_support=None, out=None,
                 out_rows_total=None, out_row_offset=0, tile=TILE_AUTO, _debug=0,
                 need_workspace=True, precision=None, basis_split=None, basis_fold=None, fb=None,
                 fb_support=None, basis_fold2=None, row_support_host=None, fft=None, out_frame_major=0,
                 basis_chain=None):
    """Validate one framed-contraction problem and fill its C argument block.
    Returns (args, out, device, keepalive tensors).  ``out_frame_major = Fp`` (> 0): the output is
    ``(B, T, Fp)``, a frame's bins contiguous and the columns ``[n_bins, Fp)`` zero (mispec.h,
    out_frame_major: the FFT path only)."""
    dev = _require_device(x, basis_re, basis_im, row_scale, row_support, out, fb, fb_support,
                          host_ok=fb is None and not _debug)
    host = dev.type == "cpu"
    if host:  # plain loops over host pointers: no workspace, no derived operands, one arithmetic
        need_workspace, basis_split, basis_fold, basis_fold2 = False, None, None, None
    x = _signal(x)
    wr = _rows(basis_re, "basis_re")
    wi = _rows(basis_im, "basis_im") if basis_im is not None else None
    if wi is not None and (wi.shape != wr.shape or wi.stride(0) != wr.stride(0)):
        raise RuntimeError("real / imaginary bases must have identical shape and layout")
    B, L = x.shape
    F, K = wr.shape
    if pad_mode == PAD_REFLECT and pad >= L:
        raise RuntimeError(
            "Argument #4: Padding size should be less than the corresponding input dimension, "
            "but got: padding (%d, %d) at dimension 2 of input %s" % (pad, pad, [B, 1, L])
        )
    T = n_frames(L, K, hop, pad)
    two = epilogue in (EPI_COMPLEX, EPI_PHASE_COSSIN)
    rows_total = F if out_rows_total is None else out_rows_total
    if fb is not None:
        # fused filterbank: the launch clears and fills a (B, n_fb, T) tensor
        fb = _f32(fb, "filterbank")
        if fb.dim() != 2 or fb.shape[1] != F or fb.stride(1) != 1:
            raise RuntimeError("fused filterbank must be (n_filters, %d) with unit column stride" % F)
        if (fb_support is None or fb_support.dtype != torch.int32
                or tuple(fb_support.shape) != (fb.shape[0], 2)):
            raise RuntimeError("fb_support must be int32 (n_filters, 2)")
        if out is not None or out_rows_total is not None or out_row_offset or two:
            raise RuntimeError("fused filterbank writes its own (B, n_filters, T) output")
        fb_support = fb_support.contiguous()
        out = alloc_out((B, fb.shape[0], T), dev)  # (cleared by the library: no ATen launch in a forward)
        rows_total = fb.shape[0]
    elif out_frame_major:
        if out is not None or out_rows_total is not None or out_row_offset or two or host:
            raise RuntimeError("frame-major output: its own (B, T, Fp) tensor, one float per bin, device only")
        out = alloc_out((B, T, int(out_frame_major)), dev)
    elif out is None:
        shape = (B, rows_total, T, 2) if two else (B, rows_total, T)
        out = alloc_out(shape, dev)
    else:
        want = (B, rows_total, T, 2) if two else (B, rows_total, T)
        if tuple(out.shape) != want or not out.is_contiguous() or out.dtype != torch.float32:
            raise RuntimeError("out must be a contiguous float32 tensor of shape %s" % (want,))
    if fb is None and (out_row_offset < 0 or out_row_offset + F > rows_total):
        raise RuntimeError("row block [%d, %d) outside the output's %d rows"
                           % (out_row_offset, out_row_offset + F, rows_total))
    if row_scale is not None:
        row_scale = _f32(row_scale, "row_scale").contiguous()
        if row_scale.numel() != F:
            raise RuntimeError("row_scale must have one entry per basis row")
    if row_support is not None:
        if row_support.dtype != torch.int32 or tuple(row_support.shape) != (F, 2):
            raise RuntimeError("row_support must be int32 (n_bins, 2)")
        # the same values in host memory (library-side launch planning): the explicit argument, else
        # the copy that features._cqt_common.SupportCache attaches to the tensor it returns (lost by
        # any slice / copy of that tensor: callers that slice pass the argument)
        support_host = row_support_host if row_support_host is not None else getattr(row_support, "host_copy", None)
        row_support = row_support.contiguous()

    E = 2 if two else 1
    a = _abi.FramedGemmArgs()
    a.struct_size = ctypes.sizeof(_abi.FramedGemmArgs)
    a.tile = int(tile)
    a.x = x.data_ptr()
    a.x_clip_stride = x.stride(0)
    a.n_clips, a.n_samples = B, L
    a.hop, a.pad, a.pad_mode, a.n_frames = int(hop), int(pad), int(pad_mode), T
    a.basis_re = wr.data_ptr()
    a.basis_im = wi.data_ptr() if wi is not None else None
    a.basis_row_stride = wr.stride(0)
    a.n_bins, a.kernel = F, K
    a.row_support = row_support.data_ptr() if row_support is not None else None
    a.row_scale = row_scale.data_ptr() if row_scale is not None else None
    a.epilogue = int(epilogue)
    a.im_sign, a.eps, a.power = float(im_sign), float(eps), float(power)
    a.out = out.data_ptr()
    a.out_clip_stride = rows_total * T * E
    a.out_row_stride = T * E
    a.out_row_offset = int(out_row_offset)
    if out_frame_major:
        a.out_clip_stride, a.out_row_stride, a.out_frame_major = T * int(out_frame_major), int(out_frame_major), 1
    a.reserved = int(_debug)  # ablation bits: honoured by libmispec_ablate.so only (framed_gemm)
    keep = [x, wr, wi, row_scale, row_support, fb, fb_support]
    if row_support is not None and support_host is not None:
        if support_host.dtype != np.int32 or support_host.shape != (F, 2) or not support_host.flags.c_contiguous:
            raise RuntimeError("row_support_host must be a C-contiguous int32 (n_bins, 2) array")
        a.row_support_host = support_host.ctypes.data
        keep.append(support_host)
    if fb is not None:
        a.fb, a.fb_support = fb.data_ptr(), fb_support.data_ptr()
        a.fb_row_stride, a.n_fb = fb.stride(0), fb.shape[0]
    if resolve_precision(precision) == "bf16x3" and need_workspace:
        if basis_split is None:  # uncached: callers with a persistent basis pass it in
            basis_split = split_basis(wr, wi)
        a.precision = PREC_BF16X3
        a.basis_split = basis_split.data_ptr()
        a.basis_split_bytes = basis_split.numel() * basis_split.element_size()
        keep.append(basis_split)
    elif basis_split is not None and need_workspace and (row_support is not None or resolve_precision(precision) == "f16x3"):
        # fp32 with the fragment-order copy of a bank with supports (frag_basis_f32): the strip kernel;
        # f16x3 with split_basis_f16 planes (staged dense kernel) or frag_basis_f16 (strip kernel)
        a.basis_split = basis_split.data_ptr()
        a.basis_split_bytes = basis_split.numel() * basis_split.element_size()
        keep.append(basis_split)
    if (basis_chain is not None and need_workspace and row_support is not None and support_host is not None
            and resolve_precision(precision) == "fp32"):
        # fp32 with the chain kernel's copy of a bank with supports (chain_basis_f32): LDS delay lines, same bits
        a.basis_chain = basis_chain.data_ptr()
        a.basis_chain_bytes = basis_chain.numel() * basis_chain.element_size()
        keep.append(basis_chain)
    if basis_fold2 is not None and need_workspace:
        # (planes, max |window|) from fold2_basis() in THIS precision: a quarter of the MFMAs
        planes, wmax = basis_fold2
        a.basis_fold2 = planes.data_ptr()
        a.basis_fold2_bytes = planes.numel() * planes.element_size()
        a.fold2_wmax = float(wmax)
        keep.append(planes)
        # window x DFT bases of 256 ... 2048 taps (powers of two) run as an FFT (fp32) unless told otherwise
        a.no_fft = 0 if (fft_enabled() if fft is None else fft) else 1
    if resolve_precision(precision) == "f16x3" and need_workspace:
        a.precision = PREC_F16X3  # (the library runs what the second fold does not cover in fp32)
    if basis_fold is not None and need_workspace:
        # (planes, folded taps) from fold_basis() in THIS precision: half the MFMAs
        planes, taps = basis_fold
        a.basis_fold = planes.data_ptr()
        a.basis_fold_bytes = planes.numel() * planes.element_size()
        a.fold_taps = int(taps)
        keep.append(planes)
    if need_workspace:
        lib = _abi.load_ablate() if a.reserved else _abi.load()
        need = lib.mispec_framed_gemm_workspace_bytes(ctypes.byref(a))
        if need < 0:
            _abi.check(int(need), lib)
        if need > 0:
            # padded edge spans; stream-ordered reuse through the caching allocator
            ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
            a.workspace, a.workspace_bytes = ws.data_ptr(), need
            keep.append(ws)
            if a.reserved:  # benchmarking build: scripts read the phase clocks behind the job counter
                global _last_workspace
                _last_workspace = (ws, need)
    return a, out, dev, keep


_last_workspace = None


def split_basis(basis_re, basis_im):
    """(hi, lo) bf16 planes of a basis for ``precision="bf16x3"`` (mispec_split_basis_bf16)."""
    dev = _require_device(basis_re, basis_im)
    wr = _rows(basis_re, "basis_re")
    wi = _rows(basis_im, "basis_im") if basis_im is not None else None
    if wi is not None and (wi.shape != wr.shape or wi.stride(0) != wr.stride(0)):
        raise RuntimeError("real / imaginary bases must have identical shape and layout")
    lib = _abi.load()
    F, K = wr.shape
    need = lib.mispec_basis_split_bytes(F, K, 1 if wi is not None else 0)
    if need < 0:
        _abi.check(int(need))
    dst = torch.empty(need // 2, dtype=torch.int16, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_split_basis_bf16(
            wr.data_ptr(), wi.data_ptr() if wi is not None else None, wr.stride(0), F, K,
            dst.data_ptr(), need, ctypes.c_void_p(stream)))
    return dst


def frag_basis_f32(basis_re, basis_im):
    """The fp32 taps of a complex bank in the strip kernel's fragment order
    (mispec_frag_basis_f32), or None for banks of more than 1024 bins: with it, the supports and
    their host copy, ``precision="fp32"`` contractions of CQT banks run on the strip kernel."""
    dev = _require_device(basis_re, basis_im)
    wr, wi = _rows(basis_re, "basis_re"), _rows(basis_im, "basis_im")
    if wi.shape != wr.shape or wi.stride(0) != wr.stride(0):
        raise RuntimeError("real / imaginary bases must have identical shape and layout")
    lib = _abi.load()
    F, K = wr.shape
    need = lib.mispec_basis_frag_bytes(F, K)
    if need < 0:
        return None
    dst = torch.empty(need // 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_frag_basis_f32(wr.data_ptr(), wi.data_ptr(), wr.stride(0), F, K,
                                             dst.data_ptr(), need, ctypes.c_void_p(stream)))
    return dst


def chain_basis_f32(basis_re, basis_im, support_host):
    """The fp32 taps of a complex bank with supports as the chain kernel consumes them
    (mispec_chain_basis_f32: 16-row x 16-tap MFMA fragments in stream order, zero taps left out), or
    None when the bank's supports do not nest (not a centred CQT bank).  With it, the supports and
    their host copy, ``precision="fp32"`` contractions at hops of 64 .. 512 (multiples of 64) keep
    the frames' samples in LDS delay lines -- the same one-FMA-chain-per-output arithmetic."""
    dev = _require_device(basis_re, basis_im)
    wr, wi = _rows(basis_re, "basis_re"), _rows(basis_im, "basis_im")
    if wi.shape != wr.shape or wi.stride(0) != wr.stride(0):
        raise RuntimeError("real / imaginary bases must have identical shape and layout")
    lib = _abi.load()
    F, K = wr.shape
    sup = np.ascontiguousarray(support_host, dtype=np.int32)
    if sup.shape != (F, 2):
        raise RuntimeError("support_host must be (n_bins, 2)")
    need = lib.mispec_basis_chain_bytes(sup.ctypes.data, F, K)
    if need < 0:
        return None
    dst = torch.empty(need // 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_chain_basis_f32(wr.data_ptr(), wi.data_ptr(), wr.stride(0), F, K, sup.ctypes.data,
                                              dst.data_ptr(), need, ctypes.c_void_p(stream)))
    return dst


def frag_basis_f16(basis_re, basis_im):
    """Scaled (hi, lo) fp16 pairs of a complex bank in the strip kernel's fragment order + the
    per-row inverse scales (mispec_frag_basis_f16), or None for banks of more than 1024 bins: with
    it, the supports and their host copy, ``precision="f16x3"`` contractions of CQT banks run on
    the strip kernel with fp32-class accuracy."""
    dev = _require_device(basis_re, basis_im)
    wr, wi = _rows(basis_re, "basis_re"), _rows(basis_im, "basis_im")
    if wi.shape != wr.shape or wi.stride(0) != wr.stride(0):
        raise RuntimeError("real / imaginary bases must have identical shape and layout")
    lib = _abi.load()
    F, K = wr.shape
    need = lib.mispec_basis_frag16_bytes(F, K)
    if need < 0:
        return None
    dst = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_frag_basis_f16(wr.data_ptr(), wi.data_ptr(), wr.stride(0), F, K,
                                             dst.data_ptr(), need, ctypes.c_void_p(stream)))
    return dst


# what the symmetric fold may neglect, relative to the largest coefficient: one ulp of the fp32
# basis (the pairs of a Fourier basis agree to the last bit or differ by one rounding), far inside
# the split's own 2^-17
FOLD_ASYMMETRY_TOL = 2.0 ** -20


def fold_basis(basis_re, basis_im, precision="bf16x3"):
    """``(planes, folded taps)`` for ``framed_gemm(..., basis_fold=...)`` when the basis is even
    (re) / odd (im) about tap K/2 -- every Fourier basis of the reference's STFT -- else None.
    The symmetry is checked numerically (``mispec_fold_basis_bf16`` / ``_f32`` report the largest
    coefficient the fold would neglect); this reads two scalars back: once per basis (the callers
    cache the result), never per forward.  The planes' format follows ``precision`` (split bf16 or
    fp32 taps, the same size) and must be handed to ``framed_gemm`` with that precision."""
    dev = _require_device(basis_re, basis_im)
    wr = _rows(basis_re.detach(), "basis_re")
    wi = _rows(basis_im.detach(), "basis_im") if basis_im is not None else None
    if wi is None or wi.shape != wr.shape or wi.stride(0) != wr.stride(0):
        return None
    F, K = wr.shape
    if K < 64 or K % 2:
        return None
    lib = _abi.load()
    with_tap0 = int(bool((wr[:, 0] != 0).any().item() or (wi[:, 0] != 0).any().item()))
    taps = lib.mispec_fold_taps(K, with_tap0)
    need = lib.mispec_basis_fold_bytes(F, K, with_tap0)
    if taps < 0 or need < 0:
        return None
    dst = torch.empty(need // 2, dtype=torch.int16, device=dev)
    stats = torch.zeros(2, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        fold = {"bf16x3": lib.mispec_fold_basis_bf16, "f16x3": lib.mispec_fold_basis_f16,
                "fp32": lib.mispec_fold_basis_f32}[resolve_precision(precision)]
        _abi.check(fold(
            wr.data_ptr(), wi.data_ptr(), wr.stride(0), F, K, with_tap0, dst.data_ptr(), need,
            stats.data_ptr(), ctypes.c_void_p(stream)))
    neglected, largest = (float(v) for v in stats.cpu())
    if not largest > 0.0 or neglected > FOLD_ASYMMETRY_TOL * largest:
        return None
    if resolve_precision(precision) == "f16x3" and largest > 2.0:
        return None  # (the fp16 pairs hold coefficient x 2^14)
    return dst, int(taps)


def fold2_basis(basis_re, basis_im, precision):
    """``(planes, max |window|)`` for ``framed_gemm(..., basis_fold2=...)`` when the basis is
    window x DFT with rows = bins 0, 1, 2, .. -- the reference's STFT basis with
    ``freq_scale='no'`` (stft.py:230-245) -- else None.  ``mispec_fold2_basis`` builds the
    quarter-folded planes from the analytic DFT and reports how far the buffers are from
    ``basis_re[0] x dft``; offered only when that is rounding noise.  Reads three scalars back,
    once per basis (the callers cache the result)."""
    dev = _require_device(basis_re, basis_im)
    wr = _rows(basis_re.detach(), "basis_re")
    wi = _rows(basis_im.detach(), "basis_im") if basis_im is not None else None
    if wi is None or wi.shape != wr.shape or wi.stride(0) != wr.stride(0):
        return None
    F, K = wr.shape
    lib = _abi.load()
    if K % 64 or not 128 <= K <= 8192:
        return None  # (the contraction on these planes also wants >= 128 bins; the FFT route takes any number)
    need = lib.mispec_basis_fold2_bytes(F, K)
    if need < 0:
        return None
    dst = torch.empty(need // 2, dtype=torch.int16, device=dev)
    stats = torch.zeros(3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_fold2_basis(
            wr.data_ptr(), wi.data_ptr(), wr.stride(0), F, K, _PRECISIONS[resolve_precision(precision)],
            dst.data_ptr(), need, stats.data_ptr(), ctypes.c_void_p(stream)))
    mismatch, largest, wmax = (float(v) for v in stats.cpu())
    if not largest > 0.0 or not wmax > 0.0 or mismatch > FOLD_ASYMMETRY_TOL * largest:
        return None
    return dst, wmax


def prepare_basis(basis_re, basis_im, precision, hop=None, fold=True):
    """Derived operands of a basis for ``framed_gemm`` in the given arithmetic, as keyword
    arguments (callers with a persistent basis cache this dict, see ``DerivedCache``): for
    "bf16x3" the split-bf16 planes; in any arithmetic, for a window x DFT basis the quarter-folded
    planes (``basis_fold2``) and for a basis with the Fourier symmetry the folded planes
    (``basis_fold``) -- the library uses them when the shape allows (include/mispec.h): a quarter /
    half of the MFMAs.  "f16x3": the folded kernels, else (complex bases of more than 64 bins) the staged
    dense kernel on ``split_basis_f16`` planes; what they do not cover runs in fp32 on the tile kernels."""
    precision = resolve_precision(precision)
    if not basis_re.is_cuda:
        return {}  # (the host path contracts the module's buffers as they are)
    out = {"basis_split": split_basis(basis_re, basis_im)} if precision == "bf16x3" else {}
    # (fold=False: a trainable basis -- every optimiser step would re-fold it, with the symmetry checks'
    # host reads, and drift away from the symmetry anyway)
    if fold and basis_im is not None and (hop is None or 8 * int(hop) >= basis_re.shape[-1]):
        folded2 = fold2_basis(basis_re, basis_im, precision)
        if folded2 is not None:
            out["basis_fold2"] = folded2
        folded = fold_basis(basis_re, basis_im, precision)
        if folded is not None:
            out["basis_fold"] = folded
    if (precision == "f16x3" and basis_im is not None and 2 * basis_re.shape[0] > 128
            and (("basis_fold2" not in out and "basis_fold" not in out) or basis_re.shape[-1] > 8192)):
        # not folded (trainable, freq_scale != 'no', hop < K / 8, ...): scaled fp16 planes for the staged dense kernel
        out["basis_split"] = split_basis_f16(basis_re, basis_im)
    return out


def describe_framed_kernel(precision, prepared):
    """Name of the dominant kernel ``framed_gemm`` launches for a dense complex basis prepared
    with ``prepare_basis`` (bench.py's report)."""
    precision = resolve_precision(precision)
    if prepared.get("basis_fold2") is not None:
        kern, mfma = {"fp32": ("framed_fold32_kernel", "v_mfma_f32_32x32x2_f32"),
                      "bf16x3": ("framed_fold_kernel", "v_mfma_f32_32x32x16_bf16"),
                      "f16x3": ("framed_fold16_kernel", "v_mfma_f32_32x32x16_f16")}[precision]
        return "%s (%s, K/4 folded taps, even / odd bins) + fold2_frames_kernel" % (kern, mfma)
    if precision != "bf16x3":
        if prepared.get("basis_fold") is not None:
            return "framed_fold32_kernel (v_mfma_f32_32x32x2_f32, K/2 folded taps) + fold_frames_kernel"
        return "framed_gemm_kernel<2,2,2,2,framed,rows,unmasked> (v_mfma_f32_32x32x2_f32)"
    if prepared.get("basis_fold") is not None:
        return "framed_fold_kernel (v_mfma_f32_32x32x16_bf16, K/2 folded taps) + fold_frames_kernel"
    return "framed_bf16x3_kernel<4,2,2,4,unmasked> (v_mfma_f32_32x32x16_bf16) + split_signal_kernel"


def framed_gemm(x, basis_re, basis_im, *, reference_kernel=False, **kw):
    """``out[b, row_offset + f, t(, 0:2)]`` <- epilogue(sum_n x_pad[b, t*hop + n] * basis[f, n]).

    Keyword arguments: hop, pad, pad_mode (``PAD_*``), epilogue (``EPI_*``), im_sign, eps, power,
    row_scale, row_support, out (pre-allocated (B, rows_total, T[, 2]) tensor: octave assembly /
    all-gather slices write in place), out_rows_total, out_row_offset, tile, precision ("fp32" /
    "bf16x3" / None = process default), basis_split (cached ``split_basis`` result), fb +
    fb_support (fused filterbank reduction: EPI_POWER with power 1 or 2; the result is
    ``(B, n_filters, T)``, see ``fused_filterbank_ok``)."""
    a, out, dev, _keep = _framed_args(x, basis_re, basis_im, need_workspace=not reference_kernel,
                                      **kw)
    if dev.type == "cpu":
        _abi.check(_abi.load().mispec_framed_gemm_host_f32(ctypes.byref(a)))
        return out
    # `_debug` (scripts/kbench.py, scripts/profile.sh) selects the benchmarking build; the product
    # library rejects a non-zero `reserved`
    lib = _abi.load_ablate() if a.reserved else _abi.load()
    fn = lib.mispec_framed_gemm_f32_ref if reference_kernel else lib.mispec_framed_gemm_f32
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(fn(ctypes.byref(a), ctypes.c_void_p(stream)), lib)
    return out


def filterbank_support(fb):
    """int32 (n_filters, 2): [first, last + 1) column with a non-zero weight per filter row, and
    how many filters cover a bin on average (plumbing for the fused filterbank epilogue; empty
    rows give [0, 0))."""
    nz = fb.detach() != 0
    F = fb.shape[1]
    idx = torch.arange(F, device=fb.device)
    first = torch.where(nz, idx, torch.full_like(idx, F)).min(1)[0]
    last = torch.where(nz, idx + 1, torch.zeros_like(idx)).max(1)[0]
    first = torch.minimum(first, last)
    sup = torch.stack((first, last), 1).to(torch.int32).contiguous()
    # bins walked per output frame, in units of the bin count: 2 for triangular (mel) filters
    coverage = float((last - first).sum().item()) / max(F, 1)
    return sup, coverage


# The fused epilogue walks a filter's band bin by bin (VALU): worth it for banded filterbanks
# (every bin under ~2 triangular mel filters); a dense one (gammatone: every bin under every
# filter) goes through the MFMA filterbank kernel, mispec_filterbank_f32, instead.
FUSED_FB_MAX_COVERAGE = 4.0


def fused_filterbank_ok(power, coverage, n_filters=1):
    """Whether ``framed_gemm(..., fb=...)`` is served (include/mispec.h: power 1 or 2, at most
    256 filters) and worthwhile (banded filters); both precisions have the fused epilogue."""
    return (n_filters <= 256 and float(power) in (1.0, 2.0)
            and 0 < coverage <= FUSED_FB_MAX_COVERAGE)


def fused_filterbank_plan(mod, fb, x, stft, power):
    """For MelSpectrogram / Gammatonegram: the ``fb_support`` table when this forward can run with
    the filterbank fused into the STFT kernel (no graph needed, banded filters: the FFT path reduces
    its LDS tile over the bands, the contraction kernels their accumulator tiles), else None."""
    if needs_grad(mod, x) or stft.freq_bins is not None or compiling() or not x.is_cuda:
        return None
    if not hasattr(mod, "_fb_support"):
        mod._fb_support = DerivedCache()
    sup, coverage = mod._fb_support.get((fb,), lambda: filterbank_support(fb))
    if not fused_filterbank_ok(power, coverage, fb.shape[0]):
        return None
    return sup


def frame_major_filterbank_plan(mod, fb, x, stft):
    """For Gammatonegram (a DENSE filterbank, which the fused reduction of fused_filterbank_plan does not serve): the
    filterbank zero-padded to ``(n_filters, Fp)``, Fp = n_bins rounded up to 32, when this forward can run as
    FFT-route power spectrogram written FRAME-MAJOR ``(B, T, Fp)`` + one framed contraction over the bins of a frame
    (``filterbank_frame_major``) -- no graph needed, device tensors, the FFT route open for this STFT
    (n_fft 1024 / 2048, all bins, frozen Fourier kernels); else None."""
    if needs_grad(mod, x) or stft.freq_bins is not None or compiling() or not x.is_cuda or not fft_enabled():
        return None
    if stft.trainable or stft.n_fft not in (1024, 2048) or fb.shape[1] != stft.n_fft // 2 + 1 or not fb.is_cuda:
        return None
    #  what the library would refuse from inside the two calls -- clips x frames of the FFT route, the int32
    # sample count of the second contraction's (T * Fp)-sample "clips" -- is checked here: the generic two-kernel path
    # serves such calls
    xs = x.reshape(-1, x.shape[-1])
    B, L = xs.shape
    T = (L + 2 * (stft.n_fft // 2 if stft.center else 0) - stft.n_fft) // stft.stride + 1
    Fp = (fb.shape[1] + 31) // 32 * 32
    if T <= 0 or B * T > (1 << 30) or T * Fp + Fp + 64 > 0x7FFFFFFF:
        return None
    if not hasattr(mod, "_fb_padded"):
        mod._fb_padded = DerivedCache()

    def build():
        F = fb.shape[1]
        out = torch.zeros((fb.shape[0], (F + 31) // 32 * 32), dtype=torch.float32, device=fb.device)
        out[:, :F] = fb.detach()
        return out

    return mod._fb_padded.get((fb,), build)


def filterbank_frame_major(fb_padded, spec_fm):
    """``out[b, m, t] = sum_k fb_padded[m, k] spec_fm[b, t, k]``: the frames of ``spec_fm`` (B, T, Fp) are the rows of the
    framed operand (hop = kernel = Fp), fp32 MFMA tile kernel (LDS-direct loads of both operands)."""
    B, T, Fp = spec_fm.shape
    return framed_gemm(spec_fm.view(B, T * Fp), fb_padded, None, hop=Fp, pad=0, pad_mode=PAD_NONE, epilogue=EPI_REAL,
                       precision="fp32")


def framed_gemm_group(problems):
    """Launch several independent framed contractions (the octaves of CQT2010v2 / VQT) as one
    grouped kernel launch; ``problems`` is a list of ``(x, basis_re, basis_im, kwargs)``.
    Falls back to one launch per problem when the library reports the group as unsupported
    (different tile shapes, more than 8 problems)."""
    if not problems:
        return
    lib = _abi.load()
    built = [_framed_args(x, wr, wi, **kw) for (x, wr, wi, kw) in problems]
    dev = built[0][2]
    if any(b[2] != dev for b in built):
        raise RuntimeError("grouped problems must live on one device")
    if dev.type == "cpu":
        for b in built:
            _abi.check(lib.mispec_framed_gemm_host_f32(ctypes.byref(b[0])))
        return
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for i in range(0, len(built), 8):
            chunk = built[i:i + 8]
            arr = (_abi.FramedGemmArgs * len(chunk))(*[b[0] for b in chunk])
            rc = lib.mispec_framed_gemm_group_f32(arr, len(chunk), stream)
            if rc == _abi.E_UNSUPPORTED:
                for b in chunk:
                    _abi.check(lib.mispec_framed_gemm_f32(ctypes.byref(b[0]), stream))
            else:
                _abi.check(rc)


def filterbank(fb, spec):
    """(M, F) x (B, F, T) -> (B, M, T)   [torch.matmul(mel_basis, spec), mel.py:188]."""
    dev = _require_device(fb, spec, host_ok=True)
    fb = _f32(fb, "filterbank").contiguous()
    spec = _f32(spec, "spectrogram").contiguous()
    if fb.dim() != 2 or spec.dim() != 3 or fb.shape[1] != spec.shape[1]:
        raise RuntimeError(
            "mat1 and mat2 shapes cannot be multiplied (%s and %s)"
            % (tuple(fb.shape), tuple(spec.shape))
        )
    B, F, T = spec.shape
    M = fb.shape[0]
    out = alloc_out((B, M, T), dev)
    lib = _abi.load()
    if dev.type == "cpu":
        _abi.check(lib.mispec_filterbank_host_f32(fb.data_ptr(), M, F, spec.data_ptr(), B, T, out.data_ptr()))
        return out
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_filterbank_f32(
            fb.data_ptr(), M, F, spec.data_ptr(), B, T, out.data_ptr(), ctypes.c_void_p(stream)))
    return out


def istft_basis(kernel_cos, kernel_sin, n_freq, onesided):
    """(n_fft, 2*n_freq) contraction basis of the inverse STFT from the module's inverse kernels
    (n_fft, 1, n_fft, 1): [cos | -sin], with the mirrored upper bins of a one-sided spectrum
    (extend_fbins, utils.py:63-70: conjugates of bins F-2 .. 1) folded onto bins 1 .. F-2."""
    C = kernel_cos.reshape(kernel_cos.shape[0], -1)
    S = kernel_sin.reshape(kernel_sin.shape[0], -1)
    N = C.shape[1]
    if not onesided:
        if n_freq != N:
            raise RuntimeError("a two-sided spectrogram needs n_fft = %d frequency bins, got %d" % (N, n_freq))
        return torch.cat((C, -S), 1).contiguous()
    if 2 * n_freq - 2 != N:
        raise RuntimeError(
            "a one-sided spectrogram needs n_fft // 2 + 1 = %d frequency bins, got %d" % (N // 2 + 1, n_freq))
    Ce, Se = C[:, :n_freq].clone(), S[:, :n_freq].clone()
    mirror = torch.arange(N - 1, N - n_freq + 1, -1, device=C.device)  # N-1 .. N-(F-2) <-> bins 1 .. F-2
    Ce[:, 1:n_freq - 1] += C[:, mirror]
    Se[:, 1:n_freq - 1] -= S[:, mirror]
    return torch.cat((Ce, -Se), 1).contiguous()


def istft_basis_is_dft(basis, n_freq):
    """Whether an ``istft_basis`` of a one-sided spectrogram is the plain inverse DFT -- rows n, columns
    [c_k cos(2 pi k n / N) | -c_k sin(2 pi k n / N)], c = 1 for the DC and Nyquist bins, 2 between (what
    ``fourier_basis(freq_bins=n_fft, freq_scale='no')`` with the mirrored bins folded in gives) -- to float32
    rounding of its entries: then ``istft`` synthesises the frames with an inverse FFT (csrc/stft_fft.inl)
    instead of contracting with the basis.  One device-side comparison; callers cache the answer with the basis."""
    N = basis.shape[0]
    if N not in (512, 1024, 2048) or n_freq != N // 2 + 1 or basis.shape[1] != 2 * n_freq or not basis.is_cuda:
        return False
    if basis.requires_grad:
        return False
    n = torch.arange(N, device=basis.device, dtype=torch.float64)[:, None]
    k = torch.arange(n_freq, device=basis.device, dtype=torch.float64)[None, :]
    ang = 2.0 * np.pi * ((n * k) % N) / N
    c = torch.full((1, n_freq), 2.0, device=basis.device, dtype=torch.float64)
    c[0, 0] = c[0, -1] = 1.0
    want = torch.cat((c * torch.cos(ang), -c * torch.sin(ang)), 1)
    return bool((basis.detach().double() - want).abs().max().item() <= 4e-7 * 2.0)


def istft(spec, basis, window, hop, start, out_len, dft=False):
    """Inverse STFT of a (B, F, T, 2) spectrogram with an ``istft_basis``: frame synthesis
    (planar contraction kernel -- or, with ``dft`` = ``istft_basis_is_dft(basis, F)`` and the FFT path
    enabled, an inverse real FFT per frame --, frames stored sample-innermost) + windowed overlap-add with
    window-sum-square normalisation (stft.py:15-63) -> (B, out_len)."""
    dev = _require_device(spec, basis, window, host_ok=True)
    spec = _f32(spec, "spectrogram").contiguous()
    basis = _f32(basis, "basis").contiguous()
    window = _f32(window, "window").reshape(-1).contiguous()
    B, F, T, two = spec.shape
    N = basis.shape[0]
    if two != 2 or basis.shape[1] != 2 * F or window.numel() != N:
        raise RuntimeError("istft: spectrogram %s, basis %s, window %s do not fit together"
                           % (tuple(spec.shape), tuple(basis.shape), tuple(window.shape)))
    out = torch.empty((B, max(int(out_len), 0)), dtype=torch.float32, device=dev)
    if out.numel() == 0:
        return out
    lib = _abi.load()
    if dev.type == "cpu":  # the library's host loops (plumbing-sized inputs)
        _abi.check(lib.mispec_istft_host_f32(spec.data_ptr(), B, F, T, basis.data_ptr(), N, window.data_ptr(),
                                             int(hop), int(start), out.data_ptr(), out.stride(0), out.shape[1]))
        return out
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if dft and fft_enabled() and _istft_fused:
            # one launch: inverse FFT per frame + overlap-add in LDS (the frames never reach HBM); bit-identical
            # to the two launches below, which serve the shapes it refuses
            rc = lib.mispec_istft_fft_f32(spec.data_ptr(), B, F, T, N, window.data_ptr(), int(hop), int(start),
                                          out.data_ptr(), out.stride(0), out.shape[1], stream)
            if rc != _abi.E_UNSUPPORTED:
                _abi.check(rc)
                return out
        frames = torch.empty((B, T, N), dtype=torch.float32, device=dev)
        if dft and fft_enabled():
            _abi.check(lib.mispec_istft_frames_fft_f32(spec.data_ptr(), B, F, T, N, frames.data_ptr(), stream))
        else:
            _abi.check(lib.mispec_istft_frames_f32(spec.data_ptr(), B, F, T, basis.data_ptr(), N,
                                                   frames.data_ptr(), stream))
        _abi.check(lib.mispec_overlap_add_f32(frames.data_ptr(), B, T, N, window.data_ptr(), int(hop),
                                              int(start), out.data_ptr(), out.stride(0), out.shape[1],
                                              stream))
    return out


class _IstftFn(torch.autograd.Function):
    """Autograd wrapper of ``istft``: gradients w.r.t. the spectrogram, the synthesis basis and the
    window (``iSTFT(trainable_window=True)``, stft.py:511-512)."""

    @staticmethod
    def forward(ctx, spec, basis, window, hop, start, out_len, dft=False):
        if not spec.is_cuda:
            raise _gpu_only(spec)  # (the host path is forward-only)
        y = istft(spec, basis, window, hop, start, out_len, dft=dft)
        ctx.save_for_backward(spec, basis, window, y if window.requires_grad else None)
        ctx.meta = (int(hop), int(start), int(out_len))
        return y

    @staticmethod
    def backward(ctx, grad_out):
        spec, basis, window, y = ctx.saved_tensors
        hop, start, out_len = ctx.meta
        dev = spec.device
        spec = _f32(spec.detach(), "spectrogram").contiguous()
        basis = _f32(basis.detach(), "basis").contiguous()
        window = _f32(window.detach(), "window").reshape(-1).contiguous()
        go = _f32(grad_out, "grad_output").contiguous()
        B, F, T, _ = spec.shape
        N = basis.shape[0]
        full = (T - 1) * hop + N
        lib = _abi.load()
        u = torch.empty((B, full), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _abi.check(lib.mispec_istft_grad_signal_f32(go.data_ptr(), go.stride(0), B, T, N,
                                                        window.data_ptr(), hop, start, out_len,
                                                        u.data_ptr(), stream))
        gspec = gbasis = None
        if ctx.needs_input_grad[0]:
            # d X[b, c, t] = sum_n basis[n, c] * win[n] * u[b, t*hop + n]: the forward framed
            # contraction of u with the window-weighted transposed synthesis basis
            bw = (basis * window[:, None]).t().contiguous()  # (2F, N)
            gspec = framed_gemm(u, bw[:F], bw[F:], hop=hop, pad=0, pad_mode=PAD_NONE,
                                epilogue=EPI_COMPLEX, im_sign=1.0, precision="fp32")
        if ctx.needs_input_grad[1]:
            # d basis[n, c] = win[n] * sum_{(b,t)} u[b, t*hop + n] * X[b, c, t]: as in
            # _FramedGemmFn, the framed contraction of the tap-major frame matrix of u with X
            BT = B * T
            ut = torch.empty((N, BT), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.mispec_frames_transpose_f32(u.data_ptr(), full, B, T, hop, N,
                                                           ut.data_ptr(), stream))
            g = spec.permute(3, 1, 0, 2).reshape(2 * F, BT).contiguous()
            dw = framed_gemm(ut.reshape(1, N * BT), g, None, hop=BT, pad=0, pad_mode=PAD_NONE,
                             epilogue=EPI_REAL, im_sign=1.0, precision="fp32")[0]  # (2F, N)
            gbasis = (dw.t() * window[:, None]).contiguous()
        gwin = None
        if ctx.needs_input_grad[2]:
            # y[b, i] = num[b, p] / wss[p], p = i + start:  num = sum_t s[b, t, p - t hop] w[p - t hop] / N,
            # wss = sum_t w[p - t hop]^2 (the division is skipped where wss <= 1e-10, stft.py:41-51), s =
            # the synthesised frames.  d w[n] = sum_{b,t} u[b, t hop + n] s[b, t, n]      (through num; u as above)
            #                                 + 2 w[n] sum_t v[t hop + n],  v[p] = -sum_b g y / wss  (through wss)
            # s comes from the frame-synthesis kernel; the rest are reductions over views (host-side
            # glue on torch ops: this gradient is not on the hot path)
            frames = torch.empty((B, T, N), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.mispec_istft_frames_f32(spec.data_ptr(), B, F, T, basis.data_ptr(), N,
                                                       frames.data_ptr(), stream))
            gwin = (u.unfold(1, N, hop) * frames).sum((0, 1))
            # (overlap-add of the squared window as col2im: conv_transpose1d here went to a MIOpen implicit-GEMM
            # backward-data kernel that faulted on the MI355X for this 1 x 1 x T problem, depending on where the
            # allocator had placed its operands)
            wss = torch.nn.functional.fold((window * window).reshape(1, N, 1).expand(1, N, T).contiguous(),
                                           (1, full), (1, N), stride=(1, hop)).reshape(-1)
            v = torch.zeros(full, dtype=torch.float32, device=dev)
            seg = wss[start:start + out_len]
            v[start:start + out_len] = torch.where(
                seg > 1e-10, -(go * y.to(torch.float32)).sum(0) / seg, torch.zeros_like(seg))
            gwin = gwin + 2.0 * window * v.unfold(0, N, hop).sum(0)
        return gspec, gbasis, gwin, None, None, None, None


def istft_autograd(spec, basis, window, hop, start, out_len, dft=False):
    if torch.is_grad_enabled() and (spec.requires_grad or basis.requires_grad or window.requires_grad):
        return _IstftFn.apply(spec, basis, window, hop, start, out_len, dft)
    return istft(spec, basis, window, hop, start, out_len, dft=dft)


def power_to_db(spec, amin, ref, top_db):
    """librosa-style power_to_db with a per-clip maximum (mel.py:263-279); (B, M, T) -> same."""
    dev = _require_device(spec, host_ok=True)
    spec = _f32(spec, "spectrogram").contiguous()
    if spec.dim() < 2:
        raise RuntimeError("power_to_db expects (batch, ...)")
    if top_db is not None and top_db < 0:
        raise ValueError("top_db must be non-negative")
    B = spec.shape[0]
    out = torch.empty_like(spec)
    lib = _abi.load()
    if dev.type == "cpu":
        _abi.check(lib.mispec_power_to_db_host_f32(spec.data_ptr(), B, spec[0].numel(), float(amin), float(ref),
                                                   -1.0 if top_db is None else float(top_db), out.data_ptr()))
        return out
    ws = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_power_to_db_f32(
            spec.data_ptr(), B, spec[0].numel(), float(amin), float(ref),
            -1.0 if top_db is None else float(top_db), out.data_ptr(), ws.data_ptr(), B * 4,
            ctypes.c_void_p(stream)))
    return out


_mfcc_fused = True


def set_mfcc_fused(enabled):
    """MFCC's tail (power_to_db + DCT) as ONE launch where no gradient is recorded (``mispec_mfcc_tail_f32``);
    ``False`` keeps the two calls.  Returns the previous setting."""
    global _mfcc_fused
    old, _mfcc_fused = _mfcc_fused, bool(enabled)
    return old


def mfcc_tail(mel, amin, ref, top_db, dct):
    """``power_to_db`` followed by the DCT (mel.py:263-307) in one launch: (B, n_mels, T) -> (B, n_mfcc, T), or
    None when the library does not serve the shape (the caller then makes the two calls)."""
    if not _mfcc_fused or not mel.is_cuda or mel.dim() != 3:
        return None
    if top_db is not None and top_db < 0:
        raise ValueError("top_db must be non-negative")
    mel = _f32(mel, "spectrogram").contiguous()
    dct = _f32(dct, "dct").contiguous()
    B, M, T = mel.shape
    K = dct.shape[0]
    if dct.dim() != 2 or dct.shape[1] != M or dct.device != mel.device:
        return None
    out = torch.empty((B, K, T), dtype=torch.float32, device=mel.device)
    lib = _abi.load()
    with torch.cuda.device(mel.device):
        stream = torch.cuda.current_stream(mel.device).cuda_stream
        rc = lib.mispec_mfcc_tail_f32(mel.data_ptr(), B, M, T, float(amin), float(ref),
                                      -1.0 if top_db is None else float(top_db), dct.data_ptr(), K, out.data_ptr(),
                                      ctypes.c_void_p(stream))
    if rc == _abi.E_UNSUPPORTED:
        return None
    _abi.check(rc)
    return out


class _PowerToDbFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, amin, ref, top_db):
        ctx.save_for_backward(spec)
        ctx.meta = (float(amin), top_db)
        return power_to_db(spec, amin, ref, top_db)

    @staticmethod
    def backward(ctx, grad_out):
        (spec,) = ctx.saved_tensors
        spec = _f32(spec.detach(), "spectrogram").contiguous()
        go = _f32(grad_out, "grad_output").contiguous()
        amin, top_db = ctx.meta
        B = spec.shape[0]
        gs = torch.empty_like(spec)
        ws = torch.empty(2 * B, dtype=torch.int32, device=spec.device)
        with torch.cuda.device(spec.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(spec.device).cuda_stream)
            _abi.check(_abi.load().mispec_power_to_db_bwd_f32(
                spec.data_ptr(), go.data_ptr(), B, spec[0].numel(), amin,
                -1.0 if top_db is None else float(top_db), gs.data_ptr(), ws.data_ptr(), 8 * B, stream))
        return gs, None, None, None


def power_to_db_autograd(spec, amin, ref, top_db):
    if torch.is_grad_enabled() and spec.requires_grad:
        if top_db is not None and top_db < 0:
            raise ValueError("top_db must be non-negative")
        return _PowerToDbFn.apply(spec, amin, ref, top_db)
    if compiling():
        from . import ops

        return ops.power_to_db(spec, float(amin), float(ref), -1.0 if top_db is None else float(top_db))
    return power_to_db(spec, amin, ref, top_db)


def fir_decimate(x, taps, stride):
    """conv1d(x, taps, stride=stride, padding=(len-1)//2)  [utils.py:73-124] -> (B, n_out)."""
    dev = _require_device(x, taps, host_ok=True)
    x = _signal(x)
    taps = _f32(taps, "filter").reshape(-1).contiguous()
    B, L = x.shape
    nt = taps.numel()
    pad = (nt - 1) // 2
    n_out = decimated_length(L, nt, int(stride))
    if n_out <= 0:
        raise RuntimeError(
            "Calculated padded input size per channel: (%d). Kernel size: (%d). "
            "Kernel size can't be greater than actual input size" % (L + 2 * pad, nt)
        )
    y = torch.empty((B, n_out), dtype=torch.float32, device=dev)
    lib = _abi.load()
    if dev.type == "cpu":
        _abi.check(lib.mispec_fir_decimate_host_f32(x.data_ptr(), x.stride(0), B, L, taps.data_ptr(), nt,
                                                    int(stride), pad, y.data_ptr(), y.stride(0), n_out))
        return y
    need = lib.mispec_fir_decimate_workspace_bytes(B, L, nt, int(stride), pad, n_out)
    if need < 0:
        _abi.check(int(need))
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev) if need > 0 else None
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_fir_decimate_f32(
            x.data_ptr(), x.stride(0), B, L, taps.data_ptr(), nt, int(stride), pad,
            y.data_ptr(), y.stride(0), n_out, ws.data_ptr() if ws is not None else None, need,
            ctypes.c_void_p(stream)))
    return y


def split_basis_f16(basis_re, basis_im):
    """Scaled (hi, lo) fp16 pairs of a complex bank as row-major planes + the per-row inverse scales
    (mispec_split_basis_f16): the banks of the fused octave kernel in ``precision="f16x3"``, and as
    ``basis_split`` of a ``precision="f16x3"`` ``framed_gemm`` the operand of the staged dense kernel
    (taps contracted in their natural order; CQT1992v2, unfolded STFT bases)."""
    dev = _require_device(basis_re, basis_im)
    wr, wi = _rows(basis_re, "basis_re"), _rows(basis_im, "basis_im")
    if wi.shape != wr.shape or wi.stride(0) != wr.stride(0):
        raise RuntimeError("real / imaginary bases must have identical shape and layout")
    lib = _abi.load()
    F, K = wr.shape
    need = lib.mispec_basis_split16_bytes(F, K)
    if need < 0:
        _abi.check(int(need))
    dst = torch.empty((need + 3) // 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _abi.check(lib.mispec_split_basis_f16(wr.data_ptr(), wi.data_ptr(), wr.stride(0), F, K,
                                              dst.data_ptr(), need, ctypes.c_void_p(stream)))
    return dst


def octave_pyramid(x, levels, *, hop, n_frames, taps, epilogue, im_sign, eps, out, x_last, _stamps=None,
                   precision="bf16x3", fir_headroom_bits=0, absmax_in=None, absmax_in_ready=False,
                   absmax_out=None):
    """One launch of the fused octave recursion (``mispec_octave_pyramid_f32``): ``levels`` is a
    list of up to three dicts ``{split, n_bins, kernel, row_offset, pad_mode, row_scale}`` or None
    (a level without a bank).  ``precision="f16x3"``: the splits are ``split_basis_f16`` results,
    ``absmax_in`` / ``absmax_out`` int32 tensors of ``32 * n_clips`` words (see include/mispec.h).
    Returns False when the library does not serve the shape."""
    dev = _require_device(x, out, x_last, taps)
    x = _signal(x)
    taps = _f32(taps, "filter").reshape(-1).contiguous()
    a = _abi.OctaveArgs()
    a.struct_size = ctypes.sizeof(_abi.OctaveArgs)
    a.n_levels = len(levels)
    a.x, a.x_clip_stride = x.data_ptr(), x.stride(0)
    a.n_clips, a.n_samples = x.shape
    a.hop, a.n_frames = int(hop), int(n_frames)
    a.taps, a.n_taps = taps.data_ptr(), taps.numel()
    a.epilogue, a.im_sign, a.eps = int(epilogue), float(im_sign), float(eps)
    keep = []
    for i, lv in enumerate(levels):
        if lv is None:
            continue
        o = a.level[i]
        sp = lv["split"]
        o.bank_split, o.bank_split_bytes = sp.data_ptr(), sp.numel() * sp.element_size()
        o.n_bins, o.kernel = int(lv["n_bins"]), int(lv["kernel"])
        o.out_row_offset, o.pad_mode = int(lv["row_offset"]), int(lv["pad_mode"])
        rs = lv.get("row_scale")
        if rs is not None:
            rs = _f32(rs, "row_scale").contiguous()
            o.row_scale = rs.data_ptr()
            keep.append(rs)
    if x_last is not None:
        a.x_last, a.x_last_clip_stride = x_last.data_ptr(), x_last.stride(0)
    a.out = out.data_ptr()
    a.out_clip_stride, a.out_row_stride = out.stride(0), out.stride(1)
    a.precision = _PRECISIONS[precision]
    if precision == "f16x3":
        a.fir_headroom_bits = int(fir_headroom_bits)
        a.absmax_in, a.absmax_in_ready = absmax_in.data_ptr(), int(bool(absmax_in_ready))
        a.absmax_out = absmax_out.data_ptr() if absmax_out is not None else None
    lib = _abi.load()
    if _stamps is not None:  # phase clock of one workgroup (scripts/kbench.py, benchmarking build)
        a.reserved = _stamps.data_ptr()
        lib = _abi.load_ablate()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.mispec_octave_pyramid_f32(ctypes.byref(a), ctypes.c_void_p(stream))
    if rc == _abi.E_UNSUPPORTED:
        return False
    _abi.check(rc)
    return True


def octave_stream(x, levels, *, hop, n_frames, taps, epilogue, im_sign, eps, out, x_last, precision="f16x3",
                  fir_headroom_bits=0, n_segments=0, _debug=None):
    """One launch of the STREAMING octave recursion (``mispec_octave_stream_f32``): ``levels`` is a list of up
    to five dicts ``{split, n_bins, kernel, row_offset, pad_mode, row_scale}`` or None (a level without a
    bank; at most four banks).  Returns False when the library does not serve the shape (the caller then
    runs ``octave_pyramid``)."""
    dev = _require_device(x, out, x_last, taps)
    x = _signal(x)
    taps = _f32(taps, "filter").reshape(-1).contiguous()
    a = _abi.OctaveStreamArgs()
    a.struct_size = ctypes.sizeof(_abi.OctaveStreamArgs)
    a.n_levels = len(levels)
    a.x, a.x_clip_stride = x.data_ptr(), x.stride(0)
    a.n_clips, a.n_samples = x.shape
    a.hop, a.n_frames = int(hop), int(n_frames)
    a.taps, a.n_taps = taps.data_ptr(), taps.numel()
    a.epilogue, a.im_sign, a.eps = int(epilogue), float(im_sign), float(eps)
    keep = []
    for i, lv in enumerate(levels):
        if lv is None:
            continue
        o = a.level[i]
        sp = lv["split"]
        o.bank_split, o.bank_split_bytes = sp.data_ptr(), sp.numel() * sp.element_size()
        o.n_bins, o.kernel = int(lv["n_bins"]), int(lv["kernel"])
        o.out_row_offset, o.pad_mode = int(lv["row_offset"]), int(lv["pad_mode"])
        rs = lv.get("row_scale")
        if rs is not None:
            rs = _f32(rs, "row_scale").contiguous()
            o.row_scale = rs.data_ptr()
            keep.append(rs)
    if x_last is not None:
        a.x_last, a.x_last_clip_stride = x_last.data_ptr(), x_last.stride(0)
    a.out = out.data_ptr()
    a.out_clip_stride, a.out_row_stride = out.stride(0), out.stride(1)
    a.precision = _PRECISIONS[precision]
    a.fir_headroom_bits = int(fir_headroom_bits) if precision == "f16x3" else 0
    a.n_segments = int(n_segments)
    lib = _abi.load()
    if _debug is not None:  # ablation bits / phase clock (scripts/stream_prof.py, benchmarking build)
        a.reserved = int(_debug)
        lib = _abi.load_ablate()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.mispec_octave_stream_f32(ctypes.byref(a), ctypes.c_void_p(stream))
    if rc == _abi.E_UNSUPPORTED:
        return False
    _abi.check(rc)
    return True


def pad_mode_id(name):
    return _PAD_MODES[name]


# ---------------------------------------------------------------------------------------
# backward of the framed contraction (include/mispec.h "Backward of ...")
# ---------------------------------------------------------------------------------------
def contract_planar(a, x, *, k, n_clips, n_cols, x_clip_stride, x_k_stride=0, x_col_stride=1,
                    k_split=0, k_split_off=0, k_offsets=None, out=None, rows_inner=False):
    """out[c, m, j] = sum_k a[m, k] * X(c, k, j) (mispec_contract_planar_f32); ``a`` is (m, k)."""
    dev = _require_device(a, x, k_offsets, out)
    a = _f32(a, "a")
    if a.dim() != 2 or a.stride(1) != 1 or a.shape[1] != k:
        raise RuntimeError("a must be (m, k) with unit inner stride")
    m = a.shape[0]
    if out is None:
        shape = (n_clips, n_cols, m) if rows_inner else (n_clips, m, n_cols)
        out = torch.empty(shape, dtype=torch.float32, device=dev)
    pa = _abi.PlanarArgs()
    pa.struct_size = ctypes.sizeof(_abi.PlanarArgs)
    pa.rows_inner = 1 if rows_inner else 0
    pa.a, pa.a_row_stride, pa.m, pa.k = a.data_ptr(), a.stride(0), m, int(k)
    pa.x, pa.x_clip_stride, pa.x_k_stride = x.data_ptr(), int(x_clip_stride), int(x_k_stride)
    pa.x_col_stride, pa.k_split, pa.k_split_off = int(x_col_stride), int(k_split), int(k_split_off)
    pa.k_offsets = k_offsets.data_ptr() if k_offsets is not None else None
    pa.n_clips, pa.n_cols = int(n_clips), int(n_cols)
    pa.out, pa.out_clip_stride = out.data_ptr(), out.stride(0)
    pa.out_row_stride = n_cols
    pa.out_col_stride = m
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _abi.check(_abi.load().mispec_contract_planar_f32(ctypes.byref(pa), stream))
    return out


def epilogue_fwd(z, epilogue, *, eps=0.0, power=2.0):
    """The contraction kernels' pointwise epilogue as a pass of its own (mispec_framed_epilogue_fwd_f32):
    ``z`` (B, F, T, 2) = a Complex output -> (B, F, T[, 2]) in ``epilogue``."""
    dev = _require_device(z)
    if z.dim() != 4 or z.shape[-1] != 2 or not z.is_contiguous() or z.dtype != torch.float32:
        raise RuntimeError("z must be a contiguous float32 (B, F, T, 2) tensor")
    B, F, T, _ = z.shape
    two = int(epilogue) in (EPI_COMPLEX, EPI_PHASE_COSSIN)
    out = torch.empty((B, F, T, 2) if two else (B, F, T), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _abi.check(_abi.load().mispec_framed_epilogue_fwd_f32(z.data_ptr(), B, F, T, int(epilogue), float(eps),
                                                              float(power), out.data_ptr(), stream))
    return out


class _FramedGemmFn(torch.autograd.Function):
    """Autograd wrapper of ``framed_gemm`` for trainable bases / differentiable inputs."""

    @staticmethod
    def forward(ctx, x, basis_re, basis_im, kw):
        if not x.is_cuda:
            raise _gpu_only(x)  # (the host path is forward-only)
        ctx.kw = dict(kw)
        # The backward needs the Complex values z of this very contraction.  With HBM to spare they are kept instead of
        # recomputed (round 6: the recomputation was 1.3 of the 6.1 ms of a cfg2-sized training step): the contraction runs
        # ONCE with the Complex epilogue and the module's output is the pointwise epilogue of z
        # (mispec_framed_epilogue_fwd_f32: the code the contraction kernels end with -- the same bits as the fused launch).
        # Not for bf16x3 (its backward recomputes z in fp32), in-place row blocks, z beyond MISPEC_SAVE_Z_MAX_BYTES, or frozen
        # Fourier bases with the quarter fold (differentiable input only: their forward may run on the FFT kernel, whose fused
        # Magnitude is not the pointwise pass's last bit -- and recomputing z there costs a tenth of a contraction).
        z = None
        epi = int(kw["epilogue"])
        plain = (kw.get("out") is None and kw.get("out_rows_total") is None and not kw.get("out_row_offset")
                 and kw.get("basis_fold2") is None)
        if plain and resolve_precision(kw.get("precision")) != "bf16x3":
            xs = _signal(x.detach())
            F = basis_re.shape[0]
            T = n_frames(xs.shape[1], basis_re.shape[-1], int(kw["hop"]), int(kw["pad"]))
            if 8 * xs.shape[0] * F * T <= int(os.environ.get("MISPEC_SAVE_Z_MAX_BYTES", str(8 << 30))):
                z = framed_gemm(x, basis_re, basis_im, **dict(kw, epilogue=EPI_COMPLEX))
        if z is None:
            ctx.save_for_backward(x, basis_re, basis_im)
            return framed_gemm(x, basis_re, basis_im, **kw)
        ctx.save_for_backward(x, basis_re, basis_im, z)
        if epi == EPI_COMPLEX:
            return z.clone()  # (the caller may write into its output; z is the backward's)
        return epilogue_fwd(z, epi, eps=float(kw.get("eps", 0.0)), power=float(kw.get("power", 2.0)))

    @staticmethod
    def backward(ctx, grad_out):
        x, basis_re, basis_im = ctx.saved_tensors[:3]
        z_saved = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        kw = ctx.kw
        lib = _abi.load()
        dev = x.device
        xs = _signal(x.detach())
        wr = _rows(basis_re.detach(), "basis_re")
        wi = _rows(basis_im.detach(), "basis_im")
        B, L = xs.shape
        F, K = wr.shape
        hop, pad, pad_mode = int(kw["hop"]), int(kw["pad"]), int(kw["pad_mode"])
        scale = kw.get("row_scale")
        im_sign = float(kw.get("im_sign", -1.0))
        # (u, v): the Complex epilogue of the same contraction IN THE FORWARD'S OWN ARITHMETIC, on its own derived
        # operands (round 5: recomputed in fp32 it was 4.7 of the 11 ms of a cfg2-sized training step whose forward --
        # f16x3, the STFT module's default -- takes 1.25; the forward's values are what the loss saw)
        # (measured against a float64 evaluation, scripts/grad_vs_float64.py: d wcos of a Magnitude STFT -- ill-conditioned where
        # a bin's |z| crosses zero -- 6.8e-4 of its maximum with the fp32 recomputation, 4.9e-4 with f16x3; bf16x3's 5e-6 of the
        # peak in z becomes 4e-2 there, so a bf16x3 forward keeps the fp32 recomputation)
        zkw = dict(kw, epilogue=EPI_COMPLEX, out=None, out_rows_total=None, out_row_offset=0)
        if resolve_precision(kw.get("precision")) == "bf16x3":
            zkw.update(precision="fp32", row_support=None)
            for k in ("basis_split", "basis_fold", "basis_fold2"):  # (planes in the forward's precision, not in fp32)
                zkw.pop(k, None)
        z = z_saved if z_saved is not None else framed_gemm(xs, wr, wi, **zkw)
        T = z.shape[2]
        go = _f32(grad_out, "grad_output").contiguous()
        need_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        need_x = ctx.needs_input_grad[0]
        # G = (dL/dacc_re, dL/dacc_im): bin-major (2, F, B, T) for d basis, frame-major
        # (B, T, 2F) for d signal -- the layouts in which each is the "basis" / the "signal" of
        # another framed contraction on the MFMA kernel
        g = torch.empty((2, F, B, T), dtype=torch.float32, device=dev) if need_w else None
        gt = torch.empty((B, T, 2 * F), dtype=torch.float32, device=dev) if need_x else None
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _abi.check(lib.mispec_framed_epilogue_bwd_f32(
                go.data_ptr(), z.data_ptr(), B, F, T, int(kw["epilogue"]), float(kw.get("eps", 0.0)),
                float(kw.get("power", 2.0)), im_sign,
                scale.data_ptr() if scale is not None else None,
                g.data_ptr() if g is not None else None, gt.data_ptr() if gt is not None else None,
                stream))
        del z
        gx = gre = gim = None
        Lp = L + 2 * pad
        BT = B * T
        if need_w:
            # d basis[row, n] = sum_{(b,t)} g[row, (b,t)] * xp[b, t*hop + n]: with the frame matrix
            # stored tap-major (xt[n, (b,t)]) this is the framed contraction of the "signal" xt
            # (frame n = its row n: hop = kernel = b*T) with the "basis" g -- over chunks of clips
            # whose tap-major frame matrix (K x b*T floats) stays below 2 GB and inside the int32
            # sizes of the argument block (CQT1992v2 cfg4: K = 16384, 862 frames -> 64 clips/chunk)
            xp = torch.empty((B, Lp), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _abi.check(lib.mispec_pad_signal_f32(xs.data_ptr(), xs.stride(0), B, L, pad, pad_mode,
                                                     xp.data_ptr(), stream))
            per = max(1, min(B, (2 ** 29) // max(K * T, 1)))
            if K * T >= 2 ** 31:
                raise RuntimeError(
                    "backward of the framed contraction: one clip's frame matrix (%d taps x %d frames) "
                    "exceeds the kernel's int32 range; shorten the clips" % (K, T))
            use16 = resolve_precision(kw.get("precision")) == "f16x3" and 2 * F > 128
            # The d-basis contraction has (2F x K) outputs and B*T taps: 8 x 8 tiles of the staged dense kernel for an STFT
            # of 2048 -- a quarter of the device.  Its sum over the clips is split into up to four chunks of clips whose
            # contractions run side by side on side streams (split-K by hand: 64 workgroups each), added in chunk order.
            n_par = 1
            if use16 and B >= 8 and not torch.cuda.is_current_stream_capturing():
                #  the side streams keep n_par chunks alive at once -- frame matrix, G, its fp16 planes, the
                # workspace, in per-stream allocator pools the main stream cannot reuse: only when all of them together
                # stay inside the budget one serial chunk had (2^29 floats of frames); else the serial loop
                # (eight chunks on the four queues: 4.57 ms per cfg2-sized step against 4.83 with four -- a chunk's 272
                # long-lived workgroups fill half of the device's 512 slots, shorter ones pack better)
                n_chunks = max(1, min(B, int(os.environ.get("MISPEC_DBASIS_CHUNKS", "8"))))
                want = (B + n_chunks - 1) // n_chunks
                if 4 * want * K * T <= (1 << 29):
                    n_par = 4
                    per = min(per, want)

            def d_basis_chunk(b0, b1):
                nb = (b1 - b0) * T
                xt = torch.empty((K, nb), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                    _abi.check(lib.mispec_frames_transpose_f32(xp[b0:b1].data_ptr(), Lp, b1 - b0, T, hop, K,
                                                               xt.data_ptr(), st))
                gc = g[:, :, b0:b1].reshape(2 * F, nb)  # (a view: rows B*T apart, the chunk's (b, t) contiguous)
                if not (use16 and nb % 2 == 0):
                    gc = gc.contiguous()
                if use16 and nb % 2 == 0:
                    # the forward's arithmetic for its adjoint too (round 5): (g_re, g_im) as the complex "basis" of the
                    # staged dense f16x3 kernel (scaled fp16 pairs per row, split here: G changes every step), Complex
                    # epilogue = (d basis_re, d basis_im) side by side
                    g0, g1 = gc[:F], gc[F:]
                    pc = framed_gemm(xt.reshape(1, K * nb), g0, g1, hop=nb, pad=0, pad_mode=PAD_NONE,
                                     epilogue=EPI_COMPLEX, im_sign=1.0, precision="f16x3",
                                     basis_split=split_basis_f16(g0, g1))[0]  # (F, K, 2)
                    return torch.cat((pc[..., 0], pc[..., 1]), 0)
                return framed_gemm(xt.reshape(1, K * nb), gc, None, hop=nb, pad=0,
                                   pad_mode=PAD_NONE, epilogue=EPI_REAL, im_sign=1.0, precision="fp32")[0]

            chunks = [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]
            dw = None
            if n_par > 1 and len(chunks) > 1:
                cur = torch.cuda.current_stream(dev)
                start = torch.cuda.Event()
                start.record(cur)
                done = []
                # n_par - 1 side streams + the current one: the device runs four hardware queues, and a fifth stream
                # shares one of them (round 6's trace: three chunks side by side, the fourth after them)
                done = [None] * len(chunks)
                for i, (b0, b1) in enumerate(chunks):
                    if i % n_par == 0:
                        continue
                    side = _side_stream(dev, i % n_par - 1)
                    side.wait_event(start)
                    with torch.cuda.stream(side):
                        part = d_basis_chunk(b0, b1)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    done[i] = (part, ev)
                for i, (b0, b1) in enumerate(chunks):
                    if i % n_par == 0:
                        done[i] = (d_basis_chunk(b0, b1), None)
                for part, ev in done:
                    if ev is not None:
                        cur.wait_event(ev)
                        part.record_stream(cur)
                    dw = part if dw is None else dw.add_(part)
            else:
                for b0, b1 in chunks:
                    part = d_basis_chunk(b0, b1)
                    dw = part if dw is None else dw.add_(part)
            gre = dw[:F].reshape(basis_re.shape)
            gim = dw[F:].reshape(basis_im.shape)
            del xp
        if need_x:
            # d frames[(b,t), n] = sum_f g_re*w_re[f,n] + g_im*w_im[f,n]: the framed contraction of
            # the "signal" gt (frame (b,t) = its row: hop = kernel = 2F) with the "basis"
            # [w_re^T | w_im^T]; the result comes out tap-major (K, B*T)
            bt = torch.cat((wr.t(), wi.t()), 1).contiguous()  # (K, 2F)
            ft = framed_gemm(gt.reshape(1, BT * 2 * F), bt, None, hop=2 * F, pad=0, pad_mode=PAD_NONE,
                             epilogue=EPI_REAL, im_sign=1.0, precision="fp32")[0]  # (K, BT)
            dxp = torch.empty((B, Lp), dtype=torch.float32, device=dev)
            dx = torch.empty((B, L), dtype=torch.float32, device=dev)
            cover = (T - 1) * hop + K  # samples the frames reach; the rest of the padded signal gets 0
            with torch.cuda.device(dev):
                if cover < Lp:
                    dxp.zero_()
                _abi.check(lib.mispec_overlap_add_f32(ft.data_ptr(), B, T, K, None, hop, -1,
                                                      dxp.data_ptr(), dxp.stride(0), min(cover, Lp),
                                                      stream))
                _abi.check(lib.mispec_unpad_adjoint_f32(dxp.data_ptr(), B, L, pad, pad_mode,
                                                        dx.data_ptr(), dx.stride(0), stream))
            gx = dx.reshape(x.shape)
        return gx, gre, gim, None


class _FilterbankFn(torch.autograd.Function):
    """Autograd wrapper of ``filterbank`` (mel / gammatone reduction, mel.py:188)."""

    @staticmethod
    def forward(ctx, fb, spec):
        if not spec.is_cuda:
            raise _gpu_only(spec)  # (the host path is forward-only)
        ctx.save_for_backward(fb, spec)
        return filterbank(fb, spec)

    @staticmethod
    def backward(ctx, grad_out):
        fb, spec = ctx.saved_tensors
        fb, spec = fb.detach(), spec.detach().contiguous()
        go = _f32(grad_out, "grad_output").contiguous()
        B, F, T = spec.shape
        M = fb.shape[0]
        gfb = gspec = None
        if ctx.needs_input_grad[1]:
            gspec = filterbank(fb.t().contiguous(), go)  # fb^T (F, M) x (B, M, T)
        if ctx.needs_input_grad[0]:
            # d fb[m, f] = sum_{(b,t)} g[b, m, t] * spec[b, f, t]
            dev = spec.device
            koff = torch.empty(B * T, dtype=torch.int64, device=dev)
            with torch.cuda.device(dev):
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                _abi.check(_abi.load().mispec_frame_offsets_i64(koff.data_ptr(), B, T, F * T, 1, stream))
            gm = go.permute(1, 0, 2).reshape(M, B * T).contiguous()
            gfb = contract_planar(gm, spec, k=B * T, n_clips=1, n_cols=F, x_clip_stride=0,
                                  x_col_stride=T, k_offsets=koff)[0].reshape(fb.shape)
        return gfb, gspec


def filterbank_autograd(fb, spec):
    if torch.is_grad_enabled() and (fb.requires_grad or spec.requires_grad):
        return _FilterbankFn.apply(fb, spec)
    if compiling():
        from . import ops

        return ops.filterbank(fb, spec)
    return filterbank(fb, spec)


class _FirDecimateFn(torch.autograd.Function):
    """Autograd wrapper of ``fir_decimate`` (gradient w.r.t. the signal; the anti-alias taps are
    fixed buffers in the reference, cqt.py:947-954)."""

    @staticmethod
    def forward(ctx, x, taps, stride):
        if not x.is_cuda:
            raise _gpu_only(x)  # (the host path is forward-only)
        ctx.save_for_backward(taps)
        ctx.stride, ctx.shape = int(stride), tuple(x.shape)
        return fir_decimate(x, taps, stride)

    @staticmethod
    def backward(ctx, grad_out):
        (taps,) = ctx.saved_tensors
        taps = _f32(taps.detach(), "filter").reshape(-1).contiguous()
        go = _f32(grad_out, "grad_output").contiguous()
        B, n_out = go.shape
        L = ctx.shape[-1]
        nt = taps.numel()
        dx = torch.empty((B, L), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            stream = ctypes.c_void_p(torch.cuda.current_stream(go.device).cuda_stream)
            _abi.check(_abi.load().mispec_fir_decimate_bwd_f32(
                go.data_ptr(), go.stride(0), B, n_out, taps.data_ptr(), nt, ctx.stride, (nt - 1) // 2,
                dx.data_ptr(), dx.stride(0), L, stream))
        return dx.reshape(ctx.shape), None, None


def fir_decimate_autograd(x, taps, stride):
    if torch.is_grad_enabled() and x.requires_grad:
        return _FirDecimateFn.apply(x, taps, stride)
    if compiling():
        from . import ops

        return ops.fir_decimate(x, taps, int(stride))
    return fir_decimate(x, taps, stride)


def _module_tensors(module):
    """Parameters AND plain tensor attributes of a module tree.  nn.DataParallel replicas hold
    their (trainable) parameters as plain attributes -- ``replica.parameters()`` is empty -- so
    ``module.parameters()`` alone would report a trainable replica as frozen."""
    for m in module.modules():
        for t in m._parameters.values():
            if t is not None:
                yield t
        for t in m.__dict__.values():
            if isinstance(t, torch.Tensor):
                yield t


def invalidate_caches(module):
    """Drop every derived operand (split / folded planes, kernel supports, effective kernels ...)
    cached on ``module`` and its children; the next forward rebuilds them.  Needed only after edits
    that bypass autograd's version counters (``tensor.data`` arithmetic)."""
    for m in module.modules():
        for v in list(m.__dict__.values()):
            for c in (v if isinstance(v, (list, tuple)) else (v,)):
                if isinstance(c, DerivedCache):
                    c.clear()
                elif hasattr(c, "__dict__"):
                    for cc in list(vars(c).values()):
                        for ccc in (cc.values() if isinstance(cc, dict) else (cc,)):
                            if isinstance(ccc, DerivedCache):
                                ccc.clear()


def needs_grad(module, x):
    """Whether this forward has to record a graph: grad mode is on and the input or any tensor
    the module computes with (parameters, or the parameter copies of a DataParallel replica)
    requires gradients."""
    return torch.is_grad_enabled() and (
        x.requires_grad or any(t.requires_grad for t in _module_tensors(module)))


def compiling():
    """Under torch.compile the forward goes through the custom ops of ``nnaudio_amd.ops`` (opaque
    to the tracer) and the host-side caches are consulted inside them, at run time."""
    return torch.compiler.is_compiling()


def framed_gemm_autograd(x, basis_re, basis_im, **kw):
    """``framed_gemm`` that records a graph when the input or the bases require gradients.
    ``support`` (CQT1992v2 under torch.compile: look the row supports up inside the custom op) is
    consumed here: neither ``framed_gemm`` nor the autograd function knows it."""
    support = bool(kw.pop("support", False))
    if torch.is_grad_enabled() and (x.requires_grad or basis_re.requires_grad
                                    or (basis_im is not None and basis_im.requires_grad)):
        if basis_im is None:
            raise NotImplementedError("backward of a real contraction is not implemented")
        return _FramedGemmFn.apply(x, basis_re, basis_im, kw)
    if compiling():
        from . import ops

        return ops.framed_gemm(
            x, basis_re, basis_im, int(kw["hop"]), int(kw["pad"]), int(kw["pad_mode"]),
            int(kw["epilogue"]), float(kw.get("im_sign", -1.0)), float(kw.get("eps", 0.0)),
            float(kw.get("power", 2.0)), kw.get("row_scale"), support,
            resolve_precision(kw.get("precision")))
    return framed_gemm(x, basis_re, basis_im, **kw)
