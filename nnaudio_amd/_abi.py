"""ctypes binding of ``libmispec.so`` (C ABI declared in ``include/mispec.h``).

There is deliberately no fallback: if the shared library is missing or does not export
the expected symbols, importing the compute path raises.  The library is built in-tree
by ``__graft_entry__.build()`` / ``python -m nnaudio_amd.build``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmispec.so")
ABLATE_LIB_PATH = os.path.join(_HERE, "csrc", "libmispec_ablate.so")  # benchmarking build

ABI_VERSION = 13
E_INVALID, E_UNSUPPORTED, E_HIP = -1, -2, -3

# enums (mirror include/mispec.h)
PAD_NONE, PAD_ZERO, PAD_REFLECT = 0, 1, 2
EPI_COMPLEX, EPI_MAGNITUDE, EPI_POWER, EPI_PHASE_ATAN2, EPI_PHASE_COSSIN, EPI_REAL = range(6)
(TILE_AUTO, TILE_128x128, TILE_32x256, TILE_64x256, TILE_128x128_TALL, TILE_192x128, TILE_256x128,
 TILE_256x128_SQ, TILE_128x256_SQ, TILE_256x256) = range(10)
PREC_F32, PREC_BF16X3, PREC_F16X3 = 0, 1, 2

EXPORTS = (
    "mispec_octave_stream_f32",
    "mispec_octave_stream_plan_of",
    "mispec_version",
    "mispec_last_error",
    "mispec_framed_gemm_f32",
    "mispec_framed_gemm_f32_ref",
    "mispec_framed_gemm_group_f32",
    "mispec_framed_gemm_workspace_bytes",
    "mispec_strip_plan",
    "mispec_basis_split_bytes",
    "mispec_split_basis_bf16",
    "mispec_basis_frag_bytes",
    "mispec_frag_basis_f32",
    "mispec_basis_chain_bytes",
    "mispec_chain_basis_f32",
    "mispec_basis_frag16_bytes",
    "mispec_basis_split16_bytes",
    "mispec_split_basis_f16",
    "mispec_frag_basis_f16",
    "mispec_fold_taps",
    "mispec_basis_fold_bytes",
    "mispec_fold_basis_bf16",
    "mispec_fold_basis_f32",
    "mispec_fold_basis_f16",
    "mispec_basis_fold2_bytes",
    "mispec_fold2_basis",
    "mispec_filterbank_f32",
    "mispec_istft_grad_signal_f32",
    "mispec_power_to_db_f32",
    "mispec_mfcc_tail_f32",
    "mispec_power_to_db_bwd_f32",
    "mispec_contract_planar_f32",
    "mispec_pad_signal_f32",
    "mispec_unpad_adjoint_f32",
    "mispec_frame_offsets_i64",
    "mispec_framed_epilogue_bwd_f32",
    "mispec_framed_epilogue_fwd_f32",
    "mispec_frames_transpose_f32",
    "mispec_istft_frames_f32",
    "mispec_istft_frames_fft_f32",
    "mispec_istft_fft_f32",
    "mispec_overlap_add_f32",
    "mispec_octave_pyramid_f32",
    "mispec_fir_decimate_f32",
    "mispec_fir_decimate_workspace_bytes",
    "mispec_fir_decimate_bwd_f32",
    "mispec_framed_gemm_host_f32",
    "mispec_filterbank_host_f32",
    "mispec_fir_decimate_host_f32",
    "mispec_power_to_db_host_f32",
    "mispec_istft_host_f32",
)


class FramedGemmArgs(ctypes.Structure):
    """struct mispec_framed_gemm_args"""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("tile", ctypes.c_int32),
        ("x", ctypes.c_void_p),
        ("x_clip_stride", ctypes.c_int64),
        ("n_clips", ctypes.c_int32),
        ("n_samples", ctypes.c_int32),
        ("hop", ctypes.c_int32),
        ("pad", ctypes.c_int32),
        ("pad_mode", ctypes.c_int32),
        ("n_frames", ctypes.c_int32),
        ("basis_re", ctypes.c_void_p),
        ("basis_im", ctypes.c_void_p),
        ("basis_row_stride", ctypes.c_int64),
        ("n_bins", ctypes.c_int32),
        ("kernel", ctypes.c_int32),
        ("row_support", ctypes.c_void_p),
        ("row_scale", ctypes.c_void_p),
        ("epilogue", ctypes.c_int32),
        ("im_sign", ctypes.c_float),
        ("eps", ctypes.c_float),
        ("power", ctypes.c_float),
        ("out", ctypes.c_void_p),
        ("out_clip_stride", ctypes.c_int64),
        ("out_row_stride", ctypes.c_int64),
        ("out_row_offset", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_int64),
        ("precision", ctypes.c_int32),
        ("reserved2", ctypes.c_int32),
        ("basis_split", ctypes.c_void_p),
        ("basis_split_bytes", ctypes.c_int64),
        ("fb", ctypes.c_void_p),
        ("fb_support", ctypes.c_void_p),
        ("fb_row_stride", ctypes.c_int64),
        ("n_fb", ctypes.c_int32),
        ("out_frame_major", ctypes.c_int32),
        ("basis_fold", ctypes.c_void_p),
        ("basis_fold_bytes", ctypes.c_int64),
        ("fold_taps", ctypes.c_int32),
        ("reserved4", ctypes.c_int32),
        ("row_support_host", ctypes.c_void_p),
        ("basis_fold2", ctypes.c_void_p),
        ("basis_fold2_bytes", ctypes.c_int64),
        ("fold2_wmax", ctypes.c_float),
        ("no_fft", ctypes.c_int32),
        ("basis_chain", ctypes.c_void_p),
        ("basis_chain_bytes", ctypes.c_int64),
    ]


class PlanarArgs(ctypes.Structure):
    """struct mispec_planar_args"""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("rows_inner", ctypes.c_int32),
        ("a", ctypes.c_void_p),
        ("a_row_stride", ctypes.c_int64),
        ("m", ctypes.c_int32),
        ("k", ctypes.c_int32),
        ("x", ctypes.c_void_p),
        ("x_clip_stride", ctypes.c_int64),
        ("x_k_stride", ctypes.c_int64),
        ("x_col_stride", ctypes.c_int32),
        ("k_split", ctypes.c_int32),
        ("k_split_off", ctypes.c_int64),
        ("k_offsets", ctypes.c_void_p),
        ("n_clips", ctypes.c_int32),
        ("n_cols", ctypes.c_int32),
        ("out", ctypes.c_void_p),
        ("out_clip_stride", ctypes.c_int64),
        ("out_row_stride", ctypes.c_int64),
        ("out_col_stride", ctypes.c_int64),
    ]


class OctaveLevel(ctypes.Structure):
    """struct mispec_octave_level"""

    _fields_ = [
        ("bank_split", ctypes.c_void_p),
        ("bank_split_bytes", ctypes.c_int64),
        ("n_bins", ctypes.c_int32),
        ("kernel", ctypes.c_int32),
        ("out_row_offset", ctypes.c_int32),
        ("pad_mode", ctypes.c_int32),
        ("row_scale", ctypes.c_void_p),
    ]


class OctaveArgs(ctypes.Structure):
    """struct mispec_octave_args"""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("n_levels", ctypes.c_int32),
        ("x", ctypes.c_void_p),
        ("x_clip_stride", ctypes.c_int64),
        ("n_clips", ctypes.c_int32),
        ("n_samples", ctypes.c_int32),
        ("hop", ctypes.c_int32),
        ("n_frames", ctypes.c_int32),
        ("taps", ctypes.c_void_p),
        ("reserved", ctypes.c_int64),
        ("n_taps", ctypes.c_int32),
        ("epilogue", ctypes.c_int32),
        ("im_sign", ctypes.c_float),
        ("eps", ctypes.c_float),
        ("level", OctaveLevel * 3),
        ("x_last", ctypes.c_void_p),
        ("x_last_clip_stride", ctypes.c_int64),
        ("out", ctypes.c_void_p),
        ("out_clip_stride", ctypes.c_int64),
        ("out_row_stride", ctypes.c_int64),
        ("precision", ctypes.c_int32),
        ("fir_headroom_bits", ctypes.c_int32),
        ("absmax_in", ctypes.c_void_p),
        ("absmax_in_ready", ctypes.c_int32),
        ("reserved2", ctypes.c_int32),
        ("absmax_out", ctypes.c_void_p),
    ]


STREAM_MAX_LEVELS = 5


class OctaveStreamArgs(ctypes.Structure):
    """struct mispec_octave_stream_args"""

    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("n_levels", ctypes.c_int32),
        ("x", ctypes.c_void_p),
        ("x_clip_stride", ctypes.c_int64),
        ("n_clips", ctypes.c_int32),
        ("n_samples", ctypes.c_int32),
        ("hop", ctypes.c_int32),
        ("n_frames", ctypes.c_int32),
        ("taps", ctypes.c_void_p),
        ("n_taps", ctypes.c_int32),
        ("epilogue", ctypes.c_int32),
        ("im_sign", ctypes.c_float),
        ("eps", ctypes.c_float),
        ("level", OctaveLevel * STREAM_MAX_LEVELS),
        ("x_last", ctypes.c_void_p),
        ("x_last_clip_stride", ctypes.c_int64),
        ("out", ctypes.c_void_p),
        ("out_clip_stride", ctypes.c_int64),
        ("out_row_stride", ctypes.c_int64),
        ("precision", ctypes.c_int32),
        ("fir_headroom_bits", ctypes.c_int32),
        ("n_segments", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class OctaveStreamPlan(ctypes.Structure):
    """struct mispec_octave_stream_plan"""

    _fields_ = [
        ("n_levels", ctypes.c_int32),
        ("frames_per_step", ctypes.c_int32),
        ("blocks_per_tile", ctypes.c_int32),
        ("n_blocks", ctypes.c_int32),
        ("n_segments", ctypes.c_int32),
        ("blocks_per_segment", ctypes.c_int32),
        ("warm_steps", ctypes.c_int32),
        ("lds_bytes", ctypes.c_int32),
        ("length", ctypes.c_int32 * STREAM_MAX_LEVELS),
        ("lookahead", ctypes.c_int32 * STREAM_MAX_LEVELS),
        ("ring_rows", ctypes.c_int32 * STREAM_MAX_LEVELS),
        ("contract_wave", ctypes.c_int32 * STREAM_MAX_LEVELS),
    ]


class MispecError(RuntimeError):
    pass


_lib = None
_lib_ablate = None


def load_ablate():
    """The benchmarking build (``python -m nnaudio_amd.build --ablate``): same entry points, with
    the ablation / A-B bits of ``reserved`` compiled in.  Never used by the modules."""
    global _lib_ablate
    if _lib_ablate is None:
        _lib_ablate = _load(ABLATE_LIB_PATH, "python -m nnaudio_amd.build --ablate")
    return _lib_ablate


def load():
    """Load (once) and return the ctypes handle; raise if the extension is not built."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, "python -m nnaudio_amd.build")
    return _lib


def _load(path, how):
    if not os.path.exists(path):
        raise MispecError(
            "%s not found at %s -- build it with `%s` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % (os.path.basename(path), path, how)
        )
    lib = ctypes.CDLL(path)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise MispecError("libmispec.so does not export %s" % name)
    lib.mispec_version.restype = ctypes.c_int
    lib.mispec_version.argtypes = []
    lib.mispec_last_error.restype = ctypes.c_char_p
    lib.mispec_last_error.argtypes = []
    for fn in (lib.mispec_framed_gemm_f32, lib.mispec_framed_gemm_f32_ref):
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(FramedGemmArgs), ctypes.c_void_p]
    lib.mispec_framed_gemm_group_f32.restype = ctypes.c_int
    lib.mispec_framed_gemm_group_f32.argtypes = [ctypes.POINTER(FramedGemmArgs), ctypes.c_int32,
                                                 ctypes.c_void_p]
    lib.mispec_framed_gemm_workspace_bytes.restype = ctypes.c_int64
    lib.mispec_framed_gemm_workspace_bytes.argtypes = [ctypes.POINTER(FramedGemmArgs)]
    lib.mispec_basis_frag_bytes.restype = ctypes.c_int64
    lib.mispec_basis_frag_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.mispec_frag_basis_f32.restype = ctypes.c_int
    lib.mispec_frag_basis_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.mispec_basis_chain_bytes.restype = ctypes.c_int64
    lib.mispec_basis_chain_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    lib.mispec_chain_basis_f32.restype = ctypes.c_int
    lib.mispec_chain_basis_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.mispec_basis_split16_bytes.restype = ctypes.c_int64
    lib.mispec_basis_split16_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.mispec_split_basis_f16.restype = ctypes.c_int
    lib.mispec_split_basis_f16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.mispec_basis_frag16_bytes.restype = ctypes.c_int64
    lib.mispec_basis_frag16_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.mispec_frag_basis_f16.restype = ctypes.c_int
    lib.mispec_frag_basis_f16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.mispec_strip_plan.restype = ctypes.c_int32
    lib.mispec_strip_plan.argtypes = [ctypes.POINTER(FramedGemmArgs), ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_int32]
    lib.mispec_basis_split_bytes.restype = ctypes.c_int64
    lib.mispec_basis_split_bytes.argtypes = [ctypes.c_int32] * 3
    lib.mispec_split_basis_bf16.restype = ctypes.c_int
    lib.mispec_split_basis_bf16.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
    ]
    lib.mispec_fold_taps.restype = ctypes.c_int32
    lib.mispec_fold_taps.argtypes = [ctypes.c_int32] * 2
    lib.mispec_basis_fold_bytes.restype = ctypes.c_int64
    lib.mispec_basis_fold_bytes.argtypes = [ctypes.c_int32] * 3
    lib.mispec_fold_basis_bf16.restype = ctypes.c_int
    lib.mispec_fold_basis_bf16.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_fold_basis_f16.restype = ctypes.c_int
    lib.mispec_fold_basis_f16.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_fold_basis_f32.restype = ctypes.c_int
    lib.mispec_fold_basis_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_basis_fold2_bytes.restype = ctypes.c_int64
    lib.mispec_basis_fold2_bytes.argtypes = [ctypes.c_int32] * 2
    lib.mispec_fold2_basis.restype = ctypes.c_int
    lib.mispec_fold2_basis.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_octave_pyramid_f32.restype = ctypes.c_int
    lib.mispec_octave_pyramid_f32.argtypes = [ctypes.POINTER(OctaveArgs), ctypes.c_void_p]
    lib.mispec_fir_decimate_workspace_bytes.restype = ctypes.c_int64
    lib.mispec_fir_decimate_workspace_bytes.argtypes = [ctypes.c_int32] * 6
    lib.mispec_filterbank_f32.restype = ctypes.c_int
    lib.mispec_filterbank_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_contract_planar_f32.restype = ctypes.c_int
    lib.mispec_contract_planar_f32.argtypes = [ctypes.POINTER(PlanarArgs), ctypes.c_void_p]
    lib.mispec_pad_signal_f32.restype = ctypes.c_int
    lib.mispec_pad_signal_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_unpad_adjoint_f32.restype = ctypes.c_int
    lib.mispec_unpad_adjoint_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
    ]
    lib.mispec_frame_offsets_i64.restype = ctypes.c_int
    lib.mispec_frame_offsets_i64.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
        ctypes.c_void_p,
    ]
    lib.mispec_framed_epilogue_bwd_f32.restype = ctypes.c_int
    lib.mispec_framed_epilogue_bwd_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_framed_epilogue_fwd_f32.restype = ctypes.c_int
    lib.mispec_framed_epilogue_fwd_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
        ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_frames_transpose_f32.restype = ctypes.c_int
    lib.mispec_frames_transpose_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_istft_frames_f32.restype = ctypes.c_int
    lib.mispec_istft_frames_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_istft_frames_fft_f32.restype = ctypes.c_int
    lib.mispec_istft_frames_fft_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_void_p,
    ]
    lib.mispec_power_to_db_host_f32.restype = ctypes.c_int
    lib.mispec_power_to_db_host_f32.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    lib.mispec_istft_host_f32.restype = ctypes.c_int
    lib.mispec_istft_host_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
    ]
    lib.mispec_istft_fft_f32.restype = ctypes.c_int
    lib.mispec_istft_fft_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p,
    ]
    lib.mispec_overlap_add_f32.restype = ctypes.c_int
    lib.mispec_overlap_add_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
        ctypes.c_void_p,
    ]
    lib.mispec_istft_grad_signal_f32.restype = ctypes.c_int
    lib.mispec_istft_grad_signal_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_void_p,
    ]
    lib.mispec_power_to_db_bwd_f32.restype = ctypes.c_int
    lib.mispec_power_to_db_bwd_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float,
        ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
    ]
    lib.mispec_mfcc_tail_f32.restype = ctypes.c_int
    lib.mispec_mfcc_tail_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
        ctypes.c_float, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
    ]
    lib.mispec_power_to_db_f32.restype = ctypes.c_int
    lib.mispec_power_to_db_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
        ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
    ]
    lib.mispec_fir_decimate_bwd_f32.restype = ctypes.c_int
    lib.mispec_fir_decimate_bwd_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_int32, ctypes.c_void_p,
    ]
    lib.mispec_fir_decimate_f32.restype = ctypes.c_int
    lib.mispec_fir_decimate_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
    ]
    lib.mispec_framed_gemm_host_f32.restype = ctypes.c_int
    lib.mispec_framed_gemm_host_f32.argtypes = [ctypes.POINTER(FramedGemmArgs)]
    lib.mispec_filterbank_host_f32.restype = ctypes.c_int
    lib.mispec_filterbank_host_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
        ctypes.c_int32, ctypes.c_void_p,
    ]
    lib.mispec_fir_decimate_host_f32.restype = ctypes.c_int
    lib.mispec_fir_decimate_host_f32.argtypes = [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
        ctypes.c_int32,
    ]
    lib.mispec_octave_stream_f32.restype = ctypes.c_int
    lib.mispec_octave_stream_f32.argtypes = [ctypes.POINTER(OctaveStreamArgs), ctypes.c_void_p]
    lib.mispec_octave_stream_plan_of.restype = ctypes.c_int
    lib.mispec_octave_stream_plan_of.argtypes = [ctypes.POINTER(OctaveStreamArgs), ctypes.c_int32,
                                                 ctypes.POINTER(OctaveStreamPlan)]
    v = lib.mispec_version()
    if v != ABI_VERSION:
        raise MispecError("libmispec ABI version %d, expected %d" % (v, ABI_VERSION))
    return lib


def check(rc, lib=None):
    if rc != 0:
        msg = (lib or load()).mispec_last_error()
        raise MispecError(
            "libmispec call failed (%d): %s" % (rc, msg.decode() if msg else "")
        )
