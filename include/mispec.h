/*
 * mispec.h -- C ABI of libmispec.so, the MI355X (gfx950) spectrogram hot path.
 *
 * The reference (KinWaiCheuk/nnAudio, pure Python) has no FFI layer: its operator
 * boundary is the set of ATen calls inside the feature modules' forward():
 *
 *   F.conv1d(x, wsin/wcos, stride=hop) + pad + |.|/stack/atan2
 *                       Installation/nnAudio/features/stft.py:278-316
 *   F.conv1d(x, cqt_kernels_real/imag, stride=hop) + sqrt(lenghts) scaling
 *                       Installation/nnAudio/features/cqt.py:740-780
 *   pad + 2x conv1d per octave (get_cqt_complex)
 *                       Installation/nnAudio/utils.py:498-521
 *   F.conv1d(x, lowpass, stride=n, padding=127) (downsampling_by_n)
 *                       Installation/nnAudio/utils.py:73-124
 *   spec ** power ; torch.matmul(mel_basis | gammatone_basis, spec)
 *                       Installation/nnAudio/features/mel.py:184-189,
 *                       Installation/nnAudio/features/gammatone.py:184-189
 *
 * Each entry point below replaces one of those call groups.  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 (or int32 where stated), owned by the
 *     caller; the library never allocates, frees or synchronises;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*) of the CURRENT device;
 *     the functions are re-entrant and thread-safe (one host thread per GPU is the
 *     reference's nn.DataParallel calling pattern, tests/test_stft.py:134-139);
 *   - return value 0 = enqueued; negative = MISPEC_E_*; mispec_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 */
#ifndef MISPEC_H
#define MISPEC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISPEC_ABI_VERSION 13

enum {
  MISPEC_OK = 0,
  MISPEC_E_INVALID = -1,     /* bad argument (null pointer, non-positive size, ...)   */
  MISPEC_E_UNSUPPORTED = -2, /* well-formed but outside what the kernels implement    */
  MISPEC_E_HIP = -3          /* a HIP runtime call failed (launch, attribute, ...)    */
};

/* how samples outside [0, n_samples) are produced while framing (never materialised) */
enum {
  MISPEC_PAD_NONE = 0,   /* center=False: every frame lies inside the signal            */
  MISPEC_PAD_ZERO = 1,   /* nn.ConstantPad1d(pad, 0)                                    */
  MISPEC_PAD_REFLECT = 2 /* nn.ReflectionPad1d(pad): mirror without repeating the edge  */
};

/* pointwise epilogue applied to (re, im) = (sum x*basis_re, im_sign * sum x*basis_im),
 * both first multiplied by row_scale[bin] when given */
enum {
  MISPEC_EPI_COMPLEX = 0,      /* out[..., 0:2] = (re, im)          stft.py:308-311, cqt.py:774-775 */
  MISPEC_EPI_MAGNITUDE = 1,    /* sqrt(re^2 + im^2 + eps)            stft.py:299-306, cqt.py:766-772 */
  MISPEC_EPI_POWER = 2,        /* sqrt(re^2 + im^2 + eps) ** power   mel.py:186 (fused pow)          */
  MISPEC_EPI_PHASE_ATAN2 = 3,  /* atan2(im + 0.0, re)                stft.py:313-316                 */
  MISPEC_EPI_PHASE_COSSIN = 4, /* (cos a, sin a), a = atan2(im, re)  cqt.py:777-780                  */
  MISPEC_EPI_REAL = 5          /* basis_im == NULL: out = scaled sum (real contraction)              */
};

/* tile shapes of the MFMA kernel (rows x frames per workgroup); 0 lets the library pick */
enum {
  MISPEC_TILE_AUTO = 0,
  MISPEC_TILE_128x128 = 1,
  MISPEC_TILE_32x256 = 2,
  MISPEC_TILE_64x256 = 3,
  MISPEC_TILE_128x128_TALL = 4, /* one wave column: every wave owns all 4 row tiles */
  MISPEC_TILE_192x128 = 5,
  MISPEC_TILE_256x128 = 6,
  MISPEC_TILE_256x128_SQ = 7, /* 2x2 waves, 128x64 per wave: one wave per SIMD           */
  MISPEC_TILE_128x256_SQ = 8, /* 2x2 waves, 64x128 per wave                              */
  MISPEC_TILE_256x256 = 9     /* 2x2 waves, 128x128 per wave (256 accumulator registers)  */
};

/* arithmetic of the framed contraction (the north star allows "MFMA bf16/fp32" at 1e-4 rel) */
enum {
  MISPEC_PREC_F32 = 0,   /* v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulate (an fmaf chain) */
  MISPEC_PREC_BF16X3 = 1 /* fp32 operands split into bf16 (hi, lo) pairs; every product is          */
                         /* a_hi*x_hi + a_hi*x_lo + a_lo*x_hi on v_mfma_f32_32x32x16_bf16 with fp32   */
                         /* accumulate: ~16 operand mantissa bits, error ~5e-6 of the spectrum peak. */
                         /* Needs basis_split; shapes it does not cover run in MISPEC_PREC_F32.      */
  , MISPEC_PREC_F16X3 = 2 /* the same three products on v_mfma_f32_32x32x16_f16 with (hi, lo) fp16 pairs */
                         /* of power-of-two SCALED operands (fp16 has 11 significant bits: 22 per      */
                         /* operand instead of bf16x3's 16; the scales keep the pairs inside fp16's     */
                         /* 5-bit exponent and are undone on the accumulators): error ~1e-7 of the     */
                         /* spectrum peak, i.e. fp32 class, at the bf16x3 MFMA count.  Served where     */
                         /* basis_fold2 / basis_fold apply; with basis_split = mispec_frag_basis_f16()  */
                         /* for banks with supports (strip kernel, hop-periodic tap order); with         */
                         /* basis_split = mispec_split_basis_f16() for any complex basis of more than   */
                         /* 64 bins (staged dense kernel, natural tap order; supports honoured); other  */
                         /* shapes run in MISPEC_PREC_F32.                                              */
};

/*
 * Framed contraction ("strided conv1d as a GEMM whose B operand is never materialised"):
 *
 *   acc_re[c,f,t] = sum_{n<kernel} X(c, t*hop - pad + n) * basis_re[f, n]
 *   acc_im[c,f,t] = sum_{n<kernel} X(c, t*hop - pad + n) * basis_im[f, n]
 *
 * X(c, p) = x[c, p] inside the signal, else per pad_mode.  Output element (c, f, t) goes to
 *   out + c*out_clip_stride + (out_row_offset + f)*out_row_stride + t*E   (E = 2 for the
 *   two-component epilogues, else 1), i.e. the reference's (batch, freq_bins, n_frames[, 2])
 *   layout when out_row_stride = n_frames*E and out_clip_stride = rows_total*n_frames*E.
 */
typedef struct mispec_framed_gemm_args {
  uint32_t struct_size;        /* sizeof(mispec_framed_gemm_args), for ABI evolution       */
  int32_t tile;                /* MISPEC_TILE_*                                            */

  const float *x;              /* (n_clips, n_samples), row stride x_clip_stride elements  */
  int64_t x_clip_stride;
  int32_t n_clips;
  int32_t n_samples;

  int32_t hop;                 /* samples between successive frames (> 0)                  */
  int32_t pad;                 /* virtual padding on each side (0 with MISPEC_PAD_NONE)    */
  int32_t pad_mode;            /* MISPEC_PAD_*                                             */
  int32_t n_frames;            /* frames per clip; caller guarantees the last frame fits   */

  const float *basis_re;       /* (n_bins, kernel), row stride basis_row_stride elements   */
  const float *basis_im;       /* same shape, or NULL for a real contraction               */
  int64_t basis_row_stride;
  int32_t n_bins;
  int32_t kernel;
  const int32_t *row_support;  /* optional (n_bins, 2) [start, stop): taps outside are     */
                               /* KNOWN to be zero in both bases and may be skipped        */
  const float *row_scale;      /* optional (n_bins,) multiplier (sqrt(lenghts) etc.)       */

  int32_t epilogue;            /* MISPEC_EPI_*                                             */
  float im_sign;               /* -1 reproduces the reference's "-conv1d(x, imag)"         */
  float eps;                   /* added under the square root (1e-8 when trainable)        */
  float power;                 /* exponent for MISPEC_EPI_POWER                            */

  float *out;
  int64_t out_clip_stride;     /* elements                                                 */
  int64_t out_row_stride;      /* elements                                                 */
  int32_t out_row_offset;      /* first output row this call writes (octave row block)     */
  int32_t reserved;            /* must be 0 (benchmark ablation bits)                      */

  void *workspace;             /* device scratch of >= mispec_framed_gemm_workspace_bytes() */
  int64_t workspace_bytes;     /* bytes; may be NULL/0 when the query returns 0            */

  int32_t precision;           /* MISPEC_PREC_*: the LOWEST precision the caller accepts   */
  int32_t reserved2;           /* must be 0                                                */
  const void *basis_split;     /* MISPEC_PREC_BF16X3: output of mispec_split_basis_bf16();  */
                               /* MISPEC_PREC_F32 (optional): of mispec_frag_basis_f32();   */
                               /* MISPEC_PREC_F16X3: of mispec_frag_basis_f16() or of       */
                               /* mispec_split_basis_f16() (told apart by their sizes)      */
  int64_t basis_split_bytes;   /* for this (basis_re, basis_im, n_bins, kernel); else NULL */

  /* Fused filterbank reduction (mel.py:184-189: matmul(mel_basis, spec ** power)) -- optional.
   * With fb != NULL the epilogue must be MISPEC_EPI_POWER with power 1 or 2, and the launch
   * computes  out[c, m, t] = sum_bin fb[m, bin] * |X[c, bin, t]|^power  for the n_fb filters into
   * `out` = (n_clips, n_fb, n_frames), contiguous -- the library clears it itself (a
   * hipMemsetAsync on `stream`) and the contraction adds (workgroups own 128-bin
   * blocks; a filter whose band crosses a block boundary receives one atomic addend per block,
   * so bands of up to 129 bins sum in an order-independent way).  The band of every filter is
   * walked bin by bin: meant for banded (mel) filterbanks.
   * fb_support[m] = [first, last+1) bin with a non-zero weight; n_fb <= 256.  Served by both
   * precisions with the automatic tile choice (MISPEC_E_UNSUPPORTED otherwise, and in grouped
   * launches) -- use mispec_filterbank_f32 on the power spectrogram then. */
  const float *fb;             /* (n_fb, n_bins), row stride fb_row_stride elements, or NULL */
  const int32_t *fb_support;   /* (n_fb, 2)                                                */
  int64_t fb_row_stride;
  int32_t n_fb;
  int32_t out_frame_major;     /* 0: out[c, row, t] as the strides above say.  1 (round 5): FRAME-MAJOR -- element  */
                               /* (c, bin, t) at out + c out_clip_stride + t out_row_stride + bin, the floats       */
                               /* [n_bins, out_row_stride) of every frame's row zeroed: the spectrum as the framed  */
                               /* operand of a contraction over the bins (Gammatonegram's dense filterbank,         */
                               /* gammatone.py:184-189).  Served by the FFT path only: kernel 1024 or 2048,         */
                               /* n_bins = kernel/2 + 1, MISPEC_EPI_POWER, no fused filterbank, out_row_offset 0,   */
                               /* n_bins <= out_row_stride <= kernel/2 + 64; MISPEC_E_UNSUPPORTED otherwise.        */

  /* Symmetric fold (optional, either precision): for a basis that is even (basis_re) / odd
   * (basis_im) about tap kernel/2 -- every Fourier basis of stft.py:230-245 -- the contraction runs
   * over kernel/2 (+1) folded taps of  x[n] + x[kernel-n]  and  x[n] - x[kernel-n]  instead of
   * `kernel` taps: half the MFMAs.  basis_fold is the output of mispec_fold_basis_bf16()
   * (MISPEC_PREC_BF16X3), mispec_fold_basis_f16() (MISPEC_PREC_F16X3) or mispec_fold_basis_f32() (MISPEC_PREC_F32: same size, fp32 taps) for
   * this (basis_re, basis_im, n_bins, kernel) -- the format must match `precision`; the CALLER
   * vouches for the symmetry (the fold routine reports what it neglects).  Used when the shape
   * allows (even kernel, dense complex basis, hop >= kernel/8, automatic tile), otherwise ignored. */
  const void *basis_fold;      /* or NULL                                                  */
  int64_t basis_fold_bytes;
  int32_t fold_taps;           /* value returned by mispec_fold_taps() for this basis      */
  int32_t reserved4;           /* must be 0                                                */

  /* The same (n_bins, 2) values as row_support, in HOST memory -- optional.  With them the
   * library can plan the strip kernel for MISPEC_PREC_BF16X3 contractions of complex bases with
   * supports (CQT banks: cqt.py:749-750 with the kernels of utils.py:457-469): the waves of a
   * workgroup are dealt out to the 16-bin row tiles in proportion to their tap ranges.  The
   * CALLER vouches that the two copies agree; without the host copy the narrow-tile kernel runs. */
  const int32_t *row_support_host;

  /* Second symmetric fold (optional, any precision): for a basis that is  window x DFT
   * (basis_re[k, n] = w[n] cos(2 pi k n / kernel), basis_im[k, n] = w[n] sin(...), k = 0 .. n_bins-1:
   * stft.py:230-245 with freq_scale='no') the contraction runs over kernel/4 (+1) taps of the four
   * combinations of  w x  at n, kernel-n, kernel/2-n, kernel/2+n  -- even and odd bins use different
   * combinations -- a QUARTER of the dense MFMAs.  basis_fold2 is the output of mispec_fold2_basis() for
   * this (basis_re, basis_im, n_bins, kernel) in this `precision`; the CALLER vouches that the basis
   * has that form (the routine reports how far it is from it).  fold2_wmax = max |w| (stats[2]).  Used
   * when the shape allows (kernel % 64 == 0, 128 .. 8192, >= 128 bins, hop >= kernel/8, no supports / row
   * scale / fused filterbank, automatic tile); takes precedence over basis_fold.  It also opens the FFT path
   * (mispec_framed_gemm_f32 below), which has none of these shape conditions but kernel = 256 .. 2048. */
  const void *basis_fold2;     /* or NULL                                                  */
  int64_t basis_fold2_bytes;
  float fold2_wmax;
  int32_t no_fft;              /* non-zero: never take the FFT path (below)                */

  /* MISPEC_PREC_F32, complex bank with supports (CQT1992v2, cqt.py:749-750), optional (ABI 12): the bank as
   * mispec_chain_basis_f32() lays it out -- 16-row x 16-tap fragments in the order the chain kernel consumes them.
   * With it, row_support, row_support_host, 64 <= hop <= 512, hop % 64 == 0 and the automatic tile the contraction runs
   * on the chain kernel: each wave keeps its 16 frames' samples in an LDS delay line (every sample enters LDS once per
   * 16 frames instead of once per frame and K stage), no workspace.  The arithmetic is unchanged: every output is ONE
   * float32 FMA chain over the taps in ascending order, the same bits as without it. */
  const void *basis_chain;     /* or NULL                                                  */
  int64_t basis_chain_bytes;
} mispec_framed_gemm_args;

/*
 * Scratch the framed contraction needs for this problem: the padded edge spans of every clip
 * (the handful of frames per clip that touch the virtual padding or run past the clip end are
 * staged there by a pre-pass so that every frame is a plain run of memory; interior frames are
 * read straight from x).  Depends only on the sizes in `args`; 0 when every frame is interior
 * (center=False with kernel % 32 == 0).  With MISPEC_PREC_BF16X3 it also holds the (hi, lo)
 * bf16 planes of the waveform and of the edge spans (4 more bytes per sample), or -- with
 * basis_fold -- the folded frames (8 bytes per folded tap and frame).
 * Negative = MISPEC_E_*.
 */
int64_t mispec_framed_gemm_workspace_bytes(const mispec_framed_gemm_args *args);

/*
 * Host-only planning query (no device work): how mispec_framed_gemm_f32 would run this problem
 * on a device with n_cu compute units when it is a MISPEC_PREC_BF16X3 contraction of a complex
 * basis with supports and row_support_host is given.  Returns the number of passes of the strip
 * kernel (0: the narrow-tile kernel runs instead; < 0: error) and writes, as far as `cap` int32
 * entries allow:  frame tiles per pass, then per pass  cost, first super-stage, super-stages,
 * slab rows, then for each of the 4 waves  row tile (16 bins; -1: idle), first tap, end tap,
 * first super-stage, end super-stage, first wave of its reduction group, waves in the group,
 * frame-tile mask.  (1 + passes * 36 entries.)
 */
int32_t mispec_strip_plan(const mispec_framed_gemm_args *args, int32_t n_cu, int32_t *plan,
                          int32_t cap);


/*
 * MISPEC_PREC_BF16X3 operand preparation for a basis: (hi, lo) bf16 planes of basis_re (and
 * basis_im), rows zero-padded to a multiple of 32 taps.  Done once per basis (the bases are the
 * modules' precomputed buffers, stft.py:230-245 / cqt.py:682-702) and again only when the basis
 * changes.  `dst` is caller-owned device memory of mispec_basis_split_bytes() bytes.
 */
int64_t mispec_basis_split_bytes(int32_t n_bins, int32_t kernel, int32_t has_im);
int mispec_split_basis_bf16(const float *basis_re, const float *basis_im,
                            int64_t basis_row_stride, int32_t n_bins, int32_t kernel,
                            void *dst, int64_t dst_bytes, void *stream);

/*
 * MISPEC_PREC_F32 counterpart for bases with supports (CQT banks): the fp32 taps in the strip
 * kernel's fragment order [16-bin tile][16-tap step][part][lane][4 taps].  Handed over in
 * `basis_split` (with precision = MISPEC_PREC_F32, row_support and row_support_host) it lets the
 * exact arithmetic run on the strip kernel too; without it the fp32 tile kernels run.
 */
int64_t mispec_basis_frag_bytes(int32_t n_bins, int32_t kernel);
int mispec_frag_basis_f32(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, void *dst, int64_t dst_bytes,
                          void *stream);

/*
 * The chain kernel's copy of a complex bank with supports (`basis_chain` above), built once per bank.
 * row_support_host: the (n_bins, 2) [start, stop) supports in HOST memory (the layout follows from them).
 *   mispec_basis_chain_bytes  size of `dst`; MISPEC_E_UNSUPPORTED when the supports of the bank's 16-row
 *                             tiles (8 bins) do not nest when ordered by length (CQT kernels are centred: they
 *                             do) or the bank has more than 640 bins -- such banks stay on the tile kernels
 *   mispec_chain_basis_f32    fills dst (one small host-synchronous copy of the plan, then a kernel on `stream`)
 */
int64_t mispec_basis_chain_bytes(const int32_t *row_support_host, int32_t n_bins, int32_t kernel);
int mispec_chain_basis_f32(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                           int32_t n_bins, int32_t kernel, const int32_t *row_support_host, void *dst,
                           int64_t dst_bytes, void *stream);

/*
 * MISPEC_PREC_F16X3 counterpart for bases with supports (CQT banks): (hi, lo) fp16 pairs of every row
 * multiplied by a power of two (its largest |tap| ends up in [2^13, 2^14)), in the strip kernel's
 * fragment order, followed by the per-row inverse factors the epilogue applies.  Handed over in
 * `basis_split` with precision = MISPEC_PREC_F16X3, row_support and row_support_host.  The signal is
 * scaled per clip (from its largest |sample|, found by a pre-pass) and split the same way.
 */
int64_t mispec_basis_frag16_bytes(int32_t n_bins, int32_t kernel);
/* ... and as row-major planes [re_hi | re_lo | im_hi | im_lo] + the inverse factors: for the levels of
 * mispec_octave_pyramid_f32 in MISPEC_PREC_F16X3, and -- as `basis_split` of a MISPEC_PREC_F16X3
 * mispec_framed_gemm_f32 with basis_split_bytes = mispec_basis_split16_bytes() exactly -- for the staged
 * dense kernel: complex bases of more than 64 bins that are not folded (CQT banks, with or without
 * row_support; trainable or non-linear STFT bases), taps contracted in their natural order */
int64_t mispec_basis_split16_bytes(int32_t n_bins, int32_t kernel);
int mispec_split_basis_f16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                           int32_t n_bins, int32_t kernel, void *dst, int64_t dst_bytes,
                           void *stream);
int mispec_frag_basis_f16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, void *dst, int64_t dst_bytes,
                          void *stream);

/*
 * Symmetric fold of a Fourier-type basis (see basis_fold above).  Folded tap j stands for the pair
 * (n, kernel - n), n = j + 1, with coefficient (re[n] + re[kernel-n]) / 2 resp.
 * (im[n] - im[kernel-n]) / 2; taps kernel/2 and 0 have no partner (tap 0 is carried only
 * with with_tap0 != 0: needed when some row has a non-zero tap 0, i.e. a window with w[0] != 0).
 *   mispec_fold_taps        folded taps per row (a multiple of 16), < 0 when the kernel cannot fold
 *   mispec_basis_fold_bytes size of `dst`
 *   mispec_fold_basis_bf16 / mispec_fold_basis_f32
 *                           build dst (split-bf16 resp. fp32 taps); stats (device, 2 floats) receives
 *                           [0] max |re[n] - re[kernel-n]| / 2, |im[n] + im[kernel-n]| / 2 over all
 *                               pairs = the largest coefficient the fold neglects,
 *                           [1] max |folded coefficient|;
 *                           offer the result to mispec_framed_gemm_f32 only when [0] is negligible
 *                           against [1] (nnaudio_amd.engine: <= 2^-20).
 */
int32_t mispec_fold_taps(int32_t kernel, int32_t with_tap0);
int64_t mispec_basis_fold_bytes(int32_t n_bins, int32_t kernel, int32_t with_tap0);
int mispec_fold_basis_bf16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                           int32_t n_bins, int32_t kernel, int32_t with_tap0, void *dst,
                           int64_t dst_bytes, float *stats, void *stream);
int mispec_fold_basis_f32(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, int32_t with_tap0, void *dst,
                          int64_t dst_bytes, float *stats, void *stream);
/* MISPEC_PREC_F16X3: (hi, lo) fp16 pairs of coefficient x 2^14 -- offer it only when stats[1] <= 2 */
int mispec_fold_basis_f16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, int32_t with_tap0, void *dst,
                          int64_t dst_bytes, float *stats, void *stream);

/*
 * Second fold of a window x DFT basis (see basis_fold2 above): the quarter-folded coefficient planes
 * from the analytic DFT, in the format of `precision` (MISPEC_PREC_*).  stats (device, 3 floats):
 *   [0] max |basis[k, n] - w[n] * dft(k, n)| over every coefficient, w = row 0 of basis_re
 *   [1] max |basis coefficient|     [2] max |w|
 * Offer the result only when [0] is rounding noise against [1] (nnaudio_amd.engine: <= 2^-20).
 */
int64_t mispec_basis_fold2_bytes(int32_t n_bins, int32_t kernel);
int mispec_fold2_basis(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                       int32_t n_bins, int32_t kernel, int32_t precision, void *dst, int64_t dst_bytes,
                       float *stats, void *stream);

/* Launch the MFMA framed contraction.  Replaces stft.py:278-316, cqt.py:740-780,
 * utils.py:498-521 (one call per octave).
 * The FFT path: with basis_fold2 (the proof that the kernels are window x DFT), kernel = 256, 512, 1024 or 2048,
 * n_bins <= kernel/2 + 1, a pointwise epilogue (with or without the fused filterbank), automatic tile and
 * no_fft == 0 the call is ONE launch that evaluates every frame's DFT as an fp32 FFT (2e-7 of the peak) whatever
 * `precision` says -- no workspace, nothing else of the argument block changes meaning.  kernel = 256 runs as
 * zero-extended frames on the 512-point transform.  kernel = 4096 (even hop and pad, no fused filterbank, not the
 * (cos, sin) phase format) runs COMPOSITE: the padded clips de-interleaved into the workspace, two 2048-point
 * transforms of the even / odd samples, the second one's flush forming the butterfly + epilogue -- workspace
 * sized by mispec_framed_gemm_workspace_bytes() like every other route. */
int mispec_framed_gemm_f32(const mispec_framed_gemm_args *args, void *stream);

/* `n` (<= 8) independent contractions in ONE launch: the octaves of CQT2010v2 / VQT
 * (cqt.py:1091-1105, vqt.py:167-188) are each too small to fill the chip.  Every problem must
 * use the same narrow tile shape (<= 64 basis rows, no per-stage row masking); otherwise
 * MISPEC_E_UNSUPPORTED is returned and the caller launches them one by one.                */
int mispec_framed_gemm_group_f32(const mispec_framed_gemm_args *args, int32_t n, void *stream);

/* Same contract, one thread per output element, no MFMA, no LDS: a slow device-side
 * cross-check used only by the test-suite to separate indexing bugs from MFMA bugs.   */
int mispec_framed_gemm_f32_ref(const mispec_framed_gemm_args *args, void *stream);

/*
 * Filterbank reduction  out[c, m, t] = sum_{f<n_freq} fb[m, f] * spec[c, f, t]
 * (torch.matmul(mel_basis, spec), mel.py:188 / gammatone.py:188).  spec is the
 * (n_clips, n_freq, n_frames) tensor produced with MISPEC_EPI_POWER; out is
 * (n_clips, n_filters, n_frames); both contiguous.
 */
int mispec_filterbank_f32(const float *fb, int32_t n_filters, int32_t n_freq,
                          const float *spec, int32_t n_clips, int32_t n_frames,
                          float *out, void *stream);

/*
 * Planar contraction  out[c, m, j] = sum_{k<K} a[m, k] * X(c, k, j)   (the kernel behind the
 * filterbank, MFCC's DCT and the inverse-STFT frame synthesis, exposed for the backward pass).
 * X(c, k, j) = x[c*x_clip_stride + koff(k) + j*x_col_stride] with
 *   koff(k) = k_offsets[k]                                  when k_offsets != NULL,
 *           = (k - k_split)*x_k_stride + k_split_off        when k_split != 0 and k >= k_split,
 *           = k*x_k_stride                                  otherwise.
 * Output element (c, m, j) goes to out + c*out_clip_stride + m*out_row_stride + j, or, with
 * rows_inner != 0, to out + c*out_clip_stride + j*out_col_stride + m (m innermost).
 */
typedef struct mispec_planar_args {
  uint32_t struct_size;
  int32_t rows_inner;
  const float *a;          /* (m, k) rows a_row_stride apart                     */
  int64_t a_row_stride;
  int32_t m;
  int32_t k;
  const float *x;
  int64_t x_clip_stride;
  int64_t x_k_stride;
  int32_t x_col_stride;    /* 0 means 1                                          */
  int32_t k_split;
  int64_t k_split_off;
  const int64_t *k_offsets; /* optional (k,) element offsets                      */
  int32_t n_clips;
  int32_t n_cols;          /* columns j per clip                                 */
  float *out;
  int64_t out_clip_stride;
  int64_t out_row_stride;  /* rows_inner == 0                                    */
  int64_t out_col_stride;  /* rows_inner != 0                                    */
} mispec_planar_args;
int mispec_contract_planar_f32(const mispec_planar_args *args, void *stream);

/*
 * Backward of the framed contraction (trainable bases: stft.py:238-242, cqt.py:698-702,
 * mel.py:158-161).  With (u, v) = (s*acc_re, s*im_sign*acc_im) -- what MISPEC_EPI_COMPLEX stores,
 * s = row_scale -- and out = E(u, v):
 *   mispec_framed_epilogue_bwd_f32: (grad_out, z = (u, v) as (B, F, T, 2)) -> g, laid out
 *       (2, F, B, T): g[0] = dL/dacc_re, g[1] = dL/dacc_im;
 *   d basis and d frames: two more framed contractions (see mispec_frames_transpose_f32 below; or
 *       mispec_contract_planar_f32 with a k-offset table from mispec_frame_offsets_i64);
 *   d signal = mispec_overlap_add_f32(window = NULL: plain overlap-add) then
 *       mispec_unpad_adjoint_f32 (mirrored positions of reflect padding folded back).
 * mispec_pad_signal_f32 materialises the (n_clips, n_samples + 2*pad) padded signal.
 */
int mispec_pad_signal_f32(const float *x, int64_t x_clip_stride, int32_t n_clips, int32_t n_samples,
                          int32_t pad, int32_t pad_mode, float *out, void *stream);
int mispec_unpad_adjoint_f32(const float *dxp, int32_t n_clips, int32_t n_samples, int32_t pad,
                             int32_t pad_mode, float *dx, int64_t dx_clip_stride, void *stream);
int mispec_frame_offsets_i64(int64_t *k_offsets, int32_t n_clips, int32_t n_frames,
                             int64_t clip_stride, int32_t hop, void *stream);
/* The pointwise epilogue alone: z (B, F, T, 2) = the Complex output of a framed contraction (row scale and im_sign
 * applied) -> out (B, F, T[, 2]) in `epilogue`, by the very code the contraction kernels end with: the same bits as
 * the fused launch.  A training step runs the contraction once with the Complex epilogue, keeps z for the backward
 * (stft.py:238-242: trainable kernels) and derives the module's output here. */
int mispec_framed_epilogue_fwd_f32(const float *z, int32_t n_clips, int32_t n_bins, int32_t n_frames,
                                   int32_t epilogue, float eps, float power, float *out, void *stream);
int mispec_framed_epilogue_bwd_f32(const float *grad_out, const float *z, int32_t n_clips,
                                   int32_t n_bins, int32_t n_frames, int32_t epilogue, float eps,
                                   float power, float im_sign, const float *row_scale, float *g,
                                   float *gt, void *stream);
/* g (2, F, B, T) and / or gt (B, T, 2F) (one frame's gradient vector contiguous, [re | im]): with
 * mispec_frames_transpose_f32 (xt[n, (c,t)] = xpad[c, t*hop + n], the frame matrix tap-major)
 * both backward contractions become calls of the framed MFMA kernel itself:
 *   d basis    = framed(signal = xt flat,  basis = g as (2F, B*T), hop = kernel = B*T, real)
 *   d frames^T = framed(signal = gt flat,  basis = [re^T | im^T] (K, 2F), hop = kernel = 2F, real)
 * and d signal = mispec_overlap_add_f32 on the tap-major d frames (window NULL, start = -1 - s). */
int mispec_frames_transpose_f32(const float *xpad, int64_t clip_stride, int32_t n_clips,
                                int32_t n_frames, int32_t hop, int32_t kernel, float *xt,
                                void *stream);

/*
 * Inverse STFT (STFTBase.inverse_stft, stft.py:15-63), two steps:
 *
 *  1. frames[c, t, n] = sum_{k<n_freq} spec[c,k,t,0]*basis[n, k] + spec[c,k,t,1]*basis[n, n_freq+k]
 *     spec is the (n_clips, n_freq, n_frames, 2) complex spectrogram; basis is (n_fft, 2*n_freq):
 *     [cos part | -sin part] of kernel_cos_inv / kernel_sin_inv, with the mirrored bins of a
 *     one-sided spectrum folded in (extend_fbins, utils.py:63-70) -- built by the caller from the
 *     module's buffers.  frames is (n_clips, n_frames, n_fft), n innermost.
 *  2. y[c, i] = (sum_t frames[c,t,n]*window[n]/n_fft) / sum_t window[n]^2,  n = i + start - t*hop,
 *     the sums over the frames that cover sample i + start; the division is skipped where the
 *     window sum is <= 1e-10 (stft.py:41-51).  start = n_fft/2 when centred, out_len = trimmed
 *     length.  out rows are out_clip_stride elements apart.  window == NULL: plain overlap-add
 *     (no window, no 1/n_fft, no normalisation): the adjoint of framing, used by the backward pass.
 */
int mispec_istft_frames_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames,
                            const float *basis, int32_t n_fft, float *frames, void *stream);
/* Step 1 as an inverse real FFT (fp32), for a one-sided spectrum (n_freq == n_fft/2 + 1) and synthesis
 * kernels that are the plain inverse DFT -- kernel_cos_inv[n, k] = cos(2 pi k n / n_fft), kernel_sin_inv
 * likewise: the caller checks its buffers once -- with n_fft = 512, 1024 or 2048: the same frames as
 * mispec_istft_frames_f32 with the folded [cos | -sin] basis, to fp32 rounding (MISPEC_E_UNSUPPORTED else). */
int mispec_istft_frames_fft_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames,
                                int32_t n_fft, float *frames, void *stream);
int mispec_overlap_add_f32(const float *frames, int32_t n_clips, int32_t n_frames, int32_t n_fft,
                           const float *window, int32_t hop, int32_t start, float *out,
                           int64_t out_clip_stride, int32_t out_len, void *stream);
/* Steps 1 + 2 in ONE launch for the inverse-FFT case (mispec_istft_frames_fft_f32's conditions, hop a multiple
 * of 64 that divides n_fft; MISPEC_E_UNSUPPORTED else): a workgroup walks a run of consecutive 8-frame tiles of a clip,
 * its waves synthesise one frame each, the windowed frames are overlap-added in LDS in the order step 2 uses
 * (bit-identical output) and only the waveform is written -- the (clip, frame, sample) tensor never exists. */
int mispec_istft_fft_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames, int32_t n_fft,
                         const float *window, int32_t hop, int32_t start, float *out, int64_t out_clip_stride,
                         int32_t out_len, void *stream);

/*
 * power_to_db of MFCC (mel.py:263-279), per clip c over its `clip_elems` values (n_mels * n_frames):
 *   l   = 10*log10(max(spec, amin)) - 10*log10(max(amin, ref))
 *   out = top_db < 0 ? l : max(l, max_over_clip(l) - top_db)
 * `out` may alias `spec`.  workspace: n_clips * 4 bytes (per-clip maxima).  The discrete cosine
 * transform that follows (mel.py:281-307) is mispec_filterbank_f32 with the DCT-II matrix.
 */
int mispec_power_to_db_f32(const float *spec, int32_t n_clips, int64_t clip_elems, float amin,
                           float ref, float top_db, float *out, void *workspace,
                           int64_t workspace_bytes, void *stream);

/* Backward of mispec_overlap_add_f32 (windowed form) w.r.t. the frame sum:
 *   u[c, p] = grad_out[c, p - start] / (n_fft * wss[p])  for p in [start, start + out_len), else 0
 * over the un-trimmed axis p in [0, (n_frames-1)*hop + n_fft); u is (n_clips, that length).  The
 * gradient of the spectrogram is then mispec_framed_gemm_f32 of u (no padding) with the
 * window-weighted transposed synthesis basis. */
int mispec_istft_grad_signal_f32(const float *grad_out, int64_t grad_clip_stride, int32_t n_clips,
                                 int32_t n_frames, int32_t n_fft, const float *window, int32_t hop,
                                 int32_t start, int32_t out_len, float *u, void *stream);

/*
 * MFCC's tail in ONE launch (mel.py:263-307 = MFCC.forward after the mel spectrogram): mispec_power_to_db_f32
 * followed by mispec_filterbank_f32 with the (n_mfcc, n_mels) DCT-II matrix `dct` (row-major), one workgroup per
 * clip: mel (n_clips, n_mels, n_frames) -> out (n_clips, n_mfcc, n_frames).  The decibel values are the ones
 * mispec_power_to_db_f32 produces (same expressions); the cosine sums run over the mel bands in order (fp32 FMA).
 * top_db < 0: no floor.  MISPEC_E_UNSUPPORTED for more than 256 mel bands (a 64-frame tile in 64 KB of LDS) or
 * n_mfcc > n_mels: use the two calls.
 */
int mispec_mfcc_tail_f32(const float *mel, int32_t n_clips, int32_t n_mels, int32_t n_frames, float amin, float ref,
                         float top_db, const float *dct, int32_t n_mfcc, float *out, void *stream);

/* Backward of mispec_power_to_db_f32 with respect to spec: elements above the per-clip floor pass
 * grad_out * 10 / (ln 10 * spec) (0 where spec <= amin); the floored ones hand their gradient to
 * the clip maximum.  workspace: n_clips * 8 bytes. */
int mispec_power_to_db_bwd_f32(const float *spec, const float *grad_out, int32_t n_clips,
                               int64_t clip_elems, float amin, float top_db, float *grad_spec,
                               void *workspace, int64_t workspace_bytes, void *stream);

/*
 * Strided FIR decimation  y[c, i] = sum_{n<n_taps} taps[n] * x[c, i*stride + n - pad]
 * with zeros outside the signal (F.conv1d(x, taps, stride, padding=pad),
 * utils.py:98-99 where pad = (n_taps-1)//2).  n_out = (n_samples + 2*pad - n_taps)/stride + 1.
 */
int mispec_fir_decimate_f32(const float *x, int64_t x_clip_stride, int32_t n_clips,
                            int32_t n_samples, const float *taps, int32_t n_taps,
                            int32_t stride, int32_t pad, float *y, int64_t y_clip_stride,
                            int32_t n_out, void *workspace, int64_t workspace_bytes,
                            void *stream);

/* Adjoint of mispec_fir_decimate_f32 with respect to the signal (backward of the octave
 * recursion): dx[c, m] = sum_i dy[c, i] * taps[m + pad - i*stride], dx is (n_clips, n_samples). */
int mispec_fir_decimate_bwd_f32(const float *dy, int64_t dy_clip_stride, int32_t n_clips,
                                int32_t n_out, const float *taps, int32_t n_taps, int32_t stride,
                                int32_t pad, float *dx, int64_t dx_clip_stride, int32_t n_samples,
                                void *stream);

/*
 * Fused octave recursion of CQT2010v2 / VQT (cqt.py:1085-1105, vqt.py:160-188: per octave one
 * downsampling_by_2, utils.py:102-124, and one get_cqt_complex, utils.py:498-521).  One launch keeps
 * up to 3 consecutive levels of the recursion resident in LDS per workgroup -- level 0 = x, level
 * l+1 = the anti-alias FIR of level l at stride 2 -- and contracts every level that has a kernel
 * bank with it (frames of `kernel` taps every hop >> l samples, centred: pad = kernel/2 with
 * pad_mode), writing rows [out_row_offset, + n_bins) of the (n_clips, rows, n_frames[, 2]) output.
 * The deepest level is also stored in fp32 to x_last (when not NULL): it is the `x` of the next
 * launch of the chain, whose level 0 then carries no bank.  All arithmetic is MISPEC_PREC_BF16X3
 * (split-bf16 operands, fp32 accumulate); between levels the signal keeps 16 significant bits.
 * Shapes: (hop >> l) a multiple of 4, kernel a multiple of 16, n_bins <= 16 per level,
 * n_taps <= 257; MISPEC_E_UNSUPPORTED otherwise (the caller then runs mispec_fir_decimate_f32 +
 * mispec_framed_gemm_f32 per octave).
 */
typedef struct mispec_octave_level {
  const void *bank_split;    /* mispec_split_basis_bf16() of this level's (n_bins, kernel) complex  */
  int64_t bank_split_bytes;  /* bank, or NULL: no contraction at this level                         */
  int32_t n_bins;
  int32_t kernel;
  int32_t out_row_offset;
  int32_t pad_mode;          /* MISPEC_PAD_ZERO or MISPEC_PAD_REFLECT                               */
  const float *row_scale;    /* (n_bins,) multiplier or NULL                                        */
} mispec_octave_level;

typedef struct mispec_octave_args {
  uint32_t struct_size;
  int32_t n_levels;            /* 1 .. 3                                                   */
  const float *x;              /* (n_clips, n_samples), rows x_clip_stride elements apart  */
  int64_t x_clip_stride;
  int32_t n_clips;
  int32_t n_samples;
  int32_t hop;                 /* frame hop at level 0; level l uses hop >> l               */
  int32_t n_frames;            /* frames per clip (the same at every level)                 */
  const float *taps;           /* anti-alias filter (n_taps,), needed when n_levels > 1     */
  int64_t reserved;            /* must be 0                                                 */
  int32_t n_taps;
  int32_t epilogue;            /* MISPEC_EPI_COMPLEX / MAGNITUDE / PHASE_COSSIN / ...       */
  float im_sign;
  float eps;
  mispec_octave_level level[3];
  float *x_last;               /* (n_clips, length of the deepest level) fp32, or NULL      */
  int64_t x_last_clip_stride;
  float *out;
  int64_t out_clip_stride;     /* elements                                                  */
  int64_t out_row_stride;      /* elements                                                  */
  int32_t precision;           /* MISPEC_PREC_BF16X3, or MISPEC_PREC_F16X3: scaled fp16 pairs --  */
                               /* then every bank_split is mispec_split_basis_f16()'s output and: */
  int32_t fir_headroom_bits;   /* ceil(log2(sum |taps|)) * (n_levels - 1): bits kept free above    */
                               /* the scaled level-0 samples for the gain of the FIRs (<= 7)       */
  void *absmax_in;             /* device, n_clips * 128 bytes: bit pattern of max |x[c, :]| per    */
                               /* clip, 32 words apart (words the CALLER has zeroed)               */
  int32_t absmax_in_ready;     /* != 0: it is there (absmax_out of the launch that wrote this x as */
                               /* its x_last); 0 (first launch of a chain): the library scales per */
                               /* work item from the item's own span inside the kernel (banks of   */
                               /* up to 192 taps, spans up to 12 K samples) or, failing that,      */
                               /* first fills absmax_in with a pass over x                         */
  int32_t reserved2;           /* must be 0                                                        */
  void *absmax_out;            /* the same for x_last, gathered while it is written (atomic max    */
                               /* into words the CALLER has zeroed), or NULL                       */
} mispec_octave_args;

int mispec_octave_pyramid_f32(const mispec_octave_args *args, void *stream);

/*
 * Streaming form of the fused octave recursion (round 4; same reference lines as
 * mispec_octave_pyramid_f32: cqt.py:1085-1105, vqt.py:160-188, utils.py:73-124, 498-521).  A workgroup
 * walks a SEGMENT of one clip in steps of 4096 level-0 samples and keeps a ring of every level
 * (level l+1 = the anti-alias FIR of level l at stride 2) in LDS: no halo is recomputed per work item,
 * up to 5 levels (4 of them with a bank) are resident, and a step has one barrier -- four waves run the
 * FIRs of all levels as one 128-column Toeplitz tile set on the matrix pipe (level l+1 a step behind
 * level l), four waves contract the levels with their banks, all of them stream the next 4096 samples in
 * (LDS-direct loads).  The deepest level goes to x_last (fp32) for the next launch of a chain.
 *   hop            a multiple of 4 << (n_levels - 1), 4096 % hop == 0, hop <= 512
 *   level[l]       bank_split = mispec_split_basis_bf16 / _f16 planes of (n_bins <= 16, kernel <= 256,
 *                  kernel % 16 == 0), or NULL (level 0 of a follow-up launch); at most 4 banks
 *   n_samples      every level with a bank needs  length >= 16 * (hop >> l) + 2 * kernel
 *   x              16-byte aligned, x_clip_stride % 4 == 0
 *   precision      MISPEC_PREC_BF16X3 or MISPEC_PREC_F16X3 (then fir_headroom_bits as for the pyramid
 *                  kernel; the power-of-two operand scale is chosen by every workgroup from the samples
 *                  it has seen, and the resident rings are rescaled when a louder chunk arrives)
 *   n_segments     segments per clip (0: the library picks ~ one workgroup per CU)
 * MISPEC_E_UNSUPPORTED for every other shape (the caller falls back to mispec_octave_pyramid_f32).
 */
#define MISPEC_STREAM_MAX_LEVELS 5
typedef struct mispec_octave_stream_args {
  uint32_t struct_size;
  int32_t n_levels;            /* 1 .. 5                                                   */
  const float *x;              /* (n_clips, n_samples), rows x_clip_stride elements apart  */
  int64_t x_clip_stride;
  int32_t n_clips;
  int32_t n_samples;
  int32_t hop;                 /* frame hop at level 0; level l uses hop >> l               */
  int32_t n_frames;            /* frames per clip (the same at every level)                 */
  const float *taps;           /* anti-alias filter (n_taps,), needed when n_levels > 1     */
  int32_t n_taps;
  int32_t epilogue;            /* MISPEC_EPI_COMPLEX / MAGNITUDE / PHASE_COSSIN / ...       */
  float im_sign;
  float eps;
  mispec_octave_level level[MISPEC_STREAM_MAX_LEVELS];
  float *x_last;               /* (n_clips, length of the deepest level) fp32, or NULL      */
  int64_t x_last_clip_stride;
  float *out;
  int64_t out_clip_stride;     /* elements                                                  */
  int64_t out_row_stride;      /* elements                                                  */
  int32_t precision;
  int32_t fir_headroom_bits;   /* F16X3: ceil(log2(sum |taps|)) * (n_levels - 1), <= 7      */
  int32_t n_segments;
  int32_t reserved;            /* must be 0                                                 */
} mispec_octave_stream_args;

int mispec_octave_stream_f32(const mispec_octave_stream_args *args, void *stream);

/* The geometry mispec_octave_stream_f32 would use (host only, no device call): checked against the
 * executable model of the schedule (scripts/octave_stream_model.py) by tests/test_octave_stream_cpu.py. */
typedef struct mispec_octave_stream_plan {
  int32_t n_levels, frames_per_step, blocks_per_tile, n_blocks, n_segments, blocks_per_segment, warm_steps;
  int32_t lds_bytes;
  int32_t length[MISPEC_STREAM_MAX_LEVELS];     /* samples of level l                        */
  int32_t lookahead[MISPEC_STREAM_MAX_LEVELS];  /* block b of level l = [blk b + c, blk (b+1) + c) */
  int32_t ring_rows[MISPEC_STREAM_MAX_LEVELS];  /* rows of 64 samples (a power of two)       */
  int32_t contract_wave[MISPEC_STREAM_MAX_LEVELS]; /* wave that contracts level l, or -1     */
} mispec_octave_stream_plan;
int mispec_octave_stream_plan_of(const mispec_octave_stream_args *args, int32_t n_cus,
                                 mispec_octave_stream_plan *plan);

/*
 * Host path: the same three operations on HOST pointers, as plain C++ loops (fp32 multiply-adds in tap
 * order, a few threads) -- so that a module whose input lives in host memory computes there, like the
 * reference's forward (stft.py:290-293: conv1d runs wherever x lives; BASELINE configs[0] is a CPU
 * case).  Same argument blocks and checks; workspace, precision, basis_split / basis_fold* are ignored;
 * no fused filterbank.  Synchronous; meant for plumbing-sized inputs, not for throughput.
 */
int mispec_framed_gemm_host_f32(const mispec_framed_gemm_args *args);
int mispec_filterbank_host_f32(const float *fb, int32_t n_filters, int32_t n_freq, const float *spec,
                               int32_t n_clips, int32_t n_frames, float *out);
/* power_to_db (mel.py:263-279) and the inverse STFT (stft.py:15-63: mispec_istft_frames_f32 + mispec_overlap_add_f32)
 * on host pointers: MFCC.forward and STFT.inverse / iSTFT.forward of CPU tensors. */
int mispec_power_to_db_host_f32(const float *spec, int32_t n_clips, int64_t clip_elems, float amin, float ref,
                                float top_db, float *out);
int mispec_istft_host_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames, const float *basis,
                          int32_t n_fft, const float *window, int32_t hop, int32_t start, float *out,
                          int64_t out_clip_stride, int32_t out_len);
int mispec_fir_decimate_host_f32(const float *x, int64_t x_clip_stride, int32_t n_clips,
                                 int32_t n_samples, const float *taps, int32_t n_taps, int32_t stride,
                                 int32_t pad, float *y, int64_t y_clip_stride, int32_t n_out);

/* Scratch bytes mispec_fir_decimate_f32 needs (zero-padded clip edges); negative = error. */
int64_t mispec_fir_decimate_workspace_bytes(int32_t n_clips, int32_t n_samples, int32_t n_taps,
                                            int32_t stride, int32_t pad, int32_t n_out);

/* ABI version of the loaded library (== MISPEC_ABI_VERSION it was built with). */
int mispec_version(void);

/* Message describing the last non-zero return on this thread ("" if none). */
const char *mispec_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MISPEC_H */
