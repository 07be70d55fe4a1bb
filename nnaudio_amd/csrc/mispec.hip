// libmispec: MI355X (gfx950 / CDNA4) spectrogram hot path.
//
// One kernel family does all the arithmetic of the reference's forward() methods
// (F.conv1d with a precomputed basis, stride = hop; see include/mispec.h for the
// reference lines each entry point replaces):
//
//   D[row, col] = sum_k A[row, k] * Bop[k, col]
//
//   A   = basis rows, interleaved (re, im) per frequency bin           (M = 2*n_bins rows)
//   Bop = the frame matrix  X(clip, t*hop - pad + k)                   (N = n_clips*n_frames cols)
//         read straight from the waveform: frames are never materialised in HBM.  Only the
//         few frames per clip that touch the virtual padding (reflect / zero) or run past
//         the clip are served from a small caller-provided workspace that a pre-pass fills
//         with the padded edge spans, so that EVERY frame is a contiguous run of memory and
//         the K loop contains no control flow around its loads.
//
// This file: the fp32 path.  The contraction runs on the matrix cores with
// v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bit-for-bit an fmaf chain, so the 1e-4 parity
// bar is met with ~1e-6), 64-wide wavefronts, 32-deep K stages double-buffered through LDS and
// filled by LDS-direct loads, and the magnitude / power / phase / complex epilogue applied to the
// accumulators before a (batch, bin, frame[,2]) store with frames innermost.
//
// Variants of the same template:
//   * A as a banded Toeplitz matrix of FIR taps  -> strided decimation (utils.py:73-124)
//   * Bop read from a planar (clip, k, t) tensor -> filterbank matmul  (mel.py:188), MFCC's DCT,
//                                                   inverse-STFT frame synthesis (stft.py:15-63)
//   * per-row [start, stop) supports             -> CQT kernels skip their zero taps
//
// The bf16x3 path (MISPEC_PREC_BF16X3: fp32 operands split into bf16 pairs, three
// v_mfma_f32_32x32x16_bf16 per product) lives in framed_bf16x3.inl (staged 256x256 kernel, split
// pre-passes, shared epilogue), framed_bf16x3_slab.inl (hop-periodic K order) and
// framed_bf16x3_narrow.inl (32-row tiles with super-stage packing for CQT banks), all included
// below; the host side of both paths, the small pointwise kernels (power_to_db, overlap-add) and
// the extern "C" entry points are at the end of this file.
//
// Ablation / A-B bits ("debug" below) exist only in the benchmarking build (-DMISPEC_ABLATE ->
// libmispec_ablate.so, used by scripts/kbench.py and scripts/profile.sh): there `reserved` of
// mispec_framed_gemm_args selects them (results are WRONG for bits 1-16, 0x40000, 0x80000):
// 1 no global loads in the K loop, 2 no LDS stores, 4 no barrier, 8 no fragment reads, 16 no MFMAs,
// 0x100 frame-tile-fastest tile order, 0x800 register-staged instead of LDS-direct loads, 0x1000
// generic Toeplitz decimator, 0x2000 no pair launch, 0x4000 bf16x3: staged kernel instead of the
// hop-periodic ones, 0x8000 one slab buffer, 0x20000 masked 192x256 slab tiles instead of narrow
// tiles, 0x40000 no epilogue, 0x80000 two K stages only, 0x100000 bf16x3: dense (unfolded) kernel,
// 0x200000 fold: chunks of clips; fold pre-pass: 0x40 no global stores, 0x80 no global loads, 0x200
// without the last bin; fused filterbank: 0x400 no walk, 0x10000 walk without stores, 0x400000 plain
// stores instead of atomics; 0x10000000 support-aware fp32 tiles: 32-row MFMA tiles (round 4) instead of 16-row.
// In the product library MISPEC_DBG() is the constant false (the branches compile away) and a
// non-zero `reserved` is rejected.
//
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <type_traits>
#include <cstring>
#include <cmath>
#include <mutex>
#include <thread>
#include <vector>

#include "mispec.h"
#include "mispec_internal.h"
#include "fft_core.h"

#ifdef MISPEC_ABLATE
#define MISPEC_DBG(p, bit) (((p).debug & (bit)) != 0)
#else
#define MISPEC_DBG(p, bit) (false)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// 4 consecutive floats with only element alignment guaranteed (hop / pad / clip length are
// arbitrary); the hardware handles dword-aligned 16-byte global loads.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// arithmetic of the split kernels (strip, fold): the operands' format
enum { FOLD_BF16X3 = 0, FOLD_F32 = 1, FOLD_F16X3 = 2 };

// (hi, lo) fp16 pairs of two floats (round to nearest even): for MISPEC_PREC_F16X3, whose operands are
// scaled by powers of two so that the pairs stay inside fp16's exponent range
__device__ __forceinline__ void f16_split2(float a, float b, unsigned &hi, unsigned &lo) {
  const f32x2 v = {a, b};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// A power-of-two scale from the largest magnitude m = f 2^e, f in [0.5, 1), of what it multiplies:
// 2^(top - e) puts the values below 2^top; exponents clamped so that scale and inverse are normal.
// (clamped so that 2^(15 - e) and 2^(e - 29 - headroom) are normal floats: finite inputs of any
// magnitude keep their scale; below 2^-96 the pairs lose bits gracefully)
__device__ __forceinline__ int absmax_exponent(float m) {
  int e = (int)((__float_as_uint(m) >> 23) & 0xff) - 126;
  return e < -96 ? -96 : (e > 127 ? 127 : e);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
constexpr int F16_TOP = 14;  // scaled operands of the strip kernel stay below 2^14
constexpr int ABSMAX_CHUNK = 16384;     // samples per workgroup of clip_absmax_kernel
constexpr int CLIP_ABSMAX_STRIDE = 32;  // unsigned words between the clips' absmax words (own cache lines)
__device__ __forceinline__ float clip_scale_of(unsigned absmax_bits) {
  return pow2f(F16_TOP - absmax_exponent(__uint_as_float(absmax_bits)));
}
__device__ __forceinline__ float clip_unscale_of(unsigned absmax_bits) {
  return pow2f(absmax_exponent(__uint_as_float(absmax_bits)) - F16_TOP);
}

constexpr int KC = MISPEC_SPLIT_KC;   // K depth of one LDS stage (mispec_internal.h: the split planes' row granule)
constexpr int LDT = 36;  // LDS row stride in floats: 16-B aligned rows, conflict-free ds_read_b128

// BMODE_PLANAR_T: planar operand, output stored rows-innermost (frame-major): the iSTFT frames
enum { BMODE_FRAMED = 0, BMODE_PLANAR = 1, BMODE_PLANAR_T = 2 };
enum { AMODE_ROWS = 0, AMODE_TOEPLITZ = 1 };
enum { STORE_FRAMES_INNER = 0, STORE_ROWS_INNER = 1 };
enum { EDGE_NONE = 0, EDGE_SPANS = 1, EDGE_FULL = 2 };

struct KParams {
  // B operand (signal / planar tensor)
  const float *x;
  long long x_clip_stride;
  long long x_k_stride;  // planar mode: distance between successive k
  int x_col_stride;      // planar mode: distance between successive columns (frames); 0 = 1
  int k_split;           // planar mode: k >= k_split reads element (k - k_split) of a second
  long long k_split_off; //   operand starting k_split_off elements further (0 = no split)
  int out_frame_stride;  // rows-innermost store: elements between successive frames
  const long long *k_offsets;  // planar mode: element offset of every k (overrides x_k_stride)
  int n_clips;
  int n_samples;
  int hop;
  int pad;
  int pad_mode;
  int n_frames;
  long long n_cols;  // n_clips * n_frames
  // padded edge spans (see EdgePlan)
  const float *edge;
  long long edge_clip_stride;
  int edge_mode;
  int n_left;   // frames t < n_left start before the signal
  int t_r0;     // frames t >= t_r0 run past the end of the signal
  int edge_ll;  // length of the left span (the right span starts there)
  // A operand
  const float *a_re;
  const float *a_im;
  long long a_row_stride;
  int n_bins;
  int K;
  const int *row_support;
  const float *row_scale;
  int toep_stride;
  int n_taps;
  // epilogue / output
  int epilogue;
  float im_sign;
  float eps;
  float power;
  float *out;
  long long out_clip_stride;
  long long out_row_stride;
  int out_fm;  // frame-major output (mispec_framed_gemm_args.out_frame_major): the FFT path's FM instances only
  int out_row_offset;
  int out_len;
  int n_tiles_m;
  int n_tiles_n;
  int n_group;  // frame tiles crossed with all row tiles before advancing (L2 blocking)
  int debug;    // ablation bits (benchmarking only), see framed_gemm_kernel
  // MISPEC_PREC_BF16X3 operands (framed_bf16x3.inl)
  const unsigned short *xs;  // hi plane of the split signal: one slot per clip, the padded clip
  long long xs_clip_stride;  // elements per clip slot (multiple of 64)
  long long xs_plane;        // hi -> lo plane distance, elements
  const unsigned short *as;  // split basis planes [re_hi | re_lo | im_hi | im_lo], each (n_bins, Ks)
  long long as_plane;
  int Ks;  // taps per split basis row (K rounded up to 32, zero filled)
  // hop-periodic K order (framed_bf16x3_slab.inl)
  int n_super;    // C = ceil(Ks / hop)
  int slab_rows;  // rows of one slab buffer (>= BN + 2*(C-1), multiple of 16)
  int slab_nbuf;  // 1 or 2 slab buffers
  int row_split;  // framed_bf16x3_narrow: workgroups per frame tile (each takes every row_split-th row tile)
  unsigned *job_counter;  // framed_bf16x3_strip: next job; zeroed by the signal split of the same call
  const unsigned short *afrag;  // framed_bf16x3_strip: the basis in fragment order, or NULL
  int split_f32;                // split_signal_kernel writes the padded clip in fp32 (fp32 strip kernel)
  // fused filterbank reduction (bf16x3_epilogue_fb): out[c, m, t] += sum_bin fb[m, bin] * |X|^power
  const float *fb;
  const int *fb_support;
  long long fb_row_stride;
  int n_fb;
  int fb_lds_floats;  // FFT route: LDS floats (after the kernel's own tables) for the packed band weights, 0: none
  int fft_row_step;   // FFT route: 512 / n_fft for the frames of n_fft = 256 (zero-extended to 512 samples), else 1
  const float *cmb_E;  // FFT route, n_fft = 4096 composite: the even samples' spectrum (n_clips, 1025, n_frames, 2) -- the Complex
                       // 2048-point instance then stores X[k] = E[k] + W^k O[k] and X[2048 - k] instead of its own tile (stft_fft.inl)
  // symmetric fold (framed_fold.inl): as = folded basis, xs = folded frames, Ks = folded taps
  const float *fold_last;  // fp32 folded (even | odd) rows of the bin the pre-pass evaluates, or NULL
  int fold_last_bin;       // that bin, relative to the problem's first bin
  int fold_tap0;           // tap 0 is carried as folded tap kernel/2
  int fold_clip0;          // pre-pass launch: first clip; contraction launch: first frame tile
  int fold_tile0;
  int fold_f32;                // MISPEC_PREC_F32: fp32 stage rows ([re | im], [E | O])
  int fold_main;               // workgroups on 256-frame tiles; the rest take 128-frame tiles
  long long fold_tail_frame0;  //   from this flat frame on
  // second fold (framed_fold2.inl): even bins contract (Ep, Om), odd bins (Em, Op) over kernel/4 taps
  int fold2;                 // contraction launch: row tiles >= fold2_tiles_e belong to the odd bins
  int fold2_tiles_e;
  int fold2_bins_e;          // even / odd bins the contraction takes
  int fold2_bins_o;
  long long fold2_as_odd;    // elements from the even bins' folded rows to the odd bins'
  long long fold2_xs_odd;    // the same for the folded frames
  int fold_arith;            // FOLD_BF16X3 / FOLD_F32 / FOLD_F16X3: format of the folded operands
  float fold_wmax;           // FOLD_F16X3: max |window| (bounds the folded samples of a frame)
  float *col_unscale;        // FOLD_F16X3: per flat frame, what undoes the operand scaling (pre-pass writes, contraction reads)
  float *col_add;            // second fold: per flat frame, tap 0's term y[0] + y[M] (even bins; odd bins: n_cols further, y[0] - y[M])
  // MISPEC_PREC_F16X3 on the strip kernel
  unsigned *clip_absmax;     // per clip: bit pattern of max |sample| (clip_absmax_kernel; atomicMax)
  const float *row_unscale;  // per bin: inverse of the power of two its basis row was multiplied with
  int split_f16;             // split_signal_kernel writes scaled fp16 pairs
};

// ---------------------------------------------------------------------------------
// Which frames of a clip are not plain runs of the waveform, and where their padded
// copies live.  A frame is "interior" when [pos, pos + Kr) lies inside [0, n_samples),
// Kr = K rounded up to a whole LDS stage (so even the K tail stage reads in bounds).
//   EDGE_SPANS: per clip [left span | right span]; left span = positions -pad .. of the
//               first n_left frames, right span = positions of frames t_r0 .. T-1
//   EDGE_FULL : short clips whose left and right edge frames overlap: the whole virtually
//               padded clip (what the reference materialises for every clip)
// ---------------------------------------------------------------------------------
struct EdgePlan {
  int mode;
  int n_left;
  int t_r0;
  long long ll;      // floats in the left span
  long long lr;      // floats in the right span
  long long stride;  // floats per clip
};

inline int round_up_kc(int k) { return mispec_split_row_taps(k); }

EdgePlan plan_edges(int n_samples, int kernel, int hop, int pad, int n_frames) {
  EdgePlan e{};
  const long long kr = round_up_kc(kernel);
  const long long T = n_frames;
  long long n_left = pad > 0 ? ((long long)pad + hop - 1) / hop : 0;
  if (n_left > T) n_left = T;
  const long long lim = (long long)n_samples - kr + pad;  // pos_t + kr <= L  <=>  t*hop <= lim
  long long t_r0 = lim >= 0 ? lim / hop + 1 : 0;
  if (t_r0 > T) t_r0 = T;
  if (n_left == 0 && t_r0 == T) {
    e.mode = EDGE_NONE;
    e.n_left = 0;
    e.t_r0 = (int)T;
    return e;
  }
  const long long full = (T - 1) * hop + kr;
  e.mode = EDGE_SPANS;
  e.n_left = (int)n_left;
  e.t_r0 = (int)t_r0;
  e.ll = n_left > 0 ? (n_left - 1) * hop + kr : 0;
  e.lr = t_r0 < T ? (T - 1 - t_r0) * hop + kr : 0;
  e.stride = e.ll + e.lr;
  if (t_r0 < n_left || e.stride >= full) {
    // overlapping edge frames, or spans larger than the padded clip itself
    e.mode = EDGE_FULL;
    e.n_left = (int)T;
    e.t_r0 = (int)T;
    e.ll = full;
    e.lr = 0;
    e.stride = full;
  }
  return e;
}

// ---------------------------------------------------------------------------------
// sample fetch with virtual padding (reflect = nn.ReflectionPad1d: no edge repeat)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float fetch_sample(const float *__restrict__ x, long long base, int pp,
                                              int L, int pad_mode, bool ok) {
  if (pad_mode == MISPEC_PAD_REFLECT) {
    pp = pp < 0 ? -pp : pp;
    pp = pp >= L ? 2 * L - 2 - pp : pp;
  }
  ok = ok && (pp >= 0) && (pp < L);
  float v = 0.f;
  if (ok) v = x[base + pp];
  return v;
}

// Pre-pass: materialise the padded edge spans of every clip (a few frames' worth of samples).
__global__ void __launch_bounds__(256) edge_fill_kernel(const KParams p, float *__restrict__ ws) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= p.edge_clip_stride) return;
  long long q;  // signal position of workspace element i
  if (p.edge_mode == EDGE_FULL || i < p.edge_ll)
    q = i - p.pad;
  else
    q = (long long)p.t_r0 * p.hop - p.pad + (i - p.edge_ll);
  float v = 0.f;
  // positions beyond the virtually padded signal only ever meet zero taps / unused columns
  if (q >= -(long long)p.pad && q < (long long)p.n_samples + p.pad)
    v = fetch_sample(p.x, (long long)c * p.x_clip_stride, (int)q, p.n_samples, p.pad_mode, true);
  ws[(long long)c * p.edge_clip_stride + i] = v;
}

// address of sample position 0 of frame (c, t): a plain run of >= Kr floats
__device__ __forceinline__ const float *frame_ptr(const KParams &p, int c, int t) {
  if (p.edge_mode != EDGE_NONE) {
    const float *e = p.edge + (long long)c * p.edge_clip_stride;
    if (t < p.n_left) return e + (long long)t * p.hop;
    if (t >= p.t_r0) return e + p.edge_ll + (long long)(t - p.t_r0) * p.hop;
  }
  return p.x + (long long)c * p.x_clip_stride + ((long long)t * p.hop - p.pad);
}

// ---------------------------------------------------------------------------------
// pointwise epilogue on one (bin, frame) pair, shared by the MFMA and the reference kernel
// ---------------------------------------------------------------------------------
template <typename P>
__device__ __forceinline__ void epilogue_store(const P &p, float *__restrict__ dst, float re, float im) {
  switch (p.epilogue) {
    case MISPEC_EPI_COMPLEX: {
      float2 v = make_float2(re, im);
      *reinterpret_cast<float2 *>(dst) = v;
    } break;
    case MISPEC_EPI_MAGNITUDE:
      dst[0] = sqrtf(re * re + im * im + p.eps);
      break;
    case MISPEC_EPI_POWER: {
      float s = re * re + im * im + p.eps;
      float r;
      if (p.power == 2.0f && p.eps == 0.f)
        r = s;
      else if (p.power == 1.0f)
        r = sqrtf(s);
      else
        r = powf(sqrtf(s), p.power);
      dst[0] = r;
    } break;
    case MISPEC_EPI_PHASE_ATAN2:
      dst[0] = atan2f(im + 0.0f, re);
      break;
    case MISPEC_EPI_PHASE_COSSIN: {
      float a = atan2f(im, re);
      float2 v = make_float2(cosf(a), sinf(a));
      *reinterpret_cast<float2 *>(dst) = v;
    } break;
    default:
      dst[0] = re;
      break;
  }
}

__device__ __forceinline__ int epilogue_width(int epi) {
  return (epi == MISPEC_EPI_COMPLEX || epi == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
}
inline int epilogue_width_host(int epi) {
  return (epi == MISPEC_EPI_COMPLEX || epi == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
}

// Barrier that publishes LDS-direct (global_load_lds) data.  These loads complete asynchronously
// under vmcnt, and the wait is stated here instead of being left to the compiler's tracking of
// LDS-DMA writes against later ds_reads: hipcc emitted the K loop's barrier of one variant of
// the bf16x3 kernel with `s_waitcnt lgkmcnt(0)` only (the variant with both first stages in
// flight before the first barrier, experiments/README.md) and that build read stale LDS.
__device__ __forceinline__ void lds_dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---------------------------------------------------------------------------------
// Fused filterbank reduction, second half (shared by the fp32 and bf16x3 epilogues): the
// workgroup's |X|^power tile P[BB bins][BN frames] (row stride BN + 4 floats) is in LDS; thread
// (filter m, frame quad) walks the bins of m's band that lie in this tile,
//   out[c, m, t] = / += sum_bin fb[m, bin] * P[bin][t],
// and stores its four frames -- or, when the band crosses the tile boundary, adds them to the
// zeroed output with hardware float atomics (device-scope float atomics are resolved in memory,
// not in the XCD's L2: kept for the few filters that need them; a band narrower than a tile
// receives at most two addends, so the sum does not depend on their order).
// b0 = first bin of the tile relative to this problem's first bin (absolute: + out_row_offset).
// ---------------------------------------------------------------------------------
template <int BB, int BN, int NT>
__device__ __forceinline__ void filterbank_from_tile(const KParams &p, float *P, const int b0,
                                                     const long long n0) {
  constexpr int RS = BN + 4;
  constexpr int QPR = BN / 4;  // frame quads per row = threads per filter
  static_assert(NT % QPR == 0, "a thread keeps its frame quad over all its filters");
  const int tid = threadIdx.x;
  const int bin_first = p.out_row_offset + b0;
  int bins_here = p.n_bins - b0;
  bins_here = bins_here < BB ? bins_here : BB;
  int2 *const sBand = reinterpret_cast<int2 *>(P + BB * RS);
  for (int m = tid; m < p.n_fb; m += NT) {
    int lo = p.fb_support[2 * m] - bin_first, hi = p.fb_support[2 * m + 1] - bin_first;
    const int whole = (lo >= 0 && hi <= bins_here) ? 0x10000 : 0;  // band inside this tile
    lo = lo < 0 ? 0 : lo;
    hi = hi > bins_here ? bins_here : hi;
    sBand[m] = make_int2(lo | whole, hi);
  }
  __syncthreads();
  const int fq = tid % QPR;
  const long long col = n0 + 4 * fq;
  int cc[4], tt[4];
  {
    const long long c0 = col < p.n_cols ? col : 0;
    int c = (int)(c0 / p.n_frames);
    int t = (int)(c0 - (long long)c * p.n_frames);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      cc[i] = c;
      tt[i] = t;
      if (++t >= p.n_frames) {
        t = 0;
        ++c;
      }
    }
  }
  if (MISPEC_DBG(p, 0x400)) return;  // benchmarking build: no walk
#pragma unroll 1
  for (int m = tid / QPR; m < p.n_fb; m += NT / QPR) {
    const int2 band = sBand[m];
    const int lo = band.x & 0xffff, hi = band.y;
    const bool whole = (band.x & 0x10000) != 0;
    if (lo >= hi) continue;
    const float *w = p.fb + (long long)m * p.fb_row_stride + bin_first;
    f32x4v sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int b = lo; b < hi; b += 4) {  // four bins per trip (independent loads); the bins past
      f32x4v q[4];                      // the band get weight 0
      float wb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int bu = b + u < hi ? b + u : hi - 1;
        q[u] = *reinterpret_cast<const f32x4v *>(P + bu * RS + 4 * fq);
        wb[u] = b + u < hi ? w[bu] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) sum[i] += wb[u] * q[u][i];
    }
    float *const orow = p.out + (long long)m * p.out_row_stride;
    if (MISPEC_DBG(p, 0x10000) && sum[0] != 12345.678f) continue;  // benchmarking build: walk, no stores
    if (whole || MISPEC_DBG(p, 0x400000)) {  // (0x400000: plain stores instead of atomics)
      if (col + 3 < p.n_cols && cc[3] == cc[0]) {
        *reinterpret_cast<f32x4u *>(orow + (long long)cc[0] * p.out_clip_stride + tt[0]) = sum;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (col + i < p.n_cols) orow[(long long)cc[i] * p.out_clip_stride + tt[i]] = sum[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (col + i < p.n_cols) unsafeAtomicAdd(orow + (long long)cc[i] * p.out_clip_stride + tt[i], sum[i]);
    }
  }
}

// ---------------------------------------------------------------------------------
// MFMA kernel.  Workgroup = WM x WN waves (256 threads); each wave owns MR x NR tiles of 32x32.
//   BM = WM*MR*32 basis rows,  BN = WN*NR*32 frames per workgroup.
// Loader geometry: thread (r32 = tid>>3, c4 = tid&7) moves the 4 consecutive K elements
// 4*c4.. of row r32 of each 32-row "pass"; a pass of the A tile is one 32-row MFMA tile, a
// pass of the B tile is 32 consecutive frames.  Every thread keeps one source pointer per pass.
//
// MISPEC_DBG bits: see the file header (benchmarking build only).
// ---------------------------------------------------------------------------------
//
// T16 (round 5; support-aware banks, every wave owning all row tiles of 32 frames: the CQT1992v2 bank in fp32): the
// same stages multiplied as 16 x 16 x 4 MFMA tiles, so that a K stage is skipped per 16 basis rows = 8 complex bins
// instead of 16.  The CQT84 bank's supports shrink by 2^(1/12) per bin: 16-bin tiles execute 1.49 x the useful
// products, 8-bin tiles 1.22 x; MFMA rate and LDS fragment traffic per product are those of the 32 x 32 x 2 tiles.
template <int WM, int WN, int MR, int NR, int BMODE, int AMODE, bool MASKED, bool GLDS, bool T16 = false>
__device__ __forceinline__ void framed_gemm_body(const KParams &p, const int wg_index,
                                                 const int wg_count) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int TR = T16 ? 16 : 32;           // rows of a support tile
  constexpr int MT = WM * MR * (T16 ? 2 : 1);  // support tiles of the workgroup
  static_assert(!T16 || (GLDS && MASKED && WM == 1 && NR == 1 && BMODE == BMODE_FRAMED && AMODE == AMODE_ROWS && MR <= 8),
                "16-row tiles: the support-aware LDS-direct instance");
  static_assert(NT == 256, "loader geometry assumes 256 threads");
  constexpr int APASS = BM / 32;
  constexpr int BPASS = BN / 32;  // framed mode
  static_assert(!GLDS || (BMODE == BMODE_FRAMED && AMODE == AMODE_ROWS), "LDS-direct loads: framed rows only");
  // LDS row stride in floats.  Register-staged path: 36 (padding keeps ds_read_b128 conflict
  // free).  LDS-direct path: the DMA writes lane-linear 16-byte pieces, so rows are the bare
  // 32 floats and the 16-byte chunks of a row are XOR-swizzled by (row >> 1) & 7 instead
  // (applied to the per-lane SOURCE address and again on the fragment reads).
  constexpr int LROW = GLDS ? KC : LDT;
  constexpr int A_STAGE = BM * LROW;
  constexpr int B_STAGE = (BMODE == BMODE_FRAMED) ? BN * LROW : KC * BN;
  // planar loader: a thread moves quads of 4 consecutive columns (one 16-byte load when the
  // four columns are adjacent in memory, four scalar loads otherwise, decided per thread once);
  // QPR quads per K row, RPP K rows per pass, VPASS passes per stage
  constexpr int QPR = BN / 4;
  constexpr int RPP = NT / QPR;
  constexpr int VPASS = KC / RPP;
  static_assert(BMODE == BMODE_FRAMED || (NT % QPR == 0 && KC % RPP == 0 && VPASS >= 1), "planar loader shape");
  constexpr int STORE_MODE = (AMODE == AMODE_TOEPLITZ || BMODE == BMODE_PLANAR_T) ? STORE_ROWS_INNER
                                                                               : STORE_FRAMES_INNER;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *sA = reinterpret_cast<float *>(smem_raw);
  float *sB = sA + 2 * A_STAGE;
  const float **sColPtr = reinterpret_cast<const float **>(sB + 2 * B_STAGE);  // [BN]
  int *sTileLo = reinterpret_cast<int *>(sColPtr + BN);
  int *sTileHi = sTileLo + MT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int r32 = tid >> 3;  // loader row inside a pass
  const int c4 = tid & 7;    // LDS 16-byte chunk this thread fills in its row
  // ... which holds this chunk of the row's K stage (identity unless swizzled)
  const int cg = GLDS ? (c4 ^ ((r32 >> 1) & 7)) : c4;

  // ---- XCD-aware tile order.  Workgroup b runs on XCD b % 8 (observed; only speed depends on
  // it) and every XCD has a private 4 MiB L2.  Each XCD gets a contiguous range of a linear
  // tile order in which `n_group` consecutive frame tiles are crossed with all row tiles.
  int tile;
  {
    const int nwg = wg_count;
    const int b = wg_index;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tile_m, tile_n;
  {
    const int G = p.n_group;
    const int per_group = G * p.n_tiles_m;
    const int full = (p.n_tiles_n / G) * per_group;
    if (tile < full) {
      const int g = tile / per_group;
      const int rest = tile - g * per_group;
      tile_m = rest / G;
      tile_n = g * G + (rest - tile_m * G);
    } else {
      const int Gt = p.n_tiles_n % G;  // > 0 here
      const int rest = tile - full;
      tile_m = rest / Gt;
      tile_n = (p.n_tiles_n / G) * G + (rest - tile_m * Gt);
    }
  }
  const int m0 = tile_m * BM;
  const long long n0 = (long long)tile_n * BN;

  const bool cplx = (AMODE == AMODE_ROWS) && p.a_im != nullptr;
  const int rpb = cplx ? 2 : 1;

  // ---- per-column source pointers and per-row-tile K ranges
  for (int j = tid; j < BN; j += NT) {
    long long col = n0 + j;
    if (col >= p.n_cols) col = 0;  // unused column: any valid frame, its results are not stored
    const int c = (int)(col / p.n_frames);
    const int t = (int)(col - (long long)c * p.n_frames);
    sColPtr[j] = (BMODE == BMODE_FRAMED)
                     ? frame_ptr(p, c, t)
                     : p.x + (long long)c * p.x_clip_stride + (long long)t * (p.x_col_stride ? p.x_col_stride : 1);
  }
  if (tid < MT) {
    const int row_lo = m0 + tid * TR;
    int lo = 0, hi = 0;
    if (AMODE == AMODE_TOEPLITZ) {
      if (row_lo < p.n_bins) hi = p.K;
    } else {
      const int bin_lo = row_lo / rpb;
      int bin_hi = (row_lo + TR + rpb - 1) / rpb;
      bin_hi = bin_hi < p.n_bins ? bin_hi : p.n_bins;
      if (bin_lo < bin_hi) {
        if (p.row_support) {
          lo = p.K;
          hi = 0;
          for (int b = bin_lo; b < bin_hi; ++b) {
            const int s = p.row_support[2 * b], e = p.row_support[2 * b + 1];
            if (e > s) {
              lo = s < lo ? s : lo;
              hi = e > hi ? e : hi;
            }
          }
          lo = lo < 0 ? 0 : lo;
          hi = hi > p.K ? p.K : hi;
          if (hi <= lo) lo = hi = 0;
        } else {
          hi = p.K;
        }
      }
    }
    sTileLo[tid] = lo;
    sTileHi[tid] = hi;
  }
  __syncthreads();

  // row-tile K ranges -> scalar registers (read once; the per-stage masks are SALU work)
  int tlo[MT], thi[MT];
  int kb = p.K, ke = 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    tlo[i] = __builtin_amdgcn_readfirstlane(sTileLo[i]);
    thi[i] = __builtin_amdgcn_readfirstlane(sTileHi[i]);
    if (thi[i] > tlo[i]) {
      kb = tlo[i] < kb ? tlo[i] : kb;
      ke = thi[i] > ke ? thi[i] : ke;
    }
  }
  kb = kb & ~(KC - 1);
  const int nstages = ke > kb ? (ke - kb + KC - 1) / KC : 0;
  // stages whose whole K range lies inside the basis rows can use 16-byte A loads; the one
  // possible K-tail stage (K % 32 != 0) is peeled out of the pipelined loop
  const bool a_tail = (AMODE == AMODE_ROWS) && nstages > 0 && (kb + nstages * KC > p.K);
  const int nloop = a_tail ? nstages - 1 : nstages;

  // which of the workgroup's row tiles intersect K stage [kc, kc+KC)
  auto stage_mask = [&](int kc) __attribute__((always_inline)) -> unsigned {
    if (!MASKED) return (1u << MT) - 1u;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      if (thi[i] > kc && tlo[i] < kc + KC) m |= 1u << i;
    return m;
  };

  // ---- per-thread source pointers (one per pass), computed once
  const float *aptr[APASS];
  if (AMODE == AMODE_ROWS) {
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      const int row = m0 + ps * 32 + r32;
      int bin = cplx ? (row >> 1) : row;
      bin = bin < p.n_bins ? bin : p.n_bins - 1;  // rows past the end feed unused accumulators
      const float *src = (cplx && (row & 1)) ? p.a_im : p.a_re;
      aptr[ps] = src + (long long)bin * p.a_row_stride + 4 * cg;
    }
  }
  const float *bptr[(BMODE == BMODE_FRAMED) ? BPASS : 4];
  const int pq = 4 * (tid % QPR);  // planar: first column of this thread's quad
  const int pk = tid / QPR;        // planar: its K row inside a pass
  bool quad = false;               // planar: the quad's columns are adjacent in memory
  if (BMODE == BMODE_FRAMED) {
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) bptr[ps] = sColPtr[ps * 32 + r32] + 4 * cg;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) bptr[e] = sColPtr[pq + e];
    quad = bptr[1] == bptr[0] + 1 && bptr[2] == bptr[0] + 2 && bptr[3] == bptr[0] + 3;
  }

  f32x4v ra[APASS];
  f32x4v rb[(BMODE == BMODE_FRAMED) ? BPASS : VPASS];
  unsigned toep_bits = 0;  // Toeplitz A: which of the 4*APASS loaded taps are inside the band

  // ---- stage loads: no control flow, every address is valid memory by construction.
  // NA = number of leading A passes (row tiles) a loop instance moves and multiplies.
  auto load_b = [&](int kc) __attribute__((always_inline)) {
    if (BMODE == BMODE_FRAMED) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps)
        rb[ps] = *reinterpret_cast<const f32x4u *>(bptr[ps] + kc);
    } else {
#pragma unroll
      for (int ps = 0; ps < VPASS; ++ps) {
        int kk = kc + ps * RPP + pk;
        kk = kk < p.K ? kk : p.K - 1;  // K tail: any finite value, the A side is zero there
        long long ko = (long long)kk * p.x_k_stride;
        if (p.k_split && kk >= p.k_split) ko = (long long)(kk - p.k_split) * p.x_k_stride + p.k_split_off;
        if (p.k_offsets) ko = p.k_offsets[kk];
        if (quad) {
          rb[ps] = *reinterpret_cast<const f32x4u *>(bptr[0] + ko);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) rb[ps][e] = bptr[e][ko];
        }
      }
    }
  };
  auto load_a = [&](int kc, auto na_tag) __attribute__((always_inline)) {
    constexpr int NA = decltype(na_tag)::value;
    if (AMODE == AMODE_ROWS) {
#pragma unroll
      for (int ps = 0; ps < NA; ++ps)
        ra[ps] = *reinterpret_cast<const f32x4u *>(aptr[ps] + kc);
    } else {
      // banded Toeplitz matrix of the FIR taps: A[row, k] = taps[k - stride*row]
      const int k = kc + 4 * c4;
      unsigned bits = 0;
#pragma unroll
      for (int ps = 0; ps < NA; ++ps) {
        const int row = m0 + ps * 32 + r32;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int tap = k + e - p.toep_stride * row;
          const bool ok = tap >= 0 && tap < p.n_taps && row < p.n_bins && (k + e) < p.K;
          ra[ps][e] = p.a_re[ok ? tap : 0];
          bits |= (ok ? 1u : 0u) << (4 * ps + e);
        }
      }
      toep_bits = bits;
    }
  };
  auto store_stage = [&](int buf, auto na_tag) __attribute__((always_inline)) {
    constexpr int NA = decltype(na_tag)::value;
    float *a = sA + buf * A_STAGE;
    float *b = sB + buf * B_STAGE;
    if (AMODE == AMODE_TOEPLITZ) {
#pragma unroll
      for (int ps = 0; ps < NA; ++ps)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          ra[ps][e] = ((toep_bits >> (4 * ps + e)) & 1u) ? ra[ps][e] : 0.f;
    }
#pragma unroll
    for (int ps = 0; ps < NA; ++ps)
      *reinterpret_cast<f32x4v *>(a + (ps * 32 + r32) * LROW + 4 * c4) = ra[ps];
    if (BMODE == BMODE_FRAMED) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps)
        *reinterpret_cast<f32x4v *>(b + (ps * 32 + r32) * LROW + 4 * c4) = rb[ps];
    } else {
#pragma unroll
      for (int ps = 0; ps < VPASS; ++ps)
        *reinterpret_cast<f32x4v *>(b + (ps * RPP + pk) * BN + pq) = rb[ps];
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  // T16: wave tile = 2*MR x 2 tiles of 16 x 16; element e of lane (l16, lq) is D[row 4*lq + e][col l16]
  typedef float f32x4a __attribute__((ext_vector_type(4)));
  constexpr int MR16 = T16 ? 2 * MR : 1;
  f32x4a acc16[MR16][2];
  if (T16) {
#pragma unroll
    for (int m = 0; m < MR16; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[m][h][e] = 0.f;
  }
  // One K stage as 16 x 16 x 4 tiles: 2 K groups of 16 taps.  v_mfma_f32_16x16x4_f32 IS an fp32 FMA chain through its four
  // k lanes in ascending order (experiments/tap_order/mfma_order.hip: bit-identical on 51 200 random results; a single
  // rounding of the exact sum matches 67 %), so with lane (l16, lq) supplying tap 16q + 4s + lq of basis row / frame l16
  // to MFMA s, an accumulator is ONE float32 FMA chain over the taps in ascending order: the arithmetic of the
  // reference's conv1d (its fixture's near-silent bins record that chain's rounding: tests/test_reference_order.py).
  // The four taps of a lane are four ds_read_b32 (chunk 4q + s of the LDS-direct layout, swizzle undone; conflict free:
  // the 64 lanes of a read cover 64 banks once).  MRA = leading 16-row tiles multiplied; the fragments of group q + 1 are
  // read under the MFMAs of group q.
  int coff[KC / 4];  // dword offset of tap chunk c in this lane's rows
  {
    const int fsw = ((lane & 15) >> 1) & 7;
#pragma unroll
    for (int c = 0; c < KC / 4; ++c) coff[c] = 4 * (c ^ fsw) + (lane >> 4);
  }
  auto mfma_stage16 = [&](int buf, unsigned mask, auto mra_tag, auto use_mask_tag) __attribute__((always_inline)) {
    constexpr int MRA = decltype(mra_tag)::value;
    constexpr bool USE_MASK = decltype(use_mask_tag)::value;
    const float *a_base = sA + buf * A_STAGE + (lane & 15) * LROW;
    const float *b_base = sB + buf * B_STAGE + (wn * 32 + (lane & 15)) * LROW;
    float av[2][MRA][4], bv[2][2][4];
    auto load_frags = [&](int q, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int off = coff[4 * q + s4];
#pragma unroll
        for (int m = 0; m < MRA; ++m) av[slot][m][s4] = a_base[m * 16 * LROW + off];
#pragma unroll
        for (int h = 0; h < 2; ++h) bv[slot][h][s4] = b_base[h * 16 * LROW + off];
      }
    };
    load_frags(0, 0);
#pragma unroll
    for (int q = 0; q < KC / 16; ++q) {
      if (q + 1 < KC / 16) load_frags(q + 1, (q + 1) & 1);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int m = 0; m < MRA; ++m) {
          if ((!USE_MASK || ((mask >> m) & 1u)) && !MISPEC_DBG(p, 16)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
              acc16[m][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][m][s4], bv[q & 1][h][s4], acc16[m][h], 0, 0, 0);
          }
        }
      }
    }
  };

  // One K stage out of LDS buffer `buf`: 4 K groups of 8; lane (li, lh) supplies k = 8q+4lh+s to
  // MFMA s of group q.  `mid0` / `mid1` run after groups 0 / 1 (LDS stores of the next stage and
  // global loads of the one after), i.e. inside the MFMA stream.  MRA = number of this wave's
  // leading row tiles that are multiplied (compile time); `mask` skips tiles at run time in the
  // generic instance.
  auto mfma_stage = [&](int buf, unsigned mask, bool frags, auto mra_tag, auto use_mask_tag,
                        auto &&mid0, auto &&mid1) __attribute__((always_inline)) {
    constexpr int MRA = decltype(mra_tag)::value;
    constexpr bool USE_MASK = decltype(use_mask_tag)::value;
    const float *a_base = sA + buf * A_STAGE + ((wm * MR) * 32 + li) * LROW + (GLDS ? 0 : 4 * lh);
    const float *b_base;
    if (BMODE == BMODE_FRAMED)
      b_base = sB + buf * B_STAGE + ((wn * NR) * 32 + li) * LROW + (GLDS ? 0 : 4 * lh);
    else
      b_base = sB + buf * B_STAGE + (4 * lh) * BN + (wn * NR) * 32 + li;
    const unsigned wmask = USE_MASK ? (mask >> (wm * MR)) : ~0u;
    const int fsw = (li >> 1) & 7;  // GLDS: chunk swizzle of this lane's rows (tile rows = li + 32k)
    f32x4v av[2][MR], bv[2][NR];
    auto load_frags = [&](int q, int slot) __attribute__((always_inline)) {
      const int off = GLDS ? 4 * ((2 * q + lh) ^ fsw) : 8 * q;
#pragma unroll
      for (int m = 0; m < MRA; ++m)
        av[slot][m] = *reinterpret_cast<const f32x4v *>(a_base + m * 32 * LROW + off);
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        if (BMODE == BMODE_FRAMED) {
          bv[slot][n] = *reinterpret_cast<const f32x4v *>(b_base + n * 32 * LROW + off);
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s) bv[slot][n][s] = b_base[(8 * q + s) * BN + n * 32];
        }
      }
    };
    if (frags) load_frags(0, 0);
#pragma unroll
    for (int q = 0; q < KC / 8; ++q) {
      if (q + 1 < KC / 8 && frags) load_frags(q + 1, (q + 1) & 1);  // prefetch under the MFMAs
#pragma unroll
      for (int m = 0; m < MRA; ++m) {
        if ((!USE_MASK || ((wmask >> m) & 1u)) && !MISPEC_DBG(p, 16)) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int n = 0; n < NR; ++n)
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][m][s], bv[q & 1][n][s],
                                                               acc[m][n], 0, 0, 0);
          }
        }
      }
      if (q == 0) mid0();
      if (q == 1) mid1();
    }
  };

  // ---- software pipeline over full stages [c0, c1) (stage c covers K [kb + c*KC, +KC)):
  //   LDS holds stage c (being multiplied) and receives stage c+1 during the same iteration;
  //   registers hold stage c+1 until it is written, then receive stage c+2.
  // NA: leading row tiles of the workgroup this instance moves through LDS; MRA: leading row
  // tiles of each wave it multiplies; USE_MASK: consult the per-stage mask at run time.
  auto run_stages = [&](int c0, int c1, auto na_tag, auto mra_tag,
                        auto use_mask_tag) __attribute__((always_inline)) {
    if (c1 <= c0) return;
    if (GLDS) {
      // LDS-direct: stage c+1 is DMA'd into the other buffer while stage c is multiplied; the
      // barrier at the end of the iteration (which carries the vmcnt(0)) publishes it.
      constexpr int NA = T16 ? (decltype(na_tag)::value + 1) / 2 : decltype(na_tag)::value;  // 32-row passes moved
      auto dma_stage = [&](int c, int buf) __attribute__((always_inline)) {
        const int kc = kb + c * KC;
        typedef __attribute__((address_space(1))) const void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
#pragma unroll
        for (int ps = 0; ps < NA; ++ps)
          __builtin_amdgcn_global_load_lds((gptr_t)(aptr[ps] + kc),
                                           (lptr_t)(sA + buf * A_STAGE + (ps * 32 + wave * 8) * LROW),
                                           16, 0, 0);
#pragma unroll
        for (int ps = 0; ps < BPASS; ++ps)
          __builtin_amdgcn_global_load_lds((gptr_t)(bptr[ps] + kc),
                                           (lptr_t)(sB + buf * B_STAGE + (ps * 32 + wave * 8) * LROW),
                                           16, 0, 0);
      };
      dma_stage(c0, 0);
      lds_dma_barrier();
      for (int c = c0; c < c1; ++c) {
        const int buf = (c - c0) & 1;
        if ((c + 1) < c1 && !MISPEC_DBG(p, 1)) dma_stage(c + 1, buf ^ 1);
        if constexpr (T16)
          mfma_stage16(buf, stage_mask(kb + c * KC), mra_tag, use_mask_tag);
        else
          mfma_stage(
              buf, stage_mask(kb + c * KC), !MISPEC_DBG(p, 8) || c == c0, mra_tag, use_mask_tag,
              [&]() __attribute__((always_inline)) {}, [&]() __attribute__((always_inline)) {});
        if (!MISPEC_DBG(p, 4)) lds_dma_barrier();
      }
      return;
    }
    if constexpr (!GLDS) {
    load_a(kb + c0 * KC, na_tag);
    load_b(kb + c0 * KC);
    store_stage(0, na_tag);
    if (c1 - c0 > 1) {
      load_a(kb + (c0 + 1) * KC, na_tag);
      load_b(kb + (c0 + 1) * KC);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
      const int buf = (c - c0) & 1;
      const bool has1 = (c + 1) < c1;
      const bool has2 = (c + 2) < c1;
      mfma_stage(
          buf, stage_mask(kb + c * KC), !MISPEC_DBG(p, 8) || c == c0, mra_tag, use_mask_tag,
          [&]() __attribute__((always_inline)) {
            if (has1 && !MISPEC_DBG(p, 2)) store_stage(buf ^ 1, na_tag);  // stage c+1: regs -> LDS
          },
          [&]() __attribute__((always_inline)) {
            if (has2 && !MISPEC_DBG(p, 1)) {  // stage c+2: HBM/L2 -> registers
              load_a(kb + (c + 2) * KC, na_tag);
              load_b(kb + (c + 2) * KC);
            }
          });
      if (!MISPEC_DBG(p, 4)) __syncthreads();
    }
    }
  };
  using std::integral_constant;
  constexpr bool PHASED = MASKED && (WM == 1) && (AMODE == AMODE_ROWS);
  constexpr int MRT = T16 ? 2 * MR : MR;  // support tiles per wave (every wave owns all of them when PHASED)
  if (!PHASED) {
    run_stages(0, nloop, integral_constant<int, T16 ? MRT : APASS>{}, integral_constant<int, MRT>{},
               integral_constant<bool, MASKED>{});
  } else {
    // Support-aware contraction, every wave owns all MR row tiles (WM == 1).  Runs of stages in
    // which exactly the first n row tiles are active (CQT banks: supports are centred and
    // shrink with the bin index, so the active set is a prefix that grows then shrinks) use a
    // loop compiled for n tiles; anything else falls back to the masked instance.
    int c = 0;
    while (c < nloop) {
      const unsigned m = stage_mask(kb + c * KC);
      int c1 = c + 1;
      while (c1 < nloop && stage_mask(kb + c1 * KC) == m) ++c1;
      bool done = false;
      if (c1 - c >= 4) {  // short runs are not worth a pipeline restart
        // one loop instance per prefix length 1 .. MR-1 (instances beyond MR are never formed)
#define MISPEC_PREFIX_RUN(N)                                                               \
  if (!done && (N) < MRT && m == ((1u << (N)) - 1u)) {                                       \
    run_stages(c, c1, integral_constant<int, ((N) < MRT ? (N) : 1)>{},                       \
               integral_constant<int, ((N) < MRT ? (N) : 1)>{}, integral_constant<bool, false>{}); \
    done = true;                                                                           \
  }
        MISPEC_PREFIX_RUN(1)
        MISPEC_PREFIX_RUN(2)
        MISPEC_PREFIX_RUN(3)
        MISPEC_PREFIX_RUN(4)
        MISPEC_PREFIX_RUN(5)
        MISPEC_PREFIX_RUN(6)
        MISPEC_PREFIX_RUN(7)
#undef MISPEC_PREFIX_RUN
      }
      if (!done)
        run_stages(c, c1, integral_constant<int, T16 ? MRT : APASS>{}, integral_constant<int, MRT>{},
                   integral_constant<bool, true>{});
      c = c1;
    }
  }
  // ---- peeled K-tail stage (only when K is not a multiple of 32): element-wise clamped A loads
  if (a_tail) {
    const int kc = kb + nloop * KC;
    const int k = kc + 4 * cg;
    if (AMODE == AMODE_ROWS) {
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (k + e) < p.K;
          const float t = (aptr[ps] - 4 * cg)[ok ? k + e : 0];
          ra[ps][e] = ok ? t : 0.f;
        }
      }
    }
    load_b(kc);  // in bounds by construction (frames are runs of >= Kr floats)
    store_stage(0, std::integral_constant<int, APASS>{});
    __syncthreads();
    if constexpr (T16)
      mfma_stage16(0, stage_mask(kc), std::integral_constant<int, MRT>{}, std::integral_constant<bool, true>{});
    else
      mfma_stage(
          0, stage_mask(kc), true, std::integral_constant<int, MR>{},
          std::integral_constant<bool, MASKED>{}, [&]() __attribute__((always_inline)) {},
          [&]() __attribute__((always_inline)) {});
  }

  // ---- epilogue.  Accumulator element e of lane (li, lh) is D[row = (e&3) + 8*(e>>2) + 4*lh][col = li].
  // Each wave restages one 32x32 tile at a time through a private LDS patch so that the
  // pointwise epilogue below is a single dynamic loop (one code instance, static register
  // indexing only in the ds_write fan-out) and so that stores are contiguous along the
  // innermost output dimension for both store modes.
  __syncthreads();  // every wave is done with the K-stage buffers
  if (!T16 && BMODE == BMODE_FRAMED && AMODE == AMODE_ROWS && MR * NR <= 4 && p.fb) {  // (automatic tiles)
    // fused filterbank reduction (mel.py:184-189): the tile's |X|^power goes to LDS, bins x
    // frames, and filterbank_from_tile reduces it over the bins of every filter's band.  A lane
    // holds re and im of a bin in adjacent accumulator elements (interleaved rows).
    constexpr int RS = BN + 4;
    float *const P = reinterpret_cast<float *>(smem_raw);
    const bool sq = p.power == 2.0f;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const int rl = (wm * MR + m) * 32 + (2 * (e2 & 1)) + 8 * (e2 >> 1) + 4 * lh;  // even row
          const int bl = rl >> 1;
          const int bin = (m0 >> 1) + bl;
          const bool bin_ok = bin < p.n_bins;
          const float sc = (p.row_scale && bin_ok) ? p.row_scale[bin] : 1.f;
          const float re = acc[m][n][2 * e2] * sc, im = acc[m][n][2 * e2 + 1] * sc;
          const float s2 = re * re + im * im + p.eps;
          P[bl * RS + (wn * NR + n) * 32 + li] = bin_ok ? (sq ? s2 : sqrtf(s2)) : 0.f;
        }
    __syncthreads();
    filterbank_from_tile<BM / 2, BN, NT>(p, P, m0 >> 1, n0);
    return;
  }
  constexpr int LDC = 33;
  float *sC = reinterpret_cast<float *>(smem_raw) + wave * (32 * LDC);
  const int E = epilogue_width(p.epilogue);
#pragma unroll 1
  for (int ti = 0; ti < MR * NR; ++ti) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        if (ti == m * NR + n) {
          if constexpr (T16) {  // (NR == 1) four 16 x 16 tiles of the 32 x 32 patch
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
              for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  sC[(16 * half + 4 * (lane >> 4) + e) * LDC + 16 * h + (lane & 15)] = acc16[T16 ? 2 * m + half : 0][h][e];
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              sC[((e & 3) + 8 * (e >> 2) + 4 * lh) * LDC + li] = acc[m][n][e];
          }
        }
      }
    }
    __syncthreads();
    const int tm = ti / NR, tn = ti - tm * NR;
    const int row_base = m0 + (wm * MR + tm) * 32;        // first basis row of this tile
    const long long col_base = n0 + (wn * NR + tn) * 32;  // first frame column of this tile
    if (STORE_MODE == STORE_ROWS_INNER) {
      // lane = row (output sample within the 32-block), iterate over the tile's 32 frames
      const int row = row_base + li;
#pragma unroll 1
      for (int it = 0; it < 16; ++it) {
        const int cl = 2 * it + lh;
        const long long col = col_base + cl;
        if (col < p.n_cols && row < p.n_bins) {
          const int c = (int)(col / p.n_frames);
          const int t = (int)(col - (long long)c * p.n_frames);
          const long long o = (long long)t * p.out_frame_stride + row;
          float v = sC[li * LDC + cl];
          if (p.row_scale) v *= p.row_scale[row];
          if (o < p.out_len) p.out[(long long)c * p.out_clip_stride + o] = v;
        }
      }
    } else {
      // lane = frame (innermost output dimension), iterate over the tile's rows
      const long long col = col_base + li;
      const bool col_ok = col < p.n_cols;
      int c = 0, t = 0;
      if (col_ok) {
        c = (int)(col / p.n_frames);
        t = (int)(col - (long long)c * p.n_frames);
      }
      float *obase = p.out + (long long)c * p.out_clip_stride + (long long)t * E;
      if (cplx) {
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
          const int rl = 2 * (2 * it + lh);  // even local row: re; rl + 1: im
          const int bin = (row_base + rl) >> 1;
          if (col_ok && bin < p.n_bins) {
            float re = sC[rl * LDC + li];
            float im = p.im_sign * sC[(rl + 1) * LDC + li];
            if (p.row_scale) {
              const float s = p.row_scale[bin];
              re *= s;
              im *= s;
            }
            epilogue_store(p, obase + (long long)(p.out_row_offset + bin) * p.out_row_stride, re,
                           im);
          }
        }
      } else {
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
          const int rl = 2 * it + lh;
          const int row = row_base + rl;
          if (col_ok && row < p.n_bins) {
            float v = sC[rl * LDC + li];
            if (p.row_scale) v *= p.row_scale[row];
            obase[(long long)(p.out_row_offset + row) * p.out_row_stride] = v;
          }
        }
      }
    }
    __syncthreads();
  }
}

template <int WM, int WN, int MR, int NR, int BMODE, int AMODE, bool MASKED, bool GLDS, bool T16 = false>
__global__ void __launch_bounds__(WM *WN * 64) framed_gemm_kernel(const KParams p) {
  framed_gemm_body<WM, WN, MR, NR, BMODE, AMODE, MASKED, GLDS, T16>(p, blockIdx.x, gridDim.x);
}

#include "framed_bf16x3.inl"
#include "framed_bf16x3_slab.inl"
#include "framed_bf16x3_narrow.inl"
#include "framed_fold.inl"
#include "framed_fold2.inl"
#include "framed_bf16x3_strip.inl"
#include "octave_pyramid.inl"
#include "stft_fft.inl"

// Several independent contractions of the same tile shape in one launch (the octaves of
// CQT2010v2 / VQT: each is a short-K, few-hundred-workgroup problem that cannot fill the chip on
// its own).  Workgroups [first[i], first[i+1]) belong to problem i.
constexpr int GROUP_MAX = 8;
struct KGroup {
  int n;
  int first[GROUP_MAX + 1];
  KParams p[GROUP_MAX];
};

template <int WM, int WN, int MR, int NR, int BMODE, int AMODE, bool MASKED, bool GLDS>
__global__ void __launch_bounds__(WM *WN * 64) framed_gemm_group_kernel(const KGroup g) {
  int pid = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.first[i]) pid = i;
  pid = __builtin_amdgcn_readfirstlane(pid);
  framed_gemm_body<WM, WN, MR, NR, BMODE, AMODE, MASKED, GLDS>(
      g.p[pid], blockIdx.x - g.first[pid], g.first[pid + 1] - g.first[pid]);
}

// Two tile shapes in one launch: workgroups [0, n_main) run the dense 128x128 contraction over
// the rows that fill whole 128-row blocks, the rest run a narrow tile over the leftover rows
// (the Nyquist bin of an n_fft/2+1 STFT).  Appended at the end of the grid, the narrow
// workgroups fill the tail of the main grid instead of paying for a launch of their own.
template <int RMR>
__global__ void __launch_bounds__(256) framed_gemm_pair_kernel(const KParams pm, const KParams pr,
                                                               const int n_main) {
  if ((int)blockIdx.x < n_main)
    framed_gemm_body<2, 2, 2, 2, BMODE_FRAMED, AMODE_ROWS, false, true>(pm, blockIdx.x, n_main);
  else
    framed_gemm_body<1, 4, RMR, 2, BMODE_FRAMED, AMODE_ROWS, false, true>(
        pr, blockIdx.x - n_main, gridDim.x - n_main);
}

// ---------------------------------------------------------------------------------
// Dedicated stride-2 FIR decimator (the octave recursion of CQT2010v2 / VQT spends most of its
// time here).  Same Toeplitz contraction as the generic kernel,
//     y[32 q + r] = sum_m T[r, m] * x[64 q + m - pad],   T[r, m] = taps[m - 2 r],  m < n_taps + 62,
// but organised around what is constant and what is reused:
//   * the whole 32 x 320 Toeplitz matrix lives in registers as MFMA A-fragments (40 K groups x
//     4 floats per lane), built once per workgroup from a zero-guarded copy of the taps in LDS;
//   * the input span of the workgroup's outputs (2 * n_out_wg + 320 samples) is loaded into LDS
//     ONCE, so every input sample is read from HBM/L2 once instead of K'/hop' = 5 times;
//     rows of 64 samples are padded to 68 floats so the hop-64 fragment reads (ds_read_b128) are
//     bank-conflict free;
//   * the K loop is 40 groups x (NRW ds_read_b128 + 4 NRW MFMA) with no barrier and no global
//     traffic; two workgroups per CU overlap one's span load with the other's MFMAs.
// Each wave owns NRW 32x32 tiles = 1024 NRW consecutive outputs; a workgroup 4096 NRW outputs
// of one clip.
// ---------------------------------------------------------------------------------
#ifndef FIR_NRW
#define FIR_NRW 1
#endif
#ifndef FIR_AREG
#define FIR_AREG false
#endif
constexpr int FIR_KG = 40;                   // K groups of 8: n_taps + 62 <= 320
constexpr int FIR_ROW = 68;                  // LDS row: 64 samples + 4 pad
template <int NRW, bool AREG>
__global__ void __launch_bounds__(256) fir_decimate2_kernel(const KParams p) {
  constexpr int OUT_WG = 4096 * NRW;                 // outputs per workgroup
  constexpr int SPAN = 2 * OUT_WG + 64 * 5;          // input samples staged (incl. K' = 320 halo)
  constexpr int ROWS = SPAN / 64;                    // 64-sample rows
  constexpr int TAPZ = 62 + 320 + 2;                 // zero-guarded taps: index i + 62, i in [-62, 320)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *sX = reinterpret_cast<float *>(smem_raw);   // [ROWS][FIR_ROW]
  float *sT = sX + ROWS * FIR_ROW;                   // [TAPZ]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int c = blockIdx.y;
  const long long o0 = (long long)blockIdx.x * OUT_WG;        // first output of this workgroup
  const long long p0 = 2 * o0 - p.pad;                        // signal position of span element 0
  const float *xc = p.x + (long long)c * p.x_clip_stride;

  // ---- stage the span (zero outside the clip) and the zero-guarded taps.  All loads are issued
  // before the first LDS store (no control flow between them); the few 16-byte pieces that
  // straddle a clip edge are patched element-wise afterwards.
  constexpr int NLD = (SPAN / 4 + 255) / 256;
  f32x4v stage[NLD];
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int i = tid + 256 * it;
    const long long q = p0 + 4LL * i;
    const bool inside = (i < SPAN / 4) && q >= 0 && q + 3 < p.n_samples;
    stage[it] = *reinterpret_cast<const f32x4u *>(xc + (inside ? q : 0));
  }
#pragma unroll
  for (int it = 0; it < NLD; ++it) {
    const int i = tid + 256 * it;
    const long long q = p0 + 4LL * i;
    if (i < SPAN / 4) {
      f32x4v v = stage[it];
      if (!(q >= 0 && q + 3 < p.n_samples)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (q + e >= 0 && q + e < p.n_samples) ? xc[q + e] : 0.f;
      }
      const int pp = 4 * i;
      *reinterpret_cast<f32x4v *>(sX + (pp >> 6) * FIR_ROW + (pp & 63)) = v;
    }
  }
  for (int i = tid; i < TAPZ; i += 256) {
    const int t = i - 62;
    sT[i] = (t >= 0 && t < p.n_taps) ? p.a_re[t] : 0.f;
  }
  __syncthreads();

  // ---- Toeplitz A-fragments: lane (r = li, lh), group g, element e <-> T[r, 8g + 4lh + e]
  f32x4v afr[FIR_KG];
#pragma unroll
  for (int g = 0; g < FIR_KG; ++g) {
    const int idx = 8 * g + 4 * lh - 2 * li + 62;  // even: 8-byte aligned pairs
    const float2 lo = *reinterpret_cast<const float2 *>(sT + idx);
    const float2 hi = *reinterpret_cast<const float2 *>(sT + idx + 2);
    afr[g] = f32x4v{lo.x, lo.y, hi.x, hi.y};
    // AREG: keep the fragment resident (otherwise the compiler re-reads it from LDS in the K
    // loop, trading 160 VGPRs for two ds_read_b64 per group and twice the occupancy)
    if (AREG) asm volatile("" : "+v"(afr[g]));
  }

  f32x16 acc[NRW];
#pragma unroll
  for (int n = 0; n < NRW; ++n)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[n][e] = 0.f;

  // B fragment of tile n: frame q = 32 (wave NRW + n) + li starts at span row q; element
  // m = 8g + 4lh + e of it sits at row q + m/64, column m%64
  const float *bbase = sX + (32 * (wave * NRW) + li) * FIR_ROW + 4 * lh;
  f32x4v bv[2][NRW];
  auto load_b = [&](int g, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NRW; ++n)
      bv[slot][n] =
          *reinterpret_cast<const f32x4v *>(bbase + (32 * n + (g >> 3)) * FIR_ROW + 8 * (g & 7));
  };
  load_b(0, 0);
#pragma unroll
  for (int g = 0; g < FIR_KG; ++g) {
    if (g + 1 < FIR_KG) load_b(g + 1, (g + 1) & 1);  // prefetch under this group's MFMAs
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int n = 0; n < NRW; ++n)
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[g][e], bv[g & 1][n][e], acc[n], 0, 0, 0);
  }

  // ---- epilogue: D[r][q] -> y[o0 + 32 (32 (wave NRW + n) + q) + r], transposed through LDS so
  // that a lane stores consecutive outputs
  __syncthreads();  // every wave is done reading the span
  constexpr int LDC = 33;
  float *sC = reinterpret_cast<float *>(smem_raw) + wave * (32 * LDC);
  float *yc = p.out + (long long)c * p.out_clip_stride;
#pragma unroll
  for (int n = 0; n < NRW; ++n) {
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sC[li * LDC + (e & 3) + 8 * (e >> 2) + 4 * lh] = acc[n][e];  // [q][r]
    __syncthreads();
    const long long ob = o0 + 1024LL * (wave * NRW + n);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int q = 2 * it + lh;
      const long long o = ob + 32 * q + li;
      if (o < p.out_len) yc[o] = sC[q * LDC + li];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// Reference kernel: one thread per output element, straight loop (test cross-check only)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) framed_gemm_ref_kernel(const KParams p) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = p.n_cols * p.n_bins;
  if (gid >= total) return;
  const long long col = gid % p.n_cols;
  const int bin = (int)(gid / p.n_cols);
  const int c = (int)(col / p.n_frames);
  const int t = (int)(col - (long long)c * p.n_frames);
  const long long base = (long long)c * p.x_clip_stride;
  const int pos = t * p.hop - p.pad;
  float re = 0.f, im = 0.f;
  const float *wr = p.a_re + (long long)bin * p.a_row_stride;
  const float *wi = p.a_im ? p.a_im + (long long)bin * p.a_row_stride : nullptr;
  for (int k = 0; k < p.K; ++k) {
    const float xv = fetch_sample(p.x, base, pos + k, p.n_samples, p.pad_mode, true);
    re = fmaf(xv, wr[k], re);
    if (wi) im = fmaf(xv, wi[k], im);
  }
  im *= p.im_sign;
  if (p.row_scale) {
    re *= p.row_scale[bin];
    im *= p.row_scale[bin];
  }
  const int E = epilogue_width(p.epilogue);
  float *dst = p.out + (long long)c * p.out_clip_stride +
               (long long)(p.out_row_offset + bin) * p.out_row_stride + (long long)t * E;
  if (wi)
    epilogue_store(p, dst, re, im);
  else
    dst[0] = re;
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}

template <int WM, int WN, int MR, int NR, int BMODE, bool GLDS>
constexpr size_t cfg_smem() {
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int LROW = GLDS ? KC : LDT;
  constexpr int A_STAGE = BM * LROW;
  constexpr int B_STAGE = (BMODE == BMODE_FRAMED) ? BN * LROW : KC * BN;
  return sizeof(float) * 2 * (A_STAGE + B_STAGE) + sizeof(const float *) * BN +
         sizeof(int) * 4 * (WM * MR);  // (K ranges of up to 2 * WM * MR support tiles)
}

// fill the tiling fields of p for a tile shape; returns the number of workgroups (or < 0)
template <int WM, int WN, int MR, int NR, int AMODE>
long long prepare_tiling(KParams &p) {
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  const int rows = AMODE == AMODE_TOEPLITZ ? p.n_bins : p.n_bins * (p.a_im ? 2 : 1);
  p.n_tiles_m = (rows + BM - 1) / BM;
  const long long tn = (p.n_cols + BN - 1) / BN;
  if (tn * p.n_tiles_m > 0x7fffffffLL) return -1;
  p.n_tiles_n = (int)tn;
  // ~64 workgroups are resident per XCD (32 CUs x 2): cross all row tiles with about
  // 64 / n_tiles_m frame tiles before advancing.  On the STFT cfg2 shape this order runs at
  // the same speed as the frame-tile-fastest one but moves 2.8x less data across the fabric
  // (rocprofv3 FETCH_SIZE 1.28e6 KB vs 3.64e6 KB per launch): the K/hop-fold re-reads of the
  // waveform hit L2 instead of the Infinity Cache.
  int g = (64 + p.n_tiles_m / 2) / p.n_tiles_m;
  if MISPEC_DBG(p, 0x100) g = 1 << 20;  // benchmarking: frame-tile-fastest order
  if (g < 1) g = 1;
  if (g > p.n_tiles_n) g = p.n_tiles_n;
  p.n_group = g;
  return tn * p.n_tiles_m;
}

// opt in to > 64 KiB of dynamic LDS once per (kernel, device)
template <typename K>
int configure_lds(K kern, size_t smem, std::atomic<unsigned long long> &configured) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(MISPEC_E_HIP, "hipGetDevice failed%s");
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(configured.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(MISPEC_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured.fetch_or(bit, std::memory_order_release);
  }
  return MISPEC_OK;
}

template <int WM, int WN, int MR, int NR, int BMODE, int AMODE = AMODE_ROWS, bool MASKED = true,
          bool GLDS = false, bool T16 = false>
int launch_cfg(KParams p, hipStream_t stream) {
  constexpr size_t smem = cfg_smem<WM, WN, MR, NR, BMODE, GLDS>();
  const long long grid = prepare_tiling<WM, WN, MR, NR, AMODE>(p);
  if (grid < 0) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  if (grid == 0) return MISPEC_OK;
  auto kern = framed_gemm_kernel<WM, WN, MR, NR, BMODE, AMODE, MASKED, GLDS, T16>;
  static std::atomic<unsigned long long> configured{0};
  int rc = configure_lds(kern, smem, configured);
  if (rc != MISPEC_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

// grouped launch of up to GROUP_MAX problems with one tile shape
template <int WM, int WN, int MR, int NR, bool MASKED, bool GLDS>
int launch_group_cfg(KParams *ps, int n, hipStream_t stream) {
  constexpr size_t smem = cfg_smem<WM, WN, MR, NR, BMODE_FRAMED, GLDS>();
  KGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    const long long grid = prepare_tiling<WM, WN, MR, NR, AMODE_ROWS>(ps[i]);
    if (grid < 0 || total + grid > 0x7fffffffLL)
      return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
    g.first[i] = (int)total;
    total += grid;
    g.p[i] = ps[i];
  }
  for (int i = n; i <= GROUP_MAX; ++i) g.first[i] = (int)total;
  if (total == 0) return MISPEC_OK;
  auto kern = framed_gemm_group_kernel<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, MASKED, GLDS>;
  static std::atomic<unsigned long long> configured{0};
  int rc = configure_lds(kern, smem, configured);
  if (rc != MISPEC_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(WM * WN * 64), smem, stream, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

// LDS-direct loads (global_load_lds_dwordx4) are used for every framed launch.  Their 16-byte
// pieces only need element (4-byte) alignment at the source: the whole GPU parity suite,
// including odd hops / pads / clip lengths, was run with this path forced on.  Bit 0x800 of the
// debug word selects the register-staged loop instead (A/B comparisons in scripts/kbench.py).
bool glds_ok(const KParams &p) { return !MISPEC_DBG(p, 0x800); }

template <int WM, int WN, int MR, int NR>
int launch_pick_mask(const KParams &p, bool masked, hipStream_t stream) {
  const bool g = glds_ok(p);
  if (masked) {
    // every wave owning all row tiles of 32 frames (the TALL shapes): supports at 16-row granularity
    if constexpr (WM == 1 && NR == 1 && MR <= 4)
      if (g && !p.fb && !MISPEC_DBG(p, 0x10000000))
        return launch_cfg<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, true, true, true>(p, stream);
    if (g) return launch_cfg<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, true, true>(p, stream);
    return launch_cfg<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, true, false>(p, stream);
  }
  if (g) return launch_cfg<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, false, true>(p, stream);
  return launch_cfg<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, false, false>(p, stream);
}

int launch_tile(const KParams &p, int tile, hipStream_t stream) {
  // skipping row tiles per K stage only pays (and is only needed) with per-row supports
  const bool masked = p.row_support != nullptr;
  switch (tile) {
    case MISPEC_TILE_128x128:
      return launch_pick_mask<2, 2, 2, 2>(p, masked, stream);
    case MISPEC_TILE_32x256:  // one row tile: the workgroup K range is the tile's range
      return launch_pick_mask<1, 4, 1, 2>(p, false, stream);
    case MISPEC_TILE_64x256:
      return launch_pick_mask<1, 4, 2, 2>(p, masked, stream);
    case MISPEC_TILE_128x128_TALL:
      return launch_pick_mask<1, 4, 4, 1>(p, masked, stream);
    case MISPEC_TILE_192x128:
      return launch_pick_mask<1, 4, 6, 1>(p, masked, stream);
    case MISPEC_TILE_256x128:
      return launch_pick_mask<1, 4, 8, 1>(p, masked, stream);
    case MISPEC_TILE_256x128_SQ:
      return launch_pick_mask<2, 2, 4, 2>(p, masked, stream);
    case MISPEC_TILE_128x256_SQ:
      return launch_pick_mask<2, 2, 2, 4>(p, masked, stream);
    case MISPEC_TILE_256x256:
      return launch_pick_mask<2, 2, 4, 4>(p, masked, stream);
    default:
      return fail(MISPEC_E_INVALID, "unknown tile id%s");
  }
}

int auto_tile(int rows, bool support) {
  // support-aware: every wave owns all row tiles of the workgroup, so skipped K stages shorten the whole workgroup
  // instead of idling some of its waves -- and (round 5) this is the instance that multiplies 16-row tiles with the taps
  // in ascending order, one float32 FMA chain per output like the reference's conv1d: banks of ANY size take it
  if (support) return MISPEC_TILE_128x128_TALL;
  if (rows <= 32) return MISPEC_TILE_32x256;
  if (rows <= 64) return MISPEC_TILE_64x256;
  return MISPEC_TILE_128x128;
}

int launch_framed(const KParams &p, int tile, hipStream_t stream) {
  const int rpb = p.a_im ? 2 : 1;
  const int rows = p.n_bins * rpb;
  if (tile != MISPEC_TILE_AUTO) return launch_tile(p, tile, stream);
  if (p.row_support || rows <= 128)
    return launch_tile(p, auto_tile(rows, p.row_support != nullptr), stream);
  // dense basis, many rows: full 128-row workgroups run the unmasked kernel; the leftover
  // rows (e.g. the Nyquist bin of an n_fft/2+1 STFT) go to a second, narrow launch instead
  // of a 17th mostly-empty row block.
  const int bins_per_wg = 128 / rpb;
  const int main_bins = (p.n_bins / bins_per_wg) * bins_per_wg;
  KParams q = p;
  q.n_bins = main_bins;
  const int rem = p.n_bins - main_bins;
  if (rem == 0) return launch_tile(q, MISPEC_TILE_128x128, stream);
  KParams r = p;
  r.n_bins = rem;
  r.a_re = p.a_re + (long long)main_bins * p.a_row_stride;
  if (p.a_im) r.a_im = p.a_im + (long long)main_bins * p.a_row_stride;
  if (p.row_scale) r.row_scale = p.row_scale + main_bins;
  r.out_row_offset = p.out_row_offset + main_bins;
  const int rem_rows = rem * rpb;
  if (rem_rows <= 64 && glds_ok(p) && !MISPEC_DBG(p, 0x2000)) {
    // one launch: main grid + narrow workgroups for the leftover rows in its tail
    const long long gm = prepare_tiling<2, 2, 2, 2, AMODE_ROWS>(q);
    const long long gr = rem_rows <= 32 ? prepare_tiling<1, 4, 1, 2, AMODE_ROWS>(r)
                                        : prepare_tiling<1, 4, 2, 2, AMODE_ROWS>(r);
    if (gm < 0 || gr < 0 || gm + gr > 0x7fffffffLL)
      return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
    constexpr size_t sm_main = cfg_smem<2, 2, 2, 2, BMODE_FRAMED, true>();
    constexpr size_t sm_r1 = cfg_smem<1, 4, 1, 2, BMODE_FRAMED, true>();
    constexpr size_t sm_r2 = cfg_smem<1, 4, 2, 2, BMODE_FRAMED, true>();
    hipError_t e;
    if (rem_rows <= 32) {
      constexpr size_t smem = sm_main > sm_r1 ? sm_main : sm_r1;
      auto kern = framed_gemm_pair_kernel<1>;
      static std::atomic<unsigned long long> configured{0};
      int rc = configure_lds(kern, smem, configured);
      if (rc != MISPEC_OK) return rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)(gm + gr)), dim3(256), smem, stream, q, r, (int)gm);
    } else {
      constexpr size_t smem = sm_main > sm_r2 ? sm_main : sm_r2;
      auto kern = framed_gemm_pair_kernel<2>;
      static std::atomic<unsigned long long> configured{0};
      int rc = configure_lds(kern, smem, configured);
      if (rc != MISPEC_OK) return rc;
      hipLaunchKernelGGL(kern, dim3((unsigned)(gm + gr)), dim3(256), smem, stream, q, r, (int)gm);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
    return MISPEC_OK;
  }
  int rc = launch_tile(q, MISPEC_TILE_128x128, stream);
  if (rc != MISPEC_OK) return rc;
  return launch_tile(r, auto_tile(rem_rows, false), stream);
}

// ---------------------------------------------------------------------------------
// iSTFT overlap-add (stft.py:33-54, utils.py:43-57): frames (clip, t, n) -> waveform
//   y[c, i] = (sum_t frames[c, t, n] * win[n] / N) / wss,   n = i + start - t*hop in [0, N),
//   wss = sum_t win[n]^2 over the same t (division skipped where wss <= 1e-10)
// One thread per output sample: a gather over the <= ceil(N/hop) frames covering it (no atomics,
// fixed summation order); consecutive lanes read consecutive n of the same frame.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) overlap_add_kernel(const float *__restrict__ frames,
                                                          const float *__restrict__ win, int N,
                                                          int hop, int n_frames, int start,
                                                          int out_len, float *__restrict__ out,
                                                          long long out_clip_stride,
                                                          long long frames_t_cols) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= out_len) return;
  const int c = blockIdx.y;
  if (frames_t_cols) {
    // plain overlap-add of frames stored tap-major: frames[n, (c,t)], frames_t_cols = n_clips*T
    const long long pos = (long long)i + start;
    int t_hi = (int)(pos / hop);
    t_hi = t_hi < n_frames - 1 ? t_hi : n_frames - 1;
    const long long t_lo_num = pos - N + 1;
    const int t_lo = t_lo_num <= 0 ? 0 : (int)((t_lo_num + hop - 1) / hop);
    float acc = 0.f;
    for (int t = t_lo; t <= t_hi; ++t)
      acc += frames[(pos - (long long)t * hop) * frames_t_cols + (long long)c * n_frames + t];
    out[(long long)c * out_clip_stride + i] = acc;
    return;
  }
  const long long pos = (long long)i + start;  // position in the un-trimmed overlap-add signal
  int t_hi = (int)(pos / hop);
  t_hi = t_hi < n_frames - 1 ? t_hi : n_frames - 1;
  long long t_lo_num = pos - N + 1;
  int t_lo = t_lo_num <= 0 ? 0 : (int)((t_lo_num + hop - 1) / hop);
  const float *f = frames + (long long)c * n_frames * N;
  const float inv_n = 1.0f / (float)N;
  float acc = 0.f, wss = 0.f;
  if (win) {
    for (int t = t_lo; t <= t_hi; ++t) {
      const int n = (int)(pos - (long long)t * hop);
      const float w = win[n];
      acc += f[(long long)t * N + n] * w * inv_n;
      wss += w * w;
    }
    if (wss > 1e-10f) acc /= wss;
  } else {  // plain overlap-add (adjoint of framing)
    for (int t = t_lo; t <= t_hi; ++t) acc += f[(long long)t * N + (int)(pos - (long long)t * hop)];
  }
  out[(long long)c * out_clip_stride + i] = acc;
}

// Adjoint of the windowed overlap-add w.r.t. its (window-weighted) frame sum: the gradient of the
// trimmed waveform scattered back onto the un-trimmed overlap-add axis and divided by N * wss,
//   u[c, p] = grad_out[c, p - start] / (N * wss[p])   for p in [start, start + out_len), else 0.
// d frames[c, t, n] = win[n] * u[c, t*hop + n], so d spectrogram is the framed contraction of u
// with the window-weighted transposed synthesis basis.
__global__ void __launch_bounds__(256) istft_grad_signal_kernel(
    const float *__restrict__ go, long long go_clip_stride, const float *__restrict__ win, int N,
    int hop, int n_frames, int start, int out_len, float *__restrict__ u, long long full) {
  const long long pos = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= full) return;
  const int c = blockIdx.y;
  float v = 0.f;
  if (pos >= start && pos < (long long)start + out_len) {
    int t_hi = (int)(pos / hop);
    t_hi = t_hi < n_frames - 1 ? t_hi : n_frames - 1;
    const long long t_lo_num = pos - N + 1;
    const int t_lo = t_lo_num <= 0 ? 0 : (int)((t_lo_num + hop - 1) / hop);
    float wss = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
      const float w = win[(int)(pos - (long long)t * hop)];
      wss += w * w;
    }
    v = go[(long long)c * go_clip_stride + (pos - start)] / (float)N;
    if (wss > 1e-10f) v /= wss;
  }
  u[(long long)c * full + pos] = v;
}

// ---------------------------------------------------------------------------------
// Backward of the framed contraction (trainable bases, stft.py:238-242 / cqt.py:698-702; SURVEY 8f
// rank 3).  With acc_re/acc_im the contraction sums, s the per-bin scale and
//   (u, v) = (s*acc_re, s*im_sign*acc_im)   -- what MISPEC_EPI_COMPLEX stores --
// the pointwise epilogue is out = E(u, v).  The backward pass is
//   1. framed_epilogue_bwd_kernel:  (grad_out, u, v) -> G = (dL/dacc_re, dL/dacc_im), laid out
//      (2, F, B, T) so that a row (component, bin) is contiguous over the flat frame axis;
//   2. d basis = G x frames^T  and  d frames = basis^T x G: the planar contraction kernel
//      (mispec_contract_planar_f32), frames read from the padded signal through a k-offset table;
//   3. d signal = overlap-add of d frames (overlap_add_kernel, no window) folded back through the
//      padding (unpad_adjoint_kernel).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pad_signal_kernel(const KParams p, float *__restrict__ out) {
  const long long Lp = (long long)p.n_samples + 2LL * p.pad;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= Lp) return;
  const int c = blockIdx.y;
  out[(long long)c * Lp + i] = fetch_sample(p.x, (long long)c * p.x_clip_stride, (int)(i - p.pad),
                                            p.n_samples, p.pad_mode, true);
}

// dx[c, i] = dxp[c, i + pad] + the padded positions that mirror onto sample i (reflect)
__global__ void __launch_bounds__(256) unpad_adjoint_kernel(const float *__restrict__ dxp, int L,
                                                            int pad, int pad_mode,
                                                            float *__restrict__ dx,
                                                            long long dx_clip_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L) return;
  const int c = blockIdx.y;
  const float *g = dxp + (long long)c * (L + 2LL * pad);
  float v = g[i + pad];
  if (pad_mode == MISPEC_PAD_REFLECT) {
    if (i >= 1 && i <= pad) v += g[pad - i];                          // position -i
    if (i <= L - 2 && i >= L - 1 - pad) v += g[pad + 2 * L - 2 - i];  // position 2L-2-i
  }
  dx[(long long)c * dx_clip_stride + i] = v;
}

// Frame matrix, transposed: xt[n, (c,t)] = xp[c, t*hop + n]  (n < N taps; (c,t) the flat frame
// axis).  32x32 tiles through LDS: reads run along n (contiguous in xp), writes along t.
__global__ void __launch_bounds__(256) frames_transpose_kernel(const float *__restrict__ xp,
                                                               long long clip_stride, int n_frames,
                                                               int hop, int N, long long n_cols,
                                                               float *__restrict__ xt) {
  __shared__ float tile[32][33];
  const long long col0 = (long long)blockIdx.x * 32;  // flat frame index
  const int n0 = blockIdx.y * 32;
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ly; r < 32; r += 8) {
    const long long col = col0 + r;
    float v = 0.f;
    if (col < n_cols && n0 + lx < N) {
      const int c = (int)(col / n_frames);
      const int t = (int)(col - (long long)c * n_frames);
      v = xp[(long long)c * clip_stride + (long long)t * hop + n0 + lx];
    }
    tile[r][lx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int r = ly; r < 32; r += 8) {
    const int n = n0 + r;
    const long long col = col0 + lx;
    if (n < N && col < n_cols) xt[(long long)n * n_cols + col] = tile[lx][r];
  }
}

__global__ void __launch_bounds__(256) frame_offsets_kernel(long long *__restrict__ koff, int n_clips,
                                                            int n_frames, long long clip_stride,
                                                            int hop) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= (long long)n_clips * n_frames) return;
  const int c = (int)(k / n_frames);
  const int t = (int)(k - (long long)c * n_frames);
  koff[k] = (long long)c * clip_stride + (long long)t * hop;
}

// the epilogue of the contraction kernels as a pass of its own: out <- epilogue(z), z = their Complex output
struct EpiParams {
  int epilogue;
  float eps, power;
};
__global__ void __launch_bounds__(256) framed_epilogue_fwd_kernel(const float *__restrict__ z, long long total, EpiParams ep,
                                                                  float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float2 v = *reinterpret_cast<const float2 *>(z + 2 * i);
  epilogue_store(ep, out + i * epilogue_width(ep.epilogue), v.x, v.y);
}

__global__ void __launch_bounds__(256) framed_epilogue_bwd_kernel(
    const float *__restrict__ go, const float *__restrict__ z, int n_clips, int n_bins, int n_frames,
    int epilogue, float eps, float power, float im_sign, const float *__restrict__ row_scale,
    float *__restrict__ g, float *__restrict__ gt) {
  const long long total = (long long)n_clips * n_bins * n_frames;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int t = (int)(i % n_frames);
  const long long bf = i / n_frames;
  const int f = (int)(bf % n_bins);
  const int c = (int)(bf / n_bins);
  const float u = z[2 * i], v = z[2 * i + 1];
  float gu = 0.f, gv = 0.f;
  const float r2 = u * u + v * v;
  switch (epilogue) {
    case MISPEC_EPI_COMPLEX:
      gu = go[2 * i];
      gv = go[2 * i + 1];
      break;
    case MISPEC_EPI_MAGNITUDE: {
      const float y = sqrtf(r2 + eps);
      if (y > 0.f) {
        gu = go[i] * u / y;
        gv = go[i] * v / y;
      }
    } break;
    case MISPEC_EPI_POWER: {
      // y = sqrt(s)^p, s = u^2 + v^2 + eps:  dy/du = p * s^(p/2 - 1) * u
      const float s0 = r2 + eps;
      if (s0 > 0.f) {
        const float k = power * powf(s0, 0.5f * power - 1.0f) * go[i];
        gu = k * u;
        gv = k * v;
      }
    } break;
    case MISPEC_EPI_PHASE_ATAN2:
      if (r2 > 0.f) {
        gu = -go[i] * v / r2;
        gv = go[i] * u / r2;
      }
      break;
    case MISPEC_EPI_PHASE_COSSIN:
      if (r2 > 0.f) {
        const float r3 = r2 * sqrtf(r2);
        const float g0 = go[2 * i], g1 = go[2 * i + 1];
        gu = (g0 * v * v - g1 * u * v) / r3;
        gv = (g1 * u * u - g0 * u * v) / r3;
      }
      break;
    default:
      break;
  }
  const float sc = row_scale ? row_scale[f] : 1.f;
  const long long plane = (long long)n_bins * n_clips * n_frames;
  const long long o = ((long long)f * n_clips + c) * n_frames + t;
  if (g) {
    g[o] = sc * gu;
    g[plane + o] = sc * im_sign * gv;
  }
  if (gt) {  // (B, T, 2F): one frame's gradient vector contiguous, [re bins | im bins]
    const long long ot = ((long long)c * n_frames + t) * (2LL * n_bins) + f;
    gt[ot] = sc * gu;
    gt[ot + n_bins] = sc * im_sign * gv;
  }
}

// adjoint of the strided FIR  y[i] = sum_n taps[n] * x[i*stride + n - pad]  (zero outside):
//   dx[m] = sum_i dy[i] * taps[m + pad - i*stride]     (a gather: <= ceil(n_taps/stride) terms)
__global__ void __launch_bounds__(256) fir_decimate_bwd_kernel(const float *__restrict__ dy,
                                                               long long dy_clip_stride, int n_out,
                                                               const float *__restrict__ taps,
                                                               int n_taps, int stride, int pad, int L,
                                                               float *__restrict__ dx,
                                                               long long dx_clip_stride) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= L) return;
  const int c = blockIdx.y;
  const long long q = (long long)m + pad;  // tap index of output 0 at this sample
  long long lo_num = q - n_taps + 1;
  int i_lo = lo_num <= 0 ? 0 : (int)((lo_num + stride - 1) / stride);
  int i_hi = (int)(q / stride);
  i_hi = i_hi < n_out - 1 ? i_hi : n_out - 1;
  const float *g = dy + (long long)c * dy_clip_stride;
  float acc = 0.f;
  for (int i = i_lo; i <= i_hi; ++i) acc += g[i] * taps[(int)(q - (long long)i * stride)];
  dx[(long long)c * dx_clip_stride + m] = acc;
}

// ---------------------------------------------------------------------------------
// power_to_db (MFCC, mel.py:263-279): HBM-bound pointwise pass with a per-clip maximum
// ---------------------------------------------------------------------------------
// per-clip maximum of max(spec, amin) (> 0: unsigned compare on the float bits is monotonic)
__global__ void __launch_bounds__(256) clip_max_kernel(const float *__restrict__ spec,
                                                       long long clip_elems, float amin,
                                                       unsigned *__restrict__ wmax) {
  const int c = blockIdx.y;
  const float *s = spec + (long long)c * clip_elems;
  float m = amin;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < clip_elems;
       i += (long long)gridDim.x * 256)
    m = fmaxf(m, s[i]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) atomicMax(&wmax[c], __float_as_uint(m));
}

__global__ void __launch_bounds__(256) clear_u32_kernel(unsigned *__restrict__ w, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) w[i] = 0u;
}

__global__ void __launch_bounds__(256) power_to_db_kernel(const float *__restrict__ spec,
                                                          long long clip_elems, float amin,
                                                          float ref, float top_db,
                                                          const unsigned *__restrict__ wmax,
                                                          float *__restrict__ out) {
  const int c = blockIdx.y;
  const long long base = (long long)c * clip_elems;
  const float off = 10.0f * log10f(fmaxf(amin, ref));
  float floor_db = -INFINITY;
  if (top_db >= 0.f) floor_db = (10.0f * log10f(__uint_as_float(wmax[c])) - off) - top_db;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < clip_elems;
       i += (long long)gridDim.x * 256) {
    const float l = 10.0f * log10f(fmaxf(spec[base + i], amin)) - off;
    out[base + i] = fmaxf(l, floor_db);
  }
}

// MFCC's tail in one launch (mel.py:263-307): power_to_db with the per-clip maximum, then the DCT -- one workgroup
// per clip.  Pass 1 reads the clip's mel spectrogram for its maximum (the floor of top_db), pass 2 walks it in
// tiles of 64 frames: n_mels x 64 decibel values in LDS (the same expressions as power_to_db_kernel), contracted
// with the (n_mfcc, n_mels) cosine matrix: wave w takes coefficients KK w .. KK w + KK - 1 (+ 8 KK ..), lane = frame;
// a wave's weights are uniform -- scalar loads straight from the matrix (10 KB: scalar cache / L2), no LDS, no VALU.
// The second read of the clip comes out of L2 (~100 KB).  Replaces clear + clip_max + power_to_db + the filterbank
// GEMM: four launches on a 28 MB tensor.  Few clips: several workgroups per clip (blockIdx.y), each with a share of
// the tiles and its own pass 1.
constexpr int MFCC_TT = 64;
template <int KK>
__global__ void __launch_bounds__(512, 6) mfcc_tail_kernel(const float *__restrict__ spec, int n_mels, int n_frames,
                                                           float amin, float ref, float top_db,
                                                           const float *__restrict__ dct, int n_mfcc,
                                                           float *__restrict__ out, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *const s_tile = reinterpret_cast<float *>(smem_raw);  // [n_mels][MFCC_TT]
  __shared__ float s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long long clip_elems = (long long)n_mels * n_frames;
  // blockIdx.y: a share of the clip's tiles (few clips: several workgroups per clip, each finds the maximum itself)
  const float *const s = spec + (long long)blockIdx.x * clip_elems;
  float *const o = out + (long long)blockIdx.x * n_mfcc * n_frames;
  const int t_first = blockIdx.y * tiles_per_wg * MFCC_TT;
  const int t_last = t_first + tiles_per_wg * MFCC_TT < n_frames ? t_first + tiles_per_wg * MFCC_TT : n_frames;
  const float off = 10.0f * log10f(fmaxf(amin, ref));
  float floor_db = -INFINITY;
  if (top_db >= 0.f) {
    // (many loads in flight per thread: a loop of single dependent-looking loads runs at one memory latency per trip)
    float m = amin;
    const long long n4 = (reinterpret_cast<uintptr_t>(s) & 15) == 0 ? clip_elems / 4 : 0;
    const f32x4v *const s4 = reinterpret_cast<const f32x4v *>(s);
    long long i = tid;
    for (; i + 3 * 512 < n4; i += 4 * 512) {
      const f32x4v a = s4[i], b = s4[i + 512], c4 = s4[i + 1024], d = s4[i + 1536];
      m = fmaxf(m, fmaxf(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])), fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3]))));
      m = fmaxf(m, fmaxf(fmaxf(fmaxf(c4[0], c4[1]), fmaxf(c4[2], c4[3])), fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]))));
    }
    for (; i < n4; i += 512) {
      const f32x4v a = s4[i];
      m = fmaxf(m, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
    }
    for (long long j = 4 * n4 + tid; j < clip_elems; j += 512) m = fmaxf(m, s[j]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])), fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7])));
    floor_db = (10.0f * log10f(m) - off) - top_db;
  }
  const int m8 = n_mels & ~7;
  for (int t0 = t_first; t0 < t_last; t0 += MFCC_TT) {
    if (t0 != t_first) __syncthreads();  // (the previous tile has been contracted)
    for (int i0 = tid; i0 < n_mels * MFCC_TT; i0 += 8 * 512) {  // (eight loads in flight per thread)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + 512 * u, m = idx >> 6, t = idx & 63;
        v[u] = (idx < n_mels * MFCC_TT && t0 + t < n_frames) ? s[(long long)m * n_frames + t0 + t] : amin;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + 512 * u;
        if (idx < n_mels * MFCC_TT) s_tile[idx] = fmaxf(10.0f * log10f(fmaxf(v[u], amin)) - off, floor_db);
      }
    }
    __syncthreads();
    for (int k0 = KK * wave; k0 < n_mfcc; k0 += 8 * KK) {  // (wave-uniform)
      float acc[KK];
      const float *d[KK];
#pragma unroll
      for (int e = 0; e < KK; ++e) {
        acc[e] = 0.f;
        d[e] = dct + (long long)(k0 + e < n_mfcc ? k0 + e : n_mfcc - 1) * n_mels;  // (rows past the end: computed, not stored)
      }
      for (int m = 0; m < m8; m += 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = s_tile[(m + u) * MFCC_TT + lane];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < KK; ++e) acc[e] = fmaf(d[e][m + u], x[u], acc[e]);
      }
      for (int m = m8; m < n_mels; ++m) {
        const float x = s_tile[m * MFCC_TT + lane];
#pragma unroll
        for (int e = 0; e < KK; ++e) acc[e] = fmaf(d[e][m], x, acc[e]);
      }
      if (t0 + lane < n_frames) {
#pragma unroll
        for (int e = 0; e < KK; ++e)
          if (k0 + e < n_mfcc) o[(long long)(k0 + e) * n_frames + t0 + lane] = acc[e];
      }
    }
  }
}

// power_to_db backward.  Elements above the per-clip floor pass their gradient through the
// logarithm; the floored ones hand theirs to the clip maximum (out = max(l, max(l) - top_db)).
__global__ void __launch_bounds__(256) power_to_db_floor_sum_kernel(
    const float *__restrict__ spec, const float *__restrict__ go, long long clip_elems, float amin,
    float top_db, const unsigned *__restrict__ wmax, float *__restrict__ wsum) {
  const int c = blockIdx.y;
  const long long base = (long long)c * clip_elems;
  const float lmax = 10.0f * log10f(__uint_as_float(wmax[c]));
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < clip_elems;
       i += (long long)gridDim.x * 256) {
    const float l = 10.0f * log10f(fmaxf(spec[base + i], amin));
    if (l < lmax - top_db) acc += go[base + i];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) atomicAdd(&wsum[c], acc);
}

__global__ void __launch_bounds__(256) power_to_db_bwd_kernel(
    const float *__restrict__ spec, const float *__restrict__ go, long long clip_elems, float amin,
    float top_db, const unsigned *__restrict__ wmax, const float *__restrict__ wsum,
    float *__restrict__ gs) {
  const int c = blockIdx.y;
  const long long base = (long long)c * clip_elems;
  const float smax = top_db >= 0.f ? __uint_as_float(wmax[c]) : 0.f;
  const float lmax = top_db >= 0.f ? 10.0f * log10f(smax) : 0.f;
  const float k = 4.3429448190325175f;  // 10 / ln(10)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < clip_elems;
       i += (long long)gridDim.x * 256) {
    const float sv = spec[base + i];
    float g = go[base + i];
    if (top_db >= 0.f) {
      const float sc = fmaxf(sv, amin);
      if (10.0f * log10f(sc) < lmax - top_db) g = 0.f;
      if (sc == smax) g += wsum[c];
    }
    gs[base + i] = sv > amin ? g * k / sv : 0.f;
  }
}

// ---------------------------------------------------------------------------------
// MISPEC_PREC_BF16X3 host side
// ---------------------------------------------------------------------------------
inline long long round_up_ll(long long v, long long m) { return (v + m - 1) / m * m; }

struct SplitPlan {
  long long edge_bytes;  // fp32 edge spans (leftover rows may run on the fp32 kernel), 256-aligned
  long long slot;        // clip slot, elements
  long long bytes;       // both planes
};

SplitPlan plan_split(const KParams &p, const EdgePlan &e) {
  SplitPlan sp{};
  sp.edge_bytes = round_up_ll(e.stride * p.n_clips * (long long)sizeof(float), 256);
  // the padded clip, and room for one hop past the last frame (rows the hop-periodic kernel
  // reads beyond it); 128-byte slots: with a power-of-two hop every frame starts on a cache line
  const long long padded = (long long)p.n_samples + 2LL * p.pad;
  const long long reach = (long long)(p.n_frames - 1) * p.hop + round_up_kc(p.K) + p.hop;
  sp.slot = round_up_ll(padded > reach ? padded : reach, 64);
  // (+ the job counter of the strip kernel, behind the planes)
  sp.bytes = 2 * sp.slot * p.n_clips * (long long)sizeof(unsigned short) + 256;
  return sp;
}

// complex banks of up to 64 row tiles carry a second copy in the fragment order of the strip
// kernel (framed_bf16x3_strip.inl): [16-bin tile][16-tap step][hi | lo][lane][8 taps]
bool basis_has_frags(int n_bins, bool has_im) { return has_im && n_bins <= 64 * 16; }
long long basis_plane_bytes(int n_bins, int kernel, bool has_im) {
  return (has_im ? 4LL : 2LL) * n_bins * round_up_kc(kernel) * (long long)sizeof(unsigned short);
}
long long basis_split_bytes(int n_bins, int kernel, bool has_im) {
  long long b = basis_plane_bytes(n_bins, kernel, has_im);
  if (basis_has_frags(n_bins, has_im))  // (+ a 4 KB block of zeros: the unit of a strip's padding)
    b += (long long)((n_bins + 15) / 16) * round_up_kc(kernel) * 128 + 4096;
  return b;
}

// does the bf16x3 kernel cover this problem?  (else it runs in fp32, which is always acceptable)
bool bf16x3_ok(const mispec_framed_gemm_args *a, const KParams &p) {
  if (a->precision != MISPEC_PREC_BF16X3 || !a->basis_split) return false;
  if (a->basis_split_bytes < basis_split_bytes(p.n_bins, p.K, p.a_im != nullptr)) return false;
  if (p.hop & 1) return false;  // frames must start at even element offsets
  const int rows = p.n_bins * (p.a_im ? 2 : 1);
  return rows > 128;  // narrower problems: the 256-row tile would be mostly empty
}

template <int WM, int WN, int MR, int NR>
constexpr size_t bf16x3_smem() {
  constexpr int NW = WM * WN;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int BMA = BM < 16 * NW ? 16 * NW : BM;
  const size_t stages = 2 * (2 * BMA + 2 * BN) * (KC * 2) + BN * sizeof(long long) + 2 * WM * MR * sizeof(int) +
                        BN * sizeof(float);  // (+ per-column factors of the F16 instances)
  const size_t patches = (size_t)NW * 32 * (NR * 32 + 4) * sizeof(float);  // bf16x3_epilogue_planar
  return stages > patches ? stages : patches;
}

// fill the tiling fields of p for a bf16x3 tile shape; returns the number of workgroups (or < 0)
template <int WM, int WN, int MR, int NR>
long long prepare_bf16x3(KParams &p) {
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  const int rows = p.n_bins * (p.a_im ? 2 : 1);
  p.n_tiles_m = (rows + BM - 1) / BM;
  const long long tn = (p.n_cols + BN - 1) / BN;
  if (tn * p.n_tiles_m > 0x7fffffffLL) return -1;
  p.n_tiles_n = (int)tn;
  // one workgroup per CU, 32 CUs per XCD: cross all row tiles with ~32 / n_tiles_m frame tiles
  int g = (32 + p.n_tiles_m / 2) / p.n_tiles_m;
  if (g < 1) g = 1;
  if (g > p.n_tiles_n) g = p.n_tiles_n;
  p.n_group = g;
  return tn * p.n_tiles_m;
}

template <int WM, int WN, int MR, int NR, bool MASKED, bool F16 = false>
int launch_bf16x3_cfg(KParams p, hipStream_t stream) {
  constexpr size_t smem = bf16x3_smem<WM, WN, MR, NR>();
  static_assert(smem <= 160 * 1024, "LDS budget");
  const long long grid = prepare_bf16x3<WM, WN, MR, NR>(p);
  if (grid < 0) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  if (grid == 0) return MISPEC_OK;
  auto kern = F16 ? framed_f16x3_kernel<WM, WN, MR, NR, MASKED> : framed_bf16x3_kernel<WM, WN, MR, NR, MASKED>;
  static std::atomic<unsigned long long> configured{0};
  int rc = configure_lds(kern, smem, configured);
  if (rc != MISPEC_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

// ---- hop-periodic (slab) variant: applicability, LDS budget, launch
template <int WM, int WN, int MR, int NR>
bool plan_slab(KParams &p, size_t &smem) {
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  if MISPEC_DBG(p, 0x4000) return false;  // benchmarking: force the staged kernel
  if (p.hop % KC != 0 || p.Ks < 2 * p.hop || p.n_frames < BN) return false;
  const int C = (p.Ks + p.hop - 1) / p.hop;
  const int rows = (BN + 2 * (C - 1) + 15) / 16 * 16;
  if (rows > SLAB_MAX_ROWS) return false;
  const size_t a = 2 * 2 * (size_t)BM * (KC * 2);
  const size_t slab = 2 * (size_t)rows * (KC * 2);
  const size_t tables = rows * sizeof(long long) + BN * sizeof(int) + 2 * WM * MR * sizeof(int);
  const size_t lds = 160 * 1024;
  int nbuf = 2;
  if (a + 2 * slab + tables > lds) nbuf = 1;
  if (a + nbuf * slab + tables > lds) return false;
  if MISPEC_DBG(p, 0x8000) nbuf = 1;  // benchmarking: single slab buffer
  p.n_super = C;
  p.slab_rows = rows;
  p.slab_nbuf = nbuf;
  smem = a + nbuf * slab + tables;
  return true;
}

template <int WM, int WN, int MR, int NR, bool MASKED>
int launch_bf16x3_slab_cfg(KParams p, size_t smem, hipStream_t stream) {
  const long long grid = prepare_bf16x3<WM, WN, MR, NR>(p);
  if (grid < 0) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  if (grid == 0) return MISPEC_OK;
  auto kern = framed_bf16x3_slab_kernel<WM, WN, MR, NR, MASKED>;
  // the LDS request varies with K / hop: raise the kernel's limit to the device maximum once
  static std::atomic<unsigned long long> configured{0};
  int rc = configure_lds(kern, 160 * 1024, configured);
  if (rc != MISPEC_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WM * WN * 64), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int launch_bf16x3_narrow(KParams p, size_t smem, hipStream_t stream) {
  if (prepare_bf16x3<1, 8, 1, 1>(p) < 0) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  smem += 2 * 64 * sizeof(int);  // per-sub-stage j ranges (hop / 32 <= 64 sub-stages)
  // one workgroup per frame tile walks all its row tiles (equal work per workgroup); when the
  // frame tiles alone cannot fill the 256 CUs the row tiles are dealt out to more workgroups
  int split = p.n_tiles_n > 0 ? 256 / p.n_tiles_n : 1;
  split = split < 1 ? 1 : (split > p.n_tiles_m ? p.n_tiles_m : split);
  p.row_split = split;
  const long long grid = (long long)p.n_tiles_n * split;
  if (grid == 0) return MISPEC_OK;
  auto kern = framed_bf16x3_narrow_kernel;
  static std::atomic<unsigned long long> configured{0};
  int rc = configure_lds(kern, 160 * 1024, configured);
  if (rc != MISPEC_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

// ---------------------------------------------------------------------------------
// strip kernel (framed_bf16x3_strip.inl): plan and launch.  The plan needs the supports on the
// host (args->row_support_host): row tiles sorted by K range are grouped into passes of up to 4
// tiles, a pass's waves are dealt out in proportion to the tiles' work, and the grouping that
// minimises the estimated makespan of n_tiles_n x passes jobs on the device's CUs wins.
// ---------------------------------------------------------------------------------
struct StripTile {
  int tile, kb, ke, jmin, jmax;
  double work;  // units (sub-stage, super-stage) inside [kb, ke)
};

inline int strip_units(int kb, int ke, int j, int hop) {  // sub-stages of super-stage j inside [kb, ke)
  const long long lo = (long long)j * hop > kb ? (long long)j * hop : kb;
  const long long hi = (long long)(j + 1) * hop < ke ? (long long)(j + 1) * hop : ke;
  return hi > lo ? (int)((hi - lo + KC - 1) / KC) : 0;
}

// waves of a group of tiles (sorted by work, descending): one each, the spare ones to the tile with
// the largest share per wave; returns the largest share
double strip_alloc(const StripTile *t, int k, int *waves) {
  for (int i = 0; i < k; ++i) waves[i] = 1;
  for (int spare = STRIP_NW - k; spare > 0; --spare) {
    int best = 0;
    for (int i = 1; i < k; ++i)
      if (t[i].work / waves[i] > t[best].work / waves[best]) best = i;
    ++waves[best];
  }
  double m = 0;
  for (int i = 0; i < k; ++i) m = t[i].work / waves[i] > m ? t[i].work / waves[i] : m;
  return m;
}

bool plan_strip(const KParams &p, const int32_t *sup, int n_slots, StripPlan &plan, bool f32 = false,
                int bn = STRIP_BN) {
  if (!sup || !p.a_im || !p.row_support) return false;
  if (p.hop % KC != 0 || p.hop > 64 * KC || p.Ks < 2 * p.hop || p.n_frames < bn) return false;
  const int hop = p.hop, sph = hop / KC;
  const int M = (p.n_bins + 15) / 16;
  if (M > STRIP_NW * STRIP_MAX_PASS) return false;
  StripTile t[STRIP_NW * STRIP_MAX_PASS];
  for (int m = 0; m < M; ++m) {
    int lo = p.K, hi = 0;
    for (int b = 16 * m; b < 16 * m + 16 && b < p.n_bins; ++b) {
      const int s0 = sup[2 * b], e0 = sup[2 * b + 1];
      if (e0 > s0) {
        lo = s0 < lo ? s0 : lo;
        hi = e0 > hi ? e0 : hi;
      }
    }
    lo = lo < 0 ? 0 : lo;
    hi = hi > p.K ? p.K : hi;
    if (hi <= lo) lo = hi = 0;
    StripTile &x = t[m];
    x.tile = m;
    x.kb = lo & ~(KC - 1);
    x.ke = hi;
    x.jmin = x.kb / hop;
    x.jmax = hi > lo ? (hi - 1) / hop : x.jmin - 1;
    x.work = 0;
    for (int j = x.jmin; j <= x.jmax; ++j) x.work += strip_units(x.kb, x.ke, j, hop);
  }
  for (int i = 1; i < M; ++i)  // insertion sort, descending work (stable)
    for (int j = i; j > 0 && t[j].work > t[j - 1].work; --j) {
      const StripTile tmp = t[j];
      t[j] = t[j - 1];
      t[j - 1] = tmp;
    }
  const long long n_tiles_n = (p.n_cols + bn - 1) / bn;
  if (n_tiles_n * STRIP_MAX_PASS > 0x7fffffffLL || p.n_cols + bn > 0x7fffffffLL) return false;
  // per-pass cost of tables, first slab, reduction and epilogue: ~10 us = 12 units at cfg4 (phase
  // clock).  (Charging the ~1.2 us of barrier + refill per sub-stage as well -- 12 + 1.45 * sph --
  // merges tiles 1 and 2 into one pass of 3 + 1 waves: measured 4 % slower, the single wave of
  // tile 2 then sets the pace of every sub-stage.)
  // (a unit of the fp32 kernel -- 128 fp32 MFMAs -- takes five times as long: the same ~10 us are
  // fewer units)
  const double ovh = (f32 ? 0.15 : 0.75) * sph;
  // slab reach of a group: span of super-stages
  auto span_of = [&](int a, int k) {
    int lo = 1 << 30, hi = -1;
    for (int i = a; i < a + k; ++i)
      if (t[i].jmax >= t[i].jmin) {
        lo = t[i].jmin < lo ? t[i].jmin : lo;
        hi = t[i].jmax > hi ? t[i].jmax : hi;
      }
    return hi >= lo ? hi - lo + 1 : 1;
  };
  // candidate caps on a pass's largest share
  double caps[STRIP_NW * STRIP_MAX_PASS * STRIP_NW + 1];
  int n_caps = 0;
  for (int m = 0; m < M; ++m)
    for (int w = 1; w <= STRIP_NW; ++w) caps[n_caps++] = t[m].work / w;
  caps[n_caps++] = 1e30;
  double best_est = 1e300;
  int best_cut[STRIP_NW * STRIP_MAX_PASS], best_np = 0;
  // (the search charges `sovh` per pass; when the true overhead is too small to keep a bank of many
  // tiles within STRIP_MAX_PASS passes it is raised until a grouping fits -- the estimate that ranks
  // the groupings always uses the true one)
  for (double sovh = ovh; best_np == 0 && sovh < 1e6; sovh = sovh * 2 + 1)
  for (int ci = 0; ci < n_caps; ++ci) {
    const double cap = caps[ci] + 1e-9;
    // dp[i]: cheapest way to cover the first i tiles (sorted) with passes whose share is <= cap
    double dp[STRIP_NW * STRIP_MAX_PASS + 1], dmax[STRIP_NW * STRIP_MAX_PASS + 1], dsum[STRIP_NW * STRIP_MAX_PASS + 1];
    int from[STRIP_NW * STRIP_MAX_PASS + 1], np[STRIP_NW * STRIP_MAX_PASS + 1];
    dp[0] = 0;
    dmax[0] = 0;
    dsum[0] = 0;
    np[0] = 0;
    for (int i = 1; i <= M; ++i) {
      dp[i] = 1e300;
      for (int k = 1; k <= STRIP_NW && k <= i; ++k) {
        if (dp[i - k] >= 1e300) continue;
        int waves[STRIP_NW];
        const double share = strip_alloc(t + i - k, k, waves);
        const int rows = (bn + 2 * (span_of(i - k, k) - 1) + 15) / 16 * 16;
        if (share > cap || rows > STRIP_MAX_ROWS) continue;
        const double c = dp[i - k] + share + sovh;
        if (c < dp[i]) {
          dp[i] = c;
          from[i] = i - k;
          const double mx = share + ovh;
          dmax[i] = dmax[i - k] > mx ? dmax[i - k] : mx;
          dsum[i] = dsum[i - k] + share + ovh;
          np[i] = np[i - k] + 1;
        }
      }
    }
    if (dp[M] >= 1e300 || np[M] > STRIP_MAX_PASS) continue;
    const double total = dsum[M] * (double)n_tiles_n / n_slots;
    const double jobs = (double)np[M] * n_tiles_n;
    const double est = jobs <= n_slots ? dmax[M] : (total > dmax[M] ? total : dmax[M]) + 0.5 * dmax[M];
    if (est < best_est) {
      best_est = est;
      best_np = np[M];
      int i = M, q = np[M];
      while (i > 0) {
        best_cut[--q] = from[i];
        i = from[i];
      }
    }
  }
  if (best_np == 0) return false;
  memset(&plan, 0, sizeof(plan));
  plan.n_pass = best_np;
  plan.nf = bn / 32;
  plan.n_tiles_n = (int)n_tiles_n;
  plan.n_jobs = (int)(n_tiles_n * best_np);
  for (int q = 0; q < best_np; ++q) {
    const int a = best_cut[q], b = q + 1 < best_np ? best_cut[q + 1] : M, k = b - a;
    StripPass &ps = plan.pass[q];
    int waves[STRIP_NW];
    const double share = strip_alloc(t + a, k, waves);
    ps.cost = (int)(share + ovh);
    int lo = 1 << 30, hi = -1;
    for (int i = a; i < b; ++i)
      if (t[i].jmax >= t[i].jmin) {
        lo = t[i].jmin < lo ? t[i].jmin : lo;
        hi = t[i].jmax > hi ? t[i].jmax : hi;
      }
    if (hi < lo) lo = hi = 0;
    ps.jbase = lo;
    ps.span = hi - lo + 1;
    ps.slab_rows = (bn + 2 * (ps.span - 1) + 15) / 16 * 16;
    ps.group = 2 * ps.slab_rows <= STRIP_MAX_ROWS && sph >= 2 ? 2 : 1;  // two slabs per buffer: half the barriers
    int w = 0;
    for (int i = a; i < b; ++i) {
      // cut the tile's super-stages into waves[i - a] runs of about equal work
      const StripTile &x = t[i];
      const int nw = waves[i - a];
      int j = x.jmin;
      double acc_w = 0;
      for (int r = 0; r < nw; ++r, ++w) {
        StripWave &sw = ps.w[w];
        sw.tile = x.tile;
        sw.kb = x.kb;
        sw.ke = x.ke;
        sw.g0 = w - r;
        sw.gsize = nw;
        sw.fmask = 0;
        for (int f = 0; f < 4; ++f)
          if (f % nw == r) sw.fmask |= 1 << f;
        sw.ja = j;
        const double goal = x.work * (r + 1) / nw;
        while (j <= x.jmax && (r == nw - 1 || acc_w + 0.5 * strip_units(x.kb, x.ke, j, hop) <= goal)) {
          acc_w += strip_units(x.kb, x.ke, j, hop);
          ++j;
        }
        sw.jb = j;
      }
    }
    for (; w < STRIP_NW; ++w) {
      StripWave &sw = ps.w[w];
      sw.tile = -1;
      sw.g0 = w;
      sw.gsize = 1;
    }
  }
  return true;
}

// plans are cached per (supports, shape): the search above costs a fraction of a millisecond
struct StripPlanKey {
  unsigned long long hash;
  int n_bins, hop, Ks, K, n_frames, n_cu;
  long long n_cols;
  int f32, pad;
};
bool strip_plan_cached(const KParams &p, const int32_t *sup, int n_cu, StripPlan &plan, bool f32 = false) {
  static std::mutex mu;
  struct Entry {
    StripPlanKey key;
    std::vector<int32_t> sup;
    StripPlan plan;
    bool ok;
  };
  static std::vector<Entry> cache;
  unsigned long long h = 1469598103934665603ull;
  for (int i = 0; i < 2 * p.n_bins; ++i) h = (h ^ (unsigned)sup[i]) * 1099511628211ull;
  const StripPlanKey key{h, p.n_bins, p.hop, p.Ks, p.K, p.n_frames, n_cu, p.n_cols, f32 ? 1 : 0, 0};
  std::lock_guard<std::mutex> lock(mu);
  for (const Entry &e : cache)
    if (memcmp(&e.key, &key, sizeof(key)) == 0 &&
        memcmp(e.sup.data(), sup, sizeof(int32_t) * 2 * p.n_bins) == 0) {
      plan = e.plan;
      return e.ok;
    }
  Entry e;
  memset(&e.key, 0, sizeof(e.key));
  e.key = key;
  e.sup.assign(sup, sup + 2 * p.n_bins);
  // 128-frame jobs; 64-frame ones when those would not fill the workgroup slots (small batches: twice
  // the jobs, half as long) or do not fit the slab (kernels of more than ~80 hops)
  e.ok = plan_strip(p, sup, n_cu, e.plan, f32, STRIP_BN);
  {
    StripPlan half;
    if (plan_strip(p, sup, n_cu, half, f32, STRIP_BN / 2)) {
      // makespan estimate: all jobs spread over the slots, but never less than the longest job; a
      // 64-frame unit is half the MFMAs of a 128-frame one at ~15 % less efficiency
      auto estimate = [&](const StripPlan &pl) {
        double sum = 0, longest = 0;
        for (int i = 0; i < pl.n_pass; ++i) {
          sum += pl.pass[i].cost;
          longest = pl.pass[i].cost > longest ? pl.pass[i].cost : longest;
        }
        const double spread = sum * pl.n_tiles_n / n_cu;
        return (spread > longest ? spread : longest) * pl.nf * (pl.nf == 2 ? 1.15 : 1.0);
      };
      if (!e.ok || estimate(half) < estimate(e.plan)) {
        e.plan = half;
        e.ok = true;
      }
    }
  }
  if (cache.size() >= 32) cache.erase(cache.begin());
  cache.push_back(e);
  plan = e.plan;
  return e.ok;
}

int device_cus() {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  return cus;
}

int launch_bf16x3_strip(KParams p, const StripPlan &plan, int n_cu, hipStream_t stream) {
  p.n_super = (p.Ks + p.hop - 1) / p.hop;
  auto kern = plan.nf == 2 ? framed_bf16x3_strip64_kernel : framed_bf16x3_strip_kernel;
  static std::atomic<unsigned long long> configured{0}, configured64{0};
  int rc = configure_lds(kern, 160 * 1024, plan.nf == 2 ? configured64 : configured);
  if (rc != MISPEC_OK) return rc;
  unsigned grid = (unsigned)(plan.n_jobs < 2 * n_cu ? plan.n_jobs : 2 * n_cu);  // two per CU
  size_t smem = STRIP_LDS_BYTES;
  if MISPEC_DBG(p, 0x4000000) {  // benchmarking: one workgroup per CU
    grid = (unsigned)(plan.n_jobs < n_cu ? plan.n_jobs : n_cu);
    smem = 100 * 1024;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(STRIP_NW * 64), smem, stream, p, plan);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

// split the waveform + edge spans into the second part of the workspace and attach it to p
int setup_split(KParams &p, const mispec_framed_gemm_args *a, hipStream_t stream) {
  const EdgePlan e = plan_edges(p.n_samples, p.K, p.hop, p.pad, p.n_frames);
  const SplitPlan sp = plan_split(p, e);
  if (!a->workspace || a->workspace_bytes < sp.edge_bytes + sp.bytes)
    return fail(MISPEC_E_INVALID,
                "workspace too small: size it with the *_workspace_bytes query%s");
  unsigned short *xs = reinterpret_cast<unsigned short *>(static_cast<char *>(a->workspace) +
                                                          sp.edge_bytes);
  p.xs = xs;
  p.xs_clip_stride = sp.slot;
  p.xs_plane = sp.slot * p.n_clips;
  p.job_counter = reinterpret_cast<unsigned *>(xs + 2 * sp.slot * p.n_clips);
  p.Ks = round_up_kc(p.K);
  p.as = static_cast<const unsigned short *>(a->basis_split);
  p.as_plane = (long long)p.n_bins * p.Ks;
  p.afrag = basis_has_frags(p.n_bins, p.a_im != nullptr) ? p.as + 4 * p.as_plane : nullptr;
  const unsigned gx = (unsigned)((sp.slot + 1023) / 1024);
  hipLaunchKernelGGL(split_signal_kernel, dim3(gx, (unsigned)p.n_clips), dim3(256), 0, stream, p,
                     xs);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(MISPEC_E_HIP, "signal split launch: %s", hipGetErrorString(err));
  return MISPEC_OK;
}

// the rows [first_bin, n_bins) of a problem as a problem of its own
KParams leftover_rows(const KParams &p, int first_bin) {
  KParams r = p;
  r.n_bins = p.n_bins - first_bin;
  r.a_re = p.a_re + (long long)first_bin * p.a_row_stride;
  if (p.a_im) r.a_im = p.a_im + (long long)first_bin * p.a_row_stride;
  if (p.row_scale) r.row_scale = p.row_scale + first_bin;
  r.out_row_offset = p.out_row_offset + first_bin;
  return r;
}

// How launch_framed_bf16x3 divides the rows: whole 256-row blocks on the bf16 pipe; a leftover
// block joins them when it is at least a quarter full (or carries supports), else its rows (the
// Nyquist bin of an n_fft/2+1 STFT) run as narrow bf16x3 tiles in the tail of the same grid
// (`pair`) or, with a forced tile shape, on the narrow fp32 kernel (the only case that needs
// the fp32 path's edge workspace).
struct Bf16x3Rows {
  int main_bins;
  bool pair, fp32_leftover;
};
Bf16x3Rows plan_bf16x3_rows(const KParams &p, int tile) {
  const int rpb = p.a_im ? 2 : 1;
  const bool masked = p.row_support != nullptr;
  const int bins_per_wg = 256 / rpb;
  Bf16x3Rows r;
  r.main_bins = (p.n_bins / bins_per_wg) * bins_per_wg;
  if (masked || (p.n_bins - r.main_bins) * rpb > 64) r.main_bins = p.n_bins;
  r.pair = tile == MISPEC_TILE_AUTO && !masked && r.main_bins > 0 && r.main_bins < p.n_bins &&
           !MISPEC_DBG(p, 0x2000);
  r.fp32_leftover = !r.pair && r.main_bins != p.n_bins;
  return r;
}

int launch_framed_bf16x3(const KParams &p, int tile, hipStream_t stream,
                         const int32_t *sup_host = nullptr) {
  const int rpb = p.a_im ? 2 : 1;
  const bool masked = p.row_support != nullptr;
  const Bf16x3Rows rows = plan_bf16x3_rows(p, tile);
  const int main_bins = rows.main_bins;
  KParams q = p;
  q.n_bins = main_bins;
  int rc;
  if (rows.pair) {
    // one launch: 256x256 workgroups + narrow ones for the leftover rows in the grid's tail
    KParams r = leftover_rows(p, main_bins);
    r.as = p.as + (long long)main_bins * p.Ks;  // same planes, first leftover bin
    const int rem_rows = r.n_bins * rpb;
    size_t sm_main = bf16x3_smem<4, 2, 2, 4>();
    const bool slab = false;  // dense 256x256 slab tiles do not fit in 256 VGPRs
    const long long gm = prepare_bf16x3<4, 2, 2, 4>(q);
    // (the fused filterbank lives in the planar epilogue: 64-row tiles for the leftover rows too)
    const bool narrow_rem = rem_rows <= 32 && !p.fb;
    const long long gr = narrow_rem ? prepare_bf16x3<1, 8, 1, 1>(r) : prepare_bf16x3<1, 8, 2, 1>(r);
    if (gm < 0 || gr < 0 || gm + gr > 0x7fffffffLL)
      return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
    const size_t sm_rem = narrow_rem ? bf16x3_smem<1, 8, 1, 1>() : bf16x3_smem<1, 8, 2, 1>();
    const size_t smem = sm_main > sm_rem ? sm_main : sm_rem;
    const dim3 grid((unsigned)(gm + gr));
    rc = MISPEC_OK;
#define MISPEC_LAUNCH_PAIR(KERN)                                                              \
  {                                                                                           \
    auto kern = KERN;                                                                         \
    static std::atomic<unsigned long long> configured{0};                                     \
    rc = configure_lds(kern, 160 * 1024, configured);                                         \
    if (rc == MISPEC_OK) hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, q, r, (int)gm); \
  }
    if (narrow_rem) {
      MISPEC_LAUNCH_PAIR(framed_bf16x3_pair_kernel<1>)
    } else {
      MISPEC_LAUNCH_PAIR(framed_bf16x3_pair_kernel<2>)
    }
#undef MISPEC_LAUNCH_PAIR
    if (rc != MISPEC_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
    return MISPEC_OK;
  }
  // Bases with supports (CQT banks): the hop-periodic narrow-tile kernel when the shape allows
  // (framed_bf16x3_narrow.inl; debug bit 0x20000 selects its predecessor with 192x256 masked
  // tiles, framed_bf16x3_slab.inl, for A/B measurements).  Otherwise the staged kernel: 256x256 workgroups of 8 waves (two per SIMD, 64x128 per wave; a 4-wave
  // layout with 128x128 per wave measured 8 % slower and does not fit without scratch).
  size_t sm = 0;
  if (masked && sup_host && tile == MISPEC_TILE_AUTO && main_bins == p.n_bins && p.job_counter && p.afrag &&
      !MISPEC_DBG(p, 0x800000)) {  // (A/B runs: the narrow-tile kernel)
    StripPlan plan;
    const int n_cu = device_cus();
    if (strip_plan_cached(q, sup_host, 2 * n_cu, plan)) {
#ifdef MISPEC_ABLATE
      if (p.debug & 0x1000000) {  // benchmarking: show the plan
        fprintf(stderr, "strip plan: %d passes x %d frame tiles on %d CUs\n", plan.n_pass, plan.n_tiles_n, n_cu);
        for (int i = 0; i < plan.n_pass; ++i) {
          const StripPass &ps = plan.pass[i];
          fprintf(stderr, "  pass %d: cost %d, super-stages [%d, %d), %d slab rows\n", i, ps.cost, ps.jbase,
                  ps.jbase + ps.span, ps.slab_rows);
          for (int w = 0; w < STRIP_NW; ++w)
            fprintf(stderr, "    wave %d: tile %d taps [%d, %d) j [%d, %d) group %d+%d fmask %x\n", w,
                    ps.w[w].tile, ps.w[w].kb, ps.w[w].ke, ps.w[w].ja, ps.w[w].jb, ps.w[w].g0, ps.w[w].gsize,
                    ps.w[w].fmask);
        }
      }
#endif
      return launch_bf16x3_strip(q, plan, n_cu, stream);
    }
  }
  if (masked && q.hop <= 64 * KC && plan_slab<2, 4, 3, 2>(q, sm))
    rc = MISPEC_DBG(p, 0x20000) ? launch_bf16x3_slab_cfg<2, 4, 3, 2, true>(q, sm, stream)  // A/B runs
                             : launch_bf16x3_narrow(q, sm, stream);
  else
    rc = masked ? launch_bf16x3_cfg<4, 2, 2, 4, true>(q, stream)
                : launch_bf16x3_cfg<4, 2, 2, 4, false>(q, stream);
  if (rc != MISPEC_OK || main_bins == p.n_bins) return rc;
  return launch_framed(leftover_rows(p, main_bins), MISPEC_TILE_AUTO, stream);
}

// ---------------------------------------------------------------------------------
// symmetric fold (framed_fold.inl): applicability, workspace, launch
// ---------------------------------------------------------------------------------
inline int fold_arith_of(int precision) {
  return precision == MISPEC_PREC_F32 ? FOLD_F32 : (precision == MISPEC_PREC_F16X3 ? FOLD_F16X3 : FOLD_BF16X3);
}

long long basis_fold_bytes(int n_bins, int kernel, int with_tap0) {
  const long long kf = fold_taps(kernel, with_tap0);
  return (long long)n_bins * kf * 8 + 2 * kf * (long long)sizeof(float);
}

struct FoldPlan {
  bool ok;
  int kf, with_tap0, main_bins;
  bool last_in_prepass;
  long long ws_bytes;
};

FoldPlan plan_fold(const mispec_framed_gemm_args *a, const KParams &p) {
  FoldPlan f{};
  if (!a->basis_fold || a->tile != MISPEC_TILE_AUTO) return f;  // (either precision: the planes' format follows it)
  if (MISPEC_DBG(p, 0x100000)) return f;  // A/B runs: the dense kernel
  if (!p.a_im || p.row_support || (p.K & 1) || p.K < 64) return f;
  if ((long long)p.hop * 8 < p.K) return f;  // folded frames cost 8 B per folded tap and frame
  if (p.K > 8192) return f;                  // the pre-pass assembles 4 frames (4 K bytes each) in LDS
  int with_tap0 = -1;
  for (int w = 0; w < 2; ++w)
    if (a->fold_taps == fold_taps(p.K, w)) with_tap0 = w;
  if (with_tap0 < 0) return f;
  if (fold_taps(p.K, 0) == fold_taps(p.K, 1)) with_tap0 = 1;  // ambiguous: the carried tap is a
                                                              // zero coefficient when not needed
  if (a->basis_fold_bytes < basis_fold_bytes(p.n_bins, p.K, with_tap0)) return f;
  if (p.n_bins < 64) return f;  // a 128-bin tile would be mostly empty
  f.kf = a->fold_taps;
  f.with_tap0 = with_tap0;
  f.last_in_prepass = p.n_bins > FOLD_BINS && p.n_bins % FOLD_BINS == 1;
  f.main_bins = f.last_in_prepass ? p.n_bins - 1 : p.n_bins;
  f.ws_bytes = p.n_cols * (long long)f.kf * 8 +
               (a->precision == MISPEC_PREC_F16X3 ? p.n_cols * (long long)sizeof(float) : 0);
  f.ok = true;
  return f;
}

// Pre-pass and contraction alternate on the caller's stream over chunks of clips whose folded frames
// (~128 MB) stay in the 256 MB Infinity Cache between the pre-pass that writes them and the
// contraction that reads them.  (Measured on the MI355X, cfg2: running the pre-pass of chunk k+1 on
// a side stream beside the contraction of chunk k -- its blocks do fit next to the contraction's --
// was SLOWER than back to back, 0.81 vs 0.78 ms: the contraction loses more to the pre-pass's
// traffic than the overlap hides.)
int launch_fold(KParams p, const mispec_framed_gemm_args *a, const FoldPlan &f, hipStream_t stream) {
  if (!a->workspace || a->workspace_bytes < f.ws_bytes)
    return fail(MISPEC_E_INVALID, "workspace too small: size it with the *_workspace_bytes query%s");
  if (p.n_cols > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  unsigned short *xf = static_cast<unsigned short *>(a->workspace);
  const unsigned short *bf = static_cast<const unsigned short *>(a->basis_fold);
  p.Ks = f.kf;
  p.fold_tap0 = f.with_tap0;
  p.fold_arith = fold_arith_of(a->precision);
  p.fold_f32 = p.fold_arith == FOLD_F32;
  p.as = bf;
  p.xs = xf;
  p.col_unscale = p.fold_arith == FOLD_F16X3
                      ? reinterpret_cast<float *>(static_cast<char *>(a->workspace) + p.n_cols * (long long)f.kf * 8)
                      : nullptr;
  const float *last_rows = reinterpret_cast<const float *>(bf + (long long)p.n_bins * f.kf * 4);
  // pre-pass: folded frames (+ the last bin); it sees the whole problem's epilogue fields
  KParams pre = p;
  pre.fold_last = f.last_in_prepass ? last_rows : nullptr;
  pre.fold_last_bin = p.n_bins - 1;
  const int pre_frames = FOLD_FR * fold_groups(f.kf);  // frames per pre-pass workgroup
  const size_t pre_smem = (size_t)pre_frames * f.kf * 8 + 12 * FOLD_FR * sizeof(float);
  {
    static std::atomic<unsigned long long> configured_pre{0};
    int rc0 = configure_lds(fold_frames_kernel, 160 * 1024, configured_pre);
    if (rc0 != MISPEC_OK) return rc0;
  }
  // main contraction over the whole 128-bin blocks (a partial last block when it is more than the
  // one bin the pre-pass took)
  p.n_bins = f.main_bins;
  p.n_tiles_m = (p.n_bins + FOLD_BINS - 1) / FOLD_BINS;
  const long long tn = (p.n_cols + FOLD_BN - 1) / FOLD_BN;
  if (tn * p.n_tiles_m > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  int g = (32 + p.n_tiles_m / 2) / p.n_tiles_m;  // one workgroup per CU, 32 CUs per XCD
  g = g < 1 ? 1 : g;
  auto kern = p.fold_f32 ? framed_fold32_kernel
                         : (p.fold_arith == FOLD_F16X3 ? framed_fold16_kernel : framed_fold_kernel);
  static std::atomic<unsigned long long> configured3[3] = {{0}, {0}, {0}};
  int rc = configure_lds(kern, 160 * 1024, configured3[p.fold_arith]);
  if (rc != MISPEC_OK) return rc;
  // the epilogues reuse the stage ring (patches: 8 waves x 32 x 132 floats, or the 128 x 260 power
  // tile + band table of the fused filterbank)
  const size_t smem = (size_t)FOLD_NBUF * FOLD_STAGE;
  // chunk = a whole number of 256-workgroup rounds of the contraction, ~128 MB of folded frames
  int q = 256;  // frame tiles that make whole rounds: 256 / gcd(256, n_tiles_m)
  for (int d = p.n_tiles_m; d % 2 == 0 && q > 1; d /= 2) q /= 2;
  const long long tile_bytes = (long long)FOLD_BN * f.kf * 8;
  long long per = (128LL << 20) / (q * tile_bytes);
  per = q * (per > 0 ? per : 1);
  if (!MISPEC_DBG(p, 0x200000)) per = tn;  // one chunk unless asked (A/B runs): see above
  long long tile0 = 0;
  int clip0 = 0;
  while (tile0 < tn) {
    // clips whose frames the next `per` frame tiles need (all the rest when little would remain)
    long long tile1 = tile0 + per;
    if (tn - tile1 < per / 2) tile1 = tn;
    long long clip1 = tile1 >= tn ? p.n_clips : (tile1 * FOLD_BN + p.n_frames - 1) / p.n_frames;
    if (clip1 > p.n_clips) clip1 = p.n_clips;
    if (clip1 > clip0) {
      KParams q1 = pre;
      q1.fold_clip0 = clip0;
      hipLaunchKernelGGL(fold_frames_kernel,
                         dim3((unsigned)((p.n_frames + pre_frames - 1) / pre_frames), (unsigned)(clip1 - clip0)),
                         dim3(256), pre_smem, stream, q1, xf);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return fail(MISPEC_E_HIP, "fold pre-pass launch: %s", hipGetErrorString(e));
      clip0 = (int)clip1;
    }
    KParams q2 = p;
    q2.fold_tile0 = (int)tile0;
    q2.n_tiles_n = (int)(tile1 - tile0);
    long long grid = (tile1 - tile0) * p.n_tiles_m;
    q2.fold_main = (int)grid;
    if (tile0 == 0 && tile1 == tn &&
        !MISPEC_DBG(p, 0x8000000)) {
      // whole rounds of the device on 256-frame tiles, the frames behind them on 128-frame tiles
      // (framed_fold_kernel); less than half a round: 128-frame tiles throughout
      const int n_cu = device_cus();
      const double rounds = (double)grid / n_cu;
      long long main_tn = tn;
      if (rounds <= 0.5) {
        main_tn = 0;
      } else if (rounds > 1.0) {
        const long long whole = (long long)rounds * n_cu / p.n_tiles_m;  // frame tiles of the whole rounds
        if (whole < tn) {
          const long long half = (p.n_cols - whole * FOLD_BN + FOLD_BN / 2 - 1) / (FOLD_BN / 2) * p.n_tiles_m;
          // a 128-frame tile costs ~0.8 of a 256-frame one (measured: the same basis rows staged for
          // half the MFMAs, the same fill and epilogue latencies)
          const double tail = 0.8 * (double)half / n_cu;
          const double mixed = (double)(whole * p.n_tiles_m) / n_cu + (tail > 0.8 ? tail : 0.8);
          if (mixed < (double)(long long)(rounds + 0.999) - 0.05) main_tn = whole;
        }
      }
      if (main_tn < tn) {
        q2.n_tiles_n = (int)main_tn;
        q2.fold_main = (int)(main_tn * p.n_tiles_m);
        q2.fold_tail_frame0 = main_tn * FOLD_BN;
        const long long tail_tn = (p.n_cols - q2.fold_tail_frame0 + FOLD_BN / 2 - 1) / (FOLD_BN / 2);
        grid = q2.fold_main + tail_tn * p.n_tiles_m;
      }
    }
    q2.n_group = g > q2.n_tiles_n ? q2.n_tiles_n : g;
    if (q2.n_group < 1) q2.n_group = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, q2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
    tile0 = tile1;
  }
  return MISPEC_OK;
}

// ---------------------------------------------------------------------------------
// second fold (framed_fold2.inl): applicability, workspace, launch
// ---------------------------------------------------------------------------------
long long basis_fold2_bytes(int n_bins, int kernel) {
  const long long kf = fold2_taps(kernel);
  return (long long)n_bins * kf * 8 + 2 * kf * (long long)sizeof(float);
}

inline bool fold2_kernel_ok(int kernel) { return kernel >= 128 && kernel <= 8192 && kernel % 64 == 0; }

struct Fold2Plan {
  bool ok;
  int kf, ne, no, main_e;
  bool last_in_prepass;
  long long ws_bytes, unscale_off;
};

Fold2Plan plan_fold2(const mispec_framed_gemm_args *a, const KParams &p) {
  Fold2Plan f{};
  if (!a->basis_fold2 || a->tile != MISPEC_TILE_AUTO) return f;
  if (MISPEC_DBG(p, 0x100000) || MISPEC_DBG(p, 0x40000000)) return f;  // A/B runs: dense / single fold
  if (!p.a_im || p.row_support || p.row_scale || p.fb || !fold2_kernel_ok(p.K)) return f;
  if ((long long)p.hop * 8 < p.K) return f;  // the folded frames cost 4 K bytes per frame
  if (a->basis_fold2_bytes < basis_fold2_bytes(p.n_bins, p.K)) return f;
  if (p.n_bins < 128) return f;  // two 128-bin tiles would be mostly empty
  f.kf = fold2_taps(p.K);
  f.ne = (p.n_bins + 1) / 2;
  f.no = p.n_bins / 2;
  f.last_in_prepass = f.ne > FOLD_BINS && f.ne % FOLD_BINS == 1;
  f.main_e = f.last_in_prepass ? f.ne - 1 : f.ne;
  // [even frames | odd frames | col_add: 2 n_cols floats | FOLD_F16X3: col_unscale: n_cols floats]
  f.unscale_off = 2 * p.n_cols * (long long)f.kf * 8 + 2 * p.n_cols * (long long)sizeof(float);
  f.ws_bytes = f.unscale_off + (a->precision == MISPEC_PREC_F16X3 ? p.n_cols * (long long)sizeof(float) : 0);
  f.ok = true;
  return f;
}

int launch_fold2(KParams p, const mispec_framed_gemm_args *a, const Fold2Plan &f, hipStream_t stream) {
  if (!a->workspace || a->workspace_bytes < f.ws_bytes)
    return fail(MISPEC_E_INVALID, "workspace too small: size it with the *_workspace_bytes query%s");
  if (p.n_cols > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  unsigned short *xf = static_cast<unsigned short *>(a->workspace);
  const unsigned short *bf = static_cast<const unsigned short *>(a->basis_fold2);
  p.Ks = f.kf;
  p.fold_arith = fold_arith_of(a->precision);
  p.fold_f32 = p.fold_arith == FOLD_F32;
  p.fold_wmax = a->fold2_wmax > 0.f ? a->fold2_wmax : 1.f;
  p.as = bf;
  p.xs = xf;
  p.fold2_as_odd = (long long)f.ne * f.kf * 4;
  p.fold2_xs_odd = p.n_cols * (long long)f.kf * 4;
  p.col_unscale = p.fold_arith == FOLD_F16X3
                      ? reinterpret_cast<float *>(static_cast<char *>(a->workspace) + f.unscale_off)
                      : nullptr;
  p.col_add = reinterpret_cast<float *>(static_cast<char *>(a->workspace) + 2 * p.n_cols * (long long)f.kf * 8);
  // output rows: the problem's block starts at out_row_offset; even / odd bins interleave from there
  p.out += (long long)p.out_row_offset * p.out_row_stride;
  p.out_row_offset = 0;
  const float *last_rows = reinterpret_cast<const float *>(bf + (long long)p.n_bins * f.kf * 4);
  KParams pre = p;
  pre.fold_last = f.last_in_prepass ? last_rows : nullptr;
  pre.fold_last_bin = 2 * (f.ne - 1);
  const int pre_frames = FOLD2_FR * (256 / fold2_tg(p.K));
  const size_t pre_smem = (size_t)pre_frames * 2 * f.kf * 8 + (2 * 4 * FOLD2_FR + 4) * sizeof(float);
  {
    static std::atomic<unsigned long long> configured_pre{0};
    int rc0 = configure_lds(fold2_frames_kernel, 160 * 1024, configured_pre);
    if (rc0 != MISPEC_OK) return rc0;
  }
  hipLaunchKernelGGL(fold2_frames_kernel,
                     dim3((unsigned)((p.n_frames + pre_frames - 1) / pre_frames), (unsigned)p.n_clips), dim3(256),
                     pre_smem, stream, pre, xf);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "fold pre-pass launch: %s", hipGetErrorString(e));

  p.fold2 = 1;
  p.fold2_bins_e = f.main_e;
  p.fold2_bins_o = f.no;
  p.fold2_tiles_e = (f.main_e + FOLD_BINS - 1) / FOLD_BINS;
  p.n_tiles_m = p.fold2_tiles_e + (f.no + FOLD_BINS - 1) / FOLD_BINS;
  const long long tn = (p.n_cols + FOLD_BN - 1) / FOLD_BN;
  if (tn * p.n_tiles_m > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  int g = (32 + p.n_tiles_m / 2) / p.n_tiles_m;  // one workgroup per CU, 32 CUs per XCD
  g = g < 1 ? 1 : g;
  auto kern = p.fold_arith == FOLD_F32 ? framed_fold32_kernel
                                       : (p.fold_arith == FOLD_F16X3 ? framed_fold16_kernel : framed_fold_kernel);
  static std::atomic<unsigned long long> configured[3] = {{0}, {0}, {0}};
  int rc = configure_lds(kern, 160 * 1024, configured[p.fold_arith]);
  if (rc != MISPEC_OK) return rc;
  const size_t smem = (size_t)FOLD_NBUF * FOLD_STAGE;
  p.fold_tile0 = 0;
  p.n_tiles_n = (int)tn;
  long long grid = tn * p.n_tiles_m;
  p.fold_main = (int)grid;
  {
    // whole rounds of the device on 256-frame tiles, the frames behind them on 128-frame tiles (as
    // launch_fold)
    const int n_cu = device_cus();
    const double rounds = (double)grid / n_cu;
    long long main_tn = tn;
    if (rounds <= 0.5) {
      main_tn = 0;
    } else if (rounds > 1.0) {
      const long long whole = (long long)rounds * n_cu / p.n_tiles_m;
      if (whole < tn) {
        const long long half = (p.n_cols - whole * FOLD_BN + FOLD_BN / 2 - 1) / (FOLD_BN / 2) * p.n_tiles_m;
        const double tail = 0.8 * (double)half / n_cu;
        const double mixed = (double)(whole * p.n_tiles_m) / n_cu + (tail > 0.8 ? tail : 0.8);
        if (mixed < (double)(long long)(rounds + 0.999) - 0.05) main_tn = whole;
      }
    }
    if (main_tn < tn) {
      p.n_tiles_n = (int)main_tn;
      p.fold_main = (int)(main_tn * p.n_tiles_m);
      p.fold_tail_frame0 = main_tn * FOLD_BN;
      const long long tail_tn = (p.n_cols - p.fold_tail_frame0 + FOLD_BN / 2 - 1) / (FOLD_BN / 2);
      grid = p.fold_main + tail_tn * p.n_tiles_m;
    }
  }
  p.n_group = g > p.n_tiles_n ? p.n_tiles_n : g;
  if (p.n_group < 1) p.n_group = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, p);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

// attach the edge workspace to p and enqueue the fill pre-pass
int setup_edges(KParams &p, void *workspace, long long workspace_bytes, hipStream_t stream) {
  const EdgePlan e = plan_edges(p.n_samples, p.K, p.hop, p.pad, p.n_frames);
  p.edge_mode = e.mode;
  p.n_left = e.n_left;
  p.t_r0 = e.t_r0;
  p.edge_ll = (int)e.ll;
  p.edge_clip_stride = e.stride;
  p.edge = nullptr;
  if (e.mode == EDGE_NONE) return MISPEC_OK;
  if (e.ll > 0x7fffffffLL || e.stride > 0x7fffffffLL)
    return fail(MISPEC_E_UNSUPPORTED, "edge span overflows int32%s");
  const long long need = e.stride * p.n_clips * (long long)sizeof(float);
  if (!workspace || workspace_bytes < need)
    return fail(MISPEC_E_INVALID,
                "workspace too small: size it with the *_workspace_bytes query%s");
  p.edge = static_cast<const float *>(workspace);
  const unsigned gx = (unsigned)((e.stride + 255) / 256);
  hipLaunchKernelGGL(edge_fill_kernel, dim3(gx, (unsigned)p.n_clips), dim3(256), 0, stream, p,
                     static_cast<float *>(workspace));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(MISPEC_E_HIP, "edge fill launch: %s", hipGetErrorString(err));
  return MISPEC_OK;
}

int fill_params(const mispec_framed_gemm_args *a, KParams &p) {
  if (!a) return fail(MISPEC_E_INVALID, "args is NULL%s");
  if (a->struct_size != sizeof(mispec_framed_gemm_args))
    return fail(MISPEC_E_INVALID, "struct_size mismatch (ABI skew)%s");
  if (!a->x || !a->basis_re || !a->out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (a->n_clips <= 0 || a->n_samples <= 0 || a->n_frames <= 0 || a->n_bins <= 0 ||
      a->kernel <= 0 || a->hop <= 0 || a->pad < 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (a->pad_mode < MISPEC_PAD_NONE || a->pad_mode > MISPEC_PAD_REFLECT)
    return fail(MISPEC_E_INVALID, "bad pad_mode%s");
  if (a->pad_mode == MISPEC_PAD_REFLECT && a->pad >= a->n_samples)
    return fail(MISPEC_E_INVALID, "reflect padding needs pad < n_samples%s");
  if (a->pad_mode == MISPEC_PAD_NONE && a->pad != 0)
    return fail(MISPEC_E_INVALID, "pad must be 0 with MISPEC_PAD_NONE%s");
  if (a->epilogue < MISPEC_EPI_COMPLEX || a->epilogue > MISPEC_EPI_REAL)
    return fail(MISPEC_E_INVALID, "bad epilogue%s");
  if ((a->epilogue == MISPEC_EPI_REAL) != (a->basis_im == nullptr))
    return fail(MISPEC_E_INVALID, "MISPEC_EPI_REAL <=> basis_im == NULL%s");
  // the last frame must end inside the (virtually padded) signal
  const long long last_end = (long long)(a->n_frames - 1) * a->hop - a->pad + a->kernel;
  if (last_end > (long long)a->n_samples + a->pad)
    return fail(MISPEC_E_INVALID, "n_frames overruns the padded signal%s");
  if ((long long)(a->n_frames - 1) * a->hop + a->kernel + KC > 0x7fffffffLL)
    return fail(MISPEC_E_UNSUPPORTED, "signal position overflows int32%s");

  memset(&p, 0, sizeof(p));
  p.x = a->x;
  p.x_clip_stride = a->x_clip_stride;
  p.n_clips = a->n_clips;
  p.n_samples = a->n_samples;
  p.hop = a->hop;
  p.pad = a->pad;
  p.pad_mode = a->pad_mode;
  p.n_frames = a->n_frames;
  p.n_cols = (long long)a->n_clips * a->n_frames;
  p.a_re = a->basis_re;
  p.a_im = a->basis_im;
  p.a_row_stride = a->basis_row_stride;
  p.n_bins = a->n_bins;
  p.K = a->kernel;
  p.row_support = a->row_support;
  p.row_scale = a->row_scale;
  p.epilogue = a->epilogue;
  p.im_sign = a->im_sign;
  p.eps = a->eps;
  p.power = a->power;
  p.out = a->out;
  p.out_clip_stride = a->out_clip_stride;
  p.out_row_stride = a->out_row_stride;
  p.out_row_offset = a->out_row_offset;
#ifdef MISPEC_ABLATE
  p.debug = a->reserved;
#else
  if (a->reserved != 0)
    return fail(MISPEC_E_INVALID, "reserved must be 0 (ablation bits exist only in libmispec_ablate.so)%s");
#endif
  if (a->precision != MISPEC_PREC_F32 && a->precision != MISPEC_PREC_BF16X3 &&
      a->precision != MISPEC_PREC_F16X3)
    return fail(MISPEC_E_INVALID, "bad precision%s");
  if (a->reserved2 != 0 || a->reserved4 != 0)
    return fail(MISPEC_E_INVALID, "reserved fields must be 0%s");
  if (a->out_frame_major != 0 && a->out_frame_major != 1)
    return fail(MISPEC_E_INVALID, "out_frame_major must be 0 or 1%s");
  p.out_fm = a->out_frame_major;
  if (a->row_support_host) {  // the caller's host copy of the supports: at least well-formed
    if (!a->row_support) return fail(MISPEC_E_INVALID, "row_support_host without row_support%s");
    for (int i = 0; i < a->n_bins; ++i) {
      const int lo = a->row_support_host[2 * i], hi = a->row_support_host[2 * i + 1];
      if (lo < 0 || hi < lo || hi > a->kernel) return fail(MISPEC_E_INVALID, "row_support_host: need 0 <= start <= stop <= kernel%s");
    }
  }
  if (a->fb) {
    if (!a->fb_support || a->n_fb <= 0)
      return fail(MISPEC_E_INVALID, "fused filterbank: fb_support and n_fb > 0 are required%s");
    if (a->epilogue != MISPEC_EPI_POWER || !(a->power == 1.0f || a->power == 2.0f))
      return fail(MISPEC_E_INVALID, "fused filterbank: MISPEC_EPI_POWER with power 1 or 2 only%s");
    if (!a->basis_im || a->row_support || a->out_row_offset != 0)
      return fail(MISPEC_E_INVALID, "fused filterbank: complex dense basis, whole output%s");
    if (a->n_fb > 256) return fail(MISPEC_E_UNSUPPORTED, "fused filterbank: at most 256 filters%s");
    p.fb = a->fb;
    p.fb_support = a->fb_support;
    p.fb_row_stride = a->fb_row_stride;
    p.n_fb = a->n_fb;
  }
  return MISPEC_OK;
}

// 32 consecutive outputs form one "frame" of the Toeplitz contraction:
//   y[32 q + r] = sum_m x[32*stride*q + m - pad] * taps[m - stride*r],  m < n_taps + 31*stride
int fir_params(KParams &p, const float *x, int64_t x_clip_stride, int32_t n_clips,
               int32_t n_samples, const float *taps, int32_t n_taps, int32_t stride, int32_t pad,
               float *y, int64_t y_clip_stride, int32_t n_out) {
  if (n_clips <= 0 || n_samples <= 0 || n_taps <= 0 || stride <= 0 || pad < 0 || n_out <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  const long long span = (long long)n_samples + 2LL * pad - n_taps;
  if (span < 0 || (long long)n_out != span / stride + 1)
    return fail(MISPEC_E_INVALID, "n_out != (n_samples + 2*pad - n_taps)/stride + 1%s");
  if ((long long)n_out * stride + n_taps + 64LL * stride > 0x7fffffffLL)
    return fail(MISPEC_E_UNSUPPORTED, "signal position overflows int32%s");
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.x_clip_stride = x_clip_stride;
  p.n_clips = n_clips;
  p.n_samples = n_samples;
  p.hop = 32 * stride;
  p.pad = pad;
  p.pad_mode = MISPEC_PAD_ZERO;
  p.n_frames = (n_out + 31) / 32;
  p.n_cols = (long long)n_clips * p.n_frames;
  p.a_re = taps;
  p.a_im = nullptr;
  p.n_bins = 32;
  p.K = n_taps + 31 * stride;
  p.toep_stride = stride;
  p.n_taps = n_taps;
  p.epilogue = MISPEC_EPI_REAL;
  p.im_sign = 1.f;
  p.out = y;
  p.out_clip_stride = y_clip_stride;
  p.out_row_stride = 0;
  p.out_len = n_out;
  p.out_frame_stride = 32;
  return MISPEC_OK;
}

}  // namespace

// ---- fp32 strip kernel (MISPEC_PREC_F32 with supports, their host copy and the fp32 fragment-order
// copy of the basis in basis_split): applicability and launch
bool strip32_ok(const mispec_framed_gemm_args *a, const KParams &p, int n_cu, StripPlan &plan) {
  if (a->precision != MISPEC_PREC_F32 || !a->basis_split || !a->row_support || !a->row_support_host ||
      !p.a_im || a->tile != MISPEC_TILE_AUTO || p.fb || MISPEC_DBG(p, 0x800000))
    return false;
  if (!basis_has_frags(p.n_bins, true) ||
      a->basis_split_bytes < (long long)((p.n_bins + 15) / 16) * round_up_kc(p.K) * 128 + 4096)
    return false;
  if (p.n_bins * 2 <= 128) return false;  // (as the split arithmetic: narrow problems stay on the tile kernels)
  KParams q = p;
  q.Ks = round_up_kc(p.K);
  return strip_plan_cached(q, a->row_support_host, 2 * n_cu, plan, true);
}

int launch_strip32(KParams p, const mispec_framed_gemm_args *a, const StripPlan &plan, int n_cu,
                   hipStream_t stream) {
  const EdgePlan e = plan_edges(p.n_samples, p.K, p.hop, p.pad, p.n_frames);
  const SplitPlan sp = plan_split(p, e);
  if (!a->workspace || a->workspace_bytes < sp.edge_bytes + sp.bytes)
    return fail(MISPEC_E_INVALID, "workspace too small: size it with the *_workspace_bytes query%s");
  float *xs = reinterpret_cast<float *>(static_cast<char *>(a->workspace) + sp.edge_bytes);
  p.xs = reinterpret_cast<const unsigned short *>(xs);
  p.xs_clip_stride = sp.slot;
  p.xs_plane = 0;
  p.split_f32 = 1;
  p.job_counter = reinterpret_cast<unsigned *>(xs + sp.slot * p.n_clips);
  p.Ks = round_up_kc(p.K);
  p.afrag = static_cast<const unsigned short *>(a->basis_split);
  const unsigned gx = (unsigned)((sp.slot + 1023) / 1024);
  hipLaunchKernelGGL(split_signal_kernel, dim3(gx, (unsigned)p.n_clips), dim3(256), 0, stream, p,
                     reinterpret_cast<unsigned short *>(xs));
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(MISPEC_E_HIP, "signal pad launch: %s", hipGetErrorString(err));
  p.n_super = (p.Ks + p.hop - 1) / p.hop;
  auto kern = plan.nf == 2 ? framed_f32_strip64_kernel : framed_f32_strip_kernel;
  static std::atomic<unsigned long long> configured{0}, configured64{0};
  int rc = configure_lds(kern, 160 * 1024, plan.nf == 2 ? configured64 : configured);
  if (rc != MISPEC_OK) return rc;
  const unsigned grid = (unsigned)(plan.n_jobs < 2 * n_cu ? plan.n_jobs : 2 * n_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(STRIP_NW * 64), (size_t)STRIP_LDS_BYTES, stream, p, plan);
  err = hipGetLastError();
  if (err != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(err));
  return MISPEC_OK;
}

// ---- MISPEC_PREC_F16X3 on the strip kernel (complex banks with supports and their host copy, basis_split
// = mispec_frag_basis_f16()): applicability and launch
static long long basis_frag16_bytes(int n_bins, int kernel) {
  return (long long)((n_bins + 15) / 16) * round_up_kc(kernel) * 128 + 4096 + 2LL * n_bins * (long long)sizeof(float);
}

bool strip16_ok(const mispec_framed_gemm_args *a, const KParams &p, int n_cu, StripPlan &plan) {
  if (a->precision != MISPEC_PREC_F16X3 || !a->basis_split || !a->row_support || !a->row_support_host ||
      !p.a_im || a->tile != MISPEC_TILE_AUTO || p.fb)
    return false;
  if (!basis_has_frags(p.n_bins, true) || a->basis_split_bytes < basis_frag16_bytes(p.n_bins, p.K)) return false;
  if (p.n_bins * 2 <= 128 || (p.hop & 1)) return false;  // (as bf16x3: narrow problems stay on the fp32 tile kernels)
  KParams q = p;
  q.Ks = round_up_kc(p.K);
  return strip_plan_cached(q, a->row_support_host, 2 * n_cu, plan);
}

// workspace: [edge spans (unused here) | (hi, lo) planes + job counter | per-clip absmax bits]
long long strip16_ws_bytes(const KParams &p, const SplitPlan &sp) {
  return sp.edge_bytes + sp.bytes + p.n_clips * (long long)(CLIP_ABSMAX_STRIDE * sizeof(unsigned));
}

int launch_strip16(KParams p, const mispec_framed_gemm_args *a, const StripPlan &plan, int n_cu,
                   hipStream_t stream) {
  const EdgePlan e = plan_edges(p.n_samples, p.K, p.hop, p.pad, p.n_frames);
  const SplitPlan sp = plan_split(p, e);
  if (!a->workspace || a->workspace_bytes < strip16_ws_bytes(p, sp))
    return fail(MISPEC_E_INVALID, "workspace too small: size it with the *_workspace_bytes query%s");
  unsigned short *xs = reinterpret_cast<unsigned short *>(static_cast<char *>(a->workspace) + sp.edge_bytes);
  p.xs = xs;
  p.xs_clip_stride = sp.slot;
  p.xs_plane = sp.slot * p.n_clips;
  p.job_counter = reinterpret_cast<unsigned *>(xs + 2 * sp.slot * p.n_clips);
  p.clip_absmax = reinterpret_cast<unsigned *>(static_cast<char *>(a->workspace) + sp.edge_bytes + sp.bytes);
  p.split_f16 = 1;
  p.Ks = round_up_kc(p.K);
  p.afrag = static_cast<const unsigned short *>(a->basis_split);
  p.row_unscale = reinterpret_cast<const float *>(
      static_cast<const char *>(a->basis_split) + (long long)((p.n_bins + 15) / 16) * p.Ks * 128 + 4096);
  if (hipMemsetAsync(p.clip_absmax, 0, (size_t)p.n_clips * CLIP_ABSMAX_STRIDE * sizeof(unsigned), stream) != hipSuccess)
    return fail(MISPEC_E_HIP, "hipMemsetAsync failed%s");
  hipLaunchKernelGGL(clip_absmax_kernel, dim3((unsigned)((p.n_samples + ABSMAX_CHUNK - 1) / ABSMAX_CHUNK), (unsigned)p.n_clips),
                     dim3(256), 0, stream, p.x, p.x_clip_stride, p.n_samples, p.clip_absmax);
  const unsigned gx = (unsigned)((sp.slot + 1023) / 1024);
  hipLaunchKernelGGL(split_signal_kernel, dim3(gx, (unsigned)p.n_clips), dim3(256), 0, stream, p, xs);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(MISPEC_E_HIP, "signal split launch: %s", hipGetErrorString(err));
  p.n_super = (p.Ks + p.hop - 1) / p.hop;
  auto kern = plan.nf == 2 ? framed_f16x3_strip64_kernel : framed_f16x3_strip_kernel;
  static std::atomic<unsigned long long> configured{0}, configured64{0};
  int rc = configure_lds(kern, 160 * 1024, plan.nf == 2 ? configured64 : configured);
  if (rc != MISPEC_OK) return rc;
  const unsigned grid = (unsigned)(plan.n_jobs < 2 * n_cu ? plan.n_jobs : 2 * n_cu);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(STRIP_NW * 64), (size_t)STRIP_LDS_BYTES, stream, p, plan);
  err = hipGetLastError();
  if (err != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(err));
  return MISPEC_OK;
}

// ---- MISPEC_PREC_F16X3 on the staged dense kernel (framed_bf16x3.inl, F16 instances): complex bases that
// are not folded -- CQT banks (row supports honoured per 32-row tile), trainable / non-linear STFT bases --
// with basis_split = mispec_split_basis_f16() (row-major planes; the strip kernel's fragment-order copy,
// mispec_frag_basis_f16(), is larger: the size tells the two apart).  Taps in their natural order.
bool dense16_ok(const mispec_framed_gemm_args *a, const KParams &p) {
  if (a->precision != MISPEC_PREC_F16X3 || !a->basis_split || !p.a_im || a->tile != MISPEC_TILE_AUTO || p.fb)
    return false;
  if (a->basis_split_bytes != basis_plane_bytes(p.n_bins, p.K, true) + 2LL * p.n_bins * (long long)sizeof(float))
    return false;
  return p.n_bins * 2 > 128 && !(p.hop & 1);  // (as bf16x3: narrow problems stay on the fp32 tile kernels)
}

int launch_dense16(KParams p, const mispec_framed_gemm_args *a, hipStream_t stream) {
  const EdgePlan e = plan_edges(p.n_samples, p.K, p.hop, p.pad, p.n_frames);
  const SplitPlan sp = plan_split(p, e);
  if (!a->workspace || a->workspace_bytes < strip16_ws_bytes(p, sp))
    return fail(MISPEC_E_INVALID, "workspace too small: size it with the *_workspace_bytes query%s");
  unsigned short *xs = reinterpret_cast<unsigned short *>(static_cast<char *>(a->workspace) + sp.edge_bytes);
  p.xs = xs;
  p.xs_clip_stride = sp.slot;
  p.xs_plane = sp.slot * p.n_clips;
  p.clip_absmax = reinterpret_cast<unsigned *>(static_cast<char *>(a->workspace) + sp.edge_bytes + sp.bytes);
  p.split_f16 = 1;
  p.Ks = round_up_kc(p.K);
  p.as = static_cast<const unsigned short *>(a->basis_split);
  p.as_plane = (long long)p.n_bins * p.Ks;
  p.row_unscale = reinterpret_cast<const float *>(static_cast<const char *>(a->basis_split) +
                                                  basis_plane_bytes(p.n_bins, p.K, true));
  if (hipMemsetAsync(p.clip_absmax, 0, (size_t)p.n_clips * CLIP_ABSMAX_STRIDE * sizeof(unsigned), stream) != hipSuccess)
    return fail(MISPEC_E_HIP, "hipMemsetAsync failed%s");
  hipLaunchKernelGGL(clip_absmax_kernel, dim3((unsigned)((p.n_samples + ABSMAX_CHUNK - 1) / ABSMAX_CHUNK), (unsigned)p.n_clips),
                     dim3(256), 0, stream, p.x, p.x_clip_stride, p.n_samples, p.clip_absmax);
  const unsigned gx = (unsigned)((sp.slot + 1023) / 1024);
  hipLaunchKernelGGL(split_signal_kernel, dim3(gx, (unsigned)p.n_clips), dim3(256), 0, stream, p, xs);
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return fail(MISPEC_E_HIP, "signal split launch: %s", hipGetErrorString(err));
  return p.row_support ? launch_bf16x3_cfg<4, 2, 2, 4, true, true>(p, stream)
                       : launch_bf16x3_cfg<4, 2, 2, 4, false, true>(p, stream);
}

// ---- helpers of the host path (mispec_*_host_f32 below)
namespace {
template <typename F>
void host_parallel_for(long long n, F &&fn) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : (nt > 16 ? 16 : nt);
  if (n < 64 || nt == 1) {
    for (long long i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<long long> next{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&]() {
      for (long long i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
    });
  for (auto &t : th) t.join();
}

inline float host_sample(const float *x, long long pos, int L, int pad_mode) {
  if (pad_mode == MISPEC_PAD_REFLECT) {
    pos = pos < 0 ? -pos : pos;
    pos = pos >= L ? 2LL * L - 2 - pos : pos;
  }
  return (pos >= 0 && pos < L) ? x[pos] : 0.f;
}

inline void host_epilogue(const mispec_framed_gemm_args *a, float *dst, float re, float im) {
  switch (a->epilogue) {
    case MISPEC_EPI_COMPLEX:
      dst[0] = re;
      dst[1] = im;
      break;
    case MISPEC_EPI_MAGNITUDE:
      dst[0] = sqrtf(re * re + im * im + a->eps);
      break;
    case MISPEC_EPI_POWER: {
      const float s2 = re * re + im * im + a->eps;
      dst[0] = (a->power == 2.0f && a->eps == 0.f) ? s2 : (a->power == 1.0f ? sqrtf(s2) : powf(sqrtf(s2), a->power));
    } break;
    case MISPEC_EPI_PHASE_ATAN2:
      dst[0] = atan2f(im + 0.0f, re);
      break;
    case MISPEC_EPI_PHASE_COSSIN: {
      const float ang = atan2f(im, re);
      dst[0] = cosf(ang);
      dst[1] = sinf(ang);
    } break;
    default:
      dst[0] = re;
      break;
  }
}
}  // namespace

// ---------------------------------------------------------------------------------
// FFT path (stft_fft.inl): window x DFT bases (the caller proves the form by handing over the
// mispec_fold2_basis() planes, which that routine only produces after checking the basis numerically) with
// n_fft = 512, 1024 or 2048 and the first n_bins <= n_fft/2 + 1 bins; any pointwise epilogue, any hop and
// padding, with or without the fused filterbank; fp32 arithmetic, so every `precision` is served.  No workspace.
// n_fft = 256 (in the reference's own STFT grid: tests/parameters.py:25) runs on the 512-point instance: the
// frame zero-extended to 512 samples has the 256-point spectrum at every second bin (1.6 - 2.1x the contraction;
// n_fft = 128 the same way measured 0.9x and stays on the contraction kernels).
// ---------------------------------------------------------------------------------
bool fft_ok(const mispec_framed_gemm_args *a, const KParams &p) {
  if (!a->basis_fold2 || a->tile != MISPEC_TILE_AUTO || a->no_fft) return false;
  if (MISPEC_DBG(p, 0x100000) || MISPEC_DBG(p, 0x40000000) || MISPEC_DBG(p, 0x08000000)) return false;  // A/B runs
  if (!p.a_im || p.row_support || p.row_scale || !fold2_kernel_ok(p.K)) return false;
  if (p.fb && (p.epilogue != MISPEC_EPI_POWER || p.out_row_offset != 0)) return false;  // (fused filterbank: in the tile flush)
  if (a->basis_fold2_bytes < basis_fold2_bytes(p.n_bins, p.K)) return false;
  if (p.epilogue < MISPEC_EPI_COMPLEX || p.epilogue > MISPEC_EPI_PHASE_COSSIN) return false;
  if (p.K != 256 && p.K != 512 && p.K != 1024 && p.K != 2048) return false;  // (128: the contraction is faster)
  if (p.n_bins > p.K / 2 + 1 || p.n_frames <= 0) return false;
  return (long long)p.n_clips * p.n_frames <= 0x3fffffffLL;
}

template <int M, int EPI, bool FB, int CEPI = -1, bool FM = false>
int launch_fft_cfg(const KParams &p, hipStream_t stream) {
  constexpr int W = (EPI == MISPEC_EPI_COMPLEX || EPI == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
  constexpr int FT = fft_tile_frames<M, W, FB>();
  const int tiles_per_clip = (p.n_frames + FT - 1) / FT;
  const long long n_tiles = (long long)p.n_clips * tiles_per_clip;
  constexpr int per_cu = fft_two_per_cu<M, W, FB>() ? 2 : 1;  // co-resident workgroups
  const size_t lds_share = 160 * 1024 / per_cu;
  long long grid = n_tiles < per_cu * device_cus() ? n_tiles : per_cu * device_cus();
  grid = (grid + 7) / 8 * 8;
  auto kern = stft_fft_kernel<M, EPI, FB, CEPI, FM>;
  static std::atomic<unsigned long long> configured{0};
  constexpr size_t smem0 = (stft_fft_smem<M, W, FB>() + 15) & ~(size_t)15;
  // fused filterbank: the band weights (a mel bank has ~8 non-zeros per filter) packed into whatever LDS the
  // instance leaves, up to 16 KB -- the tile flush then reads LDS only (they were L2 loads inside its loop)
  KParams q = p;
  q.fb_lds_floats = 0;
  q.fft_row_step = 2 * M / p.K;
  if (FB && smem0 + 2080 + 2048 <= lds_share) {
    const size_t room = lds_share - smem0 - 2080;  // (offsets of up to 256 filters + 1, a flag, their first bins)
    q.fb_lds_floats = (int)((room < 16384 ? room : 16384) / 4);
  }
  const size_t smem = smem0 + (q.fb_lds_floats ? 2080 + (size_t)q.fb_lds_floats * 4 : 0);
#if MISPEC_FFT_STAMPS
  if (const char *sp = getenv("MISPEC_FFT_STAMPS")) q.job_counter = reinterpret_cast<unsigned *>(strtoull(sp, nullptr, 0));
#endif
  int rc = configure_lds(kern, 160 * 1024, configured);
  if (rc != MISPEC_OK) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(fft_waves<M, W>() * 64), smem, stream, q, tiles_per_clip);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

template <int M>
int launch_fft_size(const KParams &p, hipStream_t stream) {
  switch (p.epilogue) {
    case MISPEC_EPI_COMPLEX:
      return launch_fft_cfg<M, MISPEC_EPI_COMPLEX, false>(p, stream);
    case MISPEC_EPI_MAGNITUDE:
      return launch_fft_cfg<M, MISPEC_EPI_MAGNITUDE, false>(p, stream);
    case MISPEC_EPI_POWER:  // (fft_ok: a filterbank comes with this epilogue only)
      return p.fb ? launch_fft_cfg<M, MISPEC_EPI_POWER, true>(p, stream)
                  : launch_fft_cfg<M, MISPEC_EPI_POWER, false>(p, stream);
    case MISPEC_EPI_PHASE_ATAN2:
      return launch_fft_cfg<M, MISPEC_EPI_PHASE_ATAN2, false>(p, stream);
    default:
      return launch_fft_cfg<M, MISPEC_EPI_PHASE_COSSIN, false>(p, stream);
  }
}

// frame-major output (mispec.h, out_frame_major): the power spectrum of a frame leaves the post-processing's registers as
// one contiguous row -- what a contraction over the bins wants as its framed operand
int launch_fft_frame_major(const KParams &p, hipStream_t stream) {
  if (p.epilogue != MISPEC_EPI_POWER || p.fb || (p.K != 1024 && p.K != 2048) || p.n_bins != p.K / 2 + 1 || p.out_row_offset != 0 ||
      p.out_row_stride < p.n_bins || p.out_row_stride > p.K / 2 + 64)
    return fail(MISPEC_E_UNSUPPORTED, "out_frame_major: kernel 1024 / 2048, all kernel/2 + 1 bins, MISPEC_EPI_POWER, no fused "
                                      "filterbank, out_row_offset 0, n_bins <= out_row_stride <= kernel/2 + 64%s");
  return p.K == 2048 ? launch_fft_cfg<1024, MISPEC_EPI_POWER, false, -1, true>(p, stream)
                     : launch_fft_cfg<512, MISPEC_EPI_POWER, false, -1, true>(p, stream);
}

int launch_fft(const KParams &p, hipStream_t stream) {
  return p.K == 2048 ? launch_fft_size<1024>(p, stream)
                     : (p.K == 1024 ? launch_fft_size<512>(p, stream) : launch_fft_size<256>(p, stream));  // (128, 256: zero-extended)
}

// ---------------------------------------------------------------------------------
// n_fft = 4096 on the FFT route (round 5; VERDICT r4: STFT(n_fft = 4096) fell to the contraction kernels at ~3 x the
// cost) -- COMPOSITE, decimation in time over the 2048-point instance:
//   y_e[n] = y[2n], y_o[n] = y[2n+1]  (y = the padded clip, w the window)
//   E = RFFT_2048(w_e y_e), O = RFFT_2048(w_o y_o)   two launches of stft_fft_kernel<1024, MISPEC_EPI_COMPLEX>
//   X[k] = E[k] + W^k O[k],  X[2048 - k] = conj(E[k] - W^k O[k]),  W = e^(-2 pi i / 4096),  k = 0 .. 1024
// Launches: fft4096_split_kernel (virtual padding applied, clips de-interleaved into the workspace -- reflect padding does
// not commute with the de-interleave when the clip length is even, so the padded streams are materialised --, the window's
// even / odd taps), then the two transforms (pad 0, hop / 2): the first writes E as a complex spectrogram into the workspace,
// the second keeps O in its LDS tile and its flush (stft_fft.inl, flush_cmb) reads E, forms the butterfly above and stores
// rows k and 2048 - k of the caller's output through the caller's pointwise epilogue.  E passes through HBM once each way
// (~2.3 x the bytes of a native instance); hop and pad must be even, no fused filterbank.
// Workspace: mispec_framed_gemm_workspace_bytes().
// ---------------------------------------------------------------------------------
namespace {
struct Fft4096Plan {
  bool ok;
  long long Lh, slot, off_xo, off_w, off_E, off_O, bytes;
};

Fft4096Plan plan_fft4096(const mispec_framed_gemm_args *a, const KParams &p) {
  Fft4096Plan pl = {};
  if (!a->basis_fold2 || a->tile != MISPEC_TILE_AUTO || a->no_fft || p.K != 4096) return pl;
  if (MISPEC_DBG(p, 0x100000) || MISPEC_DBG(p, 0x40000000) || MISPEC_DBG(p, 0x08000000)) return pl;  // A/B runs
  if (!p.a_im || p.row_support || p.row_scale || p.fb || (p.hop & 1) || (p.pad & 1)) return pl;
  if (a->basis_fold2_bytes < basis_fold2_bytes(p.n_bins, p.K)) return pl;
  // (the (cos, sin) phase format -- CQT's, never an STFT module's -- spills in the second transform's flush: contraction kernels)
  if (p.epilogue < MISPEC_EPI_COMPLEX || p.epilogue >= MISPEC_EPI_PHASE_COSSIN) return pl;
  if (p.n_bins > 2049 || p.n_frames <= 0 || (long long)p.n_clips * p.n_frames > 0x3fffffffLL) return pl;
  if (p.n_clips > 65535) return pl;  // (the split pre-pass puts the clips on gridDim.y; the contraction kernels take larger batches)
  const long long Lp = (long long)p.n_samples + 2LL * p.pad;
  pl.Lh = (Lp + 1) / 2;
  if (pl.Lh + 64 > 0x7fffffffLL) return pl;
  pl.slot = (pl.Lh + 63) / 64 * 64;
  const long long spec = (long long)p.n_clips * 1025 * p.n_frames * 2;  // floats of E (and of O)
  pl.off_xo = pl.slot * p.n_clips;
  pl.off_w = 2 * pl.off_xo;
  pl.off_E = pl.off_w + 4096;
  pl.off_O = 0;  // (the odd samples' spectrum never leaves the second transform's tile)
  pl.bytes = (pl.off_E + spec) * (long long)sizeof(float);
  pl.ok = true;
  return pl;
}

__global__ void __launch_bounds__(256) fft4096_split_kernel(const float *__restrict__ x, long long x_clip_stride, int L, int pad,
                                                            int pad_mode, const float *__restrict__ win, float *__restrict__ xe,
                                                            float *__restrict__ xo, long long slot, long long Lh,
                                                            float *__restrict__ w) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (b == 0 && j < 2048) {  // row 0 of the cosine kernels is the window itself
    w[j] = win[2 * j];
    w[2048 + j] = win[2 * j + 1];
  }
  if (j >= slot) return;
  float v[2] = {0.f, 0.f};
  if (j < Lh) {
    const float *xc = x + (long long)b * x_clip_stride;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      long long q = 2 * j + e - pad;
      if (pad_mode == MISPEC_PAD_REFLECT) {
        q = q < 0 ? -q : q;
        q = q >= L ? 2LL * L - 2 - q : q;
      }
      v[e] = (q >= 0 && q < L) ? xc[q] : 0.f;
    }
  }
  xe[(long long)b * slot + j] = v[0];
  xo[(long long)b * slot + j] = v[1];
}

int launch_fft4096(const KParams &p, const mispec_framed_gemm_args *a, const Fft4096Plan &pl, hipStream_t stream) {
  if (!a->workspace || a->workspace_bytes < pl.bytes)
    return fail(MISPEC_E_INVALID, "workspace too small: size it with the *_workspace_bytes query%s");
  float *const ws = static_cast<float *>(a->workspace);
  float *const xe = ws, *const xo = ws + pl.off_xo, *const w = ws + pl.off_w, *const E = ws + pl.off_E;
  hipLaunchKernelGGL(fft4096_split_kernel, dim3((unsigned)((pl.slot + 255) / 256), (unsigned)p.n_clips), dim3(256), 0, stream, p.x,
                     p.x_clip_stride, p.n_samples, p.pad, p.pad_mode, p.a_re, xe, xo, pl.slot, pl.Lh, w);
  {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MISPEC_E_HIP, "fft4096 split launch: %s", hipGetErrorString(e));
  }
  for (int h = 0; h < 2; ++h) {
    KParams q = p;
    q.x = h ? xo : xe;
    q.x_clip_stride = pl.slot;
    q.n_samples = (int)pl.Lh;
    q.hop = p.hop / 2;
    q.pad = 0;
    q.pad_mode = MISPEC_PAD_NONE;
    q.K = 2048;
    q.a_re = w + 2048 * h;
    q.a_im = w + 2048 * h;  // (not read)
    q.a_row_stride = 2048;
    q.fb = nullptr;
    if (h == 0) {  // E = RFFT_2048(w_e y_e): (re, im) of the DFT itself into the workspace
      q.n_bins = 1025;
      q.epilogue = MISPEC_EPI_COMPLEX;
      q.im_sign = -1.f;
      q.eps = 0.f;
      q.out = E;
      q.out_clip_stride = 1025LL * p.n_frames * 2;
      q.out_row_stride = 2LL * p.n_frames;
      q.out_row_offset = 0;
    } else {       // O stays in the transform's tile: its flush reads E and stores the caller's rows k and 2048 - k (the caller's
      q.cmb_E = E;  // epilogue, sign, bins and output strides are the ones already in q)
    }
    int rc;
    if (h == 0) {
      rc = launch_fft_cfg<1024, MISPEC_EPI_COMPLEX, false>(q, stream);
    } else {
      switch (p.epilogue) {
        case MISPEC_EPI_COMPLEX: rc = launch_fft_cfg<1024, MISPEC_EPI_COMPLEX, false, MISPEC_EPI_COMPLEX>(q, stream); break;
        case MISPEC_EPI_MAGNITUDE: rc = launch_fft_cfg<1024, MISPEC_EPI_COMPLEX, false, MISPEC_EPI_MAGNITUDE>(q, stream); break;
        case MISPEC_EPI_POWER: rc = launch_fft_cfg<1024, MISPEC_EPI_COMPLEX, false, MISPEC_EPI_POWER>(q, stream); break;
        default: rc = launch_fft_cfg<1024, MISPEC_EPI_COMPLEX, false, MISPEC_EPI_PHASE_ATAN2>(q, stream); break;
      }
    }
    if (rc != MISPEC_OK) return rc;
  }
  return MISPEC_OK;
}
}  // namespace

// MISPEC_PREC_F16X3 exists on the folded contractions, on the strip kernel and on the staged dense kernel
// (complex bases of more than 64 bins): every other shape runs in MISPEC_PREC_F32 on the tile kernels
// (operands prepared for MISPEC_PREC_F16X3 are not offered to them)
static bool f16_downgrade(const mispec_framed_gemm_args *a, const KParams &p, mispec_framed_gemm_args &local) {
  if (a->precision != MISPEC_PREC_F16X3 || plan_fold2(a, p).ok || plan_fold(a, p).ok) return false;
  StripPlan plan;
  if (strip16_ok(a, p, device_cus(), plan) || dense16_ok(a, p)) return false;
  local = *a;
  local.precision = MISPEC_PREC_F32;
  local.basis_fold2 = nullptr;
  local.basis_fold2_bytes = 0;
  local.basis_fold = nullptr;
  local.basis_fold_bytes = 0;
  local.basis_split = nullptr;
  local.basis_split_bytes = 0;
  return true;
}

// shared with the other translation units of the library (mispec_internal.h; hidden symbols)
int mispec_fail_msg(int code, const char *msg) { return fail(code, "%s", msg); }
int mispec_device_cus() { return device_cus(); }

extern "C" {

int mispec_version(void) { return MISPEC_ABI_VERSION; }

const char *mispec_last_error(void) { return g_err; }

int64_t mispec_framed_gemm_workspace_bytes(const mispec_framed_gemm_args *args) {
  KParams p;
  int rc = fill_params(args, p);
  if (rc != MISPEC_OK) return rc;
  if (fft_ok(args, p)) return 0;
  {
    const Fft4096Plan f4 = plan_fft4096(args, p);
    if (f4.ok) return f4.bytes;
  }
  mispec_framed_gemm_args local;
  if (f16_downgrade(args, p, local)) args = &local;
  if (mispec_chain_ok(args)) return 0;  // (the chain kernel resolves the virtual padding in its loads)
  const Fold2Plan f2 = plan_fold2(args, p);
  if (f2.ok) return f2.ws_bytes;
  const FoldPlan f = plan_fold(args, p);
  if (f.ok) return f.ws_bytes;
  const EdgePlan e = plan_edges(p.n_samples, p.K, p.hop, p.pad, p.n_frames);
  if (bf16x3_ok(args, p)) {
    const SplitPlan sp = plan_split(p, e);
    return sp.edge_bytes + sp.bytes;
  }
  {
    StripPlan plan;
    if (strip16_ok(args, p, device_cus(), plan) || dense16_ok(args, p)) return strip16_ws_bytes(p, plan_split(p, e));
  }
  {
    StripPlan plan;
    if (strip32_ok(args, p, device_cus(), plan)) {  // the padded fp32 copy of the clips
      const SplitPlan sp = plan_split(p, e);
      return sp.edge_bytes + sp.bytes;
    }
  }
  return e.stride * p.n_clips * (int64_t)sizeof(float);
}

int32_t mispec_strip_plan(const mispec_framed_gemm_args *args, int32_t n_cu, int32_t *plan_out,
                          int32_t cap) {
  KParams p;
  int rc = fill_params(args, p);
  if (rc != MISPEC_OK) return rc;
  if (n_cu <= 0 || cap < 0 || (cap > 0 && !plan_out)) return fail(MISPEC_E_INVALID, "bad plan buffer%s");
  mispec_framed_gemm_args local;
  if (f16_downgrade(args, p, local)) args = &local;
  StripPlan plan;
  if (args->precision == MISPEC_PREC_F16X3) {
    if (!strip16_ok(args, p, n_cu, plan)) return 0;
  } else if (args->precision == MISPEC_PREC_F32) {
    if (!strip32_ok(args, p, n_cu, plan)) return 0;
  } else {
    if (!bf16x3_ok(args, p) || !args->row_support || !args->row_support_host ||
        args->tile != MISPEC_TILE_AUTO || !basis_has_frags(p.n_bins, p.a_im != nullptr))
      return 0;
    p.Ks = round_up_kc(p.K);
    if (!strip_plan_cached(p, args->row_support_host, 2 * n_cu, plan)) return 0;
  }
  int n = 0;
  auto put = [&](int v) {
    if (n < cap) plan_out[n] = v;
    ++n;
  };
  put(plan.n_tiles_n);
  for (int i = 0; i < plan.n_pass; ++i) {
    const StripPass &ps = plan.pass[i];
    put(ps.cost), put(ps.jbase), put(ps.span), put(ps.slab_rows);
    for (int w = 0; w < STRIP_NW; ++w) {
      const StripWave &sw = ps.w[w];
      put(sw.tile), put(sw.kb), put(sw.ke), put(sw.ja), put(sw.jb), put(sw.g0), put(sw.gsize), put(sw.fmask);
    }
  }
  return plan.n_pass;
}

int mispec_framed_gemm_f32(const mispec_framed_gemm_args *args, void *stream) {
  KParams p;
  int rc = fill_params(args, p);
  if (rc != MISPEC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p.out_fm) {
    if (!fft_ok(args, p)) return fail(MISPEC_E_UNSUPPORTED, "out_frame_major: served by the FFT path only (basis_fold2, automatic tile, no_fft = 0)%s");
    return launch_fft_frame_major(p, s);
  }
  if (fft_ok(args, p)) return launch_fft(p, s);
  {
    const Fft4096Plan f4 = plan_fft4096(args, p);
    if (f4.ok) return launch_fft4096(p, args, f4, s);
  }
  mispec_framed_gemm_args local;
  if (f16_downgrade(args, p, local)) args = &local;
  if (p.fb && (args->tile != MISPEC_TILE_AUTO || MISPEC_DBG(p, 0x2000)))
    return fail(MISPEC_E_UNSUPPORTED, "fused filterbank needs the automatic tile choice%s");
  // CQT1992v2 in fp32 with the bank's chain copy: LDS delay lines instead of per-stage frame gathers (cqt_chain.hip)
  if (mispec_chain_ok(args)) return mispec_chain_launch(args, (p.debug >> 24) & 15, s);
  const Fold2Plan fold2 = plan_fold2(args, p);
  if (fold2.ok) return launch_fold2(p, args, fold2, s);
  // the fused filterbank adds into its output: cleared here (in-kernel clearing by the fold's pre-pass --
  // 32-byte row segments per workgroup -- cost 1 ms on cfg3)
  if (p.fb && hipMemsetAsync(p.out, 0, (size_t)p.n_clips * p.out_clip_stride * sizeof(float), s) != hipSuccess)
    return fail(MISPEC_E_HIP, "hipMemsetAsync failed%s");
  const FoldPlan fold = plan_fold(args, p);
  if (fold.ok) return launch_fold(p, args, fold, s);
  const bool bf16x3 = bf16x3_ok(args, p);
  if (!bf16x3) {
    StripPlan plan;
    const int n_cu = device_cus();
    if (strip16_ok(args, p, n_cu, plan)) return launch_strip16(p, args, plan, n_cu, s);
    if (dense16_ok(args, p)) return launch_dense16(p, args, s);
    if (strip32_ok(args, p, n_cu, plan)) return launch_strip32(p, args, plan, n_cu, s);
  }
  // (the bf16x3 kernels read the padded split signal, not the fp32 path's edge workspace)
  if (!bf16x3 || plan_bf16x3_rows(p, args->tile).fp32_leftover) {
    rc = setup_edges(p, args->workspace, args->workspace_bytes, s);
    if (rc != MISPEC_OK) return rc;
  }
  if (bf16x3) {
    rc = setup_split(p, args, s);
    if (rc != MISPEC_OK) return rc;
    return launch_framed_bf16x3(p, args->tile, s, args->row_support_host);
  }
  return launch_framed(p, args->tile, s);
}

int64_t mispec_basis_split_bytes(int32_t n_bins, int32_t kernel, int32_t has_im) {
  if (n_bins <= 0 || kernel <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  return basis_split_bytes(n_bins, kernel, has_im != 0);
}

int mispec_split_basis_bf16(const float *basis_re, const float *basis_im,
                            int64_t basis_row_stride, int32_t n_bins, int32_t kernel, void *dst,
                            int64_t dst_bytes, void *stream) {
  if (!basis_re || !dst) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_bins <= 0 || kernel <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (dst_bytes < basis_split_bytes(n_bins, kernel, basis_im != nullptr))
    return fail(MISPEC_E_INVALID, "dst too small: size it with mispec_basis_split_bytes%s");
  const int ks = round_up_kc(kernel);
  unsigned short *frag = nullptr;
  if (basis_has_frags(n_bins, basis_im != nullptr)) {
    // (rows past the last bin of the last 16-bin tile stay zero)
    const long long planes = basis_plane_bytes(n_bins, kernel, true);
    frag = reinterpret_cast<unsigned short *>(static_cast<char *>(dst) + planes);
    hipError_t e0 = hipMemsetAsync(frag, 0, (size_t)(basis_split_bytes(n_bins, kernel, true) - planes),
                                   static_cast<hipStream_t>(stream));
    if (e0 != hipSuccess) return fail(MISPEC_E_HIP, "basis split memset: %s", hipGetErrorString(e0));
  }
  hipLaunchKernelGGL(split_basis_kernel, dim3((unsigned)((ks + 255) / 256), (unsigned)n_bins,
                                              basis_im ? 2u : 1u),
                     dim3(256), 0, static_cast<hipStream_t>(stream), basis_re, basis_im,
                     (long long)basis_row_stride, n_bins, kernel, ks,
                     static_cast<unsigned short *>(dst), frag, static_cast<float *>(nullptr));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "basis split launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int64_t mispec_basis_frag_bytes(int32_t n_bins, int32_t kernel) {
  if (n_bins <= 0 || kernel <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!basis_has_frags(n_bins, true)) return fail(MISPEC_E_UNSUPPORTED, "more than 1024 bins%s");
  return (long long)((n_bins + 15) / 16) * round_up_kc(kernel) * 128 + 4096;
}

int mispec_frag_basis_f32(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, void *dst, int64_t dst_bytes, void *stream) {
  if (!basis_re || !basis_im || !dst) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  const int64_t need = mispec_basis_frag_bytes(n_bins, kernel);
  if (need < 0) return (int)need;
  if (dst_bytes < need) return fail(MISPEC_E_INVALID, "dst too small: size it with mispec_basis_frag_bytes%s");
  const int ks = round_up_kc(kernel);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // (rows past the last bin of the last 16-bin tile, and the block behind the tiles, stay zero)
  if (hipMemsetAsync(dst, 0, (size_t)need, s) != hipSuccess) return fail(MISPEC_E_HIP, "hipMemsetAsync failed%s");
  hipLaunchKernelGGL(split_basis_kernel, dim3((unsigned)((ks + 255) / 256), (unsigned)n_bins, 2u), dim3(256), 0,
                     s, basis_re, basis_im, (long long)basis_row_stride, n_bins, kernel, ks,
                     static_cast<unsigned short *>(nullptr), static_cast<unsigned short *>(nullptr),
                     static_cast<float *>(dst));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "basis fragment launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int64_t mispec_basis_chain_bytes(const int32_t *row_support_host, int32_t n_bins, int32_t kernel) {
  if (!row_support_host || n_bins <= 0 || kernel <= 0) return fail(MISPEC_E_INVALID, "chain basis: supports and positive sizes%s");
  for (int i = 0; i < n_bins; ++i) {
    const int lo = row_support_host[2 * i], hi = row_support_host[2 * i + 1];
    if (lo < 0 || hi < lo || hi > kernel) return fail(MISPEC_E_INVALID, "row_support_host: need 0 <= start <= stop <= kernel%s");
  }
  const int64_t b = mispec_chain_bytes_impl(row_support_host, n_bins, kernel);
  if (b < 0) return fail(MISPEC_E_UNSUPPORTED, "chain basis: the supports of the 16-row tiles do not nest, or more than 576 bins%s");
  return b;
}

int mispec_chain_basis_f32(const float *basis_re, const float *basis_im, int64_t basis_row_stride, int32_t n_bins,
                           int32_t kernel, const int32_t *row_support_host, void *dst, int64_t dst_bytes, void *stream) {
  if (!basis_re || !basis_im || !dst) return fail(MISPEC_E_INVALID, "NULL pointer%s");
  const int64_t need = mispec_basis_chain_bytes(row_support_host, n_bins, kernel);
  if (need < 0) return (int)need;
  if (basis_row_stride < kernel) return fail(MISPEC_E_INVALID, "basis_row_stride < kernel%s");
  return mispec_chain_pack_impl(basis_re, basis_im, basis_row_stride, n_bins, kernel, row_support_host, dst, dst_bytes, stream);
}

int64_t mispec_basis_split16_bytes(int32_t n_bins, int32_t kernel) {
  if (n_bins <= 0 || kernel <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  return basis_plane_bytes(n_bins, kernel, true) + 2LL * n_bins * (long long)sizeof(float);
}

int mispec_split_basis_f16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                           int32_t n_bins, int32_t kernel, void *dst, int64_t dst_bytes, void *stream) {
  if (!basis_re || !basis_im || !dst) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  const int64_t need = mispec_basis_split16_bytes(n_bins, kernel);
  if (need < 0) return (int)need;
  if (dst_bytes < need) return fail(MISPEC_E_INVALID, "dst too small: size it with mispec_basis_split16_bytes%s");
  const int ks = round_up_kc(kernel);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *unscale = reinterpret_cast<float *>(static_cast<char *>(dst) + basis_plane_bytes(n_bins, kernel, true));
  float *scale = unscale + n_bins;
  hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)n_bins), dim3(256), 0, s, basis_re, basis_im,
                     (long long)basis_row_stride, kernel, scale, unscale);
  hipLaunchKernelGGL(split_basis_kernel, dim3((unsigned)((ks + 255) / 256), (unsigned)n_bins, 2u), dim3(256), 0,
                     s, basis_re, basis_im, (long long)basis_row_stride, n_bins, kernel, ks,
                     static_cast<unsigned short *>(dst), static_cast<unsigned short *>(nullptr),
                     static_cast<float *>(nullptr), static_cast<const float *>(scale));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "basis split launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int64_t mispec_basis_frag16_bytes(int32_t n_bins, int32_t kernel) {
  if (n_bins <= 0 || kernel <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!basis_has_frags(n_bins, true)) return fail(MISPEC_E_UNSUPPORTED, "more than 1024 bins%s");
  return basis_frag16_bytes(n_bins, kernel);
}

int mispec_frag_basis_f16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, void *dst, int64_t dst_bytes, void *stream) {
  if (!basis_re || !basis_im || !dst) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  const int64_t need = mispec_basis_frag16_bytes(n_bins, kernel);
  if (need < 0) return (int)need;
  if (dst_bytes < need) return fail(MISPEC_E_INVALID, "dst too small: size it with mispec_basis_frag16_bytes%s");
  const int ks = round_up_kc(kernel);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(dst, 0, (size_t)need, s) != hipSuccess) return fail(MISPEC_E_HIP, "hipMemsetAsync failed%s");
  float *unscale = reinterpret_cast<float *>(static_cast<char *>(dst) + (long long)((n_bins + 15) / 16) * ks * 128 + 4096);
  float *scale = unscale + n_bins;
  hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)n_bins), dim3(256), 0, s, basis_re, basis_im,
                     (long long)basis_row_stride, kernel, scale, unscale);
  hipLaunchKernelGGL(split_basis_kernel, dim3((unsigned)((ks + 255) / 256), (unsigned)n_bins, 2u), dim3(256), 0,
                     s, basis_re, basis_im, (long long)basis_row_stride, n_bins, kernel, ks,
                     static_cast<unsigned short *>(nullptr), static_cast<unsigned short *>(dst),
                     static_cast<float *>(nullptr), static_cast<const float *>(scale));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "basis fragment launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int32_t mispec_fold_taps(int32_t kernel, int32_t with_tap0) {
  if (kernel < 64 || (kernel & 1)) return fail(MISPEC_E_UNSUPPORTED, "the fold needs an even kernel of >= 64 taps%s");
  return fold_taps(kernel, with_tap0 != 0);
}

int64_t mispec_basis_fold_bytes(int32_t n_bins, int32_t kernel, int32_t with_tap0) {
  if (n_bins <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (kernel < 64 || (kernel & 1)) return fail(MISPEC_E_UNSUPPORTED, "the fold needs an even kernel of >= 64 taps%s");
  return basis_fold_bytes(n_bins, kernel, with_tap0 != 0);
}

static int fold_basis_any(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, int32_t with_tap0, void *dst,
                          int64_t dst_bytes, float *stats, void *stream, int arith) {
  if (!basis_re || !basis_im || !dst) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_bins <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (kernel < 64 || (kernel & 1)) return fail(MISPEC_E_UNSUPPORTED, "the fold needs an even kernel of >= 64 taps%s");
  const int w0 = with_tap0 != 0;
  if (dst_bytes < basis_fold_bytes(n_bins, kernel, w0))
    return fail(MISPEC_E_INVALID, "dst too small: size it with mispec_basis_fold_bytes%s");
  const int kf = fold_taps(kernel, w0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned short *d = static_cast<unsigned short *>(dst);
  float *last_rows = reinterpret_cast<float *>(d + (long long)n_bins * kf * 4);
  // statistics land in the caller's buffer, or in the tail rows' place holder when not wanted
  unsigned *st = reinterpret_cast<unsigned *>(stats);
  if (st && hipMemsetAsync(st, 0, 2 * sizeof(unsigned), s) != hipSuccess)
    return fail(MISPEC_E_HIP, "hipMemsetAsync failed%s");
  if (!st) return fail(MISPEC_E_INVALID, "stats must point to 2 device floats%s");
  hipLaunchKernelGGL(fold_basis_kernel, dim3((unsigned)((kf + 255) / 256), (unsigned)n_bins), dim3(256), 0,
                     s, basis_re, basis_im, (long long)basis_row_stride, n_bins, kernel, w0, kf, d,
                     last_rows, st, arith);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "basis fold launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_fold_basis_bf16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                           int32_t n_bins, int32_t kernel, int32_t with_tap0, void *dst,
                           int64_t dst_bytes, float *stats, void *stream) {
  return fold_basis_any(basis_re, basis_im, basis_row_stride, n_bins, kernel, with_tap0, dst, dst_bytes,
                        stats, stream, FOLD_BF16X3);
}

int mispec_fold_basis_f16(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, int32_t with_tap0, void *dst,
                          int64_t dst_bytes, float *stats, void *stream) {
  return fold_basis_any(basis_re, basis_im, basis_row_stride, n_bins, kernel, with_tap0, dst, dst_bytes,
                        stats, stream, FOLD_F16X3);
}

int mispec_fold_basis_f32(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                          int32_t n_bins, int32_t kernel, int32_t with_tap0, void *dst,
                          int64_t dst_bytes, float *stats, void *stream) {
  return fold_basis_any(basis_re, basis_im, basis_row_stride, n_bins, kernel, with_tap0, dst, dst_bytes,
                        stats, stream, FOLD_F32);
}

int64_t mispec_basis_fold2_bytes(int32_t n_bins, int32_t kernel) {
  if (n_bins <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!fold2_kernel_ok(kernel))
    return fail(MISPEC_E_UNSUPPORTED, "the second fold needs a kernel of 128 .. 8192 taps, a multiple of 64%s");
  return basis_fold2_bytes(n_bins, kernel);
}

int mispec_fold2_basis(const float *basis_re, const float *basis_im, int64_t basis_row_stride,
                       int32_t n_bins, int32_t kernel, int32_t precision, void *dst, int64_t dst_bytes,
                       float *stats, void *stream) {
  if (!basis_re || !basis_im || !dst || !stats) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_bins <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!fold2_kernel_ok(kernel))
    return fail(MISPEC_E_UNSUPPORTED, "the second fold needs a kernel of 128 .. 8192 taps, a multiple of 64%s");
  if (precision != MISPEC_PREC_F32 && precision != MISPEC_PREC_BF16X3 && precision != MISPEC_PREC_F16X3)
    return fail(MISPEC_E_INVALID, "bad precision%s");
  if (dst_bytes < basis_fold2_bytes(n_bins, kernel))
    return fail(MISPEC_E_INVALID, "dst too small: size it with mispec_basis_fold2_bytes%s");
  const int kf = fold2_taps(kernel);
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned short *d = static_cast<unsigned short *>(dst);
  float *last_rows = reinterpret_cast<float *>(d + (long long)n_bins * kf * 4);
  if (hipMemsetAsync(stats, 0, 3 * sizeof(float), s) != hipSuccess)
    return fail(MISPEC_E_HIP, "hipMemsetAsync failed%s");
  const int ne = (n_bins + 1) / 2;
  hipLaunchKernelGGL(fold2_basis_kernel, dim3((unsigned)((kf + 255) / 256), (unsigned)n_bins), dim3(256), 0, s,
                     basis_re, basis_im, (long long)basis_row_stride, n_bins, kernel, kf, d, last_rows,
                     2 * (ne - 1), reinterpret_cast<unsigned *>(stats), fold_arith_of(precision));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "basis fold launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

// ---------------------------------------------------------------------------------
// Host path (CPU tensors): the same contraction + epilogue as plain C++ loops over HOST pointers, so
// that the modules run wherever their input lives, like the reference's (stft.py:290-293) -- meant
// for plumbing-sized inputs (BASELINE configs[0]: STFT 512/128 of 1 s), not for throughput.  fp32
// multiply-adds in tap order (the reference's conv1d arithmetic); rows are dealt to a few threads.
// ---------------------------------------------------------------------------------
int mispec_framed_gemm_host_f32(const mispec_framed_gemm_args *a) {
  KParams p;
  int rc = fill_params(a, p);  // (the same argument checks as the device entry)
  if (rc != MISPEC_OK) return rc;
  if (a->fb) return fail(MISPEC_E_UNSUPPORTED, "host path: no fused filterbank (use mispec_filterbank_host_f32)%s");
  if (p.out_fm) return fail(MISPEC_E_UNSUPPORTED, "out_frame_major: served by the FFT path only%s");
  const int E = epilogue_width_host(a->epilogue);
  const long long items = (long long)a->n_clips * a->n_bins;
  host_parallel_for(items, [&](long long it) {
    const int c = (int)(it / a->n_bins), f = (int)(it - (long long)c * a->n_bins);
    const float *x = a->x + (long long)c * a->x_clip_stride;
    const float *wr = a->basis_re + (long long)f * a->basis_row_stride;
    const float *wi = a->basis_im ? a->basis_im + (long long)f * a->basis_row_stride : nullptr;
    int k0 = 0, k1 = a->kernel;
    if (a->row_support) {
      k0 = a->row_support[2 * f];
      k1 = a->row_support[2 * f + 1];
    }
    const float sc = a->row_scale ? a->row_scale[f] : 1.f;
    float *orow = a->out + (long long)c * a->out_clip_stride + (long long)(a->out_row_offset + f) * a->out_row_stride;
    for (int t = 0; t < a->n_frames; ++t) {
      const long long q0 = (long long)t * a->hop - a->pad;
      float re = 0.f, im = 0.f;
      if (q0 + k0 >= 0 && q0 + k1 <= a->n_samples) {
        const float *xs = x + q0;
        for (int k = k0; k < k1; ++k) {
          re = fmaf(wr[k], xs[k], re);
          if (wi) im = fmaf(wi[k], xs[k], im);
        }
      } else {
        for (int k = k0; k < k1; ++k) {
          const float v = host_sample(x, q0 + k, a->n_samples, a->pad_mode);
          re = fmaf(wr[k], v, re);
          if (wi) im = fmaf(wi[k], v, im);
        }
      }
      host_epilogue(a, orow + (long long)t * E, re * sc, a->im_sign * im * sc);
    }
  });
  return MISPEC_OK;
}

int mispec_filterbank_host_f32(const float *fb, int32_t n_filters, int32_t n_freq, const float *spec,
                               int32_t n_clips, int32_t n_frames, float *out) {
  if (!fb || !spec || !out) return fail(MISPEC_E_INVALID, "NULL pointer%s");
  if (n_filters <= 0 || n_freq <= 0 || n_clips <= 0 || n_frames <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  host_parallel_for((long long)n_clips * n_filters, [&](long long it) {
    const int c = (int)(it / n_filters), m = (int)(it - (long long)c * n_filters);
    float *o = out + ((long long)c * n_filters + m) * n_frames;
    for (int t = 0; t < n_frames; ++t) o[t] = 0.f;
    for (int f = 0; f < n_freq; ++f) {
      const float w = fb[(long long)m * n_freq + f];
      if (w == 0.f) continue;
      const float *sp = spec + ((long long)c * n_freq + f) * n_frames;
      for (int t = 0; t < n_frames; ++t) o[t] = fmaf(w, sp[t], o[t]);
    }
  });
  return MISPEC_OK;
}

int mispec_power_to_db_host_f32(const float *spec, int32_t n_clips, int64_t clip_elems, float amin, float ref,
                                float top_db, float *out) {
  if (!spec || !out) return fail(MISPEC_E_INVALID, "NULL pointer%s");
  if (n_clips <= 0 || clip_elems <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!(amin > 0.f)) return fail(MISPEC_E_INVALID, "amin must be positive%s");
  // the arithmetic of power_to_db_kernel (mel.py:263-279), one clip per work item
  const float off = 10.0f * log10f(fmaxf(amin, fabsf(ref)));  // (|ref|: as the device entries and the reference, mel.py:276)
  host_parallel_for(n_clips, [&](long long c) {
    const float *sp = spec + c * clip_elems;
    float *o = out + c * clip_elems;
    float floor_db = -INFINITY;
    if (top_db >= 0.f) {
      float m = amin;  // (the maximum of max(spec, amin))
      for (long long i = 0; i < clip_elems; ++i) m = fmaxf(m, sp[i]);
      floor_db = (10.0f * log10f(m) - off) - top_db;
    }
    for (long long i = 0; i < clip_elems; ++i) o[i] = fmaxf(10.0f * log10f(fmaxf(sp[i], amin)) - off, floor_db);
  });
  return MISPEC_OK;
}

int mispec_istft_host_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames, const float *basis,
                          int32_t n_fft, const float *window, int32_t hop, int32_t start, float *out,
                          int64_t out_clip_stride, int32_t out_len) {
  if (!spec || !basis || !window || !out) return fail(MISPEC_E_INVALID, "NULL pointer%s");
  if (n_clips <= 0 || n_freq <= 0 || n_frames <= 0 || n_fft <= 0 || hop <= 0 || out_len <= 0 || start < 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  if ((long long)start + out_len > (long long)(n_frames - 1) * hop + n_fft)
    return fail(MISPEC_E_INVALID, "output range exceeds the overlap-add signal%s");
  // steps 1 + 2 of the inverse STFT (mispec_istft_frames_f32 + mispec_overlap_add_f32) on host pointers: a
  // clip's frames in a scratch vector, then the gather overlap-add in overlap_add_kernel's order
  const int N = n_fft, F = n_freq, T = n_frames;
  const float inv_n = 1.0f / (float)N;
  host_parallel_for(n_clips, [&](long long c) {
    std::vector<float> frames((size_t)T * N);
    const float *sc = spec + c * (long long)F * T * 2;
    for (int t = 0; t < T; ++t) {
      for (int n = 0; n < N; ++n) {
        const float *b = basis + (long long)n * 2 * F;
        float acc = 0.f;
        for (int k = 0; k < F; ++k) acc = fmaf(b[k], sc[((long long)k * T + t) * 2], acc);
        for (int k = 0; k < F; ++k) acc = fmaf(b[F + k], sc[((long long)k * T + t) * 2 + 1], acc);
        frames[(size_t)t * N + n] = acc;
      }
    }
    float *o = out + c * out_clip_stride;
    for (int i = 0; i < out_len; ++i) {
      const long long pos = (long long)i + start;
      int t_hi = (int)(pos / hop);
      t_hi = t_hi < T - 1 ? t_hi : T - 1;
      const long long t_lo_num = pos - N + 1;
      const int t_lo = t_lo_num <= 0 ? 0 : (int)((t_lo_num + hop - 1) / hop);
      float acc = 0.f, wss = 0.f;
      for (int t = t_lo; t <= t_hi; ++t) {
        const int n = (int)(pos - (long long)t * hop);
        const float w = window[n];
        acc += frames[(size_t)t * N + n] * w * inv_n;
        wss += w * w;
      }
      if (wss > 1e-10f) acc /= wss;
      o[i] = acc;
    }
  });
  return MISPEC_OK;
}

int mispec_fir_decimate_host_f32(const float *x, int64_t x_clip_stride, int32_t n_clips, int32_t n_samples,
                                 const float *taps, int32_t n_taps, int32_t stride, int32_t pad, float *y,
                                 int64_t y_clip_stride, int32_t n_out) {
  if (!x || !taps || !y) return fail(MISPEC_E_INVALID, "NULL pointer%s");
  KParams p;
  int rc = fir_params(p, x, x_clip_stride, n_clips, n_samples, taps, n_taps, stride, pad, y, y_clip_stride, n_out);
  if (rc != MISPEC_OK) return rc;
  const int chunk = 4096;
  const long long per = (n_out + chunk - 1) / chunk;
  host_parallel_for((long long)n_clips * per, [&](long long it) {
    const int c = (int)(it / per);
    const int i0 = (int)(it - (long long)c * per) * chunk, i1 = i0 + chunk < n_out ? i0 + chunk : n_out;
    const float *xc = x + (long long)c * x_clip_stride;
    float *yc = y + (long long)c * y_clip_stride;
    for (int i = i0; i < i1; ++i) {
      const long long q0 = (long long)i * stride - pad;
      float acc = 0.f;
      for (int n = 0; n < n_taps; ++n) {
        const long long q = q0 + n;
        if (q >= 0 && q < n_samples) acc = fmaf(taps[n], xc[q], acc);
      }
      yc[i] = acc;
    }
  });
  return MISPEC_OK;
}

int mispec_framed_gemm_group_f32(const mispec_framed_gemm_args *args, int32_t n, void *stream) {
  if (!args || n <= 0) return fail(MISPEC_E_INVALID, "empty group%s");
  if (n > GROUP_MAX) return fail(MISPEC_E_UNSUPPORTED, "more than 8 problems in a group%s");
  KParams ps[GROUP_MAX];
  hipStream_t s = static_cast<hipStream_t>(stream);
  int tile = -1;
  for (int i = 0; i < n; ++i) {
    int rc = fill_params(&args[i], ps[i]);
    if (rc != MISPEC_OK) return rc;
    if (ps[i].fb) return fail(MISPEC_E_UNSUPPORTED, "fused filterbank: not in grouped launches%s");
    if (ps[i].out_fm) return fail(MISPEC_E_UNSUPPORTED, "out_frame_major: served by the FFT path only%s");
    const int rows = ps[i].n_bins * (ps[i].a_im ? 2 : 1);
    const int t = args[i].tile != MISPEC_TILE_AUTO ? args[i].tile
                                                    : auto_tile(rows, false);  // (grouped launches are unmasked)
    if (i == 0) tile = t;
    if (t != tile || (t != MISPEC_TILE_32x256 && t != MISPEC_TILE_64x256))
      return fail(MISPEC_E_UNSUPPORTED,
                  "grouped launch needs one narrow tile shape (<= 64 basis rows) for every problem%s");
  }
  for (int i = 0; i < n; ++i) {
    int rc = setup_edges(ps[i], args[i].workspace, args[i].workspace_bytes, s);
    if (rc != MISPEC_OK) return rc;
  }
  if (tile == MISPEC_TILE_32x256) return launch_group_cfg<1, 4, 1, 2, false, true>(ps, n, s);
  return launch_group_cfg<1, 4, 2, 2, false, true>(ps, n, s);
}

int mispec_framed_gemm_f32_ref(const mispec_framed_gemm_args *args, void *stream) {
  KParams p;
  int rc = fill_params(args, p);
  if (rc != MISPEC_OK) return rc;
  if (p.fb) return fail(MISPEC_E_UNSUPPORTED, "fused filterbank: not in the reference kernel%s");
  if (p.out_fm) return fail(MISPEC_E_UNSUPPORTED, "out_frame_major: served by the FFT path only%s");
  const long long total = p.n_cols * p.n_bins;
  const long long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  hipLaunchKernelGGL(framed_gemm_ref_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_contract_planar_f32(const mispec_planar_args *a, void *stream) {
  if (!a) return fail(MISPEC_E_INVALID, "args is NULL%s");
  if (a->struct_size != sizeof(mispec_planar_args))
    return fail(MISPEC_E_INVALID, "struct_size mismatch (ABI skew)%s");
  if (!a->a || !a->x || !a->out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (a->m <= 0 || a->k <= 0 || a->n_clips <= 0 || a->n_cols <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  if ((long long)a->n_clips * a->n_cols > 0x7fffffffLL)
    return fail(MISPEC_E_UNSUPPORTED, "column count overflows int32%s");
  KParams p;
  memset(&p, 0, sizeof(p));
  p.x = a->x;
  p.x_clip_stride = a->x_clip_stride;
  p.x_k_stride = a->x_k_stride;
  p.x_col_stride = a->x_col_stride;
  p.k_split = a->k_split;
  p.k_split_off = a->k_split_off;
  p.k_offsets = reinterpret_cast<const long long *>(a->k_offsets);
  p.n_clips = a->n_clips;
  p.n_samples = a->n_cols;
  p.hop = 1;
  p.n_frames = a->n_cols;
  p.n_cols = (long long)a->n_clips * a->n_cols;
  p.a_re = a->a;
  p.a_im = nullptr;
  p.a_row_stride = a->a_row_stride;
  p.n_bins = a->m;
  p.K = a->k;
  p.epilogue = MISPEC_EPI_REAL;
  p.im_sign = 1.f;
  p.out = a->out;
  p.out_clip_stride = a->out_clip_stride;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->rows_inner) {
    if ((long long)a->n_cols * a->out_col_stride > 0x7fffffffLL)
      return fail(MISPEC_E_UNSUPPORTED, "one clip of the output overflows int32%s");
    p.out_frame_stride = (int)a->out_col_stride;
    p.out_len = (int)((long long)(a->n_cols - 1) * a->out_col_stride + a->m);
    if (a->m <= 32) return launch_cfg<1, 4, 1, 2, BMODE_PLANAR_T, AMODE_ROWS, false>(p, s);
    if (a->m <= 64) return launch_cfg<1, 4, 2, 2, BMODE_PLANAR_T, AMODE_ROWS, false>(p, s);
    return launch_cfg<2, 2, 2, 2, BMODE_PLANAR_T, AMODE_ROWS, false>(p, s);
  }
  p.out_row_stride = a->out_row_stride;
  if (a->m <= 32) return launch_cfg<1, 4, 1, 2, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
  if (a->m <= 64) return launch_cfg<1, 4, 2, 2, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
  return launch_cfg<2, 2, 2, 2, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
}

int mispec_pad_signal_f32(const float *x, int64_t x_clip_stride, int32_t n_clips, int32_t n_samples,
                          int32_t pad, int32_t pad_mode, float *out, void *stream) {
  if (!x || !out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_samples <= 0 || pad < 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (pad_mode < MISPEC_PAD_NONE || pad_mode > MISPEC_PAD_REFLECT || (pad_mode == MISPEC_PAD_NONE && pad))
    return fail(MISPEC_E_INVALID, "bad pad_mode%s");
  if (pad_mode == MISPEC_PAD_REFLECT && pad >= n_samples)
    return fail(MISPEC_E_INVALID, "reflect padding needs pad < n_samples%s");
  KParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.x_clip_stride = x_clip_stride;
  p.n_samples = n_samples;
  p.pad = pad;
  p.pad_mode = pad_mode;
  const long long lp = (long long)n_samples + 2LL * pad;
  hipLaunchKernelGGL(pad_signal_kernel, dim3((unsigned)((lp + 255) / 256), (unsigned)n_clips), dim3(256),
                     0, static_cast<hipStream_t>(stream), p, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_unpad_adjoint_f32(const float *dxp, int32_t n_clips, int32_t n_samples, int32_t pad,
                             int32_t pad_mode, float *dx, int64_t dx_clip_stride, void *stream) {
  if (!dxp || !dx) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_samples <= 0 || pad < 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  hipLaunchKernelGGL(unpad_adjoint_kernel, dim3((unsigned)((n_samples + 255) / 256), (unsigned)n_clips),
                     dim3(256), 0, static_cast<hipStream_t>(stream), dxp, n_samples, pad, pad_mode, dx,
                     (long long)dx_clip_stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_frame_offsets_i64(int64_t *k_offsets, int32_t n_clips, int32_t n_frames, int64_t clip_stride,
                             int32_t hop, void *stream) {
  if (!k_offsets) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_frames <= 0 || hop <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  const long long n = (long long)n_clips * n_frames;
  hipLaunchKernelGGL(frame_offsets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), reinterpret_cast<long long *>(k_offsets), n_clips,
                     n_frames, (long long)clip_stride, hop);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_frames_transpose_f32(const float *xpad, int64_t clip_stride, int32_t n_clips,
                                int32_t n_frames, int32_t hop, int32_t kernel, float *xt, void *stream) {
  if (!xpad || !xt) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_frames <= 0 || hop <= 0 || kernel <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  const long long cols = (long long)n_clips * n_frames;
  hipLaunchKernelGGL(frames_transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((kernel + 31) / 32)),
                     dim3(256), 0, static_cast<hipStream_t>(stream), xpad, (long long)clip_stride, n_frames,
                     hop, kernel, cols, xt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_framed_epilogue_fwd_f32(const float *z, int32_t n_clips, int32_t n_bins, int32_t n_frames,
                                   int32_t epilogue, float eps, float power, float *out, void *stream) {
  if (!z || !out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_bins <= 0 || n_frames <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (epilogue < MISPEC_EPI_COMPLEX || epilogue > MISPEC_EPI_PHASE_COSSIN)
    return fail(MISPEC_E_INVALID, "bad epilogue%s");
  const long long total = (long long)n_clips * n_bins * n_frames;
  if ((total + 255) / 256 > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  EpiParams ep;
  ep.epilogue = epilogue;
  ep.eps = eps;
  ep.power = power;
  hipLaunchKernelGGL(framed_epilogue_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), z, total, ep, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_framed_epilogue_bwd_f32(const float *grad_out, const float *z, int32_t n_clips, int32_t n_bins,
                                   int32_t n_frames, int32_t epilogue, float eps, float power,
                                   float im_sign, const float *row_scale, float *g, float *gt,
                                   void *stream) {
  if (!grad_out || !z || (!g && !gt)) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_bins <= 0 || n_frames <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (epilogue < MISPEC_EPI_COMPLEX || epilogue > MISPEC_EPI_PHASE_COSSIN)
    return fail(MISPEC_E_INVALID, "bad epilogue%s");
  const long long total = (long long)n_clips * n_bins * n_frames;
  if ((total + 255) / 256 > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  hipLaunchKernelGGL(framed_epilogue_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), grad_out, z, n_clips, n_bins, n_frames, epilogue,
                     eps, power, im_sign, row_scale, g, gt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_fir_decimate_bwd_f32(const float *dy, int64_t dy_clip_stride, int32_t n_clips, int32_t n_out,
                                const float *taps, int32_t n_taps, int32_t stride, int32_t pad,
                                float *dx, int64_t dx_clip_stride, int32_t n_samples, void *stream) {
  if (!dy || !taps || !dx) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_out <= 0 || n_taps <= 0 || stride <= 0 || pad < 0 || n_samples <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  hipLaunchKernelGGL(fir_decimate_bwd_kernel, dim3((unsigned)((n_samples + 255) / 256), (unsigned)n_clips),
                     dim3(256), 0, static_cast<hipStream_t>(stream), dy, (long long)dy_clip_stride, n_out,
                     taps, n_taps, stride, pad, n_samples, dx, (long long)dx_clip_stride);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_istft_frames_fft_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames,
                                int32_t n_fft, float *frames, void *stream) {
  if (!spec || !frames) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_frames <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if ((n_fft != 512 && n_fft != 1024 && n_fft != 2048) || n_freq != n_fft / 2 + 1)
    return fail(MISPEC_E_UNSUPPORTED, "inverse FFT: one-sided spectrum of n_fft = 512, 1024 or 2048%s");
  const int tiles_per_clip = (n_frames + FFT_WAVES - 1) / FFT_WAVES;
  const long long n_tiles = (long long)n_clips * tiles_per_clip;
  if (n_tiles > 0x3fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  long long grid = n_tiles < device_cus() ? n_tiles : device_cus();
  grid = (grid + 7) / 8 * 8;
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = MISPEC_OK;
#define MISPEC_LAUNCH_IFFT(MM)                                                                        \
  {                                                                                                   \
    auto kern = istft_fft_kernel<MM>;                                                                 \
    constexpr size_t smem = istft_fft_smem<MM>();                                                     \
    static std::atomic<unsigned long long> configured{0};                                             \
    rc = configure_lds(kern, smem, configured);                                                       \
    if (rc == MISPEC_OK)                                                                              \
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(FFT_WAVES * 64), smem, s, spec, n_clips,    \
                         n_frames, frames, tiles_per_clip);                                           \
  }
  if (n_fft == 2048) MISPEC_LAUNCH_IFFT(1024)
  else if (n_fft == 1024) MISPEC_LAUNCH_IFFT(512)
  else MISPEC_LAUNCH_IFFT(256)
#undef MISPEC_LAUNCH_IFFT
  if (rc != MISPEC_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_istft_fft_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames, int32_t n_fft,
                         const float *window, int32_t hop, int32_t start, float *out, int64_t out_clip_stride,
                         int32_t out_len, void *stream) {
  if (!spec || !window || !out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_frames <= 0 || hop <= 0 || out_len <= 0 || start < 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if ((n_fft != 512 && n_fft != 1024 && n_fft != 2048) || n_freq != n_fft / 2 + 1)
    return fail(MISPEC_E_UNSUPPORTED, "inverse FFT: one-sided spectrum of n_fft = 512, 1024 or 2048%s");
  if (hop % 64 || hop > n_fft || n_fft % hop)
    return fail(MISPEC_E_UNSUPPORTED, "fused inverse STFT: hop a multiple of 64 that divides n_fft%s");
  if ((long long)start + out_len > (long long)(n_frames - 1) * hop + n_fft)
    return fail(MISPEC_E_INVALID, "output range exceeds the overlap-add signal%s");
  // tiles of 8 frames whose final samples [8 hop tile, 8 hop (tile + 1)) reach the end of the output; runs of
  // consecutive tiles so that every CU has one (a run re-walks the tiles that reach into its first one)
  const long long tile_span = 8LL * hop;
  const int n_tiles_clip = (int)(((long long)start + out_len + tile_span - 1) / tile_span);
  const int n_warm = (int)((n_fft - hop + tile_span - 1) / tile_span);
  int runs = (device_cus() + n_clips - 1) / n_clips;
  const int most = n_tiles_clip / (4 * (n_warm > 0 ? n_warm : 1)) > 1 ? n_tiles_clip / (4 * (n_warm > 0 ? n_warm : 1)) : 1;
  runs = runs > most ? most : runs;  // (the re-walked tiles stay below a quarter of a run)
  runs = runs < 1 ? 1 : runs;
  const int per = (n_tiles_clip + runs - 1) / runs;
  runs = (n_tiles_clip + per - 1) / per;
  if ((long long)runs * n_clips > 0x3fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = MISPEC_OK;
#define MISPEC_LAUNCH_IOLA(MM)                                                                               \
  {                                                                                                          \
    auto kern = istft_ola_fft_kernel<MM>;                                                                    \
    const size_t smem = istft_ola_smem<MM>(hop);                                                             \
    static std::atomic<unsigned long long> configured{0};                                                    \
    rc = configure_lds(kern, 160 * 1024, configured);                                                        \
    if (rc == MISPEC_OK && smem > 160 * 1024) rc = fail(MISPEC_E_UNSUPPORTED, "fused inverse STFT: LDS%s"); \
    if (rc == MISPEC_OK)                                                                                     \
      hipLaunchKernelGGL(kern, dim3((unsigned)(runs * n_clips)), dim3(FFT_WAVES * 64), smem, s, spec, n_clips, n_frames, \
                         window, hop, start, out_len, out, (long long)out_clip_stride, runs, per, n_tiles_clip); \
  }
  if (n_fft == 2048) MISPEC_LAUNCH_IOLA(1024)
  else if (n_fft == 1024) MISPEC_LAUNCH_IOLA(512)
  else MISPEC_LAUNCH_IOLA(256)
#undef MISPEC_LAUNCH_IOLA
  if (rc != MISPEC_OK) return rc;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_istft_frames_f32(const float *spec, int32_t n_clips, int32_t n_freq, int32_t n_frames,
                            const float *basis, int32_t n_fft, float *frames, void *stream) {
  if (!spec || !basis || !frames) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_freq <= 0 || n_frames <= 0 || n_fft <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  KParams p;
  memset(&p, 0, sizeof(p));
  // K axis = [re of bins 0..F-1 | im of bins 0..F-1]; element (k, t, c) of a clip of the
  // interleaved (F, T, 2) spectrogram sits at k*2T + 2t + c
  p.x = spec;
  p.x_clip_stride = (long long)n_freq * n_frames * 2;
  p.x_k_stride = 2LL * n_frames;
  p.x_col_stride = 2;
  p.k_split = n_freq;
  p.k_split_off = 1;
  p.n_clips = n_clips;
  p.n_samples = n_frames;
  p.hop = 1;
  p.n_frames = n_frames;
  p.n_cols = (long long)n_clips * n_frames;
  p.a_re = basis;
  p.a_im = nullptr;
  p.a_row_stride = 2LL * n_freq;
  p.n_bins = n_fft;
  p.K = 2 * n_freq;
  p.epilogue = MISPEC_EPI_REAL;
  p.im_sign = 1.f;
  p.out = frames;
  p.out_clip_stride = (long long)n_frames * n_fft;
  p.out_frame_stride = n_fft;
  p.out_len = (int)((long long)n_frames * n_fft > 0x7fffffffLL ? 0x7fffffff : n_frames * n_fft);
  if ((long long)n_frames * n_fft > 0x7fffffffLL)
    return fail(MISPEC_E_UNSUPPORTED, "frames of one clip overflow int32%s");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_fft <= 32) return launch_cfg<1, 4, 1, 2, BMODE_PLANAR_T, AMODE_ROWS, false>(p, s);
  if (n_fft <= 64) return launch_cfg<1, 4, 2, 2, BMODE_PLANAR_T, AMODE_ROWS, false>(p, s);
  return launch_cfg<2, 2, 2, 2, BMODE_PLANAR_T, AMODE_ROWS, false>(p, s);
}

int mispec_overlap_add_f32(const float *frames, int32_t n_clips, int32_t n_frames, int32_t n_fft,
                           const float *window, int32_t hop, int32_t start, float *out,
                           int64_t out_clip_stride, int32_t out_len, void *stream) {
  // start < 0 (with window == NULL) flags tap-major frames (n_fft, n_clips*n_frames); the
  // overlap-add then starts at sample -(start + 1)
  const bool transposed = start < 0;
  if (transposed) {
    if (window) return fail(MISPEC_E_INVALID, "tap-major frames need window == NULL%s");
    start = -(start + 1);
  }
  if (!frames || !out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_frames <= 0 || n_fft <= 0 || hop <= 0 || out_len <= 0 || start < 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  if ((long long)start + out_len > (long long)(n_frames - 1) * hop + n_fft)
    return fail(MISPEC_E_INVALID, "output range exceeds the overlap-add signal%s");
  hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)((out_len + 255) / 256), (unsigned)n_clips),
                     dim3(256), 0, static_cast<hipStream_t>(stream), frames, window, n_fft, hop,
                     n_frames, start < 0 ? 0 : start, out_len, out, (long long)out_clip_stride,
                     transposed ? (long long)n_clips * n_frames : 0LL);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_istft_grad_signal_f32(const float *grad_out, int64_t grad_clip_stride, int32_t n_clips,
                                 int32_t n_frames, int32_t n_fft, const float *window, int32_t hop,
                                 int32_t start, int32_t out_len, float *u, void *stream) {
  if (!grad_out || !window || !u) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_frames <= 0 || n_fft <= 0 || hop <= 0 || out_len <= 0 || start < 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  const long long full = (long long)(n_frames - 1) * hop + n_fft;
  if ((long long)start + out_len > full)
    return fail(MISPEC_E_INVALID, "output range exceeds the overlap-add signal%s");
  hipLaunchKernelGGL(istft_grad_signal_kernel, dim3((unsigned)((full + 255) / 256), (unsigned)n_clips),
                     dim3(256), 0, static_cast<hipStream_t>(stream), grad_out, (long long)grad_clip_stride,
                     window, n_fft, hop, n_frames, start, out_len, u, full);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_power_to_db_f32(const float *spec, int32_t n_clips, int64_t clip_elems, float amin,
                           float ref, float top_db, float *out, void *workspace,
                           int64_t workspace_bytes, void *stream) {
  if (!spec || !out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || clip_elems <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!(amin > 0.f)) return fail(MISPEC_E_INVALID, "amin must be strictly positive%s");
  if (!workspace || workspace_bytes < (int64_t)n_clips * 4)
    return fail(MISPEC_E_INVALID, "workspace too small: n_clips * 4 bytes%s");
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned *wmax = static_cast<unsigned *>(workspace);
  long long bx = (clip_elems + 256 * 8 - 1) / (256 * 8);  // ~8 elements per thread
  bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
  const dim3 grid((unsigned)bx, (unsigned)n_clips);
  if (top_db >= 0.f) {
    hipLaunchKernelGGL(clear_u32_kernel, dim3((n_clips + 255) / 256), dim3(256), 0, s, wmax, n_clips);
    hipLaunchKernelGGL(clip_max_kernel, grid, dim3(256), 0, s, spec, (long long)clip_elems, amin, wmax);
  }
  hipLaunchKernelGGL(power_to_db_kernel, grid, dim3(256), 0, s, spec, (long long)clip_elems, amin,
                     fabsf(ref), top_db, wmax, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_mfcc_tail_f32(const float *mel, int32_t n_clips, int32_t n_mels, int32_t n_frames, float amin, float ref,
                         float top_db, const float *dct, int32_t n_mfcc, float *out, void *stream) {
  if (!mel || !dct || !out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_mels <= 0 || n_frames <= 0 || n_mfcc <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!(amin > 0.f)) return fail(MISPEC_E_INVALID, "amin must be strictly positive%s");
  if (mel == out) return fail(MISPEC_E_INVALID, "out must not alias mel%s");
  const size_t smem = (size_t)n_mels * MFCC_TT * sizeof(float);
  if (n_mfcc > n_mels || smem > 64 * 1024)
    return fail(MISPEC_E_UNSUPPORTED, "MFCC tail: at most 256 mel bands (a 64-frame tile in 64 KB of LDS), n_mfcc <= n_mels%s");
  // workgroups: one per clip and share of its 64-frame tiles, about four per CU
  const int n_tiles = (n_frames + MFCC_TT - 1) / MFCC_TT;
  long long shares = (4LL * device_cus() + n_clips - 1) / n_clips;
  shares = shares < 1 ? 1 : (shares > n_tiles ? n_tiles : shares);
  const int tiles_per_wg = (int)((n_tiles + shares - 1) / shares);
  const unsigned grid_y = (unsigned)((n_tiles + tiles_per_wg - 1) / tiles_per_wg);
  const int kk = n_mfcc >= 25 ? 4 : (n_mfcc + 7) / 8;  // coefficients per wave and trip: 8 waves x kk cover n_mfcc when it is small
  hipStream_t st = static_cast<hipStream_t>(stream);
#define MISPEC_MFCC_TAIL(KK)                                                                                          \
  {                                                                                                                   \
    static std::atomic<unsigned long long> configured{0};                                                             \
    int rc = configure_lds(mfcc_tail_kernel<KK>, 64 * 1024, configured);                                              \
    if (rc != MISPEC_OK) return rc;                                                                                   \
    hipLaunchKernelGGL(mfcc_tail_kernel<KK>, dim3((unsigned)n_clips, grid_y), dim3(512), smem, st, mel, n_mels,       \
                       n_frames, amin, fabsf(ref), top_db, dct, n_mfcc, out, tiles_per_wg);                           \
  }
  if (kk <= 1) MISPEC_MFCC_TAIL(1)
  else if (kk == 2) MISPEC_MFCC_TAIL(2)
  else if (kk == 3) MISPEC_MFCC_TAIL(3)
  else MISPEC_MFCC_TAIL(4)
#undef MISPEC_MFCC_TAIL
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_power_to_db_bwd_f32(const float *spec, const float *grad_out, int32_t n_clips,
                               int64_t clip_elems, float amin, float top_db, float *grad_spec,
                               void *workspace, int64_t workspace_bytes, void *stream) {
  if (!spec || !grad_out || !grad_spec) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || clip_elems <= 0) return fail(MISPEC_E_INVALID, "non-positive size%s");
  if (!(amin > 0.f)) return fail(MISPEC_E_INVALID, "amin must be strictly positive%s");
  if (!workspace || workspace_bytes < (int64_t)n_clips * 8)
    return fail(MISPEC_E_INVALID, "workspace too small: n_clips * 8 bytes%s");
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned *wmax = static_cast<unsigned *>(workspace);
  float *wsum = reinterpret_cast<float *>(wmax + n_clips);
  long long bx = (clip_elems + 256 * 8 - 1) / (256 * 8);
  bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
  const dim3 grid((unsigned)bx, (unsigned)n_clips);
  if (top_db >= 0.f) {
    hipLaunchKernelGGL(clear_u32_kernel, dim3((2 * n_clips + 255) / 256), dim3(256), 0, s, wmax, 2 * n_clips);
    hipLaunchKernelGGL(clip_max_kernel, grid, dim3(256), 0, s, spec, (long long)clip_elems, amin, wmax);
    hipLaunchKernelGGL(power_to_db_floor_sum_kernel, grid, dim3(256), 0, s, spec, grad_out,
                       (long long)clip_elems, amin, top_db, wmax, wsum);
  }
  hipLaunchKernelGGL(power_to_db_bwd_kernel, grid, dim3(256), 0, s, spec, grad_out, (long long)clip_elems,
                     amin, top_db, wmax, wsum, grad_spec);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int mispec_filterbank_f32(const float *fb, int32_t n_filters, int32_t n_freq, const float *spec,
                          int32_t n_clips, int32_t n_frames, float *out, void *stream) {
  if (!fb || !spec || !out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_filters <= 0 || n_freq <= 0 || n_clips <= 0 || n_frames <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  KParams p;
  memset(&p, 0, sizeof(p));
  p.x = spec;
  p.x_clip_stride = (long long)n_freq * n_frames;
  p.x_k_stride = n_frames;
  p.n_clips = n_clips;
  p.n_samples = n_frames;
  p.hop = 1;
  p.n_frames = n_frames;
  p.n_cols = (long long)n_clips * n_frames;
  p.a_re = fb;
  p.a_im = nullptr;
  p.a_row_stride = n_freq;
  p.n_bins = n_filters;
  p.K = n_freq;
  p.epilogue = MISPEC_EPI_REAL;
  p.im_sign = 1.f;
  p.out = out;
  p.out_clip_stride = (long long)n_filters * n_frames;
  p.out_row_stride = n_frames;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // (64 x 64 tiles, one accumulator tile per wave; 64 x 256 with 2 x 2 tiles per wave -- mispec_contract_planar_f32's
  // choice -- measured 0.44 against 0.265 ms for the Gammatonegram of cfg2's batch, 64 x 128 with 1 x 2 tiles 0.32)
  if (n_filters <= 32) return launch_cfg<1, 4, 1, 1, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
  if (n_filters <= 64) return launch_cfg<2, 2, 1, 1, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
  return launch_cfg<2, 2, 2, 2, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
}

int mispec_octave_pyramid_f32(const mispec_octave_args *a, void *stream) {
  if (!a) return fail(MISPEC_E_INVALID, "args is NULL%s");
  if (a->struct_size != sizeof(mispec_octave_args)) return fail(MISPEC_E_INVALID, "struct_size mismatch (ABI skew)%s");
  if (!a->x || !a->out) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (a->n_levels < 1 || a->n_levels > PYR_LEVELS) return fail(MISPEC_E_INVALID, "n_levels must be 1..3%s");
  if (a->n_clips <= 0 || a->n_samples <= 0 || a->hop <= 0 || a->n_frames <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  const int D = a->n_levels;
  if (D > 1) {
    if (!a->taps) return fail(MISPEC_E_INVALID, "taps is required when n_levels > 1%s");
    // window element m of output r is tap m - 2 r - shift, shift = 128 - (n_taps - 1) / 2 (the halo
    // of level l is 2 * halo(l + 1) + 128 samples); the band must fit the 320 columns
    if (a->n_taps <= 0 || 128 - (a->n_taps - 1) / 2 < 0 || a->n_taps + 62 + 128 - (a->n_taps - 1) / 2 > 16 * PYR_KSTEPS)
      return fail(MISPEC_E_UNSUPPORTED, "anti-alias filter too long for the fused octave kernel%s");
  }
  if (a->epilogue < MISPEC_EPI_COMPLEX || a->epilogue > MISPEC_EPI_PHASE_COSSIN)
    return fail(MISPEC_E_INVALID, "bad epilogue%s");
  PyrParams p;
  memset(&p, 0, sizeof(p));
  p.x = a->x;
  p.x_clip_stride = a->x_clip_stride;
  p.n_levels = D;
  p.n_frames = a->n_frames;
  p.n_clips = a->n_clips;
#ifdef MISPEC_ABLATE
  p.stamps = reinterpret_cast<unsigned long long *>(a->reserved);
#else
  if (a->reserved != 0) return fail(MISPEC_E_INVALID, "reserved must be 0%s");
#endif
  p.taps = a->taps;
  p.n_taps = a->n_taps;
  p.dec_pad = (a->n_taps - 1) / 2;
  p.x_last = a->x_last;
  p.x_last_stride = a->x_last_clip_stride;
  p.out = a->out;
  p.out_clip_stride = a->out_clip_stride;
  p.out_row_stride = a->out_row_stride;
  p.epilogue = a->epilogue;
  p.im_sign = a->im_sign;
  p.eps = a->eps;
  if (a->precision != MISPEC_PREC_BF16X3 && a->precision != MISPEC_PREC_F16X3)
    return fail(MISPEC_E_INVALID, "fused octave kernel: precision must be MISPEC_PREC_BF16X3 or MISPEC_PREC_F16X3%s");
  const bool f16 = a->precision == MISPEC_PREC_F16X3;
  if (f16) {
    if (!a->absmax_in) return fail(MISPEC_E_INVALID, "MISPEC_PREC_F16X3 needs absmax_in (n_clips * 128 bytes)%s");
    if (a->fir_headroom_bits < 0 || a->fir_headroom_bits > 7)
      return fail(MISPEC_E_UNSUPPORTED, "fused octave kernel: the anti-alias filter's gain leaves no fp16 headroom%s");
    p.absmax_in = static_cast<const unsigned *>(a->absmax_in);
    p.absmax_out = static_cast<unsigned *>(a->absmax_out);
    p.top = 15 - a->fir_headroom_bits;
  }
  long long L = a->n_samples;
  int need[PYR_LEVELS];
  for (int l = 0; l < D; ++l) {
    const mispec_octave_level &v = a->level[l];
    PyrLevel &o = p.lv[l];
    if (l > 0) L = (L + 2LL * p.dec_pad - a->n_taps) / 2 + 1;
    if (L <= 0) return fail(MISPEC_E_INVALID, "signal too short for this many levels%s");
    if ((a->hop % (1 << l)) || ((a->hop >> l) % 4))
      return fail(MISPEC_E_UNSUPPORTED, "fused octave kernel: hop >> level must be a multiple of 4%s");
    o.L = (int)L;
    o.hop = a->hop >> l;
    // the last frame must lie inside the (virtually padded) level
    if (v.bank_split) {
      if (v.n_bins <= 0 || v.n_bins > PYR_MAX_BINS || v.kernel < 16 || v.kernel % 16 || v.kernel > 2048)
        return fail(MISPEC_E_UNSUPPORTED, "fused octave kernel: <= 16 bins, kernel a multiple of 16%s");
      if (v.bank_split_bytes < (f16 ? mispec_basis_split16_bytes(v.n_bins, v.kernel)
                                    : basis_plane_bytes(v.n_bins, v.kernel, true)))
        return fail(MISPEC_E_INVALID, "bank_split too small%s");
      if (v.pad_mode != MISPEC_PAD_ZERO && v.pad_mode != MISPEC_PAD_REFLECT)
        return fail(MISPEC_E_INVALID, "bad pad_mode%s");
      if (v.pad_mode == MISPEC_PAD_REFLECT && v.kernel / 2 >= L)
        return fail(MISPEC_E_INVALID, "reflect padding needs kernel/2 < level length%s");
      if ((long long)(a->n_frames - 1) * o.hop > L)
        return fail(MISPEC_E_INVALID, "n_frames overruns the padded signal%s");
      o.K = v.kernel;
      o.Ks = round_up_kc(v.kernel);
      o.n_rows = v.n_bins;
      o.out_row0 = v.out_row_offset;
      o.reflect = v.pad_mode == MISPEC_PAD_REFLECT;
      o.bank = static_cast<const unsigned short *>(v.bank_split);
      o.bank_plane = (long long)v.n_bins * o.Ks;
      o.row_scale = v.row_scale;
      o.row_unscale = f16 ? reinterpret_cast<const float *>(static_cast<const char *>(v.bank_split) +
                                                            basis_plane_bytes(v.n_bins, v.kernel, true))
                          : nullptr;
      need[l] = (int)round_up_ll(v.kernel / 2 + 64, 64);
    } else {
      need[l] = 64;
    }
  }
  // halos: halo(l) = 2 halo(l+1) + 128 exactly (the FIR plan's alignment), every level >= its need
  int halo[PYR_LEVELS];
  for (int h = 64;; h += 64) {
    halo[D - 1] = h;
    bool ok = h >= need[D - 1];
    for (int l = D - 2; l >= 0; --l) {
      halo[l] = 2 * halo[l + 1] + 128;
      ok = ok && halo[l] >= need[l];
    }
    if (ok) break;
    if (h > 8192) return fail(MISPEC_E_UNSUPPORTED, "fused octave kernel: kernels too wide%s");
  }
  // frames per workgroup: ~8192 samples of level 0, within 80 KB of LDS (two workgroups per CU)
  int nf = (8192 / p.lv[0].hop + 15) / 16 * 16;
  {
    // ... but at least ~4 work items per CU: the deep launches of a chain (small hops) would
    // otherwise be a few hundred long items on 512 workgroup slots
    const long long want = 4LL * device_cus();
    const long long cap = ((long long)a->n_frames * a->n_clips / want + 15) / 16 * 16;
    if (cap < nf) nf = (int)cap;
  }
  nf = nf < 16 ? 16 : nf;
  size_t smem = 0;
  for (;; nf -= 16) {
    if (nf < 16) return fail(MISPEC_E_UNSUPPORTED, "fused octave kernel: span does not fit in LDS%s");
    smem = 0;
    for (int l = 0; l < D; ++l) {
      PyrLevel &o = p.lv[l];
      o.halo = halo[l];
      const long long n = (long long)nf * o.hop + 2LL * halo[l];
      o.rows = (int)(n / 64);
      o.lds_off = (int)smem;
      smem += (size_t)o.rows * PYR_ROW * 2;
    }
    p.tab_off = (int)smem;
    if (D > 1) smem += PYR_TAB_BYTES;
    p.scale_off = (int)smem;
    smem += 64;  // (per-item scaling: the waves' maxima)
    if (smem <= 80 * 1024) break;
  }
  p.nf = nf;
  p.n_chunks = (a->n_frames + nf - 1) / nf;
  // kernel rows resident in registers: 6 steps (192 taps) when every level fits, else 8
  int max_steps = 0;
  for (int l = 0; l < D; ++l) max_steps = p.lv[l].Ks / 32 > max_steps ? p.lv[l].Ks / 32 : max_steps;
  const bool six = max_steps <= 6;
  auto kern = f16 ? (six ? octave_pyramid_kernel<6, true> : octave_pyramid_kernel<8, true>)
                  : (six ? octave_pyramid_kernel<6, false> : octave_pyramid_kernel<8, false>);
  static std::atomic<unsigned long long> configured4[4] = {{0}, {0}, {0}, {0}};
  int rc = configure_lds(kern, 80 * 1024, configured4[(f16 ? 2 : 0) + (six ? 0 : 1)]);
  if (rc != MISPEC_OK) return rc;
  // the chain's first launch: the power of two per work item, found inside the kernel, when the span of x_0 fits
  // one batch of registers; else the clips' largest |sample| first (into zeroed words)
  p.item_scale = f16 && six && !a->absmax_in_ready && p.lv[0].rows * 64 <= PYR_NBI * 1024 ? 1 : 0;
  if (f16 && !a->absmax_in_ready && !p.item_scale) {
    hipLaunchKernelGGL(clip_absmax_kernel,
                       dim3((unsigned)((a->n_samples + ABSMAX_CHUNK - 1) / ABSMAX_CHUNK), (unsigned)a->n_clips),
                       dim3(256), 0, static_cast<hipStream_t>(stream), a->x, (long long)a->x_clip_stride,
                       a->n_samples, static_cast<unsigned *>(a->absmax_in));
  }
  // persistent workgroups, two per CU
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  const long long items = (long long)p.n_chunks * a->n_clips;
  const unsigned gx = (unsigned)(items < 2LL * cus ? items : 2LL * cus);
  hipLaunchKernelGGL(kern, dim3(gx), dim3(256), smem, static_cast<hipStream_t>(stream), p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

int64_t mispec_fir_decimate_workspace_bytes(int32_t n_clips, int32_t n_samples, int32_t n_taps,
                                            int32_t stride, int32_t pad, int32_t n_out) {
  KParams p;
  int rc = fir_params(p, nullptr, 0, n_clips, n_samples, nullptr, n_taps, stride, pad, nullptr, 0,
                      n_out);
  if (rc != MISPEC_OK) return rc;
  const EdgePlan e = plan_edges(p.n_samples, p.K, p.hop, p.pad, p.n_frames);
  return e.stride * p.n_clips * (int64_t)sizeof(float);
}

int mispec_fir_decimate_f32(const float *x, int64_t x_clip_stride, int32_t n_clips,
                            int32_t n_samples, const float *taps, int32_t n_taps, int32_t stride,
                            int32_t pad, float *y, int64_t y_clip_stride, int32_t n_out,
                            void *workspace, int64_t workspace_bytes, void *stream) {
  if (!x || !taps || !y) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  KParams p;
  int rc = fir_params(p, x, x_clip_stride, n_clips, n_samples, taps, n_taps, stride, pad, y,
                      y_clip_stride, n_out);
  if (rc != MISPEC_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (stride == 2 && n_taps + 62 <= 8 * FIR_KG && !MISPEC_DBG(p, 0x1000)) {
    // dedicated kernel: span staged once in LDS, Toeplitz taps re-read from a tiny LDS table
    constexpr int NRW = FIR_NRW;
    constexpr int OUT_WG = 4096 * NRW;
    constexpr size_t smem = sizeof(float) * ((2 * OUT_WG + 320) / 64 * FIR_ROW + 62 + 320 + 2);
    auto kern = fir_decimate2_kernel<NRW, FIR_AREG>;
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(MISPEC_E_HIP, "hipGetDevice failed%s");
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(configured.load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (e != hipSuccess)
        return fail(MISPEC_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
      configured.fetch_or(bit, std::memory_order_release);
    }
    const unsigned gx = (unsigned)((n_out + OUT_WG - 1) / OUT_WG);
    hipLaunchKernelGGL(kern, dim3(gx, (unsigned)n_clips), dim3(256), smem, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
    return MISPEC_OK;
  }
  rc = setup_edges(p, workspace, workspace_bytes, s);
  if (rc != MISPEC_OK) return rc;
  return launch_cfg<1, 4, 1, 2, BMODE_FRAMED, AMODE_TOEPLITZ, false>(p, s);
}

}  // extern "C"
