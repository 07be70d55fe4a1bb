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
//         generated on the fly from the waveform: reflect / zero padding is index
//         arithmetic inside the loader, frames are never materialised in HBM.
//
// The contraction runs on the matrix cores with v_mfma_f32_32x32x2_f32 (fp32 in, fp32
// accumulate: bit-for-bit an fmaf chain, so the 1e-4 parity bar is met with ~1e-6),
// 64-wide wavefronts, 32-deep K stages double-buffered through LDS, and the
// magnitude / power / phase / complex epilogue applied on the accumulators in
// registers before a (batch, bin, frame[,2]) store with frames innermost (coalesced).
//
// Variants of the same template:
//   * A as a banded Toeplitz matrix of FIR taps  -> strided decimation (utils.py:73-124)
//   * Bop read from a planar (clip, k, t) tensor -> filterbank matmul  (mel.py:188)
//   * per-row [start, stop) supports             -> CQT kernels skip their zero taps
//
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>

#include "mispec.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int KC = 32;   // K depth of one LDS stage
constexpr int LDT = 36;  // LDS row stride in floats: 16-B aligned rows, conflict-free ds_read_b128

enum { BMODE_FRAMED = 0, BMODE_PLANAR = 1 };
enum { AMODE_ROWS = 0, AMODE_TOEPLITZ = 1 };
enum { STORE_FRAMES_INNER = 0, STORE_ROWS_INNER = 1 };

struct KParams {
  // B operand (signal / planar tensor)
  const float *x;
  long long x_clip_stride;
  long long x_k_stride;  // planar mode: distance between successive k
  int n_clips;
  int n_samples;
  int hop;
  int pad;
  int pad_mode;
  int n_frames;
  long long n_cols;  // n_clips * n_frames
  // A operand
  const float *a_re;
  const float *a_im;
  long long a_row_stride;
  int n_bins;
  int K;
  const int *row_support;
  const float *row_scale;
  int amode;
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
  int out_row_offset;
  int store_mode;
  int out_len;
  int n_tiles_m;
  int n_tiles_n;
};

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

// ---------------------------------------------------------------------------------
// pointwise epilogue on one (bin, frame) pair, shared by the MFMA and the reference kernel
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void epilogue_store(const KParams &p, float *__restrict__ dst, float re,
                                               float im) {
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

// 4 consecutive floats with only element alignment guaranteed (hop / pad / clip length are
// arbitrary); the hardware handles dword-aligned 16-byte global loads.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// ---------------------------------------------------------------------------------
// MFMA kernel.  Workgroup = WM x WN waves; each wave owns MR x NR tiles of 32x32.
//   BM = WM*MR*32 basis rows,  BN = WN*NR*32 frames per workgroup.
// Loader geometry (256 threads): thread (r32 = tid>>3, c4 = tid&7) moves the 4 consecutive
// K elements 4*c4.. of row r32 of each 32-row "pass"; a pass of the A tile is one 32-row
// MFMA tile, a pass of the B tile is 32 consecutive frames.
// ---------------------------------------------------------------------------------
template <int WM, int WN, int MR, int NR, int BMODE, int AMODE, bool MASKED>
__global__ void __launch_bounds__(WM *WN * 64) framed_gemm_kernel(const KParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int MT = WM * MR;
  constexpr int RPP = NT / 8;  // tile rows covered by one loader pass (4 floats per thread)
  static_assert(RPP == 32, "loader geometry assumes 256 threads");
  constexpr int APASS = BM / RPP;
  constexpr int BPASS = BN / RPP;  // framed mode
  constexpr int A_STAGE = BM * LDT;
  constexpr int B_STAGE = (BMODE == BMODE_FRAMED) ? BN * LDT : KC * BN;
  constexpr int PPASS = KC * BN / NT;  // planar mode: scalar elements per thread
  static_assert(BMODE == BMODE_FRAMED || NT % BN == 0 || BN % NT == 0, "planar loader shape");
  constexpr int STORE_MODE = (AMODE == AMODE_TOEPLITZ) ? STORE_ROWS_INNER : STORE_FRAMES_INNER;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *sA = reinterpret_cast<float *>(smem_raw);
  float *sB = sA + 2 * A_STAGE;
  long long *sColBase = reinterpret_cast<long long *>(sB + 2 * B_STAGE);
  long long *sPassBase = sColBase + BN;  // [BPASS] clip offset of a single-clip pass, else -1
  int *sColPos = reinterpret_cast<int *>(sPassBase + 8);
  int *sPassPos = sColPos + BN;          // [BPASS] signal position of the pass's first frame
  int *sTileLo = sPassPos + 8;
  int *sTileHi = sTileLo + MT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int r32 = tid >> 3;  // loader row inside a pass
  const int c4 = tid & 7;    // loader K offset / 4

  // ---- XCD-aware tile order: workgroup b runs on XCD b % 8; give every XCD a contiguous
  // range of tiles with the frame-tile index fastest, so the basis rows an XCD streams
  // stay resident in its private L2 while the waveform is streamed through.
  int tile;
  {
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = b & 7, idx = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = tile / p.n_tiles_n;
  const int tile_n = tile - tile_m * p.n_tiles_n;
  const int m0 = tile_m * BM;
  const long long n0 = (long long)tile_n * BN;

  const bool cplx = (AMODE == AMODE_ROWS) && p.a_im != nullptr;
  const int rpb = cplx ? 2 : 1;

  // ---- per-column (frame) tables, per-pass fast-path descriptors, per-row-tile K ranges
  for (int j = tid; j < BN; j += NT) {
    const long long col = n0 + j;
    long long base = -1;
    int pos = 0;
    if (col < p.n_cols) {
      const int c = (int)(col / p.n_frames);
      const int t = (int)(col - (long long)c * p.n_frames);
      if (BMODE == BMODE_FRAMED) {
        base = (long long)c * p.x_clip_stride;
        pos = t * p.hop - p.pad;
      } else {
        base = (long long)c * p.x_clip_stride + t;
      }
    }
    sColBase[j] = base;
    sColPos[j] = pos;
  }
  if (BMODE == BMODE_FRAMED && tid < BPASS) {
    const long long c0 = n0 + tid * 32, c1 = c0 + 31;
    long long base = -1;
    int pos = 0;
    if (c1 < p.n_cols) {
      const int ca = (int)(c0 / p.n_frames), cb = (int)(c1 / p.n_frames);
      if (ca == cb) {
        base = (long long)ca * p.x_clip_stride;
        pos = (int)(c0 - (long long)ca * p.n_frames) * p.hop - p.pad;
      }
    }
    sPassBase[tid] = base;
    sPassPos[tid] = pos;
  }
  if (tid < MT) {
    const int row_lo = m0 + tid * 32;
    int lo = 0, hi = 0;
    if (AMODE == AMODE_TOEPLITZ) {
      if (row_lo < p.n_bins) hi = p.K;
    } else {
      const int bin_lo = row_lo / rpb;
      int bin_hi = (row_lo + 32 + rpb - 1) / rpb;
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
  const int nchunks = ke > kb ? (ke - kb + KC - 1) / KC : 0;

  // which of the workgroup's row tiles intersect K stage [kc, kc+KC)
  auto stage_mask = [&](int kc) -> unsigned {
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
      aptr[ps] = src + (long long)bin * p.a_row_stride + 4 * c4;
    }
  }
  const float *bptr[BPASS];
  int pass_ok[BPASS];   // wave-uniform: the pass's 32 frames lie in one clip
  int pass_pos[BPASS];  // wave-uniform: signal position of the pass's first frame
  if (BMODE == BMODE_FRAMED) {
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) {
      const long long pb = sPassBase[ps];
      const int pp = sPassPos[ps];
      pass_ok[ps] = __builtin_amdgcn_readfirstlane(pb >= 0 ? 1 : 0);
      pass_pos[ps] = __builtin_amdgcn_readfirstlane(pp);
      bptr[ps] = p.x + (pb < 0 ? 0 : pb) + pp + (long long)r32 * p.hop + 4 * c4;
    }
  }

  // K stages [plain_lo, plain_hi] (multiples of KC) can be loaded with unpredicated 16-byte
  // loads for every pass of both tiles: all frames of every pass in one clip, inside the signal
  int plain_lo = 0x7fffffff, plain_hi = -1;
  if (AMODE == AMODE_ROWS && BMODE == BMODE_FRAMED) {
    long long lo = 0, hi = (long long)p.K - KC;
    bool ok = true;
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) {
      ok = ok && pass_ok[ps];
      const long long a = -(long long)pass_pos[ps];
      const long long b = (long long)p.n_samples - KC - pass_pos[ps] - 31LL * p.hop;
      lo = a > lo ? a : lo;
      hi = b < hi ? b : hi;
    }
    if (ok && hi >= lo) {
      plain_lo = (int)lo;
      plain_hi = (int)hi;
    }
  }

  f32x4v ra[APASS];
  f32x4v rb[(BMODE == BMODE_FRAMED) ? BPASS : 1];
  float rp[(BMODE == BMODE_PLANAR) ? PPASS : 1];

  auto stage_is_plain = [&](int kc, unsigned amask) -> bool {
    bool ok = kc >= plain_lo && kc <= plain_hi;
    if (MASKED) ok = ok && (amask == ((1u << MT) - 1u));
    return ok;
  };

  auto load_stage = [&](int kc, unsigned amask) {
    if (stage_is_plain(kc, amask)) {
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps)
        ra[ps] = *reinterpret_cast<const f32x4u *>(aptr[ps] + kc);
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps)
        rb[ps] = *reinterpret_cast<const f32x4u *>(bptr[ps] + kc);
      return;
    }
    const int k = kc + 4 * c4;
    const bool full_k = (kc + KC) <= p.K;  // uniform
    // ---------------- A tile
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      f32x4v v = {0.f, 0.f, 0.f, 0.f};
      if ((amask >> ps) & 1u) {
        if (AMODE == AMODE_ROWS) {
          if (full_k) {
            v = *reinterpret_cast<const f32x4u *>(aptr[ps] + kc);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool ok = (k + e) < p.K;
              const float t = (aptr[ps] - 4 * c4)[ok ? k + e : 0];  // clamp into the row
              v[e] = ok ? t : 0.f;
            }
          }
        } else {
          const int row = m0 + ps * 32 + r32;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int tap = k + e - p.toep_stride * row;
            const bool ok = tap >= 0 && tap < p.n_taps && row < p.n_bins && (k + e) < p.K;
            const float t = p.a_re[ok ? tap : 0];
            v[e] = ok ? t : 0.f;
          }
        }
      }
      ra[ps] = v;
    }
    // ---------------- B tile
    if (BMODE == BMODE_FRAMED) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
        // per-pass fast path: the pass's 32 frames lie in one clip and this K stage of all of
        // them is inside the signal -> one unpredicated 16-byte load
        const int pf = pass_pos[ps];
        const bool fast = pass_ok[ps] && full_k && (pf + kc >= 0) &&
                          ((long long)pf + 31LL * p.hop + kc + KC <= (long long)p.n_samples);
        if (fast) {
          rb[ps] = *reinterpret_cast<const f32x4u *>(bptr[ps] + kc);
        } else {
          const int j = ps * 32 + r32;
          const long long base = sColBase[j];
          const int pos = sColPos[j];
          f32x4v v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int pp = pos + k + e;
            if (p.pad_mode == MISPEC_PAD_REFLECT) {
              pp = pp < 0 ? -pp : pp;
              pp = pp >= p.n_samples ? 2 * p.n_samples - 2 - pp : pp;
            }
            const bool ok = base >= 0 && (k + e) < p.K && pp >= 0 && pp < p.n_samples;
            const float t = p.x[ok ? base + pp : 0];
            v[e] = ok ? t : 0.f;
          }
          rb[ps] = v;
        }
      }
    } else {
      constexpr int KPP = (NT >= BN) ? NT / BN : 1;  // k rows per pass
      constexpr int JPP = (NT >= BN) ? 1 : BN / NT;  // column groups per k row
#pragma unroll
      for (int ps = 0; ps < PPASS; ++ps) {
        int kl, j;
        if (NT >= BN) {
          kl = ps * KPP + tid / BN;
          j = tid % BN;
        } else {
          kl = ps / JPP;
          j = (ps % JPP) * NT + tid;
        }
        const int kk = kc + kl;
        const long long base = sColBase[j];
        const bool ok = base >= 0 && kk < p.K;
        const float t = p.x[ok ? base + (long long)kk * p.x_k_stride : 0];
        rp[ps] = ok ? t : 0.f;
      }
    }
  };

  auto store_stage = [&](int buf) {
    float *a = sA + buf * A_STAGE;
    float *b = sB + buf * B_STAGE;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps)
      *reinterpret_cast<f32x4v *>(a + (ps * 32 + r32) * LDT + 4 * c4) = ra[ps];
    if (BMODE == BMODE_FRAMED) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps)
        *reinterpret_cast<f32x4v *>(b + (ps * 32 + r32) * LDT + 4 * c4) = rb[ps];
    } else {
      constexpr int KPP = (NT >= BN) ? NT / BN : 1;
      constexpr int JPP = (NT >= BN) ? 1 : BN / NT;
#pragma unroll
      for (int ps = 0; ps < PPASS; ++ps) {
        int kl, j;
        if (NT >= BN) {
          kl = ps * KPP + tid / BN;
          j = tid % BN;
        } else {
          kl = ps / JPP;
          j = (ps % JPP) * NT + tid;
        }
        b[kl * BN + j] = rp[ps];
      }
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  if (nchunks > 0) {
    unsigned mask_cur = stage_mask(kb);
    load_stage(kb, mask_cur);
    store_stage(0);
    __syncthreads();

    for (int c = 0; c < nchunks; ++c) {
      const int buf = c & 1;
      const int kc_next = kb + (c + 1) * KC;
      const bool more = (c + 1) < nchunks;
      unsigned mask_next = 0;
      if (more) {
        mask_next = stage_mask(kc_next);
        load_stage(kc_next, mask_next);  // global loads in flight under the MFMAs below
      }

      const float *a_base = sA + buf * A_STAGE + ((wm * MR) * 32 + li) * LDT + 4 * lh;
      const float *b_base;
      if (BMODE == BMODE_FRAMED)
        b_base = sB + buf * B_STAGE + ((wn * NR) * 32 + li) * LDT + 4 * lh;
      else
        b_base = sB + buf * B_STAGE + (4 * lh) * BN + (wn * NR) * 32 + li;
      const unsigned wmask = MASKED ? (mask_cur >> (wm * MR)) : ~0u;

      // fragments of K group q (8 K elements): lane (li, lh) holds k = 8q + 4lh + s for MFMA s
      f32x4v av[2][MR], bv[2][NR];
      auto load_frags = [&](int q, int slot) {
#pragma unroll
        for (int m = 0; m < MR; ++m)
          av[slot][m] = *reinterpret_cast<const f32x4v *>(a_base + m * 32 * LDT + 8 * q);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          if (BMODE == BMODE_FRAMED) {
            bv[slot][n] = *reinterpret_cast<const f32x4v *>(b_base + n * 32 * LDT + 8 * q);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) bv[slot][n][s] = b_base[(8 * q + s) * BN + n * 32];
          }
        }
      };
      load_frags(0, 0);
#pragma unroll
      for (int q = 0; q < KC / 8; ++q) {
        if (q + 1 < KC / 8) load_frags(q + 1, (q + 1) & 1);  // prefetch under this group's MFMAs
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          if (!MASKED || ((wmask >> m) & 1u)) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
              for (int n = 0; n < NR; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][m][s], bv[q & 1][n][s],
                                                                 acc[m][n], 0, 0, 0);
            }
          }
        }
      }

      if (more) store_stage(buf ^ 1);
      mask_cur = mask_next;
      __syncthreads();
    }
  }

  // ---- epilogue.  Accumulator element e of lane (li, lh) is D[row = (e&3) + 8*(e>>2) + 4*lh][col = li].
  // Each wave restages one 32x32 tile at a time through a private LDS patch so that the
  // pointwise epilogue below is a single dynamic loop (one code instance, static register
  // indexing only in the ds_write fan-out) and so that stores are contiguous along the
  // innermost output dimension for both store modes.
  __syncthreads();  // every wave is done with the K-stage buffers
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
#pragma unroll
          for (int e = 0; e < 16; ++e)
            sC[((e & 3) + 8 * (e >> 2) + 4 * lh) * LDC + li] = acc[m][n][e];
        }
      }
    }
    __syncthreads();
    const int tm = ti / NR, tn = ti - tm * NR;
    const int row_base = m0 + (wm * MR + tm) * 32;           // first basis row of this tile
    const long long col_base = n0 + (wn * NR + tn) * 32;     // first frame column of this tile
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
          const long long o = (long long)t * 32 + row;
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

template <int WM, int WN, int MR, int NR, int BMODE, int AMODE = AMODE_ROWS, bool MASKED = true>
int launch_cfg(KParams p, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int MT = WM * MR;
  constexpr int A_STAGE = BM * LDT;
  constexpr int B_STAGE = (BMODE == BMODE_FRAMED) ? BN * LDT : KC * BN;
  constexpr size_t smem = sizeof(float) * 2 * (A_STAGE + B_STAGE) + sizeof(long long) * (BN + 8) +
                          sizeof(int) * (BN + 8) + sizeof(int) * 2 * MT;

  const int rows = p.amode == AMODE_TOEPLITZ ? p.n_bins : p.n_bins * (p.a_im ? 2 : 1);
  p.n_tiles_m = (rows + BM - 1) / BM;
  const long long tn = (p.n_cols + BN - 1) / BN;
  if (tn * p.n_tiles_m > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  p.n_tiles_n = (int)tn;

  auto kern = framed_gemm_kernel<WM, WN, MR, NR, BMODE, AMODE, MASKED>;
  // opt in to > 64 KiB of dynamic LDS once per (kernel, device)
  static std::atomic<unsigned long long> configured{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(MISPEC_E_HIP, "hipGetDevice failed%s");
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(configured.load(std::memory_order_acquire) & bit)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return fail(MISPEC_E_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured.fetch_or(bit, std::memory_order_release);
  }
  const unsigned grid = (unsigned)(p.n_tiles_m * p.n_tiles_n);
  if (grid == 0) return MISPEC_OK;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MISPEC_E_HIP, "kernel launch: %s", hipGetErrorString(e));
  return MISPEC_OK;
}

template <int WM, int WN, int MR, int NR>
int launch_pick_mask(const KParams &p, bool masked, hipStream_t stream) {
  if (masked) return launch_cfg<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, true>(p, stream);
  return launch_cfg<WM, WN, MR, NR, BMODE_FRAMED, AMODE_ROWS, false>(p, stream);
}

int launch_tile(const KParams &p, int tile, hipStream_t stream) {
  // skipping row tiles per K stage only pays (and is only needed) with per-row supports
  const bool masked = p.row_support != nullptr;
  switch (tile) {
    case MISPEC_TILE_128x128:
      return launch_pick_mask<2, 2, 2, 2>(p, masked, stream);
    case MISPEC_TILE_32x256:  // one row tile: the workgroup K range is the tile's range
      return launch_cfg<1, 4, 1, 2, BMODE_FRAMED, AMODE_ROWS, false>(p, stream);
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
  if (rows <= 32) return MISPEC_TILE_32x256;
  if (rows <= 64) return MISPEC_TILE_64x256;
  if (support) {
    // support-aware: every wave owns all row tiles of the workgroup so skipped
    // K stages shorten the whole workgroup instead of idling some waves
    if (rows <= 128) return MISPEC_TILE_128x128_TALL;
    if (rows <= 192) return MISPEC_TILE_192x128;
    return MISPEC_TILE_256x128;
  }
  return MISPEC_TILE_128x128;
}

int launch_framed(const KParams &p, int tile, hipStream_t stream) {
  const int rpb = p.a_im ? 2 : 1;
  const int rows = p.n_bins * rpb;
  if (tile != MISPEC_TILE_AUTO) return launch_tile(p, tile, stream);
  if (p.row_support || rows <= 128) return launch_tile(p, auto_tile(rows, p.row_support != nullptr), stream);
  // dense basis, many rows: full 128-row workgroups run the unmasked kernel; the leftover
  // rows (e.g. the Nyquist bin of an n_fft/2+1 STFT) go to a second, narrow launch instead
  // of a 17th mostly-empty row block.
  const int bins_per_wg = 128 / rpb;
  const int main_bins = (p.n_bins / bins_per_wg) * bins_per_wg;
  KParams q = p;
  q.n_bins = main_bins;
  int rc = launch_tile(q, MISPEC_TILE_128x128, stream);
  if (rc != MISPEC_OK) return rc;
  const int rem = p.n_bins - main_bins;
  if (rem == 0) return MISPEC_OK;
  q = p;
  q.n_bins = rem;
  q.a_re = p.a_re + (long long)main_bins * p.a_row_stride;
  if (p.a_im) q.a_im = p.a_im + (long long)main_bins * p.a_row_stride;
  if (p.row_scale) q.row_scale = p.row_scale + main_bins;
  q.out_row_offset = p.out_row_offset + main_bins;
  return launch_tile(q, auto_tile(rem * rpb, false), stream);
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
  if ((long long)(a->n_frames - 1) * a->hop + a->kernel > 0x7fffffffLL)
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
  p.amode = AMODE_ROWS;
  p.epilogue = a->epilogue;
  p.im_sign = a->im_sign;
  p.eps = a->eps;
  p.power = a->power;
  p.out = a->out;
  p.out_clip_stride = a->out_clip_stride;
  p.out_row_stride = a->out_row_stride;
  p.out_row_offset = a->out_row_offset;
  p.store_mode = STORE_FRAMES_INNER;
  return MISPEC_OK;
}

}  // namespace

extern "C" {

int mispec_version(void) { return MISPEC_ABI_VERSION; }

const char *mispec_last_error(void) { return g_err; }

int mispec_framed_gemm_f32(const mispec_framed_gemm_args *args, void *stream) {
  KParams p;
  int rc = fill_params(args, p);
  if (rc != MISPEC_OK) return rc;
  return launch_framed(p, args->tile, static_cast<hipStream_t>(stream));
}

int mispec_framed_gemm_f32_ref(const mispec_framed_gemm_args *args, void *stream) {
  KParams p;
  int rc = fill_params(args, p);
  if (rc != MISPEC_OK) return rc;
  const long long total = p.n_cols * p.n_bins;
  const long long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  hipLaunchKernelGGL(framed_gemm_ref_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
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
  p.amode = AMODE_ROWS;
  p.epilogue = MISPEC_EPI_REAL;
  p.im_sign = 1.f;
  p.out = out;
  p.out_clip_stride = (long long)n_filters * n_frames;
  p.out_row_stride = n_frames;
  p.store_mode = STORE_FRAMES_INNER;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_filters <= 32) return launch_cfg<1, 4, 1, 2, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
  if (n_filters <= 64) return launch_cfg<1, 4, 2, 2, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
  return launch_cfg<2, 2, 2, 2, BMODE_PLANAR, AMODE_ROWS, false>(p, s);
}

int mispec_fir_decimate_f32(const float *x, int64_t x_clip_stride, int32_t n_clips,
                            int32_t n_samples, const float *taps, int32_t n_taps, int32_t stride,
                            int32_t pad, float *y, int64_t y_clip_stride, int32_t n_out,
                            void *stream) {
  if (!x || !taps || !y) return fail(MISPEC_E_INVALID, "NULL device pointer%s");
  if (n_clips <= 0 || n_samples <= 0 || n_taps <= 0 || stride <= 0 || pad < 0 || n_out <= 0)
    return fail(MISPEC_E_INVALID, "non-positive size%s");
  const long long span = (long long)n_samples + 2LL * pad - n_taps;
  if (span < 0 || (long long)n_out != span / stride + 1)
    return fail(MISPEC_E_INVALID, "n_out != (n_samples + 2*pad - n_taps)/stride + 1%s");
  if ((long long)n_out * stride + n_taps > 0x7fffffffLL)
    return fail(MISPEC_E_UNSUPPORTED, "signal position overflows int32%s");
  // 32 consecutive outputs form one "frame" of the Toeplitz contraction:
  //   y[32 q + r] = sum_m x[32*stride*q + m - pad] * taps[m - stride*r],  m < n_taps + 31*stride
  KParams p;
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
  p.amode = AMODE_TOEPLITZ;
  p.toep_stride = stride;
  p.n_taps = n_taps;
  p.epilogue = MISPEC_EPI_REAL;
  p.im_sign = 1.f;
  p.out = y;
  p.out_clip_stride = y_clip_stride;
  p.out_row_stride = 0;
  p.store_mode = STORE_ROWS_INNER;
  p.out_len = n_out;
  return launch_cfg<1, 4, 1, 2, BMODE_FRAMED, AMODE_TOEPLITZ, false>(p, static_cast<hipStream_t>(stream));
}

}  // extern "C"
