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

// A[row, k] for the two A modes (row is a global row index; bounds already checked for k < K)
__device__ __forceinline__ float fetch_a(const KParams &p, int row, int k) {
  if (p.amode == AMODE_TOEPLITZ) {
    int tap = k - p.toep_stride * row;
    float v = 0.f;
    if (tap >= 0 && tap < p.n_taps && row < p.n_bins) v = p.a_re[tap];
    return v;
  }
  const bool cplx = p.a_im != nullptr;
  const int bin = cplx ? (row >> 1) : row;
  const float *src = (cplx && (row & 1)) ? p.a_im : p.a_re;
  float v = 0.f;
  if (bin < p.n_bins) v = src[(long long)bin * p.a_row_stride + k];
  return v;
}

// ---------------------------------------------------------------------------------
// MFMA kernel.  Workgroup = WM x WN waves; each wave owns MR x NR tiles of 32x32.
//   BM = WM*MR*32 basis rows,  BN = WN*NR*32 frames per workgroup.
// ---------------------------------------------------------------------------------
template <int WM, int WN, int MR, int NR, int BMODE>
__global__ void __launch_bounds__(WM *WN * 64) framed_gemm_kernel(const KParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int MT = WM * MR;
  constexpr int RPP = NT / 32;  // tile rows covered by one loader pass
  constexpr int APASS = BM / RPP;
  constexpr int A_STAGE = BM * LDT;
  constexpr int B_STAGE = (BMODE == BMODE_FRAMED) ? BN * LDT : KC * BN;
  constexpr int BPASS = (BMODE == BMODE_FRAMED) ? (BN / RPP) : (KC * BN / NT);
  static_assert(32 % RPP == 0, "loader pass must not straddle a 32-row tile");
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile / loader mismatch");
  static_assert(BMODE == BMODE_FRAMED || (NT % BN == 0 || BN % NT == 0), "planar loader shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *sA = reinterpret_cast<float *>(smem_raw);
  float *sB = sA + 2 * A_STAGE;
  long long *sColBase = reinterpret_cast<long long *>(sB + 2 * B_STAGE);
  int *sColPos = reinterpret_cast<int *>(sColBase + BN);
  int *sTileLo = sColPos + BN;
  int *sTileHi = sTileLo + MT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 31;
  const int lh = lane >> 5;

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

  const bool cplx = p.a_im != nullptr;
  const int rpb = cplx ? 2 : 1;

  // ---- per-column (frame) tables and per-row-tile K ranges
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
  if (tid < MT) {
    const int row_lo = m0 + tid * 32;
    int lo = 0, hi = 0;
    if (p.amode == AMODE_TOEPLITZ) {
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

  int kb = p.K, ke = 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int lo = sTileLo[i], hi = sTileHi[i];
    if (hi > lo) {
      kb = lo < kb ? lo : kb;
      ke = hi > ke ? hi : ke;
    }
  }
  kb = __builtin_amdgcn_readfirstlane(kb) & ~(KC - 1);
  ke = __builtin_amdgcn_readfirstlane(ke);
  const int nchunks = ke > kb ? (ke - kb + KC - 1) / KC : 0;

  // which of the workgroup's row tiles intersect K stage [kc, kc+KC)
  auto stage_mask = [&](int kc) -> unsigned {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int lo = sTileLo[i], hi = sTileHi[i];
      if (hi > kc && lo < kc + KC) m |= 1u << i;
    }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
  };

  float ra[APASS];
  float rb[BPASS];

  const int lr0 = tid >> 5;  // loader row within a pass
  const int lc = tid & 31;   // loader k offset

  auto load_stage = [&](int kc, unsigned amask) {
    const int k = kc + lc;
    const bool kin = k < p.K;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      const int row = ps * RPP + lr0;
      float v = 0.f;
      if ((amask >> ((ps * RPP) >> 5)) & 1u) {
        if (kin) v = fetch_a(p, m0 + row, k);
      }
      ra[ps] = v;
    }
    if (BMODE == BMODE_FRAMED) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
        const int j = ps * RPP + lr0;
        const long long base = sColBase[j];
        const int pos = sColPos[j];
        rb[ps] = fetch_sample(p.x, base, pos + k, p.n_samples, p.pad_mode, kin && base >= 0);
      }
    } else {
      constexpr int KPP = (NT >= BN) ? NT / BN : 1;  // k rows per pass
      constexpr int JPP = (NT >= BN) ? 1 : BN / NT;  // column groups per k row
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
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
        float v = 0.f;
        if (base >= 0 && kk < p.K) v = p.x[base + (long long)kk * p.x_k_stride];
        rb[ps] = v;
      }
    }
  };

  auto store_stage = [&](int buf) {
    float *a = sA + buf * A_STAGE;
    float *b = sB + buf * B_STAGE;
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) a[(ps * RPP + lr0) * LDT + lc] = ra[ps];
    if (BMODE == BMODE_FRAMED) {
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) b[(ps * RPP + lr0) * LDT + lc] = rb[ps];
    } else {
      constexpr int KPP = (NT >= BN) ? NT / BN : 1;
      constexpr int JPP = (NT >= BN) ? 1 : BN / NT;
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
        int kl, j;
        if (NT >= BN) {
          kl = ps * KPP + tid / BN;
          j = tid % BN;
        } else {
          kl = ps / JPP;
          j = (ps % JPP) * NT + tid;
        }
        b[kl * BN + j] = rb[ps];
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
      const unsigned wmask = (mask_cur >> (wm * MR));

#pragma unroll
      for (int q = 0; q < KC / 8; ++q) {
        f32x4v av[MR], bv[NR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          if ((wmask >> m) & 1u)
            av[m] = *reinterpret_cast<const f32x4v *>(a_base + m * 32 * LDT + 8 * q);
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          if (BMODE == BMODE_FRAMED) {
            bv[n] = *reinterpret_cast<const f32x4v *>(b_base + n * 32 * LDT + 8 * q);
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) bv[n][s] = b_base[(8 * q + s) * BN + n * 32];
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            if ((wmask >> m) & 1u) {
#pragma unroll
              for (int n = 0; n < NR; ++n)
                acc[m][n] =
                    __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][s], bv[n][s], acc[m][n], 0, 0, 0);
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
    if (p.store_mode == STORE_ROWS_INNER) {
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

template <int WM, int WN, int MR, int NR, int BMODE>
int launch_cfg(KParams p, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int MT = WM * MR;
  constexpr int A_STAGE = BM * LDT;
  constexpr int B_STAGE = (BMODE == BMODE_FRAMED) ? BN * LDT : KC * BN;
  constexpr size_t smem = sizeof(float) * 2 * (A_STAGE + B_STAGE) + sizeof(long long) * BN +
                          sizeof(int) * BN + sizeof(int) * 2 * MT;

  const int rows = p.amode == AMODE_TOEPLITZ ? p.n_bins : p.n_bins * (p.a_im ? 2 : 1);
  p.n_tiles_m = (rows + BM - 1) / BM;
  const long long tn = (p.n_cols + BN - 1) / BN;
  if (tn * p.n_tiles_m > 0x7fffffffLL) return fail(MISPEC_E_UNSUPPORTED, "grid too large%s");
  p.n_tiles_n = (int)tn;

  auto kern = framed_gemm_kernel<WM, WN, MR, NR, BMODE>;
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

int launch_framed(const KParams &p, int tile, hipStream_t stream) {
  const int rows = p.amode == AMODE_TOEPLITZ ? p.n_bins : p.n_bins * (p.a_im ? 2 : 1);
  if (tile == MISPEC_TILE_AUTO) {
    if (rows <= 32)
      tile = MISPEC_TILE_32x256;
    else if (rows <= 64)
      tile = MISPEC_TILE_64x256;
    else if (p.row_support) {
      // support-aware: every wave owns all row tiles of the workgroup so skipped
      // K stages shorten the whole workgroup instead of idling some waves
      if (rows <= 128)
        tile = MISPEC_TILE_128x128_TALL;
      else if (rows <= 192)
        tile = MISPEC_TILE_192x128;
      else
        tile = MISPEC_TILE_256x128;
    } else {
      tile = MISPEC_TILE_128x128;
    }
  }
  switch (tile) {
    case MISPEC_TILE_128x128:
      return launch_cfg<2, 2, 2, 2, BMODE_FRAMED>(p, stream);
    case MISPEC_TILE_32x256:
      return launch_cfg<1, 4, 1, 2, BMODE_FRAMED>(p, stream);
    case MISPEC_TILE_64x256:
      return launch_cfg<1, 4, 2, 2, BMODE_FRAMED>(p, stream);
    case MISPEC_TILE_128x128_TALL:
      return launch_cfg<1, 4, 4, 1, BMODE_FRAMED>(p, stream);
    case MISPEC_TILE_192x128:
      return launch_cfg<1, 4, 6, 1, BMODE_FRAMED>(p, stream);
    case MISPEC_TILE_256x128:
      return launch_cfg<1, 4, 8, 1, BMODE_FRAMED>(p, stream);
    default:
      return fail(MISPEC_E_INVALID, "unknown tile id%s");
  }
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
  if (n_filters <= 32) return launch_cfg<1, 4, 1, 2, BMODE_PLANAR>(p, s);
  if (n_filters <= 64) return launch_cfg<1, 4, 2, 2, BMODE_PLANAR>(p, s);
  return launch_cfg<2, 2, 2, 2, BMODE_PLANAR>(p, s);
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
  return launch_cfg<1, 4, 1, 2, BMODE_FRAMED>(p, static_cast<hipStream_t>(stream));
}

}  // extern "C"
