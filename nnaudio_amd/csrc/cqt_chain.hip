// cqt_chain.hip -- CQT1992v2's fp32 contraction (cqt.py:749-750 with the kernels of utils.py:457-469) as ONE float32 FMA
// chain per output, with the frames' samples kept in LDS delay lines instead of being gathered per K stage.
//
// The arithmetic is that of framed_gemm_kernel<.., T16> (mispec.hip): 16 x 16 x 4 tiles on v_mfma_f32_16x16x4_f32, lane
// (col l16, k lane lq) feeding tap 4n + lq to MFMA n, so that every accumulator is one ascending fp32 FMA chain over the
// taps -- the reference's conv1d (tests/test_reference_order.py is the acceptance test: same bits).  What changes is where
// the operands come from:
//
//   * B operand (signal).  A wave owns 16 consecutive frames of one clip (one MFMA column tile).  Frame f at 16-tap
//     sub-stage s needs padded samples  (t0 + f) hop + 16 s .. + 16:  what frame f reads now, frame f - 1 reads hop / 16
//     sub-stages later.  The wave keeps that delay line in LDS: a ring of NR rows of `hop` samples (row r = samples
//     [U0 + r hop, + hop), slot r mod NR, rows 8 dwords apart in bank space so that the 16 frames of a ds_read_b128 fall on
//     different banks), refilled 64 samples at a time by global_load_lds_dword behind frame 0 -- each sample enters LDS once
//     per column tile (HBM/L2 traffic: (15 hop + K) / (16 hop) of the clip bytes per bank pass) instead of once per frame and
//     K stage.  Inside every aligned group of 16 samples the DMA stores sample 4 n + q at position 4 q + n, so a lane's four
//     taps of four successive MFMAs are ONE ds_read_b128.  Virtual padding (reflect / zero) is resolved in the DMA's
//     per-lane source address: no padded copy, no edge workspace.
//   * A operand (basis).  Prepared once per bank (mispec_chain_basis_f32) as a stream of 1 KB "bricks" -- 16 rows x 16 taps
//     in MFMA fragment order [lane][4 taps of 4 MFMAs] -- in exactly the order a workgroup consumes them; taps outside a
//     16-row tile's support are not stored at all.  The four waves of a workgroup (4 column tiles) share the bricks through
//     three LDS buffers of 9 bricks filled by global_load_lds_dwordx4 two batches ahead; a brick is one ds_read_b128 per
//     lane and feeds 4 MFMAs.
//   * Work units.  The bank's 16-row tiles (8 bins; supports nested: CQT kernels are centred) are split into row sets; a
//     workgroup = 4 column tiles x one row set, all waves doing identical work per sub-stage.  Units are issued largest
//     first so that the dispatcher's tail is made of the short ones.
//
// Bounds: MFMA (fp32 matrix pipe, 157 TFLOP/s); LDS reads 1 KB per 4 MFMAs per wave; L2 -> LDS 1 KB per brick per workgroup.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <vector>

#include "mispec.h"
#include "mispec_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CH_NMAX = 9;      // 16-row tiles per row set = bricks per batch buffer
constexpr int CH_NBUF = 3;      // brick buffers (batch b multiplies one, b + 1 has landed, b + 2 is in flight)
constexpr int CH_QMAX = 8;      // sub-stages per batch (bounds the ring's look-ahead: 3 batches <= 24 sub-stages)
constexpr int CH_MAXSETS = 8;
constexpr int CH_SKEW = 8;      // dwords between ring rows in bank space
constexpr int CH_BRICK = 1024;  // bytes
constexpr int CH_ZERO_BYTES = 256;

struct ChainSetDev {
  int n_tiles;
  int s_lo;       // first 16-tap sub-stage of the set (of its longest tile)
  int batch0;     // index of the set's first batch in the table
  int n_batches;
  long long brick0;  // first brick of the set's stream
  int tile[CH_NMAX];
  int pad_;
};

struct ChainArgs {
  const float *x;
  long long x_clip_stride;
  int n_clips, n_samples, hop, pad, pad_mode, n_frames;
  int ct_per_clip;  // 16-frame column tiles per clip
  int n_ct;         // column tiles in all
  int nr;           // ring rows
  int ring_bytes;   // per wave
  const float *zeros;
  const int *batches;  // (q << 8) | n per batch
  const float *bricks;
  const float *row_scale;
  int n_bins, epilogue;
  float im_sign, eps, power;
  float *out;
  long long out_clip_stride, out_row_stride;
  int out_row_offset;
  int n_sets;
  int debug;
  int first_wg[CH_MAXSETS + 1];
  ChainSetDev set[CH_MAXSETS];
};

// 16 / 4 bytes per lane, global -> LDS at lds_addr + 16 / 4 lane (not seen by the compiler: waits are stated by hand)
__device__ __forceinline__ void ch_dma16(const void *sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void ch_dma4(const void *src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void ch_dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// the pointwise epilogue of mispec.hip (epilogue_store): same operations in the same order
__device__ __forceinline__ void ch_epilogue_store(const ChainArgs &p, float *__restrict__ dst, float re, float im) {
  switch (p.epilogue) {
    case MISPEC_EPI_COMPLEX:
      *reinterpret_cast<float2 *>(dst) = make_float2(re, im);
      break;
    case MISPEC_EPI_MAGNITUDE:
      dst[0] = sqrtf(re * re + im * im + p.eps);
      break;
    case MISPEC_EPI_POWER: {
      const float s = re * re + im * im + p.eps;
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
      const float a = atan2f(im, re);
      *reinterpret_cast<float2 *>(dst) = make_float2(cosf(a), sinf(a));
    } break;
    default:
      dst[0] = re;
      break;
  }
}

// read cursor of a wave's ring: sub-stage s' of frame f is row (f + s' 16 / hop), offset (s' 16) mod hop
struct RingCursor {
  int slot;  // row of frame 0, mod NR
  int off;   // samples into the row
};

// One batch of q sub-stages with N active tiles: bricks [j N + m] of the batch buffer at a_addr (this lane's 16 bytes of
// brick 0), the wave's ring at ring_addr (this lane's frame row 0 + 16 lq bytes).  Fragments of sub-stage j + 1 are read
// under the MFMAs of sub-stage j.
template <int N>
__device__ __forceinline__ void chain_batch(f32x4 (&acc)[CH_NMAX], const unsigned char *smem, const unsigned a_addr, const unsigned ring_addr,
                                            const int q, RingCursor &rc, const int f, const int nr, const int hop, const int row_bytes, const int debug) {
  f32x4 a0[N], a1[N], b0, b1;
  int jl = 0;  // next sub-stage to load
  auto load = [&](f32x4(&a)[N], f32x4 &b) __attribute__((always_inline)) {
    int slot = rc.slot + f;
    slot = slot >= nr ? slot - nr : slot;
    b = *reinterpret_cast<const f32x4 *>(smem + ring_addr + slot * row_bytes + rc.off * 4);
#pragma unroll
    for (int m = 0; m < N; ++m) a[m] = *reinterpret_cast<const f32x4 *>(smem + a_addr + (jl * N + m) * CH_BRICK);
    ++jl;
    rc.off += 16;
    if (rc.off == hop) {
      rc.off = 0;
      rc.slot = rc.slot + 1 == nr ? 0 : rc.slot + 1;
    }
  };
  auto mul = [&](const f32x4(&a)[N], const f32x4 &b) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int m = 0; m < N; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][jj], b[jj], acc[m], 0, 0, 0);
  };
  load(a0, b0);
  int j = 0;
  while (true) {
    if (j + 1 < q) load(a1, b1);
    if (!(debug & 1)) mul(a0, b0);
    if (++j >= q) break;
    if (j + 1 < q) load(a0, b0);
    if (!(debug & 1)) mul(a1, b1);
    if (++j >= q) break;
  }
}

__global__ void __launch_bounds__(256) cqt_chain_kernel(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f = lane & 15, lq = lane >> 4;

  int s = 0;
  while (s + 1 < a.n_sets && (int)blockIdx.x >= a.first_wg[s + 1]) ++s;
  const ChainSetDev &S = a.set[s];
  const int g = (int)blockIdx.x - a.first_wg[s];
  const int ct_raw = 4 * g + wave;
  const bool live = ct_raw < a.n_ct;
  const int ct = live ? ct_raw : a.n_ct - 1;  // idle waves of the last group shadow the last column tile (nothing stored)
  const int clip = ct / a.ct_per_clip;
  const int t0 = (ct - clip * a.ct_per_clip) * 16;
  const float *xc = a.x + (long long)clip * a.x_clip_stride;
  const int hop = a.hop, L = a.n_samples, nr = a.nr;
  const int row_bytes = (hop + CH_SKEW) * 4;
  const int U0 = t0 * hop + 16 * S.s_lo - a.pad;  // signal position of ring sample v = 0
  const unsigned ring_base = (unsigned)(wave * a.ring_bytes);
  const unsigned a_base = (unsigned)(4 * a.ring_bytes);
  const int nb = S.n_batches;
  const int *const bt = a.batches + S.batch0;

  // ---- ring fill: block `blk` = samples v in [64 blk, +64) of the wave's delay line; inside each group of 16 the lane at
  // position 4 q + n fetches sample 4 n + q
  const int lperm = 16 * (lane >> 4) + 4 * (lane & 3) + ((lane >> 2) & 3);
  int fill_blk = 0, fill_slot = 0, fill_off = 0;
  auto fill_block = [&]() __attribute__((always_inline)) {
    int pos = U0 + 64 * fill_blk + lperm;
    if (a.pad_mode == MISPEC_PAD_REFLECT) {
      pos = pos < 0 ? -pos : pos;
      pos = pos >= L ? 2 * L - 2 - pos : pos;
    }
    const bool in = pos >= 0 && pos < L;
    pos = pos < 0 ? 0 : (pos >= L ? L - 1 : pos);
    // reflect: beyond the virtually padded clip only zero taps / unstored frames read (any finite sample will do);
    // zero / no padding: everything outside the clip is zero
    const float *src = (in || a.pad_mode == MISPEC_PAD_REFLECT) ? xc + pos : a.zeros + lane;
    ch_dma4(src, ring_base + (unsigned)(fill_slot * row_bytes + fill_off * 4));
    ++fill_blk;
    fill_off += 64;
    if (fill_off == hop) {
      fill_off = 0;
      fill_slot = fill_slot + 1 == nr ? 0 : fill_slot + 1;
    }
  };
  // ---- brick stream: batch bb -> buffer bb % 3, the batch's bricks dealt out to the four waves
  long long issue_brick = S.brick0;
  auto issue_a = [&](int bb) __attribute__((always_inline)) {
    const int e = bt[bb];
    const int cnt = (e & 0xff) * (e >> 8);
    const unsigned dst = a_base + (unsigned)((bb % CH_NBUF) * CH_NMAX * CH_BRICK);
    for (int k = wave; k < cnt; k += 4)
      ch_dma16(a.bricks + (issue_brick + k) * (CH_BRICK / 4), (unsigned)lane * 16u, dst + (unsigned)k * CH_BRICK);
    issue_brick += cnt;
  };

  const int ring_samples = nr * hop;
  if (!(a.debug & 2))
    for (int i = 0; i < ring_samples / 64; ++i) fill_block();
  if (nb > 0) issue_a(0);
  if (nb > 1) issue_a(1);
  ch_dma_barrier();

  f32x4 acc[CH_NMAX];
#pragma unroll
  for (int m = 0; m < CH_NMAX; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  RingCursor rc{0, 0};
  const unsigned ring_addr = ring_base + (unsigned)lq * 16u;
  int sdone = 0;
  for (int b = 0; b < nb; ++b) {
    const int e = bt[b];
    const int n = e & 0xff, q = e >> 8;
    if (b + 2 < nb && !(a.debug & 4)) issue_a(b + 2);
    // refill behind frame 0: sub-stages < sdone are dead
    if (!(a.debug & 2))
      while (64 * (fill_blk + 1) <= 16 * sdone + ring_samples) fill_block();
    const unsigned a_addr = a_base + (unsigned)((b % CH_NBUF) * CH_NMAX * CH_BRICK) + (unsigned)lane * 16u;
    switch (n) {
#define CH_CASE(N)                                                                        \
  case N:                                                                                 \
    chain_batch<N>(acc, smem, a_addr, ring_addr, q, rc, f, nr, hop, row_bytes, a.debug); \
    break;
      CH_CASE(1)
      CH_CASE(2)
      CH_CASE(3)
      CH_CASE(4)
      CH_CASE(5)
      CH_CASE(6)
      CH_CASE(7)
      CH_CASE(8)
      CH_CASE(9)
#undef CH_CASE
      default:
        break;
    }
    sdone += q;
    ch_dma_barrier();
  }

  // ---- epilogue: element e of lane (f, lq) of tile m is D[row 16 tile + 4 lq + e][frame t0 + f]: (re, im) of bins
  // 8 tile + 2 lq and + 1 sit in one lane
  const int t = t0 + f;
  if (!live || t >= a.n_frames || (a.debug & 8)) return;
  const int E = (a.epilogue == MISPEC_EPI_COMPLEX || a.epilogue == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
  float *const obase = a.out + (long long)clip * a.out_clip_stride + (long long)t * E;
#pragma unroll
  for (int m = 0; m < CH_NMAX; ++m) {
    if (m < S.n_tiles) {
      const int bin0 = 8 * S.tile[m] + 2 * lq;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int bin = bin0 + h;
        if (bin < a.n_bins) {
          float re = acc[m][2 * h];
          float im = a.im_sign * acc[m][2 * h + 1];
          if (a.row_scale) {
            const float sc = a.row_scale[bin];
            re *= sc;
            im *= sc;
          }
          ch_epilogue_store(a, obase + (long long)(a.out_row_offset + bin) * a.out_row_stride, re, im);
        }
      }
    }
  }
}

// brick (set stream order) <- taps: thread (lane, j) of brick i writes A[row 16 tile + (lane & 15)][16 s + 4 j + (lane >> 4)]
__global__ void __launch_bounds__(256) chain_pack_kernel(const float *__restrict__ re, const float *__restrict__ im, long long row_stride, int n_bins,
                                                         int K, const int2 *__restrict__ brick_map, long long n_bricks, float *__restrict__ dst) {
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_bricks) return;
  const int lane = threadIdx.x & 63;
  const int2 bm = brick_map[i];
  const int row = 16 * bm.x + (lane & 15);
  const int bin = row >> 1;
  const float *src = (row & 1) ? im : re;
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 16 * bm.y + 4 * j + (lane >> 4);
    v[j] = (bin < n_bins && k < K) ? src[(long long)bin * row_stride + k] : 0.f;
  }
  *reinterpret_cast<f32x4 *>(dst + i * (CH_BRICK / 4) + lane * 4) = v;
}

// ------------------------------------------------------------------------------------------------------------
// host: the plan (tiles, row sets, batches, brick order) from the supports
// ------------------------------------------------------------------------------------------------------------
struct ChainSetHost {
  int n_tiles = 0;
  int tile[CH_NMAX];
  int lo[CH_NMAX], hi[CH_NMAX];  // sub-stages
  long long brick0 = 0, n_bricks = 0;
  int batch0 = 0, n_batches = 0;
  long long cost = 0;  // bricks + a quarter of the single-tile sub-stages (dependent MFMAs issue at 40 / 32 cycles)
};

struct ChainPlan {
  bool ok = false;
  int n_sets = 0;
  ChainSetHost set[CH_MAXSETS];
  std::vector<int> batches;
  std::vector<int2> brick_map;
  long long n_bricks = 0;
  long long batch_off = 0, map_off = 0, brick_off = 0, bytes = 0;
};

int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

// `want_map`: also list (tile, sub-stage) of every brick (the pack pass only)
ChainPlan chain_plan(const int32_t *sup, int n_bins, int K, bool want_map) {
  ChainPlan pl;
  const int n_tiles = (2 * n_bins + 15) / 16;
  struct T {
    int id, lo, hi;
  };
  std::vector<T> tiles(n_tiles);
  for (int i = 0; i < n_tiles; ++i) {
    int lo = K, hi = 0;
    for (int b = 8 * i; b < 8 * i + 8 && b < n_bins; ++b) {
      const int s0 = sup[2 * b], s1 = sup[2 * b + 1];
      if (s1 > s0) {
        lo = std::min(lo, s0);
        hi = std::max(hi, s1);
      }
    }
    if (hi <= lo) lo = hi = 0;
    tiles[i] = T{i, lo / 16, (hi + 15) / 16};
  }
  // longest first; the supports must nest in that order (centred kernels): the active tiles of a sub-stage are a prefix
  std::stable_sort(tiles.begin(), tiles.end(), [](const T &x, const T &y) { return (x.hi - x.lo) > (y.hi - y.lo); });
  for (int i = 0; i + 1 < n_tiles; ++i) {
    const T &x = tiles[i], &y = tiles[i + 1];
    if (y.hi > y.lo && (y.lo < x.lo || y.hi > x.hi)) return pl;
  }
  // row sets: consecutive runs of the sorted tiles.  One set when they fit; otherwise the first `split` tiles apart from the
  // rest (the long tiles' single-tile sub-stages are the cost of separating neighbours: keep the two longest together)
  int n_sets = (n_tiles + CH_NMAX - 1) / CH_NMAX;
  if (n_sets > CH_MAXSETS) return pl;
  std::vector<int> first(1, 0);
  const int split = env_int("MISPEC_CHAIN_SPLIT", 3);
  if (n_tiles > 4 && split > 0 && split < n_tiles && n_tiles - split <= CH_NMAX && split <= CH_NMAX) {
    first.push_back(split);
  } else {
    for (int i = 1; i < n_sets; ++i) first.push_back((int)((long long)i * n_tiles / n_sets));
  }
  first.push_back(n_tiles);
  pl.n_sets = (int)first.size() - 1;
  long long bricks = 0;
  for (int s = 0; s < pl.n_sets; ++s) {
    ChainSetHost &S = pl.set[s];
    S.n_tiles = first[s + 1] - first[s];
    if (S.n_tiles > CH_NMAX) return pl;
    for (int m = 0; m < S.n_tiles; ++m) {
      const T &t = tiles[first[s] + m];
      S.tile[m] = t.id;
      S.lo[m] = t.lo;
      S.hi[m] = t.hi;
    }
    S.brick0 = bricks;
    S.batch0 = (int)pl.batches.size();
    long long solo = 0;
    int s0 = S.lo[0];
    while (s0 < S.hi[0]) {
      int n = 0;
      while (n < S.n_tiles && S.lo[n] <= s0 && s0 < S.hi[n]) ++n;
      // the run of sub-stages with this prefix: until a tile enters or leaves
      int s1 = S.hi[0];
      for (int m = 0; m < S.n_tiles; ++m) {
        if (S.lo[m] > s0) s1 = std::min(s1, S.lo[m]);
        if (S.hi[m] > s0) s1 = std::min(s1, S.hi[m]);
      }
      if (n == 1) solo += s1 - s0;
      const int qmax = std::min(CH_QMAX, CH_NMAX / n);
      for (int sb = s0; sb < s1; sb += qmax) {
        const int q = std::min(qmax, s1 - sb);
        pl.batches.push_back((q << 8) | n);
        if (want_map)
          for (int j = 0; j < q; ++j)
            for (int m = 0; m < n; ++m) pl.brick_map.push_back(make_int2(S.tile[m], sb + j));
        bricks += (long long)q * n;
      }
      s0 = s1;
    }
    S.n_bricks = bricks - S.brick0;
    S.n_batches = (int)pl.batches.size() - S.batch0;
    S.cost = S.n_bricks + solo / 4;
  }
  // largest units first: the dispatcher's tail is then made of the short ones
  std::stable_sort(pl.set, pl.set + pl.n_sets, [](const ChainSetHost &x, const ChainSetHost &y) { return x.cost > y.cost; });
  pl.n_bricks = bricks;
  pl.batch_off = CH_ZERO_BYTES;
  pl.map_off = pl.batch_off + (((long long)pl.batches.size() * 4 + 255) & ~255LL);
  pl.brick_off = (pl.map_off + bricks * 8 + 1023) & ~1023LL;
  pl.bytes = pl.brick_off + bricks * CH_BRICK;
  pl.ok = true;
  return pl;
}

int ring_rows(int hop) { return 15 + (511 + hop) / hop; }  // NR hop >= 15 hop + 511: three batches of look-ahead + one block

bool chain_shape_ok(const mispec_framed_gemm_args *a) {
  if (a->precision != MISPEC_PREC_F32 || !a->basis_chain || !a->row_support || !a->row_support_host || !a->basis_im) return false;
  if (a->tile != MISPEC_TILE_AUTO || a->fb || a->out_frame_major) return false;
  if (a->hop % 64 || a->hop < 64 || a->hop > 512) return false;
  if ((long long)a->n_frames * a->hop + a->kernel + 65536 > 0x7fffffffLL) return false;
  if ((long long)a->n_samples * 2 > 0x7fffffffLL) return false;
  return true;
}

std::atomic<unsigned long long> g_configured{0};

}  // namespace

int64_t mispec_chain_bytes_impl(const int32_t *row_support_host, int32_t n_bins, int32_t kernel) {
  const ChainPlan pl = chain_plan(row_support_host, n_bins, kernel, false);
  return pl.ok ? pl.bytes : -1;
}

int mispec_chain_pack_impl(const float *basis_re, const float *basis_im, int64_t basis_row_stride, int32_t n_bins, int32_t kernel,
                           const int32_t *row_support_host, void *dst, int64_t dst_bytes, void *stream) {
  const ChainPlan pl = chain_plan(row_support_host, n_bins, kernel, true);
  if (!pl.ok) return mispec_fail_msg(MISPEC_E_UNSUPPORTED, "chain basis: the supports of the 16-row tiles do not nest (or too many tiles)");
  if (dst_bytes != pl.bytes) return mispec_fail_msg(MISPEC_E_INVALID, "chain basis: dst_bytes != mispec_basis_chain_bytes()");
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned char *d = static_cast<unsigned char *>(dst);
  // header: zeros | batch table | brick map (host-synchronous copies, once per bank)
  if (hipMemsetAsync(d, 0, (size_t)pl.brick_off, s) != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "hipMemsetAsync failed");
  if (hipStreamSynchronize(s) != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "hipStreamSynchronize failed");
  if (!pl.batches.empty() &&
      hipMemcpy(d + pl.batch_off, pl.batches.data(), pl.batches.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
    return mispec_fail_msg(MISPEC_E_HIP, "hipMemcpy failed");
  if (pl.n_bricks > 0) {
    if (hipMemcpy(d + pl.map_off, pl.brick_map.data(), pl.brick_map.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
      return mispec_fail_msg(MISPEC_E_HIP, "hipMemcpy failed");
    const unsigned grid = (unsigned)((pl.n_bricks + 3) / 4);
    hipLaunchKernelGGL(chain_pack_kernel, dim3(grid), dim3(256), 0, s, basis_re, basis_im, (long long)basis_row_stride, n_bins, kernel,
                       reinterpret_cast<const int2 *>(d + pl.map_off), pl.n_bricks, reinterpret_cast<float *>(d + pl.brick_off));
    if (hipGetLastError() != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "chain pack launch failed");
  }
  return MISPEC_OK;
}

// 1: the chain kernel serves this call; 0: not this route
int mispec_chain_ok(const mispec_framed_gemm_args *a) {
  if (!chain_shape_ok(a)) return 0;
  const ChainPlan pl = chain_plan(a->row_support_host, a->n_bins, a->kernel, false);
  return (pl.ok && pl.bytes == a->basis_chain_bytes) ? 1 : 0;
}

int mispec_chain_launch(const mispec_framed_gemm_args *a, int debug, void *stream) {
  const ChainPlan pl = chain_plan(a->row_support_host, a->n_bins, a->kernel, false);
  if (!pl.ok || pl.bytes != a->basis_chain_bytes) return mispec_fail_msg(MISPEC_E_INVALID, "chain basis does not match this bank");
  ChainArgs k;
  memset(&k, 0, sizeof(k));
  k.x = a->x;
  k.x_clip_stride = a->x_clip_stride;
  k.n_clips = a->n_clips;
  k.n_samples = a->n_samples;
  k.hop = a->hop;
  k.pad = a->pad;
  k.pad_mode = a->pad_mode;
  k.n_frames = a->n_frames;
  k.ct_per_clip = (a->n_frames + 15) / 16;
  const long long n_ct = (long long)k.ct_per_clip * a->n_clips;
  const long long n_groups = (n_ct + 3) / 4;
  if (n_groups * pl.n_sets > 0x7fffffffLL) return mispec_fail_msg(MISPEC_E_UNSUPPORTED, "grid too large");
  k.n_ct = (int)n_ct;
  k.nr = ring_rows(a->hop);
  k.ring_bytes = k.nr * (a->hop + CH_SKEW) * 4;
  const unsigned char *blob = static_cast<const unsigned char *>(a->basis_chain);
  k.zeros = reinterpret_cast<const float *>(blob);
  k.batches = reinterpret_cast<const int *>(blob + pl.batch_off);
  k.bricks = reinterpret_cast<const float *>(blob + pl.brick_off);
  k.row_scale = a->row_scale;
  k.n_bins = a->n_bins;
  k.epilogue = a->epilogue;
  k.im_sign = a->im_sign;
  k.eps = a->eps;
  k.power = a->power;
  k.out = a->out;
  k.out_clip_stride = a->out_clip_stride;
  k.out_row_stride = a->out_row_stride;
  k.out_row_offset = a->out_row_offset;
  k.n_sets = pl.n_sets;
  k.debug = debug;
  for (int s = 0; s < pl.n_sets; ++s) {
    const ChainSetHost &S = pl.set[s];
    k.first_wg[s] = (int)(s * n_groups);
    k.set[s].n_tiles = S.n_tiles;
    k.set[s].s_lo = S.lo[0];
    k.set[s].batch0 = S.batch0;
    k.set[s].n_batches = S.n_batches;
    k.set[s].brick0 = S.brick0;
    for (int m = 0; m < S.n_tiles; ++m) k.set[s].tile[m] = S.tile[m];
  }
  for (int s = pl.n_sets; s <= CH_MAXSETS; ++s) k.first_wg[s] = (int)(pl.n_sets * n_groups);
  const size_t smem = 4 * (size_t)k.ring_bytes + (size_t)CH_NBUF * CH_NMAX * CH_BRICK;
  if (smem > 160 * 1024) return mispec_fail_msg(MISPEC_E_UNSUPPORTED, "chain kernel: LDS budget");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "hipGetDevice failed");
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(g_configured.load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(cqt_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return mispec_fail_msg(MISPEC_E_HIP, "hipFuncSetAttribute failed (chain kernel)");
    g_configured.fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL(cqt_chain_kernel, dim3((unsigned)(n_groups * pl.n_sets)), dim3(256), smem, static_cast<hipStream_t>(stream), k);
  if (hipGetLastError() != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "chain kernel launch failed");
  return MISPEC_OK;
}
