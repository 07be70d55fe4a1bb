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
//     sub-stages later.  That delay line lives in LDS: a ring of rows of `hop` samples (row r = samples [U0 + r hop, + hop),
//     rows 8 dwords apart in bank space so that the 16 frames of a ds_read_b128 fall on different banks), refilled 64
//     samples at a time by global_load_lds_dword behind its first frame.  The waves of a workgroup whose column tiles
//     follow each other inside one clip (normally all four: 64 frames) share ONE ring of 63 + ceil((511 + hop) / hop) rows --
//     each sample enters LDS once per workgroup instead of once per frame and K stage; a workgroup across a clip boundary
//     splits into one ring per run of tiles (the LDS is the same: 16 rows per wave + the look-ahead).  Inside every aligned
//     group of 16 samples the DMA stores sample 4 n + q at position 4 q + n, so a lane's four taps of four successive MFMAs
//     are ONE ds_read_b128.  Virtual padding (reflect / zero) is resolved in the DMA's per-lane source address: no padded
//     copy, no edge workspace.
//   * A operand (basis).  Prepared once per bank (mispec_chain_basis_f32) as a stream of 1 KB "bricks" -- 16 rows x 16 taps
//     in MFMA fragment order [lane][4 taps of 4 MFMAs] -- in exactly the order a workgroup consumes them; taps outside a
//     16-row tile's support are not stored at all.  The four multiplying waves of a workgroup (4 column tiles) share the
//     bricks through three LDS buffers of 10 bricks filled by global_load_lds_dwordx4 two batches ahead; a brick is one
//     ds_read_b128 per lane and feeds 4 MFMAs.
//   * Work units.  The bank's 16-row tiles (8 bins; supports nested: CQT kernels are centred) are split into row sets; a
//     workgroup = 4 column tiles x one row set, all waves doing identical work per sub-stage.  Units are issued largest
//     first so that the dispatcher's tail is made of the short ones.
//   * Wave specialization.  512 threads: waves 0-3 multiply, waves 4-7 load (one of each per SIMD).  The multiplying wave's
//     stream is MFMAs with one LDS read behind each and one barrier per batch, nothing else: a batch = Q sub-stages x N
//     tiles (N <= 10, Q <= 8, inside one ring row) is compiled per (N, Q) without control flow, its LDS reads use immediate
//     offsets, the first fragments of the next batch are read behind the barrier.  The loading wave requests the bricks of
//     batch b + 2, refills the ring blocks batch b - 1 consumed, and states "everything older has landed" as a counted
//     `s_waitcnt vmcnt(n)` before the batch's barrier.
//
// Bounds: MFMA (fp32 matrix pipe, 157 TFLOP/s); LDS reads 1 KB per 4 MFMAs per wave; L2 -> LDS 1 KB per brick per workgroup;
// and, measured, the issue rate of LDS-DMA instructions beside an MFMA stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "mispec.h"
#include "mispec_internal.h"

// development ablations (results are WRONG with any of them): -DCH_ABL=bits
//   1 no barrier, 2 no vmcnt wait, 4 no ring refill, 8 no brick requests, 16 no MFMAs
#ifndef CH_ABL
#define CH_ABL 0
#endif
#ifndef CH_STAMP_WG
#define CH_STAMP_WG 2
#endif
#ifndef CH_PRIO
#define CH_PRIO 1  // 1: the multiplying waves at s_setprio 3 (0.905 ms); 2: the loading waves (0.932); 0: neither (0.910)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CH_NMAX = 10;     // 16-row tiles per row set = bricks per batch buffer
constexpr int CH_NBUF = 3;      // brick buffers: batch b multiplies one, b + 1 is landing (published by the barrier inside batch b's
                                // last sub-stage), b + 2 is being requested
constexpr int CH_QMAX = 8;      // sub-stages per batch (the ring's look-ahead: 3 batches + 1 sub-stage <= 27)
constexpr int CH_MAXSETS = 8;
constexpr int CH_SKEW = 8;      // dwords between ring rows in bank space
constexpr int CH_BRICK = 1024;  // bytes
constexpr int CH_ZERO_BYTES = 256;
constexpr int CH_HOPS = 8;      // segment tables for hop = 64, 128 .. 512
constexpr int CH_BUF_BYTES = CH_NMAX * CH_BRICK;

struct ChainSetDev {
  int n_tiles;
  int s_lo;        // first 16-tap sub-stage of the set (of its longest tile)
  int seg0;        // the set's segments in the table of this hop
  int n_segs;
  int boff0;       // ... and its batches' first bricks (relative to brick0; 4 entries of padding)
  int n_sub;       // sub-stages of the set
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
  const int *segs;  // this hop's table.  Segment words: N | Q << 4 | count << 8 = `count` batches of Q sub-stages x N tiles
  const float *bricks;
  long long n_bricks;
  const float *row_scale;
  int n_bins, epilogue;
  float im_sign, eps, power;
  float *out;
  long long out_clip_stride, out_row_stride;
  int out_row_offset;
  int n_sets;
  int debug;
  long long *stamps;  // (CH_ABL & 64) phase clock of workgroup 0
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
__device__ __forceinline__ void ch_dma4s(const void *sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// ... the first LANES lanes only (EXEC is all ones in this kernel's main loop: set and restored here, no branch)
template <int LANES>
__device__ __forceinline__ void ch_dma4s_lanes(const void *sbase, unsigned voff, unsigned lds_addr) {
  static_assert(LANES > 0 && LANES < 64, "a partial wave");
  constexpr unsigned lo = LANES >= 32 ? 0xffffffffu : ((1u << LANES) - 1u);
  constexpr unsigned hi = LANES > 32 ? ((1u << (LANES - 32)) - 1u) : 0u;
  asm volatile(
      "s_mov_b32 exec_lo, %3\n\ts_mov_b32 exec_hi, %4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\ts_mov_b64 exec, -1"
      :
      : "v"(voff), "s"(sbase), "s"(lds_addr), "i"(lo), "i"(hi)
      : "memory", "m0");
}
template <int LANES>
__device__ __forceinline__ void ch_dma4_lanes(const void *src, unsigned lds_addr) {
  static_assert(LANES > 0 && LANES < 64, "a partial wave");
  constexpr unsigned lo = LANES >= 32 ? 0xffffffffu : ((1u << LANES) - 1u);
  constexpr unsigned hi = LANES > 32 ? ((1u << (LANES - 32)) - 1u) : 0u;
  asm volatile(
      "s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %3\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off\n\ts_mov_b64 exec, -1"
      :
      : "v"(src), "s"(lds_addr), "i"(lo), "i"(hi)
      : "memory", "m0");
}
// ... the lanes of `mask` only (EXEC is all ones in the loading waves' loop: set and restored here, no branch)
__device__ __forceinline__ void ch_dma4s_mask(const void *sbase, unsigned voff, unsigned lds_addr, unsigned long long mask) {
  asm volatile("s_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1\n\ts_mov_b64 exec, -1"
               :
               : "v"(voff), "s"(sbase), "s"(lds_addr), "s"(mask)
               : "memory", "m0");
}
// three 16-byte pieces per lane from one scalar base: lane offsets v0..v2, LDS addresses d0..d2
__device__ __forceinline__ void ch_dma16x3(const void *sbase, unsigned v0, unsigned v1, unsigned v2, unsigned d0, unsigned d1, unsigned d2) {
  asm volatile(
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
      :
      : "v"(v0), "v"(v1), "v"(v2), "s"(sbase), "s"(d0), "s"(d1), "s"(d2)
      : "memory", "m0");
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

typedef const int __attribute__((address_space(4))) *ch_seg_ptr;  // constant address space: scalar loads
typedef const volatile f32x4 __attribute__((address_space(3))) *ch_lds_pinned;  // an LDS read the compiler leaves where it is written
typedef const f32x4 __attribute__((address_space(3))) *ch_lds;

// ------------------------------------------------------------------------------------------------------------
// Workgroup = 4 multiplying waves (wave w: column tile w, all the set's tiles) + 4 loading waves (wave 4 + w: the ring of
// column tile w, a quarter of the bricks).  Waves w and 4 + w share a SIMD (dispatch order; only speed depends on it): the
// loader's address arithmetic and DMA issue fill the issue slots the MFMA stream leaves, and the MFMA stream itself is
// LDS reads + MFMAs + one barrier per batch, nothing else.
//
// Batch b (Q sub-stages x N tiles; bricks in buffer b % 3), between barrier(b - 1) and barrier(b):
//   multiplying wave: sub-stage j's 4 N MFMAs under the LDS reads of sub-stage j + 1; in the last sub-stage, after half
//     of its MFMAs, barrier(b) (all of the wave's reads of batch b are in registers) and the reads of batch b + 1's first
//     fragments (published by that barrier) under the other half.
//   loading wave: requests its bricks of batch b + 2 into buffer (b + 2) % 3 (last read in batch b - 1: free since
//     barrier(b - 1)); refills what batch b - 1 read of frame 0's ring row (same place, the samples one ring further on:
//     first needed >= 30 sub-stages later); waits until everything it requested in earlier batches has landed
//     (s_waitcnt vmcnt(this batch's requests)): the bricks of batch b + 1; barrier(b) publishes them.
// ------------------------------------------------------------------------------------------------------------

// `count` batches of Q sub-stages x N tiles on a multiplying wave.  In / out: bfirst = the signal fragment of the next
// sub-stage (always read ahead).
//
// The instruction stream is written out slot by slot: every MFMA is followed by AT MOST one LDS read or one small piece of
// address arithmetic, and a scheduling fence keeps it there.  Measured (experiments/chain/mfma_rate.hip): fragment reads
// issued as a burst in front of a sub-stage's MFMAs cost 2 .. 4.5 cycles per MFMA, address arithmetic in front of a read
// another 5 .. 10; one ds_read_b128 with an immediate offset behind each MFMA costs 0.3 .. 1 (32.3 .. 33 cycles per MFMA).
struct MulState {
  unsigned a_lane;    // (lane) LDS address of brick slot 0 of buffer 0 + 16 lane
  int buf;            // buffer of the current batch
  int off, hop, nr, row_bytes;
  int slot;           // (lane) the lane's ring row slot
  unsigned a_cur;     // (lane) this batch's bricks
  unsigned b_cur;     // (lane) this batch's first signal fragment
  unsigned row_next;  // (lane) the lane's next ring row (+ 16 lq)
};

template <int N, int Q>
__device__ __forceinline__ void mul_segment(f32x4 (&acc)[CH_NMAX], f32x4 &bfirst, MulState &c, const int count) {
  // The barrier's slot in the batch's last sub-stage: the N + 1 reads behind it want >= 4 MFMAs of cover before the next
  // batch's first MFMA needs them (N = 1: the barrier opens the sub-stage and 3 MFMAs are all there is)
  constexpr int TB = 4 * N - 5 < 0 ? 0 : (2 * N < 4 * N - 5 ? 2 * N : 4 * N - 5);
  // slots of the first sub-stage that carry the address arithmetic (behind its reads)
  // (a single sub-stage: in front of it -- its barrier slot may be the first)
  constexpr int TK0 = Q == 1 ? -1 : N + 1, TK1 = Q == 1 ? -1 : (TK0 + 1 < 4 * N ? TK0 + 1 : TK0);
  f32x4 af[N], bf = bfirst;
#pragma unroll
  for (int m = 0; m < N; ++m) af[m] = *(ch_lds_pinned)(c.a_cur + m * CH_BRICK);
  for (int i = 0; i < count; ++i) {
    f32x4 a[2][N], bb[2];
#pragma unroll
    for (int m = 0; m < N; ++m) a[0][m] = af[m];
    bb[0] = bf;
    unsigned a_nxt = 0, b_nxt = 0;
    bool row_end = false;
    if (Q == 1) {
      const int buf1 = c.buf + 1 == CH_NBUF ? 0 : c.buf + 1;
      a_nxt = c.a_lane + (unsigned)(buf1 * CH_BUF_BYTES);
      c.buf = buf1;
      const int off2 = c.off + 16;
      row_end = off2 == c.hop;
      c.off = row_end ? 0 : off2;
      b_nxt = row_end ? c.row_next : c.b_cur + 64u;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
      const int cu = j & 1, nx = cu ^ 1;
#pragma unroll
      for (int t = 0; t < 4 * N; ++t) {
        const int jj = t / N, m = t % N;
        if (j == Q - 1 && t == TB) {
          // every read of this batch is in registers (the fragments of its last sub-stage are waited for here at the
          // latest): the loading waves may overwrite its bricks and what it read of the ring; the next batch's bricks are
          // published
          if (!(CH_ABL & 1)) __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
        }
        if (CH_ABL & 16)
          acc[m][0] += a[cu][m][jj] * bb[cu][jj];
        else
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cu][m][jj], bb[cu][jj], acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // ---- the slot behind this MFMA
        if (j + 1 < Q) {
          if (t == 0) bb[nx] = *(ch_lds_pinned)(c.b_cur + 64 * (j + 1));
          if (t >= 1 && t <= N) a[nx][t - 1] = *(ch_lds_pinned)(c.a_cur + ((j + 1) * N + t - 1) * CH_BRICK);
        } else {
          if (t == TB) bb[nx] = *(ch_lds_pinned)(b_nxt);
          if (t > TB && t <= TB + N) a[nx][t - TB - 1] = *(ch_lds_pinned)(a_nxt + (t - TB - 1) * CH_BRICK);
        }
        if (j == 0 && t == TK0) {
          // where the next batch's bricks are: the next buffer
          const int buf1 = c.buf + 1 == CH_NBUF ? 0 : c.buf + 1;
          a_nxt = c.a_lane + (unsigned)(buf1 * CH_BUF_BYTES);
          c.buf = buf1;
        }
        if (j == 0 && t == TK1) {
          // ... and its first signal fragment: further along the ring row or at the start of the next one
          const int off2 = c.off + 16 * Q;
          row_end = off2 == c.hop;  // (batches never straddle ring rows)
          c.off = row_end ? 0 : off2;
          b_nxt = row_end ? c.row_next : c.b_cur + 64u * Q;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int m = 0; m < N; ++m) af[m] = a[Q & 1][m];
    bf = bb[Q & 1];
    c.a_cur = a_nxt;
    c.b_cur = b_nxt;
    if (row_end) {  // (rare: every hop / 16 sub-stages) the lane moves on to its next ring row
      const bool wrap = c.slot + 1 == c.nr;
      c.slot = wrap ? 0 : c.slot + 1;
      const bool wrap2 = c.slot + 1 == c.nr;
      c.row_next = c.row_next + (wrap2 ? (unsigned)(-(c.nr - 1) * c.row_bytes) : (unsigned)c.row_bytes);
    }
  }
  bfirst = bf;
}

// 64 (or `lanes` < 64) samples from ring sample v0 on, to LDS address dst: inside each group of 16 the lane at position
// 4 q + n fetches sample 4 n + q
template <bool REFLECT>
__device__ __forceinline__ void ch_fill(const float *xc, const float *zeros, int L, int U0, int lane, int lperm, int v0, unsigned dst, int lanes) {
  int pos = U0 + v0 + lperm;
  if (lane < lanes) {
    if (REFLECT) {
      // beyond the virtually padded clip only zero taps / unstored frames read: any finite sample will do
      pos = pos < 0 ? -pos : pos;
      pos = pos >= L ? 2 * L - 2 - pos : pos;
      pos = pos < 0 ? 0 : (pos > L - 1 ? L - 1 : pos);
      ch_dma4s(xc, (unsigned)pos * 4u, dst);
    } else {
      // zero / no padding: everything outside the clip is zero
      const bool in = pos >= 0 && pos < L;
      const float *src = in ? xc + pos : zeros + lane;
      ch_dma4(src, dst);
    }
  }
}

// What a loading (or, in the prologue, multiplying) wave knows about its clip
struct FillCtx {
  const float *xc;
  const float *zeros;
  int L, pad, U0, lane, lperm;
  unsigned lperm4, nlperm4, lane4;
};

// `lanes` (1 .. 64, wave-uniform) samples from ring sample v0 on, to LDS address dst (inside each group of 16 the lane at
// position 4 q + n fetches sample 4 n + q).  Returns the DMA instructions issued (0: nothing but unstored frames reads
// this block).  A block that lies entirely inside the clip, or entirely inside one of its mirror images / its zero padding,
// is one DMA with one vector add in front of it; only the blocks across a boundary take the per-lane arithmetic.
template <bool REFLECT>
__device__ __forceinline__ int ch_fill_block(const FillCtx &c, int v0, unsigned dst, int lanes) {
  dst = __builtin_amdgcn_readfirstlane(dst);  // (wave-uniform by construction; the compiler does not always see it)
  const int p0 = __builtin_amdgcn_readfirstlane(c.U0 + v0), p1 = p0 + lanes;  // signal positions [p0, p1)
  if (p0 >= c.L + c.pad) return 0;            // beyond the virtually padded clip: frames that are not stored
  const unsigned long long mask = lanes >= 64 ? ~0ull : ((1ull << lanes) - 1ull);
  if (p0 >= 0 && p1 <= c.L) {
    ch_dma4s_mask(c.xc, c.lperm4 + 4u * (unsigned)p0, dst, mask);
  } else if (REFLECT && p1 <= 0) {  // left mirror image: position p <- sample -p
    ch_dma4s_mask(c.xc, c.nlperm4 + 4u * (unsigned)(-p0), dst, mask);
  } else if (REFLECT && p0 >= c.L && p1 <= 2 * c.L - 1) {  // right mirror image: p <- 2 L - 2 - p
    ch_dma4s_mask(c.xc, c.nlperm4 + 4u * (unsigned)(2 * c.L - 2 - p0), dst, mask);
  } else if (!REFLECT && (p1 <= 0 || p0 >= c.L)) {
    ch_dma4s_mask(c.zeros, c.lane4, dst, mask);
  } else {
    ch_fill<REFLECT>(c.xc, c.zeros, c.L, c.U0, c.lane, c.lperm, v0, dst, lanes);
  }
  return 1;
}

// every `step`-th 64-sample block of a ring, from block `first` on
template <bool REFLECT>
__device__ __forceinline__ void ch_fill_ring(const FillCtx &c, unsigned ring_base, int ring_samples, int hop, int row_bytes, int first, int step) {
  for (int v0 = 64 * first; v0 < ring_samples; v0 += 64 * step) {
    const int slot = v0 / hop, off = v0 - slot * hop;
    ch_fill_block<REFLECT>(c, v0, ring_base + (unsigned)(slot * row_bytes + off * 4), 64);
  }
}

// The same through registers, for the prologue of a ring whose rows are multiples of 256 samples: a block of 256 samples
// inside the clip is ONE global_load_dwordx4 per lane (lane i: samples 4 i .. 4 i + 3 = group i >> 2, n = i & 3, q = 0 .. 3)
// and four ds_write_b32 to positions 16 g + 4 q + n -- a quarter of the address work of four LDS-DMA dwords per lane, which
// is what bounds the DMA fill (4 lanes per clock and CU: experiments/chain/dma_rate.hip).  Blocks across the clip's ends
// (or in the padding) take the DMA path.  Every `step`-th 256-block from block `first` on (16 of them cover the ring when
// `step` waves share the work); all loads are in flight at once.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float __attribute__((address_space(3))) *ch_lds_f32;
template <bool REFLECT>
__device__ __forceinline__ void ch_fill_ring_wide(const FillCtx &c, unsigned ring_base, int ring_samples, int hop, int row_bytes, int first, int step) {
  constexpr int NB = 16;  // blocks per wave: the ring has at most 17 x 256 (hop 256) or 16 x 512 (hop 512) samples
  const int nblk = ring_samples >> 8;
  const int sh = hop >> 9;  // 256-blocks per row: 1 << sh
  f32x4 v[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int p0 = c.U0 + 256 * (first + step * j);
    const int pc = p0 < 0 ? 0 : (p0 > c.L - 256 ? c.L - 256 : p0);  // (a block that is not inside the clip loads something valid)
    v[j] = *(const f32x4_u *)(c.xc + pc + 4 * c.lane);
  }
  const unsigned lane_off = 64u * (unsigned)(c.lane >> 2) + 4u * (unsigned)(c.lane & 3);
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int k = first + step * j;
    const int p0 = c.U0 + 256 * k;
    if (k < nblk && p0 >= 0 && p0 + 256 <= c.L) {
      const unsigned dst = ring_base + (unsigned)((k >> sh) * row_bytes + ((k & ((1 << sh) - 1)) << 10)) + lane_off;
#pragma unroll
      for (int q = 0; q < 4; ++q) *(ch_lds_f32)(dst + 16u * q) = v[j][q];
    }
  }
#pragma unroll 1
  for (int k = first; k < nblk; k += step) {
    const int p0 = c.U0 + 256 * k;
    if (p0 >= 0 && p0 + 256 <= c.L) continue;
    const unsigned dst = ring_base + (unsigned)((k >> sh) * row_bytes + ((k & ((1 << sh) - 1)) << 10));
#pragma unroll 1
    for (int b = 0; b < 4; ++b) ch_fill_block<REFLECT>(c, 256 * k + 64 * b, dst + 256u * b, 64);
  }
}

__device__ __forceinline__ void ch_wait_vmcnt(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;  // (never more than 3 + 2 + 2; a smaller count only waits longer)
  }
}

template <bool REFLECT>
__global__ void __launch_bounds__(512) cqt_chain_kernel(const ChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave & 3;  // column tile of the workgroup
  const int f = lane & 15, lq = lane >> 4;

  int s = 0;
  while (s + 1 < a.n_sets && (int)blockIdx.x >= a.first_wg[s + 1]) ++s;
  const ChainSetDev &S = a.set[s];
  const int g = (int)blockIdx.x - a.first_wg[s];
  const int ct_raw = 4 * g + cw;
  const bool live = ct_raw < a.n_ct;
  const int ct = live ? ct_raw : a.n_ct - 1;  // idle waves of the last group shadow the last column tile (nothing stored)
  const int clip = __builtin_amdgcn_readfirstlane(ct / a.ct_per_clip);  // (the division runs on the vector unit)
  const int tic = ct - clip * a.ct_per_clip;  // the wave's column tile inside its clip
  const int t0 = tic * 16;
  // Waves whose column tiles follow each other inside one clip share ONE delay line: frame 16 (w + 1) reads now what frame
  // 16 w + 15 reads a row later, so the rings of a run of `nw` waves are one ring of 16 (nw - 1) more rows, filled once
  // behind the run's first frame.  (A group across a clip boundary, or the idle waves of the last group, are runs of their
  // own: nw = 1 is a ring per wave.)
  const int lead = live ? cw - (cw < tic ? cw : tic) : cw;  // first wave of the run
  int last = live ? cw + (a.ct_per_clip - 1 - tic) : cw;
  last = last > 3 ? 3 : last;
  if (live && 4 * g + last > a.n_ct - 1) last = a.n_ct - 1 - 4 * g;
  const int nw = last - lead + 1, widx = cw - lead;
  const int hop = a.hop, nr = a.nr + 16 * (nw - 1);
  const int row_bytes = (hop + CH_SKEW) * 4;
  const unsigned ring_base = (unsigned)(lead * a.ring_bytes);  // (a run owns the LDS of its waves' rings: nw a.nr >= nr rows)
  const int U0 = (t0 - 16 * widx) * hop + 16 * S.s_lo - a.pad;   // signal position of ring sample v = 0: the run's first frame
  const unsigned a_base = (unsigned)(4 * a.ring_bytes);
  const ch_seg_ptr seg = (ch_seg_ptr)(a.segs + S.seg0);
  const int n_segs = S.n_segs;

  if ((CH_ABL & 64) && blockIdx.x == CH_STAMP_WG && wave == 0 && lane == 0) a.stamps[0] = clock64();
  if (wave >= 4) {
    // ------------------------------------------------------------------ loading wave
#if CH_PRIO == 2
    __builtin_amdgcn_s_setprio(3);  // (its batch of address arithmetic and requests must be over before its SIMD partner's batch of MFMAs)
#endif
    FillCtx fc;
    fc.xc = a.x + (long long)clip * a.x_clip_stride;
    fc.zeros = a.zeros;
    fc.L = a.n_samples;
    fc.pad = a.pad;
    fc.U0 = U0;
    fc.lane = lane;
    fc.lperm = 16 * (lane >> 4) + 4 * (lane & 3) + ((lane >> 2) & 3);
    fc.lperm4 = 4u * (unsigned)fc.lperm;
    fc.nlperm4 = 0u - fc.lperm4;
    fc.lane4 = 4u * (unsigned)lane;
    const int ring_samples = nr * hop;
    const float *bricks = a.bricks + S.brick0 * (CH_BRICK / 4);
    // its half of the ring (the even 64-sample blocks; the multiplying wave, idle until the first barrier, takes the odd ones)
    // (the 2 nw waves of the run take every 2 nw-th block of its ring)
    if ((hop & 255) == 0 && fc.L >= 256)
      ch_fill_ring_wide<REFLECT>(fc, ring_base, ring_samples, hop, row_bytes, 2 * widx, 2 * nw);
    else
      ch_fill_ring<REFLECT>(fc, ring_base, ring_samples, hop, row_bytes, 2 * widx, 2 * nw);
    // request cursor over the batches: segment, batches left in it, first brick
    int rsg = 0, rword = n_segs > 0 ? seg[0] : 0, rrem = rword >> 8;
    long long rbrick = 0;
    int rbuf = 0;
    auto request = [&]() -> int {  // this wave's bricks of the next batch to request; returns the DMAs issued
      if (rsg >= n_segs) return 0;
      const int cnt = (rword & 15) * ((rword >> 4) & 15);
      int issued = 0;
      if (!(CH_ABL & 8))
        for (int k = cw; k < cnt; k += 4) {
          ch_dma16(bricks + (rbrick + k) * (CH_BRICK / 4), (unsigned)lane * 16u, a_base + (unsigned)(rbuf * CH_BUF_BYTES + k * CH_BRICK));
          ++issued;
        }
      rbrick += cnt;
      rbuf = rbuf + 1 == CH_NBUF ? 0 : rbuf + 1;
      if (--rrem == 0) {
        ++rsg;
        rword = rsg < n_segs ? seg[rsg] : 0;
        rrem = rword >> 8;
      }
      return issued;
    };
    if ((CH_ABL & 64) && blockIdx.x == CH_STAMP_WG && wave == 4 && lane == 0) a.stamps[1] = clock64();
    request();
    request();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((CH_ABL & 64) && blockIdx.x == CH_STAMP_WG && wave == 4 && lane == 0) a.stamps[2] = clock64();
    __syncthreads();
    int slot0 = 0, off0 = 0, sdone = 0;
    // Ring refill: the part of frame 0's row that the batches up to b - 1 have read and that is not requested yet = `pend`
    // groups of 16 samples from LDS address pdst on, to be replaced by the ring samples from pv0 on.  Whole 64-sample blocks
    // only (one DMA instruction each: it is the NUMBER of DMA instructions a loading wave issues beside the MFMA stream that
    // is scarce, 3.14); what is left of a row when frame 0 leaves it goes as one partial block.
    int pend = 0;
    unsigned pdst = 0;
    int pv0 = 0;
    bool prow_end = false;
    int turn = 0;  // the run's loading waves take its blocks in turn
    long long st_issue = 0, st_wait = 0, st_barrier = 0, st_n = 0, st_prev = (CH_ABL & 64) ? clock64() : 0;
    for (int sg = 0; sg < n_segs; ++sg) {
      const int w = seg[sg];
      const int q = (w >> 4) & 15, count = w >> 8;
      for (int i = 0; i < count; ++i) {
        // Between barrier(b - 1) and barrier(b).  First the ring: what the batches up to b - 1 read of frame 0's row is dead
        // since barrier(b - 1); its replacement is first read >= 32 sub-stages after the group it replaces.  A group waits for
        // at most 3 later ones (<= 13 sub-stages with the batch that brings them), its block is requested in the iteration
        // after, waited for in the next, readable from the batch after that: <= 13 + 8 + 8 sub-stages.
        // (Bricks first and the ring blocks allowed another batch in flight -- s_waitcnt vmcnt(+ the previous iteration's
        // ring blocks) -- ran the N = 1 phases at 52 instead of 38 cycles per MFMA.)
        int issued = 0;
        if (!(CH_ABL & 4)) {
          while (pend >= 4) {
            if (turn == widx) issued += ch_fill_block<REFLECT>(fc, pv0, pdst, 64);
            turn = turn + 1 == nw ? 0 : turn + 1;
            pv0 += 64;
            pdst += 256u;
            pend -= 4;
          }
          if (prow_end && pend > 0) {
            if (turn == widx) issued += ch_fill_block<REFLECT>(fc, pv0, pdst, 16 * pend);
            turn = turn + 1 == nw ? 0 : turn + 1;
            pend = 0;
          }
        } else {
          pend = 0;
        }
        if (pend == 0) {
          pdst = ring_base + (unsigned)(slot0 * row_bytes + off0 * 4);
          pv0 = 16 * sdone + ring_samples;
        }
        pend += q;  // (batch b's part: dead at barrier(b), i.e. from the next iteration on)
        sdone += q;
        off0 += 16 * q;
        prow_end = off0 == hop;
        if (prow_end) {
          off0 = 0;
          slot0 = slot0 + 1 == nr ? 0 : slot0 + 1;
        }
        issued += request();  // batch b + 2's bricks
        // everything requested in earlier batches has landed: batch b + 1's bricks, the ring samples of batch b - 2's part
        long long tw0 = 0, tw1 = 0;
        if (CH_ABL & 64) tw0 = clock64();
        if (!(CH_ABL & 2)) ch_wait_vmcnt(issued);
        if (CH_ABL & 64) tw1 = clock64();
        if (!(CH_ABL & 1)) __syncthreads();
        if (CH_ABL & 64) {  // where a loading wave's iteration goes: issue, wait for the landing, barrier (sums; a.stamps[1000 ..])
          const long long tw2 = clock64();
          st_issue += tw0 - st_prev;
          st_wait += tw1 - tw0;
          st_barrier += tw2 - tw1;
          st_prev = tw2;
          ++st_n;
        }
      }
    }
    if ((CH_ABL & 64) && blockIdx.x == CH_STAMP_WG && lane == 0) {
      a.stamps[1000 + 4 * cw] = st_issue;
      a.stamps[1001 + 4 * cw] = st_wait;
      a.stamps[1002 + 4 * cw] = st_barrier;
      a.stamps[1003 + 4 * cw] = st_n;
    }
    return;
  }

  // -------------------------------------------------------------------- multiplying wave
  MulState c;
  c.a_lane = a_base + (unsigned)lane * 16u;
  c.buf = 0;
  c.off = 0;
  c.hop = hop;
  c.nr = nr;
  c.row_bytes = row_bytes;
  c.slot = 16 * widx + f;  // the lane's frame in the run
  c.a_cur = c.a_lane;
  c.b_cur = ring_base + (unsigned)(c.slot * row_bytes) + (unsigned)lq * 16u;
  c.row_next = c.b_cur + (c.slot + 1 == nr ? (unsigned)(-(nr - 1) * row_bytes) : (unsigned)row_bytes);
#if CH_PRIO == 1
  __builtin_amdgcn_s_setprio(3);  // (the MFMA stream before its SIMD partner's address arithmetic)
#endif
  {
    FillCtx fc;
    fc.xc = a.x + (long long)clip * a.x_clip_stride;
    fc.zeros = a.zeros;
    fc.L = a.n_samples;
    fc.pad = a.pad;
    fc.U0 = U0;
    fc.lane = lane;
    fc.lperm = 16 * (lane >> 4) + 4 * (lane & 3) + ((lane >> 2) & 3);
    fc.lperm4 = 4u * (unsigned)fc.lperm;
    fc.nlperm4 = 0u - fc.lperm4;
    fc.lane4 = 4u * (unsigned)lane;
    if ((hop & 255) == 0 && fc.L >= 256)
      ch_fill_ring_wide<REFLECT>(fc, ring_base, nr * hop, hop, row_bytes, 2 * widx + 1, 2 * nw);
    else
      ch_fill_ring<REFLECT>(fc, ring_base, nr * hop, hop, row_bytes, 2 * widx + 1, 2 * nw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();  // the ring and the first two batches are in LDS

  f32x4 acc[CH_NMAX];
#pragma unroll
  for (int m = 0; m < CH_NMAX; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bfirst = *(ch_lds_pinned)(c.b_cur);  // sub-stage 0's signal fragment

  if ((CH_ABL & 64) && blockIdx.x == CH_STAMP_WG && wave == 0 && lane == 0) a.stamps[3] = clock64();
  int w_next = n_segs > 0 ? seg[0] : 0;
  for (int sg = 0; sg < n_segs; ++sg) {
    const int w = w_next;
    w_next = seg[sg + 1 < n_segs ? sg + 1 : sg];  // (requested a segment ahead: the scalar load's latency is off the path)
    const int count = w >> 8;
    if ((CH_ABL & 64) && blockIdx.x == CH_STAMP_WG && wave == 0 && lane == 0 && sg < 200) {
      a.stamps[8 + 2 * sg] = clock64();
      a.stamps[9 + 2 * sg] = w;
    }
    switch (w & 255) {
#define CH_CASE(N, Q)                                     \
  case (N) | ((Q) << 4):                                  \
    mul_segment<N, Q>(acc, bfirst, c, count); \
    break;
      CH_CASE(1, 1) CH_CASE(1, 2) CH_CASE(1, 3) CH_CASE(1, 4) CH_CASE(1, 5) CH_CASE(1, 6) CH_CASE(1, 7) CH_CASE(1, 8)
      CH_CASE(2, 1) CH_CASE(2, 2) CH_CASE(2, 3) CH_CASE(2, 4) CH_CASE(2, 5)
      CH_CASE(3, 1) CH_CASE(3, 2) CH_CASE(3, 3)
      CH_CASE(4, 1) CH_CASE(4, 2) CH_CASE(5, 1) CH_CASE(5, 2)
      CH_CASE(6, 1) CH_CASE(7, 1) CH_CASE(8, 1) CH_CASE(9, 1) CH_CASE(10, 1)
#undef CH_CASE
      default:
        break;
    }
  }

  if ((CH_ABL & 64) && blockIdx.x == CH_STAMP_WG && wave == 0 && lane == 0) {
    a.stamps[4] = clock64();
    a.stamps[5] = n_segs;
  }
  // ---- epilogue: element e of lane (f, lq) of tile m is D[row 16 tile + 4 lq + e][frame t0 + f]: (re, im) of bins
  // 8 tile + 2 lq and + 1 sit in one lane
  const int t = t0 + f;
  if (!live || t >= a.n_frames) return;
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
// host: the plan (tiles, row sets, segments per hop, brick order) from the supports
// ------------------------------------------------------------------------------------------------------------
struct ChainSetHost {
  int n_tiles = 0;
  int tile[CH_NMAX];
  int lo[CH_NMAX], hi[CH_NMAX];  // sub-stages
  long long brick0 = 0, n_bricks = 0;
  int seg0[CH_HOPS], n_segs[CH_HOPS], boff0[CH_HOPS];  // per hop: first segment (index into the hop's table), segments, first batch offset
  long long cost = 0;  // bricks + a quarter of the single-tile sub-stages (dependent MFMAs issue at 40 / 32 cycles)
};

struct ChainPlan {
  bool ok = false;
  int n_sets = 0;
  ChainSetHost set[CH_MAXSETS];
  std::vector<int> segs[CH_HOPS];  // hop = 64 (h + 1)
  std::vector<int2> brick_map;
  long long n_bricks = 0;
  long long seg_off[CH_HOPS], map_off = 0, brick_off = 0, bytes = 0;
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
  // rest (separating neighbours costs the longer one's single-tile sub-stages: keep the longest tiles together)
  int n_sets = (n_tiles + CH_NMAX - 1) / CH_NMAX;
  if (n_sets > CH_MAXSETS) return pl;
  std::vector<int> first(1, 0);
  const int split = env_int("MISPEC_CHAIN_SPLIT", 4);
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
    std::vector<int> boffs[CH_HOPS];
    for (int h = 0; h < CH_HOPS; ++h) S.seg0[h] = (int)pl.segs[h].size();
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
      // the bricks of the run, sub-stage major (whatever the batches: their order does not depend on the hop)
      if (want_map)
        for (int sb = s0; sb < s1; ++sb)
          for (int m = 0; m < n; ++m) pl.brick_map.push_back(make_int2(S.tile[m], sb));
      // batches of the run, per hop: Q <= 7 / n sub-stages, never across a ring row (hop / 16 sub-stages from the set's start)
      // Q: what the loading waves can keep up with.  A loading wave issues ceil(Q / 4) ring blocks + ceil(n Q / 4) bricks per
      // batch at ~300 cycles each beside the MFMA stream (3.14's DMA-issue wall) against 128 n Q cycles of MFMAs:
      // n = 2, Q = 5 is (2 + 3) 300 / 1280 = 1.17 (measured 42 cycles per MFMA), Q = 4 is (1 + 2) 300 / 1024 = 0.88
      const int qmax = n == 2 ? env_int("MISPEC_CHAIN_Q2", 4) : std::min(CH_QMAX, CH_NMAX / n);
      for (int h = 0; h < CH_HOPS; ++h) {
        const int per_row = 4 * (h + 1);
        std::vector<int> &sg = pl.segs[h];
        int sb = s0;
        while (sb < s1) {
          const int row_end = S.lo[0] + ((sb - S.lo[0]) / per_row + 1) * per_row;
          const int e = std::min(s1, row_end);
          // as few batches as qmax allows, of equal length +- 1: a short remainder batch would still cost the loading
          // waves a full round of requests (measured: 1.5 .. 2 k cycles for 8 .. 16 MFMAs)
          const int len = e - sb, nb = (len + qmax - 1) / qmax;
          const int q_lo = len / nb, n_big = len % nb;
          auto push = [&](int q, int cnt) {
            if (cnt <= 0) return;
            for (int t = 0; t < cnt; ++t) boffs[h].push_back((int)(bricks - S.brick0) + ((sb - s0) + t * q) * n);
            sb += cnt * q;
            const int key = n | (q << 4);
            if ((int)sg.size() > S.seg0[h] && (sg.back() & 255) == key && (sg.back() >> 8) + cnt < (1 << 22))
              sg.back() += cnt << 8;
            else
              sg.push_back(key | (cnt << 8));
          };
          push(q_lo + 1, n_big);
          push(q_lo, nb - n_big);
        }
      }
      bricks += (long long)(s1 - s0) * n;
      s0 = s1;
    }
    for (int h = 0; h < CH_HOPS; ++h) {
      S.n_segs[h] = (int)pl.segs[h].size() - S.seg0[h];
      S.boff0[h] = (int)pl.segs[h].size();
      pl.segs[h].insert(pl.segs[h].end(), boffs[h].begin(), boffs[h].end());
      for (int t = 0; t < 4; ++t) pl.segs[h].push_back((int)(bricks - S.brick0));  // (requests past the last batch: the stream's padding)
    }
    S.n_bricks = bricks - S.brick0;
    S.cost = S.n_bricks + solo / 4;
  }
  // largest units first: the dispatcher's tail is then made of the short ones
  std::stable_sort(pl.set, pl.set + pl.n_sets, [](const ChainSetHost &x, const ChainSetHost &y) { return x.cost > y.cost; });
  pl.n_bricks = bricks;
  long long off = CH_ZERO_BYTES;
  for (int h = 0; h < CH_HOPS; ++h) {
    pl.seg_off[h] = off;
    off += ((long long)pl.segs[h].size() * 4 + 255) & ~255LL;
  }
  pl.map_off = off;
  pl.brick_off = (pl.map_off + bricks * 8 + 1023) & ~1023LL;
  pl.bytes = pl.brick_off + (bricks + 12) * CH_BRICK;  // (the request side runs up to 12 bricks past the end)
  pl.ok = true;
  return pl;
}

int ring_rows(int hop) { return 15 + (511 + hop) / hop; }  // NR hop >= 15 hop + 511: three batches of look-ahead + one block

bool chain_shape_ok(const mispec_framed_gemm_args *a) {
  if (a->precision != MISPEC_PREC_F32 || !a->basis_chain || !a->row_support || !a->row_support_host || !a->basis_im) return false;
  if (a->tile != MISPEC_TILE_AUTO || a->fb || a->out_frame_major) return false;
  if (a->hop % 64 || a->hop < 64 || a->hop > 64 * CH_HOPS) return false;
  if ((long long)a->n_frames * a->hop + a->kernel + 65536 > 0x7fffffffLL) return false;
  if ((long long)a->n_samples * 4 > 0x7fffffffLL) return false;
  return true;
}

std::atomic<unsigned long long> g_configured[2];

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
  // header: zeros | segment tables | brick map (host-synchronous copies, once per bank)
  if (hipMemsetAsync(d, 0, (size_t)pl.brick_off, s) != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "hipMemsetAsync failed");
  if (hipStreamSynchronize(s) != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "hipStreamSynchronize failed");
  for (int h = 0; h < CH_HOPS; ++h)
    if (!pl.segs[h].empty() &&
        hipMemcpy(d + pl.seg_off[h], pl.segs[h].data(), pl.segs[h].size() * 4, hipMemcpyHostToDevice) != hipSuccess)
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
  const int h = a->hop / 64 - 1;
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
  k.segs = reinterpret_cast<const int *>(blob + pl.seg_off[h]);
  k.bricks = reinterpret_cast<const float *>(blob + pl.brick_off);
  k.n_bricks = pl.n_bricks;
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
  k.debug = debug | env_int("MISPEC_CHAIN_DEBUG", 0);
  for (int s = 0; s < pl.n_sets; ++s) {
    const ChainSetHost &S = pl.set[s];
    k.first_wg[s] = (int)(s * n_groups);
    k.set[s].n_tiles = S.n_tiles;
    k.set[s].s_lo = S.lo[0];
    k.set[s].seg0 = S.seg0[h];
    k.set[s].n_segs = S.n_segs[h];
    k.set[s].boff0 = S.boff0[h];
    k.set[s].n_sub = S.hi[0] - S.lo[0];
    k.set[s].brick0 = S.brick0;
    for (int m = 0; m < S.n_tiles; ++m) k.set[s].tile[m] = S.tile[m];
  }
  for (int s = pl.n_sets; s <= CH_MAXSETS; ++s) k.first_wg[s] = (int)(pl.n_sets * n_groups);
  const size_t smem = 4 * (size_t)k.ring_bytes + (size_t)CH_NBUF * CH_BUF_BYTES;
  if (smem > 160 * 1024) return mispec_fail_msg(MISPEC_E_UNSUPPORTED, "chain kernel: LDS budget");
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "hipGetDevice failed");
  const bool reflect = a->pad_mode == MISPEC_PAD_REFLECT;
  auto kern = reflect ? cqt_chain_kernel<true> : cqt_chain_kernel<false>;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(g_configured[reflect].load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return mispec_fail_msg(MISPEC_E_HIP, "hipFuncSetAttribute failed (chain kernel)");
    g_configured[reflect].fetch_or(bit, std::memory_order_release);
  }
  static long long *stamps = nullptr;
  if (CH_ABL & 64) {
    if (!stamps) (void)hipMalloc(&stamps, 4096 * 8);
    (void)hipMemsetAsync(stamps, 0, 4096 * 8, static_cast<hipStream_t>(stream));
    k.stamps = stamps;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(n_groups * pl.n_sets)), dim3(512), smem, static_cast<hipStream_t>(stream), k);
  if (hipGetLastError() != hipSuccess) return mispec_fail_msg(MISPEC_E_HIP, "chain kernel launch failed");
  if ((CH_ABL & 64) && getenv("MISPEC_CHAIN_STAMPS")) {
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
    std::vector<long long> h(4096);
    (void)hipMemcpy(h.data(), stamps, 4096 * 8, hipMemcpyDeviceToHost);
    const long long t0 = h[0];
    fprintf(stderr, "chain stamps (cycles from start): loader ring issued %lld, ring+2 batches landed %lld, first segment %lld, epilogue %lld\n",
            h[1] - t0, h[2] - t0, h[3] - t0, h[4] - t0);
    long long prev = h[8];
    for (int i = 0; i < (int)h[5] && i < 200; ++i) {
      const long long w = h[9 + 2 * i];
      const long long nextt = (i + 1 < (int)h[5] && i + 1 < 200) ? h[8 + 2 * (i + 1)] : h[4];
      const long long n = w & 15, q = (w >> 4) & 15, cnt = w >> 8;
      fprintf(stderr, "  seg %3d: N=%lld Q=%lld count=%4lld  %8lld cycles  = %.1f per MFMA\n", i, n, q, cnt, nextt - h[8 + 2 * i], (double)(nextt - h[8 + 2 * i]) / (double)(4 * n * q * cnt));
      prev = nextt;
    }
    (void)prev;
    for (int w = 0; w < 4; ++w) {
      const double n = (double)h[1003 + 4 * w];
      if (n > 0)
        fprintf(stderr, "loading wave %d: %.0f iterations; per iteration: issue + bookkeeping %.0f, vmcnt wait %.0f, barrier %.0f cycles\n", 4 + w, n,
                h[1000 + 4 * w] / n, h[1001 + 4 * w] / n, h[1002 + 4 * w] / n);
    }
  }
  return MISPEC_OK;
}
