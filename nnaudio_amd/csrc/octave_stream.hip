// octave_stream.hip -- the octave recursion of CQT2010v2 / VQT (cqt.py:1085-1105, vqt.py:160-188,
// utils.py:73-124, 498-521) as a STREAM: mispec_octave_stream_f32 (include/mispec.h).
// Second translation unit of libmispec.so (shares mispec_internal.h with mispec.hip).
//
// The pyramid kernel (octave_pyramid.inl) gives a workgroup 16 frames of a clip and rebuilds, per work
// item, the halo of every level (28 % of its loads and FIR work at the top of the recursion, 5x at the
// bottom), in five barrier-separated phases of which no two overlap.  Here a workgroup of 8 waves walks a
// SEGMENT of one clip in steps of OS_CHUNK = 4096 level-0 samples and keeps RINGS of the levels in LDS
// (rows of 64 samples as (hi, lo) 16-bit planes, absolute position p lives in row (p >> 6) & mask):
//
//   waves 0-3 (FIR)     the FIR outputs of ALL levels of this step as ONE set of 32-output columns: level
//                       l+1 contributes 128 >> (l+1) columns of its block g-l (a step behind level l, so
//                       that everything a step reads was written in an earlier step: one barrier per step);
//                       a lane's column reads five rows of its own input ring, the Toeplitz fragments of
//                       the taps are the same for every column and live in REGISTERS (160 VGPRs; in LDS
//                       they were half of the pyramid kernel's LDS traffic).  Then the INGEST: chunk g+1
//                       (landed in a raw fp32 staging buffer during step g-1 through LDS-direct loads) is
//                       split and written into the level-0 ring, the loads of chunk g+2 are issued
//   waves 4-7 (banks)   wave 4+i contracts one level: the 16-frame tiles its block g-l completes, kernel
//                       rows in registers, fragments of a tile requested while the tile before it is
//                       contracted; the frames at the clip ends from a small PATCH holding the mirrored
//                       samples (nn.ReflectionPad1d; the ring itself keeps the zeros the FIR needs)
//   barrier
//
// scripts/octave_stream_model.py restates this schedule sample by sample (rings full of stale NaNs) and
// is checked against the plain recursion; the plan below is checked against that model
// (tests/test_octave_stream_cpu.py).  Block b of level l covers positions [blk b + c, blk (b+1) + c),
// blk = 4096 >> l, c[l] = 2 c[l+1] + 128: the look-ahead that lets level l's frames of block b and the
// FIR column that ends block b of level l+1 stay inside what is resident.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "mispec.h"
#include "mispec_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4acc __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int OS_LEVELS = MISPEC_STREAM_MAX_LEVELS;
constexpr int OS_CHUNK = 4096;      // level-0 samples per step
constexpr int OS_WAVES = 8;
constexpr int OS_THREADS = OS_WAVES * 64;
constexpr int OS_ROWB = 128;        // bytes of a ring row of one plane: 64 samples x 2
constexpr int OS_KSTEPS = 20;       // 320 columns of the Toeplitz matrix
constexpr int OS_PATCH_ROWS = 8;
constexpr int OS_PATCH_BYTES = 2 * OS_PATCH_ROWS * OS_ROWB;
constexpr int OS_STAGE_BYTES = OS_CHUNK * 4;
constexpr int OS_WARM = 2;          // warm-up steps of a segment (model: warm = 1 reads stale data)

__device__ __forceinline__ void f16_split2(float a, float b, unsigned &hi, unsigned &lo) {
  const f32x2 v = {a, b};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void bf16_split2(float a, float b, unsigned &hi, unsigned &lo) {
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const bf16x2 l = __builtin_convertvector(r, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// |v| of a finite sample, 0 of an infinite one (fmaxf drops NaNs by itself): the fp16 operand scale follows
// the FINITE samples -- an Inf poisons the frames that contain it, as in the reference, not the whole clip
__device__ __forceinline__ float os_finite_abs(float v) {
  const float a = fabsf(v);
  return a < __builtin_inff() ? a : 0.f;
}
// exponent e of m = f 2^e, f in [0.5, 1), clamped so that the scales below stay normal floats
__device__ __forceinline__ int absmax_exponent(float m) {
  int e = (int)((__float_as_uint(m) >> 23) & 0xff) - 126;
  return e < -96 ? -96 : (e > 127 ? 127 : e);
}
__device__ __forceinline__ float pow2f(int e) {
  e = e < -126 ? -126 : (e > 127 ? 127 : e);
  return __uint_as_float((unsigned)(e + 127) << 23);
}

struct OsLevel {
  int L;          // samples of this level
  int hop;        // frame hop (multiple of 4)
  int K, Ks;      // kernel width of the bank (0: none), taps per split row (K rounded up to 32)
  int n_rows;     // bins (<= 16)
  int out_row0;
  int reflect;
  int c;          // look-ahead of the level's blocks
  int mask;       // ring rows - 1
  int ring_off;   // LDS byte offset of the hi plane; lo plane: + plane
  int plane;      // ring rows * OS_ROWB
  int patch_off;  // hi plane of the patch; lo plane: + OS_PATCH_ROWS * OS_ROWB
  const unsigned short *bank;  // planes [re_hi | re_lo | im_hi | im_lo], each (n_rows, Ks)
  long long bank_plane;
  const float *row_scale;
  const float *row_unscale;    // F16
};

struct OsParams {
  const float *x;
  long long x_clip_stride;
  int n_clips, D, nf, span, n_frames;
  int n_seg, blocks_per_seg, n_blocks;
  const float *taps;
  int n_taps, dec_pad;
  OsLevel lv[OS_LEVELS];
  int c_level[4];   // level wave 4+i contracts (-1: none)
  int ingest_mode;  // who streams level 0 in: 0 waves 4-7 after their tiles, 1 waves 0-3 after their columns,
                    // 2 the bank waves WITHOUT a tile in the step, two quarters of the chunk each (os_plan)
  float *x_last;
  long long x_last_stride;
  float *out;
  long long out_clip_stride, out_row_stride;
  int epilogue;
  float im_sign, eps;
  int top;          // F16: scaled level-0 samples stay below 2^top
  int zero_bytes;   // rings + patches (start at LDS offset 0)
  int stage_off, misc_off;  // raw chunk staging (16 KB); maxima / taps staging
  unsigned long long *stamps;  // benchmarking build: phase clock of workgroup 7 (100 MHz ticks), [wave][step][12]
  int debug;                   // benchmarking build: 1 no FIR MFMAs, 2 no bank tiles, 4 no global stores, 8 no DMA, 16 / 32 bank / FIR waves at priority 0 (not 3 / 2),
                               // 64 no tile MFMAs, 128 tile fragments from one address, 256 no split + ring writes of the ingest, 512 no scale check
};

#ifdef MISPEC_ABLATE
#define OS_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (p.stamps && blockIdx.x == 7 && lane == 0 && g - g0 < 32) p.stamps[(wave * 32 + (g - g0)) * 12 + (k)] = wall_clock64(); \
  } while (0)
#define OS_DBG(bit) ((p.debug & (bit)) != 0)
#else
#define OS_STAMP(k) \
  do {              \
  } while (0)
#define OS_DBG(bit) (false)
#endif

// 16 bytes per lane, global -> LDS at m0 + 16 lane (the compiler neither sees the LDS write nor
// counts the load: the wait is stated by hand where the staging buffer is read)
__device__ __forceinline__ void os_dma16(const void *src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
}

// the four 16-byte pieces of a thread's 64 bytes of a chunk: scalar base + 32-bit lane offset, piece h at LDS m0 + 1024 h
__device__ __forceinline__ void os_dma64(const void *sbase, unsigned voff, unsigned lds_addr) {
  const char *b = static_cast<const char *>(sbase);
  asm volatile(
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %4"
      :
      : "v"(voff), "s"(b), "s"(b + 16), "s"(b + 32), "s"(b + 48), "s"(lds_addr)
      : "memory", "m0");
}

__device__ __forceinline__ void os_epilogue_store(int epi, float eps, float *dst, float re, float im) {
  switch (epi) {
    case MISPEC_EPI_COMPLEX:
      *reinterpret_cast<float2 *>(dst) = make_float2(re, im);
      break;
    case MISPEC_EPI_MAGNITUDE:
      dst[0] = sqrtf(re * re + im * im + eps);
      break;
    case MISPEC_EPI_POWER:
      dst[0] = re * re + im * im + eps;
      break;
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

// byte offset of sample `pos` inside a plane whose rows are indexed by (row - sub) & msk
__device__ __forceinline__ int os_addr(int pos, int sub, int msk) {
  const int row = ((pos >> 6) - sub) & msk;
  return row * OS_ROWB + (((((pos & 63) >> 3) ^ ((row >> 1) & 7))) << 4) + ((pos & 7) << 1);
}

template <int MAXS, bool F16>
__global__ void __launch_bounds__(OS_THREADS, 1) octave_stream_kernel(const OsParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.D;
  constexpr float TAP_SCALE = F16 ? 16384.f : 1.f;
  auto split2 = [](float a, float b, unsigned &h, unsigned &l) __attribute__((always_inline)) {
    if (F16) f16_split2(a, b, h, l);
    else bf16_split2(a, b, h, l);
  };
  auto mfma32 = [](bf16x8 a, bf16x8 b, f32x16 c) __attribute__((always_inline)) -> f32x16 {
    if (F16)
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };
  auto mfma16 = [](bf16x8 a, bf16x8 b, f32x4acc c) __attribute__((always_inline)) -> f32x4acc {
    if (F16)
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  };
  auto wave_sync = []() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // ---- the segment of this workgroup
  const int c = blockIdx.x / p.n_seg;
  const int seg = blockIdx.x - c * p.n_seg;
  const int b_a = seg * p.blocks_per_seg;
  if (b_a >= p.n_blocks) return;
  const int b_e = (b_a + p.blocks_per_seg < p.n_blocks) ? b_a + p.blocks_per_seg : p.n_blocks;
  const int g0 = b_a - OS_WARM, g_end = b_e + D - 1;
  const float *const xc = p.x + (long long)c * p.x_clip_stride;
  const int L0 = p.lv[0].L, c0 = p.lv[0].c;

  // ---- LDS: rings and patches zeroed (the FIR may multiply a slot nobody has written yet with a zero tap)
  for (int i = tid * 16; i < p.zero_bytes; i += OS_THREADS * 16)
    *reinterpret_cast<u32x4 *>(smem + i) = u32x4{0u, 0u, 0u, 0u};
  unsigned *const s_max = reinterpret_cast<unsigned *>(smem + p.misc_off);  // [2][4] bit patterns of the ingesting waves' maxima
  float *const s_taps = reinterpret_cast<float *>(smem + p.misc_off + 64);
  for (int i = tid; i < p.n_taps; i += OS_THREADS) s_taps[i] = p.taps[i];
  __syncthreads();

  // ---- streaming of level 0 in quarters of a chunk (who takes them: p.ingest_mode).  Thread ct of the four quarters
  // owns samples 16 ct .. 16 ct + 15 of every chunk; the raw floats land in the staging buffer
  // as [wave][piece][lane] 16-byte pieces (LDS-direct loads), a step before they are split.
  // F16 operand scale: the chunk that is split in step g-1 publishes its largest |sample|; step g -- its first
  // consumer -- starts by comparing it with the range of the current scale and, when it is louder (rare),
  // rescales everything resident and splits that chunk again from memory.
  unsigned char *const stage = smem + p.stage_off;
  const bool ingest_wave = p.ingest_mode == 0 ? wave >= 4 : wave < 4;  // (prologue, rescale; mode 2: the steps' ingest moves around)
  const int cw = wave & 3;
  // the quarter of a chunk this wave is streaming in (mode 2: a wave takes two, one after the other: set_quarter)
  int iq = cw, ct = 64 * cw + lane;
  const unsigned stage_lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)stage);
  unsigned stage_lds = stage_lds0 + 4096u * (unsigned)cw;
  auto chunk_pos = [&](int q) __attribute__((always_inline)) { return OS_CHUNK * q + c0 + 16 * ct; };
  auto dma_chunk = [&](int q) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int pos = chunk_pos(q) + 4 * h;
      if (pos >= 0 && pos + 4 <= L0 && !OS_DBG(8)) os_dma16(xc + pos, stage_lds + 1024 * h);
    }
  };
  auto read_chunk = [&](int q, float (&v)[16], bool staged) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int pos = chunk_pos(q) + 4 * h;
      if (staged && pos >= 0 && pos + 4 <= L0) {
        const f32x4v f = *reinterpret_cast<const f32x4v *>(stage + iq * 4096 + 1024 * h + 16 * lane);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * h + e] = f[e];
      } else {  // (the pieces that cross an end of the clip; every piece on the slow path)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * h + e] = (pos + e >= 0 && pos + e < L0) ? xc[pos + e] : 0.f;
      }
    }
  };
  auto wave_max_bits = [&](const float (&v)[16]) __attribute__((always_inline)) -> unsigned {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) m = fmaxf(m, os_finite_abs(v[e]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    return __float_as_uint(m);
  };
  auto published_max = [&](int parity) __attribute__((always_inline)) -> float {
    const u32x4 v = *reinterpret_cast<const u32x4 *>(s_max + 4 * parity);  // (one read: the check opens every step)
    const unsigned a = v[0] > v[1] ? v[0] : v[1], b = v[2] > v[3] ? v[2] : v[3];
    return __uint_as_float(a > b ? a : b);
  };
  int e_cur = 0;  // F16: every resident sample is x 2^(top - e_cur)
  auto write_chunk = [&](int q, const float (&v)[16]) __attribute__((always_inline)) {
    const float xs = F16 ? pow2f(p.top - e_cur) : 1.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      unsigned h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split2(v[8 * half + 2 * e] * xs, v[8 * half + 2 * e + 1] * xs, h[e], l[e]);
      const int a = p.lv[0].ring_off + os_addr(chunk_pos(q) + 8 * half, 0, p.lv[0].mask);
      *reinterpret_cast<u32x4 *>(smem + a) = u32x4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<u32x4 *>(smem + a + p.lv[0].plane) = u32x4{l[0], l[1], l[2], l[3]};
    }
  };
  // every resident (hi, lo) pair x 2^-k (exact up to underflow)
  auto rescale = [&](int k) __attribute__((always_inline)) {
    const float f = pow2f(-k);
    for (int i = tid * 16; i < p.zero_bytes; i += OS_THREADS * 16) {
      f16x8 v = *reinterpret_cast<const f16x8 *>(smem + i);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] * f);
      *reinterpret_cast<f16x8 *>(smem + i) = v;
    }
  };
  // start of step g, every wave: was chunk g (split a step ago, first read in this step) inside the scale's range?
  auto check_scale = [&](int g) __attribute__((always_inline)) {
    if (!F16 || g > b_e - 1 || OS_DBG(512)) return;
    const int e_need = absmax_exponent(published_max(g & 1));
    if (e_need > e_cur) {  // (workgroup-uniform: every wave reads the same four words)
      rescale(e_need - e_cur);
      e_cur = e_need;
      __syncthreads();
      if (ingest_wave) {
        float v[16];
        read_chunk(g, v, false);
        write_chunk(g, v);
      }
      __syncthreads();
    }
  };
  // s_waitcnt vmcnt(n): at most n vector-memory operations of this wave still outstanding.  The loads of a
  // chunk are older than the stores a bank wave issued since: with n = (a lower bound of) those stores the
  // loads have landed -- loads and stores of a wave retire in order on this counter (what the compiler's own
  // waitcnt insertion assumes on gfx9: SIInsertWaitcnts, one VMEM event type before gfx10).
  auto wait_loads = [&](int younger) __attribute__((always_inline)) {
    if (younger >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- prologue: chunk g0 into the ring (from memory), chunk g0 + 1 into the staging buffer, the first scale
  {
    float v[16];
    unsigned m0 = 0;
    if (ingest_wave) {
      read_chunk(g0, v, false);
      m0 = wave_max_bits(v);
      if (g0 + 1 <= b_e - 1) {
        float v1[16];
        dma_chunk(g0 + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        read_chunk(g0 + 1, v1, true);
        const unsigned m1 = wave_max_bits(v1);
        m0 = m1 > m0 ? m1 : m0;
      }
      if (F16 && lane == 0) s_max[iq] = m0, s_max[4 + iq] = m0;
    }
    if (F16) {
      __syncthreads();
      e_cur = absmax_exponent(published_max(0));
    }
    if (ingest_wave) write_chunk(g0, v);
    __syncthreads();
  }

  // ---- the ingest of one step (by the four ingesting waves)
  // ingest of level 0: thread ct owns samples 16 ct .. 16 ct + 15 of a chunk = two 16-byte pieces of ring row
  // ((c0 + 16 ct) >> 6) + 64 q (a level-0 ring has >= 128 rows: the swizzle of the row does not depend on q)
  const int ring0_off = p.lv[0].ring_off, ring0_plane = p.lv[0].plane, ring0_mask = p.lv[0].mask;
  int wr_row0, wr_col[2];
  unsigned dma_voff;
  auto set_quarter = [&](int j) __attribute__((always_inline)) {
    iq = j;
    ct = 64 * j + lane;
    stage_lds = stage_lds0 + 4096u * (unsigned)j;
    wr_row0 = (c0 + 16 * ct) >> 6;
#pragma unroll
    for (int half = 0; half < 2; ++half)
      wr_col[half] = ((((((c0 + 16 * ct) & 63) >> 3) + half) ^ ((wr_row0 >> 1) & 7)) << 4);
    dma_voff = 64u * (unsigned)ct;
  };
  set_quarter(cw);
  // the ingest of the current quarter (set_quarter), chunk inside the clip, in two halves: the staging slots into
  // registers and the request for chunk q2 into the same slots ...
  auto ingest_fetch = [&](int q2, bool more, f32x4v (&f)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 4; ++h) f[h] = *reinterpret_cast<const f32x4v *>(stage + iq * 4096 + 1024 * h + 16 * lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging slots have been read: they may be refilled
    if (more) {
      if (OS_CHUNK * q2 + c0 >= 0 && OS_CHUNK * (q2 + 1) + c0 <= L0)
        os_dma64(xc + ((long long)OS_CHUNK * q2 + c0), dma_voff, stage_lds);
      else
        dma_chunk(q2);
    }
  };
  // ... and the sixteen samples of chunk q: their largest finite magnitude for the next step's check, split, ring
  auto ingest_commit = [&](int q, const f32x4v (&f)[4]) __attribute__((always_inline)) {
    if (F16) {
      // magnitudes compared as integers (bit patterns of |x| order like the values; one AND + one MAX per sample,
      // no NaN handling): only when some |sample| >= 2^e_cur -- or is not finite -- is the true maximum formed
      unsigned mi = 0u;
#pragma unroll
      for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned a = __float_as_uint(f[h][e]) & 0x7fffffffu;
          mi = mi > a ? mi : a;
        }
      unsigned mb = 0u;
      if (__builtin_amdgcn_ballot_w64(mi >= __float_as_uint(pow2f(e_cur))) != 0ull) {
        float m = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) m = fmaxf(m, os_finite_abs(f[h][e]));
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        mb = __float_as_uint(m);
      }
      if (lane == 0) s_max[4 * (q & 1) + iq] = mb;
    }
    const float xs = F16 ? pow2f(p.top - e_cur) : 1.f;
    const int wrow = (((wr_row0 + 64 * q) & ring0_mask) << 7) + ring0_off;
#pragma unroll
    for (int half = 0; half < (OS_DBG(256) ? 0 : 2); ++half) {
      unsigned h[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split2(f[2 * half + (e >> 1)][2 * (e & 1)] * xs, f[2 * half + (e >> 1)][2 * (e & 1) + 1] * xs, h[e], l[e]);
      const int a = wrow + wr_col[half];
      *reinterpret_cast<u32x4 *>(smem + a) = u32x4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<u32x4 *>(smem + a + ring0_plane) = u32x4{l[0], l[1], l[2], l[3]};
    }
  };
  // a chunk that crosses an end of the clip: piece by piece
  auto ingest_slow = [&](int q, int q2, bool more) __attribute__((always_inline)) {
    float v16[16];
    read_chunk(q, v16, true);
    if (F16) {
      const unsigned m = wave_max_bits(v16);
      if (lane == 0) s_max[4 * (q & 1) + iq] = m;
    }
    write_chunk(q, v16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more) {
      if (OS_CHUNK * q2 + c0 >= 0 && OS_CHUNK * (q2 + 1) + c0 <= L0)
        os_dma64(xc + ((long long)OS_CHUNK * q2 + c0), dma_voff, stage_lds);
      else
        dma_chunk(q2);
    }
  };
  // modes 0 and 1: one quarter per wave and step, requested by the same wave a step ago (the wait for it is here)
  auto ingest_step = [&](int g, int &younger) __attribute__((always_inline)) {
      // chunk g + 1 (requested a step ago): staging buffer -> ring; its maximum for the check of the next step;
      // the request for chunk g + 2 into the same slots as soon as they have been read
      if (g + 1 <= b_e - 1) {
        wait_loads(younger);
        OS_STAMP(4);
        const int q = g + 1, q2 = g + 2;
        const bool more = q2 <= b_e - 1 && !OS_DBG(8);
        if (OS_CHUNK * q + c0 >= 0 && OS_CHUNK * (q + 1) + c0 <= L0) {
          // the chunk lies inside the clip (every step but the ends of the clip): no per-piece tests
          f32x4v f[4];
          ingest_fetch(q2, more, f);
          OS_STAMP(5);
          ingest_commit(q, f);
        } else {
          ingest_slow(q, q2, more);
        }
        younger = 0;
      }
  };
  // mode 2: two quarters by a bank wave that has no tile in this step; both requests go out before the samples are
  // worked on, and are waited for before the step's barrier (the next step's takers are other waves, and vmcnt is per
  // wave).  (Requesting before the scale check -- the staged floats do not depend on the scale -- keeps 32 registers
  // alive across it: spills.)
  auto ingest_pair = [&](int g, int j0, int j1) __attribute__((always_inline)) {
      if (g + 1 <= b_e - 1) {
        const int q = g + 1, q2 = g + 2;
        const bool more = q2 <= b_e - 1 && !OS_DBG(8);
        if (OS_CHUNK * q + c0 >= 0 && OS_CHUNK * (q + 1) + c0 <= L0) {
          f32x4v fa[4], fb[4];
          set_quarter(j0);
          ingest_fetch(q2, more, fa);
          set_quarter(j1);
          ingest_fetch(q2, more, fb);
          OS_STAMP(5);
          ingest_commit(q, fb);
          set_quarter(j0);
          ingest_commit(q, fa);
        } else {
          set_quarter(j0);
          ingest_slow(q, q2, more);
          set_quarter(j1);
          ingest_slow(q, q2, more);
        }
        OS_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OS_STAMP(6);
      }
  };

  if (wave < 4) {
    // =============================== FIR waves ===============================
    if (!OS_DBG(32)) __builtin_amdgcn_s_setprio(2);  // (the MFMA stream of a step goes first: measured -12 %)
    const int li = lane & 31, lh = lane >> 5;
    // Toeplitz fragments: lane (r = li, lh) of step s holds T[r, 16 s + 8 lh + e] = taps[16 s + 8 lh + e - 2 r - shift]
    bf16x8 th[OS_KSTEPS], tl[OS_KSTEPS];
    {
      const int shift = 128 - p.dec_pad;
#pragma unroll
      for (int s = 0; s < OS_KSTEPS; ++s) {
        unsigned h[4], l[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          float v[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int k = 16 * s + 8 * lh + 2 * e2 + u - 2 * li - shift;
            v[u] = (k >= 0 && k < p.n_taps) ? s_taps[k] * TAP_SCALE : 0.f;
          }
          split2(v[0], v[1], h[e2], l[e2]);
        }
        th[s] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
        tl[s] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
      }
    }
    // this lane's column: output level lam, column kap of the level's block
    int lam = 1, kap = 0;
    bool col_ok = false;
    {
      const int j = 32 * wave + li;
      int first = 0;
      for (int l = 1; l < D; ++l) {
        const int n = (OS_CHUNK >> l) / 32;
        if (j >= first && j < first + n) {
          lam = l;
          kap = j - first;
          col_ok = true;
        }
        first += n;
      }
    }
    const int in_hi = p.lv[lam - 1].ring_off, in_plane = p.lv[lam - 1].plane, in_mask = p.lv[lam - 1].mask;
    const int out_hi = p.lv[lam].ring_off, out_plane = p.lv[lam].plane, out_mask = p.lv[lam].mask;
    const int out_L = p.lv[lam].L, out_blk = OS_CHUNK >> lam, out_c = p.lv[lam].c;
    const bool to_last = col_ok && lam == D - 1 && p.x_last != nullptr;
    const int own_lo = b_a == 0 ? 0 : (OS_CHUNK >> (D - 1)) * b_a + p.lv[D - 1].c;
    int own_hi = (OS_CHUNK >> (D - 1)) * b_e + p.lv[D - 1].c;
    own_hi = own_hi < p.lv[D - 1].L ? own_hi : p.lv[D - 1].L;
    float *const xl_out = p.x_last ? p.x_last + (long long)c * p.x_last_stride : nullptr;
    const int lh16 = lh * 16;
    int fir_younger = 0;  // store instructions since the chunk that is staged was requested (the x_last stores)

    for (int g = g0; g < g_end; ++g) {
      OS_STAMP(0);
      check_scale(g);
      OS_STAMP(1);
      const int beta = g - (lam - 1);
      const bool active = col_ok && beta >= b_a - OS_WARM && beta < b_e;
      if (__builtin_amdgcn_ballot_w64(active) != 0ull && !OS_DBG(1)) {
        const int P = out_blk * (active ? beta : b_a) + out_c + 32 * kap;  // (idle lanes: any resident column)
        const int rho = (P >> 5) - 2;                                       // first input row: position 2 P - 128
        int rowoff[5], sw[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int r = (rho + j) & in_mask;
          rowoff[j] = in_hi + r * OS_ROWB;
          sw[j] = ((r >> 1) & 7) << 4;
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        bf16x8 xh[2], xl[2];
        auto frags = [&](int s, int slot) __attribute__((always_inline)) {
          const int a = rowoff[s >> 2] + ((((s & 3) * 32) + lh16) ^ sw[s >> 2]);
          xh[slot] = *reinterpret_cast<const bf16x8 *>(smem + a);
          xl[slot] = *reinterpret_cast<const bf16x8 *>(smem + a + in_plane);
        };
        frags(0, 0);
        OS_STAMP(2);
#pragma unroll
        for (int s = 0; s < OS_KSTEPS; ++s) {
          const int k = s & 1;
          if (s + 1 < OS_KSTEPS) frags(s + 1, k ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          acc = mfma32(tl[s], xh[k], acc);
          acc = mfma32(th[s], xl[k], acc);
          acc = mfma32(th[s], xh[k], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
        OS_STAMP(3);
        // acc[4 g4 + e] = y[P + 8 g4 + 4 lh + e]
        const bool inner = active && P >= 0 && P + 32 <= out_L && (!to_last || (P >= own_lo && P + 32 <= own_hi));
        const bool all_inner = __builtin_amdgcn_ballot_w64(active && !inner) == 0ull;
        // (x_last stores of this step, for the ingest's wait: four when the fast path issues them, unknown -> the
        // next wait drains the queue -- otherwise)
        if (!all_inner) fir_younger = 0;
        else if (__builtin_amdgcn_ballot_w64(active && to_last) != 0ull && !OS_DBG(4)) fir_younger += 4;
        if (all_inner) {
          // every column of the wave lies inside its level (and inside the segment's share of x_last): no per-sample tests
          if (active) {
            const float xu = F16 ? pow2f(e_cur - p.top) : 1.f;
            const int rowi = (P >> 6) & out_mask;
            const int sw = (rowi >> 1) & 7, cb = (P >> 3) & 7;
            const int base = out_hi + rowi * OS_ROWB + 8 * lh;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              float f[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) f[e] = acc[4 * g4 + e] * (1.f / TAP_SCALE);
              uint2 h, lw;
              split2(f[0], f[1], h.x, lw.x);
              split2(f[2], f[3], h.y, lw.y);
              const int a = base + (((cb + g4) ^ sw) << 4);
              *reinterpret_cast<uint2 *>(smem + a) = h;
              *reinterpret_cast<uint2 *>(smem + a + out_plane) = lw;
              if (to_last && !OS_DBG(4))
                *reinterpret_cast<f32x4u *>(xl_out + P + 8 * g4 + 4 * lh) = f32x4u{f[0] * xu, f[1] * xu, f[2] * xu, f[3] * xu};
            }
          }
        } else if (active) {
          const float xu = F16 ? pow2f(e_cur - p.top) : 1.f;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int pos = P + 8 * g4 + 4 * lh;
            float f[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              f[e] = (pos + e >= 0 && pos + e < out_L) ? acc[4 * g4 + e] * (1.f / TAP_SCALE) : 0.f;
            uint2 h, lw;
            split2(f[0], f[1], h.x, lw.x);
            split2(f[2], f[3], h.y, lw.y);
            const int a = out_hi + os_addr(pos, 0, out_mask);
            *reinterpret_cast<uint2 *>(smem + a) = h;
            *reinterpret_cast<uint2 *>(smem + a + out_plane) = lw;
            if (to_last && !OS_DBG(4) && pos + 3 >= own_lo && pos < own_hi) {
              if (pos >= own_lo && pos + 3 < own_hi) {
                *reinterpret_cast<f32x4u *>(xl_out + pos) = f32x4u{f[0] * xu, f[1] * xu, f[2] * xu, f[3] * xu};
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (pos + e >= own_lo && pos + e < own_hi) xl_out[pos + e] = f[e] * xu;
              }
            }
          }
        }
      }
      if (p.ingest_mode == 1) ingest_step(g, fir_younger);
      OS_STAMP(10);
      __syncthreads();
      OS_STAMP(11);
    }
  } else {
    // =============================== bank waves ===============================
    // (above the FIR waves' 2: in launches whose steps are tile-bound -- hops of 32 .. 4, the second launch of the cfg5
    // shard -- 96 -> 92 us; neutral where the bank waves are not the longest chain)
    if (!OS_DBG(16)) __builtin_amdgcn_s_setprio(3);
    const int my = p.c_level[cw];
    const int fn = lane & 15, kg = lane >> 4;
    bf16x8 rh[MAXS], rl[MAXS], ih[MAXS], il[MAXS];
    {
      const OsLevel &v = p.lv[my >= 0 ? my : 0];
      const int steps = my >= 0 ? v.Ks / 32 : 0;
      const int arow = (lane & 15) < v.n_rows ? (lane & 15) : (v.n_rows > 0 ? v.n_rows - 1 : 0);
      const unsigned short *are = v.bank + (long long)arow * v.Ks + 8 * kg;
      const unsigned short *aim = are + 2 * v.bank_plane;
      const bf16x8 zero = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {  // (steps past the bank's width multiply zeros: no branch in the tile loop)
        rh[s] = rl[s] = ih[s] = il[s] = zero;
        if (s < steps) {
          rh[s] = *reinterpret_cast<const bf16x8 *>(are + 32 * s);
          rl[s] = *reinterpret_cast<const bf16x8 *>(are + v.bank_plane + 32 * s);
          ih[s] = *reinterpret_cast<const bf16x8 *>(aim + 32 * s);
          il[s] = *reinterpret_cast<const bf16x8 *>(aim + v.bank_plane + 32 * s);
        }
      }
      // (round 5, scripts/isa_waits.py: values that come from global LOADS in this prologue and are used in the tile loop must be
      // pinned here -- hipcc's waitcnt pass otherwise carries "may still be in flight" into the loop and puts s_waitcnt vmcnt(0) in
      // front of their uses, where it waits for the stores the wave has just issued)
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        asm volatile("" : "+v"(rh[s]));
        asm volatile("" : "+v"(rl[s]));
        asm volatile("" : "+v"(ih[s]));
        asm volatile("" : "+v"(il[s]));
      }
    }
    const OsLevel &v = p.lv[my >= 0 ? my : 0];
    const int v_L = v.L, v_hop = v.hop, v_K = v.K, v_half = v.K / 2, v_n_rows = v.n_rows, v_row0 = v.out_row0;
    const int v_ring = v.ring_off, v_plane = v.plane, v_mask = v.mask, v_patch = v.patch_off, v_reflect = v.reflect;
    const float *const v_scale = v.row_scale, *const v_unscale = v.row_unscale;
    const int E = (p.epilogue == MISPEC_EPI_COMPLEX || p.epilogue == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
    float *const o_clip = p.out + (long long)c * p.out_clip_stride;
    const bool half_chunk = (v_hop & 7) != 0;
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    // per-bin factors of this lane's four bins (4 kg + e)
    float bsc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int bin = 4 * kg + e;
      float s = (my >= 0 && bin < v_n_rows && v_scale) ? v_scale[bin] : 1.f;
      if (F16 && my >= 0 && bin < v_n_rows) s *= v_unscale[bin];
      bsc[e] = s;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(bsc[e]));  // (pinned: see the bank fragments above)

    // the patch of one clip end: rows rho0 .. rho0 + 7 of the level with the mirrored samples beyond the end
    auto build_patch = [&](int rho0) __attribute__((always_inline)) {
      wave_sync();
      const int p0 = 64 * rho0 + 8 * lane;
      unsigned hw[4] = {0u, 0u, 0u, 0u}, lw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int q = p0 + e;
        int src = q < 0 ? -q : (q >= v_L ? 2 * (v_L - 1) - q : q);
        src = src < 0 ? 0 : (src >= v_L ? v_L - 1 : src);
        const int a = v_ring + os_addr(src, 0, v_mask);
        const unsigned hv = *reinterpret_cast<const unsigned short *>(smem + a);
        const unsigned lv2 = *reinterpret_cast<const unsigned short *>(smem + a + v_plane);
        hw[e >> 1] |= hv << (16 * (e & 1));
        lw[e >> 1] |= lv2 << (16 * (e & 1));
      }
      const int prow = lane >> 3, chunk = lane & 7;
      const int a = v_patch + prow * OS_ROWB + ((chunk ^ ((prow >> 1) & 7)) << 4);
      *reinterpret_cast<u32x4 *>(smem + a) = u32x4{hw[0], hw[1], hw[2], hw[3]};
      *reinterpret_cast<u32x4 *>(smem + a + OS_PATCH_ROWS * OS_ROWB) = u32x4{lw[0], lw[1], lw[2], lw[3]};
      wave_sync();
    };

    // element offsets of this lane's four bins at frame fn of a tile (a clip's output stays below 2^29 elements: see the launch)
    const unsigned ooff0 = (unsigned)((long long)(v_row0 + 4 * kg) * p.out_row_stride + (long long)fn * E);
    const unsigned orow = (unsigned)p.out_row_stride;  // (bin 4 kg + e: + e rows, a scalar)
    const int span = p.span, span_mask = p.span - 1, nf = p.nf, n_frames = p.n_frames, epilogue = p.epilogue;
    const float eps = p.eps, im_sign = p.im_sign;
    int younger = 0;  // store instructions this wave has issued since it requested the chunk that is staged
    // bins of this lane that exist: 4 (the fast store path), 1..3 (a narrow last group) or 0
    const int my_rows = my < 0 ? 0 : (v_n_rows - 4 * kg > 4 ? 4 : (v_n_rows - 4 * kg < 0 ? 0 : v_n_rows - 4 * kg));
    const bool narrow_rows = __builtin_amdgcn_ballot_w64(my_rows > 0 && my_rows < 4) != 0ull;
    constexpr int PF = MAXS == 6 ? 6 : 4;
    // (raw buffer over this clip's output: 32-bit byte offsets, see the size test of the launch)
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(o_clip, 0, 0x7ffffffc, 0x00020000);

    // (scalar) no frame of the tile needs mirrored samples
    auto tile_plain = [&](int tile0) __attribute__((always_inline)) -> bool {
      return !v_reflect || (tile0 * v_hop - v_half >= 0 && (tile0 + 15) * v_hop - v_half + v_K <= v_L);
    };
    // lane (frame fn, kg) holds bins 4 kg + e of frame tile0 + fn
    auto store_tile = [&](int tile0, const f32x4acc &cre, const f32x4acc &cim, float xu) __attribute__((always_inline)) {
      if (OS_DBG(4)) return;
      if (tile0 < n_frames && v_n_rows >= 4) younger += 4;  // (a lower bound of the store instructions below)
      const unsigned ot = (unsigned)(tile0 * E);
      const bool t_ok = tile0 + fn < n_frames;
      if (epilogue == MISPEC_EPI_MAGNITUDE) {
        // (the common case without the switch: lane offsets are constants, the tile's offset is scalar; v_sqrt_f32 is 1 ulp)
        if (t_ok && my_rows == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sc = bsc[e] * xu;
            const float re = cre[e] * sc, im = cim[e] * sc;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(__builtin_amdgcn_sqrtf(re * re + im * im + eps)), o_rsrc,
                                                  (int)(ooff0 * 4u), (int)((ot + e * orow) * 4u), 0);
          }
        }
      } else if (t_ok && my_rows == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc = bsc[e] * xu;
          os_epilogue_store(epilogue, eps, o_clip + (ooff0 + e * orow + ot), cre[e] * sc, im_sign * cim[e] * sc);
        }
      }
      if (narrow_rows && t_ok && my_rows < 4) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          if (e < my_rows) {
            const float sc = bsc[e] * xu;
            os_epilogue_store(epilogue, eps, o_clip + (ooff0 + e * orow + ot), cre[e] * sc, im_sign * cim[e] * sc);
          }
        }
      }
    };
    // a tile at an end of the clip (mirrored samples from the patch) or of a level whose hop is not a multiple of 8
    auto edge_tile = [&](int tile0, float xu) __attribute__((always_inline)) {
      const int w = (tile0 + fn) * v_hop - v_half;  // window start of this lane's frame
      const bool edge_l = v_reflect && w < 0;
      const bool edge_r = v_reflect && w + v_K > v_L && !edge_l;
      int base = v_ring, plane = v_plane, sub = 0, msk = v_mask;
      if (__builtin_amdgcn_ballot_w64(edge_l) != 0ull) {
        build_patch(-2);
        if (edge_l) base = v_patch, plane = OS_PATCH_ROWS * OS_ROWB, sub = -2, msk = OS_PATCH_ROWS - 1;
      } else if (__builtin_amdgcn_ballot_w64(edge_r) != 0ull) {
        // the first frame of the tile whose window passes the end of the level
        int t_e = (v_L - v_half) / v_hop + 1;
        t_e = t_e < tile0 ? tile0 : t_e;
        const int rho0 = (t_e * v_hop - v_half) >> 6;
        build_patch(rho0);
        if (edge_r) base = v_patch, plane = OS_PATCH_ROWS * OS_ROWB, sub = rho0, msk = OS_PATCH_ROWS - 1;
      }
      auto frag = [&](int pl, int pos) __attribute__((always_inline)) -> bf16x8 {
        if (!half_chunk) return *reinterpret_cast<const bf16x8 *>(smem + base + pl + os_addr(pos, sub, msk));
        u64x2 q;
        q[0] = *reinterpret_cast<const unsigned long long *>(smem + base + pl + os_addr(pos, sub, msk));
        q[1] = *reinterpret_cast<const unsigned long long *>(smem + base + pl + os_addr(pos + 4, sub, msk));
        return __builtin_bit_cast(bf16x8, q);
      };
      f32x4acc cre = {0.f, 0.f, 0.f, 0.f}, cim = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s0 = 0; s0 < MAXS; s0 += PF) {
        bf16x8 xh[PF], xl[PF];
#pragma unroll
        for (int s = 0; s < PF; ++s) {
          xh[s] = frag(0, w + 32 * (s0 + s) + 8 * kg);
          xl[s] = frag(plane, w + 32 * (s0 + s) + 8 * kg);
        }
#pragma unroll
        for (int s = 0; s < PF; ++s) {
          cre = mfma16(rl[s0 + s], xh[s], cre);
          cim = mfma16(il[s0 + s], xh[s], cim);
          cre = mfma16(rh[s0 + s], xl[s], cre);
          cim = mfma16(ih[s0 + s], xl[s], cim);
          cre = mfma16(rh[s0 + s], xh[s], cre);
          cim = mfma16(ih[s0 + s], xh[s], cim);
        }
      }
      store_tile(tile0, cre, cim, xu);
    };

    for (int g = g0; g < g_end; ++g) {
      OS_STAMP(0);
      const int beta = g - my;
      // mode 2, a step without a tile for this wave (span = 2: every other one): its own quarter of chunk g + 1 and
      // that of the neighbour, who has a tile now (os_plan pairs levels of opposite parity) -- after the scale check
      check_scale(g);
      OS_STAMP(1);
      if (p.ingest_mode == 2 && ((beta + 1) & 1) != 0) ingest_pair(g, cw, cw ^ 1);
      if (my >= 0 && beta >= b_a && beta < b_e && ((beta + 1) & span_mask) == 0 && !OS_DBG(2)) {
        const float xu = F16 ? pow2f(e_cur - p.top) : 1.f;
        const int f_first = (beta + 1 - span) * nf, f_end = (beta + 1) * nf;
        // (scalar) every tile of this step is plain -- all steps but the first and last ones of a clip
        const bool all_plain = !v_reflect || (f_first * v_hop - v_half >= 0 && (f_end - 1) * v_hop - v_half + v_K <= v_L);
        // the tiles at the ends of the clip first (one call site of the long path) ...
        if (!all_plain) {
          for (int tile0 = f_first; tile0 < f_end; tile0 += 16)
            if (!tile_plain(tile0)) edge_tile(tile0, xu);
        }
        OS_STAMP(6);
        // ... then the plain ones as one stream of (tile, step) pairs: the fragments of step s + H -- of the next
        // plain tile past the end of this one -- are requested while step s is contracted (one register set;
        // the 8-step instance has no registers to spare and requests a whole tile, then contracts it)
        constexpr int H = MAXS == 6 ? 3 : MAXS;
        auto next_plain = [&](int t) __attribute__((always_inline)) -> int {
          if (!all_plain)
            while (t < f_end && !tile_plain(t)) t += 16;
          return t < f_end ? t : -1;
        };
        // this lane's first fragment of a tile in units of 4 samples: whole 16-byte pieces when the level's hop
        // is a multiple of 8, else (hop 4: the deepest octaves) two 8-byte halves
        auto piece0 = [&](int tile0) __attribute__((always_inline)) -> int { return (((tile0 + fn) * v_hop - v_half) >> 2) + 2 * kg; };
        bf16x8 xh[H < MAXS ? MAXS : MAXS / 2], xl[H < MAXS ? MAXS : MAXS / 2];
        auto load_step = [&](int u0, int st, int slot) __attribute__((always_inline)) {
          const int u = u0 + 8 * st;
          if (!half_chunk) {
            const int q = u >> 1;
            const int r = (q >> 3) & v_mask;
            const int a = OS_DBG(128) ? v_ring : v_ring + (r << 7) + (((q ^ (r >> 1)) & 7) << 4);
            xh[slot] = *reinterpret_cast<const bf16x8 *>(smem + a);
            xl[slot] = *reinterpret_cast<const bf16x8 *>(smem + a + v_plane);
          } else {
            u64x2 h, l;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int q = (u + hh) >> 1;
              const int r = (q >> 3) & v_mask;
              const int a = v_ring + (r << 7) + (((q ^ (r >> 1)) & 7) << 4) + (((u + hh) & 1) << 3);
              h[hh] = *reinterpret_cast<const unsigned long long *>(smem + a);
              l[hh] = *reinterpret_cast<const unsigned long long *>(smem + a + v_plane);
            }
            xh[slot] = __builtin_bit_cast(bf16x8, h);
            xl[slot] = __builtin_bit_cast(bf16x8, l);
          }
        };
        if constexpr (H < MAXS) {
          int ta = next_plain(f_first);
          int q0a = ta >= 0 ? piece0(ta) : 0;
          if (ta >= 0) {
#pragma unroll
            for (int st = 0; st < H; ++st) load_step(q0a, st, st);
          }
          while (ta >= 0) {
            const int tn = next_plain(ta + 16);
            const int q0n = tn >= 0 ? piece0(tn) : 0;
            f32x4acc cre = {0.f, 0.f, 0.f, 0.f}, cim = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < MAXS; ++st) {
              if (!OS_DBG(64)) {
                cre = mfma16(rl[st], xh[st], cre);
                cim = mfma16(il[st], xh[st], cim);
                cre = mfma16(rh[st], xl[st], cre);
                cim = mfma16(ih[st], xl[st], cim);
                cre = mfma16(rh[st], xh[st], cre);
                cim = mfma16(ih[st], xh[st], cim);
              }
              __builtin_amdgcn_sched_barrier(0);  // (the request below reuses registers the MFMAs above read)
              if (st + H < MAXS) load_step(q0a, st + H, st + H);
              else if (tn >= 0) load_step(q0n, st + H - MAXS, st + H - MAXS);
              __builtin_amdgcn_sched_barrier(0);
            }
            OS_STAMP(7);
            store_tile(ta, cre, cim, xu);
            ta = tn;
            q0a = q0n;
          }
        } else {
          for (int ta = next_plain(f_first); ta >= 0; ta = next_plain(ta + 16)) {
            const int q0a = piece0(ta);
            f32x4acc cre = {0.f, 0.f, 0.f, 0.f}, cim = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s0 = 0; s0 < MAXS; s0 += MAXS / 2) {
#pragma unroll
              for (int st = 0; st < MAXS / 2; ++st) load_step(q0a, s0 + st, st);
#pragma unroll
              for (int st = 0; st < MAXS / 2; ++st) {
                cre = mfma16(rl[s0 + st], xh[st], cre);
                cim = mfma16(il[s0 + st], xh[st], cim);
                cre = mfma16(rh[s0 + st], xl[st], cre);
                cim = mfma16(ih[s0 + st], xl[st], cim);
                cre = mfma16(rh[s0 + st], xh[st], cre);
                cim = mfma16(ih[s0 + st], xh[st], cim);
              }
            }
            store_tile(ta, cre, cim, xu);
          }
        }
        OS_STAMP(8);
      }
      OS_STAMP(9);
      if (p.ingest_mode == 0) ingest_step(g, younger);  // (after the tiles: VALU work runs at half speed beside the FIR waves' MFMA stream)
      OS_STAMP(10);
      __syncthreads();
      OS_STAMP(11);
    }
  }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
struct OsPlan {
  OsParams p;
  mispec_octave_stream_plan pub;
  size_t smem;
  int max_steps;
};

int os_fail(int code, const char *msg) { return mispec_fail_msg(code, msg); }

int pow2_at_least(int v) {
  int r = 1;
  while (r < v) r *= 2;
  return r;
}

// geometry of a launch (no device access): mirrors plan_stream() of scripts/octave_stream_model.py
int os_plan(const mispec_octave_stream_args *a, int n_cus, OsPlan &pl) {
  if (!a) return os_fail(MISPEC_E_INVALID, "args is NULL");
  if (a->struct_size != sizeof(mispec_octave_stream_args)) return os_fail(MISPEC_E_INVALID, "struct_size mismatch (ABI skew)");
#ifndef MISPEC_ABLATE
  if (a->reserved != 0) return os_fail(MISPEC_E_INVALID, "reserved must be 0");
#endif
  const int D = a->n_levels;
  if (D < 1 || D > OS_LEVELS) return os_fail(MISPEC_E_INVALID, "n_levels must be 1..5");
  if (a->n_clips <= 0 || a->n_samples <= 0 || a->hop <= 0 || a->n_frames <= 0) return os_fail(MISPEC_E_INVALID, "non-positive size");
  if (a->epilogue < MISPEC_EPI_COMPLEX || a->epilogue > MISPEC_EPI_PHASE_COSSIN) return os_fail(MISPEC_E_INVALID, "bad epilogue");
  if (a->precision != MISPEC_PREC_BF16X3 && a->precision != MISPEC_PREC_F16X3)
    return os_fail(MISPEC_E_INVALID, "streaming octave kernel: precision must be MISPEC_PREC_BF16X3 or MISPEC_PREC_F16X3");
  const bool f16 = a->precision == MISPEC_PREC_F16X3;
  if (f16 && (a->fir_headroom_bits < 0 || a->fir_headroom_bits > 7))
    return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: the anti-alias filter's gain leaves no fp16 headroom");
  OsParams &p = pl.p;
  memset(&pl, 0, sizeof(pl));
  p.D = D;
  if (D > 1) {
    if (a->n_taps <= 0 || 128 - (a->n_taps - 1) / 2 < 0 || a->n_taps + 62 + 128 - (a->n_taps - 1) / 2 > 16 * OS_KSTEPS)
      return os_fail(MISPEC_E_UNSUPPORTED, "anti-alias filter too long for the streaming octave kernel");
  }
  p.n_taps = D > 1 ? a->n_taps : 0;
  p.dec_pad = D > 1 ? (a->n_taps - 1) / 2 : 0;
  if (a->hop > 512 || OS_CHUNK % a->hop || (a->hop % (4 << (D - 1))))
    return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: hop must divide 4096, be <= 512 and a multiple of 4 << (levels - 1)");
  if ((OS_CHUNK >> (D - 1)) % 32) return os_fail(MISPEC_E_UNSUPPORTED, "too many levels");
  p.nf = OS_CHUNK / a->hop;
  // the FIR waves take the ingest: with it the bank waves' chain (check, tiles) and theirs (check, columns,
  // ingest) are 1.8 and 2.5 us of a step; on the bank waves they were 2.7 and 1.7 (first launch of the cfg5 shard
  // 245 -> 223 us, second 95 -> 92; MISPEC_INGEST_MODE=0 in the benchmarking build switches back).  Mode 2 below.
  p.ingest_mode = 1;
  p.span = p.nf >= 16 ? 1 : 16 / p.nf;
  p.n_frames = a->n_frames;
  p.n_clips = a->n_clips;
  long long L = a->n_samples;
  int n_banks = 0, max_steps = 0;
  for (int l = 0; l < D; ++l) {
    const mispec_octave_level &v = a->level[l];
    OsLevel &o = p.lv[l];
    if (l > 0) L = (L + 2LL * p.dec_pad - a->n_taps) / 2 + 1;
    if (L <= 0) return os_fail(MISPEC_E_INVALID, "signal too short for this many levels");
    o.L = (int)L;
    o.hop = a->hop >> l;
    if (v.bank_split) {
      if (v.n_bins <= 0 || v.n_bins > 16 || v.kernel < 16 || v.kernel % 16 || v.kernel > 256)
        return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: <= 16 bins, kernel a multiple of 16 up to 256");
      const long long planes = 4LL * v.n_bins * mispec_split_row_taps(v.kernel) * 2;
      if (v.bank_split_bytes < planes + (f16 ? 2LL * v.n_bins * 4 : 0)) return os_fail(MISPEC_E_INVALID, "bank_split too small");
      if (v.pad_mode != MISPEC_PAD_ZERO && v.pad_mode != MISPEC_PAD_REFLECT) return os_fail(MISPEC_E_INVALID, "bad pad_mode");
      if ((long long)(a->n_frames - 1) * o.hop > L) return os_fail(MISPEC_E_INVALID, "n_frames overruns the padded signal");
      // no frame (and no 16-frame tile) may touch both ends of the level
      if (L < 16LL * o.hop + 2LL * v.kernel) return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: clip too short");
      o.K = v.kernel;
      o.Ks = mispec_split_row_taps(v.kernel);  // (the row stride mispec_split_basis_* laid the bank out with)
      o.n_rows = v.n_bins;
      o.out_row0 = v.out_row_offset;
      o.reflect = v.pad_mode == MISPEC_PAD_REFLECT;
      o.bank = static_cast<const unsigned short *>(v.bank_split);
      o.bank_plane = (long long)v.n_bins * o.Ks;
      o.row_scale = v.row_scale;
      o.row_unscale = f16 ? reinterpret_cast<const float *>(static_cast<const char *>(v.bank_split) + planes) : nullptr;
      if (n_banks >= 4) return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: at most four levels with a bank");
      pl.pub.contract_wave[l] = 4 + n_banks;
      p.c_level[n_banks++] = l;
      max_steps = o.Ks / 32 > max_steps ? o.Ks / 32 : max_steps;
    } else {
      pl.pub.contract_wave[l] = -1;
    }
  }
  for (int i = n_banks; i < 4; ++i) p.c_level[i] = -1;
  // tiles of two steps (hop 512: the first launch of a CQT2010v2 / VQT): a bank wave has a tile in every other
  // step -- level l in the steps g = l + 1 (mod 2) -- and nothing to do in between, while the FIR waves' chain
  // (columns + ingest) is the longest of the step: the bank waves without a tile take the ingest, two quarters of
  // the chunk each, when every pair of neighbours (4, 5) and (6, 7) contracts levels of opposite parity
  // (first launch of the cfg5 shard: see DESIGN.md section 3.13)
  if (p.span == 2 && n_banks == 4 && ((p.c_level[0] ^ p.c_level[1]) & 1) && ((p.c_level[2] ^ p.c_level[3]) & 1)) p.ingest_mode = 2;
#ifdef MISPEC_ABLATE
  if (const char *ev = getenv("MISPEC_INGEST_MODE")) {
    const int m = atoi(ev);
    if (m == 0 || m == 1 || (m == 2 && p.ingest_mode == 2)) p.ingest_mode = m;
  }
#endif
  pl.max_steps = max_steps;
  // look-ahead: c[l] = 2 c[l+1] + 128, c[l] >= K[l] / 2 - hop[l]
  for (int cD = 0;; cD += 32) {
    if (cD > 4096) return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: kernels too wide");
    p.lv[D - 1].c = cD;
    for (int l = D - 2; l >= 0; --l) p.lv[l].c = 2 * p.lv[l + 1].c + 128;
    bool ok = true;
    for (int l = 0; l < D; ++l) ok = ok && (!p.lv[l].K || p.lv[l].c >= p.lv[l].K / 2 - p.lv[l].hop);
    if (ok) break;
  }
  // rings: [oldest read, end of the block being written), rounded up to a power of two of rows
  size_t smem = 0;
  for (int l = 0; l < D; ++l) {
    OsLevel &o = p.lv[l];
    const int blk = OS_CHUNK >> l;
    const int newest_end = 2 * blk + o.c;
    const int back_c = (p.span - 1) * blk + (o.K ? o.K / 2 : 0);
    const int back_f = l < D - 1 ? 255 - o.c : 0;
    int back = back_c > back_f ? back_c : back_f;
    back = back > 0 ? back : 0;
    const int rows = pow2_at_least((newest_end + back + 63) / 64 + 1);
    o.mask = rows - 1;
    o.plane = rows * OS_ROWB;
    o.ring_off = (int)smem;
    smem += 2 * (size_t)o.plane;
    pl.pub.ring_rows[l] = rows;
    pl.pub.length[l] = o.L;
    pl.pub.lookahead[l] = o.c;
  }
  for (int l = 0; l < D; ++l) {
    p.lv[l].patch_off = (int)smem;
    if (p.lv[l].K) smem += OS_PATCH_BYTES;
  }
  p.zero_bytes = (int)smem;
  p.stage_off = (int)smem;
  smem += OS_STAGE_BYTES;
  p.misc_off = (int)smem;
  smem += 64 + 1280;  // maxima, taps
  if (smem > 160 * 1024) return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: rings do not fit in LDS");
  pl.smem = smem;
  // blocks (whole tiles; the deepest level is covered: x_last is written by its blocks) and segments
  int n_blocks = (a->n_frames + p.nf - 1) / p.nf;
  n_blocks = (n_blocks + p.span - 1) / p.span * p.span;
  while ((long long)(OS_CHUNK >> (D - 1)) * n_blocks + p.lv[D - 1].c < p.lv[D - 1].L) n_blocks += p.span;
  p.n_blocks = n_blocks;
  int n_seg = a->n_segments;
  if (n_seg <= 0) {
    n_seg = (n_cus + a->n_clips - 1) / a->n_clips;
    const int most = n_blocks / (2 * p.span) > 1 ? n_blocks / (2 * p.span) : 1;  // (at least two tiles of blocks per segment)
    n_seg = n_seg > most ? most : n_seg;
  }
  n_seg = n_seg < 1 ? 1 : n_seg;
  int per = (n_blocks + n_seg - 1) / n_seg;
  per = (per + p.span - 1) / p.span * p.span;
  n_seg = (n_blocks + per - 1) / per;
  p.n_seg = n_seg;
  p.blocks_per_seg = per;
  if ((long long)n_seg * a->n_clips > 0x7fffffffLL) return os_fail(MISPEC_E_UNSUPPORTED, "grid too large");
  pl.pub.n_levels = D;
  pl.pub.frames_per_step = p.nf;
  pl.pub.blocks_per_tile = p.span;
  pl.pub.n_blocks = n_blocks;
  pl.pub.n_segments = n_seg;
  pl.pub.blocks_per_segment = per;
  pl.pub.warm_steps = OS_WARM;
  pl.pub.lds_bytes = (int)smem;
  return MISPEC_OK;
}

}  // namespace

extern "C" {

int mispec_octave_stream_plan_of(const mispec_octave_stream_args *args, int32_t n_cus, mispec_octave_stream_plan *plan) {
  if (!plan) return os_fail(MISPEC_E_INVALID, "plan is NULL");
  OsPlan pl;
  const int rc = os_plan(args, n_cus > 0 ? n_cus : 256, pl);
  if (rc != MISPEC_OK) return rc;
  *plan = pl.pub;
  return MISPEC_OK;
}

int mispec_octave_stream_f32(const mispec_octave_stream_args *a, void *stream) {
  OsPlan pl;
  int rc = os_plan(a, mispec_device_cus(), pl);
  if (rc != MISPEC_OK) return rc;
  if (!a->x || !a->out) return os_fail(MISPEC_E_INVALID, "NULL device pointer");
  if (a->n_levels > 1 && !a->taps) return os_fail(MISPEC_E_INVALID, "taps is required when n_levels > 1");
  if ((reinterpret_cast<uintptr_t>(a->x) & 15) || (a->x_clip_stride & 3))
    return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: x must be 16-byte aligned with a clip stride that is a multiple of 4");
  // (the bank waves address a clip's output with 32-bit element offsets)
  for (int l = 0; l < a->n_levels; ++l) {
    const mispec_octave_level &v = a->level[l];
    if (!v.bank_split) continue;
    if (a->out_row_stride < 0 || v.out_row_offset < 0 ||
        (long long)(v.out_row_offset + v.n_bins) * a->out_row_stride + 2LL * a->n_frames + 32 >= (1LL << 29))
      return os_fail(MISPEC_E_UNSUPPORTED, "streaming octave kernel: a clip's output must stay below 2^29 elements");
  }
  // x_last receives the deepest level of every clip (16-byte stores): a clip's slot must hold it, on 4-float granules
  if (a->x_last && (a->x_last_clip_stride < pl.p.lv[a->n_levels - 1].L || (a->x_last_clip_stride & 3) != 0))
    return os_fail(MISPEC_E_INVALID, "x_last_clip_stride must be a multiple of 4 and at least the deepest level's length");
  OsParams &p = pl.p;
  p.x = a->x;
  p.x_clip_stride = a->x_clip_stride;
  p.taps = a->taps;
  p.x_last = a->x_last;
  p.x_last_stride = a->x_last_clip_stride;
  p.out = a->out;
  p.out_clip_stride = a->out_clip_stride;
  p.out_row_stride = a->out_row_stride;
  p.epilogue = a->epilogue;
  p.im_sign = a->im_sign;
  p.eps = a->eps;
  const bool f16 = a->precision == MISPEC_PREC_F16X3;
  p.top = 15 - a->fir_headroom_bits;
#ifdef MISPEC_ABLATE
  // benchmarking build: `reserved` = debug bits; the phase clock goes to the last 32 KB of x_last's tail? no:
  // MISPEC_OS_STAMPS (environment) carries the device address of an 8 x 32 x 4 array of 64-bit words
  p.debug = a->reserved;
  if (const char *sp = getenv("MISPEC_OS_STAMPS")) p.stamps = reinterpret_cast<unsigned long long *>(strtoull(sp, nullptr, 0));
#endif
  const bool six = pl.max_steps <= 6;
  auto kern = f16 ? (six ? octave_stream_kernel<6, true> : octave_stream_kernel<8, true>)
                  : (six ? octave_stream_kernel<6, false> : octave_stream_kernel<8, false>);
  static std::atomic<unsigned long long> configured[4] = {{0}, {0}, {0}, {0}};
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return os_fail(MISPEC_E_HIP, "hipGetDevice failed");
    const unsigned long long bit = 1ull << (dev & 63);
    std::atomic<unsigned long long> &cf = configured[(f16 ? 2 : 0) + (six ? 0 : 1)];
    if (!(cf.load(std::memory_order_acquire) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return os_fail(MISPEC_E_HIP, hipGetErrorString(e));
      cf.fetch_or(bit, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.n_seg * a->n_clips)), dim3(OS_THREADS), pl.smem, static_cast<hipStream_t>(stream), p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return os_fail(MISPEC_E_HIP, hipGetErrorString(e));
  return MISPEC_OK;
}

}  // extern "C"
