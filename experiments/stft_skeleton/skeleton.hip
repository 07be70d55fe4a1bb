// skeleton.hip -- what the MEMORY SKELETON of the n_fft = 2048 STFT costs on an MI355X, by access pattern, with no
// transform at all: 64 clips x 441000 samples in (113 MB), (64, 1025, 862) floats out (226 MB), frames innermost.
// The FFT kernel (nnaudio_amd/csrc/stft_fft.inl) produces all bins of ONE frame per wave and has to write rows of
// frames: a workgroup transposes a tile of FT frames through LDS and stores row segments of 4 FT bytes, 3448 bytes
// apart.  Which part of that is expensive -- the 32-byte segments, the 4 x redundant frame loads (hop = n_fft / 4),
// the lock step of loads / stores -- is measured here before a kernel is built around the answer.
//   hipcc --offload-arch=gfx950 -O3 skeleton.hip -o skeleton && ./skeleton
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int B = 64, L = 441000, NFFT = 2048, HOP = 512, F = 1025, T = 862, PAD = 1024;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() is a release fence and makes hipcc drain vmcnt
// -- every store (and LDS-direct load) in flight -- in front of s_barrier
#ifdef RAW_BARRIER
#define BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define BAR() __syncthreads()
#endif

__device__ __forceinline__ void dma16(const void *src, unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_off(const void *p) {
  typedef __attribute__((address_space(3))) void *lptr_t;
  return (unsigned)(unsigned long long)(lptr_t)p;
}

// ---- E: plain streaming copy of the same byte counts (the ceiling): read 113 MB, write 226 MB, 16 B per lane
__global__ void __launch_bounds__(512) copy_kernel(const f4 *__restrict__ x, f4 *__restrict__ out, long long n_in, long long n_out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  f4 acc = {0, 0, 0, 0};
  for (long long j = i; j < n_in; j += stride) acc += x[j];
  for (long long j = i; j < n_out; j += stride) out[j] = acc + (float)j;
}

// ---- F: stores only.  A workgroup owns tiles of FT frames x 1025 rows and writes them with 16-byte stores straight
// from registers (LPR = FT / 4 lanes per row).  RUN = 1: a workgroup walks consecutive tiles of a clip (its 32-byte
// pieces of a 128-byte line follow each other in time on ONE CU); RUN = 0: consecutive tiles of a clip belong to
// the workgroups of one XCD at the same time (as stft_fft_kernel).
template <int FT, int RUN>
__global__ void __launch_bounds__(512) store_kernel(float *__restrict__ out, int tiles_per_clip) {
  constexpr int LPR = FT / 4, RPI = 512 / LPR;
  const int tid = threadIdx.x;
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  const int fl = (tid % LPR) * 4, r0 = tid / LPR;
  auto do_tile = [&](int tile_id) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    float *oc = out + (long long)c * F * T;
    if (t0 + fl + 4 <= T) {
#pragma unroll 4
      for (int k = r0; k < F; k += RPI) *reinterpret_cast<f4u *>(oc + (long long)k * T + t0 + fl) = f4{(float)k, 1.f, 2.f, 3.f};
    }
  };
  if (RUN) {
    const int per_wg = (n_tiles + nwg - 1) / nwg;
    for (int i = 0; i < per_wg; ++i) {
      const int tile_id = blockIdx.x * per_wg + i;
      if (tile_id < n_tiles) do_tile(tile_id);
    }
  } else {
    for (int it = blockIdx.x >> 3; it < per_xcd; it += (nwg + 7) >> 3) {
      const int tile_id = (blockIdx.x & 7) * per_xcd + it;
      if (tile_id < n_tiles) do_tile(tile_id);
    }
  }
}

// ---- G: loads only.  MODE 0: every wave DMAs whole frames (8 KB each; 4 x redundant); MODE 1: the workgroup DMAs
// the tile's span once (2048 + (FT - 1) 512 samples).  Two landing buffers, the next tile requested before this one
// is "used" (a few LDS reads + a dependent add).
template <int FT, int MODE>
__global__ void __launch_bounds__(512) load_kernel(const float *__restrict__ x, float *__restrict__ sink, int tiles_per_clip) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SPAN = NFFT + (FT - 1) * HOP;                    // samples
  constexpr int LAND = MODE == 0 ? 8 * NFFT : ((SPAN + 255) / 256) * 256;  // samples per landing buffer
  float *land = reinterpret_cast<float *>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  float acc = 0.f;
  auto request = [&](int tile_id, int bufi) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    const float *xc = x + (long long)c * L;
    float *dst = land + bufi * LAND;
    if (MODE == 0) {
      for (int u = 0; u < FT / 8; ++u) {
        long long pos = (long long)(t0 + wave * (FT / 8) + u) * HOP - PAD;
        pos = pos < 0 ? 0 : (pos + NFFT > L ? L - NFFT : pos);
#pragma unroll
        for (int j = 0; j < NFFT / 256; ++j) dma16(xc + pos + 256 * j + 4 * lane, lds_off(dst + wave * NFFT + 256 * j) );
      }
    } else {
      long long pos = (long long)t0 * HOP - PAD;
      pos = pos < 0 ? 0 : (pos + SPAN > L ? L - SPAN : pos);
      for (int j = wave; j < LAND / 256; j += 8) dma16(xc + pos + 256 * j + 4 * lane, lds_off(dst + 256 * j));
    }
  };
  int it = blockIdx.x >> 3, step = 0;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  if (it < per_xcd && tile_id < n_tiles) request(tile_id, 0);
  while (it < per_xcd && tile_id < n_tiles) {
    const int it_n = it + ((nwg + 7) >> 3), tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    if (have_n) request(tile_n, (step + 1) & 1);
    if (have_n) {
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float *src = land + (step & 1) * LAND;
    for (int i = 0; i < 4; ++i) acc += src[(tid * 4 + i * 2048) % LAND];
    __syncthreads();
    it = it_n; tile_id = tile_n; ++step;
  }
  if (acc == 123.456f) sink[tid] = acc;
}

// ---- B/C: loads + LDS transpose + stores.  MODE 0: frame DMA into per-wave landing (no request ahead: the landing
// buffer is requested at the top of the step, as stft_fft_kernel); MODE 1: span DMA, requested one tile ahead (two span
// buffers).  TWO_TILES: two tile buffers and one barrier per step, else one buffer and two barriers.
// A wave "transforms" a frame by reading its 16 sample pairs per lane and writing 17 values of the tile column
// (rows lane + 64 i and 1024 - lane - 64 i: the pattern of the real post-processing).
template <int FT, int MODE, int TWO_TILES>
__global__ void __launch_bounds__(512) full_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles_per_clip, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int C = FT + 2;  // floats per tile row
  constexpr int TILE = F * C;
  constexpr int SPAN = NFFT + (FT - 1) * HOP;
  constexpr int LAND = MODE == 0 ? 8 * NFFT : ((SPAN + 255) / 256) * 256;
  constexpr int NLAND = MODE == 0 ? 1 : 2;
  constexpr int FPW = FT / 8;
  float *tiles = reinterpret_cast<float *>(smem);
  float *land = tiles + (TWO_TILES ? 2 : 1) * TILE + ((TWO_TILES ? 2 : 1) * TILE % 4 ? 4 - (TWO_TILES ? 2 : 1) * TILE % 4 : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  auto request = [&](int tile_id, int bufi, int u) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    const float *xc = x + (long long)c * L;
    float *dst = land + bufi * LAND;
    if (MODE == 0) {
      long long pos = (long long)(t0 + wave * FPW + u) * HOP - PAD;
      pos = pos < 0 ? 0 : (pos + NFFT > L ? L - NFFT : pos);
#pragma unroll
      for (int j = 0; j < NFFT / 256; ++j) dma16(xc + pos + 256 * j + 4 * lane, lds_off(dst + wave * NFFT + 256 * j));
    } else {
      long long pos = (long long)t0 * HOP - PAD;
      pos = pos < 0 ? 0 : (pos + SPAN > L ? L - SPAN : pos);
      for (int j = wave; j < LAND / 256; j += 8) dma16(xc + pos + 256 * j + 4 * lane, lds_off(dst + 256 * j));
    }
  };
  auto flush = [&](const float *tile, int tile_id) {
    constexpr int LPR = FT / 4, RPI = 512 / LPR;
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    float *oc = out + (long long)c * F * T;
    const int fl = (tid % LPR) * 4, r0 = tid / LPR;
    if (t0 + fl + 4 <= T) {
#pragma unroll 4
      for (int k = r0; k < F; k += RPI) {
        const f2 *s = reinterpret_cast<const f2 *>(tile + k * C + fl);
        const f2 lo = s[0], hi = s[1];
        *reinterpret_cast<f4u *>(oc + (long long)k * T + t0 + fl) = f4{lo.x, lo.y, hi.x, hi.y};
      }
    }
  };
  auto wait_vm = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  constexpr int LPRs = FT / 4, RPIs = 512 / LPRs;
  const int n_flush = (F - wave * 64 / LPRs + RPIs - 1) / RPIs;  // store instructions of this wave per flush
  auto transform = [&](const float *src, float *tile, int f) {
    f2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f2 *>(src + 2 * (lane + 64 * i));
    for (int s = 0; s < spin; ++s) {  // fake arithmetic: `spin` dependent fmas per value
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[(lane + 64 * i) * C + f] = v[i].x;
      tile[(1024 - lane - 64 * i) * C + f] = v[i].y + v[i + 8].x;
    }
    if (lane == 0) tile[512 * C + f] = v[8].y;
  };
  int it = blockIdx.x >> 3, step = 0, prev_tile = -1;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  if (MODE == 1 && have) request(tile_id, 0, 0);
  while (have) {
    const int it_n = it + ((nwg + 7) >> 3), tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    float *tile = tiles + (TWO_TILES ? (step & 1) * TILE : 0);
    if (MODE == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's span (requested a step ago; the stores before it are older still)
      __syncthreads();                                  // ... of every wave; and the previous tile is complete / read out
      if (have_n) request(tile_n, (step + 1) & 1, 0);
      if (TWO_TILES && prev_tile >= 0) flush(tiles + ((step & 1) ^ 1) * TILE, prev_tile);
      for (int u = 0; u < FPW; ++u) transform(land + (step & 1) * LAND + (wave * FPW + u) * HOP, tile, wave * FPW + u);
    } else {
      for (int u = 0; u < FPW; ++u) {
        request(tile_id, 0, u);
        int younger = 0;
        if (u == 0 && prev_tile >= 0 && TWO_TILES) {
          flush(tiles + ((step & 1) ^ 1) * TILE, prev_tile);
          younger = n_flush;
        }
        wait_vm(younger);
        transform(land + wave * NFFT, tile, wave * FPW + u);
      }
      if (TWO_TILES) __syncthreads();
    }
    if (!TWO_TILES) {
      __syncthreads();
      flush(tile, tile_id);
      if (MODE == 0) __syncthreads();
    }
    prev_tile = tile_id;
    it = it_n; tile_id = tile_n; have = have_n; ++step;
  }
  if (TWO_TILES && prev_tile >= 0) {
    if (MODE == 1) __syncthreads();
    flush(tiles + ((step & 1) ^ 1) * TILE, prev_tile);
  }
}

// ---- K1: FT frames per tile (FT / 8 per wave), ONE tile buffer, ONE span buffer: a wave takes the samples of all its
// frames into registers, barrier, the next tile's span is requested into the same buffer and has the whole transform
// phase to land; the flush's stores are never waited for (counted vmcnt).  XCH: with an exchange-buffer round trip per
// frame (16 ds_write_b64 + 16 ds_read_b64, then 8 + 8) as the FFT has.  Three barriers per tile.
template <int FT, int XCH>
__global__ void __launch_bounds__(512) k1_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles_per_clip, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int C = FT + 2, TILE = F * C, SPAN = NFFT + (FT - 1) * HOP, LAND = ((SPAN + 255) / 256) * 256, FPW = FT / 8;
  constexpr int XB = 1024 + 64 + 1;  // f2 elements of an exchange buffer
  float *tile = reinterpret_cast<float *>(smem);
  float *land = tile + ((TILE + 3) & ~3);
  f2 *xbuf = reinterpret_cast<f2 *>(land + LAND);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 *const buf = xbuf + wave * XB;
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  constexpr int LPR = FT / 4, RPI = 512 / LPR;
  const int n_flush = (F - wave * 64 / LPR + RPI - 1) / RPI;
  auto request = [&](int tile_id) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    long long pos = (long long)t0 * HOP - PAD;
    pos = pos < 0 ? 0 : (pos + SPAN > L ? L - SPAN : pos);
    const float *xc = x + (long long)c * L + pos;
    for (int j = wave; j < LAND / 256; j += 8) dma16(xc + 256 * j + 4 * lane, lds_off(land + 256 * j));
  };
  auto wait_vm = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  int it = blockIdx.x >> 3;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  if (have) request(tile_id);
  int younger = 0;
  while (have) {
    const int it_n = it + ((nwg + 7) >> 3), tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    wait_vm(younger);   // the span; the flush's stores (younger) may still be travelling
    BAR();    // #1: span complete, tile read out
    f2 v[FPW][16];
#pragma unroll
    for (int u = 0; u < FPW; ++u) {
      const float *src = land + (wave * FPW + u) * HOP;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[u][i] = *reinterpret_cast<const f2 *>(src + 2 * (lane + 64 * i));
    }
    BAR();    // #2: span free
    if (have_n) request(tile_n);
#pragma unroll
    for (int u = 0; u < FPW; ++u) {
      const int f = wave * FPW + u;
      for (int s = 0; s < spin / 3; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[u][i] = v[u][i] * 1.0001f + 0.5f;
      }
      if (XCH) {
#pragma unroll
        for (int i = 0; i < 16; ++i) buf[17 * lane + i] = v[u][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 16; ++i) v[u][i] = buf[lane + 64 * i + ((lane + 64 * i) >> 4)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      for (int s = 0; s < spin - spin / 3; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[u][i] = v[u][i] * 1.0001f + 0.5f;
      }
      if (XCH) {
#pragma unroll
        for (int i = 8; i < 16; ++i) buf[17 * lane + i] = v[u][i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 8; i < 16; ++i) v[u][i] = v[u][i] + buf[1088 - lane - 68 * (i - 8)];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tile[(lane + 64 * i) * C + f] = v[u][i].x;
        tile[(1024 - lane - 64 * i) * C + f] = v[u][i].y + v[u][i + 8].x;
      }
      if (lane == 0) tile[512 * C + f] = v[u][8].y;
    }
    BAR();    // #3: tile complete
    {
      const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
      float *oc = out + (long long)c * F * T;
      const int fl = (tid % LPR) * 4, r0 = tid / LPR;
      if (t0 + fl + 4 <= T) {
#pragma unroll 4
        for (int k = r0; k < F; k += RPI) {
          const f2 *s = reinterpret_cast<const f2 *>(tile + k * C + fl);
          const f2 lo = s[0], hi = s[1];
          *reinterpret_cast<f4u *>(oc + (long long)k * T + t0 + fl) = f4{lo.x, lo.y, hi.x, hi.y};
        }
      }
      younger = n_flush;
    }
    it = it_n; tile_id = tile_n; have = have_n;
  }
}

template <int FT, int XCH>
void run_k1(const char *name, const float *x, float *out, int grid, int spin) {
  constexpr int C = FT + 2, TILE = F * C, SPAN = NFFT + (FT - 1) * HOP, LAND = ((SPAN + 255) / 256) * 256;
  size_t smem = ((size_t)((TILE + 3) & ~3) + LAND) * 4 + (size_t)8 * 1089 * 8;
  if (smem > 160 * 1024) { printf("%-60s skipped (%zu B of LDS)\n", name, smem); return; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k1_kernel<FT, XCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + FT - 1) / FT;
  float ms = time_it([&] { k1_kernel<FT, XCH><<<grid, 512, smem>>>(x, out, tpc, spin); });
  printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)  LDS %zu KB grid %d spin %d\n", name, ms, 339.0e6 / ms / 1e9, smem / 1024, grid, spin);
}

// ---- X1: NW-wave workgroups, several per CU (they drift apart: one computes while the other waits for memory), FT = 8
// frames per tile = 8 / NW per wave, one tile buffer, frames DMA'd into the wave's own exchange buffer: the first frame of
// the NEXT tile is requested before the flush, the later frames of a tile travel through registers a transform ahead.
template <int NW, int XCH>
__global__ void __launch_bounds__(NW * 64) x1_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles_per_clip, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int FT = 8, C = FT + 2, TILE = F * C, FPW = FT / NW, NT = NW * 64;
  constexpr int XB = 1024 + 64 + 1;
  float *tile = reinterpret_cast<float *>(smem);
  f2 *xbuf = reinterpret_cast<f2 *>(tile + ((TILE + 3) & ~3));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 *const buf = xbuf + wave * XB;
  const unsigned buf_lds = __builtin_amdgcn_readfirstlane(lds_off(buf));
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  constexpr int LPR = FT / 4, RPI = NT / LPR;
  const int n_flush = (F - wave * 64 / LPR + RPI - 1) / RPI;
  auto frame_pos = [&](int tile_id, int u, const float *&xc) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    long long pos = (long long)(t0 + wave * FPW + u) * HOP - PAD;
    pos = pos < 0 ? 0 : (pos + NFFT > L ? L - NFFT : pos);
    xc = x + (long long)c * L + pos;
  };
  auto request = [&](int tile_id) {
    const float *xc;
    frame_pos(tile_id, 0, xc);
#pragma unroll
    for (int j = 0; j < NFFT / 256; ++j) dma16(xc + 256 * j + 4 * lane, buf_lds + 1024 * j);
  };
  auto wait_vm = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
      case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
      case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
      case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
      case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  int it = blockIdx.x >> 3;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  if (have) request(tile_id);
  int younger = 0;
  while (have) {
    const int it_n = it + ((nwg + 7) >> 3), tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    f2 v[16], vn[16];
#pragma unroll 1
    for (int u = 0; u < FPW; ++u) {
      if (u == 0) {
        wait_vm(younger);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = buf[lane + 64 * i];
        wsync();
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = vn[i];
      }
      if (u + 1 < FPW) {
        const float *xc;
        frame_pos(tile_id, u + 1, xc);
        typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
        for (int i = 0; i < 16; ++i) { const f2u t = *reinterpret_cast<const f2u *>(xc + 2 * (lane + 64 * i)); vn[i] = f2{t.x, t.y}; }
      }
      const int f = wave * FPW + u;
      for (int s = 0; s < spin / 3; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
      }
      if (XCH) {
#pragma unroll
        for (int i = 0; i < 16; ++i) buf[17 * lane + i] = v[i];
        wsync();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = buf[lane + 64 * i + ((lane + 64 * i) >> 4)];
        wsync();
      }
      for (int s = 0; s < spin - spin / 3; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
      }
      if (XCH) {
#pragma unroll
        for (int i = 8; i < 16; ++i) buf[17 * lane + i] = v[i];
        wsync();
#pragma unroll
        for (int i = 8; i < 16; ++i) v[i] = v[i] + buf[1088 - lane - 68 * (i - 8)];
        wsync();
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tile[(lane + 64 * i) * C + f] = v[i].x;
        tile[(1024 - lane - 64 * i) * C + f] = v[i].y + v[i + 8].x;
      }
      if (lane == 0) tile[512 * C + f] = v[8].y;
    }
    if (have_n) request(tile_n);  // the next tile's first frame: travels under the flush (and the other workgroup's transforms)
    __syncthreads();
    {
      const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
      float *oc = out + (long long)c * F * T;
      const int fl = (tid % LPR) * 4, r0 = tid / LPR;
      if (t0 + fl + 4 <= T) {
#pragma unroll 4
        for (int k = r0; k < F; k += RPI) {
          const f2 *sp = reinterpret_cast<const f2 *>(tile + k * C + fl);
          const f2 lo = sp[0], hi = sp[1];
          *reinterpret_cast<f4u *>(oc + (long long)k * T + t0 + fl) = f4{lo.x, lo.y, hi.x, hi.y};
        }
      }
      younger = n_flush;
    }
    __syncthreads();
    it = it_n; tile_id = tile_n; have = have_n;
  }
}

template <int NW, int XCH>
void run_x1(const char *name, const float *x, float *out, int grid, int spin) {
  constexpr int TILE = F * 10;
  size_t smem = (size_t)((TILE + 3) & ~3) * 4 + (size_t)NW * 1089 * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&x1_kernel<NW, XCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + 7) / 8;
  float ms = time_it([&] { x1_kernel<NW, XCH><<<grid, NW * 64, smem>>>(x, out, tpc, spin); });
  printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)  LDS %zu KB grid %d spin %d\n", name, ms, 339.0e6 / ms / 1e9, smem / 1024, grid, spin);
}

// ---- K2: as K1 (FT = 8, one span buffer read into registers early) but TWO tile buffers and the memory traffic SPREAD over
// the transform: the previous tile's flush in chunks (one store instruction per wave each) and the next span's DMA pieces
// are issued between quarters of the arithmetic, so that the memory system sees a steady stream instead of a burst per
// step from 256 CUs in lock step.  Exchange traffic through a HALF-size buffer in two rounds (what fits the real kernel).
template <int XCH>
__global__ void __launch_bounds__(512) k2_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles_per_clip, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int FT = 8, C = FT + 2, TILE = F * C, SPAN = NFFT + (FT - 1) * HOP, LAND = ((SPAN + 255) / 256) * 256;
  constexpr int XB = 512 + 32 + 1;  // f2 elements of a half exchange buffer
  float *tiles = reinterpret_cast<float *>(smem);
  float *land = tiles + ((2 * TILE + 3) & ~3);
  f2 *xbuf = reinterpret_cast<f2 *>(land + LAND);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 *const buf = xbuf + wave * XB;
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  constexpr int LPR = FT / 4, RPI = 512 / LPR;   // 2 lanes per row, 256 rows per iteration
  const int n_flush = (F - wave * 64 / LPR + RPI - 1) / RPI;  // 5 for wave 0, else 4
  const int n_dma = (LAND / 256 - wave + 7) / 8;             // DMA pieces of this wave per span (3 or 2)
  auto wait_vm = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  const float *nx = nullptr;  // the next span in memory
  float *poc = nullptr;       // the previous tile in memory (row 0, first frame)
  const float *ptile = nullptr;
  bool p_ok = false;
  const int fl = (tid % LPR) * 4, r0 = tid / LPR;
  auto dma_piece = [&](int q) {  // piece q of this wave's share of the next span
    if (nx && q < n_dma) { const int j = wave + 8 * q; dma16(nx + 256 * j + 4 * lane, lds_off(land + 256 * j)); }
  };
  auto flush_chunk = [&](int q) {  // chunk q of the previous tile's flush: rows r0 + 256 q
    const int k = r0 + RPI * q;
    if (poc && p_ok && k < F) {
      const f2 *sp = reinterpret_cast<const f2 *>(ptile + k * C + fl);
      const f2 lo = sp[0], hi = sp[1];
      *reinterpret_cast<f4u *>(poc + (long long)k * T + fl) = f4{lo.x, lo.y, hi.x, hi.y};
    }
  };
  auto span_of = [&](int tile_id) -> const float * {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    long long pos = (long long)t0 * HOP - PAD;
    pos = pos < 0 ? 0 : (pos + SPAN > L ? L - SPAN : pos);
    return x + (long long)c * L + pos;
  };
  int it = blockIdx.x >> 3, step = 0;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  if (have) { nx = span_of(tile_id); for (int q = 0; q < 3; ++q) dma_piece(q); }
  int younger = 0;
  while (have) {
    const int it_n = it + ((nwg + 7) >> 3), tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    float *tile = tiles + (step & 1) * TILE;
    wait_vm(younger);
    __syncthreads();  // #1: the span has landed; the previous tile is complete; the one before it has been read out
    f2 v[16];
    {
      const float *src = land + wave * HOP;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f2 *>(src + 2 * (lane + 64 * i));
    }
    __syncthreads();  // #2: span free
    nx = have_n ? span_of(tile_n) : nullptr;
    const int f = wave;
    const int q4 = spin / 4;
    for (int s = 0; s < q4; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    dma_piece(0); flush_chunk(0);
    if (XCH) {  // pass 0 -> pass 1 exchange in two halves of 32 source lanes
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
#pragma unroll
          for (int i = 0; i < 16; ++i) buf[17 * (lane & 31) + i] = v[i];
        }
        wsync();
#pragma unroll
        for (int i = 0; i < 8; ++i) v[8 * h + i] = buf[(lane + 64 * i + ((lane + 64 * i) >> 4))];
        wsync();
      }
    }
    for (int s = 0; s < q4; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    dma_piece(1); flush_chunk(1);
    for (int s = 0; s < q4; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    dma_piece(2); flush_chunk(2);
    if (XCH) {
#pragma unroll
      for (int i = 8; i < 16; ++i) buf[lane + 64 * (i - 8) + ((lane + 64 * (i - 8)) >> 4)] = v[i];
      wsync();
#pragma unroll
      for (int i = 8; i < 16; ++i) v[i] = v[i] + buf[544 - lane - 68 * (i - 8) > 0 ? 544 - lane - 68 * (i - 8) : 0];
      wsync();
    }
    for (int s = 0; s < spin - 3 * q4; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    flush_chunk(3); flush_chunk(4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[(lane + 64 * i) * C + f] = v[i].x;
      tile[(1024 - lane - 64 * i) * C + f] = v[i].y + v[i + 8].x;
    }
    if (lane == 0) tile[512 * C + f] = v[8].y;
    // what is younger than the last DMA piece of this wave: the flush chunks issued after it
    {
      const int last = n_dma - 1;  // issued with chunk `last`; chunks last .. n_flush-1 follow it
      younger = (poc && p_ok) ? (n_flush - last > 0 ? n_flush - last : 0) : 0;
      if (!nx) younger = 0;
    }
    {
      const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
      poc = out + (long long)c * F * T + t0;
      p_ok = t0 + fl + 4 <= T;
      ptile = tile;
    }
    it = it_n; tile_id = tile_n; have = have_n; ++step;
  }
  __syncthreads();
  for (int q = 0; q < 5; ++q) flush_chunk(q);
}

template <int XCH>
void run_k2(const char *name, const float *x, float *out, int grid, int spin) {
  constexpr int TILE = F * 10, SPAN = NFFT + 7 * HOP, LAND = ((SPAN + 255) / 256) * 256;
  size_t smem = ((size_t)((2 * TILE + 3) & ~3) + LAND) * 4 + (size_t)8 * 545 * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k2_kernel<XCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + 7) / 8;
  float ms = time_it([&] { k2_kernel<XCH><<<grid, 512, smem>>>(x, out, tpc, spin); });
  printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)  LDS %zu KB grid %d spin %d\n", name, ms, 339.0e6 / ms / 1e9, smem / 1024, grid, spin);
}

// ---- V3: NO workgroup barrier.  Every wave walks the workgroup's tiles on its own: frame DMA'd into its exchange buffer
// (requested as soon as the buffer is free, before the epilogue), column written into tile buffer k & 1, then its share of the
// flush of tile k - 1; the waves meet only through two pairs of LDS counters (columns written / shares flushed per buffer),
// polled with s_sleep.  Waves drift apart by up to a tile: transforms of one run beside the loads / stores of another.
template <int XCH, int HALFX>
__global__ void __launch_bounds__(512) v3_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles_per_clip, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int FT = 8, C = FT + 2, TILE = F * C;
  constexpr int XB = 1024 + 64 + 1;
  float *tiles = reinterpret_cast<float *>(smem);
  f2 *xbuf = reinterpret_cast<f2 *>(tiles + ((2 * TILE + 3) & ~3));
  unsigned *cnt = reinterpret_cast<unsigned *>(xbuf + 8 * XB);  // [0..1] columns written, [2..3] shares flushed
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 *const buf = xbuf + wave * XB;
  const unsigned buf_lds = __builtin_amdgcn_readfirstlane(lds_off(buf));
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8, stride = (nwg + 7) >> 3;
  constexpr int LPR = FT / 4, RPI = 512 / LPR;
  const int n_flush = (F - wave * 64 / LPR + RPI - 1) / RPI;
  if (tid < 4) cnt[tid] = 0;
  __syncthreads();
  auto request = [&](int tile_id) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    long long pos = (long long)(t0 + wave) * HOP - PAD;
    pos = pos < 0 ? 0 : (pos + NFFT > L ? L - NFFT : pos);
    const float *xc = x + (long long)c * L + pos;
#pragma unroll
    for (int j = 0; j < NFFT / 256; ++j) dma16(xc + 256 * j + 4 * lane, buf_lds + 1024 * j);
  };
  auto wait_vm = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  auto poll = [&](int which, unsigned target) {  // until cnt[which] >= target
    volatile unsigned *p = cnt + which;
    while (true) {
      const unsigned v = __builtin_amdgcn_readfirstlane(*p);
      if (v >= target) break;
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };
  auto bump = [&](int which) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(cnt + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto flush_share = [&](int tile_id, const float *tile) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    float *oc = out + (long long)c * F * T;
    const int fl = (tid % LPR) * 4, r0 = tid / LPR;
    if (t0 + fl + 4 <= T) {
#pragma unroll 5
      for (int k = r0; k < F; k += RPI) {
        const f2 *sp = reinterpret_cast<const f2 *>(tile + k * C + fl);
        const f2 lo = sp[0], hi = sp[1];
        *reinterpret_cast<f4u *>(oc + (long long)k * T + t0 + fl) = f4{lo.x, lo.y, hi.x, hi.y};
      }
    }
  };
  int it = blockIdx.x >> 3, k = 0, prev_tile = -1;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  if (have) request(tile_id);
  int younger = 0;
  while (have) {
    const int it_n = it + stride, tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    const int p = k & 1;
    float *tile = tiles + p * TILE;
    wait_vm(younger);
    f2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = buf[lane + 64 * i];
    wsync();
    for (int s = 0; s < spin / 3; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    if (XCH) {
#pragma unroll
      for (int i = 0; i < 16; ++i) buf[17 * lane + i] = v[i];
      wsync();
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = buf[lane + 64 * i + ((lane + 64 * i) >> 4)];
      wsync();
    }
    for (int s = 0; s < spin / 3; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    if (XCH) {
#pragma unroll
      for (int i = 8; i < 16; ++i) buf[17 * lane + i] = v[i];
      wsync();
#pragma unroll
      for (int i = 8; i < 16; ++i) v[i] = v[i] + buf[1088 - lane - 68 * (i - 8)];
      wsync();
    }
    if (have_n) request(tile_n);  // the buffer is free: the next frame travels under the epilogue and the flush
    for (int s = 0; s < spin - 2 * (spin / 3); ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    poll(2 + p, 8u * (unsigned)(k >> 1));  // tile k - 2 (same buffer) has been read out by every wave
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[(lane + 64 * i) * C + wave] = v[i].x;
      tile[(1024 - lane - 64 * i) * C + wave] = v[i].y + v[i + 8].x;
    }
    if (lane == 0) tile[512 * C + wave] = v[8].y;
    bump(p);
    younger = 0;
    if (prev_tile >= 0) {
      poll(p ^ 1, 8u * (unsigned)(((k - 1) >> 1) + 1));  // every column of tile k - 1 is there
      flush_share(prev_tile, tiles + (p ^ 1) * TILE);
      bump(2 + (p ^ 1));   // (the release fence waits for the LDS reads)
      younger = n_flush;
    }
    if (!have_n) younger = 0;
    prev_tile = tile_id;
    it = it_n; tile_id = tile_n; have = have_n; ++k;
  }
  if (prev_tile >= 0) {
    const int p = (k - 1) & 1;
    poll(p, 8u * (unsigned)(((k - 1) >> 1) + 1));
    flush_share(prev_tile, tiles + p * TILE);
  }
}

template <int XCH>
void run_v3(const char *name, const float *x, float *out, int grid, int spin) {
  constexpr int TILE = F * 10;
  size_t smem = (size_t)((2 * TILE + 3) & ~3) * 4 + (size_t)8 * 1089 * 8 + 64;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&v3_kernel<XCH, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + 7) / 8;
  float ms = time_it([&] { v3_kernel<XCH, 0><<<grid, 512, smem>>>(x, out, tpc, spin); });
  printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)  LDS %zu KB grid %d spin %d\n", name, ms, 339.0e6 / ms / 1e9, smem / 1024, grid, spin);
}

// ---- K3: as K1 (one tile, three barriers per tile, samples read into registers early) with TWO span buffers, the span of tile
// s + 2 requested at the END of step s, BEFORE the flush's stores: a CU's memory pipe is a FIFO that its stores leave at the
// HBM write rate (~7 B/clk/CU), and a load issued behind a flush burst waits for it.  Loads are always a step and a half ahead
// and never behind a store of their own step.
template <int XCH>
__global__ void __launch_bounds__(512) k3_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles_per_clip, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int FT = 8, C = FT + 2, TILE = F * C, SPAN = NFFT + (FT - 1) * HOP, LAND = ((SPAN + 255) / 256) * 256;
  constexpr int XB = 1024 + 64 + 1;
  float *tile = reinterpret_cast<float *>(smem);
  float *land = tile + ((TILE + 3) & ~3);
  f2 *xbuf = reinterpret_cast<f2 *>(land + 2 * LAND);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 *const buf = xbuf + wave * XB;
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8, stride = (nwg + 7) >> 3;
  constexpr int LPR = FT / 4, RPI = 512 / LPR;
  const int n_flush = (F - wave * 64 / LPR + RPI - 1) / RPI;
  const int n_dma = (LAND / 256 - wave + 7) / 8;
  auto request = [&](int tile_id, int bufi) {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    long long pos = (long long)t0 * HOP - PAD;
    pos = pos < 0 ? 0 : (pos + SPAN > L ? L - SPAN : pos);
    const float *xc = x + (long long)c * L + pos;
    for (int j = wave; j < LAND / 256; j += 8) dma16(xc + 256 * j + 4 * lane, lds_off(land + bufi * LAND + 256 * j));
  };
  auto wait_vm = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  int it = blockIdx.x >> 3, step = 0;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  // prologue: spans of the first two tiles
  int younger = 0;   // operations of this wave younger than the span of the tile about to be transformed
  if (have) {
    request(tile_id, 0);
    const int it1 = it + stride, t1 = (blockIdx.x & 7) * per_xcd + it1;
    if (it1 < per_xcd && t1 < n_tiles) { request(t1, 1); younger = n_dma; }
  }
  while (have) {
    const int it_n = it + stride, tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    const int it_nn = it_n + stride, tile_nn = (blockIdx.x & 7) * per_xcd + it_nn;
    const bool have_nn = have_n && it_nn < per_xcd && tile_nn < n_tiles;
    wait_vm(younger);
    BAR();    // #1: span complete, tile read out
    f2 v[16];
    {
      const float *src = land + (step & 1) * LAND + wave * HOP;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f2 *>(src + 2 * (lane + 64 * i));
    }
    for (int s = 0; s < spin / 3; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    if (XCH) {
#pragma unroll
      for (int i = 0; i < 16; ++i) buf[17 * lane + i] = v[i];
      wsync();
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = buf[lane + 64 * i + ((lane + 64 * i) >> 4)];
      wsync();
    }
    for (int s = 0; s < spin - spin / 3; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    if (XCH) {
#pragma unroll
      for (int i = 8; i < 16; ++i) buf[17 * lane + i] = v[i];
      wsync();
#pragma unroll
      for (int i = 8; i < 16; ++i) v[i] = v[i] + buf[1088 - lane - 68 * (i - 8)];
      wsync();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[(lane + 64 * i) * C + wave] = v[i].x;
      tile[(1024 - lane - 64 * i) * C + wave] = v[i].y + v[i + 8].x;
    }
    if (lane == 0) tile[512 * C + wave] = v[8].y;
    BAR();    // #2: tile complete; every wave has taken its samples: this step's span buffer is free
    if (have_nn) request(tile_nn, step & 1);   // BEFORE the stores
    {
      const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
      float *oc = out + (long long)c * F * T;
      const int fl = (tid % LPR) * 4, r0 = tid / LPR;
      if (t0 + fl + 4 <= T) {
#pragma unroll 5
        for (int k = r0; k < F; k += RPI) {
          const f2 *sp = reinterpret_cast<const f2 *>(tile + k * C + fl);
          const f2 lo = sp[0], hi = sp[1];
          *reinterpret_cast<f4u *>(oc + (long long)k * T + t0 + fl) = f4{lo.x, lo.y, hi.x, hi.y};
        }
      }
    }
    // next step waits for span(s + 1): requested at the end of step s - 1; younger: flush(s - 1) [after it], span(s + 2), flush(s)
    younger = (step >= 1 ? n_flush : 0) + (have_nn ? n_dma : 0) + n_flush;
    if (step == 0) younger = (have_nn ? n_dma : 0) + n_flush;  // span(1) was requested in the prologue, nothing after it but these
    it = it_n; tile_id = tile_n; have = have_n; ++step;
  }
}

template <int XCH>
void run_k3(const char *name, const float *x, float *out, int grid, int spin) {
  constexpr int TILE = F * 10, SPAN = NFFT + 7 * HOP, LAND = ((SPAN + 255) / 256) * 256;
  size_t smem = ((size_t)((TILE + 3) & ~3) + 2 * LAND) * 4 + (size_t)8 * 1089 * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k3_kernel<XCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + 7) / 8;
  float ms = time_it([&] { k3_kernel<XCH><<<grid, 512, smem>>>(x, out, tpc, spin); });
  printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)  LDS %zu KB grid %d spin %d\n", name, ms, 339.0e6 / ms / 1e9, smem / 1024, grid, spin);
}

// ---- R: the structure of stft_fft_kernel<1024, 1> (two tiles, one barrier per step, one frame per wave) with the samples of
// the NEXT step's frame requested into REGISTERS at the top of the step (plain loads, before the flush's stores; the flush is
// straight-line code so that the compiler's own vmcnt bookkeeping stays exact): no LDS for landing, a full step of latency cover.
template <int XCH>
__global__ void __launch_bounds__(512) r_kernel(const float *__restrict__ x, float *__restrict__ out, int tiles_per_clip, int spin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int FT = 8, C = FT + 2, TILE = F * C;
  constexpr int XB = 1024 + 64 + 1;
  float *tiles = reinterpret_cast<float *>(smem);
  f2 *xbuf = reinterpret_cast<f2 *>(tiles + ((2 * TILE + 3) & ~3));
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f2 *const buf = xbuf + wave * XB;
  const int n_tiles = B * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8, stride = (nwg + 7) >> 3;
  typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
  auto frame_ptr = [&](int tile_id) -> const float * {
    const int c = tile_id / tiles_per_clip, t0 = (tile_id - c * tiles_per_clip) * FT;
    long long pos = (long long)(t0 + wave) * HOP - PAD;
    pos = pos < 0 ? 0 : (pos + NFFT > L ? L - NFFT : pos);
    return x + (long long)c * L + pos;
  };
  auto wsync = []() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  int it = blockIdx.x >> 3, step = 0, prev_tile = -1;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  f2 vn[16];
  if (have) {
    const float *xc = frame_ptr(tile_id);
#pragma unroll
    for (int i = 0; i < 16; ++i) { const f2u t = *reinterpret_cast<const f2u *>(xc + 2 * (lane + 64 * i)); vn[i] = f2{t.x, t.y}; }
  }
  const int fl = (tid & 1) * 4, r0 = tid >> 1;
  while (have) {
    const int it_n = it + stride, tile_n = (blockIdx.x & 7) * per_xcd + it_n;
    const bool have_n = it_n < per_xcd && tile_n < n_tiles;
    float *tile = tiles + (step & 1) * TILE;
    f2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = vn[i];
    {  // the next frame: a whole step ahead, before this step's stores
      const float *xc = frame_ptr(have_n ? tile_n : tile_id);
#pragma unroll
      for (int i = 0; i < 16; ++i) { const f2u t = *reinterpret_cast<const f2u *>(xc + 2 * (lane + 64 * i)); vn[i] = f2{t.x, t.y}; }
    }
    if (prev_tile >= 0) {  // flush of the previous tile: straight-line
      const float *pt = tiles + ((step & 1) ^ 1) * TILE;
      const int c = prev_tile / tiles_per_clip, t0 = (prev_tile - c * tiles_per_clip) * FT;
      float *oc = out + (long long)c * F * T + t0 + fl;
      if (t0 + fl + 4 <= T) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = r0 + 256 * q;
          const f2 *sp = reinterpret_cast<const f2 *>(pt + k * C + fl);
          const f2 lo = sp[0], hi = sp[1];
          *reinterpret_cast<f4u *>(oc + (long long)k * T) = f4{lo.x, lo.y, hi.x, hi.y};
        }
        if (r0 == 0) {
          const f2 *sp = reinterpret_cast<const f2 *>(pt + 1024 * C + fl);
          const f2 lo = sp[0], hi = sp[1];
          *reinterpret_cast<f4u *>(oc + (long long)1024 * T) = f4{lo.x, lo.y, hi.x, hi.y};
        }
      }
    }
    for (int s = 0; s < spin / 3; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    if (XCH) {
#pragma unroll
      for (int i = 0; i < 16; ++i) buf[17 * lane + i] = v[i];
      wsync();
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = buf[lane + 64 * i + ((lane + 64 * i) >> 4)];
      wsync();
    }
    for (int s = 0; s < spin - spin / 3; ++s) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + 0.5f;
    }
    if (XCH) {
#pragma unroll
      for (int i = 8; i < 16; ++i) buf[17 * lane + i] = v[i];
      wsync();
#pragma unroll
      for (int i = 8; i < 16; ++i) v[i] = v[i] + buf[1088 - lane - 68 * (i - 8)];
      wsync();
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[(lane + 64 * i) * C + wave] = v[i].x;
      tile[(1024 - lane - 64 * i) * C + wave] = v[i].y + v[i + 8].x;
    }
    if (lane == 0) tile[512 * C + wave] = v[8].y;
    BAR();
    prev_tile = tile_id;
    it = it_n; tile_id = tile_n; have = have_n; ++step;
  }
  if (prev_tile >= 0) {
    const float *pt = tiles + ((step & 1) ^ 1) * TILE;
    const int c = prev_tile / tiles_per_clip, t0 = (prev_tile - c * tiles_per_clip) * FT;
    float *oc = out + (long long)c * F * T + t0 + fl;
    if (t0 + fl + 4 <= T) {
      for (int k = r0; k < F; k += 256) {
        const f2 *sp = reinterpret_cast<const f2 *>(pt + k * C + fl);
        const f2 lo = sp[0], hi = sp[1];
        *reinterpret_cast<f4u *>(oc + (long long)k * T) = f4{lo.x, lo.y, hi.x, hi.y};
      }
    }
  }
}

template <int XCH>
void run_r(const char *name, const float *x, float *out, int grid, int spin) {
  constexpr int TILE = F * 10;
  size_t smem = (size_t)((2 * TILE + 3) & ~3) * 4 + (size_t)8 * 1089 * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&r_kernel<XCH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + 7) / 8;
  float ms = time_it([&] { r_kernel<XCH><<<grid, 512, smem>>>(x, out, tpc, spin); });
  printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)  LDS %zu KB grid %d spin %d\n", name, ms, 339.0e6 / ms / 1e9, smem / 1024, grid, spin);
}

template <typename Fn>
float time_it(Fn &&fn, int reps = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) fn();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

template <int FT, int MODE, int TWO>
void run_full(const char *name, const float *x, float *out, int grid, int spin = 0) {
  constexpr int C = FT + 2, TILE = F * C, SPAN = NFFT + (FT - 1) * HOP;
  constexpr int LAND = MODE == 0 ? 8 * NFFT : ((SPAN + 255) / 256) * 256;
  size_t smem = ((size_t)(TWO ? 2 : 1) * TILE + 4 + (size_t)(MODE == 0 ? 1 : 2) * LAND) * 4;
  if (smem > 160 * 1024) { printf("%-60s skipped (%zu B of LDS)\n", name, smem); return; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&full_kernel<FT, MODE, TWO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + FT - 1) / FT;
  float ms = time_it([&] { full_kernel<FT, MODE, TWO><<<grid, 512, smem>>>(x, out, tpc, spin); });
  printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)  LDS %zu KB grid %d spin %d\n", name, ms, 339.0e6 / ms / 1e9, smem / 1024, grid, spin);
}
template <int FT, int RUN>
void run_store(const char *name, float *out, int grid) {
  const int tpc = (T + FT - 1) / FT;
  float ms = time_it([&] { store_kernel<FT, RUN><<<grid, 512>>>(out, tpc); });
  printf("%-60s %.4f ms  (%.2f TB/s on 226 MB)\n", name, ms, 226.2e6 / ms / 1e9);
}
template <int FT, int MODE>
void run_load(const char *name, const float *x, float *sink, int grid) {
  constexpr int SPAN = NFFT + (FT - 1) * HOP;
  constexpr int LAND = MODE == 0 ? 8 * NFFT : ((SPAN + 255) / 256) * 256;
  size_t smem = (size_t)2 * LAND * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&load_kernel<FT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int tpc = (T + FT - 1) / FT;
  float ms = time_it([&] { load_kernel<FT, MODE><<<grid, 512, smem>>>(x, sink, tpc); });
  printf("%-60s %.4f ms  (%.2f TB/s on 113 MB)\n", name, ms, 112.9e6 / ms / 1e9);
}

int main() {
  float *x, *out, *sink;
  const long long n_in = (long long)B * L, n_out = (long long)B * F * T;
  CK(hipMalloc(&x, n_in * 4 + 4096)); CK(hipMalloc(&out, n_out * 4 + 4096)); CK(hipMalloc(&sink, 4096));
  std::vector<float> h(n_in);
  for (long long i = 0; i < n_in; ++i) h[i] = (float)((i * 2654435761u) & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(x, h.data(), n_in * 4, hipMemcpyHostToDevice));
  {
    float ms = time_it([&] { copy_kernel<<<2048, 512>>>(reinterpret_cast<const f4 *>(x), reinterpret_cast<f4 *>(out), n_in / 4, n_out / 4); });
    printf("%-60s %.4f ms  (%.2f TB/s on 339 MB)\n", "E  streaming copy 113 MB in + 226 MB out", ms, 339.0e6 / ms / 1e9);
  }
#ifdef RAW_BARRIER
  printf("barriers: s_waitcnt lgkmcnt(0); s_barrier\n");
#else
  printf("barriers: __syncthreads()\n");
#endif
  for (int spin : {0, 16, 33, 48}) {
    run_full<8, 0, 1>("A  frame loads at the top, two tiles, FT 8 (__syncthreads)", x, out, 256, spin);
    run_k1<8, 1>("K1 FT 8 + exchange traffic", x, out, 256, spin);
    run_k3<1>("K3 two spans, requested before the flush, + exchange", x, out, 256, spin);
    run_r<1>("R  next frame into registers a step ahead + exchange", x, out, 256, spin);
    run_v3<1>("V3 no barriers, LDS counters, + exchange traffic", x, out, 256, spin);
  }
  return 0;
}
