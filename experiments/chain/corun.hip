// do LDS-DMA loads issue slower beside an MFMA + ds_read stream on the same SIMD?  8 waves: 0-3 multiply (N chains,
// fragments from LDS, one read per MFMA gap), 4-7 issue global_load_lds_dwordx4 / _dword with at most 4 in flight
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const volatile f32x4 __attribute__((address_space(3))) *lds4;
__device__ __forceinline__ void dma16s(const void *sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma4s(const void *sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
template <int N, int MODE>  // MODE bit 0: multiplying waves run; bit 1: loading waves run; bit 2: dword instead of dwordx4
__global__ void __launch_bounds__(512) k(const float *x, float *out, long long *cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) {
    if (MODE & 1) {
      f32x4 acc[N], a[2][N], b[2];
      for (int m = 0; m < N; ++m) acc[m] = f32x4{0, 0, 0, 0}, a[0][m] = a[1][m] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
      b[0] = b[1] = f32x4{1.f, 1.f, 1.f, 1.f};
      const unsigned base = wave * 16384 + lane * 16;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int t = 0; t < 4 * N; ++t) {
            acc[t % N] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][t % N][t / N], b[j][t / N], acc[t % N], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (!(MODE & 8)) {
              if (t == 0) b[j ^ 1] = *(lds4)(base + 1024 * j);
              if (t >= 1 && t <= N) a[j ^ 1][t - 1] = *(lds4)(base + 1024 * (j + t));
            }
            if ((MODE & 16) && t == 4 * N - 1 && (j == 0 || N > 1)) dma16s(x + wave * 65536, (unsigned)(((2 * i + j) & 63) * 1024 + lane * 16), 65536 + wave * 8192u + (unsigned)(((2 * i + j) & 7) * 1024));
            __builtin_amdgcn_sched_barrier(0);
          }
      }
      float s = 0;
      for (int m = 0; m < N; ++m) s += acc[m][0];
      out[threadIdx.x] = s;
    }
    if (threadIdx.x == 0) cyc[0] = clock64() - t0;
  } else {
    if (MODE & 2) {
      if (MODE & 32) __builtin_amdgcn_s_setprio(3);
      const unsigned wbase = 65536 + (wave - 4) * 8192u;
      const float *xs = x + (wave - 4) * 65536;
      const int n = iters * 8 * N * 32 / 250;  // about one DMA per 250 cycles of the multiplying waves' time
      for (int i = 0; i < n; ++i) {
        if (MODE & 4) dma4s(xs, (unsigned)((i & 255) * 256 + lane * 4), wbase + (unsigned)((i & 7) * 256));
        else dma16s(xs, (unsigned)((i & 63) * 1024 + lane * 16), wbase + (unsigned)((i & 7) * 1024));
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 256) {
        cyc[1] = clock64() - t0;
        cyc[2] = n;
      }
    }
  }
}
template <int N, int MODE>
void run(const char *name, const float *x) {
  float *out;
  long long *cyc, c[3] = {0, 0, 0};
  (void)hipMalloc(&out, 4096);
  (void)hipMalloc(&cyc, 24);
  (void)hipMemset(cyc, 0, 24);
  const int iters = 4000;
  (void)hipFuncSetAttribute((const void *)k<N, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<N, MODE><<<1, 512, 160 * 1024>>>(x, out, cyc, iters);
  k<N, MODE><<<1, 512, 160 * 1024>>>(x, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(c, cyc, 24, hipMemcpyDeviceToHost);
  printf("%-40s N=%d: %6.1f cycles per MFMA", name, N, (MODE & 1) ? (double)c[0] / (iters * 8.0 * N) : 0.0);
  if (MODE & 2) printf(" | loading wave: %6.1f cycles per DMA (%lld DMAs)", (double)c[1] / c[2], c[2]);
  printf("\n");
}
int main() {
  float *x;
  (void)hipMalloc(&x, 1 << 24);
  (void)hipMemset(x, 0, 1 << 24);
  run<1, 1>("multiply only", x);
  run<1, 2>("load only (dwordx4)", x);
  run<1, 6>("load only (dword)", x);
  run<1, 3>("both (dwordx4)", x);
  run<1, 7>("both (dword)", x);
  run<4, 1>("multiply only", x);
  run<4, 3>("both (dwordx4)", x);
  run<4, 7>("both (dword)", x);
  run<1, 11>("both, multiply without LDS reads", x);
  run<4, 11>("both, multiply without LDS reads", x);
  run<1, 17>("multiply + own DMA every 8 MFMAs", x);
  run<4, 17>("multiply + own DMA every 16 MFMAs", x);
  run<1, 35>("both, loading waves at priority 3", x);
  run<4, 35>("both, loading waves at priority 3", x);
  run<2, 1>("multiply only", x);
  run<2, 17>("multiply + own DMA every 8 MFMAs", x);
  return 0;
}
