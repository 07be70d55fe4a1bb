// latency / issue rate of LDS-DMA loads as the chain kernel's loading waves use them (one wave, L2-resident source)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void dma4(const void *src, unsigned lds) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma4s(const void *sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma16s(const void *sbase, unsigned voff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds) : "memory", "m0");
}
// MODE 0: dword, lanes in order; 1: dword, the chain kernel's permutation inside groups of 16; 2: dwordx4 in order
// DEP 1: wait for every load (latency); 0: 8 in flight (issue rate)
template <int MODE, int DEP>
__global__ void __launch_bounds__(512) k(const float *x, long long *cyc, int iters) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wbase = wv * 8192u;
  x += wv * 65536;
  const int lperm = MODE == 1 ? 16 * (lane >> 4) + 4 * (lane & 3) + ((lane >> 2) & 3) : lane;
  // warm
  for (int i = 0; i < iters; ++i) {
    if (MODE == 2) dma16s(x, (unsigned)(i * 1024 + lane * 16), wbase); else dma4s(x, (unsigned)(i * 256 + lperm * 4), wbase);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 2) dma16s(x, (unsigned)((i & 63) * 1024 + lane * 16), wbase + (unsigned)((i & 7) * 1024)); else dma4s(x, (unsigned)((i & 255) * 256 + lperm * 4), wbase + (unsigned)((i & 7) * 256));
    if (DEP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int DEP>
void run(const char *name, const float *x, int threads = 64) {
  long long *cyc, c;
  (void)hipMalloc(&cyc, 8);
  const int iters = 512;
  k<MODE, DEP><<<1, threads, 65536>>>(x, cyc, iters);
  k<MODE, DEP><<<1, threads, 65536>>>(x, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-52s %d waves: %8.1f cycles per instruction per wave = %.1f B/clk/CU\n", name, threads / 64, (double)c / iters, (MODE == 2 ? 1024.0 : 256.0) * (threads / 64) * iters / (double)c);
  (void)hipFree(cyc);
}
int main() {
  float *x;
  (void)hipMalloc(&x, 1 << 24);
  (void)hipMemset(x, 0, 1 << 24);
  run<0, 1>("dword, lanes in order, one at a time (latency)", x);
  run<1, 1>("dword, permuted inside 16s, one at a time (latency)", x);
  run<2, 1>("dwordx4, one at a time (latency)", x);
  run<0, 0>("dword, lanes in order, 8 in flight", x);
  run<1, 0>("dword, permuted inside 16s, 8 in flight", x);
  run<2, 0>("dwordx4, 8 in flight", x);
  for (int t : {128, 256, 512}) {
    run<2, 0>("dwordx4, 8 in flight", x, t);
    run<1, 0>("dword permuted, 8 in flight", x, t);
  }
  return 0;
}
