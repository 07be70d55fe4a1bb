// cycles per v_mfma_f32_16x16x4_f32 for the instruction patterns of the chain kernel (one wave per SIMD, 4 waves per CU)
//   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(3))) *lds4;

template <int N, int MODE>  // MODE 0: registers only; 1: + one ds_read_b128 per 4 MFMAs per tile (+1), prefetched; 2: accumulators renamed (d != c)
__global__ void __launch_bounds__(512) k(float *out, long long *cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  f32x4 acc[N];
  for (int m = 0; m < N; ++m) acc[m] = f32x4{0, 0, 0, 0};
  f32x4 a[2][N], b[2];
  for (int m = 0; m < N; ++m) a[0][m] = a[1][m] = f32x4{1.f + lane, 2.f, 3.f, 4.f};
  b[0] = b[1] = f32x4{1.f, 1.f, 1.f, 1.f};
  const unsigned base = (threadIdx.x >> 6) * 16384 + lane * 16;  // (8 waves x 16 KB <= 160 KB)
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (MODE == 1) {
        __builtin_amdgcn_sched_barrier(0);
        b[j ^ 1] = *(lds4)(base + 1024 * j);
#pragma unroll
        for (int m = 0; m < N; ++m) a[j ^ 1][m] = *(lds4)(base + 1024 * (j + m + 1));
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int m = 0; m < N; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][m][jj], b[j][jj], acc[m], 0, 0, 0);
          if (MODE == 2) {  // one fragment read of the next sub-stage behind each of the first N + 1 MFMAs
            const int idx = jj * N + m;
            __builtin_amdgcn_sched_barrier(0);
            if (idx == 0) b[j ^ 1] = *(lds4)(base + 1024 * j);
            if (idx >= 1 && idx <= N) a[j ^ 1][idx - 1] = *(lds4)(base + 1024 * (j + idx));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
  }
  const long long t1 = clock64();
  float s = 0;
  for (int m = 0; m < N; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int N, int MODE>
void run(const char *name, int grid, int threads = 256) {
  float *out;
  long long *cyc;
  (void)hipMalloc(&out, grid * 512 * 4);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipFuncSetAttribute((const void *)k<N, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<N, MODE><<<grid, threads, 160 * 1024>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<N, MODE><<<grid, threads, 160 * 1024>>>(out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 8 * N * (threads / 256);
  printf("%-34s N=%d grid=%4d: %.1f clock64 ticks / MFMA, %.2f ns / MFMA (%.1f cycles at 2.4 GHz)\n", name, N, grid, c / n, ms * 1e6 / n, ms * 1e6 / n * 2.4);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int grid : {1, 256}) {
    run<1, 0>("registers, 1 chain", grid);
    run<2, 0>("registers, 2 chains", grid);
    run<4, 0>("registers, 4 chains", grid);
    run<8, 0>("registers, 8 chains", grid);
    run<1, 1>("LDS fragments, 1 chain", grid);
    run<2, 1>("LDS fragments, 2 chains", grid);
    run<4, 1>("LDS fragments, 4 chains", grid);
    run<8, 1>("LDS fragments, 8 chains", grid);
    run<1, 2>("LDS fragments interleaved, 1 chain", grid);
    run<2, 2>("LDS fragments interleaved, 2 chains", grid);
    run<4, 2>("LDS fragments interleaved, 4 chains", grid);
    run<8, 2>("LDS fragments interleaved, 8 chains", grid);
    run<1, 1>("2 waves/SIMD: LDS burst, 1 chain each", grid, 512);
    run<2, 1>("2 waves/SIMD: LDS burst, 2 chains each", grid, 512);
    run<4, 1>("2 waves/SIMD: LDS burst, 4 chains each", grid, 512);
    run<1, 2>("2 waves/SIMD: interleaved, 1 chain each", grid, 512);
    run<2, 2>("2 waves/SIMD: interleaved, 2 chains each", grid, 512);
    run<4, 2>("2 waves/SIMD: interleaved, 4 chains each", grid, 512);
  }
  return 0;
}
