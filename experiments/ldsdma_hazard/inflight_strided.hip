// inflight.hip with the access pattern of the framed kernels: one LDS-DMA instruction moves 16 rows
// x 64 bytes (4 lanes per row, rows ROW_STRIDE bytes apart in global memory), N instructions per
// wave in flight, 8 waves.
//   hipcc --offload-arch=gfx950 -O2 inflight_strided.hip -o inflight_strided && ./inflight_strided
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
constexpr int ROW_STRIDE = 1024;  // bytes between consecutive rows (frames hop = 512 bf16)

template <int N>
__global__ void __launch_bounds__(512) probe(const unsigned char *src, unsigned *bad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row16 = lane >> 2, chunk = lane & 3;
  const unsigned char *base = src + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int piece = wave * N + i;  // 16 rows each
    __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)(piece * 16 + row16) * ROW_STRIDE + chunk * 16 + (i & 1) * 64),
                                     (lptr_t)(lds + piece * 1024), 16, 0, 0);
  }
  __syncthreads();
  int wrong = 0;
  for (int i = 0; i < N; ++i) {
    const int piece = wave * N + i;
    const uint4 got = *reinterpret_cast<const uint4 *>(lds + piece * 1024 + lane * 16);
    const uint4 want = *reinterpret_cast<const uint4 *>(base + (size_t)(piece * 16 + row16) * ROW_STRIDE + chunk * 16 + (i & 1) * 64);
    wrong += (got.x != want.x) + (got.y != want.y) + (got.z != want.z) + (got.w != want.w);
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int N>
void run(const unsigned char *d_src, unsigned *d_bad, int blocks) {
  hipMemset(d_bad, 0, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((probe<N>), dim3(blocks), dim3(512), 8 * N * 1024, 0, d_src, d_bad);
  unsigned b = 0;
  hipMemcpy(&b, d_bad, 4, hipMemcpyDeviceToHost);
  printf("%2d strided LDS-DMA loads in flight per wave, %4d workgroups: %u wrong dwords (%s)\n", N, blocks, b,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t n = (size_t)1024 * 4096 + (size_t)8 * 16 * 16 * ROW_STRIDE + 4096;
  std::vector<unsigned> h(n / 4);
  for (size_t i = 0; i < n / 4; ++i) h[i] = (unsigned)(i * 2654435761u);
  unsigned char *d_src;
  unsigned *d_bad;
  hipMalloc(&d_src, n);
  hipMalloc(&d_bad, 4);
  hipMemcpy(d_src, h.data(), n, hipMemcpyHostToDevice);
  for (int blocks : {1, 256, 1024}) {
    run<8>(d_src, d_bad, blocks);
    run<16>(d_src, d_bad, blocks);
  }
  return 0;
}
