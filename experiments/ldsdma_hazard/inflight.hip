// How many global_load_lds_dwordx4 can one wave / one workgroup keep in flight?  512 threads; wave w
// issues N back-to-back 1-KiB LDS-DMA loads into its own LDS region (N KiB per wave), waits for
// vmcnt(0), barrier, and every thread checks the LDS contents against global memory.
//   hipcc --offload-arch=gfx950 -O2 inflight.hip -o inflight && ./inflight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int N>
__global__ void __launch_bounds__(512) probe(const unsigned *src, unsigned *bad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned *base = src + (size_t)blockIdx.x * 8 * N * 256 + (size_t)wave * N * 256;
#pragma unroll
  for (int i = 0; i < N; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(base + i * 256 + lane * 4),
                                     (lptr_t)(lds + (wave * N + i) * 1024), 16, 0, 0);
  __syncthreads();
  int wrong = 0;
  const unsigned *l = reinterpret_cast<const unsigned *>(lds) + wave * N * 256;
  for (int i = lane; i < N * 256; i += 64) wrong += l[i] != base[i];
  if (wrong) atomicAdd(bad, wrong);
}

template <int N>
void run(const unsigned *d_src, unsigned *d_bad, int blocks) {
  hipMemset(d_bad, 0, 4);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(probe<N>, dim3(blocks), dim3(512), 8 * N * 1024, 0, d_src, d_bad);
  unsigned b = 0;
  hipMemcpy(&b, d_bad, 4, hipMemcpyDeviceToHost);
  printf("%2d LDS-DMA loads in flight per wave (%3d KiB per workgroup), %d workgroups: %u wrong dwords (%s)\n", N,
         8 * N, blocks, b, hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t n = (size_t)1024 * 8 * 16 * 256;
  std::vector<unsigned> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned)(i * 2654435761u);
  unsigned *d_src, *d_bad;
  hipMalloc(&d_src, n * 4);
  hipMalloc(&d_bad, 4);
  hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int blocks : {1, 256, 1024}) {
    run<4>(d_src, d_bad, blocks);
    run<8>(d_src, d_bad, blocks);
    run<12>(d_src, d_bad, blocks);
    run<16>(d_src, d_bad, blocks);
  }
  return 0;
}
