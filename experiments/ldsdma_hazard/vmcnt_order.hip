// Do a wave's vector-memory LOADS and STORES retire in order on the one vmcnt counter of gfx950?
// The FFT and streaming-octave kernels (csrc/stft_fft.inl fft_wait_vm, csrc/octave_stream.hip wait_loads) issue
// LDS-direct loads, then K global stores, and wait with `s_waitcnt vmcnt(K)` before reading the loaded bytes
// from LDS: correct only if the K younger stores cannot be counted off before the older loads.  (That is what
// hipcc's own waitcnt insertion assumes for gfx9: SIInsertWaitcnts keeps loads and stores on one in-order VM_CNT
// event list; gfx10+ split the stores off into vscnt because there they do NOT stay ordered.)
//
// Probe: every wave issues one global_load_lds_dwordx4 from a COLD line (a fresh 1-KiB piece of a 1-GiB buffer per
// iteration: HBM latency), then K stores to a HOT line (its own 256 bytes: L2 hits, the fastest stores there
// are), waits vmcnt(W) and compares the LDS bytes with the expected pattern.
//   W = K      the kernels' wait: 0 stale reads expected if loads and stores retire in order
//   W = K + 1  control (one operation too many allowed outstanding -- the load itself may still be in flight):
//              stale reads MUST show up, otherwise the probe proves nothing
//   hipcc --offload-arch=gfx950 -O2 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int K, int W>
__global__ void __launch_bounds__(256) probe(const unsigned *cold, unsigned *hot, unsigned *stale, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4][256];
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)&lds[wave][0]);
  unsigned *myhot = hot + ((size_t)blockIdx.x * 4 + wave) * 64;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    // poison the destination, make sure the poison has landed
    *reinterpret_cast<uint4 *>(&lds[wave][lane * 4]) = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const size_t piece = ((size_t)it * gridDim.x * 4 + (size_t)blockIdx.x * 4 + wave) * 9973 % (1u << 20);  // 1-KiB pieces of 1 GiB
    const unsigned *src = cold + piece * 256 + lane * 4;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
#pragma unroll
    for (int k = 0; k < K; ++k)
      asm volatile("global_store_dword %0, %1, off" : : "v"(myhot + lane), "v"((unsigned)(it + k)) : "memory");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(W) : "memory");
    const uint4 got = *reinterpret_cast<const uint4 *>(&lds[wave][lane * 4]);
    const unsigned e0 = (unsigned)((piece * 256 + lane * 4) * 2654435761u);
    bad += got.x != e0;  // (cold[i] = i * 2654435761)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (bad) atomicAdd(stale, bad);
}

template <int K, int W>
void run(const unsigned *cold, unsigned *hot, unsigned *stale, const char *what) {
  hipMemset(stale, 0, 4);
  hipLaunchKernelGGL((probe<K, W>), dim3(1024), dim3(256), 0, 0, cold, hot, stale, 200);
  unsigned b = 0;
  hipMemcpy(&b, stale, 4, hipMemcpyDeviceToHost);
  printf("1 LDS-DMA load then %2d stores, s_waitcnt vmcnt(%2d) [%s]: %u stale reads of %d (%s)\n", K, W, what, b,
         1024 * 256 * 200, hipGetErrorString(hipGetLastError()));
}

__global__ void fill(unsigned *p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (unsigned)(i * 2654435761u);
}

int main() {
  const size_t n = (size_t)1 << 28;  // 1 GiB of dwords
  unsigned *cold, *hot, *stale;
  hipMalloc(&cold, n * 4);
  hipMalloc(&hot, 1024 * 4 * 64 * 4);
  hipMalloc(&stale, 4);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, cold, n);
  hipDeviceSynchronize();
  run<4, 4>(cold, hot, stale, "the kernels' wait");
  run<4, 5>(cold, hot, stale, "control: too weak");
  run<12, 12>(cold, hot, stale, "the kernels' wait");
  run<12, 13>(cold, hot, stale, "control: too weak");
  run<1, 1>(cold, hot, stale, "the kernels' wait");
  run<1, 2>(cold, hot, stale, "control: too weak");
  return 0;
}
