// Follow-up to hazard.hip: the same address-VGPR overwrite, but behind a VMEM queue that is already
// busy: every wave of a 512-thread workgroup first issues PRE LDS-DMA loads (distinct address
// registers), then one more whose address VGPR pair is advanced by 4 KiB right after issue.
//   hipcc --offload-arch=gfx950 -O2 hazard2.hip -o hazard2 && ./hazard2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int PRE, int NOPS>
__global__ void __launch_bounds__(512) probe(const unsigned *src, unsigned *bad, unsigned *from_next) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned *base = src + (size_t)blockIdx.x * 8 * 32 * 256 + (size_t)wave * 32 * 256;
#pragma unroll
  for (int i = 0; i < PRE; ++i)
    __builtin_amdgcn_global_load_lds((gptr_t)(base + (i + 1) * 256 + lane * 4),
                                     (lptr_t)(lds + (wave * (PRE + 1) + i + 1) * 1024), 16, 0, 0);
  const unsigned *p = base + lane * 4;
  unsigned lds_dst = (unsigned)(size_t)(lds + wave * (PRE + 1) * 1024);
  asm volatile(
      "s_mov_b32 m0, %1\n"
      "s_nop 1\n"
      "global_load_lds_dwordx4 %0, off\n"
      ".rept %2\n s_nop 0\n .endr\n"
      "v_lshl_add_u64 %0, %0, 0, %3\n"
      : "+v"(p)
      : "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "n"(NOPS), "s"(4096ull)
      : "memory");
  __syncthreads();
  const unsigned *l = reinterpret_cast<const unsigned *>(lds) + wave * (PRE + 1) * 256;
  int wrong = 0, nxt = 0;
  for (int i = lane; i < 256; i += 64) {
    wrong += l[i] != base[i];
    nxt += l[i] == base[i + 1024];
  }
  if (wrong) atomicAdd(bad, wrong);
  if (nxt) atomicAdd(from_next, nxt);
  if (threadIdx.x == 0 && p == nullptr) bad[1] = 1;
}

template <int PRE, int NOPS>
void run(const unsigned *d_src, unsigned *d_bad, int blocks) {
  hipMemset(d_bad, 0, 12);
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe<PRE, NOPS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      160 * 1024);
  hipLaunchKernelGGL((probe<PRE, NOPS>), dim3(blocks), dim3(512), 8 * (PRE + 1) * 1024, 0, d_src, d_bad, d_bad + 2);
  unsigned b[3] = {0, 0, 0};
  hipMemcpy(b, d_bad, 12, hipMemcpyDeviceToHost);
  printf("%2d loads queued ahead, %2d nops before the overwrite, %4d workgroups: %6u wrong dwords, %6u equal to src+4096 (%s)\n",
         PRE, NOPS, blocks, b[0], b[2], hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t n = (size_t)1024 * 8 * 32 * 256 + 4096;
  std::vector<unsigned> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned)(i * 2654435761u);
  unsigned *d_src, *d_bad;
  hipMalloc(&d_src, n * 4);
  hipMalloc(&d_bad, 12);
  hipMemcpy(d_src, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int blocks : {1, 1024}) {
    run<0, 0>(d_src, d_bad, blocks);
    run<4, 0>(d_src, d_bad, blocks);
    run<8, 0>(d_src, d_bad, blocks);
    run<15, 0>(d_src, d_bad, blocks);
    run<8, 4>(d_src, d_bad, blocks);
    run<8, 16>(d_src, d_bad, blocks);
    run<8, 64>(d_src, d_bad, blocks);
    run<15, 64>(d_src, d_bad, blocks);
  }
  return 0;
}
