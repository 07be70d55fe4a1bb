// Does gfx950's global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) keep reading its address VGPRs
// after issue?  One wave: lane l loads 16 bytes from src + 16*l into LDS, and the address VGPR
// pair is advanced by 4 KiB by a VALU instruction placed NOPS wait states after the load.  If the
// DMA re-reads its address registers on later passes, part of the data comes from src + 4096.
//   hipcc --offload-arch=gfx950 -O2 hazard.hip -o hazard && ./hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NOPS>
__global__ void probe(const unsigned *src, unsigned *out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
  const unsigned *p = src + 4 * threadIdx.x;
  unsigned lds_base = (unsigned)(size_t)lds;  // LDS byte address (low 32 bits of the generic ptr)
  asm volatile(
      "s_mov_b32 m0, %1\n"
      "s_nop 1\n"
      "global_load_lds_dwordx4 %0, off\n"
      ".rept %2\n s_nop 0\n .endr\n"
      "v_lshl_add_u64 %0, %0, 0, %3\n"
      "s_waitcnt vmcnt(0)\n"
      : "+v"(p)
      : "s"(__builtin_amdgcn_readfirstlane(lds_base)), "n"(NOPS), "s"(4096ull)
      : "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[4 * threadIdx.x + i] = lds[4 * threadIdx.x + i];
  if (threadIdx.x == 0) out[256] = (unsigned)(size_t)p;  // keep p alive
}

template <int NOPS>
int run(const unsigned *d_src, unsigned *d_out, const std::vector<unsigned> &h) {
  hipMemset(d_out, 0xff, 257 * 4);
  hipLaunchKernelGGL(probe<NOPS>, dim3(1), dim3(64), 0, 0, d_src, d_out);
  std::vector<unsigned> o(257);
  hipMemcpy(o.data(), d_out, 257 * 4, hipMemcpyDeviceToHost);
  int bad = 0, from_next = 0;
  for (int i = 0; i < 256; ++i) {
    if (o[i] != h[i]) ++bad;
    if (o[i] == h[i + 1024]) ++from_next;
  }
  printf("nops %2d: %3d of 256 dwords wrong, %3d of them equal to src+4096\n", NOPS, bad, from_next);
  return bad;
}

int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 0x1000000u + i;
  unsigned *d_src, *d_out;
  hipMalloc(&d_src, h.size() * 4);
  hipMalloc(&d_out, 257 * 4);
  hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0>(d_src, d_out, h);
  run<1>(d_src, d_out, h);
  run<2>(d_src, d_out, h);
  run<3>(d_src, d_out, h);
  run<4>(d_src, d_out, h);
  run<6>(d_src, d_out, h);
  run<8>(d_src, d_out, h);
  run<12>(d_src, d_out, h);
  run<16>(d_src, d_out, h);
  run<24>(d_src, d_out, h);
  run<32>(d_src, d_out, h);
  return 0;
}
