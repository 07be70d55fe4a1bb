#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_a(float a, float b, float s, unsigned &hi, unsigned &lo) {
  const f32x2 v = {a * s, b * s};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split_m(float a, float b, float s, unsigned &hi, unsigned &lo) {
  const f32x2 v = {a * s, b * s};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(s), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(s), "v"(hi));
  lo = l;
}
__global__ void k(const float *x, float s, unsigned *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h, l, h2, l2;
  split_a(x[2 * i], x[2 * i + 1], s, h, l);
  split_m(x[2 * i], x[2 * i + 1], s, h2, l2);
  o[4 * i] = h; o[4 * i + 1] = l; o[4 * i + 2] = h2; o[4 * i + 3] = l2;
}
int main() {
  const int n = 1 << 20;
  std::vector<float> x(2 * n);
  unsigned seed = 12345;
  for (auto &v : x) { seed = seed * 1664525u + 1013904223u; unsigned b = seed; float f; 
    // random exponents in a wide range + random mantissas, plus specials
    b = (b & 0x807fffffu) | ((100u + (seed >> 9) % 50u) << 23); memcpy(&f, &b, 4); v = f; }
  x[0] = 0.f; x[1] = -0.f; x[2] = INFINITY; x[3] = NAN; x[4] = 65504.f; x[5] = 1e-8f; x[6] = 70000.f; x[7] = -70000.f;
  float *dx; unsigned *dout; hipMalloc(&dx, 8 * n); hipMalloc(&dout, 16 * n);
  hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
  for (float s : {1.0f, 0.25f, 1024.f, 3.0517578125e-05f}) {
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, s, dout, n);
    std::vector<unsigned> o(4 * n); hipMemcpy(o.data(), dout, 16 * n, hipMemcpyDeviceToHost);
    long bad = 0, badnan = 0;
    for (int i = 0; i < n; ++i) { if (o[4*i] != o[4*i+2]) ++bad; if (o[4*i+1] != o[4*i+3]) { ++bad; if (i < 4) ++badnan; } }
    printf("scale %g: %ld mismatching words of %d pairs (first 4 pairs hold specials: %ld of them)\n", s, bad, n, badnan);
  }
  return 0;
}
