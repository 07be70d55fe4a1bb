// How do v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 round?  D = C + sum_k A[i][k] B[k][j] on random operands whose
// products have widely spread exponents, compared bit for bit with host models: an fp32 FMA chain through k ascending /
// descending, one rounding of the exact sum ("fused"), and products rounded to fp32 before an fp32 add chain.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_order.hip -o mfma_order
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k32(const float *A, const float *B, const float *C, float *D) {  // A[32][2], B[2][32], C/D[32][32]
  const int l = threadIdx.x, li = l & 31, lh = l >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = C[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + li];
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * 2 + lh], B[lh * 32 + li], acc, 0, 0, 0);
  for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + li] = acc[e];
}
__global__ void k16(const float *A, const float *B, const float *C, float *D) {  // A[16][4], B[4][16], C/D[16][16]
  const int l = threadIdx.x, l16 = l & 15, lq = l >> 4;
  f32x4 acc;
  for (int e = 0; e < 4; ++e) acc[e] = C[(4 * lq + e) * 16 + l16];
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[l16 * 4 + lq], B[lq * 16 + l16], acc, 0, 0, 0);
  for (int e = 0; e < 4; ++e) D[(4 * lq + e) * 16 + l16] = acc[e];
}

static float rnd(int spread) {
  const float m = 1.f + (float)rand() / RAND_MAX;
  const int e = rand() % (2 * spread + 1) - spread;
  return (rand() & 1 ? -1.f : 1.f) * ldexpf(m, e);
}

template <int N, int K>
void run(const char *name, void (*kern)(const float *, const float *, const float *, float *)) {
  long n = 0, up = 0, down = 0, fused = 0, prod = 0, pairs = 0;
  float *dA, *dB, *dC, *dD;
  hipMalloc(&dA, N * K * 4); hipMalloc(&dB, N * K * 4); hipMalloc(&dC, N * N * 4); hipMalloc(&dD, N * N * 4);
  std::vector<float> A(N * K), B(N * K), C(N * N), D(N * N);
  for (int trial = 0; trial < 200; ++trial) {
    for (auto &v : A) v = rnd(6);
    for (auto &v : B) v = rnd(6);
    for (auto &v : C) v = rnd(8);
    hipMemcpy(dA, A.data(), N * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), N * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), N * N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, N * N * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const float c = C[i * N + j], d = D[i * N + j];
        float u = c, dn = c, pr = c;
        long double ex = c;
        for (int k = 0; k < K; ++k) {
          u = fmaf(A[i * K + k], B[k * N + j], u);
          dn = fmaf(A[i * K + K - 1 - k], B[(K - 1 - k) * N + j], dn);
          pr = pr + A[i * K + k] * B[k * N + j];
          ex += (long double)A[i * K + k] * B[k * N + j];
        }
        float pw = c;  // pairs: (p0 + p1) exactly, then + (p2 + p3) ...
        for (int k = 0; k < K; k += 2) pw = (float)((double)pw + ((double)A[i * K + k] * B[k * N + j] + (double)A[i * K + k + 1] * B[(k + 1) * N + j]));
        ++n;
        up += d == u; down += d == dn; fused += d == (float)ex; prod += d == pr; pairs += d == pw;
      }
  }
  printf("%s: %ld results | FMA chain k ascending %.4f | k descending %.4f | one rounding of the exact sum %.4f | "
         "rounded products, add chain %.4f | exact pairs, chain of pairs %.4f\n",
         name, n, (double)up / n, (double)down / n, (double)fused / n, (double)prod / n, (double)pairs / n);
}

int main() {
  srand(1);
  run<32, 2>("v_mfma_f32_32x32x2_f32", k32);
  run<16, 4>("v_mfma_f32_16x16x4_f32", k16);
  return 0;
}
