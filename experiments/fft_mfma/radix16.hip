// "one honest experiment with the idle matrix pipe: the first radix-16 pass as a constant 16 x 16
// complex DFT fragment on v_mfma_f32_16x16x4_f32 -- record the number either way".
//
// The first pass of the 1024-point complex transform of stft_fft.inl is 64 DFTs of 16 points per frame (lane m: the points
// z[m + 64 j], j = 0 .. 15).  Two stand-alone versions of that pass, LDS -> registers -> LDS, one frame per wave and pass:
//   VALU: each lane reads its 16 complex points (32 ds_read_b32... here 8 ds_read_b128 of a [lane][32] layout), runs a
//         radix-4 x radix-4 DFT-16 in registers, writes 16 complex points back.
//   MFMA: the same DFTs as the real product [Wr -Wi; Wi Wr] (32 x 32) x [Xr; Xi] (32 x 64): 2 row tiles x 4 column groups x
//         8 k-slices = 64 v_mfma_f32_16x16x4_f32 (exact fp32), B fragments from LDS in [column][k group][k slice] order
//         (2 ds_read_b128 per column group), D written back from the accumulator layout.
// Reported per frame: cycles of the wave (s_memtime), for 1 / 2 / 4 waves per SIMD, and the maximum difference of the two.
//   hipcc --offload-arch=gfx950 -O3 radix16.hip -o radix16 && ./radix16
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct cf {
  float re, im;
};
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf mul_mi(cf a) { return {a.im, -a.re}; }  // * (-i)

// forward DFT-4 of (a0..a3), in place
__device__ __forceinline__ void dft4(cf &a0, cf &a1, cf &a2, cf &a3) {
  const cf s0 = cadd(a0, a2), d0 = csub(a0, a2), s1 = cadd(a1, a3), d1 = mul_mi(csub(a1, a3));
  a0 = cadd(s0, s1);
  a2 = csub(s0, s1);
  a1 = cadd(d0, d1);
  a3 = csub(d0, d1);
}

// X[k] = sum_j x[j] w^(jk), w = exp(-2 pi i / 16): radix 4 x 4 (j = 4 j1 + j0, k = k0 + 4 k1)
__device__ __forceinline__ void dft16(cf (&x)[16], const cf (&tw)[16]) {
#pragma unroll
  for (int j0 = 0; j0 < 4; ++j0) dft4(x[j0], x[4 + j0], x[8 + j0], x[12 + j0]);  // over j1 -> index k0 at position 4 k0 + j0
#pragma unroll
  for (int k0 = 1; k0 < 4; ++k0)
#pragma unroll
    for (int j0 = 1; j0 < 4; ++j0) x[4 * k0 + j0] = cmul(x[4 * k0 + j0], tw[k0 * j0]);
#pragma unroll
  for (int k0 = 0; k0 < 4; ++k0) dft4(x[4 * k0], x[4 * k0 + 1], x[4 * k0 + 2], x[4 * k0 + 3]);  // over j0 -> k1: X[k0 + 4 k1] at 4 k0 + k1
}

// LDS layouts (floats, per wave 64 x 32 = 8 KB in, 8 KB out), both free of bank conflicts:
//   VALU : [j][lane m][re, im] in, [k][lane][re, im] out (ds_read_b64 / ds_write_b64)
//   MFMA : [column group cg][half of the k slices][lane (k group, column)][4] in, [cg][re / im][lane][4 rows] out (b128)
// (the pass before -- the window multiply and the load from the frame -- writes whichever layout the pass wants)
template <int MODE>
__global__ void __launch_bounds__(256) pass_kernel(const float *__restrict__ in, float *__restrict__ out, long long *__restrict__ cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *const buf = smem + wave * 4096;  // 2048 floats in, 2048 out
  for (int i = lane; i < 2048; i += 64) buf[i] = in[i];
  __syncthreads();
  cf tw[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) tw[t] = {cosf(-2.f * 3.14159265358979f * t / 16.f), sinf(-2.f * 3.14159265358979f * t / 16.f)};
  // MFMA A fragments: A[row][k], row tile r (0, 1), k slice ks: lane (row = l & 15, k = 4 ks + (l >> 4))
  float afrag[2][8];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int row = 16 * r + (lane & 15), k = 4 * ks + (lane >> 4);
      const int kk = row & 15, j = k & 15;
      const float c = cosf(-2.f * 3.14159265358979f * (float)((kk * j) & 15) / 16.f), s = sinf(-2.f * 3.14159265358979f * (float)((kk * j) & 15) / 16.f);
      // [Wr -Wi; Wi Wr]: rows < 16 give the real parts, columns < 16 multiply the real inputs
      afrag[r][ks] = (row < 16) ? (k < 16 ? c : -s) : (k < 16 ? s : c);
    }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      cf x[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(buf + (j * 64 + lane) * 2);
        x[j] = {v[0], v[1]};
      }
      dft16(x, tw);
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) {  // position 4 k0 + k1 holds X[k0 + 4 k1]
        const int k = (pos >> 2) + 4 * (pos & 3);
        *reinterpret_cast<f32x2 *>(buf + 2048 + (k * 64 + lane) * 2) = f32x2{x[pos].re, x[pos].im};
      }
    } else {
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) {
        // B fragments: lane (column c = l & 15, k group kg = l >> 4) holds k = 4 ks + kg, ks = 0 .. 7: [cg][half][lane][4]
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(buf + ((cg * 2 + 0) * 64 + lane) * 4);
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(buf + ((cg * 2 + 1) * 64 + lane) * 4);
        f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const float b = ks < 4 ? b0[ks] : b1[ks - 4];
          d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[0][ks], b, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[1][ks], b, d1, 0, 0, 0);
        }
        // D: lane (column c, rows 4 kg + e); row tile 0 = Re X[k], 1 = Im X[k]: [cg][r][lane][4]
        *reinterpret_cast<f32x4 *>(buf + 2048 + ((cg * 2 + 0) * 64 + lane) * 4) = d0;
        *reinterpret_cast<f32x4 *>(buf + 2048 + ((cg * 2 + 1) * 64 + lane) * 4) = d1;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  __syncthreads();
  if (blockIdx.x == 0 && wave == 0)
    for (int i = lane; i < 2048; i += 64) out[i] = buf[2048 + i];
}

int main() {
  const int iters = 2000;
  std::vector<float> z(2048), valu_in(2048), mfma_in(2048);
  for (int i = 0; i < 2048; ++i) z[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;  // z[(m, j, re/im)]
  for (int m = 0; m < 64; ++m)
    for (int j = 0; j < 16; ++j)
      for (int e = 0; e < 2; ++e) {
        const float v = z[(m * 16 + j) * 2 + e];
        valu_in[(j * 64 + m) * 2 + e] = v;
        const int k = j + 16 * e, ks = k >> 2, kg = k & 3, cg = m >> 4, c = m & 15, l = kg * 16 + c;
        mfma_in[((cg * 2 + (ks >> 2)) * 64 + l) * 4 + (ks & 3)] = v;
      }
  float *din, *dout;
  long long *dc;
  hipMalloc(&din, 8192);
  hipMalloc(&dout, 8192);
  hipMalloc(&dc, 1024 * 4 * 8);
  std::vector<float> res[2];
  for (int mode = 0; mode < 2; ++mode) {
    hipMemcpy(din, mode ? mfma_in.data() : valu_in.data(), 8192, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; wps *= 2) {  // waves per SIMD: 4 waves per workgroup = one per SIMD; wps workgroups per CU (64 KB of LDS each: two fit)
      const int grid = 256 * wps;
      auto k = mode ? pass_kernel<1> : pass_kernel<0>;
      hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 4 * 4096 * 4, 0, din, dout, dc, iters);
      hipDeviceSynchronize();
      std::vector<long long> c(grid * 4);
      hipMemcpy(c.data(), dc, grid * 4 * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (long long v : c) s += (double)v;
      printf("%s pass, %d wave(s) per SIMD: %7.1f cycles per frame and wave -> %7.1f SIMD cycles per frame\n", mode ? "MFMA" : "VALU", wps,
             s / c.size() / iters, s / c.size() / iters / wps);
    }
    res[mode].resize(2048);
    hipMemcpy(res[mode].data(), dout, 8192, hipMemcpyDeviceToHost);
  }
  // reference in double
  double worst[2] = {0, 0}, peak = 0;
  for (int m = 0; m < 64; ++m)
    for (int k = 0; k < 16; ++k) {
      double re = 0, im = 0;
      for (int j = 0; j < 16; ++j) {
        const double a = -2.0 * M_PI * ((j * k) & 15) / 16.0, xr = z[(m * 16 + j) * 2], xi = z[(m * 16 + j) * 2 + 1];
        re += xr * cos(a) - xi * sin(a);
        im += xr * sin(a) + xi * cos(a);
      }
      peak = fmax(peak, fmax(fabs(re), fabs(im)));
      {
        worst[0] = fmax(worst[0], fmax(fabs(res[0][(k * 64 + m) * 2] - re), fabs(res[0][(k * 64 + m) * 2 + 1] - im)));
        const int cg = m >> 4, c = m & 15, kg = k >> 2, e = k & 3, l = kg * 16 + c;
        worst[1] = fmax(worst[1], fmax(fabs(res[1][((cg * 2 + 0) * 64 + l) * 4 + e] - re), fabs(res[1][((cg * 2 + 1) * 64 + l) * 4 + e] - im)));
      }
    }
  printf("max |error| against float64: VALU %.2e, MFMA %.2e (peak %.2f)\n", worst[0], worst[1], peak);
  return 0;
}
