// Which split arithmetic reaches fp32-class accuracy on the bf16/f16 matrix pipe?  (round 3)
// One wave computes D[32 frames][32 bins] = sum_n x[f][n] * w[b][n], K = 2048 taps of a hann-windowed
// cosine basis, in several arithmetics and compares with a float64 evaluation on the host:
//   fp32     v_mfma_f32_32x32x2_f32
//   bf16x3   (hi, lo) bf16 pairs, products lo*hi + hi*lo + hi*hi          (what round 2 ships)
//   bf16x6   (hi, mid, lo) bf16 triples, 6 products
//   f16x3    (hi, lo) fp16 pairs of power-of-two SCALED operands, 3 products on v_mfma_f32_32x32x16_f16
//   f16x3u   the same without scaling (what fp16's 5 exponent bits cost)
// Columns: 0-7 a sine exactly on bin 400 (bins 100.. are ~silent: dynamic range), 8-15 a sine between
// bins, 16-23 white noise, 24-31 a quiet sine (1e-3) + noise at 1e-6.
// build: hipcc --offload-arch=gfx950 -O2 split_accuracy.hip -o split_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 2048, NB = 32, NF = 32;

__device__ inline __bf16 to_bf16(float v) { return (__bf16)v; }

// mode: 0 fp32, 1 bf16x3, 2 bf16x6, 3 f16x3 scaled, 4 f16x3 unscaled
__global__ void __launch_bounds__(64) contract(const float *__restrict__ x, const float *__restrict__ w,
                                              float *__restrict__ out, int mode, const float *xs,
                                              const float *ws) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float *xr = x + li * K, *wr = w + li * K;
  const float sx = (mode == 3) ? xs[li] : 1.f, sw = (mode == 3) ? ws[li] : 1.f;
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xr[k + lh], wr[k + lh], acc, 0, 0, 0);
  } else if (mode == 1 || mode == 2) {
    for (int k = 0; k < K; k += 16) {
      bf16x8 xh, xm, xl, ah, am, al;
      for (int j = 0; j < 8; ++j) {
        float v = xr[k + 8 * lh + j];
        __bf16 h = to_bf16(v); float r = v - (float)h; __bf16 m = to_bf16(r); float r2 = r - (float)m;
        xh[j] = h; xm[j] = m; xl[j] = to_bf16(r2);
        v = wr[k + 8 * lh + j];
        h = to_bf16(v); r = v - (float)h; m = to_bf16(r); r2 = r - (float)m;
        ah[j] = h; am[j] = m; al[j] = to_bf16(r2);
      }
      if (mode == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, am, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ah, acc, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, al, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, am, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, am, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ah, acc, 0, 0, 0);
      }
    }
  } else {
    for (int k = 0; k < K; k += 16) {
      f16x8 xh, xl, ah, al;
      for (int j = 0; j < 8; ++j) {
        float v = xr[k + 8 * lh + j] * sx;
        _Float16 h = (_Float16)v; xh[j] = h; xl[j] = (_Float16)(v - (float)h);
        v = wr[k + 8 * lh + j] * sw;
        h = (_Float16)v; ah[j] = h; al[j] = (_Float16)(v - (float)h);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, al, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, ah, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, ah, acc, 0, 0, 0);
    }
  }
  // acc[e] = D[frame (e&3) + 8 (e>>2) + 4 lh][bin li]
  for (int e = 0; e < 16; ++e) {
    const int f = (e & 3) + 8 * (e >> 2) + 4 * lh;
    float v = acc[e];
    if (mode == 3) v = v / (xs[f] * sw);
    out[f * NB + li] = v;
  }
}

int main() {
  std::vector<float> x(NF * K), w(NB * K), xs(NF), ws(NB);
  std::vector<double> ref(NF * NB);
  int bins[NB];
  bins[0] = 400; bins[1] = 399; bins[2] = 401;
  for (int b = 3; b < NB; ++b) bins[b] = 97 + b;
  for (int b = 0; b < NB; ++b)
    for (int n = 0; n < K; ++n) {
      const double win = 0.5 - 0.5 * cos(2 * M_PI * n / K);
      const float c = (float)cos(2 * M_PI * ((long long)bins[b] * n % K) / K);
      w[b * K + n] = c * (float)win;  // fp32 product like stft.py:230-232
    }
  srand(1);
  auto rnd = []() { double s = 0; for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX; return s - 6.0; };
  for (int f = 0; f < NF; ++f)
    for (int n = 0; n < K; ++n) {
      double v;
      const double ph = 0.3 * f;
      if (f < 8) v = cos(2 * M_PI * 400.0 * n / K + ph);
      else if (f < 16) v = cos(2 * M_PI * 400.37 * n / K + ph);
      else if (f < 24) v = 0.3 * rnd();
      else v = 1e-3 * cos(2 * M_PI * 400.0 * n / K + ph) + 1e-6 * rnd();
      x[f * K + n] = (float)v;
    }
  for (int f = 0; f < NF; ++f) {
    float m = 0; for (int n = 0; n < K; ++n) m = fmaxf(m, fabsf(x[f * K + n]));
    int e; frexpf(m, &e); xs[f] = ldexpf(1.f, 15 - e);  // max |x| * scale in [2^14, 2^15)
  }
  for (int b = 0; b < NB; ++b) {
    float m = 0; for (int n = 0; n < K; ++n) m = fmaxf(m, fabsf(w[b * K + n]));
    int e; frexpf(m, &e); ws[b] = ldexpf(1.f, 15 - e);
  }
  for (int f = 0; f < NF; ++f)
    for (int b = 0; b < NB; ++b) {
      double s = 0;
      for (int n = 0; n < K; ++n) s += (double)x[f * K + n] * (double)w[b * K + n];
      ref[f * NB + b] = s;
    }
  float *dx, *dw, *dout, *dxs, *dws;
  hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dout, NF * NB * 4);
  hipMalloc(&dxs, NF * 4); hipMalloc(&dws, NB * 4);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dxs, xs.data(), NF * 4, hipMemcpyHostToDevice);
  hipMemcpy(dws, ws.data(), NB * 4, hipMemcpyHostToDevice);
  const char *names[5] = {"fp32", "bf16x3", "bf16x6", "f16x3", "f16x3u"};
  std::vector<float> out(NF * NB);
  for (int mode = 0; mode < 5; ++mode) {
    hipLaunchKernelGGL(contract, dim3(1), dim3(64), 0, 0, dx, dw, dout, mode, dxs, dws);
    hipMemcpy(out.data(), dout, NF * NB * 4, hipMemcpyDeviceToHost);
    // per column group: peak of the group, max error over all bins, max error over the silent bins
    printf("%-7s", names[mode]);
    for (int g = 0; g < 4; ++g) {
      double peak = 0, eall = 0, esil = 0, vsil = 0;
      for (int f = 8 * g; f < 8 * g + 8; ++f)
        for (int b = 0; b < NB; ++b) {
          const double r = ref[f * NB + b], e = fabs(out[f * NB + b] - r);
          peak = fmax(peak, fabs(r)); eall = fmax(eall, e);
          if (b >= 3) { esil = fmax(esil, e); vsil = fmax(vsil, fabs(r)); }
        }
      printf("  | g%d peak %.3e err/peak %.2e silent: |v|/peak %.1e err/peak %.2e (%.0f dB)", g, peak,
             eall / peak, vsil / peak, esil / peak, 20 * log10(esil / peak + 1e-300));
    }
    printf("\n");
  }
  return 0;
}
