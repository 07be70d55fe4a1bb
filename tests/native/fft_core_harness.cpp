// Host check of nnaudio_amd/csrc/fft_core.h: the 64 lanes of a wave run one after the other, the exchange
// buffer is a plain array; compared with a float64 DFT.  Built and run by tests/test_fft_core_cpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_core.h"

using namespace fftcore;

template <int M, int PASS>
void run_pass(cf (*x)[M / 64], std::vector<cf> &buf, bool reload) {
  cf tw[tw_count<M, PASS>() + 1];
  for (int lane = 0; lane < 64; ++lane) {
    for (int i = 0; i < tw_count<M, PASS>(); ++i) {
      const double t = 2.0 * M_PI * (double)tw_turns<M, PASS>(lane, i);
      tw[i] = cf{(float)cos(t), (float)sin(t)};
    }
    stockham_pass<M, PASS>(x[lane], lane, [&](int q, int r) { return tw[q * (radix_of<M, PASS>() - 1) + r - 1]; },
                           [&](int o, cf v) { buf[pad(o)] = v; });
  }
  if (reload)
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < M / 64; ++i) x[lane][i] = buf[pad(lane + 64 * i)];
}

// taps < N: a frame of `taps` samples zero-extended to N (how n_fft = 128 / 256 run on the 512-point
// instance): bin k of the taps-point DFT is bin k N / taps of the N-point one
template <int M>
double check(unsigned seed, int taps = 2 * M) {
  constexpr int N = 2 * M, P = M / 64;
  std::vector<double> y(N);
  srand(seed);
  for (int n = 0; n < N; ++n) y[n] = n < taps ? (double)rand() / RAND_MAX * 2 - 1 : 0.0;
  static cf x[64][P];
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < P; ++i) {
      const int m = lane + 64 * i;
      x[lane][i] = cf{(float)y[2 * m], (float)y[2 * m + 1]};
    }
  std::vector<cf> buf(padded_size<M>());
  run_pass<M, 0>(x, buf, true);
  run_pass<M, 1>(x, buf, true);
  run_pass<M, 2>(x, buf, Radix<M>::n > 3);
  if constexpr (Radix<M>::n > 3) run_pass<M, 3>(x, buf, false);
  // buf holds Z in natural order, and so do the lanes' slots
  std::vector<double> xr(M + 1), xi(M + 1);
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < P; ++i) {  // every bin on its own
      const int k = lane + 64 * i;
      const cf zk = x[lane][i];
      if (zk.x != buf[pad(k)].x || zk.y != buf[pad(k)].y) return 1e9;
      const cf zm = buf[pad((M - k) & (M - 1))];
      const double t = -2.0 * M_PI * k / N;
      const cf X = real_post(zk, zm, cf{(float)(0.5 * cos(t)), (float)(0.5 * sin(t))});
      xr[k] = X.x;
      xi[k] = X.y;
      if (k == 0) {
        xr[M] = zk.x - zk.y;
        xi[M] = 0;
      }
    }
  // the kernel's way: pairs (k, M - k) from the lower half of a lane's slots, the mirror read at
  // pad(M - lane) - 68 i, bin M/2 from lane 0's slot P/2 -- must give the same numbers
  for (int lane = 0; lane < 64; ++lane) {
    const int pm0 = pad(M - lane);
    for (int i = 0; i < P / 2; ++i) {
      const int k = lane + 64 * i;
      if (pm0 - 68 * i != pad(M - k)) return 2e9;
      cf zm = buf[pm0 - 68 * i];
      if (k == 0) zm = x[0][0];
      const double t = -2.0 * M_PI * k / N;
      cf xk, xm;
      real_post_pair(x[lane][i], zm, cf{(float)(0.5 * cos(t)), (float)(0.5 * sin(t))}, xk, xm);
      if (fabs(xk.x - xr[k]) > 1e-6 * (1 + fabs(xr[k])) || fabs(xk.y - xi[k]) > 1e-6 * (1 + fabs(xi[k]))) return 3e9;
      if (fabs(xm.x - xr[M - k]) > 1e-5 * (1 + fabs(xr[M - k])) || fabs(xm.y - xi[M - k]) > 1e-5 * (1 + fabs(xi[M - k]))) return 4e9;
    }
  }
  if (fabs(x[0][P / 2].x - xr[M / 2]) > 1e-5 * (1 + fabs(xr[M / 2])) || fabs(-x[0][P / 2].y - xi[M / 2]) > 1e-5 * (1 + fabs(xi[M / 2]))) return 5e9;
  double err = 0, peak = 0;
  const int step = N / taps;
  for (int k = 0; k <= taps / 2; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < taps; ++n) {
      const double t = -2.0 * M_PI * (double)((long long)k * n % taps) / taps;
      re += (double)(float)y[n] * cos(t);
      im += (double)(float)y[n] * sin(t);
    }
    err = fmax(err, hypot(xr[k * step] - re, xi[k * step] - im));
    peak = fmax(peak, hypot(re, im));
  }
  return err / peak;
}

// inverse direction: half spectrum -> N real samples through the same passes
template <int M>
double check_inverse(unsigned seed) {
  constexpr int N = 2 * M, P = M / 64;
  std::vector<double> gr(M + 1), gi(M + 1);
  srand(seed + 77);
  for (int k = 0; k <= M; ++k) {
    gr[k] = (double)rand() / RAND_MAX * 2 - 1;
    gi[k] = (k == 0 || k == M) ? 0.0 : (double)rand() / RAND_MAX * 2 - 1;
  }
  static cf x[64][P];
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < P; ++i) {
      const int k = lane + 64 * i;
      const double t = 2.0 * M_PI * k / N;
      x[lane][i] = real_pre_conj(cf{(float)gr[k], (float)gi[k]}, cf{(float)gr[M - k], (float)gi[M - k]},
                                 cf{(float)cos(t), (float)sin(t)});
    }
  std::vector<cf> buf(padded_size<M>());
  run_pass<M, 0>(x, buf, true);
  run_pass<M, 1>(x, buf, true);
  run_pass<M, 2>(x, buf, Radix<M>::n > 3);
  if constexpr (Radix<M>::n > 3) run_pass<M, 3>(x, buf, false);
  double err = 0, peak = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < P; ++i) {
      const int m = lane + 64 * i;
      const double y0 = x[lane][i].x, y1 = -x[lane][i].y;  // z = conj(FFT(conj Z))
      for (int e = 0; e < 2; ++e) {
        const int n = 2 * m + e;
        double ref = 0;
        for (int k = 0; k < N; ++k) {
          const int kk = k <= M ? k : N - k;
          const double re = (double)(float)gr[kk], im = (k <= M ? 1.0 : -1.0) * (double)(float)gi[kk];
          const double t = 2.0 * M_PI * (double)((long long)k * n % N) / N;
          ref += re * cos(t) - im * sin(t);
        }
        err = fmax(err, fabs((e ? y1 : y0) - ref));
        peak = fmax(peak, fabs(ref));
      }
    }
  return err / peak;
}

int main() {
  int bad = 0;
  for (unsigned seed = 1; seed <= 3; ++seed) {
    const double e1024 = check<1024>(seed), e512 = check<512>(seed), e256 = check<256>(seed);
    printf("seed %u: N=2048 %.2e  N=1024 %.2e  N=512 %.2e (max |d| / peak)\n", seed, e1024, e512, e256);
    bad += !(e1024 < 5e-7) + !(e512 < 5e-7) + !(e256 < 5e-7);
    const double z256 = check<256>(seed, 256), z128 = check<256>(seed, 128);
    printf("        zero-extended to 512: n_fft=256 %.2e  n_fft=128 %.2e\n", z256, z128);
    bad += !(z256 < 5e-7) + !(z128 < 5e-7);
    const double i1024 = check_inverse<1024>(seed), i512 = check_inverse<512>(seed), i256 = check_inverse<256>(seed);
    printf("        inverse: N=2048 %.2e  N=1024 %.2e  N=512 %.2e\n", i1024, i512, i256);
    bad += !(i1024 < 1e-6) + !(i512 < 1e-6) + !(i256 < 1e-6);
  }
  return bad ? 1 : 0;
}
