// fft_core.h -- lane-level arithmetic of the STFT's FFT path (stft_fft.inl): small DFTs on registers,
// the Stockham passes of a complex FFT of M = 64 P points held P per lane, and the real-input
// post-processing.  Plain C++ (no LDS / lane intrinsics in here): the same functions are compiled for the
// host by tests/fft_core_harness.cpp, which runs the 64 lanes one after the other and compares with a
// float64 DFT -- the index arithmetic is tested without a GPU.
//
// Replaces, for window x DFT bases, the two conv1d of STFT.forward (stft.py:290-293: every bin a dot
// product of n_fft taps) by X[k] = sum_n w[n] x[n] e^(-2 pi i k n / N) evaluated as an FFT.
#pragma once

#if defined(__HIPCC__)
#define FFT_HD __host__ __device__ __forceinline__
#else
#define FFT_HD inline
#endif

namespace fftcore {

typedef float cf __attribute__((ext_vector_type(2)));  // (re, im)

FFT_HD cf cmul(cf a, cf w) { return cf{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
FFT_HD cf mul_mi(cf a) { return cf{a.y, -a.x}; }  // a * (-i)

// ---- DFT_R on registers, natural order in and out:  v[k] <- sum_r v[r] e^(-2 pi i r k / R)
FFT_HD void dft2(cf &a, cf &b) {
  const cf t = a - b;
  a = a + b;
  b = t;
}

FFT_HD void dft4(cf (&v)[4]) {
  const cf s02 = v[0] + v[2], d02 = v[0] - v[2];
  const cf s13 = v[1] + v[3], d13 = mul_mi(v[1] - v[3]);
  v[0] = s02 + s13;
  v[1] = d02 + d13;
  v[2] = s02 - s13;
  v[3] = d02 - d13;
}

// decimation in time: DFT_R from the DFT_{R/2} of the even and of the odd inputs
FFT_HD void dft8(cf (&v)[8]) {
  cf e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
  dft4(e);
  dft4(o);
  constexpr float h = 0.70710678118654752f;
  const cf t1 = cf{(o[1].x + o[1].y) * h, (o[1].y - o[1].x) * h};   // o1 * e^(-i pi/4)
  const cf t2 = mul_mi(o[2]);                                        // o2 * (-i)
  const cf t3 = cf{(o[3].y - o[3].x) * h, -(o[3].x + o[3].y) * h};  // o3 * e^(-3 i pi/4)
  v[0] = e[0] + o[0];
  v[4] = e[0] - o[0];
  v[1] = e[1] + t1;
  v[5] = e[1] - t1;
  v[2] = e[2] + t2;
  v[6] = e[2] - t2;
  v[3] = e[3] + t3;
  v[7] = e[3] - t3;
}

FFT_HD void dft16(cf (&v)[16]) {
  cf e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    e[i] = v[2 * i];
    o[i] = v[2 * i + 1];
  }
  dft8(e);
  dft8(o);
  constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f;  // cos, sin of pi/8
  constexpr float h = 0.70710678118654752f;
  const cf w[8] = {cf{1.f, 0.f}, cf{c1, -s1}, cf{h, -h}, cf{s1, -c1}, cf{0.f, -1.f}, cf{-s1, -c1}, cf{-h, -h}, cf{-c1, -s1}};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    cf t;
    if (k == 0)
      t = o[0];
    else if (k == 4)
      t = mul_mi(o[4]);
    else if (k == 2)
      t = cf{(o[2].x + o[2].y) * h, (o[2].y - o[2].x) * h};
    else if (k == 6)
      t = cf{(o[6].y - o[6].x) * h, -(o[6].x + o[6].y) * h};
    else
      t = cmul(o[k], w[k]);
    v[k] = e[k] + t;
    v[k + 8] = e[k] - t;
  }
}

template <int R>
FFT_HD void dft(cf (&v)[R]) {
  static_assert(R == 2 || R == 4 || R == 8 || R == 16, "radix");
  if constexpr (R == 2) dft2(v[0], v[1]);
  if constexpr (R == 4) dft4(v);
  if constexpr (R == 8) dft8(v);
  if constexpr (R == 16) dft16(v);
}

// ---- exchange buffer: element o of a pass's output lives at pad(o) (one spare element per 16: the writes
// of a pass -- lane stride R elements -- and its reads -- lane stride 1 -- are both spread over the banks)
FFT_HD int pad(int o) { return o + (o >> 4); }
template <int M>
constexpr int padded_size() {  // (+ 1: pad(M), which lane 0 addresses as the mirror of bin 0 and never uses)
  return M + (M >> 4) + 1;
}

// Radix plan of the M-point complex FFT on 64 lanes, P = M / 64 points per lane (lane l holds the
// elements l + 64 i of the pass's input):  M = 1024: 16 x 16 x 4,  512: 8 x 8 x 8,  256: 4 x 4 x 4 x 4.
template <int M>
struct Radix;
template <>
struct Radix<1024> {
  static constexpr int n = 3, r0 = 16, r1 = 16, r2 = 4, r3 = 1;
};
template <>
struct Radix<512> {
  static constexpr int n = 3, r0 = 8, r1 = 8, r2 = 8, r3 = 1;
};
template <>
struct Radix<256> {
  static constexpr int n = 4, r0 = 4, r1 = 4, r2 = 4, r3 = 4;
};
template <int M, int PASS>
constexpr int radix_of() {
  return PASS == 0 ? Radix<M>::r0 : PASS == 1 ? Radix<M>::r1 : PASS == 2 ? Radix<M>::r2 : Radix<M>::r3;
}
template <int M, int PASS>
constexpr int ns_of() {  // product of the radices of the passes before PASS
  int ns = 1;
  if (PASS > 0) ns *= Radix<M>::r0;
  if (PASS > 1) ns *= Radix<M>::r1;
  if (PASS > 2) ns *= Radix<M>::r2;
  return ns;
}
// twiddles of a pass a lane keeps: (P / R) butterflies x (R - 1) factors
template <int M, int PASS>
constexpr int tw_count() {
  return PASS == 0 ? 0 : (M / 64 / radix_of<M, PASS>()) * (radix_of<M, PASS>() - 1);
}
template <int M>
constexpr int tw_total() {
  return tw_count<M, 1>() + tw_count<M, 2>() + (Radix<M>::n > 3 ? tw_count<M, 3>() : 0);
}
template <int M, int PASS>
constexpr int tw_offset() {
  return PASS <= 1 ? 0 : PASS == 2 ? tw_count<M, 1>() : tw_count<M, 1>() + tw_count<M, 2>();
}

// fraction (in turns, negative) of twiddle `idx` of pass PASS for a lane: butterfly q = idx / (R - 1),
// factor r = idx % (R - 1) + 1:  W = e^(-2 pi i r k / (NS R)),  k = (lane + 64 q) mod NS
template <int M, int PASS>
FFT_HD float tw_turns(int lane, int idx) {
  constexpr int R = radix_of<M, PASS>(), NS = ns_of<M, PASS>();
  const int q = idx / (R - 1), r = idx % (R - 1) + 1;
  const int k = (lane + 64 * q) & (NS - 1);
  return -(float)(r * k) / (float)(NS * R);
}

// One Stockham pass on a lane's P elements: butterfly q takes the elements q + r (P / R), r = 0 .. R-1
// (input indices j + r M / R, j = lane + 64 q), multiplies them by W^(r k), k = j mod NS, transforms, and
// sends output r to index (j - k) R + k + r NS.  `twf(q, r)` supplies W^(r k) of butterfly q (r >= 1; see
// tw_turns), `store(o, value)` receives the unpadded index.  The values also stay in x (same slots): after
// the LAST pass slot i holds output lane + 64 i -- the natural-order spectrum.
template <int M, int PASS, typename Tw, typename Store>
FFT_HD void stockham_pass(cf (&x)[M / 64], int lane, Tw &&twf, Store &&store) {
  constexpr int P = M / 64, R = radix_of<M, PASS>(), NS = ns_of<M, PASS>(), Q = P / R;
  static_assert(P % R == 0, "a lane holds whole butterflies");
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    cf v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = x[q + r * Q];
    if constexpr (NS > 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) v[r] = cmul(v[r], twf(q, r));
    }
    dft<R>(v);
    const int j = lane + 64 * q, k = j & (NS - 1);
    const int base = (j - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      x[q + r * Q] = v[r];
      store(base + r * NS, v[r]);
    }
  }
}

// Real-input post-processing: Z = FFT_M of z[m] = y[2m] + i y[2m+1]; the N = 2M point spectrum of y is
//   X[k] = (Zk + conj(Zm)) / 2 - (i / 2) e^(-2 pi i k / N) (Zk - conj(Zm)),  Zm = Z[(M - k) mod M]
// `wh` = e^(-2 pi i k / N) / 2.  (k = 0 gives X[0] = Re Z0 + Im Z0; the Nyquist bin is Re Z0 - Im Z0.)
FFT_HD cf real_post(cf zk, cf zm, cf wh) {
  const cf a = cf{zk.x + zm.x, zk.y - zm.y} * 0.5f;
  const cf b = cf{zk.x - zm.x, zk.y + zm.y};
  const cf wb = cmul(b, wh);
  return cf{a.x + wb.y, a.y - wb.x};
}

// ... for the pair (k, M - k) at once: A and B of bin M - k are conj(A), -conj(B) and its factor is
// -conj(w): half the multiplications.  With zm = zk at k = 0 the pair is (X[0], Nyquist bin); bin M/2 is its
// own mirror: X[M/2] = conj(Z[M/2]).
FFT_HD void real_post_pair(cf zk, cf zm, cf wh, cf &xk, cf &xm) {
  const cf a = cf{zk.x + zm.x, zk.y - zm.y} * 0.5f;
  const cf b = cf{zk.x - zm.x, zk.y + zm.y};
  const cf wb = cmul(b, wh);
  xk = cf{a.x + wb.y, a.y - wb.x};
  xm = cf{a.x - wb.y, -(a.y + wb.x)};
}

// Inverse direction (iSTFT): the N real samples y[n] = sum_{k=0}^{N-1} G[k] e^(+2 pi i k n / N) of a Hermitian
// spectrum given by its half G[0 .. M] are  y[2m] = Re z[m], y[2m+1] = Im z[m]  with  z = the UNNORMALISED
// inverse M-point FFT of  Z[k] = (Gk + conj(Gm)) + i e^(+2 pi i k / N) (Gk - conj(Gm)),  Gm = G[M - k],
// k = 0 .. M-1.  `w` = e^(+2 pi i k / N).  The inverse FFT is run as conj(FFT(conj Z)): this returns conj Z.
FFT_HD cf real_pre_conj(cf gk, cf gm, cf w) {
  const cf a = cf{gk.x + gm.x, gk.y - gm.y};
  const cf b = cf{gk.x - gm.x, gk.y + gm.y};
  const cf wb = cmul(b, w);            // i * wb = (-wb.y, wb.x)
  return cf{a.x - wb.y, -(a.y + wb.x)};  // conj(a + i wb)
}

}  // namespace fftcore
