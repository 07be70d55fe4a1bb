// framed_fold2.inl -- the SECOND symmetric fold of a Fourier basis: the contraction over a QUARTER of
// the taps.  Included by mispec.hip after framed_fold.inl (whose contraction kernel it reuses).
//
// The reference's STFT basis is window x DFT: wcos[k, n] = w[n] cos(2 pi k n / N), wsin likewise
// (utils.py:379-389, stft.py:230-232; N = n_fft, k = 0 .. freq_bins-1).  Moving the window to the
// signal, y[n] = w[n] x_t[n], the bare DFT rows are (anti)symmetric about N/2 AND about N/4:
//     cos(2 pi k (N/2 - n) / N) = (-1)^k cos(2 pi k n / N),   sin(...) = (-1)^(k+1) sin(...)
// With E[n] = y[n] + y[N-n], O[n] = y[n] - y[N-n] (the first fold, framed_fold.inl), M = N/2, Q = N/4:
//     Ep[n] = E[n] + E[M-n]    Em[n] = E[n] - E[M-n]    Op[n] = O[n] + O[M-n]    Om[n] = O[n] - O[M-n]
//     even k:  re = sum_n Ep[n] cos(2 pi k n / N)      im = sum_n Om[n] sin(2 pi k n / N)
//     odd  k:  re = sum_n Em[n] cos(...)               im = sum_n Op[n] sin(...)        n = 0 .. Q
// -- a quarter of the dense MFMAs.  Folded slot j < Q stands for n = j + 1 (the four samples n, N-n,
// M-n, M+n): exactly Q = N/4 slots.  At n = Q the pair (n, M-n) is one tap: the general formulas give
// twice E[Q] / O[Q] and the folded basis carries half the coefficient there.  Tap n = 0 (samples 0 and
// M; cos = 1 for every bin, sin = 0) is not a slot: the pre-pass writes  y[0] + y[M]  (even bins) and
// y[0] - y[M]  (odd bins) per frame to col_add, and the contraction adds them to its real accumulators
// before the epilogue (a 33rd K stage for this one tap cost 3 % of the MFMAs).
//
// Nothing is assumed about the basis: fold2_basis_kernel compares EVERY coefficient of the module's
// buffers with  row 0 of the cos basis (= the window) x the analytic DFT value  and reports the
// largest difference; the caller offers the folded planes only when that is rounding noise
// (engine.fold2_basis).  The window may be any window (no symmetry needed).
//
// Operands (the stage rows of framed_fold.inl, 128 B = 16 slots x 4 planes or 2 fp32 planes):
//   folded basis : [even bins: ceil(F/2) rows | odd bins: floor(F/2) rows], row = per stage
//                  [cos_hi 16 | cos_lo 16 | sin_hi 16 | sin_lo 16]; then the fp32 (cos | sin) rows of
//                  the bin the pre-pass evaluates.  FOLD_F16X3: fp16 pairs of coefficient x 2^14.
//   folded frames: [even: n_cols rows of (Ep | Om)] [odd: n_cols rows of (Em | Op)], written every
//                  call by fold2_frames_kernel; FOLD_F16X3: fp16 pairs of value x 2^s, s chosen per
//                  workgroup from the largest |sample| it reads so that the folded values stay below
//                  2^15; col_unscale[frame] = 2^-(s + 14) undoes both scalings on the accumulators.
// The contraction is framed_fold_kernel & co. with p.fold2 set: its row tiles are the even bins'
// then the odd bins', output rows interleaved (framed_fold_tile).

constexpr int FOLD2_FR = 2;          // frames per thread group of the pre-pass

__host__ __device__ inline int fold2_taps(int kernel) {
  return (kernel / 4 + FOLD_KC - 1) / FOLD_KC * FOLD_KC;  // kernel % 64 == 0: exactly kernel / 4
}
// threads per frame group of the pre-pass: one quad of slots per thread and trip
__host__ __device__ inline int fold2_tg(int kernel) {
  const int quads = kernel / 16;  // paired quads (Q / 4)
  return quads <= 64 ? 64 : (quads <= 128 ? 128 : 256);
}

// ---------------------------------------------------------------------------------
// basis -> quarter-folded planes from the ANALYTIC DFT (+ how far the buffers are from
// window x DFT).  grid (ceil(Kf / 256), n_bins).  stats[0] = max |buffer - w * dft| over all
// coefficients, stats[1] = max |buffer|, stats[2] = max |window| (float bit patterns, atomicMax).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold2_basis_kernel(const float *__restrict__ re,
                                                          const float *__restrict__ im,
                                                          long long row_stride, int n_bins, int N, int Kf,
                                                          unsigned short *__restrict__ dst,
                                                          float *__restrict__ last_rows, int last_bin,
                                                          unsigned *__restrict__ stats, int arith) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;
  const int Q = N >> 2, M = N >> 1;
  float mism = 0.f, amax = 0.f, wmax = 0.f;
  if (j < Kf) {
    const int n = j < Q ? j + 1 : -1;
    float c = 0.f, s = 0.f;
    if (n >= 0) {
      const double ang = 2.0 * (double)(((long long)k * n) % N) / (double)N;  // in units of pi
      double cd = cospi(ang), sd = sinpi(ang);
      if (n == Q) {  // the pair (n, M - n) is one tap
        cd *= 0.5;
        sd *= 0.5;
      }
      c = (float)cd;
      s = (float)sd;
      // the buffers at the (up to) four samples this slot stands for
      // (+ samples 0 and M, the tap the contraction adds from col_add: checked by slot 0's thread)
      const int pos[6] = {n, N - n, M - n, M + n, j == 0 ? 0 : -1, j == 0 ? M : -1};
      const float *wr = re + (long long)k * row_stride;
      const float *wi = im + (long long)k * row_stride;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int m = pos[u];
        if (m < 0 || m >= N) continue;
        const double a = 2.0 * (double)(((long long)k * m) % N) / (double)N;
        const float w = re[m];  // row 0 of the cos basis: cos(0) = 1, i.e. the window
        const float r1 = wr[m], i1 = wi[m];
        mism = fmaxf(mism, fmaxf(fabsf(r1 - (float)((double)w * cospi(a))), fabsf(i1 - (float)((double)w * sinpi(a)))));
        amax = fmaxf(amax, fmaxf(fabsf(r1), fabsf(i1)));
        wmax = fmaxf(wmax, fabsf(w));
      }
    }
    const int par = k & 1, r = k >> 1, ne = (n_bins + 1) >> 1;
    unsigned short *row = dst + ((long long)(par ? ne + r : r) * (Kf / FOLD_KC) + j / FOLD_KC) * (FOLD_ROWB / 2);
    const int u = j % FOLD_KC;
    if (arith == FOLD_F32) {
      float *frow = reinterpret_cast<float *>(row);
      frow[u] = c;
      frow[16 + u] = s;
    } else {
      unsigned ch, cl, sh, sl;
      if (arith == FOLD_F16X3) {
        unsigned h2, l2;
        f16_split2(c * FOLD_ASCALE, s * FOLD_ASCALE, h2, l2);
        ch = h2 & 0xffff, sh = h2 >> 16, cl = l2 & 0xffff, sl = l2 >> 16;
      } else {
        bf16_split(c, ch, cl);
        bf16_split(s, sh, sl);
      }
      row[u] = (unsigned short)ch;
      row[16 + u] = (unsigned short)cl;
      row[32 + u] = (unsigned short)sh;
      row[48 + u] = (unsigned short)sl;
    }
    if (k == last_bin) {
      last_rows[j] = c;
      last_rows[Kf + j] = s;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mism = fmaxf(mism, __shfl_xor(mism, d));
    amax = fmaxf(amax, __shfl_xor(amax, d));
    wmax = fmaxf(wmax, __shfl_xor(wmax, d));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&stats[0], __float_as_uint(mism));
    atomicMax(&stats[1], __float_as_uint(amax));
    atomicMax(&stats[2], __float_as_uint(wmax));
  }
}

// ---------------------------------------------------------------------------------
// Pre-pass: one workgroup per FOLD2_FR * G consecutive frames of a clip, G = 256 / TG thread groups
// (TG = fold2_tg(N)) of FOLD2_FR frames each.  Thread i of a group owns slots 4 i .. 4 i + 3 (+ 4 TG
// per trip) of each of its frames: four 16-byte loads per frame (x_t[n ..] and x_t[M+n ..] forwards,
// x_t[N-n ..] and x_t[M-n ..] backwards), the window applied from row 0 of the cos basis, the four
// combinations in fp32, split (bf16 / scaled fp16) or kept (fp32), assembled in LDS as in
// fold_frames_kernel and stored as two runs of memory per frame (even / odd operand).
// With p.fold_last the LAST even bin (the Nyquist bin of an n_fft/2+1 STFT) is evaluated here in fp32.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold2_frames_kernel(const KParams p, unsigned short *__restrict__ dst) {
  const int N = p.K, M = N >> 1, Q = N >> 2, Kf = p.Ks;
  const int TG = fold2_tg(N), G = 256 / TG;
  const int grp = threadIdx.x / TG, gt = threadIdx.x - grp * TG;
  const int c = p.fold_clip0 + blockIdx.y;
  const int tw0 = blockIdx.x * (FOLD2_FR * G);
  const int nfw = p.n_frames - tw0 < FOLD2_FR * G ? p.n_frames - tw0 : FOLD2_FR * G;
  const int t0 = tw0 + grp * FOLD2_FR;
  int nf = p.n_frames - t0;
  nf = nf < FOLD2_FR ? nf : FOLD2_FR;
  const float *x = p.x + (long long)c * p.x_clip_stride;
  const float *win = p.a_re;  // row 0 of the cos basis
  const long long col0 = (long long)c * p.n_frames + tw0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int rows_f = Kf / FOLD_KC;  // staging rows per frame and operand
  float *const red = reinterpret_cast<float *>(smem_raw + (size_t)FOLD2_FR * G * 2 * Kf * 8);  // [2][4][FR] + [4]
  const float *le = p.fold_last, *lo = le ? le + Kf : nullptr;
  float pe[FOLD2_FR], po[FOLD2_FR];
#pragma unroll
  for (int f = 0; f < FOLD2_FR; ++f) pe[f] = po[f] = 0.f;
  const long long qa = (long long)t0 * p.hop - p.pad;
  const bool interior = qa >= 0 && qa + (long long)(nf - 1) * p.hop + N <= p.n_samples;

  // one trip's values: slots j0 .. j0 + 3 of the group's frames
  // (returns the largest magnitude among them)
  auto compute = [&](int j0, float (&ep)[FOLD2_FR][4], float (&em)[FOLD2_FR][4], float (&op)[FOLD2_FR][4],
                     float (&om)[FOLD2_FR][4]) __attribute__((always_inline)) -> float {
    float mx = 0.f;
    {  // slots of n = j0+1 .. j0+4 <= Q: samples n, N-n, M-n, M+n
      const f32x4u wA = *reinterpret_cast<const f32x4u *>(win + j0 + 1);
      const f32x4u wB = *reinterpret_cast<const f32x4u *>(win + N - j0 - 4);
      const f32x4u wC = *reinterpret_cast<const f32x4u *>(win + M - j0 - 4);
      const f32x4u wD = *reinterpret_cast<const f32x4u *>(win + M + j0 + 1);
      float A[FOLD2_FR][4], B[FOLD2_FR][4], C[FOLD2_FR][4], D[FOLD2_FR][4];
      if (interior) {
        f32x4u va[FOLD2_FR], vb[FOLD2_FR], vc[FOLD2_FR], vd[FOLD2_FR];
#pragma unroll
        for (int f = 0; f < FOLD2_FR; ++f) {
          const long long q0 = qa + (long long)(f < nf ? f : 0) * p.hop;
          va[f] = *reinterpret_cast<const f32x4u *>(x + q0 + j0 + 1);
          vb[f] = *reinterpret_cast<const f32x4u *>(x + q0 + N - j0 - 4);
          vc[f] = *reinterpret_cast<const f32x4u *>(x + q0 + M - j0 - 4);
          vd[f] = *reinterpret_cast<const f32x4u *>(x + q0 + M + j0 + 1);
        }
#pragma unroll
        for (int f = 0; f < FOLD2_FR; ++f)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            A[f][i] = va[f][i];
            B[f][i] = vb[f][i];
            C[f][i] = vc[f][i];
            D[f][i] = vd[f][i];
          }
      } else {
#pragma unroll
        for (int f = 0; f < FOLD2_FR; ++f) {
          const long long q0 = qa + (long long)(f < nf ? f : 0) * p.hop;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const long long cb = (long long)c * p.x_clip_stride;
            A[f][i] = fetch_sample(p.x, cb, (int)(q0 + j0 + 1 + i), p.n_samples, p.pad_mode, true);
            B[f][i] = fetch_sample(p.x, cb, (int)(q0 + N - j0 - 4 + i), p.n_samples, p.pad_mode, true);
            C[f][i] = fetch_sample(p.x, cb, (int)(q0 + M - j0 - 4 + i), p.n_samples, p.pad_mode, true);
            D[f][i] = fetch_sample(p.x, cb, (int)(q0 + M + j0 + 1 + i), p.n_samples, p.pad_mode, true);
          }
        }
      }
#pragma unroll
      for (int f = 0; f < FOLD2_FR; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = A[f][i] * wA[i], b = B[f][3 - i] * wB[3 - i];
          const float cc = C[f][3 - i] * wC[3 - i], d = D[f][i] * wD[i];
          const float s1 = a + b, s2 = cc + d, d1 = a - b, d2 = cc - d;
          ep[f][i] = s1 + s2;
          em[f][i] = s1 - s2;
          op[f][i] = d1 + d2;
          om[f][i] = d1 - d2;
          mx = fmaxf(fmaxf(mx, fmaxf(fabsf(ep[f][i]), fabsf(em[f][i]))), fmaxf(fabsf(op[f][i]), fabsf(om[f][i])));
        }
    }
    return mx;
  };
  // four slots of the two planes of one staging row R (u = first slot inside the row's stage)
  auto stage_quad = [&](int R, int u, const float (&c0)[4], const float (&c1)[4], float scale)
                        __attribute__((always_inline)) {
    const int sw = R & 7;
    unsigned char *r = smem_raw + (size_t)R * FOLD_ROWB;
    if (p.fold_arith == FOLD_F32) {
      const f32x4v v0 = {c0[0], c0[1], c0[2], c0[3]}, v1 = {c1[0], c1[1], c1[2], c1[3]};
      *reinterpret_cast<f32x4v *>(r + (((u >> 2)) ^ sw) * 16) = v0;
      *reinterpret_cast<f32x4v *>(r + ((4 + (u >> 2)) ^ sw) * 16) = v1;
    } else {
      uint2 h0, l0, h1, l1;
      if (p.fold_arith == FOLD_F16X3) {
        f16_split2(c0[0] * scale, c0[1] * scale, h0.x, l0.x);
        f16_split2(c0[2] * scale, c0[3] * scale, h0.y, l0.y);
        f16_split2(c1[0] * scale, c1[1] * scale, h1.x, l1.x);
        f16_split2(c1[2] * scale, c1[3] * scale, h1.y, l1.y);
      } else {
        bf16_split2(c0[0], c0[1], h0.x, l0.x);
        bf16_split2(c0[2], c0[3], h0.y, l0.y);
        bf16_split2(c1[0], c1[1], h1.x, l1.x);
        bf16_split2(c1[2], c1[3], h1.y, l1.y);
      }
      const int h = u >> 3, sub = (u & 4) * 2;
      *reinterpret_cast<uint2 *>(r + ((0 + h) ^ sw) * 16 + sub) = h0;
      *reinterpret_cast<uint2 *>(r + ((2 + h) ^ sw) * 16 + sub) = l0;
      *reinterpret_cast<uint2 *>(r + ((4 + h) ^ sw) * 16 + sub) = h1;
      *reinterpret_cast<uint2 *>(r + ((6 + h) ^ sw) * 16 + sub) = l1;
    }
  };
  // the last even bin's partial sums, the split and the staging of one trip's values
  auto emit = [&](int j0, const float (&ep)[FOLD2_FR][4], const float (&em)[FOLD2_FR][4],
                  const float (&op)[FOLD2_FR][4], const float (&om)[FOLD2_FR][4], float scale)
                  __attribute__((always_inline)) {
    f32x4v we = {0.f, 0.f, 0.f, 0.f}, wo = {0.f, 0.f, 0.f, 0.f};
    if (le) {
      we = *reinterpret_cast<const f32x4v *>(le + j0);
      wo = *reinterpret_cast<const f32x4v *>(lo + j0);
    }
#pragma unroll
    for (int f = 0; f < FOLD2_FR; ++f) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // the last even bin: (Ep, Om)
        pe[f] = fmaf(we[i], ep[f][i], pe[f]);
        po[f] = fmaf(wo[i], om[f][i], po[f]);
      }
      const int u = j0 % FOLD_KC;
      // staging rows of (frame, operand, stage); operand 0 = (Ep | Om), 1 = (Em | Op)
      stage_quad(((grp * FOLD2_FR + f) * 2 + 0) * rows_f + j0 / FOLD_KC, u, ep[f], om[f], scale);
      stage_quad(((grp * FOLD2_FR + f) * 2 + 1) * rows_f + j0 / FOLD_KC, u, em[f], op[f], scale);
    }
  };
  // workgroup maximum of a per-thread value (the idle groups of the last block take part)
  auto wg_max = [&](float m) __attribute__((always_inline)) -> float {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) red[2 * 4 * FOLD2_FR + (threadIdx.x >> 6)] = m;
    __syncthreads();
    return fmaxf(fmaxf(red[2 * 4 * FOLD2_FR], red[2 * 4 * FOLD2_FR + 1]),
                 fmaxf(red[2 * 4 * FOLD2_FR + 2], red[2 * 4 * FOLD2_FR + 3]));
  };

  // tap n = 0 of the group's frames: y[0] +- y[M], added by the contraction to the real parts of the
  // even / odd bins (and here to the last even bin's sum)
  if (gt == 0 && nf > 0) {
#pragma unroll
    for (int f = 0; f < FOLD2_FR; ++f) {
      if (f < nf) {
        const long long q0 = qa + (long long)f * p.hop, cb = (long long)c * p.x_clip_stride;
        const float y0 = win[0] * fetch_sample(p.x, cb, (int)q0, p.n_samples, p.pad_mode, true);
        const float yM = win[M] * fetch_sample(p.x, cb, (int)(q0 + M), p.n_samples, p.pad_mode, true);
        const long long col = (long long)c * p.n_frames + t0 + f;
        p.col_add[col] = y0 + yM;
        p.col_add[p.n_cols + col] = y0 - yM;
        if (le) pe[f] += y0 + yM;  // (its coefficient: cos(0) = 1)
      }
    }
  }
  if (p.fold_arith == FOLD_F16X3 && Kf <= 4 * TG) {
    // ---- FOLD_F16X3, one trip per thread: all values are formed first, the workgroup's largest
    // magnitude m = f 2^e, f in [0.5, 1), gives the scale 2^(15-e) that keeps the fp16 pairs below
    // 2^15, then they are split and staged
    float ep0[FOLD2_FR][4], em0[FOLD2_FR][4], op0[FOLD2_FR][4], om0[FOLD2_FR][4];
    const int j0 = 4 * gt;
    float m = 0.f;
    const bool live = nf > 0 && j0 < Kf;  // (short kernels: more threads than quads)
    if (live) m = compute(j0, ep0, em0, op0, om0);
    m = wg_max(m);
    const int e = absmax_exponent(m);
    const float scale = pow2f(15 - e);
    if ((int)threadIdx.x < nfw) p.col_unscale[col0 + threadIdx.x] = pow2f(e - 15 - 14);  // also the basis' 2^14
    if (live) emit(j0, ep0, em0, op0, om0, scale);
  } else {
    // ---- any other case: FOLD_F16X3 takes its scale from the largest |sample| the workgroup reads
    // (x max |window|: the four-sample combinations stay below 2^(e+2))
    float scale = 1.f;
    if (p.fold_arith == FOLD_F16X3) {
      const long long wa = (long long)tw0 * p.hop - p.pad, wb = wa + (long long)(nfw - 1) * p.hop + N;
      float m = 0.f;
      if (wa >= 0 && wb <= p.n_samples) {
        long long q = wa + 4 * threadIdx.x;
        for (; q + 4 <= wb; q += 1024) {
          const f32x4u v = *reinterpret_cast<const f32x4u *>(x + q);
          m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (; q < wb; ++q) m = fmaxf(m, fabsf(x[q]));
      } else {
        for (long long q = wa + threadIdx.x; q < wb; q += 256)
          m = fmaxf(m, fabsf(fetch_sample(p.x, (long long)c * p.x_clip_stride, (int)q, p.n_samples, p.pad_mode, true)));
      }
      m = wg_max(m) * p.fold_wmax;
      const int e = absmax_exponent(m);
      scale = pow2f(13 - e);
      if ((int)threadIdx.x < nfw) p.col_unscale[col0 + threadIdx.x] = pow2f(e - 13 - 14);  // also the basis' 2^14
    }
    for (int j0 = 4 * gt; j0 < Kf && nf > 0; j0 += 4 * TG) {
      float ep[FOLD2_FR][4], em[FOLD2_FR][4], op[FOLD2_FR][4], om[FOLD2_FR][4];
      (void)compute(j0, ep, em, op, om);
      emit(j0, ep, em, op, om, scale);
    }
  }
  if (le) {
#pragma unroll
    for (int f = 0; f < FOLD2_FR; ++f) {
      const float se = wave_sum_f32(pe[f]), so = wave_sum_f32(po[f]);
      if ((threadIdx.x & 63) == 0) {
        red[(0 * 4 + (threadIdx.x >> 6)) * FOLD2_FR + f] = se;
        red[(1 * 4 + (threadIdx.x >> 6)) * FOLD2_FR + f] = so;
      }
    }
  }
  __syncthreads();
  {
    // per frame Kf pieces of 16 bytes: Kf / 2 of the even operand, then Kf / 2 of the odd one
    const int per = Kf / 2;
    f32x4v *out_e = reinterpret_cast<f32x4v *>(dst + col0 * ((long long)Kf * 4));
    f32x4v *out_o = reinterpret_cast<f32x4v *>(dst + p.fold2_xs_odd + col0 * ((long long)Kf * 4));
    const f32x4v *src = reinterpret_cast<const f32x4v *>(smem_raw);
    const int pieces = nfw * Kf;
    for (int i = threadIdx.x; i < pieces; i += 256) {
      const int fw = i / Kf, rest = i - fw * Kf;
      const int par = rest >= per, pc = rest - par * per;
      (par ? out_o : out_e)[(long long)fw * per + pc] = src[i ^ ((i >> 3) & 7)];
    }
  }
  if (!le) return;
  const int bin = p.fold_last_bin;  // relative to the problem's first bin
  const float sc = p.row_scale ? p.row_scale[bin] : 1.f;
  const int wpg = 4 / G;  // waves per group
  auto frame_value = [&](int fw, float &re, float &im) __attribute__((always_inline)) {
    const int g = fw / FOLD2_FR, f = fw - g * FOLD2_FR;
    re = im = 0.f;
    for (int w = g * wpg; w < (g + 1) * wpg; ++w) {
      re += red[(0 * 4 + w) * FOLD2_FR + f];
      im += red[(1 * 4 + w) * FOLD2_FR + f];
    }
    re *= sc;
    im *= p.im_sign * sc;
  };
  if ((int)threadIdx.x < nfw) {  // one thread per frame
    float re, im;
    frame_value(threadIdx.x, re, im);
    const int E = epilogue_width(p.epilogue);
    float *d = p.out + (long long)c * p.out_clip_stride +
               (long long)(p.out_row_offset + bin) * p.out_row_stride + (long long)(tw0 + threadIdx.x) * E;
    epilogue_store(p, d, re, im);
  }
}
