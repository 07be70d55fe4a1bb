// framed_fold.inl -- MISPEC_PREC_BF16X3 for Fourier-type bases: the contraction over HALF the taps.
// Included by mispec.hip inside its anonymous namespace, after framed_bf16x3.inl (shares KParams,
// the XCD-aware tile order, the split helpers and the planar / fused-filterbank epilogues).
//
// The reference's STFT bases (utils.py:379-389 times the centred periodic window, stft.py:230-232)
// are even (cos rows) / odd (sin rows) about tap N/2, N = n_fft:
//     wcos[k, N-n] = wcos[k, n],   wsin[k, N-n] = -wsin[k, n]          (1 <= n < N/2)
// so with  E_t[n] = x_t[n] + x_t[N-n],  O_t[n] = x_t[n] - x_t[N-n]  (x_t = frame t of the padded
// signal; E = O = x_t[N/2] at n = N/2 and x_t[0] at n = 0, the two taps without a partner)
//     re[k, t] = sum_{n in fold} wcos[k, n] * E_t[n],      im[k, t] = sum_{n in fold} wsin[k, n] * O_t[n]
// over N/2 (+1) taps instead of N: HALF the MFMAs of the dense contraction for the same result up to
// rounding (the caller verifies the symmetry of the basis numerically before offering this path,
// engine.fold_basis; the coefficient actually used is the mean of the pair).
//
// Operands (bf16, 128-byte "stage rows": 16 folded taps x 4 planes, so one K stage of one row is one
// cache line and one LDS-DMA piece is 8 whole lines):
//   folded basis : per bin, per stage s : [re_hi 16 | re_lo 16 | im_hi 16 | im_lo 16]   (fold_basis_kernel,
//                  once per basis)  + the fp32 folded rows of the LAST bin (the Nyquist bin of an
//                  n_fft/2+1 STFT, which the pre-pass evaluates itself)
//   folded frames: per flat frame (clip, t), per stage s : [E_hi 16 | E_lo 16 | O_hi 16 | O_lo 16]
//                  (fold_frames_kernel, every call: reads the fp32 waveform with the virtual
//                  padding, 8 B out per folded tap; frames do not overlap any more, hop is free)
// Folded tap j <-> n = j + 1 for j < N/2 (n = 1 .. N/2), j = N/2 <-> n = 0 (only when some basis row
// has a non-zero tap 0, i.e. a window with w[0] != 0), then zero taps up to a multiple of 16.
//
// Main kernel: 128 bins x 256 frames per workgroup, 8 waves as 2 x 4 (wave = 64 bins x 64 frames, re
// and im accumulators of the same bins: acc[m][0][n] / acc[m][1][n]), K stage = 16 folded taps = ONE MFMA step,
// LDS stage = 128 A rows + 256 X rows of 128 B (48 KB), ring of 3.  An iteration is two halves:
//     re half : 12 MFMAs acc[m][0][n] += E x A_re   (3 split terms x 2 x 2 tiles) while the im-half
//               fragments of this stage are read
//     barrier : stage c+1 has landed (s_waitcnt vmcnt(6): stage c+2 may still be in flight), every
//               wave has read what it needs from stage c
//     im half : 12 MFMAs acc[m][1][n] += O x A_im while stage c+3 is DMA'd into the buffer of stage c and
//               the re-half fragments of stage c+1 are read
// Rows are XOR-swizzled in 16-byte chunks by (row >> 1) & 7 (applied to the DMA source address and to
// the fragment reads): conflict-free ds_read_b128 (MI355X_MICROARCH.md, LDS table).

constexpr int FOLD_KC = 16;                 // folded taps per stage
constexpr float FOLD_ASCALE = 16384.f;      // FOLD_F16X3: basis coefficients x 2^14 (undone with the frames' scale)
constexpr int FOLD_ROWB = 128;              // bytes of one stage row: 4 planes x 16 bf16
constexpr int FOLD_BINS = 128;              // bins per workgroup
constexpr int FOLD_BN = 256;                // frames per workgroup
constexpr int FOLD_NBUF = 3;
constexpr int FOLD_A_ST = FOLD_BINS * FOLD_ROWB;            // 16 KB
constexpr int FOLD_X_ST = FOLD_BN * FOLD_ROWB;              // 32 KB
constexpr int FOLD_STAGE = FOLD_A_ST + FOLD_X_ST;           // 48 KB
constexpr int FOLD_DMA_PER_WAVE = (FOLD_STAGE / 1024) / 8;  // LDS-DMA pieces per wave and stage (6)

// (hi, lo) of two floats with the hardware conversion (v_cvt_pk_bf16_f32, round to nearest even):
// the same values as bf16_split for every finite input inside the bf16 range
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bf16_split2(float a, float b, unsigned &hi, unsigned &lo) {
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r = v - __builtin_convertvector(h, f32x2);
  const bf16x2 l = __builtin_convertvector(r, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}

__host__ __device__ inline int fold_taps(int kernel, int with_tap0) {
  const int k = kernel / 2 + (with_tap0 ? 1 : 0);
  return (k + FOLD_KC - 1) / FOLD_KC * FOLD_KC;
}

// original tap of folded tap j (or -1 for the zero padding)
__device__ __forceinline__ int fold_tap_of(int j, int N, int with_tap0) {
  const int H = N >> 1;
  if (j < H) return j + 1;
  if (j == H && with_tap0) return 0;
  return -1;
}

// ---------------------------------------------------------------------------------
// basis -> folded split planes (+ asymmetry statistics).  grid (ceil(Kf/256), n_bins)
// stats[0] = max over pairs of |wr[n] - wr[N-n]| / 2 and |wi[n] + wi[N-n]| / 2 (what the fold
// neglects), stats[1] = max |coefficient|, both as float bit patterns (atomicMax on unsigned).
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fold_basis_kernel(const float *__restrict__ re,
                                                         const float *__restrict__ im,
                                                         long long row_stride, int n_bins, int N,
                                                         int with_tap0, int Kf,
                                                         unsigned short *__restrict__ dst,
                                                         float *__restrict__ last_rows,
                                                         unsigned *__restrict__ stats, int arith) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int bin = blockIdx.y;
  float asym = 0.f, amax = 0.f;
  if (j < Kf) {
    const int n = fold_tap_of(j, N, with_tap0);
    const float *wr = re + (long long)bin * row_stride;
    const float *wi = im + (long long)bin * row_stride;
    float ae = 0.f, ao = 0.f;
    if (n >= 0) {
      if (n > 0 && n < (N >> 1)) {
        const float r1 = wr[n], r2 = wr[N - n], i1 = wi[n], i2 = wi[N - n];
        ae = 0.5f * (r1 + r2);
        ao = 0.5f * (i1 - i2);
        asym = fmaxf(fabsf(0.5f * (r1 - r2)), fabsf(0.5f * (i1 + i2)));
      } else {  // n = N/2 or n = 0: no partner, E = O = the sample itself
        ae = wr[n];
        ao = wi[n];
      }
      amax = fmaxf(fabsf(ae), fabsf(ao));
    }
    unsigned short *row = dst + ((long long)bin * (Kf / FOLD_KC) + j / FOLD_KC) * (FOLD_ROWB / 2);
    const int u = j % FOLD_KC;
    if (arith == FOLD_F32) {  // MISPEC_PREC_F32: the same 128-byte stage rows as [re 16 floats | im 16 floats]
      float *frow = reinterpret_cast<float *>(row);
      frow[u] = ae;
      frow[16 + u] = ao;
    } else {
      unsigned eh, el, oh, ol;
      if (arith == FOLD_F16X3) {  // (hi, lo) fp16 pairs of coefficient x 2^14 (the caller checks |c| <= 2)
        unsigned h2, l2;
        f16_split2(ae * FOLD_ASCALE, ao * FOLD_ASCALE, h2, l2);
        eh = h2 & 0xffff, oh = h2 >> 16, el = l2 & 0xffff, ol = l2 >> 16;
      } else {
        bf16_split(ae, eh, el);
        bf16_split(ao, oh, ol);
      }
      row[u] = (unsigned short)eh;
      row[16 + u] = (unsigned short)el;
      row[32 + u] = (unsigned short)oh;
      row[48 + u] = (unsigned short)ol;
    }
    if (bin == n_bins - 1) {
      last_rows[j] = ae;
      last_rows[Kf + j] = ao;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    asym = fmaxf(asym, __shfl_xor(asym, d));
    amax = fmaxf(amax, __shfl_xor(amax, d));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&stats[0], __float_as_uint(asym));
    atomicMax(&stats[1], __float_as_uint(amax));
  }
}

// ---------------------------------------------------------------------------------
// Pre-pass: one workgroup per FOLD_FR * G consecutive frames of a clip, G = fold_groups(Kf) thread
// groups of 256 / G threads with FOLD_FR frames each (short frames: 1024 taps keep one group of 256
// threads busy, 512 taps two groups of 128, ...).  Thread i of a group owns folded taps 4 i .. 4 i + 3
// (+ 4 * group size per trip) of each of its frames: 16-byte loads of x_t[n ..] forwards and
// x_t[N-n ..] backwards, all frames' loads issued before the first use (element-wise with the
// virtual padding for the frames that touch a clip edge), E / O in fp32, split, assembled in LDS and
// stored as consecutive 16-byte pieces (a frame's 8 Kf bytes are one run of memory).  The LDS
// staging rows (128 B = all 32 banks) keep their 16-byte pieces XOR-swizzled by row & 7: a wave's
// 8-byte writes cover 16 rows x the same two pieces, unswizzled a 16-way bank conflict (3.7e7
// conflict cycles per cfg3 step).  With p.fold_last the LAST bin (Nyquist) is evaluated here too, in
// plain fp32 FMAs on the same E / O, through the full pointwise epilogue -- instead of a row block
// of its own in the contraction.
// ---------------------------------------------------------------------------------
constexpr int FOLD_FR = 4;

// sum over the 64 lanes, the same value in every lane: quad permutes, half-row and row mirrors (DPP,
// no LDS crossbar), then the four rows through scalar registers
__device__ __forceinline__ float wave_sum_f32(float v) {
#define FOLD_DPP_ADD(ctrl) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
  FOLD_DPP_ADD(0xB1);   // quad_perm [1, 0, 3, 2]
  FOLD_DPP_ADD(0x4E);   // quad_perm [2, 3, 0, 1]
  FOLD_DPP_ADD(0x141);  // row_half_mirror
  FOLD_DPP_ADD(0x140);  // row_mirror
#undef FOLD_DPP_ADD
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

__host__ __device__ inline int fold_groups(int Kf) { return Kf <= 256 ? 4 : (Kf <= 512 ? 2 : 1); }

__global__ void __launch_bounds__(256) fold_frames_kernel(const KParams p,
                                                          unsigned short *__restrict__ dst) {
  const int N = p.K, H = N >> 1, Kf = p.Ks;
  const int G = fold_groups(Kf), TG = 256 / G;
  const int grp = threadIdx.x / TG, gt = threadIdx.x - grp * TG;
  const int c = p.fold_clip0 + blockIdx.y;
  const int tw0 = blockIdx.x * (FOLD_FR * G);  // first frame of the workgroup
  const int nfw = p.n_frames - tw0 < FOLD_FR * G ? p.n_frames - tw0 : FOLD_FR * G;
  const int t0 = tw0 + grp * FOLD_FR;          // first frame of this thread's group
  int nf = p.n_frames - t0;                    // frames of the group (<= 0: an idle group of the last block)
  nf = nf < FOLD_FR ? nf : FOLD_FR;
  const float *x = p.x + (long long)c * p.x_clip_stride;
  const long long col0 = (long long)c * p.n_frames + tw0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // (benchmarking build, 0x200: without the Nyquist bin)
  const float *le = MISPEC_DBG(p, 0x200) ? nullptr : p.fold_last, *lo = le ? le + Kf : nullptr;
  float pe[FOLD_FR], po[FOLD_FR];
#pragma unroll
  for (int f = 0; f < FOLD_FR; ++f) pe[f] = po[f] = 0.f;
  // all frames of the group interior? (then every load is a plain 16-byte run)
  const long long qa = (long long)t0 * p.hop - p.pad;
  const bool interior = qa >= 0 && qa + (long long)(nf - 1) * p.hop + N <= p.n_samples;
  const int rows_f = Kf / FOLD_KC;  // staging rows per frame
  // one trip's values: folded taps j0 .. j0 + 3 of the group's frames (returns their largest magnitude)
  auto compute = [&](int j0, float (&e)[FOLD_FR][4], float (&o)[FOLD_FR][4]) __attribute__((always_inline)) -> float {
    if (interior && j0 + 4 <= H) {  // taps n = j0+1 .. j0+4 <= N/2: paired, except n = N/2 itself
      f32x4u fw[FOLD_FR], bw[FOLD_FR];
#pragma unroll
      for (int f = 0; f < FOLD_FR; ++f) {
        const long long q0 = qa + (long long)(f < nf ? f : 0) * p.hop;
        if (MISPEC_DBG(p, 0x80)) {  // benchmarking build: no global loads
          fw[f] = f32x4u{(float)j0, 1.f, 2.f, (float)f};
          bw[f] = f32x4u{3.f, (float)gt, 2.f, 1.f};
          continue;
        }
        fw[f] = *reinterpret_cast<const f32x4u *>(x + q0 + j0 + 1);
        bw[f] = *reinterpret_cast<const f32x4u *>(x + q0 + N - j0 - 4);
      }
      // (the quad that ends at n = N/2 used to take the element-wise path below: one lane per
      // frame group, but its wave then walked 32 dependent loads in every workgroup -- the pre-pass
      // ran at the same speed with all its other loads and stores removed)
      const bool mid = j0 + 4 == H;  // tap N/2 has no partner: E = O = the sample itself
#pragma unroll
      for (int f = 0; f < FOLD_FR; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          e[f][i] = fw[f][i] + bw[f][3 - i];
          o[f][i] = fw[f][i] - bw[f][3 - i];
          if (i == 3 && mid) e[f][i] = o[f][i] = fw[f][i];
        }
    } else if (interior && j0 >= H) {  // behind the paired taps: tap 0 (if carried), then zero padding
#pragma unroll
      for (int f = 0; f < FOLD_FR; ++f) {
        const long long q0 = qa + (long long)(f < nf ? f : 0) * p.hop;
        const float a = (j0 == H && p.fold_tap0) ? x[q0] : 0.f;
        e[f][0] = o[f][0] = a;
        e[f][1] = e[f][2] = e[f][3] = 0.f;
        o[f][1] = o[f][2] = o[f][3] = 0.f;
      }
    } else {
#pragma unroll
      for (int f = 0; f < FOLD_FR; ++f) {
        const long long q0 = qa + (long long)(f < nf ? f : 0) * p.hop;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int n = fold_tap_of(j0 + i, N, p.fold_tap0);
          float a = 0.f, b = 0.f;
          if (n >= 0)
            a = fetch_sample(p.x, (long long)c * p.x_clip_stride, (int)(q0 + n), p.n_samples, p.pad_mode, true);
          if (n > 0 && n < H) {
            b = fetch_sample(p.x, (long long)c * p.x_clip_stride, (int)(q0 + N - n), p.n_samples, p.pad_mode, true);
            e[f][i] = a + b;
            o[f][i] = a - b;
          } else {
            e[f][i] = a;
            o[f][i] = a;
          }
        }
      }
    }
    float mx = 0.f;
    if (p.fold_arith == FOLD_F16X3) {
#pragma unroll
      for (int f = 0; f < FOLD_FR; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) mx = fmaxf(mx, fmaxf(fabsf(e[f][i]), fabsf(o[f][i])));
    }
    return mx;
  };
  // the last bin's partial sums, the split and the staging of one trip's values
  auto emit = [&](int j0, const float (&e)[FOLD_FR][4], const float (&o)[FOLD_FR][4], float scale)
                  __attribute__((always_inline)) {
    f32x4v we = {0.f, 0.f, 0.f, 0.f}, wo = {0.f, 0.f, 0.f, 0.f};
    if (le) {
      we = *reinterpret_cast<const f32x4v *>(le + j0);
      wo = *reinterpret_cast<const f32x4v *>(lo + j0);
    }
#pragma unroll
    for (int f = 0; f < FOLD_FR; ++f) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pe[f] = fmaf(we[i], e[f][i], pe[f]);
        po[f] = fmaf(wo[i], o[f][i], po[f]);
      }
      // staging row of (frame, stage) and its swizzle; u = first of this thread's 4 taps in the stage
      const int R = (grp * FOLD_FR + f) * rows_f + j0 / FOLD_KC, sw = R & 7, u = j0 % FOLD_KC;
      unsigned char *r = smem_raw + (size_t)R * FOLD_ROWB;
      if (p.fold_f32) {  // MISPEC_PREC_F32: stage row = [E 16 floats | O 16 floats], pieces of 4 floats
        const f32x4v ev = {e[f][0], e[f][1], e[f][2], e[f][3]}, ov = {o[f][0], o[f][1], o[f][2], o[f][3]};
        *reinterpret_cast<f32x4v *>(r + (((u >> 2)) ^ sw) * 16) = ev;
        *reinterpret_cast<f32x4v *>(r + ((4 + (u >> 2)) ^ sw) * 16) = ov;
      } else {  // [E_hi 16 | E_lo 16 | O_hi 16 | O_lo 16] bf16 / scaled fp16: plane P = pieces 2P, 2P+1 of 8 taps
        uint2 eh, el, oh, ol;
        if (p.fold_arith == FOLD_F16X3) {
          f16_split2(e[f][0] * scale, e[f][1] * scale, eh.x, el.x);
          f16_split2(e[f][2] * scale, e[f][3] * scale, eh.y, el.y);
          f16_split2(o[f][0] * scale, o[f][1] * scale, oh.x, ol.x);
          f16_split2(o[f][2] * scale, o[f][3] * scale, oh.y, ol.y);
        } else {
          bf16_split2(e[f][0], e[f][1], eh.x, el.x);
          bf16_split2(e[f][2], e[f][3], eh.y, el.y);
          bf16_split2(o[f][0], o[f][1], oh.x, ol.x);
          bf16_split2(o[f][2], o[f][3], oh.y, ol.y);
        }
        const int h = u >> 3, sub = (u & 4) * 2;
        *reinterpret_cast<uint2 *>(r + ((0 + h) ^ sw) * 16 + sub) = eh;
        *reinterpret_cast<uint2 *>(r + ((2 + h) ^ sw) * 16 + sub) = el;
        *reinterpret_cast<uint2 *>(r + ((4 + h) ^ sw) * 16 + sub) = oh;
        *reinterpret_cast<uint2 *>(r + ((6 + h) ^ sw) * 16 + sub) = ol;
      }
    }
  };
  float(*red)[4][FOLD_FR] =
      reinterpret_cast<float(*)[4][FOLD_FR]>(smem_raw + (size_t)FOLD_FR * G * Kf * 8);
  // workgroup maximum of a per-thread value (slot [2] of the reduction area; idle groups take part)
  auto wg_max = [&](float m) __attribute__((always_inline)) -> float {
    float *rm = &red[2][0][0];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) rm[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
  };
  if (p.fold_arith == FOLD_F16X3 && Kf == 4 * TG) {
    // MISPEC_PREC_F16X3, one trip per thread: the values first, then the scale 2^(15-e) from the
    // workgroup's largest magnitude m = f 2^e (the fp16 pairs stay below 2^15), then split and stage
    float e[FOLD_FR][4], o[FOLD_FR][4];
    float m = 0.f;
    if (nf > 0) m = compute(4 * gt, e, o);
    m = wg_max(m);
    const int ex = absmax_exponent(m);
    if ((int)threadIdx.x < nfw) p.col_unscale[col0 + threadIdx.x] = pow2f(ex - 15 - 14);  // also the basis' 2^14
    if (nf > 0) emit(4 * gt, e, o, pow2f(15 - ex));
  } else {
    float scale = 1.f;
    if (p.fold_arith == FOLD_F16X3) {
      // any other shape: the scale from the largest |sample| the workgroup reads (|E|, |O| <= 2 max)
      const long long wa = (long long)tw0 * p.hop - p.pad, wb = wa + (long long)(nfw - 1) * p.hop + N;
      float m = 0.f;
      for (long long q = wa + threadIdx.x; q < wb; q += 256)
        m = fmaxf(m, fabsf(fetch_sample(p.x, (long long)c * p.x_clip_stride, (int)q, p.n_samples, p.pad_mode, true)));
      const int ex = absmax_exponent(wg_max(m));
      scale = pow2f(14 - ex);
      if ((int)threadIdx.x < nfw) p.col_unscale[col0 + threadIdx.x] = pow2f(ex - 14 - 14);
    }
    for (int j0 = 4 * gt; j0 < Kf && nf > 0; j0 += 4 * TG) {
      float e[FOLD_FR][4], o[FOLD_FR][4];
      (void)compute(j0, e, o);
      emit(j0, e, o, scale);
    }
  }
  // the last bin's partial sums: wave totals to LDS before the barrier the staging needs anyway
  if (le) {
#pragma unroll
    for (int f = 0; f < FOLD_FR; ++f) {
      const float se = wave_sum_f32(pe[f]), so = wave_sum_f32(po[f]);
      if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6][f] = se;
        red[1][threadIdx.x >> 6][f] = so;
      }
    }
  }
  __syncthreads();
  {
    f32x4v *out = reinterpret_cast<f32x4v *>(dst + col0 * ((long long)Kf * 4));
    const int pieces = nfw * (Kf / 2);  // Kf * 8 bytes per frame = Kf / 2 pieces of 16
    const f32x4v *src = reinterpret_cast<const f32x4v *>(smem_raw);
    for (int i = threadIdx.x; i < pieces; i += 256) {
      if (MISPEC_DBG(p, 0x40) && src[i][0] != 12345.678f) continue;  // benchmarking build: no global stores
      out[i] = src[i ^ ((i >> 3) & 7)];
    }
  }
  if (!le) return;
  const int bin = p.fold_last_bin;  // relative to the problem's first bin
  const float sc = p.row_scale ? p.row_scale[bin] : 1.f;
  const int wpg = 4 / G;  // waves per group
  auto frame_value = [&](int fw, float &re, float &im) __attribute__((always_inline)) {
    const int g = fw / FOLD_FR, f = fw - g * FOLD_FR;
    re = im = 0.f;
    for (int w = g * wpg; w < (g + 1) * wpg; ++w) {
      re += red[0][w][f];
      im += red[1][w][f];
    }
    re *= sc;
    im *= p.im_sign * sc;
  };
  if (p.fb) {
    // fused filterbank: out[c, m, t] += fb[m, bin] * |z|^power for the filters that weigh this
    // bin (the main kernel treats their bands as crossing a tile boundary: atomic addends there too)
    const int babs = p.out_row_offset + bin;
    for (int m = threadIdx.x; m < p.n_fb; m += 256) {
      if (p.fb_support[2 * m] <= babs && babs < p.fb_support[2 * m + 1]) {
        const float w = p.fb[(long long)m * p.fb_row_stride + babs];
        if (w != 0.f) {
          for (int fw = 0; fw < nfw; ++fw) {
            float re, im;
            frame_value(fw, re, im);
            const float s2 = re * re + im * im + p.eps;
            const float pw = p.power == 2.0f ? s2 : sqrtf(s2);
            unsafeAtomicAdd(p.out + (long long)c * p.out_clip_stride + (long long)m * p.out_row_stride + tw0 + fw, w * pw);
          }
        }
      }
    }
  } else if ((int)threadIdx.x < nfw) {  // one thread per frame
    float re, im;
    frame_value(threadIdx.x, re, im);
    const int E = epilogue_width(p.epilogue);
    float *d = p.out + (long long)c * p.out_clip_stride +
               (long long)(p.out_row_offset + bin) * p.out_row_stride + (long long)(tw0 + threadIdx.x) * E;
    epilogue_store(p, d, re, im);
  }
}

// Barrier that publishes LDS-direct data, leaving the newest N loads of this wave in flight.  Stated
// as instructions: __syncthreads() carries a workgroup-scope fence, for which hipcc waits for ALL
// outstanding LDS-direct loads (vmcnt(0)) when it sees one issued earlier in the same block.
template <int N>
__device__ __forceinline__ void lds_dma_barrier_keep() {
  static_assert(N == 0 || N == 4 || N == 6, "immediate of the s_waitcnt below");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One tile of the contraction: 128 bins x (64 * NR) frames.  NR = 4 is the kernel described above;
// NR = 2 (128 frames: 32 KB stages, 4 DMA pieces per wave, half the MFMAs per barrier) serves the
// frames left over behind the last whole round of workgroups -- see framed_fold_kernel.
// F32 = the same kernel in the exact arithmetic (MISPEC_PREC_F32): stage rows of [re | im] resp.
// [E | O] fp32 taps, a lane's fragment = taps 8 lh .. 8 lh + 7 (two 16-byte chunks), eight
// v_mfma_f32_32x32x2_f32 per 16 taps and frame tile (MFMA p contracts taps p and 8 + p) instead of
// three bf16 ones: the dense fp32 kernel's MFMA count halved, everything else unchanged.
// ARITH = FOLD_F16X3: the bf16x3 kernel on v_mfma_f32_32x32x16_f16 -- the operands are (hi, lo) fp16
// pairs of power-of-two scaled values (22 significant bits instead of 16; framed_fold2.inl), and the
// accumulators are multiplied by the frames' col_unscale before the epilogue.
template <int NR, int ARITH>
__device__ __forceinline__ void framed_fold_body(const KParams &p, const int tile_m, const long long n0) {
  constexpr bool F32 = ARITH == FOLD_F32;
  // wave layout: 2 x 4 waves, a wave owns MRW = 2 bin tiles x NRW = NR / 2 frame tiles (64 bins x 64
  // frames of the 256-frame tile), re and im accumulators of both: per stage and wave 2 (MRW + NRW)
  // = 8 fragment reads per half for 12 MFMAs.  (Until round 2's last day: 4 x 2 waves of 32 bins x
  // 128 frames, 10 reads per half -- with the stage DMA more LDS cycles than MFMA cycles.)
  constexpr int MRW = 2, WM = 2, WN = 4, NW = 8;
  constexpr int BN = 64 * NR;                            // frames of the tile
  constexpr int NRW = BN / (32 * WN);                    // frame tiles per wave
  static_assert(NRW >= 1, "tile too narrow for 4 wave columns");
  constexpr int X_ST = BN * FOLD_ROWB;                   // bytes of the frame rows of a stage
  constexpr int STAGE = FOLD_A_ST + X_ST;
  constexpr int XJ = BN / 64;                            // frame-row DMA pieces per wave
  constexpr int DMA_PER_WAVE = 2 + XJ;
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int b0 = tile_m * FOLD_BINS;                  // first bin of the tile
  const int nst = p.Ks / FOLD_KC;
  const long long row_el = (long long)nst * (FOLD_ROWB / 2);  // elements per basis / frame row

  // ---- DMA geometry: a piece = 8 rows x 128 B; lane -> (row r8 = lane >> 3, slot = lane & 7),
  // which receives chunk  slot ^ ((row >> 1) & 7)  of its row (row = index inside the tile)
  const int r8 = lane >> 3, slot = lane & 7;
  const unsigned short *aptr[2], *xptr[XJ];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = (j * NW + wave) * 8 + r8;  // A row = bin of the tile
    int bin = b0 + row;
    bin = bin < p.n_bins ? bin : p.n_bins - 1;  // bins past the end feed unused accumulators
    aptr[j] = p.as + (long long)bin * row_el + 8 * (slot ^ ((row >> 1) & 7));
  }
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int row = (j * NW + wave) * 8 + r8;  // X row = frame of the tile
    long long col = n0 + row;
    col = col < p.n_cols ? col : 0;  // unused column: any valid frame, never stored
    xptr[j] = p.xs + col * row_el + 8 * (slot ^ ((row >> 1) & 7));
  }
  auto dma_stage = [&](int s, int buf) __attribute__((always_inline)) {
    unsigned char *st = smem_raw + buf * STAGE;
    const int so = s * (FOLD_ROWB / 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(aptr[j] + so), (lptr_t)(st + (j * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < XJ; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(xptr[j] + so),
                                       (lptr_t)(st + FOLD_A_ST + (j * NW + wave) * 1024), 16, 0, 0);
  };

  f32x16 acc[MRW][2][NRW];  // [bin tile][re / im][frame tile]
#pragma unroll
  for (int m = 0; m < MRW; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int n = 0; n < NRW; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][h][n][e] = 0.f;

  // ---- fragments.  Lane (li, lh) supplies row li and taps 8 lh .. 8 lh + 7 of the step: chunk
  // 2 * plane + lh of the row, at slot  chunk ^ ((li >> 1) & 7)
  const int fsw = (li >> 1) & 7;
  const int a_row = ((wm * MRW) * 32 + li) * FOLD_ROWB;
  const int x_row = FOLD_A_ST + ((wn * NRW) * 32 + li) * FOLD_ROWB;
  bf16x8 fa[2][2][MRW], fx[2][2][NRW];  // [half][hi / lo][tile]
  auto load_frags = [&](int buf, auto half_tag) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half_tag)::value;  // 0: (A_re, E), 1: (A_im, O)
    const unsigned char *st = smem_raw + buf * STAGE;
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int off = 16 * ((F32 ? 4 * HALF + 2 * lh + pl : 4 * HALF + 2 * pl + lh) ^ fsw);
#pragma unroll
      for (int m = 0; m < MRW; ++m)
        fa[HALF][pl][m] = *reinterpret_cast<const bf16x8 *>(st + a_row + m * 32 * FOLD_ROWB + off);
#pragma unroll
      for (int n = 0; n < NRW; ++n)
        fx[HALF][pl][n] = *reinterpret_cast<const bf16x8 *>(st + x_row + n * 32 * FOLD_ROWB + off);
    }
  };
  // the 3 * MRW * NRW MFMAs of one half; small terms first, an accumulator is revisited after
  // MRW * NRW - 1 others
  auto mfma_half = [&](auto half_tag) __attribute__((always_inline)) {
    constexpr int HALF = decltype(half_tag)::value;
    if (F32) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
          for (int n = 0; n < NRW; ++n) {
            const float a = __builtin_bit_cast(f32x4v, fa[HALF][t >> 2][m])[t & 3];
            const float x = __builtin_bit_cast(f32x4v, fx[HALF][t >> 2][n])[t & 3];
            acc[m][HALF][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, a, acc[m][HALF][n], 0, 0, 0);
          }
      return;
    }
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int m = 0; m < MRW; ++m)
#pragma unroll
        for (int n = 0; n < NRW; ++n) {
          const bf16x8 a = fa[HALF][term == 0 ? 1 : 0][m];
          const bf16x8 x = fx[HALF][term == 1 ? 1 : 0][n];
          if (ARITH == FOLD_F16X3)
            acc[m][HALF][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, a), acc[m][HALF][n], 0, 0, 0);
          else
            acc[m][HALF][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, a, acc[m][HALF][n], 0, 0, 0);
        }
  };
  auto interleave = [&](auto n_mfma_tag, auto n_ds_tag, auto n_vm_tag) __attribute__((always_inline)) {
    constexpr int NM = decltype(n_mfma_tag)::value;
    constexpr int ND = decltype(n_ds_tag)::value;
    constexpr int NV = decltype(n_vm_tag)::value;
    constexpr int NMD = NM - NM / 4;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      if ((i + 1) * NV / NM != i * NV / NM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (i < NMD && (i + 1) * ND / NMD != i * ND / NMD)
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  using std::integral_constant;
  typedef integral_constant<int, 0> i0;
  typedef integral_constant<int, 1> i1;
  typedef integral_constant<int, (F32 ? 8 : 3) * MRW * NRW> n_mfma;
  typedef integral_constant<int, 2 * (MRW + NRW)> n_reads;
  typedef integral_constant<int, DMA_PER_WAVE> n_dma;

  // One iteration = stage c out of buffer `buf`.  Waves w and w + 4 share a SIMD (a workgroup's
  // waves are dealt to the SIMDs cyclically): so that the two never sit in their LDS-DMA issue at
  // the same time (a piece costs its wave 60-150 issue cycles, during which only the OTHER wave can
  // feed the matrix pipe), the "early" waves 0-3 issue stage c+3 in the im half of iteration c (into
  // the buffer of stage c, free since this iteration's barrier) and the "late" waves 4-7 issue
  // stage c+2 in the re half (into the buffer of stage c-1, free since the previous barrier).
  // Either way a stage is issued >= 1.5 iterations before the barrier that publishes it, and at
  // that barrier a wave may leave exactly its newest stage in flight (vmcnt(6)).
  // NEXT: stage c+1 exists; KEEP: loads left in flight at the barrier.
  const bool late = wave >= NW / 2;
  auto stage_iter = [&](int c, int buf, int buf_prev, int buf_next, bool dma_late, bool dma_early,
                        auto next_tag, auto keep_tag) __attribute__((always_inline)) {
    constexpr bool NEXT = decltype(next_tag)::value;
    if (!MISPEC_DBG(p, 8)) load_frags(buf, i1{});
    if (dma_late && !MISPEC_DBG(p, 1)) dma_stage(c + 2, buf_prev);
    mfma_half(i0{});
    interleave(n_mfma{}, n_reads{}, i0{});
    if (!MISPEC_DBG(p, 4)) lds_dma_barrier_keep<decltype(keep_tag)::value>();
    if (dma_early && !MISPEC_DBG(p, 1)) dma_stage(c + FOLD_NBUF, buf);
    if (NEXT && !MISPEC_DBG(p, 8)) load_frags(buf_next, i0{});
    mfma_half(i1{});
    interleave(n_mfma{}, integral_constant<int, NEXT ? n_reads::value : 0>{}, i0{});
  };
  typedef integral_constant<bool, true> yes;
  typedef integral_constant<bool, false> no;
  typedef integral_constant<int, DMA_PER_WAVE> keep1;
  if (nst > 0) {
    dma_stage(0, 0);
    if (nst > 1) dma_stage(1, 1);
    if (nst > 2) dma_stage(2, 2);
    // stage 0 landed: at most the later stages' loads outstanding (DMA_PER_WAVE each)
    if (nst > 2) {
      if (DMA_PER_WAVE == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (nst > 1) {
      if (DMA_PER_WAVE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    load_frags(0, i0{});
    auto nxt = [](int b) { return b == FOLD_NBUF - 1 ? 0 : b + 1; };
    auto prv = [](int b) { return b == 0 ? FOLD_NBUF - 1 : b - 1; };
    int c = 0, buf = 0;
    for (; c + 2 < nst; ++c) {  // stages c+1 and c+2 exist; c+2 is in flight at the barrier
      stage_iter(c, buf, prv(buf), nxt(buf), late && c >= 1, !late && c + FOLD_NBUF < nst, yes{}, keep1{});
      buf = nxt(buf);
    }
    if (c + 1 < nst) {  // only c+1 left: wait for everything
      stage_iter(c, buf, prv(buf), nxt(buf), false, false, yes{}, i0{});
      buf = nxt(buf);
      ++c;
    }
    stage_iter(c, buf, buf, buf, false, false, no{}, i0{});
    __syncthreads();  // every wave is done with the stage buffers (the epilogue reuses them)
  }
  if (MISPEC_DBG(p, 0x40000)) {  // ablation: no epilogue (keep the accumulators alive)
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MRW; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < NRW; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) sum += acc[m][h][n][e];
    if (sum == 12345.678f) p.out[0] = sum;
    return;
  }
  if (ARITH == FOLD_F16X3 || p.col_add) {
    // FOLD_F16X3: undo the operand scaling; second fold: add tap 0's term to the real parts.
    // acc[..][n][4 g + i] belongs to frame 32 n + 8 g + 4 lh + i of the wave's block
#pragma unroll
    for (int n = 0; n < NRW; ++n)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const long long col = n0 + (wn * NRW + n) * 32 + 8 * g + 4 * lh;
        f32x4v u = {1.f, 1.f, 1.f, 1.f}, ad = {0.f, 0.f, 0.f, 0.f};
        if (col + 3 < p.n_cols) {
          // (4-byte aligned vectors: the arrays start wherever n_cols puts them)
          if (ARITH == FOLD_F16X3) u = *reinterpret_cast<const f32x4u *>(p.col_unscale + col);
          if (p.col_add) ad = *reinterpret_cast<const f32x4u *>(p.col_add + col);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const long long ci = col + i < p.n_cols ? col + i : 0;
            if (ARITH == FOLD_F16X3) u[i] = p.col_unscale[ci];
            if (p.col_add) ad[i] = p.col_add[ci];
          }
        }
#pragma unroll
        for (int m = 0; m < MRW; ++m)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (ARITH == FOLD_F16X3) {
              acc[m][0][n][4 * g + i] = fmaf(acc[m][0][n][4 * g + i], u[i], ad[i]);
              acc[m][1][n][4 * g + i] *= u[i];
            } else {
              acc[m][0][n][4 * g + i] += ad[i];
            }
          }
      }
  }
  // the epilogues of the dense planar kernel, once per bin tile of the wave (32-bin block
  // wm MRW + m, frame block wn of NRW tiles)
  if (p.fb) {
    constexpr int RS = BN + 4;
    float *const P = reinterpret_cast<float *>(smem_raw);
#pragma unroll
    for (int m = 0; m < MRW; ++m) bf16x3_fb_write<RS, NRW>(p, acc[m], P, b0, wm * MRW + m, wn);
    filterbank_from_tile<FOLD_BINS, BN, NW * 64>(p, P, b0, n0);
  } else {
#pragma unroll
    for (int m = 0; m < MRW; ++m)
      bf16x3_epilogue_planar<NRW>(p, acc[m], 2 * b0, n0, smem_raw, wm * MRW + m, wn);
  }
}


// The grid: p.fold_main workgroups take 256-frame tiles (XCD-aware order, whole rounds of the
// device's CUs), the rest take 128-frame tiles of the frames behind them -- the last, partial round
// of a 256-frame tiling costs a whole round (Mel cfg3: 864 tiles = 3.375 rounds ran as 4), and
// problems of less than half a round double their parallelism (STFT cfg2's shape at B = 4: 0.10 ->
// 0.07 ms).  A 128-frame tile costs ~0.8 of a 256-frame one, so the host (launch_fold) uses them
// only where that still pays: cfg3 -4 %, cfg2 (6.75 rounds) unchanged.
// p.fold2: the row tiles are those of two problems, the even bins (operands: the even folded rows /
// frames, output rows 0, 2, ..) and the odd bins (rows 1, 3, ..) -- see framed_fold2.inl
template <int NR, int ARITH>
__device__ __forceinline__ void framed_fold_tile(const KParams &p, int tile_m, const long long n0) {
  if (!p.fold2) {
    framed_fold_body<NR, ARITH>(p, tile_m, n0);
    return;
  }
  KParams q = p;
  q.out_row_stride = 2 * p.out_row_stride;
  q.n_bins = p.fold2_bins_e;
  if (tile_m >= p.fold2_tiles_e) {
    tile_m -= p.fold2_tiles_e;
    q.n_bins = p.fold2_bins_o;
    q.as = p.as + p.fold2_as_odd;
    q.xs = p.xs + p.fold2_xs_odd;
    q.out = p.out + p.out_row_stride;
    q.col_add = p.col_add + p.n_cols;
  }
  framed_fold_body<NR, ARITH>(q, tile_m, n0);
}

template <int ARITH>
__device__ __forceinline__ void framed_fold_grid(const KParams &p) {
  const int b = blockIdx.x;
  if (b < p.fold_main) {
    // ---- XCD-aware tile order (as framed_gemm_body)
    int tile;
    {
      const int nwg = p.fold_main;
      const int q = nwg >> 3, r = nwg & 7;
      const int xcd = b & 7, idx = b >> 3;
      tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tile_m, tile_n;
    {
      const int G = p.n_group;
      const int per_group = G * p.n_tiles_m;
      const int full = (p.n_tiles_n / G) * per_group;
      if (tile < full) {
        const int g = tile / per_group;
        const int rest = tile - g * per_group;
        tile_m = rest / G;
        tile_n = g * G + (rest - tile_m * G);
      } else {
        const int Gt = p.n_tiles_n % G;
        const int rest = tile - full;
        tile_m = rest / Gt;
        tile_n = (p.n_tiles_n / G) * G + (rest - tile_m * Gt);
      }
    }
    tile_n += p.fold_tile0;  // (this launch's chunk of frame tiles)
    framed_fold_tile<4, ARITH>(p, tile_m, (long long)tile_n * FOLD_BN);
  } else {
    // bin blocks fastest: the workgroups that run side by side share their frame rows in L2
    const int t = b - p.fold_main;
    const int tile_n = t / p.n_tiles_m, tile_m = t - tile_n * p.n_tiles_m;
    framed_fold_tile<2, ARITH>(p, tile_m, p.fold_tail_frame0 + (long long)tile_n * (FOLD_BN / 2));
  }
}

__global__ void __launch_bounds__(512) framed_fold_kernel(const KParams p) { framed_fold_grid<FOLD_BF16X3>(p); }
__global__ void __launch_bounds__(512) framed_fold32_kernel(const KParams p) { framed_fold_grid<FOLD_F32>(p); }
__global__ void __launch_bounds__(512) framed_fold16_kernel(const KParams p) { framed_fold_grid<FOLD_F16X3>(p); }

