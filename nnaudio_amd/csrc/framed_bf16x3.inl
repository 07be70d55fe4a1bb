// framed_bf16x3.inl -- MISPEC_PREC_BF16X3: the framed contraction on the bf16 matrix pipe.
// Included by mispec.hip inside its anonymous namespace (shares KParams, the edge plan, the
// XCD-aware tile order and the pointwise epilogue with the fp32 kernel).
//
// gfx950 has no fast fp32 MFMA (v_mfma_f32_32x32x2_f32: 157 TFLOP/s) but a 16x faster bf16 one
// (v_mfma_f32_32x32x16_bf16: 2.5 PFLOP/s).  Every fp32 operand v is split once into two bf16
// numbers  v ~= hi + lo  (hi = rne(v), lo = rne(v - hi): 16 significant bits, full fp32 exponent
// range) and every product is taken as
//        a*x ~= a_lo*x_hi + a_hi*x_lo + a_hi*x_hi          (fp32 accumulate on the MFMA)
// i.e. three bf16 MFMAs per fp32 MFMA-equivalent, a 16/3 = 5.3x higher matrix-pipe ceiling.
// The dropped a_lo*x_lo term and the split error are ~2^-17 per product; measured error is
// ~5e-6 of the spectrum peak against 6e-7 for the fp32 path (tests/test_gpu_parity.py), inside
// the 1e-4 bar of the north star.
//
// Operands (all bf16, prepared by the two pre-pass kernels below):
//   basis : planes [re_hi | re_lo | im_hi | im_lo], each (n_bins, Ks), Ks = K rounded up to 32
//           with zero taps -> no K-tail handling; split once per basis (cached by the caller)
//   signal: planes [hi | lo], each (n_clips, S): a clip slot holds the PADDED clip (what the
//           reference materialises with nn.ReflectionPad1d / ConstantPad1d, stft.py:279-289)
//           followed by zeros: frame t is the plain run  slot[t*hop .. t*hop + Ks)  at an EVEN
//           element offset (hop even), so 16-byte LDS-direct pieces start 4-byte aligned and
//           there is no edge handling anywhere in the contraction.
//
// Workgroup = WM x WN waves, wave tile MR x NR MFMA tiles of 32x32, K stage = 32 taps.
// LDS stage = [A_hi | A_lo | X_hi | X_lo], rows of 64 B (32 bf16), double buffered, filled by
// global_load_lds_dwordx4 (one instruction = 16 rows x 64 B).  The four 16-byte chunks of a row
// are XOR-swizzled by (row >> 2) & 3 -- applied to the per-lane SOURCE address of the DMA and
// again on the fragment reads -- which makes every ds_read_b128 lane group conflict free
// (rows r, r+4, r+8, r+12 share banks; MI355X_MICROARCH.md, LDS table).
// The loads complete under vmcnt: every barrier that publishes a stage is lds_dma_barrier()
// (mispec.hip), which states the s_waitcnt vmcnt(0) instead of leaving it to the compiler.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne_bits(float v) {
  const unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;  // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// v ~= hi + lo, both bf16 (returned as bit patterns)
__device__ __forceinline__ void bf16_split(float v, unsigned &hi, unsigned &lo) {
  hi = bf16_rne_bits(v);
  const float hf = __uint_as_float(hi << 16);
  const float r = v - hf;
  // |v| beyond the bf16 range rounds hi to inf: keep lo finite (inf - inf would be NaN)
  lo = ((hi & 0x7f80u) == 0x7f80u) ? 0u : bf16_rne_bits(r);
}

// basis rows -> (hi, lo) planes, zero-padded to Ks taps.  grid (ceil(Ks/256), n_bins, 1 or 2)
__global__ void __launch_bounds__(256) split_basis_kernel(const float *__restrict__ re,
                                                          const float *__restrict__ im,
                                                          long long row_stride, int n_bins, int K,
                                                          int Ks, unsigned short *__restrict__ dst,
                                                          unsigned short *__restrict__ frag,
                                                          float *__restrict__ frag32,
                                                          const float *__restrict__ row_scale = nullptr) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= Ks) return;
  const int bin = blockIdx.y;
  const int z = blockIdx.z;
  const float *src = z ? im : re;
  const float v = k < K ? src[(long long)bin * row_stride + k] : 0.f;
  if (row_scale) {
    // MISPEC_PREC_F16X3: (hi, lo) fp16 pairs of the row x its power of two -- in the strip kernel's
    // fragment order (frag) and / or as row-major planes (dst: the octave kernel's banks)
    unsigned h2, l2;
    f16_split2(v * row_scale[bin], 0.f, h2, l2);
    if (dst) {
      const long long plane = (long long)n_bins * Ks;
      const long long o = (long long)bin * Ks + k;
      dst[(2 * z) * plane + o] = (unsigned short)(h2 & 0xffff);
      dst[(2 * z + 1) * plane + o] = (unsigned short)(l2 & 0xffff);
    }
    if (frag) {
      const long long tile = bin >> 4;
      const int lane = 2 * (bin & 15) + z + 32 * ((k >> 3) & 1);
      const long long f = (((tile * (Ks >> 4) + (k >> 4)) * 2) * 64 + lane) * 8 + (k & 7);
      frag[f] = (unsigned short)(h2 & 0xffff);
      frag[f + 64 * 8] = (unsigned short)(l2 & 0xffff);
    }
    return;
  }
  if (frag32) {
    // MISPEC_PREC_F32: fragment order of the strip kernel with fp32 taps -- tile of 16 bins (row =
    // 2 * bin + component), 16-tap step, part = tap / 4 % 2, lane = row + 32 * (tap / 8 % 2), 4 taps
    const long long tile = bin >> 4;
    const int lane = 2 * (bin & 15) + z + 32 * ((k >> 3) & 1);
    frag32[(((tile * (Ks >> 4) + (k >> 4)) * 2 + ((k >> 2) & 1)) * 64 + lane) * 4 + (k & 3)] = v;
    return;
  }
  unsigned hi, lo;
  bf16_split(v, hi, lo);
  const long long plane = (long long)n_bins * Ks;
  const long long o = (long long)bin * Ks + k;
  dst[(2 * z) * plane + o] = (unsigned short)hi;
  dst[(2 * z + 1) * plane + o] = (unsigned short)lo;
  if (frag) {
    // fragment order of framed_bf16x3_strip.inl: tile of 16 bins (row = 2 * bin + component),
    // 16-tap step, [hi | lo], lane = row + 32 * (tap / 8 % 2), 8 taps per lane
    const long long tile = bin >> 4;
    const int lane = 2 * (bin & 15) + z + 32 * ((k >> 3) & 1);
    const long long f = (((tile * (Ks >> 4) + (k >> 4)) * 2) * 64 + lane) * 8 + (k & 7);
    frag[f] = (unsigned short)hi;
    frag[f + 64 * 8] = (unsigned short)lo;
  }
}

// padded clips -> (hi, lo) planes.  One thread = 4 consecutive elements of a clip slot
// (S % 8 == 0); element i of a slot is the padded signal at position i - pad, zero beyond it.
// grid (ceil(S/1024), n_clips)
__global__ void __launch_bounds__(256) split_signal_kernel(const KParams p,
                                                           unsigned short *__restrict__ dst) {
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && p.job_counter) *p.job_counter = 0u;
  if (i0 >= p.xs_clip_stride) return;
  const int c = blockIdx.y;
  const float *x = p.x + (long long)c * p.x_clip_stride;
  const long long q0 = i0 - p.pad;  // signal position of the first element
  float v[4];
  if (q0 >= 0 && q0 + 4 <= p.n_samples) {
    const f32x4u t = *reinterpret_cast<const f32x4u *>(x + q0);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = t[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long q = q0 + e;
      float t = 0.f;
      if (q >= -(long long)p.pad && q < (long long)p.n_samples + p.pad)
        t = fetch_sample(p.x, (long long)c * p.x_clip_stride, (int)q, p.n_samples, p.pad_mode, true);
      v[e] = t;
    }
  }
  if (p.split_f32) {  // MISPEC_PREC_F32 (strip kernel): the padded clip itself, fp32
    const f32x4v t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4v *>(reinterpret_cast<float *>(dst) + (long long)c * p.xs_clip_stride + i0) = t;
    return;
  }
  if (p.split_f16) {  // MISPEC_PREC_F16X3: (hi, lo) fp16 pairs of the clip x its power of two
    const float sc = clip_scale_of(p.clip_absmax[(long long)c * CLIP_ABSMAX_STRIDE]);
    uint2 h2, l2;
    f16_split2(v[0] * sc, v[1] * sc, h2.x, l2.x);
    f16_split2(v[2] * sc, v[3] * sc, h2.y, l2.y);
    unsigned short *o16 = dst + (long long)c * p.xs_clip_stride + i0;
    *reinterpret_cast<uint2 *>(o16) = h2;
    *reinterpret_cast<uint2 *>(o16 + p.xs_plane) = l2;
    return;
  }
  u16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned hi, lo;
    bf16_split(v[e], hi, lo);
    h[e] = (unsigned short)hi;
    l[e] = (unsigned short)lo;
  }
  unsigned short *o = dst + (long long)c * p.xs_clip_stride + i0;
  *reinterpret_cast<u16x4 *>(o) = h;
  *reinterpret_cast<u16x4 *>(o + p.xs_plane) = l;
}

// |v| of a finite sample, 0 of an infinite one (fmaxf drops NaNs by itself): the operand scale of a clip is
// chosen for its FINITE samples -- a stray Inf then poisons the frames that contain it, as in the reference,
// instead of pushing every other sample of the clip below fp16's range
__device__ __forceinline__ float finite_abs(float v) {
  const float a = fabsf(v);
  return a < __builtin_inff() ? a : 0.f;
}

// MISPEC_PREC_F16X3: bit pattern of max |x[c, :]| per clip over its finite samples (the padding mirrors or
// zero-fills the clip: the same bound holds for the padded clip).  grid (chunks of ABSMAX_CHUNK samples, n_clips), ONE
// atomic per workgroup, every clip's word in a 128-byte line of its own (CLIP_ABSMAX_STRIDE): with an
// atomic per wave on adjacent words the kernel spent 0.2 ms queueing 27 000 device-scope atomics on
// two cache lines.  dst is zeroed by the caller (hipMemsetAsync) -- positive floats order like their
// bit patterns.
__global__ void __launch_bounds__(256) clip_absmax_kernel(const float *__restrict__ x, long long clip_stride,
                                                          int n_samples, unsigned *__restrict__ dst) {
  const int c = blockIdx.y;
  const float *xc = x + (long long)c * clip_stride;
  const long long q0 = (long long)blockIdx.x * ABSMAX_CHUNK + 4 * threadIdx.x;
  float m = 0.f;
  f32x4u v[ABSMAX_CHUNK / 1024];
#pragma unroll
  for (int u = 0; u < ABSMAX_CHUNK / 1024; ++u) {
    const long long q = q0 + 1024 * u;
    v[u] = f32x4u{0.f, 0.f, 0.f, 0.f};
    if (q + 4 <= n_samples) {
      v[u] = *reinterpret_cast<const f32x4u *>(xc + q);
    } else {
      for (long long i = q; i < n_samples; ++i) m = fmaxf(m, finite_abs(xc[i]));
    }
  }
#pragma unroll
  for (int u = 0; u < ABSMAX_CHUNK / 1024; ++u)
    m = fmaxf(fmaxf(m, fmaxf(finite_abs(v[u][0]), finite_abs(v[u][1]))), fmaxf(finite_abs(v[u][2]), finite_abs(v[u][3])));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    if (m > 0.f) atomicMax(dst + (long long)c * CLIP_ABSMAX_STRIDE, __float_as_uint(m));
  }
}

// MISPEC_PREC_F16X3: per basis row (bin), the power of two that puts its largest |re|, |im| below
// 2^14 and its inverse.  grid (n_bins), one workgroup per row.
__global__ void __launch_bounds__(256) row_scale_kernel(const float *__restrict__ re, const float *__restrict__ im,
                                                        long long row_stride, int K, float *__restrict__ scale,
                                                        float *__restrict__ unscale) {
  const int bin = blockIdx.x;
  float m = 0.f;
  for (int k = threadIdx.x; k < K; k += 256)
    m = fmaxf(m, fmaxf(fabsf(re[(long long)bin * row_stride + k]), fabsf(im[(long long)bin * row_stride + k])));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  __shared__ float sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    const int e = absmax_exponent(m);
    scale[bin] = pow2f(F16_TOP - e);
    unscale[bin] = pow2f(e - F16_TOP);
  }
}

// ---------------------------------------------------------------------------------
// Epilogue shared by the bf16x3 kernels: pointwise epilogue on the wave's MR x NR accumulator
// tiles and the (batch, bin, frame[,2]) store.  (m0, n0) = first basis row / first flat frame
// column of the workgroup tile; smem_raw = the workgroup's LDS (free at this point).
// ---------------------------------------------------------------------------------
template <int WM, int WN, int MR, int NR>
__device__ __forceinline__ void bf16x3_epilogue(const KParams &p, f32x16 (&acc)[MR][NR], const int m0,
                                                const long long n0, unsigned char *smem_raw) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const bool cplx = p.a_im != nullptr;
  // ---- epilogue.  Accumulator element e of lane (li, lh) is
  // D[row = (e&3) + 8*(e>>2) + 4*lh][col = li]: with interleaved (re, im) rows a lane holds both
  // parts of 8 bins of a 32x32 tile (elements e, e+1 for even e), so Complex / Magnitude / Power
  // are formed and stored from registers, frames innermost (32 consecutive frames per half wave).
  const int E = epilogue_width(p.epilogue);
  const bool direct = cplx && (p.epilogue == MISPEC_EPI_COMPLEX || p.epilogue == MISPEC_EPI_MAGNITUDE ||
                               p.epilogue == MISPEC_EPI_POWER);
  if (direct) {
    // Row-major store order: the NR column blocks of a row (NR*32 consecutive frames, one
    // contiguous run of the output) are written back to back, so that L2 sees whole lines.
    // Only the first block's (clip, frame) is kept; block n is 32*n frames further along the
    // flat frame axis, i.e. in the same clip or (per lane) a later one.
    const long long col0 = n0 + (wn * NR) * 32 + li;
    int c0 = 0, t0 = 0;
    {
      const long long cc = col0 < p.n_cols ? col0 : 0;
      c0 = (int)(cc / p.n_frames);
      t0 = (int)(cc - (long long)c0 * p.n_frames);
    }
    float *const orow = p.out + (long long)p.out_row_offset * p.out_row_stride;
    auto store_all = [&](auto epi_tag) __attribute__((always_inline)) {
      constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const int bin0 = ((m0 + (wm * MR + m) * 32) >> 1) + 2 * lh;
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const int bin = bin0 + (e2 & 1) + 4 * (e2 >> 1);
          const bool bin_ok = bin < p.n_bins;
          const float sc = (p.row_scale && bin_ok) ? p.row_scale[bin] : 1.f;
          float *const obin = orow + (long long)bin * p.out_row_stride;
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            int c = c0, t = t0 + 32 * n;
            while (t >= p.n_frames) {
              t -= p.n_frames;
              ++c;
            }
            if (bin_ok && col0 + 32 * n < p.n_cols) {
              const float re = acc[m][n][2 * e2] * sc;
              const float im = p.im_sign * acc[m][n][2 * e2 + 1] * sc;
              float *dst = obin + (long long)c * p.out_clip_stride + (long long)t * E;
              if (EPI == MISPEC_EPI_COMPLEX) {
                *reinterpret_cast<float2 *>(dst) = make_float2(re, im);
              } else if (EPI == MISPEC_EPI_MAGNITUDE) {
                dst[0] = sqrtf(re * re + im * im + p.eps);
              } else {
                epilogue_store(p, dst, re, im);  // MISPEC_EPI_POWER
              }
            }
          }
          // keep the 8*MR row groups in program order: hoisting all their address arithmetic
          // above the first store costs more registers than the kernel has (it must not spill)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (p.epilogue == MISPEC_EPI_COMPLEX)
      store_all(std::integral_constant<int, MISPEC_EPI_COMPLEX>{});
    else if (p.epilogue == MISPEC_EPI_MAGNITUDE)
      store_all(std::integral_constant<int, MISPEC_EPI_MAGNITUDE>{});
    else
      store_all(std::integral_constant<int, MISPEC_EPI_POWER>{});
  } else {
    // phase epilogues / real bases: one 32x32 tile at a time through a wave-private LDS patch
    // (a single code instance of the transcendental epilogues), as framed_gemm_body
    constexpr int LDC = 33;
    float *sC = reinterpret_cast<float *>(smem_raw) + wave * (32 * LDC);
#pragma unroll 1
    for (int ti = 0; ti < MR * NR; ++ti) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
          if (ti == m * NR + n) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
              sC[((e & 3) + 8 * (e >> 2) + 4 * lh) * LDC + li] = acc[m][n][e];
          }
      __syncthreads();
      const int tm = ti / NR, tn = ti - tm * NR;
      const int row_base = m0 + (wm * MR + tm) * 32;
      const long long col = n0 + (wn * NR + tn) * 32 + li;
      const bool col_ok = col < p.n_cols;
      int c = 0, t = 0;
      if (col_ok) {
        c = (int)(col / p.n_frames);
        t = (int)(col - (long long)c * p.n_frames);
      }
      float *obase = p.out + (long long)c * p.out_clip_stride + (long long)t * E;
      if (cplx) {
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
          const int rl = 2 * (2 * it + lh);  // even local row: re; rl + 1: im
          const int bin = (row_base + rl) >> 1;
          if (col_ok && bin < p.n_bins) {
            float re = sC[rl * LDC + li];
            float im = p.im_sign * sC[(rl + 1) * LDC + li];
            if (p.row_scale) {
              const float sc = p.row_scale[bin];
              re *= sc;
              im *= sc;
            }
            epilogue_store(p, obase + (long long)(p.out_row_offset + bin) * p.out_row_stride, re, im);
          }
        }
      } else {
#pragma unroll 1
        for (int it = 0; it < 16; ++it) {
          const int rl = 2 * it + lh;
          const int row = row_base + rl;
          if (col_ok && row < p.n_bins) {
            float v = sC[rl * LDC + li];
            if (p.row_scale) v *= p.row_scale[row];
            obase[(long long)(p.out_row_offset + row) * p.out_row_stride] = v;
          }
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------
// Epilogue of the dense staged kernel ("planar" accumulators).  The dense kernel multiplies
// frames x basis (MFMA operands swapped) and lays a complex basis out as 32 re rows followed by
// the 32 im rows of the same bins, so that
//   acc[0][n][e] = re, acc[1][n][e] = im   of bin (m0/2 + 32*wm + li)
//   at flat frame n0 + 32*(wn*NR + n) + 8*(e>>2) + 4*lh + (e&3):
// a lane owns both parts of one bin and four consecutive frames per register quad.  Complex /
// Magnitude / Power (and real bases) are formed in registers, transposed through a wave-private
// LDS patch and stored with 16-byte stores that cover whole 512-byte row segments.
// ---------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// (wm, wn): the 32-bin block and the NR*32-frame block of the workgroup tile that `acc` holds -- the
// wave's own coordinates in the dense kernel; the folded kernel's waves own two bin blocks and call
// this once per block.
template <int NR>
__device__ __forceinline__ void bf16x3_epilogue_planar(const KParams &p, f32x16 (&acc)[2][NR], const int m0,
                                                       const long long n0, unsigned char *smem_raw,
                                                       const int wm, const int wn) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int lh = lane >> 5;
  const bool cplx = p.a_im != nullptr;
  const int E = epilogue_width(p.epilogue);
  float *const orow = p.out + (long long)p.out_row_offset * p.out_row_stride;
  // POWER: |z|^2 and |z| directly; other exponents take the generic path below
  const bool pow_sq = p.epilogue == MISPEC_EPI_POWER && p.power == 2.0f && p.eps == 0.f;
  const bool pow_1 = p.epilogue == MISPEC_EPI_POWER && p.power == 1.0f;
  const bool direct = !cplx || p.epilogue == MISPEC_EPI_COMPLEX || p.epilogue == MISPEC_EPI_MAGNITUDE ||
                      pow_sq || pow_1;
  if (direct) {
    // The final values go through a wave-private LDS patch [32 rows][NR*32 frames' worth of
    // floats] (ds_write_b128 by (bin, frame quad), ds_read_b128 by (row, 4 consecutive floats)),
    // so that one store instruction writes whole rows of the wave's tile: 64/LPR rows x
    // NR*128 bytes.  KIND: 0 real basis (pass = row tile m: rows m0 + 32*(2*wm + m) ..),
    // 1 Complex (pass = half of the wave's frames), 2 sqrt(|z|^2 + eps), 3 |z|^2.
    constexpr int RS = NR * 32 + 4;  // patch row stride in floats (16-byte rows, banks spread)
    constexpr int LPR = NR * 8;      // lanes per patch row when every lane reads 4 floats
    float *const P = reinterpret_cast<float *>(smem_raw) + wave * (32 * RS);
    const int lr = lane / LPR, lc = lane % LPR;
    auto store_all = [&](auto kind_tag) __attribute__((always_inline)) {
      constexpr int KIND = decltype(kind_tag)::value;
      constexpr int W = KIND == 1 ? 2 : 1;     // floats per frame
      constexpr int NPASS = KIND <= 1 ? 2 : 1;
      constexpr int FP = NR * 32 / W;          // frames per pass
      constexpr int FPL = 4 / W;               // frames per lane in the read phase
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        {
          const int bin = KIND == 0 ? m0 + (wm * 2 + ps) * 32 + li : (m0 >> 1) + wm * 32 + li;
          const float sc = (p.row_scale && bin < p.n_bins) ? p.row_scale[bin] : 1.f;
          const float sci = sc * p.im_sign;
#pragma unroll
          for (int n = 0; n < NR; ++n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              if (KIND == 1 && (32 * n + 8 * g) / FP != ps) continue;
              float v[4 * W];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const float re = acc[KIND == 0 ? ps : 0][n][4 * g + i] * sc;
                const float im = KIND == 0 ? 0.f : acc[1][n][4 * g + i] * sci;
                if (KIND == 0) v[i] = re;
                if (KIND == 1) {
                  v[2 * i] = re;
                  v[2 * i + 1] = im;
                }
                if (KIND == 2) v[i] = sqrtf(re * re + im * im + p.eps);
                if (KIND == 3) v[i] = re * re + im * im;
              }
              const int f = 32 * n + 8 * g + 4 * lh - (KIND == 1 ? ps * FP : 0);  // frame in the pass
              float *d = P + li * RS + f * W;
#pragma unroll
              for (int h = 0; h < W; ++h) {
                f32x4 q = {v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
                *reinterpret_cast<f32x4 *>(d + 4 * h) = q;
              }
            }
          }
        }
        // rows back out: lane (lr, lc) handles floats 4*lc .. 4*lc+3 of rows lr, lr + 64/LPR, ..
        const long long col = n0 + (wn * NR) * 32 + (KIND == 1 ? ps * FP : 0) + lc * FPL;
        const bool col_ok = col < p.n_cols;
        int c = 0, t = 0;
        if (col_ok) {
          c = (int)(col / p.n_frames);
          t = (int)(col - (long long)c * p.n_frames);
        }
        const bool fast = col_ok && t + FPL - 1 < p.n_frames;  // the lane's frames lie in one clip
        const int bin0 = KIND == 0 ? m0 + (wm * 2 + ps) * 32 : (m0 >> 1) + wm * 32;
        float *const obase = orow + (long long)c * p.out_clip_stride + (long long)t * W;
#pragma unroll 4
        for (int r = lr; r < 32; r += 64 / LPR) {
          const int bin = bin0 + r;
          if (col_ok && bin < p.n_bins) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(P + r * RS + 4 * lc);
            float *dst = obase + (long long)bin * p.out_row_stride;
            if (fast) {
              *reinterpret_cast<f32x4u *>(dst) = q;
            } else {  // straddles a clip boundary or the end of the batch
#pragma unroll
              for (int f = 0; f < FPL; ++f) {
                int cc = c, tt = t + f;
                while (tt >= p.n_frames) {
                  tt -= p.n_frames;
                  ++cc;
                }
                if (col + f < p.n_cols) {
                  float *d1 = orow + (long long)bin * p.out_row_stride + (long long)cc * p.out_clip_stride +
                              (long long)tt * W;
#pragma unroll
                  for (int h = 0; h < W; ++h) d1[h] = q[W * f + h];
                }
              }
            }
          }
        }
      }
    };
    if (!cplx)
      store_all(std::integral_constant<int, 0>{});
    else if (p.epilogue == MISPEC_EPI_COMPLEX)
      store_all(std::integral_constant<int, 1>{});
    else if (pow_sq)
      store_all(std::integral_constant<int, 3>{});
    else
      store_all(std::integral_constant<int, 2>{});  // Magnitude, Power with exponent 1
    return;
  }
  // phase epilogues, general exponents: one (re, im) tile pair at a time through wave-private LDS patches indexed
  // [bin][frame] (a single code instance of the transcendental epilogues), frames innermost
  constexpr int LDC = 33;
  float *sRe = reinterpret_cast<float *>(smem_raw) + wave * (2 * 32 * LDC);
  float *sIm = sRe + 32 * LDC;
#pragma unroll 1
  for (int tn = 0; tn < NR; ++tn) {
#pragma unroll
    for (int n = 0; n < NR; ++n)
      if (tn == n) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int fr = (e & 3) + 8 * (e >> 2) + 4 * lh;
          sRe[li * LDC + fr] = acc[0][n][e];
          sIm[li * LDC + fr] = acc[1][n][e];
        }
      }
    __syncthreads();
    const long long col = n0 + (wn * NR + tn) * 32 + li;
    const bool col_ok = col < p.n_cols;
    int c = 0, t = 0;
    if (col_ok) {
      c = (int)(col / p.n_frames);
      t = (int)(col - (long long)c * p.n_frames);
    }
    float *obase = p.out + (long long)c * p.out_clip_stride + (long long)t * E;
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
      const int rl = 2 * it + lh;
      const int bin = (m0 >> 1) + wm * 32 + rl;
      if (col_ok && bin < p.n_bins) {
        float re = sRe[rl * LDC + li];
        float im = p.im_sign * sIm[rl * LDC + li];
        if (p.row_scale) {
          const float sc = p.row_scale[bin];
          re *= sc;
          im *= sc;
        }
        epilogue_store(p, obase + (long long)(p.out_row_offset + bin) * p.out_row_stride, re, im);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// Fused filterbank epilogue of the dense planar kernel (mel.py:184-189: matmul(mel_basis,
// spec ** power) without the (batch, bins, frames) intermediate).  The workgroup's
// |X|^power tile -- WM*32 bins x WN*NR*32 frames -- is written to LDS once (a lane holds re and
// im of one bin for four consecutive frames per register quad: one 16-byte LDS write each);
// filterbank_from_tile (mispec.hip) then reduces it over the bins of every filter's band.
// Bins are absolute: p.out_row_offset is the first bin of a leftover-row problem.
// ---------------------------------------------------------------------------------
// One wave's part of the |X|^power tile: its bins b0 + 32 wm + li, its frames 32 (wn NR + n) + ...
template <int RS, int NR>
__device__ __forceinline__ void bf16x3_fb_write(const KParams &p, f32x16 (&acc)[2][NR], float *P, const int b0,
                                                const int wm, const int wn) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int bl = wm * 32 + li;
  const bool bin_ok = b0 + bl < p.n_bins;
  const float sc = (p.row_scale && bin_ok) ? p.row_scale[b0 + bl] : 1.f;
  const bool sq = p.power == 2.0f;
#pragma unroll
  for (int n = 0; n < NR; ++n) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float re = acc[0][n][4 * g + i] * sc;
        const float im = acc[1][n][4 * g + i] * sc;
        const float s2 = re * re + im * im + p.eps;
        v[i] = bin_ok ? (sq ? s2 : sqrtf(s2)) : 0.f;
      }
      *reinterpret_cast<f32x4 *>(P + bl * RS + (wn * NR + n) * 32 + 8 * g + 4 * lh) = v;
    }
  }
}

template <int WM, int WN, int NR>
__device__ __forceinline__ void bf16x3_epilogue_fb(const KParams &p, f32x16 (&acc)[2][NR], const int m0,
                                                   const long long n0, unsigned char *smem_raw) {
  constexpr int BN = WN * NR * 32;
  constexpr int BB = WM * 32;   // bins per workgroup tile
  constexpr int RS = BN + 4;    // patch row stride in floats
  constexpr int NT = WM * WN * 64;
  static_assert(NT % (BN / 4) == 0, "a thread keeps its frame quad over all its items");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float *const P = reinterpret_cast<float *>(smem_raw);
  const int b0 = m0 >> 1;  // first bin of the tile (relative to this problem's first bin)
  bf16x3_fb_write<RS, NR>(p, acc, P, b0, wave / WN, wave % WN);
  filterbank_from_tile<BB, BN, NT>(p, P, b0, n0);
}

// F16 = MISPEC_PREC_F16X3: the same staged kernel on v_mfma_f32_32x32x16_f16 -- the planes hold (hi, lo)
// fp16 pairs of every basis row x its power of two (mispec_split_basis_f16) and of every padded clip x
// the power of two of its largest |sample| (clip_absmax_kernel + split_signal_kernel); the accumulators
// are multiplied by the two inverse factors before the epilogue.  Taps are contracted in their natural
// order (the hop-periodic kernels -- strip, narrow, slab -- sum aliased combs whose partial sums are
// orders of magnitude larger than a silent bin's result: DESIGN.md 3.10).
template <int WM, int WN, int MR, int NR, bool MASKED, bool F16 = false>
__device__ __forceinline__ void framed_bf16x3_body(const KParams &p, const int wg_index,
                                                   const int wg_count) {
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int MT = WM * MR;
  // dense tiles: frames x basis product with (re rows | im rows) row tiles, see
  // bf16x3_epilogue_planar
  constexpr bool PLANAR = !MASKED && MR == 2;
  constexpr int ROWB = KC * 2;  // bytes of one row of one plane in a stage
  // one DMA instruction moves 16 rows; every wave issues the same number of them, so a narrow
  // A tile is padded to 16*NW rows (the extra rows re-read the last basis row, never multiplied)
  constexpr int BMA = BM < 16 * NW ? 16 * NW : BM;
  constexpr int A_PL = BMA * ROWB;  // bytes of one A plane
  constexpr int X_PL = BN * ROWB;
  constexpr int STAGE = 2 * A_PL + 2 * X_PL;
  constexpr int AJ = BMA / (16 * NW);  // DMA instructions per wave, per plane, per stage
  constexpr int XJ = BN / (16 * NW);
  static_assert(BMA % (16 * NW) == 0 && BN % (16 * NW) == 0, "DMA geometry");
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *sStage = smem_raw;  // [2][STAGE]
  long long *sColOff = reinterpret_cast<long long *>(smem_raw + 2 * STAGE);  // [BN]
  int *sTileLo = reinterpret_cast<int *>(sColOff + BN);
  int *sTileHi = sTileLo + MT;
  float *sColUnscale = reinterpret_cast<float *>(sTileHi + MT);  // [BN] (F16)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int row16 = lane >> 2;                    // DMA: row inside a 16-row piece
  const int cg = (lane & 3) ^ ((lane >> 4) & 3);  // DMA: global chunk that lands in slot lane & 3

  // ---- XCD-aware tile order (as framed_gemm_body)
  int tile;
  {
    const int nwg = wg_count, b = wg_index;
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
  const int m0 = tile_m * BM;
  const long long n0 = (long long)tile_n * BN;
  const bool cplx = p.a_im != nullptr;
  const int rpb = cplx ? 2 : 1;

  // ---- per-frame element offsets into the split signal, per-row-tile K ranges
  for (int j = tid; j < BN; j += NT) {
    long long col = n0 + j;
    if (col >= p.n_cols) col = 0;  // unused column: any valid frame, its results are not stored
    const int c = (int)(col / p.n_frames);
    const int t = (int)(col - (long long)c * p.n_frames);
    sColOff[j] = (long long)c * p.xs_clip_stride + (long long)t * p.hop;
    if (F16) sColUnscale[j] = clip_unscale_of(p.clip_absmax[(long long)c * CLIP_ABSMAX_STRIDE]);
  }
  if (tid < MT) {
    const int row_lo = m0 + tid * 32;
    int lo = 0, hi = 0;
    const bool pl = PLANAR && cplx;  // row tile (wm, m): part m of bins m0/2 + 32*wm ..+32
    const int bin_lo = pl ? (m0 >> 1) + (tid / MR) * 32 : row_lo / rpb;
    int bin_hi = pl ? bin_lo + 32 : (row_lo + 32 + rpb - 1) / rpb;
    bin_hi = bin_hi < p.n_bins ? bin_hi : p.n_bins;
    if (bin_lo < bin_hi) {
      if (p.row_support) {
        lo = p.K;
        hi = 0;
        for (int b = bin_lo; b < bin_hi; ++b) {
          const int s = p.row_support[2 * b], e = p.row_support[2 * b + 1];
          if (e > s) {
            lo = s < lo ? s : lo;
            hi = e > hi ? e : hi;
          }
        }
        lo = lo < 0 ? 0 : lo;
        hi = hi > p.K ? p.K : hi;
        if (hi <= lo) lo = hi = 0;
      } else {
        hi = p.K;
      }
    }
    sTileLo[tid] = lo;
    sTileHi[tid] = hi;
  }
  __syncthreads();

  int tlo[MT], thi[MT];
  int kb = p.K, ke = 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    tlo[i] = __builtin_amdgcn_readfirstlane(sTileLo[i]);
    thi[i] = __builtin_amdgcn_readfirstlane(sTileHi[i]);
    if (thi[i] > tlo[i]) {
      kb = tlo[i] < kb ? tlo[i] : kb;
      ke = thi[i] > ke ? thi[i] : ke;
    }
  }
  kb = kb & ~(KC - 1);
  int nstages = ke > kb ? (ke - kb + KC - 1) / KC : 0;  // Ks covers the last partial stage
  if (PLANAR && MISPEC_DBG(p, 0x80000)) nstages = nstages < 2 ? nstages : 2;  // ablation: no K loop
  auto stage_mask = [&](int kc) __attribute__((always_inline)) -> unsigned {
    if (!MASKED) return (1u << MT) - 1u;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      if (thi[i] > kc && tlo[i] < kc + KC) m |= 1u << i;
    return m;
  };

  // ---- DMA source pointers (hi planes; lo = + plane distance), one per 16-row piece
  const unsigned short *aptr[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int rt = (j * NW + wave) * 16 + row16;  // row of the workgroup tile
    const int row = m0 + rt;
    int bin = cplx ? (row >> 1) : row;
    bool im_row = cplx && (row & 1);
    if (PLANAR && cplx) {
      bin = (m0 >> 1) + (rt / (MR * 32)) * 32 + (rt & 31);
      im_row = (rt >> 5) & 1;
    }
    bin = bin < p.n_bins ? bin : p.n_bins - 1;  // rows past the end feed unused accumulators
    const long long comp = im_row ? 2 * p.as_plane : 0;
    aptr[j] = p.as + comp + (long long)bin * p.Ks + 8 * cg;
  }
  const unsigned short *xptr[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) xptr[j] = p.xs + sColOff[(j * NW + wave) * 16 + row16] + 8 * cg;

  auto dma_stage = [&](int kc, int buf, unsigned am) __attribute__((always_inline)) {
    unsigned char *st = sStage + buf * STAGE;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      // 16-row piece j*NW + wave lies in row tile (j*NW + wave) / 2; an inactive row tile is
      // all zeros in this stage and is not multiplied: fetch one hot row instead
      const bool on = !MASKED || (j * NW + wave) >= 2 * MT || ((am >> ((j * NW + wave) >> 1)) & 1u);
      const unsigned short *src = on ? aptr[j] + kc : p.as + 8 * cg;
      unsigned char *d = st + (j * NW + wave) * 16 * ROWB;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + p.as_plane), (lptr_t)(d + A_PL), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const unsigned short *src = xptr[j] + kc;
      unsigned char *d = st + 2 * A_PL + (j * NW + wave) * 16 * ROWB;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + p.xs_plane), (lptr_t)(d + X_PL), 16, 0, 0);
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  // A K stage is two 16-deep MFMA steps.  Lane (li, lh) supplies row li of a 32-row tile and
  // taps 8*lh .. 8*lh+7 of the step: one ds_read_b128 per (tile, plane) and step.
  const int fsw = (li >> 2) & 3;
  const int a_off = ((wm * MR) * 32 + li) * ROWB;
  const int x_off = 2 * A_PL + ((wn * NR) * 32 + li) * ROWB;
  bf16x8 ah[2][MR], al[2][MR], xh[2][NR], xl[2][NR];  // slot = step of the stage
  auto load_frags = [&](int buf, int q) __attribute__((always_inline)) {
    const unsigned char *st = sStage + buf * STAGE;
    const int off = 16 * ((2 * q + lh) ^ fsw);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      ah[q][m] = *reinterpret_cast<const bf16x8 *>(st + a_off + m * 32 * ROWB + off);
      al[q][m] = *reinterpret_cast<const bf16x8 *>(st + a_off + A_PL + m * 32 * ROWB + off);
    }
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      xh[q][n] = *reinterpret_cast<const bf16x8 *>(st + x_off + n * 32 * ROWB + off);
      xl[q][n] = *reinterpret_cast<const bf16x8 *>(st + x_off + X_PL + n * 32 * ROWB + off);
    }
  };
  // the 3 * MR * NR MFMAs of one step; small terms first, and each accumulator is revisited
  // only after MR*NR - 1 other MFMAs
  auto mfma_step = [&](int q, unsigned mask) __attribute__((always_inline)) {
    const unsigned wmask = MASKED ? (mask >> (wm * MR)) : ~0u;
#pragma unroll
    for (int term = 0; term < 3; ++term) {
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        if (!MASKED || ((wmask >> m) & 1u)) {
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            const bf16x8 a = term == 0 ? al[q][m] : ah[q][m];
            const bf16x8 x = term == 1 ? xl[q][n] : xh[q][n];
            if constexpr (F16) {
              typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
              const f16x8_t a16 = __builtin_bit_cast(f16x8_t, a), x16 = __builtin_bit_cast(f16x8_t, x);
              acc[m][n] = PLANAR ? __builtin_amdgcn_mfma_f32_32x32x16_f16(x16, a16, acc[m][n], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, x16, acc[m][n], 0, 0, 0);
            } else {
              acc[m][n] = PLANAR ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, a, acc[m][n], 0, 0, 0)
                                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x, acc[m][n], 0, 0, 0);
            }
          }
        }
      }
    }
  };
  // scheduling hints for one half stage: spread N_DS LDS reads and N_VM LDS-DMA loads evenly
  // through its N_MFMA MFMAs instead of leaving them in front of an idle matrix pipe
  auto interleave = [&](auto n_mfma_tag, auto n_ds_tag, auto n_vm_tag) __attribute__((always_inline)) {
    constexpr int NM = decltype(n_mfma_tag)::value;
    constexpr int ND = decltype(n_ds_tag)::value;
    constexpr int NV = decltype(n_vm_tag)::value;
    // the reads are consumed right after this half: have them issued over its first 3/4 so
    // that their latency is covered by the remaining MFMAs
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
  typedef integral_constant<int, 3 * MR * NR> n_mfma;       // MFMAs per step
  typedef integral_constant<int, 2 * (MR + NR)> n_reads;    // fragment reads per step
  typedef integral_constant<int, 2 * (AJ + XJ)> n_dma;      // DMA instructions per stage
  typedef integral_constant<int, 0> none;

  // ---- K loop.  One barrier per stage, placed between its two steps:
  //   first half  : MFMAs of step 0 (fragments already in registers) while the step-1 fragments
  //                 of this stage are read from buffer b
  //   barrier     : every wave has read all it needs from buffer b, and stage c+1 (DMA'd a whole
  //                 stage earlier) has landed in buffer b^1
  //   second half : MFMAs of step 1 while stage c+2 is DMA'd into buffer b and the step-0
  //                 fragments of stage c+1 are read from buffer b^1
  // so neither the LDS latency nor the DMA issue ever sits in front of an idle matrix pipe.
  // DMA / NEXT: whether stages c+2 / c+1 exist (compile time: no control flow inside a half).
  auto stage_iter = [&](int c, auto dma_tag, auto next_tag) __attribute__((always_inline)) {
    constexpr bool DMA = decltype(dma_tag)::value;
    constexpr bool NEXT = decltype(next_tag)::value;
    const int buf = c & 1;
    const int kc = kb + c * KC;
    const unsigned mask = stage_mask(kc);
    load_frags(buf, 1);
    mfma_step(0, mask);
    if (!MASKED) interleave(n_mfma{}, n_reads{}, none{});
    lds_dma_barrier();
    if (DMA) dma_stage(kc + 2 * KC, buf, stage_mask(kc + 2 * KC));
    if (NEXT) load_frags(buf ^ 1, 0);
    mfma_step(1, mask);
    if (!MASKED)
      interleave(n_mfma{}, integral_constant<int, NEXT ? n_reads::value : 0>{},
                 integral_constant<int, DMA ? n_dma::value : 0>{});
  };
  if (nstages > 0) {
    dma_stage(kb, 0, stage_mask(kb));
    lds_dma_barrier();
    if (nstages > 1) dma_stage(kb + KC, 1, stage_mask(kb + KC));
    load_frags(0, 0);
    int c = 0;
    for (; c + 2 < nstages; ++c) stage_iter(c, integral_constant<bool, true>{}, integral_constant<bool, true>{});
    if (c + 1 < nstages) stage_iter(c++, integral_constant<bool, false>{}, integral_constant<bool, true>{});
    stage_iter(c, integral_constant<bool, false>{}, integral_constant<bool, false>{});
    __syncthreads();  // every wave is done with the stage buffers (the epilogue reuses them)
  }

  if constexpr (F16) {  // take the operands' powers of two off the accumulators
    if constexpr (PLANAR) {
      // acc[part][n][e]: bin (m0 / rpb) + 32 wm + li, frame column 32 (wn NR + n) + 8 (e >> 2) + 4 lh + (e & 3)
      int bin = m0 / rpb + 32 * wm + li;
      bin = bin < p.n_bins ? bin : p.n_bins - 1;
      const float ru = p.row_unscale[bin];
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4v cu = *reinterpret_cast<const f32x4v *>(sColUnscale + 32 * (wn * NR + n) + 8 * g + 4 * lh);
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m][n][4 * g + e] *= ru * cu[e];
        }
    } else {
      // acc[m][n][e]: basis row m0 + 32 (wm MR + m) + 8 (e >> 2) + 4 lh + (e & 3), frame column 32 (wn NR + n) + li
      float cu[NR];
#pragma unroll
      for (int n = 0; n < NR; ++n) cu[n] = sColUnscale[32 * (wn * NR + n) + li];
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          int bin = (m0 + 32 * (wm * MR + m) + 8 * (e >> 2) + 4 * lh + (e & 3)) / rpb;
          bin = bin < p.n_bins ? bin : p.n_bins - 1;
          const float ru = p.row_unscale[bin];
#pragma unroll
          for (int n = 0; n < NR; ++n) acc[m][n][e] *= ru * cu[n];
        }
    }
    __syncthreads();  // (the table of column factors lies where the epilogue's patches go)
  }

  if constexpr (PLANAR) {
    if MISPEC_DBG(p, 0x40000) {  // ablation: no epilogue (keep the accumulators alive)
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
          for (int e = 0; e < 16; ++e) s += acc[m][n][e];
      if (s == 12345.678f) p.out[0] = s;
      return;
    }
    if (p.fb)
      bf16x3_epilogue_fb<WM, WN, NR>(p, acc, m0, n0, smem_raw);
    else
      bf16x3_epilogue_planar<NR>(p, acc, m0, n0, smem_raw, wave / WN, wave % WN);
  }
  else
    bf16x3_epilogue<WM, WN, MR, NR>(p, acc, m0, n0, smem_raw);
}

template <int WM, int WN, int MR, int NR, bool MASKED>
__global__ void __launch_bounds__(WM *WN * 64) framed_bf16x3_kernel(const KParams p) {
  framed_bf16x3_body<WM, WN, MR, NR, MASKED>(p, blockIdx.x, gridDim.x);
}

template <int WM, int WN, int MR, int NR, bool MASKED>
__global__ void __launch_bounds__(WM *WN * 64) framed_f16x3_kernel(const KParams p) {
  framed_bf16x3_body<WM, WN, MR, NR, MASKED, true>(p, blockIdx.x, gridDim.x);
}

// Dense basis whose row count is not a multiple of 256 (the n_fft/2+1 bins of an STFT): the
// leftover rows run as narrow 32- or 64-row workgroups in the tail of the main grid, where
// they fill the CUs the last partial round of 256x256 tiles leaves idle.
template <int RMR>
__global__ void __launch_bounds__(512) framed_bf16x3_pair_kernel(const KParams pm, const KParams pr,
                                                                  const int n_main) {
  if ((int)blockIdx.x < n_main)
    framed_bf16x3_body<4, 2, 2, 4, false>(pm, blockIdx.x, n_main);
  else
    framed_bf16x3_body<1, 8, RMR, 1, false>(pr, blockIdx.x - n_main, gridDim.x - n_main);
}
