// octave_pyramid.inl -- the octave recursion of CQT2010v2 / VQT (cqt.py:1085-1105, vqt.py:160-188,
// utils.py:73-124, 498-521) with the decimated signals resident in LDS.
// Included by mispec.hip inside its anonymous namespace (uses bf16_split2, epilogue_store).
//
// The reference halves the signal once per octave through HBM (conv1d with the 256-tap anti-alias
// filter, stride 2) and runs two small conv1d per octave on each copy.  Here one workgroup owns
// `nf` consecutive output frames of one clip and keeps, for up to three consecutive levels
// l = 0, 1, 2 of the recursion (x_0 = the launch's input, x_{l+1} = downsampling_by_2(x_l)), the span
// of x_l that these frames -- and the levels below -- depend on, as split-bf16 (hi, lo) planes in LDS:
//
//   A  load the span of x_0 (fp32, zero outside the clip), split, store to LDS
//   B  for l = 0, 1: x_{l+1} = FIR(x_l) on the matrix pipe, straight from LDS into LDS
//         y[32 q + r] = sum_m T[r, m] * x_l[64 q + m],  T[r, m] = taps[m - 2 r - 1]   (32 x 320 banded
//         Toeplitz matrix; its MFMA fragments are 16-byte runs of four displaced copies of the taps
//         that every workgroup builds once in LDS);
//         three bf16 MFMAs per product (split operands), fp32 accumulate, 16 significant bits
//         kept for the next level; the deepest level is also written to HBM in fp32 (the part
//         this workgroup owns) for the next launch of the chain;
//      then the reflected samples the frames of level l need beyond the clip ends replace the zeros
//      the FIR needed there (nn.ReflectionPad1d of get_cqt_complex, utils.py:505-517)
//   C  per level, the 2 x 12 kernel rows x nf frames contraction from the LDS span
//      (v_mfma_f32_16x16x32_bf16: 16 bins x 16 frames per tile, re and im tiles in the same lanes),
//      per-bin scale, Magnitude / Complex / Phase epilogue, (batch, bin, frame[,2]) store.
//
// A chain of launches covers the octaves (three levels per launch; the deepest level of a launch is
// the input of the next), so the only intermediate signals in HBM are x_2, x_4, x_6: the 339 MB
// input is read once instead of decimated through HBM seven times and re-read by eight contractions.
//
// Workgroups are persistent (two per CU, each looping over (clip, frame chunk) work items): the tap
// tables are built once per workgroup and the wave -> level assignment of phase C is fixed (the
// Toeplitz fragments as per-wave register sets from L2 would cost 160 KB per item against the 40 KB
// of signal an item reads).
// LDS layout of a level: rows of 64 samples (128 B) with XOR-swizzled 16-byte chunks, hi plane then
// lo plane; the FIR's fragment reads (lane stride = one row) are bank-conflict free.

constexpr int PYR_ROW = 128;        // bytes per LDS row: 64 bf16 samples, 16-byte chunks XOR-swizzled
constexpr int PYR_LEVELS = 3;       // resident levels per launch
constexpr int PYR_KSTEPS = 20;      // K' = 320 columns of the Toeplitz matrix
constexpr int PYR_MAX_BINS = 16;    // kernel rows per component and level (one 16-row MFMA tile)
constexpr int PYR_TAB_LEN = 384;    // elements per shifted copy of the tap table
constexpr int PYR_TAB_STRIDE = 832; // bytes between copies (52 chunks: conflict-free fragment reads)
constexpr int PYR_TAB_BYTES = 2 * 4 * PYR_TAB_STRIDE;  // hi copies, lo copies
constexpr int PYR_NT = 1;           // column tiles of the FIR a wave multiplies at a time
constexpr int PYR_NB = 8;           // 16-byte loads per thread and batch while fetching a span of x_0
constexpr int PYR_NBI = 12;         // ... in one batch, per-item scaling (spans up to 12 K samples)

struct PyrLevel {
  int L;         // length of x_l
  int hop;       // frame hop at this level (multiple of 4; multiples of 8 read whole 16-byte chunks)
  int K;         // kernel width of this level's bank (0: no contraction at this level)
  int Ks;        // taps per split bank row (K rounded up to 32)
  int n_rows;    // bins at this level (<= 16)
  int out_row0;  // first output row of this level's bins
  int reflect;   // frames use mirror padding (else zeros)
  int halo;      // samples resident before position t0 * hop (multiple of 64)
  int rows;      // LDS rows of 64 samples
  int lds_off;   // byte offset of the hi plane (lo plane: + rows * PYR_ROW)
  const unsigned short *bank;  // split planes [re_hi | re_lo | im_hi | im_lo], each (n_rows, Ks)
  long long bank_plane;        // plane distance, elements
  const float *row_scale;      // (n_rows,) or NULL
  const float *row_unscale;    // F16: inverse of the power of two each bank row was multiplied with
};

struct PyrParams {
  const float *x;  // level 0: (n_clips, L0)
  long long x_clip_stride;
  int n_clips;
  int n_levels;    // 1 .. PYR_LEVELS
  int nf;          // frames per work item (multiple of 16)
  int n_chunks;    // work items per clip
  int n_frames;
  const float *taps;  // anti-alias filter
  int n_taps;
  int dec_pad;     // (n_taps - 1) / 2
  int tab_off;     // LDS byte offset of the tap tables
  PyrLevel lv[PYR_LEVELS];
  float *x_last;   // fp32 copy of the deepest level (n_clips, lv[n_levels-1].L) or NULL
  long long x_last_stride;
  float *out;
  long long out_clip_stride, out_row_stride;
  int epilogue;
  float im_sign, eps;
  unsigned long long *stamps;  // benchmarking build: phase clock of workgroup 7 (100 MHz ticks)
  // MISPEC_PREC_F16X3 (F16 instances): scaled fp16 pairs instead of bf16 pairs
  const unsigned *absmax_in;   // per clip (CLIP_ABSMAX_STRIDE apart): bit pattern of max |x[c, :]|
  unsigned *absmax_out;        // the same for x_last, gathered while it is written (atomicMax), or NULL
  int top;                     // level-0 samples are scaled below 2^top (headroom for the FIR's gain)
  int item_scale;              // F16, first launch of a chain: the power of two is chosen per work item, from the
                               // largest |sample| of its own span (no pass over the clips beforehand)
  int scale_off;               // LDS byte offset of the four per-wave maxima
};

#ifdef MISPEC_ABLATE
#define PYR_STAMP()                                                      \
  do {                                                                   \
    if (p.stamps && blockIdx.x == 7 && tid == 0 && n_stamp < 64) p.stamps[n_stamp++] = wall_clock64(); \
  } while (0)
#else
#define PYR_STAMP() \
  do {              \
  } while (0)
#endif

typedef float f32x4acc __attribute__((ext_vector_type(4)));

// byte offset of sample i of a level: rows of 64 samples, the 16-byte chunks of a row XOR-swizzled
// by (row >> 1) & 7 (sixteen consecutive rows read at the same in-row offset hit 16 different
// 16-byte bank groups)
__device__ __forceinline__ int pyr_addr(int i) {
  const int row = i >> 6;
  return row * PYR_ROW + ((((i & 63) >> 3) ^ ((row >> 1) & 7)) << 4) + ((i & 7) << 1);
}

// MAXS = 32-tap steps of the kernel rows a wave keeps in registers (8: banks up to 256 taps; 6: up to
// 192 -- the reference's banks once their zero margins are trimmed -- 32 VGPRs fewer); wider banks
// stream the remaining steps from L2.
// F16 = MISPEC_PREC_F16X3: every bf16 pair becomes an fp16 pair of a power-of-two scaled value --
//   signal   x 2^(top - e_c), e_c from the clip's largest |sample| (absmax_in); the FIR keeps that scale
//            (its taps carry 2^14, taken off the accumulators before the outputs are split again)
//   bank     every row x its own power of two (mispec_split_basis_f16; inverse in row_unscale)
// and the epilogue multiplies by the inverse factors.  x_last is stored unscaled (fp32) and its per-clip
// absmax gathered on the way (absmax_out): the next launch of the chain needs no pass over it.
template <int MAXS, bool F16>
__global__ void __launch_bounds__(256, 2) octave_pyramid_kernel(const PyrParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = p.n_levels;
  auto split2 = [](float a, float b, unsigned &h, unsigned &l) __attribute__((always_inline)) {
    if (F16) f16_split2(a, b, h, l);
    else bf16_split2(a, b, h, l);
  };
  constexpr float TAP_SCALE = F16 ? 16384.f : 1.f;  // taps (|t| < 1) x 2^14
  typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
  auto mfma32 = [](bf16x8 a, bf16x8 b, f32x16 c) __attribute__((always_inline)) -> f32x16 {
    if (F16)
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };
  auto mfma16 = [](bf16x8 a, bf16x8 b, f32x4acc c) __attribute__((always_inline)) -> f32x4acc {
    if (F16)
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  };

  // ---- once per (persistent) workgroup: tap tables for the Toeplitz fragments.  Lane (r, lh) of
  // step s needs taps[16 s + 8 lh - 2 r - shift .. + 8): copy cp = r & 3 holds the taps displaced
  // by 64 + 2 cp + shift, so that every such run starts on a 16-byte boundary.
  const int shift = 128 - p.dec_pad;
  if (D > 1) {
    unsigned short *th = reinterpret_cast<unsigned short *>(smem_raw + p.tab_off);
    unsigned short *tl = reinterpret_cast<unsigned short *>(smem_raw + p.tab_off + 4 * PYR_TAB_STRIDE);
    for (int i = tid; i < 4 * (PYR_TAB_LEN / 2); i += 256) {
      const int cp = i / (PYR_TAB_LEN / 2), e = 2 * (i - cp * (PYR_TAB_LEN / 2));
      float v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = e + u - 64 - 2 * cp - shift;
        v[u] = (idx >= 0 && idx < p.n_taps) ? p.taps[idx] * TAP_SCALE : 0.f;
      }
      unsigned h, l;
      split2(v[0], v[1], h, l);
      *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(th) + cp * PYR_TAB_STRIDE + 2 * e) = h;
      *reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(tl) + cp * PYR_TAB_STRIDE + 2 * e) = l;
    }
  }

  // ---- once per workgroup: this wave's level in phase C (the waves are dealt to the contracting
  // levels)
  int my_level = -1, my_rank = 0, my_peers = 1;
  {
    int nc = 0;
    for (int l = 0; l < D; ++l) nc += p.lv[l].K > 0;
    int seen = 0;
    for (int l = 0; l < D; ++l) {
      if (p.lv[l].K <= 0) continue;
      const int w0 = seen * 4 / nc, w1 = (seen + 1) * 4 / nc;  // waves [w0, w1) serve this level
      if (wave >= w0 && wave < w1) {
        my_level = l;
        my_rank = wave - w0;
        my_peers = w1 - w0;
      }
      ++seen;
    }
    if (nc > 4 || nc == 0) my_level = -1;
  }
  const int fn = lane & 15, kg = lane >> 4;  // 16x16x32 B / D operand: frame of the tile, k block
  // a wave keeps its level's 16 (padded) kernel rows x K taps as MFMA A fragments in registers for
  // all its work items (re-read per item they cost 96 KB of L2 traffic per item: measured slower)
  bf16x8 rh[MAXS], rl[MAXS], ih[MAXS], il[MAXS];
  const unsigned short *are = nullptr, *aim = nullptr;
  if (my_level >= 0) {
    const PyrLevel &v = p.lv[my_level];
    const int steps = v.Ks / 32;
    const int arow = (lane & 15) < v.n_rows ? (lane & 15) : v.n_rows - 1;
    are = v.bank + (long long)arow * v.Ks + 8 * kg;
    aim = are + 2 * v.bank_plane;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      const int ss = s < steps ? s : 0;
      rh[s] = *reinterpret_cast<const bf16x8 *>(are + 32 * ss);
      rl[s] = *reinterpret_cast<const bf16x8 *>(are + v.bank_plane + 32 * ss);
      ih[s] = *reinterpret_cast<const bf16x8 *>(aim + 32 * ss);
      il[s] = *reinterpret_cast<const bf16x8 *>(aim + v.bank_plane + 32 * ss);
    }
  }

  const int n_items = p.n_clips * p.n_chunks;
  int n_stamp = 0;
  (void)n_stamp;
  PYR_STAMP();  // 0: set-up done
  constexpr int NB = PYR_NB;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int c = item / p.n_chunks;
    const int t0 = (item - c * p.n_chunks) * p.nf;
    float xscale = 1.f, xunscale = 1.f;  // F16: the clip's power of two and its inverse
    if (F16 && !p.item_scale) {
      const int e = absmax_exponent(__uint_as_float(p.absmax_in[(long long)c * CLIP_ABSMAX_STRIDE]));
      xscale = pow2f(p.top - e);
      xunscale = pow2f(e - p.top);
    }
    float last_max = 0.f;  // F16: largest |x_last| this thread stores
    __syncthreads();  // the previous item's phase C is done with the spans (and the tables are built)
    PYR_STAMP();  // item start

    // ---- A: span of x_0 -> split planes, in batches of 8 loads per thread in flight (the span is
    // ~40 KB: one load at a time would leave the workgroup waiting on HBM latency ten times over)
    if (F16 && MAXS == 6 && p.item_scale) {  // (the 8-step instance has no registers to spare for the batch)
      // the whole span in one batch of registers; its largest |sample| -- wave maxima through LDS, one more
      // barrier -- picks the item's power of two (the pass over the clips that found one per clip cost 0.067 ms
      // of a 0.61 ms step: every sample had to be seen before the first could be split)
      const PyrLevel &v = p.lv[0];
      const long long a0 = (long long)t0 * v.hop - v.halo;
      const float *x = p.x + (long long)c * p.x_clip_stride;
      unsigned char *hi = smem_raw + v.lds_off, *lo = hi + v.rows * PYR_ROW;
      const int n = v.rows * 64;
      f32x4v f[PYR_NBI];
      float m = 0.f;
#pragma unroll
      for (int b = 0; b < PYR_NBI; ++b) {
        const int i = 4 * tid + 1024 * b;
        const long long g = a0 + i;
        const bool inside = i < n && g >= 0 && g + 3 < v.L;
        f[b] = *reinterpret_cast<const f32x4u *>(x + (inside ? g : 0));
        if (!inside) {
#pragma unroll
          for (int e = 0; e < 4; ++e) f[b][e] = (i < n && g + e >= 0 && g + e < v.L) ? x[g + e] : 0.f;
        }
      }
#pragma unroll
      for (int b = 0; b < PYR_NBI; ++b)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, finite_abs(f[b][e]));
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
      float *const s_max = reinterpret_cast<float *>(smem_raw + p.scale_off);
      if (lane == 0) s_max[wave] = m;
      __syncthreads();
      m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
      const int e2 = absmax_exponent(m);
      xscale = pow2f(p.top - e2);
      xunscale = pow2f(e2 - p.top);
#pragma unroll
      for (int b = 0; b < PYR_NBI; ++b) {
        const int i = 4 * tid + 1024 * b;
        if (i < n) {
          uint2 h, l;
          split2(f[b][0] * xscale, f[b][1] * xscale, h.x, l.x);
          split2(f[b][2] * xscale, f[b][3] * xscale, h.y, l.y);
          const int ad = pyr_addr(i);
          *reinterpret_cast<uint2 *>(hi + ad) = h;
          *reinterpret_cast<uint2 *>(lo + ad) = l;
        }
      }
    } else {
    {
      const PyrLevel &v = p.lv[0];
      const long long a0 = (long long)t0 * v.hop - v.halo;
      const float *x = p.x + (long long)c * p.x_clip_stride;
      unsigned char *hi = smem_raw + v.lds_off, *lo = hi + v.rows * PYR_ROW;
      const int n = v.rows * 64;
      for (int i0 = 4 * tid; i0 < n; i0 += 1024 * NB) {
        f32x4v f[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int i = i0 + 1024 * b;
          const long long g = a0 + i;
          const bool inside = i < n && g >= 0 && g + 3 < v.L;
          f[b] = *reinterpret_cast<const f32x4u *>(x + (inside ? g : 0));
          if (!inside) {
#pragma unroll
            for (int e = 0; e < 4; ++e) f[b][e] = (i < n && g + e >= 0 && g + e < v.L) ? x[g + e] : 0.f;
          }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int i = i0 + 1024 * b;
          if (i < n) {
            uint2 h, l;
            split2(f[b][0] * xscale, f[b][1] * xscale, h.x, l.x);
            split2(f[b][2] * xscale, f[b][3] * xscale, h.y, l.y);
            const int ad = pyr_addr(i);
            *reinterpret_cast<uint2 *>(hi + ad) = h;
            *reinterpret_cast<uint2 *>(lo + ad) = l;
          }
        }
      }
    }
    }
    __syncthreads();
    PYR_STAMP();  // span in LDS

    // mirrored samples beyond the clip ends, for the frames of level l (after the FIR has consumed
    // the zeros there)
    auto reflect_fixup = [&](int l) __attribute__((always_inline)) {
      const PyrLevel &v = p.lv[l];
      if (!v.reflect || v.K <= 0) return;
      const long long a = (long long)t0 * v.hop - v.halo;
      const int n = v.rows * 64, half = v.K / 2;
      if (a >= 0 && a + n <= v.L) return;  // interior work item
      unsigned char *hi = smem_raw + v.lds_off, *lo = hi + v.rows * PYR_ROW;
      for (int i = tid; i < n; i += 256) {
        const long long g = a + i;
        long long src = -1;
        if (g < 0 && g >= -half) src = -g;
        if (g >= v.L && g < (long long)v.L + half) src = 2LL * (v.L - 1) - g;
        if (src >= 0) {
          const long long j = src - a;
          if (j >= 0 && j < n) {
            const int as = pyr_addr((int)j), ad = pyr_addr(i);
            *reinterpret_cast<unsigned short *>(hi + ad) = *reinterpret_cast<const unsigned short *>(hi + as);
            *reinterpret_cast<unsigned short *>(lo + ad) = *reinterpret_cast<const unsigned short *>(lo + as);
          }
        }
      }
    };

    // ---- B: x_{l+1} = FIR(x_l), LDS -> LDS
    if (D > 1) {
      const int li = lane & 31, lh = lane >> 5;
      const int cp = li & 3;
      // Toeplitz fragment of step s: table position 16 s + 8 lh - 2 (li - cp) + 64 in copy cp
      const unsigned char *tbh = smem_raw + p.tab_off + cp * PYR_TAB_STRIDE + 2 * (8 * lh - 2 * (li - cp) + 64);
      const unsigned char *tbl = tbh + 4 * PYR_TAB_STRIDE;
      for (int l = 0; l + 1 < D; ++l) {
        // (every field used inside the loops below is copied out of the kernel-argument block first:
        // read in place -- p.lv[l] with a run-time l -- each use is a scalar load whose
        // s_waitcnt lgkmcnt(0) also drains the LDS queue; positions fit 32 bits)
        const PyrLevel &vi = p.lv[l], &vo = p.lv[l + 1];
        const int vi_rows = vi.rows, vo_rows = vo.rows, vo_L = vo.L, vo_hop = vo.hop;
        const unsigned char *ihi = smem_raw + vi.lds_off, *ilo = ihi + vi_rows * PYR_ROW;
        unsigned char *ohi = smem_raw + vo.lds_off, *olo = ohi + vo_rows * PYR_ROW;
        const int ao = t0 * vo_hop - vo.halo;  // global index of output 0
        const int n_out = vo_rows * 64;
        const int tiles = (n_out + 1023) / 1024;
        float *const xl_out = (l + 2 == D && p.x_last) ? p.x_last + (long long)c * p.x_last_stride : nullptr;
        const bool last = xl_out != nullptr;
        const int own_lo = t0 * vo_hop, own_hi = own_lo + p.nf * vo_hop;
        // A wave multiplies PYR_NT column tiles at a time -- tile, tile + 4, ... -- with the next
        // step's fragments requested a step ahead.  Tiles past the end compute on clamped columns
        // and are not written.  (Measured: one workgroup per CU with each wave alone on its SIMD and
        // three tiles at a time is slower -- a lone wave issues an MFMA every ~50-90 cycles.)
        for (int tile = wave; tile < tiles; tile += 4 * PYR_NT) {
          int qv[PYR_NT], qr[PYR_NT];
#pragma unroll
          for (int u = 0; u < PYR_NT; ++u) {
            qv[u] = (tile + 4 * u) * 32 + li;  // this lane's column: outputs 32 q .. 32 q + 31
            // column q reads rows q .. q + 4 of the input level; columns past the end of the level
            // are clamped to a valid row
            qr[u] = qv[u] + 5 <= vi.rows ? qv[u] : vi.rows - 5;
          }
          f32x16 acc[PYR_NT];
#pragma unroll
          for (int u = 0; u < PYR_NT; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;
          bf16x8 th[2], tl[2], xh[2][PYR_NT], xl[2][PYR_NT];
          auto frags = [&](int s, int slot) __attribute__((always_inline)) {
            th[slot] = *reinterpret_cast<const bf16x8 *>(tbh + 32 * s);
            tl[slot] = *reinterpret_cast<const bf16x8 *>(tbl + 32 * s);
#pragma unroll
            for (int u = 0; u < PYR_NT; ++u) {
              const int row = qr[u] + (s >> 2);
              const int off = row * PYR_ROW + ((((s & 3) * 2 + lh) ^ ((row >> 1) & 7)) << 4);
              xh[slot][u] = *reinterpret_cast<const bf16x8 *>(ihi + off);
              xl[slot][u] = *reinterpret_cast<const bf16x8 *>(ilo + off);
            }
          };
          frags(0, 0);
#pragma unroll
          for (int s = 0; s < PYR_KSTEPS; ++s) {
            const int k = s & 1;
            // (pinned: hipcc otherwise sinks every fragment read to just before its first use, and
            // the wave -- alone on its SIMD -- waits out the LDS latency in front of each MFMA)
            if (s + 1 < PYR_KSTEPS) frags(s + 1, k ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PYR_NT; ++u) acc[u] = mfma32(tl[k], xh[k][u], acc[u]);
#pragma unroll
            for (int u = 0; u < PYR_NT; ++u) acc[u] = mfma32(th[k], xl[k][u], acc[u]);
#pragma unroll
            for (int u = 0; u < PYR_NT; ++u) acc[u] = mfma32(th[k], xh[k][u], acc[u]);
            __builtin_amdgcn_sched_barrier(0);
          }
          // acc[e] = y[32 q + r], r = (e & 3) + 8 (e >> 2) + 4 lh: four consecutive outputs per quad
#pragma unroll
          for (int u = 0; u < PYR_NT; ++u) {
            const int q = qv[u];
            if (tile + 4 * u < tiles && q * 32 < n_out) {
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4) {
                const int o = 32 * q + 8 * g4 + 4 * lh;  // relative output index of the quad
                const long long g = ao + o;
                float f[4];  // (F16: still carrying the clip's scale; the taps' 2^14 comes off here)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  f[e] = (g + e >= 0 && g + e < vo.L) ? acc[u][4 * g4 + e] * (1.f / TAP_SCALE) : 0.f;
                uint2 h, lw;
                split2(f[0], f[1], h.x, lw.x);
                split2(f[2], f[3], h.y, lw.y);
                const int ad = pyr_addr(o);
                *reinterpret_cast<uint2 *>(ohi + ad) = h;
                *reinterpret_cast<uint2 *>(olo + ad) = lw;
                if (last && g >= own_lo && g < own_hi) {
                  float *d = p.x_last + (long long)c * p.x_last_stride + g;
                  if (F16) {  // unscaled in HBM; its absmax for the next launch
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      f[e] *= xunscale;
                      last_max = fmaxf(last_max, finite_abs(f[e]));
                    }
                  }
                  if (g + 3 < vo.L) {
                    *reinterpret_cast<f32x4u *>(d) = f32x4u{f[0], f[1], f[2], f[3]};
                  } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                      if (g + e < vo.L) d[e] = f[e];
                  }
                }
              }
            }
          }
        }
        __syncthreads();  // level l+1 complete, level l no longer needed by the FIR
        PYR_STAMP();  // FIR level done
        reflect_fixup(l);
      }
    }
    reflect_fixup(D - 1);
    if (F16 && p.absmax_out && D > 1) {  // (one atomic per wave and item: ~10^4 per launch, on n_clips lines)
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) last_max = fmaxf(last_max, __shfl_xor(last_max, d));
      if (lane == 0 && last_max > 0.f)
        atomicMax(p.absmax_out + (long long)c * CLIP_ABSMAX_STRIDE, __float_as_uint(last_max));
    }
    __syncthreads();
    PYR_STAMP();  // fix-ups done

    // ---- C: this wave's level: kernel rows x frames from the LDS span (16 bins x 16 frames per tile)
    if (my_level >= 0) {
      const PyrLevel &v = p.lv[my_level];
      const unsigned char *hi = smem_raw + v.lds_off, *lo = hi + v.rows * PYR_ROW;
      const int steps = v.Ks / 32;
      const int ftiles = p.nf / 16;
      const int v_hop = v.hop, v_wofs = v.halo - v.K / 2, v_n_rows = v.n_rows, v_row0 = v.out_row0;
      const float *const v_scale = v.row_scale;
      const long long o_row = p.out_row_stride;
      float *const o_clip = p.out + (long long)c * p.out_clip_stride;
      const int n_frames = p.n_frames;
      const float im_sign = p.im_sign;
      KParams ep{};
      ep.epilogue = p.epilogue;
      ep.eps = p.eps;
      ep.power = 2.f;
      const int E = epilogue_width(p.epilogue);
      // a frame's 8-sample fragments start on 16-byte chunks when the hop is a multiple of 8; with
      // a hop of 4 (the bottom octave of an 8-octave bank at hop 512) every other frame starts in
      // the middle of one, and its fragment is the upper half of one (swizzled) chunk + the lower
      // half of the next: two 8-byte reads per plane
      const bool half_chunk = (v_hop & 7) != 0;
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      auto frag = [&](const unsigned char *plane, int i) __attribute__((always_inline)) -> bf16x8 {
        if (!half_chunk) return *reinterpret_cast<const bf16x8 *>(plane + pyr_addr(i));
        u64x2 v;
        v[0] = *reinterpret_cast<const unsigned long long *>(plane + pyr_addr(i));
        v[1] = *reinterpret_cast<const unsigned long long *>(plane + pyr_addr(i + 4));
        return __builtin_bit_cast(bf16x8, v);
      };
      for (int ft = my_rank; ft < ftiles; ft += my_peers) {
        const int t = t0 + ft * 16 + fn;
        // window start of frame t inside the span: (t - t0) hop + halo - K/2
        const int w = (ft * 16 + fn) * v_hop + v_wofs;
        f32x4acc cre = {0.f, 0.f, 0.f, 0.f}, cim = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
          if (s < steps) {
            const bf16x8 xh = frag(hi, w + 32 * s + 8 * kg);
            const bf16x8 xl = frag(lo, w + 32 * s + 8 * kg);
            cre = mfma16(rl[s], xh, cre);
            cim = mfma16(il[s], xh, cim);
            cre = mfma16(rh[s], xl, cre);
            cim = mfma16(ih[s], xl, cim);
            cre = mfma16(rh[s], xh, cre);
            cim = mfma16(ih[s], xh, cim);
          }
        }
        for (int s = MAXS; s < steps; ++s) {  // banks wider than 256 taps: rows streamed from L2
          const bf16x8 xh = frag(hi, w + 32 * s + 8 * kg);
          const bf16x8 xl = frag(lo, w + 32 * s + 8 * kg);
          const bf16x8 arh = *reinterpret_cast<const bf16x8 *>(are + 32 * s);
          const bf16x8 arl = *reinterpret_cast<const bf16x8 *>(are + v.bank_plane + 32 * s);
          const bf16x8 aih = *reinterpret_cast<const bf16x8 *>(aim + 32 * s);
          const bf16x8 ail = *reinterpret_cast<const bf16x8 *>(aim + v.bank_plane + 32 * s);
          cre = mfma16(arl, xh, cre);
          cim = mfma16(ail, xh, cim);
          cre = mfma16(arh, xl, cre);
          cim = mfma16(aih, xl, cim);
          cre = mfma16(arh, xh, cre);
          cim = mfma16(aih, xh, cim);
        }
        // lane (frame fn, kg) holds bins 4 kg + e
        if (t < n_frames) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int bin = 4 * kg + e;
            if (bin < v_n_rows) {
              float sc = v_scale ? v_scale[bin] : 1.f;
              if (F16) sc *= xunscale * v.row_unscale[bin];
              float *d = o_clip + (long long)(v_row0 + bin) * o_row + (long long)t * E;
              epilogue_store(ep, d, cre[e] * sc, im_sign * cim[e] * sc);
            }
          }
        }
      }
    }
  }
}
