// framed_bf16x3_packed.inl -- hop-periodic bf16x3 kernel for bases with per-row supports
// (CQT banks), with super-stage packing.  Included by mispec.hip after framed_bf16x3_slab.inl
// (whose header explains the hop-periodic K order: tap k = j*hop + 32*s, slab per sub-stage s).
//
// In a CQT bank the supports are centred and shrink with the bin index, so over most of the K
// range only the first one, two or three 32-row tiles of a 192-row workgroup tile are active.  A
// stage that carries all six row tiles then does 12-36 MFMAs per wave for ~400 other
// instructions and the kernel is instruction-issue bound.  Here the six 32-row
// slots of an A stage buffer hold "units" (row tile m, super-stage j + jj) instead:
//     na active tiles (a prefix), na <= 3  ->  6/na consecutive super-stages per barrier interval
// unit u = jj*na + m, its X fragments are slab rows + (j + jj).  MFMA work and A traffic per
// barrier stay those of a full stage; the number of barrier intervals drops up to 6x.  Stages
// with more (or non-prefix) active tiles run one super-stage with per-tile masks as before.
//
// A barrier interval is a sequence of "X-steps" (jj, 16-tap step q): its X fragments are shared
// by the interval's units.  Fragments are double buffered per X-step; the single barrier sits
// before the last X-step: everything has been read from the A buffer by then (it becomes the
// target of the DMA two intervals ahead) and the next interval's data has landed, so its first
// fragments are read under the last X-step's MFMAs.
//
// Fixed shape: 192x256 tile, 2x4 waves (96x64 per wave); waves w and w+4 share a SIMD, so the
// packed intervals (all work on the row half of waves 0-3) still use all four matrix pipes.

__device__ __forceinline__ void framed_bf16x3_packed_body(const KParams &p, const int wg_index,
                                                          const int wg_count) {
  constexpr int WM = 2, WN = 4, MR = 3, NR = 2;
  constexpr bool MASKED = true;
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int MT = WM * MR;
  constexpr int ROWB = KC * 2;       // bytes of one row of one plane in a stage
  constexpr int A_PL = BM * ROWB;    // bytes of one A plane
  constexpr int A_STAGE = 2 * A_PL;  // [hi | lo]
  constexpr int APIECES = BM / 16;   // 16-row DMA pieces of an A plane
  constexpr int AJ = (APIECES + NW - 1) / NW;
  static_assert(NW == 8, "slab DMA geometry assumes 8 waves");
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  const int slab_rows = p.slab_rows;      // multiple of 16, <= SLAB_MAX_ROWS
  const int SL_PL = slab_rows * ROWB;     // bytes of one slab plane
  const int SLAB = 2 * SL_PL;             // [hi | lo]
  const int spieces = slab_rows / 16;
  const int C = p.n_super;
  const int SPH = p.hop / KC;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *sA = smem_raw;                 // [2][A_STAGE]
  unsigned char *sS = smem_raw + 2 * A_STAGE;   // [slab_nbuf][SLAB]
  long long *sRowOff = reinterpret_cast<long long *>(sS + p.slab_nbuf * SLAB);  // [slab_rows]
  int *sColRow = reinterpret_cast<int *>(sRowOff + slab_rows);                  // [BN]
  int *sTileLo = sColRow + BN;
  int *sTileHi = sTileLo + MT;

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

  // ---- the tile's (at most two) runs of consecutive frames, slab row tables, K ranges
  const int c0 = (int)(n0 / p.n_frames);
  const int t0 = (int)(n0 - (long long)c0 * p.n_frames);
  const int len0 = (p.n_frames - t0) < BN ? (p.n_frames - t0) : BN;  // columns in the first run
  const int rows0 = len0 + C - 1;                                   // slab rows of the first run
  for (int j = tid; j < BN; j += NT) sColRow[j] = j < len0 ? j : j + (C - 1);
  for (int r = tid; r < slab_rows; r += NT) {
    int c = c0, f = t0 + r;
    if (r >= rows0) {
      c = c0 + 1;
      f = r - rows0;
    }
    // rows past the tile's last column (or of a clip past the batch) feed unused columns only
    c = c < p.n_clips ? c : p.n_clips - 1;
    const int fmax = p.n_frames - 1 + C - 1;
    f = f < fmax ? f : fmax;
    sRowOff[r] = (long long)c * p.xs_clip_stride + (long long)f * p.hop;
  }
  if (tid < MT) {
    const int row_lo = m0 + tid * 32;
    int lo = 0, hi = 0;
    const int bin_lo = row_lo / rpb;
    int bin_hi = (row_lo + 32 + rpb - 1) / rpb;
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
  auto stage_mask = [&](int kc) __attribute__((always_inline)) -> unsigned {
    if (!MASKED) return (1u << MT) - 1u;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
      if (thi[i] > kc && tlo[i] < kc + KC) m |= 1u << i;
    return m;
  };
  // stages of sub-stage s: j in [j_lo(s), j_hi(s)], tap k = j*hop + 32*s in [kb, ke)
  auto j_lo = [&](int s) __attribute__((always_inline)) -> int {
    const int num = kb - KC * s;
    return num <= 0 ? 0 : (num + p.hop - 1) / p.hop;
  };
  auto j_hi = [&](int s) __attribute__((always_inline)) -> int {
    const int num = ke - 1 - KC * s;
    return num < 0 ? -1 : num / p.hop;
  };

  // ---- per-lane constants of the A DMA: lane (row16, chunk) of a 16-row piece of row tile m,
  // half h reads bin  lane_bin + TB*m + HB*h  (clamped to the last bin: rows past the end feed
  // unused accumulators) of plane pair `lane_comp`
  const int TB = 32 / rpb, HB = 16 / rpb;
  const int lane_bin = m0 / rpb + row16 / rpb;
  const long long lane_comp = ((cplx && (row16 & 1)) ? 2 * p.as_plane : 0) + 8 * cg;
  const unsigned short *sptr[SLAB_SJ];
#pragma unroll
  for (int j = 0; j < SLAB_SJ; ++j) {
    const int pj = j * NW + wave;
    const int row = (pj < spieces ? pj : 0) * 16 + row16;
    sptr[j] = p.xs + sRowOff[row] + 8 * cg;
  }
  int xrow[NR];  // slab row of this lane's column in each of the wave's column blocks (j = 0)
#pragma unroll
  for (int n = 0; n < NR; ++n) xrow[n] = sColRow[(wn * NR + n) * 32 + li];

  // ---- barrier intervals: position (s, j), packing nae in {1, 2, 3} (jb = 6/nae super-stages,
  // the first nae row tiles) or 6 (one super-stage, tiles per `mask`)
  struct Iv {
    int s, j, hi, k, nae, jb;
    unsigned mask;
    bool valid;
  };
  auto classify = [&](Iv &iv) __attribute__((always_inline)) {
    iv.k = iv.j * p.hop + KC * iv.s;
    iv.mask = stage_mask(iv.k);
    iv.nae = 6;
    iv.jb = 1;
    const unsigned m = iv.mask;
    const int na = m == 1u ? 1 : m == 3u ? 2 : m == 7u ? 3 : 0;
    if (na) {
      const int jb = 6 / na;
      bool same = iv.j + jb - 1 <= iv.hi;
      for (int i = 1; same && i < jb; ++i) same = stage_mask(iv.k + i * p.hop) == m;
      if (same) {
        iv.nae = na;
        iv.jb = jb;
      }
    }
  };
  auto iv_seek = [&](Iv &iv) __attribute__((always_inline)) {  // first valid position at s >= iv.s
    for (; iv.s < SPH; ++iv.s) {
      iv.j = j_lo(iv.s);
      iv.hi = j_hi(iv.s);
      if (iv.hi >= iv.j) break;
    }
    iv.valid = iv.s < SPH;
    if (iv.valid) classify(iv);
  };
  auto iv_next = [&](Iv iv) __attribute__((always_inline)) -> Iv {
    if (!iv.valid) return iv;
    iv.j += iv.jb;
    if (iv.j <= iv.hi) {
      classify(iv);
    } else {
      ++iv.s;
      iv_seek(iv);
    }
    return iv;
  };

  auto dma_a = [&](const Iv &iv, int buf) __attribute__((always_inline)) {
    unsigned char *st = sA + buf * A_STAGE;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int pj = jj * NW + wave;  // 16-row piece: slot u = pj / 2, half h = pj % 2
      if (pj < 12) {
        const int u = pj >> 1, h = pj & 1;
        int mu, ju;  // unit u = (row tile mu, super-stage j + ju)
        if (iv.nae == 6) {
          mu = u;
          ju = 0;
        } else if (iv.nae == 1) {
          mu = 0;
          ju = u;
        } else if (iv.nae == 2) {
          mu = u & 1;
          ju = u >> 1;
        } else {
          mu = u >= 3 ? u - 3 : u;
          ju = u >= 3 ? 1 : 0;
        }
        // an inactive row tile is all zeros in this stage and is not multiplied: fetch one hot
        // row instead of streaming zeros through L2
        const bool on = iv.nae != 6 || ((iv.mask >> mu) & 1u);
        int bin = lane_bin + TB * mu + HB * h;
        bin = bin < p.n_bins ? bin : p.n_bins - 1;
        const unsigned short *src =
            on ? p.as + lane_comp + (long long)bin * p.Ks + (iv.k + ju * p.hop) : p.as + 8 * cg;
        unsigned char *d = st + pj * 16 * ROWB;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + p.as_plane), (lptr_t)(d + A_PL), 16, 0, 0);
      }
    }
  };
  auto dma_slab = [&](int s, int sbuf) __attribute__((always_inline)) {
    unsigned char *st = sS + sbuf * SLAB;
#pragma unroll
    for (int j = 0; j < SLAB_SJ; ++j) {
      const int pj = j * NW + wave;
      if (pj < spieces) {
        const unsigned short *src = sptr[j] + KC * s;
        unsigned char *d = st + pj * 16 * ROWB;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(src + p.xs_plane), (lptr_t)(d + SL_PL), 16, 0, 0);
      }
    }
  };

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  // fragments of one X-step, double buffered (slot = X-step parity)
  const int fsw = (li >> 2) & 3;
  bf16x8 ah[2][MR], al[2][MR], xh[2][NR], xl[2][NR];
  // A fragments of slots u0 .. u0+cnt-1 (step q) -> ah/al[slot][0..cnt-1]
  auto load_a = [&](int abuf, int u0, auto cnt_tag, int q, int slot) __attribute__((always_inline)) {
    constexpr int CNT = decltype(cnt_tag)::value;
    const unsigned char *sa = sA + abuf * A_STAGE + (u0 * 32 + li) * ROWB + 16 * ((2 * q + lh) ^ fsw);
#pragma unroll
    for (int m = 0; m < CNT; ++m) {
      ah[slot][m] = *reinterpret_cast<const bf16x8 *>(sa + m * 32 * ROWB);
      al[slot][m] = *reinterpret_cast<const bf16x8 *>(sa + A_PL + m * 32 * ROWB);
    }
  };
  auto load_x = [&](int sbuf, int jsup, int q, int slot) __attribute__((always_inline)) {
    const unsigned char *ss = sS + sbuf * SLAB;
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      const int row = xrow[n] + jsup;
      const unsigned char *r = ss + row * ROWB + 16 * ((2 * q + lh) ^ ((row >> 2) & 3));
      xh[slot][n] = *reinterpret_cast<const bf16x8 *>(r);
      xl[slot][n] = *reinterpret_cast<const bf16x8 *>(r + SL_PL);
    }
  };
  // MFMAs of one X-step: accumulator tile m gets A fragment m of `slot`, for the m in `mmask`
  auto mfma_x = [&](int slot, auto cnt_tag, unsigned mmask) __attribute__((always_inline)) {
    constexpr int CNT = decltype(cnt_tag)::value;
#pragma unroll
    for (int term = 0; term < 3; ++term) {
#pragma unroll
      for (int m = 0; m < CNT; ++m) {
        if ((mmask >> m) & 1u) {
#pragma unroll
          for (int n = 0; n < NR; ++n) {
            const bf16x8 a = term == 0 ? al[slot][m] : ah[slot][m];
            const bf16x8 x = term == 1 ? xl[slot][n] : xh[slot][n];
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x, acc[m][n], 0, 0, 0);
          }
        }
      }
    }
  };

  using std::integral_constant;
  Iv cur{0, 0, -1, 0, 6, 1, 0u, false};
  if (ke > kb) iv_seek(cur);
  if (cur.valid) {
    const bool two = p.slab_nbuf == 2;
    Iv nx1 = iv_next(cur);
    Iv nx2 = iv_next(nx1);
    int abuf = 0;       // A buffer of the current interval
    int sbuf = 0;       // slab buffer of the current interval
    bool pref = false;  // the slab of the next sub-stage is already in flight / landed
    dma_slab(cur.s, 0);
    dma_a(cur, 0);
    __syncthreads();
    if (nx1.valid) dma_a(nx1, 1);
    // first X-step of an interval: A slots 3*wm .. +3 (one super-stage) or 0 .. 3 (packed: the
    // slots past its nae units belong to later X-steps and are simply not multiplied)
    auto load_first = [&](const Iv &iv, int ab, int sb) __attribute__((always_inline)) {
      load_a(ab, iv.nae == 6 ? 3 * wm : 0, integral_constant<int, MR>{}, 0, 0);
      load_x(sb, iv.j, 0, 0);
    };
    load_first(cur, 0, 0);

    // The loop runs over super-stages; an interval is 6/nae (or one) consecutive iterations.
    // Per iteration: X-step q = 0 sits in fragment slot 0; request q = 1 into slot 1 under the
    // MFMAs of slot 0; then request the NEXT super-stage's q = 0 into slot 0 under the MFMAs of
    // slot 1 -- the same straight-line code whether that super-stage belongs to this interval or
    // opens the next one; only the barrier + DMA bookkeeping in between is conditional (it
    // defines no vector registers, which keeps the accumulators out of control-flow merges).
    typedef integral_constant<int, MR> cnt;
    int jj = 0;  // super-stage inside the current interval
    while (cur.valid) {
      const bool one = cur.nae == 6;
      const int u0 = one ? 3 * wm : jj * cur.nae;
      // this wave's live accumulator tiles: per `mask`, or the first nae of waves 0-3 (waves 4-7
      // own row tiles 3-5)
      const unsigned mm = one ? ((cur.mask >> (3 * wm)) & 7u)
                              : (wm == 0 ? (1u << cur.nae) - 1u : 0u);
      load_a(abuf, u0, cnt{}, 1, 1);
      load_x(sbuf, cur.j + jj, 1, 1);
      mfma_x(0, cnt{}, mm);
      const bool last = jj + 1 == cur.jb;
      // where the next super-stage's first fragments live
      int nab = abuf, nu0 = u0 + cur.nae, nrow = cur.j + jj + 1;
      bool more = true;
      if (last) {
        // every fragment of this interval has been requested: the barrier (which waits for
        // them) frees its A buffer, and publishes the next interval's data
        __syncthreads();
        const bool switching = nx1.valid && nx1.s != cur.s;  // the next interval opens a new slab
        if (two && !pref && nx1.valid && !switching) {
          // first interval of a slab with more to come: prefetch the next slab (if any interval
          // is left for it) into the spare buffer; it lands during the remaining intervals
          Iv probe = cur;
          probe.s = cur.s + 1;
          iv_seek(probe);
          if (probe.valid) dma_slab(probe.s, sbuf ^ 1);
          pref = true;
        }
        if (switching) {
          if (two && pref) {
            sbuf ^= 1;  // prefetched during the first interval of the current slab, landed since
          } else {
            // no prefetched slab (one buffer, or a one-interval slab): every wave is past its
            // last read of the current slab, so fetch the next one now -- in place, or into
            // the spare buffer -- and wait for it: one exposed DMA latency per slab
            const int tb = two ? (sbuf ^ 1) : sbuf;
            dma_slab(nx1.s, tb);
            __syncthreads();
            sbuf = tb;
          }
          pref = false;
        }
        if (nx2.valid) dma_a(nx2, abuf);
        nab = abuf ^ 1;
        nu0 = nx1.nae == 6 ? 3 * wm : 0;
        nrow = nx1.j;
        more = nx1.valid;
      }
      if (more) {
        load_a(nab, nu0, cnt{}, 0, 0);
        load_x(sbuf, nrow, 0, 0);
      }
      mfma_x(1, cnt{}, mm);
      if (last) {
        cur = nx1;
        nx1 = nx2;
        nx2 = iv_next(nx2);
        abuf ^= 1;
        jj = 0;
      } else {
        ++jj;
      }
    }
    __syncthreads();  // every wave is done with the LDS buffers (the epilogue may reuse them)
  }

  bf16x3_epilogue<WM, WN, MR, NR>(p, acc, m0, n0, smem_raw);
}

__global__ void __launch_bounds__(512) framed_bf16x3_packed_kernel(const KParams p) {
  framed_bf16x3_packed_body(p, blockIdx.x, gridDim.x);
}
