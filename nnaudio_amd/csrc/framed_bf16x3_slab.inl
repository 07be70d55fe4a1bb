// framed_bf16x3_slab.inl -- hop-periodic ("slab") K order of the bf16x3 framed contraction.
// Included by mispec.hip after framed_bf16x3.inl (same operands, same epilogue).
//
// With hop % 32 == 0 write a tap index as k = j*hop + 32*s (s < hop/32 "sub-stage", j "super-
// stage").  The 32 taps of stage (s, j) of frame t are the padded samples
//        slot[(t + j)*hop + 32*s .. +32)
// i.e. stage (s, j) of frame t and stage (s, j') of frame t + j - j' are the SAME 64 bytes.  So the
// K loop runs s outer / j inner: for one s the "slab"
//        X_s[r] = slot[(t0 + r)*hop + 32*s .. +32),     r < BN + C - 1,   C = ceil(Ks / hop)
// is DMA'd into LDS once and serves all C stages of that s (frame t0 + i reads row i + j), each of
// which only streams its A tile.  The split waveform is then read about once per workgroup
// instead of K/hop times from L2 (4x for the n_fft=2048 / hop=512 STFT, 64x for the 84-bin CQT),
// which is what bounds the staged bf16x3 kernel (its matrix pipe is 5x faster than the fp32 one,
// its L2 -> LDS path is not).
//
// A frame tile may straddle one clip boundary (n_frames >= BN): its columns then form two runs
// of consecutive frames, each with its own C - 1 extra rows; column j reads slab row
// j + (C-1)*[j in second run] + super-stage.
//
// LDS: [A stage 0 | A stage 1 | slab buffer 0 | (slab buffer 1) | row / column tables].
// Pipeline: the mid-stage barrier scheme of framed_bf16x3_body for the A tiles; the next slab
// is prefetched into the spare buffer during the first stage of the current one (two buffers),
// or fetched in place after the last reads of the current one (one buffer: one exposed DMA
// latency per slab, i.e. per C stages).

constexpr int SLAB_SJ = 3;                    // slab DMA instructions per wave, per plane
constexpr int SLAB_MAX_ROWS = 16 * 8 * SLAB_SJ;  // 384 rows

template <int WM, int WN, int MR, int NR, bool MASKED>
__device__ __forceinline__ void framed_bf16x3_slab_body(const KParams &p, const int wg_index,
                                                        const int wg_count) {
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
  int n_stages = 0;
  if (ke > kb)
    for (int s = 0; s < SPH; ++s) {
      const int a = j_lo(s), b = j_hi(s);
      if (b >= a) n_stages += b - a + 1;
    }

  // ---- DMA source pointers (hi planes; lo = + plane distance)
  const unsigned short *aptr[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int row = m0 + (j * NW + wave) * 16 + row16;
    int bin = cplx ? (row >> 1) : row;
    bin = bin < p.n_bins ? bin : p.n_bins - 1;  // rows past the end feed unused accumulators
    const long long comp = (cplx && (row & 1)) ? 2 * p.as_plane : 0;
    aptr[j] = p.as + comp + (long long)bin * p.Ks + 8 * cg;
  }
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

  auto dma_a = [&](int kc, int buf, unsigned am) __attribute__((always_inline)) {
    unsigned char *st = sA + buf * A_STAGE;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int pj = j * NW + wave;  // 16-row piece; row tile pj / 2
      if (APIECES % NW == 0 || pj < APIECES) {
        // an inactive row tile is all zeros in this stage and is not multiplied: fetch one hot
        // row instead of streaming zeros through L2
        const bool on = !MASKED || ((am >> (pj >> 1)) & 1u);
        const unsigned short *src = on ? aptr[j] + kc : p.as + 8 * cg;
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

  // fragments: slot = step of the stage (two 16-deep MFMA steps per stage)
  const int fsw = (li >> 2) & 3;
  const int a_off = ((wm * MR) * 32 + li) * ROWB;
  bf16x8 ah[2][MR], al[2][MR], xh[2][NR], xl[2][NR];
  auto load_frags = [&](int abuf, int sbuf, int jsup, int q) __attribute__((always_inline)) {
    const unsigned char *sa = sA + abuf * A_STAGE + a_off;
    const unsigned char *ss = sS + sbuf * SLAB;
    const int off = 16 * ((2 * q + lh) ^ fsw);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      ah[q][m] = *reinterpret_cast<const bf16x8 *>(sa + m * 32 * ROWB + off);
      al[q][m] = *reinterpret_cast<const bf16x8 *>(sa + A_PL + m * 32 * ROWB + off);
    }
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      const int row = xrow[n] + jsup;
      const unsigned char *r = ss + row * ROWB + 16 * ((2 * q + lh) ^ ((row >> 2) & 3));
      xh[q][n] = *reinterpret_cast<const bf16x8 *>(r);
      xl[q][n] = *reinterpret_cast<const bf16x8 *>(r + SL_PL);
    }
  };
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
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, x, acc[m][n], 0, 0, 0);
          }
        }
      }
    }
  };
  auto interleave = [&](auto n_mfma_tag, auto n_ds_tag, auto n_vm_tag) __attribute__((always_inline)) {
    constexpr int NM = decltype(n_mfma_tag)::value;
    constexpr int ND = decltype(n_ds_tag)::value;
    constexpr int NV = decltype(n_vm_tag)::value;
    constexpr int NMD = NM - NM / 4;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if ((i + 1) * NV / NM != i * NV / NM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      if (i < NMD && (i + 1) * ND / NMD != i * ND / NMD)
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
  };
  using std::integral_constant;
  typedef integral_constant<int, 3 * MR * NR> n_mfma;
  typedef integral_constant<int, 2 * (MR + NR)> n_reads;
  typedef integral_constant<int, (APIECES % NW == 0) ? 2 * AJ : 0> n_dma;  // exact only when unguarded
  typedef integral_constant<int, 0> none;

  // ---- stage iterator over (s, j), s outer
  struct It {
    int s, j, hi;
  };
  auto it_first = [&]() __attribute__((always_inline)) -> It {
    It it{0, 0, -1};
    for (; it.s < SPH; ++it.s) {
      it.j = j_lo(it.s);
      it.hi = j_hi(it.s);
      if (it.hi >= it.j) break;
    }
    return it;
  };
  auto it_next = [&](It it) __attribute__((always_inline)) -> It {
    if (it.j < it.hi) {
      ++it.j;
      return it;
    }
    for (++it.s; it.s < SPH; ++it.s) {
      it.j = j_lo(it.s);
      it.hi = j_hi(it.s);
      if (it.hi >= it.j) break;
    }
    return it;  // s == SPH: exhausted (callers bound the walk with n_stages)
  };
  auto tap = [&](const It &it) __attribute__((always_inline)) -> int { return it.j * p.hop + KC * it.s; };

  if (n_stages > 0) {
    const bool two = p.slab_nbuf == 2;
    It cur = it_first();
    It nx1 = it_next(cur);  // stage i+1 (valid iff i + 1 < n_stages)
    It nx2 = it_next(nx1);  // stage i+2
    int sbuf = 0;           // slab buffer of the current stage
    bool pref = false;      // the slab of the next sub-stage is already in flight / landed
    dma_slab(cur.s, 0);
    dma_a(tap(cur), 0, stage_mask(tap(cur)));
    lds_dma_barrier();
    if (n_stages > 1) dma_a(tap(nx1), 1, stage_mask(tap(nx1)));
    load_frags(0, 0, cur.j, 0);

    // One stage.  DMA / NEXT: stages i+2 / i+1 exist (compile time, as framed_bf16x3_body).
    auto stage_iter = [&](int i, auto dma_tag, auto next_tag) __attribute__((always_inline)) {
      constexpr bool DMA = decltype(dma_tag)::value;
      constexpr bool NEXT = decltype(next_tag)::value;
      const int abuf = i & 1;
      const int kc = tap(cur);
      const unsigned mask = stage_mask(kc);
      // first half: step-1 fragments of this stage under the MFMAs of step 0
      load_frags(abuf, sbuf, cur.j, 1);
      mfma_step(0, mask);
      if (!MASKED) interleave(n_mfma{}, n_reads{}, none{});
      lds_dma_barrier();
      // second half
      const bool switching = NEXT && nx1.s != cur.s;  // stage i+1 opens the next slab
      if (two && !pref && NEXT && !switching) {
        // first stage of a slab with more stages to come: prefetch the next slab (if any stage
        // is left for it) into the spare buffer; it lands during the remaining stages
        It probe = cur;
        probe.j = probe.hi;
        const It nslab = it_next(probe);
        if (nslab.s < SPH) dma_slab(nslab.s, sbuf ^ 1);
        pref = true;
      }
      if (switching) {
        if (two && pref) {
          sbuf ^= 1;  // prefetched during the first stage of the current slab, landed since
        } else {
          // no prefetched slab (one buffer, or a one-stage slab): every wave is past its last
          // read of the current slab (barrier above), so fetch the next one now -- in place, or
          // into the spare buffer -- and wait for it: one exposed DMA latency per slab.  (A
          // second copy of the MFMA step here, to overlap that latency, costs the kernel its
          // register budget.)
          const int tb = two ? (sbuf ^ 1) : sbuf;
          dma_slab(nx1.s, tb);
          lds_dma_barrier();
          sbuf = tb;
        }
        pref = false;
      }
      if (DMA) dma_a(tap(nx2), abuf, stage_mask(tap(nx2)));
      if (NEXT) load_frags(abuf ^ 1, sbuf, nx1.j, 0);
      mfma_step(1, mask);
      if (!MASKED)
        interleave(n_mfma{}, integral_constant<int, NEXT ? n_reads::value : 0>{},
                   integral_constant<int, DMA ? n_dma::value : 0>{});
      cur = nx1;
      nx1 = nx2;
      nx2 = it_next(nx2);
    };
    int i = 0;
    for (; i + 2 < n_stages; ++i) stage_iter(i, integral_constant<bool, true>{}, integral_constant<bool, true>{});
    if (i + 1 < n_stages) stage_iter(i++, integral_constant<bool, false>{}, integral_constant<bool, true>{});
    stage_iter(i, integral_constant<bool, false>{}, integral_constant<bool, false>{});
    __syncthreads();  // every wave is done with the LDS buffers (the epilogue may reuse them)
  }

  bf16x3_epilogue<WM, WN, MR, NR>(p, acc, m0, n0, smem_raw);
}

template <int WM, int WN, int MR, int NR, bool MASKED>
__global__ void __launch_bounds__(WM *WN * 64) framed_bf16x3_slab_kernel(const KParams p) {
  framed_bf16x3_slab_body<WM, WN, MR, NR, MASKED>(p, blockIdx.x, gridDim.x);
}

