// framed_bf16x3_narrow.inl -- hop-periodic bf16x3 kernel for bases with per-row supports (CQT
// banks): 32-row tiles with super-stage packing.  Included by mispec.hip after
// framed_bf16x3_slab.inl (whose header explains the hop-periodic K order: tap k = j*hop + 32*s,
// one LDS slab per sub-stage s).
//
// In a CQT bank the supports are centred and shrink by 2x per octave, so a workgroup that owns
// many row tiles spends most of its K range with one or two of them active; per-stage masks keep
// the MFMA count right but every stage still pays its barrier, DMA bookkeeping and a cascade of
// scalar branches, and the kernel ends up issue bound at ~19 % MFMA busy.
//
// Here every workgroup owns ONE 32-row tile (16 bins) x 256 frames and runs the dense contraction
// over that tile's own K range [kb, ke) -- no masks, no branches around MFMAs.  To keep the work
// per barrier that of a wide tile, the six 32-row slots of an A stage buffer hold the tile's rows
// at six consecutive super-stages j .. j+5 ("units"): one barrier interval = 6 super-stages x 2
// MFMA steps, all read from the same slab (unit jj reads slab rows + j + jj).  Workgroups of the
// long low-frequency tiles run 100+ intervals, those of the short high-frequency tiles a few;
// the grid is ordered longest first and the hardware scheduler balances it.
//
// 8 waves, wave w owns frames 32w .. 32w+31 of the tile (one 32x32 accumulator tile, kept as two
// independent partial sums so that consecutive MFMAs never wait on each other).  A barrier
// interval is a sequence of X-steps (super-stage jj, 16-tap step q); fragments are double
// buffered per X-step; the single barrier sits before the last X-step, when everything has been
// read from the A buffer (it becomes the target of the DMA two intervals ahead) and the next
// interval's data has landed, so that its first fragments are read under the last MFMAs.

__device__ __forceinline__ void framed_bf16x3_narrow_body(const KParams &p, const int wg_index,
                                                          const int wg_count) {
  constexpr int WM = 1, WN = 8, MR = 1, NR = 1;
  constexpr bool MASKED = false;
  constexpr int UNITS = 6;  // super-stages per barrier interval = 32-row slots of an A buffer
  constexpr int NW = WM * WN;
  constexpr int NT = NW * 64;
  constexpr int BM = WM * MR * 32;
  constexpr int BN = WN * NR * 32;
  constexpr int MT = WM * MR;
  constexpr int ROWB = KC * 2;       // bytes of one row of one plane in a stage
  constexpr int A_PL = UNITS * 32 * ROWB;  // bytes of one A plane: UNITS slots of 32 rows
  constexpr int A_STAGE = 2 * A_PL;  // [hi | lo]
  constexpr int APIECES = UNITS * 2;  // 16-row DMA pieces of an A plane
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
  int *sJlo = sTileHi + MT;  // [SPH] super-stage range of every sub-stage of the current row tile
  int *sJhi = sJlo + 64;     // (SPH <= 64: launch_bf16x3_narrow)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN;
  const int wn = wave % WN;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int row16 = lane >> 2;                    // DMA: row inside a 16-row piece
  const int cg = (lane & 3) ^ ((lane >> 4) & 3);  // DMA: global chunk that lands in slot lane & 3

  // ---- a workgroup owns one frame tile and every row_split-th row tile (all of them when the
  // frame tiles alone fill the chip): the per-frame-tile tables below are built once, and with
  // row_split == 1 every workgroup does the same amount of work
  const int tile_n = wg_index / p.row_split;
  const int first_m = wg_index - tile_n * p.row_split;
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
  const unsigned short *sptr[SLAB_SJ];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < SLAB_SJ; ++j) {
    const int pj = j * NW + wave;
    const int row = (pj < spieces ? pj : 0) * 16 + row16;
    sptr[j] = p.xs + sRowOff[row] + 8 * cg;
  }
  const int xrow = sColRow[wn * 32 + li];  // slab row of this lane's frame at super-stage 0

  for (int tile_m = first_m; tile_m < p.n_tiles_m; tile_m += p.row_split) {
  const int m0 = tile_m * BM;
  // ---- K range of the row tile: union of its bins' supports (one lane per bin, wave reduction)
  int kb = 0, ke = 0;
  {
    const int bin_lo = m0 / rpb;
    int bin_hi = (m0 + 32 + rpb - 1) / rpb;
    bin_hi = bin_hi < p.n_bins ? bin_hi : p.n_bins;
    int lo = p.K, hi = 0;
    if (p.row_support) {
      const int b = bin_lo + li;
      if (b < bin_hi) {
        const int s0 = p.row_support[2 * b], e0 = p.row_support[2 * b + 1];
        if (e0 > s0) {
          lo = s0;
          hi = e0;
        }
      }
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) {
        const int ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
      }
      lo = lo < 0 ? 0 : lo;
      hi = hi > p.K ? p.K : hi;
    } else if (bin_lo < bin_hi) {
      lo = 0;
      hi = p.K;
    }
    kb = __builtin_amdgcn_readfirstlane(lo) & ~(KC - 1);
    ke = __builtin_amdgcn_readfirstlane(hi);
  }
  // stages of sub-stage s: j in [j_lo(s), j_hi(s)], tap k = j*hop + 32*s in [kb, ke): tabulated
  // once per row tile (the divisions would otherwise sit in every interval's bookkeeping)
  __syncthreads();  // the previous row tile's readers of the table are done
  if (tid < SPH) {
    const int lo_num = kb - KC * tid, hi_num = ke - 1 - KC * tid;
    sJlo[tid] = lo_num <= 0 ? 0 : (lo_num + p.hop - 1) / p.hop;
    sJhi[tid] = (ke <= kb || hi_num < 0) ? -1 : hi_num / p.hop;
  }
  __syncthreads();
  auto j_lo = [&](int s) __attribute__((always_inline)) -> int {
    return __builtin_amdgcn_readfirstlane(sJlo[s]);
  };
  auto j_hi = [&](int s) __attribute__((always_inline)) -> int {
    return __builtin_amdgcn_readfirstlane(sJhi[s]);
  };

  // ---- per-lane constants of the A DMA: lane (row16, chunk) of half h of the tile reads bin
  // lane_bin + HB*h (clamped to the last bin: rows past the end feed unused accumulator rows)
  const int HB = 16 / rpb;
  const int lane_bin = m0 / rpb + row16 / rpb;
  const long long lane_comp = ((cplx && (row16 & 1)) ? 2 * p.as_plane : 0) + 8 * cg;
  // both of a wave's pieces (pj = wave, wave + 8) are the same half h = wave & 1 of the tile
  const unsigned short *arow;  // the lane's basis row, tap 0
  {
    int bin = lane_bin + HB * (wave & 1);
    bin = bin < p.n_bins ? bin : p.n_bins - 1;
    arow = p.as + lane_comp + (long long)bin * p.Ks;
  }
  // ---- barrier intervals: sub-stage s, super-stages j .. j+jb-1 (jb <= UNITS), first tap k
  struct Iv {
    int s, j, hi, k, jb;
    bool valid;
  };
  auto iv_fill = [&](Iv &iv) __attribute__((always_inline)) {
    const int left = iv.hi - iv.j + 1;
    iv.jb = left < UNITS ? left : UNITS;
    iv.k = iv.j * p.hop + KC * iv.s;
  };
  auto iv_seek = [&](Iv &iv) __attribute__((always_inline)) {  // first valid position at s >= iv.s
    for (; iv.s < SPH; ++iv.s) {
      iv.j = j_lo(iv.s);
      iv.hi = j_hi(iv.s);
      if (iv.hi >= iv.j) break;
    }
    iv.valid = iv.s < SPH;
    if (iv.valid) iv_fill(iv);
  };
  auto iv_next = [&](Iv iv) __attribute__((always_inline)) -> Iv {
    if (!iv.valid) return iv;
    iv.j += iv.jb;
    if (iv.j <= iv.hi) {
      iv_fill(iv);
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
      const int pj = jj * NW + wave;  // 16-row piece: unit u = pj / 2, half pj % 2 = wave % 2
      if (pj < APIECES) {
        const int u = pj >> 1;
        // units past the interval's last super-stage are not multiplied: fetch a hot row
        const int ku = u < iv.jb ? iv.k + u * p.hop : iv.k;
        const unsigned short *src = arow + ku;
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

  // two partial sums: the cross terms and the hi*hi term accumulate independently
  f32x16 acc2[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc2[0][e] = acc2[1][e] = 0.f;

  // fragments of one X-step (A unit, slab row offset, 16-tap step q), requested two X-steps ahead
  // of their MFMAs: three slots
  const int fsw = (li >> 2) & 3;
  bf16x8 ah[3], al[3], xh[3], xl[3];
  auto load_frags = [&](int abuf, int unit, int sbuf, int jsup, int q, int slot) __attribute__((always_inline)) {
    const unsigned char *sa = sA + abuf * A_STAGE + (unit * 32 + li) * ROWB + 16 * ((2 * q + lh) ^ fsw);
    ah[slot] = *reinterpret_cast<const bf16x8 *>(sa);
    al[slot] = *reinterpret_cast<const bf16x8 *>(sa + A_PL);
    const int row = xrow + jsup;
    const unsigned char *r = sS + sbuf * SLAB + row * ROWB + 16 * ((2 * q + lh) ^ ((row >> 2) & 3));
    xh[slot] = *reinterpret_cast<const bf16x8 *>(r);
    xl[slot] = *reinterpret_cast<const bf16x8 *>(r + SL_PL);
  };
  auto mfma_x = [&](int slot) __attribute__((always_inline)) {
    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[slot], xh[slot], acc2[0], 0, 0, 0);
    acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[slot], xh[slot], acc2[1], 0, 0, 0);
    acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[slot], xl[slot], acc2[0], 0, 0, 0);
  };

  Iv cur{0, 0, -1, 0, 1, false};
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
    lds_dma_barrier();
    if (nx1.valid) dma_a(nx1, 1);
    load_frags(0, 0, 0, cur.j, 0, 0);
    load_frags(0, 0, 0, cur.j, 1, 1);

    // One interval per iteration, always as the same straight-line code: X-step x = (super-stage
    // x/2, step x%2), x < 12, lives in fragment slot x%3 and is requested two X-steps before its
    // MFMAs (12 % 3 == 0: the next interval's X-steps 0 and 1 land in slots 0 and 1 again).
    // Super-stages past a short interval's last one (end of a sub-stage's j range, short
    // high-frequency tiles) are requested like the others -- their A slots hold a valid dummy
    // row -- but not multiplied.  The barrier sits before the last two X-steps: everything of
    // this interval has been requested by then, and their MFMAs cover the barrier's bookkeeping
    // and the next interval's first fragment reads.
    constexpr int NX = 2 * UNITS;
    static_assert(NX % 3 == 0, "fragment slot rotation");
    while (cur.valid) {
      const int j0 = cur.j, jb = cur.jb;
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        if (x == NX - 2) {
          // every fragment of this interval has been requested: the barrier (which waits for
          // them) frees its A buffer, and publishes the next interval's data
          lds_dma_barrier();
          const bool switching = nx1.valid && nx1.s != cur.s;  // the next interval opens a new slab
          if (two && !pref && nx1.valid && !switching) {
            // first interval of a slab with more to come: prefetch the next slab (if any
            // interval is left for it) into the spare buffer; it lands during the remaining ones
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
              lds_dma_barrier();
              sbuf = tb;
            }
            pref = false;
          }
          if (nx2.valid) dma_a(nx2, abuf);
        }
        if (x + 2 < NX)
          load_frags(abuf, (x + 2) / 2, sbuf, j0 + (x + 2) / 2, (x + 2) % 2, (x + 2) % 3);
        else if (nx1.valid)  // X-steps 0 and 1 of the next interval (its super-stage 0)
          load_frags(abuf ^ 1, 0, sbuf, nx1.j, x + 2 - NX, (x + 2) % 3);
        if (x / 2 < jb) mfma_x(x % 3);
      }
      cur = nx1;
      nx1 = nx2;
      nx2 = iv_next(nx2);
      abuf ^= 1;
    }
    __syncthreads();  // every wave is done with the LDS buffers (the epilogue may reuse them)
  }

  f32x16 acc[MR][NR];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[0][0][e] = acc2[0][e] + acc2[1][e];
  bf16x3_epilogue<WM, WN, MR, NR>(p, acc, m0, n0, smem_raw);
  }  // row tiles
}

__global__ void __launch_bounds__(512) framed_bf16x3_narrow_kernel(const KParams p) {
  framed_bf16x3_narrow_body(p, blockIdx.x, gridDim.x);
}
