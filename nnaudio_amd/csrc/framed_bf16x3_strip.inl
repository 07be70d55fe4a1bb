// framed_bf16x3_strip.inl -- kernel for bases with per-row supports (CQT banks) whose basis
// fragments never pass through LDS; split-bf16 (bf16x3) and, same code, fp32 arithmetic.  Included by mispec.hip after framed_fold.inl; the
// hop-periodic K order (tap k = j*hop + 32*s, one LDS slab per sub-stage s) is explained in
// framed_bf16x3_slab.inl, the predecessor of this kernel in framed_bf16x3_narrow.inl.
//
// The narrow-tile kernel gives a wave 32 rows x 32 frames: four 16-byte LDS fragment reads per
// three MFMAs (170 B/clk of LDS traffic per CU at full matrix rate -- more than the LDS delivers)
// plus a 24-instruction LDS-DMA of the basis stage per barrier interval.  Here a wave owns a
// STRIP: one 32-row tile (16 bins, re/im interleaved) x 128 frames x a range [ja, jb) of the
// tile's super-stages:
//   * its basis fragments are used by no other wave of the workgroup, so they are loaded straight
//     from global memory (L2: the non-zero part of a CQT bank is a few MB) into registers, 16 taps
//     x 32 rows per instruction, three units ahead -- no LDS traffic, no barrier for the basis;
//   * one fragment pair of the basis serves four frame tiles: 8 LDS reads per 12 MFMAs;
//   * the only shared data is the slab of the signal (128 + span - 1 rows of 32 taps), double
//     buffered, one barrier per sub-stage (per two where two slabs fit one buffer).
// The four waves of a workgroup take strips of the SAME 128 frames: a host-made plan (plan_strip,
// mispec.hip) groups the row tiles into passes and deals the waves of a pass out in proportion to
// the tiles' K ranges (cfg4: tile 0 on four waves, tile 1 on four, tile 2 on four, tiles 3-5 as
// 2 + 1 + 1).
// Waves that share a row tile add their partial sums through LDS at the end of the pass; the
// complex / magnitude / power epilogues are formed in registers, the phase ones go through the
// shared bf16x3_epilogue (one code instance of the transcendentals).  Jobs (pass, frame tile) are handed
// out longest first through an atomic counter to TWO persistent workgroups per CU (76 KB of LDS
// each): every SIMD hosts one wave of each, so the serial parts of a job -- tables, the first slab,
// the barrier of every sub-stage, reduction and epilogue -- run under the other workgroup's MFMAs
// (with one 8-wave workgroup per CU they cost 0.11 ms of a 0.41 ms kernel).
//
// The basis is read in "fragment order" (written next to the split planes by split_basis_kernel):
// [16-bin tile][16-tap step][hi | lo][lane][8 taps], lane = row + 32 * (tap / 8 % 2) -- the MFMA's
// own operand layout, so a load instruction fetches one contiguous kilobyte.  (Loading the
// fragments from the row-major planes -- 64 pieces of 16 bytes in 32 rows per instruction -- made
// the kernel address-bound in the texture path: 0.17 ms of 0.43.)

constexpr int STRIP_BN = 128;        // frame columns of a workgroup
constexpr int STRIP_NW = 4;          // waves of a workgroup
constexpr int STRIP_MAX_ROWS = 288;  // slab rows (STRIP_BN + 2 * (span - 1), rounded up to 16)
constexpr int STRIP_MAX_PASS = 8;
constexpr int STRIP_SJ = (STRIP_MAX_ROWS / 16 + STRIP_NW - 1) / STRIP_NW;  // slab DMA pieces per wave and plane
constexpr int STRIP_RED_TILE = STRIP_NW * 16 * 64 * 4;  // one 32 x 32 partial sum of each wave
// [two slab buffers | later: two reduction buffers + the epilogue's patches] [tables]
constexpr int STRIP_RED_BYTES = 2 * STRIP_MAX_ROWS * 64 * 2;
constexpr int STRIP_PLAN_BYTES = 1280;  // >= sizeof(StripPlan): the plan's copy in LDS
constexpr int STRIP_LDS_BYTES = STRIP_RED_BYTES + STRIP_MAX_ROWS * 8 + STRIP_BN * 4 + 64 + STRIP_PLAN_BYTES;
static_assert(2 * STRIP_LDS_BYTES <= 160 * 1024, "two workgroups per CU");

struct StripWave {
  int tile;    // 32-row tile (-1: the wave idles through this pass)
  int kb, ke;  // tap range of the tile (kb a multiple of 32)
  int ja, jb;  // the wave's super-stages
  int g0, gsize;  // waves g0 .. g0+gsize-1 share the tile
  int fmask;   // bit f: this wave reduces and stores frame tile f of the row tile
};
struct StripPass {
  StripWave w[STRIP_NW];
  int jbase, span;  // super-stages [jbase, jbase + span) are read from the slab
  int slab_rows;    // multiple of 16
  int cost;
  int group;        // sub-stages per slab buffer (2 when two slabs fit: half the barriers), else 1
};
struct StripPlan {
  int n_pass, n_tiles_n, n_jobs;
  int nf;  // frame tiles of a job (4: 128 frames; 2: 64)
  StripPass pass[STRIP_MAX_PASS];
};
static_assert(sizeof(StripPlan) <= 1280, "STRIP_PLAN_BYTES");

// 16-byte-per-lane LDS-direct load, as instructions: hipcc knows that the builtin writes LDS and, in
// a loop that also reads LDS, drains vmcnt before every fragment read -- with it the basis
// fragments requested two units ahead.  Here the slab loads are ordered by hand (strip_barrier);
// the compiler's own vmcnt waits for the basis registers only get stricter by loads it cannot see.
__device__ __forceinline__ void strip_lds_dma16(const void *src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(src), "s"(lds_addr)
               : "memory", "m0");
}

// s_waitcnt vmcnt(N) lgkmcnt(0); s_barrier -- as instructions (see lds_dma_barrier_keep)
__device__ __forceinline__ void strip_barrier(int younger_units) {
  if (younger_units >= 3)
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if (younger_units == 2)
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else if (younger_units == 1)
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// F32 = the same kernel in the exact arithmetic (MISPEC_PREC_F32; same plan, same LDS layout,
// same job structure): the slab holds fp32 samples -- a row's 32 taps as two 64-byte halves in the
// two planes (taps 0-15 | 16-31: the DMA reads both from the padded fp32 copy of the clip, 64 bytes
// apart), a lane's fragments of a 16-tap step are taps 8 lh .. 8 lh + 3 and .. + 7 of the step's
// plane (chunks 2 lh and 2 lh + 1 where the split arithmetic reads the hi and the lo plane) -- the
// basis fragments are fp32 in the same [tile][16-tap step][part][lane][16 bytes] order, and a
// step is 8 x 4 v_mfma_f32_32x32x2_f32 (MFMA t of a frame tile contracts taps t and 8 + t).
// ARITH = FOLD_F16X3 (MISPEC_PREC_F16X3): the split kernel on v_mfma_f32_32x32x16_f16 -- the planes of
// the signal are (hi, lo) fp16 pairs of the padded clip x a power of two chosen per clip from its largest
// |sample| (clip_absmax_kernel), the fragment-order basis holds fp16 pairs of each row x a power of two
// (row_scale_kernel); the direct epilogues multiply by the two inverse factors (the phase epilogues
// do not care about a positive common factor of re and im).
template <int ARITH, int NF>
__device__ __forceinline__ void framed_strip_body(const KParams &p, const StripPlan &plan) {
  constexpr bool F32 = ARITH == FOLD_F32;
  constexpr int NW = STRIP_NW;
  constexpr int BN = 32 * NF;  // frame columns of a job: NF frame tiles per wave (4; 2 for small problems)
  constexpr int ROWB = KC * 2;                    // bytes of one slab row of one plane
  constexpr int SL_PL = STRIP_MAX_ROWS * ROWB;    // hi -> lo plane of a slab buffer
  constexpr int SLAB = 2 * SL_PL;
  static_assert(2 * SLAB <= STRIP_RED_BYTES && 4 * STRIP_RED_TILE <= STRIP_RED_BYTES && NW * 32 * 33 * 4 <= STRIP_RED_BYTES,
                "slabs, then partial sums + epilogue patches, share the first region");
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char *sS = smem_raw;  // [2][SLAB]; later the partial sums
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)smem_raw);
  long long *sRowOff = reinterpret_cast<long long *>(smem_raw + STRIP_RED_BYTES);  // [STRIP_MAX_ROWS]
  int *sColRow = reinterpret_cast<int *>(sRowOff + STRIP_MAX_ROWS);                // [STRIP_BN]
  int *sJob = sColRow + STRIP_BN;
  // the plan, copied out of the kernel-argument block once: a job reads a dozen of its fields, and
  // dependent scalar loads from the argument block cost microseconds per job
  int *sPlanRaw = sJob + 16;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int row16 = lane >> 2;                    // DMA: row inside a 16-row piece
  const int cg = (lane & 3) ^ ((lane >> 4) & 3);  // DMA: global chunk that lands in slot lane & 3
  const int hop = p.hop;
  const int SPH = hop / KC;
  const int C = p.n_super;
  const int n_tiles_n = plan.n_tiles_n;
  // bytes from a slab row's first plane to its second one in the source (split arithmetic: the lo
  // plane of the split signal; fp32: the second 16 taps of the row), and per 32-tap sub-stage
  const long long src_plane_bytes = F32 ? 64 : p.xs_plane * 2;
  constexpr int SRC_STAGE_BYTES = F32 ? 128 : 64;
  // benchmarking build only (constants otherwise): 8 no slab DMA, 16 no reduction / epilogue, 32 no
  // units (the unit loop itself is the product's: ablation branches inside it change its schedule)
  const bool ab_dma = MISPEC_DBG(p, 8), ab_epi = MISPEC_DBG(p, 16);

  for (int i = tid; i < (int)(sizeof(StripPlan) / sizeof(int)); i += NW * 64)
    sPlanRaw[i] = reinterpret_cast<const int *>(&plan)[i];
  if (tid == 0) sJob[0] = (int)atomicAdd(p.job_counter, 1u);
  __syncthreads();
  int job = __builtin_amdgcn_readfirstlane(sJob[0]);
  const StripPlan &lplan = *reinterpret_cast<const StripPlan *>(sPlanRaw);
  const int n_frames = p.n_frames, n_clips = p.n_clips;
  const long long xs_clip_stride = p.xs_clip_stride;
#ifdef MISPEC_ABLATE
  // phase clock of one job (benchmarking build, bit 0x2000000): 100 MHz stamps of workgroup 7's
  // first job, written behind the job counter
  // (bits 28-29: which pass the stamped job belongs to)
  const int stamp_pass = (p.debug >> 28) & 3;
  bool stamp_armed = MISPEC_DBG(p, 0x2000000) && blockIdx.x == 7, stamp_on = false;
  unsigned long long *stamps = reinterpret_cast<unsigned long long *>(p.job_counter + 2);
#define STRIP_STAMP(i) \
  if (stamp_on && tid == 0) stamps[i] = wall_clock64();
#else
#define STRIP_STAMP(i)
#endif
  const int n_jobs = plan.n_jobs;
  while (job < n_jobs) {
#ifdef MISPEC_ABLATE
    stamp_on = stamp_armed && job / n_tiles_n == stamp_pass;
    if (stamp_on) stamp_armed = false;
#endif
    STRIP_STAMP(0)
    const int pass_i = job / n_tiles_n;
    const int tile_n = job - pass_i * n_tiles_n;
    const StripPass &ps = lplan.pass[pass_i];
    const int jbase = __builtin_amdgcn_readfirstlane(ps.jbase), span = __builtin_amdgcn_readfirstlane(ps.span);
    const int slab_rows = __builtin_amdgcn_readfirstlane(ps.slab_rows);
    const int spieces = slab_rows / 16;
    // sub-stages per slab buffer: sub-stage s lives in buffer (s >> gsh) & 1, rows (s & gsh) * slab_rows ..
    const int gsh = __builtin_amdgcn_readfirstlane(ps.group) - 1;
    const int NG = (SPH + gsh) >> gsh;  // slab groups = barriers of the job
    const StripWave &wv = ps.w[wave];
    const int tile_m = __builtin_amdgcn_readfirstlane(wv.tile);
    const int kb = __builtin_amdgcn_readfirstlane(wv.kb), ke = __builtin_amdgcn_readfirstlane(wv.ke);
    const int ja = __builtin_amdgcn_readfirstlane(wv.ja), jb = __builtin_amdgcn_readfirstlane(wv.jb);
    const int g0 = __builtin_amdgcn_readfirstlane(wv.g0), gsize = __builtin_amdgcn_readfirstlane(wv.gsize);
    const int fmask = tile_m < 0 ? 0 : __builtin_amdgcn_readfirstlane(wv.fmask);

    // ---- the frame tile's (at most two) runs of consecutive frames, slab row table
    const long long n0 = (long long)tile_n * BN;
    // (n_cols < 2^31: launch_bf16x3_strip)
    const int c0 = (int)((unsigned)n0 / (unsigned)n_frames);
    const int t0 = (int)((unsigned)n0 - (unsigned)c0 * (unsigned)n_frames);
    const int len0 = (n_frames - t0) < BN ? (n_frames - t0) : BN;
    {
      const int rows0 = len0 + span - 1;
      if (tid < BN) sColRow[tid] = tid < len0 ? tid : tid + (span - 1);
      for (int r = tid; r < slab_rows; r += NW * 64) {
        int c = c0, f = t0 + r + jbase;
        if (r >= rows0) {
          c = c0 + 1;
          f = r - rows0 + jbase;
        }
        // rows past the tile's last column (or of a clip past the batch) feed unused columns only
        c = c < n_clips ? c : n_clips - 1;
        const int fmax = n_frames - 1 + C - 1;
        f = f < fmax ? f : fmax;
        sRowOff[r] = (long long)c * xs_clip_stride + (long long)f * hop;
      }
    }
    // ---- super-stage range of every sub-stage of this wave's strip: lane s holds sub-stage s
    int vJlo, vJhi;
    {
      const int lo_num = kb - KC * lane, hi_num = ke - 1 - KC * lane;
      int jl = lo_num <= 0 ? 0 : (lo_num + hop - 1) / hop;
      int jh = (ke <= kb || hi_num < 0) ? -1 : hi_num / hop;
      jl = jl > ja ? jl : ja;
      jh = jh < jb - 1 ? jh : jb - 1;
      if (tile_m < 0 || lane >= SPH || MISPEC_DBG(p, 32)) jh = -1;  // (32: benchmarking, no units)
      vJlo = jl;
      vJhi = jh;
    }
    __syncthreads();
    STRIP_STAMP(1)
    const char *sptr[STRIP_SJ];
#pragma unroll
    for (int j = 0; j < STRIP_SJ; ++j) {
      const int pj = j * NW + wave;
      const int row = (pj < spieces ? pj : 0) * 16 + row16;
      sptr[j] = reinterpret_cast<const char *>(p.xs) + sRowOff[row] * (F32 ? 4 : 2) + 16 * cg;
    }
    int xrow[NF];  // slab row of this lane's frame of frame tile f at super-stage jbase
#pragma unroll
    for (int f = 0; f < NF; ++f) xrow[f] = sColRow[32 * f + li];

    // ---- the strip's basis fragments: (tile, 16-tap step) -> [hi | lo][lane][8 taps], so a unit
    // (two steps) is 4 KB of consecutive memory and every load a contiguous kilobyte
    const unsigned short *arow =
        p.afrag + (long long)(tile_m < 0 ? 0 : tile_m) * (p.Ks >> 4) * 1024 + lane * 8;
    const unsigned short *azero = p.afrag + (long long)((p.n_bins + 15) >> 4) * (p.Ks >> 4) * 1024 + lane * 8;

    auto dma_slab = [&](int g, int buf) __attribute__((always_inline)) {
      if (ab_dma) return;
      for (int sub = 0; sub <= gsh; ++sub) {
        const int s = (g << gsh) + sub;
        if (s >= SPH) break;
#pragma unroll
        for (int j = 0; j < STRIP_SJ; ++j) {
          const int pj = j * NW + wave;
          if (pj < spieces) {
            const char *src = sptr[j] + SRC_STAGE_BYTES * s;
            const unsigned d = lds0 + buf * SLAB + (sub * slab_rows + pj * 16) * ROWB;
            strip_lds_dma16(src, d);
            strip_lds_dma16(src + src_plane_bytes, d + SL_PL);
          }
        }
      }
    };

    // ---- units (sub-stage s, super-stage j) of the strip, in execution order
    struct It {
      int s, j, hi;
      bool valid;
    };
    auto it_seek = [&](It &it) __attribute__((always_inline)) {
      for (; it.s < SPH; ++it.s) {
        it.j = __builtin_amdgcn_readlane(vJlo, it.s);
        it.hi = __builtin_amdgcn_readlane(vJhi, it.s);
        if (it.hi >= it.j) break;
      }
      it.valid = it.s < SPH;
    };
    auto it_next = [&](It it) __attribute__((always_inline)) -> It {
      if (!it.valid) return it;
      if (++it.j > it.hi) {
        ++it.s;
        it_seek(it);
      }
      return it;
    };

    // basis fragments of a unit: [q][hi, lo], three slots (units u, u + 1, u + 2 in flight: the
    // bank competes with the streaming slabs for the L2, a fragment load is often a miss)
    bf16x8 ah[3][2], al[3][2];
    auto a_src = [&](const It &it) __attribute__((always_inline)) -> const unsigned short * {
      // past the strip's last unit: the block of zeros behind the last tile -- the number of loads in
      // flight stays fixed, and a unit executed on them adds nothing
      return it.valid ? arow + (long long)(it.j * SPH + it.s) * 2048 : azero;
    };
    auto load_a1 = [&](auto slot_tag, const unsigned short *src, int part) __attribute__((always_inline)) {
      constexpr int S = decltype(slot_tag)::value;
      if (part == 0) ah[S][0] = *reinterpret_cast<const bf16x8 *>(src);
      if (part == 1) al[S][0] = *reinterpret_cast<const bf16x8 *>(src + 512);
      if (part == 2) ah[S][1] = *reinterpret_cast<const bf16x8 *>(src + 1024);
      if (part == 3) al[S][1] = *reinterpret_cast<const bf16x8 *>(src + 1536);
    };
    auto load_a = [&](auto slot_tag, const It &it) __attribute__((always_inline)) {
      const unsigned short *src = a_src(it);
#pragma unroll
      for (int part = 0; part < 4; ++part) load_a1(slot_tag, src, part);
    };
    // slab fragments of one 16-tap step, four frame tiles x [hi, lo]: step q lives in set q and is
    // requested during the MFMAs of the step before.  xa[f] = LDS byte address of the unit's step-0
    // hi fragment of frame tile f; step 1 is the same address with bit 5 flipped (chunk 2q + lh,
    // XOR-swizzled), the lo plane SL_PL further.
    typedef __attribute__((address_space(3))) const bf16x8 *lfrag_t;
    bf16x8 xh[2][NF], xl[2][NF];
    unsigned xa[NF];
    auto x_base = [&](int s) __attribute__((always_inline)) -> unsigned {  // slab of sub-stage s
      return lds0 + ((s >> gsh) & 1) * SLAB + (s & gsh) * slab_rows * ROWB;
    };
    auto x_addrs = [&](unsigned (&dst)[NF], int s, int dj) __attribute__((always_inline)) {
      const unsigned base = x_base(s);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int row = xrow[f] + dj;  // (the slabs of a buffer start at multiples of 16 rows: same swizzle)
        dst[f] = base + row * ROWB + 16 * ((F32 ? 2 * lh : lh) ^ ((row >> 2) & 3));
      }
    };
    // address of fragment `part` of step q: split arithmetic -- chunk 2 q + lh of the hi (part 0) /
    // lo (part 1) plane; fp32 -- chunk 2 lh + part of plane q
    auto x_frag = [&](unsigned a0, int q, int part) __attribute__((always_inline)) -> unsigned {
      return F32 ? (a0 ^ (16 * part)) + q * SL_PL : (a0 ^ (32 * q)) + part * SL_PL;
    };
    auto load_set = [&](auto q_tag, const unsigned (&a0)[NF]) __attribute__((always_inline)) {
      constexpr int Q = decltype(q_tag)::value;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        xh[Q][f] = *(lfrag_t)x_frag(a0[f], Q, 0);
        xl[Q][f] = *(lfrag_t)x_frag(a0[f], Q, 1);
      }
    };
    f32x16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
    // One step = 12 MFMAs of (slot S, step Q) in a fixed order -- frame tile innermost, so that
    // MFMAs on the same accumulator are four apart -- with, pinned behind the first eight, the
    // eight fragment reads of the NEXT step (set NQ, addresses an[f] ^ NX) and, behind the last four,
    // `tail(f)`: an address computation or a basis load.  Nothing moves (sched_barrier): a wave
    // alone on its SIMD keeps the matrix pipe busy while it issues the reads (issued as a block in
    // front of the MFMAs they cost 2 x 125 cycles per unit, and the lo fragments, requested inside
    // their own step, another 2 x 150 of waiting); left to hipcc's scheduler the MFMAs get
    // reordered into dependent pairs.
    auto step = [&](auto slot_tag, auto q_tag, const unsigned (&an)[NF], auto &&tail)
                    __attribute__((always_inline)) {
      constexpr int S = decltype(slot_tag)::value;
      constexpr int Q = decltype(q_tag)::value;
      constexpr int NQ = 1 - Q;
      constexpr int NM = (F32 ? 8 : 3) * NF;  // MFMAs of the step
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        const int f = i % NF, t = i / NF;
        if (F32) {
          // MFMA t contracts taps t and 8 + t of the step: element t & 3 of part t >> 2
          const f32x4v a = __builtin_bit_cast(f32x4v, t < 4 ? ah[S][Q] : al[S][Q]);
          const f32x4v x = __builtin_bit_cast(f32x4v, t < 4 ? xh[Q][f] : xl[Q][f]);
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t & 3], x[t & 3], acc[f], 0, 0, 0);
        } else if (ARITH == FOLD_F16X3) {
          const bf16x8 a = t == 0 ? al[S][Q] : ah[S][Q];
          const bf16x8 x = t == 2 ? xl[Q][f] : xh[Q][f];
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                         __builtin_bit_cast(f16x8, x), acc[f], 0, 0, 0);
        } else if (t == 0) {
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[S][Q], xh[Q][f], acc[f], 0, 0, 0);
        } else if (t == 1) {
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[S][Q], xh[Q][f], acc[f], 0, 0, 0);
        } else {
          acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[S][Q], xl[Q][f], acc[f], 0, 0, 0);
        }
        if (i < NF) xh[NQ][i] = *(lfrag_t)x_frag(an[i], NQ, 0);
        else if (i < 2 * NF) xl[NQ][i - NF] = *(lfrag_t)x_frag(an[i - NF], NQ, 1);
        if (i >= NM - 4) tail(i - (NM - 4));  // four slots behind the last four MFMAs
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    typedef std::integral_constant<int, 0> i0;
    typedef std::integral_constant<int, 1> i1;

    It cur{0, 0, -1, false};
    it_seek(cur);
    It n1 = it_next(cur);
    It n2 = it_next(n1);
    It n3 = it_next(n2);
    int cur_g = 0;      // slab group (1 or 2 sub-stages: one buffer) being read
    int done_in_s = 0;  // units (= groups of 4 basis loads) issued since the last slab DMA
    // end of slab group g for this wave: its pieces of group g+1 have landed (they were issued before
    // the basis loads of the units counted in done_in_s), everybody is done reading group g, whose
    // buffer takes group g+2
    auto transition = [&]() __attribute__((always_inline)) {
      strip_barrier(done_in_s);
      STRIP_STAMP(3 + cur_g)
      if (cur_g + 2 < NG) dma_slab(cur_g + 2, cur_g & 1);
      ++cur_g;
      done_in_s = 0;
    };
    dma_slab(0, 0);
    if (NG > 1) dma_slab(1, 1);
    typedef std::integral_constant<int, 2> i2;
    load_a(i0{}, cur);
    load_a(i1{}, n1);
    load_a(i2{}, n2);
    strip_barrier(0);
    STRIP_STAMP(2)
    {
      const int target = cur.valid ? cur.s >> gsh : NG;
      while (cur_g < target) transition();
    }
    if (cur.valid) {
      x_addrs(xa, cur.s, cur.j - jbase);
      load_set(i0{}, xa);
    }

    // one unit = two steps of 12 MFMAs; the slot's basis registers then take unit u + 3
    auto unit = [&](auto slot_tag) __attribute__((always_inline)) {
      const int dj = cur.valid ? cur.j - jbase : 0;
      if (!cur.valid) x_addrs(xa, 0, 0);  // the padding unit of a strip: any row will do
      // the next unit reads the same slab buffer: its first fragments can be requested now
      const bool same = n1.valid && (n1.s >> gsh) == (cur.s >> gsh);
      // ---- step 0 (+ the reads of step 1, + the addresses of the next unit's step 0: its own row
      // once more when the next unit opens another slab -- a branch here makes hipcc keep both
      // generations of a fragment set alive, with copies)
      unsigned xn[NF];
      const int djn = same ? n1.j - jbase : dj;
      const unsigned basen = x_base(same ? n1.s : (cur.valid ? cur.s : 0));
      __builtin_amdgcn_sched_barrier(0);
      step(slot_tag, i0{}, xa, [&](int k) __attribute__((always_inline)) {
        if (k < NF) {
          const int row = xrow[k] + djn;
          xn[k] = basen + row * ROWB + 16 * ((F32 ? 2 * lh : lh) ^ ((row >> 2) & 3));
        }
      });
      // ---- step 1 (+ the reads of the next unit's step 0, + this slot's next basis fragments)
      const unsigned short *asrc = a_src(n3);
      // (the last four MFMAs still read ah[S][1] -- fp32: al[S][1] --, reloaded behind the very last one)
      step(slot_tag, i1{}, xn, [&](int k) __attribute__((always_inline)) {
        load_a1(slot_tag, asrc, F32 ? k : (k == 2 ? 3 : (k == 3 ? 2 : k)));
      });
      ++done_in_s;
      if (!same) {
        const int target = n1.valid ? n1.s >> gsh : NG;
        while (cur_g < target) transition();
        if (n1.valid) {
          x_addrs(xn, n1.s, n1.j - jbase);
          load_set(i0{}, xn);
        }
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) xa[f] = xn[f];
      cur = n1;
      n1 = n2;
      n2 = n3;
      n3 = it_next(n3);
    };
    // (three units per iteration, whatever the strip's length: a unit past the end multiplies the
    // zero block -- an early exit between them makes hipcc copy the 64 accumulator registers)
    while (cur.valid) {
      unit(i0{});
      unit(i1{});
      unit(i2{});
    }
    // (a strip with an odd number of units has just read slab fragments for its padding unit:
    // nobody may overwrite the slabs with partial sums before that)
    STRIP_STAMP(19)
    __syncthreads();
    STRIP_STAMP(20)
    // the next job is requested here -- loads return in order, so anywhere earlier the basis
    // fragments would queue up behind the atomic -- and looked at after the epilogue
    int next_job = 0;
    if (tid == 0) next_job = (int)atomicAdd(p.job_counter, 1u);

    // ---- partial sums of the waves that share a row tile: all of them through LDS at once (64 KB
    // of the dead slab buffers), every wave adds up the frame tiles the plan gave it (fmask) and
    // runs the epilogue on them
    if (!ab_epi) {
      f32x4 *red = reinterpret_cast<f32x4 *>(smem_raw);  // [wave][frame tile][4 rows of e][lane]
      // (a lambda per frame tile, not a loop: the accumulators must stay in registers)
      auto put = [&](auto f_tag) __attribute__((always_inline)) {
        constexpr int f = decltype(f_tag)::value;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[f][4 * e4 + e];
          red[((wave * 4 + f) * 4 + e4) * 64 + lane] = v;
        }
      };
      auto sum = [&](auto f_tag) __attribute__((always_inline)) {
        constexpr int f = decltype(f_tag)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
        if (fmask >> f & 1) {
          for (int w2 = g0; w2 < g0 + gsize; ++w2) {
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
              const f32x4 v = red[((w2 * 4 + f) * 4 + e4) * 64 + lane];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[f][4 * e4 + e] += v[e];
            }
          }
        }
      };
      const bool direct = p.epilogue == MISPEC_EPI_COMPLEX || p.epilogue == MISPEC_EPI_MAGNITUDE ||
                          p.epilogue == MISPEC_EPI_POWER;  // the epilogue's register path: no barriers
      // register path of the pointwise epilogue for this kernel's layout -- lane (li, lh) holds (re, im)
      // of bins 2*lh + {0, 1} + 4*k, k < 4, of frame 32*f + li: the same arithmetic as
      // bf16x3_epilogue's, with the (clip, frame) of a column taken from the tile's two runs instead
      // of a 64-bit division per call
      const int E = epilogue_width(p.epilogue);
      auto store_direct = [&](auto f_tag) __attribute__((always_inline)) {
        constexpr int f = decltype(f_tag)::value;
        const int j = 32 * f + li;
        const bool in0 = j < len0;
        const int c = in0 ? c0 : c0 + 1, t = in0 ? t0 + j : j - len0;
        const bool col_ok = n0 + j < p.n_cols;
        const int bin0 = tile_m * 16 + 2 * lh;
        float *ob = p.out + (long long)c * p.out_clip_stride + (long long)t * E +
                    (long long)(p.out_row_offset + bin0) * p.out_row_stride;
        float cu = 1.f;  // FOLD_F16X3: what undoes the clip's operand scale
        if (ARITH == FOLD_F16X3) cu = clip_unscale_of(p.clip_absmax[(long long)(c < p.n_clips ? c : 0) * CLIP_ABSMAX_STRIDE]);
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const int db = (e2 & 1) + 4 * (e2 >> 1);
          const bool ok = col_ok && bin0 + db < p.n_bins;
          float sc = (p.row_scale && ok) ? p.row_scale[bin0 + db] : 1.f;
          if (ARITH == FOLD_F16X3) sc *= cu * (ok ? p.row_unscale[bin0 + db] : 1.f);
          const float re = acc[f][2 * e2] * sc;
          const float im = p.im_sign * acc[f][2 * e2 + 1] * sc;
          float *dst = ob + (long long)db * p.out_row_stride;
          if (ok) {
            if (p.epilogue == MISPEC_EPI_COMPLEX)
              *reinterpret_cast<float2 *>(dst) = make_float2(re, im);
            else if (p.epilogue == MISPEC_EPI_MAGNITUDE)
              dst[0] = sqrtf(re * re + im * im + p.eps);
            else
              epilogue_store(p, dst, re, im);  // MISPEC_EPI_POWER
          }
        }
      };
      auto store = [&](auto f_tag) __attribute__((always_inline)) {
        constexpr int f = decltype(f_tag)::value;
        const bool mine = fmask >> f & 1;
        if (direct) {
          if (mine) store_direct(f_tag);
          return;
        }
        // the phase epilogues go through bf16x3_epilogue's LDS path, which synchronises: every wave
        // calls it (it places wave w at columns n0 + 32*w: hand it this wave's frame tile)
        f32x16 a1[1][1];
        a1[0][0] = acc[f];
        bf16x3_epilogue<1, 8, 1, 1>(p, a1, mine ? tile_m * 32 : (1 << 24), n0 + 32 * f - 32 * wave, smem_raw);
      };
      typedef std::integral_constant<int, 3> i3;
      put(i0{}), put(i1{});
      if constexpr (NF > 2) put(i2{}), put(i3{});
      __syncthreads();
      sum(i0{}), sum(i1{});
      if constexpr (NF > 2) sum(i2{}), sum(i3{});
      __syncthreads();  // the partial sums are dead: the epilogue's LDS path may use their place
      store(i0{}), store(i1{});
      if constexpr (NF > 2) store(i2{}), store(i3{});
    }
    __syncthreads();  // the epilogue is done with the LDS
    STRIP_STAMP(21)
#ifdef MISPEC_ABLATE
    stamp_on = false;
#endif
    if (tid == 0) sJob[0] = next_job;
    __syncthreads();
    job = __builtin_amdgcn_readfirstlane(sJob[0]);
  }
}

// (NF = 2: jobs of 64 frames -- twice as many, half as long -- for problems whose 128-frame jobs
// would not fill the device's workgroup slots, and for banks whose slab needs the rows)
__global__ void __launch_bounds__(256, 2) framed_bf16x3_strip_kernel(const KParams p, const StripPlan plan) {
  framed_strip_body<FOLD_BF16X3, 4>(p, plan);
}
__global__ void __launch_bounds__(256, 2) framed_bf16x3_strip64_kernel(const KParams p, const StripPlan plan) {
  framed_strip_body<FOLD_BF16X3, 2>(p, plan);
}
__global__ void __launch_bounds__(256, 2) framed_f32_strip_kernel(const KParams p, const StripPlan plan) {
  framed_strip_body<FOLD_F32, 4>(p, plan);
}
__global__ void __launch_bounds__(256, 2) framed_f32_strip64_kernel(const KParams p, const StripPlan plan) {
  framed_strip_body<FOLD_F32, 2>(p, plan);
}
__global__ void __launch_bounds__(256, 2) framed_f16x3_strip_kernel(const KParams p, const StripPlan plan) {
  framed_strip_body<FOLD_F16X3, 4>(p, plan);
}
__global__ void __launch_bounds__(256, 2) framed_f16x3_strip64_kernel(const KParams p, const StripPlan plan) {
  framed_strip_body<FOLD_F16X3, 2>(p, plan);
}
