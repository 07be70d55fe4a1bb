// stft_fft.inl -- STFT.forward (stft.py:279-316) for window x DFT bases as an FFT.
// Included by mispec.hip inside its anonymous namespace (uses KParams, epilogue_store).
//
// The reference contracts every frame with 2 x (n_fft/2+1) kernels of n_fft taps (two conv1d); when
// the kernels are  w[n] cos(2 pi k n / N), w[n] sin(2 pi k n / N)  (what fold2_basis_kernel verifies
// numerically once per basis: freq_scale='no', any window) the same numbers are the first bins of the
// N-point DFT of the windowed frame, and a radix FFT needs ~35 N log2 N / 16 flops per frame instead of
// 4 N (N/2+1): cfg2 (N = 2048) 4 GFLOP per step instead of 460 -- the step becomes a streaming pass over
// the clips (read once, through L2) and the spectrogram (written once), fp32 throughout.
//
// One wave transforms one frame at a time: the N-point real FFT as an M = N/2 point complex FFT of
// z[m] = y[2m] + i y[2m+1] (P = M/64 points per lane; Stockham passes of radix 16/16/4, 8/8/8 or 4/4/4/4
// on registers, exchanged through a padded LDS buffer of the wave's own: no workgroup barrier inside a
// frame) followed by the real-input post-processing (fft_core.h).  The twiddles, the window and the
// post-processing factors of a lane are the same for every frame: registers, set up once by a persistent
// workgroup (one per CU, 8 waves).
//
// Output is (clip, bin, frame[, 2]) with frames innermost while a wave produces all bins of ONE frame:
// the workgroup collects a tile of M bins x FT frames in LDS (columns rotated by the row so that neither
// the workgroup collects a tile of M bins x FT frames in LDS (rows of 16 .. 64 floats + 2 of padding: the
// per-frame writes of 32 consecutive bins spread over the banks) and stores it with one 16-byte store per
// lane (4 frames of a row; with 4-byte stores -- a 64-byte row segment per 16 lanes -- the store
// instructions themselves were a quarter of the kernel: ~16 cycles each in the address path, 256 of them
// per tile and CU).  Every wave requests the samples of its next frame BEFORE it stores its share of the
// finished tile, so the loads travel under the stores.  With a filterbank (mel.py:184-189) the tile -- all
// the bins of its frames -- is reduced over every filter's band right there, one thread per (filter, frame):
// the (clip, bin, frame) spectrogram is never written and no atomics are needed.
//
// LDS: tile (M + 1) x (16384 / M + 2) floats = 68-72 KB + 8 x (M + M/16 + 1) x 8 B exchange buffers + 2 x 8 M
// bytes for the window pairs and the post-processing factors (as per-lane registers next to the pass-1
// twiddles they made the N = 2048 instance spill) + the 2 KB twiddle table of pass 1: 158 KB.  The
// N = 2048 / one-float-per-output instance (the bench step) trades those two tables for a second tile buffer
// of 8 frames -- see fft_two_buffers.

constexpr int FFT_WAVES = 8;
// n_fft = 2048 with one float per output: TWO tile buffers of 8 frames (one frame per wave and step), so that a
// step is  request samples | store the previous tile | transform into the other buffer | ONE barrier  and the
// stores drain under the transforms; the window pairs and post-processing factors then live in registers (the
// second buffer takes their LDS).  Other instances: one buffer of 16 .. 64 frames, two barriers per tile.
// (A/B builds, scripts/build_variant.py: MISPEC_FFT2048_MODE 1 = ONE tile buffer of 16 frames -- two frames per wave, 64-byte row
// segments, two barriers per tile --, tables in registers as in mode 0; 2 = workgroups of FOUR waves, two frames each, one
// 8-frame tile buffer, TWO workgroups per CU)
#ifndef MISPEC_FB_UNROLL  // bins of a filter's band whose LDS reads are in flight together (flush_fb: the loop is a chain of LDS latencies)
#define MISPEC_FB_UNROLL 4
#endif
#ifndef MISPEC_FFT2048_MODE
#define MISPEC_FFT2048_MODE 0
#endif
#ifndef MISPEC_FFT_ROWSWAP  // (A/B: 0 = the second exchange of the 1024-point transform through LDS like the first, instead of row swaps)
#define MISPEC_FFT_ROWSWAP 1
#endif
// Phase clock (variant builds only: -DMISPEC_FFT_STAMPS=1, scripts/fft_stamps.py): lane 0 of waves 0 and NW / 2 of workgroup 8
// writes s_memtime at the phase boundaries of tile steps 4 .. 11 to the buffer named by the environment variable
// MISPEC_FFT_STAMPS (launch_fft_cfg hands it over in KParams::job_counter, which this kernel does not use otherwise).
#ifndef MISPEC_FFT_STAMPS
#define MISPEC_FFT_STAMPS 0
#endif
#if MISPEC_FFT_STAMPS
#define FFT_STAMP(k)                                                                                              \
  do {                                                                                                            \
    if (p.job_counter && blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == NW / 2) && step >= 4 && step < 12) \
      reinterpret_cast<unsigned long long *>(p.job_counter)[(((wave != 0) * 8 + (step - 4)) * 16) + (k)] =        \
          __builtin_readcyclecounter();                                                                           \
  } while (0)
#else
#define FFT_STAMP(k) do { } while (0)
#endif
template <int M, int W>
constexpr int fft_waves() {  // waves of a workgroup (mode 2: FOUR, two frames each, and two workgroups per CU that drift apart)
  return (M == 1024 && W == 1 && MISPEC_FFT2048_MODE == 2) ? 4 : FFT_WAVES;
}
template <int M, int W>
constexpr bool fft_reg_tables() {  // window pairs and post-processing factors in registers (their LDS goes to the tiles)
  return M == 1024 && W == 1;
}
template <int M, int W>
constexpr bool fft_two_buffers() {
  return M == 1024 && W == 1 && MISPEC_FFT2048_MODE == 0;  // (n_fft = 1024 with 2 x 16 frames: Mel cfg3 0.116 -> 0.121 ms, not kept)
}
// Frames per tile of the one-buffer instances (FB: the instance with the fused filterbank).  Small tiles make
// the workgroup small enough -- LDS and, with __launch_bounds__' second argument, 128 VGPRs -- for TWO
// workgroups per CU (fft_two_per_cu): four waves per SIMD, and one workgroup reduces / stores its tile while
// the other transforms.  The choices are measured ones (scripts/fft_variants_time.py).
#ifndef MISPEC_FFT512_FT_W1
#define MISPEC_FFT512_FT_W1 16
#endif
#ifndef MISPEC_FFT512_FT_W2
#define MISPEC_FFT512_FT_W2 16
#endif
#ifndef MISPEC_FFT512_FT_FB
#define MISPEC_FFT512_FT_FB 8  // (16 frames leave no LDS for the packed band weights next to a second workgroup)
#endif
#ifndef MISPEC_FFT256_FT_W1
#define MISPEC_FFT256_FT_W1 32
#endif
#ifndef MISPEC_FFT256_FT_W2
#define MISPEC_FFT256_FT_W2 16
#endif
#ifndef MISPEC_FFT256_FT_FB
#define MISPEC_FFT256_FT_FB 32
#endif
template <int M, int W, bool FB = false>
constexpr int fft_tile_row() {  // floats per tile row: the tile's frames x W + 2 of padding
  return (M == 1024 ? ((W == 1 && MISPEC_FFT2048_MODE == 1) ? 2 * FFT_WAVES : FFT_WAVES)
                    : M == 512 ? (FB ? MISPEC_FFT512_FT_FB : W == 1 ? MISPEC_FFT512_FT_W1 : MISPEC_FFT512_FT_W2)
                               : (FB ? MISPEC_FFT256_FT_FB : W == 1 ? MISPEC_FFT256_FT_W1 : MISPEC_FFT256_FT_W2)) * W + 2;
}
template <int M, int W, bool FB = false>
constexpr size_t stft_fft_smem() {
  return (size_t)(fft_two_buffers<M, W>() ? 2 : 1) * (M + 1) * fft_tile_row<M, W, FB>() * 4 +
         (size_t)fft_waves<M, W>() * fftcore::padded_size<M>() * 8 + (fft_reg_tables<M, W>() ? 0 : 2 * (size_t)M * 8) +
         (size_t)fftcore::radix_of<M, 0>() * (fftcore::radix_of<M, 1>() - 1) * 8;  // (+ the twiddle table of pass 1)
}
template <int M, int W, bool FB = false>
constexpr bool fft_two_per_cu() {  // two workgroups fit the CU's 160 KB (FB: with 6 KB each for packed band weights)
  return 2 * (((stft_fft_smem<M, W, FB>() + 15) & ~(size_t)15) + (FB ? 6144 : 0)) <= 160 * 1024;
}
template <int M, int EPI, bool FB>
constexpr int fft_min_waves() {  // waves per SIMD the register allocation must leave room for
  constexpr int W = (EPI == MISPEC_EPI_COMPLEX || EPI == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
  return (fft_two_per_cu<M, W, FB>() ? 2 : 1) * fft_waves<M, W>() / 4;
}

// 16 bytes per lane, global -> LDS at m0 + 16 lane, as instructions: the loads of a frame stay invisible to
// the compiler's s_waitcnt placement -- with ordinary loads requested before the tile's stores it waited for
// vmcnt(0) in front of the first use (the store loop's trip count is not a constant), i.e. for the stores to
// drain: 0.05 ms of a 0.19 ms kernel.  The wait is stated by hand (fft_wait_vm): loads and stores of a
// wave retire in order on one counter, so "at most as many operations outstanding as stores were issued
// after the loads" means the loads have landed.
__device__ __forceinline__ void fft_dma16(const void *src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void fft_wait_vm(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: case 13: case 14: case 15: case 16: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
  }
}

// Exchange between the second and the third pass of the 1024-point transform without LDS: lane 16 g + k holds
// outputs 256 g + k + 16 r (r = register), the radix-4 pass wants index lane + 64 i in slot i -- for each
// c = r >> 2 a 4 x 4 transpose between the wave's four 16-lane rows (g) and the registers 4 c + a, which two
// rounds of gfx950's row swaps do (v_permlane32_swap: upper half of one register <-> lower half of the
// other; v_permlane16_swap: odd rows <-> even rows): 32 VALU instructions instead of 16 LDS writes + 16 LDS
// reads and their latency (the LDS pipe was the busiest unit of the kernel: 56 % against 39 % for the VALU).
__device__ __forceinline__ void fft_swap32(fftcore::cf &a, fftcore::cf &b) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const u2 y = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = fftcore::cf{__uint_as_float(x[0]), __uint_as_float(y[0])};
  b = fftcore::cf{__uint_as_float(x[1]), __uint_as_float(y[1])};
}
__device__ __forceinline__ void fft_swap16(fftcore::cf &a, fftcore::cf &b) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  const u2 x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const u2 y = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = fftcore::cf{__uint_as_float(x[0]), __uint_as_float(y[0])};
  b = fftcore::cf{__uint_as_float(x[1]), __uint_as_float(y[1])};
}
__device__ __forceinline__ void fft_rows_to_slots(fftcore::cf (&x)[16]) {
  fftcore::cf y[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    fftcore::cf r0 = x[4 * c], r1 = x[4 * c + 1], r2 = x[4 * c + 2], r3 = x[4 * c + 3];
    fft_swap32(r0, r2);
    fft_swap32(r1, r3);
    fft_swap16(r0, r1);
    fft_swap16(r2, r3);
    y[c] = r0;       // slot 4 g + c <- what row g held
    y[4 + c] = r1;
    y[8 + c] = r2;
    y[12 + c] = r3;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = y[i];
}

// the pointwise epilogue with a compile-time kind (the shared epilogue_store switches at run time: 16 copies
// of the switch per frame made the kernel's control flow -- and its register allocation -- unmanageable)
template <int EPI>
__device__ __forceinline__ void fft_epilogue(const KParams &p, float re, float im, float &v0, float &v1) {
  v1 = 0.f;
  if constexpr (EPI == MISPEC_EPI_COMPLEX) {
    v0 = re;
    v1 = im;
  } else if constexpr (EPI == MISPEC_EPI_MAGNITUDE) {
    v0 = __builtin_amdgcn_sqrtf(re * re + im * im + p.eps);  // (v_sqrt_f32: 1 ulp; sqrtf's fix-up code is 12 instructions per bin)
  } else if constexpr (EPI == MISPEC_EPI_POWER) {
    const float s = re * re + im * im + p.eps;
    v0 = (p.power == 2.0f && p.eps == 0.f) ? s : (p.power == 1.0f ? sqrtf(s) : powf(sqrtf(s), p.power));
  } else if constexpr (EPI == MISPEC_EPI_PHASE_ATAN2) {
    v0 = atan2f(im + 0.0f, re);
  } else if constexpr (EPI == MISPEC_EPI_PHASE_COSSIN) {
    const float a = atan2f(im, re);
    v0 = cosf(a);
    v1 = sinf(a);
  } else {
    v0 = re;
  }
}

// ... and from the squared magnitude s = re^2 + im^2 + eps (Magnitude / Power: the post-processing below forms it packed)
template <int EPI>
__device__ __forceinline__ float fft_epilogue_sq(const KParams &p, float s) {
  if constexpr (EPI == MISPEC_EPI_MAGNITUDE)
    return __builtin_amdgcn_sqrtf(s);  // (v_sqrt_f32: 1 ulp; sqrtf's fix-up code is 12 instructions per bin)
  else
    return (p.power == 2.0f && p.eps == 0.f) ? s : (p.power == 1.0f ? sqrtf(s) : powf(sqrtf(s), p.power));
}

// Real-input post-processing of the pair (k, M - k) as EIGHT packed instructions (round 5: the compiler's version of
// fft_core.h real_post_pair + the epilogue was 23 per pair, and this kernel is bound by the instructions its two waves per SIMD
// can issue, not by the vector pipe's throughput).  The spectrum arrives HALVED (the window registers / table carry the factor
// 1/2 of the formula) and w = e^(-2 pi i k / N) in full:
//   S = zk + conj(zm), D = zk - conj(zm), wb = D w;   X[k] = (S.x + wb.y, S.y - wb.x),  X[M - k] = (S.x - wb.y, -(S.y + wb.x))
// fft_post_pair_sq returns (|X[k]|^2, |X[M - k]|^2) + eps, formed from the re parts of both and the im parts of both side by
// side; fft_post_pair_c the two complex values.  op_sel / op_sel_hi pick the 32-bit half of a source that feeds the low / high
// lane of the packed operation, neg_lo / neg_hi negate it (checked against the plain formula: experiments/pk_post).
__device__ __forceinline__ fftcore::cf fft_post_pair_sq(fftcore::cf zk, fftcore::cf zm, fftcore::cf w, fftcore::cf eps2) {
  fftcore::cf S, D, t, wb, X, Y, q, r;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(S) : "v"(zk), "v"(zm));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(D) : "v"(zk), "v"(zm));
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(D), "v"(w));                                                // (D.x w.x, D.y w.x)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(wb) : "v"(D), "v"(w), "v"(t));  // + (-D.y w.y, D.x w.y)
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(X) : "v"(S), "v"(wb));                    // (S.x + wb.y, S.x - wb.y)
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(Y) : "v"(S), "v"(wb));                    // (S.y - wb.x, S.y + wb.x)
  asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(q) : "v"(Y), "v"(eps2));
  asm("v_pk_fma_f32 %0, %1, %1, %2" : "=v"(r) : "v"(X), "v"(q));
  return r;
}
__device__ __forceinline__ void fft_post_pair_c(fftcore::cf zk, fftcore::cf zm, fftcore::cf w, fftcore::cf &xk, fftcore::cf &xm) {
  fftcore::cf S, D, t, wb;
  asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(S) : "v"(zk), "v"(zm));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(D) : "v"(zk), "v"(zm));
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(D), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(wb) : "v"(D), "v"(w), "v"(t));
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(xk) : "v"(S), "v"(wb));                    // (S.x + wb.y, S.y - wb.x)
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(xm) : "v"(S), "v"(wb));       // (S.x - wb.y, -S.y - wb.x)
}

// frames of a tile
template <int M, int W, bool FB = false>
constexpr int fft_tile_frames() {
  return (fft_tile_row<M, W, FB>() - 2) / W;
}

// M = n_fft / 2; EPI = the epilogue (W = floats per output element: 2 for Complex / Phase as (cos, sin)); FB = with
// the fused filterbank (p.fb; EPI = MISPEC_EPI_POWER)
// CEPI >= 0: the n_fft = 4096 composite's second transform (see cmb below), CEPI = the CALLER'S epilogue
// FM (round 5): FRAME-MAJOR output (mispec_framed_gemm_args.out_frame_major: element (c, bin, t) at
// out + c out_clip_stride + t out_row_stride + bin; the floats [n_bins, out_row_stride) of a frame's row are zeroed) -- the
// spectrum leaves the registers of the post-processing directly, 256 contiguous bytes per store instruction, no tile, no
// flush, no workgroup barrier in the tile loop.  For consumers that contract over the bins of a frame (the dense
// filterbank of Gammatonegram: the power spectrogram is then the framed operand of the contraction kernels).
template <int M, int EPI, bool FB, int CEPI = -1, bool FM = false>
__global__ void __launch_bounds__((fft_waves<M, (EPI == MISPEC_EPI_COMPLEX || EPI == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1>() * 64), (fft_min_waves<M, EPI, FB>()))
    stft_fft_kernel(const KParams p, const int tiles_per_clip) {
  using namespace fftcore;
  constexpr int N = 2 * M, P = M / 64;
  constexpr int W = (EPI == MISPEC_EPI_COMPLEX || EPI == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
  constexpr int FT = fft_tile_frames<M, W, FB>();  // frames per tile
  constexpr int NW = fft_waves<M, W>();            // waves of the workgroup
  constexpr int FPW = FT / NW;          // frames per wave and tile
  constexpr int C = fft_tile_row<M, W, FB>();  // floats per tile row (FT * W + 2)
  constexpr bool DB = fft_two_buffers<M, W>();
  constexpr bool RT = fft_reg_tables<M, W>();
  constexpr int TILE_FLOATS = (M + 1) * C;     // rows 0 .. M (the Nyquist bin)
  constexpr int FFT_TILE_BYTES = (DB ? 2 : 1) * TILE_FLOATS * 4;
  static_assert(FPW >= 1 && (C & 1) == 0, "tile geometry");
  static_assert(!FM || (W == 1 && !FB && CEPI < 0 && M >= 512), "frame-major output: Magnitude / Power of the full spectrum");
  // n_fft = 256 (or any power of two below) on the 512-point instance: the frame is zero-extended to N samples (a window of n_fft
  // taps followed by zeros), whose spectrum has the n_fft-point bins at every RS-th row of the tile
  constexpr bool ZP = M == 256;
  const int RS = ZP ? p.fft_row_step : 1;
  const int half_taps = ZP ? p.K / 2 : M;  // window pairs that exist
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *const tiles = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cf *const buf = reinterpret_cast<cf *>(smem_raw + FFT_TILE_BYTES) + wave * padded_size<M>();

  // ---- window pairs (w[2m], w[2m+1]) / 2 and post-processing factors e^(-2 pi i k / N): per-workgroup tables,
  // or (two tile buffers) the lane's own in registers
  cf *const s_win = reinterpret_cast<cf *>(smem_raw + FFT_TILE_BYTES) + NW * padded_size<M>();
  cf *const s_wh = s_win + M;
  typedef __attribute__((address_space(3))) void *lptr_t;
  const unsigned buf_lds = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lptr_t)buf);
  cf wn[RT ? P : 1], whr[RT ? P / 2 : 1];
  if constexpr (RT) {
#pragma unroll
    for (int i = 0; i < P; ++i) wn[i] = 0.5f * *reinterpret_cast<const cf *>(p.a_re + 2 * (lane + 64 * i));  // (window / 2: see fft_post_pair_sq)
    // The window pairs are the only values of the tile loop that come from global LOADS the compiler knows about.
    // Pin them here: hipcc's waitcnt pass otherwise carries "wn[i] may still be in flight" into the loop (the loop
    // header merges the prologue's state) and places s_waitcnt vmcnt(6) / (2) / (0) in front of the window
    // multiplications of EVERY frame -- i.e. right behind the hand-stated fft_wait_vm the wave waited for the
    // flush's stores it had just issued (round 5: found in the ISA; the stores' acknowledgements take ~2 us).
#pragma unroll
    for (int i = 0; i < P; ++i) asm volatile("" : "+v"(wn[i]));
#pragma unroll
    for (int i = 0; i < P / 2; ++i) {
      float sn, cs;
      sincospif(-(float)(lane + 64 * i) / (float)M, &sn, &cs);
      whr[i] = cf{cs, sn};
    }
  } else {
    for (int m = tid; m < M; m += NW * 64) {
      // row 0 of the cosine kernels is the window itself
      s_win[m] = m < half_taps ? 0.5f * *reinterpret_cast<const cf *>(p.a_re + 2 * m) : cf{0.f, 0.f};  // (window / 2: see fft_post_pair_sq)
      float sn, cs;
      sincospif(-(float)m / (float)M, &sn, &cs);
      s_wh[m] = cf{cs, sn};
    }
    __syncthreads();
  }
  // twiddles of pass 1, W^(r k) with k = lane mod R0 for every butterfly of the lane: a table of R0 x (R1 - 1)
  // factors (as registers they are 30 VGPRs of the N = 2048 instance, which then has no room to request the
  // next frame's samples while it transforms this one)
  constexpr int R0 = radix_of<M, 0>(), R1 = radix_of<M, 1>();
  cf *const s_tw1 = RT ? s_win : s_wh + M;
  for (int i = tid; i < R0 * (R1 - 1); i += NW * 64) {
    const int k = i / (R1 - 1), r = i % (R1 - 1) + 1;
    float sn, cs;
    sincospif(-2.f * (float)(r * k) / (float)(R0 * R1), &sn, &cs);
    s_tw1[i] = cf{cs, sn};
  }
  __syncthreads();
  const cf *const tw1 = s_tw1 + (lane & (R0 - 1)) * (R1 - 1);
  // ---- per-lane constants: the twiddles of the later passes
  cf tw[tw_total<M>() > 0 ? tw_total<M>() : 1];
  auto fill_tw = [&](auto pass_tag) __attribute__((always_inline)) {
    constexpr int PASS = decltype(pass_tag)::value;
#pragma unroll
    for (int i = 0; i < tw_count<M, PASS>(); ++i) {
      float s, c;
      sincospif(2.f * tw_turns<M, PASS>(lane, i), &s, &c);
      tw[tw_offset<M, PASS>() + i] = cf{c, s};
    }
  };
  fill_tw(std::integral_constant<int, 2>{});
  if constexpr (Radix<M>::n > 3) fill_tw(std::integral_constant<int, 3>{});
  auto twf1 = [&](int, int r) __attribute__((always_inline)) { return tw1[r - 1]; };
  auto twf2 = [&](int q, int r) __attribute__((always_inline)) { return tw[tw_offset<M, 2>() + q * (radix_of<M, 2>() - 1) + r - 1]; };
  auto twf3 = [&](int q, int r) __attribute__((always_inline)) { return tw[tw_offset<M, 3>() + q * (radix_of<M, 3>() - 1) + r - 1]; };
  auto twf0 = [](int, int) __attribute__((always_inline)) { return cf{1.f, 0.f}; };

  const int n_tiles = p.n_clips * tiles_per_clip;
  const int hop = p.hop, L = p.n_samples, T = p.n_frames;
  // n_fft = 4096 composite (mispec.hip launch_fft4096): this instance transforms the ODD samples; its tile (O) is not stored, the
  // flush reads E (p.cmb_E) and writes X[k] = E[k] + W^k O[k] and X[2048 - k] = conj(E[k] - W^k O[k]) through the caller's epilogue
  constexpr bool CMB_OK = CEPI >= 0;
  static_assert(!CMB_OK || (M == 1024 && EPI == MISPEC_EPI_COMPLEX && !FB), "the composite's second transform is the Complex 2048-point instance");
  constexpr bool cmb = CMB_OK;
  const int n_rows = cmb ? M + 1 : (p.n_bins < M + 1 ? p.n_bins : M + 1);  // rows of the tile that are stored
  // wave-level ordering of the exchange buffer: the LDS executes a wave's instructions in order; only the
  // compiler has to be kept from moving accesses across
  auto wave_sync = []() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto store = [&](int o, cf v) __attribute__((always_inline)) { buf[pad(o)] = v; };
  auto reload = [&](cf (&x)[P]) __attribute__((always_inline)) {
    wave_sync();
#pragma unroll
    for (int i = 0; i < P; ++i) x[i] = buf[pad(lane + 64 * i)];
    wave_sync();
  };
  // tile -> memory: a lane stores 4 floats of a row (4 / W frames)
  auto flush = [&](const float *tile, float *oc, int t0) __attribute__((always_inline)) {
    constexpr int LPR = FT * W / 4;               // lanes per row
    constexpr int RPI = NW * 64 / LPR;     // rows per iteration
    const int fl = (tid % LPR) * (4 / W), r0 = tid / LPR;  // first frame of the lane's quad
    if (t0 + fl < T && !MISPEC_DBG(p, 0x1)) {
      const bool whole = t0 + fl + 4 / W <= T;
#pragma unroll 4
      for (int k = r0; k < n_rows; k += RPI) {
        const cf *src = reinterpret_cast<const cf *>(tile + k * RS * C + fl * W);
        const cf lo = src[0], hi = src[1];
        float *d = oc + (long long)k * p.out_row_stride + (long long)(t0 + fl) * W;
        if (MISPEC_DBG(p, 0x8)) d = p.out + (long long)k * p.out_row_stride + (long long)fl * W;  // benchmarking: every tile onto the first one (no HBM write stream)
        if (whole) {
          *reinterpret_cast<f32x4u *>(d) = f32x4u{lo.x, lo.y, hi.x, hi.y};
        } else {  // the clip's last frames
          const float v[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (t0 + fl + e / W < T) d[e] = v[e];
        }
      }
    }
  };

  // ---- n_fft = 4096 composite: rows of E for the tile just transformed, requested at the end of its step (ahead of the barrier;
  // the flush at the top of the next step uses them); lane = (row r0 + 128 j, frame pair fl)
  constexpr int CNIT = CMB_OK ? (M + 1 + NW * 64 / (FT * W / 4) - 1) / (NW * 64 / (FT * W / 4)) : 1;
  f32x4v ce[CNIT];
  auto cmb_request = [&](int c, int t0) __attribute__((always_inline)) {
    constexpr int LPR = FT * W / 4, RPI = NW * 64 / LPR;
    const int fl = (tid % LPR) * (4 / W), r0 = tid / LPR;
#pragma unroll
    for (int j = 0; j < CNIT; ++j) {
      const int k = r0 + RPI * j;
      ce[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      if (k <= M && t0 + fl < T) {
        const float *src = p.cmb_E + (((long long)c * (M + 1) + k) * T + t0 + fl) * 2;
        if (t0 + fl + 1 < T) {
          const f32x4u v = *reinterpret_cast<const f32x4u *>(src);
          ce[j] = f32x4v{v[0], v[1], v[2], v[3]};
        } else {
          ce[j] = f32x4v{src[0], src[1], 0.f, 0.f};
        }
      }
    }
  };
  auto cmb_epilogue = [&](float re, float im, float &v0, float &v1) __attribute__((always_inline)) {
    fft_epilogue<CMB_OK ? CEPI : MISPEC_EPI_COMPLEX>(p, re, im, v0, v1);  // (the CALLER'S epilogue; this instance's own -- Complex -- only shapes the tile)
  };
  auto flush_cmb = [&](const float *tile, float *oc, int t0) __attribute__((always_inline)) {
    constexpr int LPR = FT * W / 4, RPI = NW * 64 / LPR;
    const int fl = (tid % LPR) * (4 / W), r0 = tid / LPR;
    if (t0 + fl >= T || MISPEC_DBG(p, 0x1)) return;
    constexpr int wo = (CEPI == MISPEC_EPI_COMPLEX || CEPI == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;  // floats per output
    const float ims = -p.im_sign;
    const bool two = t0 + fl + 1 < T;
#pragma unroll
    for (int j = 0; j < CNIT; ++j) {
      const int k = r0 + RPI * j;
      if (k > M) continue;
      const cf *src = reinterpret_cast<const cf *>(tile + k * C + fl * W);
      const cf o0 = src[0], o1 = src[1];
      // W^k = e^(-2 pi i k / 4096): the post-processing table holds e^(-2 pi i m / 2048) -- every second power
      cf wk = s_wh[k >> 1];
      if (k & 1) wk = cmul(wk, cf{0.99999882345170188f, -0.0015339801862847655f});  // x e^(-2 pi i / 4096)
      const cf e0 = cf{ce[j][0], ce[j][1]}, e1 = cf{ce[j][2], ce[j][3]};
      const cf w0 = cmul(o0, wk), w1 = cmul(o1, wk);
      float v[2][2][2];  // [row k | row 2048 - k][frame][component]
      cmb_epilogue(e0.x + w0.x, ims * (e0.y + w0.y), v[0][0][0], v[0][0][1]);
      cmb_epilogue(e1.x + w1.x, ims * (e1.y + w1.y), v[0][1][0], v[0][1][1]);
      cmb_epilogue(e0.x - w0.x, ims * -(e0.y - w0.y), v[1][0][0], v[1][0][1]);  // conj(E - W O)
      cmb_epilogue(e1.x - w1.x, ims * -(e1.y - w1.y), v[1][1][0], v[1][1][1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = h ? 2 * M - k : k;
        if (row >= p.n_bins || (h && k == M)) continue;
        float *d = oc + (long long)row * p.out_row_stride + (long long)(t0 + fl) * wo;
        if (wo == 2) {
          *reinterpret_cast<cf *>(d) = cf{v[h][0][0], v[h][0][1]};
          if (two) *reinterpret_cast<cf *>(d + 2) = cf{v[h][1][0], v[h][1][1]};
        } else {
          d[0] = v[h][0][0];
          if (two) d[1] = v[h][1][0];
        }
      }
    }
  };

  // ---- fused filterbank: the band weights packed into LDS (s_fboff[m] = start of filter m's band [lo, hi)), when
  // they fit what the launch reserved; built once by the persistent workgroup
  int *const s_fboff = reinterpret_cast<int *>(smem_raw + ((stft_fft_smem<M, W, FB>() + 15) & ~(size_t)15));
  int *const s_fblo = s_fboff + 260;  // first bin of filter m's band
  float *const s_fbw = reinterpret_cast<float *>(s_fboff + 520);
  bool fb_packed = false;
  if (FB && p.fb_lds_floats > 0) {
    for (int m = tid; m < p.n_fb; m += NW * 64) {
      int lo = p.fb_support[2 * m], hi = p.fb_support[2 * m + 1];
      lo = lo < 0 ? 0 : lo;
      hi = hi > n_rows ? n_rows : hi;
      s_fboff[m + 1] = hi > lo ? hi - lo : 0;
      s_fblo[m] = lo;
    }
    __syncthreads();
    if (tid == 0) {
      int acc = 0;
      s_fboff[0] = 0;
      for (int m = 0; m < p.n_fb; ++m) {
        acc += s_fboff[m + 1];
        s_fboff[m + 1] = acc;
      }
      s_fboff[257] = acc <= p.fb_lds_floats ? 1 : 0;
    }
    __syncthreads();
    fb_packed = s_fboff[257] != 0;
    if (fb_packed) {
      for (int m = wave; m < p.n_fb; m += NW) {
        int lo = p.fb_support[2 * m], hi = p.fb_support[2 * m + 1];
        lo = lo < 0 ? 0 : lo;
        hi = hi > n_rows ? n_rows : hi;
        const float *w = p.fb + (long long)m * p.fb_row_stride;
        float *d = s_fbw + s_fboff[m];
        for (int b = lo + lane; b < hi; b += 64) d[b - lo] = w[b];
      }
    }
    __syncthreads();
  }
  // tile -> filterbank outputs: out[c, m, t] = sum over the band of fb[m, bin] * tile[bin][t].  Packed weights:
  // a thread takes TWO frames of a filter (8-byte tile reads, one weight read per bin), else consecutive lanes
  // take consecutive frames of a filter and the weight is a broadcast load from memory
  auto flush_fb = [&](const float *tile, float *oc, int t0) __attribute__((always_inline)) {
    if (MISPEC_DBG(p, 0x1)) return;
    if (fb_packed) {
      for (int idx = tid; idx < p.n_fb * (FT / 2); idx += NW * 64) {
        const int m = idx / (FT / 2), fl = 2 * (idx - m * (FT / 2));
        const int lo = s_fblo[m];
        const int nb = s_fboff[m + 1] - s_fboff[m];
        const float *w = s_fbw + s_fboff[m];
        const int CR = C * RS;
        const float *tr = tile + lo * CR + fl;
        cf sum = cf{0.f, 0.f};
#pragma unroll MISPEC_FB_UNROLL
        for (int b = 0; b < nb; ++b) sum += w[b] * *reinterpret_cast<const cf *>(tr + b * CR);
        float *d = oc + (long long)m * p.out_row_stride + t0 + fl;
        if (t0 + fl < T) d[0] = sum.x;
        if (t0 + fl + 1 < T) d[1] = sum.y;
      }
      return;
    }
    for (int idx = tid; idx < p.n_fb * FT; idx += NW * 64) {
      const int m = idx / FT, fl = idx - m * FT;
      int lo = p.fb_support[2 * m], hi = p.fb_support[2 * m + 1];
      lo = lo < 0 ? 0 : lo;
      hi = hi > n_rows ? n_rows : hi;
      const float *w = p.fb + (long long)m * p.fb_row_stride;
      float sum = 0.f;
#pragma unroll 4
      for (int b = lo; b < hi; ++b) sum += w[b] * tile[b * RS * C + fl];
      if (t0 + fl < T) oc[(long long)m * p.out_row_stride + t0 + fl] = sum;
    }
  };

  // persistent workgroup; consecutive tiles of a clip stay on one XCD (workgroup b runs on XCD b % 8)
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  if (MISPEC_DBG(p, 0x20)) {  // benchmarking: workgroups out of phase (quarters of a tile time)
    for (int z = 0; z < ((blockIdx.x >> 3) & 3); ++z) __builtin_amdgcn_s_sleep(120);
  }
  float *prev_oc = nullptr;  // the tile waiting to be stored
  int prev_t0 = 0, step = 0;
  const int n_flush_stores = [&]() {
    constexpr int LPR = FT * W / 4, RPI = NW * 64 / LPR;
    const int r_min = wave * 64 / LPR;
    return (n_rows > r_min && !MISPEC_DBG(p, 0x1)) ? (n_rows - r_min + RPI - 1) / RPI : 0;
  }();
  bool pre = false;          // the first frame of this wave in the coming tile has been requested already (see below)
  for (int it = blockIdx.x >> 3; it < per_xcd; it += (nwg + 7) >> 3) {
    const int tile_id = (blockIdx.x & 7) * per_xcd + it;
    if (tile_id >= n_tiles) continue;  // (workgroup-uniform)
    // the tile after this one (tile ids grow with `it`: behind the first one past the end there is none)
    const int tile_nx = (it + ((nwg + 7) >> 3) < per_xcd) ? tile_id + ((nwg + 7) >> 3) : n_tiles;
    const int c = tile_id / tiles_per_clip;
    const int t0 = (tile_id - c * tiles_per_clip) * FT;
    const float *const xc = p.x + (long long)c * p.x_clip_stride;
    float *const oc = p.out + (long long)c * p.out_clip_stride + (long long)p.out_row_offset * p.out_row_stride;
    float *const tile = tiles + (DB ? (step & 1) * TILE_FLOATS : 0);
    const float *const prev_tile = tiles + (DB ? ((step & 1) ^ 1) * TILE_FLOATS : 0);
    cf xn[P];             // samples of the wave's next frame of this tile, requested a frame ahead
    bool fast_n = false;
    FFT_STAMP(0);
#pragma unroll 1
    for (int u = 0; u < FPW; ++u) {
      const int f = wave * FPW + u, t = t0 + f;
      // ---- the frame: y[n] = w[n] x[t hop - pad + n], packed as z[m] = (y[2m], y[2m+1]); the samples are
      // requested first, the previous tile is stored while they travel
      const long long pos0 = (long long)t * hop - p.pad;
      cf x[P];
      const bool live = t < T;
      const bool fast = u == 0 ? (live && pos0 >= 0 && pos0 + N <= L) : fast_n;
      if (u == 0 && fast && !pre) {  // the frame as it lies in memory -> the wave's exchange buffer (idle here), 1 KB per instruction
#pragma unroll
        for (int j = 0; j < N / 256; ++j)
          fft_dma16((MISPEC_DBG(p, 0x40) ? p.x : xc + pos0) + 256 * j + 4 * lane, buf_lds + 1024 * j);  // (0x40: every frame = the first 8 KB)
      }
      const bool had_pre = u == 0 && pre;
      if (u == 0) pre = false;
      int younger = 0;  // store instructions this wave issues after the loads
      if (u == 0) {
        if (prev_oc) {
          if constexpr (FB) {
            // (the reduction's stores are not counted -- their number depends on the bands --: a frame requested early is
            // waited for BEFORE them, when nothing younger than it is in flight; one requested just now after them, with them)
            if (had_pre) fft_wait_vm(0);
            flush_fb(prev_tile, prev_oc, prev_t0);
          } else if (cmb) {
            if (had_pre) fft_wait_vm(0);  // (loads and stores of this flush are not counted: as with the fused filterbank)
            flush_cmb(prev_tile, prev_oc, prev_t0);
          } else {
            flush(prev_tile, prev_oc, prev_t0);
            younger += n_flush_stores;
          }
        }
        if constexpr (!DB && !FM) __syncthreads();  // the only buffer: everyone has read the tile before it is refilled
        if (FM && had_pre) younger += P + 1;  // (frame-major: the P + 1 stores of the previous frame followed its request)
      }
      FFT_STAMP(1);
      if (!live) continue;
      if (fast && u == 0) {
        if (!((FB || cmb) && had_pre && prev_oc)) fft_wait_vm(younger);
        FFT_STAMP(2);
        const cf *const plain = buf;
#pragma unroll
        for (int i = 0; i < P; ++i) x[i] = plain[lane + 64 * i];
        wave_sync();
      } else if (fast) {
#pragma unroll
        for (int i = 0; i < P; ++i) x[i] = xn[i];
      } else {
        // edge frames (a few per clip): sample by sample through the exchange buffer
#pragma unroll 1
        for (int m = lane; m < M; m += 64) {
          float v[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            long long q = pos0 + 2 * m + e;
            if (p.pad_mode == MISPEC_PAD_REFLECT) {
              q = q < 0 ? -q : q;
              q = q >= L ? 2LL * L - 2 - q : q;
            }
            v[e] = (q >= 0 && q < L) ? xc[q] : 0.f;
          }
          buf[pad(m)] = cf{v[0], v[1]};
        }
        reload(x);
      }
      if (u + 1 < FPW) {  // the next frame of this wave: plain loads, a whole transform ahead of their use
        const long long pos_n = pos0 + hop;
        fast_n = t + 1 < T && pos_n >= 0 && pos_n + N <= L;
        if (fast_n) {
          typedef float cfu __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
          for (int i = 0; i < P; ++i) {
            const cfu v = *reinterpret_cast<const cfu *>(xc + pos_n + 2 * (lane + 64 * i));
            xn[i] = cf{v.x, v.y};
          }
        }
      }
#pragma unroll
      for (int i = 0; i < P; ++i) x[i] = x[i] * (RT ? wn[RT ? i : 0] : s_win[lane + 64 * i]);
      if (ZP && RS > 1) {  // the samples behind the frame are not part of it, whatever they are (Inf x 0)
#pragma unroll
        for (int i = 0; i < P; ++i) x[i] = lane + 64 * i < half_taps ? x[i] : cf{0.f, 0.f};
      }
      // ---- M-point complex FFT   (benchmarking build: 0x4 skips the passes, 0x2 the post-processing, 0x1 the stores)
      FFT_STAMP(3);
      if (!MISPEC_DBG(p, 0x4)) {
        stockham_pass<M, 0>(x, lane, twf0, store);
        FFT_STAMP(4);
        reload(x);
        FFT_STAMP(5);
        if constexpr (M == 1024 && MISPEC_FFT_ROWSWAP) {
          stockham_pass<M, 1>(x, lane, twf1, [](int, cf) {});
          FFT_STAMP(6);
          fft_rows_to_slots(x);
        } else {
          stockham_pass<M, 1>(x, lane, twf1, store);
          reload(x);
        }
        FFT_STAMP(7);
        // (the last pass leaves the spectrum in the lanes' slots; the buffer only serves the mirrored reads of
        // the post-processing, which touch the upper half)
        stockham_pass<M, 2>(x, lane, twf2, [&](int o, cf v) __attribute__((always_inline)) {
          if (Radix<M>::n > 3 || o >= M / 2) buf[pad(o)] = v;
        });
        if constexpr (Radix<M>::n > 3) {
          reload(x);
          stockham_pass<M, 3>(x, lane, twf3, store);
        }
      }
      wave_sync();
      FFT_STAMP(8);
      // ---- real-input post-processing of the pairs (k, M - k), k = lane + 64 i < M/2, epilogue, into the
      // tile.  Every address is a per-lane base + a compile-time multiple of i: mirror Z[M - k] at
      // pad(M - lane) - 68 i, rows k and M - k of the tile.
      if (!MISPEC_DBG(p, 0x2)) {
        const cf *const zmp = buf + pad(M - lane);
        const cf *const whp = s_wh + lane;
        float *const ta = tile + lane * C + W * f;
        float *const tb = tile + (M - lane) * C + W * f;
        const float ims = cmb ? 1.f : -p.im_sign;  // (composite: the tile holds O itself)
        float *const fm_row = p.out + (long long)c * p.out_clip_stride + (long long)t * p.out_row_stride;  // (FM)
        // the mirrored values first: behind them the exchange buffer is idle until this wave's next frame, and the FIRST
        // frame of its NEXT tile is requested right here (round 5) -- it travels under the post-processing, the barrier
        // and the flush instead of being asked for at the top of the next step and waited for at once (the wave's own
        // buffer: no other wave is concerned; the request still precedes the flush's stores, so fft_wait_vm's count of
        // younger operations is what it was)
        // (only the instances with two waves per SIMD: those that run two workgroups per CU live within 128 VGPRs, where
        // the eight values held across the request spill, and have four waves per SIMD to cover the latency anyway)
        constexpr bool EARLY = fft_min_waves<M, EPI, FB>() <= 2;
        cf zmv[EARLY ? P / 2 : 1];
        if constexpr (EARLY) {
#pragma unroll
          for (int i = 0; i < P / 2; ++i) zmv[i] = zmp[-68 * i];
        }
        if (EARLY && u == FPW - 1 && tile_nx < n_tiles && !MISPEC_DBG(p, 0x80)) {
          const int cn = tile_nx / tiles_per_clip;
          const int tn = (tile_nx - cn * tiles_per_clip) * FT + wave * FPW;
          const long long posn = (long long)tn * hop - p.pad;
          if (tn < T && posn >= 0 && posn + N <= L) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the reads above have left the buffer)
            const float *const src = (MISPEC_DBG(p, 0x40) ? p.x : p.x + (long long)cn * p.x_clip_stride + posn) + 4 * lane;
#pragma unroll
            for (int j = 0; j < N / 256; ++j) fft_dma16(src + 256 * j, buf_lds + 1024 * j);
            pre = true;
          }
        }
        FFT_STAMP(9);
#pragma unroll
        for (int i = 0; i < P / 2; ++i) {
          cf zm = EARLY ? zmv[EARLY ? i : 0] : zmp[-68 * i];
          if (i == 0) zm = lane == 0 ? x[0] : zm;  // bin 0 pairs with itself: (X[0], Nyquist bin)
          const cf wk = RT ? whr[RT ? i : 0] : whp[64 * i];
          float a0, a1 = 0.f, b0, b1 = 0.f;
          if constexpr (EPI == MISPEC_EPI_MAGNITUDE || EPI == MISPEC_EPI_POWER) {
            const cf sq = fft_post_pair_sq(x[i], zm, wk, cf{p.eps, p.eps});
            a0 = fft_epilogue_sq<EPI>(p, sq.x);
            b0 = fft_epilogue_sq<EPI>(p, sq.y);
          } else {
            cf xk, xm;
            fft_post_pair_c(x[i], zm, wk, xk, xm);
            fft_epilogue<EPI>(p, xk.x, ims * xk.y, a0, a1);
            fft_epilogue<EPI>(p, xm.x, ims * xm.y, b0, b1);
          }
          if constexpr (FM) {  // rows k = lane + 64 i and M - k of THIS frame: two runs of 64 consecutive floats
            fm_row[lane + 64 * i] = a0;
            fm_row[M - lane - 64 * i] = b0;
          } else if constexpr (W == 2) {
            *reinterpret_cast<cf *>(ta + 64 * C * i) = cf{a0, a1};
            *reinterpret_cast<cf *>(tb - 64 * C * i) = cf{b0, b1};
          } else {
            ta[64 * C * i] = a0;
            tb[-64 * C * i] = b0;
          }
        }
        // bin M/2 is its own mirror: X = conj(Z[M/2]), lane 0's slot P/2
        float h0, h1;
        fft_epilogue<EPI>(p, 2.f * x[P / 2].x, -ims * 2.f * x[P / 2].y, h0, h1);  // (the halved spectrum)
        if constexpr (FM) {  // ONE store instruction: lane 0 the bin M/2, lanes 1 .. the zeros behind bin M (the row's padding)
          const int kz = M + lane;
          if (lane == 0 || kz < (int)p.out_row_stride) fm_row[lane == 0 ? M / 2 : kz] = lane == 0 ? h0 : 0.f;
        } else if (lane == 0) {
          float *th = tile + (M / 2) * C + W * f;
          th[0] = h0;
          if constexpr (W == 2) th[1] = h1;
        }
      }
      wave_sync();  // the mirrored reads are done before the next frame's first pass overwrites the buffer
    }
    FFT_STAMP(10);
    // (composite: E's rows for THIS tile, used by the flush at the top of the next step -- requested here, behind the transform:
    // held across it they made the instance spill; they travel under the barrier and the next step's waits)
    if (cmb) cmb_request(c, t0);
    FFT_STAMP(11);
    if constexpr (!FM) __syncthreads();  // the tile is complete (two buffers: and the other one has been read out)
    FFT_STAMP(12);
    prev_oc = FM ? nullptr : oc;
    prev_t0 = t0;
    ++step;
  }
  if (prev_oc) {
    const float *const last = tiles + (DB ? ((step & 1) ^ 1) * TILE_FLOATS : 0);
    if constexpr (FB)
      flush_fb(last, prev_oc, prev_t0);
    else if (cmb)
      flush_cmb(last, prev_oc, prev_t0);
    else
      flush(last, prev_oc, prev_t0);
  }
}

// ---------------------------------------------------------------------------------
// Inverse direction: frame synthesis of the inverse STFT (STFTBase.inverse_stft, stft.py:15-63, step 1 of
// mispec_istft_frames_f32) for DFT synthesis kernels and a one-sided spectrum, as an inverse real FFT:
//   frames[c, t, n] = sum_{k=0}^{N-1} Xh[k] e^(+2 pi i k n / N),  Xh = the Hermitian extension of spec[c, :, t]
// (what the contraction with [cos | -sin] and the mirrored bins folded in computes; the imaginary parts of
// the DC and Nyquist bins do not contribute).  Mirror image of stft_fft_kernel: the workgroup gathers a tile of
// M + 1 bins x 8 frames of the (clip, bin, frame, 2) spectrogram into LDS (64-byte row segments; the next
// tile's rows are requested into registers before this tile is transformed), every wave pre-processes one
// frame into the M-point spectrum Z (fft_core.h: real_pre_conj), runs the same Stockham passes -- the
// inverse transform as conj(FFT(conj Z)) -- and stores its 2 M samples straight from registers (sample
// innermost: 512 contiguous bytes per instruction).  The windowed overlap-add stays mispec_overlap_add_f32.
// ---------------------------------------------------------------------------------
template <int M>
constexpr size_t istft_fft_smem() {
  return (size_t)(M + 1) * 18 * 4 + (size_t)FFT_WAVES * fftcore::padded_size<M>() * 8 +
         (size_t)fftcore::radix_of<M, 0>() * (fftcore::radix_of<M, 1>() - 1) * 8;
}

template <int M>
__global__ void __launch_bounds__(FFT_WAVES * 64) istft_fft_kernel(const float *__restrict__ spec, const int n_clips,
                                                                    const int n_frames, float *__restrict__ frames,
                                                                    const int tiles_per_clip) {
  using namespace fftcore;
  constexpr int N = 2 * M, P = M / 64, F = M + 1;
  constexpr int FT = FFT_WAVES;  // frames per tile: one per wave
  constexpr int C = 18;          // floats per tile row: 8 frames x (re, im) + 2 of padding
  constexpr int RPI = FFT_WAVES * 64 / 4, NIT = (F + RPI - 1) / RPI;  // gather: 4 lanes per row, 2 frames per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *const tile = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cf *const buf = reinterpret_cast<cf *>(smem_raw + F * C * 4) + wave * padded_size<M>();
  constexpr int R0 = radix_of<M, 0>(), R1 = radix_of<M, 1>();
  cf *const s_tw1 = reinterpret_cast<cf *>(smem_raw + F * C * 4) + FFT_WAVES * padded_size<M>();
  for (int i = tid; i < R0 * (R1 - 1); i += FFT_WAVES * 64) {
    const int k = i / (R1 - 1), r = i % (R1 - 1) + 1;
    float sn, cs;
    sincospif(-2.f * (float)(r * k) / (float)(R0 * R1), &sn, &cs);
    s_tw1[i] = cf{cs, sn};
  }
  const cf *const tw1 = s_tw1 + (lane & (R0 - 1)) * (R1 - 1);
  cf tw[tw_total<M>() > 0 ? tw_total<M>() : 1], wpre[P];
  auto fill_tw = [&](auto pass_tag) __attribute__((always_inline)) {
    constexpr int PASS = decltype(pass_tag)::value;
#pragma unroll
    for (int i = 0; i < tw_count<M, PASS>(); ++i) {
      float s, c;
      sincospif(2.f * tw_turns<M, PASS>(lane, i), &s, &c);
      tw[tw_offset<M, PASS>() + i] = cf{c, s};
    }
  };
  fill_tw(std::integral_constant<int, 2>{});
  if constexpr (Radix<M>::n > 3) fill_tw(std::integral_constant<int, 3>{});
#pragma unroll
  for (int i = 0; i < P; ++i) {  // e^(+2 pi i k / N), k = lane + 64 i
    float sn, cs;
    sincospif((float)(lane + 64 * i) / (float)M, &sn, &cs);
    wpre[i] = cf{cs, sn};
  }
  auto twf0 = [](int, int) __attribute__((always_inline)) { return cf{1.f, 0.f}; };
  auto twf1 = [&](int, int r) __attribute__((always_inline)) { return tw1[r - 1]; };
  auto twf2 = [&](int q, int r) __attribute__((always_inline)) { return tw[tw_offset<M, 2>() + q * (radix_of<M, 2>() - 1) + r - 1]; };
  auto twf3 = [&](int q, int r) __attribute__((always_inline)) { return tw[tw_offset<M, 3>() + q * (radix_of<M, 3>() - 1) + r - 1]; };
  auto wave_sync = []() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto store = [&](int o, cf v) __attribute__((always_inline)) { buf[pad(o)] = v; };
  auto reload = [&](cf (&x)[P]) __attribute__((always_inline)) {
    wave_sync();
#pragma unroll
    for (int i = 0; i < P; ++i) x[i] = buf[pad(lane + 64 * i)];
    wave_sync();
  };

  const int T = n_frames;
  const int n_tiles = n_clips * tiles_per_clip;
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8, stride = (nwg + 7) >> 3;
  const int q2 = 2 * (tid & 3), r0 = tid >> 2;  // gather: this lane's two frames of a row, its first row
  // the rows of a tile this lane gathers: 16 bytes each (frames t0 + q2, t0 + q2 + 1 of bin r0 + RPI j)
  f32x4v g[NIT];
  auto gather = [&](int tile_id) __attribute__((always_inline)) {
    const int c = tile_id / tiles_per_clip;
    const int t0 = (tile_id - c * tiles_per_clip) * FT;
    const float *const sc = spec + (long long)c * F * T * 2;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int k = r0 + RPI * j;
      g[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      if (k < F && t0 + q2 < T) {
        const float *src = sc + ((long long)k * T + t0 + q2) * 2;
        if (t0 + q2 + 1 < T)
          g[j] = *reinterpret_cast<const f32x4u *>(src);
        else
          g[j] = f32x4v{src[0], src[1], 0.f, 0.f};
      }
    }
  };
  auto scatter = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int k = r0 + RPI * j;
      if (k < F) {
        cf *d = reinterpret_cast<cf *>(tile + k * C + 2 * q2);
        d[0] = cf{g[j][0], g[j][1]};
        d[1] = cf{g[j][2], g[j][3]};
      }
    }
  };
  int it = blockIdx.x >> 3;
  int tile_id = (blockIdx.x & 7) * per_xcd + it;
  bool have = it < per_xcd && tile_id < n_tiles;
  if (have) gather(tile_id);
  __syncthreads();  // (the twiddle table)
  while (have) {
    scatter();
    __syncthreads();
    const int cur = tile_id;
    it += stride;
    tile_id = (blockIdx.x & 7) * per_xcd + it;
    have = it < per_xcd && tile_id < n_tiles;
    if (have) gather(tile_id);  // the next tile travels while this one is transformed
    const int c = cur / tiles_per_clip;
    const int t = (cur - c * tiles_per_clip) * FT + wave;
    if (t < T) {
      cf x[P];
      const cf *const col = reinterpret_cast<const cf *>(tile) + wave;  // row k at col[k * (C / 2)]
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int k = lane + 64 * i;
        cf gk = col[k * (C / 2)], gm = col[(M - k) * (C / 2)];
        if (i == 0 && lane == 0) {  // DC and Nyquist: real
          gk.y = 0.f;
          gm.y = 0.f;
        }
        x[i] = real_pre_conj(gk, gm, wpre[i]);
      }
      stockham_pass<M, 0>(x, lane, twf0, store);
      reload(x);
      if constexpr (M == 1024) {
        stockham_pass<M, 1>(x, lane, twf1, [](int, cf) {});
        fft_rows_to_slots(x);
      } else {
        stockham_pass<M, 1>(x, lane, twf1, store);
        reload(x);
      }
      if constexpr (Radix<M>::n > 3) {
        stockham_pass<M, 2>(x, lane, twf2, store);
        reload(x);
        stockham_pass<M, 3>(x, lane, twf3, [](int, cf) {});
      } else {
        stockham_pass<M, 2>(x, lane, twf2, [](int, cf) {});
      }
      wave_sync();
      // (round 5: the next tile's rows -- plain loads requested before this transform -- are pinned HERE, in front of this
      // iteration's stores: hipcc otherwise waits for them with vmcnt(0) at the top of the next iteration, i.e. for the stores
      // it has just issued -- scripts/isa_waits.py)
#pragma unroll
      for (int j = 0; j < NIT; ++j) asm volatile("" : "+v"(g[j]));
      float *const out = frames + ((long long)c * T + t) * N;
#pragma unroll
      for (int i = 0; i < P; ++i)  // z = conj(FFT(conj Z)): y[2m] = Re, y[2m + 1] = -Im
        *reinterpret_cast<cf *>(out + 2 * (lane + 64 * i)) = cf{x[i].x, -x[i].y};
    }
    __syncthreads();  // the tile has been read
  }
}

// ---------------------------------------------------------------------------------
// Inverse STFT in ONE launch: frame synthesis (above) + windowed overlap-add + window-sum-square
// normalisation (stft.py:33-54, utils.py:43-57) without the (clip, frame, sample) round trip through HBM
// (452 MB written and re-read for a cfg2-sized inverse).  A workgroup owns a RUN of consecutive 8-frame
// tiles of one clip and walks it in order:
//   * every wave transforms one frame of the tile (as istft_fft_kernel) and writes  frame[n] * win[n] / N  into
//     its own exchange buffer (free once the transform is over);
//   * barrier; thread j sums samples q = j, j + 512, ... of the tile's span [0, 8 hop + N - hop): the CARRY of
//     the tiles before (partial sums that were waiting for their later frames) + the tile's frames in
//     ascending order -- the summation order of overlap_add_kernel, so the fused result is bit-identical
//     to the two-launch one -- ; the first 8 hop samples are final (divided by the window sum, trimmed to
//     [start, start + out_len), stored), the rest is the next tile's carry.  8 hop is a multiple of the 512
//     threads (hop % 64 == 0), so the thread that reads carry[q] is the one that rewrites it: no second
//     carry buffer, no race.  hop divides N (a power of two): no divisions in the sums.
//   * a run that does not start the clip first walks the ceil((N - hop) / (8 hop)) tiles before it without
//     storing (their frames build its first carry); the last run walks one tile past the clip's frames,
//     which flushes the carry.
// ---------------------------------------------------------------------------------
template <int M>
constexpr size_t istft_ola_smem(int hop) {
  return ((istft_fft_smem<M>() + 15) & ~(size_t)15) + (size_t)2 * M * 4 + (size_t)(2 * M - hop) * 4 + (size_t)hop * 4;
}

template <int M>
__global__ void __launch_bounds__(FFT_WAVES * 64) istft_ola_fft_kernel(const float *__restrict__ spec, const int n_clips,
                                                                        const int n_frames, const float *__restrict__ win,
                                                                        const int hop, const int start, const int out_len,
                                                                        float *__restrict__ out, const long long out_clip_stride,
                                                                        const int runs_per_clip, const int tiles_per_run,
                                                                        const int n_tiles_clip) {
  using namespace fftcore;
  constexpr int N = 2 * M, P = M / 64, F = M + 1;
  constexpr int FT = FFT_WAVES;  // frames per tile: one per wave
  constexpr int C = 18;          // floats per tile row: 8 frames x (re, im) + 2 of padding
  constexpr int NT = FFT_WAVES * 64;
  constexpr int RPI = NT / 4, NIT = (F + RPI - 1) / RPI;  // gather: 4 lanes per row, 2 frames per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *const tile = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cf *const buf0 = reinterpret_cast<cf *>(smem_raw + F * C * 4);
  cf *const buf = buf0 + wave * padded_size<M>();
  constexpr int R0 = radix_of<M, 0>(), R1 = radix_of<M, 1>();
  cf *const s_tw1 = buf0 + FFT_WAVES * padded_size<M>();
  float *const s_win = reinterpret_cast<float *>(smem_raw + ((istft_fft_smem<M>() + 15) & ~(size_t)15));
  float *const s_carry = s_win + N;
  const int n_carry = N - hop;
  float *const s_wss = s_carry + n_carry;  // [hop]: window sum of a sample all of whose K frames exist
  const int hop_log2 = 31 - __builtin_clz((unsigned)hop), K = N / hop;
  for (int i = tid; i < R0 * (R1 - 1); i += NT) {
    const int k = i / (R1 - 1), r = i % (R1 - 1) + 1;
    float sn, cs;
    sincospif(-2.f * (float)(r * k) / (float)(R0 * R1), &sn, &cs);
    s_tw1[i] = cf{cs, sn};
  }
  for (int i = tid; i < N; i += NT) s_win[i] = win[i];
  for (int i = tid; i < n_carry; i += NT) s_carry[i] = 0.f;
  for (int r = tid; r < hop; r += NT) {  // (ascending frame = descending n: the order of the sums it stands for)
    float a = 0.f;
    for (int kk = 0; kk < K; ++kk) {
      const float w = win[r + (K - 1 - kk) * hop];
      a += w * w;
    }
    s_wss[r] = a;
  }
  const cf *const tw1 = s_tw1 + (lane & (R0 - 1)) * (R1 - 1);
  cf tw[tw_total<M>() > 0 ? tw_total<M>() : 1], wpre[P];
  auto fill_tw = [&](auto pass_tag) __attribute__((always_inline)) {
    constexpr int PASS = decltype(pass_tag)::value;
#pragma unroll
    for (int i = 0; i < tw_count<M, PASS>(); ++i) {
      float s, c;
      sincospif(2.f * tw_turns<M, PASS>(lane, i), &s, &c);
      tw[tw_offset<M, PASS>() + i] = cf{c, s};
    }
  };
  fill_tw(std::integral_constant<int, 2>{});
  if constexpr (Radix<M>::n > 3) fill_tw(std::integral_constant<int, 3>{});
#pragma unroll
  for (int i = 0; i < P; ++i) {  // e^(+2 pi i k / N), k = lane + 64 i
    float sn, cs;
    sincospif((float)(lane + 64 * i) / (float)M, &sn, &cs);
    wpre[i] = cf{cs, sn};
  }
  auto twf0 = [](int, int) __attribute__((always_inline)) { return cf{1.f, 0.f}; };
  auto twf1 = [&](int, int r) __attribute__((always_inline)) { return tw1[r - 1]; };
  auto twf2 = [&](int q, int r) __attribute__((always_inline)) { return tw[tw_offset<M, 2>() + q * (radix_of<M, 2>() - 1) + r - 1]; };
  auto twf3 = [&](int q, int r) __attribute__((always_inline)) { return tw[tw_offset<M, 3>() + q * (radix_of<M, 3>() - 1) + r - 1]; };
  auto wave_sync = []() __attribute__((always_inline)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto store = [&](int o, cf v) __attribute__((always_inline)) { buf[pad(o)] = v; };
  auto reload = [&](cf (&x)[P]) __attribute__((always_inline)) {
    wave_sync();
#pragma unroll
    for (int i = 0; i < P; ++i) x[i] = buf[pad(lane + 64 * i)];
    wave_sync();
  };

  const int T = n_frames;
  const int c = blockIdx.x / runs_per_clip;
  const int run = blockIdx.x - c * runs_per_clip;
  const int ta = run * tiles_per_run;                       // first tile this run stores
  const int tb = (ta + tiles_per_run < n_tiles_clip) ? ta + tiles_per_run : n_tiles_clip;
  if (ta >= tb) return;
  const int n_warm = (n_carry + FT * hop - 1) / (FT * hop);  // tiles whose frames reach into tile ta
  const int t_first = ta - n_warm > 0 ? ta - n_warm : 0;
  const float *const sc = spec + (long long)c * F * T * 2;
  float *const oc = out + (long long)c * out_clip_stride;
  const float inv_n = 1.0f / (float)N;
  // the windowed frame of wave j: N floats in its exchange buffer, from a 16-byte aligned address (a buffer is
  // an odd number of 8-byte elements long, and so is the tile in front of them)
  auto frame_buf = [&](int j) __attribute__((always_inline)) -> float * {
    return reinterpret_cast<float *>(buf0 + j * padded_size<M>()) + 2 * ((F * C / 2 + j * padded_size<M>()) & 1);
  };
  // (scalar) the 16-byte form of the overlap-add: see the sums below
  const bool vec4 = (hop & 255) == 0 && ((start & 3) == 0) && ((out_clip_stride & 3) == 0) &&
                    ((reinterpret_cast<unsigned long long>(out) & 15) == 0);
  const int q2 = 2 * (tid & 3), r0 = tid >> 2;  // gather: this lane's two frames of a row, its first row
  f32x4v g[NIT];
  auto gather = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int k = r0 + RPI * j;
      g[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      if (k < F && t0 + q2 < T) {
        const float *src = sc + ((long long)k * T + t0 + q2) * 2;
        if (t0 + q2 + 1 < T)
          g[j] = *reinterpret_cast<const f32x4u *>(src);
        else
          g[j] = f32x4v{src[0], src[1], 0.f, 0.f};
      }
    }
  };
  auto scatter = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int k = r0 + RPI * j;
      if (k < F) {
        cf *d = reinterpret_cast<cf *>(tile + k * C + 2 * q2);
        d[0] = cf{g[j][0], g[j][1]};
        d[1] = cf{g[j][2], g[j][3]};
      }
    }
  };
  gather(t_first * FT);
  __syncthreads();  // (the tables, the zeroed carry)
  for (int tl = t_first; tl < tb; ++tl) {
    const int t0 = tl * FT;
    scatter();
    __syncthreads();
    if (tl + 1 < tb) gather((tl + 1) * FT);  // the next tile travels while this one is transformed
    const int t = t0 + wave;
    if (t < T) {
      cf x[P];
      const cf *const col = reinterpret_cast<const cf *>(tile) + wave;  // row k at col[k * (C / 2)]
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int k = lane + 64 * i;
        cf gk = col[k * (C / 2)], gm = col[(M - k) * (C / 2)];
        if (i == 0 && lane == 0) {  // DC and Nyquist: real
          gk.y = 0.f;
          gm.y = 0.f;
        }
        x[i] = real_pre_conj(gk, gm, wpre[i]);
      }
      stockham_pass<M, 0>(x, lane, twf0, store);
      reload(x);
      if constexpr (M == 1024) {
        stockham_pass<M, 1>(x, lane, twf1, [](int, cf) {});
        fft_rows_to_slots(x);
      } else {
        stockham_pass<M, 1>(x, lane, twf1, store);
        reload(x);
      }
      if constexpr (Radix<M>::n > 3) {
        stockham_pass<M, 2>(x, lane, twf2, store);
        reload(x);
        stockham_pass<M, 3>(x, lane, twf3, [](int, cf) {});
      } else {
        stockham_pass<M, 2>(x, lane, twf2, [](int, cf) {});
      }
      wave_sync();
      // z = conj(FFT(conj Z)): y[2m] = Re, y[2m + 1] = -Im; windowed and scaled as overlap_add_kernel does it
      float *const fb = frame_buf(wave);
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const int n = 2 * (lane + 64 * i);
        const cf w = *reinterpret_cast<const cf *>(s_win + n);
        *reinterpret_cast<cf *>(fb + n) = cf{x[i].x * w.x * inv_n, -x[i].y * w.y * inv_n};
      }
    }
    // (round 5: the next tile's rows -- plain loads requested before this transform -- are pinned HERE, in front of the
    // overlap-add's stores: hipcc otherwise waits for them with vmcnt(0) at the top of the next iteration, i.e. for the
    // stores it has just issued -- scripts/isa_waits.py)
#pragma unroll
    for (int j = 0; j < NIT; ++j) asm volatile("" : "+v"(g[j]));
    __syncthreads();  // the frames of the tile stand in the exchange buffers
    {
      const bool emit = tl >= ta;
      const int final_n = FT * hop, span = final_n + n_carry;
      const long long base = (long long)t0 * hop;  // position of q = 0 on the un-trimmed overlap-add axis
      // hop is a power of two that divides N: sample q = b hop + r of the span is covered by the K = N / hop frames
      // j = b - K + 1 .. b of the tile (those that exist), at n = r + (b - j) hop; ascending j = the order of
      // overlap_add_kernel.  (carry[q] is rewritten by the thread that reads it, in a later iteration of this
      // loop: FT * hop % NT == 0)
      // hop % 256 == 0 (8 hop a multiple of 4 NT) and a 16-byte aligned output: a thread takes FOUR CONSECUTIVE
      // samples (same block b: one 16-byte LDS read per frame, one 16-byte store), the carry entries it reads are
      // again the ones it rewrites later
      if (vec4) {
        for (int q = 4 * tid; q < span; q += 4 * NT) {
          f32x4v acc = q < n_carry ? *reinterpret_cast<const f32x4v *>(s_carry + q) : f32x4v{0.f, 0.f, 0.f, 0.f};
          const int b = q >> hop_log2, r = q & (hop - 1);
          for (int kk = 0; kk < K; ++kk) {
            const int j = b - K + 1 + kk;
            if (j >= 0 && j < FT && t0 + j < T)
              acc += *reinterpret_cast<const f32x4v *>(frame_buf(j) + r + (K - 1 - kk) * hop);
          }
          if (q < final_n) {
            const long long i = base + q - start;
            if (emit && i + 3 >= 0 && i < out_len) {
              const int t_hi = t0 + b, t_lo = t_hi - K + 1;
              f32x4v wss;
              if (t_lo >= 0 && t_hi < T) {
                wss = *reinterpret_cast<const f32x4v *>(s_wss + r);
              } else {
                wss = f32x4v{0.f, 0.f, 0.f, 0.f};
                for (int kk = 0; kk < K; ++kk) {
                  const int tt = t_lo + kk;
                  if (tt >= 0 && tt < T) {
                    const f32x4v w = *reinterpret_cast<const f32x4v *>(s_win + r + (K - 1 - kk) * hop);
                    wss += w * w;
                  }
                }
              }
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (wss[e] > 1e-10f) acc[e] /= wss[e];
              if (i >= 0 && i + 3 < out_len) {
                *reinterpret_cast<f32x4v *>(oc + i) = acc;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (i + e >= 0 && i + e < out_len) oc[i + e] = acc[e];
              }
            }
          } else {
            *reinterpret_cast<f32x4v *>(s_carry + q - final_n) = acc;
          }
        }
        __syncthreads();
        continue;
      }
      // (four samples of a thread at a time: their sums are independent chains of LDS round trips)
      constexpr int U = 4;
      for (int q0 = tid; q0 < span; q0 += U * NT) {
        float acc[U];
        int bb[U], rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int q = q0 + u * NT;
          acc[u] = q < n_carry ? s_carry[q] : 0.f;  // (q >= span: below n_carry only if inside the span)
          bb[u] = q >> hop_log2;
          rr[u] = q & (hop - 1);
        }
        for (int kk = 0; kk < K; ++kk) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int j = bb[u] - K + 1 + kk;
            if (q0 + u * NT < span && j >= 0 && j < FT && t0 + j < T)
              acc[u] += frame_buf(j)[rr[u] + (K - 1 - kk) * hop];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int q = q0 + u * NT;
          if (q >= span) continue;
          float a = acc[u];
          if (q < final_n) {
            const long long i = base + q - start;
            if (emit && i >= 0 && i < out_len) {
              // the window sum over ALL frames of the clip that cover the sample (ascending, as overlap_add_kernel):
              // frames t0 + b - K + 1 .. t0 + b; where they all exist the sum is a function of r alone (s_wss)
              const int t_hi = t0 + bb[u], t_lo = t_hi - K + 1;
              float wss;
              if (t_lo >= 0 && t_hi < T) {
                wss = s_wss[rr[u]];
              } else {
                wss = 0.f;
                for (int kk = 0; kk < K; ++kk) {
                  const int tt = t_lo + kk;
                  if (tt >= 0 && tt < T) {
                    const float w = s_win[rr[u] + (K - 1 - kk) * hop];
                    wss += w * w;
                  }
                }
              }
              if (wss > 1e-10f) a /= wss;
              oc[i] = a;
            }
          } else {
            s_carry[q - final_n] = a;
          }
        }
      }
    }
    __syncthreads();  // the exchange buffers (and the carry) may be rewritten
  }
}
