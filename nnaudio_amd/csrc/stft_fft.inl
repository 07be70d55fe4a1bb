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
// the workgroup collects a tile of M bins x FT frames in LDS (64 KB; columns rotated by the row so that
// neither the per-frame writes nor the per-row reads pile up on a bank) and then stores whole row
// segments (64-128 bytes).  The Nyquist bin goes straight to memory.
//
// LDS: 64 KB tile + 8 x (M + M/16) x 8 B exchange buffers + 2 x 8 M bytes for the window pairs and the
// post-processing factors (M = 1024: 150 KB; as per-lane registers they cost 64 VGPRs and the N = 2048
// instance spilled).

constexpr int FFT_WAVES = 8;
constexpr int FFT_TILE_FLOATS = 16384;  // 64 KB: M rows x (16384 / M) columns

template <int M>
constexpr size_t stft_fft_smem() {
  return (size_t)FFT_TILE_FLOATS * sizeof(float) + (size_t)FFT_WAVES * fftcore::padded_size<M>() * 8 + 2 * (size_t)M * 8;
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
    v0 = sqrtf(re * re + im * im + p.eps);
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

// M = n_fft / 2; EPI = the epilogue (W = floats per output element: 2 for Complex / Phase as (cos, sin))
template <int M, int EPI>
__global__ void __launch_bounds__(FFT_WAVES * 64) stft_fft_kernel(const KParams p, const int tiles_per_clip) {
  using namespace fftcore;
  constexpr int N = 2 * M, P = M / 64;
  constexpr int W = (EPI == MISPEC_EPI_COMPLEX || EPI == MISPEC_EPI_PHASE_COSSIN) ? 2 : 1;
  constexpr int C = FFT_TILE_FLOATS / M;  // floats per tile row
  constexpr int FT = C / W;               // frames per tile
  constexpr int FPW = FT / FFT_WAVES;     // frames per wave and tile
  static_assert(FPW >= 1, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float *const tile = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  cf *const buf = reinterpret_cast<cf *>(smem_raw + FFT_TILE_FLOATS * sizeof(float)) + wave * padded_size<M>();

  // ---- per-workgroup tables: window pairs (w[2m], w[2m+1]) and e^(-2 pi i m / N) / 2
  cf *const s_win = reinterpret_cast<cf *>(smem_raw + FFT_TILE_FLOATS * sizeof(float)) + FFT_WAVES * padded_size<M>();
  cf *const s_wh = s_win + M;
  for (int m = tid; m < M; m += FFT_WAVES * 64) {
    s_win[m] = *reinterpret_cast<const cf *>(p.a_re + 2 * m);  // row 0 of the cosine kernels is the window itself
    float s, c;
    sincospif(-(float)m / (float)M, &s, &c);
    s_wh[m] = cf{0.5f * c, 0.5f * s};
  }
  __syncthreads();
  // ---- per-lane constants: the twiddles of the passes
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
  fill_tw(std::integral_constant<int, 1>{});
  fill_tw(std::integral_constant<int, 2>{});
  if constexpr (Radix<M>::n > 3) fill_tw(std::integral_constant<int, 3>{});

  const int n_tiles = p.n_clips * tiles_per_clip;
  const int hop = p.hop, L = p.n_samples, T = p.n_frames;
  const int n_rows = p.n_bins < M ? p.n_bins : M;  // rows of the tile that are stored
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

  // persistent workgroup; consecutive tiles of a clip stay on one XCD (workgroup b runs on XCD b % 8)
  const int nwg = gridDim.x, per_xcd = (n_tiles + 7) / 8;
  for (int it = blockIdx.x >> 3; it < per_xcd; it += (nwg + 7) >> 3) {
    const int tile_id = (blockIdx.x & 7) * per_xcd + it;
    if (tile_id < n_tiles) {
      const int c = tile_id / tiles_per_clip;
      const int t0 = (tile_id - c * tiles_per_clip) * FT;
      const float *const xc = p.x + (long long)c * p.x_clip_stride;
      float *const oc = p.out + (long long)c * p.out_clip_stride + (long long)p.out_row_offset * p.out_row_stride;
#pragma unroll 1
      for (int u = 0; u < FPW; ++u) {
        const int f = wave * FPW + u, t = t0 + f;
        if (t >= T) break;
        // ---- the frame: y[n] = w[n] x[t hop - pad + n], packed as z[m] = (y[2m], y[2m+1])
        const long long pos0 = (long long)t * hop - p.pad;
        cf x[P];
        const bool fast = pos0 >= 0 && pos0 + N <= L && ((reinterpret_cast<unsigned long long>(xc + pos0) & 7) == 0);
        if (fast) {
#pragma unroll
          for (int i = 0; i < P; ++i) x[i] = *reinterpret_cast<const cf *>(xc + pos0 + 2 * (lane + 64 * i));
        } else {
          // edge frames (a few per clip) and odd alignments: sample by sample through the exchange buffer
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
#pragma unroll
        for (int i = 0; i < P; ++i) x[i] = x[i] * s_win[lane + 64 * i];
        // ---- M-point complex FFT
        stockham_pass<M, 0>(x, lane, tw, store);
        reload(x);
        stockham_pass<M, 1>(x, lane, tw + tw_offset<M, 1>(), store);
        reload(x);
        stockham_pass<M, 2>(x, lane, tw + tw_offset<M, 2>(), store);
        if constexpr (Radix<M>::n > 3) {
          reload(x);
          stockham_pass<M, 3>(x, lane, tw + tw_offset<M, 3>(), store);
        }
        wave_sync();
        // ---- real-input post-processing, epilogue, into the tile
#pragma unroll
        for (int i = 0; i < P; ++i) {
          const int k = lane + 64 * i;
          const cf zm = buf[pad((M - k) & (M - 1))];
          const cf X = real_post(x[i], zm, s_wh[k]);
          float v0, v1;
          fft_epilogue<EPI>(p, X.x, -p.im_sign * X.y, v0, v1);
          const int col = (f * W + W * k) & (C - 1);
          if constexpr (W == 2)
            *reinterpret_cast<cf *>(tile + k * C + col) = cf{v0, v1};
          else
            tile[k * C + col] = v0;
        }
        if (p.n_bins > M && lane == 0) {  // Nyquist bin: Re Z0 - Im Z0
          float *d = oc + (long long)M * p.out_row_stride + (long long)t * W;
          float v0, v1;
          fft_epilogue<EPI>(p, x[0].x - x[0].y, 0.f, v0, v1);
          d[0] = v0;
          if constexpr (W == 2) d[1] = v1;
        }
        wave_sync();  // the mirrored reads are done before the next frame's first pass overwrites the buffer
      }
      __syncthreads();
      // ---- tile -> memory: FT lanes per row, W floats per lane
      constexpr int RPI = FFT_WAVES * 64 / FT;  // rows per iteration
      const int fl = tid % FT, r0 = tid / FT;
      if (t0 + fl < T) {
#pragma unroll 4
        for (int k = r0; k < n_rows; k += RPI) {
          const int col = (fl * W + W * k) & (C - 1);
          float *d = oc + (long long)k * p.out_row_stride + (long long)(t0 + fl) * W;
          if constexpr (W == 2)
            *reinterpret_cast<cf *>(d) = *reinterpret_cast<const cf *>(tile + k * C + col);
          else
            *d = tile[k * C + col];
        }
      }
    }
    __syncthreads();
  }
}
