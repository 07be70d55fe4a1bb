// mispec_internal.h -- what the translation units of libmispec.so share besides include/mispec.h.
// Hidden visibility: none of this is part of the C ABI.
#pragma once
#include <cstddef>
#include <cstdint>

#define MISPEC_HIDDEN __attribute__((visibility("hidden")))

// records the message mispec_last_error() returns on this thread, returns `code`
MISPEC_HIDDEN int mispec_fail_msg(int code, const char *msg);
// compute units of the current device (cached per device)
MISPEC_HIDDEN int mispec_device_cus();

// Taps per row of the split basis / bank planes (mispec_split_basis_*: a row is its kernel rounded up to the K depth of an
// LDS stage, zero filled).  ONE definition for every translation unit that strides those planes: mispec.hip writes them,
// octave_stream.hip reads the banks (the second unit had the 32 spelled out).
constexpr int MISPEC_SPLIT_KC = 32;
constexpr int mispec_split_row_taps(int kernel) { return (kernel + MISPEC_SPLIT_KC - 1) / MISPEC_SPLIT_KC * MISPEC_SPLIT_KC; }

// cqt_chain.hip: the chain kernel of CQT1992v2's fp32 contraction (basis_chain of mispec_framed_gemm_args)
struct mispec_framed_gemm_args;
MISPEC_HIDDEN int64_t mispec_chain_bytes_impl(const int32_t *row_support_host, int32_t n_bins, int32_t kernel);
MISPEC_HIDDEN int mispec_chain_pack_impl(const float *basis_re, const float *basis_im, int64_t basis_row_stride, int32_t n_bins,
                                         int32_t kernel, const int32_t *row_support_host, void *dst, int64_t dst_bytes, void *stream);
MISPEC_HIDDEN int mispec_chain_ok(const mispec_framed_gemm_args *a);  // 1: the chain kernel serves this call
MISPEC_HIDDEN int mispec_chain_launch(const mispec_framed_gemm_args *a, int debug, void *stream);
