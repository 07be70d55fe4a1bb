// mispec_internal.h -- what the translation units of libmispec.so share besides include/mispec.h.
// Hidden visibility: none of this is part of the C ABI.
#pragma once
#include <cstddef>

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
