// mispec_internal.h -- what the translation units of libmispec.so share besides include/mispec.h.
// Hidden visibility: none of this is part of the C ABI.
#pragma once
#include <cstddef>

#define MISPEC_HIDDEN __attribute__((visibility("hidden")))

// records the message mispec_last_error() returns on this thread, returns `code`
MISPEC_HIDDEN int mispec_fail_msg(int code, const char *msg);
// compute units of the current device (cached per device)
MISPEC_HIDDEN int mispec_device_cus();
