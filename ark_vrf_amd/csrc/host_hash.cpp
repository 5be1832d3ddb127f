// host_hash.cpp -- the multi-buffer SHA-512 of host_sha512_mb.h as a unit of its own, compiled by g++ (see the header for why).
#define AVRF_SHA_MB_IMPL
#include "host_sha512_mb.h"

namespace avrf {
AVRF_MB_TARGET void sha512_weights_x8(WeightJob *const *jobs, int count) { sha512_weights_x8_impl(jobs, count); }
AVRF_MB16_TARGET void sha512_weights_x16(WeightJob *const *jobs, int count) { sha512_weights_x16_impl(jobs, count); }
}  // namespace avrf
