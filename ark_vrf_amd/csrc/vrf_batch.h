// vrf_batch.h -- device-side batch preparation for the Thin / Pedersen batch verifiers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "msm.h"

namespace avrf {

// A staged batch in HBM.  Offsets are exclusive prefix sums (n + 1 entries).
struct BatchDev {
  const uint8_t *pks_xy;    // n x 64 (thin only)
  const uint8_t *ios_xy;    // tot_io x 128 (input_xy || output_xy)
  const uint32_t *io_off;   // n + 1
  const uint8_t *ads;       // concatenated additional data
  const uint32_t *ad_off;   // n + 1
  const uint8_t *proofs;    // thin: n x 96 (R_xy || s); pedersen: n x 256
  uint32_t n;
};

struct Seed64 { uint64_t w[8]; };  // a SHA-512 digest as big-endian words

enum { FLAG_RANGE = 1, FLAG_IDENTITY = 2, FLAG_SCALAR = 4 };

// thin::BatchVerifier::prepare (src/thin.rs:209-226) for every item: c_j (4 x u32 LE) and
// z_{j,i} for i >= 1 (4 x u32 each, indexed by io_off[j] + i - 1); plus the identity / range
// checks of src/thin.rs:266-271 into *flags.
void launch_thin_prepare(int suite, const BatchDev &b, uint32_t *d_c, uint32_t *d_z, uint32_t *d_flags, hipStream_t st);

// Builds the MSM of src/thin.rs:282-317: scalars (n_terms x 8 u32, plain) and precomputed bases,
// given the weight-transcript seed.  n_terms = 2 n + 2 tot_io + 1.
void launch_thin_terms(int suite, const BatchDev &b, const Seed64 &seed, const uint32_t *d_c, const uint32_t *d_z,
                       uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st);

}  // namespace avrf
