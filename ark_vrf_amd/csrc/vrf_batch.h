// vrf_batch.h -- device-side batch preparation for the Thin / Pedersen batch verifiers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "msm.h"

namespace avrf {

struct BatchDev;
struct Seed64;

// thin::BatchVerifier::prepare (src/thin.rs:209-226) for every item: c_j (4 x u32 LE) and
// z_{j,i} for i >= 1 (4 x u32 each, indexed by io_off[j] + i - 1); plus the identity / range
// checks of src/thin.rs:266-271 into *flags.
void launch_thin_prepare(int suite, const BatchDev &b, uint32_t *d_c, uint32_t *d_z, uint32_t *d_flags, hipStream_t st);

// Builds the MSM of src/thin.rs:282-317: scalars (n_terms x 8 u32, plain) and precomputed bases,
// given the weight-transcript seed.  n_terms = 2 n + 2 tot_io + 1.
void launch_thin_terms(int suite, const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint32_t *d_z,
                       uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st);

// pedersen::BatchItem::new (src/pedersen.rs:276-293) and the (5N+2)-term MSM of :369-418.
void launch_ped_prepare(int suite, const BatchDev &b, uint32_t *d_c, uint8_t *d_merged, uint32_t *d_flags, hipStream_t st);
void launch_ped_terms(int suite, const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint8_t *d_merged,
                      uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st);

// independent per-item kernels (vrf_single.hip)
// fixed-base tables of G and BLINDING_BASE (2 x 32 x 256 te_pre = 1.5 MB), see proto_dev.h te_smul_fixed
void launch_fixed_table(int suite, te_pre_raw *d_tab, hipStream_t st);
void launch_smul(int suite, const uint8_t *d_scalars, const uint8_t *d_points_xy, uint32_t n, uint8_t *d_out, uint32_t *d_flags,
                 const te_pre_raw *d_fixed, hipStream_t st);
void launch_thin_prove(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny = false);
// one proof split around the MSM engine (vrf_single.hip k_thin_prove_begin / _end): terms of R = k G + sum (k z_i) I_i, then the
// challenge and the response from the normalised R; d_state holds the nonce and the transcript between the two
size_t thin_prove_state_bytes(int suite);
void launch_thin_prove_begin(int suite, const BatchDev &b, uint32_t *d_scalars, struct te_pre_raw *d_pre, uint8_t *d_state, hipStream_t st, bool tiny);
void launch_thin_prove_end(int suite, const BatchDev &b, const uint8_t *d_state, const uint8_t *d_rxy, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny);
// one Pedersen proof the same way (k_ped_prove_begin / _mid / _end): Yb, then R and Ok as two scalar vectors over {G, B, I_0 ..}
size_t ped_prove_state_bytes(int suite);
void launch_ped_prove_begin(int suite, const BatchDev &b, uint32_t *d_scalars, struct te_pre_raw *d_pre, uint8_t *d_state, uint32_t *d_wts, hipStream_t st);
void launch_ped_prove_mid(int suite, const BatchDev &b, uint32_t *d_scalars, struct te_pre_raw *d_pre, uint8_t *d_state, const uint32_t *d_wts, const uint8_t *d_yb, hipStream_t st);
void launch_ped_prove_end(int suite, const BatchDev &b, const uint8_t *d_state, const uint8_t *d_pts, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st);
void launch_tiny_verify(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st);   // proofs: n x 48 (c16 || s32)
void launch_thin_verify(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st);
// few items, ONE I/O pair each, twisted-Edwards suites: an item spread over 32 lanes (vrf_single.hip "few items"); false = the
// suite has no such kernel (nothing was launched).  d_status[j] = AVRF_WAVE_FALLBACK for an item with a degenerate point: the
// caller re-runs the call on the lane-per-item kernel.
enum { AVRF_WAVE_ITEMS_MAX = 2048, AVRF_WAVE_FALLBACK = -99 };
bool launch_thin_verify_wave(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st);
bool launch_thin_prove_wave(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, int32_t *d_status, hipStream_t st, bool tiny = false);
bool launch_tiny_verify_wave(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st);
bool launch_ped_verify_wave(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st);     // one item per wave (two equations)
bool launch_ped_prove_wave(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, int32_t *d_status, hipStream_t st);
void launch_ped_prove(int suite, const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st);
void launch_ped_verify(int suite, const BatchDev &b, int32_t *d_status, hipStream_t st);

void launch_hash_to_curve(int suite, const uint8_t *d_data, const uint32_t *d_off, uint32_t n, uint8_t *d_out, int32_t *d_status, hipStream_t st);
void launch_decompress_strided(int suite, const uint8_t *d_in, uint32_t in_stride, uint32_t n, uint8_t *d_out, uint32_t out_stride, int validate, uint32_t *d_flags, hipStream_t st);
void launch_output_hash(int suite, const uint8_t *d_in_xy, uint32_t n, uint32_t len, uint8_t *d_out, hipStream_t st);     // Output::hash, src/lib.rs:605-609
void launch_secret_from_seed(int suite, const uint8_t *d_seeds, uint32_t n, uint8_t *d_sk, uint32_t *d_flags, hipStream_t st);   // Secret::from_seed, src/lib.rs:346-369
void launch_decompress(int suite, const uint8_t *d_in, uint32_t n, uint8_t *d_out, int validate, int32_t *d_status, hipStream_t st);
void launch_compress(int suite, const uint8_t *d_in, uint32_t n, uint8_t *d_out, hipStream_t st);
// Validate::Yes on xy points laid out as records (see k_validate_xy): level 1 on-curve, 2 + prime-order subgroup
// d_item_off (device, n_items + 1 exclusive prefix sums): the records are I/O pairs of items with different pair counts and
// d_rec_status is indexed by ITEM -- each lane finds the item of its pair by binary search (one launch whatever the counts)
void launch_validate_xy(int suite, const uint8_t *d_base, uint32_t stride, uint32_t ppr, uint32_t nrec, int level, uint32_t *d_flags,
                        int32_t *d_rec_status, hipStream_t st, const uint32_t *d_item_off = nullptr, uint32_t n_items = 0);

// Per-suite launch tables.  vrf_batch.hip / vrf_single.hip are compiled once per suite (-DAVRF_TU_SUITE=<id>), each such unit
// holding the kernels of one suite and the explicit instantiation of these two structs for it; the unit compiled without
// the macro holds only the launch_* dispatchers above, which resolve to the instantiations at link time (csrc/Makefile).
template <class S> struct BatchOps {
  static void thin_prepare(const BatchDev &b, uint32_t *d_c, uint32_t *d_z, uint32_t *d_flags, hipStream_t st);
  static void thin_terms(const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint32_t *d_z,
                         uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st);
  static void ped_prepare(const BatchDev &b, uint32_t *d_c, uint8_t *d_merged, uint32_t *d_flags, hipStream_t st);
  static void ped_terms(const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint8_t *d_merged,
                        uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st);
};
template <class S> struct SingleOps {
  static void fixed_table(te_pre_raw *d_tab, hipStream_t st);
  static void smul(const uint8_t *d_scalars, const uint8_t *d_points_xy, uint32_t n, uint8_t *d_out, uint32_t *d_flags,
                   const te_pre_raw *d_fixed, hipStream_t st);
  static void thin_prove(const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny);
  static size_t prove_state_bytes();
  static void thin_prove_begin(const BatchDev &b, uint32_t *d_scalars, te_pre_raw *d_pre, uint8_t *d_state, hipStream_t st, bool tiny);
  static void thin_prove_end(const BatchDev &b, const uint8_t *d_state, const uint8_t *d_rxy, uint8_t *d_proofs_out, uint32_t *d_flags, hipStream_t st, bool tiny);
  static size_t ped_state_bytes();
  static void ped_prove_begin(const BatchDev &b, uint32_t *d_scalars, te_pre_raw *d_pre, uint8_t *d_state, uint32_t *d_wts, hipStream_t st);
  static void ped_prove_mid(const BatchDev &b, uint32_t *d_scalars, te_pre_raw *d_pre, uint8_t *d_state, const uint32_t *d_wts, const uint8_t *d_yb, hipStream_t st);
  static void ped_prove_end(const BatchDev &b, const uint8_t *d_state, const uint8_t *d_pts, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st);
  static void tiny_verify(const BatchDev &b, int32_t *d_status, hipStream_t st);
  static void thin_verify(const BatchDev &b, int32_t *d_status, hipStream_t st);
  static bool thin_verify_wave(const BatchDev &b, int32_t *d_status, hipStream_t st);
  static bool thin_prove_wave(const BatchDev &b, uint8_t *d_proofs_out, uint32_t *d_flags, int32_t *d_status, hipStream_t st, bool tiny);
  static bool tiny_verify_wave(const BatchDev &b, int32_t *d_status, hipStream_t st);
  static bool ped_verify_wave(const BatchDev &b, int32_t *d_status, hipStream_t st);
  static bool ped_prove_wave(const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, int32_t *d_status, hipStream_t st);
  static void ped_prove(const BatchDev &b, uint8_t *d_proofs_out, uint8_t *d_blind, uint32_t *d_flags, hipStream_t st);
  static void ped_verify(const BatchDev &b, int32_t *d_status, hipStream_t st);
  static void hash_to_curve(const uint8_t *d_data, const uint32_t *d_off, uint32_t n, uint8_t *d_out, int32_t *d_status, hipStream_t st);
  static void decompress(const uint8_t *d_in, uint32_t n, uint8_t *d_out, int validate, int32_t *d_status, hipStream_t st);
  static void validate_xy(const uint8_t *d_base, uint32_t stride, uint32_t ppr, uint32_t nrec, int level, uint32_t *d_flags,
                          int32_t *d_rec_status, hipStream_t st, const uint32_t *d_item_off, uint32_t n_items);
  static void compress(const uint8_t *d_in, uint32_t n, uint8_t *d_out, hipStream_t st);
  static void decompress_strided(const uint8_t *d_in, uint32_t in_stride, uint32_t n, uint8_t *d_out, uint32_t out_stride, int validate, uint32_t *d_flags, hipStream_t st);
  static void output_hash(const uint8_t *d_in, uint32_t n, uint32_t len, uint8_t *d_out, hipStream_t st);
  static void secret_from_seed(const uint8_t *d_seeds, uint32_t n, uint8_t *d_sk, uint32_t *d_flags, hipStream_t st);
};

}  // namespace avrf
