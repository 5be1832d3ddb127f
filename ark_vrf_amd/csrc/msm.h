// msm.h -- Pippenger bucket MSM over twisted-Edwards curves for gfx950: kernel declarations
// and the host-side engine.  Takes over `VariableBaseMSM::msm_unchecked` at
// src/thin.rs:319, src/pedersen.rs:420, src/utils/common.rs:410-411.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "host_te.h"

namespace avrf {

struct te_pre_raw { uint32_t w[24]; };   // x | y | k, Montgomery, 8 x u32 each
struct te_ext_raw { uint32_t w[32]; };   // x | y | t | z

// thrown by the device engines on a failed HIP call; every extern "C" entry point catches it and returns AVRF_ERR_NO_DEVICE
struct HipFailure { hipError_t err; const char *file; int line; };

struct MsmPlan {
  int c;         // window bits
  int nwin;      // number of windows: nwin * c >= scalar bits + 1
  int nb;        // buckets per window = 2^(c-1) (signed digits)
  int lpb;       // entries per lane of the last k_accumulate launch (read back from the device plan)
};
MsmPlan msm_plan(size_t n, int scalar_bits);

// Device workspace owned by a context; grows on demand, never shrinks.
struct MsmWorkspace {
  uint16_t *keys = nullptr;      // nwin * n      bucket | sign << 15
  uint32_t *sorted = nullptr;    // nwin * n      term index | sign << 31, grouped by (window, bucket)
  uint32_t *hist = nullptr;      // nwin * ntiles * nb   per-tile histograms -> per-tile prefixes
  uint32_t *cnts = nullptr;      // nwin * nb     entries per bucket
  uint32_t *offsets = nullptr;   // nwin * nb     first entry of the bucket in sorted[]
  uint32_t *win_tot = nullptr;   // nwin          entries per window
  uint32_t *lane_base = nullptr; // nwin + 1      first k_accumulate lane of every window
  uint32_t *heavy = nullptr;     // nwin * nb     buckets fed by many lanes (summed by k_heavy_sum)
  uint32_t *plan_dev = nullptr;  // {entries per lane, lanes used, heavy buckets}
  uint32_t *plan_host = nullptr; // pinned copy of plan_dev
  // accumulator arrays (layout of the curve policy: te_ext 128 B, G1 XYZZ 4 * Fq)
  uint32_t *buckets = nullptr;   // nwin * nb
  uint32_t *rc = nullptr;        // nwin * (rows + cols) partial sums of the bucket reduction
  uint32_t *part = nullptr;      // lanes + buckets slots: partial sum of lane t for bucket g at slot t + g
  uint32_t *bits = nullptr;      // nwin * c
  uint32_t *bits_host = nullptr; // pinned
  size_t cap_n = 0, cap_slots = 0, cap_buckets = 0, cap_bits = 0, cap_part = 0, cap_hist = 0, cap_vwin = 0;   // cap_buckets.. in bytes
  // HIP events bracketing the dominant kernel (k_accumulate) on the launch stream
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double accum_ms_total = 0; uint64_t accum_launches = 0; float accum_ms_last = 0;
  MsmPlan last_plan = {0, 0, 0, 0};
  int wsum_lg = 4;               // scale 2^lg of the third point of a window triple (TE window sums)
  // an enqueued launch chain whose results have not been collected yet (msm_te_enqueue / msm_te_finish)
  MsmPlan pending_plan = {0, 0, 0, 0}; int pending_ret = 0; size_t pending_n = 0; bool pending_armed = false;
  void ensure(size_t n, const MsmPlan &p, size_t acc_bytes, size_t batch, size_t lanes_max, size_t part_bytes = 0);   // part_bytes: one slot of `part` (0: acc_bytes)
  void release();
};

// Results of ONE enqueued twisted-Edwards chain kept outside the workspace, so that several chains can be queued behind each
// other on one lane (stream + workspace): the device-side workspace is reused in stream order, what comes back to the host
// (window sums, the plan, the timing events of the dominant kernel) lands here.  pool.hip gives every slot one.
struct MsmPending {
  uint32_t *bits_host = nullptr, *plan_host = nullptr; size_t cap_bytes = 0;   // pinned
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  MsmPlan plan = {0, 0, 0, 0}; int ret = 0, wsum_lg = 4; size_t n = 0; bool armed = false;
  void ensure(size_t bytes);
  void release();
};

// MSM of n precomputed device points (Montgomery form) with n plain 256-bit scalars (8 x u32 LE,
// already < r), on `stream`.  Result: extended point on the host.  suite: 0 Bandersnatch, 1 Baby-JubJub.
int msm_te_device(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n,
                  MsmWorkspace &ws, hipStream_t stream, HostExt *out);
// the same in two halves: msm_te_enqueue launches the whole kernel chain and the copies back on `stream` and returns without
// waiting; msm_te_finish waits for the stream and folds the window sums on the host.  One chain in flight per workspace.
// pend != nullptr: the chain's results go to *pend (see MsmPending); msm_te_finish then does NOT wait for the stream -- the
// caller has seen an event recorded behind the chain complete.
int msm_te_enqueue(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, MsmPending *pend = nullptr);
int msm_te_finish(int suite, MsmWorkspace &ws, hipStream_t stream, HostExt *out, MsmPending *pend = nullptr);
// two or more scalar vectors over the SAME n <= 2048 bases (vector v's scalars start at element v * n) through the single-launch
// form: out[v] = sum_t s_{v,t} P_t.  Synchronises the stream.  Returns -1 when the shape is not the single-launch one.
int msm_te_small_vectors(int suite, const te_pre_raw *d_pre, const uint32_t *d_scalars, size_t n, size_t n_vectors, MsmWorkspace &ws, hipStream_t stream,
                         HostExt *out);
bool msm_te_pending_supported(int suite);   // external results exist for the window-sum form of the chain only

// G1 MSM over a short-Weierstrass curve (curve: 0 BLS12-381, 1 BN254): d_bases = n Montgomery affine
// points (2 * Fq words each, (0,0) = infinity), d_scalars = n plain 256-bit scalars (< r).
// out_xy: canonical affine x || y little-endian (2 * FQ_BYTES), all-zero for the point at infinity.
// batch > 1: `batch` scalar vectors over the same bases (vector b starts at element b * scalar_stride; 0 = n),
// `batch` results in out_xy.
int msm_g1_device(int curve, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n, MsmWorkspace &ws, hipStream_t stream, uint8_t *out_xy,
                  size_t batch = 1, size_t scalar_stride = 0);
// Fixed-base form for MSMs over one shared base set (the KZG SRS): build_g1_table fills
// table[w * n + i] = 2^(c w) * P_i (nwin = ceil((Fr bits + 1) / c) rows); msm_g1_fixed_device then treats all
// windows of a scalar vector as ONE bucket set (n may be smaller than the table's row length `table_stride`).
// d_base_idx != nullptr: sparse form -- entry i of vector b multiplies table base d_base_idx[b * n + i] instead of base i.
void build_g1_table(int curve, const uint32_t *d_bases, size_t n, int c, int nwin, uint32_t *d_table, hipStream_t stream);
int msm_g1_fixed_device(int curve, const uint32_t *d_table, int table_c, size_t table_stride, const uint32_t *d_scalars, size_t n,
                        size_t scalar_stride, MsmWorkspace &ws, hipStream_t stream, uint8_t *out_xy, size_t batch,
                        const uint32_t *d_base_idx = nullptr, int scalars_mont = 0);
// (scalars_mont: the scalar vectors hold Montgomery limbs -- 1 = BLS12-381 Fr, 2 = BN254 Fr -- and the digit kernel converts on load)
// Fixed-base form over a table of ALL multiples (msm.hip "fixed-base MSM over a table of ALL multiples"): for HBM-rich parts.  The table of
// a KZG SRS is built once per (device, SRS) and shared; a commitment is one gathered point per (coefficient, row), no buckets.
struct G1DirectTable {
  uint32_t *d = nullptr;            // affine Montgomery points, 2 * Fq words each; (w, m, i) = (m + 1) 2^(c w) P_i at off[w] + m * n + i
  int curve = 0, c = 0, rows = 0;   // rows: digit rows + the carry row where the recoding can carry out of the top one
  size_t n = 0;                     // bases per row
  uint64_t off[32] = {};            // first point of row w
  uint32_t mult[32] = {};           // multiples held for row w (1 .. mult[w])
  uint64_t points = 0, bytes = 0;
};
size_t g1_direct_table_shape(int curve, size_t n, int c, G1DirectTable *t);      // fills *t (no allocation), returns the table's bytes
void build_g1_direct_table(int curve, const uint32_t *d_bases, size_t n, int c, G1DirectTable *t, hipStream_t stream);   // allocates t->d, synchronises
void free_g1_direct_table(G1DirectTable *t);
// `batch` vectors of n Montgomery-or-plain scalars (vector b starts at element b * scalar_stride) -> `batch` affine results in out_xy
// (d_base_idx != nullptr: sparse form, as msm_g1_fixed_device's)
int msm_g1_direct_device(const G1DirectTable &t, const uint32_t *d_scalars, size_t n, size_t scalar_stride, MsmWorkspace &ws, hipStream_t stream,
                         uint8_t *out_xy, size_t batch, int scalars_mont = 0, const uint32_t *d_base_idx = nullptr);
// n compressed G1 points (FQ_BYTES each, ark-serialize) -> canonical x || y little-endian + ok[i] (0 invalid, 1 point, 2 infinity)
void launch_g1_decompress(int curve, const uint8_t *d_comp, size_t n, uint8_t *d_out_xy, uint8_t *d_ok, hipStream_t stream);
void launch_g1_bases(int curve, const uint8_t *d_xy, size_t n, uint32_t *d_out, uint32_t *d_flag, hipStream_t stream);
// sets bit 1 of *d_flag when one of the n Montgomery affine bases (as launch_g1_bases writes them) is outside the prime-order
// subgroup (BLS12-381: endomorphism test; BN254: cofactor 1, no-op)
// (d_rec_status != nullptr: record i / ppr of a failing point gets status 2)
void launch_g1_subgroup_check(int curve, const uint32_t *d_bases, size_t n, uint32_t *d_flag, hipStream_t stream, int32_t *d_rec_status = nullptr,
                              uint32_t ppr = 1);
// per item (16-lane group): point 0 = sum_{t < split} s_t P_t, point 1 = sum_{split <= t < tpi} s_t P_t over the item's tpi <= 16
// (base, scalar) pairs; out: n_items x 2 Montgomery affine points.  glv_split_scalars (BLS12-381 only): every 32-byte scalar slot
// holds k mod z^2 | (k div z^2) << 128 (g1_glv_split_bls) and the kernel takes the 128-doubling GLV chain.
void launch_g1_lincomb(int curve, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n_items, uint32_t tpi, uint32_t split, uint32_t *d_out,
                       hipStream_t stream, bool glv_split_scalars = false);
// k (256-bit little-endian limbs, < 2^255) -> k mod z^2 in limbs 0..1, k div z^2 in limbs 2..3, z = 0xd201000000010000 (host)
void g1_glv_split_bls(uint64_t k[4]);

// canonical affine bytes (x||y LE32) -> te_pre (device); flags[i] |= 1 if a coordinate >= q, |= 2 if off-curve (when check_curve)
// (mont_in: the coordinates are Montgomery limbs already -- avrf_msm_te_mont)
void launch_pre_from_affine(int suite, const uint8_t *d_xy, size_t n, te_pre_raw *d_pre, uint32_t *d_flag,
                            int check_curve, hipStream_t stream, int mont_in = 0);
// n scalars as Montgomery limbs of the suite's Fr -> plain 256-bit integers, in place; *d_flag |= 4 for a limb value >= r
void launch_scalars_from_mont(int suite, uint32_t *d_scalars, size_t n, uint32_t *d_flag, hipStream_t stream);

}  // namespace avrf
