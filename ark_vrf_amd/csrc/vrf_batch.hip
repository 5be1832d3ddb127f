// vrf_batch.hip -- per-item Fiat-Shamir hashing and MSM term construction on the device.
//
//   k_thin_prepare : thin::BatchVerifier::prepare, src/thin.rs:209-226
//                    (vrf_transcript_scalars_with_schnorr src/utils/common.rs:251-258,
//                     vrf_transcript_base :159-173, DelinearizeScalars :335-369, challenge :270-280)
//   k_thin_terms   : the per-item part of thin::BatchVerifier::verify, src/thin.rs:287-313
//   k_g_final      : the shared-generator term, src/thin.rs:303,316-317
// One lane per batch item; SHA-512 state in VGPRs (sha512_dev.h); scalar-field products in
// 8 x u32 Montgomery form (fp256.h).
#include "vrf_batch.h"
#include "proto_dev.h"
#include "suite_dispatch.h"

namespace avrf {
#ifdef AVRF_TU_SUITE      // per-suite unit: the kernels of one suite + BatchOps<S> (vrf_batch.h)

template <class S>
__global__ void __launch_bounds__(128)
k_thin_prepare(BatchDev b, uint32_t *__restrict__ c_out, uint32_t *__restrict__ z_out, uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n) return;
  uint32_t io0 = b.io_off[j], io1 = b.io_off[j + 1], m = io1 - io0;
  uint32_t ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  uint32_t f = 0;
  suite_tr<S> h; tr_init(h);
  for (int i = 0; i < S::SUITE_ID_LEN; i++) tr_byte(h, S::SUITE_ID[i]);       // Transcript::new(SUITE_ID)
  tr_byte(h, DS_THIN);                                                         // common.rs:166
  tr_u64le(h, (uint64_t)m + 1);                                                // absorb_ios :377-383, Schnorr pair first
  absorb_generator<S>(h);                                                    // chain_ios :231-240: (G, pk)
  // short-Weierstrass presentation, one pair (the common shape): the 33-byte forms of pk, I, O and R with ONE inversion
  sw_enc enc[4]; bool have_enc = false;
  if constexpr (S::SW_CODEC) if (m == 1) {
    const uint8_t *p = b.ios_xy + 128 * (size_t)io0, *pr = b.proofs + 96 * (size_t)j, *pk = b.pks_xy + 64 * (size_t)j;
    const fp xs[4] = {fp_load_le(pk), fp_load_le(p), fp_load_le(p + 64), fp_load_le(pr)};
    const fp ys[4] = {fp_load_le(pk + 32), fp_load_le(p + 32), fp_load_le(p + 96), fp_load_le(pr + 32)};
    sw_encode_te_many<S, 4>(xs, ys, enc); have_enc = true;
  }
  {
    fp x = fp_load_le(b.pks_xy + 64 * (size_t)j), y = fp_load_le(b.pks_xy + 64 * (size_t)j + 32);
    f |= point_flags<S>(x, y);                                                 // thin.rs:266-271
    if (have_enc) absorb_sw_enc(h, enc[0]); else absorb_point_xy<S>(h, x, y);
  }
  for (uint32_t i = 0; i < m; i++) {
    const uint8_t *p = b.ios_xy + 128 * (size_t)(io0 + i);
    fp x = fp_load_le(p), y = fp_load_le(p + 32);
    f |= point_flags<S>(x, y); if (have_enc) absorb_sw_enc(h, enc[1]); else absorb_point_xy<S>(h, x, y);
    x = fp_load_le(p + 64); y = fp_load_le(p + 96);
    f |= point_flags<S>(x, y); if (have_enc) absorb_sw_enc(h, enc[2]); else absorb_point_xy<S>(h, x, y);
  }
  tr_u64le(h, (uint64_t)adl);                                                  // common.rs:169-170
  tr_bytes(h, b.ads + ad0, adl);
  if (m) {                                                                     // DelinearizeScalars :345-363
    auto rd = delin_seed(h);
    for (uint32_t i = 0; i < m; i++) {
      uint32_t w[4]; rd_chunk16(rd, i, w);
      uint4 *o = reinterpret_cast<uint4 *>(z_out + 4 * (size_t)(io0 + i));
      *o = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
  {                                                                            // challenge :270-280
    const uint8_t *pr = b.proofs + 96 * (size_t)j;
    fp x = fp_load_le(pr), y = fp_load_le(pr + 32), s = fp_load_le(pr + 64);
    if (point_flags<S>(x, y) & FLAG_RANGE) f |= FLAG_RANGE;                    // R may be the identity (thin.rs:95-99)
    if (ge_p<Fr>(s)) f |= FLAG_SCALAR;
    tr_byte(h, DS_CHALLENGE);
    if (have_enc) absorb_sw_enc(h, enc[3]); else absorb_point_xy<S>(h, x, y);
    auto rd = tr_reader(h);
    uint32_t w[4]; rd_chunk16(rd, 0, w);
    uint4 *o = reinterpret_cast<uint4 *>(c_out + 4 * (size_t)j);
    *o = make_uint4(w[0], w[1], w[2], w[3]);
    if (b.records) {                                                           // LE32(c) || LE32(s) of the weight transcript, thin.rs:274-279
      uint4 *r = reinterpret_cast<uint4 *>(b.records + 64 * (size_t)j);
      const uint4 *sp = reinterpret_cast<const uint4 *>(pr + 64);
      r[0] = make_uint4(w[0], w[1], w[2], w[3]); r[1] = make_uint4(0, 0, 0, 0); r[2] = sp[0]; r[3] = sp[1];
    }
  }
  if (f) atomicOr(flags, f);
}

AVRF_DI fp fp_from_u128(const uint32_t *w) {
  fp a = fp_zero(); uint4 v = *reinterpret_cast<const uint4 *>(w);
  a.v[0] = v.x; a.v[1] = v.y; a.v[2] = v.z; a.v[3] = v.w; return a;
}
template <class S> AVRF_DI void emit_term(uint32_t *scalars, te_pre *pre, uint32_t t, const fp &scalar_plain, const uint8_t *xy) {
  using Fq = typename S::Fq;
  store_fp(scalars + 8 * (size_t)t, scalar_plain);
  fp x = fp_to_mont<Fq>(fp_load_le(xy)), y = fp_to_mont<Fq>(fp_load_le(xy + 32));
  store_pre(pre + t, te_make_pre<S>(x, y));
}

template <class S>
__global__ void __launch_bounds__(128)
k_thin_terms(BatchDev b, Seed64 seed, uint64_t j0, const uint32_t *__restrict__ c_in, const uint32_t *__restrict__ z_in,
             uint32_t *__restrict__ scalars, te_pre *__restrict__ pre, uint32_t *__restrict__ gpart) {
  using Fr = typename S::Fr;
  __shared__ fp red[128];
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  fp ws = fp_zero();
  if (j < b.n) {
    uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0;
    // w_j = challenge_scalar(&mut t): 16 bytes of the weight stream (thin.rs:289, common.rs:72-76)
    // (j0 = global index of this shard's first item when one batch is split over several GPUs)
    uint64_t gj = j0 + j;
    uint32_t ww[4];
    if (b.weights) { const uint4 v = *reinterpret_cast<const uint4 *>(b.weights + 16 * (size_t)j); ww[0] = v.x; ww[1] = v.y; ww[2] = v.z; ww[3] = v.w; }
    else { uint64_t blk[8]; sha512_xof_block(seed.w, gj >> 2, blk); digest_le128(blk, (int)(gj & 3), ww); }
    fp w_plain = fp_zero(); w_plain.v[0] = ww[0]; w_plain.v[1] = ww[1]; w_plain.v[2] = ww[2]; w_plain.v[3] = ww[3];
    fp w = fp_to_mont<Fr>(w_plain);
    fp c = fp_to_mont<Fr>(fp_from_u128(c_in + 4 * (size_t)j));
    const uint8_t *pr = b.proofs + 96 * (size_t)j;
    fp s = fp_to_mont<Fr>(fp_load_le(pr + 64));
    fp wc = fp_mul<Fr>(w, c);                               // thin.rs:291-292
    ws = fp_mul<Fr>(w, s);
    uint32_t t = 2 * j + 2 * io0;
    emit_term<S>(scalars, pre, t, w_plain, pr);                                           // (R_j, w_j)        :295-296
    emit_term<S>(scalars, pre, t + 1, fp_from_mont<Fr>(wc), b.pks_xy + 64 * (size_t)j);   // (pk_j, w c z0)    :299-300, z0 = 1
    for (uint32_t i = 0; i < m; i++) {                                                    // :306-312
      fp z = fp_to_mont<Fr>(fp_from_u128(z_in + 4 * (size_t)(io0 + i)));
      const uint8_t *p = b.ios_xy + 128 * (size_t)(io0 + i);
      emit_term<S>(scalars, pre, t + 2 + 2 * i, fp_from_mont<Fr>(fp_mul<Fr>(wc, z)), p + 64);             // (O_i, w c z_i)
      emit_term<S>(scalars, pre, t + 3 + 2 * i, fp_from_mont<Fr>(fp_neg<Fr>(fp_mul<Fr>(ws, z))), p);      // (I_i, -w s z_i)
    }
  }
  // block sum of w_j s_j z0 (Montgomery form), thin.rs:303
  red[threadIdx.x] = ws;
  __syncthreads();
  for (int s2 = 64; s2 >= 1; s2 >>= 1) {
    if ((int)threadIdx.x < s2) red[threadIdx.x] = fp_add<Fr>(red[threadIdx.x], red[threadIdx.x + s2]);
    __syncthreads();
  }
  if (threadIdx.x == 0) store_fp(gpart + 8 * (size_t)blockIdx.x, red[0]);
}

// g = -(sum of partials); last term (G, g)   (thin.rs:303,316-317)
template <class S>
__global__ void __launch_bounds__(256)
k_g_final(const uint32_t *__restrict__ gpart, uint32_t nparts, uint32_t *__restrict__ scalars, te_pre *__restrict__ pre,
          uint32_t t_last, int which_base) {
  using Fr = typename S::Fr; using Fq = typename S::Fq;
  __shared__ fp red[256];
  fp acc = fp_zero();
  for (uint32_t i = threadIdx.x; i < nparts; i += 256) acc = fp_add<Fr>(acc, load_fp(gpart + 8 * (size_t)i));
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s2 = 128; s2 >= 1; s2 >>= 1) {
    if ((int)threadIdx.x < s2) red[threadIdx.x] = fp_add<Fr>(red[threadIdx.x], red[threadIdx.x + s2]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    store_fp(scalars + 8 * (size_t)t_last, fp_from_mont<Fr>(fp_neg<Fr>(red[0])));
    te_pre g;
    if (which_base == 0) { g.x = fp_const<Fq>(S::G_X); g.y = fp_const<Fq>(S::G_Y); g.k = fp_const<Fq>(S::G_K); }
    else { g.x = fp_const<Fq>(S::B_X); g.y = fp_const<Fq>(S::B_Y); g.k = fp_const<Fq>(S::B_K); }
    store_pre(pre + t_last, g);
  }
}


// ---------------------------------------------------------------- Pedersen batch

// pedersen::BatchItem::new (src/pedersen.rs:276-293): challenge c_j and the merged I/O pair
// (written as canonical xy to merged_xy[j], 128 bytes) for every item.
template <class S>
__global__ void __launch_bounds__(128)
k_ped_prepare(BatchDev b, uint32_t *__restrict__ c_out, uint8_t *__restrict__ merged_xy, uint32_t *__restrict__ flags) {
  using Fr = typename S::Fr;
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n) return;
  uint32_t io0 = b.io_off[j], m = b.io_off[j + 1] - io0, ad0 = b.ad_off[j], adl = b.ad_off[j + 1] - ad0;
  const uint8_t *ios = b.ios_xy + 128 * (size_t)io0, *pr = b.proofs + 256 * (size_t)j;
  suite_tr<S> t; uint32_t f = 0;
  tr_base<S>(t, DS_PEDERSEN, false, nullptr, ios, m, b.ads + ad0, adl, &f);   // io identity -> io_identity flag (:278)
  uint8_t *mo = merged_xy + 128 * (size_t)j;
  if (m == 1) {
    const uint4 *src = reinterpret_cast<const uint4 *>(ios); uint4 *dst = reinterpret_cast<uint4 *>(mo);
#pragma unroll
    for (int i = 0; i < 8; i++) dst[i] = src[i];
  } else if (m == 0) {
    fp zero = fp_zero(), one = fp_zero(); one.v[0] = S::SW_NATIVE ? 0 : 1;      // identity: (0, 1), or (0, 0) for a short-Weierstrass suite
    fp_store_le(mo, zero); fp_store_le(mo + 32, one); fp_store_le(mo + 64, zero); fp_store_le(mo + 96, one);
  } else {                                                                     // merge_ios, common.rs:389-419
    auto dseed = delin_seed(t);
    te_ext im = te_identity<S>(), om = te_identity<S>();
    for (uint32_t i = 0; i < m; i++) {
      te_pre pi = pre_from_xy<S>(ios + 128 * (size_t)i), po = pre_from_xy<S>(ios + 128 * (size_t)i + 64);
      if (i == 0) { im = te_madd<S>(im, pi); om = te_madd<S>(om, po); }
      else { fp z = xof128(dseed, i - 1); im = te_add<S>(im, te_smul<S>(pi, z, 128)); om = te_add<S>(om, te_smul<S>(po, z, 128)); }
    }
    te_aff ia = te_to_aff<S>(im), oa = te_to_aff<S>(om);
    store_xy<S>(mo, ia); store_xy<S>(mo + 64, oa);
  }
  fp ybx = fp_load_le(pr), yby = fp_load_le(pr + 32), rx = fp_load_le(pr + 64), ry = fp_load_le(pr + 96);
  fp okx = fp_load_le(pr + 128), oky = fp_load_le(pr + 160);
  f |= point_flags<S>(ybx, yby);                                               // pk_com.is_zero() (:348-353)
  f |= (point_flags<S>(rx, ry) | point_flags<S>(okx, oky)) & FLAG_RANGE;
  if (ge_p<Fr>(fp_load_le(pr + 192)) || ge_p<Fr>(fp_load_le(pr + 224))) f |= FLAG_SCALAR;
  if constexpr (S::SW_CODEC) {                                                 // the three proof points' 33-byte forms, one inversion
    const fp xs[3] = {ybx, rx, okx}, ys[3] = {yby, ry, oky};
    sw_enc enc[3]; sw_encode_te_many<S, 3>(xs, ys, enc);
    absorb_sw_enc(t, enc[0]); tr_byte(t, DS_CHALLENGE); absorb_sw_enc(t, enc[1]); absorb_sw_enc(t, enc[2]);
  } else {
    absorb_point_xy<S>(t, ybx, yby);                                           // :280
    tr_byte(t, DS_CHALLENGE); absorb_point_xy<S>(t, rx, ry); absorb_point_xy<S>(t, okx, oky);
  }
  fp c = challenge_finish(t);                                                  // :281
  *reinterpret_cast<uint4 *>(c_out + 4 * (size_t)j) = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]);
  if (b.records) {                                                             // LE32(c) || LE32(s) || LE32(sb), pedersen.rs:361-367
    uint4 *r = reinterpret_cast<uint4 *>(b.records + 96 * (size_t)j);
    const uint4 *sp = reinterpret_cast<const uint4 *>(pr + 192);
    r[0] = make_uint4(c.v[0], c.v[1], c.v[2], c.v[3]); r[1] = make_uint4(0, 0, 0, 0); r[2] = sp[0]; r[3] = sp[1]; r[4] = sp[2]; r[5] = sp[3];
  }
  if (f) atomicOr(flags, f);
}

// per-item part of pedersen::BatchVerifier::verify, src/pedersen.rs:369-410
template <class S>
__global__ void __launch_bounds__(128)
k_ped_terms(BatchDev b, Seed64 seed, uint64_t j0, const uint32_t *__restrict__ c_in, const uint8_t *__restrict__ merged_xy,
            uint32_t *__restrict__ scalars, te_pre *__restrict__ pre, uint32_t *__restrict__ gpart, uint32_t *__restrict__ bpart) {
  using Fr = typename S::Fr;
  __shared__ fp red[128];
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  fp us = fp_zero(), usb = fp_zero();
  if (j < b.n) {
    // 32 squeezed bytes per item: t = bytes[0..16], u = bytes[16..32]   (:373-381)
    // (j0: global index of the shard's first item when one batch is split over several GPUs)
    fp t_plain, u_plain;
    if (b.weights) { t_plain = fp_from_u128(reinterpret_cast<const uint32_t *>(b.weights + 32 * (size_t)j)); u_plain = fp_from_u128(reinterpret_cast<const uint32_t *>(b.weights + 32 * (size_t)j + 16)); }
    else {
      ShaReader rd; for (int i = 0; i < 8; i++) rd.seed[i] = seed.w[i]; rd.have = 0;
      t_plain = xof128(rd, (uint32_t)(2 * (j0 + j))); u_plain = xof128(rd, (uint32_t)(2 * (j0 + j) + 1));
    }
    fp tt = fp_to_mont<Fr>(t_plain), uu = fp_to_mont<Fr>(u_plain);
    fp c = fp_to_mont<Fr>(fp_from_u128(c_in + 4 * (size_t)j));
    const uint8_t *pr = b.proofs + 256 * (size_t)j, *mo = merged_xy + 128 * (size_t)j;
    fp s = fp_to_mont<Fr>(fp_load_le(pr + 192)), sb = fp_to_mont<Fr>(fp_load_le(pr + 224));
    uint32_t k = 5 * j;
    emit_term<S>(scalars, pre, k, fp_from_mont<Fr>(fp_mul<Fr>(tt, c)), mo + 64);                    // (O, t c)      :388-389
    emit_term<S>(scalars, pre, k + 1, t_plain, pr + 128);                                            // (Ok, t)       :391-392
    emit_term<S>(scalars, pre, k + 2, fp_from_mont<Fr>(fp_neg<Fr>(fp_mul<Fr>(tt, s))), mo);         // (I, -t s)     :394-395
    emit_term<S>(scalars, pre, k + 3, fp_from_mont<Fr>(fp_mul<Fr>(uu, c)), pr);                     // (Yb, u c)     :398-399
    emit_term<S>(scalars, pre, k + 4, u_plain, pr + 64);                                             // (R, u)        :401-402
    us = fp_mul<Fr>(uu, s); usb = fp_mul<Fr>(uu, sb);                                                // :405-406
  }
  red[threadIdx.x] = us;
  __syncthreads();
  for (int s2 = 64; s2 >= 1; s2 >>= 1) {
    if ((int)threadIdx.x < s2) red[threadIdx.x] = fp_add<Fr>(red[threadIdx.x], red[threadIdx.x + s2]);
    __syncthreads();
  }
  if (threadIdx.x == 0) store_fp(gpart + 8 * (size_t)blockIdx.x, red[0]);
  __syncthreads();
  red[threadIdx.x] = usb;
  __syncthreads();
  for (int s2 = 64; s2 >= 1; s2 >>= 1) {
    if ((int)threadIdx.x < s2) red[threadIdx.x] = fp_add<Fr>(red[threadIdx.x], red[threadIdx.x + s2]);
    __syncthreads();
  }
  if (threadIdx.x == 0) store_fp(bpart + 8 * (size_t)blockIdx.x, red[0]);
}

template <class S> void BatchOps<S>::ped_prepare(const BatchDev &b, uint32_t *d_c, uint8_t *d_merged, uint32_t *d_flags, hipStream_t st) {
  dim3 g((b.n + 127) / 128), blk(128);
  hipLaunchKernelGGL(k_ped_prepare<S>, g, blk, 0, st, b, d_c, d_merged, d_flags);
}
template <class S> void BatchOps<S>::ped_terms(const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint8_t *d_merged,
                                               uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st) {
  dim3 g((b.n + 127) / 128), blk(128);
  uint32_t *bp = d_gpart + 8 * (size_t)g.x;
  hipLaunchKernelGGL(k_ped_terms<S>, g, blk, 0, st, b, seed, j0, d_c, d_merged, d_scalars, (te_pre *)d_pre, d_gpart, bp);
  hipLaunchKernelGGL(k_g_final<S>, dim3(1), dim3(256), 0, st, d_gpart, g.x, d_scalars, (te_pre *)d_pre, n_terms - 2, 0);
  hipLaunchKernelGGL(k_g_final<S>, dim3(1), dim3(256), 0, st, bp, g.x, d_scalars, (te_pre *)d_pre, n_terms - 1, 1);
}
template <class S> void BatchOps<S>::thin_prepare(const BatchDev &b, uint32_t *d_c, uint32_t *d_z, uint32_t *d_flags, hipStream_t st) {
  dim3 g((b.n + 127) / 128), blk(128);
  hipLaunchKernelGGL(k_thin_prepare<S>, g, blk, 0, st, b, d_c, d_z, d_flags);
}
template <class S> void BatchOps<S>::thin_terms(const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint32_t *d_z,
                                                uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st) {
  dim3 g((b.n + 127) / 128), blk(128);
  hipLaunchKernelGGL(k_thin_terms<S>, g, blk, 0, st, b, seed, j0, d_c, d_z, d_scalars, (te_pre *)d_pre, d_gpart);
  hipLaunchKernelGGL(k_g_final<S>, dim3(1), dim3(256), 0, st, d_gpart, g.x, d_scalars, (te_pre *)d_pre, n_terms - 1, 0);
}
template struct BatchOps<suite_by_id<AVRF_TU_SUITE>::type>;

#else   // run-time dispatch unit

#define AVRF_BATCH(suite, CALL) with_suite((suite), [&](auto tag_) { using S_ = typename decltype(tag_)::type; BatchOps<S_>::CALL; })

void launch_ped_prepare(int suite, const BatchDev &b, uint32_t *d_c, uint8_t *d_merged, uint32_t *d_flags, hipStream_t st) {
  if (!b.n) return;
  AVRF_BATCH(suite, ped_prepare(b, d_c, d_merged, d_flags, st));
}
void launch_ped_terms(int suite, const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint8_t *d_merged,
                      uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st) {
  if (!b.n) return;
  AVRF_BATCH(suite, ped_terms(b, seed, j0, d_c, d_merged, d_scalars, d_pre, d_gpart, n_terms, st));
}
void launch_thin_prepare(int suite, const BatchDev &b, uint32_t *d_c, uint32_t *d_z, uint32_t *d_flags, hipStream_t st) {
  if (!b.n) return;
  AVRF_BATCH(suite, thin_prepare(b, d_c, d_z, d_flags, st));
}
void launch_thin_terms(int suite, const BatchDev &b, const Seed64 &seed, uint64_t j0, const uint32_t *d_c, const uint32_t *d_z,
                       uint32_t *d_scalars, te_pre_raw *d_pre, uint32_t *d_gpart, uint32_t n_terms, hipStream_t st) {
  if (!b.n) return;
  AVRF_BATCH(suite, thin_terms(b, seed, j0, d_c, d_z, d_scalars, d_pre, d_gpart, n_terms, st));
}
#endif

}  // namespace avrf
