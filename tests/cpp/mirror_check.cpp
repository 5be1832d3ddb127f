// Exercises the C++ mirror of the reference interface (include/avrf.hpp) the way the reference's
// own tests do (src/thin.rs:333-384 prove_verify / batch_verify; src/pedersen.rs:434-488).
// Usage: mirror_check <suite> <sk_hex> <input_xy_hex> <ad_hex> [<srs file> <ring pks xy hex> <key index>];
// prints hex lines checked by pytest.  With the ring arguments it also runs ring::{prove, verify, BatchVerifier}
// (src/ring.rs:1017-1140 prove_verify / prove_verify_batch).
#include <cstdio>
#include <cstring>
#include <string>
#include "avrf.hpp"

static std::vector<uint8_t> unhex(const char *s) {
  std::vector<uint8_t> v; size_t n = strlen(s);
  for (size_t i = 0; i + 1 < n; i += 2) { unsigned x; sscanf(s + i, "%2x", &x); v.push_back((uint8_t)x); }
  return v;
}
template <class T> static void put(const char *k, const T &a) { printf("%s=", k); for (uint8_t b : a) printf("%02x", b); printf("\n"); }

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  using namespace avrf;
  Suite su((Suite::Id)atoi(argv[1]));
  Scalar sk; auto skb = unhex(argv[2]); std::copy(skb.begin(), skb.end(), sk.begin());
  Point in; auto inb = unhex(argv[3]); std::copy(inb.begin(), inb.end(), in.begin());
  auto adb = unhex(argv[4]); std::string ad(adb.begin(), adb.end());

  Secret secret = Secret::from_scalar(su, sk);
  Public pub = secret.public_key;
  VrfIo io = secret.vrf_io(su, in);
  put("pk", pub.point); put("output", io.output);
  put("beta", output_hash(su, io.output));                               // Output::hash::<32>, src/lib.rs:605-609
  if (argc > 5) {                                                        // Secret::from_seed, src/lib.rs:346-369
    std::array<uint8_t, 32> seed{}; auto sb = unhex(argv[5]); std::copy(sb.begin(), sb.end(), seed.begin());
    Secret fs = Secret::from_seed(su, seed);
    put("seed_sk", fs.scalar); put("seed_pk", fs.public_key.point);
  }

  tiny::Proof yp = tiny::prove(su, secret, {io}, ad);                    // src/tiny.rs tests: prove_verify
  put("tiny_c", yp.c); put("tiny_s", yp.s);
  printf("tiny_verify=%d\n", tiny::verify(su, pub, {io}, ad, yp));
  printf("tiny_verify_bad_ad=%d\n", tiny::verify(su, pub, {io}, ad + "x", yp));
  { tiny::Proof b = yp; b.c[3] ^= 1; printf("tiny_verify_bad_c=%d\n", tiny::verify(su, pub, {io}, ad, b)); }

  thin::Proof tp = thin::prove(su, secret, {io}, ad);
  put("thin_r", tp.r); put("thin_s", tp.s);
  printf("thin_verify=%d\n", thin::verify(su, pub, {io}, ad, tp));
  printf("thin_verify_bad_ad=%d\n", thin::verify(su, pub, {io}, ad + "x", tp));
  thin::BatchVerifier tb(su);
  printf("thin_batch_empty=%d\n", tb.verify());
  for (int i = 0; i < 3; i++) tb.push(pub, {io}, ad, tp);
  printf("thin_batch=%d\n", tb.verify());
  thin::Proof bad = tp; bad.s[0] ^= 1;
  tb.push(pub, {io}, ad, bad);
  printf("thin_batch_bad=%d\n", tb.verify());
  {  // the same two batches through the pool (native host threads, batches in flight together)
    thin::BatchVerifier good(su);
    for (int i = 0; i < 3; i++) good.push(pub, {io}, ad, tp);
    thin::VerifierPool pool(atoi(argv[1]), 0, 4, 2, 2, 8);
    const uint64_t t1 = pool.submit(good), t2 = pool.submit(tb), t3 = pool.submit(good);
    printf("thin_pool=%d%d%d\n", pool.wait(t1), pool.wait(t2), pool.wait(t3));
  }

  auto pp = pedersen::prove(su, secret, {io}, ad);
  put("ped_pk_com", pp.first.pk_com); put("ped_r", pp.first.r); put("ped_ok", pp.first.ok);
  put("ped_s", pp.first.s); put("ped_sb", pp.first.sb); put("ped_blinding", pp.second);
  printf("ped_verify=%d\n", pedersen::verify(su, {io}, ad, pp.first));
  pedersen::BatchVerifier pb(su);
  for (int i = 0; i < 2; i++) pb.push({io}, ad, pp.first);
  printf("ped_batch=%d\n", pb.verify());
  pedersen::Proof pbad = pp.first; pbad.sb[1] ^= 2;
  pb.push({io}, ad, pbad);
  printf("ped_batch_bad=%d\n", pb.verify());
  if (argc >= 8) {
    std::vector<uint8_t> srs; { FILE *f = fopen(argv[5], "rb"); if (!f) return 3; uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) srs.insert(srs.end(), buf, buf + k); fclose(f); }
    auto ringb = unhex(argv[6]);
    std::vector<Public> pks(ringb.size() / 64);
    for (size_t i = 0; i < pks.size(); i++) std::copy(ringb.begin() + 64 * i, ringb.begin() + 64 * (i + 1), pks[i].point.begin());
    ring::RingSetup setup;
    printf("ring_setup_too_big=%d\n", ring::RingSetup::from_pcs_params(su, 1000000, srs, &setup));
    printf("ring_setup=%d\n", ring::RingSetup::from_pcs_params(su, pks.size(), srs, &setup));
    ring::RingKey key;
    printf("ring_index=%d\n", ring::RingKey::index(setup, pks, &key));
    put("ring_commitment", key.commitment());
    ring::RingProver prover{&key, (uint32_t)atoi(argv[7])};
    ring::RingVerifier verifier{&setup, key.commitment()};
    ring::Proof rp = ring::prove(su, secret, {io}, ad, prover);
    put("ring_proof", rp.ring_proof); put("ring_ped_pk_com", rp.pedersen_proof.pk_com);
    printf("ring_verify=%d\n", ring::verify(su, {io}, ad, rp, verifier));
    printf("ring_verify_bad_ad=%d\n", ring::verify(su, {io}, ad + "x", rp, verifier));
    ring::Proof other = rp; other.ring_proof[other.ring_proof.size() - 40] ^= 4;
    printf("ring_verify_bad_proof=%d\n", ring::verify(su, {io}, ad, other, verifier) != 0);
    prover.hiding = true;
    ring::Proof hp = ring::prove(su, secret, {io}, ad, prover);
    printf("ring_hiding_differs=%d\n", hp.ring_proof != rp.ring_proof);
    ring::BatchVerifier rb(su, verifier);
    rb.push(verifier, {io}, ad, rp); rb.push(verifier, {io}, ad, hp);
    printf("ring_batch=%d\n", rb.verify());
    rb.push(verifier, {io}, ad, other);
    printf("ring_batch_bad=%d\n", rb.verify() != 0);
  }
  return 0;
}
