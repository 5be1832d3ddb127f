// Exercises the C++ mirror of the reference interface (include/avrf.hpp) the way the reference's
// own tests do (src/thin.rs:333-384 prove_verify / batch_verify; src/pedersen.rs:434-488).
// Usage: mirror_check <suite> <sk_hex> <input_xy_hex> <ad_hex>; prints hex lines checked by pytest.
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
  return 0;
}
