// suite_dispatch.h -- run-time suite id -> compile-time Suite (trait Suite, src/lib.rs:177-250; src/suites/*.rs)
#pragma once
#include "consts_gen.h"

namespace avrf {

constexpr int AVRF_N_SUITES = 8;   // 0 Bandersnatch-SHA512-ELL2, 1 BabyJubJub-SHA512-TAI, 2 JubJub-SHA512-TAI, 3 Ed25519-SHA512-TAI, 4 Bandersnatch-SW-SHA512-TAI, 5 Bandersnatch-SHAKE128-ELL2, 6 Testing-SHA256-TAI, 7 Secp256r1-SHA256-TAI
template <class S> struct SuiteTag { using type = S; };

template <class F> inline auto with_suite(int suite, F &&f) {
  switch (suite) {
    case 1: return f(SuiteTag<SuiteBabyJubJub>{});
    case 2: return f(SuiteTag<SuiteJubJub>{});
    case 3: return f(SuiteTag<SuiteEd25519>{});
    case 4: return f(SuiteTag<SuiteBandersnatchSW>{});
    case 5: return f(SuiteTag<SuiteBandersnatchShake>{});
    case 6: return f(SuiteTag<SuiteTesting>{});
    case 7: return f(SuiteTag<SuiteSecp256r1>{});
    default: return f(SuiteTag<SuiteBandersnatch>{});
  }
}
// compile-time id -> Suite (the per-suite translation units, -DAVRF_TU_SUITE=<id>)
template <int K> struct suite_by_id;
template <> struct suite_by_id<0> { using type = SuiteBandersnatch; };
template <> struct suite_by_id<1> { using type = SuiteBabyJubJub; };
template <> struct suite_by_id<2> { using type = SuiteJubJub; };
template <> struct suite_by_id<3> { using type = SuiteEd25519; };
template <> struct suite_by_id<4> { using type = SuiteBandersnatchSW; };
template <> struct suite_by_id<5> { using type = SuiteBandersnatchShake; };
template <> struct suite_by_id<6> { using type = SuiteTesting; };
template <> struct suite_by_id<7> { using type = SuiteSecp256r1; };
// RingSuite::Pairing (src/suites/{bandersnatch,jubjub}.rs: BLS12-381; baby_jubjub.rs: BN254): 0 BLS12-381, 1 BN254
inline int pairing_curve_of(int suite) { return suite == 1 ? 1 : 0; }
// trait RingSuite (src/ring.rs:97-150) is implemented for the suites whose base field is a pairing curve's scalar field
inline bool ring_suite(int suite) { return (suite >= 0 && suite <= 2) || suite == 4 || suite == 5; }
// serialize_compressed size of the suite's Affine: 32 (twisted Edwards: y and the sign of x), 33 for the short-Weierstrass presentation
inline int point_len_of(int suite) { return (suite == 4 || suite == 7) ? 33 : 32; }

}  // namespace avrf
