// suite_dispatch.h -- run-time suite id -> compile-time Suite (trait Suite, src/lib.rs:177-250; src/suites/*.rs)
#pragma once
#include "consts_gen.h"

namespace avrf {

constexpr int AVRF_N_SUITES = 3;   // 0 Bandersnatch-SHA512-ELL2, 1 BabyJubJub-SHA512-TAI, 2 JubJub-SHA512-TAI
template <class S> struct SuiteTag { using type = S; };

template <class F> inline auto with_suite(int suite, F &&f) {
  switch (suite) {
    case 1: return f(SuiteTag<SuiteBabyJubJub>{});
    case 2: return f(SuiteTag<SuiteJubJub>{});
    default: return f(SuiteTag<SuiteBandersnatch>{});
  }
}
// RingSuite::Pairing (src/suites/{bandersnatch,jubjub}.rs: BLS12-381; baby_jubjub.rs: BN254): 0 BLS12-381, 1 BN254
inline int pairing_curve_of(int suite) { return suite == 1 ? 1 : 0; }

}  // namespace avrf
