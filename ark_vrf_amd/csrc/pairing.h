// pairing.h -- device pairing-product checks (pairing.hip): per-setup line tables of the fixed G2 arguments and the launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace avrf {

// indices into the per-curve constant block (Fp elements, Montgomery form) the kernels read
enum : int {
  PC_XI0 = 0,               // xi0 (xi = xi0 + u)
  PC_KAPPA,                 // xi0^2 + 1
  PC_2XI,                   // 2 xi0
  PC_2XI_KAPPA,             // 2 xi0 (xi0^2 + 1)
  PC_4XI2_MINUS_KAPPA,      // 4 xi0^2 - (xi0^2 + 1)
  PC_FROB2,                 // gamma^k, k < 6, gamma = xi^((p^2-1)/6) in Fp
  PC_FROB1 = PC_FROB2 + 6,  // xi^(k (p-1)/6) in Fp2 as (a, b), k < 6
  PC_COUNT = PC_FROB1 + 12
};

struct PairingTables {
  int curve = 0;                 // 0 BLS12-381, 1 BN254
  uint32_t steps = 0, nq = 0, words = 0;
  uint32_t *d_cst = nullptr;     // PC_COUNT x N words
  uint32_t *d_tab = nullptr;     // nq x steps x 36 x N words: {T0[12], TX[12], TY[12]} per step
  // g2_raw: nq `powers_in_g2` entries exactly as in an arkworks URS file (src/ring.rs:380-393).  Throws HipFailure.
  void build(int curve, const uint8_t *g2_raw, size_t nq, hipStream_t stream);
  void release();
};

// d_pts: n x nq affine G1 points, Montgomery x | y (as launch_g1_bases writes them; (0, 0) = infinity);
// d_ok[i] = 1 iff prod_q e(P_iq, Q_q) == 1
void launch_pairing_check(const PairingTables &pt, const uint32_t *d_pts, size_t n, int32_t *d_ok, hipStream_t stream);

}  // namespace avrf
