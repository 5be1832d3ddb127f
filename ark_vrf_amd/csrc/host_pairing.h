// host_pairing.h -- host-side pairing check for the ring verifier: e(A, g2) * e(B, tau*g2) == 1.
//
// Counterpart of arkworks `Pairing::multi_pairing` + final exponentiation reached from
// `RingVerifier::verify` (src/ring.rs:242) and the multi-ring batch verifier (src/ring.rs:731).
// A ring *batch* verification needs exactly two Miller loops and one final exponentiation however
// many proofs it covers (the per-proof work is the G1 MSM, which runs on the GPU), so the pairing is
// kept on the host (SURVEY.md §7 "Pairing on GPU ... 2 pairings per batch -- run on CPU").
//
// Tower: Fp2 = Fp[u]/(u^2+1), xi = XI0 + u, Fp6 = Fp2[v]/(v^3 - xi), Fp12 = Fp6[w]/(w^2 - v).
// Plain ate pairing: Miller loop over t - 1 with affine points on the sextic twist, lines evaluated
// at P in G1 in sparse form, then f^((p^12-1)/r).  Product code (not the oracle).
#pragma once
#include "host_g1.h"

namespace avrf {

template <class C> struct HostPairing {
  using Fp = HostFieldN<typename C::Fq>;
  using El = typename Fp::El;
  struct F2 { El a, b; };                    // a + b u
  struct F6 { F2 c0, c1, c2; };              // c0 + c1 v + c2 v^2
  struct F12 { F6 c0, c1; };                 // c0 + c1 w
  struct G2 { F2 x, y; bool inf; };

  static F2 f2(const El &a, const El &b) { F2 r; r.a = a; r.b = b; return r; }
  static F2 f2_zero() { return f2(Fp::zero(), Fp::zero()); }
  static F2 f2_one() { return f2(Fp::one(), Fp::zero()); }
  static bool f2_is_zero(const F2 &x) { return Fp::is_zero(x.a) && Fp::is_zero(x.b); }
  static bool f2_eq(const F2 &x, const F2 &y) { return Fp::eq(x.a, y.a) && Fp::eq(x.b, y.b); }
  static F2 f2_add(const F2 &x, const F2 &y) { return f2(Fp::add(x.a, y.a), Fp::add(x.b, y.b)); }
  static F2 f2_sub(const F2 &x, const F2 &y) { return f2(Fp::sub(x.a, y.a), Fp::sub(x.b, y.b)); }
  static F2 f2_neg(const F2 &x) { return f2(Fp::neg(x.a), Fp::neg(x.b)); }
  static F2 f2_mul(const F2 &x, const F2 &y) {
    El t0 = Fp::mul(x.a, y.a), t1 = Fp::mul(x.b, y.b);
    El t2 = Fp::mul(Fp::add(x.a, x.b), Fp::add(y.a, y.b));
    return f2(Fp::sub(t0, t1), Fp::sub(Fp::sub(t2, t0), t1));
  }
  static F2 f2_sqr(const F2 &x) { return f2_mul(x, x); }
  static F2 f2_scale(const F2 &x, const El &k) { return f2(Fp::mul(x.a, k), Fp::mul(x.b, k)); }
  static F2 f2_inv(const F2 &x) {
    El n = Fp::inv(Fp::add(Fp::sqr(x.a), Fp::sqr(x.b)));
    return f2(Fp::mul(x.a, n), Fp::neg(Fp::mul(x.b, n)));
  }
  static El small(int k) { El r = Fp::zero(), one = Fp::one(); for (int i = 0; i < k; i++) r = Fp::add(r, one); return r; }
  static F2 f2_mul_xi(const F2 &x) {          // (a + b u)(XI0 + u) = (XI0 a - b) + (a + XI0 b) u
    static const El k = small(C::XI0);
    return f2(Fp::sub(Fp::mul(x.a, k), x.b), Fp::add(x.a, Fp::mul(x.b, k)));
  }
  static F6 f6(const F2 &a, const F2 &b, const F2 &c) { F6 r; r.c0 = a; r.c1 = b; r.c2 = c; return r; }
  static F6 f6_zero() { return f6(f2_zero(), f2_zero(), f2_zero()); }
  static F6 f6_add(const F6 &x, const F6 &y) { return f6(f2_add(x.c0, y.c0), f2_add(x.c1, y.c1), f2_add(x.c2, y.c2)); }
  static F6 f6_sub(const F6 &x, const F6 &y) { return f6(f2_sub(x.c0, y.c0), f2_sub(x.c1, y.c1), f2_sub(x.c2, y.c2)); }
  static F6 f6_mul(const F6 &x, const F6 &y) {
    F2 v0 = f2_mul(x.c0, y.c0), v1 = f2_mul(x.c1, y.c1), v2 = f2_mul(x.c2, y.c2);
    F2 t0 = f2_sub(f2_sub(f2_mul(f2_add(x.c1, x.c2), f2_add(y.c1, y.c2)), v1), v2);
    F2 t1 = f2_sub(f2_sub(f2_mul(f2_add(x.c0, x.c1), f2_add(y.c0, y.c1)), v0), v1);
    F2 t2 = f2_sub(f2_sub(f2_mul(f2_add(x.c0, x.c2), f2_add(y.c0, y.c2)), v0), v2);
    return f6(f2_add(v0, f2_mul_xi(t0)), f2_add(t1, f2_mul_xi(v2)), f2_add(t2, v1));
  }
  static F6 f6_mul_v(const F6 &x) { return f6(f2_mul_xi(x.c2), x.c0, x.c1); }
  static F12 f12_one() { F12 r; r.c0 = f6(f2_one(), f2_zero(), f2_zero()); r.c1 = f6_zero(); return r; }
  static F12 f12_mul(const F12 &x, const F12 &y) {
    F6 v0 = f6_mul(x.c0, y.c0), v1 = f6_mul(x.c1, y.c1);
    F12 r;
    r.c1 = f6_sub(f6_sub(f6_mul(f6_add(x.c0, x.c1), f6_add(y.c0, y.c1)), v0), v1);
    r.c0 = f6_add(v0, f6_mul_v(v1));
    return r;
  }
  static F6 f6_neg(const F6 &x) { return f6(f2_neg(x.c0), f2_neg(x.c1), f2_neg(x.c2)); }
  static F6 f6_scale2(const F6 &x, const F2 &k) { return f6(f2_mul(x.c0, k), f2_mul(x.c1, k), f2_mul(x.c2, k)); }
  static F6 f6_inv(const F6 &x) {                                   // (A + B v + C v^2) / (c0 A + xi (c2 B + c1 C))
    F2 A = f2_sub(f2_sqr(x.c0), f2_mul_xi(f2_mul(x.c1, x.c2)));
    F2 B = f2_sub(f2_mul_xi(f2_sqr(x.c2)), f2_mul(x.c0, x.c1));
    F2 Cc = f2_sub(f2_sqr(x.c1), f2_mul(x.c0, x.c2));
    F2 Fd = f2_add(f2_mul(x.c0, A), f2_mul_xi(f2_add(f2_mul(x.c2, B), f2_mul(x.c1, Cc))));
    return f6_scale2(f6(A, B, Cc), f2_inv(Fd));
  }
  static F12 f12_conj(const F12 &x) { F12 r; r.c0 = x.c0; r.c1 = f6_neg(x.c1); return r; }     // x^(p^6)
  static F12 f12_inv(const F12 &x) {                                // (a - b w) / (a^2 - b^2 v)
    F6 d = f6_inv(f6_sub(f6_mul(x.c0, x.c0), f6_mul_v(f6_mul(x.c1, x.c1))));
    F12 r; r.c0 = f6_mul(x.c0, d); r.c1 = f6_neg(f6_mul(x.c1, d)); return r;
  }
  // x^(p^2): the coefficient of w^k (w^6 = xi; order c0.c0, c1.c0, c0.c1, c1.c1, c0.c2, c1.c2) times gamma^k,
  // gamma = xi^((p^2-1)/6) in Fp
  static F12 f12_frob2(const F12 &x) {
    static const El g1 = Fp::from32(C::FROB2_GAMMA), g2 = Fp::mul(g1, g1), g3 = Fp::mul(g2, g1), g4 = Fp::mul(g3, g1), g5 = Fp::mul(g4, g1);
    F12 r;
    r.c0 = f6(x.c0.c0, f2_scale(x.c0.c1, g2), f2_scale(x.c0.c2, g4));
    r.c1 = f6(f2_scale(x.c1.c0, g1), f2_scale(x.c1.c1, g3), f2_scale(x.c1.c2, g5));
    return r;
  }
  // x^p: the coefficient of w^k is conjugated in Fp2 and multiplied by gamma1^k, gamma1 = xi^((p-1)/6)
  static F12 f12_frob1(const F12 &x) {
    auto g = [](int k) { return f2(Fp::from32(C::FROB1_GAMMA[k - 1][0]), Fp::from32(C::FROB1_GAMMA[k - 1][1])); };
    auto cj = [](const F2 &a) { return f2(a.a, Fp::neg(a.b)); };
    static const F2 g1 = g(1), g2 = g(2), g3 = g(3), g4 = g(4), g5 = g(5);
    F12 r;
    r.c0 = f6(cj(x.c0.c0), f2_mul(cj(x.c0.c1), g2), f2_mul(cj(x.c0.c2), g4));
    r.c1 = f6(f2_mul(cj(x.c1.c0), g1), f2_mul(cj(x.c1.c1), g3), f2_mul(cj(x.c1.c2), g5));
    return r;
  }
  // Squaring in the cyclotomic subgroup (Granger-Scott, "Faster squaring in the cyclotomic subgroup of sixth degree
  // extensions"): three Fp4 squarings, 18 Fp multiplications instead of 54.  Valid only after the easy part.
  static void fp4_sqr(const F2 &a, const F2 &b, F2 &t0, F2 &t1) {    // (a + b y)^2 with y^2 = xi
    F2 ab = f2_mul(a, b);
    t0 = f2_sub(f2_sub(f2_mul(f2_add(a, b), f2_add(f2_mul_xi(b), a)), ab), f2_mul_xi(ab));
    t1 = f2_add(ab, ab);
  }
  static F12 f12_cyclo_sqr(const F12 &x) {
    const F2 &z0 = x.c0.c0, &z4 = x.c0.c1, &z3 = x.c0.c2, &z2 = x.c1.c0, &z1 = x.c1.c1, &z5 = x.c1.c2;
    F2 t0, t1, t2, t3, t4, t5;
    fp4_sqr(z0, z1, t0, t1); fp4_sqr(z2, z3, t2, t3); fp4_sqr(z4, z5, t4, t5);
    auto minus2plus3 = [](const F2 &t, const F2 &z) { F2 d = f2_sub(t, z); return f2_add(f2_add(d, d), t); };   // 3t - 2z
    auto plus2plus3 = [](const F2 &t, const F2 &z) { F2 d = f2_add(t, z); return f2_add(f2_add(d, d), t); };    // 3t + 2z
    F12 r;
    r.c0.c0 = minus2plus3(t0, z0); r.c1.c1 = plus2plus3(t1, z1);
    r.c1.c0 = plus2plus3(f2_mul_xi(t5), z2); r.c0.c2 = minus2plus3(t4, z3);
    r.c0.c1 = minus2plus3(t2, z4); r.c1.c2 = plus2plus3(t3, z5);
    return r;
  }
  // a^x for a in the cyclotomic subgroup (inverse = conjugate), x the curve parameter (negative for BLS12-381)
  static F12 f12_pow_x(const F12 &a) {
    F12 r = a;                                                       // top bit of x
    int top = 63; while (!((C::X_ABS >> top) & 1)) top--;
    for (int bit = top - 1; bit >= 0; bit--) { r = f12_cyclo_sqr(r); if ((C::X_ABS >> bit) & 1) r = f12_mul(r, a); }
    return C::X_NEG ? f12_conj(r) : r;
  }
  // f^((p^12-1)/r) up to a power coprime to r: easy part (p^6-1)(p^2+1) by conjugation, inversion and a p^2-Frobenius;
  // hard part (p^4-p^2+1)/r by plain square-and-multiply, or for BLS12 three times it as
  // (x-1)^2 (x+p) (x^2+p^2-1) + 3 (Hayashida-Hayasaka-Teruya): five 64-bit exponentiations.  Only "== 1" is used.
  static F12 final_exp(const F12 &f) {
    F12 t = f12_mul(f12_conj(f), f12_inv(f));
    t = f12_mul(f12_frob2(t), t);
    if (C::X_CHAIN) {
      F12 a = f12_mul(f12_pow_x(t), f12_conj(t));                     // t^(x-1)
      F12 b = f12_mul(f12_pow_x(a), f12_conj(a));                     // t^((x-1)^2)
      F12 c = f12_mul(f12_pow_x(b), f12_frob1(b));                    // ^(x+p)
      F12 d = f12_mul(f12_mul(f12_pow_x(f12_pow_x(c)), f12_frob2(c)), f12_conj(c));   // ^(x^2+p^2-1)
      return f12_mul(d, f12_mul(t, f12_mul(t, t)));                   // * t^3
    }
    F12 out = t;                                                     // top bit; t is in the cyclotomic subgroup
    for (int bit = C::HARD_EXP_BITS - 2; bit >= 0; bit--) {
      out = f12_cyclo_sqr(out);
      if ((C::HARD_EXP[bit >> 6] >> (bit & 63)) & 1) out = f12_mul(out, t);
    }
    return out;
  }
  static bool f12_is_one(const F12 &x) {
    return Fp::eq(x.c0.c0.a, Fp::one()) && Fp::is_zero(x.c0.c0.b) && f2_is_zero(x.c0.c1) && f2_is_zero(x.c0.c2) &&
           f2_is_zero(x.c1.c0) && f2_is_zero(x.c1.c1) && f2_is_zero(x.c1.c2);
  }

  // line through twist points with slope lam, passing (tx, ty), evaluated at P = (px, py) in G1
  static F12 line(const F2 &lam, const F2 &tx, const F2 &ty, const El &px, const El &py) {
    F2 c = f2_sub(f2_mul(lam, tx), ty);            // lam x' - y'
    F2 m = f2_neg(f2_scale(lam, px));              // -lam xP
    F2 yp = f2(py, Fp::zero());
    F12 r;
    if (C::MTWIST) { r.c0 = f6(c, m, f2_zero()); r.c1 = f6(f2_zero(), yp, f2_zero()); }      // (l * w^3): c + m v + yP v w
    else { r.c0 = f6(yp, f2_zero(), f2_zero()); r.c1 = f6(m, c, f2_zero()); }                // yP + m w + c v w
    return r;
  }

  // ---- fixed G2 arguments: the verifier's two G2 points (g2, tau g2) never change for a setup, so the whole G2 side of the
  // Miller loop -- the affine doublings / additions with their Fp2 inversions (an Fp inversion costs as much as ten Fp12
  // products) -- is done once: per step the slope lam and c = lam x' - y' of the line.  With the table a step is one Fp12
  // squaring and one sparse product per pair (device counterpart: k_g2_lines, pairing.hip).
  struct G2Lines { std::vector<F2> lam, c; size_t live_steps = 0; bool inf = true; };
  static G2Lines g2_lines(const G2 &q) {
    G2Lines t; t.inf = q.inf;
    if (q.inf) return t;
    static const El three = small(3), two = small(2);
    F2 rx = q.x, ry = q.y;
    bool live = true;
    auto push = [&](const F2 &lam) { t.lam.push_back(lam); t.c.push_back(f2_sub(f2_mul(lam, rx), ry)); };
    for (int bit = C::ATE_LOOP_BITS - 2; bit >= 0 && live; bit--) {
      if (f2_is_zero(ry)) { live = false; break; }
      F2 lam = f2_mul(f2_scale(f2_sqr(rx), three), f2_inv(f2_scale(ry, two)));
      push(lam);
      F2 nx = f2_sub(f2_sqr(lam), f2_add(rx, rx));
      ry = f2_sub(f2_mul(lam, f2_sub(rx, nx)), ry); rx = nx;
      if ((C::ATE_LOOP[bit >> 6] >> (bit & 63)) & 1) {
        if (f2_eq(rx, q.x)) { live = false; break; }                 // cannot happen for points of order r
        lam = f2_mul(f2_sub(q.y, ry), f2_inv(f2_sub(q.x, rx)));
        push(lam);
        nx = f2_sub(f2_sub(f2_sqr(lam), rx), q.x);
        ry = f2_sub(f2_mul(lam, f2_sub(rx, nx)), ry); rx = nx;
      }
    }
    t.live_steps = t.lam.size();
    return t;
  }
  // (a + b w)^2 = (a + b)(a + v b) - ab - v ab + 2 ab w: two Fp6 products
  static F12 f12_sqr(const F12 &x) {
    F6 ab = f6_mul(x.c0, x.c1);
    F6 t = f6_mul(f6_add(x.c0, x.c1), f6_add(x.c0, f6_mul_v(x.c1)));
    F12 r; r.c0 = f6_sub(f6_sub(t, ab), f6_mul_v(ab)); r.c1 = f6_add(ab, ab);
    return r;
  }
  // x * (u0 + u1 v) and x * (u1 v) in Fp6 (v^3 = xi)
  static F6 f6_mul_01(const F6 &x, const F2 &u0, const F2 &u1) {
    F2 a = f2_mul(x.c0, u0), b = f2_mul(x.c1, u1);
    F2 c1 = f2_sub(f2_sub(f2_mul(f2_add(x.c0, x.c1), f2_add(u0, u1)), a), b);
    return f6(f2_add(a, f2_mul_xi(f2_mul(x.c2, u1))), c1, f2_add(f2_mul(x.c2, u0), b));
  }
  static F6 f6_mul_1(const F6 &x, const F2 &u1) { return f6(f2_mul_xi(f2_mul(x.c2, u1)), f2_mul(x.c0, u1), f2_mul(x.c1, u1)); }
  // f * line, the line in the sparse shape `line()` builds: M-twist (c + m v) + (yP v) w, D-twist (yP) + (m + c v) w
  static F12 f12_mul_line(const F12 &f, const F2 &c, const F2 &m, const El &py) {
    const F2 yp = f2(py, Fp::zero());
    F12 r;
    if (C::MTWIST) {
      F6 v0 = f6_mul_01(f.c0, c, m), v1 = f6_mul_1(f.c1, yp);
      F6 t = f6_mul_01(f6_add(f.c0, f.c1), c, f2_add(m, yp));
      r.c1 = f6_sub(f6_sub(t, v0), v1); r.c0 = f6_add(v0, f6_mul_v(v1));
    } else {
      F6 v0 = f6(f2_scale(f.c0.c0, py), f2_scale(f.c0.c1, py), f2_scale(f.c0.c2, py)), v1 = f6_mul_01(f.c1, m, c);
      F6 t = f6_mul_01(f6_add(f.c0, f.c1), f2_add(yp, m), c);
      r.c1 = f6_sub(f6_sub(t, v0), v1); r.c0 = f6_add(v0, f6_mul_v(v1));
    }
    return r;
  }
  // prod_i e(P_i, Q_i) == 1 with the Q_i given as line tables
  static bool product_is_one_lines(const El *px, const El *py, const bool *pinf, const G2Lines *tabs, int n) {
    F12 f = f12_one();
    std::vector<size_t> pos(n, 0);
    for (int bit = C::ATE_LOOP_BITS - 2; bit >= 0; bit--) {
      f = f12_sqr(f);
      const int steps = 1 + (int)((C::ATE_LOOP[bit >> 6] >> (bit & 63)) & 1);
      for (int st = 0; st < steps; st++)
        for (int i = 0; i < n; i++) {
          if (pinf[i] || tabs[i].inf || pos[i] >= tabs[i].live_steps) continue;
          const size_t k = pos[i]++;
          f = f12_mul_line(f, tabs[i].c[k], f2_neg(f2_scale(tabs[i].lam[k], px[i])), py[i]);
        }
    }
    return f12_is_one(final_exp(f));
  }

  // the same with one Miller loop per pair, each on its own thread of `pf` (pf(n, fn) runs fn(0..n-1)): a pair then pays its own
  // squarings of f, but for the verifier's two pairs the loop takes 63 squarings + 69 line products on the critical path
  // instead of 63 + 138 (a lone verification is a latency case: 1.04 -> 0.85 ms for the check)
  template <class ParFor>
  static bool product_is_one_lines_par(const El *px, const El *py, const bool *pinf, const G2Lines *tabs, int n, ParFor pf) {
    std::vector<F12> fs(n);
    pf((size_t)n, [&](size_t i) {
      F12 f = f12_one();
      if (!(pinf[i] || tabs[i].inf)) {
        size_t pos = 0;
        for (int bit = C::ATE_LOOP_BITS - 2; bit >= 0; bit--) {
          f = f12_sqr(f);
          const int steps = 1 + (int)((C::ATE_LOOP[bit >> 6] >> (bit & 63)) & 1);
          for (int st = 0; st < steps && pos < tabs[i].live_steps; st++, pos++)
            f = f12_mul_line(f, tabs[i].c[pos], f2_neg(f2_scale(tabs[i].lam[pos], px[i])), py[i]);
        }
      }
      fs[i] = f;
    });
    F12 f = fs[0];
    for (int i = 1; i < n; i++) f = f12_mul(f, fs[i]);
    return f12_is_one(final_exp(f));
  }

  // prod_i e(P_i, Q_i) == 1 ?   P_i affine G1 (Montgomery x, y; inf flag), Q_i affine G2 on the twist
  static bool product_is_one(const El *px, const El *py, const bool *pinf, const G2 *q, int n) {
    std::vector<F2> rx(n), ry(n);
    std::vector<bool> live(n);
    for (int i = 0; i < n; i++) { live[i] = !pinf[i] && !q[i].inf; rx[i] = q[i].x; ry[i] = q[i].y; }
    F12 f = f12_one();
    static const El three = small(3), two = small(2);
    // all slopes of one step share ONE Fp2 inversion (Montgomery's trick over the live pairs)
    std::vector<F2> den(n), pre(n);
    auto batch_inv = [&]() {                                         // den[i] <- 1 / den[i] for live i (all non-zero)
      F2 run = f2_one();
      for (int i = 0; i < n; i++) if (live[i]) { pre[i] = run; run = f2_mul(run, den[i]); }
      F2 inv = f2_inv(run);
      for (int i = n - 1; i >= 0; i--) if (live[i]) { F2 d = den[i]; den[i] = f2_mul(inv, pre[i]); inv = f2_mul(inv, d); }
    };
    for (int bit = C::ATE_LOOP_BITS - 2; bit >= 0; bit--) {
      f = f12_mul(f, f);
      for (int i = 0; i < n; i++) if (live[i]) { if (f2_is_zero(ry[i])) live[i] = false; else den[i] = f2_scale(ry[i], two); }
      batch_inv();
      for (int i = 0; i < n; i++) {
        if (!live[i]) continue;
        F2 lam = f2_mul(f2_scale(f2_sqr(rx[i]), three), den[i]);      // 3 x^2 / (2 y)
        f = f12_mul(f, line(lam, rx[i], ry[i], px[i], py[i]));
        F2 nx = f2_sub(f2_sqr(lam), f2_add(rx[i], rx[i]));
        ry[i] = f2_sub(f2_mul(lam, f2_sub(rx[i], nx)), ry[i]); rx[i] = nx;
      }
      if ((C::ATE_LOOP[bit >> 6] >> (bit & 63)) & 1) {
        for (int i = 0; i < n; i++) if (live[i]) { if (f2_eq(rx[i], q[i].x)) live[i] = false; else den[i] = f2_sub(q[i].x, rx[i]); }   // cannot happen for points of order r
        batch_inv();
        for (int i = 0; i < n; i++) {
          if (!live[i]) continue;
          F2 lam = f2_mul(f2_sub(q[i].y, ry[i]), den[i]);
          f = f12_mul(f, line(lam, rx[i], ry[i], px[i], py[i]));
          F2 nx = f2_sub(f2_sub(f2_sqr(lam), rx[i]), q[i].x);
          ry[i] = f2_sub(f2_mul(lam, f2_sub(rx[i], nx)), ry[i]); rx[i] = nx;
        }
      }
    }
    F12 out = final_exp(f);                                          // f^((p^12 - 1) / r)
    return f12_is_one(out);
  }

  // affine G2 arithmetic on the twist (y^2 = x^3 + b'), only for tau * g2 when an SRS is generated (Kzg::setup)
  static G2 g2_dbl(const G2 &p) {
    if (p.inf || f2_is_zero(p.y)) { G2 r = p; r.inf = true; return r; }
    static const El three = small(3), two = small(2);
    F2 lam = f2_mul(f2_scale(f2_sqr(p.x), three), f2_inv(f2_scale(p.y, two)));
    G2 r; r.inf = false;
    r.x = f2_sub(f2_sqr(lam), f2_add(p.x, p.x));
    r.y = f2_sub(f2_mul(lam, f2_sub(p.x, r.x)), p.y);
    return r;
  }
  static G2 g2_add(const G2 &a, const G2 &b) {
    if (a.inf) return b;
    if (b.inf) return a;
    if (f2_eq(a.x, b.x)) { if (f2_eq(a.y, b.y)) return g2_dbl(a); G2 r = a; r.inf = true; return r; }
    F2 lam = f2_mul(f2_sub(b.y, a.y), f2_inv(f2_sub(b.x, a.x)));
    G2 r; r.inf = false;
    r.x = f2_sub(f2_sub(f2_sqr(lam), a.x), b.x);
    r.y = f2_sub(f2_mul(lam, f2_sub(a.x, r.x)), a.y);
    return r;
  }
  static G2 g2_mul(const G2 &p, const uint64_t k[4]) {               // k: plain 256-bit scalar, little-endian limbs
    G2 r = p; r.inf = true;
    for (int i = 255; i >= 0; i--) { r = g2_dbl(r); if ((k[i >> 6] >> (i & 63)) & 1) r = g2_add(r, p); }
    return r;
  }
  // inverse of g2_decode: one `powers_in_g2` entry (serialize_uncompressed)
  static void g2_encode(const G2 &p, uint8_t *b) {
    constexpr int B = 8 * Fp::L;
    memset(b, 0, 4 * B);
    if (B == 48) {
      if (p.inf) { b[0] = 0x40; return; }
      El v[4] = {Fp::from_mont(p.x.b), Fp::from_mont(p.x.a), Fp::from_mont(p.y.b), Fp::from_mont(p.y.a)};
      for (int k = 0; k < 4; k++) { uint8_t le[48]; memcpy(le, v[k].l, B); for (int i = 0; i < B; i++) b[k * B + i] = le[B - 1 - i]; }
    } else {
      if (p.inf) { b[4 * B - 1] = 0x40; return; }
      El v[4] = {Fp::from_mont(p.x.a), Fp::from_mont(p.x.b), Fp::from_mont(p.y.a), Fp::from_mont(p.y.b)};
      for (int k = 0; k < 4; k++) memcpy(b + k * B, v[k].l, B);
      // arkworks SWFlags: y > -y, Fp2 ordered by (c1, c0)
      El half = Fp::from32(C::Fq::HALF), t; const El &key = Fp::is_zero(v[3]) ? v[2] : v[3];
      if (Fp::subb(t, half, key)) b[4 * B - 1] |= 0x80;
    }
  }

  // ---- compressed G2 encodings (ark-serialize `serialize_compressed` of a URS / RingSetup, src/ring.rs:484-521)
  static F2 f2_pow(const F2 &x, const El &e) {                       // e: plain integer
    F2 r = f2_one();
    for (int i = 64 * Fp::L - 1; i >= 0; i--) { r = f2_sqr(r); if ((e.l[i / 64] >> (i % 64)) & 1) r = f2_mul(r, x); }
    return r;
  }
  // square root in Fp2 for p = 3 mod 4 (Adj, Rodriguez-Henriquez, "Square root computation over even extension fields", Alg. 9)
  static bool f2_sqrt(const F2 &a, F2 *out) {
    if (f2_is_zero(a)) { *out = a; return true; }
    El pm3_4 = Fp::P(), pm1_2 = Fp::P(), three = Fp::zero(), onei = Fp::zero(); three.l[0] = 3; onei.l[0] = 1;
    Fp::subb(pm3_4, pm3_4, three); Fp::subb(pm1_2, pm1_2, onei);
    for (int k = 0; k < 2; k++) for (int i = 0; i < Fp::L; i++) pm3_4.l[i] = (pm3_4.l[i] >> 1) | (i + 1 < Fp::L ? pm3_4.l[i + 1] << 63 : 0);
    for (int i = 0; i < Fp::L; i++) pm1_2.l[i] = (pm1_2.l[i] >> 1) | (i + 1 < Fp::L ? pm1_2.l[i + 1] << 63 : 0);
    const F2 a1 = f2_pow(a, pm3_4), x0 = f2_mul(a1, a), alpha = f2_mul(a1, x0);
    F2 x;
    if (f2_eq(alpha, f2_neg(f2_one()))) x = f2(Fp::neg(x0.b), x0.a);                            // u * x0
    else x = f2_mul(f2_pow(f2_add(f2_one(), alpha), pm1_2), x0);
    if (!f2_eq(f2_sqr(x), a)) return false;
    *out = x; return true;
  }
  static F2 twist_b() {                                              // y^2 = x^3 + b': b' = b xi (M-twist) or b / xi (D-twist)
    const F2 xi = f2(small(C::XI0), Fp::one()), b = f2(Fp::from32(C::B), Fp::zero());
    return C::MTWIST ? f2_mul(b, xi) : f2_mul(b, f2_inv(xi));
  }
  // "y is the lexicographically largest of {y, -y}" with Fp2 ordered by (c1, c0) (ark-ff QuadExtField Ord; zcash encoding)
  static bool f2_is_largest(const F2 &y) {
    El half = Fp::from32(C::Fq::HALF), t;
    El b = Fp::from_mont(y.b), a = Fp::from_mont(y.a);
    return Fp::subb(t, half, Fp::is_zero(b) ? a : b) != 0;
  }
  static void g2_encode_compressed(const G2 &p, uint8_t *b) {
    constexpr int B = 8 * Fp::L;
    memset(b, 0, 2 * B);
    if (B == 48) {
      if (p.inf) { b[0] = 0xC0; return; }
      El v[2] = {Fp::from_mont(p.x.b), Fp::from_mont(p.x.a)};
      for (int k = 0; k < 2; k++) { uint8_t le[48]; memcpy(le, v[k].l, B); for (int i = 0; i < B; i++) b[k * B + i] = le[B - 1 - i]; }
      b[0] |= 0x80; if (f2_is_largest(p.y)) b[0] |= 0x20;
    } else {
      if (p.inf) { b[2 * B - 1] = 0x40; return; }
      El v[2] = {Fp::from_mont(p.x.a), Fp::from_mont(p.x.b)};
      for (int k = 0; k < 2; k++) memcpy(b + k * B, v[k].l, B);
      if (f2_is_largest(p.y)) b[2 * B - 1] |= 0x80;
    }
  }
  static bool g2_decode_compressed(const uint8_t *b, G2 *out) {
    constexpr int B = 8 * Fp::L;
    El v[2]; bool big, inf;
    if (B == 48) {
      if (!(b[0] & 0x80)) return false;
      inf = b[0] & 0x40; big = b[0] & 0x20;
      for (int k = 0; k < 2; k++) { uint8_t le[48]; for (int i = 0; i < B; i++) le[i] = b[k * B + B - 1 - i]; if (k == 0) le[B - 1] &= 0x1f; memcpy(v[k].l, le, B); }
      { El t = v[0]; v[0] = v[1]; v[1] = t; }                        // -> (c0, c1)
    } else {
      inf = b[2 * B - 1] & 0x40; big = b[2 * B - 1] & 0x80;
      for (int k = 0; k < 2; k++) { uint8_t le[32]; memcpy(le, b + k * B, B); if (k == 1) le[B - 1] &= 0x3f; memcpy(v[k].l, le, B); }
    }
    out->inf = inf;
    if (inf) { out->x = f2_zero(); out->y = f2_zero(); return !big && Fp::is_zero(v[0]) && Fp::is_zero(v[1]); }
    { El t; if (Fp::subb(t, v[0], Fp::P()) == 0 || Fp::subb(t, v[1], Fp::P()) == 0) return false; }   // coordinate >= p
    out->x = f2(Fp::to_mont(v[0]), Fp::to_mont(v[1]));
    F2 y;
    if (!f2_sqrt(f2_add(f2_mul(f2_sqr(out->x), out->x), twist_b()), &y)) return false;
    if (f2_is_largest(y) != big) y = f2_neg(y);
    out->y = y;
    return true;
  }

  // G2 from the `powers_in_g2` bytes of an arkworks URS file
  static bool g2_decode(const uint8_t *b, G2 *out) {
    constexpr int B = 8 * Fp::L;
    El v[4];
    if (B == 48) {                                                     // zcash: x.c1 || x.c0 || y.c1 || y.c0, big-endian
      out->inf = (b[0] & 0x40) != 0;
      for (int k = 0; k < 4; k++) { uint8_t le[48]; for (int i = 0; i < B; i++) le[i] = b[k * B + B - 1 - i]; if (k == 0) le[B - 1] &= 0x1f; memcpy(v[k].l, le, B); }
      out->x = f2(Fp::to_mont(v[1]), Fp::to_mont(v[0])); out->y = f2(Fp::to_mont(v[3]), Fp::to_mont(v[2]));
    } else {                                                           // arkworks: x.c0 || x.c1 || y.c0 || y.c1, little-endian
      out->inf = (b[4 * B - 1] & 0x40) != 0;
      for (int k = 0; k < 4; k++) { uint8_t le[32]; memcpy(le, b + k * B, B); if (k == 3) le[B - 1] &= 0x3f; memcpy(v[k].l, le, B); }
      out->x = f2(Fp::to_mont(v[0]), Fp::to_mont(v[1])); out->y = f2(Fp::to_mont(v[2]), Fp::to_mont(v[3]));
    }
    for (int k = 0; k < 4; k++) if (Fp::geq(v[k], Fp::P())) return false;   // non-canonical coordinate
    return true;
  }
  // y^2 = x^3 + b' on the twist (b' = b xi for the M-type twist of BLS12-381, b / xi for the D-type twist of BN254)
  static bool g2_on_twist(const G2 &q) {
    if (q.inf) return true;
    const F2 b = f2(Fp::from32(C::B), Fp::zero());
    const F2 bt = C::MTWIST ? f2_mul_xi(b) : f2_mul(b, f2_inv(f2_mul_xi(f2_one())));
    return f2_eq(f2_sqr(q.y), f2_add(f2_mul(f2_sqr(q.x), q.x), bt));
  }
};

}  // namespace avrf
