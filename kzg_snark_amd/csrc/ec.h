// ec.h -- G1 group law for y^2 = x^3 + b (a = 0) in extended Jacobian "XYZZ"
// coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), shared by device kernels and
// the host finishing code.  This is the engine's own group law: the reference
// delegates to py_ecc's homogeneous projective add/double (bound at
// kzg.py:27-35, used at kzg.py:115-116); results agree as affine points.
//
// Infinity is ZZ == 0 (all four coordinates zero).  All edge cases are exact:
// P + O, O + P, P + P (falls through to doubling), P + (-P) = O, 2P with y = 0.
// Formulas: madd-2008-s / add-2008-s / dbl-2008-s-1 / mdbl-2008-s-1 (EFD).
#pragma once
#include "field.h"

namespace kzg {

template <class C>
struct Affine {
  Fe<typename C::Fp> x, y;   // Montgomery form
  bool inf;
};

template <class C>
struct XYZZ {
  Fe<typename C::Fp> x, y, zz, zzz;
};

template <class C>
struct Ec {
  using F = typename C::Fp;
  using Fd = Field<F>;
  using E = Fe<F>;
  using A = Affine<C>;
  using P = XYZZ<C>;

  static KZG_HD P infinity() {
    P r;
    r.x = Fd::zero(); r.y = Fd::zero(); r.zz = Fd::zero(); r.zzz = Fd::zero();
    return r;
  }
  static KZG_HD bool is_inf(const P& p) { return Fd::is_zero(p.zz); }

  static KZG_HD P from_affine(const A& a) {
    if (a.inf) return infinity();
    P r;
    r.x = a.x; r.y = a.y; r.zz = Fd::one(); r.zzz = Fd::one();
    return r;
  }

  // 2 * (affine point)
  static KZG_HD P dbl_affine(const E& x, const E& y) {
    if (Fd::is_zero(y)) return infinity();
    const E U = Fd::dbl(y);
    const E V = Fd::sqr(U);
    const E W = Fd::mul(U, V);
    const E S = Fd::mul(x, V);
    const E xx = Fd::sqr(x);
    const E M = Fd::add(Fd::dbl(xx), xx);
    P r;
    r.x = Fd::sub(Fd::sqr(M), Fd::dbl(S));
    r.y = Fd::sub(Fd::mul(M, Fd::sub(S, r.x)), Fd::mul(W, y));
    r.zz = V;
    r.zzz = W;
    return r;
  }

  static KZG_HD P dbl(const P& p) {
    if (is_inf(p) || Fd::is_zero(p.y)) return infinity();
    const E U = Fd::dbl(p.y);
    const E V = Fd::sqr(U);
    const E W = Fd::mul(U, V);
    const E S = Fd::mul(p.x, V);
    const E xx = Fd::sqr(p.x);
    const E M = Fd::add(Fd::dbl(xx), xx);
    P r;
    r.x = Fd::sub(Fd::sqr(M), Fd::dbl(S));
    r.y = Fd::sub(Fd::mul(M, Fd::sub(S, r.x)), Fd::mul(W, p.y));
    r.zz = Fd::mul(V, p.zz);
    r.zzz = Fd::mul(W, p.zzz);
    return r;
  }

  // acc + (x2, y2) with the affine operand known to be finite
  static KZG_HD P madd(const P& a, const E& x2, const E& y2) {
    if (is_inf(a)) {
      P r;
      r.x = x2; r.y = y2; r.zz = Fd::one(); r.zzz = Fd::one();
      return r;
    }
    const E U2 = Fd::mul(x2, a.zz);
    const E S2 = Fd::mul(y2, a.zzz);
    const E Pp = Fd::sub(U2, a.x);
    const E R = Fd::sub(S2, a.y);
    if (Fd::is_zero(Pp)) {
      if (Fd::is_zero(R)) return dbl_affine(x2, y2);
      return infinity();
    }
    const E PP = Fd::sqr(Pp);
    const E PPP = Fd::mul(Pp, PP);
    const E Q = Fd::mul(a.x, PP);
    P r;
    r.x = Fd::sub(Fd::sub(Fd::sqr(R), PPP), Fd::dbl(Q));
    r.y = Fd::mul2(R, Fd::sub(Q, r.x), Fd::neg(a.y), PPP);     // R*(Q - X3) - Y1*PPP, one reduction
    r.zz = Fd::mul(a.zz, PP);
    r.zzz = Fd::mul(a.zzz, PPP);
    return r;
  }

  // madd for an accumulator known to be finite; `finite` is cleared when the sum is infinity.
  // The hot loop of the MSM: tracks infinity in a flag instead of testing ZZ every time, and
  // forms every difference lazily (Field::sub_carry: a - b + K*p, carried, never reduced).
  // Value ranges, as multiples of p (limbs always normalised):
  //   in:  X1 < 8, Y1, ZZ1, ZZZ1 < 2 (products), x2 < 1, y2 <= 1
  //   Pp = U2 - X1 + 8p < 10     R = S2 - Y1 + 2p < 4      PP, PPP, Q, RR < 2 (products)
  //   X3 = RR - (PPP + 2Q) + 6p < 8      Q - X3 + 8p < 10      Y3 = mul2(...) < 2
  // Largest operand products: Pp^2 < 100 p^2, mul2: 4*10 + 2*2 = 44 p^2 -- below R*p for both
  // base fields (R/p = 168 for BN254, 630 for BLS12-381).  Zero tests go through a product, which
  // is weak-normal: Pp = 0 mod p  <=>  PP = 0 mod p (p prime), likewise R and RR.
  // Consumers of the accumulator (add, dbl, to_affine) use X only as a mul/sqr operand.
  // `neg`: add -(x2, y2) instead.  y2 is the table's canonical coordinate; its sign enters the lazy
  // difference R = (+-S2) - Y1 (Field::sub_carry_cneg) instead of a carried negation of y2 of its own.
  static KZG_HD P madd_finite(const P& a, const E& x2, const E& y2, bool neg, bool& finite) {
    const E U2 = Fd::mul(x2, a.zz);
    const E S2 = Fd::mul(y2, a.zzz);
    const E Pp = Fd::template sub_carry<8>(U2, a.x);
    const E R = Fd::template sub_carry_cneg<2>(S2, neg, a.y);      // S2 < 2p: 2p - S2 in (0, 2p], R < 4p as before
    const E PP = Fd::sqr(Pp);
    if (Fd::is_zero_weak(PP)) {
      if (Fd::is_zero_weak(Fd::sqr(R))) {
        const P d = dbl_affine(x2, Fd::cneg_canonical(y2, neg));
        finite = !is_inf(d);
        return d;
      }
      finite = false;
      return infinity();
    }
    const E PPP = Fd::mul(Pp, PP);
    const E Q = Fd::mul(a.x, PP);
    P r;
    r.x = Fd::template sub_carry<6>(Fd::sqr(R), Fd::add_twice_carry(PPP, Q));
    r.y = Fd::mul2(R, Fd::template sub_carry<8>(Q, r.x), Fd::neg_weak(a.y), PPP);   // R*(Q - X3) - Y1*PPP
    r.zz = Fd::mul(a.zz, PP);
    r.zzz = Fd::mul(a.zzz, PPP);
    return r;
  }

  static KZG_HD P add(const P& a, const P& b) {
    if (is_inf(a)) return b;
    if (is_inf(b)) return a;
    const E U1 = Fd::mul(a.x, b.zz);
    const E U2 = Fd::mul(b.x, a.zz);
    const E S1 = Fd::mul(a.y, b.zzz);
    const E S2 = Fd::mul(b.y, a.zzz);
    const E Pp = Fd::sub(U2, U1);
    const E R = Fd::sub(S2, S1);
    if (Fd::is_zero(Pp)) {
      if (Fd::is_zero(R)) return dbl(a);
      return infinity();
    }
    const E PP = Fd::sqr(Pp);
    const E PPP = Fd::mul(Pp, PP);
    const E Q = Fd::mul(U1, PP);
    P r;
    r.x = Fd::sub(Fd::sub(Fd::sqr(R), PPP), Fd::dbl(Q));
    r.y = Fd::mul2(R, Fd::sub(Q, r.x), Fd::neg(S1), PPP);
    r.zz = Fd::mul(Fd::mul(a.zz, b.zz), PP);
    r.zzz = Fd::mul(Fd::mul(a.zzz, b.zzz), PPP);
    return r;
  }

  // XYZZ -> affine (one field inversion; host finishing and SRS table build)
  static KZG_HD A to_affine(const P& p) {
    A r;
    if (is_inf(p)) {
      r.x = Fd::zero(); r.y = Fd::zero(); r.inf = true;
      return r;
    }
    const E u = Fd::inv(Fd::mul(p.zz, p.zzz));          // Z^-5
    r.x = Fd::mul(p.x, Fd::mul(u, p.zzz));              // X * Z^-2
    r.y = Fd::mul(p.y, Fd::mul(u, p.zz));               // Y * Z^-3
    r.inf = false;
    return r;
  }

  // y^2 == x^3 + b (Montgomery-form inputs)
  static KZG_HD bool on_curve(const E& x, const E& y) {
    E b;
#pragma unroll
    for (int j = 0; j < F::N; ++j) b.l[j] = C::B_MONT[j];
    return Fd::eq(Fd::sqr(y), Fd::add(Fd::mul(Fd::sqr(x), x), b));
  }
};

}  // namespace kzg
