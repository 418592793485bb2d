// Host-only test shim: exposes kzg_snark_amd/csrc/field.h (the same header the
// gfx950 kernels use) through a tiny C interface so tests/test_field_host.py can
// check it against Python integers without a GPU.  Test infrastructure only.
#include "../../kzg_snark_amd/csrc/field.h"
#include "../../kzg_snark_amd/csrc/ec.h"
using namespace kzg;

template <class F>
static void op(int which, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  using Fd = Field<F>;
  auto A = Fd::to_mont(Fd::from_words(a));
  auto B = Fd::to_mont(Fd::from_words(b));
  typename Fd::E R;
  switch (which) {
    case 0: R = Fd::mul(A, B); break;
    case 1: R = Fd::add(A, B); break;
    case 2: R = Fd::sub(A, B); break;
    case 3: R = Fd::inv(A); break;
    case 4: R = Fd::neg(A); break;
    case 5: R = Fd::dbl(A); break;
    case 6: {  // chain: ((a+b)*(a-b) + a) * b - mixes weak-normal values
      auto s = Fd::add(A, B), d = Fd::sub(A, B);
      R = Fd::sub(Fd::mul(Fd::add(Fd::mul(s, d), A), B), Fd::neg(B));
      break;
    }
    case 7: R = Fd::sqr(A); break;
    case 8: R = Fd::mul2(A, B, Fd::add(A, B), Fd::sub(B, A)); break;   // a*b + (a+b)*(b-a)
    case 9: R = Fd::sqr(Fd::sub(A, B)); break;
    case 10: R = Fd::neg_weak(Fd::sub(A, B)); break;                       // -(a - b)
    case 11: R = Fd::cneg_canonical(Fd::reduce(A), true); break;            // -a
    case 12: R = Fd::cneg_canonical(Fd::reduce(A), false); break;           // a
    case 13: R = Fd::is_zero_weak(Fd::sub(A, B)) == Fd::is_zero(Fd::sub(A, B))
                     ? (Fd::is_zero_weak(Fd::sub(A, B)) ? Fd::one() : Fd::zero()) : Fd::add(Fd::one(), Fd::one()); break;
    case 14: {  // lazy forms at the top of their ranges: (a + 2b) * (a - b)
      auto x3 = Fd::template sub_carry<6>(A, Fd::add_twice_carry(B, A));    // -a - b + 6p  (< 8p)
      auto d = Fd::template sub_carry<8>(B, x3);                            // a + 2b + 2p (< 10p)
      R = Fd::mul(d, Fd::template sub_carry<2>(A, B));
      break;
    }
    case 15: R = Fd::sqr(Fd::template sub_carry<8>(A, Fd::template sub_carry<6>(B, A))); break;   // (2a - b)^2
    case 16: R = Fd::mul2(Fd::template sub_carry<2>(A, B), Fd::template sub_carry<8>(A, Fd::template sub_carry<6>(B, A)),
                          Fd::neg_weak(A), B); break;                       // (a - b)(2a - b) - a*b
    case 17: {  // reduce_wide on a lazily accumulated value: 16a + 15b (< 62p), and on small ones
      auto x = A;
      for (int i = 0; i < 4; ++i) x = Fd::carry(Fd::add_lazy(Fd::add_lazy(x, x), B));
      R = Fd::reduce_wide(x);
      if (!(Fd::eq(Fd::reduce_wide(A), A) && Fd::eq(Fd::reduce_wide(Fd::carry(Fd::add_lazy(A, B))), Fd::add(A, B))))
        R = Fd::zero();
      // the result must be canonical: reducing again changes nothing, limb for limb
      auto again = Fd::reduce(R);
      for (int j = 0; j < F::N; ++j) if (again.l[j] != R.l[j]) R = Fd::zero();
      break;
    }
    case 18: R = Fd::mul(Fd::template sub_carry_cneg<2>(Fd::mul(A, Fd::one()), true, Fd::mul(B, Fd::one())), Fd::one()); break;    // -a - b
    case 19: R = Fd::mul(Fd::template sub_carry_cneg<2>(Fd::mul(A, Fd::one()), false, Fd::mul(B, Fd::one())), Fd::one()); break;   //  a - b
    case 20: {  // dot<6>: a*b + (a+b)(a-b) + b*b + a*a + (a-b)*b + (a+b)*a, weak-normal operands, one reduction
      const typename Fd::E x[6] = {A, Fd::add(A, B), B, A, Fd::sub(A, B), Fd::add(A, B)};
      const typename Fd::E y[6] = {B, Fd::sub(A, B), B, A, B, A};
      R = Fd::template dot<6>(x, y);
      break;
    }
    case 21: {  // dot<1> == mul, dot<2> == mul2, dot<3>
      const typename Fd::E x[3] = {A, Fd::add(A, B), Fd::neg_weak(B)};
      const typename Fd::E y[3] = {B, Fd::sub(B, A), A};
      R = Fd::add(Fd::template dot<3>(x, y), Fd::add(Fd::template dot<1>(x, y), Fd::template dot<2>(x, y)));
      if (!Fd::eq(Fd::template dot<1>(x, y), Fd::mul(A, B)) || !Fd::eq(Fd::template dot<2>(x, y), Fd::mul2(x[0], y[0], x[1], y[1])))
        R = Fd::zero();
      break;
    }
    default: R = Fd::zero();
  }
  Fd::to_words(Fd::from_mont(R), out);
}

extern "C" int shim_field_op(int field, int which, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  switch (field) {
    case 0: op<BnFr>(which, a, b, out); return 0;
    case 1: op<BnFp>(which, a, b, out); return 0;
    case 2: op<BlsFr>(which, a, b, out); return 0;
    case 3: op<BlsFp>(which, a, b, out); return 0;
  }
  return -1;
}
extern "C" int shim_is_zero(int field, const uint32_t* a) {
  switch (field) {
    case 0: return Field<BnFr>::is_zero(Field<BnFr>::from_words(a));
    case 1: return Field<BnFp>::is_zero(Field<BnFp>::from_words(a));
    case 2: return Field<BlsFr>::is_zero(Field<BlsFr>::from_words(a));
    case 3: return Field<BlsFp>::is_zero(Field<BlsFp>::from_words(a));
  }
  return -1;
}

// ---- ec.h ---------------------------------------------------------------------------------
template <class C>
static int ec_op(int which, const uint32_t* p1, int inf1, const uint32_t* p2, int inf2, uint32_t* out, int* out_inf) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  using E = Ec<C>;
  auto load = [](const uint32_t* w, int inf) {
    Affine<C> a;
    a.inf = inf != 0;
    a.x = inf ? Fd::zero() : Fd::to_mont(Fd::from_words(w));
    a.y = inf ? Fd::zero() : Fd::to_mont(Fd::from_words(w + F::NW));
    return a;
  };
  const Affine<C> A = load(p1, inf1), B = load(p2, inf2);
  XYZZ<C> R;
  switch (which) {
    case 0: R = B.inf ? E::from_affine(A) : E::madd(E::from_affine(A), B.x, B.y); break;
    case 1: R = E::add(E::from_affine(A), E::from_affine(B)); break;
    case 2: R = E::dbl(E::from_affine(A)); break;
    case 4: {  // the MSM inner loop form: flag-tracked accumulator, A + B - B + B + A (passes through infinity? no: through A)
      bool fin = true;
      XYZZ<C> t = E::from_affine(A);
      auto step = [&](const Affine<C>& q, bool negate) {      // the loop body of msm_accumulate_kernel
        const auto y = Fd::reduce(q.y);
        const auto x = Fd::reduce(q.x);
        if (!fin) { t.x = x; t.y = Fd::cneg_canonical(y, negate); fin = true; }
        else {
          t = E::madd_finite(t, x, y, negate, fin);
          if (!fin) { t.zz = Fd::one(); t.zzz = Fd::one(); }
        }
      };
      step(B, false); step(B, true); step(A, true);      // A + B - B - A = O
      if (fin) return -2;
      step(B, false); step(B, false); step(A, false);    // O + B + B (doubling) + A
      R = fin ? t : E::infinity();
      break;
    }
    case 5: {  // a long flag-tracked chain: the lazily reduced X of madd_finite over many steps
      bool fin = true;
      XYZZ<C> t = E::from_affine(A);
      for (int i = 0; i < 48; ++i) {
        const Affine<C>& q = (i % 3 == 0) ? A : B;
        const bool negate = (i % 5 == 4);
        const auto y = Fd::reduce(q.y);
        const auto x = Fd::reduce(q.x);
        if (!fin) { t.x = x; t.y = Fd::cneg_canonical(y, negate); fin = true; }
        else {
          t = E::madd_finite(t, x, y, negate, fin);
          if (!fin) { t.zz = Fd::one(); t.zzz = Fd::one(); }
        }
      }
      // the consumers of such an accumulator: a full addition and a doubling
      R = fin ? E::dbl(E::add(t, E::from_affine(B))) : E::infinity();
      break;
    }
    case 3: {  // ((A + B) + B) + A with a projective accumulator
      XYZZ<C> t = E::madd(E::from_affine(A), B.x, B.y);
      t = E::madd(t, B.x, B.y);
      R = E::add(t, E::from_affine(A));
      break;
    }
    default: return -1;
  }
  const Affine<C> r = E::to_affine(R);
  *out_inf = r.inf ? 1 : 0;
  if (!r.inf) {
    Fd::to_words(Fd::from_mont(r.x), out);
    Fd::to_words(Fd::from_mont(r.y), out + F::NW);
  }
  return 0;
}
extern "C" int shim_ec_op(int curve, int which, const uint32_t* p1, int inf1, const uint32_t* p2, int inf2,
                          uint32_t* out, int* out_inf) {
  if (curve == 0) return ec_op<Bn254>(which, p1, inf1, p2, inf2, out, out_inf);
  if (curve == 1) return ec_op<Bls12_381>(which, p1, inf1, p2, inf2, out, out_inf);
  return -1;
}
