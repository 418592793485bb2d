// field.h -- prime-field arithmetic for the KZG engine, shared by gfx950 device
// code and the host side of the C-ABI library.
//
// Representation ("unsaturated limbs"): an element is N limbs of L bits held in
// 32-bit words, value = sum l[j] * 2^(L*j).  L = 29 (N = 9) for the ~255-bit
// fields, L = 30 (N = 13) for the 381-bit BLS12-381 base field.  The point of the
// slack bits is the multiplier: on gfx950 a 32x32+64 multiply-add
// (v_mad_u64_u32) issues at the same rate as an FP64 FMA (measured,
// tools/microbench/int_rates.hip) but has no carry-in, so with saturated
// 32-bit limbs every partial product needs extra carry instructions (hipcc's
// CIOS for 12 limbs: 288 mads + 295 64-bit adds + 620 moves).  With L-bit limbs
// a whole operand-scanning Montgomery product accumulates into 64-bit columns
// without ever overflowing, so each partial product is exactly one
// v_mad_u64_u32 and nothing else.
//
// Invariant of every value returned by this header ("weak-normal"): limbs
// l[0..N-2] < 2^L, top limb small and non-negative, value in [0, 2p).
// reduce() brings a value to the canonical range [0, p).
//
// Montgomery radix R = 2^(L*N); 4p < R for all four fields, so mul() maps
// weak-normal inputs to a weak-normal output:  (ab + mp)/R < 4p^2/R + p < 2p.
//
// Column bound: a 64-bit column absorbs CAP = 2^(64-2L) - 1 products of two L-bit
// limbs.  A Montgomery row adds two products per column (a_j*b_i and m*p_j), so
// the sliding window is carry-normalised every CAP/2 rows: never for L = 29, N = 9
// (CAP = 63), once (after row 7) for L = 30, N = 13 (CAP = 15).  Inputs must have
// normalised limbs (< 2^L), which every function of this header guarantees.
#pragma once
#include <stdint.h>
#include "curve_constants.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KZG_HD __host__ __device__ __forceinline__
#else
#define KZG_HD inline __attribute__((always_inline))
#endif

namespace kzg {

// c + a*b: one v_mad_u64_u32.  On the device every partial sum is also shown to an EMPTY asm statement as an input:
// LLVM's Reassociate pass only rewrites chains whose interior sums have a single use, so the column stays ONE
// multiply-add chain seeded with the carry instead of a chain from zero joined to the carry by a v_lshl_add_u64
// (one instruction per column).  Nothing is emitted for the statement and, unlike an asm DEFINITION, an asm use
// draws no hazard nops.
static KZG_HD uint64_t mad_wide(uint32_t a, uint32_t b, uint64_t c) {
  const uint64_t r = c + (uint64_t)a * b;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KZG_NO_CHAIN_PIN)
  asm volatile("" ::"v"(r));
#endif
  return r;
}

template <class F>
struct Fe {
  uint32_t l[F::N];
};

template <class F>
struct Field {
  static constexpr int L = F::L;
  static constexpr int N = F::N;
  static constexpr int NW = F::NW;
  static constexpr uint32_t MASK = F::MASK;
  // products of two normalised limbs that fit one 64-bit column (with room for carries)
  static constexpr int CAP = (2 * L >= 64) ? 0 : (int)((1ull << (64 - 2 * L)) - 1);
  static_assert(CAP >= 15, "limb width too large for 64-bit columns");
  // Exact capacity of a product-scanning column: FIT = 2^(64-2L) products of two limbs <= 2^L - 1 plus the carry of
  // the previous column still fit 64 bits:  FIT (2^L - 1)^2 + 2^(64-L) = 2^64 - FIT 2^(L+1) + FIT + 2^(64-L) < 2^64
  // because FIT 2^(L+1) = 2^(65-L) > 2^(64-L) + FIT.  (L = 30: 16 products; L = 29: 64.)
  static constexpr int FIT = CAP + 1;
  using E = Fe<F>;

  // carry-normalise a window of 64-bit columns in place (value unchanged)
  template <int M>
  static KZG_HD void normalize_cols(uint64_t (&w)[M]) {
#pragma unroll
    for (int j = 0; j < M - 1; ++j) {
      const uint64_t c = w[j] >> L;
      w[j] &= (uint64_t)MASK;
      w[j + 1] += c;
    }
  }

  static KZG_HD E zero() {
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = 0;
    return r;
  }
  // Montgomery form of 1
  static KZG_HD E one() {
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = F::R1[j];
    return r;
  }
  // the plain integer 1 (mul(a, raw_one()) converts out of Montgomery form)
  static KZG_HD E raw_one() {
    E r = zero();
    r.l[0] = 1;
    return r;
  }
  static KZG_HD E r2() {
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = F::R2[j];
    return r;
  }

  // canonical saturated little-endian words (NW x 32 bit) -> limbs.  The value
  // is taken as is (no reduction): callers pass values < p.
  static KZG_HD E from_words(const uint32_t* w) {
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int bit = j * L;
      const int k = bit >> 5, sh = bit & 31;
      uint64_t lo = (k < NW) ? w[k] : 0u;
      uint64_t hi = (k + 1 < NW) ? w[k + 1] : 0u;
      r.l[j] = (uint32_t)(((hi << 32) | lo) >> sh) & MASK;
    }
    return r;
  }
  // limbs -> canonical words.  Requires a canonical element (reduce() first).
  static KZG_HD void to_words(const E& a, uint32_t* w) {
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      // word k covers bits [32k, 32k+32)
      const int j0 = (32 * k) / L;           // first limb touching the word
      const int off0 = 32 * k - j0 * L;      // bit offset inside limb j0
      uint64_t acc = (uint64_t)a.l[j0] >> off0;
      int have = L - off0;
      if (j0 + 1 < N) {
        acc |= (uint64_t)a.l[j0 + 1] << have;
        have += L;
      }
      if (have < 32 && j0 + 2 < N) acc |= (uint64_t)a.l[j0 + 2] << have;
      w[k] = (uint32_t)acc;
    }
  }

  // ---- multiplication: product scanning (column by column) --------------------------------------
  // Every simple integer instruction of gfx950 issues at the rate of v_mad_u64_u32 (measured,
  // tools/microbench/int_rates.hip), so what counts is the NUMBER of instructions.  Column k of the
  // Montgomery product collects  sum_(i+j=k) a_i b_j + sum_(i+j=k) m_i p_j  on top of the carry of
  // column k-1: the carry is the initial value of the column's multiply-add chain (no separate
  // 64-bit addition per column), the low half of the columns yields the quotient digits m_k, the
  // high half the result limbs.  (hipcc re-associates each column's sum so that the incoming carry is
  // added last -- one v_lshl_add_u64 per column remains; forcing a single chain with inline-asm
  // multiply-adds was worth another 0.4 % and was not kept)
  //
  // Column capacity: products of two L-bit limbs are < 2^(2L); a 64-bit column holds FIT of them
  // next to the carry.  Columns with more products (L = 30, N = 13: the 9 middle ones) are split:
  // the partial sum is cut into its low L bits (which stay in the chain) and its upper part (added
  // to the outgoing carry).  `UA` = how many units one a*b product may take (operands with limbs
  // up to 2^(L+1) from add_lazy count 2 or 4).
  template <int UNITS_AB = 1>
  static KZG_HD E mul(const E& a, const E& b) {
    uint32_t m[N];
    E r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
      const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
      const int nab = hi - lo + 1;                       // a*b products of this column
      const int nmp = k < N ? k + 1 : nab;               // m*p products (incl. m_k p_0 in the low half)
      const bool split = nab * UNITS_AB + nmp > FIT;
      uint64_t upper = 0;
#pragma unroll
      for (int i = lo; i <= hi; ++i) acc = mad_wide(a.l[i], b.l[k - i], acc);
      if (split) { upper = acc >> L; acc &= (uint64_t)MASK; }
#pragma unroll
      for (int i = lo; i <= hi; ++i)
        if (i < k || k >= N) acc = mad_wide(m[i], F::P[k - i], acc);
      if (k < N) {
        m[k] = ((uint32_t)acc * F::N0) & MASK;
        acc = mad_wide(m[k], F::P[0], acc);
      } else {
        r.l[k - N] = (uint32_t)acc & MASK;
      }
      acc >>= L;
      if (split) acc += upper;
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
  }
  // Montgomery square: off-diagonal products once, against the doubled operand.
  static KZG_HD E sqr(const E& a) {
    uint32_t m[N], a2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) a2[j] = a.l[j] << 1;
    E r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
      const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
      const int nab = hi - lo + 1;                       // units: a doubled product counts twice
      const int nmp = k < N ? k + 1 : nab;
      const bool split = nab + nmp > FIT;
      uint64_t upper = 0;
#pragma unroll
      for (int i = lo; i <= hi; ++i) {
        const int j = k - i;
        if (i < j) acc = mad_wide(a.l[i], a2[j], acc);
        else if (i == j) acc = mad_wide(a.l[i], a.l[i], acc);
      }
      if (split) { upper = acc >> L; acc &= (uint64_t)MASK; }
#pragma unroll
      for (int i = lo; i <= hi; ++i)
        if (i < k || k >= N) acc = mad_wide(m[i], F::P[k - i], acc);
      if (k < N) {
        m[k] = ((uint32_t)acc * F::N0) & MASK;
        acc = mad_wide(m[k], F::P[0], acc);
      } else {
        r.l[k - N] = (uint32_t)acc & MASK;
      }
      acc >>= L;
      if (split) acc += upper;
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
  }

  // a*b + c*d with ONE Montgomery reduction (weak-normal in and out: (4p^2+4p^2)/R + p < 2p
  // needs 8p < R, true for all four fields).
  static KZG_HD E mul2(const E& a, const E& b, const E& c, const E& d) {
    static_assert(F::BITS + 3 <= L * N, "mul2 needs 8p < R");
    uint32_t m[N];
    E r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
      const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
      const int nab = hi - lo + 1;
      const int nmp = k < N ? k + 1 : nab;
      const bool split1 = 2 * nab > FIT;                 // between a*b and c*d
      const bool split2 = (split1 ? nab : 2 * nab) + nmp > FIT;       // before m*p
      uint64_t upper = 0;
#pragma unroll
      for (int i = lo; i <= hi; ++i) acc = mad_wide(a.l[i], b.l[k - i], acc);
      if (split1) { upper = acc >> L; acc &= (uint64_t)MASK; }
#pragma unroll
      for (int i = lo; i <= hi; ++i) acc = mad_wide(c.l[i], d.l[k - i], acc);
      if (split2) { upper += acc >> L; acc &= (uint64_t)MASK; }
#pragma unroll
      for (int i = lo; i <= hi; ++i)
        if (i < k || k >= N) acc = mad_wide(m[i], F::P[k - i], acc);
      if (k < N) {
        m[k] = ((uint32_t)acc * F::N0) & MASK;
        acc = mad_wide(m[k], F::P[0], acc);
      } else {
        r.l[k - N] = (uint32_t)acc & MASK;
      }
      acc >>= L;
      if (split1 || split2) acc += upper;
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
  }

  // sum_{t<K} a[t]*b[t] with ONE Montgomery reduction: K*N^2 + N^2 + N multiply-adds instead of K*(2N^2 + N) --
  // the linear combinations of KZG.open (kzg.py:148-150) and of the prover's r(X).  Weak-normal (< 2p) in and out:
  // (K*4p^2 + m*p)/R < 2p needs 4K*p <= R.  A column takes the a*b products of one term after the other and is cut
  // (low L bits stay in the chain, the rest joins the outgoing carry) whenever the next group would pass its capacity.
  template <int K>
  static KZG_HD E dot(const E* a, const E* b) {
    static_assert(K >= 1 && F::BITS + 2 + (K > 8 ? 4 : K > 4 ? 3 : K > 2 ? 2 : K > 1 ? 1 : 0) <= L * N, "dot needs 4K*p <= R");
    uint32_t m[N];
    E r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N - 1; ++k) {
      const int lo = k < N ? 0 : k - N + 1, hi = k < N ? k : N - 1;
      const int nab = hi - lo + 1;                       // a*b products of ONE term in this column
      const int nmp = k < N ? k + 1 : nab;
      uint64_t upper = 0;
      int used = 1;                                      // the incoming carry (< 2^(64-L)) counts as one product
#pragma unroll
      for (int t = 0; t < K; ++t) {
        if (used + nab > FIT) { upper += acc >> L; acc &= (uint64_t)MASK; used = 1; }
#pragma unroll
        for (int i = lo; i <= hi; ++i) acc = mad_wide(a[t].l[i], b[t].l[k - i], acc);
        used += nab;
      }
      if (used + nmp > FIT) { upper += acc >> L; acc &= (uint64_t)MASK; used = 1; }
#pragma unroll
      for (int i = lo; i <= hi; ++i)
        if (i < k || k >= N) acc = mad_wide(m[i], F::P[k - i], acc);
      if (k < N) {
        m[k] = ((uint32_t)acc * F::N0) & MASK;
        acc = mad_wide(m[k], F::P[0], acc);
      } else {
        r.l[k - N] = (uint32_t)acc & MASK;
      }
      acc >>= L;
      acc += upper;
    }
    r.l[N - 1] = (uint32_t)acc;
    return r;
  }

  // a + b, weak-normal in and out
  static KZG_HD E add(const E& a, const E& b) {
    E s, t;
    uint32_t cs = 0;
    int32_t ct = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const uint32_t x = a.l[j] + b.l[j];
      const uint32_t sj = x + cs;
      const int32_t tj = (int32_t)(x - F::P2[j]) + ct;
      if (j < N - 1) {
        cs = sj >> L;
        s.l[j] = sj & MASK;
        ct = tj >> L;
        t.l[j] = (uint32_t)tj & MASK;
      } else {
        s.l[j] = sj;
        t.l[j] = (uint32_t)tj;
      }
    }
    const bool neg = (int32_t)t.l[N - 1] < 0;
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = neg ? s.l[j] : t.l[j];
    return r;
  }
  // a - b, weak-normal in and out
  static KZG_HD E sub(const E& a, const E& b) {
    E u, t;
    int32_t cu = 0, ct = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int32_t x = (int32_t)a.l[j] - (int32_t)b.l[j];
      const int32_t tj = x + ct;
      const int32_t uj = x + (int32_t)F::P2[j] + cu;
      if (j < N - 1) {
        ct = tj >> L;
        t.l[j] = (uint32_t)tj & MASK;
        cu = uj >> L;
        u.l[j] = (uint32_t)uj & MASK;
      } else {
        t.l[j] = (uint32_t)tj;
        u.l[j] = (uint32_t)uj;
      }
    }
    const bool neg = (int32_t)t.l[N - 1] < 0;
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = neg ? u.l[j] : t.l[j];
    return r;
  }
  // ---- lazy forms (NTT butterflies): no value reduction, limbs may exceed 2^L --------------
  // a + b limb-wise.  Caller keeps limbs below 2^32.
  static KZG_HD E add_lazy(const E& a, const E& b) {
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = a.l[j] + b.l[j];
    return r;
  }
  // a - b + 4p limb-wise, for a normalised b < 2p (e.g. a mul() output): every limb of the
  // redistributed constant P4R dominates the matching limb of b, so no limb goes negative.
  static KZG_HD E sub_lazy4(const E& a, const E& b) {
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = a.l[j] + (F::P4R[j] - b.l[j]);
    return r;
  }
  // carry propagation: limbs back below 2^L (the top limb absorbs the excess), value unchanged
  static KZG_HD E carry(const E& a) {
    E r;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      const uint32_t t = a.l[j] + c;
      r.l[j] = t & MASK;
      c = t >> L;
    }
    r.l[N - 1] = a.l[N - 1] + c;
    return r;
  }

  // ---- lazy forms with normalised limbs (the MSM inner loop) --------------------------------
  // mul / sqr / mul2 only need normalised limbs and a product of the operand VALUES below R*p
  // (R/p > 160 for the two base fields), not weak-normal operands.  So between multiplications
  // a difference is formed as a - b + K*p (K*p >= b, nothing else to decide) and carried, with
  // no conditional subtraction and no signed borrow chain: value in [0, a + K*p), limbs < 2^L.
  template <int K>
  static KZG_HD const uint32_t* pkr() {
    static_assert(K == 2 || K == 4 || K == 6 || K == 8, "no redistributed constant for this multiple of p");
    if constexpr (K == 2) return F::P2R;
    else if constexpr (K == 4) return F::P4R;
    else if constexpr (K == 6) return F::P6R;
    else return F::P8R;
  }
  // a - b + K*p for normalised a, b with b <= K*p.  Every lower limb of the redistributed K*p
  // dominates a normalised limb; the top limb may wrap below zero on the way and is put right by
  // the incoming carry (arithmetic mod 2^32, true value non-negative).
  template <int K>
  static KZG_HD E sub_carry(const E& a, const E& b) {
    const uint32_t* kp = pkr<K>();
    E r;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      const uint32_t t = a.l[j] + (kp[j] - b.l[j]) + c;
      r.l[j] = t & MASK;
      c = t >> L;
    }
    r.l[N - 1] = a.l[N - 1] + (kp[N - 1] - b.l[N - 1]) + c;
    return r;
  }
  // (neg ? 2p - a : a) - b + K*p for normalised a < 2p and b <= K*p: the conditional negation of a is folded
  // into the carried difference (2 instructions per limb instead of a carried negation of its own).
  template <int K>
  static KZG_HD E sub_carry_cneg(const E& a, bool neg, const E& b) {
    const uint32_t* kp = pkr<K>();
    E r;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const uint32_t aj = neg ? F::P2R[j] - a.l[j] : a.l[j];
      const uint32_t t = aj + (kp[j] - b.l[j]) + c;
      if (j < N - 1) { r.l[j] = t & MASK; c = t >> L; } else { r.l[j] = t; }
    }
    return r;
  }
  // a + b + b for normalised operands: value a + 2b, limbs normalised
  static KZG_HD E add_twice_carry(const E& a, const E& b) {
    E r;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      const uint32_t t = a.l[j] + 2 * b.l[j] + c;
      r.l[j] = t & MASK;
      c = t >> L;
    }
    r.l[N - 1] = a.l[N - 1] + 2 * b.l[N - 1] + c;
    return r;
  }

  static KZG_HD E dbl(const E& a) { return add(a, a); }
  static KZG_HD E neg(const E& a) { return sub(zero(), a); }

  // [0, 2p) -> [0, p)
  static KZG_HD E reduce(const E& a) {
    E t;
    int32_t ct = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int32_t tj = (int32_t)a.l[j] - (int32_t)F::P[j] + ct;
      if (j < N - 1) {
        ct = tj >> L;
        t.l[j] = (uint32_t)tj & MASK;
      } else {
        t.l[j] = (uint32_t)tj;
      }
    }
    const bool neg = (int32_t)t.l[N - 1] < 0;
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = neg ? a.l[j] : t.l[j];
    return r;
  }
  // [0, 2^(L*N)) with normalised limbs -> [0, p): the canonical value of a lazily accumulated sum
  // (NTT butterflies: up to 49p) without a multiplication.  Quotient estimate from the top limb:
  // q = floor(top * floor(2^52 / (ptop + 1)) / 2^52) with ptop = top limb of p never exceeds
  // floor(x / p) and falls short of it by at most 1 (the estimate loses < 2^-13 + rounding), so
  // x - q*p is in [0, 2p) and one conditional subtraction finishes.
  static KZG_HD E reduce_wide(const E& a) {
    constexpr uint64_t PTOP1 = (uint64_t)F::P[N - 1] + 1;
    constexpr uint64_t M = (1ull << 52) / PTOP1;
    static_assert(PTOP1 > (1ull << 20), "top limb of p too small for the 52-bit reciprocal");
    const uint32_t q = (uint32_t)(((uint64_t)a.l[N - 1] * M) >> 52);
    E r;
    int64_t c = 0;
#pragma unroll
    for (int j = 0; j < N - 1; ++j) {
      const int64_t t = (int64_t)a.l[j] - (int64_t)((uint64_t)q * F::P[j]) + c;
      r.l[j] = (uint32_t)t & MASK;
      c = t >> L;
    }
    r.l[N - 1] = (uint32_t)((int64_t)a.l[N - 1] - (int64_t)((uint64_t)q * F::P[N - 1]) + c);
    return reduce(r);
  }
  static KZG_HD bool is_zero(const E& a) {
    const E r = reduce(a);
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) acc |= r.l[j];
    return acc == 0;
  }
  static KZG_HD bool eq(const E& a, const E& b) { return is_zero(sub(a, b)); }
  // zero test without the reduction: a weak-normal value is 0 mod p iff it is the integer 0 or
  // the integer p, and normalised limbs represent an integer uniquely
  static KZG_HD bool is_zero_weak(const E& a) {
    uint32_t z = 0, q = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) { z |= a.l[j]; q |= a.l[j] ^ F::P[j]; }
    return z == 0 || q == 0;
  }
  // 2p - a for weak-normal a: normalised limbs, value in (0, 2p] (fine as a mul operand)
  static KZG_HD E neg_weak(const E& a) {
    E r;
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int32_t t = (int32_t)F::P2[j] - (int32_t)a.l[j] + c;
      if (j < N - 1) { c = t >> L; r.l[j] = (uint32_t)t & MASK; } else { r.l[j] = (uint32_t)t; }
    }
    return r;
  }
  // flag ? p - a : a  for CANONICAL a (table coordinates): result weak-normal in [0, p]
  static KZG_HD E cneg_canonical(const E& a, bool flag) {
    E r;
    int32_t c = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int32_t t = (int32_t)F::P[j] - (int32_t)a.l[j] + c;
      uint32_t v;
      if (j < N - 1) { c = t >> L; v = (uint32_t)t & MASK; } else { v = (uint32_t)t; }
      r.l[j] = flag ? v : a.l[j];
    }
    return r;
  }

  static KZG_HD E to_mont(const E& a) { return mul(a, r2()); }
  static KZG_HD E from_mont(const E& a) { return reduce(mul(a, raw_one())); }

  static KZG_HD E select(bool c, const E& a, const E& b) {
    E r;
#pragma unroll
    for (int j = 0; j < N; ++j) r.l[j] = c ? a.l[j] : b.l[j];
    return r;
  }

  // a^e for a Montgomery-form a; e given as `nw` saturated 32-bit words.
  // Not unrolled: used off the hot path (table construction, host finishing).
  static KZG_HD E pow_words(const E& a, const uint32_t* e, int nw) {
    E r = one();
    for (int k = nw - 1; k >= 0; --k) {
      for (int bit = 31; bit >= 0; --bit) {
        r = mul(r, r);
        if ((e[k] >> bit) & 1u) r = mul(r, a);
      }
    }
    return r;
  }
  // a^(p-2) (Montgomery form in and out); inv(0) = 0
  static KZG_HD E inv(const E& a) {
    uint32_t e[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) e[k] = F::PW[k];
    uint32_t borrow = 2;  // e = p - 2
    for (int k = 0; k < NW && borrow; ++k) {
      const uint32_t old = e[k];
      e[k] = old - borrow;
      borrow = old < borrow ? 1u : 0u;
    }
    return pow_words(a, e, NW);
  }
};

}  // namespace kzg
