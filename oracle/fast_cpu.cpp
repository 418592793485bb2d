// oracle/fast_cpu.cpp -- TEST / BENCH INFRASTRUCTURE, NOT PRODUCT CODE.
//
// The "fair" CPU baseline of BASELINE.md section 3 / SURVEY.md section 8d-2: what a competent
// multi-core CPU implementation of the same job costs, timed beside the GPU figure by bench.py
// (`cpu_baseline_optimised`).  It is NOT a restatement of the reference's algorithms (that is
// oracle/kzg_oracle.c, the recursive FFT and the naive double-and-add commit) and it is never a
// fallback of the product: only bench.py's cpu_baseline leg and tests/ load it.
//
//   * fields: Montgomery form on saturated 64-bit limbs (4 for both scalar fields and BN254's Fp,
//     6 for BLS12-381's Fp), CIOS multiplication on unsigned __int128;
//   * NTT: iterative radix-2 decimation in time, twiddle table, OpenMP over butterflies.  At the
//     level that merges halves of length m/2 butterfly i uses (w^(n/m))^i, so the result is the
//     reference recursion's (fft_ff.py:15-37) for every w, like the engine's;
//   * MSM: Pippenger with signed 16-bit windows, one bucket set per window, XYZZ mixed additions,
//     running-sum bucket reduction; windows are dealt to the OpenMP threads, the window sums are
//     combined by Horner;
//   * key generation: fixed-base 8-bit windows per point, batch inversion to affine.
//
// Parity: tests/test_fast_cpu.py compares it with oracle/py_oracle.py (and the trapdoor identity)
// before bench.py is allowed to quote it.  A tuned assembly library (blst-class, batched-affine
// buckets) would be a further ~2-4x faster per core; the figure is labelled accordingly.
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef unsigned __int128 u128;

template <int N>
struct Fld {
  uint64_t p[N];
  uint64_t n0;          // -p^-1 mod 2^64
  uint64_t r1[N];       // R mod p   (Montgomery one)
  uint64_t r2[N];       // R^2 mod p
};

template <int N>
static inline bool geq(const uint64_t* a, const uint64_t* b) {
  for (int i = N - 1; i >= 0; --i) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}
template <int N>
static inline uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t borrow = 0;
  for (int i = 0; i < N; ++i) {
    const u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
template <int N>
static inline uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t carry = 0;
  for (int i = 0; i < N; ++i) {
    const u128 s = (u128)a[i] + b[i] + carry;
    r[i] = (uint64_t)s;
    carry = (uint64_t)(s >> 64);
  }
  return carry;
}
template <int N>
static inline void f_add(const Fld<N>& F, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t t[N];
  const uint64_t c = add_n<N>(t, a, b);
  if (c || geq<N>(t, F.p)) sub_n<N>(t, t, F.p);
  memcpy(r, t, sizeof(t));
}
template <int N>
static inline void f_sub(const Fld<N>& F, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t t[N];
  if (sub_n<N>(t, a, b)) add_n<N>(t, t, F.p);
  memcpy(r, t, sizeof(t));
}
template <int N>
static inline void f_mul(const Fld<N>& F, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t t[N + 2];
  for (int i = 0; i < N + 2; ++i) t[i] = 0;
  for (int i = 0; i < N; ++i) {
    u128 c = 0;
    for (int j = 0; j < N; ++j) {
      c += (u128)a[j] * b[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[N];
    t[N] = (uint64_t)c;
    t[N + 1] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * F.n0;
    c = (u128)m * F.p[0] + t[0];
    c >>= 64;
    for (int j = 1; j < N; ++j) {
      c += (u128)m * F.p[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[N];
    t[N - 1] = (uint64_t)c;
    t[N] = t[N + 1] + (uint64_t)(c >> 64);
  }
  if (t[N] || geq<N>(t, F.p)) sub_n<N>(t, t, F.p);
  memcpy(r, t, N * 8);
}
template <int N>
static inline bool f_is_zero(const uint64_t* a) {
  uint64_t acc = 0;
  for (int i = 0; i < N; ++i) acc |= a[i];
  return acc == 0;
}
template <int N>
static void f_pow(const Fld<N>& F, uint64_t* r, const uint64_t* a, const uint64_t* e, int ne) {
  uint64_t acc[N], base[N];
  memcpy(acc, F.r1, N * 8);
  memcpy(base, a, N * 8);
  for (int k = 0; k < ne; ++k)
    for (int b = 0; b < 64; ++b) {
      if ((e[k] >> b) & 1) f_mul<N>(F, acc, acc, base);
      f_mul<N>(F, base, base, base);
    }
  memcpy(r, acc, N * 8);
}
template <int N>
static void f_inv(const Fld<N>& F, uint64_t* r, const uint64_t* a) {   // a^(p-2)
  uint64_t e[N], two[N] = {2};
  sub_n<N>(e, F.p, two);
  f_pow<N>(F, r, a, e, N);
}
template <int N>
static void f_to_mont(const Fld<N>& F, uint64_t* r, const uint64_t* a) { f_mul<N>(F, r, a, F.r2); }
template <int N>
static void f_from_mont(const Fld<N>& F, uint64_t* r, const uint64_t* a) {
  uint64_t one[N] = {1};
  f_mul<N>(F, r, a, one);
}
template <int N>
static Fld<N> make_field(const uint64_t* p) {
  Fld<N> F;
  memcpy(F.p, p, N * 8);
  uint64_t inv = 1;                                   // Newton: inv = p^-1 mod 2^64
  for (int i = 0; i < 6; ++i) inv *= 2 - p[0] * inv;
  F.n0 = (uint64_t)0 - inv;
  uint64_t x[N] = {1};                                // 2^k mod p by doubling
  auto dbl = [&](uint64_t* v) {
    uint64_t t[N];
    const uint64_t c = add_n<N>(t, v, v);
    if (c || geq<N>(t, p)) sub_n<N>(t, t, p);
    memcpy(v, t, N * 8);
  };
  for (int i = 0; i < 64 * N; ++i) dbl(x);
  memcpy(F.r1, x, N * 8);
  for (int i = 0; i < 64 * N; ++i) dbl(x);
  memcpy(F.r2, x, N * 8);
  return F;
}

// ---- curves ------------------------------------------------------------------------------------
static const uint64_t BN_P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t BN_R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t BLS_P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                  0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t BLS_R[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const uint64_t BN_G[8] = {1, 0, 0, 0, 2, 0, 0, 0};
static const uint64_t BLS_G[12] = {
    0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull, 0xc3688c4f9774b905ull,
    0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull,
    0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull, 0xfcf5e095d5d00af6ull,
    0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};

// ---- G1 in XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ = 0) -----------------
template <int N>
struct Pt {
  uint64_t x[N], y[N], zz[N], zzz[N];
};
template <int N>
static inline void pt_inf(Pt<N>& p) { memset(&p, 0, sizeof(p)); }
template <int N>
static inline bool pt_is_inf(const Pt<N>& p) { return f_is_zero<N>(p.zz); }

template <int N>
static inline void pt_dbl_affine(const Fld<N>& F, Pt<N>& r, const uint64_t* x, const uint64_t* y) {
  // dbl-2008-s-1 with ZZ1 = ZZZ1 = 1 (a = 0):  U = 2Y, V = U^2, W = U V, S = X V, M = 3 X^2
  uint64_t U[N], V[N], W[N], S[N], M[N], t[N];
  if (f_is_zero<N>(y)) { pt_inf(r); return; }
  f_add<N>(F, U, y, y);
  f_mul<N>(F, V, U, U);
  f_mul<N>(F, W, U, V);
  f_mul<N>(F, S, x, V);
  f_mul<N>(F, t, x, x);
  f_add<N>(F, M, t, t);
  f_add<N>(F, M, M, t);
  f_mul<N>(F, t, M, M);
  f_sub<N>(F, t, t, S);
  f_sub<N>(F, r.x, t, S);
  f_sub<N>(F, t, S, r.x);
  f_mul<N>(F, t, M, t);
  uint64_t wy[N];
  f_mul<N>(F, wy, W, y);
  f_sub<N>(F, r.y, t, wy);
  memcpy(r.zz, V, N * 8);
  memcpy(r.zzz, W, N * 8);
}
template <int N>
static inline void pt_dbl(const Fld<N>& F, Pt<N>& r, const Pt<N>& p) {
  if (pt_is_inf(p) || f_is_zero<N>(p.y)) { pt_inf(r); return; }
  uint64_t U[N], V[N], W[N], S[N], M[N], t[N], x3[N], y3[N];
  f_add<N>(F, U, p.y, p.y);
  f_mul<N>(F, V, U, U);
  f_mul<N>(F, W, U, V);
  f_mul<N>(F, S, p.x, V);
  f_mul<N>(F, t, p.x, p.x);
  f_add<N>(F, M, t, t);
  f_add<N>(F, M, M, t);
  f_mul<N>(F, t, M, M);
  f_sub<N>(F, t, t, S);
  f_sub<N>(F, x3, t, S);
  f_sub<N>(F, t, S, x3);
  f_mul<N>(F, t, M, t);
  f_mul<N>(F, y3, W, p.y);
  f_sub<N>(F, y3, t, y3);
  uint64_t zz[N], zzz[N];
  f_mul<N>(F, zz, V, p.zz);
  f_mul<N>(F, zzz, W, p.zzz);
  memcpy(r.x, x3, N * 8); memcpy(r.y, y3, N * 8); memcpy(r.zz, zz, N * 8); memcpy(r.zzz, zzz, N * 8);
}
// r = p + (x2, y2) affine, neg: use -y2   (madd-2008-s, all special cases exact)
template <int N>
static inline void pt_madd(const Fld<N>& F, Pt<N>& p, const uint64_t* x2, const uint64_t* y2in, bool neg) {
  uint64_t y2[N];
  if (neg && !f_is_zero<N>(y2in)) sub_n<N>(y2, F.p, y2in); else memcpy(y2, y2in, N * 8);
  if (pt_is_inf(p)) {
    memcpy(p.x, x2, N * 8); memcpy(p.y, y2, N * 8); memcpy(p.zz, F.r1, N * 8); memcpy(p.zzz, F.r1, N * 8);
    return;
  }
  uint64_t U2[N], S2[N], P[N], R[N];
  f_mul<N>(F, U2, x2, p.zz);
  f_mul<N>(F, S2, y2, p.zzz);
  f_sub<N>(F, P, U2, p.x);
  f_sub<N>(F, R, S2, p.y);
  if (f_is_zero<N>(P)) {
    if (f_is_zero<N>(R)) pt_dbl_affine<N>(F, p, x2, y2); else pt_inf(p);
    return;
  }
  uint64_t PP[N], PPP[N], Q[N], t[N], x3[N];
  f_mul<N>(F, PP, P, P);
  f_mul<N>(F, PPP, P, PP);
  f_mul<N>(F, Q, p.x, PP);
  f_mul<N>(F, t, R, R);
  f_sub<N>(F, t, t, PPP);
  f_sub<N>(F, t, t, Q);
  f_sub<N>(F, x3, t, Q);
  f_sub<N>(F, t, Q, x3);
  f_mul<N>(F, t, R, t);
  uint64_t u[N];
  f_mul<N>(F, u, p.y, PPP);
  f_sub<N>(F, p.y, t, u);
  memcpy(p.x, x3, N * 8);
  f_mul<N>(F, p.zz, p.zz, PP);
  f_mul<N>(F, p.zzz, p.zzz, PPP);
}
template <int N>
static inline void pt_add(const Fld<N>& F, Pt<N>& p, const Pt<N>& q) {   // add-2008-s
  if (pt_is_inf(q)) return;
  if (pt_is_inf(p)) { p = q; return; }
  uint64_t U1[N], U2[N], S1[N], S2[N], P[N], R[N];
  f_mul<N>(F, U1, p.x, q.zz);
  f_mul<N>(F, U2, q.x, p.zz);
  f_mul<N>(F, S1, p.y, q.zzz);
  f_mul<N>(F, S2, q.y, p.zzz);
  f_sub<N>(F, P, U2, U1);
  f_sub<N>(F, R, S2, S1);
  if (f_is_zero<N>(P)) {
    if (f_is_zero<N>(R)) { Pt<N> d; pt_dbl<N>(F, d, p); p = d; } else pt_inf(p);
    return;
  }
  uint64_t PP[N], PPP[N], Q[N], t[N], x3[N], u[N];
  f_mul<N>(F, PP, P, P);
  f_mul<N>(F, PPP, P, PP);
  f_mul<N>(F, Q, U1, PP);
  f_mul<N>(F, t, R, R);
  f_sub<N>(F, t, t, PPP);
  f_sub<N>(F, t, t, Q);
  f_sub<N>(F, x3, t, Q);
  f_sub<N>(F, t, Q, x3);
  f_mul<N>(F, t, R, t);
  f_mul<N>(F, u, S1, PPP);
  f_sub<N>(F, p.y, t, u);
  memcpy(p.x, x3, N * 8);
  f_mul<N>(F, t, p.zz, q.zz);
  f_mul<N>(F, p.zz, t, PP);
  f_mul<N>(F, t, p.zzz, q.zzz);
  f_mul<N>(F, p.zzz, t, PPP);
}
// XYZZ -> canonical affine words; returns infinity flag
template <int N>
static bool pt_to_affine(const Fld<N>& F, const Pt<N>& p, uint64_t* xy) {
  if (pt_is_inf(p)) { memset(xy, 0, 2 * N * 8); return true; }
  uint64_t zi[N], z2[N], t[N];
  f_inv<N>(F, zi, p.zzz);                 // 1/ZZZ
  f_mul<N>(F, t, zi, p.zz);               // ZZ/ZZZ = 1/Z
  f_mul<N>(F, z2, t, t);                  // 1/ZZ
  f_mul<N>(F, t, p.x, z2);
  f_from_mont<N>(F, xy, t);
  f_mul<N>(F, t, p.y, zi);
  f_from_mont<N>(F, xy + N, t);
  return false;
}

// ---- NTT -------------------------------------------------------------------------------------------
static int ntt_run(const uint64_t* rmod, uint64_t* data, size_t n, const uint64_t* w, int inverse, int threads) {
  constexpr int N = 4;
  if (n == 0 || (n & (n - 1))) return -1;
  const Fld<N> F = make_field<N>(rmod);
  int log_n = 0;
  while (((size_t)1 << log_n) < n) ++log_n;
  uint64_t wm[N];
  f_to_mont<N>(F, wm, w);
  if (inverse) f_inv<N>(F, wm, wm);                                        // fft_ff.py:53
  // twiddles tw[i] = w^i, i < n/2 (Montgomery form), built in parallel blocks
  const size_t half = n / 2;
  std::vector<uint64_t> tw((half ? half : 1) * N);
  const size_t BLK = 1024;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long long b0 = 0; b0 < (long long)((half + BLK - 1) / BLK); ++b0) {
    const size_t s = (size_t)b0 * BLK, e = s + BLK < half ? s + BLK : half;
    uint64_t cur[N], ex[1] = {(uint64_t)s};
    f_pow<N>(F, cur, wm, ex, 1);
    for (size_t i = s; i < e; ++i) {
      memcpy(&tw[i * N], cur, N * 8);
      f_mul<N>(F, cur, cur, wm);
    }
  }
  // bit-reversed placement (values stay in standard form: x * (w R) / R = x w)
  std::vector<uint64_t> buf(n * N);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long long i = 0; i < (long long)n; ++i) {
    size_t rv = 0;
    for (int b = 0; b < log_n; ++b) rv |= (((size_t)i >> b) & 1) << (log_n - 1 - b);
    memcpy(&buf[rv * N], data + (size_t)i * N, N * 8);
  }
  for (int s = 1; s <= log_n; ++s) {
    const size_t m = (size_t)1 << s, h = m >> 1, stride = n >> s;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long bf = 0; bf < (long long)half; ++bf) {
      const size_t i = (size_t)bf & (h - 1), g = (size_t)bf >> (s - 1);
      uint64_t* x = &buf[(g * m + i) * N];
      uint64_t* y = x + h * N;
      uint64_t t[N];
      f_mul<N>(F, t, y, &tw[i * stride * N]);
      f_sub<N>(F, y, x, t);
      f_add<N>(F, x, x, t);
    }
  }
  if (inverse) {
    uint64_t nn[N] = {(uint64_t)n}, ninv[N];
    f_to_mont<N>(F, nn, nn);
    f_inv<N>(F, ninv, nn);                                                // fft_ff.py:57-58
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long long i = 0; i < (long long)n; ++i) f_mul<N>(F, &buf[(size_t)i * N], &buf[(size_t)i * N], ninv);
  }
  memcpy(data, buf.data(), n * N * 8);
  return 0;
}

// ---- key generation: [tau^i G1], affine canonical ------------------------------------------------------
template <int N>
static int setup_run(const uint64_t* pmod, const uint64_t* rmod, const uint64_t* gen, const uint64_t* tau, size_t n,
                     uint64_t* out_xy, int threads) {
  const Fld<N> F = make_field<N>(pmod);
  const Fld<4> R = make_field<4>(rmod);
  // fixed-base table: tab[j][d] = d * 2^(8j) G (affine Montgomery), d = 1..255
  std::vector<uint64_t> tab((size_t)32 * 256 * 2 * N);
  {
    uint64_t gx[N], gy[N];
    f_to_mont<N>(F, gx, gen);
    f_to_mont<N>(F, gy, gen + N);
    Pt<N> base;
    memcpy(base.x, gx, N * 8); memcpy(base.y, gy, N * 8); memcpy(base.zz, F.r1, N * 8); memcpy(base.zzz, F.r1, N * 8);
    for (int j = 0; j < 32; ++j) {
      uint64_t bxy[2 * N];   // affine of base, Montgomery
      {
        uint64_t zi[N], z2[N], t[N];
        f_inv<N>(F, zi, base.zzz);
        f_mul<N>(F, t, zi, base.zz);
        f_mul<N>(F, z2, t, t);
        f_mul<N>(F, bxy, base.x, z2);
        f_mul<N>(F, bxy + N, base.y, zi);
      }
      Pt<N> acc;
      pt_inf(acc);
      for (int d = 1; d < 256; ++d) {
        pt_madd<N>(F, acc, bxy, bxy + N, false);
        uint64_t zi[N], z2[N], t[N];
        f_inv<N>(F, zi, acc.zzz);
        f_mul<N>(F, t, zi, acc.zz);
        f_mul<N>(F, z2, t, t);
        uint64_t* e = &tab[((size_t)j * 256 + d) * 2 * N];
        f_mul<N>(F, e, acc.x, z2);
        f_mul<N>(F, e + N, acc.y, zi);
      }
      for (int q = 0; q < 8; ++q) { Pt<N> d2; pt_dbl<N>(F, d2, base); base = d2; }
    }
  }
  uint64_t taum[4];
  f_to_mont<4>(R, taum, tau);
  const size_t BLK = 256;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
  for (long long b0 = 0; b0 < (long long)((n + BLK - 1) / BLK); ++b0) {
    const size_t s = (size_t)b0 * BLK, e = s + BLK < n ? s + BLK : n;
    uint64_t cur[4], ex[1] = {(uint64_t)s};
    f_pow<4>(R, cur, taum, ex, 1);
    std::vector<Pt<N>> pts(e - s);
    for (size_t i = s; i < e; ++i) {
      uint64_t sc[4];
      f_from_mont<4>(R, sc, cur);
      Pt<N> acc;
      pt_inf(acc);
      for (int j = 0; j < 32; ++j) {
        const unsigned d = (unsigned)((sc[j >> 3] >> (8 * (j & 7))) & 0xff);
        if (d) { const uint64_t* t = &tab[((size_t)j * 256 + d) * 2 * N]; pt_madd<N>(F, acc, t, t + N, false); }
      }
      pts[i - s] = acc;
      f_mul<4>(R, cur, cur, taum);
    }
    // batch inversion of the ZZZ's of the block (tau^i G is never infinity for tau != 0; guard anyway)
    std::vector<uint64_t> pref((e - s) * N);
    uint64_t run[N];
    memcpy(run, F.r1, N * 8);
    for (size_t i = 0; i < e - s; ++i) {
      memcpy(&pref[i * N], run, N * 8);
      if (!pt_is_inf(pts[i])) f_mul<N>(F, run, run, pts[i].zzz);
    }
    uint64_t inv[N];
    f_inv<N>(F, inv, run);
    for (size_t i = e - s; i-- > 0;) {
      uint64_t* o = out_xy + (s + i) * 2 * N;
      if (pt_is_inf(pts[i])) { memset(o, 0, 2 * N * 8); continue; }
      uint64_t zi[N], t[N], z2[N];
      f_mul<N>(F, zi, inv, &pref[i * N]);          // 1/ZZZ_i
      f_mul<N>(F, inv, inv, pts[i].zzz);
      f_mul<N>(F, t, zi, pts[i].zz);
      f_mul<N>(F, z2, t, t);
      f_mul<N>(F, t, pts[i].x, z2);
      f_from_mont<N>(F, o, t);
      f_mul<N>(F, t, pts[i].y, zi);
      f_from_mont<N>(F, o + N, t);
    }
  }
  return 0;
}

// ---- MSM: signed 16-bit windows --------------------------------------------------------------------------
template <int N>
static int msm_run(const uint64_t* pmod, const uint64_t* xy, const uint8_t* inf, size_t n, const uint64_t* scalars,
                   uint64_t* out_xy, uint8_t* out_inf, int threads) {
  constexpr int C = 16, NWIN = 16, NB = 1 << (C - 1);
  const Fld<N> F = make_field<N>(pmod);
  // points to Montgomery form once
  std::vector<uint64_t> pm(n * 2 * N);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long long i = 0; i < (long long)n; ++i) {
    f_to_mont<N>(F, &pm[(size_t)i * 2 * N], xy + (size_t)i * 2 * N);
    f_to_mont<N>(F, &pm[(size_t)i * 2 * N + N], xy + (size_t)i * 2 * N + N);
  }
  // signed digits: d_j in [-2^15, 2^15]; scalars < 2^255 so the top window absorbs the last carry
  std::vector<int32_t> dig(n * NWIN);
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long long i = 0; i < (long long)n; ++i) {
    const uint64_t* s = scalars + (size_t)i * 4;
    int carry = 0;
    for (int j = 0; j < NWIN; ++j) {
      int v = (int)((s[j >> 2] >> (16 * (j & 3))) & 0xffff) + carry;
      carry = 0;
      if (v > NB && j < NWIN - 1) { v -= 1 << C; carry = 1; }
      dig[(size_t)i * NWIN + j] = v;
    }
  }
  std::vector<Pt<N>> wsum(NWIN);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (int j = 0; j < NWIN; ++j) {
    std::vector<Pt<N>> bucket(NB + 1);
    for (auto& b : bucket) pt_inf(b);
    for (size_t i = 0; i < n; ++i) {
      const int d = dig[i * NWIN + j];
      if (d == 0 || (inf && inf[i])) continue;                          // kzg.py:113-114: zero coefficients skipped
      const uint64_t* q = &pm[i * 2 * N];
      pt_madd<N>(F, bucket[d > 0 ? d : -d], q, q + N, d < 0);
    }
    Pt<N> run, sum;
    pt_inf(run);
    pt_inf(sum);
    for (int k = NB; k >= 1; --k) {                                    // sum_k k * B_k by running sums
      pt_add<N>(F, run, bucket[k]);
      pt_add<N>(F, sum, run);
    }
    wsum[j] = sum;
  }
  Pt<N> acc;
  pt_inf(acc);
  for (int j = NWIN - 1; j >= 0; --j) {
    for (int b = 0; b < C; ++b) { Pt<N> d2; pt_dbl<N>(F, d2, acc); acc = d2; }
    pt_add<N>(F, acc, wsum[j]);
  }
  *out_inf = pt_to_affine<N>(F, acc, out_xy) ? 1 : 0;
  return 0;
}

extern "C" {

int fc_fp_limbs(int curve) { return curve == 0 ? 4 : (curve == 1 ? 6 : 0); }
int fc_max_threads(void) { return omp_get_max_threads(); }

int fc_ntt(int curve, uint64_t* data, size_t n, const uint64_t* w, int inverse, int threads) {
  if (curve != 0 && curve != 1) return -1;
  return ntt_run(curve == 0 ? BN_R : BLS_R, data, n, w, inverse, threads > 0 ? threads : 1);
}
int fc_setup(int curve, const uint64_t* tau, size_t n, uint64_t* out_xy, int threads) {
  if (threads <= 0) threads = 1;
  if (curve == 0) return setup_run<4>(BN_P, BN_R, BN_G, tau, n, out_xy, threads);
  if (curve == 1) return setup_run<6>(BLS_P, BLS_R, BLS_G, tau, n, out_xy, threads);
  return -1;
}
int fc_msm(int curve, const uint64_t* xy, const uint8_t* inf, size_t n, const uint64_t* scalars, uint64_t* out_xy,
           uint8_t* out_inf, int threads) {
  if (threads <= 0) threads = 1;
  if (curve == 0) return msm_run<4>(BN_P, xy, inf, n, scalars, out_xy, out_inf, threads);
  if (curve == 1) return msm_run<6>(BLS_P, xy, inf, n, scalars, out_xy, out_inf, threads);
  return -1;
}

}  // extern "C"
