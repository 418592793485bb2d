/*
 * oracle/kzg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's hot path with the reference's own
 * algorithms (swusjask/kzg-snark):
 *   fft_ff / ifft_ff            recursive radix-2 DIT with even/odd copies  fft_ff.py:3-58
 *   KZG.commit                  one double-and-add scalar-mul per coefficient
 *                               plus a running add, zero coefficients skipped  kzg.py:108-116
 *   KZG.open                    sum xi^(i+1) p_i, Horner, synthetic division   kzg.py:147-157
 *   KZG.setup (G1 half)         tau^i * G1 by independent scalar-muls          kzg.py:69-72
 * and py_ecc's published homogeneous-projective group law (see py_oracle.py:
 * restated from the textbook formulas, py_ecc itself is not present).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the shipped package never does.  PARITY UNPINNED by the
 * reference (no golden vectors exist there); this file is pinned
 * against oracle/py_oracle.py, public curve constants / known-answer points and
 * algebraic identities in tests/test_oracle_pins.py.
 *
 * Arithmetic: Montgomery form on 64-bit limbs with unsigned __int128 (4 limbs
 * for the ~255-bit fields, 6 for BLS12-381's Fp).  Single-threaded, like the
 * reference.  Values cross the API as canonical little-endian 64-bit limbs.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define MAXL 6

typedef struct {
  int n;               /* limbs */
  uint64_t p[MAXL];    /* modulus */
  uint64_t n0;         /* -p^-1 mod 2^64 */
  uint64_t r1[MAXL];   /* R mod p */
  uint64_t r2[MAXL];   /* R^2 mod p */
} field_t;

typedef struct {
  field_t fr, fp;
  uint64_t b_mont[MAXL];   /* curve coefficient b, Montgomery form */
  uint64_t gx[MAXL], gy[MAXL]; /* generator, canonical */
} curve_t;

/* ---- multi-limb helpers ------------------------------------------------------------ */
static int ge(const uint64_t* a, const uint64_t* b, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  uint64_t br = 0;
  for (int i = 0; i < n; ++i) {
    u128 d = (u128)a[i] - b[i] - br;
    r[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
static uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  uint64_t c = 0;
  for (int i = 0; i < n; ++i) {
    u128 s = (u128)a[i] + b[i] + c;
    r[i] = (uint64_t)s;
    c = (uint64_t)(s >> 64);
  }
  return c;
}
static int is_zero_n(const uint64_t* a, int n) {
  uint64_t acc = 0;
  for (int i = 0; i < n; ++i) acc |= a[i];
  return acc == 0;
}

static void f_add(const field_t* f, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t t[MAXL];
  uint64_t c = add_n(t, a, b, f->n);
  if (c || ge(t, f->p, f->n)) sub_n(t, t, f->p, f->n);
  memcpy(r, t, f->n * 8);
}
static void f_sub(const field_t* f, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  uint64_t t[MAXL];
  if (sub_n(t, a, b, f->n)) add_n(t, t, f->p, f->n);
  memcpy(r, t, f->n * 8);
}
/* Montgomery product (CIOS) */
static void f_mul(const field_t* f, uint64_t* r, const uint64_t* a, const uint64_t* b) {
  const int n = f->n;
  uint64_t t[MAXL + 2];
  memset(t, 0, sizeof(t));
  for (int i = 0; i < n; ++i) {
    uint64_t c = 0;
    for (int j = 0; j < n; ++j) {
      u128 cur = (u128)a[j] * b[i] + t[j] + c;
      t[j] = (uint64_t)cur;
      c = (uint64_t)(cur >> 64);
    }
    u128 cur = (u128)t[n] + c;
    t[n] = (uint64_t)cur;
    t[n + 1] = (uint64_t)(cur >> 64);
    uint64_t m = t[0] * f->n0;
    cur = (u128)m * f->p[0] + t[0];
    c = (uint64_t)(cur >> 64);
    for (int j = 1; j < n; ++j) {
      cur = (u128)m * f->p[j] + t[j] + c;
      t[j - 1] = (uint64_t)cur;
      c = (uint64_t)(cur >> 64);
    }
    cur = (u128)t[n] + c;
    t[n - 1] = (uint64_t)cur;
    t[n] = t[n + 1] + (uint64_t)(cur >> 64);
  }
  if (t[n] || ge(t, f->p, n)) sub_n(t, t, f->p, n);
  memcpy(r, t, n * 8);
}
static void f_to_mont(const field_t* f, uint64_t* r, const uint64_t* a) { f_mul(f, r, a, f->r2); }
static void f_from_mont(const field_t* f, uint64_t* r, const uint64_t* a) {
  uint64_t one[MAXL] = {1, 0, 0, 0, 0, 0};
  f_mul(f, r, a, one);
}
/* a^e, e given as n limbs (Montgomery in/out) */
static void f_pow(const field_t* f, uint64_t* r, const uint64_t* a, const uint64_t* e) {
  uint64_t acc[MAXL], base[MAXL];
  memcpy(acc, f->r1, f->n * 8);
  memcpy(base, a, f->n * 8);
  for (int i = f->n - 1; i >= 0; --i)
    for (int b = 63; b >= 0; --b) {
      f_mul(f, acc, acc, acc);
      if ((e[i] >> b) & 1) f_mul(f, acc, acc, base);
    }
  memcpy(r, acc, f->n * 8);
}
static void f_inv(const field_t* f, uint64_t* r, const uint64_t* a) {
  uint64_t e[MAXL], two[MAXL] = {2, 0, 0, 0, 0, 0};
  sub_n(e, f->p, two, f->n);
  f_pow(f, r, a, e);
}

/* field setup from the modulus alone */
static void field_init(field_t* f, const uint64_t* p, int n) {
  memset(f, 0, sizeof(*f));
  f->n = n;
  memcpy(f->p, p, n * 8);
  uint64_t inv = 1;                      /* Newton: p^-1 mod 2^64 */
  for (int i = 0; i < 6; ++i) inv *= 2 - p[0] * inv;
  f->n0 = (uint64_t)0 - inv;
  /* R mod p and R^2 mod p by repeated doubling */
  uint64_t x[MAXL] = {1, 0, 0, 0, 0, 0};
  for (int i = 0; i < 2 * 64 * n; ++i) {
    uint64_t c = add_n(x, x, x, n);
    if (c || ge(x, p, n)) sub_n(x, x, p, n);
    if (i == 64 * n - 1) memcpy(f->r1, x, n * 8);
  }
  memcpy(f->r2, x, n * 8);
}

/* ---- curves ---------------------------------------------------------------------------- */
static curve_t g_curves[2];
static int g_init = 0;

static const uint64_t BN_R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t BN_P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t BLS_R[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
static const uint64_t BLS_P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                  0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t BLS_GX[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                                   0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
static const uint64_t BLS_GY[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                                   0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};

static void curves_init(void) {
  if (g_init) return;
  curve_t* bn = &g_curves[0];
  field_init(&bn->fr, BN_R, 4);
  field_init(&bn->fp, BN_P, 4);
  uint64_t three[MAXL] = {3, 0, 0, 0, 0, 0};
  f_to_mont(&bn->fp, bn->b_mont, three);
  memset(bn->gx, 0, sizeof(bn->gx)); memset(bn->gy, 0, sizeof(bn->gy));
  bn->gx[0] = 1; bn->gy[0] = 2;
  curve_t* bl = &g_curves[1];
  field_init(&bl->fr, BLS_R, 4);
  field_init(&bl->fp, BLS_P, 6);
  uint64_t four[MAXL] = {4, 0, 0, 0, 0, 0};
  f_to_mont(&bl->fp, bl->b_mont, four);
  memcpy(bl->gx, BLS_GX, 48); memcpy(bl->gy, BLS_GY, 48);
  g_init = 1;
}
static const curve_t* get_curve(int id) {
  curves_init();
  return (id == 0 || id == 1) ? &g_curves[id] : NULL;
}

/* ---- fft_ff.py ---------------------------------------------------------------------------- */
/* values in Montgomery form; w Montgomery.  fft_ff.py:15-37 */
static void fft_rec(const field_t* f, uint64_t* out, const uint64_t* in, size_t n, const uint64_t* w) {
  const int L = f->n;
  if (n == 1) { memcpy(out, in, L * 8); return; }                     /* :16-17 */
  /* No length check in the reference: coeffs[0::2] has ceil(n/2) entries, coeffs[1::2] floor(n/2),
   * the loop runs n // 2 butterflies, and for odd n result[n-1] keeps its initial F(0). */
  size_t h = n / 2, he = n - h;
  uint64_t* even = (uint64_t*)malloc(he * L * 8);
  uint64_t* odd = (uint64_t*)malloc(h * L * 8);
  uint64_t* ef = (uint64_t*)malloc(he * L * 8);
  uint64_t* of = (uint64_t*)malloc(h * L * 8);
  for (size_t i = 0; i < he; ++i) memcpy(even + i * L, in + (2 * i) * L, L * 8);      /* :20 */
  for (size_t i = 0; i < h; ++i) memcpy(odd + i * L, in + (2 * i + 1) * L, L * 8);    /* :21 */
  uint64_t w2[MAXL];
  f_mul(f, w2, w, w);                                                 /* :24 */
  fft_rec(f, ef, even, he, w2);                                       /* :25 */
  fft_rec(f, of, odd, h, w2);                                         /* :26 */
  memset(out, 0, n * L * 8);                                          /* :29 */
  uint64_t wp[MAXL], t[MAXL];
  memcpy(wp, f->r1, L * 8);                                           /* :30 */
  for (size_t i = 0; i < h; ++i) {                                    /* :32-35 */
    f_mul(f, t, wp, of + i * L);
    f_add(f, out + i * L, ef + i * L, t);
    f_sub(f, out + (i + h) * L, ef + i * L, t);
    f_mul(f, wp, wp, w);
  }
  free(even); free(odd); free(ef); free(of);
}

/* data: n canonical Fr elements (4 limbs each), in place.  inverse: fft_ff.py:51-58 */
int oracle_fft(int curve_id, uint64_t* data, size_t n, const uint64_t* w, int inverse) {
  const curve_t* cv = get_curve(curve_id);
  if (!cv || n == 0) return -1;   /* n == 0: the reference recursion never terminates */
  const field_t* f = &cv->fr;
  uint64_t* in = (uint64_t*)malloc(n * 32);
  uint64_t* out = (uint64_t*)malloc(n * 32);
  for (size_t i = 0; i < n; ++i) f_to_mont(f, in + 4 * i, data + 4 * i);
  uint64_t wm[MAXL];
  f_to_mont(f, wm, w);
  if (inverse) f_inv(f, wm, wm);                                      /* :53 */
  fft_rec(f, out, in, n, wm);
  if (inverse) {
    uint64_t nn[MAXL] = {0}, ninv[MAXL];
    nn[0] = (uint64_t)n;
    f_to_mont(f, nn, nn);
    f_inv(f, ninv, nn);                                               /* :57 */
    for (size_t i = 0; i < n; ++i) f_mul(f, out + 4 * i, out + 4 * i, ninv);   /* :58 */
  }
  for (size_t i = 0; i < n; ++i) f_from_mont(f, data + 4 * i, out + 4 * i);
  free(in); free(out);
  return 0;
}

/* ---- py_ecc-shaped G1 (homogeneous projective, Montgomery coordinates) ------------------------- */
typedef struct { uint64_t x[MAXL], y[MAXL], z[MAXL]; } pt_t;

static void pt_inf(const field_t* f, pt_t* r) {   /* (1, 1, 0) */
  memcpy(r->x, f->r1, f->n * 8); memcpy(r->y, f->r1, f->n * 8); memset(r->z, 0, sizeof(r->z));
}
static void f_muls(const field_t* f, uint64_t* r, const uint64_t* a, int k) {   /* r = k*a, small k */
  uint64_t acc[MAXL];
  memset(acc, 0, sizeof(acc));
  for (int i = 0; i < k; ++i) f_add(f, acc, acc, a);
  memcpy(r, acc, f->n * 8);
}
static void pt_double(const field_t* f, pt_t* r, const pt_t* p) {
  uint64_t W[MAXL], S[MAXL], B[MAXL], H[MAXL], SS[MAXL], t[MAXL], u[MAXL];
  f_mul(f, t, p->x, p->x); f_muls(f, W, t, 3);                 /* W = 3x^2 */
  f_mul(f, S, p->y, p->z);                                     /* S = yz */
  f_mul(f, t, p->x, p->y); f_mul(f, B, t, S);                  /* B = xyS */
  f_mul(f, t, W, W); f_muls(f, u, B, 8); f_sub(f, H, t, u);    /* H = W^2 - 8B */
  f_mul(f, SS, S, S);
  pt_t o;
  f_mul(f, t, H, S); f_muls(f, o.x, t, 2);                     /* x' = 2HS */
  f_muls(f, t, B, 4); f_sub(f, t, t, H); f_mul(f, t, W, t);    /* W(4B - H) */
  f_mul(f, u, p->y, p->y); f_mul(f, u, u, SS); f_muls(f, u, u, 8);
  f_sub(f, o.y, t, u);                                         /* y' = W(4B-H) - 8y^2S^2 */
  f_mul(f, t, S, SS); f_muls(f, o.z, t, 8);                    /* z' = 8S^3 */
  *r = o;
}
static void pt_add(const field_t* f, pt_t* r, const pt_t* p1, const pt_t* p2) {
  if (is_zero_n(p1->z, f->n) || is_zero_n(p2->z, f->n)) { *r = is_zero_n(p2->z, f->n) ? *p1 : *p2; return; }
  uint64_t U1[MAXL], U2[MAXL], V1[MAXL], V2[MAXL];
  f_mul(f, U1, p2->y, p1->z); f_mul(f, U2, p1->y, p2->z);
  f_mul(f, V1, p2->x, p1->z); f_mul(f, V2, p1->x, p2->z);
  const int veq = memcmp(V1, V2, f->n * 8) == 0, ueq = memcmp(U1, U2, f->n * 8) == 0;
  if (veq && ueq) { pt_double(f, r, p1); return; }
  if (veq) { pt_inf(f, r); return; }
  uint64_t U[MAXL], V[MAXL], VV[MAXL], VV2[MAXL], VVV[MAXL], W[MAXL], A[MAXL], t[MAXL], u[MAXL];
  f_sub(f, U, U1, U2); f_sub(f, V, V1, V2);
  f_mul(f, VV, V, V); f_mul(f, VV2, VV, V2); f_mul(f, VVV, V, VV);
  f_mul(f, W, p1->z, p2->z);
  f_mul(f, t, U, U); f_mul(f, t, t, W); f_sub(f, t, t, VVV); f_muls(f, u, VV2, 2); f_sub(f, A, t, u);
  pt_t o;
  f_mul(f, o.x, V, A);
  f_sub(f, t, VV2, A); f_mul(f, t, U, t); f_mul(f, u, VVV, U2); f_sub(f, o.y, t, u);
  f_mul(f, o.z, VVV, W);
  *r = o;
}
/* scalar given as 4 canonical limbs; py_ecc's recursion unrolled (see py_oracle.multiply) */
static void pt_mul(const field_t* f, pt_t* r, const pt_t* p, const uint64_t* k) {
  int top = -1;
  for (int i = 255; i >= 0; --i) if ((k[i >> 6] >> (i & 63)) & 1) { top = i; break; }
  if (top < 0) { pt_inf(f, r); return; }
  pt_t* dbl = (pt_t*)malloc((top + 1) * sizeof(pt_t));
  dbl[0] = *p;
  for (int i = 1; i <= top; ++i) pt_double(f, &dbl[i], &dbl[i - 1]);
  pt_t acc = dbl[top];
  for (int i = top - 1; i >= 0; --i)
    if ((k[i >> 6] >> (i & 63)) & 1) pt_add(f, &acc, &acc, &dbl[i]);
  free(dbl);
  *r = acc;
}
static void pt_from_affine(const field_t* f, pt_t* r, const uint64_t* xy, int inf) {
  if (inf) { pt_inf(f, r); return; }
  memset(r, 0, sizeof(*r));
  f_to_mont(f, r->x, xy); f_to_mont(f, r->y, xy + f->n);
  memcpy(r->z, f->r1, f->n * 8);
}
static void pt_to_affine(const field_t* f, const pt_t* p, uint64_t* xy, uint8_t* inf) {
  if (is_zero_n(p->z, f->n)) { memset(xy, 0, 2 * f->n * 8); *inf = 1; return; }
  uint64_t zi[MAXL], t[MAXL];
  f_inv(f, zi, p->z);
  f_mul(f, t, p->x, zi); f_from_mont(f, xy, t);
  f_mul(f, t, p->y, zi); f_from_mont(f, xy + f->n, t);
  *inf = 0;
}

int oracle_fp_limbs(int curve_id) { const curve_t* cv = get_curve(curve_id); return cv ? cv->fp.n : 0; }

/* out = k * P (affine in, affine out) */
int oracle_g1_mul(int curve_id, const uint64_t* xy, int inf, const uint64_t* k, uint64_t* out_xy, uint8_t* out_inf) {
  const curve_t* cv = get_curve(curve_id);
  if (!cv) return -1;
  pt_t p, r;
  pt_from_affine(&cv->fp, &p, xy, inf);
  pt_mul(&cv->fp, &r, &p, k);
  pt_to_affine(&cv->fp, &r, out_xy, out_inf);
  return 0;
}

/* kzg.py:69-72: ck[i] = multiply(G1, tau^i), i = 0..n-1, affine out */
int oracle_setup(int curve_id, const uint64_t* tau, size_t n, uint64_t* out_xy) {
  const curve_t* cv = get_curve(curve_id);
  if (!cv) return -1;
  const field_t* fp = &cv->fp; const field_t* fr = &cv->fr;
  uint64_t gxy[2 * MAXL];
  memcpy(gxy, cv->gx, fp->n * 8); memcpy(gxy + fp->n, cv->gy, fp->n * 8);
  pt_t g; pt_from_affine(fp, &g, gxy, 0);
  uint64_t tm[MAXL], pw[MAXL], k[MAXL];
  f_to_mont(fr, tm, tau);
  memcpy(pw, fr->r1, fr->n * 8);
  for (size_t i = 0; i < n; ++i) {
    uint8_t inf;
    if (i == 0) { memcpy(out_xy, gxy, 2 * fp->n * 8); }
    else {
      f_mul(fr, pw, pw, tm);
      f_from_mont(fr, k, pw);
      pt_t r; pt_mul(fp, &r, &g, k);
      pt_to_affine(fp, &r, out_xy + i * 2 * fp->n, &inf);
    }
  }
  return 0;
}

/* kzg.py:108-116.  ck: n_ck affine points (+ optional inf flags); coeffs: n canonical scalars */
int oracle_commit(int curve_id, const uint64_t* ck_xy, const uint8_t* ck_inf, size_t n_ck, const uint64_t* coeffs,
                  size_t n, uint64_t* out_xy, uint8_t* out_inf) {
  const curve_t* cv = get_curve(curve_id);
  if (!cv) return -1;
  const field_t* fp = &cv->fp;
  /* degree check on the normalised polynomial (kzg.py:103-106) */
  while (n > 0 && is_zero_n(coeffs + 4 * (n - 1), 4)) --n;
  if (n > n_ck) return -4;
  pt_t acc; pt_inf(fp, &acc);                                         /* :109 */
  for (size_t i = 0; i < n; ++i) {                                    /* :112 */
    const uint64_t* c = coeffs + 4 * i;
    if (is_zero_n(c, 4)) continue;                                    /* :113-114 */
    pt_t p, term;
    pt_from_affine(fp, &p, ck_xy + i * 2 * fp->n, ck_inf ? ck_inf[i] : 0);
    pt_mul(fp, &term, &p, c);                                         /* :115 */
    pt_add(fp, &acc, &acc, &term);                                    /* :116 */
  }
  pt_to_affine(fp, &acc, out_xy, out_inf);
  return 0;
}

/* kzg.py:147-154: combined = sum xi^(i+1) p_i; quotient by (X - z).  polys: k arrays `stride` apart.
 * quot receives max_len - 1 coefficients (zero-padded), eval the value combined(z); returns max_len */
long oracle_open_quotient(int curve_id, const uint64_t* polys, const size_t* lens, size_t k, size_t stride,
                          const uint64_t* z, const uint64_t* xi, uint64_t* quot, uint64_t* eval) {
  const curve_t* cv = get_curve(curve_id);
  if (!cv) return -1;
  const field_t* f = &cv->fr;
  size_t n = 0;
  for (size_t i = 0; i < k; ++i) if (lens[i] > n) n = lens[i];
  memset(eval, 0, 32);
  if (n == 0) return 0;
  uint64_t* comb = (uint64_t*)calloc(n, 32);
  uint64_t xim[MAXL], xp[MAXL], zm[MAXL], t[MAXL], c[MAXL];
  f_to_mont(f, xim, xi); f_to_mont(f, zm, z);
  memcpy(xp, f->r1, 32);
  for (size_t i = 0; i < k; ++i) {                                    /* :148-150 */
    f_mul(f, xp, xp, xim);
    for (size_t j = 0; j < lens[i]; ++j) {
      f_to_mont(f, c, polys + (i * stride + j) * 4);
      f_mul(f, t, xp, c);
      f_add(f, comb + 4 * j, comb + 4 * j, t);
    }
  }
  uint64_t carry[MAXL];                                               /* :154 synthetic division */
  memset(carry, 0, sizeof(carry));
  for (size_t j = n; j-- > 0;) {
    f_mul(f, t, carry, zm);
    f_add(f, carry, comb + 4 * j, t);
    if (j >= 1) f_from_mont(f, quot + 4 * (j - 1), carry);
    else f_from_mont(f, eval, carry);
  }
  free(comb);
  return (long)n;
}
