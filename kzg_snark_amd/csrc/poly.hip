// poly.hip -- the polynomial half of KZG.open (kzg.py:122-159) on gfx950:
//
//   combined(X) = sum_i xi^(i+1) * p_i(X)                     kzg.py:147-150
//   witness(X)  = (combined(X) - combined(z)) // (X - z)      kzg.py:153-154
//
// Division by (X - z) is synthetic division: with S_j = sum_{i>=j} c_i z^(i-j)
// (suffix Horner values, S_j = c_j + z*S_{j+1}) the quotient is q_{j-1} = S_j for
// j >= 1 and combined(z) = S_0.  The recurrence is evaluated in two tile passes (below,
// "The opening's scan as TWO tile passes"); kzg_fr_poly_eval is the first of them without its stores.
// The MSM of the quotient (msm.hip) dominates open() by far.
// These kernels are bound by dependent Horner chains and by memory latency, not by instruction issue: the chain pin of
// field.h (an asm volatile per multiply-add) would only keep the scheduler from hoisting the next loads.
#define KZG_NO_CHAIN_PIN 1
#include "internal.h"
#include "msm.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace kzg {

namespace {

constexpr uint32_t LC = 32;       // elements per thread of the vector primitives (batch inversion, powers, prefix product)
// Chunk of the suffix-Horner scans (open, poly_eval): what one thread walks serially, and the group of lanes whose
// products one DPP sum collapses (8 limbs of 29 bits stay below 2^32).
constexpr uint32_t SC_LOG = 3;
constexpr uint32_t SC = 1u << SC_LOG;
constexpr uint32_t MAXK = 64;     // polynomials per open()

constexpr int FRN = 9;            // both scalar fields: 9 x 29-bit limbs
struct LincombArgs {              // travels in the kernel arguments (2.6 KB): no staging copy, no host synchronisation
  const uint32_t* polys;     // k polynomials, `stride` elements apart, canonical words
  uint64_t stride;
  uint32_t k;
  uint32_t lens[MAXK];
  uint32_t xipow[MAXK * FRN];   // xi^(i+1), Montgomery form
};
struct FrArg {                    // one field element (Montgomery limbs) as a kernel argument
  uint32_t l[FRN];
};

template <class F>
__device__ __forceinline__ Fe<F> load_words(const uint32_t* p) {
  const uint4* g = reinterpret_cast<const uint4*>(p);
  const uint4 lo = g[0], hi = g[1];
  const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return Field<F>::from_words(w);
}
template <class F>
__device__ __forceinline__ void store_words(uint32_t* p, const Fe<F>& v) {
  uint32_t w[8];
  Field<F>::to_words(Field<F>::reduce(v), w);
  uint4* g = reinterpret_cast<uint4*>(p);
  g[0] = make_uint4(w[0], w[1], w[2], w[3]);
  g[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
template <class F>
__device__ __forceinline__ Fe<F> load_limbs(const uint32_t* p) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < F::N; ++j) r.l[j] = p[j];
  return r;
}
template <class F>
__device__ __forceinline__ void store_limbs(uint32_t* p, const Fe<F>& v) {
#pragma unroll
  for (int j = 0; j < F::N; ++j) p[j] = v.l[j];
}

// sum over up to DOT_G terms c_g * x_g with ONE Montgomery reduction (Field::dot): 81 multiply-adds per term plus 74
// for the group, against 155 per term for separate products -- the combination kernels are bound by exactly these
// multiply-adds (rocprofv3, profiles/r03_open_pmc.csv: 31 M VALU wave-instructions per 2^20 x 6 combination before).
constexpr uint32_t DOT_G = 6;
template <class F>
__device__ __forceinline__ Fe<F> dot_upto(uint32_t cnt, const Fe<F>* c, const Fe<F>* x) {
  using Fd = Field<F>;
  switch (cnt) {                                             // uniform over the launch
    case 1: return Fd::template dot<1>(c, x);
    case 2: return Fd::template dot<2>(c, x);
    case 3: return Fd::template dot<3>(c, x);
    case 4: return Fd::template dot<4>(c, x);
    case 5: return Fd::template dot<5>(c, x);
    default: return Fd::template dot<6>(c, x);
  }
}

// out[t] = sum_i xipow[i] * p_i[t]   (xipow in Montgomery form => result in standard form)
template <class F>
__global__ void lincomb_kernel(LincombArgs a, uint32_t* out, uint32_t n) {
  using Fd = Field<F>;
  static_assert(F::N == FRN, "scalar fields have 9 limbs");
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fe<F> acc = Fd::zero();
  for (uint32_t i0 = 0; i0 < a.k; i0 += DOT_G) {
    const uint32_t cnt = min(DOT_G, a.k - i0);
    Fe<F> c[DOT_G], x[DOT_G];
#pragma unroll
    for (uint32_t g = 0; g < DOT_G; ++g) {
      const uint32_t i = i0 + g;
      const bool on = g < cnt && t < a.lens[i];               // shorter polynomials read as zero-padded
      c[g] = on ? load_words<F>(a.polys + (a.stride * i + t) * 8) : Fd::zero();
      x[g] = load_limbs<F>(a.xipow + (g < cnt ? i : i0) * F::N);
    }
    const Fe<F> part = dot_upto<F>(cnt, c, x);
    acc = i0 ? Fd::add(acc, part) : part;
  }
  store_words<F>(out + (size_t)t * 8, acc);
}

__global__ void any_nonzero_kernel(const uint32_t* words, size_t from_elem, size_t to_elem, uint32_t* flag) {
  const size_t i = from_elem + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= to_elem) return;
  uint32_t acc = 0;
  for (int k = 0; k < 8; ++k) acc |= words[i * 8 + k];
  if (acc) atomicOr(flag, 1u);
}

// =====================================================================================
// The opening's scan as TWO tile passes (round 4; rounds 1-3 collapsed chunks of 8 level by level: thirteen launches
// for a 2^20 opening, eleven of them 7-9 us dependent chains on a handful of waves).
//
// A workgroup of TB threads owns a TILE of T = 8 TB consecutive coefficients c_(bT) .. c_(bT+T-1) (thread q owns the
// chunk of 8 at local offset 8q).  With z != 0 the suffix values S_j = sum_(i >= j) c_i z^(i-j) are plain suffix SUMS of
// scaled terms, S_j = z^-j sum_(i >= j) c_i z^i, and the scaling is applied hierarchically so that no power table is
// longer than a tile:
//
//   pass 1 (tile_combine_kernel): comb = sum_i xi^(i+1) p_i (kzg.py:148-150), coalesced (thread t takes the local
//     elements t, t + TB, ..); e = comb * z^(pos mod 8); the 8 products of a chunk are added across 8 adjacent lanes by
//     DPP (lazily: 8 limbs of 29 bits stay below 2^32) -> h_q = sum_(r<8) c_(8q+r) z^r; g_q = h_q z^(8q); the tile's
//     aggregate H_b = sum_q g_q = sum_(i<T) c_(bT+i) z^i.  Written: comb (32 B per coefficient), g (36 B per chunk),
//     H (36 B per tile).  Extra workgroups at the head of the grid build the two tables pass 2 reads: zinv[q] = z^(-8q)
//     (q <= TB) and W[k] = z^(T(k+1)), from a few generator powers the host passes in the kernel arguments.
//   pass 2 (tile_fill_kernel): every tile sums the aggregates ABOVE it itself, Q_b = z^T S_((b+1)T) = sum_(k>=0) H_(b+1+k) W[k]
//     -- one product per tile pair, 1.3 * 10^5 products at 2^20, spread over all workgroups: there is no serial chain
//     over the tiles and no launch per level; beyond 1024 tiles the aggregates of 64 tiles are first folded into one
//     (tile_group_kernel) so that a tile never sums more than 63 + ntiles/64 terms.  Inside the tile a wave-wide suffix
//     sum of the g_q (additions only) gives the carry into chunk q, S_(bT+8(q+1)) = zinv[q+1] (sum_(q'>q) g_q' + Q_b),
//     and each thread walks its 8 coefficients in LDS from there (S_j = c_j + z S_(j+1)); the suffix values leave
//     coalesced: S_0 = combined(z), S_j = quotient coefficient j-1 (kzg.py:153-154).
//
// Products per coefficient: the combination's k (one reduction, Field::dot) + 1 (e) + 1 (walk) + (2 + 1 + ~1)/8.
// Traffic: k + 1 reads of 32 B, 2 writes of 32 B per coefficient + 9 B per coefficient of g -> (k+3)/(k+1) of the
// algorithmic bytes, the floor of a combine-then-scan structure.  z = 0 has no inverse: then S_j = c_j, a copy.
// =====================================================================================
constexpr uint32_t SG_LOG = 6, SG = 1u << SG_LOG;      // tiles per group of the two-level aggregate sum
constexpr uint32_t TILE_MAXK = 16;                     // polynomials per tile_combine launch (more: combined first)
constexpr uint32_t TILE_GW = 22;                       // generator powers z^(T 2^s) for the W table: 2^22 tiles
constexpr uint32_t TILE_DIRECT_MAX = 1024;             // up to this many tiles every tile sums all aggregates above it
#ifndef KZG_TILE_ITER
#define KZG_TILE_ITER 4
#endif
// Issue priority of the two tile kernels.  Beside an accumulate kernel they hold a wave slot and ~110 VGPRs per SIMD
// that the reduce stage's 168-VGPR waves then cannot get: at priority 0 the combination of opening p + 1 sat there for
// 1.4 ms and rc1 / rc2 / prep_binsort stretched 4-17x (profiles/r04_open_async_timeline_prio0.txt); raised, it is gone after 0.48 ms:
// 448 vs 430 pipelined opens/s, same box, alternating (profiles/r04_open_async_ab.txt).
#ifndef KZG_TILE_PRIO
#define KZG_TILE_PRIO 3
#endif
constexpr uint32_t TILE_ITER = KZG_TILE_ITER;          // coefficients per thread of the combination (tile_combine_kernel)

struct TileLincomb {
  const uint32_t* polys;
  uint64_t stride;
  uint32_t k;
  uint32_t lens[TILE_MAXK];
  uint32_t xipow[TILE_MAXK * FRN];    // xi^(i+1), Montgomery form
};
struct TileScanArgs {                 // all Montgomery form; 2.5 KB of the 4 KB kernel-argument limit with TileLincomb
  uint32_t zs[SC * FRN];              // z^r, r < 8
  uint32_t zq_lo[16 * FRN];           // z^(8c), c < 16         z^(8q) = zq_lo[q & 15] * zq_hi[q >> 4]
  uint32_t zq_hi[16 * FRN];           // z^(128c), c < 16
  uint32_t ginv[9 * FRN];             // z^(-8 * 2^s), s <= 8   -> zinv[q], q <= 256
  uint32_t gw[TILE_GW * FRN];         // z^(T * 2^s)            -> W[k] = z^(T(k+1))
};

// sum over the 8 lanes of an aligned group (every lane of the group ends with the total): two quad permutes and a
// half-row mirror, no LDS
__device__ __forceinline__ uint32_t lane8_sum(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);     // quad_perm:[1,0,3,2]
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);     // quad_perm:[2,3,0,1]
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);    // row_half_mirror
  return x;
}
// 8 weak-normal values -> their canonical sum (8 * 2p < 2^261, limbs < 2^32), in every lane of the group
template <class F>
__device__ __forceinline__ Fe<F> lane8_sum_reduced(const Fe<F>& v) {
  Fe<F> s;
#pragma unroll
  for (int j = 0; j < F::N; ++j) s.l[j] = lane8_sum(v.l[j]);
  return Field<F>::reduce_wide(Field<F>::carry(s));
}
// Sum of one weak-normal value per thread over the workgroup (all threads call it); valid in thread 0 only.
// red: (TB / 8 + TB / 64) elements of LDS.
template <class F, uint32_t TB>
__device__ __forceinline__ Fe<F> block_sum(const Fe<F>& v, uint32_t* red) {
  using Fd = Field<F>;
  const uint32_t tid = threadIdx.x;
  const Fe<F> s1 = lane8_sum_reduced<F>(v);
  if ((tid & 7) == 0) store_limbs<F>(red + (tid >> 3) * F::N, s1);
  __syncthreads();
  const Fe<F> x = tid < TB / 8 ? load_limbs<F>(red + tid * F::N) : Fd::zero();
  const Fe<F> s2 = lane8_sum_reduced<F>(x);
  if ((tid & 7) == 0 && tid < TB / 8) store_limbs<F>(red + (TB / 8 + (tid >> 3)) * F::N, s2);
  __syncthreads();
  Fe<F> r = Fd::zero();
  if (tid == 0)
    for (uint32_t u = 0; u < TB / 64; ++u) r = Fd::add(r, load_limbs<F>(red + (TB / 8 + u) * F::N));
  return r;
}
// product of the generators g[s] over the set bits s of e (Montgomery form)
template <class F>
__device__ __forceinline__ Fe<F> pow_from_generators(const uint32_t* g, uint32_t e) {
  Fe<F> p = Field<F>::one();
  for (uint32_t s = 0; (e >> s) != 0; ++s)
    if ((e >> s) & 1u) p = Field<F>::mul(p, load_limbs<F>(g + s * F::N));
  return p;
}

// TB chunks (T = 8 TB coefficients) per tile, TK = T / ITER threads per workgroup: the kernel waits for memory, so the
// coefficients of a tile are spread over MORE threads than the fill uses (2 waves per SIMD at one thread per chunk:
// 77 us for the 2^20 x 6 combination; profiles/r04a_open_kernel_stats.csv), each taking ITER strided elements.
// STORE = false: the aggregates alone (kzg_fr_poly_eval: nothing but H and the tables is written).
template <class F, uint32_t TB, uint32_t ITER, bool STORE = true>
__global__ __launch_bounds__(TB * SC / ITER) void tile_combine_kernel(TileLincomb a, TileScanArgs ts, uint32_t* comb,
                                                                      uint32_t n, uint32_t ntiles, uint32_t* G,
                                                                      uint32_t* H, uint32_t* zinv, uint32_t* W) {
  using Fd = Field<F>;
  constexpr uint32_t T = TB * SC, TK = T / ITER;
  static_assert(F::N == FRN && SC == 8 && TB % 64 == 0 && TB <= 256 && TK % 64 == 0 && TK >= TB && TK <= 1024, "tile layout");
  __shared__ uint32_t hs[TB * FRN];                         // chunk sums (lazy limbs)
  __shared__ uint32_t zsh[SC * FRN];
  __shared__ uint32_t red[(TK / 8 + TK / 64) * FRN];
  const uint32_t tid = threadIdx.x;
#if KZG_TILE_PRIO
  __builtin_amdgcn_s_setprio(KZG_TILE_PRIO);
#endif
  const uint32_t ntab = gridDim.x - ntiles;
  if (blockIdx.x < ntab) {       // table workgroups (nothing of pass 1 reads these): FIRST in the grid, so that their
    const uint32_t t = blockIdx.x * TK + tid;               // chains of <= 20 products start at once and end early
    if (t <= TB) store_limbs<F>(zinv + (size_t)t * F::N, pow_from_generators<F>(ts.ginv, t));
    if (t < ntiles) store_limbs<F>(W + (size_t)t * F::N, pow_from_generators<F>(ts.gw, t + 1));
    return;
  }
  const uint32_t b = blockIdx.x - ntab, base = b * T;
  if (tid < SC * FRN) zsh[tid] = ts.zs[tid];
  __syncthreads();
  const Fe<F> zr = load_limbs<F>(zsh + (tid & (SC - 1)) * F::N);
#pragma unroll 1
  for (uint32_t it = 0; it < ITER; ++it) {
    const uint32_t i = it * TK + tid, t = base + i;
    Fe<F> acc = Fd::zero();
    if (t < n) {
      for (uint32_t i0 = 0; i0 < a.k; i0 += DOT_G) {
        const uint32_t cnt = min(DOT_G, a.k - i0);
        Fe<F> c[DOT_G], x[DOT_G];
#pragma unroll
        for (uint32_t g = 0; g < DOT_G; ++g) {
          const uint32_t p = i0 + g;
          const bool on = g < cnt && t < a.lens[g < cnt ? p : i0];           // shorter polynomials read as zero-padded
          c[g] = on ? load_words<F>(a.polys + (a.stride * p + t) * 8) : Fd::zero();
          x[g] = load_limbs<F>(a.xipow + (g < cnt ? p : i0) * F::N);
        }
        const Fe<F> part = dot_upto<F>(cnt, c, x);
        acc = i0 ? Fd::add(acc, part) : part;
      }
      if (STORE) store_words<F>(comb + (size_t)t * 8, acc);
    }
    const Fe<F> e = Fd::mul(acc, zr);                                         // 0 beyond the end
    Fe<F> s;
#pragma unroll
    for (int j = 0; j < F::N; ++j) s.l[j] = lane8_sum(e.l[j]);
    if ((tid & 7) == 0) store_limbs<F>(hs + (i >> 3) * F::N, s);
  }
  __syncthreads();
  Fe<F> g = Fd::zero();
  if (tid < TB) {                                                             // whole waves: TB is a multiple of 64
    const Fe<F> hq = Fd::reduce_wide(Fd::carry(load_limbs<F>(hs + tid * F::N)));
    const Fe<F> zq = Fd::mul(load_limbs<F>(ts.zq_lo + (tid & 15) * F::N), load_limbs<F>(ts.zq_hi + (tid >> 4) * F::N));
    g = Fd::mul(hq, zq);
    if (STORE) store_limbs<F>(G + ((size_t)b * TB + tid) * F::N, g);
  }
  const Fe<F> hb = block_sum<F, TK>(g, red);
  if (tid == 0) store_limbs<F>(H + (size_t)b * F::N, hb);
}

// A[s] = sum_(i < 64) H[64 s + i] z^(T i): one wave per group of tiles
template <class F>
__global__ __launch_bounds__(64) void tile_group_kernel(const uint32_t* H, const uint32_t* W, uint32_t ntiles,
                                                        uint32_t* A) {
  using Fd = Field<F>;
  static_assert(SG == 64, "one wave per group");
  const uint32_t s = blockIdx.x, i = threadIdx.x, bt = s * SG + i;
  Fe<F> v = Fd::zero();
  if (bt < ntiles) {
    v = load_limbs<F>(H + (size_t)bt * F::N);
    if (i) v = Fd::mul(v, load_limbs<F>(W + (size_t)(i - 1) * F::N));
  }
  const Fe<F> x = lane8_sum_reduced<F>(v);
  Fe<F> r = x;
#pragma unroll 1
  for (uint32_t u = 1; u < 8; ++u) {
    Fe<F> t;
#pragma unroll
    for (int j = 0; j < F::N; ++j) t.l[j] = (uint32_t)__shfl((int)x.l[j], (int)(u * 8));
    r = Fd::add(r, t);
  }
  if (i == 0) store_limbs<F>(A + (size_t)s * F::N, r);
}

// Q_b = sum over everything above tile b, each term once: tiles of b's own group, then whole groups (A != null), then
// the caller's carry as the aggregate of a virtual tile `ntiles` (sharded open).  All threads call; valid in thread 0.
template <class F, uint32_t TB>
__device__ __forceinline__ Fe<F> tile_carry_sum(uint32_t b, uint32_t ntiles, const uint32_t* H, const uint32_t* A,
                                                uint32_t nsuper, const FrArg& hv, uint32_t has_hv, const uint32_t* W,
                                                uint32_t* red) {
  using Fd = Field<F>;
  const uint32_t tid = threadIdx.x;
  Fe<F> acc = Fd::zero();
  if (!A) {
    for (uint32_t bp = b + 1 + tid; bp < ntiles; bp += TB)
      acc = Fd::add(acc, Fd::mul(load_limbs<F>(H + (size_t)bp * F::N), load_limbs<F>(W + (size_t)(bp - b - 1) * F::N)));
  } else {
    const uint32_t s = b >> SG_LOG, gend = min((s + 1) << SG_LOG, ntiles);
    for (uint32_t bp = b + 1 + tid; bp < gend; bp += TB)
      acc = Fd::add(acc, Fd::mul(load_limbs<F>(H + (size_t)bp * F::N), load_limbs<F>(W + (size_t)(bp - b - 1) * F::N)));
    for (uint32_t sp = s + 1 + tid; sp < nsuper; sp += TB)
      acc = Fd::add(acc, Fd::mul(load_limbs<F>(A + (size_t)sp * F::N),
                                 load_limbs<F>(W + (size_t)((sp << SG_LOG) - b - 1) * F::N)));
  }
  if (has_hv && tid == TB - 1)
    acc = Fd::add(acc, Fd::mul(load_limbs<F>(hv.l), load_limbs<F>(W + (size_t)(ntiles - b - 1) * F::N)));
  return block_sum<F, TB>(acc, red);
}

template <class F, uint32_t TB>
__global__ __launch_bounds__(TB) void tile_fill_kernel(const uint32_t* comb, uint32_t n, uint32_t ntiles,
                                                       const uint32_t* G, const uint32_t* H, const uint32_t* A,
                                                       uint32_t nsuper, FrArg hv, uint32_t has_hv, FrArg zarg,
                                                       const uint32_t* zinv, const uint32_t* W, uint32_t* S_base) {
  using Fd = Field<F>;
  constexpr uint32_t T = TB * SC, NWAVE = TB / 64;
  __shared__ uint4 st[T * 2 + TB];                          // 32 B per coefficient + one pad quad per chunk
  __shared__ uint32_t red[(TB / 8 + TB / 64) * FRN];
  __shared__ uint32_t wsum[NWAVE * FRN];
  __shared__ uint32_t qsh[FRN];
#if KZG_TILE_PRIO
  __builtin_amdgcn_s_setprio(KZG_TILE_PRIO);
#endif
  const uint32_t tid = threadIdx.x, b = blockIdx.x, e0 = b * T;
  const uint4* gin = reinterpret_cast<const uint4*>(comb);
  for (uint32_t i = tid; i < T * 2; i += TB) {
    const uint32_t e = i >> 1;
    if (e0 + e < n) st[i + (e >> SC_LOG)] = gin[(size_t)(e0 + e) * 2 + (i & 1)];
  }
  const Fe<F> q0 = tile_carry_sum<F, TB>(b, ntiles, H, A, nsuper, hv, has_hv, W, red);
  if (tid == 0) store_limbs<F>(qsh, q0);
  // suffix sums of the tile's scaled chunk values: inclusive inside the wave by shuffles, the waves above from LDS
  const uint32_t lane = tid & 63, wave = tid >> 6;
  Fe<F> incl = load_limbs<F>(G + ((size_t)b * TB + tid) * F::N);
#pragma unroll 1
  for (uint32_t d = 1; d < 64; d <<= 1) {
    Fe<F> t;
#pragma unroll
    for (int j = 0; j < F::N; ++j) t.l[j] = (uint32_t)__shfl_down((int)incl.l[j], d);
    const Fe<F> s = Fd::add(incl, t);
    incl = Fd::select(lane + d < 64, s, incl);
  }
  if (lane == 0) store_limbs<F>(wsum + wave * F::N, incl);
  Fe<F> ex;                                                 // sum over the chunks after this one
#pragma unroll
  for (int j = 0; j < F::N; ++j) {
    const uint32_t t = (uint32_t)__shfl_down((int)incl.l[j], 1);
    ex.l[j] = lane == 63 ? 0u : t;
  }
  __syncthreads();                                          // st, qsh, wsum complete
  for (uint32_t w = wave + 1; w < NWAVE; ++w) ex = Fd::add(ex, load_limbs<F>(wsum + w * F::N));
  Fe<F> acc = Fd::mul(Fd::add(ex, load_limbs<F>(qsh)), load_limbs<F>(zinv + (size_t)(tid + 1) * F::N));
  const uint32_t j0 = e0 + tid * SC;
  if (j0 < n) {
    const Fe<F> z = load_limbs<F>(zarg.l);
    for (uint32_t j = j0 + SC; j-- > j0;) {
      if (j >= n) {                                         // the chunk that holds the end: zero coefficients above it
        if (has_hv) acc = Fd::mul(acc, z);
        continue;
      }
      const uint32_t q = (j - e0) * 2 + tid;                // (j - e0) >> SC_LOG == tid
      const uint4 lo = st[q], hi = st[q + 1];
      const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      acc = Fd::add(Fd::from_words(w), Fd::mul(acc, z));
      uint32_t o[8];
      Fd::to_words(Fd::reduce(acc), o);
      st[q] = make_uint4(o[0], o[1], o[2], o[3]);
      st[q + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
  }
  __syncthreads();
  uint4* gout = reinterpret_cast<uint4*>(S_base);
  for (uint32_t i = tid; i < T * 2; i += TB) {
    const uint32_t e = i >> 1;
    if (e0 + e < n) gout[(size_t)(e0 + e) * 2 + (i & 1)] = st[i + (e >> SC_LOG)];
  }
}

// S_0 alone (the slice evaluation of a sharded open): H_0 + sum_k H_(1+k) W[k]
template <class F, uint32_t TB>
__global__ __launch_bounds__(TB) void tile_eval_kernel(uint32_t ntiles, const uint32_t* H, const uint32_t* A,
                                                       uint32_t nsuper, const uint32_t* W, uint32_t* out_words) {
  using Fd = Field<F>;
  __shared__ uint32_t red[(TB / 8 + TB / 64) * FRN];
  FrArg none{};
  const Fe<F> q0 = tile_carry_sum<F, TB>(0, ntiles, H, A, nsuper, none, 0, W, red);
  if (threadIdx.x == 0) store_words<F>(out_words, Fd::add(q0, load_limbs<F>(H)));
}

// Buffer layout of poly_tmp[0] (canonical words, 8 per element), cap = n + 1:
//   comb[0 .. cap)   combined polynomial
//   eval             S_0
//   quot[0 .. cap)   S_1, S_2, ...        => [eval, quot...] is the contiguous vector S_0, S_1, ...
// scan_tmp: g (9 words per chunk), H (per tile, + 1), A (per group of 64 tiles), zinv (TB + 1), W (per tile).
template <class F>
struct TilePlan {
  uint32_t tb = 0, ntiles = 0, ntab = 0, nsuper = 0;        // nsuper = 0: every tile sums all aggregates above it
  TileScanArgs ts;
  FrArg z;                                                    // Montgomery form
  Fe<F> z_inv;                                                // Montgomery form
  uint32_t *G = nullptr, *H = nullptr, *A = nullptr, *zinv = nullptr, *W = nullptr;
};

template <class F>
static FrArg fr_arg(const Fe<F>& v) {
  FrArg a;
  memcpy(a.l, v.l, F::N * 4);
  return a;
}
static bool words_are_zero(const uint32_t* w) {
  uint32_t acc = 0;
  for (int i = 0; i < 8; ++i) acc |= w[i];
  return acc == 0;
}

// Tile width: 256 threads (2048 coefficients, 65 KiB of LDS in the fill) when the GPU is the opening's; 128 threads
// (33 KiB) while an accumulate kernel of the commit pipeline holds 114 of a CU's 160 KiB, so that the fill still gets
// a workgroup onto every CU (the transform makes the same choice, ntt.hip).  c->tune_open_tb fixes it (tests).
static uint32_t open_tile_threads(Ctx* c) {
  if (c->tune_open_tb == 128 || c->tune_open_tb == 256) return (uint32_t)c->tune_open_tb;
  static const int fixed = [] { const char* e = getenv("KZG_OPEN_TB"); return e ? atoi(e) : 0; }();      // experiments
  if (fixed == 128 || fixed == 256) return (uint32_t)fixed;
  return msm_accumulate_in_flight(c) ? 128u : 256u;
}

template <class F>
int tile_plan(Ctx* c, size_t n, const uint32_t* z_words, uint32_t tb, TilePlan<F>* p, DevBuf* buf = nullptr,
              bool with_chunks = true) {
  using Fd = Field<F>;
  const uint32_t T = tb * SC, tk = T / TILE_ITER;
  p->tb = tb;
  p->ntiles = (uint32_t)((n + T - 1) / T);
  p->ntab = (std::max(tb + 1, p->ntiles) + tk - 1) / tk;
  const uint32_t direct_max = c->tune_open_direct_max > 0 ? (uint32_t)c->tune_open_direct_max : TILE_DIRECT_MAX;
  p->nsuper = p->ntiles > direct_max ? (p->ntiles + SG - 1) / SG : 0;
  const Fe<F> z = Fd::to_mont(Fd::from_words(z_words));
  p->z = fr_arg<F>(z);
  Fe<F> zj = Fd::one();
  for (uint32_t j = 0; j < SC; ++j) { memcpy(&p->ts.zs[j * F::N], zj.l, F::N * 4); zj = Fd::mul(zj, z); }
  const Fe<F> z8 = zj;
  zj = Fd::one();
  for (uint32_t j = 0; j < 16; ++j) { memcpy(&p->ts.zq_lo[j * F::N], zj.l, F::N * 4); zj = Fd::mul(zj, z8); }
  const Fe<F> z128 = zj;
  zj = Fd::one();
  for (uint32_t j = 0; j < 16; ++j) { memcpy(&p->ts.zq_hi[j * F::N], zj.l, F::N * 4); zj = Fd::mul(zj, z128); }
  Fe<F> g = z128;                                             // z^(8 * 16) -> z^(8 * tb) = z^T
  for (uint32_t q = 16; q < tb; q <<= 1) g = Fd::mul(g, g);
  for (uint32_t s = 0; s < TILE_GW; ++s) { memcpy(&p->ts.gw[s * F::N], g.l, F::N * 4); g = Fd::mul(g, g); }
  p->z_inv = Fd::inv(z);
  g = p->z_inv;
  for (uint32_t q = 0; q < SC_LOG; ++q) g = Fd::mul(g, g);    // z^-8
  for (uint32_t s = 0; s < 9; ++s) { memcpy(&p->ts.ginv[s * F::N], g.l, F::N * 4); g = Fd::mul(g, g); }
  const size_t chunk_words = with_chunks ? (size_t)p->ntiles * tb : 0;
  const size_t words = (chunk_words + (p->ntiles + 1) + (p->nsuper + 1) + (tb + 1) + p->ntiles) * F::N;
  DevBuf& b = buf ? *buf : c->scan_tmp;
  int rc = ensure_buf(c, b, words * 4);
  if (rc) return rc;
  p->G = static_cast<uint32_t*>(b.p);
  p->H = p->G + chunk_words * F::N;
  p->A = p->H + (size_t)(p->ntiles + 1) * F::N;
  p->zinv = p->A + (size_t)(p->nsuper + 1) * F::N;
  p->W = p->zinv + (size_t)(tb + 1) * F::N;
  return KZG_OK;
}

// plain combination: out[t] = sum_i xi^(i+1) p_i[t]  (any k <= 64)
template <class F>
int launch_lincomb(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                   const uint32_t* xi_words, uint32_t* d_out, size_t n) {
  using Fd = Field<F>;
  LincombArgs la{};
  la.polys = d_polys; la.stride = stride; la.k = (uint32_t)k;
  const Fe<F> xi = Fd::to_mont(Fd::from_words(xi_words));
  Fe<F> xp = Fd::one();
  for (size_t i = 0; i < k; ++i) {
    xp = Fd::mul(xp, xi);                                   // xi^(i+1): kzg.py:148-150
    memcpy(&la.xipow[i * F::N], xp.l, F::N * 4);
    la.lens[i] = (uint32_t)lens[i];
  }
  hipLaunchKernelGGL(lincomb_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, la, d_out,
                     (uint32_t)n);
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

// pass 1: combination + chunk / tile aggregates + tables.  More than TILE_MAXK polynomials are combined first and
// pass 1 then runs over the combination itself (weight one).
template <class F, uint32_t TB>
int launch_tile_combine(Ctx* c, const TilePlan<F>& p, const uint32_t* d_polys, const size_t* lens, size_t k,
                        size_t stride, const uint32_t* xi_words, uint32_t* d_comb, size_t n) {
  using Fd = Field<F>;
  TileLincomb la{};
  if (k > TILE_MAXK) {
    int rc = launch_lincomb<F>(c, d_polys, lens, k, stride, xi_words, d_comb, n);
    if (rc) return rc;
    la.polys = d_comb; la.stride = n; la.k = 1; la.lens[0] = (uint32_t)n;
    const Fe<F> one = Fd::one();
    memcpy(&la.xipow[0], one.l, F::N * 4);
  } else {
    la.polys = d_polys; la.stride = stride; la.k = (uint32_t)k;
    const Fe<F> xi = Fd::to_mont(Fd::from_words(xi_words));
    Fe<F> xp = Fd::one();
    for (size_t i = 0; i < k; ++i) {
      xp = Fd::mul(xp, xi);                                 // xi^(i+1): kzg.py:148-150
      memcpy(&la.xipow[i * F::N], xp.l, F::N * 4);
      la.lens[i] = (uint32_t)lens[i];
    }
  }
  hipLaunchKernelGGL((tile_combine_kernel<F, TB, TILE_ITER>), dim3(p.ntiles + p.ntab), dim3(TB * SC / TILE_ITER), 0,
                     c->stream, la, p.ts, d_comb, (uint32_t)n, p.ntiles, p.G, p.H, p.zinv, p.W);
  KZG_HIP(c, hipGetLastError());
  if (p.nsuper) {
    hipLaunchKernelGGL(tile_group_kernel<F>, dim3(p.nsuper), dim3(64), 0, c->stream, p.H, p.W, p.ntiles, p.A);
    KZG_HIP(c, hipGetLastError());
  }
  return KZG_OK;
}

template <class F, uint32_t TB>
int launch_tile_fill(Ctx* c, const TilePlan<F>& p, const uint32_t* d_comb, size_t n, const FrArg* hv, uint32_t* d_S) {
  FrArg none{};
  hipLaunchKernelGGL((tile_fill_kernel<F, TB>), dim3(p.ntiles), dim3(TB), 0, c->stream, d_comb, (uint32_t)n, p.ntiles,
                     p.G, p.H, p.nsuper ? p.A : (const uint32_t*)nullptr, p.nsuper, hv ? *hv : none, hv ? 1u : 0u, p.z,
                     p.zinv, p.W, d_S);
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

template <class F>
int check_open_args(Ctx* c, const size_t* lens, size_t k, size_t stride, size_t* n_out) {
  if (k > MAXK) return set_err(c, KZG_ERR_ARG, "kzg_open: more than 64 polynomials");
  size_t n = 0;
  for (size_t i = 0; i < k; ++i) {
    if (lens[i] > stride) return set_err(c, KZG_ERR_ARG, "kzg_open: lens[i] > stride");
    n = std::max(n, lens[i]);
  }
  if (n >= (1ull << 31) - 1) return set_err(c, KZG_ERR_ARG, "kzg_open: polynomial too long");
  *n_out = n;
  return KZG_OK;
}

// witness of kzg.py:153-154: eval <- S_0, quotient coefficient j-1 <- S_j.  ONE "open_poly" span per opening.
// eval_out: host memory that receives S_0 when `sync` (the call then waits for the stream); otherwise nothing is
// copied and nothing waits -- the pipelined open fetches the 32 bytes below the quotient itself (msm.hip).
template <class F>
int open_quotient_t(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                    const uint32_t* z_words, const uint32_t* xi_words, uint32_t** d_quot_out, size_t* quot_len,
                    uint64_t* eval_out, bool sync) {
  if (eval_out) memset(eval_out, 0, 32);
  *quot_len = 0;
  *d_quot_out = nullptr;
  size_t n = 0;
  int rc = check_open_args<F>(c, lens, k, stride, &n);
  if (rc) return rc;
  if (n == 0) return KZG_OK;     // all polynomials zero: witness 0, evaluation 0
  if ((rc = ensure_buf(c, c->poly_tmp[0], (2 * (n + 1) + 1) * 32))) return rc;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  uint32_t* d_eval = d_comb + (n + 1) * 8;
  {
    ProfScope ps(c, "open_poly");
    if (words_are_zero(z_words)) {         // S_j = c_j: the combination IS the vector of suffix values
      if ((rc = launch_lincomb<F>(c, d_polys, lens, k, stride, xi_words, d_eval, n))) return rc;
    } else {
      TilePlan<F> p;
      const uint32_t tb = open_tile_threads(c);
      if ((rc = tile_plan<F>(c, n, z_words, tb, &p))) return rc;
      rc = tb == 128 ? launch_tile_combine<F, 128>(c, p, d_polys, lens, k, stride, xi_words, d_comb, n)
                     : launch_tile_combine<F, 256>(c, p, d_polys, lens, k, stride, xi_words, d_comb, n);
      if (rc) return rc;
      rc = tb == 128 ? launch_tile_fill<F, 128>(c, p, d_comb, n, nullptr, d_eval)
                     : launch_tile_fill<F, 256>(c, p, d_comb, n, nullptr, d_eval);
      if (rc) return rc;
    }
  }
  if (sync && eval_out) {
    KZG_HIP(c, hipMemcpyAsync(eval_out, d_eval, 32, hipMemcpyDeviceToHost, c->stream));
    KZG_HIP(c, hipStreamSynchronize(c->stream));
  }
  *d_quot_out = d_eval + 8;
  *quot_len = n - 1;
  return KZG_OK;
}

// Sharded open, step 1: combine this rank's coefficient slices and evaluate the slice polynomial (local indexing)
// at z.  The combination, its chunk values and tile aggregates stay on the device for step 2.
template <class F>
int open_shard_begin_t(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                       const uint32_t* z_words, const uint32_t* xi_words, uint64_t* chunk_eval_out) {
  memset(chunk_eval_out, 0, 32);
  size_t n = 0;
  int rc = check_open_args<F>(c, lens, k, stride, &n);
  if (rc) return rc;
  c->open_shard_n = n;
  c->open_shard_tb = 0;
  if (n == 0) return KZG_OK;
  if ((rc = ensure_buf(c, c->poly_tmp[0], (2 * (n + 1) + 1) * 32))) return rc;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  uint32_t* d_eval = d_comb + (n + 1) * 8;
  {
    ProfScope ps(c, "open_shard_poly");
    if (words_are_zero(z_words)) {         // slice(0) = its constant coefficient
      if ((rc = launch_lincomb<F>(c, d_polys, lens, k, stride, xi_words, d_comb, n))) return rc;
      KZG_HIP(c, hipMemcpyAsync(chunk_eval_out, d_comb, 32, hipMemcpyDeviceToHost, c->stream));
    } else {
      TilePlan<F> p;
      const uint32_t tb = open_tile_threads(c);
      if ((rc = tile_plan<F>(c, n, z_words, tb, &p))) return rc;
      rc = tb == 128 ? launch_tile_combine<F, 128>(c, p, d_polys, lens, k, stride, xi_words, d_comb, n)
                     : launch_tile_combine<F, 256>(c, p, d_polys, lens, k, stride, xi_words, d_comb, n);
      if (rc) return rc;
      if (tb == 128)
        hipLaunchKernelGGL((tile_eval_kernel<F, 128>), dim3(1), dim3(128), 0, c->stream, p.ntiles, p.H,
                           p.nsuper ? p.A : (const uint32_t*)nullptr, p.nsuper, p.W, d_eval);
      else
        hipLaunchKernelGGL((tile_eval_kernel<F, 256>), dim3(1), dim3(256), 0, c->stream, p.ntiles, p.H,
                           p.nsuper ? p.A : (const uint32_t*)nullptr, p.nsuper, p.W, d_eval);
      KZG_HIP(c, hipGetLastError());
      c->open_shard_tb = tb;
      KZG_HIP(c, hipMemcpyAsync(chunk_eval_out, d_eval, 32, hipMemcpyDeviceToHost, c->stream));
    }
  }
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  return KZG_OK;
}

// step 2: the carry S_hi of the ranks above enters the fill as the aggregate of a virtual tile above the slice
// (exact: S_j of the slice depends on what lies above only through S_hi).  Returns the vector to commit.
template <class F>
int open_shard_finish_t(Ctx* c, const uint32_t* z_words, const uint32_t* carry_words, int first_rank,
                        uint32_t** d_vec_out, size_t* vec_len, uint64_t* eval_out) {
  using Fd = Field<F>;
  memset(eval_out, 0, 32);
  *d_vec_out = nullptr;
  *vec_len = 0;
  const size_t n = c->open_shard_n;
  if (n == 0) return KZG_OK;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  uint32_t* d_eval = d_comb + (n + 1) * 8;
  int rc;
  {
    ProfScope ps(c, "open_shard_poly");
    if (words_are_zero(z_words)) {         // S_j = c_j for every j below the slice's end
      KZG_HIP(c, hipMemcpyAsync(d_eval, d_comb, n * 32, hipMemcpyDeviceToDevice, c->stream));
    } else {
      const uint32_t tb = c->open_shard_tb;
      if (tb != 128 && tb != 256) return set_err(c, KZG_ERR_ARG, "kzg_open_shard_finish without kzg_open_shard_begin");
      TilePlan<F> p;
      if ((rc = tile_plan<F>(c, n, z_words, tb, &p))) return rc;       // same z, same tiles: same buffers and tables
      // virtual tile `ntiles`: its aggregate is S at index ntiles * T, i.e. carry * z^-(ntiles * T - n)
      const size_t gap = (size_t)p.ntiles * tb * SC - n;
      Fe<F> hv = Fd::from_words(carry_words), zi = p.z_inv;
      for (size_t e = gap; e; e >>= 1) {
        if (e & 1) hv = Fd::mul(hv, zi);
        zi = Fd::mul(zi, zi);
      }
      const FrArg hva = fr_arg<F>(hv);
      rc = tb == 128 ? launch_tile_fill<F, 128>(c, p, d_comb, n, &hva, d_eval)
                     : launch_tile_fill<F, 256>(c, p, d_comb, n, &hva, d_eval);
      if (rc) return rc;
    }
  }
  KZG_HIP(c, hipMemcpyAsync(eval_out, d_eval, 32, hipMemcpyDeviceToHost, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  if (first_rank) {            // S_0 is the evaluation; the quotient slice is S_1 .. S_(n-1)
    *d_vec_out = d_eval + 8;
    *vec_len = n - 1;
  } else {                     // S_lo is the quotient coefficient lo-1: commit S_lo .. S_(hi-1)
    *d_vec_out = d_eval;
    *vec_len = n;
  }
  return KZG_OK;
}

}  // namespace

int open_quotient_device(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                         const uint32_t* z_words, const uint32_t* xi_words, uint32_t** d_quot_out, size_t* quot_len,
                         uint64_t* eval_out, bool sync) {
  return c->curve == 0 ? open_quotient_t<BnFr>(c, d_polys, lens, k, stride, z_words, xi_words, d_quot_out, quot_len,
                                               eval_out, sync)
                       : open_quotient_t<BlsFr>(c, d_polys, lens, k, stride, z_words, xi_words, d_quot_out, quot_len,
                                                eval_out, sync);
}

int open_shard_begin_device(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                            const uint32_t* z_words, const uint32_t* xi_words, uint64_t* chunk_eval_out) {
  return c->curve == 0 ? open_shard_begin_t<BnFr>(c, d_polys, lens, k, stride, z_words, xi_words, chunk_eval_out)
                       : open_shard_begin_t<BlsFr>(c, d_polys, lens, k, stride, z_words, xi_words, chunk_eval_out);
}
int open_shard_finish_device(Ctx* c, const uint32_t* z_words, const uint32_t* carry_words, int first_rank,
                             uint32_t** d_vec_out, size_t* vec_len, uint64_t* eval_out) {
  return c->curve == 0 ? open_shard_finish_t<BnFr>(c, z_words, carry_words, first_rank, d_vec_out, vec_len, eval_out)
                       : open_shard_finish_t<BlsFr>(c, z_words, carry_words, first_rank, d_vec_out, vec_len, eval_out);
}

// true iff any of the elements [from, to) of a canonical-word array is non-zero
int device_any_nonzero(Ctx* c, const uint32_t* d_words, size_t from, size_t to, bool* out) {
  *out = false;
  if (from >= to) return KZG_OK;
  uint32_t* d_flag = nullptr;
  KZG_HIP(c, hipMalloc(reinterpret_cast<void**>(&d_flag), 4));
  hipMemsetAsync(d_flag, 0, 4, c->stream);
  hipLaunchKernelGGL(any_nonzero_kernel, dim3((uint32_t)((to - from + 255) / 256)), dim3(256), 0, c->stream, d_words,
                     from, to, d_flag);
  uint32_t f = 0;
  hipMemcpyAsync(&f, d_flag, 4, hipMemcpyDeviceToHost, c->stream);
  hipError_t e = hipStreamSynchronize(c->stream);
  hipFree(d_flag);
  if (e != hipSuccess) return set_err(c, KZG_ERR_HIP, "any_nonzero", e);
  *out = f != 0;
  return KZG_OK;
}

}  // namespace kzg

// =====================================================================================
// Device polynomial / vector primitives over Fr (include/kzg_mi355x.h "kzg_fr_*").  What the
// reference's callers get from Sage's dense polynomial arithmetic (plonk/prover.py:243-316:
// accumulator ratios, products, division by Z_H on a coset) expressed as data-parallel passes
// over device-resident coefficient / evaluation vectors.
// =====================================================================================
namespace kzg {
namespace {

enum : int { VEC_ADD = 0, VEC_SUB = 1, VEC_MUL = 2 };

template <class F>
__global__ void vec_binary_kernel(int op, size_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  using Fd = Field<F>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe<F> x = load_words<F>(a + i * 8), y = load_words<F>(b + i * 8);
  Fe<F> r;
  if (op == VEC_ADD) r = Fd::add(x, y);
  else if (op == VEC_SUB) r = Fd::sub(x, y);
  else r = Fd::mul(Fd::to_mont(x), y);          // (xR)*y/R = x*y
  store_words<F>(out + i * 8, r);
}

// Everything a lincomb launch needs travels in the kernel arguments (3.1 KB of the 4 KB limit): no
// staging copy, no host synchronisation, so a chain of vector operations is enqueued without stalls.
constexpr int FR_LIMBS = 9;       // both scalar fields: 9 x 29-bit limbs
struct ScalarLincombArgs {
  const uint32_t* ptr[MAXK];
  uint32_t len[MAXK];
  uint32_t scal[MAXK * FR_LIMBS];   // s_j, Montgomery form
  uint32_t k;
};
// out[i] = sum_j s_j * p_j[i]  (s_j in Montgomery form; p_j shorter than n count as zero-padded); groups of DOT_G
// terms share one Montgomery reduction
template <class F>
__global__ void vec_lincomb_kernel(ScalarLincombArgs a, uint32_t* out, uint32_t n) {
  using Fd = Field<F>;
  static_assert(F::N == FR_LIMBS, "scalar fields have 9 limbs");
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fe<F> acc = Fd::zero();
  for (uint32_t j0 = 0; j0 < a.k; j0 += DOT_G) {
    const uint32_t cnt = min(DOT_G, a.k - j0);
    Fe<F> c[DOT_G], x[DOT_G];
#pragma unroll
    for (uint32_t g = 0; g < DOT_G; ++g) {
      const uint32_t j = j0 + g;
      const bool on = g < cnt && t < a.len[j];
      c[g] = on ? load_words<F>(a.ptr[j] + (size_t)t * 8) : Fd::zero();
      x[g] = load_limbs<F>(a.scal + (g < cnt ? j : j0) * F::N);
    }
    const Fe<F> part = dot_upto<F>(cnt, c, x);
    acc = j0 ? Fd::add(acc, part) : part;
  }
  store_words<F>(out + (size_t)t * 8, acc);
}

// out[i] = a[i] * c * s^i : thread handles LC consecutive i (one pow per chunk, then a running product)
struct PowArgs {
  uint32_t s[FR_LIMBS], c[FR_LIMBS];   // Montgomery form, in the kernel arguments
  uint32_t step[FR_LIMBS];             // s^(threads of the launch)
};
// Thread t takes the elements t, t + T, t + 2T, .. (T = threads of the launch): consecutive lanes touch consecutive
// 32-byte elements (round 2 gave a thread LC consecutive elements -- a 1 KB lane stride, four times the bytes through
// the memory system and 0.5 ms per 2^22 elements in the prover's coset shifts).  p = c * s^t by square-and-multiply
// once, then p *= s^T per element (sc.step, computed by the host).
template <class F>
__global__ void vec_mul_powers_kernel(size_t n, const uint32_t* a, PowArgs sc, uint32_t* out) {
  using Fd = Field<F>;
  static_assert(F::N == FR_LIMBS, "scalar fields have 9 limbs");
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fe<F> b = load_limbs<F>(sc.s), p = load_limbs<F>(sc.c);   // p = c (Montgomery)
  for (size_t bits = t; bits; bits >>= 1) {                 // p = c * s^t
    if (bits & 1u) p = Fd::mul(p, b);
    b = Fd::mul(b, b);
  }
  const Fe<F> step = load_limbs<F>(sc.step);
  for (size_t i = t; i < n; i += T) {
    store_words<F>(out + i * 8, Fd::mul(load_words<F>(a + i * 8), p));
    p = Fd::mul(p, step);
  }
}

// out[i] = a[i]^-1 (0 -> 0), Fermat: one exponentiation per element.  Used when out aliases a.
template <class F>
__global__ void vec_inverse_fermat_kernel(size_t n, const uint32_t* a, uint32_t* out) {
  using Fd = Field<F>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Fe<F> x = Fd::to_mont(load_words<F>(a + i * 8));
  store_words<F>(out + i * 8, Fd::from_mont(Fd::inv(x)));
}
// The same by batch inversion (Montgomery's trick) over LC elements per thread -- the elements t, t + T, t + 2T, ..
// (T = threads of the launch), so that consecutive lanes touch consecutive elements (round 2 batched LC CONSECUTIVE
// elements: a 1 KB lane stride).  The output buffer first receives the running products of the non-zero elements
// (Montgomery form), one exponentiation inverts the batch product, and the backward sweep peels the inverses off:
// 5 multiplications per element + 1/LC of an exponentiation instead of a whole one (~380).
template <class F>
__global__ void vec_inverse_kernel(size_t n, const uint32_t* a, uint32_t* out) {
  using Fd = Field<F>;
  const size_t T = (size_t)gridDim.x * blockDim.x;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fe<F> acc = Fd::one();
  size_t last = t;
  for (size_t i = t; i < n; i += T) {
    store_words<F>(out + i * 8, acc);                       // product of the non-zero elements of the batch before i
    const Fe<F> x = Fd::to_mont(load_words<F>(a + i * 8));
    if (!Fd::is_zero(x)) acc = Fd::mul(acc, x);
    last = i;
  }
  Fe<F> inv = Fd::inv(acc);                                 // acc is a product of non-zero elements (or one)
  for (size_t i = last;; i -= T) {
    const Fe<F> x = Fd::to_mont(load_words<F>(a + i * 8));
    if (Fd::is_zero(x)) {
      store_words<F>(out + i * 8, Fd::zero());
    } else {
      const Fe<F> pre = load_words<F>(out + i * 8);
      store_words<F>(out + i * 8, Fd::from_mont(Fd::mul(inv, pre)));
      inv = Fd::mul(inv, x);
    }
    if (i == t) break;
  }
}

// Exclusive prefix product out[i] = prod_{j<i} a[j], three steps with chunks of LC:
//   1 chunk products (and level-up while more than LC chunks remain), 2 serial scan of the top,
//   3 refill.  Values travel in Montgomery form between the steps.
template <class F, bool WORDS_IN>
__global__ void chunk_prod_kernel(const uint32_t* in, uint32_t m, uint32_t* prod) {
  using Fd = Field<F>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j0 = t * LC;
  if (j0 >= m) return;
  const uint32_t j1 = min(j0 + LC, m);
  Fe<F> acc = Fd::one();
  for (uint32_t j = j0; j < j1; ++j)
    acc = Fd::mul(acc, WORDS_IN ? Fd::to_mont(load_words<F>(in + (size_t)j * 8)) : load_limbs<F>(in + (size_t)j * F::N));
  store_limbs<F>(prod + (size_t)t * F::N, acc);
}
template <class F>
__global__ void top_prefix_kernel(const uint32_t* in, uint32_t m, uint32_t* pre) {   // pre[j] = prod_{i<j} in[i]
  using Fd = Field<F>;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  Fe<F> acc = Fd::one();
  for (uint32_t j = 0; j < m; ++j) {
    store_limbs<F>(pre + (size_t)j * F::N, acc);
    acc = Fd::mul(acc, load_limbs<F>(in + (size_t)j * F::N));
  }
}
// pre_out[j] for j in chunk t = pre_up[t] * prod_{chunk start <= i < j} in[i]
template <class F, bool FINAL>
__global__ void chunk_prefix_fill_kernel(const uint32_t* in, uint32_t m, const uint32_t* pre_up, uint32_t* pre_out) {
  using Fd = Field<F>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j0 = t * LC;
  if (j0 >= m) return;
  const uint32_t j1 = min(j0 + LC, m);
  Fe<F> acc = load_limbs<F>(pre_up + (size_t)t * F::N);
  for (uint32_t j = j0; j < j1; ++j) {
    if (FINAL) {
      store_words<F>(pre_out + (size_t)j * 8, Fd::from_mont(acc));
      acc = Fd::mul(acc, Fd::to_mont(load_words<F>(in + (size_t)j * 8)));
    } else {
      store_limbs<F>(pre_out + (size_t)j * F::N, acc);
      acc = Fd::mul(acc, load_limbs<F>(in + (size_t)j * F::N));
    }
  }
}

template <class F>
int vec_binary_t(Ctx* c, int op, size_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  if (n == 0) return KZG_OK;
  hipLaunchKernelGGL(vec_binary_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, op, n, a, b, out);
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

template <class F>
int vec_lincomb_t(Ctx* c, size_t n, size_t k, const uint32_t* const* ptrs, const size_t* lens, const uint32_t* scalars,
                  uint32_t* out) {
  using Fd = Field<F>;
  if (k > MAXK) return set_err(c, KZG_ERR_ARG, "kzg_fr_vec_lincomb: more than 64 terms");
  if (n == 0) return KZG_OK;
  if (n >= (1ull << 32)) return set_err(c, KZG_ERR_ARG, "vector too long");
  ScalarLincombArgs la{};
  la.k = (uint32_t)k;
  for (size_t j = 0; j < k; ++j) {
    const Fe<F> s = Fd::to_mont(Fd::from_words(scalars + j * 8));
    memcpy(&la.scal[j * F::N], s.l, F::N * 4);
    la.ptr[j] = ptrs[j];
    la.len[j] = (uint32_t)std::min(lens[j], n);
  }
  hipLaunchKernelGGL(vec_lincomb_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, la, out,
                     (uint32_t)n);
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

template <class F>
int vec_mul_powers_t(Ctx* c, size_t n, const uint32_t* a, const uint32_t* s_words, const uint32_t* c_words,
                     uint32_t* out) {
  using Fd = Field<F>;
  if (n == 0) return KZG_OK;
  const Fe<F> s = Fd::to_mont(Fd::from_words(s_words)), cc = Fd::to_mont(Fd::from_words(c_words));
  PowArgs sc;
  memcpy(sc.s, s.l, F::N * 4);
  memcpy(sc.c, cc.l, F::N * 4);
  // LC elements per thread; the thread count is a power of two so that s^T is log2(T) squarings on the host
  uint32_t lt = 7;
  while (((size_t)LC << lt) < n && lt < 24) ++lt;
  const size_t T = (size_t)1 << lt;
  Fe<F> step = s;
  for (uint32_t q = 0; q < lt; ++q) step = Fd::mul(step, step);
  memcpy(sc.step, step.l, F::N * 4);
  hipLaunchKernelGGL(vec_mul_powers_kernel<F>, dim3((uint32_t)(T / 128)), dim3(128), 0, c->stream, n, a, sc, out);
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

template <class F>
int vec_inverse_t(Ctx* c, size_t n, const uint32_t* a, uint32_t* out) {
  if (n == 0) return KZG_OK;
  if (a == out) {     // in place: the batch kernel needs the output buffer as scratch next to the input
    hipLaunchKernelGGL(vec_inverse_fermat_kernel<F>, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, c->stream, n, a,
                       out);
  } else {
    const size_t chunks = (n + LC - 1) / LC;
    hipLaunchKernelGGL(vec_inverse_kernel<F>, dim3((uint32_t)((chunks + 63) / 64)), dim3(64), 0, c->stream, n, a, out);
  }
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

template <class F>
int vec_prefix_product_t(Ctx* c, size_t n, const uint32_t* a, uint32_t* out) {
  if (n == 0) return KZG_OK;
  if (n >= (1ull << 31)) return set_err(c, KZG_ERR_ARG, "vector too long");
  std::vector<uint32_t> m{(uint32_t)n};
  while (m.back() > LC) m.push_back((m.back() + LC - 1) / LC);
  const size_t nl = m.size();
  size_t total = 0;
  for (size_t l = 1; l < nl; ++l) total += m[l];
  int rc;
  if ((rc = ensure_buf(c, c->poly_tmp[2], (total + 1) * F::N * 4))) return rc;      // chunk products per level
  if ((rc = ensure_buf(c, c->poly_tmp[3], (total + LC + 1) * F::N * 4))) return rc; // prefixes per level
  uint32_t* d_p = static_cast<uint32_t*>(c->poly_tmp[2].p);
  uint32_t* d_q = static_cast<uint32_t*>(c->poly_tmp[3].p);
  std::vector<uint32_t*> pp(nl, nullptr), qq(nl, nullptr);
  {
    uint32_t* x = d_p; uint32_t* y = d_q;
    for (size_t l = 1; l < nl; ++l) { pp[l] = x; x += (size_t)m[l] * F::N; qq[l] = y; y += (size_t)m[l] * F::N; }
  }
  auto grid = [](uint32_t chunks) { return dim3((chunks + 127) / 128); };
  if (nl == 1) {
    // a single chunk: its incoming prefix is 1
    const Fe<F> one = Field<F>::one();
    KZG_HIP(c, hipMemcpyAsync(d_q, one.l, F::N * 4, hipMemcpyHostToDevice, c->stream));
    KZG_HIP(c, hipStreamSynchronize(c->stream));
    hipLaunchKernelGGL((chunk_prefix_fill_kernel<F, true>), dim3(1), dim3(64), 0, c->stream, a, m[0], d_q, out);
  } else {
    hipLaunchKernelGGL((chunk_prod_kernel<F, true>), grid(m[1]), dim3(128), 0, c->stream, a, m[0], pp[1]);
    for (size_t l = 1; l + 1 < nl; ++l)
      hipLaunchKernelGGL((chunk_prod_kernel<F, false>), grid(m[l + 1]), dim3(128), 0, c->stream, pp[l], m[l], pp[l + 1]);
    hipLaunchKernelGGL(top_prefix_kernel<F>, dim3(1), dim3(64), 0, c->stream, pp[nl - 1], m[nl - 1], qq[nl - 1]);
    for (size_t l = nl - 2; l >= 1; --l)
      hipLaunchKernelGGL((chunk_prefix_fill_kernel<F, false>), grid(m[l + 1]), dim3(128), 0, c->stream, pp[l], m[l],
                         qq[l + 1], qq[l]);
    hipLaunchKernelGGL((chunk_prefix_fill_kernel<F, true>), grid(m[1]), dim3(128), 0, c->stream, a, m[0], qq[1], out);
  }
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

// p(z) for a coefficient vector: pass 1 of the opening's scan without its stores (tile aggregates only), then the
// one-workgroup sum of the aggregates -- two launches (rounds 1-3 collapsed chunks of 8 level by level: seven)
template <class F>
int poly_eval_t(Ctx* c, size_t n, const uint32_t* a, const uint32_t* z_words, uint64_t* out) {
  using Fd = Field<F>;
  memset(out, 0, 32);
  if (n == 0) return KZG_OK;
  if (n >= (1ull << 31) - 1) return set_err(c, KZG_ERR_ARG, "vector too long");
  if (n == 1 || words_are_zero(z_words)) {      // the constant coefficient
    KZG_HIP(c, hipMemcpyAsync(out, a, 32, hipMemcpyDeviceToHost, c->stream));
    KZG_HIP(c, hipStreamSynchronize(c->stream));
    return KZG_OK;
  }
  TilePlan<F> p;
  int rc = tile_plan<F>(c, n, z_words, 256, &p, &c->poly_tmp[2], /*with_chunks=*/false);
  if (rc) return rc;
  if ((rc = ensure_buf(c, c->poly_tmp[3], 32))) return rc;
  uint32_t* d_val = static_cast<uint32_t*>(c->poly_tmp[3].p);
  TileLincomb la{};
  la.polys = a; la.stride = n; la.k = 1; la.lens[0] = (uint32_t)n;
  const Fe<F> one = Fd::one();
  memcpy(&la.xipow[0], one.l, F::N * 4);
  hipLaunchKernelGGL((tile_combine_kernel<F, 256, TILE_ITER, false>), dim3(p.ntiles + p.ntab), dim3(256 * SC / TILE_ITER),
                     0, c->stream, la, p.ts, (uint32_t*)nullptr, (uint32_t)n, p.ntiles, (uint32_t*)nullptr, p.H, p.zinv,
                     p.W);
  if (p.nsuper)
    hipLaunchKernelGGL(tile_group_kernel<F>, dim3(p.nsuper), dim3(64), 0, c->stream, p.H, p.W, p.ntiles, p.A);
  hipLaunchKernelGGL((tile_eval_kernel<F, 256>), dim3(1), dim3(256), 0, c->stream, p.ntiles, p.H,
                     p.nsuper ? p.A : (const uint32_t*)nullptr, p.nsuper, p.W, d_val);
  KZG_HIP(c, hipGetLastError());
  KZG_HIP(c, hipMemcpyAsync(out, d_val, 32, hipMemcpyDeviceToHost, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  return KZG_OK;
}

}  // namespace

#define KZG_FR_DISPATCH(fn, ...) (c->curve == 0 ? fn<BnFr>(__VA_ARGS__) : fn<BlsFr>(__VA_ARGS__))
int fr_vec_binary(Ctx* c, int op, size_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  return KZG_FR_DISPATCH(vec_binary_t, c, op, n, a, b, out);
}
int fr_vec_lincomb(Ctx* c, size_t n, size_t k, const uint32_t* const* ptrs, const size_t* lens, const uint32_t* scalars,
                   uint32_t* out) {
  return KZG_FR_DISPATCH(vec_lincomb_t, c, n, k, ptrs, lens, scalars, out);
}
int fr_vec_mul_powers(Ctx* c, size_t n, const uint32_t* a, const uint32_t* s, const uint32_t* cc, uint32_t* out) {
  return KZG_FR_DISPATCH(vec_mul_powers_t, c, n, a, s, cc, out);
}
int fr_vec_inverse(Ctx* c, size_t n, const uint32_t* a, uint32_t* out) { return KZG_FR_DISPATCH(vec_inverse_t, c, n, a, out); }
int fr_vec_prefix_product(Ctx* c, size_t n, const uint32_t* a, uint32_t* out) {
  return KZG_FR_DISPATCH(vec_prefix_product_t, c, n, a, out);
}
int fr_poly_eval(Ctx* c, size_t n, const uint32_t* a, const uint32_t* z, uint64_t* out) {
  return KZG_FR_DISPATCH(poly_eval_t, c, n, a, z, out);
}

}  // namespace kzg
