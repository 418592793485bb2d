// msm.hip -- KZG commitment = multi-scalar multiplication sum_i p_i * ck[i] on gfx950.
//
// Replaces the hot loop of the reference's KZG.commit (kzg.py:112-116: one
// py_ecc double-and-add `multiply` per coefficient plus a running `add`) with a
// Pippenger bucket method laid out for 288 GB of HBM:
//
//   SRS load (once per commitment key): for every point P_i the table holds the
//     W = ceil(256/c) multiples 2^(c*j) * P_i in affine Montgomery form, one
//     128-byte record each (BLS12-381; one cache line per gather).  With the
//     multiples precomputed ALL windows feed ONE set of 2^(c-1) buckets: no
//     per-window bucket sets, no doublings between windows at commit time.
//     c = 20 (13 windows, 2^19 buckets) for keys of >= 2^18 points, else c = 16.
//   commit, three stages on three internal streams (prep(p+1) | accumulate(p) | reduce(p-1) share
//   the GPU; see "Commit pipeline" below and DESIGN.md 4.2):
//     P prep        msm_prep.hip: scalar -> W signed c-bit digits (zero digits dropped,
//                   kzg.py:113-114), table indices grouped by bucket, buckets ordered by length,
//                   slices of <= SEG entries
//     A accumulate  one lane per slice: gathers records (LDS-DMA prefetch) and runs mixed XYZZ
//                   additions; persistent grid, 2 waves per SIMD        <- the dominant kernel
//     B finalize    a few lanes per multi-slice bucket fold slice partials (most buckets are one
//                   slice and were written by A)
//       reduce      sum_v v*B_v with v = hi*2^LO + lo:  2^LO * sum_hi hi*R_hi +
//                   sum_lo lo*C_lo from row sums R and column sums C of the bucket
//                   matrix (2 additions per bucket, tree-shaped), then bit-plane
//                   sums of the short R and C vectors
//       host        Horner over the ~20 partial points, one inversion to affine
#include <cstring>
#include <cstdlib>
#include <string.h>
#include <algorithm>
#include <vector>
#include <type_traits>
#include <utility>
#include "internal.h"
#include "ec.h"
#include "msm.h"
#include "msm_prep.h"

namespace kzg {

namespace {

// Window configuration.  Scalars are < 2^255, so the top digit never wraps.
template <int WB>
struct Win {
  static constexpr int BITS = WB;
  static constexpr int NWIN = (256 + WB - 1) / WB;            // 16 -> 16 windows, 20 -> 13
  static constexpr uint32_t NB = 1u << (WB - 1);              // bucket k holds digit magnitude k+1
  static constexpr uint32_t SEG = WB >= 20 ? 64 : 32;         // max entries per accumulate thread
  static constexpr uint32_t FIN = WB >= 20 ? 2 : 8;           // lanes per bucket in finalize
  static constexpr int LO = WB / 2;                           // bucket matrix: 2^HI rows x 2^LO columns
  static constexpr int HI = WB - 1 - LO;
  static constexpr int NPART = HI + LO + 1;                   // points handed to the host
};
constexpr int MAX_NPART = 24;

template <class C> struct Rec {
  static constexpr int WORDS = C::REC_WORDS;
  static constexpr int N = C::Fp::N;
  static constexpr int FLAG = 2 * N;   // word index of the flags (bit 0: infinity)
};

// x, y and the flag word of a record (bit 0: point at infinity)
template <class C>
__device__ __forceinline__ uint32_t load_rec(const uint32_t* recs, size_t idx, Fe<typename C::Fp>& x,
                                             Fe<typename C::Fp>& y) {
  constexpr int N = C::Fp::N;
  const uint32_t* p = recs + idx * Rec<C>::WORDS;
  uint32_t flag;
  constexpr int Q = (2 * N + 3) / 4;                 // 16-byte loads covering x and y
  uint32_t w[4 * Q];
  if constexpr (4 * Q <= Rec<C>::WORDS && (Rec<C>::WORDS % 4) == 0) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < Q; ++i) {
      const uint4 v = q[i];
      w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    if constexpr (Rec<C>::FLAG < 4 * Q) flag = w[Rec<C>::FLAG]; else flag = p[Rec<C>::FLAG];
  } else {
    const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int i = 0; i < (2 * N) / 2; ++i) {
      const uint2 v = q[i];
      w[2 * i] = v.x; w[2 * i + 1] = v.y;
    }
    flag = p[Rec<C>::FLAG];
  }
#pragma unroll
  for (int j = 0; j < N; ++j) { x.l[j] = w[j]; y.l[j] = w[N + j]; }
  return flag;
}

template <class C>
__device__ __forceinline__ void store_rec(uint32_t* recs, size_t idx, const Fe<typename C::Fp>& x,
                                          const Fe<typename C::Fp>& y, bool inf) {
  constexpr int N = C::Fp::N;
  uint32_t* p = recs + idx * Rec<C>::WORDS;
#pragma unroll
  for (int j = 0; j < N; ++j) { p[j] = inf ? 0u : x.l[j]; p[N + j] = inf ? 0u : y.l[j]; }
#pragma unroll
  for (int j = 2 * N; j < Rec<C>::WORDS; ++j) p[j] = 0;
  p[Rec<C>::FLAG] = inf ? 1u : 0u;
}

// XYZZ points are 4*N words = a whole number of 16-byte quads; every array of them is 16-byte aligned
template <class C>
__device__ __forceinline__ XYZZ<C> load_xyzz(const uint32_t* base, size_t idx) {
  constexpr int N = C::Fp::N;
  static_assert((4 * N) % 4 == 0, "XYZZ must be a whole number of quads");
  const uint4* p = reinterpret_cast<const uint4*>(base + idx * 4 * N);
  uint32_t w[4 * N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const uint4 v = p[q];
    w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
  XYZZ<C> r;
#pragma unroll
  for (int j = 0; j < N; ++j) { r.x.l[j] = w[j]; r.y.l[j] = w[N + j]; r.zz.l[j] = w[2 * N + j]; r.zzz.l[j] = w[3 * N + j]; }
  return r;
}
template <class C>
__device__ __forceinline__ void store_xyzz(uint32_t* base, size_t idx, const XYZZ<C>& v) {
  constexpr int N = C::Fp::N;
  uint32_t w[4 * N];
#pragma unroll
  for (int j = 0; j < N; ++j) { w[j] = v.x.l[j]; w[N + j] = v.y.l[j]; w[2 * N + j] = v.zz.l[j]; w[3 * N + j] = v.zzz.l[j]; }
  uint4* p = reinterpret_cast<uint4*>(base + idx * 4 * N);
#pragma unroll
  for (int q = 0; q < N; ++q) p[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
}
template <class C>
__device__ __forceinline__ XYZZ<C> shfl_xor_xyzz(const XYZZ<C>& v, int mask) {
  constexpr int N = C::Fp::N;
  XYZZ<C> r;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    r.x.l[j] = __shfl_xor(v.x.l[j], mask);
    r.y.l[j] = __shfl_xor(v.y.l[j], mask);
    r.zz.l[j] = __shfl_xor(v.zz.l[j], mask);
    r.zzz.l[j] = __shfl_xor(v.zzz.l[j], mask);
  }
  return r;
}

// ---- SRS table construction ------------------------------------------------------

// canonical affine words (x | y, NW 32-bit words each) -> window-0 records
template <class C>
__global__ void srs_import_kernel(const uint32_t* xy, const uint8_t* inf, uint32_t* recs, size_t n,
                                  uint32_t* bad_count) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool is_inf = inf && inf[i];
  Fe<F> x = Fd::zero(), y = Fd::zero();
  if (!is_inf) {
    uint32_t wx[F::NW], wy[F::NW];
#pragma unroll
    for (int k = 0; k < F::NW; ++k) { wx[k] = xy[i * 2 * F::NW + k]; wy[k] = xy[i * 2 * F::NW + F::NW + k]; }
    x = Fd::reduce(Fd::to_mont(Fd::from_words(wx)));
    y = Fd::reduce(Fd::to_mont(Fd::from_words(wy)));
    if (!Ec<C>::on_curve(x, y)) atomicAdd(bad_count, 1u);
  }
  store_rec<C>(recs, i, x, y, is_inf);
}

// records of window j -> window j+1: multiply every point by 2^win_bits
template <class C>
__global__ __launch_bounds__(128) void srs_window_kernel(const uint32_t* src, uint32_t* dst, size_t n, int win_bits) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<F> x, y;
  load_rec<C>(src, i, x, y);
  const bool is_inf = src[i * Rec<C>::WORDS + Rec<C>::FLAG] & 1u;
  XYZZ<C> p = is_inf ? Ec<C>::infinity() : Ec<C>::dbl_affine(x, y);
  for (int d = 1; d < win_bits; ++d) p = Ec<C>::dbl(p);
  const Affine<C> a = Ec<C>::to_affine(p);
  store_rec<C>(dst, i, Fd::reduce(a.x), Fd::reduce(a.y), a.inf);
}

// window-0 records -> canonical affine words (kzg_srs_export)
template <class C>
__global__ void srs_export_kernel(const uint32_t* recs, size_t start, size_t count, uint32_t* xy, uint8_t* inf) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  Fe<F> x, y;
  load_rec<C>(recs, start + i, x, y);
  const bool is_inf = recs[(start + i) * Rec<C>::WORDS + Rec<C>::FLAG] & 1u;
  uint32_t wx[F::NW], wy[F::NW];
  Fd::to_words(Fd::from_mont(x), wx);
  Fd::to_words(Fd::from_mont(y), wy);
#pragma unroll
  for (int k = 0; k < F::NW; ++k) {
    xy[i * 2 * F::NW + k] = is_inf ? 0u : wx[k];
    xy[i * 2 * F::NW + F::NW + k] = is_inf ? 0u : wy[k];
  }
  inf[i] = is_inf ? 1 : 0;
}

// fixed-base table for kzg_srs_generate: tab[j][d-1] = d * 2^(8j) * G, d = 1..255, j = 0..31.
// Block j, thread d-1; every thread recomputes 2^(8j) G (cheap, one-time, fully parallel).
template <class C>
__global__ __launch_bounds__(256) void gen_table_kernel(const uint32_t* g_rec, uint32_t* tab) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  const uint32_t j = blockIdx.x;       // window of 8 bits
  const uint32_t d = threadIdx.x + 1;
  if (d > 255) return;
  Fe<F> gx, gy;
  load_rec<C>(g_rec, 0, gx, gy);
  XYZZ<C> base;
  base.x = gx; base.y = gy; base.zz = Fd::one(); base.zzz = Fd::one();
  for (uint32_t q = 0; q < 8 * j; ++q) base = Ec<C>::dbl(base);
  XYZZ<C> acc = Ec<C>::infinity();
  for (int bit = 7; bit >= 0; --bit) {
    acc = Ec<C>::dbl(acc);
    if ((d >> bit) & 1u) acc = Ec<C>::add(acc, base);
  }
  const Affine<C> a = Ec<C>::to_affine(acc);
  store_rec<C>(tab, (size_t)j * 255 + (d - 1), Fd::reduce(a.x), Fd::reduce(a.y), a.inf);
}

// SRS generation (kzg.py:70-72): record i = tau^i * G by fixed-base windows of 8 bits.
// powers: canonical words of tau^i (computed by pow kernel below)
template <class C>
__global__ __launch_bounds__(128) void srs_generate_kernel(const uint32_t* tab, const uint32_t* tau_mont,
                                                           uint32_t* recs, size_t start, size_t n, size_t run_len,
                                                           size_t inner_stride, size_t outer_stride) {
  using F = typename C::Fp;
  using Fr = typename C::Fr;
  using Fd = Field<F>;
  using Frd = Field<Fr>;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // s = tau^i (square-and-multiply on the index bits), canonical words
  Fe<Fr> b, acc = Frd::one();
#pragma unroll
  for (int j = 0; j < Fr::N; ++j) b.l[j] = tau_mont[j];
  // record i holds tau^e, e = start + (i / run_len) * outer_stride + (i % run_len) * inner_stride
  // (a contiguous range: run_len = n, inner_stride = 1)
  for (size_t bits = start + (i / run_len) * outer_stride + (i % run_len) * inner_stride; bits; bits >>= 1) {
    if (bits & 1u) acc = Frd::mul(acc, b);
    b = Frd::mul(b, b);
  }
  uint32_t s[Fr::NW];
  Frd::to_words(Frd::from_mont(acc), s);
  XYZZ<C> p = Ec<C>::infinity();
  for (int j = 0; j < 32; ++j) {
    const uint32_t d = (s[j >> 2] >> (8 * (j & 3))) & 0xffu;
    if (d) {
      Fe<F> x, y;
      load_rec<C>(tab, (size_t)j * 255 + (d - 1), x, y);
      p = Ec<C>::madd(p, x, y);
    }
  }
  const Affine<C> a = Ec<C>::to_affine(p);
  store_rec<C>(recs, i, Fd::reduce(a.x), Fd::reduce(a.y), a.inf);
}

// ---- commit pipeline ----------------------------------------------------------------

// Record staging for the accumulate kernel: the 16-byte quads of a record that hold x, y and the
// flag word go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR destination, so the next
// record is in flight during the whole mixed addition although the kernel sits at its VGPR budget).
// One wave-instruction writes 64 lanes x 16 B contiguously: the image of a wave is [quad][lane][4 words].
template <class C>
struct RecStage {
  static constexpr int N = C::Fp::N;
  static constexpr int Q = (2 * N + 1 + 3) / 4;            // quads covering x, y, flag
  static constexpr int WAVE_WORDS = Q * 64 * 4;
  static_assert(4 * Q <= Rec<C>::WORDS && (Rec<C>::WORDS % 4) == 0, "records are whole 16-byte quads");

  // The instruction's immediate offset moves the global source AND the LDS destination
  // (tools/microbench/glds_offset.hip), so one address register pair serves all quads: quad q is
  // requested at offset 16q with the LDS base pulled back by the same 16q.
  static __device__ __forceinline__ void issue(const uint32_t* recs, uint32_t idx, uint32_t* stage) {
    const uint32_t* src = recs + (size_t)idx * Rec<C>::WORDS;
    issue_quads(src, stage, std::make_integer_sequence<int, Q>());
  }
  template <int... Qs>
  static __device__ __forceinline__ void issue_quads(const uint32_t* src, uint32_t* stage,
                                                     std::integer_sequence<int, Qs...>) {
    (__builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) uint32_t*)(stage + Qs * (256 - 4)), 16,
                                      16 * Qs, 0),
     ...);
  }
  // call after `s_waitcnt vmcnt(0)`; returns the flag word
  static __device__ __forceinline__ uint32_t read(const uint32_t* stage, uint32_t lane, Fe<typename C::Fp>& x,
                                                  Fe<typename C::Fp>& y) {
    uint32_t w[4 * Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const uint4 t = *reinterpret_cast<const uint4*>(stage + q * 256 + lane * 4);
      w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) { x.l[j] = w[j]; y.l[j] = w[N + j]; }
    return w[Rec<C>::FLAG];
  }
};

// Persistent kernel: the grid is a fixed number of waves per SIMD (msm_enqueue), each wave draws
// chunks of 64 consecutive slices (one per lane; equal lengths, longest first) from a counter until
// none are left.  Capping its share of every SIMD is what lets the next polynomial's prep kernels
// and the previous one's reduce stage run beside it: a grid of one workgroup per 128 slices keeps
// every wave slot refilled for 2 ms and starves them (measured: the sort made no progress).
// Inner loop, software-pipelined one entry deep over two LDS stages per wave: while entry e is
// added, the record of entry e+1 travels to the other stage and the table index of entry e+2 to a
// register.
template <class C, int WB>
__global__ __launch_bounds__(128, 3) void msm_accumulate_kernel(const uint32_t* recs, const uint32_t* vals,
                                                             const uint32_t* bstart, const uint32_t* order,
                                                             const uint32_t* slice_off, uint32_t* partials,
                                                             uint32_t* buckets, uint32_t* chunk_counter,
                                                             const uint32_t* chunk_rank, unsigned long long* clk) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  using St = RecStage<C>;
  constexpr uint32_t NB = Win<WB>::NB;
  __shared__ __attribute__((aligned(16))) uint32_t stage_all[2 * 2 * St::WAVE_WORDS];   // [wave][stage]
  const uint32_t total = slice_off[NB];
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t* stage = stage_all + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (2 * St::WAVE_WORDS);
  // profiling only (clk is null otherwise): wave 0 of workgroup 0 reports the shader clock it ran at, as the
  // ratio of s_memtime (shader clock) to s_memrealtime (100 MHz) ticks over its lifetime
  const bool probe = clk != nullptr && blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0;
  const long long c0 = probe ? clock64() : 0, w0 = probe ? wall_clock64() : 0;
  for (;;) {
    uint32_t chunk = 0;
    if (lane == 0) chunk = atomicAdd(chunk_counter, 1u);
    chunk = __builtin_amdgcn_readfirstlane(chunk);
    if ((uint64_t)chunk * 64 >= total) {               // every wave reaches this: the counter only grows
      if (probe && lane == 0) {
        atomicAdd(clk, (unsigned long long)(clock64() - c0));
        atomicAdd(clk + 1, (unsigned long long)(wall_clock64() - w0));
      }
      break;
    }
    const uint32_t t = chunk * 64 + lane;
    if (t >= total) continue;
    // rank of the bucket of slice t: largest r with slice_off[r] <= t.  Every non-empty bucket has a
    // slice, so it lies within `lane` ranks of the chunk's first one (chunk_rank, from stage P).
    uint32_t lo = chunk_rank[chunk], hi = min(lo + lane + 1u, NB);
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (slice_off[mid] <= t) lo = mid; else hi = mid;
    }
    const uint32_t r = lo;
    const uint32_t k = order[r];
    const uint32_t s = t - slice_off[r];
    const uint32_t ns = slice_off[r + 1] - slice_off[r];
    const uint32_t b0 = bstart[k];
    const uint32_t len = bstart[k + 1] - b0;
    uint32_t e = b0 + (uint32_t)(((uint64_t)s * len) / ns);
    const uint32_t e1 = b0 + (uint32_t)(((uint64_t)(s + 1) * len) / ns);
    // a bucket that is a single slice (the common case) is final; the others go through finalize.
    // One register carries the destination through the loop: bit 31 = "bucket k", else partial t.
    const uint32_t dst = ns == 1 ? (k | 0x80000000u) : t;
    // The accumulator starts as "infinity" (finite == false) with ZZ = ZZZ = 1 already in place: the first
    // finite entry then only copies its coordinates (the Montgomery one is not re-materialised per iteration).
    XYZZ<C> acc;
    acc.x = Fd::zero(); acc.y = Fd::zero(); acc.zz = Fd::one(); acc.zzz = Fd::one();
    bool finite = false;                       // accumulator is the point at infinity
    uint32_t v_cur = 0, v_next = 0;
    uint32_t it = 0;                            // the same for every lane still in the loop
    if (e < e1) {
      v_cur = vals[e];
      if (e + 1 < e1) v_next = vals[e + 1];
      St::issue(recs, v_cur & 0x7fffffffu, stage);       // stage 0; the previous chunk's reads have completed
    }
    while (e < e1) {
      // record e is in stage it & 1, v_next in its register; every earlier LDS read has returned
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const uint32_t* cur = stage + (it & 1u) * St::WAVE_WORDS;
      const uint32_t v = v_cur;
      ++e;
      ++it;
      if (e < e1) {
        v_cur = v_next;
        St::issue(recs, v_cur & 0x7fffffffu, stage + (it & 1u) * St::WAVE_WORDS);
        if (e + 1 < e1) v_next = vals[e + 1];
      }
      Fe<F> x, y;
      const uint32_t flag = St::read(cur, lane, x, y);    // table coordinates are canonical
      if (flag & 1u) continue;                            // key point at infinity contributes nothing
      if (!finite) {
        acc.x = x; acc.y = Fd::cneg_canonical(y, v >> 31);
        finite = true;
      } else {
        acc = Ec<C>::madd_finite(acc, x, y, (v >> 31) != 0, finite);
        if (!finite) { acc.zz = Fd::one(); acc.zzz = Fd::one(); }     // back to "infinity": restore the invariant
      }
    }
    if (!finite) acc = Ec<C>::infinity();
    store_xyzz<C>((dst >> 31) ? buckets : partials, dst & 0x7fffffffu, acc);
  }
}

// The reduce-stage kernels run beside the persistent accumulate kernel, which holds 2 waves x 160
// VGPRs (153 used, BLS12-381) of every SIMD: 192 of the 512 are left, so the reduce kernels are
// compiled to the 168 of a 3-waves-per-SIMD budget and one of their waves fits every SIMD.
#ifndef KZG_SIDE_WAVES
#define KZG_SIDE_WAVES 3
#endif
#define KZG_SIDE_VGPRS __attribute__((amdgpu_waves_per_eu(KZG_SIDE_WAVES, KZG_SIDE_WAVES)))
#ifndef KZG_REDUCE_PRIO
#define KZG_REDUCE_PRIO 3
#endif

// Skewed scalars (many equal or small coefficients) put thousands of slices into a few buckets.
// Buckets come in length order, so those are the first ranks: the first HEAVY_RANKS ranks whose
// slice count exceeds HEAVY_NS are folded by a whole 256-thread block each, the rest by FIN lanes.
constexpr uint32_t HEAVY_RANKS = 256;
constexpr uint32_t HEAVY_NS = 64;

template <class C, int WB>
__global__ __launch_bounds__(256) KZG_SIDE_VGPRS void msm_finalize_heavy_kernel(const uint32_t* partials, const uint32_t* order,
                                                                 const uint32_t* slice_off, uint32_t* buckets) {
  __builtin_amdgcn_s_setprio(KZG_REDUCE_PRIO);   // short stage beside the long accumulate kernel: win instruction issue
  constexpr int N = C::Fp::N;
  __shared__ __attribute__((aligned(16))) uint32_t xch[3 * 4 * N];
  const uint32_t r = blockIdx.x;
  const uint32_t p0 = slice_off[r], p1 = slice_off[r + 1];
  if (p1 - p0 <= HEAVY_NS) return;                 // whole block exits together
  XYZZ<C> acc = Ec<C>::infinity();
  for (uint32_t p = p0 + threadIdx.x; p < p1; p += blockDim.x) acc = Ec<C>::add(acc, load_xyzz<C>(partials, p));
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) acc = Ec<C>::add(acc, shfl_xor_xyzz<C>(acc, m));
  const uint32_t wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0 && wave) store_xyzz<C>(xch, wave - 1, acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 0; k < 3; ++k) acc = Ec<C>::add(acc, load_xyzz<C>(xch, k));
    store_xyzz<C>(buckets, order[r], acc);
  }
}

// FIN lanes per bucket (in length order) fold the slice partials of the buckets that have more than
// one slice, and empty buckets are set to infinity.  Single-slice buckets were written by the
// accumulate kernel itself.  Slice counts do not increase with the rank, so a wave whose first and
// last ranks both have exactly one slice has nothing to do - with uniform scalars, nearly all.
template <class C, int WB>
__global__ __launch_bounds__(128) KZG_SIDE_VGPRS void msm_finalize_kernel(const uint32_t* partials, const uint32_t* order,
                                                           const uint32_t* slice_off, uint32_t* buckets) {
  __builtin_amdgcn_s_setprio(KZG_REDUCE_PRIO);   // short stage beside the long accumulate kernel: win instruction issue
  constexpr uint32_t FIN = Win<WB>::FIN, NB = Win<WB>::NB;
  static_assert(NB % (64 / FIN) == 0, "a wave covers whole ranks inside the bucket range");
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = gt / FIN, g = gt % FIN;
  if (r >= NB) return;                               // whole waves (NB*FIN is a multiple of 64)
  const uint32_t r_first = (gt & ~63u) / FIN, r_last = (gt | 63u) / FIN;
  if (slice_off[r_first + 1] - slice_off[r_first] == 1 && slice_off[r_last + 1] - slice_off[r_last] == 1) return;
  const uint32_t p0 = slice_off[r], p1 = slice_off[r + 1];
  const uint32_t cnt = p1 - p0;
  const bool mine = cnt != 1 && !(r < HEAVY_RANKS && cnt > HEAVY_NS);   // heavy buckets: the other kernel
  XYZZ<C> acc = Ec<C>::infinity();
  if (mine)
    for (uint32_t p = p0 + g; p < p1; p += FIN) acc = Ec<C>::add(acc, load_xyzz<C>(partials, p));
#pragma unroll
  for (int m = 1; m < (int)FIN; m <<= 1) acc = Ec<C>::add(acc, shfl_xor_xyzz<C>(acc, m));
  if (g == 0 && mine) store_xyzz<C>(buckets, order[r], acc);
}

// Row and column sums of the bucket matrix (entry v = hi*2^LO + lo is buckets[v-1]; v = 0 is
// the empty digit), in two steps with every lane busy:
//   step 1: a lane adds RC_CH entries of one column (blockIdx.y = 0) or one row (= 1) serially;
//           consecutive lanes touch consecutive buckets.  colpart[chunk][lo], rowpart[hi][chunk].
//   step 2: RC_L2 lanes fold the partials of one column / row (serial + log2(RC_L2) shuffle levels).
constexpr uint32_t RC_CH = 8;      // serial depth of step 1 (latency: the stage is a chain of point additions)
constexpr uint32_t RC_L2 = 16;

template <class C, int WB>
__global__ __launch_bounds__(128) KZG_SIDE_VGPRS void msm_rc1_kernel(const uint32_t* buckets, uint32_t* colpart,
                                                      uint32_t* rowpart) {
  __builtin_amdgcn_s_setprio(KZG_REDUCE_PRIO);   // short stage beside the long accumulate kernel: win instruction issue
  constexpr int LO = Win<WB>::LO;
  constexpr uint32_t NCHR = (1u << LO) / RC_CH;            // chunks per row (interleaved)
  const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
  if (L >= Win<WB>::NB / RC_CH) return;
  XYZZ<C> acc = Ec<C>::infinity();
  if (blockIdx.y == 0) {          // column lo, rows chunk*RC_CH .. +RC_CH-1
    const uint32_t lo = L & ((1u << LO) - 1), chunk = L >> LO;
    for (uint32_t e = 0; e < RC_CH; ++e) {
      const uint32_t v = ((chunk * RC_CH + e) << LO) + lo;
      if (v) acc = Ec<C>::add(acc, load_xyzz<C>(buckets, v - 1));
    }
    store_xyzz<C>(colpart, L, acc);
  } else {                        // row hi, columns c, c + NCHR, c + 2*NCHR, ...
    const uint32_t c = L & (NCHR - 1), hi = L / NCHR;
    for (uint32_t e = 0; e < RC_CH; ++e) {
      const uint32_t v = (hi << LO) + e * NCHR + c;
      if (v) acc = Ec<C>::add(acc, load_xyzz<C>(buckets, v - 1));
    }
    store_xyzz<C>(rowpart, L, acc);
  }
}

template <class C, int WB>
__global__ __launch_bounds__(128) KZG_SIDE_VGPRS void msm_rc2_kernel(const uint32_t* colpart, const uint32_t* rowpart,
                                                      uint32_t* colsum, uint32_t* rowsum) {
  __builtin_amdgcn_s_setprio(KZG_REDUCE_PRIO);   // short stage beside the long accumulate kernel: win instruction issue
  constexpr int LO = Win<WB>::LO, HI = Win<WB>::HI;
  constexpr uint32_t NCHC = (1u << HI) / RC_CH, NCHR = (1u << LO) / RC_CH;
  const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t vec = gt / RC_L2, g = gt % RC_L2;         // vectors 0..2^LO-1: columns, then rows
  XYZZ<C> acc = Ec<C>::infinity();
  const bool is_col = vec < (1u << LO);
  const uint32_t row = vec - (1u << LO);
  if (is_col) {
    for (uint32_t q = g; q < NCHC; q += RC_L2) acc = Ec<C>::add(acc, load_xyzz<C>(colpart, (q << LO) + vec));
  } else if (row < (1u << HI)) {
    for (uint32_t q = g; q < NCHR; q += RC_L2) acc = Ec<C>::add(acc, load_xyzz<C>(rowpart, row * NCHR + q));
  }
#pragma unroll
  for (int m = 1; m < (int)RC_L2; m <<= 1) acc = Ec<C>::add(acc, shfl_xor_xyzz<C>(acc, m));
  if (g == 0) {
    if (is_col) store_xyzz<C>(colsum, vec, acc);
    else if (row < (1u << HI)) store_xyzz<C>(rowsum, row, acc);
  }
}

// bit-plane sums of the row-sum and column-sum vectors: block b < HI: T = sum of rowsum[i]
// with bit b of i set; block HI + b: same over colsum; last block: the top bucket (v = 2^(WB-1)).
// Two waves per block: short serial part, 6 shuffle levels, one LDS hop.
template <class C, int WB>
__global__ __launch_bounds__(128) KZG_SIDE_VGPRS void msm_planes_kernel(const uint32_t* rowsum, const uint32_t* colsum,
                                                         const uint32_t* buckets, uint32_t* out) {
  __builtin_amdgcn_s_setprio(KZG_REDUCE_PRIO);   // short stage beside the long accumulate kernel: win instruction issue
  constexpr int LO = Win<WB>::LO, HI = Win<WB>::HI, N = C::Fp::N;
  __shared__ __attribute__((aligned(16))) uint32_t xch[4 * N];
  const uint32_t blk = blockIdx.x, tid = threadIdx.x;
  XYZZ<C> acc = Ec<C>::infinity();
  if (blk < (uint32_t)(HI + LO)) {
    const bool cols = blk >= (uint32_t)HI;
    const uint32_t b = cols ? blk - HI : blk;
    const uint32_t members = (cols ? (1u << LO) : (1u << HI)) >> 1;
    const uint32_t* src = cols ? colsum : rowsum;
    for (uint32_t q = tid; q < members; q += blockDim.x) {
      const uint32_t i = ((q >> b) << (b + 1)) | (1u << b) | (q & ((1u << b) - 1));   // q-th index with bit b set
      acc = Ec<C>::add(acc, load_xyzz<C>(src, i));
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc = Ec<C>::add(acc, shfl_xor_xyzz<C>(acc, m));
    if (tid == 64) store_xyzz<C>(xch, 0, acc);
    __syncthreads();
    if (tid == 0) acc = Ec<C>::add(acc, load_xyzz<C>(xch, 0));
  } else {
    acc = load_xyzz<C>(buckets, Win<WB>::NB - 1);
  }
  if (tid == 0) store_xyzz<C>(out, blk, acc);
}

template <class C>
size_t rec_bytes() { return (size_t)C::REC_WORDS * 4; }

int pick_win_bits(size_t n) { return n >= (1u << 18) ? 20 : 16; }
int nwin_of(int wb) { return (256 + wb - 1) / wb; }

}  // namespace

// ---- host-side drivers ----------------------------------------------------------------

template <class C>
static int srs_build_windows(Ctx* c, Srs* s) {
  const size_t n = s->n;
  const uint32_t blocks = (uint32_t)((n + 127) / 128);
  for (int j = 1; j < s->nwin; ++j) {
    hipLaunchKernelGGL(srs_window_kernel<C>, dim3(blocks), dim3(128), 0, c->stream,
                       s->recs + (size_t)(j - 1) * n * C::REC_WORDS, s->recs + (size_t)j * n * C::REC_WORDS, n,
                       s->win_bits);
    KZG_HIP(c, hipGetLastError());
  }
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  return KZG_OK;
}

static Srs* srs_alloc(Ctx* c, size_t n) {
  Srs* s = new Srs();
  s->n = n;
  s->curve = c->curve;
  s->win_bits = pick_win_bits(n);
  s->nwin = nwin_of(s->win_bits);
  return s;
}

template <class C>
static int srs_load_t(Ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, Srs** out) {
  using F = typename C::Fp;
  if (n == 0 || n * (size_t)16 >= (1ull << 31)) return set_err(c, KZG_ERR_ARG, "kzg_srs_load_g1: bad size");
  Srs* s = srs_alloc(c, n);
  const size_t table_bytes = (size_t)s->nwin * n * rec_bytes<C>();
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->recs), table_bytes);
  if (e != hipSuccess) { delete s; return set_err(c, KZG_ERR_ALLOC, "hipMalloc(SRS table)", e); }
  uint32_t* d_xy = nullptr; uint8_t* d_inf = nullptr; uint32_t* d_bad = nullptr;
  const size_t xy_bytes = n * 2 * F::NW * 4;
  auto cleanup = [&]() { hipFree(d_xy); hipFree(d_inf); hipFree(d_bad); };
  auto fail = [&](int rc) { cleanup(); hipFree(s->recs); delete s; return rc; };
  if (hipMalloc(reinterpret_cast<void**>(&d_xy), xy_bytes) != hipSuccess) return fail(set_err(c, KZG_ERR_ALLOC, "hipMalloc"));
  if (hipMalloc(reinterpret_cast<void**>(&d_bad), 4) != hipSuccess) return fail(set_err(c, KZG_ERR_ALLOC, "hipMalloc"));
  if (inf && hipMalloc(reinterpret_cast<void**>(&d_inf), n) != hipSuccess) return fail(set_err(c, KZG_ERR_ALLOC, "hipMalloc"));
  hipMemcpyAsync(d_xy, xy, xy_bytes, hipMemcpyHostToDevice, c->stream);
  if (inf) hipMemcpyAsync(d_inf, inf, n, hipMemcpyHostToDevice, c->stream);
  hipMemsetAsync(d_bad, 0, 4, c->stream);
  hipLaunchKernelGGL(srs_import_kernel<C>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d_xy, d_inf,
                     s->recs, n, d_bad);
  uint32_t bad = 0;
  hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream);
  e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(set_err(c, KZG_ERR_HIP, "srs import", e));
  if (bad) return fail(set_err(c, KZG_ERR_ARG, "kzg_srs_load_g1: point not on the curve"));
  cleanup();
  int rc = srs_build_windows<C>(c, s);
  if (rc) { hipFree(s->recs); delete s; return rc; }
  *out = s;
  return KZG_OK;
}

template <class C>
static int srs_generate_t(Ctx* c, const uint32_t* tau_words, size_t start, size_t n, const uint64_t* gen_xy,
                          Srs** out, size_t run_len, size_t inner_stride, size_t outer_stride) {
  using F = typename C::Fp;
  using Fr = typename C::Fr;
  if (n == 0 || n * (size_t)16 >= (1ull << 31)) return set_err(c, KZG_ERR_ARG, "kzg_srs_generate: bad size");
  Srs* s = srs_alloc(c, n);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->recs), (size_t)s->nwin * n * rec_bytes<C>());
  if (e != hipSuccess) { delete s; return set_err(c, KZG_ERR_ALLOC, "hipMalloc(SRS table)", e); }
  uint32_t *d_g = nullptr, *d_tab = nullptr, *d_tau = nullptr, *d_xy = nullptr, *d_bad = nullptr;
  auto cleanup = [&]() { hipFree(d_g); hipFree(d_tab); hipFree(d_tau); hipFree(d_xy); hipFree(d_bad); };
  auto fail = [&](int rc) { cleanup(); hipFree(s->recs); delete s; return rc; };
  if (hipMalloc(reinterpret_cast<void**>(&d_g), rec_bytes<C>()) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&d_tab), 32 * 255 * rec_bytes<C>()) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&d_tau), Fr::N * 4) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&d_xy), 2 * F::NW * 4) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&d_bad), 4) != hipSuccess)
    return fail(set_err(c, KZG_ERR_ALLOC, "hipMalloc"));
  const Fe<Fr> tau = Field<Fr>::to_mont(Field<Fr>::from_words(tau_words));
  hipMemcpy(d_tau, tau.l, Fr::N * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_xy, gen_xy, 2 * F::NW * 4, hipMemcpyHostToDevice);
  hipMemsetAsync(d_bad, 0, 4, c->stream);
  hipLaunchKernelGGL(srs_import_kernel<C>, dim3(1), dim3(64), 0, c->stream, d_xy, (const uint8_t*)nullptr, d_g,
                     (size_t)1, d_bad);
  hipLaunchKernelGGL(gen_table_kernel<C>, dim3(32), dim3(256), 0, c->stream, d_g, d_tab);
  hipLaunchKernelGGL(srs_generate_kernel<C>, dim3((uint32_t)((n + 127) / 128)), dim3(128), 0, c->stream, d_tab, d_tau,
                     s->recs, start, n, run_len, inner_stride, outer_stride);
  e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(set_err(c, KZG_ERR_HIP, "srs generate", e));
  cleanup();
  int rc = srs_build_windows<C>(c, s);
  if (rc) { hipFree(s->recs); delete s; return rc; }
  *out = s;
  return KZG_OK;
}

static const uint64_t GEN_BN254[8] = {1, 0, 0, 0, 2, 0, 0, 0};
static const uint64_t GEN_BLS[12] = {
    0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull, 0xc3688c4f9774b905ull,
    0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull,
    0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull, 0xfcf5e095d5d00af6ull,
    0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};

int srs_load(Ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, Srs** out) {
  return c->curve == 0 ? srs_load_t<Bn254>(c, xy, inf, n, out) : srs_load_t<Bls12_381>(c, xy, inf, n, out);
}
int srs_generate(Ctx* c, const uint64_t* tau, size_t start, size_t n, Srs** out, size_t run_len, size_t inner_stride,
                 size_t outer_stride) {
  const uint32_t* t = reinterpret_cast<const uint32_t*>(tau);
  if (run_len == 0) { run_len = n ? n : 1; inner_stride = 1; outer_stride = 0; }
  return c->curve == 0 ? srs_generate_t<Bn254>(c, t, start, n, GEN_BN254, out, run_len, inner_stride, outer_stride)
                       : srs_generate_t<Bls12_381>(c, t, start, n, GEN_BLS, out, run_len, inner_stride, outer_stride);
}
void srs_free(Srs* s) {
  if (!s) return;
  hipFree(s->recs);
  delete s;
}

template <class C>
static int srs_export_t(Ctx* c, const Srs* s, size_t start, size_t count, uint64_t* xy, uint8_t* inf) {
  using F = typename C::Fp;
  if (start + count > s->n) return set_err(c, KZG_ERR_ARG, "kzg_srs_export: range");
  if (count == 0) return KZG_OK;
  uint32_t* d_xy = nullptr; uint8_t* d_inf = nullptr;
  KZG_HIP(c, hipMalloc(reinterpret_cast<void**>(&d_xy), count * 2 * F::NW * 4));
  KZG_HIP(c, hipMalloc(reinterpret_cast<void**>(&d_inf), count));
  hipLaunchKernelGGL(srs_export_kernel<C>, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, c->stream, s->recs,
                     start, count, d_xy, d_inf);
  hipMemcpyAsync(xy, d_xy, count * 2 * F::NW * 4, hipMemcpyDeviceToHost, c->stream);
  hipMemcpyAsync(inf, d_inf, count, hipMemcpyDeviceToHost, c->stream);
  hipError_t e = hipStreamSynchronize(c->stream);
  hipFree(d_xy); hipFree(d_inf);
  if (e != hipSuccess) return set_err(c, KZG_ERR_HIP, "srs export", e);
  return KZG_OK;
}
int srs_export(Ctx* c, const Srs* s, size_t start, size_t count, uint64_t* xy, uint8_t* inf) {
  return c->curve == 0 ? srs_export_t<Bn254>(c, s, start, count, xy, inf)
                       : srs_export_t<Bls12_381>(c, s, start, count, xy, inf);
}

// Commit pipeline: three stages on three internal streams, up to four polynomials in flight.
//   P  prep        digits, two-step partition, bucket order, slices (memory-bound, msm_prep.hip)
//   A  accumulate  the mixed-addition kernel                        (ALU-bound, persistent)
//   B  reduce      finalize, row/column sums, bit planes, copy-out  (latency-bound)
// prep(p+1), accumulate(p) and reduce(p-1) run concurrently ON THE SAME SIMDs: A holds 2 waves x
// 160 VGPRs of each, P and B workgroups are sized to fit into what is left.  The context's stream only
// carries ordering: P waits for everything enqueued on it before the call (the scalars), and it
// waits for P to have consumed the scalars, so later work on the context's stream (the next NTT)
// can neither race with prep nor queue behind accumulate.  Every buffer belongs to a slot; the
// host finishes a polynomial (Horner + one inversion) when its slot is recycled or on flush.
constexpr int NSLOT = 4;

struct MsmSlot {
  DevBuf prep_ws, vals, bstart, order, slice_off, counter, chunk_rank;              // prep
  DevBuf scal;                 // the polynomial's scalars, copied at enqueue (KZG_COMMIT_COPY_SCALARS)
  DevBuf partials, buckets, rowsum, colsum, rowpart, colpart, tb;
  void* h_tb = nullptr;        // pinned host copy of the partial points (+ 32 bytes behind them: P(z) of a pipelined open)
  uint64_t* out_eval = nullptr;   // where that P(z) goes when the slot is retired
  hipEvent_t ev_in = nullptr;  // inputs ready on the context's stream
  hipEvent_t ev_p = nullptr;   // prep done
  hipEvent_t ev_a = nullptr;   // accumulate done
  hipEvent_t ev_b = nullptr;   // reduce done (h_tb ready)
  bool pending = false;
  int win_bits = 0;
  uint64_t* out_xy = nullptr;
  uint8_t* out_inf = nullptr;
};

struct MsmWork {
  MsmSlot slot[NSLOT];
  hipStream_t stream_p = nullptr, stream_a = nullptr, stream_b = nullptr;
  int next = 0;
  uint32_t acc_blocks = 0;     // grid of the persistent accumulate kernel
};

static MsmWork* get_work(Ctx* c) {
  if (!c->msm_work) c->msm_work = new MsmWork();
  return static_cast<MsmWork*>(c->msm_work);
}
// true while a queued polynomial's accumulate kernel has not finished (its persistent workgroups hold 114 of a CU's
// 160 KiB of LDS): what another kernel launched now would have to fit beside (ntt.hip: tile size)
bool msm_accumulate_in_flight(Ctx* c) {
  MsmWork* w = static_cast<MsmWork*>(c->msm_work);
  if (!w) return false;
  for (auto& sl : w->slot)
    if (sl.pending && sl.ev_a && hipEventQuery(sl.ev_a) == hipErrorNotReady) return true;
  (void)hipGetLastError();
  return false;
}

void msm_free_work(Ctx* c) {
  MsmWork* w = static_cast<MsmWork*>(c->msm_work);
  if (!w) return;
  for (auto& sl : w->slot) {
    for (DevBuf* b : {&sl.scal, &sl.prep_ws, &sl.vals, &sl.bstart, &sl.order, &sl.slice_off, &sl.counter, &sl.chunk_rank, &sl.partials,
                      &sl.buckets, &sl.rowsum, &sl.colsum, &sl.rowpart, &sl.colpart, &sl.tb})
      hipFree(b->p);
    if (sl.h_tb) hipHostFree(sl.h_tb);
    for (hipEvent_t e : {sl.ev_in, sl.ev_p, sl.ev_a, sl.ev_b})
      if (e) hipEventDestroy(e);
  }
  for (hipStream_t st : {w->stream_p, w->stream_a, w->stream_b})
    if (st) hipStreamDestroy(st);
  delete w;
  c->msm_work = nullptr;
}

template <class C, int WB>
static int msm_enqueue(Ctx* c, const Srs* s, const uint32_t* d_scalars, uint32_t n, MsmWork* w, int slot_idx) {
  using W = Win<WB>;
  constexpr size_t PT = 4 * C::Fp::N * 4;   // bytes of one XYZZ
  constexpr uint32_t NB = W::NB;
  MsmSlot& sl = w->slot[slot_idx];
  const uint32_t m = n * W::NWIN;
  const uint32_t max_slices = m / W::SEG + NB + 1;
  int rc;
  if ((rc = ensure_buf(c, sl.prep_ws, msm_prep_workspace_bytes(n, WB)))) return rc;
  if ((rc = ensure_buf(c, sl.vals, (size_t)m * 4))) return rc;
  if ((rc = ensure_buf(c, sl.bstart, (size_t)(NB + 2) * 4))) return rc;
  if ((rc = ensure_buf(c, sl.order, (size_t)NB * 4))) return rc;
  if ((rc = ensure_buf(c, sl.slice_off, (size_t)(NB + 2) * 4))) return rc;
  if ((rc = ensure_buf(c, sl.partials, (size_t)max_slices * PT))) return rc;
  if ((rc = ensure_buf(c, sl.buckets, (size_t)NB * PT))) return rc;
  if ((rc = ensure_buf(c, sl.rowsum, ((size_t)1 << W::HI) * PT))) return rc;
  if ((rc = ensure_buf(c, sl.colsum, ((size_t)1 << W::LO) * PT))) return rc;
  if ((rc = ensure_buf(c, sl.rowpart, (size_t)(NB / RC_CH) * PT))) return rc;
  if ((rc = ensure_buf(c, sl.colpart, (size_t)(NB / RC_CH) * PT))) return rc;
  if ((rc = ensure_buf(c, sl.tb, (size_t)MAX_NPART * PT))) return rc;
  if ((rc = ensure_buf(c, sl.counter, 256))) return rc;
  const uint32_t nchunk_max = max_slices / 64 + 1;
  if ((rc = ensure_buf(c, sl.chunk_rank, (size_t)nchunk_max * 4))) return rc;
  if (!sl.h_tb) KZG_HIP(c, hipHostMalloc(&sl.h_tb, MAX_NPART * 4 * 16 * 4 + 32));
  for (hipEvent_t* e : {&sl.ev_in, &sl.ev_p, &sl.ev_a, &sl.ev_b})
    if (!*e) KZG_HIP(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
  if (!w->stream_a) {
    KZG_HIP(c, hipStreamCreateWithFlags(&w->stream_a, hipStreamNonBlocking));
    KZG_HIP(c, hipStreamCreateWithFlags(&w->stream_p, hipStreamNonBlocking));
    KZG_HIP(c, hipStreamCreateWithFlags(&w->stream_b, hipStreamNonBlocking));
    // accumulate: 2 waves on every SIMD (4 workgroups of 2 waves per CU); KZG_ACC_WGS_PER_CU overrides
    int cus = 0;
    KZG_HIP(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    const char* ew = getenv("KZG_ACC_WGS_PER_CU");
    const int per_cu = ew && atoi(ew) > 0 ? atoi(ew) : 4;
    w->acc_blocks = (uint32_t)std::max(cus, 1) * (uint32_t)per_cu;
  }
  sl.win_bits = WB;

  auto* vals = static_cast<uint32_t*>(sl.vals.p);
  auto* bstart = static_cast<uint32_t*>(sl.bstart.p);
  auto* order = static_cast<uint32_t*>(sl.order.p);
  auto* slice_off = static_cast<uint32_t*>(sl.slice_off.p);
  hipStream_t sp = w->stream_p, sa = w->stream_a, sb = w->stream_b;

  // ---- stage P: prep
  // The scalars are first copied into the slot (n x 32 bytes on the context's stream, ~25 us at 2^20): the caller's
  // buffer is free as soon as that copy has run, so what it enqueues next on the context's stream -- the next
  // transform, the next opening's polynomial stage -- does not queue behind this polynomial's prep (~0.5 ms).
  static const bool copy_scalars = [] { const char* e = getenv("KZG_COMMIT_COPY_SCALARS"); return !(e && atoi(e) == 0); }();
  if (copy_scalars) {
    if ((rc = ensure_buf(c, sl.scal, (size_t)n * 32))) return rc;
    KZG_HIP(c, hipMemcpyAsync(sl.scal.p, d_scalars, (size_t)n * 32, hipMemcpyDeviceToDevice, c->stream));
    d_scalars = static_cast<const uint32_t*>(sl.scal.p);
  }
  KZG_HIP(c, hipEventRecord(sl.ev_in, c->stream));
  KZG_HIP(c, hipStreamWaitEvent(sp, sl.ev_in, 0));
#ifdef KZG_TIMING_SKIP_PREP   // timing experiment only (results are wrong): reuse the slot's previous prep output
  static int prep_runs = 0;
  if (prep_runs++ >= 2 * NSLOT) { KZG_HIP(c, hipMemsetAsync(sl.counter.p, 0, 4, sp)); } else
#endif
  if ((rc = msm_prep_enqueue(c, sp, WB, d_scalars, n, (uint32_t)s->n, W::SEG, sl.prep_ws.p, vals, bstart, order,
                             slice_off, static_cast<uint32_t*>(sl.counter.p),
                             static_cast<uint32_t*>(sl.chunk_rank.p), nchunk_max)))
    return rc;
  KZG_HIP(c, hipEventRecord(sl.ev_p, sp));
  if (!copy_scalars) KZG_HIP(c, hipStreamWaitEvent(c->stream, sl.ev_p, 0));   // the scalars are free again from here on

  // ---- stage A: accumulate
  KZG_HIP(c, hipStreamWaitEvent(sa, sl.ev_p, 0));
  {
    ProfScope ps(c, "msm_accumulate", sa);
    hipLaunchKernelGGL((msm_accumulate_kernel<C, WB>), dim3(std::min(w->acc_blocks, (max_slices + 127) / 128)),
                       dim3(128), 0, sa, s->recs, vals, bstart, order, slice_off,
                       static_cast<uint32_t*>(sl.partials.p), static_cast<uint32_t*>(sl.buckets.p),
                       static_cast<uint32_t*>(sl.counter.p), static_cast<const uint32_t*>(sl.chunk_rank.p),
                       c->prof_on ? c->clk_probe : nullptr);
  }
  KZG_HIP(c, hipGetLastError());
  KZG_HIP(c, hipEventRecord(sl.ev_a, sa));

  // ---- stage B: reduce
  KZG_HIP(c, hipStreamWaitEvent(sb, sl.ev_a, 0));
#ifndef KZG_TIMING_SKIP_REDUCE   // timing experiment only (results are wrong)
  {
    ProfScope ps(c, "msm_finalize", sb);
    hipLaunchKernelGGL((msm_finalize_heavy_kernel<C, WB>), dim3(std::min<uint32_t>(HEAVY_RANKS, NB)), dim3(256), 0, sb,
                       static_cast<uint32_t*>(sl.partials.p), order, slice_off, static_cast<uint32_t*>(sl.buckets.p));
    hipLaunchKernelGGL((msm_finalize_kernel<C, WB>), dim3((uint32_t)(((uint64_t)NB * W::FIN + 127) / 128)), dim3(128),
                       0, sb, static_cast<uint32_t*>(sl.partials.p), order, slice_off,
                       static_cast<uint32_t*>(sl.buckets.p));
  }
  {
    ProfScope ps(c, "msm_reduce", sb);
    hipLaunchKernelGGL((msm_rc1_kernel<C, WB>), dim3((NB / RC_CH + 127) / 128, 2), dim3(128), 0, sb,
                       static_cast<uint32_t*>(sl.buckets.p), static_cast<uint32_t*>(sl.colpart.p),
                       static_cast<uint32_t*>(sl.rowpart.p));
    hipLaunchKernelGGL((msm_rc2_kernel<C, WB>), dim3((((1u << W::LO) + (1u << W::HI)) * RC_L2 + 127) / 128),
                       dim3(128), 0, sb, static_cast<uint32_t*>(sl.colpart.p), static_cast<uint32_t*>(sl.rowpart.p),
                       static_cast<uint32_t*>(sl.colsum.p), static_cast<uint32_t*>(sl.rowsum.p));
    hipLaunchKernelGGL((msm_planes_kernel<C, WB>), dim3(W::NPART), dim3(128), 0, sb,
                       static_cast<uint32_t*>(sl.rowsum.p), static_cast<uint32_t*>(sl.colsum.p),
                       static_cast<uint32_t*>(sl.buckets.p), static_cast<uint32_t*>(sl.tb.p));
  }
#endif
  KZG_HIP(c, hipGetLastError());
  KZG_HIP(c, hipMemcpyAsync(sl.h_tb, sl.tb.p, W::NPART * PT, hipMemcpyDeviceToHost, sb));
  KZG_HIP(c, hipEventRecord(sl.ev_b, sb));
  return KZG_OK;
}

// host: 2^LO * sum_b 2^b TR_b + sum_b 2^b TC_b + 2^(WB-1) * top; to affine, canonical words
template <class C, int WB>
static void msm_finish_host(const void* h_tb, uint64_t* out_xy, uint8_t* out_inf) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  using W = Win<WB>;
  constexpr int N = F::N;
  const uint32_t* p = static_cast<const uint32_t*>(h_tb);
  auto pt = [&](int idx) {
    XYZZ<C> t;
    memcpy(t.x.l, p + (size_t)idx * 4 * N, N * 4);
    memcpy(t.y.l, p + (size_t)idx * 4 * N + N, N * 4);
    memcpy(t.zz.l, p + (size_t)idx * 4 * N + 2 * N, N * 4);
    memcpy(t.zzz.l, p + (size_t)idx * 4 * N + 3 * N, N * 4);
    return t;
  };
  // acc = top * 2^(HI-1) ... folded into one Horner chain over descending powers of two:
  //   total = sum_{b<HI} 2^(LO+b) TR_b + sum_{b<LO} 2^b TC_b + 2^(WB-1) top,   WB-1 = HI+LO
  XYZZ<C> acc = pt(W::HI + W::LO);                       // coefficient of 2^(HI+LO)
  for (int e = W::HI + W::LO - 1; e >= 0; --e) {
    acc = Ec<C>::dbl(acc);
    if (e >= W::LO) acc = Ec<C>::add(acc, pt(e - W::LO));          // TR_{e-LO}
    if (e < W::LO) acc = Ec<C>::add(acc, pt(W::HI + e));           // TC_e
  }
  const Affine<C> a = Ec<C>::to_affine(acc);
  uint32_t* o = reinterpret_cast<uint32_t*>(out_xy);
  if (a.inf) {
    memset(o, 0, 2 * F::NW * 4);
    *out_inf = 1;
    return;
  }
  Fd::to_words(Fd::from_mont(a.x), o);
  Fd::to_words(Fd::from_mont(a.y), o + F::NW);
  *out_inf = 0;
}

template <class C>
static int msm_retire(Ctx* c, MsmSlot& sl) {
  if (!sl.pending) { sl.out_eval = nullptr; return KZG_OK; }
  KZG_HIP(c, hipEventSynchronize(sl.ev_b));
  if (sl.win_bits == 20) msm_finish_host<C, 20>(sl.h_tb, sl.out_xy, sl.out_inf);
  else msm_finish_host<C, 16>(sl.h_tb, sl.out_xy, sl.out_inf);
  if (sl.out_eval) {      // copied on the context's stream before this slot's prep was allowed to start
    memcpy(sl.out_eval, static_cast<const char*>(sl.h_tb) + MAX_NPART * 4 * 16 * 4, 32);
    sl.out_eval = nullptr;
  }
  sl.pending = false;
  return KZG_OK;
}

template <class C>
static int commit_flush_t(Ctx* c) {
  MsmWork* w = get_work(c);
  int rc = KZG_OK;
  for (int k = 0; k < NSLOT; ++k) {               // drain in issue order
    MsmSlot& sl = w->slot[(w->next + k) % NSLOT];
    int r2 = msm_retire<C>(c, sl);
    if (rc == KZG_OK) rc = r2;
  }
  if (rc == KZG_OK) KZG_HIP(c, hipStreamSynchronize(c->stream));
  return rc;
}

template <class C>
static int commit_t(Ctx* c, const Srs* s, const uint32_t* d_scalars, const size_t* lens, size_t n_polys,
                    size_t stride, uint64_t* out_xy, uint8_t* out_inf, bool drain, const uint32_t* d_eval,
                    uint64_t* out_eval) {
  using F = typename C::Fp;
  MsmWork* w = get_work(c);
  for (size_t p = 0; p < n_polys; ++p) {
    if (lens[p] > s->n) return set_err(c, KZG_ERR_DEGREE, "polynomial longer than the commitment key");
    if (lens[p] > stride) return set_err(c, KZG_ERR_ARG, "kzg_commit: lens[p] > stride");
  }
  int rc = KZG_OK;
  for (size_t p = 0; p < n_polys && rc == KZG_OK; ++p) {
    uint64_t* o = out_xy + p * 2 * (F::NW / 2);
    if (lens[p] == 0) {   // zero polynomial: Z1 (kzg.py:109)
      memset(o, 0, 2 * F::NW * 4);
      out_inf[p] = 1;
      if (d_eval && out_eval) {     // nothing to pipeline behind: fetch P(z) now
        KZG_HIP(c, hipMemcpyAsync(out_eval, d_eval, 32, hipMemcpyDeviceToHost, c->stream));
        KZG_HIP(c, hipStreamSynchronize(c->stream));
      }
      continue;
    }
    const int si = w->next;
    MsmSlot& sl = w->slot[si];
    w->next = (w->next + 1) % NSLOT;
    if ((rc = msm_retire<C>(c, sl))) break;       // recycle: its stage B has long finished
    uint64_t* slot_eval = nullptr;
    if (d_eval && out_eval) {
      // P(z) rides with the slot: the copy is ordered on the context's stream BEFORE ev_in, which every stage of
      // this polynomial waits for, so it has landed when the slot's last event (ev_b) has
      if (!sl.h_tb) KZG_HIP(c, hipHostMalloc(&sl.h_tb, MAX_NPART * 4 * 16 * 4 + 32));
      KZG_HIP(c, hipMemcpyAsync(static_cast<char*>(sl.h_tb) + MAX_NPART * 4 * 16 * 4, d_eval, 32,
                                hipMemcpyDeviceToHost, c->stream));
      slot_eval = out_eval;
    }
    const uint32_t* sc = d_scalars + p * stride * 8;
    rc = s->win_bits == 20 ? msm_enqueue<C, 20>(c, s, sc, (uint32_t)lens[p], w, si)
                           : msm_enqueue<C, 16>(c, s, sc, (uint32_t)lens[p], w, si);
    if (rc) break;                                 // nothing of this polynomial stays behind in the slot
    sl.pending = true;
    sl.out_eval = slot_eval;                       // host pointers are adopted only once the work is queued
    sl.out_xy = o;
    sl.out_inf = out_inf + p;
  }
  if (drain || rc != KZG_OK) {
    int r2 = commit_flush_t<C>(c);
    if (rc == KZG_OK) rc = r2;
  }
  return rc;
}

int commit_flush(Ctx* c) { return c->curve == 0 ? commit_flush_t<Bn254>(c) : commit_flush_t<Bls12_381>(c); }

int commit_device(Ctx* c, const Srs* s, const uint32_t* d_scalars, const size_t* lens, size_t n_polys,
                  size_t stride, uint64_t* out_xy, uint8_t* out_inf, bool drain, const uint32_t* d_eval,
                  uint64_t* out_eval) {
  if (s->curve != c->curve) return set_err(c, KZG_ERR_ARG, "SRS belongs to another curve");
  if (d_eval && n_polys != 1) return set_err(c, KZG_ERR_ARG, "an evaluation rides with exactly one polynomial");
  return c->curve == 0
             ? commit_t<Bn254>(c, s, d_scalars, lens, n_polys, stride, out_xy, out_inf, drain, d_eval, out_eval)
             : commit_t<Bls12_381>(c, s, d_scalars, lens, n_polys, stride, out_xy, out_inf, drain, d_eval, out_eval);
}

}  // namespace kzg
