// poly.hip -- the polynomial half of KZG.open (kzg.py:122-159) on gfx950:
//
//   combined(X) = sum_i xi^(i+1) * p_i(X)                     kzg.py:147-150
//   witness(X)  = (combined(X) - combined(z)) // (X - z)      kzg.py:153-154
//
// Division by (X - z) is synthetic division: with S_j = sum_{i>=j} c_i z^(i-j)
// (suffix Horner values, S_j = c_j + z*S_{j+1}) the quotient is q_{j-1} = S_j for
// j >= 1 and combined(z) = S_0.  The first-order recurrence is evaluated in
// parallel by chunking: chunks of SC coefficients are collapsed bottom-up into
// one value each (a polynomial in z^SC of 1/SC the length; repeated until <= SC
// values remain), the short top level is solved directly, and the suffix values
// are pushed back down, each thread re-walking its chunk from the carry above.
// Cost: ~2 Montgomery multiplications per coefficient, all data HBM-streamed
// twice.  The MSM of the quotient (msm.hip) dominates open() by far.
// These kernels are bound by dependent Horner chains and by memory latency, not by instruction issue: the chain pin of
// field.h (an asm volatile per multiply-add) would only keep the scheduler from hoisting the next loads.
#define KZG_NO_CHAIN_PIN 1
#include "internal.h"
#include "msm.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace kzg {

namespace {

constexpr uint32_t LC = 32;       // elements per thread of the vector primitives (batch inversion, powers, prefix product)
// Chunk of the suffix-Horner scans (open, poly_eval): their cost is the DEPENDENT chain of SC multiply-adds per
// thread, not bandwidth -- measured at 2^20, k = 6: 151 us with chunks of 32, 129 with 16, 114 with 8 (more, shorter
// levels; the batch inversion above wants the opposite, hence two constants).
#ifndef KZG_POLY_SC_LOG
#define KZG_POLY_SC_LOG 3
#endif
constexpr uint32_t SC_LOG = KZG_POLY_SC_LOG;
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

// The same with the bottom level of the evaluation fused in (round 3): beside comb[t] the workgroup also leaves
// h1[c] = sum_{j < SC} comb[c*SC + j] * z^j for its 256 / SC chunks, so the scan does not read the combination a second
// time for that.  Each thread multiplies its coefficient by z^(t mod SC) (table in LDS), the SC products of a chunk are
// added lazily by its first lane (SC * 2p stays far below 2^261 and SC limbs of 29 bits below 2^32) and reduced once.
struct ZPowArgs {
  uint32_t l[SC * FRN];      // z^0 .. z^(SC-1), Montgomery form
};
template <class F>
__global__ __launch_bounds__(256) void lincomb_eval_kernel(LincombArgs a, ZPowArgs zp, uint32_t* out, uint32_t n,
                                                           uint32_t* h1) {
  using Fd = Field<F>;
  static_assert(F::N == FRN && 256 % SC == 0 && SC <= 8, "chunk layout of the fused evaluation");
  __shared__ uint32_t zs[SC * FRN];
  __shared__ uint32_t es[256 * FRN];                         // stride 9 words: conflict-free across consecutive lanes
  const uint32_t tid = threadIdx.x;
  const uint32_t t = blockIdx.x * 256 + tid;
  if (tid < SC * FRN) zs[tid] = zp.l[tid];
  Fe<F> acc = Fd::zero();
  if (t < n) {
    for (uint32_t i0 = 0; i0 < a.k; i0 += DOT_G) {
      const uint32_t cnt = min(DOT_G, a.k - i0);
      Fe<F> c[DOT_G], x[DOT_G];
#pragma unroll
      for (uint32_t g = 0; g < DOT_G; ++g) {
        const uint32_t i = i0 + g;
        const bool on = g < cnt && t < a.lens[i];
        c[g] = on ? load_words<F>(a.polys + (a.stride * i + t) * 8) : Fd::zero();
        x[g] = load_limbs<F>(a.xipow + (g < cnt ? i : i0) * F::N);
      }
      const Fe<F> part = dot_upto<F>(cnt, c, x);
      acc = i0 ? Fd::add(acc, part) : part;
    }
    store_words<F>(out + (size_t)t * 8, acc);
  }
  __syncthreads();
  const Fe<F> e = Fd::mul(acc, load_limbs<F>(zs + (tid & (SC - 1)) * F::N));      // 0 beyond the polynomial's end
  store_limbs<F>(es + tid * F::N, e);
  __syncthreads();
  if ((tid & (SC - 1)) == 0 && t < n) {
    Fe<F> sum = e;
#pragma unroll
    for (uint32_t j = 1; j < SC; ++j) sum = Fd::add_lazy(sum, load_limbs<F>(es + (tid + j) * F::N));
    store_limbs<F>(h1 + (size_t)(t / SC) * F::N, Fd::reduce_wide(Fd::carry(sum)));
  }
}

// bottom-up: h[t] = sum_{j in chunk t} c_j * z^(j - t*SC)
template <class F, bool WORDS_IN>
__global__ void chunk_eval_kernel(const uint32_t* in, uint32_t m, FrArg zpow, uint32_t* h) {
  using Fd = Field<F>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j0 = t * SC;
  if (j0 >= m) return;
  const uint32_t j1 = min(j0 + SC, m);
  const Fe<F> z = load_limbs<F>(zpow.l);
  Fe<F> acc = Fd::zero();
  for (uint32_t j = j1; j-- > j0;) {
    const Fe<F> c = WORDS_IN ? load_words<F>(in + (size_t)j * 8) : load_limbs<F>(in + (size_t)j * F::N);
    acc = Fd::add(c, Fd::mul(acc, z));
  }
  store_limbs<F>(h + (size_t)t * F::N, acc);
}

// top level (m <= SC): S[j] = c_j + z*S[j+1], S[m] = 0; one thread
template <class F>
__global__ void top_suffix_kernel(const uint32_t* in, uint32_t m, FrArg zpow, uint32_t* S) {
  using Fd = Field<F>;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const Fe<F> z = load_limbs<F>(zpow.l);
  Fe<F> acc = Fd::zero();
  store_limbs<F>(S + (size_t)m * F::N, acc);
  for (uint32_t j = m; j-- > 0;) {
    acc = Fd::add(load_limbs<F>(in + (size_t)j * F::N), Fd::mul(acc, z));
    store_limbs<F>(S + (size_t)j * F::N, acc);
  }
}

// top-down: S[j] for j in chunk t, starting from the carry S_up[t+1].
// FINAL: input is the coefficient array (words); writes quotient q[j-1] = S_j and eval = S_0.
template <class F, bool FINAL>
__global__ void chunk_fill_kernel(const uint32_t* in, uint32_t m, FrArg zpow, const uint32_t* S_up,
                                  uint32_t* S_out, uint32_t* quot, uint32_t* eval_out) {
  using Fd = Field<F>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j0 = t * SC;
  if (j0 >= m) return;
  const uint32_t j1 = min(j0 + SC, m);
  const Fe<F> z = load_limbs<F>(zpow.l);
  Fe<F> acc = load_limbs<F>(S_up + (size_t)(t + 1) * F::N);
  if (!FINAL && t == 0 && j1 == m) { /* single chunk: nothing above */ }
  for (uint32_t j = j1; j-- > j0;) {
    const Fe<F> c = FINAL ? load_words<F>(in + (size_t)j * 8) : load_limbs<F>(in + (size_t)j * F::N);
    acc = Fd::add(c, Fd::mul(acc, z));
    if (FINAL) {
      if (j >= 1) store_words<F>(quot + (size_t)(j - 1) * 8, acc);
      else store_words<F>(eval_out, acc);
    } else {
      store_limbs<F>(S_out + (size_t)j * F::N, acc);
    }
  }
  if (!FINAL && j1 == m) store_limbs<F>(S_out + (size_t)m * F::N, Fd::zero());
}

__global__ void any_nonzero_kernel(const uint32_t* words, size_t from_elem, size_t to_elem, uint32_t* flag) {
  const size_t i = from_elem + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= to_elem) return;
  uint32_t acc = 0;
  for (int k = 0; k < 8; ++k) acc |= words[i * 8 + k];
  if (acc) atomicOr(flag, 1u);
}

// The last fill (coefficients in, quotient out) through LDS (round 3): a workgroup loads its FILL_EPB consecutive
// coefficients with fully coalesced 16-byte accesses, every thread walks its SC-element chunk in LDS (one pad quad per
// chunk spreads the 256-byte lane stride over the banks), and the suffix values leave coalesced again -- S[j] goes to
// S_base + j (S_0 = the evaluation, S_j = quotient coefficient j-1: one contiguous vector, see the buffer layout).
// The round-2 kernel read and wrote 32-byte elements at 256-byte lane strides: 1.7x / 1.55x the bytes (r03_open_pmc.csv).
constexpr uint32_t FILL_TB = 128, FILL_EPB = FILL_TB * SC;
template <class F>
__global__ __launch_bounds__(FILL_TB) void chunk_fill_final_kernel(const uint32_t* in, uint32_t m, FrArg zpow,
                                                                    const uint32_t* S_up, uint32_t* S_base) {
  using Fd = Field<F>;
  __shared__ uint4 st[FILL_EPB * 2 + FILL_TB];
  const uint32_t tid = threadIdx.x, e0 = blockIdx.x * FILL_EPB;
  const uint4* gin = reinterpret_cast<const uint4*>(in);
  for (uint32_t i = tid; i < FILL_EPB * 2; i += FILL_TB) {
    const uint32_t e = i >> 1;
    if (e0 + e < m) st[i + (e >> SC_LOG)] = gin[(size_t)(e0 + e) * 2 + (i & 1)];
  }
  __syncthreads();
  const uint32_t t = blockIdx.x * FILL_TB + tid;
  const uint32_t j0 = t * SC;
  if (j0 < m) {
    const uint32_t j1 = min(j0 + SC, m);
    const Fe<F> z = load_limbs<F>(zpow.l);
    Fe<F> acc = load_limbs<F>(S_up + (size_t)(t + 1) * F::N);
    for (uint32_t j = j1; j-- > j0;) {
      const uint32_t q = (j - e0) * 2 + tid;                 // (j - e0) >> SC_LOG == tid
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
  for (uint32_t i = tid; i < FILL_EPB * 2; i += FILL_TB) {
    const uint32_t e = i >> 1;
    if (e0 + e < m) gout[(size_t)(e0 + e) * 2 + (i & 1)] = st[i + (e >> SC_LOG)];
  }
}

// Buffer layout of poly_tmp[0] (canonical words, 8 per element), cap = n + 1:
//   comb[0 .. cap)   combined polynomial (+ one optional appended top coefficient)
//   eval             S_0
//   quot[0 .. cap)   S_1, S_2, ...        => [eval, quot...] is the contiguous vector S_0, S_1, ...
// total entries of the levels above the coefficients (h_1 .. h_(nl-1)) for n coefficients
static size_t scan_level_entries(size_t n) {
  size_t total = 0;
  for (size_t m = n; m > SC;) { m = (m + SC - 1) / SC; total += m; }
  return total;
}

// z_words given: the bottom level of the evaluation at z is fused into the combination (h_1 lands at the start of
// poly_tmp[2], where open_scan_t(.., h1_ready = true) expects it).
template <class F>
int open_combine_t(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                   const uint32_t* xi_words, size_t* n_out, const uint32_t* z_words = nullptr) {
  using Fd = Field<F>;
  if (k > MAXK) return set_err(c, KZG_ERR_ARG, "kzg_open: more than 64 polynomials");
  size_t n = 0;
  for (size_t i = 0; i < k; ++i) {
    if (lens[i] > stride) return set_err(c, KZG_ERR_ARG, "kzg_open: lens[i] > stride");
    n = std::max(n, lens[i]);
  }
  *n_out = n;
  if (n == 0) return KZG_OK;
  if (n >= (1ull << 31) - 1) return set_err(c, KZG_ERR_ARG, "kzg_open: polynomial too long");
  int rc;
  if ((rc = ensure_buf(c, c->poly_tmp[0], (2 * (n + 1) + 1) * 32))) return rc;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  ProfScope ps(c, "open_poly");
  LincombArgs la{};
  la.polys = d_polys; la.stride = stride; la.k = (uint32_t)k;
  const Fe<F> xi = Fd::to_mont(Fd::from_words(xi_words));
  Fe<F> xp = Fd::one();
  for (size_t i = 0; i < k; ++i) {
    xp = Fd::mul(xp, xi);                                   // xi^(i+1): kzg.py:148-150
    memcpy(&la.xipow[i * F::N], xp.l, F::N * 4);
    la.lens[i] = (uint32_t)lens[i];
  }
  if (z_words && n > SC) {
    if ((rc = ensure_buf(c, c->poly_tmp[2], (scan_level_entries(n) + 1) * F::N * 4))) return rc;
    ZPowArgs zp{};
    const Fe<F> z = Fd::to_mont(Fd::from_words(z_words));
    Fe<F> zj = Fd::one();
    for (uint32_t j = 0; j < SC; ++j) {
      const Fe<F> zr = Fd::reduce(zj);
      memcpy(&zp.l[j * F::N], zr.l, F::N * 4);
      zj = Fd::mul(zj, z);
    }
    hipLaunchKernelGGL(lincomb_eval_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, la, zp,
                       d_comb, (uint32_t)n, static_cast<uint32_t*>(c->poly_tmp[2].p));
  } else {
    hipLaunchKernelGGL(lincomb_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, la, d_comb,
                       (uint32_t)n);
  }
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

// Suffix-Horner scan of comb[0 .. n) (n >= 1) at z: eval <- S_0, quot[j-1] <- S_j.  `cap` fixes the layout.
// eval_out: host memory that receives S_0 (the call then synchronises the stream), or nullptr: nothing is copied and
// nothing waits -- the pipelined open fetches the 32 bytes at comb + cap*8 itself (msm.hip, commit_device).
template <class F>
int open_scan_t(Ctx* c, size_t n, size_t cap, const uint32_t* z_words, uint64_t* eval_out, bool sync = true,
                bool h1_ready = false) {
  using Fd = Field<F>;
  std::vector<uint32_t> m{(uint32_t)n};
  while (m.back() > SC) m.push_back((m.back() + SC - 1) / SC);
  const size_t nl = m.size();
  std::vector<FrArg> zp(nl);                              // z^(SC^l), Montgomery form, passed by value
  {
    Fe<F> t = Fd::to_mont(Fd::from_words(z_words));
    for (size_t l = 0; l < nl; ++l) {
      memcpy(zp[l].l, t.l, F::N * 4);
      for (uint32_t q = 0; q < SC_LOG; ++q) t = Fd::mul(t, t);        // ^(2^SC_LOG) = ^SC
    }
  }
  size_t hl_total = 0, sl_total = 0;
  for (size_t l = 1; l < nl; ++l) { hl_total += m[l]; sl_total += m[l] + 1; }
  int rc;
  if ((rc = ensure_buf(c, c->poly_tmp[2], (hl_total + 1) * F::N * 4))) return rc;
  if ((rc = ensure_buf(c, c->poly_tmp[3], (sl_total + 2) * F::N * 4))) return rc;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  uint32_t* d_eval = d_comb + cap * 8;
  uint32_t* d_quot = d_eval + 8;
  uint32_t* d_h = static_cast<uint32_t*>(c->poly_tmp[2].p);
  uint32_t* d_S = static_cast<uint32_t*>(c->poly_tmp[3].p);

  ProfScope ps(c, "open_poly");
  auto zpow = [&](size_t l) { return zp[l]; };
  std::vector<uint32_t*> hptr(nl, nullptr), sptr(nl, nullptr);
  {
    uint32_t* hp = d_h; uint32_t* sp = d_S;
    for (size_t l = 1; l < nl; ++l) { hptr[l] = hp; hp += (size_t)m[l] * F::N; sptr[l] = sp; sp += (size_t)(m[l] + 1) * F::N; }
  }
  (void)d_quot;
  if (nl == 1) {
    // n <= SC: a single chunk; its carry is zero.  Use a one-entry zero suffix array.
    KZG_HIP(c, hipMemsetAsync(d_S, 0, 2 * F::N * 4, c->stream));
    hipLaunchKernelGGL(chunk_fill_final_kernel<F>, dim3(1), dim3(FILL_TB), 0, c->stream, d_comb, m[0], zpow(0), d_S,
                       d_eval);
  } else {
    if (!h1_ready)     // (the fused combination of open_quotient_t has left h_1 already)
      hipLaunchKernelGGL((chunk_eval_kernel<F, true>), dim3((m[1] + 127) / 128), dim3(128), 0, c->stream, d_comb, m[0],
                         zpow(0), hptr[1]);
    for (size_t l = 1; l + 1 < nl; ++l)
      hipLaunchKernelGGL((chunk_eval_kernel<F, false>), dim3((m[l + 1] + 127) / 128), dim3(128), 0, c->stream, hptr[l],
                         m[l], zpow(l), hptr[l + 1]);
    hipLaunchKernelGGL(top_suffix_kernel<F>, dim3(1), dim3(64), 0, c->stream, hptr[nl - 1], m[nl - 1], zpow(nl - 1),
                       sptr[nl - 1]);
    for (size_t l = nl - 2; l >= 1; --l)
      hipLaunchKernelGGL((chunk_fill_kernel<F, false>), dim3((m[l + 1] + 127) / 128), dim3(128), 0, c->stream, hptr[l],
                         m[l], zpow(l), sptr[l + 1], sptr[l], (uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(chunk_fill_final_kernel<F>, dim3((m[0] + FILL_EPB - 1) / FILL_EPB), dim3(FILL_TB), 0, c->stream,
                       d_comb, m[0], zpow(0), sptr[1], d_eval);
  }
  KZG_HIP(c, hipGetLastError());
  if (eval_out) {
    KZG_HIP(c, hipMemcpyAsync(eval_out, d_eval, 32, hipMemcpyDeviceToHost, c->stream));
    KZG_HIP(c, hipStreamSynchronize(c->stream));
  }
  (void)sync;
  return KZG_OK;
}

template <class F>
int open_quotient_t(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                    const uint32_t* z_words, const uint32_t* xi_words, uint32_t** d_quot_out, size_t* quot_len,
                    uint64_t* eval_out, bool sync) {
  if (eval_out) memset(eval_out, 0, 32);
  *quot_len = 0;
  *d_quot_out = nullptr;
  size_t n = 0;
  int rc = open_combine_t<F>(c, d_polys, lens, k, stride, xi_words, &n, z_words);
  if (rc) return rc;
  if (n == 0) return KZG_OK;     // all polynomials zero: witness 0, evaluation 0
  if ((rc = open_scan_t<F>(c, n, n + 1, z_words, sync ? eval_out : nullptr, sync, /*h1_ready=*/n > SC))) return rc;
  *d_quot_out = static_cast<uint32_t*>(c->poly_tmp[0].p) + (n + 1) * 8 + 8;
  *quot_len = n - 1;
  return KZG_OK;
}

// Sharded open, step 1: combine this rank's coefficient slices and evaluate the slice polynomial
// (local indexing) at z.
template <class F>
int open_shard_begin_t(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                       const uint32_t* z_words, const uint32_t* xi_words, uint64_t* chunk_eval_out) {
  memset(chunk_eval_out, 0, 32);
  size_t n = 0;
  int rc = open_combine_t<F>(c, d_polys, lens, k, stride, xi_words, &n);
  if (rc) return rc;
  c->open_shard_n = n;
  if (n == 0) return KZG_OK;
  return open_scan_t<F>(c, n, n + 1, z_words, chunk_eval_out);
}

// step 2: with the carry S_hi of the ranks above appended as an extra top coefficient the scan
// yields S_lo .. S_(hi-1) exactly (linearity).  Returns the vector to commit.
template <class F>
int open_shard_finish_t(Ctx* c, const uint32_t* z_words, const uint32_t* carry_words, int first_rank,
                        uint32_t** d_vec_out, size_t* vec_len, uint64_t* eval_out) {
  memset(eval_out, 0, 32);
  *d_vec_out = nullptr;
  *vec_len = 0;
  const size_t n = c->open_shard_n;
  if (n == 0) return KZG_OK;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  KZG_HIP(c, hipMemcpyAsync(d_comb + n * 8, carry_words, 32, hipMemcpyHostToDevice, c->stream));
  int rc = open_scan_t<F>(c, n + 1, n + 1, z_words, eval_out);
  if (rc) return rc;
  uint32_t* d_eval = d_comb + (n + 1) * 8;
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

// p(z) for a coefficient vector: the bottom-up half of the open() scan
template <class F>
int poly_eval_t(Ctx* c, size_t n, const uint32_t* a, const uint32_t* z_words, uint64_t* out) {
  using Fd = Field<F>;
  memset(out, 0, 32);
  if (n == 0) return KZG_OK;
  if (n >= (1ull << 31)) return set_err(c, KZG_ERR_ARG, "vector too long");
  std::vector<uint32_t> m{(uint32_t)n};
  while (m.back() > 1) m.push_back((m.back() + SC - 1) / SC);
  const size_t nl = m.size();
  std::vector<FrArg> zp(nl);                              // z^(SC^l), passed to the kernels by value
  {
    Fe<F> t = Fd::to_mont(Fd::from_words(z_words));
    for (size_t l = 0; l < nl; ++l) {
      memcpy(zp[l].l, t.l, F::N * 4);
      for (uint32_t q = 0; q < SC_LOG; ++q) t = Fd::mul(t, t);
    }
  }
  size_t total = 0;
  for (size_t l = 1; l < nl; ++l) total += m[l];
  int rc;
  if ((rc = ensure_buf(c, c->poly_tmp[2], (total + 1) * F::N * 4))) return rc;
  uint32_t* d_h = static_cast<uint32_t*>(c->poly_tmp[2].p);
  if (nl == 1) {     // n == 1: the value is the coefficient itself
    KZG_HIP(c, hipMemcpyAsync(out, a, 32, hipMemcpyDeviceToHost, c->stream));
    KZG_HIP(c, hipStreamSynchronize(c->stream));
    return KZG_OK;
  }
  std::vector<uint32_t*> hp(nl, nullptr);
  { uint32_t* x = d_h; for (size_t l = 1; l < nl; ++l) { hp[l] = x; x += (size_t)m[l] * F::N; } }
  hipLaunchKernelGGL((chunk_eval_kernel<F, true>), dim3((m[1] + 127) / 128), dim3(128), 0, c->stream, a, m[0],
                     zp[0], hp[1]);
  for (size_t l = 1; l + 1 < nl; ++l)
    hipLaunchKernelGGL((chunk_eval_kernel<F, false>), dim3((m[l + 1] + 127) / 128), dim3(128), 0, c->stream, hp[l],
                       m[l], zp[l], hp[l + 1]);
  KZG_HIP(c, hipGetLastError());
  // the top level holds one weak-normal value (standard form): canonicalise on the host
  uint32_t top[F::N];
  KZG_HIP(c, hipMemcpyAsync(top, hp[nl - 1], F::N * 4, hipMemcpyDeviceToHost, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  Fe<F> v;
  memcpy(v.l, top, F::N * 4);
  Fd::to_words(Fd::reduce(v), reinterpret_cast<uint32_t*>(out));
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
