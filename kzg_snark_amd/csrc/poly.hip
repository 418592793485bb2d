// poly.hip -- the polynomial half of KZG.open (kzg.py:122-159) on gfx950:
//
//   combined(X) = sum_i xi^(i+1) * p_i(X)                     kzg.py:147-150
//   witness(X)  = (combined(X) - combined(z)) // (X - z)      kzg.py:153-154
//
// Division by (X - z) is synthetic division: with S_j = sum_{i>=j} c_i z^(i-j)
// (suffix Horner values, S_j = c_j + z*S_{j+1}) the quotient is q_{j-1} = S_j for
// j >= 1 and combined(z) = S_0.  The first-order recurrence is evaluated in
// parallel by chunking: chunks of LC coefficients are collapsed bottom-up into
// one value each (a polynomial in z^LC of 1/LC the length; repeated until <= LC
// values remain), the short top level is solved directly, and the suffix values
// are pushed back down, each thread re-walking its chunk from the carry above.
// Cost: ~2 Montgomery multiplications per coefficient, all data HBM-streamed
// twice.  The MSM of the quotient (msm.hip) dominates open() by far.
#include "internal.h"
#include "msm.h"
#include <algorithm>
#include <cstring>
#include <vector>

namespace kzg {

namespace {

constexpr uint32_t LC = 32;       // coefficients per chunk
constexpr uint32_t MAXK = 64;     // polynomials per open()

struct LincombArgs {
  const uint32_t* polys;     // k polynomials, `stride` elements apart, canonical words
  uint64_t stride;
  uint32_t k;
  uint32_t lens[MAXK];
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

// out[t] = sum_i xipow[i] * p_i[t]   (xipow in Montgomery form => result in standard form)
template <class F>
__global__ void lincomb_kernel(LincombArgs a, const uint32_t* xipow, uint32_t* out, uint32_t n) {
  using Fd = Field<F>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  Fe<F> acc = Fd::zero();
  for (uint32_t i = 0; i < a.k; ++i) {
    if (t < a.lens[i]) {
      const Fe<F> c = load_words<F>(a.polys + (a.stride * i + t) * 8);
      acc = Fd::add(acc, Fd::mul(c, load_limbs<F>(xipow + i * F::N)));
    }
  }
  store_words<F>(out + (size_t)t * 8, acc);
}

// bottom-up: h[t] = sum_{j in chunk t} c_j * z^(j - t*LC)
template <class F, bool WORDS_IN>
__global__ void chunk_eval_kernel(const uint32_t* in, uint32_t m, const uint32_t* zpow, uint32_t* h) {
  using Fd = Field<F>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j0 = t * LC;
  if (j0 >= m) return;
  const uint32_t j1 = min(j0 + LC, m);
  const Fe<F> z = load_limbs<F>(zpow);
  Fe<F> acc = Fd::zero();
  for (uint32_t j = j1; j-- > j0;) {
    const Fe<F> c = WORDS_IN ? load_words<F>(in + (size_t)j * 8) : load_limbs<F>(in + (size_t)j * F::N);
    acc = Fd::add(c, Fd::mul(acc, z));
  }
  store_limbs<F>(h + (size_t)t * F::N, acc);
}

// top level (m <= LC): S[j] = c_j + z*S[j+1], S[m] = 0; one thread
template <class F>
__global__ void top_suffix_kernel(const uint32_t* in, uint32_t m, const uint32_t* zpow, uint32_t* S) {
  using Fd = Field<F>;
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const Fe<F> z = load_limbs<F>(zpow);
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
__global__ void chunk_fill_kernel(const uint32_t* in, uint32_t m, const uint32_t* zpow, const uint32_t* S_up,
                                  uint32_t* S_out, uint32_t* quot, uint32_t* eval_out) {
  using Fd = Field<F>;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t j0 = t * LC;
  if (j0 >= m) return;
  const uint32_t j1 = min(j0 + LC, m);
  const Fe<F> z = load_limbs<F>(zpow);
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

// Buffer layout of poly_tmp[0] (canonical words, 8 per element), cap = n + 1:
//   comb[0 .. cap)   combined polynomial (+ one optional appended top coefficient)
//   eval             S_0
//   quot[0 .. cap)   S_1, S_2, ...        => [eval, quot...] is the contiguous vector S_0, S_1, ...
template <class F>
int open_combine_t(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                   const uint32_t* xi_words, size_t* n_out) {
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
  const Fe<F> xi = Fd::to_mont(Fd::from_words(xi_words));
  std::vector<uint32_t> hs(k * F::N);
  Fe<F> xp = Fd::one();
  for (size_t i = 0; i < k; ++i) {
    xp = Fd::mul(xp, xi);
    memcpy(&hs[i * F::N], xp.l, F::N * 4);
  }
  int rc;
  if ((rc = ensure_buf(c, c->poly_tmp[0], (2 * (n + 1) + 1) * 32))) return rc;
  if ((rc = ensure_buf(c, c->poly_tmp[1], (k + 16) * F::N * 4))) return rc;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  uint32_t* d_sc = static_cast<uint32_t*>(c->poly_tmp[1].p);
  if (k) {
    KZG_HIP(c, hipMemcpyAsync(d_sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, c->stream));
    KZG_HIP(c, hipStreamSynchronize(c->stream));     // hs is a local vector
  }
  ProfScope ps(c, "open_poly");
  LincombArgs la{};
  la.polys = d_polys; la.stride = stride; la.k = (uint32_t)k;
  for (size_t i = 0; i < k; ++i) la.lens[i] = (uint32_t)lens[i];
  hipLaunchKernelGGL(lincomb_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, la, d_sc, d_comb,
                     (uint32_t)n);
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

// Suffix-Horner scan of comb[0 .. n) (n >= 1) at z: eval <- S_0, quot[j-1] <- S_j.  `cap` fixes the layout.
template <class F>
int open_scan_t(Ctx* c, size_t n, size_t cap, const uint32_t* z_words, uint64_t* eval_out) {
  using Fd = Field<F>;
  std::vector<uint32_t> m{(uint32_t)n};
  while (m.back() > LC) m.push_back((m.back() + LC - 1) / LC);
  const size_t nl = m.size();
  const Fe<F> z = Fd::to_mont(Fd::from_words(z_words));
  std::vector<uint32_t> hs(nl * F::N);
  Fe<F> zp = z;
  for (size_t l = 0; l < nl; ++l) {
    memcpy(&hs[l * F::N], zp.l, F::N * 4);
    for (int q = 0; q < 5; ++q) zp = Fd::mul(zp, zp);     // ^32 = ^LC
  }
  static_assert(LC == 32, "zp update assumes LC = 2^5");
  size_t hl_total = 0, sl_total = 0;
  for (size_t l = 1; l < nl; ++l) { hl_total += m[l]; sl_total += m[l] + 1; }
  int rc;
  if ((rc = ensure_buf(c, c->poly_tmp[2], (hl_total + 1) * F::N * 4))) return rc;
  if ((rc = ensure_buf(c, c->poly_tmp[3], (sl_total + 2 + nl) * F::N * 4))) return rc;
  uint32_t* d_comb = static_cast<uint32_t*>(c->poly_tmp[0].p);
  uint32_t* d_eval = d_comb + cap * 8;
  uint32_t* d_quot = d_eval + 8;
  uint32_t* d_h = static_cast<uint32_t*>(c->poly_tmp[2].p);
  uint32_t* d_S = static_cast<uint32_t*>(c->poly_tmp[3].p);
  uint32_t* d_zp = d_S + (sl_total + 2) * F::N;           // z^(LC^l) limbs behind the suffix arrays
  KZG_HIP(c, hipMemcpyAsync(d_zp, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));

  ProfScope ps(c, "open_poly");
  auto zpow = [&](size_t l) { return d_zp + l * F::N; };
  std::vector<uint32_t*> hptr(nl, nullptr), sptr(nl, nullptr);
  {
    uint32_t* hp = d_h; uint32_t* sp = d_S;
    for (size_t l = 1; l < nl; ++l) { hptr[l] = hp; hp += (size_t)m[l] * F::N; sptr[l] = sp; sp += (size_t)(m[l] + 1) * F::N; }
  }
  if (nl == 1) {
    // n <= LC: a single chunk; its carry is zero.  Use a one-entry zero suffix array.
    KZG_HIP(c, hipMemsetAsync(d_S, 0, 2 * F::N * 4, c->stream));
    hipLaunchKernelGGL((chunk_fill_kernel<F, true>), dim3(1), dim3(64), 0, c->stream, d_comb, m[0], zpow(0), d_S,
                       (uint32_t*)nullptr, d_quot, d_eval);
  } else {
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
    hipLaunchKernelGGL((chunk_fill_kernel<F, true>), dim3((m[1] + 127) / 128), dim3(128), 0, c->stream, d_comb, m[0],
                       zpow(0), sptr[1], (uint32_t*)nullptr, d_quot, d_eval);
  }
  KZG_HIP(c, hipGetLastError());
  KZG_HIP(c, hipMemcpyAsync(eval_out, d_eval, 32, hipMemcpyDeviceToHost, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  return KZG_OK;
}

template <class F>
int open_quotient_t(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                    const uint32_t* z_words, const uint32_t* xi_words, uint32_t** d_quot_out, size_t* quot_len,
                    uint64_t* eval_out) {
  memset(eval_out, 0, 32);
  *quot_len = 0;
  *d_quot_out = nullptr;
  size_t n = 0;
  int rc = open_combine_t<F>(c, d_polys, lens, k, stride, xi_words, &n);
  if (rc) return rc;
  if (n == 0) return KZG_OK;     // all polynomials zero: witness 0, evaluation 0
  if ((rc = open_scan_t<F>(c, n, n + 1, z_words, eval_out))) return rc;
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
                         uint64_t* eval_out) {
  return c->curve == 0
             ? open_quotient_t<BnFr>(c, d_polys, lens, k, stride, z_words, xi_words, d_quot_out, quot_len, eval_out)
             : open_quotient_t<BlsFr>(c, d_polys, lens, k, stride, z_words, xi_words, d_quot_out, quot_len, eval_out);
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
