// api.hip -- the extern "C" surface of libkzg_mi355x.so (include/kzg_mi355x.h).
#include "internal.h"
#include "ec.h"
#include "msm.h"
#include "../../include/kzg_mi355x.h"
#include <cstdio>
#include <cstring>
#include <new>

struct kzg_ctx {
  kzg::Ctx c;
};
struct kzg_srs {
  kzg::Srs* s;
};

namespace kzg {
int device_any_nonzero(Ctx* c, const uint32_t* d_words, size_t from, size_t to, bool* out);   // poly.hip
}

namespace kzg {

int set_err(Ctx* c, int code, const char* what, hipError_t e) {
  if (c) {
    c->err = what ? what : "";
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}

ProfScope::ProfScope(Ctx* ctx, const char* name, hipStream_t stream) : c(ctx) {
  if (!c->prof_on) return;
  st = stream ? stream : c->stream;
  for (auto& sp : c->prof)
    if (sp.name == name) { span = &sp; break; }
  if (!span) {
    c->prof.reserve(64);   // spans are referenced by pointer while alive
    c->prof.push_back(ProfSpan{name, {}, 0, 0});
    span = &c->prof.back();
  }
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { span = nullptr; return; }
  hipEventRecord(e0, st);
}
ProfScope::~ProfScope() {
  if (!span) return;
  hipEventRecord(e1, st);
  span->pending.emplace_back(e0, e1);
}

int ensure_buf(Ctx* c, DevBuf& b, size_t bytes) {
  if (b.cap >= bytes) return KZG_OK;
  if (b.p) {
    KZG_HIP(c, hipStreamSynchronize(c->stream));
    KZG_HIP(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    b.p = nullptr;
    return set_err(c, KZG_ERR_ALLOC, "hipMalloc", e);
  }
  b.cap = bytes;
  return KZG_OK;
}

}  // namespace kzg

namespace kzg {
// sum of n affine points on the HOST (XYZZ accumulator, mixed additions of ec.h -- exact for O, P + P, P - P --, one
// inversion at the end): the "reduce" of partial MSM results that RCCL has no operator for
template <class C>
static int g1_sum_t(const uint64_t* xy, const uint8_t* inf, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
  using F = typename C::Fp;
  using Fd = Field<F>;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(xy);
  XYZZ<C> acc = Ec<C>::infinity();
  for (size_t i = 0; i < n; ++i) {
    if (inf && inf[i]) continue;
    const Fe<F> x = Fd::from_words(w + i * 2 * F::NW), y = Fd::from_words(w + i * 2 * F::NW + F::NW);
    // coordinates must be canonical field elements of a point on the curve
    for (int half = 0; half < 2; ++half) {
      const uint32_t* q = w + i * 2 * F::NW + half * F::NW;
      bool below = false;
      for (int k = F::NW - 1; k >= 0 && !below; --k) {
        if (q[k] < F::PW[k]) below = true;
        else if (q[k] > F::PW[k]) return KZG_ERR_ARG;
      }
      if (!below) return KZG_ERR_ARG;
    }
    const Fe<F> xm = Fd::reduce(Fd::to_mont(x)), ym = Fd::reduce(Fd::to_mont(y));
    if (!Ec<C>::on_curve(xm, ym)) return KZG_ERR_ARG;
    acc = Ec<C>::madd(acc, xm, ym);
  }
  const Affine<C> a = Ec<C>::to_affine(acc);
  uint32_t* o = reinterpret_cast<uint32_t*>(out_xy);
  if (a.inf) {
    memset(o, 0, 2 * F::NW * 4);
    *out_inf = 1;
    return KZG_OK;
  }
  Fd::to_words(Fd::from_mont(a.x), o);
  Fd::to_words(Fd::from_mont(a.y), o + F::NW);
  *out_inf = 0;
  return KZG_OK;
}
}  // namespace kzg

using namespace kzg;

extern "C" {

int kzg_g1_sum(int curve_id, const uint64_t* xy, const uint8_t* inf, size_t n, uint64_t* out_xy, uint8_t* out_inf) {
  if ((n && !xy) || !out_xy || !out_inf) return KZG_ERR_ARG;
  if (curve_id == KZG_CURVE_BN254) return g1_sum_t<Bn254>(xy, inf, n, out_xy, out_inf);
  if (curve_id == KZG_CURVE_BLS12_381) return g1_sum_t<Bls12_381>(xy, inf, n, out_xy, out_inf);
  return KZG_ERR_ARG;
}

int kzg_abi_version(void) { return 1; }

int kzg_fp_limbs(int curve_id) {
  if (curve_id == KZG_CURVE_BN254) return 4;
  if (curve_id == KZG_CURVE_BLS12_381) return 6;
  return 0;
}

int kzg_ctx_create(int curve_id, int device_id, kzg_ctx** out) {
  if (!out) return KZG_ERR_ARG;
  *out = nullptr;
  if (curve_id != KZG_CURVE_BN254 && curve_id != KZG_CURVE_BLS12_381) return KZG_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev)
    return KZG_ERR_NODEV;   // no CPU fallback by design
  if (hipSetDevice(device_id) != hipSuccess) return KZG_ERR_NODEV;
  kzg_ctx* h = new (std::nothrow) kzg_ctx();
  if (!h) return KZG_ERR_ALLOC;
  h->c.curve = curve_id;
  h->c.device = device_id;
  if (hipStreamCreateWithFlags(&h->c.stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return KZG_ERR_HIP;
  }
  h->c.own_stream = true;
  *out = h;
  return KZG_OK;
}

void kzg_ctx_destroy(kzg_ctx* ctx) {
  if (!ctx) return;
  Ctx* c = &ctx->c;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  ntt_free_domains(c);
  msm_free_work(c);
  hipFree(c->ntt_scratch.p);
  hipFree(c->io.p);
  hipFree(c->clk_probe);
  hipFree(c->scan_tmp.p);
  for (auto& b : c->poly_tmp) hipFree(b.p);
  for (auto s : c->aux_streams) hipStreamDestroy(s);
  for (auto e : c->aux_events) hipEventDestroy(e);
  if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
  delete ctx;
}

const char* kzg_last_error(const kzg_ctx* ctx) { return ctx ? ctx->c.err.c_str() : "null context"; }

int kzg_ctx_set_stream(kzg_ctx* ctx, void* hip_stream) {
  if (!ctx) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  if (hip_stream) {
    // HIP offers no validation of a stream handle: every entry point, hipStreamQuery included, dereferences it (a
    // readable buffer that is no stream crashed the process on ROCm 7.2, profiles/r03_stream_query_crash.log, and a
    // query on a capturing stream would invalidate the capture).  So nothing is asked of the runtime here: the two
    // documented aliases (hipStreamLegacy, hipStreamPerThread) are taken as such, any other small integer or
    // misaligned value cannot be a runtime object and is refused, and for everything else the caller guarantees a
    // live hipStream_t of this process.
    const uintptr_t hv = reinterpret_cast<uintptr_t>(hip_stream);
    const bool alias = hip_stream == static_cast<void*>(hipStreamLegacy) || hip_stream == static_cast<void*>(hipStreamPerThread);
    if (!alias && (hv < 65536 || (hv & 7)))
      return set_err(c, KZG_ERR_ARG, "kzg_ctx_set_stream: not a stream handle (small integer or misaligned value)");
    if (c->own_stream && c->stream) {
      KZG_HIP(c, hipStreamSynchronize(c->stream));
      KZG_HIP(c, hipStreamDestroy(c->stream));
    }
    // hipStreamLegacy names HIP's null stream (torch's default stream).  Internally that is the plain
    // null handle, which every runtime entry point accepts (event record / wait on the alias do not).
    c->stream = hip_stream == static_cast<void*>(hipStreamLegacy) ? nullptr : static_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
  } else if (!c->own_stream) {
    KZG_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return KZG_OK;
}

int kzg_ctx_set_tuning(kzg_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  const std::string k(key);
  if (k == "ntt_tile_log") {
    if (value != 0 && (value < 8 || value > 12)) return set_err(c, KZG_ERR_ARG, "ntt_tile_log: 0 or 8..12");
    c->tune_ntt_tile_log = (int)value;
  } else if (k == "open_tile_threads") {
    if (value != 0 && value != 128 && value != 256) return set_err(c, KZG_ERR_ARG, "open_tile_threads: 0, 128 or 256");
    c->tune_open_tb = (int)value;
  } else if (k == "open_direct_tiles") {
    if (value < 0 || value > (1 << 20)) return set_err(c, KZG_ERR_ARG, "open_direct_tiles: 0 .. 2^20");
    c->tune_open_direct_max = (int)value;
  } else {
    return set_err(c, KZG_ERR_ARG, "kzg_ctx_set_tuning: unknown key");
  }
  return KZG_OK;
}

int kzg_ctx_synchronize(kzg_ctx* ctx) {
  if (!ctx) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  return KZG_OK;
}

int kzg_ntt_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                   uint32_t batch) {
  if (!ctx || !d_data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return ntt_run_device(c, static_cast<uint32_t*>(d_data), log_n, reinterpret_cast<const uint32_t*>(w),
                        inverse ? 1 : 0, batch);
}

int kzg_ntt_columns_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                           uint64_t n_cols, uint64_t col_base) {
  if (!ctx || !d_data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return ntt_partial_device(c, static_cast<uint32_t*>(d_data), log_n, reinterpret_cast<const uint32_t*>(w),
                            inverse ? 1 : 0, 0, n_cols, col_base);
}

int kzg_ntt_rows_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                        uint64_t n_rows) {
  if (!ctx || !d_data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return ntt_partial_device(c, static_cast<uint32_t*>(d_data), log_n, reinterpret_cast<const uint32_t*>(w),
                            inverse ? 1 : 0, 1, n_rows, 0);
}

int kzg_ntt_rows_twist_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                              uint64_t n_rows, uint64_t row_base) {
  if (!ctx || !d_data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return ntt_partial_device(c, static_cast<uint32_t*>(d_data), log_n, reinterpret_cast<const uint32_t*>(w),
                            inverse ? 1 : 0, 2, n_rows, row_base);
}

int kzg_ntt_columns_plain_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                                 uint64_t n_cols) {
  if (!ctx || !d_data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return ntt_partial_device(c, static_cast<uint32_t*>(d_data), log_n, reinterpret_cast<const uint32_t*>(w),
                            inverse ? 1 : 0, 3, n_cols, 0);
}

int kzg_ntt_rows_exchange_device(kzg_ctx* ctx, const void* d_src, void* d_dst, uint32_t log_n, const uint64_t w[4],
                                 int inverse, uint64_t n_rows, uint32_t world, int blocked_out) {
  if (!ctx || !d_src || !d_dst || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return ntt_rows_exchange_device(c, static_cast<const uint32_t*>(d_src), static_cast<uint32_t*>(d_dst), log_n,
                                  reinterpret_cast<const uint32_t*>(w), inverse ? 1 : 0, n_rows, world, blocked_out);
}

int kzg_ntt(kzg_ctx* ctx, uint64_t* data, uint32_t log_n, const uint64_t w[4], int inverse) {
  if (!ctx || !data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  if (log_n > 24) return set_err(c, KZG_ERR_ARG, "kzg_ntt: log_n > 24 not supported");
  KZG_HIP(c, hipSetDevice(c->device));
  const size_t bytes = (size_t)32 << log_n;
  int rc = ensure_buf(c, c->io, bytes);
  if (rc) return rc;
  KZG_HIP(c, hipMemcpyAsync(c->io.p, data, bytes, hipMemcpyHostToDevice, c->stream));
  rc = ntt_run_device(c, static_cast<uint32_t*>(c->io.p), log_n, reinterpret_cast<const uint32_t*>(w),
                      inverse ? 1 : 0, 1);
  if (rc) return rc;
  KZG_HIP(c, hipMemcpyAsync(data, c->io.p, bytes, hipMemcpyDeviceToHost, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  return KZG_OK;
}

int kzg_fft_ff_any_device(kzg_ctx* ctx, void* d_data, size_t n, const uint64_t w[4], int inverse) {
  if (!ctx || !d_data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  if (n && !(n & (n - 1))) {   // power of two: the tiled kernels compute the same recursion
    uint32_t log_n = 0;
    while ((1ull << log_n) < n) ++log_n;
    return ntt_run_device(c, static_cast<uint32_t*>(d_data), log_n, reinterpret_cast<const uint32_t*>(w),
                          inverse ? 1 : 0, 1);
  }
  return fft_ragged_device(c, static_cast<uint32_t*>(d_data), n, reinterpret_cast<const uint32_t*>(w), inverse ? 1 : 0);
}

int kzg_fft_ff_any(kzg_ctx* ctx, uint64_t* data, size_t n, const uint64_t w[4], int inverse) {
  if (!ctx || !data || !w) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  if (n == 0 || n > ((size_t)1 << 24)) return set_err(c, KZG_ERR_ARG, "kzg_fft_ff_any: length must be in [1, 2^24]");
  KZG_HIP(c, hipSetDevice(c->device));
  const size_t bytes = n * 32;
  int rc = ensure_buf(c, c->io, bytes);
  if (rc) return rc;
  KZG_HIP(c, hipMemcpyAsync(c->io.p, data, bytes, hipMemcpyHostToDevice, c->stream));
  rc = kzg_fft_ff_any_device(ctx, c->io.p, n, w, inverse);
  if (rc) return rc;
  KZG_HIP(c, hipMemcpyAsync(data, c->io.p, bytes, hipMemcpyDeviceToHost, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));
  return KZG_OK;
}

int kzg_srs_load_g1(kzg_ctx* ctx, const uint64_t* xy, const uint8_t* inf, size_t n, kzg_srs** out) {
  if (!ctx || !xy || !out) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  *out = nullptr;
  KZG_HIP(c, hipSetDevice(c->device));
  Srs* s = nullptr;
  int rc = srs_load(c, xy, inf, n, &s);
  if (rc) return rc;
  *out = new kzg_srs{s};
  return KZG_OK;
}

int kzg_srs_generate_range(kzg_ctx* ctx, const uint64_t tau[4], size_t start, size_t n, kzg_srs** out) {
  if (!ctx || !tau || !out) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  *out = nullptr;
  KZG_HIP(c, hipSetDevice(c->device));
  Srs* s = nullptr;
  int rc = srs_generate(c, tau, start, n, &s);
  if (rc) return rc;
  *out = new kzg_srs{s};
  return KZG_OK;
}

int kzg_srs_generate_strided(kzg_ctx* ctx, const uint64_t tau[4], size_t start, size_t n, size_t run_len,
                             size_t inner_stride, size_t outer_stride, kzg_srs** out) {
  if (!ctx || !tau || !out || run_len == 0) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  *out = nullptr;
  KZG_HIP(c, hipSetDevice(c->device));
  Srs* s = nullptr;
  int rc = srs_generate(c, tau, start, n, &s, run_len, inner_stride, outer_stride);
  if (rc) return rc;
  *out = new kzg_srs{s};
  return KZG_OK;
}

int kzg_srs_generate(kzg_ctx* ctx, const uint64_t tau[4], size_t n, kzg_srs** out) {
  return kzg_srs_generate_range(ctx, tau, 0, n, out);
}

int kzg_srs_export(kzg_ctx* ctx, const kzg_srs* srs, size_t start, size_t count, uint64_t* xy, uint8_t* inf) {
  if (!ctx || !srs || !xy || !inf) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return srs_export(c, srs->s, start, count, xy, inf);
}

size_t kzg_srs_size(const kzg_srs* srs) { return srs ? srs->s->n : 0; }

void kzg_srs_free(kzg_srs* srs) {
  if (!srs) return;
  srs_free(srs->s);
  delete srs;
}

int kzg_commit_device(kzg_ctx* ctx, const kzg_srs* srs, const void* d_scalars, const size_t* lens, size_t n_polys,
                      size_t stride, uint64_t* out_xy, uint8_t* out_inf) {
  if (!ctx || !srs || !lens || !out_xy || !out_inf || (n_polys && !d_scalars)) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return commit_device(c, srs->s, static_cast<const uint32_t*>(d_scalars), lens, n_polys, stride, out_xy, out_inf);
}

int kzg_commit_device_async(kzg_ctx* ctx, const kzg_srs* srs, const void* d_scalars, const size_t* lens,
                            size_t n_polys, size_t stride, uint64_t* out_xy, uint8_t* out_inf) {
  if (!ctx || !srs || !lens || !out_xy || !out_inf || (n_polys && !d_scalars)) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return commit_device(c, srs->s, static_cast<const uint32_t*>(d_scalars), lens, n_polys, stride, out_xy, out_inf,
                       false);
}

int kzg_commit_flush(kzg_ctx* ctx) {
  if (!ctx) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return commit_flush(c);
}

int kzg_commit(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* scalars, const size_t* lens, size_t n_polys,
               size_t stride, uint64_t* out_xy, uint8_t* out_inf) {
  if (!ctx || !srs || !lens || !out_xy || !out_inf || (n_polys && stride && !scalars)) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  const size_t bytes = n_polys * stride * 32;
  int rc = ensure_buf(c, c->io, bytes ? bytes : 32);
  if (rc) return rc;
  // One polynomial at a time: the copy of polynomial p + 1 is ordered on the context's stream behind the prep of
  // polynomial p (which is all that reads p's scalars) and so travels while p is being accumulated.
  const size_t fpw = (c->curve == 0 ? Bn254::Fp::NW : Bls12_381::Fp::NW) / 2;   // 64-bit words per coordinate
  auto* io = static_cast<uint32_t*>(c->io.p);
  for (size_t p = 0; p < n_polys; ++p) {   // all arguments are checked before any work is queued
    if (lens[p] > srs->s->n) return set_err(c, KZG_ERR_DEGREE, "polynomial longer than the commitment key");
    if (lens[p] > stride) return set_err(c, KZG_ERR_ARG, "kzg_commit: lens[p] > stride");
  }
  for (size_t p = 0; p < n_polys; ++p) {
    if (lens[p])
      KZG_HIP(c, hipMemcpyAsync(io + p * stride * 8, scalars + p * stride * 4, lens[p] * 32, hipMemcpyHostToDevice,
                                c->stream));
    rc = commit_device(c, srs->s, io + p * stride * 8, lens + p, 1, stride, out_xy + p * 2 * fpw, out_inf + p, false);
    if (rc) { commit_flush(c); return rc; }
  }
  return commit_flush(c);
}

int kzg_open_device(kzg_ctx* ctx, const kzg_srs* srs, const void* d_polys, const size_t* lens, size_t k,
                    size_t stride, const uint64_t z[4], const uint64_t xi[4], uint64_t* out_xy, uint8_t* out_inf,
                    uint64_t* eval_out) {
  if (!ctx || !srs || !z || !xi || !out_xy || !out_inf || (k && (!lens || !d_polys))) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  uint32_t* d_quot = nullptr;
  size_t qlen = 0;
  uint64_t ev[4];
  int rc = open_quotient_device(c, static_cast<const uint32_t*>(d_polys), lens, k, stride,
                                reinterpret_cast<const uint32_t*>(z), reinterpret_cast<const uint32_t*>(xi), &d_quot,
                                &qlen, ev);
  if (rc) return rc;
  if (eval_out) memcpy(eval_out, ev, 32);
  const size_t srs_n = srs->s->n;
  if (qlen > srs_n) {
    // The reference checks the DEGREE of the witness (kzg.py:103 via :157): coefficients
    // beyond the key are fine as long as they are zero.
    bool nz = false;
    rc = device_any_nonzero(c, d_quot, srs_n, qlen, &nz);
    if (rc) return rc;
    if (nz) return set_err(c, KZG_ERR_DEGREE, "witness polynomial longer than the commitment key");
    qlen = srs_n;
  }
  return commit_device(c, srs->s, d_quot, &qlen, 1, qlen ? qlen : 1, out_xy, out_inf);
}

int kzg_open_device_async(kzg_ctx* ctx, const kzg_srs* srs, const void* d_polys, const size_t* lens, size_t k,
                          size_t stride, const uint64_t z[4], const uint64_t xi[4], uint64_t* out_xy, uint8_t* out_inf,
                          uint64_t* eval_out) {
  if (!ctx || !srs || !z || !xi || !out_xy || !out_inf || !eval_out || (k && (!lens || !d_polys))) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  uint32_t* d_quot = nullptr;
  size_t qlen = 0;
  int rc = open_quotient_device(c, static_cast<const uint32_t*>(d_polys), lens, k, stride,
                                reinterpret_cast<const uint32_t*>(z), reinterpret_cast<const uint32_t*>(xi), &d_quot,
                                &qlen, eval_out, /*sync=*/false);
  if (rc) return rc;
  if (!d_quot) {   // every polynomial empty: witness 0, evaluation 0 (already in eval_out)
    memset(out_xy, 0, (size_t)kzg_fp_limbs(c->curve) * 16);
    *out_inf = 1;
    memset(eval_out, 0, 32);
    return KZG_OK;
  }
  const size_t srs_n = srs->s->n;
  if (qlen > srs_n) {   // rare: the degree check of kzg.py:103 needs the coefficients beyond the key, which synchronises
    bool nz = false;
    rc = device_any_nonzero(c, d_quot, srs_n, qlen, &nz);
    if (rc) return rc;
    if (nz) return set_err(c, KZG_ERR_DEGREE, "witness polynomial longer than the commitment key");
    qlen = srs_n;
  }
  // S_0 = combined(z) sits right below the quotient in the scan's buffer (poly.hip layout)
  return commit_device(c, srs->s, d_quot, &qlen, 1, qlen ? qlen : 1, out_xy, out_inf, /*drain=*/false, d_quot - 8,
                       eval_out);
}

int kzg_open(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* polys, const size_t* lens, size_t k, size_t stride,
             const uint64_t z[4], const uint64_t xi[4], uint64_t* out_xy, uint8_t* out_inf, uint64_t* eval_out) {
  if (!ctx || !srs || !z || !xi || !out_xy || !out_inf || (k && (!lens || !polys))) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  const size_t bytes = k * stride * 32;
  int rc = ensure_buf(c, c->io, bytes ? bytes : 32);
  if (rc) return rc;
  if (bytes) KZG_HIP(c, hipMemcpyAsync(c->io.p, polys, bytes, hipMemcpyHostToDevice, c->stream));
  return kzg_open_device(ctx, srs, c->io.p, lens, k, stride, z, xi, out_xy, out_inf, eval_out);
}

int kzg_open_shard_begin(kzg_ctx* ctx, const void* d_polys, const size_t* lens, size_t k, size_t stride,
                         const uint64_t z[4], const uint64_t xi[4], uint64_t* chunk_eval_out) {
  if (!ctx || !z || !xi || !chunk_eval_out || (k && (!lens || !d_polys))) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  return open_shard_begin_device(c, static_cast<const uint32_t*>(d_polys), lens, k, stride,
                                 reinterpret_cast<const uint32_t*>(z), reinterpret_cast<const uint32_t*>(xi),
                                 chunk_eval_out);
}

int kzg_open_shard_finish(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t z[4], const uint64_t carry[4],
                          int first_rank, uint64_t* out_xy, uint8_t* out_inf, uint64_t* eval_out) {
  if (!ctx || !srs || !z || !carry || !out_xy || !out_inf) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipSetDevice(c->device));
  uint32_t* d_vec = nullptr;
  size_t len = 0;
  uint64_t ev[4];
  int rc = open_shard_finish_device(c, reinterpret_cast<const uint32_t*>(z), reinterpret_cast<const uint32_t*>(carry),
                                    first_rank, &d_vec, &len, ev);
  if (rc) return rc;
  if (eval_out) memcpy(eval_out, ev, 32);
  if (len > srs->s->n) return set_err(c, KZG_ERR_DEGREE, "quotient slice longer than the key shard");
  return commit_device(c, srs->s, d_vec, &len, 1, len ? len : 1, out_xy, out_inf);
}

#define KZG_VEC_ENTER()                          \
  if (!ctx) return KZG_ERR_ARG;                  \
  Ctx* c = &ctx->c;                              \
  KZG_HIP(c, hipSetDevice(c->device));

int kzg_fr_vec_op(kzg_ctx* ctx, int op, size_t n, const void* d_a, const void* d_b, void* d_out) {
  KZG_VEC_ENTER();
  if (op < 0 || op > 2 || (n && (!d_a || !d_b || !d_out))) return KZG_ERR_ARG;
  return fr_vec_binary(c, op, n, static_cast<const uint32_t*>(d_a), static_cast<const uint32_t*>(d_b),
                       static_cast<uint32_t*>(d_out));
}

int kzg_fr_vec_lincomb(kzg_ctx* ctx, size_t n, size_t k, const void* const* d_ptrs, const size_t* lens,
                       const uint64_t* scalars, void* d_out) {
  KZG_VEC_ENTER();
  if (k && (!d_ptrs || !lens || !scalars)) return KZG_ERR_ARG;
  if (n && !d_out) return KZG_ERR_ARG;
  return fr_vec_lincomb(c, n, k, reinterpret_cast<const uint32_t* const*>(d_ptrs), lens,
                        reinterpret_cast<const uint32_t*>(scalars), static_cast<uint32_t*>(d_out));
}

int kzg_fr_vec_mul_powers(kzg_ctx* ctx, size_t n, const void* d_a, const uint64_t s[4], const uint64_t c0[4],
                          void* d_out) {
  KZG_VEC_ENTER();
  if (!s || !c0 || (n && (!d_a || !d_out))) return KZG_ERR_ARG;
  return fr_vec_mul_powers(c, n, static_cast<const uint32_t*>(d_a), reinterpret_cast<const uint32_t*>(s),
                           reinterpret_cast<const uint32_t*>(c0), static_cast<uint32_t*>(d_out));
}

int kzg_fr_vec_inverse(kzg_ctx* ctx, size_t n, const void* d_a, void* d_out) {
  KZG_VEC_ENTER();
  if (n && (!d_a || !d_out)) return KZG_ERR_ARG;
  return fr_vec_inverse(c, n, static_cast<const uint32_t*>(d_a), static_cast<uint32_t*>(d_out));
}

int kzg_fr_vec_prefix_product(kzg_ctx* ctx, size_t n, const void* d_a, void* d_out) {
  KZG_VEC_ENTER();
  if (n && (!d_a || !d_out)) return KZG_ERR_ARG;
  return fr_vec_prefix_product(c, n, static_cast<const uint32_t*>(d_a), static_cast<uint32_t*>(d_out));
}

int kzg_fr_poly_eval(kzg_ctx* ctx, size_t n, const void* d_a, const uint64_t z[4], uint64_t out[4]) {
  KZG_VEC_ENTER();
  if (!z || !out || (n && !d_a)) return KZG_ERR_ARG;
  return fr_poly_eval(c, n, static_cast<const uint32_t*>(d_a), reinterpret_cast<const uint32_t*>(z), out);
}

int kzg_prof_enable(kzg_ctx* ctx, int on) {
  if (!ctx) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  if (on && !c->clk_probe) {
    KZG_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->clk_probe), 32));
    KZG_HIP(c, hipMemset(c->clk_probe, 0, 32));
  }
  c->prof_on = on != 0;
  return KZG_OK;
}

int kzg_prof_reset(kzg_ctx* ctx) {
  if (!ctx) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipDeviceSynchronize());
  for (auto& sp : c->prof) {
    for (auto& pr : sp.pending) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
    sp.pending.clear();
    sp.total_ms = 0;
    sp.count = 0;
  }
  if (c->clk_probe) KZG_HIP(c, hipMemset(c->clk_probe, 0, 32));
  return KZG_OK;
}

int kzg_prof_read(kzg_ctx* ctx, const char* name, double* total_ms, uint64_t* count) {
  if (!ctx || !name || !total_ms || !count) return KZG_ERR_ARG;
  Ctx* c = &ctx->c;
  KZG_HIP(c, hipDeviceSynchronize());
  *total_ms = 0;
  *count = 0;
  const bool acc_clk = std::string(name) == "msm_accumulate_shader_mhz", ntt_clk = std::string(name) == "ntt_pass_shader_mhz";
  if (std::string(name) == "ntt_tile_log") {   // not a span: log2 of the LDS tile the last transform took
    *total_ms = (double)c->last_ntt_tile_log;
    *count = c->last_ntt_tile_log ? 1 : 0;
    return KZG_OK;
  }
  if (acc_clk || ntt_clk) {   // not a span: the shader clock (MHz) the kernel's probing wave ran at
    unsigned long long t[4] = {0, 0, 0, 0};
    if (c->clk_probe) KZG_HIP(c, hipMemcpy(t, c->clk_probe, 32, hipMemcpyDeviceToHost));
    const unsigned long long* q = t + (ntt_clk ? 2 : 0);
    if (q[1]) { *total_ms = 100.0 * (double)q[0] / (double)q[1]; *count = 1; }
    return KZG_OK;
  }
  for (auto& sp : c->prof) {
    if (sp.name != name) continue;
    for (auto& pr : sp.pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { sp.total_ms += ms; sp.count += 1; }
      hipEventDestroy(pr.first);
      hipEventDestroy(pr.second);
    }
    sp.pending.clear();
    *total_ms = sp.total_ms;
    *count = sp.count;
    return KZG_OK;
  }
  return KZG_OK;
}

}  // extern "C"
