// api.hip -- the extern "C" surface of libkzg_mi355x.so (include/kzg_mi355x.h).
#include "internal.h"
#include "../../include/kzg_mi355x.h"
#include <cstdio>
#include <cstring>
#include <new>

struct kzg_ctx {
  kzg::Ctx c;
};

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

using namespace kzg;

extern "C" {

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
  hipFree(c->ntt_scratch.p);
  hipFree(c->io.p);
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
    if (c->own_stream && c->stream) {
      KZG_HIP(c, hipStreamSynchronize(c->stream));
      KZG_HIP(c, hipStreamDestroy(c->stream));
    }
    c->stream = static_cast<hipStream_t>(hip_stream);
    c->own_stream = false;
  } else if (!c->own_stream) {
    KZG_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
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

}  // extern "C"
