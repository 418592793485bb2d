// internal.h -- library-private declarations shared by the translation units of
// libkzg_mi355x.so (api.hip, ntt.hip, msm.hip, poly.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "field.h"

namespace kzg {

constexpr int KZG_OK = 0;
constexpr int KZG_ERR_ARG = -1;      // bad argument
constexpr int KZG_ERR_HIP = -2;      // HIP runtime error (see kzg_last_error)
constexpr int KZG_ERR_NODEV = -3;    // no usable gfx950 device
constexpr int KZG_ERR_DEGREE = -4;   // polynomial longer than the SRS (kzg.py:103-106)
constexpr int KZG_ERR_ALLOC = -5;

// A cached evaluation domain: everything the NTT kernels need for one
// (log_n, w, direction).  Built once per distinct root (callers reuse the same
// domain for every transform of a proof: plonk/encoder.py:49).
struct NttDomain {
  uint32_t log_n = 0;
  int inverse = 0;
  uint32_t w[8] = {0};        // caller's root, canonical words
  uint32_t kmax = 0;          // stage table holds (w^(n/2^kmax))^e, e < 2^(kmax-1)
  uint32_t h = 0;             // twist exponent split e = hi*2^h + lo
  uint32_t* d_stage = nullptr;   // [2^(kmax-1)][9] Montgomery limbs
  uint32_t* d_twA = nullptr;     // [2^(log_n-h)][9]  (w^(2^h))^u  (x n^-1 when inverse)
  uint32_t* d_twB = nullptr;     // [2^h][9]          w^l
  uint32_t* d_scale = nullptr;   // [9] n^-1 for a single-pass inverse transform; null: the last pass only reduces
  uint32_t* d_twist = nullptr;   // [n][9] w^(t*v) for two-pass sizes (row t of N2 entries); only with KZG_NTT_TWIST_TABLE=1
  uint64_t last_use = 0;
};

// Optional per-kernel timing with HIP events on the context's stream (bench.py's roofline
// leg): each named span accumulates elapsed ms and a launch count.
struct ProfSpan {
  std::string name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double total_ms = 0;
  uint64_t count = 0;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct Ctx {
  int curve = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  std::vector<NttDomain> domains;
  uint64_t tick = 0;
  uint32_t ntt_lds_attr_set = 0;   // per ntt_pass_kernel instantiation: hipFuncAttributeMaxDynamicSharedMemorySize applies per device
  DevBuf ntt_scratch;     // pass-1 output of two-pass transforms
  DevBuf io;              // staging for the host-pointer entry points
  DevBuf poly_tmp[4];     // open(): combined polynomial + suffix values; scratch of the vector primitives
  DevBuf scan_tmp;        // open(): chunk values, tile aggregates, power tables (kept between the shard calls)
  size_t open_shard_n = 0;                // slice length between kzg_open_shard_begin / _finish
  uint32_t open_shard_tb = 0;             // tile width the slice's aggregates were formed with
  // kzg_ctx_set_tuning: 0 = the library's own choice
  int tune_ntt_tile_log = 0;              // LDS tile of the transform (8..12)
  int tune_open_tb = 0;                   // threads per tile of the opening's scan (128 | 256)
  int tune_open_direct_max = 0;           // tiles up to which every tile sums all aggregates above it
  int last_ntt_tile_log = 0;              // what the last transform used (kzg_prof_read "ntt_tile_log")
  void* msm_work = nullptr;               // MsmWork (msm.hip)
  bool prof_on = false;
  std::vector<ProfSpan> prof;
  unsigned long long* clk_probe = nullptr;   // [4] shader-clock / 100 MHz ticks of one wave of msm_accumulate ([0..1]) and of ntt_pass ([2..3]); profiling only
  std::vector<hipStream_t> aux_streams;   // commit pipeline
  std::vector<hipEvent_t> aux_events;
};

int set_err(Ctx* c, int code, const char* what, hipError_t e = hipSuccess);
// RAII span: records an event pair around the launches issued while it is alive (no-op when profiling is off)
struct ProfScope {
  Ctx* c;
  ProfSpan* span = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
  ProfScope(Ctx* ctx, const char* name, hipStream_t stream = nullptr);
  ~ProfScope();
};
int ensure_buf(Ctx* c, DevBuf& b, size_t bytes);

#define KZG_HIP(c, call)                                         \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) return set_err((c), KZG_ERR_HIP, #call, e__); \
  } while (0)

// msm.hip: is an accumulate kernel of the commit pipeline queued or running?
bool msm_accumulate_in_flight(Ctx* c);

// ntt.hip
int ntt_run_device(Ctx* c, uint32_t* d_data, uint32_t log_n, const uint32_t* w_words, int inverse,
                   uint32_t batch);
int ntt_partial_device(Ctx* c, uint32_t* d_data, uint32_t log_n, const uint32_t* w_words, int inverse, int rows_pass,
                       uint64_t count, uint64_t col_base);
int ntt_rows_exchange_device(Ctx* c, const uint32_t* d_src, uint32_t* d_dst, uint32_t log_n, const uint32_t* w_words,
                             int inverse, uint64_t n_rows, uint32_t world, int blocked_out);
int fft_ragged_device(Ctx* c, uint32_t* d_data, uint64_t n, const uint32_t* w_words, int inverse);
void ntt_free_domains(Ctx* c);

// poly.hip: device vector / polynomial primitives over Fr
int fr_vec_binary(Ctx* c, int op, size_t n, const uint32_t* a, const uint32_t* b, uint32_t* out);
int fr_vec_lincomb(Ctx* c, size_t n, size_t k, const uint32_t* const* ptrs, const size_t* lens, const uint32_t* scalars,
                   uint32_t* out);
int fr_vec_mul_powers(Ctx* c, size_t n, const uint32_t* a, const uint32_t* s, const uint32_t* cc, uint32_t* out);
int fr_vec_inverse(Ctx* c, size_t n, const uint32_t* a, uint32_t* out);
int fr_vec_prefix_product(Ctx* c, size_t n, const uint32_t* a, uint32_t* out);
int fr_poly_eval(Ctx* c, size_t n, const uint32_t* a, const uint32_t* z, uint64_t* out);

}  // namespace kzg
