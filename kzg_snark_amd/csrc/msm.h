// msm.h -- library-private interface of msm.hip (SRS tables and the commit pipeline).
#pragma once
#include "internal.h"

namespace kzg {

// Device-resident commitment key: the reference's `ck` = [tau^i G1] (kzg.py:70-72),
// expanded to NWIN window multiples per point (layout: msm.hip header comment).
struct Srs {
  size_t n = 0;
  int curve = 0;
  int win_bits = 16;          // Pippenger window c: 20 for keys of >= 2^18 points, else 16
  int nwin = 16;              // ceil(256 / c) table windows
  uint32_t* recs = nullptr;   // [nwin][n] records of Curve::REC_WORDS words
};

int srs_load(Ctx* c, const uint64_t* xy, const uint8_t* inf, size_t n, Srs** out);
// run_len = 0: the contiguous range tau^(start + i); otherwise record i = tau^(start + (i / run_len) * outer_stride +
// (i % run_len) * inner_stride)
int srs_generate(Ctx* c, const uint64_t* tau, size_t start, size_t n, Srs** out, size_t run_len = 0,
                 size_t inner_stride = 1, size_t outer_stride = 0);
int srs_export(Ctx* c, const Srs* s, size_t start, size_t count, uint64_t* xy, uint8_t* inf);
void srs_free(Srs* s);

// One MSM per polynomial; scalars device-resident, results to host memory (synchronises).
// drain = false leaves up to four polynomials in flight; their outputs are written when their
// slot is recycled by a later call or by commit_flush().
// d_eval / out_eval (pipelined open, n_polys == 1): 32 bytes at d_eval -- written by work already enqueued on the
// context's stream -- are delivered to out_eval when the polynomial's slot is retired, together with its point.
int commit_device(Ctx* c, const Srs* s, const uint32_t* d_scalars, const size_t* lens, size_t n_polys,
                  size_t stride, uint64_t* out_xy, uint8_t* out_inf, bool drain = true,
                  const uint32_t* d_eval = nullptr, uint64_t* out_eval = nullptr);
int commit_flush(Ctx* c);
void msm_free_work(Ctx* c);

// poly.hip: combined = sum_i xi^(i+1) p_i; quotient (combined - combined(z)) / (X - z).
// d_quot receives max_len-1 coefficients (canonical words); eval_out the value combined(z).
int open_quotient_device(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                         const uint32_t* z_words, const uint32_t* xi_words, uint32_t** d_quot_out,
                         size_t* quot_len, uint64_t* eval_out, bool sync = true);

int open_shard_begin_device(Ctx* c, const uint32_t* d_polys, const size_t* lens, size_t k, size_t stride,
                            const uint32_t* z_words, const uint32_t* xi_words, uint64_t* chunk_eval_out);
int open_shard_finish_device(Ctx* c, const uint32_t* z_words, const uint32_t* carry_words, int first_rank,
                             uint32_t** d_vec_out, size_t* vec_len, uint64_t* eval_out);

}  // namespace kzg
