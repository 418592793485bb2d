/*
 * kzg_mi355x.h -- C ABI of libkzg_mi355x.so, the MI355X (gfx950) engine behind the
 * hot path of swusjask/kzg-snark:
 *
 *     fft_ff / ifft_ff / fft_ff_interpolation        (reference fft_ff.py:3,39,60)
 *     KZG.commit / KZG.open                           (reference kzg.py:80,122)
 *     KZG.setup's [tau^i G1] table                    (reference kzg.py:56-78)
 *
 * The reference has no FFI of its own (it is pure Python on SageMath + py_ecc);
 * these entry points are what a ctypes binding placed inside those five
 * functions calls (INTEGRATION.md shows the stubs).  Plain pointers and sizes
 * only; no exceptions cross the boundary.
 *
 * Conventions
 *  - Every function returns 0 on success or a negative KZG_ERR_* code;
 *    kzg_last_error(ctx) gives the message of the last failure on that context.
 *  - Field elements cross the boundary in CANONICAL form (integers < modulus,
 *    not Montgomery), little-endian 64-bit limbs: 4 limbs (32 B) for the scalar
 *    field Fr of either curve and for BN254's Fp, 6 limbs (48 B) for BLS12-381's
 *    Fp.  Scalars and NTT data MUST be reduced (< r): the facade does
 *    int(x) % r exactly like the reference's int(coeff) / Fq(x) coercions.
 *  - G1 points cross as affine (x, y) = 2*FP_LIMBS limbs plus a separate
 *    infinity flag byte (the reference's Z1, kzg.py:43).
 *  - "host" entry points take host pointers, copy in/out and synchronise.
 *    "_device" entry points take device pointers (>= 32-byte aligned), enqueue on
 *    the context's stream and return without synchronising.
 *  - A context is bound to one GPU and one stream; it is not thread-safe (one
 *    context per thread).  The library owns all device memory behind handles.
 *  - There is NO CPU fallback: without a gfx950 device kzg_ctx_create fails
 *    with KZG_ERR_NODEV.
 */
#ifndef KZG_MI355X_H
#define KZG_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZG_CURVE_BN254 0      /* reference default, kzg.py:18,26 */
#define KZG_CURVE_BLS12_381 1  /* kzg.py:31 */

#define KZG_OK 0
#define KZG_ERR_ARG (-1)
#define KZG_ERR_HIP (-2)
#define KZG_ERR_NODEV (-3)
#define KZG_ERR_DEGREE (-4) /* polynomial longer than the SRS: ValueError at kzg.py:103-106 */
#define KZG_ERR_ALLOC (-5)

typedef struct kzg_ctx kzg_ctx;
typedef struct kzg_srs kzg_srs;

/* ABI version of this header (bumped on incompatible change). */
int kzg_abi_version(void);

/* Limbs (uint64) per base-field element for a curve: 4 (BN254) or 6 (BLS12-381); 0 if unknown.
 * Curve selection mirrors KZG.__init__ (kzg.py:26-37). */
int kzg_fp_limbs(int curve_id);

/* Create / destroy a context on HIP device `device_id`. */
int kzg_ctx_create(int curve_id, int device_id, kzg_ctx** out);
void kzg_ctx_destroy(kzg_ctx* ctx);
const char* kzg_last_error(const kzg_ctx* ctx);

/* Use an existing hipStream_t (e.g. torch's current stream) for all work of this context.
 * NULL restores the context's own (non-blocking) stream; to run on HIP's null stream -- torch's
 * default stream -- pass hipStreamLegacy.  A context that keeps its own stream is NOT ordered with
 * work the caller enqueues elsewhere: synchronise, or bind the producer's stream.
 * The handle must be a live hipStream_t of this process (or hipStreamLegacy / hipStreamPerThread): HIP has no way
 * to validate one, so only values that cannot be runtime objects (other small integers, misaligned values) are
 * refused with KZG_ERR_ARG; the stream must outlive its use by the context. */
int kzg_ctx_set_stream(kzg_ctx* ctx, void* hip_stream);
/* Block until everything enqueued on the context's stream has finished. */
int kzg_ctx_synchronize(kzg_ctx* ctx);
/* Fix a choice the library otherwise makes from what is resident on the GPU (tests and A/B timing; results never
 * depend on it).  value 0 restores the library's choice.  Keys:
 *   "ntt_tile_log"       log2 of the transform's LDS tile, 8..12 (default: 11 alone, 10 beside an accumulate kernel)
 *   "open_tile_threads"  threads per tile of the opening's scan, 128 | 256 (default: 256 alone, 128 beside one)
 *   "open_direct_tiles"  tile count up to which every tile sums all tile aggregates above it (default 1024) */
int kzg_ctx_set_tuning(kzg_ctx* ctx, const char* key, int64_t value);

/* ---- NTT: replaces fft_ff (fft_ff.py:3-37) and ifft_ff (fft_ff.py:39-58) -------------
 * data: n = 2^log_n elements of Fr, natural order in and out, transformed in place.
 * w: the caller's root, exactly the `w` argument of fft_ff / ifft_ff.  The result is that
 * of the reference recursion for any w (primitive or not).  inverse != 0 => ifft_ff:
 * transform with w^-1, then scale by n^-1. */
int kzg_ntt(kzg_ctx* ctx, uint64_t* data, uint32_t log_n, const uint64_t w[4], int inverse);
/* Same on `batch` consecutive device-resident arrays of n elements each. */
int kzg_ntt_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                   uint32_t batch);

/* fft_ff / ifft_ff for ANY list length n >= 1 (fft_ff.py:3-58 never checks it): powers of two go
 * to the kernels above; other lengths reproduce the reference recursion level by level -- slices of
 * ceil(n/2) and floor(n/2) elements, n//2 butterflies, result[n-1] left at zero for odd n
 * (fft_ff.py:20-35) -- which is what marlin/prover.py:439-449 receives when it passes list(row_A)
 * with trailing zeros dropped.  inverse: root w^-1, scale by F(n)^-1 (fft_ff.py:53-58).
 * n = 0 is KZG_ERR_ARG (the reference recurses without end). */
int kzg_fft_ff_any(kzg_ctx* ctx, uint64_t* data, size_t n, const uint64_t w[4], int inverse);
int kzg_fft_ff_any_device(kzg_ctx* ctx, void* d_data, size_t n, const uint64_t w[4], int inverse);

/* The two local halves of the multi-GPU four-step transform of n = 2^log_n = N1*N2 elements
 * (N1 = 2^ceil(log_n/2)); kzg_snark_amd/sharding.py exchanges the data between them with
 * all-to-all transposes.  Requires log_n > 12.
 *   columns: d_data is an [N1][n_cols] row-major matrix holding global columns col_base ..
 *            col_base+n_cols-1 of the N1 x N2 view; transformed in place down the columns and
 *            multiplied by the twist w^(row * global column) -- and by n^-1 when inverse (the scale
 *            of fft_ff.py:57-58 rides on the twist; the two halves are only meaningful as a pair).
 *   rows:    d_data is an [n_rows][N2] matrix (n_rows rows of the twisted matrix); transformed in
 *            place along the rows, natural order.
 * n_cols / n_rows: powers of two. */
int kzg_ntt_columns_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                           uint64_t n_cols, uint64_t col_base);
int kzg_ntt_rows_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                        uint64_t n_rows);
/* The same transform taken the other way round: input in the TRANSPOSED layout (what the rows pass above leaves:
 * row rho of the [N1][N2] matrix holds the elements b*N1 + rho, b = 0..N2-1), output in natural order --
 *   rows_twist:    N2-point transforms along n_rows rows (global rows row_base ..), each output (rho, beta) multiplied
 *                  by w^(rho*beta) (and by n^-1 when inverse); in place;
 *   (all-to-all: rows -> whole columns)
 *   columns_plain: N1-point transforms down n_cols columns of an [N1][n_cols] matrix, no twist; element (alpha, beta)
 *                  is then X[alpha*N2 + beta]: an all-to-all back to rows gives every rank a contiguous range.
 * Two all-to-alls instead of the four that "bring the data into range order, then transform" costs.  Unlike the
 * passes above this pair relies on w^n = 1: w must be a primitive 2^log_n-th root of unity (KZG_ERR_ARG otherwise).
 * Requires log_n > 12. */
int kzg_ntt_rows_twist_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                              uint64_t n_rows, uint64_t row_base);
int kzg_ntt_columns_plain_device(kzg_ctx* ctx, void* d_data, uint32_t log_n, const uint64_t w[4], int inverse,
                                 uint64_t n_cols);
/* The rows half reading and writing the all-to-all buffers directly (no repacking copies):
 *   d_src  [world][n_rows][N2/world]  what the columns -> rows all-to-all delivers: block h holds
 *          columns h*N2/world .. of this rank's n_rows rows;
 *   d_dst  blocked_out != 0: [world][n_rows][N2/world] over the OUTPUT index b (block h = outputs
 *          h*N2/world ..), the send buffer of the all-to-all that restores natural order;
 *          blocked_out == 0: [n_rows][N2] plain rows -- the "transposed" result layout: row t of
 *          this rank holds result indices b*N1 + t, b = 0..N2-1 (commit against a key shard in the
 *          same order, kzg_srs_generate_strided; two all-to-alls per transform instead of three).
 * Out of place (d_src != d_dst); world: power of two dividing N2. */
int kzg_ntt_rows_exchange_device(kzg_ctx* ctx, const void* d_src, void* d_dst, uint32_t log_n, const uint64_t w[4],
                                 int inverse, uint64_t n_rows, uint32_t world, int blocked_out);

/* ---- commitment key: the `ck` list of KZG.setup / KZG.commit (kzg.py:56-78, 80) ------
 * kzg_srs_load_g1 uploads n affine G1 points (xy: n x 2*FP_LIMBS limbs; inf: n flag bytes or
 * NULL) and expands them into the engine's device table (msm.hip).  Points are checked to be
 * on the curve (KZG_ERR_ARG otherwise).  The handle replaces passing `ck` on every call. */
int kzg_srs_load_g1(kzg_ctx* ctx, const uint64_t* xy, const uint8_t* inf, size_t n, kzg_srs** out);
/* kzg_srs_generate builds [tau^i * G1], i = 0..n-1, on the device: the G1 half of KZG.setup
 * (kzg.py:70-72) with the secret supplied by the caller (the reference samples it at :67). */
int kzg_srs_generate(kzg_ctx* ctx, const uint64_t tau[4], size_t n, kzg_srs** out);
/* The slice [tau^(start+i) * G1], i = 0..n-1: one rank's shard of a key partitioned by
 * coefficient range across GPUs. */
int kzg_srs_generate_range(kzg_ctx* ctx, const uint64_t tau[4], size_t start, size_t n, kzg_srs** out);
/* Key points in a strided order: point i = tau^e * G1 with
 *   e = start + (i / run_len) * outer_stride + (i % run_len) * inner_stride.
 * The shard a rank needs to commit the "transposed" output of the distributed inverse NTT without
 * reordering it: rows t0..t0+R-1 of N2 coefficients each, coefficient (t, b) having index b*N1 + t
 * => start = t0, run_len = N2, inner_stride = N1, outer_stride = 1, n = R*N2. */
int kzg_srs_generate_strided(kzg_ctx* ctx, const uint64_t tau[4], size_t start, size_t n, size_t run_len,
                             size_t inner_stride, size_t outer_stride, kzg_srs** out);
/* Read points [start, start+count) back as canonical affine coordinates. */
int kzg_srs_export(kzg_ctx* ctx, const kzg_srs* srs, size_t start, size_t count, uint64_t* xy, uint8_t* inf);
size_t kzg_srs_size(const kzg_srs* srs);
void kzg_srs_free(kzg_srs* srs);

/* ---- KZG.commit (kzg.py:80-120) --------------------------------------------------------
 * n_polys coefficient arrays, `stride` elements apart, polynomial p having lens[p] coefficients
 * (low to high).  One affine point per polynomial: out_xy[p] (2*FP_LIMBS limbs), out_inf[p] = 1
 * for the point at infinity (zero polynomial, kzg.py:109).  lens[p] > kzg_srs_size(srs) returns
 * KZG_ERR_DEGREE -- the ValueError of kzg.py:103-106 -- before any work is queued.  Zero coefficients
 * contribute nothing (kzg.py:113-114).  The scalars are in host memory: polynomial p + 1 is copied to
 * the device while polynomial p is being accumulated. */
int kzg_commit(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* scalars, const size_t* lens, size_t n_polys,
               size_t stride, uint64_t* out_xy, uint8_t* out_inf);
/* Same with device-resident scalars; results are still written to host memory (one point per
 * polynomial), so the call synchronises the stream. */
int kzg_commit_device(kzg_ctx* ctx, const kzg_srs* srs, const void* d_scalars, const size_t* lens, size_t n_polys,
                      size_t stride, uint64_t* out_xy, uint8_t* out_inf);

/* Pipelined form: returns after enqueueing; up to four polynomials stay in flight across calls.
 * The scalars are copied at enqueue, in the order of the context's stream: work queued on that
 * stream afterwards (the next transform into the same buffer, say) may overwrite them, anything
 * else must wait for the stream.  out_xy / out_inf stay valid until kzg_commit_flush() returns
 * (or until a later call on this context has recycled the slot); results are written by the host
 * thread inside those calls. */
int kzg_commit_device_async(kzg_ctx* ctx, const kzg_srs* srs, const void* d_scalars, const size_t* lens,
                            size_t n_polys, size_t stride, uint64_t* out_xy, uint8_t* out_inf);
int kzg_commit_flush(kzg_ctx* ctx);

/* ---- KZG.open (kzg.py:122-159) -----------------------------------------------------------
 * combined = sum_i xi^(i+1) * polys[i]  (first polynomial scaled by xi, kzg.py:148-150);
 * witness = (combined - combined(z)) // (X - z); the proof is commit(witness).
 * eval_out (optional, 4 limbs) receives combined(z). */
int kzg_open(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t* polys, const size_t* lens, size_t k, size_t stride,
             const uint64_t z[4], const uint64_t xi[4], uint64_t* out_xy, uint8_t* out_inf, uint64_t* eval_out);
int kzg_open_device(kzg_ctx* ctx, const kzg_srs* srs, const void* d_polys, const size_t* lens, size_t k,
                    size_t stride, const uint64_t z[4], const uint64_t xi[4], uint64_t* out_xy, uint8_t* out_inf,
                    uint64_t* eval_out);

/* Pipelined form of kzg_open_device: the combine / evaluate / divide kernels and the MSM of the witness are only
 * enqueued (the MSM shares the commit pipeline's slots with kzg_commit_device_async); out_xy, out_inf and eval_out
 * (required, 4 limbs) are written when kzg_commit_flush() returns or a later call recycles the slot.  The input
 * polynomials may be overwritten by later work on the context's stream as soon as the call has returned. */
int kzg_open_device_async(kzg_ctx* ctx, const kzg_srs* srs, const void* d_polys, const size_t* lens, size_t k,
                          size_t stride, const uint64_t z[4], const uint64_t xi[4], uint64_t* out_xy, uint8_t* out_inf,
                          uint64_t* eval_out);

/* ---- KZG.open on ONE polynomial set partitioned by coefficient range across GPUs ----------------
 * Rank g holds coefficients [lo_g, hi_g) of every polynomial (the same ranges for all) and a key
 * shard.  kzg_open_shard_begin combines the slices (sum xi^(i+1) p_i) and returns the slice
 * polynomial's value H_g = sum_j c_(lo_g+j) z^j.  The ranks exchange the H_g (one field element
 * each); rank g's carry is S_(hi_g) = sum_(g' > g) H_g' * z^(lo_g' - hi_g).  kzg_open_shard_finish
 * lets the carry enter the scan above the slice's top coefficient and commits the quotient slice
 * (it continues from what _begin left on the device: no other open on this context in between):
 *   first_rank != 0: coefficients S_1 .. S_(hi-1) against key points 0 ..      (eval_out = P(z))
 *   otherwise      : coefficients S_lo .. S_(hi-1) against key points lo-1 ..   -- the caller's shard
 *                    must therefore START at global index lo_g - 1 (kzg_srs_generate_range).
 * The partial points of all ranks add up to the opening proof. */
int kzg_open_shard_begin(kzg_ctx* ctx, const void* d_polys, const size_t* lens, size_t k, size_t stride,
                         const uint64_t z[4], const uint64_t xi[4], uint64_t* chunk_eval_out);
int kzg_open_shard_finish(kzg_ctx* ctx, const kzg_srs* srs, const uint64_t z[4], const uint64_t carry[4],
                          int first_rank, uint64_t* out_xy, uint8_t* out_inf, uint64_t* eval_out);

/* Sum of n affine G1 points, on the HOST (no context, no GPU): what every rank does with the partial commitments /
 * partial opening proofs the others computed over their coefficient ranges -- the group law is not an RCCL reduction
 * operator, so the partial points are all-gathered as 97-byte records and added here (one inversion per sum).
 * xy: n x 2*FP_LIMBS canonical limbs; inf: n flags or NULL; KZG_ERR_ARG for a coordinate >= p or a point off the
 * curve.  Replaces the running kzg.add of kzg.py:116 over the ranks' results. */
int kzg_g1_sum(int curve_id, const uint64_t* xy, const uint8_t* inf, size_t n, uint64_t* out_xy, uint8_t* out_inf);

/* ---- device vector / polynomial primitives over Fr ------------------------------------------------
 * What the reference's callers do with Sage's dense polynomials between the transforms and the
 * commitments (plonk/prover.py:243-316: accumulator ratios, products, division by Z_H on a coset),
 * as passes over device-resident vectors of canonical 32-byte elements.  All enqueue on the context's
 * stream; out may alias an input for the element-wise ones.
 *   vec_op          out[i] = a[i] (+ | - | *) b[i]            op: 0 add, 1 sub, 2 mul
 *   vec_lincomb     out[i] = sum_j scalars[j] * p_j[i]         (p_j of lens[j] < n read as zero-padded)
 *   vec_mul_powers  out[i] = a[i] * c0 * s^i                   (coset shift of a coefficient vector)
 *   vec_inverse     out[i] = a[i]^-1, 0 -> 0
 *   vec_prefix_product  out[i] = prod_(j<i) a[j]               (exclusive; out[0] = 1)
 *   poly_eval       out = sum_i a[i] z^i                       (synchronises) */
int kzg_fr_vec_op(kzg_ctx* ctx, int op, size_t n, const void* d_a, const void* d_b, void* d_out);
int kzg_fr_vec_lincomb(kzg_ctx* ctx, size_t n, size_t k, const void* const* d_ptrs, const size_t* lens,
                       const uint64_t* scalars, void* d_out);
int kzg_fr_vec_mul_powers(kzg_ctx* ctx, size_t n, const void* d_a, const uint64_t s[4], const uint64_t c0[4],
                          void* d_out);
int kzg_fr_vec_inverse(kzg_ctx* ctx, size_t n, const void* d_a, void* d_out);
int kzg_fr_vec_prefix_product(kzg_ctx* ctx, size_t n, const void* d_a, void* d_out);
int kzg_fr_poly_eval(kzg_ctx* ctx, size_t n, const void* d_a, const uint64_t z[4], uint64_t out[4]);

/* ---- measurement hooks (bench.py) -----------------------------------------------------------
 * When enabled, the library brackets its kernels with HIP events on the stream each one runs on.
 * Span names: "ntt_pass", "msm_partition1", "msm_partition2", "msm_order", "msm_accumulate",
 * "msm_finalize", "msm_reduce", "open_poly" (ONE span per kzg_open*: combination, scan and division),
 * "open_shard_poly" (one per kzg_open_shard_begin and one per _finish).  kzg_prof_read synchronises the
 * stream and returns the accumulated milliseconds and span count of one name since the last kzg_prof_reset.
 * Two names are not spans: "msm_accumulate_shader_mhz" and "ntt_pass_shader_mhz" return (in *total_ms) the shader
 * clock in MHz the accumulate / NTT kernel ran at since the last reset -- s_memtime over s_memrealtime ticks of its
 * first wave -- and *count = 1 when a launch has reported, 0 otherwise; "ntt_tile_log" returns log2 of the LDS
 * tile the last two-pass transform took. */
int kzg_prof_enable(kzg_ctx* ctx, int on);
int kzg_prof_reset(kzg_ctx* ctx);
int kzg_prof_read(kzg_ctx* ctx, const char* name, double* total_ms, uint64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* KZG_MI355X_H */
