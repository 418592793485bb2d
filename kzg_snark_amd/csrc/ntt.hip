// ntt.hip -- radix-2 NTT / INTT over the scalar field Fr on gfx950.
//
// Replaces the recursion of the reference's fft_ff (fft_ff.py:3-37) and the
// w^-1 / n^-1 wrapper ifft_ff (fft_ff.py:39-58).  Semantics are those of the
// recursion, for ANY w (primitive or not): at the level that merges two halves
// of length m/2 the butterfly i uses the twiddle (w^(n/m))^i,
//     out[i] = E[i] + tw*O[i],   out[i + m/2] = E[i] - tw*O[i]     (fft_ff.py:32-35)
// which the iterative decimation-in-time form (bit-reversed placement, stages
// m = 2, 4, .., n) reproduces operation for operation.
//
// Decomposition (n = N1*N2, x viewed as an N1 x N2 row-major matrix):
//   pass 1: N1-point transforms down the columns (root w^N2), then element
//           (t, v) is multiplied by w^(t*v) (product of two small-table entries);
//   pass 2: N2-point transforms along the rows (root w^N1); row t, output
//           index b lands at out[b*N1 + t].
// The twist w^(t*v) is an exact field identity for every w (the t-dependent
// factor of every later-stage twiddle only depends on the input position), so
// the result equals the recursion's for non-primitive w as well.
//
// Each workgroup owns one LDS tile of up to 4096 elements (C lines of 2^k
// elements; 36 B per element: 9 limbs of 29 bits, stride 9 words => conflict-free
// ds_read_b32/ds_write_b32 across consecutive lanes).  The k levels run out of LDS
// as fused radix-4 steps (two levels per LDS round trip and barrier; one thread =
// one 4-point butterfly; a leading radix-2 step when k is odd); HBM is touched once
// for the load and once for the store of each pass.  Data stays in standard
// (non-Montgomery) form: twiddles are kept in Montgomery form, so
// mont_mul(x, tw*R) = x*tw.
//
// Arithmetic inside a pass is lazy: a level maps (x, y) to (x + t, x - t + 4p) with
// t = y*tw < 2p, limbs are only carry-normalised, values grow by at most 4p per
// level (<= 49p after 12 levels; the 9x29-bit representation holds 70p for
// BLS12-381 Fr and mul() accepts any operand below R = 2^261).  The multiplication
// that ends every pass -- the twist w^(t*v) after pass 1, the scale (1 or n^-1) after
// the last pass -- brings the value back below 2p, one conditional subtraction makes
// it canonical.
#include "internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace kzg {

namespace {

constexpr int TILE_LOG = 12;            // largest LDS tile: 4096 elements * 36 B = 144 KiB (one workgroup per CU)
constexpr int TILE_LOG_2WG = 11;        // preferred tile: 2048 elements = 72 KiB, TWO workgroups per CU, so one
                                        // workgroup's HBM load/store phases overlap the other's butterfly levels
                                        // (SQ counters of the one-per-CU version: 44 % of wave cycles parked)
constexpr int FRN = 9;                  // limbs of both scalar fields

struct NttPassArgs {
  const uint32_t* src;
  uint32_t* dst;
  const uint32_t* stage;   // stage twiddles (Montgomery), root order 2^kmax
  const uint32_t* twist;   // pass 1: table of w^(pos*col), [line length][row pitch] entries, or nullptr
  const uint32_t* twA;     // default: w^(pos*col) = twA[e >> h] * twB[e & (2^h - 1)], e = pos*col, from the two small
  const uint32_t* twB;     //   factor tables (build_domain); `twist` is only set under KZG_NTT_TWIST_TABLE=1
  const uint32_t* scale;   // last pass: final multiplier (Montgomery 1 or n^-1), or nullptr
  uint64_t twist_pitch;    // entries per twist-table row (= N2)
  uint32_t k;              // log2 line length
  uint32_t logC;           // log2 lines per tile
  uint32_t kmax;
  uint32_t h;              // twist split
  uint32_t c_fast_load;    // consecutive threads walk lines first when loading
  uint32_t c_fast_store;   // ... and when storing
  uint64_t col_base;       // global index of this matrix's first column (distributed column pass)
  uint64_t ld_line, ld_pos, tile_ld;
  uint64_t st_line, st_pos, tile_st;
  // Exchange layouts of the distributed transform: element `pos` of a line lives in block
  // pos >> shift (blocks `*_hi` elements apart), at pos & mask inside it.  Plain layouts: shift = 31.
  uint64_t ld_hi, st_hi;
  uint32_t ld_shift, st_shift;
  uint64_t batch_stride;   // elements between transforms of a batch (blockIdx.y)
  uint32_t pair_tiles;     // g > 0: remap blockIdx so that each group of 2^g adjacent tiles lands on one XCD
  unsigned long long* clk; // profiling only (null otherwise): [0] += shader-clock ticks, [1] += 100 MHz ticks of one wave
};

// LDS word address of element `pos` of line `line`.  Elements are 9 words; one
// pad word per 32 elements makes the bit-reversed scatter of the load phase
// (positions c + u*LEN/32) hit 32 different banks, and a one-word skew per line
// separates the lines of a tile (whose 9*LEN-word stride is a multiple of 32).
__device__ __forceinline__ uint32_t lds_addr(uint32_t line, uint32_t k, uint32_t pos) {
  const uint32_t e = (line << k) + pos;
  return e * FRN + (e >> 5) + line;
}
template <class F>
__device__ __forceinline__ Fe<F> lds_get(const uint32_t* lds, uint32_t addr) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < F::N; ++j) r.l[j] = lds[addr + j];
  return r;
}
template <class F>
__device__ __forceinline__ void lds_put(uint32_t* lds, uint32_t addr, const Fe<F>& v) {
#pragma unroll
  for (int j = 0; j < F::N; ++j) lds[addr + j] = v.l[j];
}
template <class F>
__device__ __forceinline__ Fe<F> glb_get_limbs(const uint32_t* p) {
  Fe<F> r;
#pragma unroll
  for (int j = 0; j < F::N; ++j) r.l[j] = p[j];
  return r;
}

// How a pass ends (compile-time, one instantiation each: no epilogue branches, no dead registers):
enum : int {
  EPI_REDUCE = 0,   // last pass of a forward / two-pass inverse transform: canonical value of the lazy sum
  EPI_FACTOR = 1,   // pass 1: twist w^(pos*col) = twA[e >> h] * twB[e & (2^h - 1)], applied as two multiplications
  EPI_TABLE = 2,    // pass 1 under KZG_NTT_TWIST_TABLE=1: one multiplication by the full-table entry
  EPI_SCALE = 3,    // single-pass inverse: n^-1 (fft_ff.py:57-58)
};

template <class F>
struct Tw3 {
  Fe<F> a, b, c;
};

// what a butterfly step writes back is carry-normalised (limbs < 2^L again; the value keeps growing lazily)
template <class F>
__device__ __forceinline__ void put_out(uint32_t* lds, uint32_t addr, const Fe<F>& v) {
  lds_put<F>(lds, addr, Field<F>::carry(v));
}

// Fused levels (s, s+1): butterfly `rem` of line `bline` works on the positions q0 + {0, h, 2h, 3h}, h = 2^(s-1).
template <class F>
__device__ __forceinline__ void radix4_step(uint32_t* lds, const Tw3<F>& tw, uint32_t bline, uint32_t k, uint32_t rem,
                                            uint32_t s) {
  using Fd = Field<F>;
  const uint32_t h = 1u << (s - 1);
  const uint32_t i = rem & (h - 1), g = rem >> (s - 1);
  const uint32_t q0 = (g << (s + 1)) + i;
  const uint32_t a0_ = lds_addr(bline, k, q0), a1_ = lds_addr(bline, k, q0 + h);
  const uint32_t a2_ = lds_addr(bline, k, q0 + 2 * h), a3_ = lds_addr(bline, k, q0 + 3 * h);
  const Fe<F> x0 = lds_get<F>(lds, a0_), x2 = lds_get<F>(lds, a2_);
  const Fe<F> t1 = Fd::mul(lds_get<F>(lds, a1_), tw.a), t3 = Fd::mul(lds_get<F>(lds, a3_), tw.a);
  const Fe<F> b0 = Fd::add_lazy(x0, t1), b1 = Fd::sub_lazy4(x0, t1);                               // level s
  const Fe<F> u2 = Fd::mul(Fd::add_lazy(x2, t3), tw.b), u3 = Fd::mul(Fd::sub_lazy4(x2, t3), tw.c);
  put_out<F>(lds, a0_, Fd::add_lazy(b0, u2));                                                // level s+1
  put_out<F>(lds, a2_, Fd::sub_lazy4(b0, u2));
  put_out<F>(lds, a1_, Fd::add_lazy(b1, u3));
  put_out<F>(lds, a3_, Fd::sub_lazy4(b1, u3));
}

// Levels 1 and 2: the only twiddle that is not 1 is w^(n/4) = stage[2^(kmax-2)], the same for every butterfly.
template <class F>
__device__ __forceinline__ void first_step(uint32_t* lds, const NttPassArgs& a, uint32_t bline, uint32_t k, uint32_t rem) {
  using Fd = Field<F>;
  const uint32_t q0 = rem << 2;
  const uint32_t a0_ = lds_addr(bline, k, q0), a1_ = lds_addr(bline, k, q0 + 1);
  const uint32_t a2_ = lds_addr(bline, k, q0 + 2), a3_ = lds_addr(bline, k, q0 + 3);
  const Fe<F> x0 = lds_get<F>(lds, a0_), x2 = lds_get<F>(lds, a2_);
  const Fe<F> t1 = lds_get<F>(lds, a1_), t3 = lds_get<F>(lds, a3_);
  const Fe<F> b0 = Fd::add_lazy(x0, t1), b1 = Fd::sub_lazy4(x0, t1);
  const Fe<F> u2 = Fd::carry(Fd::add_lazy(x2, t3));             // twiddle 1: only normalise for sub_lazy4
  const Fe<F> u3 = Fd::mul(Fd::sub_lazy4(x2, t3), glb_get_limbs<F>(a.stage + (size_t)(1u << (a.kmax - 2)) * F::N));
  put_out<F>(lds, a0_, Fd::add_lazy(b0, u2));
  put_out<F>(lds, a2_, Fd::sub_lazy4(b0, u2));
  put_out<F>(lds, a1_, Fd::add_lazy(b1, u3));
  put_out<F>(lds, a3_, Fd::sub_lazy4(b1, u3));
}

template <class F, bool PLAIN, int EPI>
__global__ __launch_bounds__(1024) void ntt_pass_kernel(NttPassArgs a) {
  using Fd = Field<F>;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
#ifndef KZG_NTT_PRIO
#define KZG_NTT_PRIO 3
#endif
  // like prep and the reduce stage: a short kernel that wins instruction issue beside the long accumulate kernel
  // (alone it changes nothing; in the commit pipeline at B = 4: 416-417 vs 409-410 commits/s on one box, r02c)
  __builtin_amdgcn_s_setprio(KZG_NTT_PRIO);
  const uint32_t T = blockDim.x, tid = threadIdx.x;
  // profiling only: wave 0 of workgroup (0, 0) reports the shader clock it ran at -- s_memtime (shader clock) against
  // s_memrealtime (100 MHz) over its lifetime, like the accumulate kernel does (the clock is power-managed)
  const bool probe = a.clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid < 64;
  const long long pc0 = probe ? clock64() : 0, pw0 = probe ? wall_clock64() : 0;
  const uint32_t k = a.k, logC = a.logC;
  const uint32_t LEN = 1u << k, C = 1u << logC, TILE = LEN << logC;
  // Adjacent tiles share 128-byte lines when a tile holds fewer than 4 columns; workgroups b and
  // b + 8 run on the same XCD (round-robin dispatch), so give them adjacent tiles (speed only).
  uint64_t tile = blockIdx.x;
  if (a.pair_tiles) {          // groups of 2^g tiles (g = pair_tiles) on one XCD
    const uint32_t g = a.pair_tiles, span = 3 + g;
    const uint32_t q = blockIdx.x >> span, rr = blockIdx.x & ((1u << span) - 1);
    tile = ((uint64_t)q << span) + ((rr & 7) << g) + (rr >> 3);
  }
  const uint32_t* src = a.src + (a.batch_stride * blockIdx.y + tile * a.tile_ld) * 8;
  uint32_t* dst = a.dst + (a.batch_stride * blockIdx.y + tile * a.tile_st) * 8;
  // element offset of (line, pos); the exchange layouts of the distributed transform (PLAIN = false) cut a
  // line into blocks of 2^shift elements that lie `hi` elements apart
  auto ld_off = [&](uint32_t line, uint32_t pos) -> uint64_t {
    if constexpr (PLAIN) return line * a.ld_line + pos * a.ld_pos;
    else return line * a.ld_line + (pos & ((1u << a.ld_shift) - 1u)) * a.ld_pos + (pos >> a.ld_shift) * a.ld_hi;
  };
  auto st_off = [&](uint32_t line, uint32_t pos) -> uint64_t {
    if constexpr (PLAIN) return line * a.st_line + pos * a.st_pos;
    else return line * a.st_line + (pos & ((1u << a.st_shift) - 1u)) * a.st_pos + (pos >> a.st_shift) * a.st_hi;
  };

  // ---- load: canonical words -> limbs, bit-reversed placement inside each line
  for (uint32_t idx = tid; idx < TILE; idx += T) {
    uint32_t line, pos;
    if (a.c_fast_load) { line = idx & (C - 1); pos = idx >> logC; }
    else               { pos = idx & (LEN - 1); line = idx >> k; }
    const uint4* g = reinterpret_cast<const uint4*>(src + ld_off(line, pos) * 8);
    const uint4 lo = g[0], hi = g[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const uint32_t rpos = k ? (__brev(pos) >> (32 - k)) : 0u;
    lds_put<F>(lds, lds_addr(line, k, rpos), Fd::from_words(w));
  }

  // Butterfly bf of a fused step (levels s, s+1) works on the positions q0 + {0, h, 2h, 3h} of its line, h = 2^(s-1),
  // with the twiddles (w^(n/2^s))^i = stage[i << (kmax - s)] of level s and stage[{i, i + h} << (kmax - s - 1)] of
  // level s+1.  Every launch gives a tile at least TILE/4 threads: one butterfly per thread and step.
  const uint32_t bf = tid;
  const bool active = bf < TILE / 4;
  const uint32_t bline = k >= 2 ? bf >> (k - 2) : 0u, rem = bf & (LEN / 4 - 1);   // rem is in range for every thread
  auto load_tw = [&](uint32_t s) {
    const uint32_t i = rem & ((1u << (s - 1)) - 1);
    Tw3<F> t;
    t.a = glb_get_limbs<F>(a.stage + (size_t)(i << (a.kmax - s)) * F::N);
    t.b = glb_get_limbs<F>(a.stage + (size_t)(i << (a.kmax - s - 1)) * F::N);
    t.c = glb_get_limbs<F>(a.stage + (size_t)((i + (1u << (s - 1))) << (a.kmax - s - 1)) * F::N);
    return t;
  };
  __syncthreads();

  // ---- levels 1..k in LDS.  (Leaving the LAST step's outputs un-carried for a multiplying epilogue -- mul<5> takes
  // such limbs -- saves 27 instructions per element and pass, but hipcc then keeps every partial sum of the step's
  // products alive for the pinned chains of field.h and spills 1 KB per lane; measured in round 3, not kept.)
  uint32_t s = 1;
  if (k & 1u) {   // odd k: one radix-2 level first (level 1: every twiddle is w^0 = 1)
    for (uint32_t b2 = tid; b2 < TILE / 2; b2 += T) {
      const uint32_t line = b2 >> (k - 1);
      const uint32_t blk = b2 & (LEN / 2 - 1);
      const uint32_t p0 = lds_addr(line, k, blk << 1), p1 = lds_addr(line, k, (blk << 1) + 1);
      const Fe<F> x = lds_get<F>(lds, p0), y = lds_get<F>(lds, p1);
      lds_put<F>(lds, p0, Fd::carry(Fd::add_lazy(x, y)));
      lds_put<F>(lds, p1, Fd::carry(Fd::sub_lazy4(x, y)));
    }
    __syncthreads();
    s = 2;
  }
  if (k >= 2) {
    Tw3<F> tw;
    if (s == 1) {   // levels 1 and 2: the only twiddle that is not 1 is w^(n/4), the same for every butterfly
      if (k > 2) tw = load_tw(3);   // in flight across this step's arithmetic and barrier
      if (active) first_step<F>(lds, a, bline, k, rem);
      __syncthreads();
      s = 3;
    } else {
      tw = load_tw(2);
    }
    for (; s + 2 < k + 1; s += 2) {   // every step but the last
      const Tw3<F> cur = tw;
      tw = load_tw(s + 2);
      if (active) radix4_step<F>(lds, cur, bline, k, rem, s);
      __syncthreads();
    }
    if (s < k + 1) {
      if (active) radix4_step<F>(lds, tw, bline, k, rem, s);
      __syncthreads();
    }
  }

  // ---- store: the twist multiplication (or n^-1) reduces the lazy value below 2p; a factor of 1 is a plain reduction
  for (uint32_t idx = tid; idx < TILE; idx += T) {
    uint32_t line, pos;
    if (a.c_fast_store) { line = idx & (C - 1); pos = idx >> logC; }
    else                { pos = idx & (LEN - 1); line = idx >> k; }
    Fe<F> x = lds_get<F>(lds, lds_addr(line, k, pos));
    if constexpr (EPI == EPI_TABLE) {
      const uint64_t col = a.col_base + tile * C + line;
      // weak-normal (< 2p < 2^256) is enough between the passes: the next pass starts its lazy
      // levels from it (2p + 4p per level stays below the 64p that reduce_wide accepts)
      x = Fd::mul(x, glb_get_limbs<F>(a.twist + ((size_t)pos * a.twist_pitch + col) * F::N));
    } else if constexpr (EPI == EPI_FACTOR) {
      // x * twA[hi] * twB[lo] as two multiplications (both factors canonical, so each product is back below 2p):
      // forming the twiddle first costs the same two products plus a reduction of it
      const uint64_t e = (uint64_t)pos * (a.col_base + tile * C + line);
      x = Fd::mul(x, glb_get_limbs<F>(a.twA + (size_t)(e >> a.h) * F::N));
      x = Fd::mul(x, glb_get_limbs<F>(a.twB + (size_t)(e & ((1ull << a.h) - 1)) * F::N));
    } else if constexpr (EPI == EPI_SCALE) {
      x = Fd::reduce(Fd::mul(x, glb_get_limbs<F>(a.scale)));   // single-pass inverse: n^-1
    } else {
      x = Fd::reduce_wide(x);                                     // factor 1: reduce the lazy sum directly
    }
    uint32_t w[8];
    Fd::to_words(x, w);
    uint4* g = reinterpret_cast<uint4*>(dst + st_off(line, pos) * 8);
    g[0] = make_uint4(w[0], w[1], w[2], w[3]);
    g[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
  if (probe && tid == 0) {
    atomicAdd(a.clk, (unsigned long long)(clock64() - pc0));
    atomicAdd(a.clk + 1, (unsigned long long)(wall_clock64() - pw0));
  }
}

// one launch: picks the instantiation for the pass's layout and epilogue
template <class F>
int launch_pass(Ctx* c, const NttPassArgs& a, uint64_t tiles, uint32_t batch, uint32_t threads) {
  const uint32_t tile_elems = 1u << (a.k + a.logC);
  if (threads < tile_elems / 4 || threads > 1024) return set_err(c, KZG_ERR_ARG, "ntt: a tile needs one thread per butterfly");
  const size_t lds_bytes = ((size_t)tile_elems * F::N + tile_elems / 32 + (1u << a.logC) + 1) * 4;
  const bool plain = a.ld_shift == 31 && a.st_shift == 31;
  const int epi = a.twist ? EPI_TABLE : a.twA ? EPI_FACTOR : a.scale ? EPI_SCALE : EPI_REDUCE;
  void (*kern)(NttPassArgs) = nullptr;
#define KZG_NTT_PICK(P, E) if (plain == P && epi == E) kern = ntt_pass_kernel<F, P, E>;
  KZG_NTT_PICK(true, EPI_REDUCE) KZG_NTT_PICK(true, EPI_FACTOR) KZG_NTT_PICK(true, EPI_TABLE) KZG_NTT_PICK(true, EPI_SCALE)
  KZG_NTT_PICK(false, EPI_REDUCE) KZG_NTT_PICK(false, EPI_FACTOR) KZG_NTT_PICK(false, EPI_TABLE) KZG_NTT_PICK(false, EPI_SCALE)
#undef KZG_NTT_PICK
  const uint32_t slot = (plain ? 0u : 4u) + (uint32_t)epi;
  if (!(c->ntt_lds_attr_set & (1u << slot))) {   // allow > 64 KiB of dynamic LDS (gfx950: 160 KiB per CU); per device, so per context
    KZG_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
    c->ntt_lds_attr_set |= (1u << slot);
  }
  ProfScope ps(c, "ntt_pass");
  NttPassArgs launch_args = a;
  launch_args.clk = (c->prof_on && c->clk_probe) ? c->clk_probe + 2 : nullptr;
  hipLaunchKernelGGL(kern, dim3((uint32_t)tiles, batch), dim3(threads), lds_bytes, c->stream, launch_args);
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

// out[e] = mult * base^e  (all Montgomery form), e < count
template <class F>
__global__ void pow_table_kernel(uint32_t* out, const uint32_t* base_mult, uint32_t count) {
  using Fd = Field<F>;
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  Fe<F> b = glb_get_limbs<F>(base_mult);
  Fe<F> acc = glb_get_limbs<F>(base_mult + F::N);
  for (uint32_t bits = e; bits; bits >>= 1) {
    if (bits & 1u) acc = Fd::mul(acc, b);
    b = Fd::mul(b, b);
  }
  acc = Fd::reduce(acc);   // canonical: the lazy butterflies rely on twiddles < p
#pragma unroll
  for (int j = 0; j < F::N; ++j) out[(size_t)e * F::N + j] = acc.l[j];
}

// twist[t * N2 + v] = A[e >> h] * B[e & (2^h - 1)], e = t*v  (w^(t*v) in Montgomery form)
template <class F>
__global__ void twist_table_kernel(uint32_t* out, const uint32_t* twA, const uint32_t* twB, uint32_t h,
                                   uint32_t log_n2, uint64_t n) {
  using Fd = Field<F>;
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint64_t t = idx >> log_n2, v = idx & ((1ull << log_n2) - 1);
  const uint64_t e = t * v;
  const Fe<F> r = Fd::reduce(Fd::mul(glb_get_limbs<F>(twA + (size_t)(e >> h) * F::N),
                                     glb_get_limbs<F>(twB + (size_t)(e & ((1ull << h) - 1)) * F::N)));
#pragma unroll
  for (int j = 0; j < F::N; ++j) out[(size_t)idx * F::N + j] = r.l[j];
}

template <class F>
int build_domain(Ctx* c, NttDomain& d) {
  using Fd = Field<F>;
  const uint32_t log_n = d.log_n;
  Fe<F> w = Fd::to_mont(Fd::from_words(d.w));
  if (d.inverse) w = Fd::inv(w);                       // fft_ff.py:53
  // n^-1 (fft_ff.py:57): n as a field element, inverted
  uint32_t nw[8] = {0};
  nw[log_n >> 5] = 1u << (log_n & 31);
  const Fe<F> ninv = Fd::inv(Fd::to_mont(Fd::from_words(nw)));

  const bool two_pass = log_n > (uint32_t)TILE_LOG;
  d.kmax = two_pass ? (log_n + 1) / 2 : log_n;
  d.h = (log_n + 1) / 2;
  auto pow2k = [&](Fe<F> x, uint32_t sq) { for (uint32_t i = 0; i < sq; ++i) x = Fd::mul(x, x); return x; };

  struct Job { uint32_t** dst; Fe<F> base; Fe<F> mult; uint32_t count; };
  std::vector<Job> jobs;
  if (d.kmax >= 1) jobs.push_back({&d.d_stage, pow2k(w, log_n - d.kmax), Fd::one(), 1u << (d.kmax - 1)});
  if (two_pass) {
    // the n^-1 of an inverse transform (fft_ff.py:57-58) rides on the twist: twist = n^-1 * w^(t*v)
    jobs.push_back({&d.d_twA, pow2k(w, d.h), d.inverse ? ninv : Fd::one(), 1u << (log_n - d.h)});
    jobs.push_back({&d.d_twB, w, Fd::one(), 1u << d.h});
  }
  uint32_t* d_tmp = nullptr;
  KZG_HIP(c, hipMalloc(&d_tmp, 2 * F::N * 4));
  for (auto& j : jobs) {
    KZG_HIP(c, hipMalloc(j.dst, (size_t)j.count * F::N * 4));
    uint32_t hb[2 * F::N];
    memcpy(hb, j.base.l, F::N * 4);
    memcpy(hb + F::N, j.mult.l, F::N * 4);
    KZG_HIP(c, hipMemcpyAsync(d_tmp, hb, sizeof(hb), hipMemcpyHostToDevice, c->stream));
    KZG_HIP(c, hipStreamSynchronize(c->stream));   // hb is a stack buffer
    hipLaunchKernelGGL(pow_table_kernel<F>, dim3((j.count + 255) / 256), dim3(256), 0, c->stream, *j.dst,
                       d_tmp, j.count);
    KZG_HIP(c, hipGetLastError());
    KZG_HIP(c, hipStreamSynchronize(c->stream));
  }
  KZG_HIP(c, hipFree(d_tmp));
  // The twist w^(t*v) comes from the two factor tables (e = t*v = hi * 2^h + lo: twA[hi] * twB[lo], 2 x 2^10 entries at
  // 2^20) -- one more multiplication per element than a full [N1][N2] table, but measured at the same time (112-115 us
  // either way at 2^20: the table's 36 MiB of reads per transform stalled as much as the multiplication costs) with
  // half the reads (67.9 vs 135.8 MB per transform, rocprofv3 FETCH_SIZE) and no 36 MiB (2^24: 576 MiB) per cached
  // domain.  KZG_NTT_TWIST_TABLE=1 brings the full table back (A/B).
  static const bool twist_table = [] { const char* e = getenv("KZG_NTT_TWIST_TABLE"); return e && atoi(e) != 0; }();
  if (two_pass && twist_table) {   // full twist table (one multiplication per element between the passes)
    const uint32_t k2 = log_n - (log_n + 1) / 2;
    const uint64_t n = 1ull << log_n;
    KZG_HIP(c, hipMalloc(&d.d_twist, (size_t)n * F::N * 4));
    hipLaunchKernelGGL(twist_table_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d.d_twist,
                       d.d_twA, d.d_twB, d.h, k2, n);
    KZG_HIP(c, hipGetLastError());
    KZG_HIP(c, hipStreamSynchronize(c->stream));
    KZG_HIP(c, hipFree(d.d_twA));
    KZG_HIP(c, hipFree(d.d_twB));
    d.d_twA = d.d_twB = nullptr;
  }
  // Last pass: a forward transform, and an inverse one whose n^-1 is already in the twist, end with the
  // plain reduction of the lazy value (d_scale stays null); only the single-pass inverse multiplies by
  // n^-1 (fft_ff.py:57-58).
  d.d_scale = nullptr;
  if (d.inverse && !two_pass) {
    const Fe<F> last = Fd::reduce(ninv);
    KZG_HIP(c, hipMalloc(&d.d_scale, F::N * 4));
    KZG_HIP(c, hipMemcpy(d.d_scale, last.l, F::N * 4, hipMemcpyHostToDevice));
  }
  return KZG_OK;
}

// Preferred log2(tile elements).  Alone, 2048-element tiles (72 KiB, two 512-thread workgroups per CU) are fastest
// (TILE_LOG_2WG).  While an accumulate kernel of the commit pipeline is queued or running, its four persistent
// workgroups hold 114 of a CU's 160 KiB of LDS for ~2 ms: a 72 KiB tile then only gets CUs in that kernel's tail and the
// NEXT accumulate kernel starts short of workgroups, whereas a 1024-element tile (36 KiB, 256 threads, one wave of <= 122
// VGPRs per SIMD) fits beside them.  Round 3, same box, alternating, bench.py at B = 4: 464.6 / 460.9 commits/s with
// 1024-element tiles against 432.3 / 439.2 (the NTT alone: 100.4 vs 99.0 us, HBM traffic per transform unchanged at
// 2.04x algorithmic) -- so the tile follows what is resident.  kzg_ctx_set_tuning(ctx, "ntt_tile_log", 8 .. 12) or
// KZG_NTT_TILE_LOG fixes it (tests, experiments); kzg_prof_read(ctx, "ntt_tile_log") tells what the last transform took.
constexpr int TILE_LOG_BESIDE_MSM = 10;
static uint32_t tile_log_pref(Ctx* c) {
  static const int fixed = [] { const char* e = getenv("KZG_NTT_TILE_LOG"); return e ? atoi(e) : 0; }();
  const int forced = c->tune_ntt_tile_log ? c->tune_ntt_tile_log : fixed;     // kzg_ctx_set_tuning("ntt_tile_log") first
  const int x = forced ? forced : (msm_accumulate_in_flight(c) ? TILE_LOG_BESIDE_MSM : TILE_LOG_2WG);
  c->last_ntt_tile_log = std::min(std::max(x, 8), TILE_LOG);
  return (uint32_t)c->last_ntt_tile_log;
}

template <class F>
int launch_passes(Ctx* c, const NttDomain& d, uint32_t* d_data, uint32_t batch) {
  const uint32_t log_n = d.log_n;
  const uint64_t n = 1ull << log_n;
  auto launch = [&](const NttPassArgs& a, uint64_t tiles) -> int {
    const uint32_t tile_elems = 1u << (a.k + a.logC);
    return launch_pass<F>(c, a, tiles, batch, std::min<uint32_t>(1024u, std::max<uint32_t>(64u, tile_elems / 4)));
  };
  if (log_n <= (uint32_t)TILE_LOG) {
    NttPassArgs a{};
    a.ld_shift = a.st_shift = 31;
    a.src = d_data; a.dst = d_data; a.stage = d.d_stage; a.twist = nullptr; a.twist_pitch = 0;
    a.pair_tiles = 0;
    a.scale = d.d_scale; a.k = log_n; a.logC = 0; a.kmax = d.kmax; a.h = 0; a.c_fast_load = 0;
    a.c_fast_store = 1; a.col_base = 0;
    a.ld_line = 0; a.ld_pos = 1; a.tile_ld = 0; a.st_line = 0; a.st_pos = 1; a.tile_st = 0;
    a.batch_stride = n;
    return launch(a, 1);
  }
  const uint32_t k1 = (log_n + 1) / 2, k2 = log_n - k1;
  const uint64_t N1 = 1ull << k1, N2 = 1ull << k2;
  const uint32_t tile_pref = tile_log_pref(c);       // one choice for both passes of the transform
  int rc = ensure_buf(c, c->ntt_scratch, (size_t)n * batch * 32);
  if (rc) return rc;
  uint32_t* scratch = static_cast<uint32_t*>(c->ntt_scratch.p);
  {  // pass 1: columns of the N1 x N2 matrix, twist by w^(t*v)
    const uint32_t tl = std::max<uint32_t>(k1, tile_pref);
    const uint32_t logC = std::min<uint32_t>(tl - k1, k2);
    NttPassArgs a{};
    a.ld_shift = a.st_shift = 31;
    a.pair_tiles = (logC < 2 && ((N2 >> logC) % (8u << (2 - logC)) == 0)) ? 2 - logC : 0;
    a.src = d_data; a.dst = scratch; a.stage = d.d_stage; a.twist = d.d_twist; a.twist_pitch = N2;
    a.twA = d.d_twist ? nullptr : d.d_twA; a.twB = d.d_twB;
    a.scale = nullptr;
    a.k = k1; a.logC = logC; a.kmax = d.kmax; a.h = d.h; a.c_fast_load = 1; a.c_fast_store = 1; a.col_base = 0;
    a.ld_line = 1; a.ld_pos = N2; a.tile_ld = 1ull << logC;
    a.st_line = 1; a.st_pos = N2; a.tile_st = 1ull << logC;
    a.batch_stride = n;
    rc = launch(a, N2 >> logC);
    if (rc) return rc;
  }
  {  // pass 2: rows; row t, output index b -> out[b*N1 + t]
    const uint32_t tl = std::max<uint32_t>(k2, tile_pref);
    const uint32_t logC = std::min<uint32_t>(tl - k2, k1);
    NttPassArgs a{};
    a.ld_shift = a.st_shift = 31;
    a.pair_tiles = (logC < 2 && ((N1 >> logC) % (8u << (2 - logC)) == 0)) ? 2 - logC : 0;
    a.src = scratch; a.dst = d_data; a.stage = d.d_stage; a.twist = nullptr; a.twist_pitch = 0;
    a.scale = d.d_scale;
    a.k = k2; a.logC = logC; a.kmax = d.kmax; a.h = 0; a.c_fast_load = 0; a.c_fast_store = 1; a.col_base = 0;
    a.ld_line = N2; a.ld_pos = 1; a.tile_ld = N2 << logC;
    a.st_line = 1; a.st_pos = N1; a.tile_st = 1ull << logC;
    a.batch_stride = n;
    rc = launch(a, N1 >> logC);
    if (rc) return rc;
  }
  return KZG_OK;
}

template <class F>
int get_domain(Ctx* c, uint32_t log_n, const uint32_t* w_words, int inverse, NttDomain** out);

// Distributed four-step NTT, local halves (kzg_snark_amd/sharding.py moves the data between them):
//   columns: in-place N1-point transforms down the n_cols columns of an [N1][n_cols] matrix whose
//            first column is global column col_base, followed by the twist w^(t * global column)
//            (times n^-1 for the inverse transform);
//   rows:    in-place N2-point transforms along the n_rows rows of an [n_rows][N2] matrix, natural
//            order in and out, ending with the reduction of the lazily accumulated values.
//   rows (exchange form): src is what the columns -> rows all-to-all delivers, [world][n_rows][N2/world]
//            (block h = columns h*N2/world ..); dst is either the same blocked shape over the OUTPUT index
//            (ready for the all-to-all that restores natural order) or plain [n_rows][N2] rows.
template <class F>
int ntt_partial_t(Ctx* c, uint32_t* d_data, uint32_t log_n, const uint32_t* w_words, int inverse, int rows_pass,
                  uint64_t count, uint64_t col_base, uint32_t* d_dst = nullptr, uint32_t world = 1,
                  int blocked_out = 0) {
  if (log_n <= (uint32_t)TILE_LOG) return set_err(c, KZG_ERR_ARG, "distributed NTT needs log_n > 12");
  const uint32_t k1 = (log_n + 1) / 2, k2 = log_n - k1;
  const uint64_t N1 = 1ull << k1, N2 = 1ull << k2;
  if (count == 0 || (count & (count - 1))) return set_err(c, KZG_ERR_ARG, "row/column count must be a power of two");
  {
    const bool col_pass = rows_pass == 0 || rows_pass == 3;
    const uint64_t first = (rows_pass == 0 || rows_pass == 2) ? col_base : 0;       // global index of the first line
    if (first + count > (col_pass ? N2 : N1)) return set_err(c, KZG_ERR_ARG, "row/column range");
  }
  NttDomain* dom = nullptr;
  int rc = get_domain<F>(c, log_n, w_words, inverse, &dom);
  if (rc) return rc;
  uint32_t lc = 0;
  while ((1ull << (lc + 1)) <= count) ++lc;
  // rows_pass: 0 columns + twist, 1 rows, 2 rows + twist by (global row) x (output index), 3 columns without twist.
  // 2 then 3 is the transform taken the other way round -- input in the TRANSPOSED layout (row rho of the [N1][N2]
  // matrix holds the indices b N1 + rho), output in natural order: X[alpha N2 + beta] = sum_rho w^(rho alpha N2)
  // [ w^(rho beta) sum_b x[b N1 + rho] w^(b beta N1) ].  That identity needs w^n = 1 (the decimation-in-time form of
  // modes 0 / 1 holds for every w, this one for the roots of unity the provers pass), so mode 2 checks it.
  const bool cols = rows_pass == 0 || rows_pass == 3, twisted = rows_pass == 0 || rows_pass == 2;
  if (rows_pass == 2 && !dom->d_twA)
    return set_err(c, KZG_ERR_ARG, "kzg_ntt_rows_twist_device needs the twist factor tables (KZG_NTT_TWIST_TABLE is set)");
  if (rows_pass == 2) {
    using Fd = Field<F>;
    Fe<F> t = Fd::to_mont(Fd::from_words(w_words));
    for (uint32_t q = 0; q + 1 < log_n; ++q) t = Fd::mul(t, t);                 // w^(n/2)
    if (!Fd::is_zero(Fd::add(t, Fd::one())))
      return set_err(c, KZG_ERR_ARG, "kzg_ntt_rows_twist_device: w must be a primitive 2^log_n-th root of unity");
  }
  NttPassArgs a{};
  a.ld_shift = a.st_shift = 31;
  a.src = d_data; a.dst = d_dst ? d_dst : d_data; a.stage = dom->d_stage; a.kmax = dom->kmax; a.h = 0; a.batch_stride = 0;
  a.pair_tiles = 0;
  uint64_t tiles;
  if (cols) {
    const uint32_t logC = std::min<uint32_t>(TILE_LOG - k1, lc);
    a.twist = twisted ? dom->d_twist : nullptr; a.twist_pitch = N2; a.scale = nullptr; a.col_base = col_base;
    a.twA = (twisted && !dom->d_twist) ? dom->d_twA : nullptr; a.twB = dom->d_twB; a.h = dom->h;
    a.k = k1; a.logC = logC; a.c_fast_load = 1; a.c_fast_store = 1;
    a.ld_line = 1; a.ld_pos = count; a.tile_ld = 1ull << logC;
    a.st_line = 1; a.st_pos = count; a.tile_st = 1ull << logC;
    tiles = count >> logC;
  } else {
    const uint32_t logC = std::min<uint32_t>(TILE_LOG - k2, lc);
    a.twist = nullptr; a.twist_pitch = 0; a.scale = dom->d_scale; a.col_base = 0;
    if (twisted) { a.twA = dom->d_twA; a.twB = dom->d_twB; a.h = dom->h; a.col_base = col_base; a.scale = nullptr; }
    a.k = k2; a.logC = logC; a.c_fast_load = 0; a.c_fast_store = 0;
    a.ld_line = N2; a.ld_pos = 1; a.tile_ld = N2 << logC;
    a.st_line = N2; a.st_pos = 1; a.tile_st = N2 << logC;
    if (d_dst) {   // exchange form
      if (world == 0 || (world & (world - 1)) || world > N2) return set_err(c, KZG_ERR_ARG, "world must be a power of two <= N2");
      if (d_dst == d_data) return set_err(c, KZG_ERR_ARG, "rows exchange pass is out of place");
      uint32_t lw = 0;
      while ((N2 >> (lw + 1)) >= world) ++lw;           // W = N2 / world = 2^lw
      const uint64_t W = 1ull << lw;
      a.ld_line = W; a.tile_ld = W << logC; a.ld_shift = lw; a.ld_hi = count * W;
      if (blocked_out) { a.st_line = W; a.tile_st = W << logC; a.st_shift = lw; a.st_hi = count * W; }
    }
    tiles = count >> logC;
  }
  const uint32_t tile_elems = 1u << (a.k + a.logC);
  return launch_pass<F>(c, a, tiles, 1, std::min<uint32_t>(1024u, std::max<uint32_t>(64u, tile_elems / 2)));
}

template <class F>
int ntt_run_t(Ctx* c, uint32_t* d_data, uint32_t log_n, const uint32_t* w_words, int inverse, uint32_t batch) {
  if (log_n == 0 || batch == 0) return KZG_OK;   // n == 1: fft_ff.py:16-17 returns the input; 1^-1 = 1
  NttDomain* dom = nullptr;
  int rc = get_domain<F>(c, log_n, w_words, inverse, &dom);
  if (rc) return rc;
  return launch_passes<F>(c, *dom, d_data, batch);
}

template <class F>
int get_domain(Ctx* c, uint32_t log_n, const uint32_t* w_words, int inverse, NttDomain** out) {
  NttDomain* dom = nullptr;
  for (auto& d : c->domains)
    if (d.log_n == log_n && d.inverse == inverse && memcmp(d.w, w_words, 32) == 0) { dom = &d; break; }
  if (!dom) {
    if (c->domains.size() >= 8) {   // evict least recently used
      size_t victim = 0;
      for (size_t i = 1; i < c->domains.size(); ++i)
        if (c->domains[i].last_use < c->domains[victim].last_use) victim = i;
      NttDomain& v = c->domains[victim];
      hipStreamSynchronize(c->stream);
      hipFree(v.d_stage); hipFree(v.d_twA); hipFree(v.d_twB); hipFree(v.d_scale); hipFree(v.d_twist);
      c->domains.erase(c->domains.begin() + victim);
    }
    NttDomain d;
    d.log_n = log_n; d.inverse = inverse;
    memcpy(d.w, w_words, 32);
    int rc = build_domain<F>(c, d);
    if (rc) return rc;
    c->domains.push_back(d);
    dom = &c->domains.back();
  }
  dom->last_use = ++c->tick;
  *out = dom;
  return KZG_OK;
}


// ---- arbitrary (non-power-of-two) lengths: the reference recursion level by level ---------------
// fft_ff (fft_ff.py:15-37) never checks the length.  With n not a power of two its even/odd
// slices have lengths ceil(n/2) and floor(n/2), the loop runs n//2 butterflies, and -- for odd n --
// the last even value is dropped and result[n-1] stays F(0).  The output is not a transform of
// anything, but it is what the reference returns (marlin/prover.py:439-449 passes list(row_A),
// whose length is whatever survives Sage's dropping of trailing zeros), so the engine returns the
// same values.  Off the hot path: one thread per output element and level, no LDS staging.
//
// The node at depth d reached by the slice choices (b0, .., b_(d-1)) holds the inputs j = off + k*2^d,
// off = sum b_i 2^i, so it has L = ceil((n - off) / 2^d) elements; element k of its result is kept
// at position off + k*2^d of a length-n buffer (the union of its children's positions).  A node with
// L = 1 returns its input (fft_ff.py:16-17).  Depth d uses the root w^(2^d) (fft_ff.py:24).
template <class F>
__global__ void fft_ragged_level_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint64_t n,
                                        uint32_t d, const uint32_t* __restrict__ wd, const uint32_t* __restrict__ scale) {
  using Fd = Field<F>;
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const uint64_t step = 1ull << d, off = p & (step - 1), k = p >> d;
  const uint64_t len = (n - off + step - 1) >> d;
  auto load = [&](uint64_t pos) {
    const uint4* g = reinterpret_cast<const uint4*>(src + pos * 8);
    const uint4 lo = g[0], hi = g[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return Fd::from_words(w);
  };
  Fe<F> out;
  if (len == 1) {
    out = load(p);                                          // fft_ff.py:16-17
  } else {
    const uint64_t h = len >> 1;                            // n // 2 butterflies (fft_ff.py:32)
    if (k >= 2 * h) {
      out = Fd::zero();                                     // odd length: result[n-1] keeps F(0) (fft_ff.py:29)
    } else {
      const uint64_t i = k < h ? k : k - h;
      const Fe<F> e = load(off + ((2 * i) << d));           // even_fft[i]
      const Fe<F> o = load(off + ((2 * i + 1) << d));       // odd_fft[i]
      Fe<F> b = glb_get_limbs<F>(wd);                       // (w^(2^d))^i, Montgomery form
      Fe<F> tw = Fd::one();
      for (uint64_t bits = i; bits; bits >>= 1) {
        if (bits & 1u) tw = Fd::mul(tw, b);
        b = Fd::mul(b, b);
      }
      const Fe<F> t = Fd::mul(o, tw);                       // standard form * Montgomery form = standard form
      out = k < h ? Fd::add(e, t) : Fd::sub(e, t);          // fft_ff.py:33-34
    }
  }
  if (scale) out = Fd::mul(out, glb_get_limbs<F>(scale));   // ifft_ff: n^-1 (fft_ff.py:57-58)
  out = Fd::reduce(out);
  uint32_t w[8];
  Fd::to_words(out, w);
  uint4* g = reinterpret_cast<uint4*>(dst + p * 8);
  g[0] = make_uint4(w[0], w[1], w[2], w[3]);
  g[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <class F>
int fft_ragged_t(Ctx* c, uint32_t* d_data, uint64_t n, const uint32_t* w_words, int inverse) {
  using Fd = Field<F>;
  uint32_t depth = 0;
  while ((1ull << depth) < n) ++depth;
  Fe<F> w = Fd::to_mont(Fd::from_words(w_words));
  if (inverse) w = Fd::inv(w);                                                     // fft_ff.py:53
  std::vector<uint32_t> host((size_t)(depth + 1) * F::N);
  for (uint32_t d = 0; d < depth; ++d) {
    const Fe<F> wr = Fd::reduce(w);
    memcpy(&host[(size_t)d * F::N], wr.l, F::N * 4);
    w = Fd::mul(w, w);                                                             // fft_ff.py:24
  }
  if (inverse) {   // F(n)^-1 (fft_ff.py:57); n < 2^64 < r
    const uint32_t nw[8] = {(uint32_t)n, (uint32_t)(n >> 32), 0, 0, 0, 0, 0, 0};
    const Fe<F> ninv = Fd::reduce(Fd::inv(Fd::to_mont(Fd::from_words(nw))));
    memcpy(&host[(size_t)depth * F::N], ninv.l, F::N * 4);
  }
  int rc = ensure_buf(c, c->ntt_scratch, (size_t)n * 32 + host.size() * 4 + 64);
  if (rc) return rc;
  uint32_t* scratch = static_cast<uint32_t*>(c->ntt_scratch.p);
  uint32_t* d_tw = scratch + (((size_t)n * 8 + 15) & ~(size_t)15);
  KZG_HIP(c, hipMemcpyAsync(d_tw, host.data(), host.size() * 4, hipMemcpyHostToDevice, c->stream));
  KZG_HIP(c, hipStreamSynchronize(c->stream));   // `host` is a local buffer
  uint32_t* bufs[2] = {d_data, scratch};
  int cur = 0;
  for (int d = (int)depth - 1; d >= 0; --d) {
    const uint32_t* scale = (inverse && d == 0) ? d_tw + (size_t)depth * F::N : nullptr;
    hipLaunchKernelGGL(fft_ragged_level_kernel<F>, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream,
                       bufs[cur], bufs[cur ^ 1], n, (uint32_t)d, d_tw + (size_t)d * F::N, scale);
    KZG_HIP(c, hipGetLastError());
    cur ^= 1;
  }
  if (cur == 1) KZG_HIP(c, hipMemcpyAsync(d_data, scratch, (size_t)n * 32, hipMemcpyDeviceToDevice, c->stream));
  return KZG_OK;
}

}  // namespace

int fft_ragged_device(Ctx* c, uint32_t* d_data, uint64_t n, const uint32_t* w_words, int inverse) {
  if (n == 0) return set_err(c, KZG_ERR_ARG, "fft_ff of an empty list: the reference recursion never terminates");
  if (n > (1ull << 24)) return set_err(c, KZG_ERR_ARG, "kzg_fft_ff_any: n > 2^24 not supported");
  if (n == 1) return KZG_OK;                       // fft_ff.py:16-17; F(1)^-1 = 1
  return c->curve == 0 ? fft_ragged_t<BnFr>(c, d_data, n, w_words, inverse)
                       : fft_ragged_t<BlsFr>(c, d_data, n, w_words, inverse);
}

int ntt_run_device(Ctx* c, uint32_t* d_data, uint32_t log_n, const uint32_t* w_words, int inverse, uint32_t batch) {
  if (log_n > 24) return set_err(c, KZG_ERR_ARG, "kzg_ntt: log_n > 24 not supported");
  if (c->curve == 0) return ntt_run_t<BnFr>(c, d_data, log_n, w_words, inverse, batch);
  return ntt_run_t<BlsFr>(c, d_data, log_n, w_words, inverse, batch);
}

int ntt_partial_device(Ctx* c, uint32_t* d_data, uint32_t log_n, const uint32_t* w_words, int inverse, int rows_pass,
                       uint64_t count, uint64_t col_base) {
  if (log_n > 24) return set_err(c, KZG_ERR_ARG, "kzg_ntt: log_n > 24 not supported");
  return c->curve == 0 ? ntt_partial_t<BnFr>(c, d_data, log_n, w_words, inverse, rows_pass, count, col_base)
                       : ntt_partial_t<BlsFr>(c, d_data, log_n, w_words, inverse, rows_pass, count, col_base);
}

int ntt_rows_exchange_device(Ctx* c, const uint32_t* d_src, uint32_t* d_dst, uint32_t log_n, const uint32_t* w_words,
                             int inverse, uint64_t n_rows, uint32_t world, int blocked_out) {
  if (log_n > 24) return set_err(c, KZG_ERR_ARG, "kzg_ntt: log_n > 24 not supported");
  uint32_t* src = const_cast<uint32_t*>(d_src);
  return c->curve == 0
             ? ntt_partial_t<BnFr>(c, src, log_n, w_words, inverse, 1, n_rows, 0, d_dst, world, blocked_out)
             : ntt_partial_t<BlsFr>(c, src, log_n, w_words, inverse, 1, n_rows, 0, d_dst, world, blocked_out);
}

void ntt_free_domains(Ctx* c) {
  for (auto& d : c->domains) {
    hipFree(d.d_stage); hipFree(d.d_twA); hipFree(d.d_twB); hipFree(d.d_scale); hipFree(d.d_twist);
  }
  c->domains.clear();
}

}  // namespace kzg
