// msm_prep.hip -- stage P of the commit pipeline: scalars -> bucket-sorted table indices.
//
// Replaces, for the MSM of msm.hip, what the reference does implicitly by walking the
// coefficient list (kzg.py:112-116): every scalar is cut into W signed c-bit digits (zero digits
// are dropped: the zero skip of kzg.py:113-114) and the (bucket, table index) entries are grouped
// by bucket.  It is an MSD counting sort in two partition steps, written for running BESIDE the
// persistent accumulate kernel of the previous polynomial: 256-thread workgroups, < 64 VGPRs,
// <= 4 KiB of LDS and raised wave priority, so that its waves are placed in the registers the
// accumulate kernel leaves free and win instruction issue when they have something to do (a
// library radix sort with 1024-thread workgroups was measured to make no progress there).
//
//   partition 1   (one segment: all scalars, chunks of CH1 scalars)
//     count1      digits of a chunk -> LDS histogram over NBIN bins (high bucket bits)
//     row_scan    per bin: exclusive prefix over chunks;  bins: bin starts, chunk list for step 2
//     scatter1    digits again -> 8-byte entries (low bucket bits | table index, sign) into the bin
//   partition 2   (one segment per bin)
//     binsort     a bin of <= STAGE_CAP entries (with uniform scalars: every bin the twelve full windows fill alone,
//                 12 n / NBIN = 6.1 K entries at 2^20; the short TOP window -- bits 240..254, 13-15 significant bits --
//                 adds n / 32 .. n / 128 entries to each of the lowest 32 .. 128 bins, which therefore outgrow the
//                 stage and take the chunked kernels below: ~1/13 of the entries) is sorted by ONE workgroup: LDS
//                 histogram over its BPB buckets -> bstart[]; ranks by LDS atomics; the table indices are placed in
//                 an LDS stage of the bin's size and leave as one contiguous, fully coalesced run -- the HBM write
//                 is the payload (round 2 scattered 4-byte elements to 1024 open runs per workgroup, which the L2
//                 handed on as partial lines: WRITE_SIZE 7.4x the payload);
//     count2 / scan2 / scatter2   the chunked form of the same (chunks of CH2 entries, histogram per chunk, prefix
//                 over chunks and buckets, scattered stores) for a bin that outgrows the stage: skewed scalars only
//   order         buckets by length class (255 - min(len, 255): longest first), the same
//                 count / row_scan / scatter scheme with 256 classes; slices of <= SEG entries
//                 per bucket and their exclusive scan (slice_off, two-level, same kernels); arms
//                 the work counter.  No library call anywhere in the stage.
//
// Entries of one bucket land in no fixed order (LDS atomics hand out the ranks).  The group law
// in ec.h is exact for every case, so the bucket sum - and the affine result - does not depend on it.
#include <cstring>
#include <string.h>
#include <algorithm>
#include "internal.h"
#include "msm_prep.h"

namespace kzg {
namespace {

template <int WB, int LOBV = 8>
struct PW {
  static constexpr int NWIN = (256 + WB - 1) / WB;
  static constexpr uint32_t NB = 1u << (WB - 1);
  static constexpr int LOB = LOBV;                        // low bucket bits: buckets per bin
  static constexpr uint32_t BPB = 1u << LOB;
  static constexpr uint32_t NBIN = NB >> LOB;             // LOB = 8: 2048 (c = 20) / 128 (c = 16) bins -- a uniform
                                                          // 2^20-scalar commit puts 6.6 K entries into each bin;
                                                          // LOB = 7 / 6 for up to 2^21 / 2^22 scalars (lob_for)
};
constexpr uint32_t TPB = 256;     // threads per workgroup, every kernel here
#ifndef KZG_PREP_PRIO
#define KZG_PREP_PRIO 3
#endif
#ifndef KZG_PREP_CH1
#define KZG_PREP_CH1 4096
#endif
constexpr uint32_t CH1 = KZG_PREP_CH1;    // scalars per partition-1 chunk
#ifndef KZG_PREP_CH2
#define KZG_PREP_CH2 8192
#endif
constexpr uint32_t CH2 = KZG_PREP_CH2;    // entries per partition-2 chunk (bins beyond the stage)
#ifndef KZG_PREP_STAGE_CAP
#define KZG_PREP_STAGE_CAP 7168
#endif
#ifndef KZG_PREP_SWEEPS
#define KZG_PREP_SWEEPS 2
#endif
// Bin ranges the partition-1 scatter walks its chunk for.  Two sweeps of 1024 bins: partition 1 alone 0.190 -> 0.169 ms,
// +1.5 % commits/s (same box, alternating, twice) at an unchanged WRITE_SIZE (374 -> 379 MB per commit; four sweeps
// write 282 MB and are no faster, eight are slower: the digits are extracted once per sweep) --
// profiles/r03_scatter1_ab.txt, profiles/r03_pmc.csv
constexpr uint32_t SWEEPS = KZG_PREP_SWEEPS;
constexpr uint32_t STAGE_CAP = KZG_PREP_STAGE_CAP;   // entries a bin may have to be sorted in LDS: 28 KiB of table indices
constexpr uint32_t CHL = 2048;    // buckets per ordering chunk
constexpr uint32_t NCLS = 256;    // length classes

__device__ __forceinline__ void side_priority() { __builtin_amdgcn_s_setprio(KZG_PREP_PRIO); }

// scalar i (8 little-endian words, < 2^255) -> signed digits; f(window, bucket = |d|-1, negative)
// for every non-zero digit.  The top digit never wraps (msm.hip, Win).
template <int WB, class Fn>
__device__ __forceinline__ void for_each_digit(const uint32_t* scalars, uint32_t i, Fn&& f) {
  const uint4* sp = reinterpret_cast<const uint4*>(scalars + (size_t)i * 8);
  const uint4 lo4 = sp[0], hi4 = sp[1];
  const uint32_t w[9] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w, 0u};
  uint32_t carry = 0;
#pragma unroll
  for (int j = 0; j < PW<WB>::NWIN; ++j) {
    const int bit = j * WB, k = bit >> 5, sh = bit & 31;
    const uint64_t two = ((uint64_t)w[k + 1] << 32) | w[k];
    uint32_t d = ((uint32_t)(two >> sh) & ((1u << WB) - 1)) + carry;
    uint32_t neg = 0;
    if (d > (1u << (WB - 1))) { d = (1u << WB) - d; neg = 1; carry = 1; } else { carry = 0; }
    if (d) f((uint32_t)j, d - 1, neg);
  }
}

// inclusive scan of one value per thread over the 256 threads of the workgroup; `sh` has TPB words
__device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* sh) {
  const uint32_t tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
#pragma unroll
  for (uint32_t off = 1; off < TPB; off <<= 1) {
    const uint32_t t = tid >= off ? sh[tid - off] : 0u;
    __syncthreads();
    sh[tid] += t;
    __syncthreads();
  }
  return sh[tid];
}

// ---- partition 1 ------------------------------------------------------------------

// hist1[bin][chunk] = entries of the chunk that fall into the bin
template <int WB, int LOBV>
__global__ __launch_bounds__(TPB) void prep_count1_kernel(const uint32_t* scalars, uint32_t n, uint32_t nchunk,
                                                          uint32_t* hist1) {
  using P = PW<WB, LOBV>;
  side_priority();
  __shared__ uint32_t hist[P::NBIN];
  for (uint32_t b = threadIdx.x; b < P::NBIN; b += TPB) hist[b] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * CH1;
  for (uint32_t it = 0; it < CH1 / TPB; ++it) {
    const uint32_t i = base + it * TPB + threadIdx.x;
    if (i < n) for_each_digit<WB>(scalars, i, [&](uint32_t, uint32_t key, uint32_t) { atomicAdd(&hist[key >> P::LOB], 1u); });
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < P::NBIN; b += TPB) hist1[(size_t)b * nchunk + blockIdx.x] = hist[b];
}

// one workgroup per row: exclusive prefix over the `len` entries of the row (in place), row total
__global__ __launch_bounds__(TPB) void prep_row_scan_kernel(uint32_t* hist, uint32_t len, uint32_t* row_total) {
  side_priority();
  __shared__ uint32_t sh[TPB];
  uint32_t* row = hist + (size_t)blockIdx.x * len;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < len; base += TPB) {
    const uint32_t idx = base + threadIdx.x;
    const uint32_t v = idx < len ? row[idx] : 0u;
    const uint32_t incl = block_inclusive_scan(v, sh);
    if (idx < len) row[idx] = carry + incl - v;
    carry += sh[TPB - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}

// one workgroup: bin starts (exclusive scan of the bin totals), the partition-2 chunk list
// (chunk_base[s] .. chunk_base[s+1]: chunks of bin s; seg_of_chunk[g] = s), bstart[NB] = #entries
template <int WB, int LOBV>
__global__ __launch_bounds__(TPB) void prep_bins_kernel(const uint32_t* bin_total, uint32_t* bin_start,
                                                        uint32_t* chunk_base, uint32_t* seg_of_chunk,
                                                        uint32_t* nchunk2, uint32_t* bstart_top) {
  using P = PW<WB, LOBV>;
  side_priority();
  __shared__ uint32_t sh[TPB];
  uint32_t size_carry = 0, chunk_carry = 0;
  for (uint32_t base = 0; base < P::NBIN; base += TPB) {
    const uint32_t s = base + threadIdx.x;
    const uint32_t size = s < P::NBIN ? bin_total[s] : 0u;
    const uint32_t nc = size > STAGE_CAP ? (size + CH2 - 1) / CH2 : 0u;    // a bin that fits the stage needs no chunks
    const uint32_t size_incl = block_inclusive_scan(size, sh);
    const uint32_t size_tot = sh[TPB - 1];
    __syncthreads();
    const uint32_t nc_incl = block_inclusive_scan(nc, sh);
    const uint32_t nc_tot = sh[TPB - 1];
    __syncthreads();
    if (s < P::NBIN) {
      bin_start[s] = size_carry + size_incl - size;
      const uint32_t c0 = chunk_carry + nc_incl - nc;
      chunk_base[s] = c0;
      for (uint32_t q = 0; q < nc; ++q) seg_of_chunk[c0 + q] = s;
    }
    size_carry += size_tot;
    chunk_carry += nc_tot;
  }
  if (threadIdx.x == 0) {
    bin_start[P::NBIN] = size_carry;
    chunk_base[P::NBIN] = chunk_carry;
    *nchunk2 = chunk_carry;
    *bstart_top = size_carry;
  }
}

// entry = low bucket bits << 32 | sign << 31 | table index (window * srs_n + i)
//
// Write combining in pairs (round 4; measured, NOT the default).  A chunk leaves ~26 entries in each of 2048 bins, and
// every scattered 8-byte store costs a 32-byte sector by the time its line leaves the L2 (WRITE_SIZE 379 MB for 109 MB
// of entries, profiles/r03_pmc.csv).  With KZG_PREP_PAIR=1 entries leave in PAIRS: a bin has one 8-byte slot in LDS; an
// entry is either deposited in the empty slot or takes the entry waiting there (compare-and-swap on the slot, no lane
// ever waits for another) and the two go out as ONE 16-byte store at two consecutive positions of the bin; what is
// left in the slots when a sweep ends is stored singly.  16 KiB of slots + 8 KiB of cursors fit beside the accumulate
// workgroups.  Result (profiles/r04_scatter_pair_ab.txt, one box, alternating, bench.py --mode batch --steps 30):
// WRITE_SIZE 378 -> 302 MB per commit (a 16-byte store still costs a sector, two when it straddles one), partition 1
// alone 0.169 -> 0.188-0.195 ms (two LDS compare-and-swaps per entry), commits/s 462.1 / 464.1 with pairs against
// 463.4 / 463.3 without: the pipeline does not notice 76 MB of writes, so the plain scatter stays.
#ifndef KZG_PREP_PAIR
#define KZG_PREP_PAIR 0
#endif
// two entries at p[0], p[1] as one 16-byte store (p is 8-byte aligned; gfx950 global stores need dword alignment only)
__device__ __forceinline__ void store_pair(uint64_t* p, unsigned long long a, unsigned long long b) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(8)));
  u32x4 v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
  *reinterpret_cast<u32x4*>(p) = v;
}
template <int WB, int LOBV>
__global__ __launch_bounds__(TPB) void prep_scatter1_kernel(const uint32_t* scalars, uint32_t n, uint32_t srs_n,
                                                            uint32_t nchunk, const uint32_t* hist1,
                                                            const uint32_t* bin_start, uint64_t* ent) {
  using P = PW<WB, LOBV>;
  constexpr bool PAIR = KZG_PREP_PAIR && P::NBIN <= 2048;
  constexpr unsigned long long EMPTY = ~0ull;             // a real entry's upper word is a bucket index < BPB
  side_priority();
  __shared__ uint32_t cur[P::NBIN];
  __shared__ unsigned long long slot[PAIR ? P::NBIN : 1];
  const uint32_t chunk = blockIdx.x;
  for (uint32_t b = threadIdx.x; b < P::NBIN; b += TPB) {
    cur[b] = bin_start[b] + hist1[(size_t)b * nchunk + chunk];
    if (PAIR) slot[b] = EMPTY;
  }
  __syncthreads();
  const uint32_t base = chunk * CH1;
  // the chunk is walked once per range of NBIN / SWEEPS bins, so that a workgroup has fewer runs open at a time and
  // more of its stores meet in L2 before their line leaves (its scalars come from L2 after the first walk; the
  // digits are extracted again)
  for (uint32_t sweep = 0; sweep < SWEEPS; ++sweep) {
    const uint32_t b_lo = sweep * (P::NBIN / SWEEPS), b_hi = b_lo + P::NBIN / SWEEPS;
    for (uint32_t it = 0; it < CH1 / TPB; ++it) {
      const uint32_t i = base + it * TPB + threadIdx.x;
      if (i < n)
        for_each_digit<WB>(scalars, i, [&](uint32_t j, uint32_t key, uint32_t neg) {
          const uint32_t bin = key >> P::LOB;
          if (SWEEPS > 1 && (bin < b_lo || bin >= b_hi)) return;
          const unsigned long long mine =
              ((unsigned long long)(key & (P::BPB - 1)) << 32) | (unsigned long long)((j * srs_n + i) | (neg << 31));
          if constexpr (PAIR) {
            for (;;) {
              const unsigned long long seen = atomicCAS(&slot[bin], EMPTY, mine);
              if (seen == EMPTY) break;                                       // deposited: a later entry takes it along
              if (atomicCAS(&slot[bin], seen, EMPTY) == seen) {               // took the waiting entry: the pair leaves
                const uint32_t pos = atomicAdd(&cur[bin], 2u);
                store_pair(ent + pos, seen, mine);
                break;
              }
            }
          } else {
            const uint32_t pos = atomicAdd(&cur[bin], 1u);
            ent[pos] = mine;
          }
        });
    }
    if constexpr (PAIR) {     // singles of this sweep's bins
      __syncthreads();
      for (uint32_t b = b_lo + threadIdx.x; b < b_hi; b += TPB) {
        const unsigned long long e = slot[b];
        if (e != EMPTY) {
          ent[atomicAdd(&cur[b], 1u)] = e;
          slot[b] = EMPTY;
        }
      }
      __syncthreads();
    }
  }
}

// ---- partition 2 ------------------------------------------------------------------

struct Chunk2 {
  uint32_t seg, e0, e1;
};
__device__ __forceinline__ Chunk2 chunk2_of(uint32_t g, const uint32_t* bin_start, const uint32_t* chunk_base,
                                            const uint32_t* seg_of_chunk) {
  Chunk2 c;
  c.seg = seg_of_chunk[g];
  const uint32_t lc = g - chunk_base[c.seg];
  c.e0 = bin_start[c.seg] + lc * CH2;
  c.e1 = min(c.e0 + CH2, bin_start[c.seg + 1]);
  return c;
}

// hist2[chunk][bucket-in-bin]
template <int WB, int LOBV>
__global__ __launch_bounds__(TPB) void prep_count2_kernel(const uint64_t* ent, const uint32_t* bin_start,
                                                          const uint32_t* chunk_base, const uint32_t* seg_of_chunk,
                                                          const uint32_t* nchunk2, uint32_t* hist2) {
  using P = PW<WB, LOBV>;
  side_priority();
  const uint32_t g = blockIdx.x;
  if (g >= *nchunk2) return;
  __shared__ uint32_t hist[P::BPB];
  for (uint32_t b = threadIdx.x; b < P::BPB; b += TPB) hist[b] = 0;
  __syncthreads();
  const Chunk2 c = chunk2_of(g, bin_start, chunk_base, seg_of_chunk);
  for (uint32_t e = c.e0 + threadIdx.x; e < c.e1; e += TPB) atomicAdd(&hist[(uint32_t)(ent[e] >> 32)], 1u);
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < P::BPB; b += TPB) hist2[(size_t)g * P::BPB + b] = hist[b];
}

// one workgroup per bin: hist2[chunk][b] -> entries of bucket b in earlier chunks of the bin;
// bstart[bin * BPB + b] = first sorted entry of the bucket
template <int WB, int LOBV>
__global__ __launch_bounds__(TPB) void prep_scan2_kernel(uint32_t* hist2, const uint32_t* bin_start,
                                                         const uint32_t* chunk_base, uint32_t* bstart) {
  using P = PW<WB, LOBV>;
  constexpr uint32_t PER = P::BPB >= TPB ? P::BPB / TPB : 1;     // buckets a thread owns in the scan (threads beyond
  side_priority();                                                //   the bin's buckets own none)
  __shared__ uint32_t tot[P::BPB];
  __shared__ uint32_t sh[TPB];
  const uint32_t s = blockIdx.x;
  const uint32_t c0 = chunk_base[s], c1 = chunk_base[s + 1];
  if (c0 == c1) return;                                   // sorted by prep_binsort_kernel (or empty)
  for (uint32_t b = threadIdx.x; b < P::BPB; b += TPB) {
    uint32_t run = 0;
    for (uint32_t c = c0; c < c1; ++c) {
      const uint32_t v = hist2[(size_t)c * P::BPB + b];
      hist2[(size_t)c * P::BPB + b] = run;
      run += v;
    }
    tot[b] = run;
  }
  __syncthreads();
  const bool own = threadIdx.x * PER < P::BPB;
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t q = 0; q < PER; ++q) mine += own ? tot[threadIdx.x * PER + q] : 0u;
  const uint32_t incl = block_inclusive_scan(mine, sh);
  uint32_t run = bin_start[s] + incl - mine;
  if (own) {
#pragma unroll
    for (uint32_t q = 0; q < PER; ++q) {
      bstart[(size_t)s * P::BPB + threadIdx.x * PER + q] = run;
      run += tot[threadIdx.x * PER + q];
    }
  }
}

template <int WB, int LOBV>
__global__ __launch_bounds__(TPB) void prep_scatter2_kernel(const uint64_t* ent, const uint32_t* bin_start,
                                                            const uint32_t* chunk_base, const uint32_t* seg_of_chunk,
                                                            const uint32_t* nchunk2, const uint32_t* hist2,
                                                            const uint32_t* bstart, uint32_t* vals) {
  using P = PW<WB, LOBV>;
  side_priority();
  const uint32_t g = blockIdx.x;
  if (g >= *nchunk2) return;
  __shared__ uint32_t cur[P::BPB];
  const Chunk2 c = chunk2_of(g, bin_start, chunk_base, seg_of_chunk);
  for (uint32_t b = threadIdx.x; b < P::BPB; b += TPB)
    cur[b] = bstart[(size_t)c.seg * P::BPB + b] + hist2[(size_t)g * P::BPB + b];
  __syncthreads();
  for (uint32_t e = c.e0 + threadIdx.x; e < c.e1; e += TPB) {
    const uint64_t v = ent[e];
    const uint32_t pos = atomicAdd(&cur[(uint32_t)(v >> 32)], 1u);
    vals[pos] = (uint32_t)v;
  }
}

// One workgroup sorts a whole bin (<= STAGE_CAP entries) through registers and LDS: every thread keeps its <= EPT
// entries in registers, LDS histogram over the bin's buckets, exclusive scan -> bstart[], ranks by LDS atomics, table
// indices staged in LDS at their final offset inside the bin, then one contiguous copy to vals[bin_start ..).
// Reads every entry once and writes each output byte once, in full lines.
template <int WB, int LOBV>
__global__ __launch_bounds__(TPB) void prep_binsort_kernel(const uint64_t* ent, const uint32_t* bin_start,
                                                           uint32_t* bstart, uint32_t* vals) {
  using P = PW<WB, LOBV>;
  constexpr uint32_t PER = P::BPB >= TPB ? P::BPB / TPB : 1;   // buckets a thread owns in the scan
  constexpr uint32_t EPT = STAGE_CAP / TPB;               // entries per thread
  static_assert(STAGE_CAP % TPB == 0, "the stage is a whole number of entries per thread");
  side_priority();
  __shared__ uint32_t cur[P::BPB];
  __shared__ uint32_t sh[TPB];
  __shared__ uint32_t stage[STAGE_CAP];
  const uint32_t s = blockIdx.x;
  const uint32_t e0 = bin_start[s], e1 = bin_start[s + 1];
  const uint32_t size = e1 - e0;
  if (size > STAGE_CAP) return;                           // the chunked kernels take this bin
  for (uint32_t b = threadIdx.x; b < P::BPB; b += TPB) cur[b] = 0;
  uint64_t mine_e[EPT];
#pragma unroll
  for (uint32_t q = 0; q < EPT; ++q) {
    const uint32_t e = e0 + q * TPB + threadIdx.x;
    mine_e[q] = e < e1 ? ent[e] : ~0ull;                  // ~0: no entry (a real key is < BPB)
  }
  __syncthreads();
#pragma unroll
  for (uint32_t q = 0; q < EPT; ++q)
    if ((uint32_t)(mine_e[q] >> 32) < P::BPB) atomicAdd(&cur[(uint32_t)(mine_e[q] >> 32)], 1u);
  __syncthreads();
  const bool own = threadIdx.x * PER < P::BPB;            // (a bin of fewer than TPB buckets: the other threads own none)
  uint32_t mine = 0;
#pragma unroll
  for (uint32_t q = 0; q < PER; ++q) mine += own ? cur[threadIdx.x * PER + q] : 0u;
  const uint32_t incl = block_inclusive_scan(mine, sh);
  uint32_t run = incl - mine;
  if (own) {
#pragma unroll
    for (uint32_t q = 0; q < PER; ++q) {
      const uint32_t cnt = cur[threadIdx.x * PER + q];
      bstart[(size_t)s * P::BPB + threadIdx.x * PER + q] = e0 + run;
      cur[threadIdx.x * PER + q] = run;                   // becomes the running rank inside the bin
      run += cnt;
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t q = 0; q < EPT; ++q)
    if ((uint32_t)(mine_e[q] >> 32) < P::BPB) stage[atomicAdd(&cur[(uint32_t)(mine_e[q] >> 32)], 1u)] = (uint32_t)mine_e[q];
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < size; i += TPB) vals[e0 + i] = stage[i];
}

// ---- buckets in length order ----------------------------------------------------------

__device__ __forceinline__ uint32_t length_class(uint32_t len) { return 255u - min(len, 255u); }

template <int WB>
__global__ __launch_bounds__(TPB) void prep_lcount_kernel(const uint32_t* bstart, uint32_t nchunk, uint32_t* histl) {
  using P = PW<WB>;
  side_priority();
  __shared__ uint32_t hist[NCLS];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * CHL;
  for (uint32_t it = 0; it < CHL / TPB; ++it) {
    const uint32_t k = base + it * TPB + threadIdx.x;
    if (k < P::NB) atomicAdd(&hist[length_class(bstart[k + 1] - bstart[k])], 1u);
  }
  __syncthreads();
  histl[(size_t)threadIdx.x * nchunk + blockIdx.x] = hist[threadIdx.x];
}

// one workgroup: class_start = exclusive scan of the NCLS (= TPB) class totals
__global__ __launch_bounds__(TPB) void prep_classes_kernel(const uint32_t* class_total, uint32_t* class_start) {
  side_priority();
  __shared__ uint32_t sh[TPB];
  const uint32_t v = class_total[threadIdx.x];
  const uint32_t incl = block_inclusive_scan(v, sh);
  class_start[threadIdx.x] = incl - v;
}

template <int WB>
__global__ __launch_bounds__(TPB) void prep_lscatter_kernel(const uint32_t* bstart, uint32_t nchunk,
                                                            const uint32_t* histl, const uint32_t* class_start,
                                                            uint32_t* order) {
  using P = PW<WB>;
  side_priority();
  __shared__ uint32_t cur[NCLS];
  cur[threadIdx.x] = class_start[threadIdx.x] + histl[(size_t)threadIdx.x * nchunk + blockIdx.x];
  __syncthreads();
  const uint32_t base = blockIdx.x * CHL;
  for (uint32_t it = 0; it < CHL / TPB; ++it) {
    const uint32_t k = base + it * TPB + threadIdx.x;
    if (k < P::NB) order[atomicAdd(&cur[length_class(bstart[k + 1] - bstart[k])], 1u)] = k;
  }
}

// slice_off = exclusive scan of ns over the NB + 1 ranks (ns[r] = slices of the r-th bucket in length
// order, ns[NB] = 0 so that slice_off[NB] is the total), in three launches of this file's own kernels:
//   ns_totals   per block of TPB ranks: the block's number of slices
//   row_scan    one workgroup: exclusive prefix over the block totals (prep_row_scan_kernel, one row)
//   slice_off   per block: ns again, scanned inside the block, plus the block's prefix
template <int WB>
__device__ __forceinline__ uint32_t slices_of_rank(const uint32_t* bstart, const uint32_t* order, uint32_t seg, uint32_t r) {
  if (r >= PW<WB>::NB) return 0u;
  const uint32_t k = order[r];
  return (bstart[k + 1] - bstart[k] + seg - 1) / seg;
}
template <int WB>
__global__ __launch_bounds__(TPB) void prep_ns_totals_kernel(const uint32_t* bstart, const uint32_t* order, uint32_t seg,
                                                             uint32_t* blk_total) {
  side_priority();
  __shared__ uint32_t sh[TPB];
  const uint32_t v = slices_of_rank<WB>(bstart, order, seg, blockIdx.x * TPB + threadIdx.x);
  block_inclusive_scan(v, sh);
  if (threadIdx.x == 0) blk_total[blockIdx.x] = sh[TPB - 1];
}
// also arms the accumulate kernel's work counter
template <int WB>
__global__ __launch_bounds__(TPB) void prep_slice_off_kernel(const uint32_t* bstart, const uint32_t* order, uint32_t seg,
                                                             const uint32_t* blk_prefix, uint32_t* slice_off,
                                                             uint32_t* chunk_counter) {
  side_priority();
  __shared__ uint32_t sh[TPB];
  const uint32_t r = blockIdx.x * TPB + threadIdx.x;
  const uint32_t v = slices_of_rank<WB>(bstart, order, seg, r);
  const uint32_t incl = block_inclusive_scan(v, sh);
  if (r <= PW<WB>::NB) slice_off[r] = blk_prefix[blockIdx.x] + incl - v;
  if (r == 0) *chunk_counter = 0;
}

// chunk_rank[c] = rank (position in length order) of the bucket that owns slice 64*c, the first slice
// of the accumulate kernel's work chunk c: the kernel's lanes then search only the 64 ranks from there
// instead of all NB (every non-empty bucket has at least one slice, so slice 64*c + l belongs to one of
// the ranks chunk_rank[c] .. chunk_rank[c] + l).
template <int WB>
__global__ __launch_bounds__(TPB) void prep_chunk_rank_kernel(const uint32_t* slice_off, uint32_t nchunk_max,
                                                              uint32_t* chunk_rank) {
  constexpr uint32_t NB = PW<WB>::NB;
  side_priority();
  const uint32_t c = blockIdx.x * TPB + threadIdx.x;
  if (c >= nchunk_max) return;
  const uint32_t t = c * 64u;
  uint32_t lo = 0, hi = NB;
  if (t < slice_off[NB]) {
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (slice_off[mid] <= t) lo = mid; else hi = mid;
    }
  }
  chunk_rank[c] = lo;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <int WB, int LOBV>
struct Layout {
  using P = PW<WB, LOBV>;
  uint32_t m, nchunk1, nchunk2_max, nchunkl;
  size_t off_ent, off_hist1, off_bin_total, off_bin_start, off_chunk_base, off_seg_of_chunk, off_nchunk2, off_hist2,
      off_histl, off_class_total, off_class_start, off_ns, total;
  explicit Layout(uint32_t n) {
    m = n * (uint32_t)P::NWIN;
    nchunk1 = (n + CH1 - 1) / CH1;
    nchunk2_max = m / CH2 + P::NBIN + 1;
    nchunkl = (P::NB + CHL - 1) / CHL;
    size_t t = 0;
    auto take = [&](size_t bytes) { const size_t o = t; t += align256(bytes); return o; };
    off_ent = take((size_t)m * 8);
    off_hist1 = take((size_t)P::NBIN * nchunk1 * 4);
    off_bin_total = take((size_t)P::NBIN * 4);
    off_bin_start = take((size_t)(P::NBIN + 1) * 4);
    off_chunk_base = take((size_t)(P::NBIN + 1) * 4);
    off_seg_of_chunk = take((size_t)nchunk2_max * 4);
    off_nchunk2 = take(4);
    off_hist2 = take((size_t)nchunk2_max * P::BPB * 4);
    off_histl = take((size_t)NCLS * nchunkl * 4);
    off_class_total = take(NCLS * 4);
    off_class_start = take(NCLS * 4);
    off_ns = take((size_t)((P::NB + 1 + TPB - 1) / TPB + 2) * 4);      // block totals of the slice counts (+ their sum)
    total = t;
  }
};

template <int WB, int LOBV>
int prep_enqueue_t(Ctx* c, hipStream_t sp, const uint32_t* d_scalars, uint32_t n, uint32_t srs_n, uint32_t seg,
                   void* ws, uint32_t* vals, uint32_t* bstart, uint32_t* order, uint32_t* slice_off,
                   uint32_t* chunk_counter, uint32_t* chunk_rank, uint32_t nchunk_max) {
  using P = PW<WB, LOBV>;
  const Layout<WB, LOBV> L(n);
  char* w = static_cast<char*>(ws);
  auto* ent = reinterpret_cast<uint64_t*>(w + L.off_ent);
  auto* hist1 = reinterpret_cast<uint32_t*>(w + L.off_hist1);
  auto* bin_total = reinterpret_cast<uint32_t*>(w + L.off_bin_total);
  auto* bin_start = reinterpret_cast<uint32_t*>(w + L.off_bin_start);
  auto* chunk_base = reinterpret_cast<uint32_t*>(w + L.off_chunk_base);
  auto* seg_of_chunk = reinterpret_cast<uint32_t*>(w + L.off_seg_of_chunk);
  auto* nchunk2 = reinterpret_cast<uint32_t*>(w + L.off_nchunk2);
  auto* hist2 = reinterpret_cast<uint32_t*>(w + L.off_hist2);
  auto* histl = reinterpret_cast<uint32_t*>(w + L.off_histl);
  auto* class_total = reinterpret_cast<uint32_t*>(w + L.off_class_total);
  auto* class_start = reinterpret_cast<uint32_t*>(w + L.off_class_start);
  auto* ns = reinterpret_cast<uint32_t*>(w + L.off_ns);
  {
    ProfScope ps(c, "msm_partition1", sp);
    hipLaunchKernelGGL((prep_count1_kernel<WB, LOBV>), dim3(L.nchunk1), dim3(TPB), 0, sp, d_scalars, n, L.nchunk1, hist1);
    hipLaunchKernelGGL(prep_row_scan_kernel, dim3(P::NBIN), dim3(TPB), 0, sp, hist1, L.nchunk1, bin_total);
    hipLaunchKernelGGL((prep_bins_kernel<WB, LOBV>), dim3(1), dim3(TPB), 0, sp, bin_total, bin_start, chunk_base, seg_of_chunk,
                       nchunk2, bstart + P::NB);
    hipLaunchKernelGGL((prep_scatter1_kernel<WB, LOBV>), dim3(L.nchunk1), dim3(TPB), 0, sp, d_scalars, n, srs_n, L.nchunk1,
                       hist1, bin_start, ent);
  }
  KZG_HIP(c, hipGetLastError());
  {
    ProfScope ps(c, "msm_partition2", sp);
    hipLaunchKernelGGL((prep_binsort_kernel<WB, LOBV>), dim3(P::NBIN), dim3(TPB), 0, sp, ent, bin_start, bstart, vals);
    hipLaunchKernelGGL((prep_count2_kernel<WB, LOBV>), dim3(L.nchunk2_max), dim3(TPB), 0, sp, ent, bin_start, chunk_base,
                       seg_of_chunk, nchunk2, hist2);
    hipLaunchKernelGGL((prep_scan2_kernel<WB, LOBV>), dim3(P::NBIN), dim3(TPB), 0, sp, hist2, bin_start, chunk_base, bstart);
    hipLaunchKernelGGL((prep_scatter2_kernel<WB, LOBV>), dim3(L.nchunk2_max), dim3(TPB), 0, sp, ent, bin_start, chunk_base,
                       seg_of_chunk, nchunk2, hist2, bstart, vals);
  }
  KZG_HIP(c, hipGetLastError());
  {
    ProfScope ps(c, "msm_order", sp);
    hipLaunchKernelGGL(prep_lcount_kernel<WB>, dim3(L.nchunkl), dim3(TPB), 0, sp, bstart, L.nchunkl, histl);
    hipLaunchKernelGGL(prep_row_scan_kernel, dim3(NCLS), dim3(TPB), 0, sp, histl, L.nchunkl, class_total);
    hipLaunchKernelGGL(prep_classes_kernel, dim3(1), dim3(TPB), 0, sp, class_total, class_start);
    hipLaunchKernelGGL(prep_lscatter_kernel<WB>, dim3(L.nchunkl), dim3(TPB), 0, sp, bstart, L.nchunkl, histl,
                       class_start, order);
    const uint32_t nblk = (P::NB + 1 + TPB - 1) / TPB;
    hipLaunchKernelGGL(prep_ns_totals_kernel<WB>, dim3(nblk), dim3(TPB), 0, sp, bstart, order, seg, ns);
    hipLaunchKernelGGL(prep_row_scan_kernel, dim3(1), dim3(TPB), 0, sp, ns, nblk, ns + nblk);
    hipLaunchKernelGGL(prep_slice_off_kernel<WB>, dim3(nblk), dim3(TPB), 0, sp, bstart, order, seg, ns, slice_off,
                       chunk_counter);
    hipLaunchKernelGGL(prep_chunk_rank_kernel<WB>, dim3((nchunk_max + TPB - 1) / TPB), dim3(TPB), 0, sp, slice_off,
                       nchunk_max, chunk_rank);
  }
  KZG_HIP(c, hipGetLastError());
  return KZG_OK;
}

}  // namespace

// Bucket bits per bin for an n-scalar commit: as many bins as keep a bin of uniform scalars (n * windows / bins entries)
// within the sort stage, up to 8192 (the partition-1 histograms of a workgroup live in LDS: 32 KiB at 8192 bins).
static int lob_for(uint32_t n, int win_bits) {
  if (win_bits != 20) return 8;
  // the bins that only the full windows fill (all but the lowest few dozen) must fit the bin sort's stage: their mean
  // m / NBIN (m = entries of all 13 windows: an upper bound of what the 12 full ones put there) + 4 sigma of the 2^20
  // case; the bins the short top window overloads take the chunked kernels whatever the bin count
  const uint64_t m = (uint64_t)n * PW<20>::NWIN;
  for (int lob = 8; lob > 6; --lob)
    if (m / (PW<20>::NB >> lob) + 4 * 82 <= STAGE_CAP) return lob;
  return 6;
}

size_t msm_prep_workspace_bytes(uint32_t n, int win_bits) {
  if (win_bits != 20) return Layout<16, 8>(n).total;
  switch (lob_for(n, win_bits)) {
    case 8: return Layout<20, 8>(n).total;
    case 7: return Layout<20, 7>(n).total;
    default: return Layout<20, 6>(n).total;
  }
}

int msm_prep_enqueue(Ctx* c, hipStream_t sp, int win_bits, const uint32_t* d_scalars, uint32_t n, uint32_t srs_n,
                     uint32_t seg, void* ws, uint32_t* vals, uint32_t* bstart, uint32_t* order, uint32_t* slice_off,
                     uint32_t* chunk_counter, uint32_t* chunk_rank, uint32_t nchunk_max) {
#define KZG_PREP_GO(WBV, LOBV) prep_enqueue_t<WBV, LOBV>(c, sp, d_scalars, n, srs_n, seg, ws, vals, bstart, order, slice_off, \
                                                        chunk_counter, chunk_rank, nchunk_max)
  if (win_bits != 20) return KZG_PREP_GO(16, 8);
  switch (lob_for(n, win_bits)) {
    case 8: return KZG_PREP_GO(20, 8);
    case 7: return KZG_PREP_GO(20, 7);
    default: return KZG_PREP_GO(20, 6);
  }
#undef KZG_PREP_GO
}

}  // namespace kzg
