// msm_prep.h -- library-private interface of msm_prep.hip (stage P of the commit pipeline).
#pragma once
#include "internal.h"

namespace kzg {

// Bytes of scratch one polynomial of n scalars needs in stage P (entries, histograms, chunk lists).
size_t msm_prep_workspace_bytes(uint32_t n, int win_bits);

// Enqueues stage P on `sp`: the n scalars (8 words each, canonical) -> vals[0 .. bstart[NB]) table
// indices (window * srs_n + i, bit 31 = negate) grouped by bucket, bstart[0 .. NB] bucket bounds,
// order[0 .. NB) buckets by decreasing length, slice_off[0 .. NB] exclusive scan of the slice
// counts ceil(len / seg) in that order; *chunk_counter = 0; chunk_rank[c] (c < nchunk_max) = position in
// that order of the bucket owning slice 64*c.  NB = 2^(win_bits - 1).
int msm_prep_enqueue(Ctx* c, hipStream_t sp, int win_bits, const uint32_t* d_scalars, uint32_t n, uint32_t srs_n,
                     uint32_t seg, void* ws, uint32_t* vals, uint32_t* bstart, uint32_t* order, uint32_t* slice_off,
                     uint32_t* chunk_counter, uint32_t* chunk_rank, uint32_t nchunk_max);

}  // namespace kzg
