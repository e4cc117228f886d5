// variants.hpp -- per-dimension kernel tables. The pairwise summation tree is unrolled at compile time, so each
// supported vector dimension gets its own translation unit (search_d<D>.hip), built in parallel by the Makefile.
#pragma once
#include <stddef.h>

// kind: which search_kernel instantiation
//   0 M1 (ADC filter + exact), per-query table in LDS, 1 wave per workgroup
//   1 exact traversal (M2, M4, M3 without PQ, builder), 1 wave per workgroup
//   2 ADC-only traversal (M3 with PQ), per-query table
//   3 M1, codebook shared in LDS, 8 waves per workgroup      (D <= 128 only, else nullptr)
//   4 (retired)
//   5 ADC-only traversal, codebook shared in LDS, 8 waves    (D <= 128 only)
//   6, 7 (retired: superseded by 9)
//   8 exact traversal, vectors landed in LDS, 8 waves                                          (D = 128)
//   9 M1, vectors landed in LDS, 24 rows/burst, 12 waves                                       (D = 128)
//  10 (retired: the 12-wave form of 11)
//  11 M1, BYTE vectors landed in LDS (lossless copy of integer-valued data), 64 rows/burst, 16 waves (D = 128;
//     128 VGPRs: a few spilled dwords, 7 % faster than 12 waves once the row bytes are quartered)
//  12 exact traversal (M2, M4, M3 without PQ) on BYTE vectors, 64 rows/burst, 16 waves          (D = 128)
//  13 = 11 with BYTE queries as well (every component of the batch an integer in [0, 255]): v_dot4_u32_u8 distances
//  14 = 12 with byte queries
// (tried in round 2 and removed: 16-wave forms of 3 and 5 for D <= 96 -- twice the queries in flight, 30 % slower than the
//  per-query table at the c4 shape: that kernel is bound by its number of memory requests, not by latency)
// sizeclass: result capacity <= 64 / 128 / 256 / 512 / 1024 (the last one: one-wavefront-per-workgroup variants 0, 1, 2 only)
#define DR_NUM_KINDS 15
#define DR_NUM_SIZECLASS 5
#define DR_MAX_CAPACITY 1024u
struct DimKernels {
    int D;
    const void *search[DR_NUM_KINDS][DR_NUM_SIZECLASS];
    const void *exact;
    const void *bruteforce;
    const void *prune;
    const void *nearest_pivot;   // aux_kernels.hpp nearest_pivot_kernel (bit order of the visited bitmap)
    const void *search_f64;   // M1 / M2 with float64 queries (the CLI path), search_f64.hpp
    const void *rerank;       // aux_kernels.hpp rerank_kernel (DR_MODE_PQ + DR_F_RERANK)
};
static const int DR_KIND_NW[DR_NUM_KINDS] = { 1, 1, 1, 8, 16, 8, 8, 16, 8, 12, 12, 16, 16, 16, 16 };
static const bool DR_KIND_CB[DR_NUM_KINDS] = { false, false, false, true, true, true, false, false, false, false, false, false, false, false, false };   // codebook copied to LDS
static const int DR_KIND_RB[DR_NUM_KINDS] = { 0, 0, 0, 0, 0, 0, 32, 16, 32, 24, 64, 64, 64, 64, 64 };                                       // rows per LDS burst
static const bool DR_KIND_LUT[DR_NUM_KINDS] = { true, false, true, false, false, false, false, false, false, false, false, false, false, false, false };     // per-query table in LDS
static const bool DR_KIND_PQ[DR_NUM_KINDS] = { true, false, true, true, true, true, true, true, false, true, true, true, false, true, false };
static const bool DR_KIND_U8[DR_NUM_KINDS] = { false, false, false, false, false, false, false, false, false, false, true, true, true, true, true };    // byte rows
static const bool DR_KIND_QB[DR_NUM_KINDS] = { false, false, false, false, false, false, false, false, false, false, false, false, false, true, true };  // byte queries

const DimKernels *dr_dim_kernels(int D);

#define DR_DECLARE_DIM(DD) const DimKernels *dr_dim_kernels_##DD();
DR_DECLARE_DIM(32)
DR_DECLARE_DIM(64)
DR_DECLARE_DIM(96)
DR_DECLARE_DIM(128)
DR_DECLARE_DIM(256)
DR_DECLARE_DIM(768)
DR_DECLARE_DIM(960)
DR_DECLARE_DIM(1536)
