// variants.hpp -- per-dimension kernel tables. The pairwise summation tree is unrolled at compile time, so each
// supported vector dimension gets its own translation unit (search_d<D>.hip), built in parallel by the Makefile.
#pragma once
#include <stddef.h>

// A search-kernel VARIANT is one instantiation family of search_kernel (search_kernel.hpp). Variants are known by a stable
// id (dr_timing.variant, dr_debug_force_kind, the committed profiles); ids of variants that were retired (4, 6, 7, 10:
// superseded forms) are simply absent from the table below -- a kernel table row is its POSITION in DR_KINDS, not its id.
//   0 M1 (ADC filter + exact), per-query table in LDS, 1 wave per workgroup
//   1 exact traversal (M2, M4, M3 without PQ, builder), 1 wave per workgroup
//   2 ADC-only traversal (M3 with PQ, DR_MODE_PQ), per-query table in LDS
//   3 M1, codebook shared in LDS, 8 waves per workgroup      (D <= 128)
//   5 ADC-only traversal, codebook shared in LDS, 8 waves    (D <= 128)
//   8 exact traversal, vectors landed in LDS, 8 waves                                          (D = 128)
//   9 M1, vectors landed in LDS, 24 rows/burst, 12 waves                                       (D = 128)
//  11 M1, BYTE vectors landed in LDS (lossless copy of integer-valued data), 64 rows/burst, 16 waves (D = 128;
//     128 VGPRs: a few spilled dwords, 7 % faster than 12 waves once the row bytes are quartered)
//  12 exact traversal (M2, M4, M3 without PQ) on BYTE vectors, 64 rows/burst, 16 waves          (D = 128)
//  13 = 11 with BYTE queries as well (every component of the batch an integer in [0, 255]): v_dot4_u32_u8 distances
//  14 = 12 with byte queries
//  15 = 2 with the table rows of the LAST 16 sub-quantisers held in VGPRs (64 registers, looked up with ds_bpermute) and
//     only the first m - 16 rows in LDS: at m = 32 a table costs 16 KiB of LDS instead of 32, so 8 wavefronts fit a CU
//     instead of 4 (two per SIMD: the traversal is bound by instruction latency, not by memory)  (m % 16 == 0, m >= 32)
//  18 (not a row of DR_KINDS: its own kernel family, latency_kernel.hpp; the engine's choice for launches of <= 256 queries on long rows -- M1 above 960 dimensions, exact traversals above 256 --, DR_LAT_ALL=1 / forced elsewhere) M1 / exact traversals with a WORKGROUP of eight wavefronts per
//     query: scoring ahead of the decisions, visited ids in LDS -- the handful of queries of one request (round 5)
//  16 = 11, 17 = 13 in workgroups of FOUR wavefronts (same code, same 16 wavefronts per CU as four workgroups): a batch
//     smaller than the chip's 4096 wavefront slots -- the 1250-query slice of an 8-GPU strong-scaling job, a coalesced
//     handful of requests -- spreads over all 256 CUs instead of filling ceil(nq / 16) of them (round 4)
// (tried in round 2 and removed: 16-wave forms of 3 and 5 for D <= 96 -- twice the queries in flight, 30 % slower than the
//  per-query table at the c4 shape: that kernel is bound by its number of memory requests, not by latency)
// sizeclass: result capacity <= 64 / 128 / 256 / 512 / 1024 (the last one: one-wavefront-per-workgroup variants only)
// A/B build (VERDICT r5 item 2 i: "more requests in flight"): -DDR_AB_RB17=32 -DDR_AB_MINW17=6 turns variant 17 (byte rows + byte queries in
// 4-wavefront workgroups) into a 24-wavefronts-per-CU form -- bursts of 32 rows (4 KiB landing area per wavefront), six workgroups per CU, 80 VGPRs.
#ifndef DR_AB_RB17
#define DR_AB_RB17 64
#endif
#ifndef DR_AB_MINW17
#define DR_AB_MINW17 4
#endif
struct KindDesc {
    int id;
    int nw;       // wavefronts per workgroup
    bool cb;      // codebook copied to LDS
    int rb;       // rows per LDS burst (0: no row landing)
    bool lut;     // per-query table (built for the batch by lut_build_kernel)
    bool pq;      // reads PQ data
    bool u8, qb;  // byte rows, byte queries
    int treg;     // table rows (sub-quantisers) held in registers instead of LDS
};
static const KindDesc DR_KINDS[] = {
    { 0, 1, false, 0, true, true, false, false, 0 },
    { 1, 1, false, 0, false, false, false, false, 0 },
    { 2, 1, false, 0, true, true, false, false, 0 },
    { 3, 8, true, 0, false, true, false, false, 0 },
    { 5, 8, true, 0, false, true, false, false, 0 },
    { 8, 8, false, 32, false, false, false, false, 0 },
    { 9, 12, false, 24, false, true, false, false, 0 },
    { 11, 16, false, 64, false, true, true, false, 0 },
    { 12, 16, false, 64, false, false, true, false, 0 },
    { 13, 16, false, 64, false, true, true, true, 0 },
    { 14, 16, false, 64, false, false, true, true, 0 },
    { 15, 1, false, 0, true, true, false, false, 16 },
    { 16, 4, false, 64, false, true, true, false, 0 },
    { 17, 4, false, DR_AB_RB17, false, true, true, true, 0 },
};
#define DR_NUM_KINDS ((int)(sizeof(DR_KINDS) / sizeof(DR_KINDS[0])))
#define DR_MAX_KIND_ID 17
#define DR_NUM_SIZECLASS 5
#define DR_MAX_CAPACITY 1024u
// position of variant `id` in DR_KINDS (= its row in DimKernels::search), or -1
static inline int dr_kind_pos(int id)
{
    for (int i = 0; i < DR_NUM_KINDS; i++) if (DR_KINDS[i].id == id) return i;
    return -1;
}
// the small-workgroup twin of a variant (same kernel in 4-wavefront workgroups), or -1
static inline int dr_small_twin(int id) { return id == 11 ? 16 : id == 13 ? 17 : -1; }
struct DimKernels {
    int D;
    const void *search[sizeof(DR_KINDS) / sizeof(DR_KINDS[0])][DR_NUM_SIZECLASS];     // [position in DR_KINDS][sizeclass]; nullptr: not built for this dimension
    const void *exact;
    const void *bruteforce;
    const void *prune;
    const void *prune_multi;      // prune_kernel<D, true> (multi-pick form), nullptr where the split row form does not exist
    const void *nearest_pivot;   // aux_kernels.hpp nearest_pivot_kernel (bit order of the visited bitmap)
    const void *search_f64;   // M1 / M2 with float64 queries (the CLI path), search_f64.hpp
    const void *search_seq_f32;   // the same literal kernel on float32 queries (band policy 2: the reference's coin flip itself)
    const void *rerank;       // aux_kernels.hpp rerank_kernel (DR_MODE_PQ + DR_F_RERANK)
    const void *latency[2][DR_NUM_SIZECLASS];   // latency_kernel.hpp lat_kernel (variant 18: a workgroup per query): [0] exact traversals, [1] M1
};

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
