// variants.hpp -- per-dimension kernel tables. The pairwise summation tree is unrolled at compile time, so each
// supported vector dimension gets its own translation unit (search_d<D>.hip), built in parallel by the Makefile.
#pragma once
#include <stddef.h>

// kind: 0 = M1 (ADC filter + exact), 1 = exact traversal (M2, M4, M3 without PQ), 2 = ADC-only traversal (M3)
// sizeclass: result capacity <= 64 / 128 / 256 / 512
#define DR_NUM_SIZECLASS 4
struct DimKernels {
    int D;
    const void *search[3][DR_NUM_SIZECLASS];
    const void *exact;
    const void *bruteforce;
    const void *prune;
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
