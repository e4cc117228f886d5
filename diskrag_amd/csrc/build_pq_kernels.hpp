// build_pq_kernels.hpp -- Vamana construction over PQ code words only (no stored vectors). Dimension-independent:
// included by engine.hip only.
#pragma once
#include "build_kernels.hpp"

// ---- PQ-only construction (BASELINE config c5: the vectors of a 1.25e8 x 1536 shard are never stored) ---------------
// Every distance the builder needs is a sum of entries of the centroid-pair table S[j][a][b] = |C_j[a] - C_j[b]|^2
// (each entry in A2's summation order, the sum over j in A3's order): d(p, c) = sum_j S[j][code_p[j]][code_c[j]], the
// symmetric PQ distance the reference has as pq_distance_fast_cython (pydiskann/cython_utils.pyx:26-51). The reference
// itself never builds from codes ("always exact distances at build time", vamana_graph.py:405): this is the engine's
// own construction for shards whose vectors cannot be held, checked by graph quality, not by parity.

// S[j][a][:] = table row of sub-quantiser j for the query sub-vector C_j[a]; one block per (j, a), thread b.
__global__ __launch_bounds__(256) void sdc_table_kernel(const float *__restrict__ codebook, u32 m, u32 sd, float *__restrict__ sdc)
{
    const u32 ja = blockIdx.x, jq = ja >> 8, b = threadIdx.x;
    const float *ca = codebook + (size_t)ja * sd;
    const float *cbp = codebook + ((size_t)jq * 256 + b) * sd;
    sdc[(size_t)ja * 256 + b] = pw_run_lane(cbp, ca, (int)sd);
}

// nearest-centroid codes for `rows` row-major vectors (a streamed chunk; DiskANNPQ.encode, fast_pq.py:245-267)
__global__ void pq_assign_rows_kernel(const float *__restrict__ x, u64 rows, u32 D, u32 m, u32 sd,
                                      const float *__restrict__ codebook, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *cb = reinterpret_cast<float *>(smem);
    const u32 jq = blockIdx.y;
    for (u32 e = threadIdx.x; e < 256 * sd; e += blockDim.x) cb[e] = codebook[(size_t)jq * 256 * sd + e];
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (u64)gridDim.x * blockDim.x) {
        float v[128];
        for (u32 t = 0; t < sd; t++) v[t] = x[i * D + jq * sd + t];
        float best = 3.4e38f;
        u32 bi = 0;
        for (u32 c = 0; c < 256; c++) {
            float s2 = 0.0f;
            for (u32 t = 0; t < sd; t++) { const float d = v[t] - cb[c * sd + t]; s2 += d * d; }
            if (s2 < best) { best = s2; bi = c; }
        }
        out[i * m + jq] = (u8)bi;
    }
}

// per sub-quantiser histogram of the code words (the mean of the decoded vectors follows from it: medoid)
__global__ void code_histogram_kernel(const u8 *__restrict__ codes, u64 n, u32 m, u32 *__restrict__ hist /*[m][256]*/)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n * m; i += (u64)gridDim.x * blockDim.x)
        atomicAdd(&hist[(i % m) * 256 + codes[i]], 1u);
}

#define DR_PRUNE_PQ_MAXC 320    // candidates whose code words are cached in LDS (L_build + row slots of the c5 shapes)

struct PrunePQParams {
    const u8 *codes; const float *sdc; u32 m;
    u32 *adjb; u32 *deg; u32 RX, R; float alpha;
    const u32 *points; u32 npoints;
    const u64 *res_keys; const u32 *res_n; u32 cap;
    u32 *fwd; u32 *fwd_n;
};

// prune_kernel's twin on code words: one wavefront per point, the point's (then each pick's) m table rows staged in LDS,
// one lane per candidate for the m-term sums.
__global__ __launch_bounds__(64) void prune_pq_kernel(const PrunePQParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *rows = reinterpret_cast<float *>(smem);                                    // [m][256]
    u64 *keyA = reinterpret_cast<u64 *>(smem + (size_t)p.m * 1024);                   // [MAXC]
    u64 *keyB = keyA + DR_PRUNE_MAXC;
    u32 *raw = reinterpret_cast<u32 *>(keyB + DR_PRUNE_MAXC);
    u32 *keep = raw + DR_PRUNE_MAXC;
    u32 *outsel = keep + DR_PRUNE_MAXC;                                               // [256]
    u32 *cslot2 = outsel + 256;                                                       // [MAXC] second slot array of the compaction
    u8 *ccode = reinterpret_cast<u8 *>(cslot2 + DR_PRUNE_MAXC);                       // [DR_PRUNE_PQ_MAXC][m] candidates' code words
    const int lane = lane_id();
    // (code words are read as whole 16-byte pieces, and every loop over their bytes is unrolled: the m row loads / table
    // reads that depend on a piece are issued together, and no register array is indexed dynamically)
    const bool wide = (p.m & 15u) == 0 && p.m <= 64;
    auto stage_rows = [&](const u8 *cd) {       // cd: a code word (global memory, or the LDS copy of a candidate's)
        WSYNC();
        if (wide) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i * 16 < (int)p.m) {
                    const uint4 w = reinterpret_cast<const uint4 *>(cd)[i];
                    const u32 words[4] = { w.x, w.y, w.z, w.w };
                    float4 r[16];
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        const u32 c = (words[t >> 2] >> (8 * (t & 3))) & 255u;
                        r[t] = reinterpret_cast<const float4 *>(p.sdc + ((size_t)(i * 16 + t) * 256 + c) * 256)[lane];
                    }
#pragma unroll
                    for (int t = 0; t < 16; t++) reinterpret_cast<float4 *>(rows + (size_t)(i * 16 + t) * 256)[lane] = r[t];
                }
            }
        } else {
            for (u32 jq = 0; jq < p.m; jq++)
                reinterpret_cast<float4 *>(rows + (size_t)jq * 256)[lane] =
                    reinterpret_cast<const float4 *>(p.sdc + ((size_t)jq * 256 + cd[jq]) * 256)[lane];
        }
        WSYNC();
    };
    // sum_j rows[j][code[j]], A3's order; `cd` points at a code word (LDS copy of a candidate's, or global)
    auto dist_code = [&](const u8 *cd) {
        float s2 = 0.0f;
        if (wide) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i * 16 < (int)p.m) {
                    const uint4 w = reinterpret_cast<const uint4 *>(cd)[i];
                    const u32 words[4] = { w.x, w.y, w.z, w.w };
                    float tv[16];
#pragma unroll
                    for (int t = 0; t < 16; t++) tv[t] = rows[(i * 16 + t) * 256 + ((words[t >> 2] >> (8 * (t & 3))) & 255u)];
#pragma unroll
                    for (int t = 0; t < 16; t++) s2 = f_add(s2, tv[t]);
                }
            }
        } else {
            for (u32 jq = 0; jq < p.m; jq++) s2 = f_add(s2, rows[jq * 256 + cd[jq]]);
        }
        return s2;
    };
    for (u32 pi = blockIdx.x; pi < p.npoints; pi += gridDim.x) {
        const u32 pt = p.points[pi];
        WSYNC();
        int nraw = 0;
        if (p.res_keys) {
            const int n = (int)p.res_n[pi];
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                u32 id = 0xFFFFFFFFu;
                if (i < n) id = ~(u32)p.res_keys[(size_t)pi * p.cap + i];
                const bool ok = (i < n) && id != pt;
                const u64 mm = __ballot(ok);
                if (ok) raw[nraw + __popcll(mm & lanemask_lt())] = id;
                nraw += __popcll(mm);
            }
        }
        {
            const int dn = (int)min(p.deg[pt], p.RX);
            for (int base = 0; base < dn; base += 64) {
                const int i = base + lane;
                u32 id = 0xFFFFFFFFu;
                if (i < dn) id = p.adjb[(size_t)pt * p.RX + i];
                const bool ok = (i < dn) && id != pt && id != 0xFFFFFFFFu;
                const u64 mm = __ballot(ok);
                if (ok) raw[nraw + __popcll(mm & lanemask_lt())] = id;
                nraw += __popcll(mm);
            }
        }
        if (nraw > DR_PRUNE_PQ_MAXC) nraw = DR_PRUNE_PQ_MAXC;
        stage_rows(p.codes + (size_t)pt * p.m);
        for (int i = lane; i < nraw; i += 64) keyA[i] = ((u64)__float_as_uint(dist_code(p.codes + (size_t)raw[i] * p.m)) << 32) | raw[i];
        WSYNC();
        for (int base = 0; base < nraw; base += 64) {
            const int i = base + lane;
            if (i < nraw) {
                const u64 ki = keyA[i];
                int rank = 0;
                for (int t = 0; t < nraw; t++) { const u64 kt = keyA[t]; rank += (kt < ki || (kt == ki && t < i)) ? 1 : 0; }
                keyB[rank] = ki;
            }
        }
        WSYNC();
        int na = 0;
        for (int base = 0; base < nraw; base += 64) {
            const int i = base + lane;
            const bool ok = (i < nraw) && (i == 0 || keyB[i] != keyB[i - 1]);
            const u64 mm = __ballot(ok);
            if (ok) keyA[na + __popcll(mm & lanemask_lt())] = keyB[i];
            na += __popcll(mm);
        }
        WSYNC();
        // the surviving candidates' code words are cached in LDS once (slot = sorted position); the slot travels with the
        // key through the compactions, so the pick loop reads no code word from memory
        u32 *slotA = raw, *slotB = cslot2;
        for (int i = lane; i < na; i += 64) {
            const u8 *g = p.codes + (size_t)(u32)keyA[i] * p.m;
            if (wide) { for (u32 w4 = 0; w4 < p.m / 16; w4++) reinterpret_cast<uint4 *>(ccode + (size_t)i * p.m)[w4] = reinterpret_cast<const uint4 *>(g)[w4]; }
            else { for (u32 jq = 0; jq < p.m; jq++) ccode[(size_t)i * p.m + jq] = g[jq]; }
            slotA[i] = (u32)i;
        }
        WSYNC();
        u64 *cur = keyA, *nxt = keyB;
        u32 *scur = slotA, *snxt = slotB;
        int nsel = 0;
        while (na > 0 && nsel < (int)p.R) {
            const u32 star = (u32)cur[0];
            if (lane == 0) outsel[nsel] = star;
            nsel++;
            if (na == 1 || nsel >= (int)p.R) break;
            stage_rows(ccode + (size_t)scur[0] * p.m);
            for (int i = 1 + lane; i < na; i += 64) {
                const u64 kc = cur[i];
                keep[i] = (f_mul(p.alpha, dist_code(ccode + (size_t)scur[i] * p.m)) <= key_dist(kc)) ? 0u : 1u;   // pruned when alpha * d(p*, c) <= d(p, c)
            }
            WSYNC();
            int nn = 0;
            for (int base = 1; base < na; base += 64) {
                const int i = base + lane;
                const bool ok = (i < na) && keep[i] != 0u;
                const u64 mm = __ballot(ok);
                if (ok) { const int o = nn + __popcll(mm & lanemask_lt()); nxt[o] = cur[i]; snxt[o] = scur[i]; }
                nn += __popcll(mm);
            }
            WSYNC();
            u64 *t = cur; cur = nxt; nxt = t;
            u32 *ts = scur; scur = snxt; snxt = ts;
            na = nn;
        }
        WSYNC();
        for (int s2 = lane; s2 < (int)p.RX; s2 += 64) p.adjb[(size_t)pt * p.RX + s2] = (s2 < nsel) ? outsel[s2] : 0xFFFFFFFFu;
        if (p.fwd) for (int s2 = lane; s2 < (int)p.R; s2 += 64) p.fwd[(size_t)pi * p.R + s2] = (s2 < nsel) ? outsel[s2] : 0xFFFFFFFFu;
        if (lane == 0) {
            p.deg[pt] = (u32)nsel;
            if (p.fwd_n) p.fwd_n[pi] = (u32)nsel;
        }
    }
}

// ---- C8 scalar kernels (pydiskann/cython_utils.pyx:18-24 l2_distance_fast_cython, :53-70 cosine_similarity_cython) -------
// Row pairs x[i], y[i]: squared L2, and the cosine DISTANCE 1 - dot / (|x| |y|) (0.0 when either norm is 0). One wavefront
// per pair; the reference accumulates in float32 in index order but is compiled -ffast-math (order unpinned, its own
// test allows rtol 1e-5): here the lanes' partial sums are combined by a butterfly.
__global__ __launch_bounds__(64) void scalar_pairs_kernel(const float *__restrict__ x, const float *__restrict__ y, u32 n, u32 D,
                                                          float *__restrict__ out_l2, float *__restrict__ out_cos)
{
    const u32 lane = threadIdx.x;
    for (u32 i = blockIdx.x; i < n; i += gridDim.x) {
        float l2 = 0.0f, dot = 0.0f, nx = 0.0f, ny = 0.0f;
        for (u32 t = lane; t < D; t += 64) {
            const float a = x[(size_t)i * D + t], b = y[(size_t)i * D + t];
            l2 += (a - b) * (a - b); dot += a * b; nx += a * a; ny += b * b;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { l2 += __shfl_xor(l2, o); dot += __shfl_xor(dot, o); nx += __shfl_xor(nx, o); ny += __shfl_xor(ny, o); }
        if (lane == 0) {
            if (out_l2) out_l2[i] = l2;
            if (out_cos) out_cos[i] = (nx == 0.0f || ny == 0.0f) ? 0.0f : (float)(1.0 - (double)dot / (sqrt((double)nx) * sqrt((double)ny)));
        }
    }
}
