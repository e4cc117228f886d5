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

// 16 terms of a centroid-pair sum from table rows held in registers: tv[jj * 4 + k] of lane l = row_jj[l * 4 + k] (the lane's
// float4 of the 1 KiB row), the entry of code c is register c & 3 of lane c >> 2 (ds_bpermute; every lane takes part)
DEV float sdc_reg16(const float *tv, const uint4 cw, float s)
{
    const u32 words[4] = { cw.x, cw.y, cw.z, cw.w };
    float t[16];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int jj = u * 4 + b;
            const u32 c = (words[u] >> (8 * b)) & 255u;
            const int addr = (int)(c & 252u);                       // lane (c >> 2), in bytes
            const u32 r0 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 0]));
            const u32 r1 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 1]));
            const u32 r2 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 2]));
            const u32 r3 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 3]));
            const u32 lo = (c & 1u) ? r1 : r0, hi = (c & 1u) ? r3 : r2;
            t[jj] = __uint_as_float((c & 2u) ? hi : lo);
            if (b == 3) __builtin_amdgcn_sched_barrier(0);         // (16 lookups in flight, not 128: the rows take half the registers)
        }
#pragma unroll
    for (int i = 0; i < 16; i++) s = f_add(s, t[i]);
    return s;
}

// prune_kernel's twin on code words: one wavefront per point, the m table rows of the point (then of each pick) staged, one
// lane per candidate for the m-term sums. M16 = m / 16 in {1, 2}: the rows live in REGISTERS (64 per 16 sub-quantisers,
// lookups by ds_bpermute) and the block's LDS holds only the candidate lists: 8 wavefronts per CU instead of 2 (the kernel
// waits on the row loads of pick after pick; its rate is the number of points in flight). M16 = 0: any m, rows in LDS.
// Same sums in the same order either way: the graph does not depend on the form.
template <int M16>
__global__ __launch_bounds__(64, M16 == 2 ? 2 : 1) void prune_pq_kernel(const PrunePQParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KC = DR_PRUNE_PQ_MAXC;                                              // candidates per prune (nraw is clipped to it)
    float *rows = reinterpret_cast<float *>(smem);                                    // [m][256] (M16 == 0)
    u64 *keyA = reinterpret_cast<u64 *>(smem + (M16 ? 0 : (size_t)p.m * 1024));       // [KC]
    u64 *keyB = keyA + KC;
    u32 *raw = reinterpret_cast<u32 *>(keyB + KC);
    u32 *outsel = raw + KC;                                                           // [256]
    u32 *cslot2 = outsel + 256;                                                       // [KC] second slot array of the compaction
    u8 *ccode = reinterpret_cast<u8 *>(cslot2 + KC);                                  // [KC][m] candidates' code words
    const int lane = lane_id();
    float tv[M16 ? M16 * 64 : 1];
    // (code words are read as whole 16-byte pieces, and every loop over their bytes is unrolled: the m row loads / table
    // reads that depend on a piece are issued together, and no register array is indexed dynamically)
    const bool wide = (p.m & 15u) == 0 && p.m <= 64;
    auto stage_rows = [&](const u8 *cd) {       // cd: ONE code word for the whole wavefront (global memory, or the LDS copy of a candidate's)
        WSYNC();
        if constexpr (M16 > 0) {
#pragma unroll
            for (int i = 0; i < M16; i++) {
                const uint4 w = reinterpret_cast<const uint4 *>(cd)[i];
                const u32 words[4] = { (u32)__builtin_amdgcn_readfirstlane((int)w.x), (u32)__builtin_amdgcn_readfirstlane((int)w.y),
                                       (u32)__builtin_amdgcn_readfirstlane((int)w.z), (u32)__builtin_amdgcn_readfirstlane((int)w.w) };
#pragma unroll
                for (int t = 0; t < 16; t++) {
                    const u32 c = (words[t >> 2] >> (8 * (t & 3))) & 255u;
                    const float4 r = reinterpret_cast<const float4 *>(p.sdc + ((size_t)(i * 16 + t) * 256 + c) * 256)[lane];
                    tv[(i * 16 + t) * 4 + 0] = r.x; tv[(i * 16 + t) * 4 + 1] = r.y; tv[(i * 16 + t) * 4 + 2] = r.z; tv[(i * 16 + t) * 4 + 3] = r.w;
                }
            }
        } else if (wide) {
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
    // sum_j rows[j][code[j]], A3's order; `cd` points at the lane's code word (LDS copy of a candidate's, or global). With the
    // rows in registers EVERY lane has to call it (a lane without a candidate passes any valid code word).
    auto dist_code = [&](const u8 *cd) {
        float s2 = 0.0f;
        if constexpr (M16 > 0) {
#pragma unroll
            for (int i = 0; i < M16; i++) s2 = sdc_reg16(tv + i * 64, reinterpret_cast<const uint4 *>(cd)[i], s2);
        } else if (wide) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (i * 16 < (int)p.m) {
                    const uint4 w = reinterpret_cast<const uint4 *>(cd)[i];
                    const u32 words[4] = { w.x, w.y, w.z, w.w };
                    float tt[16];
#pragma unroll
                    for (int t = 0; t < 16; t++) tt[t] = rows[(i * 16 + t) * 256 + ((words[t >> 2] >> (8 * (t & 3))) & 255u)];
#pragma unroll
                    for (int t = 0; t < 16; t++) s2 = f_add(s2, tt[t]);
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
        for (int base = 0; base < nraw; base += 64) {       // (whole wavefront in every trip: see dist_code)
            const int i = base + lane;
            const u32 id = (i < nraw) ? raw[i] : pt;
            const float dpc = dist_code(p.codes + (size_t)id * p.m);
            keyA[i] = ((u64)__float_as_uint(dpc) << 32) | id;       // (i < KC, a multiple of 64: slots past nraw are never read)
        }
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
            // distance of every remaining candidate to the pick and the compaction of the survivors in one sweep (the
            // ballot also keeps the sums out of a branch: the looked-up values are consumed as they arrive)
            int nn = 0;
            for (int base = 1; base < na; base += 64) {
                const int i = base + lane;
                const bool have = i < na;
                const u64 kc = cur[have ? i : 0];
                const u32 sl = scur[have ? i : 0];
                const float dsc = dist_code(ccode + (size_t)sl * p.m);
                const bool ok = have && !(f_mul(p.alpha, dsc) <= key_dist(kc));          // pruned when alpha * d(p*, c) <= d(p, c)
                const u64 mm = __ballot(ok);
                if (ok) { const int o = nn + __popcll(mm & lanemask_lt()); nxt[o] = kc; snxt[o] = sl; }
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
