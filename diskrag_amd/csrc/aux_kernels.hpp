// aux_kernels.hpp -- per-dimension template kernels besides the search kernel: exact distances for listed nodes
// and brute-force top-k (recall ground truth). Instantiated in search_d<D>.hip.
#pragma once
#include "search_kernel.hpp"

// ---- kernel-level entry points ----------------------------------------------------------------------------
// exact squared distances out[q][i] for node_ids[i]; one wave scores 8 nodes per pass (same device function as
// the search kernel).
template <int D> __global__ __launch_bounds__(64) void exact_kernel(const float *__restrict__ vecp,
        const float *__restrict__ queries_p, u32 nq, const u32 *__restrict__ ids, u32 n, float *__restrict__ out)
{
    constexpr bool QREG = (D <= 256);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qperm = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x & 63, j = lane & 7, oct = lane >> 3;
    const u32 q = blockIdx.y;
    const float *qpg = queries_p + (size_t)q * D;
    QueryRegs<D> qreg;
    if constexpr (QREG) load_query_regs<0, D, D>(qpg, j, qreg);
    else { for (int i = lane; i < D; i += 64) qperm[i] = qpg[i]; }
    WSYNC();
    for (u32 base = blockIdx.x * 8; base < n; base += gridDim.x * 8) {
        const u32 idx = min(base + (u32)oct, n - 1);
        const float e = pw_row_stream<0, D, D, QREG>(vecp + (size_t)ids[idx] * D, &qreg, qperm, j);
        if (j == 0 && base + oct < n) out[(size_t)q * n + base + oct] = e;
    }
}

// DR_MODE_PQ + DR_F_RERANK (config c3: "PQ traversal + full-precision rerank of the L list"): exact
// squared L2 (A1 order, the same device function as everywhere) of every entry of a query's final result list, then the
// k best in (distance, id) order. One WORKGROUP per query -- one wavefront for batches that fill the chip, eight for the handful of
// queries of a request (round 5: a lone wavefront streams the 100 rows of an L = 100 list pass after pass, 13 passes of 6-KiB rows at D = 1536;
// eight take two each) --, 8 stored vectors per wavefront pass; ranks by counting in LDS.
template <int D> __global__ __launch_bounds__(512) void rerank_kernel(const float *__restrict__ vecp,
        const float *__restrict__ queries_p, u32 nq, const u64 *__restrict__ res_keys, const u32 *__restrict__ res_n, u32 cap,
        u32 k, u32 *__restrict__ out_ids, float *__restrict__ out_dist, u32 *__restrict__ out_count, KStats *__restrict__ stats, u32 ip,
        const float *__restrict__ queries, const u32 *__restrict__ perm, const u32 *__restrict__ id_map, u32 top)
{
    // (id_map != nullptr: the disk tier -- `vecp` holds only the rows of this batch's lists, fetched from index.dat: the keys carry a row's
    //  position in that buffer and id_map gives the node it is; distances are ranked with the NODE id, as everywhere)
    // (queries_p == nullptr: no chain-major copy of the batch was made -- the handful of queries of a request; the element permutation is applied
    //  here from `queries` / `perm`, one launch less in front of the answer)
    // ip (DR_F_IP): unit-norm data, out_dist = |q - v|^2 / 2 = 1 - <q, v>; a query whose squared norm is not 1 (+- 1e-3) gets NaN + status bit 4
    constexpr bool QREG = (D <= 256);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qperm = reinterpret_cast<float *>(smem);
    u64 *keys = reinterpret_cast<u64 *>(smem + (QREG ? 0 : (size_t)D * 4));
    const int lane = threadIdx.x & 63, j = lane & 7, oct = lane >> 3;
    const u32 wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    for (u32 q = blockIdx.x; q < nq; q += gridDim.x) {
        const float *qpg = (queries_p ? queries_p : queries) + (size_t)q * D;      // (the ip check below sums squares: any element order)
        QueryRegs<D> qreg;
        __syncthreads();
        if constexpr (QREG) { if (queries_p) load_query_regs<0, D, D>(qpg, j, qreg); else load_query_regs_orig<0, D, D>(qpg, j, qreg); }
        else if (queries_p) { for (int i = threadIdx.x; i < D; i += blockDim.x) qperm[i] = qpg[i]; }
        else { for (int i = threadIdx.x; i < D; i += blockDim.x) qperm[perm[i]] = qpg[i]; }
        __syncthreads();
        const u32 n = min(min(res_n[q], cap), top ? top : cap);      // (DR_F_RERANK_TOP: the list is in (ADC, id) order -- its first `top` entries)
        const u64 *rk = res_keys + (size_t)q * cap;
        for (u32 base = wave * 8; base < n; base += 8 * nwv) {
            const u32 idx = min(base + (u32)oct, n - 1);
            const u32 rowi = ~(u32)rk[idx];
            const u32 id = id_map ? id_map[rowi] : rowi;
            const float e = pw_row_stream<0, D, D, QREG>(vecp + (size_t)rowi * D, &qreg, qperm, j);
            if (j == 0 && base + oct < n) keys[base + oct] = ((u64)__float_as_uint(e) << 32) | id;   // e >= 0: bits order = value order
        }
        __syncthreads();
        const u32 kout = min(k, n);
        bool q_unit = true;
        if (ip) {
            double s2 = 0.0;
            for (int i = lane; i < D; i += 64) s2 += (double)qpg[i] * (double)qpg[i];      // (a permutation of the query: the same sum)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
            q_unit = s2 > 1.0 - 1e-3 && s2 < 1.0 + 1e-3;
        }
        for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
            const u64 mine = keys[i];
            u32 r = 0;
            for (u32 t = 0; t < n; t++) r += (keys[t] < mine) ? 1u : 0u;        // ids are distinct: a total order
            if (r < kout) {
                out_ids[(size_t)q * k + r] = (u32)mine;
                out_dist[(size_t)q * k + r] = !ip ? key_dist(mine) : q_unit ? f_mul(key_dist(mine), 0.5f) : __uint_as_float(0x7FC00000u);
            }
        }
        for (u32 i = kout + threadIdx.x; i < k; i += blockDim.x) { out_ids[(size_t)q * k + i] = 0xFFFFFFFFu; out_dist[(size_t)q * k + i] = __uint_as_float(0x7FC00000u); }
        if (threadIdx.x == 0) { out_count[q] = kout; stats[q].exact += n; if (!q_unit) stats[q].status |= 16u; }
    }
}

// brute-force exact top-k (recall ground truth). One wave per query streams all N stored vectors, 8 per pass,
// and keeps the k best in LDS (k <= 64), ties broken towards the smaller id.
// With `part` != nullptr the rows are cut into gridDim.y slices (small batches on large tables: one wavefront per query
// leaves the chip empty): block (q, s) scans rows [s*N/S, (s+1)*N/S) and leaves its k keys (distance bits << 32 | id,
// ~0 = empty) in part[q][s][k]; topk_merge_kernel (engine_kernels.hpp) folds the slices.
template <int D> __global__ __launch_bounds__(64) void bruteforce_kernel(const float *__restrict__ vecp, u64 N,
        const float *__restrict__ queries_p, u32 nq, u32 k, u32 *__restrict__ out_ids, float *__restrict__ out_dist,
        u64 *__restrict__ part)
{
    constexpr bool QREG = (D <= 256);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qperm = reinterpret_cast<float *>(smem);
    u64 *best = reinterpret_cast<u64 *>(smem + (QREG ? 0 : (size_t)D * 4));
    const int lane = threadIdx.x & 63, j = lane & 7, oct = lane >> 3;
    const u32 q = blockIdx.x;
    if (q >= nq) return;
    const float *qpg = queries_p + (size_t)q * D;
    QueryRegs<D> qreg;
    if constexpr (QREG) load_query_regs<0, D, D>(qpg, j, qreg);
    else { for (int i = lane; i < D; i += 64) qperm[i] = qpg[i]; }
    if (lane < (int)k) best[lane] = ~0ull;
    WSYNC();
    int bn = 0;
    float W = __uint_as_float(0x7F800000u);
    const u64 row_lo = part ? (N * blockIdx.y) / gridDim.y : 0ull;
    const u64 row_hi = part ? (N * (blockIdx.y + 1)) / gridDim.y : N;
    for (u64 base = row_lo; base < row_hi; base += 8) {
        const u64 row = base + oct < row_hi ? base + oct : row_hi - 1;
        const float e = pw_row_stream<0, D, D, QREG>(vecp + row * D, &qreg, qperm, j);
        const bool cand = (j == 0) && (base + oct < row_hi) && (bn < (int)k || e < W);
        u64 cm = __ballot(cand);
        while (cm) {
            const int f = __ffsll((long long)cm) - 1;
            cm &= cm - 1;
            const float ef = __shfl(e, f);
            const u32 idf = (u32)(base + (f >> 3));
            if (bn < (int)k || ef < W) {
                const u64 key = ((u64)__float_as_uint(ef) << 32) | idf;
                // insert into ascending list of at most k keys (one lane per slot)
                const u64 mine = (lane < bn) ? best[lane] : ~0ull;
                const int pos = __popcll(__ballot(lane < bn && mine < key));
                WSYNC();
                if (lane < bn && lane >= pos && lane + 1 < (int)k) best[lane + 1] = mine;
                if (lane == 0 && pos < (int)k) best[pos] = key;
                WSYNC();
                if (bn < (int)k) bn++;
                if (bn == (int)k) W = key_dist(best[k - 1]);
            }
        }
    }
    WSYNC();
    if (lane < (int)k) {
        const u64 key = lane < bn ? best[lane] : ~0ull;
        if (part) { part[((size_t)q * gridDim.y + blockIdx.y) * k + lane] = key; return; }
        out_ids[(size_t)q * k + lane] = lane < bn ? (u32)key : 0xFFFFFFFFu;
        out_dist[(size_t)q * k + lane] = lane < bn ? key_dist(key) : __uint_as_float(0x7FC00000u);
    }
}

// Nearest of P pivot rows for every stored vector (a coarse cluster label; used only to give the visited bitmap a
// locality-preserving bit order, see engine.hip build_bit_order: any labelling is correct, a good one is fast).
// One wavefront per stored vector, persistent over the rows; the pivots (P*D*4 bytes) stay L2-resident.
template <int D> __global__ __launch_bounds__(64) void nearest_pivot_kernel(const float *__restrict__ vecp, u64 N,
        const float *__restrict__ pivots_p, u32 P, u32 *__restrict__ label)
{
    constexpr bool QREG = (D <= 256);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qperm = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x & 63, j = lane & 7, oct = lane >> 3;
    for (u64 row = blockIdx.x; row < N; row += gridDim.x) {
        const float *qpg = vecp + row * D;
        QueryRegs<D> qreg;
        if constexpr (QREG) load_query_regs<0, D, D>(qpg, j, qreg);
        else { WSYNC(); for (int i = lane; i < D; i += 64) qperm[i] = qpg[i]; WSYNC(); }
        float best = __uint_as_float(0x7F800000u);
        u32 besti = 0;
        for (u32 base = 0; base < P; base += 8) {
            const u32 pi = min(base + (u32)oct, P - 1);
            const float e = pw_row_stream<0, D, D, QREG>(pivots_p + (size_t)pi * D, &qreg, qperm, j);
            if (e < best) { best = e; besti = pi; }
        }
        // every lane of an octet holds the octet's result: reduce over the 8 octets
        u64 key = ((u64)__float_as_uint(best) << 32) | besti;
#pragma unroll
        for (int sh = 8; sh < 64; sh <<= 1) { const u64 o = __shfl_xor(key, sh); key = o < key ? o : key; }
        if (lane == 0) label[row] = (u32)key;
    }
}
