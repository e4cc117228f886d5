// build_kernels.hpp -- batched Vamana construction on the GPU (row N1 of SURVEY.md 8f).
//
// What the reference does (pydiskann/cython_utils.pyx:269-369): two passes over a random permutation, alpha = 1
// then alpha; per point: greedy search from the medoid with list size L (:371-433), robust prune of
// (search results U current out-neighbours) to R (:435-492), then for every new out-neighbour add the reverse
// edge and re-prune that neighbour when it exceeds R (:338-357). It is serial and time-seeded (Q15), so large
// graphs cannot be bit-matched; search parity is pinned on reference-built graphs, this builder only has to
// produce a graph of the same kind (recall is what is checked).
//
// GPU form: points are inserted in batches (sizes double up to a cap, so early batches see a graph that already
// contains their predecessors). Per batch:
//   1. search_kernel (exact traversal, result capacity L) with each batch point as the query
//   2. prune_kernel: one wavefront per point; candidates sorted by (distance, id); greedy selection with the
//      alive list compacted in LDS after every pick; writes the forward row
//   3. reverse_edges_kernel: atomic append of p into each selected neighbour's row (rows have R + slack slots)
//   4. prune_kernel again for the rows that grew past R
#pragma once
#include "search_kernel.hpp"

#define DR_PRUNE_MAXC 448   // max candidates per prune: L_build (<= 256) + row slots (<= 192)

struct PruneParams {
    const float *vecp;
    u32 *adjb;            // [N][RX]
    u32 *deg;             // [N]
    u32 RX, R;
    float alpha;
    const u32 *points;    // nodes to prune
    u32 npoints;
    const u64 *res_keys;  // [npoints][cap] search results (dist bits << 32 | ~id) or nullptr
    const u32 *res_n;
    u32 cap;
    u32 *fwd;             // [npoints][R] selected neighbours (PAD padded) for the reverse-edge pass, or nullptr
    u32 *fwd_n;           // [npoints]
    u32 multi;            // 1: the launch uses prune_kernel<D, true> (the multi-pick form, D <= 256 split form)
};

// MULTI: the multi-pick form of step 4 (below) is compiled in -- its own instantiation, because the staged picks' registers
// would cost the plain form occupancy at D = 256.
template <int D, bool MULTI = false> __global__ __launch_bounds__(64) void prune_kernel(const PruneParams p)
{
    constexpr bool QREG = (D <= 256);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qperm = reinterpret_cast<float *>(smem);                                   // D floats when !QREG
    u64 *keyA = reinterpret_cast<u64 *>(smem + (QREG ? 0 : (size_t)D * 4));           // [MAXC]
    u64 *keyB = keyA + DR_PRUNE_MAXC;                                                 // [MAXC]
    u32 *raw = reinterpret_cast<u32 *>(keyB + DR_PRUNE_MAXC);                         // [MAXC]
    u32 *keep = raw + DR_PRUNE_MAXC;                                                  // [MAXC]
    u32 *outsel = keep + DR_PRUNE_MAXC;                                               // [256]
    // multi-pick form: distances of every candidate to the TP staged picks, [TP][MAXC] (only allocated when p.multi)
    constexpr bool MULTI_OK = MULTI && QREG && split_form_ok<D>();
    constexpr int TP = (D <= 128) ? 4 : 2;
    float *epick = reinterpret_cast<float *>(outsel + 256);
    const int lane = lane_id(), j = lane & 7, oct = lane >> 3;

    for (u32 pi = blockIdx.x; pi < p.npoints; pi += gridDim.x) {
        const u32 pt = p.points[pi];
        WSYNC();
        // ---- 1. raw candidate ids: search results, then the current row
        int nraw = 0;
        if (p.res_keys) {
            const int n = (int)p.res_n[pi];
            for (int base = 0; base < n; base += 64) {
                const int i = base + lane;
                u32 id = 0xFFFFFFFFu;
                if (i < n) id = ~(u32)p.res_keys[(size_t)pi * p.cap + i];
                const bool ok = (i < n) && id != pt;
                const u64 m = __ballot(ok);
                if (ok) raw[nraw + __popcll(m & lanemask_lt())] = id;
                nraw += __popcll(m);
            }
        }
        {
            const int dn = (int)min(p.deg[pt], p.RX);
            for (int base = 0; base < dn; base += 64) {
                const int i = base + lane;
                u32 id = 0xFFFFFFFFu;
                if (i < dn) id = p.adjb[(size_t)pt * p.RX + i];
                const bool ok = (i < dn) && id != pt && id != 0xFFFFFFFFu;   // host guarantees cap + RX <= DR_PRUNE_MAXC
                const u64 m = __ballot(ok);
                if (ok) raw[nraw + __popcll(m & lanemask_lt())] = id;
                nraw += __popcll(m);
            }
        }
        if (nraw > DR_PRUNE_MAXC) nraw = DR_PRUNE_MAXC;
        WSYNC();

        // ---- 2. distances to the point itself -> keys (dist bits << 32 | id)
        QueryRegs<D> qreg;
        {
            const float *qp = p.vecp + (size_t)pt * D;
            if constexpr (QREG) load_query_regs<0, D, D>(qp, j, qreg);
            else { for (int i = lane; i < D; i += 64) qperm[i] = qp[i]; WSYNC(); }
        }
        for (int base = 0; base < nraw; base += 8) {
            const int idx = min(base + oct, nraw - 1);
            const u32 id = raw[idx];
            const float e = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)id * D, &qreg, qperm, j);
            if (j == 0 && base + oct < nraw) keyA[base + oct] = ((u64)__float_as_uint(e) << 32) | id;
        }
        WSYNC();

        // ---- 3. rank sort by (distance, id) into keyB, duplicates adjacent; then drop duplicates into keyA
        for (int base = 0; base < nraw; base += 64) {
            const int i = base + lane;
            if (i < nraw) {
                const u64 ki = keyA[i];
                int rank = 0;
                for (int t = 0; t < nraw; t++) {
                    const u64 kt = keyA[t];
                    rank += (kt < ki || (kt == ki && t < i)) ? 1 : 0;
                }
                keyB[rank] = ki;
            }
        }
        WSYNC();
        int na = 0;
        for (int base = 0; base < nraw; base += 64) {
            const int i = base + lane;
            const bool ok = (i < nraw) && (i == 0 || keyB[i] != keyB[i - 1]);
            const u64 m = __ballot(ok);
            if (ok) keyA[na + __popcll(m & lanemask_lt())] = keyB[i];
            na += __popcll(m);
        }
        WSYNC();

        // ---- 4. greedy selection (robust_prune_fast_cython, cython_utils.pyx:459-486)
        u64 *cur = keyA, *nxt = keyB;
        int nsel = 0;
        if constexpr (MULTI_OK) {
            // Multi-pick form (same picks, same order): the sequential form streams every remaining candidate's row once per
            // pick, and the kernel runs at the memory system's limit doing so. Here the first TP remaining candidates -- the
            // only ones that can become the next picks -- are staged in registers and ONE pass over the rows scores every
            // candidate against all of them; the picks are then resolved from those distances, one after the other, until the
            // next survivor is not a staged one.
            while (na > 0 && nsel < (int)p.R) {
                const int S = min(TP, na);
                QueryRegs<D> sq[TP];
#pragma unroll
                for (int t = 0; t < TP; t++) if (t < S) load_query_regs<0, D, D>(p.vecp + (size_t)(u32)cur[t] * D, j, sq[t]);
                for (int base = 1; base < na; base += 8) {
                    const int idx = min(base + oct, na - 1);
                    RowRegs<D> rr;
                    row_load<0, D, D>(p.vecp + (size_t)(u32)cur[idx] * D, j, rr);
#pragma unroll
                    for (int t = 0; t < TP; t++) {
                        if (t < S) {
                            const float e = row_reduce<0, D, D>(rr, sq[t]);
                            if (j == 0 && base + oct < na) epick[t * DR_PRUNE_MAXC + idx] = e;
                        }
                    }
                }
                for (int i = lane; i < na; i += 64) keep[i] = 1u;
                WSYNC();
                int t = 0, restart = -1;      // t: position in cur of the pick being applied; restart: first survivor that is not staged
                for (;;) {
                    if (lane == 0) outsel[nsel] = (u32)cur[t];
                    nsel++;
                    if (nsel >= (int)p.R) { na = 0; break; }
                    int nx = -1;              // first survivor after t
                    for (int base = t + 1; base < na; base += 64) {
                        const int i = base + lane;
                        bool alive = false;
                        if (i < na) {
                            alive = keep[i] != 0u;
                            if (alive && f_mul(p.alpha, epick[t * DR_PRUNE_MAXC + i]) <= key_dist(cur[i])) { alive = false; keep[i] = 0u; }   // pruned when alpha * d(p*, c) <= d(p, c)
                        }
                        const u64 mm = __ballot(alive);
                        if (nx < 0 && mm) nx = base + __builtin_ctzll(mm);
                    }
                    WSYNC();
                    if (nx < 0) { na = 0; break; }
                    if (nx < S) { t = nx; continue; }
                    restart = nx;
                    break;
                }
                if (restart >= 0) {           // survivors from `restart` on, in order, become the new list
                    int nn = 0;
                    for (int base = restart; base < na; base += 64) {
                        const int i = base + lane;
                        const bool ok = (i < na) && keep[i] != 0u;
                        const u64 mm = __ballot(ok);
                        if (ok) nxt[nn + __popcll(mm & lanemask_lt())] = cur[i];
                        nn += __popcll(mm);
                    }
                    WSYNC();
                    u64 *tq = cur; cur = nxt; nxt = tq;
                    na = nn;
                }
            }
        }
        while (na > 0 && nsel < (int)p.R) {
            const u64 k0 = cur[0];
            const u32 star = (u32)k0;
            if (lane == 0) outsel[nsel] = star;
            nsel++;
            if (na == 1 || nsel >= (int)p.R) break;
            {
                const float *qp = p.vecp + (size_t)star * D;
                if constexpr (QREG) load_query_regs<0, D, D>(qp, j, qreg);
                else { WSYNC(); for (int i = lane; i < D; i += 64) qperm[i] = qp[i]; WSYNC(); }
            }
            const int nrem = na - 1;
            for (int base = 0; base < nrem; base += 8) {
                const int idx = 1 + min(base + oct, nrem - 1);
                const u64 kc = cur[idx];
                const float e = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)(u32)kc * D, &qreg, qperm, j);
                // pruned when alpha * d(p*, c) <= d(p, c)
                if (j == 0 && base + oct < nrem) keep[idx] = (f_mul(p.alpha, e) <= key_dist(kc)) ? 0u : 1u;
            }
            WSYNC();
            int nn = 0;
            for (int base = 1; base < na; base += 64) {
                const int i = base + lane;
                const bool ok = (i < na) && keep[i] != 0u;
                const u64 m = __ballot(ok);
                if (ok) nxt[nn + __popcll(m & lanemask_lt())] = cur[i];
                nn += __popcll(m);
            }
            WSYNC();
            u64 *t = cur; cur = nxt; nxt = t;
            na = nn;
        }
        WSYNC();

        // ---- 5. write the row
        for (int s = lane; s < (int)p.RX; s += 64) p.adjb[(size_t)pt * p.RX + s] = (s < nsel) ? outsel[s] : 0xFFFFFFFFu;
        if (p.fwd) for (int s = lane; s < (int)p.R; s += 64) p.fwd[(size_t)pi * p.R + s] = (s < nsel) ? outsel[s] : 0xFFFFFFFFu;
        if (lane == 0) {
            p.deg[pt] = (u32)nsel;
            if (p.fwd_n) p.fwd_n[pi] = (u32)nsel;
        }
    }
}
