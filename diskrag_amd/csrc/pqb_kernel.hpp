// pqb_kernel.hpp -- DR_MODE_PQB: the engine's batch-per-step PQ-only beam search (round 5). One 64-lane wavefront per query,
// persistent wavefronts, independent of the vector dimension (an ADC-only traversal never touches a stored vector).
//
// No reference counterpart: the reference's only PQ-only traversal, beam_search_with_pq (pydiskann/vamana_graph.py:535-605),
// keeps a k-sized heap and trims its frontier from the wrong end (quirk Q9); it is served bit-exactly as DR_MODE_M3. What IS the
// reference's: the table T = compute_distance_table(q) (pq/fast_pq.py:294-318, built by lut_build_kernel in A2's order) and the
// distance s = 0f; s += T[j][code_j] in strict order of j (asymmetric_distance_sq, fast_pq.py:320-328). The traversal is SURVEY.md
// 8a row E's batch-per-expansion form; the test suite holds it bit for bit to a plain CPU restatement of these rules (tests/test_gpu_pqb.py):
//
//   key(i) = (bits of the squared ADC of node i) << 32 | i        a TOTAL order: nothing below depends on evaluation order
//   list   = at most L keys, ascending, each entry live or not (expanded / trimmed)
//   step   = the p = min(pops, #live, max_steps - steps) smallest live entries are marked expanded; every neighbour slot of their
//            rows that is the first occurrence of its id in its row is scored; a key enters the candidate SET iff (the list is
//            not full or key < the list's largest key) and it is neither in the list nor in the set already (all against the
//            list as the step found it); list = the L smallest of list + set, newcomers live; beam_width > 0: only the
//            beam_width smallest live entries stay live (heapq.nsmallest on the frontier, search_engine.py:477-479)
//   stop   = no live entry, or min(10 L, N) expanded nodes (search_engine.py:429)
//
// There is no visited set (a node scored before and not in the list now was rejected or evicted at a largest key >= today's: it
// can never enter again, so membership in the list is the whole test) and therefore no sequential walk to emulate: where
// DR_MODE_PQ spends half of its instructions deciding what the reference's neighbour-by-neighbour loop would have done (three
// binary searches over the list, candidate masks, a fixed point, the insert log for the tie-order pass), a step here is
//   pop (scalar bit operations on the live masks) -> rows -> code words -> ADC -> one ballot against the largest key ->
//   per surviving candidate: its rank in the list by wave-wide compares -> ONE scatter / gather merge through LDS.
// Expanding several frontier entries per step (pops; DiskANN's beam) fills the 64 lanes when rows are narrow (R = 32: two rows
// per ADC pass) and halves the number of DEPENDENT memory round trips per query.
//
// Where a query's state lives: list keys in VGPRs (one per lane per 64-entry chunk), live masks in SGPRs (one 64-bit mask per
// chunk: pop and trim are scalar code), the per-query table split between LDS (rows 0 .. m-TREG-1) and VGPRs (the last TREG rows,
// looked up with ds_bpermute: search_kernel.hpp adc_reg16), the merge scratch in LDS.
#pragma once
#include "search_kernel.hpp"

struct PqbParams {
    const u32 *adj;          // [N][R]
    const u64 *first;        // [N][ceil(R/64)] bit s: slot s holds a real id, first occurrence in its row (first_mask_kernel)
    const u8 *codes;         // [N][m]
    const u8 *nbcodes;       // [N][R][m] inline neighbour codes, or nullptr
    const float *lut_g;      // [nq][m][256] the batch's tables (lut_build_kernel)
    u64 N;
    u32 R, m, medoid, nq, k, cap, bw, pops, max_steps;
    u32 rs_shift;            // log2 of the lane stride of a row inside a step: next_pow2(R)
    u32 *counter;            // query ticket counter (monotonic, as search_kernel)
    u32 ticket_base;
    u64 *res_keys;           // [nq][cap] ascending (dist bits << 32 | ~id): what rerank_kernel reads
    u32 *res_n;
    KStats *stats;
    u32 *out_ids;            // [nq][k]
    float *out_dist;         // [nq][k]
    u32 *out_count;
};

#define PQB_NOTLIVE 0x80000000ull      // bit 31 of the id word carries "not live" through the merge scratch (N < 2^31)

// squared ADC of one code word for a compile-time m = 16 * M16, rows 0 .. m-TREG-1 from LDS, the last TREG from registers
// (tv[jj*4 + v] of lane l = T[m-TREG+jj][64 v + l]); strict order of j. Whole-wave call (ds_bpermute).
template <int M16, int TREG>
DEV float pqb_adc(const float *lut, const float (&tv)[TREG > 0 ? TREG * 4 : 1], const uint4 (&cw)[M16])
{
    constexpr int M = M16 * 16, ML = M - TREG;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < M16; w++) {
        u32 words[4] = { cw[w].x, cw[w].y, cw[w].z, cw[w].w };
        // four sub-quantisers (one code word) at a time: their lookups (4 LDS reads, or 16 ds_bpermute) in flight together, then
        // the strict sum. The empty asm makes the next word's decoding wait for this sum: left alone the compiler issues all
        // 64 permutes of a piece first and keeps their results (and the 32 decoded indices and select masks) live at once.
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (w + g > 0) asm volatile("" : "+v"(words[g]) : "v"(s));
            float t[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int jq = w * 16 + g * 4 + b;
                const u32 c = (words[g] >> (8 * b)) & 255u;
                if (jq < ML) t[b] = lut[jq * 256 + c];
                else {
                    const int jj = jq - ML;
                    const int addr = (int)((c & 63u) << 2);
                    const u32 r0 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 0]));
                    const u32 r1 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 1]));
                    const u32 r2 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 2]));
                    const u32 r3 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 3]));
                    const u32 lo = (c & 64u) ? r1 : r0, hi = (c & 64u) ? r3 : r2;
                    t[b] = __uint_as_float((c & 128u) ? hi : lo);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; b++) s = f_add(s, t[b]);
        }
    }
    return s;
}
// any m (M16 = 0): whole table in LDS, bytes read one by one from global memory
DEV float pqb_adc_generic(const float *lut, const u8 *__restrict__ code, u32 m)
{
    float s = 0.0f;
    if ((m & 3u) == 0) {
        const u32 *c4 = reinterpret_cast<const u32 *>(code);
#pragma unroll 1
        for (u32 w = 0; w < m / 4; w++) s = adc_from_lut(lut, c4[w], w * 4, s);
    } else {
#pragma unroll 1
        for (u32 jq = 0; jq < m; jq++) s = f_add(s, lut[jq * 256 + code[jq]]);
    }
    return s;
}

// NCHR  list capacity in 64-entry chunks          NC    64-lane passes per step (>= ceil(pops * next_pow2(R) / 64))
// M16   m / 16 (0: any m, generic ADC)            TREG  table rows held in registers
template <int NCHR, int NC, int M16, int TREG>
__global__ __launch_bounds__(64, (TREG >= 24) ? 3 : 2) void pqb_search_kernel(const PqbParams p)
{
    static_assert(TREG == 0 || (M16 > 0 && TREG <= M16 * 16 && TREG % 8 == 0), "register rows need a compile-time m");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const u32 m_lds = p.m - (u32)TREG;
    float *lut = reinterpret_cast<float *>(smem);
    u64 *mk = reinterpret_cast<u64 *>(smem + (size_t)m_lds * 1024);           // merge scratch [NCHR * 64]
    const u32 slot_id = (u32)__builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const u32 nslots = gridDim.x;
    const int cap = (int)p.cap;
    const u32 nwords = (p.R + 63) / 64;
    const u32 rs_mask = (1u << p.rs_shift) - 1u;

    u32 qi = slot_id;
    for (u32 round = 0; round < p.nq && qi < p.nq; ++round) {
        // ---- the query's table: m KiB built by lut_build_kernel, landed 1 KiB per wave instruction; the last TREG rows to registers
        float tv[TREG > 0 ? TREG * 4 : 1];
        {
            const float *tg = p.lut_g + (size_t)qi * p.m * 256 + lane * 4;
            for (u32 e = 0; e < m_lds * 256; e += 256)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tg + e),
                    (__attribute__((address_space(3))) void *)(lut + e), 16, 0, 0);
            if constexpr (TREG > 0) {
                const float *tr = p.lut_g + ((size_t)qi * p.m + m_lds) * 256 + lane;
#pragma unroll
                for (int i = 0; i < TREG * 4; i++) tv[i] = tr[i * 64];
            } else tv[0] = 0.0f;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WSYNC();
        }
        RegList<NCHR> rk;
        u64 live[NCHR];
#pragma unroll
        for (int c = 0; c < NCHR; c++) { rk.v[c] = ~0ull; live[c] = 0ull; }
        int rn = 0;
        u32 steps = 0, nevals = 0, nins = 0, status = 0;

        // ---- steps. The first one is the SEED: nothing is popped, lane 0 of pass 0 scores the start node (the one copy of the
        // ADC code serves both) and the merge puts it into the empty list.
        bool seed = true;
        for (;;) {
            int np = 0;
            u32 mypop = 0u;
            if (!seed) {
                int nlive = 0;
#pragma unroll
                for (int c = 0; c < NCHR; c++) nlive += __popcll(live[c]);
                if (nlive == 0 || steps >= p.max_steps) break;
                if ((u64)steps > p.N + 64) { status |= DR_ST_INTERNAL; break; }
                np = min(min((int)p.pops, nlive), (int)(p.max_steps - steps));
                // pop: the np smallest live entries (scalar bit operations); lane i keeps the id of pop i
#pragma unroll 1
                for (int i = 0; i < np; i++) {
                    int pos = -1;
#pragma unroll
                    for (int c = 0; c < NCHR; c++) {
                        if (pos < 0 && live[c] != 0ull) { pos = c * 64 + __ffsll((long long)live[c]) - 1; live[c] &= live[c] - 1ull; }
                    }
                    const u32 id = (u32)list_get<NCHR>(rk, pos);
                    mypop = (lane == i) ? id : mypop;
                }
                steps += (u32)np;
            }

            // rows: pass t covers lanes g = 64 t + lane of the step; row ni = g >> rs_shift, slot = g & rs_mask
            u32 nid[NC]; u64 fw[NC]; bool valid[NC]; u32 slot_[NC], cur_[NC];
#pragma unroll
            for (int t = 0; t < NC; t++) {
                const u32 g = (u32)(t * 64 + lane);
                const u32 ni = g >> p.rs_shift;
                const u32 slot = g & rs_mask;
                const u32 sl = min(slot, p.R - 1);
                slot_[t] = sl;
                if (seed) {
                    valid[t] = (g == 0u); cur_[t] = 0u; nid[t] = p.medoid; fw[t] = 1ull;
                } else {
                    valid[t] = ni < (u32)np && slot < p.R;
                    const u32 cur = (u32)__builtin_amdgcn_ds_bpermute((int)(min(ni, (u32)max(np, 1) - 1u) << 2), (int)mypop);
                    cur_[t] = cur;
                    nid[t] = p.adj[(size_t)cur * p.R + sl];
                    fw[t] = p.first[(size_t)cur * nwords + (sl >> 6)];
                }
            }
            // code words (inline: beside the row, no dependency on the ids; else a gather behind them)
            uint4 cw[NC][M16 > 0 ? M16 : 1];
            bool act[NC];
#pragma unroll
            for (int t = 0; t < NC; t++) {
                act[t] = valid[t] && ((fw[t] >> (slot_[t] & 63u)) & 1ull) != 0ull;
                if constexpr (M16 > 0) {
                    const u8 *code = (p.nbcodes && !seed) ? p.nbcodes + ((size_t)cur_[t] * p.R + slot_[t]) * p.m : p.codes + (size_t)(act[t] ? nid[t] : 0u) * p.m;
#pragma unroll
                    for (int w = 0; w < M16; w++) cw[t][w] = reinterpret_cast<const uint4 *>(code)[w];
                }
            }
            const bool full = (rn == cap);
            const u64 wk = full ? list_get<NCHR>(rk, cap - 1) : ~0ull;
            u64 key[NC], cm[NC];
#pragma unroll
            for (int t = 0; t < NC; t++) {
                float e;
                if constexpr (M16 > 0) e = pqb_adc<M16, TREG>(lut, tv, cw[t]);
                else {
                    const u8 *code = (p.nbcodes && !seed) ? p.nbcodes + ((size_t)cur_[t] * p.R + slot_[t]) * p.m : p.codes + (size_t)(act[t] ? nid[t] : 0u) * p.m;
                    e = pqb_adc_generic(lut, code, p.m);
                }
                key[t] = ((u64)__float_as_uint(e) << 32) | nid[t];
                nevals += (u32)__popcll(__ballot(act[t]));
                cm[t] = __ballot(act[t] && key[t] < wk);
            }

            // candidates one by one: dropped if in the list (or met before in this step); else its rank in the list, and every
            // lane counts the accepted keys below its own key / its list keys (the merge positions)
            u64 acc[NC];
            u32 rT[NC], rA[NC], sT[NCHR];
#pragma unroll
            for (int t = 0; t < NC; t++) { acc[t] = 0ull; rT[t] = 0u; rA[t] = 0u; }
#pragma unroll
            for (int c = 0; c < NCHR; c++) sT[c] = 0u;
            int nacc = 0;
#pragma unroll
            for (int t = 0; t < NC; t++) {
#pragma unroll 1
                while (cm[t] != 0ull) {
                    const int f = __ffsll((long long)cm[t]) - 1;
                    cm[t] &= cm[t] - 1ull;
                    const u64 kf = readlane64(key[t], f);
                    u64 eq = 0ull;
                    int clt = 0;
#pragma unroll
                    for (int c = 0; c < NCHR; c++) {
                        eq |= __ballot(rk.v[c] == kf);
                        clt += __popcll(__ballot(rk.v[c] < kf));
                    }
                    if (np > 1) {           // the same node through two of the step's rows: the later copies leave the set
#pragma unroll
                        for (int u = t; u < NC; u++) cm[u] &= ~__ballot(key[u] == kf);
                    }
                    if (eq != 0ull) continue;
                    acc[t] |= 1ull << f;
                    nacc++;
                    rT[t] = (lane == f) ? (u32)clt : rT[t];
#pragma unroll
                    for (int u = 0; u < NC; u++) rA[u] += (kf < key[u]) ? 1u : 0u;
#pragma unroll
                    for (int c = 0; c < NCHR; c++) sT[c] += (kf < rk.v[c]) ? 1u : 0u;
                }
            }
            if (nacc > 0) {
                // ONE merge through LDS: list entries move up by the accepted keys below them, accepted keys land at
                // (list keys below) + (accepted keys below); "not live" travels in bit 31 of the id word
#pragma unroll
                for (int c = 0; c < NCHR; c++) {
                    const int idx = c * 64 + lane;
                    const int npos = idx + (int)sT[c];
                    const u64 nl = ((live[c] >> lane) & 1ull) ? 0ull : PQB_NOTLIVE;
                    if (idx < rn && npos < cap) mk[npos] = rk.v[c] | nl;
                }
#pragma unroll
                for (int t = 0; t < NC; t++) {
                    const int npos = (int)(rT[t] + rA[t]);
                    if (((acc[t] >> lane) & 1ull) && npos < cap) mk[npos] = key[t];
                }
                WSYNC();
                const int rn2 = min(rn + nacc, cap);
#pragma unroll
                for (int c = 0; c < NCHR; c++) {
                    const int idx = c * 64 + lane;
                    const u64 v = (idx < rn2) ? mk[idx] : ~0ull;
                    live[c] = __ballot(idx < rn2 && (v & PQB_NOTLIVE) == 0ull);
                    rk.v[c] = (idx < rn2) ? (v & ~PQB_NOTLIVE) : ~0ull;
                }
                WSYNC();
                rn = rn2;
                nins += (u32)nacc;
            }
            seed = false;
            // frontier trim: only the beam_width smallest live entries stay live
            if (p.bw != 0u) {
                int nl = 0;
#pragma unroll
                for (int c = 0; c < NCHR; c++) nl += __popcll(live[c]);
                if (nl > (int)p.bw) {
                    int keep = (int)p.bw;
#pragma unroll
                    for (int c = 0; c < NCHR; c++) {
                        const int cnt = __popcll(live[c]);
                        if (keep >= cnt) keep -= cnt;
                        else {
                            const int rank = __popcll(live[c] & lanemask_lt());
                            live[c] &= ~__ballot(((live[c] >> lane) & 1ull) != 0ull && rank >= keep);
                            keep = 0;
                        }
                    }
                }
            }
        }

        // ---- results: keys for the rerank pass, the k best (ids, squared ADC), counters
        {
#pragma unroll
            for (int c = 0; c < NCHR; c++) {
                const int i = c * 64 + lane;
                if (i < rn) p.res_keys[(size_t)qi * cap + i] = (rk.v[c] & 0xFFFFFFFF00000000ull) | (u32)(~(u32)rk.v[c]);
            }
            const int kout = min((int)p.k, rn);
#pragma unroll
            for (int c = 0; c < NCHR; c++) {
                const int i = c * 64 + lane;
                if (i < (int)p.k) {
                    p.out_ids[(size_t)qi * p.k + i] = (i < kout) ? (u32)rk.v[c] : 0xFFFFFFFFu;
                    p.out_dist[(size_t)qi * p.k + i] = (i < kout) ? key_dist(rk.v[c]) : __uint_as_float(0x7FC00000u);
                }
            }
            for (int i = NCHR * 64 + lane; i < (int)p.k; i += 64) {
                p.out_ids[(size_t)qi * p.k + i] = 0xFFFFFFFFu;
                p.out_dist[(size_t)qi * p.k + i] = __uint_as_float(0x7FC00000u);
            }
            if (lane == 0) {
                p.res_n[qi] = (u32)rn;
                p.out_count[qi] = (u32)kout;
                KStats st;
                st.steps = steps; st.visited = nevals; st.exact = 0u; st.pq = nevals; st.status = status;
                st.inserts = nins; st.pq_evaluated = nevals; st.adj_prefetch_hits = 0u;
                p.stats[qi] = st;
            }
        }
        {
            u32 t = 0;
            if (lane == 0) t = atomicAdd(p.counter, 1u);
            qi = (u32)__builtin_amdgcn_readfirstlane((int)t) - p.ticket_base + nslots;
        }
    }
}
