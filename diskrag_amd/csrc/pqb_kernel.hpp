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
//   per surviving candidate: its rank in the list (a 4-ary search over the staged list) and among the accepted keys -> ONE scatter / gather merge
//   through LDS. (-DPQB_CANDIDATES_V2: round 6's form of that phase -- first search level out of the registers, membership folded into the last
//   level, the accepted keys ranked by a v_readlane loop: 2 dependent LDS round trips instead of 5, measured 0-3 % slower.)
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
    u64 *phase;              // [nq][8] cycle sums (DR_PHASE_TIMING builds only): 0 table landing, 1 pop, 2 rows, 3 code words, 4 ADC,
                             // 5 candidates, 6 merge + trim, 7 output
};

// A list key in the merge scratch: distance bits << 32 | id << 1 | not-live. The state bit is the LOWEST bit, so for a candidate key
// (state bit 0) `entry < key` and `entry > key` hold or fail exactly as for the pair (distance, id), and `(entry ^ key) <= 1` says
// "same node" -- the staged list can be searched without masking anything. N < 2^31.
#define PQB_NOTLIVE 1ull

// (wave_incl_scan_u32: search_kernel.hpp)

// squared ADC of one code word for a compile-time m = 16 * M16, rows 0 .. m-TREG-1 from LDS, the last TREG from registers
// (tv[jj*4 + v] of lane l = T[m-TREG+jj][64 v + l]); strict order of j. Whole-wave call (ds_bpermute).
template <int M16, int TREG, int GW>
DEV float pqb_adc(const float *lut, const float (&tv)[TREG > 0 ? TREG * 4 : 1], const uint4 (&cw)[M16])
{
    constexpr int M = M16 * 16, ML = M - TREG;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < M16; w++) {
        u32 words[4] = { cw[w].x, cw[w].y, cw[w].z, cw[w].w };
        // GW code words (4 GW sub-quantisers) at a time: their lookups (LDS reads, or four ds_bpermute each) in flight together, then
        // the strict sum. The empty asm makes the next group's decoding wait for this sum: left alone the compiler issues all
        // 64 permutes of a piece first and keeps their results (and the 32 decoded indices and select masks) live at once.
#pragma unroll
        for (int g = 0; g < 4; g += GW) {
            if (w + g > 0) {
#pragma unroll
                for (int u = 0; u < GW; u++) asm volatile("" : "+v"(words[g + u]) : "v"(s));
            }
            float t[4 * GW];
#pragma unroll
            for (int b = 0; b < 4 * GW; b++) {
                const int jq = w * 16 + g * 4 + b;
                const u32 c = (words[g + (b >> 2)] >> (8 * (b & 3))) & 255u;
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
            for (int b = 0; b < 4 * GW; b++) s = f_add(s, t[b]);
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
// LDS per wavefront (pqb_lds_bytes): table rows | staged list mk[NCHR*64] u64 | rank histogram [NCHR*64] u32 | accepted keys
// cbuf[NC*64 + 4] u64 | popped ids [16] u32
static inline size_t pqb_lds_bytes(uint32_t m, int treg, int nchr, int nc)
{
    return (size_t)(m - (uint32_t)treg) * 1024 + (size_t)nchr * 512 + (size_t)nchr * 256 + ((size_t)nc * 64 + 4) * 8 + 64;
}

template <int NCHR, int NC, int M16, int TREG>
// (wavefronts per SIMD the registers must allow: 3 with 24 of 32 rows in registers -- 8 KiB of LDS per wavefront, 12 per CU --; 4 for the
// small table of m = 16 with half of it in registers; m = 64: the LDS rows decide -- 32 of 64 rows in LDS are 4 wavefronts per CU, one per SIMD, so the
// registers need not be cut to 168 (round 6: that cap cost 64-970 bytes of scratch per lane), 24 / 16 rows in LDS are 6 / 8 per CU at <= 256 registers)
#ifdef PQB_FORCE_WAVES4      // A/B: the <= 128-register form -- four wavefronts per SIMD whatever it spills
__global__ __launch_bounds__(64, 4) void pqb_search_kernel(const PqbParams p)
#else
__global__ __launch_bounds__(64, (M16 == 1 && TREG == 8) ? 4 : (M16 == 4) ? ((TREG >= 48) ? 2 : 1) : (M16 == 3) ? ((TREG >= 32) ? 2 : 1) : (TREG >= 24) ? 3 : 2) void pqb_search_kernel(const PqbParams p)
#endif
{
    static_assert(TREG == 0 || (M16 > 0 && TREG <= M16 * 16 && TREG % 8 == 0), "register rows need a compile-time m");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const u32 m_lds = p.m - (u32)TREG;
    float *lut = reinterpret_cast<float *>(smem);
    u64 *mk = reinterpret_cast<u64 *>(smem + (size_t)m_lds * 1024);           // the list, staged: always the registers' copy (+ state bits)
    u32 *hist = reinterpret_cast<u32 *>(mk + NCHR * 64);                       // [NCHR*64] accepted keys per list rank; zero between steps
    u64 *cbuf = reinterpret_cast<u64 *>(hist + NCHR * 64);                     // [NC*64 + 4] the step's accepted keys, compacted
    u32 *popb = reinterpret_cast<u32 *>(cbuf + NC * 64 + 4);                   // [16] ids of the step's popped nodes
    const u32 slot_id = (u32)__builtin_amdgcn_readfirstlane((int)blockIdx.x);
    const u32 nslots = gridDim.x;
    const int cap = (int)p.cap;
    const u32 nwords = (p.R + 63) / 64;
    const u32 rs_mask = (1u << p.rs_shift) - 1u;
    // code words decoded together in the ADC: two (8 lookups in flight) when one pass per step leaves the registers for it
#ifdef PQB_FORCE_GW
    constexpr int GW = PQB_FORCE_GW;       // A/B builds
#else
    constexpr int GW = (NC == 1 && (TREG < 24 || NCHR <= 2)) ? 2 : 1;
#endif
#pragma unroll
    for (int c = 0; c < NCHR; c++) hist[c * 64 + lane] = 0u;

    u32 qi = slot_id;
    for (u32 round = 0; round < p.nq && qi < p.nq; ++round) {
        PH_BEGIN();
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
        PH(0);
        RegList<NCHR> rk;           // keys in registers: distance bits << 32 | id << 1 (state bit clear; ~0 = unused)
        u64 live[NCHR];             // live entries, one bit per list position (wave-uniform: scalar registers)
#pragma unroll
        for (int c = 0; c < NCHR; c++) { rk.v[c] = ~0ull; live[c] = 0ull; }
        int rn = 0;
        u32 steps = 0, nevals = 0, nins = 0, status = 0;

        // ---- steps. The first one is the SEED: nothing is popped, lane 0 of pass 0 scores the start node (the one copy of the
        // ADC code serves both) and the merge puts it into the empty list.
        bool seed = true;
        for (;;) {
            int np = 0;
            if (!seed) {
                int nlive = 0;
#pragma unroll
                for (int c = 0; c < NCHR; c++) nlive += __popcll(live[c]);
                if (nlive == 0 || steps >= p.max_steps) break;
                if ((u64)steps > p.N + 64) { status |= DR_ST_INTERNAL; break; }
                np = min(min((int)p.pops, nlive), (int)(p.max_steps - steps));
                // pop: the np smallest live entries -- a live entry's rank among the live ones is a prefix popcount of the masks
                int base = 0;
#pragma unroll
                for (int c = 0; c < NCHR; c++) {
                    const u64 lm = live[c];
                    const int rank = base + __popcll(lm & lanemask_lt());
                    const bool sel = ((lm >> lane) & 1ull) != 0ull && rank < np;
                    if (sel) popb[rank] = (u32)rk.v[c] >> 1;
                    live[c] = lm & ~__ballot(sel);
                    base += __popcll(lm);
                }
                steps += (u32)np;
                WSYNC();
            }
            PH(1);

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
                    const u32 cur = popb[min(ni, (u32)max(np, 1) - 1u)];
                    cur_[t] = cur;
                    nid[t] = p.adj[(size_t)cur * p.R + sl];
                    fw[t] = p.first[(size_t)cur * nwords + (sl >> 6)];
                }
            }
            PH(2);
            bool act[NC];
#pragma unroll
            for (int t = 0; t < NC; t++) act[t] = valid[t] && ((fw[t] >> (slot_[t] & 63u)) & 1ull) != 0ull;
            // code words (inline: beside the row, no dependency on the ids; else a gather behind them)
            uint4 cw[NC][M16 > 0 ? M16 : 1];
#pragma unroll
            for (int t = 0; t < NC; t++) {
                if constexpr (M16 > 0) {
                    const u8 *code = (p.nbcodes && !seed) ? p.nbcodes + ((size_t)cur_[t] * p.R + slot_[t]) * p.m : p.codes + (size_t)(act[t] ? nid[t] : 0u) * p.m;
#pragma unroll
                    for (int w = 0; w < M16; w++) cw[t][w] = reinterpret_cast<const uint4 *>(code)[w];
                }
            }
            PH(3);
            const bool full = (rn == cap);
            const u64 wk = full ? list_get<NCHR>(rk, cap - 1) : ~0ull;
            u64 key[NC], cm[NC];
            u64 anyc = 0ull;
#pragma unroll
            for (int t = 0; t < NC; t++) {
                float e;
                if constexpr (M16 > 0) e = pqb_adc<M16, TREG, GW>(lut, tv, cw[t]);
                else {
                    const u8 *code = (p.nbcodes && !seed) ? p.nbcodes + ((size_t)cur_[t] * p.R + slot_[t]) * p.m : p.codes + (size_t)(act[t] ? nid[t] : 0u) * p.m;
                    e = pqb_adc_generic(lut, code, p.m);
                }
                key[t] = ((u64)__float_as_uint(e) << 32) | ((u64)nid[t] << 1);
                nevals += (u32)__popcll(__ballot(act[t]));
                cm[t] = __ballot(act[t] && key[t] < wk);
                anyc |= cm[t];
            }
            PH(4);

#ifndef PQB_CANDIDATES_V2
            // ---- candidates (key below the list's largest key, or any while the list fills), all lanes at once:
            // (1) rank in the list = binary search over the staged list; the entry found there says whether the node is IN the list;
            // (2) several rows per step: the same node through two rows -- the later copy leaves (a loop over the compacted keys);
            // (3) the accepted keys are compacted into LDS; every accepted lane counts the accepted keys below its own (broadcast
            //     reads, no dependency between them), and a histogram of the list ranks, prefix-summed, tells every list entry how
            //     many accepted keys lie below it;
            // (4) ONE scatter / gather merge through the staged list.
            if (anyc != 0ull) {
                constexpr int QIT = (NCHR == 1) ? 3 : (NCHR <= 4) ? 4 : 5;      // 4^QIT >= NCHR * 64
                u64 acc[NC];
                int lb[NC], ci[NC];
                int nacc = 0;
#pragma unroll
                for (int t = 0; t < NC; t++) {
                    // #(list keys < key): a 4-ary search over the staged list padded with +inf to 4^QIT entries -- three independent
                    // LDS reads per level, QIT DEPENDENT round trips instead of the 2 QIT of a binary search
                    int lo = 0;
                    bool inl = false;
                    if (rn > 0) {
#pragma unroll
                        for (int it = 0; it < QIT; it++) {
                            const int st = 1 << (2 * (QIT - 1 - it));
                            const int i1 = lo + st - 1, i2 = lo + 2 * st - 1, i3 = lo + 3 * st - 1;
                            const u64 v1 = mk[min(i1, rn - 1)], v2 = mk[min(i2, rn - 1)], v3 = mk[min(i3, rn - 1)];
                            const int c1 = (i1 < rn && v1 < key[t]) ? 1 : 0, c2 = (i2 < rn && v2 < key[t]) ? 1 : 0, c3 = (i3 < rn && v3 < key[t]) ? 1 : 0;
                            lo += (c1 + c2 + c3) * st;
                        }
                        const u64 vv = mk[min(lo, rn - 1)];
                        inl = lo < rn && (vv ^ key[t]) <= 1ull;
                    }
                    lb[t] = lo;
                    acc[t] = cm[t] & __ballot(!inl);
                    ci[t] = nacc + __popcll(acc[t] & lanemask_lt());
                    nacc += __popcll(acc[t]);
                }
                if (np > 1 && nacc > 1) {
                    // the same node through two of the step's rows: only its first copy (in lane order) stays
#pragma unroll
                    for (int t = 0; t < NC; t++) if ((acc[t] >> lane) & 1ull) cbuf[ci[t]] = key[t];
                    WSYNC();
                    bool dup[NC];
#pragma unroll
                    for (int t = 0; t < NC; t++) dup[t] = false;
#pragma unroll 8
                    for (int jq = 0; jq < nacc; jq++) {
                        const u64 kj = cbuf[jq];
#pragma unroll
                        for (int t = 0; t < NC; t++) dup[t] = dup[t] || (kj == key[t] && jq < ci[t]);
                    }
                    WSYNC();
                    nacc = 0;
#pragma unroll
                    for (int t = 0; t < NC; t++) {
                        acc[t] &= ~__ballot(dup[t]);
                        ci[t] = nacc + __popcll(acc[t] & lanemask_lt());
                        nacc += __popcll(acc[t]);
                    }
                }
                if (nacc > 0) {
                    bool isacc[NC];
#pragma unroll
                    for (int t = 0; t < NC; t++) {
                        isacc[t] = ((acc[t] >> lane) & 1ull) != 0ull;
                        if (isacc[t]) { cbuf[ci[t]] = key[t]; atomicAdd(&hist[lb[t]], 1u); }
                    }
                    if (lane < 4) cbuf[nacc + lane] = ~0ull;          // (the count loop below reads four keys per trip)
                    WSYNC();
                    u32 rA[NC];
#pragma unroll
                    for (int t = 0; t < NC; t++) rA[t] = 0u;
#pragma unroll 1
                    for (int jq = 0; jq < nacc; jq += 4) {
                        const u64 k0 = cbuf[jq], k1 = cbuf[jq + 1], k2 = cbuf[jq + 2], k3 = cbuf[jq + 3];
#pragma unroll
                        for (int t = 0; t < NC; t++)
                            rA[t] += (k0 < key[t] ? 1u : 0u) + (k1 < key[t] ? 1u : 0u) + (k2 < key[t] ? 1u : 0u) + (k3 < key[t] ? 1u : 0u);
                    }
                    // list entry i moves up by the accepted keys below it = those whose list rank is <= i
                    u32 carry = 0u;
#pragma unroll
                    for (int c = 0; c < NCHR; c++) {
                        const int idx = c * 64 + lane;
                        const u32 inc = wave_incl_scan_u32(hist[idx]) + carry;
                        hist[idx] = 0u;
                        carry = readlane32(inc, 63);
                        const int npos = idx + (int)inc;
                        const u64 nl = ((live[c] >> lane) & 1ull) ? 0ull : PQB_NOTLIVE;
                        if (idx < rn && npos < cap) mk[npos] = rk.v[c] | nl;
                    }
#pragma unroll
                    for (int t = 0; t < NC; t++) {
                        const int npos = lb[t] + (int)rA[t];
                        if (isacc[t] && npos < cap) mk[npos] = key[t];
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
            }
#else      // A/B of round 6 (profiles/r06/ab/ab_pqb_candidates_phase_*.jsonl): fewer dependent LDS round trips, 0-3 % SLOWER -- not the default
            // ---- candidates (key below the list's largest key, or any while the list fills), all lanes at once:
            // (1) rank in the list. First level WITHOUT a memory access: the list is sorted in the registers, its entries at the fixed positions
            //     SEG - 1, 2 SEG - 1, ... are read out with v_readlane (wave-uniform: scalar operands of the compares) and cut the list into K0
            //     segments; then LEV 4-ary levels over the staged list inside the lane's segment (three independent LDS reads per level). The
            //     last level reads a fourth entry, so the entry AT the rank -- which says whether the node is IN the list -- needs no round trip
            //     of its own: LEV dependent LDS round trips (2 for lists of up to 256 entries) where round 5 took 5-6;
            // (2) rank among the accepted keys: a loop over the accepted LANES (the mask is wave-uniform, the key comes out of its register
            //     with v_readlane: no LDS). The same loop sees a node accepted through two of the step's rows (equal keys): the later copy
            //     leaves and the loop runs once more, on the rare steps where that happens. Long candidate sets (a filling list) count
            //     through LDS, four keys per trip, as round 5 did for all;
            // (3) a histogram of the list ranks, prefix-summed, tells every list entry how many accepted keys lie below it;
            // (4) ONE scatter / gather merge through the staged list.
            if (anyc != 0ull) {
                constexpr int K0 = (NCHR * 4 < 16) ? NCHR * 4 : 16;          // first-level segments
                constexpr int SEG = NCHR * 64 / K0;                          // 16 (lists of up to 256 entries), 32, 64
                constexpr int LEV = (SEG <= 16) ? 2 : 3;                     // 4^LEV >= SEG
                u64 acc[NC];
                int lb[NC];
                int nacc = 0;
                u64 spl[K0 - 1];
#pragma unroll
                for (int i = 0; i < K0 - 1; i++) spl[i] = readlane64(rk.v[((i + 1) * SEG - 1) >> 6], ((i + 1) * SEG - 1) & 63);      // (unused entries are ~0: never below a key)
#pragma unroll
                for (int t = 0; t < NC; t++) {
                    int lo = 0;
                    bool inl = false;
                    if (rn > 0) {
#pragma unroll
                        for (int i = 0; i < K0 - 1; i++) lo += (spl[i] < key[t]) ? SEG : 0;
#pragma unroll
                        for (int it = 0; it < LEV - 1; it++) {
                            const int st = 1 << (2 * (LEV - 1 - it));
                            const int i1 = lo + st - 1, i2 = lo + 2 * st - 1, i3 = lo + 3 * st - 1;
                            const u64 v1 = mk[min(i1, rn - 1)], v2 = mk[min(i2, rn - 1)], v3 = mk[min(i3, rn - 1)];
                            const int c1 = (i1 < rn && v1 < key[t]) ? 1 : 0, c2 = (i2 < rn && v2 < key[t]) ? 1 : 0, c3 = (i3 < rn && v3 < key[t]) ? 1 : 0;
                            lo += (c1 + c2 + c3) * st;
                        }
                        // last level: the rank is lo ... lo + 3, and the entry at the rank is one of the four read here
                        const u64 e0 = mk[min(lo, rn - 1)], e1 = mk[min(lo + 1, rn - 1)], e2 = mk[min(lo + 2, rn - 1)], e3 = mk[min(lo + 3, rn - 1)];
                        const int c = ((lo < rn && e0 < key[t]) ? 1 : 0) + ((lo + 1 < rn && e1 < key[t]) ? 1 : 0) + ((lo + 2 < rn && e2 < key[t]) ? 1 : 0);
                        const u64 vv = (c == 0) ? e0 : (c == 1) ? e1 : (c == 2) ? e2 : e3;
                        lo += c;
                        inl = lo < rn && (vv ^ key[t]) <= 1ull;
                    }
                    lb[t] = lo;
                    acc[t] = cm[t] & __ballot(!inl);
                    nacc += __popcll(acc[t]);
                }
                if (nacc > 0) {
#ifdef PQB_LOOP_MAX
                    constexpr int LOOP_MAX = PQB_LOOP_MAX;        // A/B
#else
                    constexpr int LOOP_MAX = 24;          // accepted keys counted by the readlane loop (more: through LDS)
#endif
                    bool isacc[NC];
                    u32 rA[NC];
#pragma unroll 1
                    for (int again = 0; again < 2; again++) {
                        bool dupt[NC];
#pragma unroll
                        for (int t = 0; t < NC; t++) { isacc[t] = ((acc[t] >> lane) & 1ull) != 0ull; rA[t] = 0u; dupt[t] = false; }
                        if (nacc <= LOOP_MAX) {
#pragma unroll
                            for (int u = 0; u < NC; u++) {
                                u64 mrem = acc[u];
                                while (mrem != 0ull) {
                                    const int j = __builtin_ctzll(mrem);
                                    mrem &= mrem - 1ull;
                                    const u64 kj = readlane64(key[u], j);
#pragma unroll
                                    for (int t = 0; t < NC; t++) {
                                        rA[t] += (kj < key[t]) ? 1u : 0u;
                                        // equal keys = the same node through two of the step's rows: the copy in the earlier (pass, lane) stays
                                        if (t > u) dupt[t] = dupt[t] || kj == key[t];
                                        else if (t == u) dupt[t] = dupt[t] || (kj == key[t] && j < lane);
                                    }
                                }
                            }
                        } else {
                            int ci[NC];
                            int base = 0;
#pragma unroll
                            for (int t = 0; t < NC; t++) {
                                ci[t] = base + __popcll(acc[t] & lanemask_lt());
                                base += __popcll(acc[t]);
                                if (isacc[t]) cbuf[ci[t]] = key[t];
                            }
                            if (lane < 4) cbuf[nacc + lane] = ~0ull;          // (four keys per trip)
                            WSYNC();
#pragma unroll 1
                            for (int jq = 0; jq < nacc; jq += 4) {
                                const u64 k0 = cbuf[jq], k1 = cbuf[jq + 1], k2 = cbuf[jq + 2], k3 = cbuf[jq + 3];
#pragma unroll
                                for (int t = 0; t < NC; t++) {
                                    rA[t] += (k0 < key[t] ? 1u : 0u) + (k1 < key[t] ? 1u : 0u) + (k2 < key[t] ? 1u : 0u) + (k3 < key[t] ? 1u : 0u);
                                    dupt[t] = dupt[t] || (k0 == key[t] && jq < ci[t]) || (k1 == key[t] && jq + 1 < ci[t]) || (k2 == key[t] && jq + 2 < ci[t]) ||
                                              (k3 == key[t] && jq + 3 < ci[t]);
                                }
                            }
                            WSYNC();
                        }
                        if (np <= 1 || again == 1) break;          // (one row per step: its first-occurrence slots hold distinct nodes)
                        u64 anyd = 0ull;
                        int nacc2 = 0;
#pragma unroll
                        for (int t = 0; t < NC; t++) {
                            const u64 dm = __ballot(dupt[t]) & acc[t];
                            anyd |= dm;
                            acc[t] &= ~dm;
                            nacc2 += __popcll(acc[t]);
                        }
                        if (anyd == 0ull) break;
                        nacc = nacc2;          // later copies left: the counts are taken again without them
                    }
#pragma unroll
                    for (int t = 0; t < NC; t++) if (isacc[t]) atomicAdd(&hist[lb[t]], 1u);
                    WSYNC();
                    // list entry i moves up by the accepted keys below it = those whose list rank is <= i
                    u32 carry = 0u;
#pragma unroll
                    for (int c = 0; c < NCHR; c++) {
                        const int idx = c * 64 + lane;
                        const u32 inc = wave_incl_scan_u32(hist[idx]) + carry;
                        hist[idx] = 0u;
                        carry = readlane32(inc, 63);
                        const int npos = idx + (int)inc;
                        const u64 nl = ((live[c] >> lane) & 1ull) ? 0ull : PQB_NOTLIVE;
                        if (idx < rn && npos < cap) mk[npos] = rk.v[c] | nl;
                    }
#pragma unroll
                    for (int t = 0; t < NC; t++) {
                        const int npos = lb[t] + (int)rA[t];
                        if (isacc[t] && npos < cap) mk[npos] = key[t];
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
            }
#endif
            PH(5);
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
            PH(6);
        }

        // ---- results: keys for the rerank pass, the k best (ids, squared ADC), counters
        {
#pragma unroll
            for (int c = 0; c < NCHR; c++) {
                const int i = c * 64 + lane;
                if (i < rn) p.res_keys[(size_t)qi * cap + i] = (rk.v[c] & 0xFFFFFFFF00000000ull) | (u32)(~((u32)rk.v[c] >> 1));
            }
            const int kout = min((int)p.k, rn);
#pragma unroll
            for (int c = 0; c < NCHR; c++) {
                const int i = c * 64 + lane;
                if (i < (int)p.k) {
                    p.out_ids[(size_t)qi * p.k + i] = (i < kout) ? ((u32)rk.v[c] >> 1) : 0xFFFFFFFFu;
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
        PH(7);
        PH_END(qi);
        {
            u32 t = 0;
            if (lane == 0) t = atomicAdd(p.counter, 1u);
            qi = (u32)__builtin_amdgcn_readfirstlane((int)t) - p.ticket_base + nslots;
        }
    }
}
