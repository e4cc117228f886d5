// search_kernel.hpp -- the beam-search kernel: one 64-lane wavefront per query, persistent wavefronts.
//
// Reference control flow restated (file:line into Jolara-ai/diskrag):
//   M1 search_engine.py:398-506, M2 pydiskann/vamana_graph.py:719-760, M3 :535-605, M4 :607-640.
//   (mode 5 = DR_MODE_PQ, no reference counterpart: M1's loop with the squared ADC as the only distance.)
// All four are the same loop -- pop the closest frontier node, score its unvisited neighbours in stored
// order, insert the accepted ones into a bounded result list and the frontier, optionally trim the frontier --
// and differ in distance function, result-list capacity, stop rule and trim rule (SearchParams).
//
// Where a query's state lives:
//   VGPRs  result list (which doubles as the frontier: a per-entry "popped" flag) as a sorted array of 64-bit keys
//          (distance bits << 32 | ~id), one element per lane per 64-entry chunk; a whole expansion's accepted
//          neighbours enter it with ONE scatter/gather merge through LDS; single inserts (start node, tie side
//          list) are a ballot plus one wave shift per chunk; the query (chain-major) for D <= 256
//   LDS    PQ data for ADC: either the whole codebook shared by all wavefronts of the workgroup (D <= 128:
//          4*256*D bytes <= 128 KiB; table entries T[j][c] are recomputed per neighbour in the reference's order,
//          which lets 8-16 queries share a CU instead of the 4 that per-query tables allow), or the per-query
//          table T[m][256] (larger D); the landing area of the row variants (the stored vectors of an expansion arrive
//          by global_load_lds); a 64-entry staging area per expansion; the word filter of the visited set
//   HBM    exact visited set: per wavefront slot one word per 24 bit positions + an 8-bit stamp of the query that wrote
//          it (plain load / plain store, lanes that share a word combined in LDS, never cleared between queries);
//          accepted-insert log per query (tie replay in finalize)
//
// Sequential semantics kept exactly: the neighbours of one expansion are scored in parallel (distances do not
// depend on list state); which of them the reference would have scored and accepted, walking them in stored order
// with the worst distance W in effect at each position and the rerank policy A4 (search_engine.py:381-397), is
// then computed without a walk -- see "decisions" in the main loop: counts against the old list by binary search,
// counts against earlier neighbours by bit masks, and a fixed point that converges in one or two ballots.
#pragma once
#include "variants.hpp"
#include "numerics.hpp"

// A wavefront's LDS operations execute in issue order, so cross-lane hand-offs through LDS inside ONE wave only
// need the compiler not to reorder them.
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// Diagnostic build (-DDR_PHASE_TIMING): per-phase shader-clock sums per query (SearchParams::phase). Each stamp
// drains the wave's memory queues first, so the build is slower than the real one: read shares, not totals.
#ifdef DR_PHASE_TIMING
#define PH_STAMP(t) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define PH_BEGIN() u64 ph_acc[8] = {0,0,0,0,0,0,0,0}; u64 ph_t0, ph_t1; PH_STAMP(ph_t0)
#define PH(i) do { PH_STAMP(ph_t1); ph_acc[i] += ph_t1 - ph_t0; ph_t0 = ph_t1; } while (0)
#define PH_END(qi) do { if (lane == 0 && p.phase) for (int i_ = 0; i_ < 8; i_++) p.phase[(size_t)(qi) * 8 + i_] = ph_acc[i_]; } while (0)
#elif defined(DR_PHASE_MARK)
// (-DDR_PHASE_MARK: the phase boundaries as comments in the otherwise unchanged ISA -- scripts/phase_budget.py counts the
// instructions between them)
#define PH_BEGIN() asm volatile("; DR_PHASE_BEGIN")
#define PH(i) asm volatile("; DR_PHASE_END " #i)
#define PH_END(qi) do {} while (0)
#else
#define PH_BEGIN() do {} while (0)
#define PH(i) do {} while (0)
#define PH_END(qi) do {} while (0)
#endif

// (-DDR_PHASE_TIMING -DDR_DEC_SUB: the decision pass by the path a row takes, over slots that are otherwise nearly empty -- 4 = rows without a
//  candidate, 7 = every candidate accepted, 6 = verdicts in closed form, 0 = the general path; each stamp also counts its rows in bits 36.. of the slot: scripts/exp_phase_single.py)
#if defined(DR_PHASE_TIMING) && defined(DR_DEC_SUB)
#define PHD(i) do { PH(i); ph_acc[i] += 1ull << 36; } while (0)
#else
#define PHD(i) do {} while (0)
#endif

#define DR_ST_CAND_OVERFLOW 2u
#define DR_ST_LOG_OVERFLOW 4u
#define DR_ST_INTERNAL 8u      // a loop guard fired (never expected; bounds every loop so a bug cannot hang the GPU)

enum DistKind { DIST_EXACT = 0, DIST_ADC_SQ = 2 };

// A/B switch (round 4): score the rows of a byte-row burst instruction by instruction as they land instead of after the whole burst
#ifndef DR_BURST_PIPE
#define DR_BURST_PIPE 0
#endif
template <int K> DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory"); }

struct KStats { u32 steps, visited, exact, pq, status, inserts, pq_evaluated, adj_prefetch_hits; };

struct SearchParams {
    const float *vecp;       // [N][D] chain-major
    const u8 *vec8;          // [N][D] the same vectors as bytes when every component is an integer in [0, 255] (else nullptr)
    const u32 *adj;          // [N][R]
    const u32 *adjr;         // [N][R] bit position of each neighbour in the visited bitmap (nullptr: the id itself)
    const u64 *first;        // [N][ceil(R/64)] bit s: slot s is a real id and its first occurrence in the row
    const u32 *deg;          // build mode (first == nullptr): rows hold deg[i] distinct ids, the rest is DR_PAD
    const u8 *codes;         // [N][m]
    const u8 *nbcodes;       // [N][R][m] code words of every adjacency slot's neighbour, or nullptr (dr_index_inline_codes)
    const float *codebook;   // [m][256][sd]
    const float *lut_g;      // [nq][m][256] the per-query tables T[j][c] of the batch (lut_build_kernel), per-query-table variants
    const float *queries;    // [nq][D] original element order
    const float *queries_p;  // [nq][D] chain-major
    u64 N;
    u32 D, R, m, sd, medoid, nq;
    u32 medoid_pos;          // bit position of the medoid
    u32 mode, k, cap, L, bw, policy, flags;
    u32 norm;                // 1: traversal metric is sqrt(squared L2) (M2, M4: np.linalg.norm)
    u32 max_steps;           // M1: min(10L, N); others: 0xFFFFFFFF
    u32 *vis;                // [slots][vis_words] visited sets: words of 24 position bits + an 8-bit query stamp (see below)
    u32 vis_words;           // ceil(N / 24), rounded up to a multiple of 4
    u32 *vis_epoch;          // [slots] stamp of the last query each slot served (persists across launches)
    u32 *counter;            // [2]: query ticket counter (monotonic: a launch draws exactly nq tickets), tie-list length
    u32 ticket_base;         // value of the ticket counter when this launch starts
    u64 *res_keys;           // [nq][cap] ascending (dist bits << 32 | ~id)
    u32 *res_n;              // [nq]
    KStats *stats;           // [nq]
    u32 *tie_list;           // queries whose first k results hold equal sort keys (finalize replays their heap)
    u32 *tie_count;
    u64 *log;                // [nq][logcap] accepted inserts in order (dist bits << 32 | id)
    u32 logcap;
    u32 *out_ids;            // [nq][k]   (nullptr in build mode)
    float *out_dist;         // [nq][k]
    u32 *out_count;          // [nq]
    u64 *phase;              // [nq][8] cycle sums (DR_PHASE_TIMING builds only)
    u32 *tie_flag;           // small blocking calls: a word in the host's result slab set when a query was listed for the tie-order pass (else nullptr)
    const float *pq_ub;      // [nq] precomputed sqrt-ADC upper bounds (pq_bound_kernel) or nullptr
    const float *vnorm2;     // [N] squared norms of the stored vectors (DR_F_COSINE: M3 with distance_metric='cosine') or nullptr
    // builder over a PQ-only shard (no stored vectors): query qi IS the stored point build_pts[qi], known by its code
    // word only, and its table T[j][c] = |C_j[c] - C_j[code_j]|^2 is m rows of the centroid-pair table sdc[m][256][256]
    const float *sdc;
    const u32 *build_pts;
    // 1: no visited set (round 4; the engine's own ADC traversals -- DR_MODE_PQ and the PQ-only builder's searches). Every
    // first-occurrence neighbour of an expansion is scored; one that would enter the list is looked up IN the list (same id
    // => same ADC distance => same 64-bit key) and dropped if it is there. That is exact: a node scored before and not in the
    // list now was rejected or evicted at a worst distance W' >= W(now) and can never be accepted again, and while the list
    // is filling every scored node is in it -- so ids, distances and the accepted-insert sequence are those of the
    // visited-set form, while the visited words (a load per test, a store per new node, 20 MB per wavefront slot on a
    // 1.25e8-point shard), their bit-position twin of the adjacency and the grouping passes are gone. stats.visited / stats.pq
    // then count EVALUATIONS (a node met again after it fell out of the list is scored again). vis / adjr may be null.
    // bit 1 (with bit 0): the ids of the predicted next pop's adjacency row are landed in LDS during the running expansion
    // (R <= 128), so that on a hit the expansion starts with its code-word gathers instead of waiting for the row first.
    // bit 2: the adjacency prefetch of the byte-query variants WITH a second chance (A/B switch DR_REPREFETCH=1; measured: hits 58 -> 97 %,
    // kernel 1 % slower -- profiles/r04/ab/ab_c2_second_chance_prefetch.jsonl).
    u32 novis;
    u32 vh_bits;             // latency_kernel.hpp: log2 of the slots of the visited-id hash set in LDS (+ option bits)
    const u32 *perm;         // [D] position of original element e in the chain-major layout (latency_kernel.hpp permutes a query itself when queries_p is null)
};

DEV u32 lane_id() { return threadIdx.x & 63; }
DEV u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }
DEV float key_dist(u64 key) { return __uint_as_float((u32)(key >> 32)); }

// ---- cross-lane helpers -----------------------------------------------------------------------------------
DEV u32 readlane32(u32 x, int l) { return (u32)__builtin_amdgcn_readlane((int)x, __builtin_amdgcn_readfirstlane(l)); }
DEV u64 readlane64(u64 x, int l)
{
    const int ll = __builtin_amdgcn_readfirstlane(l);
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)x, ll);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), ll);
    return ((u64)hi << 32) | lo;
}
// lane l <- x[l-1]; lane 0 <- carry   (DPP wave_shr:1, full-wave shift on GFX9)
DEV u64 wave_shr1(u64 x, u64 carry)
{
    const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)carry, (int)(u32)x, 0x138, 0xf, 0xf, false);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(carry >> 32), (int)(u32)(x >> 32), 0x138, 0xf, 0xf, false);
    return ((u64)hi << 32) | lo;
}
// lane l <- x[l+1]; lane 63 <- carry  (DPP wave_shl:1)
DEV u64 wave_shl1(u64 x, u64 carry)
{
    const u32 lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)carry, (int)(u32)x, 0x130, 0xf, 0xf, false);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(carry >> 32), (int)(u32)(x >> 32), 0x130, 0xf, 0xf, false);
    return ((u64)hi << 32) | lo;
}

// min over the wave of a u32 (identity 0xFFFFFFFF), result broadcast: DPP inclusive scan (row_shr 1,2,4,8, then
// row_bcast 15 / 31), total in lane 63.
DEV u32 wave_min_u32(u32 x)
{
    const int idn = -1;
#define DR_DPP_MIN(ctrl, rmask) x = min(x, (u32)__builtin_amdgcn_update_dpp(idn, (int)x, ctrl, rmask, 0xf, false))
    DR_DPP_MIN(0x111, 0xf);
    DR_DPP_MIN(0x112, 0xf);
    DR_DPP_MIN(0x114, 0xf);
    DR_DPP_MIN(0x118, 0xf);
    DR_DPP_MIN(0x142, 0xa);
    DR_DPP_MIN(0x143, 0xc);
#undef DR_DPP_MIN
    return readlane32(x, 63);
}

// max over the wave of a u32 (identity 0), result broadcast (search_kernel.hpp wave_min_u32's scan with max)
// inclusive prefix sum over the wavefront's lanes (rows of 16 by shifts, then the row totals broadcast forward)
DEV u32 wave_incl_scan_u32(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return x;
}

DEV u32 wave_max_u32(u32 x)
{
#define DR_DPP_MAX(ctrl, rmask) x = max(x, (u32)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rmask, 0xf, false))
    DR_DPP_MAX(0x111, 0xf);
    DR_DPP_MAX(0x112, 0xf);
    DR_DPP_MAX(0x114, 0xf);
    DR_DPP_MAX(0x118, 0xf);
    DR_DPP_MAX(0x142, 0xa);
    DR_DPP_MAX(0x143, 0xc);
#undef DR_DPP_MAX
    return readlane32(x, 63);
}

// ---- register-resident sorted lists -----------------------------------------------------------------------
// Ascending array of NCH*64 keys; lane l of chunk c holds index c*64+l. Result list key = dist bits << 32 | ~id
// (last element = what heapq pops from the reference's max-heap of (-dist, id): largest distance, smallest id
// among equals); frontier key = dist bits << 32 | id (first element = heappop of the min-heap of (dist, id)).
template <int NCH> struct RegList { u64 v[NCH]; };

template <int NCH> DEV u64 list_get(const RegList<NCH> &L, int idx)
{
    u64 r = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++)
        if ((idx >> 6) == c) r = readlane64(L.v[c], idx & 63);
    return r;
}

// Inserts key (n entries, capacity cap <= NCH*64). When full, the largest entry falls off the end (or the key
// itself, if it is the largest): reported through dropped/did_drop. Returns the new length.
template <int NCH> DEV int list_insert(RegList<NCH> &L, int n, int cap, u64 key, u64 &dropped, bool &did_drop)
{
    const int lane = lane_id();
    int pos = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) pos += __popcll(__ballot((c * 64 + lane) < n && L.v[c] < key));
    did_drop = (n == cap);
    if (did_drop) {
        if (pos >= cap) { dropped = key; return n; }
        dropped = list_get<NCH>(L, cap - 1);
    }
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--) {
        const u64 carry = (c > 0) ? readlane64(L.v[c > 0 ? c - 1 : 0], 63) : 0ull;
        const u64 prev = wave_shr1(L.v[c], carry);
        const int idx = c * 64 + lane;
        const bool take_prev = idx > pos && idx <= n && idx < cap;
        L.v[c] = (idx == pos) ? key : (take_prev ? prev : L.v[c]);
    }
    return n < cap ? n + 1 : cap;
}

template <int NCH> DEV void list_pop_front(RegList<NCH> &L)
{
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u64 carry = (c + 1 < NCH) ? readlane64(L.v[c + 1 < NCH ? c + 1 : c], 0) : ~0ull;
        L.v[c] = wave_shl1(L.v[c], carry);
    }
}

// Per-entry state of the result list: bit0 = expanded, bit1 = trimmed from the frontier ("dead"). The frontier of
// the reference (a heap of (dist, id)) is the set of result entries with state 0, plus the few entries that were
// evicted from the results while tied with the new worst distance (side list), plus a COUNT of evicted entries
// that are worse than every result (they can only end the search). One sorted insert per accepted neighbour.
template <int NCH> struct FlagList { u32 v[NCH]; };

DEV u32 wave_shr1_u32(u32 x, u32 carry)
{
    return (u32)__builtin_amdgcn_update_dpp((int)carry, (int)x, 0x138, 0xf, 0xf, false);
}
template <int NCH> DEV u32 flag_get(const FlagList<NCH> &F, int idx)
{
    u32 r = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++)
        if ((idx >> 6) == c) r = readlane32(F.v[c], idx & 63);
    return r;
}
template <int NCH> DEV void flag_or(FlagList<NCH> &F, int idx, u32 bits)
{
    const int lane = lane_id();
#pragma unroll
    for (int c = 0; c < NCH; c++) F.v[c] |= (c == (idx >> 6) && lane == (idx & 63)) ? bits : 0u;
}

// list_insert for the flagged result list: the new entry gets state 0; reports the evicted entry and its state.
template <int NCH>
DEV int list_insert_f(RegList<NCH> &L, FlagList<NCH> &F, int n, int cap, u64 key, u64 &dropped, u32 &dflag, bool &did_drop)
{
    const int lane = lane_id();
    int pos = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) pos += __popcll(__ballot((c * 64 + lane) < n && L.v[c] < key));
    did_drop = (n == cap);
    dflag = 0;
    if (did_drop) {
        if (pos >= cap) { dropped = key; return n; }
        dropped = list_get<NCH>(L, cap - 1);
        dflag = flag_get<NCH>(F, cap - 1);
    }
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--) {
        const u64 carry = (c > 0) ? readlane64(L.v[c > 0 ? c - 1 : 0], 63) : 0ull;
        const u32 carryf = (c > 0) ? readlane32(F.v[c > 0 ? c - 1 : 0], 63) : 0u;
        const u64 prev = wave_shr1(L.v[c], carry);
        const u32 prevf = wave_shr1_u32(F.v[c], carryf);
        const int idx = c * 64 + lane;
        const bool take_prev = idx > pos && idx <= n && idx < cap;
        L.v[c] = (idx == pos) ? key : (take_prev ? prev : L.v[c]);
        F.v[c] = (idx == pos) ? 0u : (take_prev ? prevf : F.v[c]);
    }
    return n < cap ? n + 1 : cap;
}

// Frontier order is (dist asc, id asc); the list is (dist asc, id DESC), so inside a run of equal distances the
// frontier order is the list order reversed. Index of the first / last live entry in frontier order, or -1.
template <int NCH> DEV int frontier_first(const RegList<NCH> &L, const FlagList<NCH> &F, int n)
{
    const int lane = lane_id();
    int pidx = -1;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u64 m = __ballot(c * 64 + lane < n && F.v[c] == 0u);
        if (m != 0ull && pidx < 0) pidx = c * 64 + __ffsll((long long)m) - 1;
    }
    if (pidx < 0) return -1;
    const u32 dp = (u32)(list_get<NCH>(L, pidx) >> 32);
    int q = pidx;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u64 m = __ballot(c * 64 + lane < n && F.v[c] == 0u && (u32)(L.v[c] >> 32) == dp);
        if (m != 0ull) q = max(q, c * 64 + 63 - __clzll((long long)m));
    }
    return q;
}
template <int NCH> DEV int frontier_last(const RegList<NCH> &L, const FlagList<NCH> &F, int n)
{
    const int lane = lane_id();
    int q = -1;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u64 m = __ballot(c * 64 + lane < n && F.v[c] == 0u);
        if (m != 0ull) q = c * 64 + 63 - __clzll((long long)m);
    }
    if (q < 0) return -1;
    const u32 dq = (u32)(list_get<NCH>(L, q) >> 32);
    int r = q;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const u64 m = __ballot(c * 64 + lane < n && F.v[c] == 0u && (u32)(L.v[c] >> 32) == dq);
        if (m != 0ull) r = min(r, c * 64 + __ffsll((long long)m) - 1);
    }
    return r;
}
// result key (dist << 32 | ~id) -> frontier key (dist << 32 | id)
DEV u64 fkey(u64 k) { return (k & 0xFFFFFFFF00000000ull) | (u32)(~(u32)k); }

// ---- PQ pieces -------------------------------------------------------------------------------------------
// A2 for a compile-time sub_dim: U table entries per lane per trip, every codebook load of a trip issued before the
// first use (the per-entry loop of the generic form below exposes one memory round trip per element).
template <int SD>
DEV void build_lut_sd(float *lut, const float *__restrict__ codebook, const float *q, u32 total)
{
    constexpr int U = SD <= 8 ? 8 : (SD <= 16 ? 4 : (SD <= 32 ? 2 : 1));
    for (u32 e0 = lane_id(); e0 < total; e0 += 64 * U) {
        float c[U][SD];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 e = min(e0 + 64u * u, total - 1);
            const float *src = codebook + (size_t)e * SD;
            if constexpr (SD % 4 == 0) {
#pragma unroll
                for (int i = 0; i < SD; i += 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(src + i);
                    c[u][i] = v.x; c[u][i + 1] = v.y; c[u][i + 2] = v.z; c[u][i + 3] = v.w;
                }
            } else if constexpr (SD % 2 == 0) {
#pragma unroll
                for (int i = 0; i < SD; i += 2) {
                    const float2 v = *reinterpret_cast<const float2 *>(src + i);
                    c[u][i] = v.x; c[u][i + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int i = 0; i < SD; i++) c[u][i] = src[i];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 e = e0 + 64u * u;
            if (e < total) lut[e] = pw_run_regs<SD>(c[u], q + (e >> 8) * SD);
        }
    }
}

// A2: whole table for one query, entries spread over the wave. q in original order (LDS). (The search kernels no longer
// build their tables: lut_build_kernel below fills them for the whole batch at full occupancy; this form serves the flat
// scan and the PQ-only builder's companions.)
DEV void build_lut_wave(float *lut, const float *__restrict__ codebook, const float *q, u32 m, u32 sd)
{
    const u32 total = m * 256;
    switch (sd) {
    case 2: build_lut_sd<2>(lut, codebook, q, total); return;
    case 3: build_lut_sd<3>(lut, codebook, q, total); return;
    case 4: build_lut_sd<4>(lut, codebook, q, total); return;
    case 6: build_lut_sd<6>(lut, codebook, q, total); return;
    case 8: build_lut_sd<8>(lut, codebook, q, total); return;
    case 12: build_lut_sd<12>(lut, codebook, q, total); return;
    case 16: build_lut_sd<16>(lut, codebook, q, total); return;
    case 24: build_lut_sd<24>(lut, codebook, q, total); return;
    case 32: build_lut_sd<32>(lut, codebook, q, total); return;
    case 48: build_lut_sd<48>(lut, codebook, q, total); return;
    default: break;
    }
    for (u32 e = lane_id(); e < total; e += 64) {
        const u32 jq = e >> 8;
        lut[e] = pw_run_lane(codebook + (size_t)e * sd, q + jq * sd, (int)sd);
    }
}

// A3 from a per-query table: T[j][code_j] summed in strict sub-quantiser order (fast_pq.py:325-326).
DEV float adc_from_lut(const float *lut, u32 cw, u32 jbase, float s)
{
#pragma unroll
    for (int b = 0; b < 4; b++) s = f_add(s, lut[(jbase + b) * 256 + ((cw >> (8 * b)) & 255u)]);
    return s;
}
// A3 with the table entry recomputed from the codebook in LDS: same value as A2 would have stored (A2 order).
// SD4: sub_dim == 4 (one ds_read_b128 per centroid); otherwise the generic run (numpy order for any sub_dim).
template <bool SD4>
DEV float adc_from_codebook(const float *cb, const float *q, u32 sd, u32 cw, u32 jbase, float s)
{
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const u32 jq = jbase + b;
        const u32 c = (cw >> (8 * b)) & 255u;
        float t;
        if constexpr (SD4) {
            const float4 cv = *reinterpret_cast<const float4 *>(cb + ((size_t)jq * 256 + c) * 4);
            const float4 qv = *reinterpret_cast<const float4 *>(q + jq * 4);
            t = f_add(f_add(f_add(f_add(0.0f, sqd(cv.x, qv.x)), sqd(cv.y, qv.y)), sqd(cv.z, qv.z)), sqd(cv.w, qv.w));
        } else {
            t = pw_run_lane(cb + ((size_t)jq * 256 + c) * sd, q + jq * sd, (int)sd);
        }
        s = f_add(s, t);
    }
    return s;
}

// Squared ADC of one code word (m bytes at `code`), split in two so that the caller can put other loads between
// the code-word loads and their use. The sums consume four sub-quantisers at a time in ROLLED loops: unrolling
// them lets the compiler hoist every LDS read and costs ~190 VGPRs.
DEV void adc_load_codes(uint4 &w0, uint4 &w1, uint4 &w2, uint4 &w3, const u8 *__restrict__ code, u32 m)
{
    if ((m & 15u) == 0 && m <= 64) {
        const uint4 *c4 = reinterpret_cast<const uint4 *>(code);
        const int m16 = (int)(m / 16);
        w0 = c4[0];
        w1 = w0; w2 = w0; w3 = w0;
        if (m16 > 1) w1 = c4[1];
        if (m16 > 2) w2 = c4[2];
        if (m16 > 3) w3 = c4[3];
    }
}

// A3 from the per-query table for one 16-byte code piece: all 16 table reads issued together, then the strict
// sequential sum (one LDS wait per piece instead of one per 4 entries).
DEV float adc_lut16(const float *lut, const uint4 cw, u32 jbase, float s)
{
    const u32 words[4] = { cw.x, cw.y, cw.z, cw.w };
    float t[16];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int b = 0; b < 4; b++) t[u * 4 + b] = lut[(jbase + u * 4 + b) * 256 + ((words[u] >> (8 * b)) & 255u)];
#pragma unroll
    for (int i = 0; i < 16; i++) s = f_add(s, t[i]);
    return s;
}

// A3 for the split table (variant 15): the rows of the LAST 16 sub-quantisers live in registers -- tv[jj*4 + v] of lane l
// holds T[m-16+jj][64 v + l] -- and an entry T[j][c] is fetched from lane c & 63 of register jj*4 + (c >> 6) with
// ds_bpermute (four of them per sub-quantiser, one per register, then a select on c >> 6); the other rows are read from
// LDS as before. The same 16 values per piece, added in the same strict order: the same bits. Must be called by the
// whole wavefront (an inactive source lane reads as 0).
DEV float adc_reg16(const float (&tv)[64], const uint4 cw, float s)
{
    const u32 words[4] = { cw.x, cw.y, cw.z, cw.w };
    float t[16];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int jj = u * 4 + b;
            const u32 c = (words[u] >> (8 * b)) & 255u;
            const int addr = (int)((c & 63u) << 2);
            const u32 r0 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 0]));
            const u32 r1 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 1]));
            const u32 r2 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 2]));
            const u32 r3 = (u32)__builtin_amdgcn_ds_bpermute(addr, (int)__float_as_uint(tv[jj * 4 + 3]));
            const u32 lo = (c & 64u) ? r1 : r0, hi = (c & 64u) ? r3 : r2;
            t[jj] = __uint_as_float((c & 128u) ? hi : lo);
        }
#pragma unroll
    for (int i = 0; i < 16; i++) s = f_add(s, t[i]);
    return s;
}
// m in { 32, 48, 64 }: pieces 0 .. m/16 - 2 from the LDS rows, the last piece from the register rows
DEV float adc_split16(const float *lut, const float (&tv)[64], const uint4 c0, const uint4 c1v, const uint4 c2, const uint4 c3, u32 m)
{
    const uint4 cl = (m == 32u) ? c1v : (m == 48u) ? c2 : c3;
    float s = adc_lut16(lut, c0, 0, 0.0f);
    if (m > 32u) s = adc_lut16(lut, c1v, 16, s);
    if (m > 48u) s = adc_lut16(lut, c2, 32, s);
    return adc_reg16(tv, cl, s);
}

template <bool CBLDS>
DEV float adc_compute(const float *tab, const float *q, u32 sd, const uint4 c0, const uint4 c1v, const uint4 c2,
                      const uint4 c3, const u8 *__restrict__ code, u32 m)
{
    float s = 0.0f;
    if constexpr (!CBLDS) {
        if ((m & 15u) == 0 && m <= 64) {
            s = adc_lut16(tab, c0, 0, s);
            if (m > 16) s = adc_lut16(tab, c1v, 16, s);
            if (m > 32) s = adc_lut16(tab, c2, 32, s);
            if (m > 48) s = adc_lut16(tab, c3, 48, s);
            return s;
        }
    }
    if ((m & 15u) == 0 && m <= 64) {
        const int m16 = (int)(m / 16);
        const bool sd4 = (sd == 4);
#pragma unroll 1
        for (int w = 0; w < m16; w++) {
            uint4 cw = c0;
            if (w == 1) cw = c1v;
            if (w == 2) cw = c2;
            if (w == 3) cw = c3;
#pragma unroll 1
            for (int t = 0; t < 4; t++) {
                u32 word = cw.x;
                if (t == 1) word = cw.y;
                if (t == 2) word = cw.z;
                if (t == 3) word = cw.w;
                const u32 jbase = (u32)(w * 16 + t * 4);
                if constexpr (CBLDS) {
                    if (sd4) s = adc_from_codebook<true>(tab, q, sd, word, jbase, s);
                    else s = adc_from_codebook<false>(tab, q, sd, word, jbase, s);
                } else {
                    s = adc_from_lut(tab, word, jbase, s);
                }
            }
        }
    } else if ((m & 3u) == 0) {
        const u32 *c1 = reinterpret_cast<const u32 *>(code);
#pragma unroll 1
        for (u32 w = 0; w < m / 4; w++) {
            if constexpr (CBLDS) s = adc_from_codebook<false>(tab, q, sd, c1[w], w * 4, s);
            else s = adc_from_lut(tab, c1[w], w * 4, s);
        }
    } else {
#pragma unroll 1
        for (u32 jq = 0; jq < m; jq++) {
            const u32 c = code[jq];
            if constexpr (CBLDS) s = f_add(s, pw_run_lane(tab + ((size_t)jq * 256 + c) * sd, q + jq * sd, (int)sd));
            else s = f_add(s, tab[jq * 256 + c]);
        }
    }
    return s;
}

template <bool CBLDS>
DEV float adc_lane(const float *tab, const float *q, u32 sd, const u8 *__restrict__ code, u32 m)
{
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
    adc_load_codes(w0, w1, w2, w3, code, m);
    return adc_compute<CBLDS>(tab, q, sd, w0, w1, w2, w3, code, m);
}

// max over lanes, result in every lane
DEV float wave_max(float x)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x = fmaxf(x, __shfl_xor(x, o));
    return x;
}

// ---- the kernel -------------------------------------------------------------------------------------------
// D      vector dimension (compile time: the pairwise tree is unrolled)
// FILTER M1's ADC + rerank policy            KIND   traversal metric
// NCHR   result capacity in 64-entry chunks
// NW     wavefronts (= concurrent queries) per workgroup
// CBLDS  ADC from the codebook shared in LDS (true) or from a per-query table (false)
// RB     0, or: rows per burst of the LDS-landing variant -- the stored vectors of an expansion are moved HBM -> LDS
//        by global_load_lds (no VGPR destination), RB rows in flight per wavefront with one wait, and the octets
//        read their chain-major groups back with ds_read_b128. The codebook then stays in global memory (L2) and
//        is only touched on the rare expansions whose ADC cannot be skipped (CBLDS must be true: the table
//        entries are recomputed from the codebook, wherever it lives).
// U8     the landing variant on the lossless byte copy of integer-valued vectors (D = 128): 128-byte rows, a whole
//        expansion per burst.
// QB     byte rows AND byte queries (integer-valued queries in [0, 255], checked per batch by the engine): distances
//        by v_dot4_u32_u8 -- the integer sum equals the reference's float32 sum bit for bit (see the burst code).
// A4 as a threshold on the worst distance: returns the bits of x = the largest float W >= 0 with f_mul(W, thr) <= pq.
// f_mul(., thr) is monotone, so the reference's test `pq < thr * W` (search_engine.py:390-395) holds exactly for
// W > x. `ok` is cleared if the fix-up did not reach the boundary (reported through stats.status, never silent).
// Cosine distance of the reference's in-memory M3 (compute_query_distance(metric='cosine'), vamana_graph.py:324-329 ->
// cosine_similarity_cython, cython_utils.pyx:53-70: 1 - dot / (sqrt(nx) sqrt(ny)), 0 when a norm is 0) from the squared L2
// distance the kernel already has: dot = (nx + ny - |x - y|^2) / 2, in double. The reference sums in float32 under
// -ffast-math (order unpinned, held to 1e-5 like its own test); a value that rounds below zero is returned as 0 (the
// lists order non-negative float bits).
DEV float cosine_from_l2(float l2, float nx, float ny)
{
    if (nx == 0.0f || ny == 0.0f) return 0.0f;
    const double dot = 0.5 * ((double)nx + (double)ny - (double)l2);
    const double d = 1.0 - dot / (__builtin_sqrt((double)nx) * __builtin_sqrt((double)ny));
    return d > 0.0 ? (float)d : 0.0f;
}

DEV u32 a4_threshold_bits(float pq, float thr, bool &ok) {
    if (!(pq < __builtin_inff())) return 0x7F800000u;    // never passes
    u32 cb = __float_as_uint(pq / thr);
#pragma unroll 1
    for (int it = 0; it < 8; it++) { if (f_mul(__uint_as_float(cb + 1u), thr) <= pq) cb++; else break; }
#pragma unroll 1
    for (int it = 0; it < 8; it++) { if (cb != 0u && f_mul(__uint_as_float(cb), thr) > pq) cb--; else break; }
    ok = ok && f_mul(__uint_as_float(cb), thr) <= pq && f_mul(__uint_as_float(cb + 1u), thr) > pq;
    return cb;
}

// ---- a finished query: result keys, the k best (ids, distances), tie detection for the tie-order pass, counters
template <int NCHR>
DEV void write_results(const SearchParams &p, u32 qi, int cap, u32 kmode, bool has_out, bool has_ties, const RegList<NCHR> &rk, int rn,
                       u32 steps, u32 nvisited, u32 nexact, u32 npq, u32 status, u32 ninserts, u32 npq_eval, u32 npre_hit)
{
    const int lane = lane_id();
#pragma unroll
    for (int c = 0; c < NCHR; c++) {
        const int i = c * 64 + lane;
        if (i < rn) p.res_keys[(size_t)qi * cap + i] = rk.v[c];
    }
    const int kout = min((int)p.k, rn);
    bool t = false;
#pragma unroll
    for (int c = 0; c < NCHR; c++) {
        const int i = c * 64 + lane;
        // tie detection on the sort key of the final stable sort (distance; sqrt(distance) for M3)
        const u64 nextk = wave_shl1(rk.v[c], (c + 1 < NCHR) ? readlane64(rk.v[c + 1 < NCHR ? c + 1 : c], 0) : ~0ull);
        float a = key_dist(rk.v[c]), b = key_dist(nextk);
        if (kmode == 3u) { a = f_sqrt(a); b = f_sqrt(b); }
        if (i < kout && i + 1 < rn && a == b) t = true;
        if (has_out) {
            if (i < (int)p.k) {
                p.out_ids[(size_t)qi * p.k + i] = (i < kout) ? ~(u32)rk.v[c] : 0xFFFFFFFFu;
                p.out_dist[(size_t)qi * p.k + i] = (i < kout) ? a : __uint_as_float(0x7FC00000u);
            }
        }
    }
    if (has_out) for (int i = NCHR * 64 + lane; i < (int)p.k; i += 64) {
        p.out_ids[(size_t)qi * p.k + i] = 0xFFFFFFFFu;
        p.out_dist[(size_t)qi * p.k + i] = __uint_as_float(0x7FC00000u);
    }
    const bool anyt = __ballot(t) != 0ull;
    if (lane == 0) {
        p.res_n[qi] = (u32)rn;
        if (p.out_count) p.out_count[qi] = (u32)kout;
        if (anyt && has_ties) { p.tie_list[atomicAdd(p.tie_count, 1u)] = qi; if (p.tie_flag) *p.tie_flag = 1u; }
        KStats st;
        st.steps = steps; st.visited = nvisited; st.exact = nexact; st.pq = npq; st.status = status;
        st.inserts = ninserts; st.pq_evaluated = npq_eval; st.adj_prefetch_hits = npre_hit;
        p.stats[qi] = st;
    }
}

template <int D, bool FILTER, int KIND, int NCHR, int NW, bool CBLDS, int RB = 0, bool U8 = false, bool QB = false, int TREG = 0>
DEV void search_body(const SearchParams &p)
{
    static_assert(TREG == 0 || (TREG == 16 && KIND == DIST_ADC_SQ && !FILTER && !CBLDS && RB == 0), "register table rows: the ADC-only per-query-table variant");
    constexpr bool QREG = (D <= 256);
    constexpr bool SPLIT = QREG && split_form_ok<D>();
    constexpr bool ROWLDS = RB > 0;
    static_assert(!ROWLDS || (CBLDS && SPLIT && (64 % (D / 4)) == 0), "row landing needs whole rows per instruction");
    constexpr int NP = !SPLIT ? 1 : (NW >= 16 ? 1 : ((NW >= 8 || D >= 256) ? 2 : 4));   // row passes in flight (D = 256: 32 VGPRs per pass)
    constexpr bool NEED_PQ = FILTER || KIND == DIST_ADC_SQ;
    constexpr bool SPEC_CODES = NEED_PQ && !ROWLDS;   // code words fetched beside the visited test (see there)
    // the rerank-policy kernels only ever serve M1 (squared distances, trim rule of search_engine.py:477-479): folding the
    // mode at compile time drops the other variants' branches and their scalar registers from the hot loop
    const u32 kmode = FILTER ? 1u : p.mode;
    const u32 knorm = FILTER ? 0u : p.norm;
    // (they are never used by the builder either: first-occurrence masks, outputs, tie list and the per-query ADC bounds
    // are always there)
    const bool has_first = FILTER ? true : (p.first != nullptr);
    const bool has_out = FILTER ? true : (p.out_ids != nullptr);
    const bool has_ties = FILTER ? true : (p.tie_list != nullptr);
    const bool kcos = !FILTER && KIND == DIST_EXACT && p.vnorm2 != nullptr;      // traversal metric: cosine distance (M3)
    const bool novis = !FILTER && KIND == DIST_ADC_SQ && (p.novis & 1u) != 0u;   // no visited set (SearchParams::novis)
    const bool rowpre = novis && (p.novis & 2u) != 0u && p.R <= 128u;             // ... and the next row's ids prefetched into LDS

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int j = lane & 7, oct = lane >> 3;

    // ---- LDS carve-up: [shared codebook] then one region per wavefront (all offsets multiples of 16 bytes)
    size_t off = 0;
    float *cb_lds = reinterpret_cast<float *>(smem);
    if constexpr (NEED_PQ && CBLDS && !ROWLDS) off += (size_t)256 * D * 4;
    constexpr size_t MERGE_BYTES = (size_t)NCHR * 64 * 12;   // merge scratch: NCHR*64 keys (u64) + states (u32)
    constexpr size_t ROW_BYTES = U8 ? (size_t)D : (size_t)D * 4;   // a landed row: bytes (lossless, see below) or floats
    // buckets of the visited-set grouping table ({word, lane} entries in the idle landing area / merge scratch)
    constexpr size_t VSCR = (RB > 0) ? (size_t)RB * ROW_BYTES : MERGE_BYTES;
    constexpr int VT = VSCR >= 8192 ? 1024 : VSCR >= 4096 ? 512 : VSCR >= 2048 ? 256 : VSCR >= 1024 ? 128 : 64;
    static_assert((size_t)VT * 8 <= VSCR, "grouping table must fit the scratch area");
    static_assert(!U8 || (ROWLDS && D == 128), "byte rows: the 12-wave landing variant at D = 128");
    static_assert(!QB || U8, "byte queries go with byte rows");
    static_assert(!ROWLDS || (size_t)RB * ROW_BYTES >= MERGE_BYTES, "the row landing area doubles as merge scratch");
    // the query in its original element order is needed in LDS only where table entries are recomputed per neighbour
    // from the codebook; the per-query table is built once, straight from global memory (at D = 1536, m = 32 that is
    // the 6 KiB between three and four wavefronts per CU)
    constexpr bool QORIG_LDS = NEED_PQ && CBLDS && !ROWLDS;     // (the landing variants read it from global memory: their
                                                                 // ADC runs on the rare expansions that cannot skip it)
    // "has this visited word been written by the running query?" -- a 4096-bit filter per wavefront: a word that was not
    // is stale or zero whatever it holds, so its load is skipped (more than a third of the tests). Everywhere except
    // the per-query-table variants at large D, whose LDS is full.
    constexpr int VB_BITS = (ROWLDS || CBLDS || D <= 256) ? 4096 : 0;
    // adjacency row of the predicted next pop prefetched into LDS: the byte-query variants only -- on variant 11 (16
    // floats of query per lane live) the few extra live values spill in the hot loop: 1.62 -> 2.01 ms
    // (measured on the rerank-policy-live M1 at D = 96 and on the ADC-only traversals as well, round 2: 58 % hits and
    // 3-6 % SLOWER -- those kernels are bound by the number of memory requests, not by the length of the chain)
    constexpr bool ADJPRE = QB;
    // the chain-major query copy in LDS (large D) is only read by exact distances
    constexpr bool QPERM_LDS = !QREG && KIND != DIST_ADC_SQ;
    constexpr size_t ADJPRE_BYTES = 528;    // 64 ids + 64 bit positions + the 8-byte mask word, padded to 16
    const u32 m_lds = p.m - (u32)TREG;      // table rows (sub-quantisers) kept in LDS; the last TREG rows live in registers
    const size_t per_wave = ((NEED_PQ && !CBLDS) ? (size_t)m_lds * 256 * 4 : 0) + (QORIG_LDS ? (size_t)D * 4 : 0) + (QPERM_LDS ? (size_t)D * 4 : 0) + 512 +
                            (size_t)VB_BITS / 8 + (ADJPRE ? ADJPRE_BYTES : 0) + (ROWLDS ? (size_t)RB * ROW_BYTES : MERGE_BYTES);
    unsigned char *wbase = smem + off + (size_t)wave * per_wave;
    size_t woff = 0;
    float *lut = reinterpret_cast<float *>(wbase);
    if constexpr (NEED_PQ && !CBLDS) woff += (size_t)m_lds * 256 * 4;
    float *qorig_lds = reinterpret_cast<float *>(wbase + woff);
    if constexpr (QORIG_LDS) woff += (size_t)D * 4;
    float *qperm = reinterpret_cast<float *>(wbase + woff);
    if constexpr (QPERM_LDS) woff += (size_t)D * 4;
    u32 *nb_id = reinterpret_cast<u32 *>(wbase + woff);
    woff += 256;
    float *nb_e = reinterpret_cast<float *>(wbase + woff);
    woff += 256;
    u32 *blm = reinterpret_cast<u32 *>(wbase + woff);          // [VB_BITS / 32]
    woff += (size_t)VB_BITS / 8;
    u32 *pre_buf = reinterpret_cast<u32 *>(wbase + woff);      // [64 ids][64 positions][2 mask words]
    if constexpr (ADJPRE) woff += ADJPRE_BYTES;
    float *rowbuf = reinterpret_cast<float *>(wbase + woff);   // [RB][D] landing area (ROWLDS)
    u64 *mk = reinterpret_cast<u64 *>(wbase + woff);           // merge scratch (shares the landing area: rows are
    u32 *mf = reinterpret_cast<u32 *>(mk + NCHR * 64);         // consumed before the decisions start)

    if constexpr (NEED_PQ && CBLDS && !ROWLDS) {
        const float4 *src = reinterpret_cast<const float4 *>(p.codebook);
        float4 *dst = reinterpret_cast<float4 *>(cb_lds);
        for (u32 i = threadIdx.x; i < 64u * D; i += 64 * NW) dst[i] = src[i];
        __syncthreads();
    }
    const float *pq_tab = ROWLDS ? p.codebook : (CBLDS ? cb_lds : lut);

    // Query schedule: wavefront slot s starts with query s; every later query is a ticket (one atomicAdd by lane 0,
    // broadcast with readfirstlane), so slots that drew short queries keep pulling work and a batch that is not a
    // multiple of the slot count does not end in a mostly idle last round (10 000 queries on 3 072 slots: -19 %
    // kernel time against the static s, s + slots, ... schedule). The loop is a counted loop on a scalar with the
    // ticket as a second scalar condition: the exit is an s_cbranch_scc (checked in the ISA of every variant) and
    // the trip count is bounded by nq whatever the counter holds.
    const u32 slot_id = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * NW + wave));
    const u32 nslots = gridDim.x * NW;
    // Visited set of the running query: one word per 24 bit positions, the top byte holding the STAMP of the query that
    // wrote it. A word whose stamp is not the running query's is an empty word, so nothing is ever cleared between
    // queries (the clears and the log they needed were a quarter of the kernel's memory requests); the stamp counts
    // 1..255 per slot and the slot's words are wiped once per 255 queries.
    u32 *vbm = p.vis + (size_t)slot_id * p.vis_words;
    u32 vstamp = novis ? 0u : (u32)__builtin_amdgcn_readfirstlane((int)p.vis_epoch[slot_id]);
    const int cap = (int)p.cap;
    const u32 nwords = (p.R + 63) / 64;

    u32 qi = slot_id;
    for (u32 round = 0; round < p.nq && qi < p.nq; ++round) {

        PH_BEGIN();
        // ---- per-query setup
        QueryRegs<D> qreg;
        float tv[TREG > 0 ? TREG * 4 : 1];
        const float *qorig = QORIG_LDS ? qorig_lds : p.queries + (size_t)qi * D;
        if (!(KIND == DIST_ADC_SQ && p.sdc != nullptr)) {      // (the PQ-only builder has no query vectors)
            const float *qg = p.queries + (size_t)qi * D;
            const float *qpg = p.queries_p + (size_t)qi * D;
            for (int i = lane; i < D; i += 64) {
                if constexpr (QORIG_LDS) qorig_lds[i] = qg[i];
                if constexpr (QPERM_LDS) qperm[i] = qpg[i];
            }
            if constexpr (QREG) {
                // (register variants read the original-order batch when no chain-major copy was made: dr_search_submit at
                // D <= 256 skips permute_queries_kernel; the builder hands chain-major rows only)
                if (p.queries_p != nullptr) load_query_regs<0, D, D>(qpg, j, qreg);
                else load_query_regs_orig<0, D, D>(qg, j, qreg);
            }
        }
        float qn2 = 0.0f;
        if (kcos) {
            const float *qg = p.queries + (size_t)qi * D;
            for (int i = lane; i < D; i += 64) qn2 += qg[i] * qg[i];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) qn2 += __shfl_xor(qn2, o);
        }
        if constexpr (VB_BITS > 0) { for (int i = lane; i < VB_BITS / 32; i += 64) blm[i] = 0u; }
        // Byte queries (the engine selects this variant only when EVERY component of EVERY query of the batch is an
        // integer in [0, 255], like the rows): the same 16 chain steps of lane j packed into four words, and sum q^2.
        u32 qb[4] = { 0u, 0u, 0u, 0u };
        int qq = 0;
        if constexpr (QB) {
#pragma unroll
            for (int t = 0; t < 16; t++) qb[t >> 2] |= ((u32)qreg.v[t] & 255u) << (8 * (t & 3));
            u32 s2 = 0u;
#pragma unroll
            for (int w = 0; w < 4; w++) s2 = __builtin_amdgcn_udot4(qb[w], qb[w], s2, false);
            qq = (int)s2;
            qq = octet_combine_i32(qq);
        }
        WSYNC();
        if constexpr (NEED_PQ && !CBLDS) {
            if (KIND == DIST_ADC_SQ && p.sdc != nullptr) {
                // the point's code word in whole 16-byte pieces, then the table rows of a piece, sixteen loads in flight
                // (every loop over the bytes is unrolled: no register array is indexed dynamically)
                const u8 *mycodes = p.codes + (size_t)p.build_pts[qi] * p.m;
                if ((p.m & 15u) == 0 && p.m <= 64) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        if (i * 16 < (int)m_lds) {
                            const uint4 w = reinterpret_cast<const uint4 *>(mycodes)[i];
                            const u32 words[4] = { w.x, w.y, w.z, w.w };
                            float4 r[16];
#pragma unroll
                            for (int t = 0; t < 16; t++) {
                                const u32 c = (words[t >> 2] >> (8 * (t & 3))) & 255u;
                                r[t] = reinterpret_cast<const float4 *>(p.sdc + ((size_t)(i * 16 + t) * 256 + c) * 256)[lane];
                            }
#pragma unroll
                            for (int t = 0; t < 16; t++) reinterpret_cast<float4 *>(lut + (size_t)(i * 16 + t) * 256)[lane] = r[t];
                        }
                    }
                    if constexpr (TREG > 0) {
                        // the last piece's rows go to registers: tv[t*4 + v] of lane l = T[m-16+t][64 v + l]
                        const uint4 w = reinterpret_cast<const uint4 *>(mycodes)[m_lds >> 4];
                        const u32 words[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
                        for (int t = 0; t < 16; t++) {
                            const u32 c = (words[t >> 2] >> (8 * (t & 3))) & 255u;
                            const float *row = p.sdc + ((size_t)(m_lds + t) * 256 + c) * 256 + lane;
#pragma unroll
                            for (int v = 0; v < 4; v++) tv[t * 4 + v] = row[v * 64];
                        }
                    }
                } else if constexpr (TREG == 0) {
                    for (u32 jq = 0; jq < p.m; jq++)
                        reinterpret_cast<float4 *>(lut + (size_t)jq * 256)[lane] =
                            reinterpret_cast<const float4 *>(p.sdc + ((size_t)jq * 256 + mycodes[jq]) * 256)[lane];
                }
            } else {
                // the query's table was built by lut_build_kernel (A2 for the whole batch, every CU busy): land its
                // m KiB in LDS, 1 KiB per wave instruction, all in flight, one wait
                const float *tg = p.lut_g + (size_t)qi * p.m * 256 + lane * 4;
                for (u32 e = 0; e < m_lds * 256; e += 256)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tg + e),
                        (__attribute__((address_space(3))) void *)(lut + e), 16, 0, 0);
                if constexpr (TREG > 0) {
                    // rows m-16 .. m-1 stay in registers: tv[jj*4 + v] of lane l = T[m-16+jj][64 v + l]
                    const float *tr = p.lut_g + ((size_t)qi * p.m + m_lds) * 256 + lane;
#pragma unroll
                    for (int i = 0; i < TREG * 4; i++) tv[i] = tr[i * 64];
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            WSYNC();
        }
        // Upper bound of sqrt(ADC) over ALL code words for this query: sum_j max_c T[j][c] accumulated in the same
        // j order as A3 (float addition and sqrt are monotone, so the bound holds in float arithmetic too). When
        // it is below 0.8 * (a lower bound of the worst result distance during an expansion), the rerank policy A4
        // returns True for every neighbour of that expansion and the ADC need not be evaluated -- on SIFT-scale
        // data (squared distances ~1e4-1e5, sqrt(ADC) ~1e2) that is every expansion (quirk Q1). Exact: results and
        // counters are unchanged; stats.pq_evaluated says how many ADC sums were really computed.
        float pq_ub = __uint_as_float(0x7F800000u);
        if constexpr (FILTER) pq_ub = p.pq_ub[qi];   // pq_bound_kernel, once per uploaded batch

        u32 npq_eval = 0;
        u32 steps = 0, nvisited = 0, nexact = 0, npq = 0, status = 0, ninserts = 0;
        u32 pre_id = 0xFFFFFFFFu;   // node whose adjacency row is (being) landed in pre_buf; none at query start
        u32 pre_db = 0xFFFFFFFFu;   // distance bits of that node (the prediction a better new neighbour replaces)
        u32 npre_hit = 0;
        const bool pre_on = ADJPRE && has_first && p.adjr != nullptr && p.R == 64u;
#ifdef DR_TRACE_VIS
        u32 trn = 0;
#endif
        // next stamp; after 255 queries the slot's words are wiped and the count restarts
        if (!novis && vstamp >= 255u) {
            uint4 *vb4 = reinterpret_cast<uint4 *>(vbm);
            for (u32 i = lane; i < p.vis_words / 4; i += 64) vb4[i] = make_uint4(0, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            vstamp = 0u;
        }
        vstamp++;
        const u32 vtag = vstamp << 24;
        int rn = 0, cnT = 0, tn = 0;   // results; live (unexpanded, untrimmed) result entries; tie side list
        u32 junk = 0;   // evicted frontier entries that are worse than every result (only their count matters)
        RegList<NCHR> rk;
        FlagList<NCHR> fl;
        RegList<1> tl;   // frontier keys of entries evicted from the results while tied with the worst distance
#pragma unroll
        for (int c = 0; c < NCHR; c++) { rk.v[c] = ~0ull; fl.v[c] = 0u; }
        tl.v[0] = ~0ull;
        u64 *qlog = p.log + (size_t)qi * p.logcap;

        // ---- start node (search_engine.py:416-426)
        {
            const u32 start = p.medoid;
            if (lane == 0 && !novis) {
                const u32 sp = p.adjr ? p.medoid_pos : start;
                const u32 sw = __umulhi(sp, 0xAAAAAAABu) >> 4;            // sp / 24
                vbm[sw] = vtag | (1u << (sp - sw * 24u));                // first word of this query: whatever was there is stale
                if constexpr (VB_BITS > 0) { const u32 bh = (sw * 0x9E3779B1u) >> 20; blm[bh >> 5] = 1u << (bh & 31); }
            }
            nvisited = 1;
            float d0;
            if constexpr (KIND == DIST_ADC_SQ) {
                if constexpr (TREG > 0) {
                    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
                    adc_load_codes(w0, w1, w2, w3, p.codes + (size_t)start * p.m, p.m);
                    d0 = adc_split16(pq_tab, tv, w0, w1, w2, w3, p.m);
                } else {
                    d0 = adc_lane<CBLDS>(pq_tab, qorig, p.sd, p.codes + (size_t)start * p.m, p.m);
                }
                npq++;
            } else {
                d0 = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)start * D, &qreg, qperm, j);
                d0 = __uint_as_float(readlane32(__float_as_uint(d0), 0));
                if (kcos) d0 = cosine_from_l2(d0, p.vnorm2[start], qn2);
                if (knorm) d0 = f_sqrt(d0);
                nexact++;
            }
            const u32 db = __float_as_uint(d0);
            u64 dr; u32 df; bool dd;
            rn = list_insert_f<NCHR>(rk, fl, 0, cap, ((u64)db << 32) | (u32)(~start), dr, df, dd);
            cnT = 1;
            if (lane == 0 && p.logcap > 0) qlog[0] = ((u64)db << 32) | start;
            ninserts = 1;
        }
        PH(0);

        // ---- main loop
        while ((cnT + tn > 0 || junk > 0) && steps < p.max_steps) {
            if ((u64)steps > p.N + 8) { status |= DR_ST_INTERNAL; break; }   // every expansion pops a distinct node
            steps++;
            if (cnT + tn == 0) break;   // only junk left: the reference pops it and stops (it is worse than W)
            // heappop(candidates): the smaller of the first live result entry and the side-list head
            u64 ckey;
            {
                const int ia = frontier_first<NCHR>(rk, fl, rn);
                const u64 ka = (ia >= 0) ? fkey(list_get<NCHR>(rk, ia)) : ~0ull;
                const u64 kb = (tn > 0) ? readlane64(tl.v[0], 0) : ~0ull;
                if (ka <= kb) { ckey = ka; flag_or<NCHR>(fl, ia, 1u); cnT--; }
                else { ckey = kb; list_pop_front<1>(tl); tn--; }
            }
            const float cd = key_dist(ckey);
            const u32 cur = (u32)ckey;
            // Was this node's adjacency row prefetched during the previous expansion? (static graph data: valid whatever
            // happened to the lists since)
            const bool pre_hit = ADJPRE && pre_on && cur == pre_id;
            if (pre_hit) npre_hit++;
            {
                const float W = key_dist(list_get<NCHR>(rk, rn - 1));
                bool stop;
                if (kmode == 3u) stop = (cd > W) && (rn == cap);
                else if (kmode == 4u) stop = (cd > W);
                else stop = (rn >= cap) && (cd > W);
                if (stop) break;
            }
            PH(1);

            // (no visited set) the ids of this node's row, landed in LDS by the previous expansion if it predicted this pop; then the
            // same for the next pop: the best frontier entry that is left now (this expansion's neighbours may still beat it). The
            // landing area is the 512-byte id / distance staging of the exact traversals, idle in an ADC-only kernel. Static
            // graph data: valid whatever happens to the lists meanwhile.
            u32 *rowpre_buf = nb_id;
            bool rowpre_hit = false;
            u32 hit_ids0 = 0u, hit_ids1 = 0u;
            if (rowpre) {
                rowpre_hit = cur == pre_id;
                if (rowpre_hit) {
                    npre_hit++;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    hit_ids0 = rowpre_buf[lane]; hit_ids1 = rowpre_buf[64 + lane];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // read before the next prefetch reuses the area
                }
                pre_id = 0xFFFFFFFFu;
                const int ia2 = frontier_first<NCHR>(rk, fl, rn);
                const u64 ka2 = (ia2 >= 0) ? fkey(list_get<NCHR>(rk, ia2)) : ~0ull;
                const u64 kb2 = (tn > 0) ? readlane64(tl.v[0], 0) : ~0ull;
                const u64 kn = ka2 <= kb2 ? ka2 : kb2;
                if (kn != ~0ull) {
                    pre_id = (u32)kn;
                    const u32 *gi = p.adj + (size_t)pre_id * p.R;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gi + min((u32)lane, p.R - 1)),
                        (__attribute__((address_space(3))) void *)rowpre_buf, 4, 0, 0);
                    if (p.R > 64u)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gi + min(64u + (u32)lane, p.R - 1)),
                            (__attribute__((address_space(3))) void *)(rowpre_buf + 64), 4, 0, 0);
                }
            }
            for (u32 cbase = 0; cbase < p.R; cbase += 64) {
                const u32 slot = cbase + lane;
                // One memory round trip for the whole row: ids, bit positions and the mask/degree word are loaded
                // through pointers selected up front, so no load waits behind a branch on another load's register.
                // The visited bitmap is indexed by a locality-preserving bit order (neighbours of one node share a
                // few cache lines instead of touching 64 different ones); the positions travel with the row.
                const u32 *idrow = p.adj + (size_t)cur * p.R;
                const u32 *posrow = p.adjr ? p.adjr + (size_t)cur * p.R : idrow;
                // (the degree array of the build mode is read as the aligned 8 bytes around deg[cur]: same load shape
                // as the mask word, so the compiler keeps one straight-line group of loads; deg has N + 1 entries)
                const u64 *auxp = has_first ? p.first + (size_t)cur * nwords + (cbase >> 6)
                                          : reinterpret_cast<const u64 *>(p.deg + (cur & ~1u));
                const u32 sl = min(slot, p.R - 1);
                u32 nbid_l, nbpos_l;
                u64 aux_w;
                uint4 cw0 = make_uint4(0, 0, 0, 0), cw1 = cw0, cw2 = cw0, cw3 = cw0;
                // No visited set AND inline neighbour codes: where a slot's code word lives depends on the popped node alone, so
                // the row's code words leave WITH its ids and mask -- ONE memory round trip per expansion instead of two (ids, then
                // the code words of those ids), R*m contiguous bytes instead of R scattered gathers. (A slot that turns out to be a
                // pad or a repeat costs its 32 bytes of a line that is read anyway.)
                const bool codes_with_row = SPEC_CODES && novis && p.nbcodes != nullptr;
                if (ADJPRE && pre_hit) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    nbid_l = pre_buf[lane]; nbpos_l = pre_buf[64 + lane];
                    aux_w = *reinterpret_cast<const u64 *>(pre_buf + 128);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // read before the next prefetch reuses the area
                } else if (rowpre_hit) { nbid_l = cbase ? hit_ids1 : hit_ids0; nbpos_l = 0u; aux_w = auxp[0]; }
                else { nbid_l = idrow[sl]; nbpos_l = novis ? 0u : posrow[sl]; aux_w = auxp[0]; }
                if constexpr (SPEC_CODES) { if (codes_with_row) adc_load_codes(cw0, cw1, cw2, cw3, p.nbcodes + ((size_t)cur * p.R + sl) * p.m, p.m); }
                // Predict the next pop -- the best frontier entry that is left now (this expansion's neighbours may still
                // beat it) -- and land ITS adjacency row in LDS: no VGPR destination, nobody waits for it, and when the
                // prediction holds the next expansion starts without its first global round trip.
                if constexpr (ADJPRE) {
                    pre_id = 0xFFFFFFFFu; pre_db = 0xFFFFFFFFu;
                    if (pre_on) {
                        const int ia2 = frontier_first<NCHR>(rk, fl, rn);
                        const u64 ka2 = (ia2 >= 0) ? fkey(list_get<NCHR>(rk, ia2)) : ~0ull;
                        const u64 kb2 = (tn > 0) ? readlane64(tl.v[0], 0) : ~0ull;
                        const u64 kn = ka2 <= kb2 ? ka2 : kb2;
                        if (kn != ~0ull) {
                            pre_id = (u32)kn; pre_db = (u32)(kn >> 32);
                            const u32 *gi = p.adj + (size_t)pre_id * 64 + lane, *gp = p.adjr + (size_t)pre_id * 64 + lane;
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gi,
                                (__attribute__((address_space(3))) void *)pre_buf, 4, 0, 0);
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                (__attribute__((address_space(3))) void *)(pre_buf + 64), 4, 0, 0);
                            if (lane < 2) {
                                const u32 *gm = reinterpret_cast<const u32 *>(p.first + (size_t)pre_id) + lane;
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gm,
                                    (__attribute__((address_space(3))) void *)(pre_buf + 128), 4, 0, 0);
                            }
                        }
                    }
                }
                const u32 nbid = slot < p.R ? nbid_l : 0xFFFFFFFFu;
                const u32 nbpos = slot < p.R ? nbpos_l : 0xFFFFFFFFu;
                const u64 aux = has_first ? aux_w : (u64)(u32)(aux_w >> ((cur & 1u) * 32));
                bool active;
                if (has_first) active = ((aux >> lane) & 1ull) != 0ull;
                else active = slot < min((u32)aux, p.R) && nbid != 0xFFFFFFFFu;
                PH(2);
                // visited test-and-set: one atomic round trip; duplicates inside a row were removed by `first`
#ifdef DR_TRACE_VIS
                // diagnostic build: the bit positions tested by the first queries, one 0xFFFFFFFF marker per expansion
                if (p.phase && qi < 256u) {
                    u32 *tr = reinterpret_cast<u32 *>(p.phase) + (size_t)qi * 16384;
                    const u64 am = __ballot(active);
                    const u32 na = (u32)__popcll(am);
                    if (trn + na + 1 < 16384u) {
                        if (lane == 0) tr[1 + trn] = 0xFFFFFFFFu;
                        if (active) tr[2 + trn + __popcll(am & lanemask_lt())] = nbpos;
                        trn += na + 1;
                        if (lane == 0) tr[0] = trn;
                    }
                }
#endif
                bool isnew = false;
                // ---- visited test-and-set WITHOUT atomics and without clears.
                // The slot's set is private to this wavefront, so the test is a plain load (sc1: served by the L2, never by
                // a stale L1 line) and the set a plain store; a word stamped by an earlier query reads as empty. What an
                // atomic gave for free -- several lanes of one instruction hitting the same word -- is done in LDS: lanes
                // are grouped by word (hash table of {word, lane} entries in the idle landing / merge scratch: a lane
                // that reads back its own word has found its group's leader; groups that lost their bucket to another
                // word retry with the next hash, then one at a time), the group's bits are OR-ed into the leader's LDS
                // word, and only leaders store. (Memory-side atomics and the per-query clears measured slower: the
                // kernel is bound by the chip's RATE of random requests, ~55 G/s, and this form issues the fewest.)
                const u32 vw = __umulhi(nbpos, 0xAAAAAAABu) >> 4;        // nbpos / 24
                const u32 vbit = 1u << (nbpos - vw * 24u);
                u32 vraw = 0u;
                const u32 hsh = vw * 0x9E3779B1u;
                const u32 bh = hsh >> 20;                                // filter bit of this word
                bool vneed = active && !novis;
                if constexpr (VB_BITS > 0) vneed = vneed && ((blm[bh >> 5] >> (bh & 31)) & 1u) != 0u;
                if (vneed) vraw = __hip_atomic_load(&vbm[vw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // Code words of the row's neighbours fetched BESIDE the visited test instead of after it (one dependent
                // round trip less per expansion) whenever their ADC may be needed: always for the ADC traversals; for M1
                // unless the bound that lets the expansion skip the ADC already holds with every active lane counted as
                // new (a superset of the final need_adc below, so the registers are loaded whenever they are used). The
                // few code words of neighbours that turn out to be visited are wasted requests (~15 %).
                // (inline neighbour codes: the slot's code word sits in the block beside the adjacency row -- one coalesced
                // read of R*m bytes per expansion instead of a scattered m-byte gather per neighbour)
                const u8 *mycode = p.nbcodes ? p.nbcodes + ((size_t)cur * p.R + min(slot, p.R - 1)) * p.m : p.codes + (size_t)nbid * p.m;
                if constexpr (SPEC_CODES) {
                    bool spec = true;
                    if constexpr (FILTER) {
                        const int nact = __popcll(__ballot(active));
                        if (rn + nact <= (int)p.L) spec = false;
                        else if (NCHR == 1 && rn == cap && (p.novis & 8u) != 0u) spec = false;       // ("ask later": the code words are fetched if the sharper test fails)
                        else if (rn == cap && cap - 1 - nact >= 0) {
                            const float Wlow = key_dist(list_get<NCHR>(rk, cap - 1 - nact));
                            if (pq_ub < f_mul(Wlow, 0.8f)) spec = false;
                        }
                    }
                    // (no visited set: the gather only needs a real id, not the first-occurrence mask -- on a prefetch hit the mask
                    // word is still on its way while the code words are requested)
                    const bool codelane = novis ? (slot < p.R && nbid < (u32)p.N) : active;
                    if (spec && codelane && !codes_with_row) adc_load_codes(cw0, cw1, cw2, cw3, mycode, p.m);
                }
                if (novis) isnew = active;      // every neighbour is scored; the list itself says which are already in it (decisions)
                else {
                    u32 vldr = (u32)lane;
                    u32 *vacc = reinterpret_cast<u32 *>(nb_e);         // idle until the distances are written
                    u64 *vtab = reinterpret_cast<u64 *>(mk);           // idle until the rows land / the merge
                    vacc[lane] = 0u;
                    bool vpend = active;
                    u64 pendm = __ballot(vpend);
#pragma unroll 1
                    for (int vr = 0; vr < 2 && pendm != 0ull; vr++) {
                        const u32 hb = (hsh >> (vr ? 8 : 20)) & (u32)(VT - 1);
                        if (vpend) vtab[hb] = ((u64)vw << 32) | (u32)lane;
                        WSYNC();
                        const u64 ent = vtab[hb];
                        WSYNC();
                        if (vpend && (u32)(ent >> 32) == vw) { vldr = (u32)ent; vpend = false; atomicOr(&vacc[vldr], vbit); }
                        pendm = __ballot(vpend);
                    }
#pragma unroll 1
                    while (pendm != 0ull) {      // leftovers (two hash collisions in a row): one group per trip
                        const int f = __ffsll((long long)pendm) - 1;
                        const u32 wf = readlane32(vw, f);
                        if (vpend && vw == wf) { vldr = (u32)f; vpend = false; atomicOr(&vacc[f], vbit); }
                        pendm = __ballot(vpend);
                    }
                    WSYNC();
                    const u32 gbits = vacc[lane];
                    WSYNC();
                    const u32 vold = ((vraw >> 24) == vstamp) ? (vraw & 0x00FFFFFFu) : 0u;
                    isnew = active && (vold & vbit) == 0u;
                    // (store -> later load of the same word: both are served by the L2 in this wave's issue order, and
                    // every expansion that stored has since waited for row loads issued after its stores)
                    if (active && vldr == (u32)lane && (gbits & ~vold) != 0u) {
                        vbm[vw] = vtag | vold | gbits;
                        if constexpr (VB_BITS > 0) atomicOr(&blm[bh >> 5], 1u << (bh & 31));
                    }
                }
                const u64 newmask = __ballot(isnew);
                const int nnew = __popcll(newmask);
                if (nnew == 0) continue;
                nvisited += nnew;
                PH(3);

                // PQ kernels: every lane stays on its own neighbour (stored order = lane order; the decisions below only
                // need that order) and only the rows to fetch are compacted, once, after the ADC has said which are
                // needed. Exact traversals: every new neighbour is fetched, so the new ids are compacted right here and
                // lanes 0 .. nnew-1 take them (the sparse form costs those kernels 18-25 VGPRs, i.e. a wavefront per
                // SIMD at D = 96).
                u32 myid = nbid;
                if constexpr (!NEED_PQ) {
                    if (isnew) nb_id[__popcll(newmask & lanemask_lt())] = nbid;
                    WSYNC();
                    myid = nb_id[lane < nnew ? lane : 0];
                    isnew = lane < nnew;
                }
                float pq_d = 0.0f, e = 0.0f;
                // Is the ADC value of this expansion's neighbours needed at all? (A4 is provably True for all of
                // them when the list cannot fill up during the expansion, or when pq_ub clears the threshold for
                // the smallest worst-distance the expansion can reach: at most nnew results get replaced.)
                bool need_adc = NEED_PQ;
                if constexpr (FILTER) {
                    if (rn + nnew <= (int)p.L) need_adc = false;
                    else if (rn == cap && cap - 1 - nnew >= 0) {
                        const float Wlow = key_dist(list_get<NCHR>(rk, cap - 1 - nnew));
                        if (pq_ub < f_mul(Wlow, 0.8f)) need_adc = false;
                    }
                }
                // Round 5, "ask later" (SearchParams::novis bit 3, set by the engine while the index is not known to keep the policy busy): when the
                // list is full and neither test above can prove the policy true -- lists shorter than a row, the API's L = 20 -- every new row is
                // fetched FIRST, as if it were proven. With the exact distances known a sharper test exists: while the list is full its worst
                // distance only shrinks, so a neighbour is accepted only if it is below today's worst -- c' of them at most (those lanes) --, at
                // most c' results are replaced and the worst distance stays >= list[cap - 1 - c'] for the whole row. If pq_ub clears THAT
                // threshold the policy was true for every neighbour and the ADC is never evaluated; if not, it is evaluated now (one more
                // dependent round trip, on the rare row) and the decisions count the policy as ever -- with every new row scored, which they
                // never look at for lanes the policy rejects. Exact either way; SIFT-scale data at L = 20: the kernel time halves.
                bool late_adc = false;
                // (lists of at most 64 entries only -- the one list-size class whose lists a row can replace whole; the longer classes, the bench
                // kernel among them, keep their code and their registers)
                if constexpr (FILTER && NCHR == 1) { if (need_adc && (p.novis & 8u) != 0u && rn == cap) { late_adc = true; need_adc = false; } }
                bool all_pass = !need_adc;
                // A4 live: ADC first. It turns into a per-neighbour threshold x on the worst result distance (A4 passes
                // iff W > x), and while the list is full W only shrinks during the expansion, so a neighbour with
                // x >= W now can never pass: its stored vector is not fetched at all (the reference would not score it
                // either). The rows to score are compacted in stored order.
                if constexpr (NEED_PQ && !SPEC_CODES) { if (need_adc && isnew) adc_load_codes(cw0, cw1, cw2, cw3, mycode, p.m); }
                float adc_s = 0.0f;
                u32 xbits = 0u;               // bits of the A4 threshold; 0 when A4 is proven true
                bool rowlane = isnew;
                if constexpr (FILTER) {
                    if (need_adc) {
                        if (isnew) adc_s = adc_compute<CBLDS>(pq_tab, qorig, p.sd, cw0, cw1, cw2, cw3, mycode, p.m);
                        pq_d = f_sqrt(adc_s);   // asymmetric_distance = sqrt (fast_pq.py:330-333)
                        bool ok = true;
                        xbits = a4_threshold_bits(pq_d, p.policy == 0u ? 1.2f : 0.8f, ok);
                        if (__ballot(isnew && !ok) != 0ull) status |= DR_ST_INTERNAL;
                        if (rn == cap) {
                            const u32 W0b = (u32)(list_get<NCHR>(rk, rn - 1) >> 32);
                            rowlane = isnew && xbits < W0b;
                        }
                        PH(4);
                    }
                }
                int nrow = nnew, myrow = lane;
                if constexpr (NEED_PQ) {
                    const u64 rowmask = __ballot(rowlane);
                    nrow = __popcll(rowmask);
                    myrow = __popcll(rowmask & lanemask_lt());
                    if constexpr (KIND != DIST_ADC_SQ) {
                        if (rowlane) nb_id[myrow] = myid;
                        WSYNC();
                    }
                }
                if constexpr (KIND != DIST_ADC_SQ) {
                    if constexpr (ROWLDS && U8) {
                        // Byte rows. SIFT-type descriptors are integers in [0, 255] stored as float32: the engine
                        // keeps a second, lossless copy as bytes (built only if EVERY component of the index
                        // qualifies), laid out so that lane j of an octet finds the 16 steps of its accumulator
                        // chain in one 16-byte piece (position j*16 + t = element 8t + j). A row is 128 bytes
                        // instead of 512, a wavefront instruction lands 8 rows, and the whole expansion is ONE
                        // burst. v_cvt_f32_ubyte gives back exactly the stored float, so the sum below is the same
                        // arithmetic on the same values as the float path: bit-identical distances.
                        const unsigned char *rowbuf8 = reinterpret_cast<const unsigned char *>(rowbuf);
                        for (int b0 = 0; b0 < nrow; b0 += RB) {
                            const int nb = min(RB, nrow - b0);
                            u32 rid[RB / 8];
#pragma unroll
                            for (int r = 0; r < RB; r += 8) rid[r / 8] = nb_id[min(b0 + r + (lane >> 3), nrow - 1)];
#pragma unroll
                            for (int r = 0; r < RB; r += 8) {
                                // (DR_BURST_PIPE: every instruction of the burst is issued -- the ones past the last row re-read it, an L2
                                // hit -- so that the waits below are compile-time counts)
                                if (DR_BURST_PIPE || r < nb) {
                                    const u8 *g = p.vec8 + (size_t)rid[r / 8] * D + (lane & 7) * 16;
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                        (__attribute__((address_space(3))) void *)(const_cast<unsigned char *>(rowbuf8) + (size_t)r * D), 16, 0, 0);
                                }
                            }
                            if (!DR_BURST_PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            auto pass = [&](const int r8) {
                                const int row = min(r8 + oct, nb - 1);
                                const uint4 w = *reinterpret_cast<const uint4 *>(rowbuf8 + (size_t)row * D + j * 16);
                                const u32 words[4] = { w.x, w.y, w.z, w.w };
                                float ev;
                                if constexpr (QB) {
                                    // Integers in [0, 255] on both sides: every difference, square and partial sum
                                    // of the reference's float32 computation is an integer below 128 * 255^2 < 2^24,
                                    // i.e. exact in ANY order, so its result is float(sum (x - q)^2) -- computed here
                                    // as x.x - 2 q.x + q.q with 8 byte dot products instead of 63 float operations.
                                    u32 qx = 0u, xx = 0u;
#pragma unroll
                                    for (int w4 = 0; w4 < 4; w4++) {
                                        qx = __builtin_amdgcn_udot4(words[w4], qb[w4], qx, false);
                                        xx = __builtin_amdgcn_udot4(words[w4], words[w4], xx, false);
                                    }
                                    ev = (float)(octet_combine_i32((int)xx - 2 * (int)qx) + qq);
                                } else {
                                    float r = 0.0f;
#pragma unroll
                                    for (int t = 0; t < 16; t++) {
                                        const float v = (float)((words[t >> 2] >> (8 * (t & 3))) & 255u);
                                        const float sq = sqd(v, qreg.v[t]);
                                        r = (t == 0) ? sq : f_add(r, sq);
                                    }
                                    ev = octet_combine(r);
                                }
                                if (knorm) ev = f_sqrt(ev);
                                if (j == 0 && r8 + oct < nb) nb_e[b0 + r8 + oct] = ev;
                            };
                            if constexpr (DR_BURST_PIPE != 0) {
                                // pass i scores rows 8i .. 8i + 7 as soon as THEIR instruction has landed (loads return in order: at most
                                // RB/8 - 1 - i newer ones may still be in flight), while the rest of the burst is still arriving
                                static_assert(RB == 64, "eight instructions per burst");
#define DR_BURST_PASS(I) if ((I) * 8 < nb) { wait_vmcnt<7 - (I)>(); pass((I) * 8); }
                                DR_BURST_PASS(0) DR_BURST_PASS(1) DR_BURST_PASS(2) DR_BURST_PASS(3) DR_BURST_PASS(4) DR_BURST_PASS(5) DR_BURST_PASS(6) DR_BURST_PASS(7)
#undef DR_BURST_PASS
                                wait_vmcnt<0>();      // (the instructions past the last row)
                            } else {
                                for (int r8 = 0; r8 < nb; r8 += 8) pass(r8);
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                    } else if constexpr (ROWLDS) {
                        constexpr int LPR = D / 4;          // lanes (16-byte pieces) per row
                        constexpr int RPI = 64 / LPR;       // rows per wave instruction (1 KiB)
                        for (int b0 = 0; b0 < nrow; b0 += RB) {
                            const int nb = min(RB, nrow - b0);
                            // all rows of the burst in flight, no VGPR destination. The row ids of every instruction
                            // are read first (one LDS wait for the burst instead of one per instruction).
                            u32 rid[RB / RPI];
#pragma unroll
                            for (int r = 0; r < RB; r += RPI) rid[r / RPI] = nb_id[min(b0 + r + lane / LPR, nrow - 1)];
#pragma unroll
                            for (int r = 0; r < RB; r += RPI) {
                                if (r < nb) {
                                    const float *g = p.vecp + (size_t)rid[r / RPI] * D + (lane % LPR) * 4;
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                        (__attribute__((address_space(3))) void *)(rowbuf + (size_t)r * D), 16, 0, 0);
                                }
                            }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            for (int r8 = 0; r8 < nb; r8 += 8) {
                                const int row = min(r8 + oct, nb - 1);
                                RowRegs<D> rr;
                                row_load<0, D, D>(rowbuf + (size_t)row * D, j, rr);
                                float ev = row_reduce<0, D, D>(rr, qreg);
                                if (knorm) ev = f_sqrt(ev);
                                if (j == 0 && r8 + oct < nb) nb_e[b0 + r8 + oct] = ev;
                            }
                            // the landing area is rewritten by the next burst: its reads above have been consumed
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        }
                    } else if constexpr (SPLIT) {
                        // rolling ring of NP row buffers: pass r is reduced while passes r+1 .. r+NP-1 are in flight
                        RowRegs<D> rr[NP];
#pragma unroll
                        for (int r = 0; r < NP; r++) {
                            if (r * 8 < nrow) {
                                const int idx = min(r * 8 + oct, nrow - 1);
                                row_load<0, D, D>(p.vecp + (size_t)nb_id[idx] * D, j, rr[r]);
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 8; r++) {
                            if (r * 8 < nrow) {
                                float ev = row_reduce<0, D, D>(rr[r % NP], qreg);
                                if (knorm) ev = f_sqrt(ev);
                                if (j == 0 && r * 8 + oct < nrow) nb_e[r * 8 + oct] = ev;
                                if ((r + NP) * 8 < nrow) {
                                    const int idx = min((r + NP) * 8 + oct, nrow - 1);
                                    row_load<0, D, D>(p.vecp + (size_t)nb_id[idx] * D, j, rr[r % NP]);
                                }
                            }
                        }
                    } else if constexpr (ChunkCfg<D>::ok && !QREG) {
                        // large rows (D = 768, 960, 1536): the software pipeline of numerics.hpp -- NBUF - 1 chunks of the
                        // running row (or of the next pass's row) in flight behind the chunk being reduced
                        ChunkRegs<D> cbuf[ChunkCfg<D>::NBUF];
                        const int npass = (nrow + 7) >> 3;
                        if (npass > 0) {
                            const float *rp = p.vecp + (size_t)nb_id[min(oct, nrow - 1)] * D;
                            chunk_prologue<D, 0>(rp, j, cbuf);
                            float cres[ChunkCfg<D>::NC];
                            // one loop, two bodies (a pass with / without a next pass to prefetch for): the loop header
                            // is only reached with the same NBUF - 1 chunks in flight, so the wait counts stay exact
#pragma unroll 1
                            for (int r0 = 0; r0 < npass; r0++) {
                                if (r0 + 1 < npass) {
                                    const float *rnext = p.vecp + (size_t)nb_id[min((r0 + 1) * 8 + oct, nrow - 1)] * D;
                                    chunk_pass<D, 0, true>(rp, rnext, j, cbuf, qperm, cres);
                                    rp = rnext;
                                } else {
                                    chunk_pass<D, 0, false>(rp, rp, j, cbuf, qperm, cres);
                                }
                                float ev = chunk_tree<0, ChunkCfg<D>::NC>(cres);
                                if (knorm) ev = f_sqrt(ev);
                                if (j == 0 && r0 * 8 + oct < nrow) nb_e[r0 * 8 + oct] = ev;
                            }
                        }
                    } else {
                        for (int r0 = 0; r0 * 8 < nrow; r0++) {
                            const int idx = min(r0 * 8 + oct, nrow - 1);
                            float ev = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)nb_id[idx] * D, &qreg, qperm, j);
                            if (knorm) ev = f_sqrt(ev);
                            if (j == 0 && r0 * 8 + oct < nrow) nb_e[r0 * 8 + oct] = ev;
                        }
                    }
                    WSYNC();
                    e = rowlane ? nb_e[myrow] : __builtin_inff();
                    if (kcos && rowlane) e = cosine_from_l2(e, p.vnorm2[myid], qn2);
                    if constexpr (FILTER) {
                        if constexpr (NCHR == 1) if (late_adc) {
                            const u32 Wb = (u32)(list_get<NCHR>(rk, rn - 1) >> 32);
                            const int cpr = __popcll(__ballot(isnew && __float_as_uint(e) < Wb));
                            bool proven = false;
                            if (cap - 1 - cpr >= 0) proven = pq_ub < f_mul(key_dist(list_get<NCHR>(rk, cap - 1 - cpr)), 0.8f);
                            if (!proven) {
                                if (isnew) {
                                    adc_load_codes(cw0, cw1, cw2, cw3, mycode, p.m);
                                    adc_s = adc_compute<CBLDS>(pq_tab, qorig, p.sd, cw0, cw1, cw2, cw3, mycode, p.m);
                                }
                                pq_d = f_sqrt(adc_s);
                                bool ok = true;
                                xbits = a4_threshold_bits(pq_d, p.policy == 0u ? 1.2f : 0.8f, ok);
                                if (__ballot(isnew && !ok) != 0ull) status |= DR_ST_INTERNAL;
                                need_adc = true; all_pass = false;
                            }
                        }
                        npq += nnew;            // the reference counts one PQ distance per new neighbour
                        if (need_adc) npq_eval += nnew;
                    }
                    if constexpr (ADJPRE) {
                        // Second chance for the prediction (round 4, A/B switch, OFF by default): the row landed above belongs to the best
                        // frontier entry that was left when this expansion began -- 42 % of the time one of THIS expansion's neighbours is
                        // closer and is popped instead. Its distance is known now: land ITS row over the stale one (same area; a
                        // wavefront's loads return in order, so the later one wins) and let the decision pass hide the round trip.
                        // Measured: hits 58 -> 97 % of the expansions and the kernel 1 % SLOWER -- the length of an expansion's chain of
                        // round trips is not what bounds this kernel.
                        if (pre_on && (p.novis & 4u) != 0u) {
                            const u32 eb = rowlane ? __float_as_uint(e) : 0xFFFFFFFFu;      // (distances are >= 0: the bits order like the values)
                            const u32 mb = wave_min_u32(eb);
                            if (mb < pre_db) {
                                const int fb = __ffsll((long long)__ballot(eb == mb)) - 1;
                                pre_id = readlane32(myid, fb); pre_db = mb;
                                const u32 *gi = p.adj + (size_t)pre_id * 64 + lane, *gp = p.adjr + (size_t)pre_id * 64 + lane;
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gi,
                                    (__attribute__((address_space(3))) void *)pre_buf, 4, 0, 0);
                                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gp,
                                    (__attribute__((address_space(3))) void *)(pre_buf + 64), 4, 0, 0);
                                if (lane < 2) {
                                    const u32 *gm = reinterpret_cast<const u32 *>(p.first + (size_t)pre_id) + lane;
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gm,
                                        (__attribute__((address_space(3))) void *)(pre_buf + 128), 4, 0, 0);
                                }
                            }
                        }
                    }
                } else {
                    if constexpr (!SPEC_CODES) { if (isnew) adc_load_codes(cw0, cw1, cw2, cw3, mycode, p.m); }
                    if constexpr (TREG > 0) {
                        // every lane takes part (ds_bpermute reads 0 from inactive source lanes); lanes without a new
                        // neighbour look up whatever their code registers hold and are ignored below
                        adc_s = adc_split16(pq_tab, tv, cw0, cw1, cw2, cw3, p.m);
                    } else {
                        if (isnew) adc_s = adc_compute<CBLDS>(pq_tab, qorig, p.sd, cw0, cw1, cw2, cw3, mycode, p.m);
                    }
                    e = adc_s;
                    npq += nnew; npq_eval += nnew;
                    PH(4);
                }
                PH(5);

                // ---- decisions
                // The reference walks the new neighbours in stored order: i is scored iff A4 passes against the worst
                // distance W_i in effect at its position, and accepted iff moreover e_i < W_i (or the list is not
                // full); W_i = the cap-th smallest of S_i = old list + neighbours accepted before i. With
                // x_i = A4's threshold (0 when A4 is proven true) and t_i = max(e_i, x_i):
                //     scored_i   <=>  #(S_i <= x_i) < cap          accepted_i <=>  #(S_i <= t_i) < cap.
                // No walk is needed. (1) #(old <= .) is a binary search over the list, staged in the merge scratch.
                // (2) ONE pass over the candidate lanes records, as bit masks per lane, which earlier candidates j have
                // e_j <= t_i (Mt) / <= x_i (Mx), and which candidate keys lie below this lane's key or list keys.
                // (3) accepted = the unique fixed point of  A = { i : ub_t_i + |Mt_i & A| < cap }: the status of i
                // depends on earlier lanes only, so iterating from "all candidates" is exact after at most
                // (#candidates) rounds and in practice after two (without A4 the first round already is the answer:
                // a neighbour rejected earlier had e_j >= W_j >= W_i > e_i). Each round is a popcount and a ballot.
                // (4) the merge ranks are popcounts of the key masks against A. ONE merge through LDS follows.
                {
                    const bool count_pass = FILTER && !all_pass;
                    if constexpr (KIND != DIST_ADC_SQ) { if (!count_pass) nexact += nnew; }
                    const u32 ebits = __float_as_uint(e);
                    const u32 tbits = max(ebits, xbits);
                    const bool full0 = (rn == cap);
                    const u32 W0b = (u32)(list_get<NCHR>(rk, rn - 1) >> 32);
                    // lanes that can still be scored (superset of those that can be accepted)
                    const u64 cm = __ballot(isnew && (!full0 || (count_pass ? xbits : ebits) < W0b));
                    const u64 mykey = ((u64)ebits << 32) | (u32)(~myid);
                    int na = 0;
                    u64 accmask = 0ull;
                    int rT = 0, rA = 0;     // merge ranks: list keys below an accepted key, accepted keys below it
                    u32 sT[NCHR];           // accepted keys below a list key
#pragma unroll
                    for (int ch = 0; ch < NCHR; ch++) sT[ch] = 0u;
                    // When the policy is proven true for the row (or absent) and EVERY candidate is accepted whatever the order, none of the machinery below
                    // is needed -- latency_kernel.hpp's first decision path, derived there (round 5; first in the 4-wavefront workgroups of the small
                    // launches: one query at L = 100 0.298 -> 0.281 ms; then everywhere: interleaved A/B on one box, profiles/r05/ab/ab_accept_all_path.log --
                    // value 8.79-8.96 -> 9.12-9.22 M QPS, resident 7.61 -> 8.00 M, the float-row kernel 2.10 -> 2.07 ms; registers unchanged): with c candidates and d = rn + c - cap list
                    // entries to drop, all candidates below list[rn - d] mean c accepts and d evictions, and one loop over the candidates gives the merge
                    // ranks (evicted live entries: counted, or -- tied with the new worst distance -- kept in the side list, as in the general path).
                    bool fast_done = false;
                    if constexpr (KIND != DIST_ADC_SQ) {
                        // (With the policy counting -- live data -- the same argument runs on t_i = max(e_i, x_i): a candidate is a lane whose threshold x_i
                        //  is below the worst distance; if every candidate's t_i is below list[rn - d], each has #(S_i <= t_i) <= (rn - d) + (c - 1) < cap:
                        //  accepted, and scored because x_i <= t_i; the new lanes that are no candidates fail A4 at every W_i <= W_0. So all c are scored
                        //  and accepted: nexact += c.)
                        if (cm != 0ull) {      // (any number of candidates: a filling list accepts its whole row)
                            const int c = __popcll(cm);
                            const bool iscand = ((cm >> lane) & 1ull) != 0ull;
                            u64 am = cm;            // the accepted candidates
                            u32 nsc = (u32)c;       // the scored ones (where the policy counts)
                            constexpr bool closed = true;
                            {
                                const int d0 = max(0, rn + c - cap);
                                // the candidates at or above list[rn - d0] (tbits == ebits where the policy does not count): none -> all accepted
                                u64 um = 0ull;
                                if (d0 > 0) {
                                    const u32 kd = (u32)(list_get<NCHR>(rk, rn - d0) >> 32);
                                    um = __ballot(iscand && tbits >= kd);
                                }
                                if (um != 0ull) {
                                    // Some candidate may be turned away or pushed out again. Where the policy does not count (proven true for the row, or
                                    // absent) the verdicts are still a closed form -- the first round of the fixed point below, which is its answer then:
                                    // a neighbour turned away earlier had e_j >= W_j >= W_i, so it is never among the e_j <= e_i of an accepted i, and
                                    // counting it cannot rescue a rejected one --
                                    //     accepted_i  <=>  #(old entries with distance <= e_i) + #(earlier candidates with e_j <= e_i) < cap:
                                    // one compare + ballot per list chunk and one over the candidate lanes, no masks, no LDS -- and only for the candidates
                                    // at or above list[rn - d0]: one below it has at most cap - c old entries and c - 1 candidates under it. (Round 5, last
                                    // session: at the bench point 12 of a query's 45 rows -- the ones that fill and settle the list, 32 candidates each --
                                    // took the general path at ~11 000 cycles each, a fifth of a lone query's time: profiles/r05/phase_shares_decision_paths.txt.
                                    // Interleaved A/B, profiles/r05/ab/ab_closed_form_verdicts.log: resident 8.02-8.05 -> 8.31-8.35 M QPS, value +2-4 %, one
                                    // query at L = 100 0.254 -> 0.242 ms of kernel; same bits.)
                                    // Where the policy counts the same walk is exact with t_i = max(e_i, x_i) in place of e_i once it runs in STORED order over
                                    // the candidates at or above list[rn - d0] and counts only neighbours accepted so far (the ones below that entry are
                                    // accepted whatever happens, the ones walked earlier have their verdict): #(S_i <= t_i) itself, no fixed point. Scored
                                    // (the policy's count): the same count at x_i.
                                    {
                                        am = cm & ~um;
                                        nsc = (u32)__popcll(am);
                                        for (u64 mm = um; mm != 0ull; mm &= mm - 1ull) {
                                            const int f = __ffsll((long long)mm) - 1;
                                            const u32 tf = readlane32(tbits, f);
                                            const bool earlier = ((am >> lane) & 1ull) != 0ull && lane < f;
                                            u32 nle = (u32)__popcll(__ballot(earlier && ebits <= tf));
#pragma unroll
                                            for (int ch = 0; ch < NCHR; ch++) nle += (u32)__popcll(__ballot((u32)(rk.v[ch] >> 32) <= tf));      // (unused slots hold ~0)
                                            if (count_pass) {
                                                const u32 xf = readlane32(xbits, f);
                                                u32 nx = (u32)__popcll(__ballot(earlier && ebits <= xf));
#pragma unroll
                                                for (int ch = 0; ch < NCHR; ch++) nx += (u32)__popcll(__ballot((u32)(rk.v[ch] >> 32) <= xf));
                                                nsc += (nx < (u32)cap) ? 1u : 0u;
                                            }
                                            am |= (nle < (u32)cap) ? (1ull << f) : 0ull;
                                        }
                                    }
                                }
                            }
                            if (closed) {
                                fast_done = true;
                                if (count_pass) nexact += nsc;
                                if (am != 0ull) {
                                    const int nacc = __popcll(am);
                                    const int d = max(0, rn + nacc - cap);
                                    const bool isacc = ((am >> lane) & 1ull) != 0ull;
                                    u32 lessc = 0u, rTc = 0u, sTc[NCHR];
#pragma unroll
                                    for (int ch = 0; ch < NCHR; ch++) sTc[ch] = 0u;
                                    if (nacc <= 12) {
                                        for (u64 mm = am; mm != 0ull; mm &= mm - 1ull) {
                                            const int f = __ffsll((long long)mm) - 1;
                                            const u64 kf = readlane64(mykey, f);
                                            lessc += (kf < mykey) ? 1u : 0u;
                                            u32 cnt = 0u;
#pragma unroll
                                            for (int ch = 0; ch < NCHR; ch++) {
                                                sTc[ch] += (kf < rk.v[ch]) ? 1u : 0u;
                                                cnt += (u32)__popcll(__ballot(rk.v[ch] < kf));      // (unused slots hold ~0)
                                            }
                                            rTc = (lane == f) ? cnt : rTc;
                                        }
                                    } else {
                                        // many accepted candidates (the rows that fill the list): their ranks among the OLD keys by the per-lane binary search
                                        // over the list staged in LDS -- one pass for all of them instead of a ballot per candidate and chunk --, and the loop
                                        // keeps only its compares (no result travels to the scalar unit: ~60 instead of ~150 cycles per candidate for a lone wave)
#pragma unroll
                                        for (int ch = 0; ch < NCHR; ch++) {
                                            if (ch * 64 + lane < rn) mk[ch * 64 + lane] = rk.v[ch];
                                            mf[ch * 64 + lane] = 0u;
                                        }
                                        WSYNC();
                                        int lo = 0, hi = rn;
                                        if (isacc) {
                                            constexpr int ITER = (NCHR == 1) ? 7 : (NCHR == 2) ? 8 : (NCHR == 4) ? 9 : (NCHR == 8) ? 10 : 11;
#pragma unroll
                                            for (int it = 0; it < ITER; it++) {
                                                const int m1 = (lo + hi) >> 1;
                                                const u64 v1 = mk[min(m1, rn - 1)];
                                                if (lo < hi) { if (v1 < mykey) lo = m1 + 1; else hi = m1; }
                                            }
                                            // an accepted key with `lo` old keys below it sits below old[lo], old[lo + 1], ...: the old entries' shifts are the
                                            // running sum of this histogram (the state scratch is idle until the merge)
                                            if (lo < rn) atomicAdd(&mf[lo], 1u);
                                        }
                                        rTc = (u32)lo;
                                        for (u64 mm = am; mm != 0ull; mm &= mm - 1ull) {
                                            const int f = __ffsll((long long)mm) - 1;
                                            const u64 kf = readlane64(mykey, f);
                                            lessc += (kf < mykey) ? 1u : 0u;
                                        }
                                        WSYNC();
                                        {
                                            u32 carry = 0u;
#pragma unroll
                                            for (int ch = 0; ch < NCHR; ch++) {
                                                const u32 sc = wave_incl_scan_u32(mf[ch * 64 + lane]) + carry;
                                                sTc[ch] = sc;
                                                if (ch + 1 < NCHR) carry = readlane32(sc, 63);
                                            }
                                        }
                                        WSYNC();      // every search has read the staged list and the histogram before the merge scatters over them
                                    }
                                    {
                                        const u32 o = ninserts + (u32)__popcll(am & lanemask_lt());
                                        if (isacc && o < p.logcap) qlog[o] = ((u64)ebits << 32) | myid;
                                        if (ninserts + (u32)nacc > p.logcap && p.logcap > 0) status |= DR_ST_LOG_OVERFLOW;
                                        ninserts += (u32)nacc;
                                    }
                                    const int rn2 = rn + nacc - d;
                                    int npT[NCHR];
#pragma unroll
                                    for (int ch = 0; ch < NCHR; ch++) {
                                        const int idx = ch * 64 + lane;
                                        npT[ch] = idx + (int)sTc[ch];
                                        if (idx < rn && npT[ch] < cap) { mk[npT[ch]] = rk.v[ch]; mf[npT[ch]] = fl.v[ch]; }
                                    }
                                    const int npA = (int)(rTc + lessc);
                                    if (isacc && npA < cap) { mk[npA] = mykey; mf[npA] = 0u; }
                                    WSYNC();
                                    // the d entries pushed out (old ones, or candidates accepted and pushed out again by later ones): live ones stay in the
                                    // reference's frontier -- worse than every result: only counted; tied with the new worst distance (the cut fell inside a
                                    // run of equal distances): side list, as in the general path below
                                    int nlive_out = 0, nout = 0;
                                    if (d > 0) {
                                        // (with a beam width the entries at the end of the list have long been trimmed from the frontier: usually none of the
                                        //  pushed-out ones is live and one ballot per chunk says so)
                                        const u32 Wfb = (u32)(mk[rn2 - 1] >> 32);
#pragma unroll
                                        for (int ch = 0; ch < NCHR; ch++) {
                                            const bool out = (ch * 64 + lane < rn) && npT[ch] >= cap && fl.v[ch] == 0u;
                                            const u64 lo_ = __ballot(out);
                                            if (lo_ != 0ull) {
                                                const u32 db = (u32)(rk.v[ch] >> 32);
                                                nlive_out += __popcll(lo_);
                                                junk += (u32)__popcll(__ballot(out && db > Wfb));
                                                u64 tm = __ballot(out && db <= Wfb);
                                                while (tm != 0ull) {
                                                    const int f = __ffsll((long long)tm) - 1;
                                                    tm &= tm - 1ull;
                                                    if (tn < 64) { u64 d2; bool dd2; tn = list_insert<1>(tl, tn, 64, fkey(readlane64(rk.v[ch], f)), d2, dd2); }
                                                    else status |= DR_ST_CAND_OVERFLOW;
                                                }
                                            }
                                        }
                                        const bool outc = isacc && npA >= cap;
                                        const u64 om = __ballot(outc);
                                        if (om != 0ull) {
                                            nout = __popcll(om);
                                            junk += (u32)__popcll(__ballot(outc && ebits > Wfb));
                                            u64 tm = __ballot(outc && ebits <= Wfb);
                                            while (tm != 0ull) {
                                                const int f = __ffsll((long long)tm) - 1;
                                                tm &= tm - 1ull;
                                                if (tn < 64) { u64 d2; bool dd2; tn = list_insert<1>(tl, tn, 64, fkey(readlane64(mykey, f)), d2, dd2); }
                                                else status |= DR_ST_CAND_OVERFLOW;
                                            }
                                        }
                                    }
#pragma unroll
                                    for (int ch = 0; ch < NCHR; ch++) {
                                        const int idx = ch * 64 + lane;
                                        rk.v[ch] = (idx < rn2) ? mk[idx] : ~0ull;
                                        fl.v[ch] = (idx < rn2) ? mf[idx] : 0u;
                                    }
                                    cnT += nacc - nout - nlive_out;
                                    rn = rn2;
                                    WSYNC();
                                }
                                if (am == cm) PHD(7); else PHD(6);
                            }
                        }
                    }
                    if (cm == 0ull) PHD(4);
                    if (cm != 0ull && !fast_done) {
                        // (1) counts against the old list. Few candidates (the steady state of a full list: only
                        // neighbours that beat the worst entry are candidates): one wave-wide compare per list chunk
                        // and candidate inside the mask loop below -- the list is sorted in REGISTERS, a ballot counts
                        // it, nothing waits for LDS. Many candidates (a filling list, up to 64): the per-lane binary
                        // search over the list staged in LDS, 7-11 dependent reads but one pass for all lanes. Same
                        // counts either way. (One-wavefront workgroups only: the 16-wave variants have no registers to spare.)
                        constexpr bool BALLOT_COUNTS = (NW == 1);
                        // (break-even: a candidate costs 3 compares + counts per list chunk, the search 7-11 dependent LDS reads)
                        const bool by_ballot = BALLOT_COUNTS && __popcll(cm) * (NCHR + 2) <= 48;
                        if (!by_ballot) {
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) if (ch * 64 + lane < rn) mk[ch * 64 + lane] = rk.v[ch];
                            WSYNC();
                        }
                        const bool iscand = ((cm >> lane) & 1ull) != 0ull;
                        int lb_lo = 0, lb_hi = rn, ut_lo = 0, ut_hi = rn, ux_lo = 0, ux_hi = rn;
                        bool isdup = false;
                        const u64 key_ut = ((u64)tbits << 32) | 0xFFFFFFFFull, key_ux = ((u64)xbits << 32) | 0xFFFFFFFFull;
                        if (iscand && !by_ballot) {
                            constexpr int ITER = (NCHR == 1) ? 7 : (NCHR == 2) ? 8 : (NCHR == 4) ? 9 : (NCHR == 8) ? 10 : 11;
#pragma unroll
                            for (int it = 0; it < ITER; it++) {
                                const int m1 = (lb_lo + lb_hi) >> 1, m2 = (ut_lo + ut_hi) >> 1;
                                const u64 v1 = mk[min(m1, rn - 1)], v2 = mk[min(m2, rn - 1)];
                                if (lb_lo < lb_hi) { if (v1 < mykey) lb_lo = m1 + 1; else lb_hi = m1; }
                                if (ut_lo < ut_hi) { if (v2 <= key_ut) ut_lo = m2 + 1; else ut_hi = m2; }
                                if (count_pass) {
                                    const int m3 = (ux_lo + ux_hi) >> 1;
                                    const u64 v3 = mk[min(m3, rn - 1)];
                                    if (ux_lo < ux_hi) { if (v3 <= key_ux) ux_lo = m3 + 1; else ux_hi = m3; }
                                }
                            }
                            // no visited set: is this key in the list already? (lb_lo = list keys below it)
                            if (novis) isdup = lb_lo < rn && mk[min(lb_lo, rn - 1)] == mykey;
                        }
                        // (the candidate masks are built as two 32-bit halves -- candidates in lanes 0..31, then 32..63 -- so
                        // that every update is one select and one OR instead of two of each on a 64-bit register pair)
                        u64 Mt = 0ull, Mx = 0ull, lessm = 0ull, oldm[NCHR];
#pragma unroll
                        for (int ch = 0; ch < NCHR; ch++) oldm[ch] = 0ull;
#pragma unroll
                        for (int half = 0; half < 2; half++) {
                            u32 mt = 0u, mx = 0u, ls = 0u, om[NCHR];
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) om[ch] = 0u;
                            for (u32 mm = (u32)(cm >> (32 * half)); mm != 0u; mm &= mm - 1u) {
                                const int fl = __ffs((int)mm) - 1;
                                const int f = fl + 32 * half;
                                const u32 ef = readlane32(ebits, f);
                                const u64 kf = readlane64(mykey, f);
                                const u32 bit = 1u << fl;
                                mt |= (f < lane && ef <= tbits) ? bit : 0u;
                                if (count_pass) mx |= (f < lane && ef <= xbits) ? bit : 0u;
                                ls |= (kf < mykey) ? bit : 0u;
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) om[ch] |= (kf < rk.v[ch]) ? bit : 0u;
                                if constexpr (BALLOT_COUNTS) {
                                    if (by_ballot) {
                                        // candidate f's counts (unused list slots hold ~0: above every key and every bound)
                                        const u64 kut = ((u64)readlane32(tbits, f) << 32) | 0xFFFFFFFFull;
                                        const u64 kux = ((u64)readlane32(xbits, f) << 32) | 0xFFFFFFFFull;
                                        int c_lb = NCHR * 64, c_ut = 0, c_ux = 0;
#pragma unroll
                                        for (int ch = 0; ch < NCHR; ch++) {
                                            c_lb -= __popcll(__ballot(kf < rk.v[ch]));       // keys are distinct: list entries below kf
                                            c_ut += __popcll(__ballot(rk.v[ch] <= kut));
                                            if (count_pass) c_ux += __popcll(__ballot(rk.v[ch] <= kux));
                                        }
                                        if (novis) {      // (unused list slots hold ~0, never a candidate's key)
                                            u64 eqm = 0ull;
#pragma unroll
                                            for (int ch = 0; ch < NCHR; ch++) eqm |= __ballot(rk.v[ch] == kf);
                                            isdup = (lane == f) ? (eqm != 0ull) : isdup;
                                        }
                                        lb_lo = (lane == f) ? c_lb : lb_lo;
                                        ut_lo = (lane == f) ? c_ut : ut_lo;
                                        if (count_pass) ux_lo = (lane == f) ? c_ux : ux_lo;
                                    }
                                }
                            }
                            Mt |= (u64)mt << (32 * half); Mx |= (u64)mx << (32 * half); lessm |= (u64)ls << (32 * half);
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) oldm[ch] |= (u64)om[ch] << (32 * half);
                        }
                        // a lane can be accepted only if its own exact distance beats the current worst (when full)
                        const bool canacc = iscand && !isdup && (!full0 || tbits < W0b);
                        accmask = __ballot(canacc);
                        for (int round = 0; round < 66; round++) {
                            const u64 nxt = __ballot(canacc && (u32)ut_lo + (u32)__popcll(Mt & accmask) < (u32)cap);
                            if (nxt == accmask) break;
                            accmask = nxt;
                            if (round == 65) status |= DR_ST_INTERNAL;
                        }
                        na = __popcll(accmask);
                        if (count_pass)
                            nexact += (u32)__popcll(__ballot(iscand && (u32)ux_lo + (u32)__popcll(Mx & accmask) < (u32)cap));
                        rT = lb_lo;
                        rA = __popcll(lessm & accmask);
#pragma unroll
                        for (int ch = 0; ch < NCHR; ch++) sT[ch] = (u32)__popcll(oldm[ch] & accmask);
                        WSYNC();    // every search has read the staged list before the merge scatters over it
                    }
                    if (na > 0) {
                        const bool isacc = ((accmask >> lane) & 1ull) != 0ull;
                        // accepted-insert log, in stored order (finalize replays the reference's heap from it)
                        {
                            const u32 o = ninserts + (u32)__popcll(accmask & lanemask_lt());
                            if (isacc && o < p.logcap) qlog[o] = ((u64)ebits << 32) | myid;
                            if (ninserts + (u32)na > p.logcap && p.logcap > 0) status |= DR_ST_LOG_OVERFLOW;
                            ninserts += (u32)na;
                        }
                        const int rn2 = min(rn + na, cap);
                        // scatter to merged positions
                        int npT[NCHR];
#pragma unroll
                        for (int ch = 0; ch < NCHR; ch++) {
                            const int idx = ch * 64 + lane;
                            npT[ch] = idx + (int)sT[ch];
                            if (idx < rn && npT[ch] < cap) { mk[npT[ch]] = rk.v[ch]; mf[npT[ch]] = fl.v[ch]; }
                        }
                        const int npA = rT + rA;
                        if (isacc && npA < cap) { mk[npA] = mykey; mf[npA] = 0u; }
                        WSYNC();
                        const u32 Wfb = (u32)(mk[rn2 - 1] >> 32);
                        // entries pushed out: live ones stay in the reference's frontier (search_engine.py:469-474):
                        // worse than every result -> only counted; tied with the new worst distance -> side list
                        int nlive_out = 0;
#pragma unroll
                        for (int ch = 0; ch < NCHR; ch++) {
                            const bool out = (ch * 64 + lane < rn) && npT[ch] >= cap && fl.v[ch] == 0u;
                            const u32 db = (u32)(rk.v[ch] >> 32);
                            nlive_out += __popcll(__ballot(out));
                            junk += (u32)__popcll(__ballot(out && db > Wfb));
                            u64 tm = __ballot(out && db <= Wfb);
                            while (tm != 0ull) {
                                const int f = __ffsll((long long)tm) - 1;
                                tm &= tm - 1ull;
                                if (tn < 64) { u64 d2; bool dd2; tn = list_insert<1>(tl, tn, 64, fkey(readlane64(rk.v[ch], f)), d2, dd2); }
                                else status |= DR_ST_CAND_OVERFLOW;
                            }
                        }
                        {
                            const bool out = isacc && npA >= cap;
                            const int nout = __popcll(__ballot(out));
                            junk += (u32)__popcll(__ballot(out && ebits > Wfb));
                            u64 tm = __ballot(out && ebits <= Wfb);
                            while (tm != 0ull) {
                                const int f = __ffsll((long long)tm) - 1;
                                tm &= tm - 1ull;
                                if (tn < 64) { u64 d2; bool dd2; tn = list_insert<1>(tl, tn, 64, fkey(readlane64(mykey, f)), d2, dd2); }
                                else status |= DR_ST_CAND_OVERFLOW;
                            }
                            cnT += na - nout - nlive_out;
                        }
                        // gather the merged list back into registers
#pragma unroll
                        for (int ch = 0; ch < NCHR; ch++) {
                            const int idx = ch * 64 + lane;
                            rk.v[ch] = (idx < rn2) ? mk[idx] : ~0ull;
                            fl.v[ch] = (idx < rn2) ? mf[idx] : 0u;
                        }
                        rn = rn2;
                        WSYNC();
                    }
                    if (cm != 0ull && !fast_done) PHD(0);
                }
            }
            PH(6);
            // ---- frontier trim
            if (kmode == 1u || kmode == 2u || kmode == 5u) {
                // candidates = heapq.nsmallest(beam_width, candidates) (search_engine.py:477-479)
                if (p.bw != 0u && (u32)(cnT + tn) + junk > p.bw) {
                    u32 excess = (u32)(cnT + tn) + junk - p.bw;
                    const u32 rj = excess < junk ? excess : junk;   // junk entries are the largest
                    junk -= rj; excess -= rj;
                    // drop the `excess` largest frontier entries. Common case (side list empty, the cut does not
                    // fall inside a run of equal distances): mark the last `excess` live result entries in one
                    // pass; otherwise one at a time (frontier order reverses the list order inside such runs).
                    bool done_trim = false;
                    if (tn == 0 && excess > 0 && (int)excess <= cnT) {
                        const int keep = cnT - (int)excess;      // live entries that stay
                        int base = 0;
                        u32 kd[NCHR]; bool kill[NCHR];
                        u32 d_lastkept = 0xFFFFFFFFu, d_firstkill = 0xFFFFFFFEu;
#pragma unroll
                        for (int ch = 0; ch < NCHR; ch++) {
                            const bool live = (ch * 64 + lane < rn) && fl.v[ch] == 0u;
                            const u64 lm = __ballot(live);
                            const int rank = base + __popcll(lm & lanemask_lt());
                            kill[ch] = live && rank >= keep;
                            kd[ch] = (u32)(rk.v[ch] >> 32);
                            const u64 m1 = __ballot(live && rank == keep - 1), m2 = __ballot(live && rank == keep);
                            if (m1 != 0ull) d_lastkept = readlane32(kd[ch], __ffsll((long long)m1) - 1);
                            if (m2 != 0ull) d_firstkill = readlane32(kd[ch], __ffsll((long long)m2) - 1);
                            base += __popcll(lm);
                        }
                        if (keep == 0 || d_lastkept != d_firstkill) {
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) fl.v[ch] |= kill[ch] ? 2u : 0u;
                            cnT -= (int)excess;
                            done_trim = true;
                        }
                    }
                    if (!done_trim)
                    for (u32 t = 0; t < excess; t++) {
                        // drop the largest frontier entry: last live result entry or side-list tail
                        const int ia = frontier_last<NCHR>(rk, fl, rn);
                        const u64 ka = (ia >= 0) ? fkey(list_get<NCHR>(rk, ia)) : 0ull;
                        const u64 kb = (tn > 0) ? readlane64(tl.v[0], tn - 1) : 0ull;
                        if (ia >= 0 && (tn == 0 || ka > kb)) { flag_or<NCHR>(fl, ia, 2u); cnT--; }
                        else tn--;
                    }
                }
            } else if (kmode == 3u) {
                // while len(beam) > beam_width: heappop(beam)   (vamana_graph.py:592-593, Q9: pops the BEST)
                if ((u32)(cnT + tn) + junk > p.bw) {
                    u32 excess = (u32)(cnT + tn) + junk - p.bw;
                    const u32 rl = excess < (u32)(cnT + tn) ? excess : (u32)(cnT + tn);
                    for (u32 t = 0; t < rl; t++) {
                        const int ia = frontier_first<NCHR>(rk, fl, rn);
                        const u64 ka = (ia >= 0) ? fkey(list_get<NCHR>(rk, ia)) : ~0ull;
                        const u64 kb = (tn > 0) ? readlane64(tl.v[0], 0) : ~0ull;
                        if (ka <= kb) { flag_or<NCHR>(fl, ia, 2u); cnT--; }
                        else { list_pop_front<1>(tl); tn--; }
                    }
                    excess -= rl;
                    junk -= excess;
                }
            }
        }

        // ---- write results
        write_results<NCHR>(p, qi, cap, kmode, has_out, has_ties, rk, rn, steps, nvisited, nexact, npq, status, ninserts, npq_eval, npre_hit);
        PH(7);
        PH_END(qi);
        {
            u32 t = 0;
            if (lane == 0) t = atomicAdd(p.counter, 1u);
            qi = (u32)__builtin_amdgcn_readfirstlane((int)t) - p.ticket_base + nslots;
        }
    }
    if (lane == 0 && !novis) p.vis_epoch[slot_id] = vstamp;
}

template <int D, bool FILTER, int KIND, int NCHR, int NW, bool CBLDS, int RB = 0, bool U8 = false, bool QB = false, int TREG = 0>
// (minimum wavefronts per SIMD: 2 for the one-wavefront workgroups at D <= 256, which are register-limited -- forcing 3 on
// the exact traversals (168 VGPRs, 25 spilled) measured 0...+5 % slower at the c4 shape; the large dimensions hold a
// 32-KiB table or the row pipeline's buffers: 1)
// the register-table variant exists to run two wavefronts per SIMD: 256 registers each; the 4-wavefront workgroups of the
// byte-row landing variants (16, 17) sit four to a CU like the 16 wavefronts of ONE workgroup of 11 / 13: 4 per SIMD, 128 registers)
__global__ __launch_bounds__(64 * NW, ((NW == 1 && D <= 256) || TREG > 0) ? 2 : (NW == 4 && RB > 0 && U8) ? (QB ? DR_AB_MINW17 : 4) : 1) void search_kernel(const SearchParams p)
{
    search_body<D, FILTER, KIND, NCHR, NW, CBLDS, RB, U8, QB, TREG>(p);
}
