// search_kernel.hpp -- the beam-search kernel: one 64-lane wavefront per query, persistent workgroups.
//
// Reference control flow restated (file:line into Jolara-ai/diskrag):
//   M1 search_engine.py:398-506, M2 pydiskann/vamana_graph.py:719-760, M3 :535-605, M4 :607-640.
// All four are the same loop -- pop the closest frontier node, score its unvisited neighbours in stored
// order, insert the accepted ones into a bounded result list and the frontier, optionally trim the frontier --
// and differ in distance function, result-list capacity, stop rule and trim rule (SearchParams).
//
// Per-query state:
//   LDS   PQ distance table T[m][256] f32 (A2), the query, result list and frontier as sorted arrays of
//         64-bit keys (distance bits << 32 | id), a 64-entry staging area for one expansion
//   HBM   exact visited set: open-addressed table of (generation << 32 | id) words per workgroup
//         (generation tags make clearing unnecessary); accepted-insert log per query (tie replay)
//
// Sequential semantics kept exactly: the neighbours of one expansion are scored in parallel (distances do not
// depend on list state) and then DECIDED in stored order by a wave-uniform loop that runs once per accepted
// insert; lanes between two accepted inserts evaluate the rerank policy A4 (search_engine.py:381-397) with
// the worst-distance W in effect at their position.
#pragma once
#include "numerics.hpp"

// A workgroup is ONE wavefront: LDS operations of a wave execute in issue order, so cross-lane hand-offs through
// LDS only need the compiler not to reorder them. DR_HEAVY_SYNC swaps in a full barrier for debugging.
#ifdef DR_HEAVY_SYNC
#define WSYNC() __syncthreads()
#else
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

#define DR_ST_VIS_OVERFLOW 1u
#define DR_ST_CAND_OVERFLOW 2u
#define DR_ST_LOG_OVERFLOW 4u

enum DistKind { DIST_EXACT = 0, DIST_ADC_SQ = 2 };

struct KStats { u32 steps, visited, exact, pq, status, inserts; };

struct SearchParams {
    const float *vecp;       // [N][D] chain-major
    const u32 *adj;          // [N][R]
    const u64 *first;        // [N][ceil(R/64)] bit s: slot s is a real id and its first occurrence in the row
    const u32 *deg;          // build mode (first == nullptr): rows hold deg[i] distinct ids, the rest is DR_PAD
    const u8 *codes;         // [N][m]
    const float *codebook;   // [m][256][sd]
    const float *queries;    // [nq][D] original element order
    const float *queries_p;  // [nq][D] chain-major
    u64 N;
    u32 D, R, m, sd, medoid, nq;
    u32 mode, k, cap, L, bw, policy, flags;
    u32 norm;                // 1: traversal metric is sqrt(squared L2) (M2, M4: np.linalg.norm)
    u32 max_steps;           // M1: min(10L, N); others: 0xFFFFFFFF
    u32 capC;                // frontier ring capacity (power of two)
    u64 *vis;                // [grid][vis_slots]
    u32 vis_slots;           // power of two
    u32 vis_limit;           // max entries before overflow is flagged
    u32 *vis_gen;            // [grid]
    u32 *counter;            // next query ticket
    u64 *res_keys;           // [nq][cap] ascending (dist bits << 32 | ~id)
    u32 *res_n;              // [nq]
    KStats *stats;           // [nq]
    u32 *tie;                // [nq] 1 = finalize must replay the heap
    u64 *log;                // [nq][logcap] accepted inserts in order (dist bits << 32 | id)
    u32 logcap;
};

DEV u32 lane_id() { return threadIdx.x & 63; }
DEV u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }
DEV u32 hash_id(u32 id) { return id * 2654435761u; }

// ---- visited set ------------------------------------------------------------------------------------------
// Lanes hold distinct ids (duplicates inside a row are removed by the `first` mask). Returns true when the id
// was not in the set and has now been added.
DEV bool wave_visit(u64 *tab, u32 mask, u32 gen, u32 id, bool active)
{
    bool isnew = false, done = !active;
    u32 slot = (hash_id(id) >> 7) & mask;
    const u64 mine = ((u64)gen << 32) | id;
    while (__ballot(!done) != 0ull) {
        if (!done) {
            const u64 cur = __hip_atomic_load(&tab[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((u32)(cur >> 32) == gen) {
                if ((u32)cur == id) done = true;
                else slot = (slot + 1) & mask;
            } else {
                const u64 old = atomicCAS(&tab[slot], cur, mine);
                if (old == cur) { done = true; isnew = true; }
            }
        }
    }
    return isnew;
}

// ---- sorted lists in LDS ---------------------------------------------------------------------------------
// Result list: ascending linear array, key = dist bits << 32 | ~id, so the last element is the one heapq would
// pop from the reference's max-heap of (-dist, id): largest distance, smallest id among equals.
// Returns the new length. `evicted` receives the dropped key when the list was full.
template <int NCH> DEV int res_insert(u64 *a, int n, int cap, u64 key, bool &did_evict)
{
    const int lane = lane_id();
    u64 v[NCH];
    int pos = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int i = c * 64 + lane;
        v[c] = (i < n) ? a[i] : ~0ull;
        pos += __popcll(__ballot(i < n && v[c] < key));
    }
    did_evict = (n == cap);
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--) {
        const int i = c * 64 + lane;
        if (i < n && i >= pos && i + 1 < cap) a[i + 1] = v[c];
    }
    if (lane == 0 && pos < cap) a[pos] = key;
    return n < cap ? n + 1 : cap;
}

// Frontier: ascending ring (key = dist bits << 32 | id), logical index i lives at (head + i) & (capC - 1).
// Pop-min advances head; when full the largest entry falls off the end (caller checks it was junk).
template <int NCH> DEV int cand_insert(u64 *a, int head, int n, int capC, u64 key, u64 &dropped, bool &did_drop)
{
    const int lane = lane_id();
    const int msk = capC - 1;
    u64 v[NCH];
    int pos = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const int i = c * 64 + lane;
        v[c] = (i < n) ? a[(head + i) & msk] : ~0ull;
        pos += __popcll(__ballot(i < n && v[c] < key));
    }
    did_drop = false;
    if (n == capC) {
        // full: the largest element (logical n-1) is dropped, unless the new key is itself the largest
        did_drop = true;
        const int src = (head + n - 1) & msk;
        dropped = a[src];
        if (pos >= n) { dropped = key; return n; }
        n = n - 1;
    }
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--) {
        const int i = c * 64 + lane;
        if (i < n && i >= pos) a[(head + i + 1) & msk] = v[c];
    }
    if (lane == 0) a[(head + pos) & msk] = key;
    return n + 1;
}

DEV float key_dist(u64 key) { return __uint_as_float((u32)(key >> 32)); }

// ---- PQ pieces -------------------------------------------------------------------------------------------
// A2: whole table for one query, entries spread over the wave. q in original order (LDS).
DEV void build_lut_wave(float *lut, const float *__restrict__ codebook, const float *q, u32 m, u32 sd)
{
    const u32 total = m * 256;
    for (u32 e = lane_id(); e < total; e += 64) {
        const u32 jq = e >> 8;
        lut[e] = pw_run_lane(codebook + (size_t)e * sd, q + jq * sd, (int)sd);
    }
}

// A3: squared ADC of one code word, strict sequential f32 order over the sub-quantisers (fast_pq.py:325-326).
DEV float adc_lane(const float *lut, const u8 *__restrict__ code, u32 m)
{
    float s = 0.0f;
    if ((m & 15u) == 0) {
        const uint4 *c4 = reinterpret_cast<const uint4 *>(code);
        for (u32 w = 0; w < m / 16; w++) {
            const uint4 cw = c4[w];
            const u32 ws[4] = { cw.x, cw.y, cw.z, cw.w };
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const u32 j = w * 16 + t * 4 + b;
                    s = f_add(s, lut[j * 256 + ((ws[t] >> (8 * b)) & 255u)]);
                }
            }
        }
    } else if ((m & 3u) == 0) {
        const u32 *c1 = reinterpret_cast<const u32 *>(code);
        for (u32 w = 0; w < m / 4; w++) {
            const u32 cw = c1[w];
#pragma unroll
            for (int b = 0; b < 4; b++) s = f_add(s, lut[(w * 4 + b) * 256 + ((cw >> (8 * b)) & 255u)]);
        }
    } else {
        for (u32 j = 0; j < m; j++) s = f_add(s, lut[j * 256 + code[j]]);
    }
    return s;
}

// ---- the kernel -------------------------------------------------------------------------------------------
// D: vector dimension (compile time: the pairwise tree is unrolled). FILTER: M1's ADC + rerank policy.
// KIND: traversal metric. NCHR/NCHC: result/frontier capacity in 64-entry chunks.
template <int D, bool FILTER, int KIND, int NCHR, int NCHC>
__global__ __launch_bounds__(64) void search_kernel(const SearchParams p)
{
    constexpr bool QREG = (D <= 256);
    constexpr bool SPLIT = QREG && split_form_ok<D>();
    constexpr int NP = SPLIT ? (D <= 128 ? 8 : 4) : 1;   // row passes (8 rows each) kept in flight
    constexpr bool NEED_LUT = FILTER || KIND == DIST_ADC_SQ;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int j = lane & 7, oct = lane >> 3;

    // LDS carve-up (all offsets multiples of 16 bytes)
    size_t off = 0;
    float *lut = reinterpret_cast<float *>(smem + off);
    off += NEED_LUT ? (size_t)p.m * 256 * 4 : 0;
    float *qorig = reinterpret_cast<float *>(smem + off);
    off += (size_t)D * 4;
    float *qperm = reinterpret_cast<float *>(smem + off);
    off += QREG ? 0 : (size_t)D * 4;
    u64 *rk = reinterpret_cast<u64 *>(smem + off);
    off += (size_t)NCHR * 64 * 8;
    u64 *ck = reinterpret_cast<u64 *>(smem + off);
    off += (size_t)NCHC * 64 * 8;
    u32 *nb_id = reinterpret_cast<u32 *>(smem + off);
    off += 64 * 4;
    float *nb_e = reinterpret_cast<float *>(smem + off);
    off += 64 * 4;

    u64 *vtab = p.vis + (size_t)blockIdx.x * p.vis_slots;
    const u32 vmask = p.vis_slots - 1;
    u32 gen = p.vis_gen[blockIdx.x];
    const int cap = (int)p.cap;
    const int capC = (int)p.capC;
    const u32 nwords = (p.R + 63) / 64;

    for (;;) {
        u32 qi = 0;
        if (lane == 0) qi = atomicAdd(p.counter, 1u);
        qi = __shfl(qi, 0);
        if (qi >= p.nq) break;
        gen++;

        // ---- per-query setup
        QueryRegs<D> qreg;
        {
            const float *qg = p.queries + (size_t)qi * D;
            const float *qpg = p.queries_p + (size_t)qi * D;
            for (int i = lane; i < D; i += 64) {
                qorig[i] = qg[i];
                if constexpr (!QREG) qperm[i] = qpg[i];
            }
            if constexpr (QREG) load_query_regs<0, D, D>(qpg, j, qreg);
        }
        WSYNC();
        if constexpr (NEED_LUT) {
            build_lut_wave(lut, p.codebook, qorig, p.m, p.sd);
            WSYNC();
        }

        u32 steps = 0, nvisited = 0, nexact = 0, npq = 0, status = 0, ninserts = 0;
        int rn = 0, cn = 0, chead = 0;
        u32 junk = 0;   // frontier entries dropped because they were worse than every result (see cand_insert)
        u64 *qlog = p.log + (size_t)qi * p.logcap;

        // distance of one node in the traversal metric, computed by octet 0, returned in every lane
        auto node_dist = [&](u32 id) -> float {
            float e;
            if constexpr (KIND == DIST_ADC_SQ) {
                e = adc_lane(lut, p.codes + (size_t)id * p.m, p.m);
                npq++;
            } else {
                e = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)id * D, &qreg, qperm, j);
                e = __shfl(e, 0);
                if (p.norm) e = f_sqrt(e);
                nexact++;
            }
            return e;
        };

        // ---- start node (search_engine.py:416-426)
        {
            const u32 start = p.medoid;
            wave_visit(vtab, vmask, gen, start, lane == 0);
            nvisited = 1;
            const float d0 = node_dist(start);
            const u32 db = __float_as_uint(d0);
            if (lane == 0) {
                rk[0] = ((u64)db << 32) | (u32)(~start);
                ck[0] = ((u64)db << 32) | start;
                if (p.logcap > 0) qlog[0] = ((u64)db << 32) | start;
            }
            rn = 1; cn = 1; chead = 0; ninserts = 1;
        }
        WSYNC();

        // ---- main loop
        while ((cn > 0 || junk > 0) && steps < p.max_steps) {
            steps++;
            if (cn == 0) break;   // only junk left: the reference pops it and stops (it is worse than W)
            const u64 ckey = ck[chead & (capC - 1)];
            chead = (chead + 1) & (capC - 1);
            cn--;
            const float cd = key_dist(ckey);
            const u32 cur = (u32)ckey;
            {
                const float W = key_dist(rk[rn - 1]);
                bool stop;
                if (p.mode == 3u) stop = (cd > W) && (rn == cap);
                else if (p.mode == 4u) stop = (cd > W);
                else stop = (rn >= cap) && (cd > W);
                if (stop) break;
            }

            for (u32 cbase = 0; cbase < p.R; cbase += 64) {
                if (nvisited + 64u > p.vis_limit) { status |= DR_ST_VIS_OVERFLOW; break; }
                const u32 slot = cbase + lane;
                u32 nbid = 0xFFFFFFFFu;
                if (slot < p.R) nbid = p.adj[(size_t)cur * p.R + slot];
                bool active;
                if (p.first) {
                    const u64 fm = p.first[(size_t)cur * nwords + (cbase >> 6)];
                    active = ((fm >> lane) & 1ull) != 0ull;
                } else {
                    active = slot < min(p.deg[cur], p.R) && nbid != 0xFFFFFFFFu;
                }
                const bool isnew = wave_visit(vtab, vmask, gen, nbid, active);
                const u64 newmask = __ballot(isnew);
                const int nnew = __popcll(newmask);
                if (nnew == 0) continue;
                if (isnew) nb_id[__popcll(newmask & lanemask_lt())] = nbid;
                nvisited += nnew;
                WSYNC();

                const u32 myid = nb_id[lane < nnew ? lane : 0];
                float pq_d = 0.0f, e = 0.0f;
                if constexpr (NEED_LUT) {
                    float s = 0.0f;
                    if (lane < nnew) s = adc_lane(lut, p.codes + (size_t)myid * p.m, p.m);
                    if constexpr (FILTER) pq_d = f_sqrt(s);   // asymmetric_distance = sqrt (fast_pq.py:330-333)
                    else e = s;
                    npq += nnew;
                }
                if constexpr (KIND != DIST_ADC_SQ) {
                    // exact distances: octet `oct` scores neighbour r*8+oct of pass r
                    if constexpr (SPLIT) {
                        RowRegs<D> rr[NP];
#pragma unroll
                        for (int r = 0; r < NP; r++) {
                            if (r * 8 < nnew) {
                                const int idx = min(r * 8 + oct, nnew - 1);
                                row_load<0, D, D>(p.vecp + (size_t)nb_id[idx] * D, j, rr[r]);
                            }
                        }
#pragma unroll
                        for (int r = 0; r < NP; r++) {
                            if (r * 8 < nnew) {
                                float ev = row_reduce<0, D, D>(rr[r], qreg);
                                if (p.norm) ev = f_sqrt(ev);
                                if (j == 0 && r * 8 + oct < nnew) nb_e[r * 8 + oct] = ev;
                            }
                        }
                        for (int r0 = NP; r0 * 8 < nnew; r0++) {   // only when NP*8 < 64
                            const int idx = min(r0 * 8 + oct, nnew - 1);
                            float ev = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)nb_id[idx] * D, &qreg, qperm, j);
                            if (p.norm) ev = f_sqrt(ev);
                            if (j == 0 && r0 * 8 + oct < nnew) nb_e[r0 * 8 + oct] = ev;
                        }
                    } else {
                        for (int r0 = 0; r0 * 8 < nnew; r0++) {
                            const int idx = min(r0 * 8 + oct, nnew - 1);
                            float ev = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)nb_id[idx] * D, &qreg, qperm, j);
                            if (p.norm) ev = f_sqrt(ev);
                            if (j == 0 && r0 * 8 + oct < nnew) nb_e[r0 * 8 + oct] = ev;
                        }
                    }
                    WSYNC();
                    e = nb_e[lane < nnew ? lane : 0];
                }

                // ---- decisions in stored order
                bool pending = lane < nnew;
                for (;;) {
                    const float W = key_dist(rk[rn - 1]);
                    bool pass = true;
                    if constexpr (FILTER) {
                        // _should_compute_exact_distance (search_engine.py:381-397); f32 products as numpy does
                        pass = (rn < (int)p.L) || (pq_d < f_mul(W, 0.8f)) ||
                               ((pq_d < f_mul(W, 1.2f)) && p.policy == 0u);
                    }
                    const bool acc = pending && pass && (rn < cap || e < W);
                    const u64 am = __ballot(acc);
                    if (am == 0ull) {
                        if constexpr (FILTER) nexact += __popcll(__ballot(pending && pass));
                        else if constexpr (KIND != DIST_ADC_SQ) nexact += __popcll(__ballot(pending));
                        break;
                    }
                    const int f = __ffsll((long long)am) - 1;
                    const bool upto = pending && lane <= f;
                    if constexpr (FILTER) nexact += __popcll(__ballot(upto && pass));
                    else if constexpr (KIND != DIST_ADC_SQ) nexact += __popcll(__ballot(upto));
                    pending = pending && lane > f;
                    const float ef = __shfl(e, f);
                    const u32 idf = __shfl(myid, f);
                    const u32 eb = __float_as_uint(ef);
                    bool ev;
                    rn = res_insert<NCHR>(rk, rn, cap, ((u64)eb << 32) | (u32)(~idf), ev);
                    u64 dropped = 0; bool dd;
                    cn = cand_insert<NCHC>(ck, chead, cn, capC, ((u64)eb << 32) | idf, dropped, dd);
                    if (lane == 0) {
                        if (ninserts < p.logcap) qlog[ninserts] = ((u64)eb << 32) | idf;
                    }
                    if (ninserts >= p.logcap) status |= DR_ST_LOG_OVERFLOW;
                    ninserts++;
                    WSYNC();
                    if (dd) {
                        // the dropped frontier entry must be worse than every result, otherwise it could still
                        // have been expanded by the reference
                        const float Wn = key_dist(rk[rn - 1]);
                        if (rn >= cap && key_dist(dropped) > Wn) junk++;
                        else status |= DR_ST_CAND_OVERFLOW;
                    }
                }
            }

            if (status & DR_ST_VIS_OVERFLOW) break;
            // ---- frontier trim
            if (p.mode == 1u || p.mode == 2u) {
                // candidates = heapq.nsmallest(beam_width, candidates) (search_engine.py:477-479)
                if (p.bw != 0u && (u32)cn + junk > p.bw) {
                    u32 excess = (u32)cn + junk - p.bw;
                    const u32 rj = excess < junk ? excess : junk;   // junk entries are the largest
                    junk -= rj; excess -= rj;
                    cn -= (int)excess;
                }
            } else if (p.mode == 3u) {
                // while len(beam) > beam_width: heappop(beam)   (vamana_graph.py:592-593, Q9)
                if ((u32)cn + junk > p.bw) {
                    u32 excess = (u32)cn + junk - p.bw;
                    const u32 rl = excess < (u32)cn ? excess : (u32)cn;
                    chead = (chead + (int)rl) & (capC - 1);
                    cn -= (int)rl; excess -= rl;
                    junk -= excess;
                }
            }
        }

        // ---- write results
        WSYNC();
        for (int i = lane; i < rn; i += 64) p.res_keys[(size_t)qi * cap + i] = rk[i];
        // tie detection on the sort key of the final stable sort (distance; sqrt(distance) for M3)
        bool t = false;
        {
            const int lim = min((int)p.k, rn);
            for (int i = lane; i < lim; i += 64) {
                if (i + 1 < rn) {
                    float a = key_dist(rk[i]), b = key_dist(rk[i + 1]);
                    if (p.mode == 3u) { a = f_sqrt(a); b = f_sqrt(b); }
                    if (a == b) t = true;
                }
            }
        }
        const bool anyt = __ballot(t) != 0ull;
        if (lane == 0) {
            p.res_n[qi] = (u32)rn;
            p.tie[qi] = anyt ? 1u : 0u;
            KStats st;
            st.steps = steps; st.visited = nvisited; st.exact = nexact; st.pq = npq; st.status = status;
            st.inserts = ninserts;
            p.stats[qi] = st;
        }
        WSYNC();
    }
    if (lane == 0) p.vis_gen[blockIdx.x] = gen;
}
