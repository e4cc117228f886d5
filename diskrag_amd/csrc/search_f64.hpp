// search_f64.hpp -- M1 / M2 for float64 queries: the `diskrag search` CLI path.
//
// diskrag.py:194 hands the engine `np.array(list_of_floats)`, a float64 query, so everything numpy computes from
// it is float64 (quirk Q8): exact distances (search_engine.py:374-379), the table rows of A2 before they are
// stored as float32 (fast_pq.py:307-316), the worst distance W and the 0.8 / 1.2 products of A4 (:390-395), the
// heap keys, and the distances returned. The batched engine (search_kernel.hpp) packs a float32 distance and an
// id into one 64-bit key; float64 distances do not fit, and the CLI asks one query at a time, so this path is a
// separate, plain kernel: one wavefront per query, distances computed by the whole wavefront (octet per stored
// vector, numpy's pairwise order in double), and the reference's two heaps kept literally -- CPython's heapq sift
// rules on (key, id) tuples -- by lane 0 in LDS. Exact by construction, including the heap-array tie order of the
// final stable sort (search_engine.py:483-488); throughput is not the point here (a query takes ~1 ms).
//
// Round 5: the kernel is a template on the query's arithmetic type. The float32 instantiation exists for ONE thing the batched engine
// cannot do: the LITERAL rerank policy A4 -- in the 0.8 - 1.2 band the reference flips a coin, np.random.random() < 0.2, on numpy's global
// MT19937 stream (search_engine.py:393-395, quirk Q2). The batched kernels decide a whole expansion at once and serve the two
// deterministic branches of that coin (band_policy 0 / 1); here lane 0 walks the neighbours in stored order anyway, so band_policy
// 2 | seed0 << 8 draws from the same generator -- init_genrand(seed0 + query index), 53-bit doubles from two 32-bit draws, exactly numpy's
// legacy np.random.seed / np.random.random -- at the same points of the walk, for float32 (dr_search_batch) and float64
// (dr_search_batch_f64) queries. Goldens: the reference run UNPATCHED (tests/golden/gen_golden_coinflip.py).
//
// Round 6: D = 0 is the GENERIC instantiation -- the dimension is a run-time parameter, the stored rows are in their original element order (or,
// on an index of a built dimension, read through the chain-major position table), numpy's pairwise tree is evaluated from D at run time by
// pw_run_rt (numerics.hpp), one LANE per neighbour, and the kernel serves all four reference traversals: M1, M2 and -- D = 0 only -- M3
// (beam_search_with_pq, vamana_graph.py:535-605: k-sized heap, the trim that pops the BEST candidates, quirk Q9; with DR_F_USE_PQ the squared ADC
// is the distance) and M4 (greedy_search, :607-640). It is what an index whose dimension has no compiled kernels is searched with
// (pydiskann's functions take any D); DR_FORCE_GENERIC=1 runs it on a built dimension, where tests/test_gpu_shapes.py holds it bit for bit to
// the compiled trees. Correct, not fast: one wavefront per query, lane-serial sums.
#pragma once
#include "search_kernel.hpp"

// numpy's legacy generator (numpy/random/src/mt19937/mt19937.c): state in LDS, driven by lane 0
DEV void mt_seed(u32 *mt, u32 seed)
{
    mt[0] = seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (u32)i;
    mt[624] = 624u;
}
DEV u32 mt_next32(u32 *mt)
{
    if (mt[624] == 624u) {
        for (int i = 0; i < 624; i++) {
            const u32 y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7FFFFFFFu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
        }
        mt[624] = 0u;
    }
    u32 y = mt[mt[624]++];
    y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
    return y;
}
DEV double mt_random(u32 *mt)
{
    const u32 a = mt_next32(mt) >> 5, b = mt_next32(mt) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

struct F64Params {
    const float *vecp; const u32 *adj; const u64 *first; const u32 *deg; const u8 *codes; const float *codebook;
    const u32 *perm; const void *queries;      // REAL[nq][D]
    u64 N; u32 D, R, m, sd, medoid, nq;
    u32 mode, k, cap, L, bw, policy, max_steps;
    u32 *vis; u32 vis_words; u32 cand_cap;
    u32 *out_ids; void *out_dist; u32 *out_count; KStats *stats;      // out_dist: REAL[nq][k]
    u32 q0;     // index of the launch's first query in the caller's batch (the coin flip is seeded per query)
    u32 flags;  // DR_F_USE_PQ (M3), DR_F_SQDIST (M4): the generic instantiation (D = 0) serves every mode
    u32 chain_major;   // generic instantiation: the stored rows are a built dimension's chain-major rows (read through perm) / 0: original element order
};

// (key, id) tuple order: first differing element decides (ids are unique inside a heap)
template <typename REAL> DEV bool tup_lt(REAL ka, u32 ia, REAL kb, u32 ib) { return ka < kb || (ka == kb && ia < ib); }

// Lib/heapq.py _siftdown: bubble heap[pos] up towards startpos
template <typename REAL> DEV void hq_siftdown(REAL *hk, u32 *hi, int startpos, int pos)
{
    const REAL nk = hk[pos]; const u32 ni = hi[pos];
    while (pos > startpos) {
        const int parent = (pos - 1) >> 1;
        const REAL pk = hk[parent]; const u32 pi = hi[parent];
        if (tup_lt(nk, ni, pk, pi)) { hk[pos] = pk; hi[pos] = pi; pos = parent; continue; }
        break;
    }
    hk[pos] = nk; hi[pos] = ni;
}
// Lib/heapq.py _siftup: move the smaller child up until a leaf, then sift the item down from there
template <typename REAL> DEV void hq_siftup(REAL *hk, u32 *hi, int n, int pos)
{
    const int startpos = pos;
    const REAL nk = hk[pos]; const u32 ni = hi[pos];
    int child = 2 * pos + 1;
    while (child < n) {
        const int right = child + 1;
        if (right < n && !tup_lt(hk[child], hi[child], hk[right], hi[right])) child = right;
        hk[pos] = hk[child]; hi[pos] = hi[child];
        pos = child;
        child = 2 * pos + 1;
    }
    hk[pos] = nk; hi[pos] = ni;
    hq_siftdown(hk, hi, startpos, pos);
}
template <typename REAL> DEV void hq_push(REAL *hk, u32 *hi, int &n, REAL k, u32 id) { hk[n] = k; hi[n] = id; n++; hq_siftdown(hk, hi, 0, n - 1); }
template <typename REAL> DEV void hq_pop(REAL *hk, u32 *hi, int &n, REAL &k, u32 &id)
{
    n--;
    const REAL lk = hk[n]; const u32 li = hi[n];
    if (n > 0) { k = hk[0]; id = hi[0]; hk[0] = lk; hi[0] = li; hq_siftup(hk, hi, n, 0); }
    else { k = lk; id = li; }
}

// the type's own arithmetic: numpy's pairwise sums (A1, A2), products and square roots in REAL
template <typename REAL> struct SeqNum;
template <> struct SeqNum<double> {
    template <int D> static DEV double row(const float *r, const double *q, int j) { return pw_row_stream64<0, D, D>(r, q, j); }
    static DEV double run(const float *c, const double *q, int n) { return pw_run_lane64(c, q, n); }
    static DEV double mul(double a, double b) { return d_mul(a, b); }
    static DEV double sqrt_(double a) { return __builtin_sqrt(a); }
    static DEV double nan_() { return __longlong_as_double(0x7FF8000000000000ll); }
    static DEV double root_of_sq(double a) { return __builtin_sqrt(a); }
};
template <> struct SeqNum<float> {
    template <int D> static DEV float row(const float *r, const float *q, int j) { return pw_row_stream<0, D, D, false>(r, nullptr, q, j); }
    static DEV float run(const float *c, const float *q, int n) { return pw_run_lane(c, q, n); }
    static DEV float mul(float a, float b) { return f_mul(a, b); }
    static DEV float sqrt_(float a) { return f_sqrt(a); }
    static DEV float nan_() { return __uint_as_float(0x7FC00000u); }
    static DEV float root_of_sq(float a) { return (float)__builtin_sqrt((double)a); }      // M3 without PQ returns sqrt(d) taken in double (vamana_graph.py:598)
};

template <int D, typename REAL>
__global__ __launch_bounds__(64) void search_seq_kernel(const F64Params p)
{
    using NUM = SeqNum<REAL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem64[];
    const int lane = (int)lane_id();
    const int oct = lane >> 3, j = lane & 7;
    const u32 qi = blockIdx.x;
    if (qi >= p.nq) return;
    constexpr bool GEN = (D == 0);
    const int Dn = GEN ? (int)p.D : D;
    const bool pq = (p.mode == 1u);
    // (the generic instantiation only) M3 / M4: their own capacity, stop rule, trim and result order; M3 with PQ scores by the squared ADC alone
    const bool m3 = GEN && p.mode == 3u, m4 = GEN && p.mode == 4u;
    const bool adc_only = m3 && (p.flags & 1u) != 0u;
    const bool rooted = (p.mode == 2u) || (m4 && (p.flags & 2u) == 0u);          // np.linalg.norm: the square root of the pairwise sum
    const bool need_lut = pq || adc_only;
    // LDS carve-up
    REAL *qperm = reinterpret_cast<REAL *>(smem64);                  // chain-major query
    REAL *qorig = qperm + Dn;                                        // original order (table rows)
    REAL *res_k = qorig + Dn;                                        // results heap: keys = -distance
    REAL *cand_k = res_k + (p.cap + 2);                              // candidates heap: keys = distance   (cap + 2: keeps 8-byte alignment for either type)
    REAL *nb_e = cand_k + p.cand_cap;
    u32 *res_i = reinterpret_cast<u32 *>(nb_e + 64);
    u32 *cand_i = res_i + (p.cap + 2);
    u32 *nb_id = cand_i + p.cand_cap;
    float *nb_pq = reinterpret_cast<float *>(nb_id + 64);
    float *lut = nb_pq + 64;                                         // m*256 floats when pq
    u32 *mt = reinterpret_cast<u32 *>(lut + (need_lut ? (size_t)p.m * 256 : 0));   // [625] MT19937 state + position (the literal coin flip)
    const bool coin = (p.policy & 0xFFu) == 2u;
    u32 *vbm = p.vis + (size_t)qi * p.vis_words;                     // zeroed by the host before the launch

    for (int i = lane; i < Dn; i += 64) {
        const REAL v = reinterpret_cast<const REAL *>(p.queries)[(size_t)qi * Dn + i];
        qorig[i] = v;
        qperm[p.perm[i]] = v;
    }
    WSYNC();
    if (need_lut) {
        // A2 in float64, stored as float32 (fast_pq.py:307-316)
        const u32 total = p.m * 256;
        for (u32 e = lane; e < total; e += 64) {
            if constexpr (GEN) lut[e] = (float)pw_run_rt<REAL>(p.codebook + (size_t)e * p.sd, nullptr, qorig + (e >> 8) * p.sd, (int)p.sd);
            else lut[e] = (float)NUM::run(p.codebook + (size_t)e * p.sd, qorig + (e >> 8) * p.sd, (int)p.sd);
        }
        WSYNC();
    }
    // squared ADC of a node, A3's strict order (fast_pq.py:320-328): one lane, one code word
    auto adc_sq = [&](u32 node) -> float {
        const u8 *code = p.codes + (size_t)node * p.m;
        float s = 0.0f;
        for (u32 jj = 0; jj < p.m; jj++) s = f_add(s, lut[jj * 256 + code[jj]]);
        return s;
    };
    // (generic) the position table of a built dimension's chain-major rows, or none: rows in original order
    const u32 *rowpos = nullptr;
    if constexpr (GEN) rowpos = p.chain_major ? p.perm : nullptr;
    const u32 cand_cap = p.cand_cap;
    if (coin && lane == 0) mt_seed(mt, (p.policy >> 8) + p.q0 + qi);     // np.random.seed(seed0 + query index)

    u32 steps = 0, nvisited = 0, nexact = 0, npq = 0, status = 0;
    int rn = 0, cn = 0;
    const int cap = (int)p.cap;

    // start node (search_engine.py:416-426, vamana_graph.py:724-729)
    {
        const u32 start = p.medoid;
        if (lane == 0) atomicOr(&vbm[start >> 5], 1u << (start & 31));
        nvisited = 1;
        REAL d0;
        if constexpr (GEN) {
            if (adc_only) { d0 = (REAL)adc_sq(start); npq = 1; }
            else { d0 = pw_run_rt<REAL>(p.vecp + (size_t)start * Dn, rowpos, qorig, Dn); if (rooted) d0 = NUM::sqrt_(d0); nexact = 1; }
        } else {
            d0 = NUM::template row<(D > 0 ? D : 8)>(p.vecp + (size_t)start * D, qperm, j);
            if (!pq) d0 = NUM::sqrt_(d0);
            nexact = 1;
        }
        if (lane == 0) { hq_push(cand_k, cand_i, cn, d0, start); hq_push(res_k, res_i, rn, -d0, start); }
        cn = 1; rn = 1;
        WSYNC();
    }
    const u32 nwords = (p.R + 63) / 64;
    bool stop = false;
    while (cn > 0 && steps < p.max_steps && !stop) {
        steps++;
        // pop + stop rule (search_engine.py:434-439, vamana_graph.py:731-735) by lane 0, broadcast through LDS
        if (lane == 0) {
            REAL ck; u32 ci; int n = cn;
            hq_pop(cand_k, cand_i, n, ck, ci);
            const REAL W = -res_k[0];
            nb_id[0] = ci;
            nb_id[1] = ((m4 || rn >= cap) && ck > W) ? 1u : 0u;      // (M4 stops on the worst distance whatever the list holds, vamana_graph.py:621-623)
        }
        cn--;
        WSYNC();
        const u32 cur = nb_id[0];
        stop = nb_id[1] != 0u;
        WSYNC();
        if (stop) break;
        for (u32 cbase = 0; cbase < p.R; cbase += 64) {
            const u32 slot = cbase + lane;
            u32 nbid = 0xFFFFFFFFu;
            if (slot < p.R) nbid = p.adj[(size_t)cur * p.R + slot];
            const u64 aux = p.first ? p.first[(size_t)cur * nwords + (cbase >> 6)] : (u64)p.deg[cur];
            bool active;
            if (p.first) active = ((aux >> lane) & 1ull) != 0ull;
            else active = slot < min((u32)aux, p.R) && nbid != 0xFFFFFFFFu;
            bool isnew = false;
            if (active) {
                const u32 bit = 1u << (nbid & 31);
                isnew = (atomicOr(&vbm[nbid >> 5], bit) & bit) == 0u;
            }
            const u64 newmask = __ballot(isnew);
            const int nnew = __popcll(newmask);
            if (nnew == 0) continue;
            if (isnew) nb_id[__popcll(newmask & lanemask_lt())] = nbid;
            nvisited += nnew;
            WSYNC();
            if (pq) {
                // A3: strict sequential float32 sum over the sub-quantisers, then sqrt (fast_pq.py:320-333)
                if (lane < nnew) nb_pq[lane] = f_sqrt(adc_sq(nb_id[lane]));
                npq += nnew;
            }
            if constexpr (GEN) {
                // one lane per new neighbour: its whole sum, the tree walked at run time (M3 with PQ: the squared ADC IS the distance)
                if (lane < nnew) {
                    REAL ev;
                    if (adc_only) ev = (REAL)adc_sq(nb_id[lane]);
                    else { ev = pw_run_rt<REAL>(p.vecp + (size_t)nb_id[lane] * Dn, rowpos, qorig, Dn); if (rooted) ev = NUM::sqrt_(ev); }
                    nb_e[lane] = ev;
                }
                if (adc_only) npq += nnew;
            } else {
                // exact distances of all new neighbours, 8 per pass (the reference scores only those A4 lets through;
                // scoring the others changes nothing but work, the counter below follows the reference)
                for (int r0 = 0; r0 < nnew; r0 += 8) {
                    const int idx = min(r0 + oct, nnew - 1);
                    REAL ev = NUM::template row<(D > 0 ? D : 8)>(p.vecp + (size_t)nb_id[idx] * D, qperm, j);
                    if (!pq) ev = NUM::sqrt_(ev);
                    if (j == 0 && r0 + oct < nnew) nb_e[r0 + oct] = ev;
                }
            }
            WSYNC();
            // the reference's neighbour loop, literally, on lane 0 (search_engine.py:449-474; vamana_graph.py:741-750)
            if (lane == 0) {
                int n_c = cn, n_r = rn;
                u32 ex = 0;
                for (int i = 0; i < nnew; i++) {
                    const REAL W = -res_k[0];
                    if (pq) {
                        const REAL pd = (REAL)nb_pq[i];
                        bool pass;
                        if (n_r < (int)p.L) pass = true;
                        else if (pd < NUM::mul((REAL)0.8, W)) pass = true;
                        else if (pd < NUM::mul((REAL)1.2, W)) pass = coin ? (mt_random(mt) < 0.2) : (p.policy == 0u);     // np.random.random() < 0.2
                        else pass = false;
                        if (!pass) continue;
                    }
                    if (!adc_only) ex++;
                    const REAL e = nb_e[i];
                    if (n_r < cap || e < W) {
                        if (n_c >= (int)cand_cap) { status |= DR_ST_CAND_OVERFLOW; continue; }
                        hq_push(cand_k, cand_i, n_c, e, nb_id[i]);
                        hq_push(res_k, res_i, n_r, -e, nb_id[i]);
                        if (n_r > cap) { REAL dk; u32 di; hq_pop(res_k, res_i, n_r, dk, di); }
                    }
                }
                nb_id[0] = (u32)n_c; nb_id[1] = (u32)n_r; nb_id[2] = ex; nb_id[3] = status;
            }
            WSYNC();
            cn = (int)nb_id[0]; rn = (int)nb_id[1]; nexact += nb_id[2]; status |= nb_id[3];
            WSYNC();
        }
        // candidates = heapq.nsmallest(beam_width, candidates); heapify (search_engine.py:477-479,
        // vamana_graph.py:753-755): the bw smallest tuples in ascending order, which already is a heap
        if (m3) {
            // vamana_graph.py:586-593: while len(candidates) > beam_width: heappop(candidates) -- the BEST candidates leave (quirk Q9)
            if (cn > (int)p.bw) {
                if (lane == 0) { int n = cn; while (n > (int)p.bw) { REAL dk; u32 di; hq_pop(cand_k, cand_i, n, dk, di); } }
                cn = (int)p.bw;
                WSYNC();
            }
        } else if (!m4 && p.bw != 0u && cn > (int)p.bw) {
            if (lane == 0) {
                for (int a = 0; a < (int)p.bw; a++) {
                    int best = a;
                    for (int b = a + 1; b < cn; b++)
                        if (tup_lt(cand_k[b], cand_i[b], cand_k[best], cand_i[best])) best = b;
                    const REAL tk = cand_k[a]; const u32 ti = cand_i[a];
                    cand_k[a] = cand_k[best]; cand_i[a] = cand_i[best];
                    cand_k[best] = tk; cand_i[best] = ti;
                }
            }
            cn = (int)p.bw;
            WSYNC();
        }
    }

    // result extraction by lane 0: M1 stable sort of the heap ARRAY by distance only (search_engine.py:483-488);
    // M2 sorted() on full (distance, id) tuples (vamana_graph.py:758)
    // (generic: M3 / M4 sort like M1 -- stable, by the key alone, vamana_graph.py:596-598 / :640; M3's key and returned distance are sqrt(d):
    //  with PQ the float32 root, whose ties between different sums keep heap-array order; without, the root taken in double -- monotone and
    //  injective on float32 sums, so the order is the sums' own and the root is applied afterwards)
    const bool by_key = pq || m3 || m4;
    if (lane == 0) {
        for (int i = 0; i < rn; i++) res_k[i] = -res_k[i];
        if (adc_only) for (int i = 0; i < rn; i++) res_k[i] = (REAL)f_sqrt((float)res_k[i]);
        for (int i = 1; i < rn; i++) {        // insertion sort: stable
            const REAL kk = res_k[i]; const u32 ii = res_i[i];
            int b = i - 1;
            while (b >= 0 && (by_key ? (res_k[b] > kk) : tup_lt(kk, ii, res_k[b], res_i[b]))) {
                res_k[b + 1] = res_k[b]; res_i[b + 1] = res_i[b]; b--;
            }
            res_k[b + 1] = kk; res_i[b + 1] = ii;
        }
        if (m3 && !adc_only) for (int i = 0; i < rn; i++) res_k[i] = NUM::root_of_sq(res_k[i]);
    }
    WSYNC();
    const int kout = min(rn, (int)p.k);
    for (int i = lane; i < (int)p.k; i += 64) {
        p.out_ids[(size_t)qi * p.k + i] = (i < kout) ? res_i[i] : 0xFFFFFFFFu;
        reinterpret_cast<REAL *>(p.out_dist)[(size_t)qi * p.k + i] = (i < kout) ? res_k[i] : NUM::nan_();
    }
    if (lane == 0) {
        p.out_count[qi] = (u32)kout;
        KStats st;
        st.steps = steps; st.visited = nvisited; st.exact = nexact; st.pq = npq; st.status = status;
        st.inserts = 0; st.pq_evaluated = npq; st.adj_prefetch_hits = 0;
        p.stats[qi] = st;
    }
}
