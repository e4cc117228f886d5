/* diskrag_oracle.c -- CPU restatement of the reference's graph-search hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT. Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg may load this library, and only as the checker / reported CPU baseline. The product path
 * (diskrag_amd/, libdiskrag_hip.so) never links, imports or calls it and has no CPU fallback.
 *
 * What is restated (file:line into the reference, Jolara-ai/diskrag):
 *   A1 exact squared L2            search_engine.py:374-379        np.sum(diff*diff): numpy pairwise order
 *   A2 PQ distance table           pydiskann/pq/fast_pq.py:294-318 per-row np.sum, n = sub_dim
 *   A3 ADC                         pydiskann/pq/fast_pq.py:320-333 sequential f32 sum over sub-quantisers, sqrt
 *   A4 rerank policy               search_engine.py:381-397
 *   M1 PQ-accelerated disk search  search_engine.py:398-506
 *   M2 exact disk beam search      pydiskann/vamana_graph.py:719-760
 *   M3 in-memory beam search       pydiskann/vamana_graph.py:535-605 (+ distance dispatch :301-329)
 *   M4 in-memory greedy search     pydiskann/vamana_graph.py:607-640; Cython twin cython_utils.pyx:72-122
 *   heapq                          CPython Lib/heapq.py (tie order of the returned lists depends on it, Q11)
 *   mode 5 (ORC_PQ)                NOT in the reference: M1's loop with the squared ADC as the only distance (the
 *                                  engine's flagged PQ-only traversal, an intentional divergence from quirk Q9); it is
 *                                  the checker for DR_MODE_PQ and is pinned only through the pieces it shares with
 *                                  M1 (loop, heaps) and M3-with-PQ (ADC distance), both golden-pinned
 *
 *   mode 6 (ORC_PQB)               NOT in the reference either: the batch-per-step statement of the PQ-only traversal
 *                                   -- pqb_search_one below; built from the golden-pinned A2 table and
 *                                  A3 sum; the checker for DR_MODE_PQB
 *
 * Parity pinning: every function here is checked in tests/test_oracle_golden.py against golden vectors that
 * tests/golden/gen_golden.py produced by running the reference itself (imported from /root/reference in the
 * dev container). M1 and M3-with-PQ are pinned bit-exactly (ids, float bits, counters). M2/M4 use
 * np.linalg.norm -> BLAS sdot and M3-without-PQ uses a -ffast-math Cython loop; their summation order is
 * library/compiler specific, so those are pinned on ids with a near-tie allowance and on distances to 1e-4
 * relative ("parity unpinned at bit level" for those three distance functions only).
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off; no -ffast-math, no FMA contraction).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_PAD 0xFFFFFFFFu
#define ORC_M1 1u
#define ORC_M2 2u
#define ORC_M3 3u
#define ORC_M4 4u
#define ORC_PQ 5u         /* engine mode DR_MODE_PQ (no reference counterpart): M1's loop on squared ADC distances only */
#define ORC_PQB 6u        /* engine mode DR_MODE_PQB (no reference counterpart): batch-per-step ADC beam search, pqb_search_one below */
#define ORC_F_USE_PQ 1u   /* M3: use_pq=True */
#define ORC_F_CYTHON 2u   /* M4: greedy_search_cython twin (squared L2 via l2_distance_fast_cython) */
#define ORC_F_QUERY_F64 4u
#define ORC_F_RERANK 16u  /* ORC_PQ: exact squared L2 (A1) of the final list, (distance, id) order */
#define ORC_F_NO_VISITED_SET 64u /* ORC_PQ: the statement without a visited set (engine flag DR_F_NO_VISITED_SET): same ids and distances; the counters count evaluations */
#define ORC_F_COSINE 32u  /* M3 without PQ: distance_metric='cosine' (cosine_similarity_cython, cython_utils.pyx:53-70) */
#define ORC_F_IP 128u      /* with ORC_F_RERANK: engine flag DR_F_IP -- unit-norm data, the returned distance is |q - v|^2 / 2 = 1 - <q, v> (no reference counterpart) */
#define ORC_F_RERANK_TOP_SHIFT 12u /* ORC_PQB with ORC_F_RERANK: bits 12..21 = rerank only the n list entries with the smallest ADC keys (0 = all), engine flag DR_F_RERANK_TOP(n) */
#define ORC_F_POPS_SHIFT 8u /* ORC_PQB: bits 8..11 = frontier entries expanded per step (0 = 1), engine flag DR_F_POPS(n) */
#define ORC_F_PAIRWISE 8u  /* squared-L2 modes: use the numpy pairwise order (what the device computes) instead of the
                             sequential Cython loop, whose -ffast-math order is unpinned anyway */

typedef struct {
    uint64_t N;
    uint32_t D, R, m, medoid;
    const float *vectors;    /* [N][D]                         index.dat records, vector part  */
    const uint32_t *adj;     /* [N][R]  ORC_PAD slots skipped  index.dat records, neighbour part (0-padded on disk) */
    const uint8_t *codes;    /* [N][m] or NULL                 pq_codes.bin */
    const float *codebook;   /* [m][256][D/m] or NULL          kmeans_list[j].cluster_centers_ */
} orc_index;

/* numpy's legacy global generator (np.random.seed(int) -> init_genrand; np.random.random() -> 53-bit double from two 32-bit draws:
 * numpy/random/src/mt19937/mt19937.c, mt19937_seed / mt19937_gen / mt19937_next_double): what the reference's coin flip draws from
 * (search_engine.py:393-395, quirk Q2). band policy 2 | seed0 << 8 = "np.random.seed(seed0 + query index) before each query". */
typedef struct { uint32_t mt[624]; int pos; } orc_mt;
static void orc_mt_seed(orc_mt *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->pos = 624;
}
static uint32_t orc_mt_next32(orc_mt *g)
{
    if (g->pos == 624) {
        for (int i = 0; i < 624; i++) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7FFFFFFFu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908B0DFu : 0u);
        }
        g->pos = 0;
    }
    uint32_t y = g->mt[g->pos++];
    y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680u; y ^= (y << 15) & 0xEFC60000u; y ^= y >> 18;
    return y;
}
static double orc_mt_random(orc_mt *g)
{
    const uint32_t a = orc_mt_next32(g) >> 5, b = orc_mt_next32(g) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
/* test seam: the first n doubles of np.random.seed(seed); np.random.random() */
void orc_mt_doubles(uint32_t seed, uint32_t n, double *out)
{
    orc_mt g; orc_mt_seed(&g, seed);
    for (uint32_t i = 0; i < n; i++) out[i] = orc_mt_random(&g);
}

static inline float sqrt_real_f32(float x) { return sqrtf(x); }
static inline double sqrt_real_f64(double x) { return sqrt(x); }

#define REAL float
#define SFX f32
#include "oracle_core.inc"
#undef REAL
#undef SFX

#define REAL double
#define SFX f64
#include "oracle_core.inc"
#undef REAL
#undef SFX

/* ---- mode 6 (ORC_PQB): the engine's batch-per-step PQ-only beam search ------------------------------------------
 * NOT in the reference (its only PQ-only traversal, beam_search_with_pq, vamana_graph.py:535-605, keeps a k-sized heap and
 * trims its frontier from the wrong end, quirk Q9). This is SURVEY.md 8a row E's batch-per-expansion form, stated on a
 * TOTAL order so that nothing depends on evaluation order:
 *   distance      d(i) = squared ADC of node i: T = compute_distance_table(q) (fast_pq.py:294-318), s = 0f; s += T[j][code_j]
 *                 in strict order of j (fast_pq.py:320-328) -- the golden-pinned A2 / A3
 *   key(i)        (bits of d(i), i): distances are sums of squares (>= +0), so float order = bit order; ids break ties
 *   list          at most L entries in ascending key order, each live / not live (expanded or trimmed)
 *   pops          frontier entries expanded per step: flags bits 8..11, 0 = max(1, 64 / next_pow2(R)) (rows that fill 64 slots)
 *   step          take the p = min(pops, #live, max_steps - steps) smallest live entries, mark them expanded;
 *                 score every neighbour slot of their rows that is the first occurrence of its id in ITS row (ORC_PAD
 *                 skipped; the 0-pads of a disk row are neighbour 0 once, quirk Q3); a scored key enters the candidate SET
 *                 iff (list not full or key < the list's largest key) and it is neither in the list nor in the set --
 *                 all tests against the list as it was when the step began; list = the L smallest of list + set, new
 *                 entries live; then, with beam_width > 0, only the beam_width smallest live entries stay live
 *                 (heapq.nsmallest on the frontier, search_engine.py:477-479)
 *   stop          no live entry, or max_steps = min(10 L, N) expanded nodes (search_engine.py:429)
 * There is no visited set: a node that was scored and is not in the list now was rejected or evicted at a largest key >=
 * today's and can never enter again, so membership in the list is the whole test (nodes met again are scored again; the
 * counters count evaluations). Output: the k smallest keys (squared ADC, id); with ORC_F_RERANK the whole list is scored
 * with the exact squared L2 (A1, search_engine.py:374-379) and the k best in (distance, id) order are returned.
 * stats = { expanded nodes, evaluations, exact (rerank), evaluations }. */
typedef struct { uint32_t db, id; int live; } pqb_ent;
static int pqb_less(uint32_t d1, uint32_t i1, uint32_t d2, uint32_t i2) { return d1 != d2 ? d1 < d2 : i1 < i2; }
static int pqb_cmp(const void *x, const void *y)
{
    const pqb_ent *a = (const pqb_ent *)x, *b = (const pqb_ent *)y;
    if (pqb_less(a->db, a->id, b->db, b->id)) return -1;
    if (pqb_less(b->db, b->id, a->db, a->id)) return 1;
    return 0;
}
static int pqb_search_one(const orc_index *ix, const float *q, uint32_t k, uint32_t L, uint32_t bw, uint32_t flags,
                          uint32_t *out_ids, double *out_dist, uint32_t *out_count, uint32_t *stats)
{
    const uint64_t N = ix->N;
    const uint32_t D = ix->D, R = ix->R, m = ix->m;
    if (!ix->codes || !ix->codebook || m == 0 || D % m) return -2;
    if (k == 0 || L == 0 || ix->medoid >= N) return -3;
    uint32_t pops = (flags >> ORC_F_POPS_SHIFT) & 15u;
    if (pops == 0) { uint32_t rs = 1; while (rs < R) rs <<= 1; pops = rs >= 64 ? 1 : 64 / rs; }   /* default: the rows that fill 64 neighbour slots */
    float *lut = (float *)malloc((size_t)m * 256 * sizeof(float));
    build_lut_f32(ix->codebook, q, m, D / m, lut);
    pqb_ent *list = (pqb_ent *)malloc(((size_t)L + (size_t)pops * R + 1) * sizeof(pqb_ent));
    size_t n = 0;
    uint32_t steps = 0, nevals = 0, nexact = 0, ndup = 0;
    const uint64_t max_steps = (uint64_t)L * 10 < N ? (uint64_t)L * 10 : N;
#define PQB_ADC(i, out) do { float s_ = 0.0f; const uint8_t *c_ = ix->codes + (size_t)(i) * m;             \
        for (uint32_t j_ = 0; j_ < m; j_++) s_ += lut[j_ * 256 + c_[j_]];                                  \
        memcpy(&(out), &s_, 4); nevals++; } while (0)
    {
        uint32_t d0;
        PQB_ADC(ix->medoid, d0);
        list[0].db = d0; list[0].id = ix->medoid; list[0].live = 1; n = 1;
    }
    uint32_t *popped = (uint32_t *)malloc(pops * sizeof(uint32_t));
    for (;;) {
        size_t nlive = 0;
        for (size_t i = 0; i < n; i++) nlive += (size_t)list[i].live;
        if (nlive == 0 || steps >= max_steps) break;
        uint32_t p = pops;
        if (p > nlive) p = (uint32_t)nlive;
        if ((uint64_t)p > max_steps - steps) p = (uint32_t)(max_steps - steps);
        for (uint32_t t = 0, i = 0; t < p; i++) if (list[i].live) { list[i].live = 0; popped[t++] = list[i].id; }
        steps += p;
        const size_t n_old = n;
        const int full = (n_old >= L);
        const pqb_ent worst = list[n_old - 1];
        size_t nc = 0;                                    /* the candidate set lives behind the list */
        for (uint32_t t = 0; t < p; t++) {
            const uint32_t *nbrs = ix->adj + (size_t)popped[t] * R;
            for (uint32_t s = 0; s < R; s++) {
                const uint32_t nb = nbrs[s];
                if (nb == ORC_PAD) continue;
                if (nb >= N) { free(lut); free(list); free(popped); return -4; }
                int seen = 0;
                for (uint32_t u = 0; u < s && !seen; u++) seen = (nbrs[u] == nb);
                if (seen) continue;
                uint32_t db;
                PQB_ADC(nb, db);
                if (full && !pqb_less(db, nb, worst.db, worst.id)) continue;
                int dup = 0;
                for (size_t u = 0; u < n_old + nc && !dup; u++) dup = (list[u].id == nb);
                if (dup) { ndup++; continue; }
                list[n_old + nc].db = db; list[n_old + nc].id = nb; list[n_old + nc].live = 1; nc++;
            }
        }
        n = n_old + nc;
        qsort(list, n, sizeof(pqb_ent), pqb_cmp);
        if (n > L) n = L;
        if (bw) { size_t kept = 0; for (size_t i = 0; i < n; i++) if (list[i].live) { if (kept >= bw) list[i].live = 0; else kept++; } }
    }
#undef PQB_ADC
    if (flags & ORC_F_RERANK) {
        if (!ix->vectors) { free(lut); free(list); free(popped); return -2; }
        const uint32_t top = (flags >> ORC_F_RERANK_TOP_SHIFT) & 1023u;       /* the list is in (ADC, id) order: its first `top` entries */
        if (top && n > top) n = top;
        for (size_t i = 0; i < n; i++) {
            const float e = pw_sqdiff_f32(ix->vectors + (size_t)list[i].id * D, q, D);
            memcpy(&list[i].db, &e, 4); nexact++;
        }
        qsort(list, n, sizeof(pqb_ent), pqb_cmp);
    }
    const uint32_t cnt = (uint32_t)(n < k ? n : k);
    for (uint32_t i = 0; i < cnt; i++) {
        float d; memcpy(&d, &list[i].db, 4);
        if ((flags & ORC_F_RERANK) && (flags & ORC_F_IP)) d = d * 0.5f;
        out_ids[i] = list[i].id; out_dist[i] = (double)d;
    }
    for (uint32_t i = cnt; i < k; i++) { out_ids[i] = ORC_PAD; out_dist[i] = NAN; }
    *out_count = cnt;
    if (stats) { stats[0] = steps; stats[1] = nevals; stats[2] = nexact; stats[3] = nevals; }
    /* diagnostic (scripts/exp_pqb_revisits.py): how many scored slots were nodes that sat IN the list at that moment */
    if (stats && getenv("ORC_PQB_DIAG") && !(flags & ORC_F_RERANK)) stats[2] = ndup;
    free(lut); free(list); free(popped);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ C API */

/* numpy-order squared L2 between a stored vector and a query (A1). */
float orc_sqdist_f32(const float *v, const float *q, uint32_t n) { return pw_sqdiff_f32(v, q, n); }
double orc_sqdist_f64(const float *v, const double *q, uint32_t n) { return pw_sqdiff_f64(v, q, n); }

/* A2: lut[m][256] */
void orc_build_lut_f32(const float *codebook, const float *q, uint32_t m, uint32_t sd, float *lut)
{ build_lut_f32(codebook, q, m, sd, lut); }
void orc_build_lut_f64(const float *codebook, const double *q, uint32_t m, uint32_t sd, float *lut)
{ build_lut_f64(codebook, q, m, sd, lut); }

/* A3: squared ADC and its sqrt for n codes */
void orc_adc(const float *lut, const uint8_t *codes, uint64_t n, uint32_t m, float *out_sq, float *out_sqrt)
{
    for (uint64_t i = 0; i < n; i++) {
        float s = 0.0f;
        for (uint32_t j = 0; j < m; j++) s += lut[j * 256 + codes[i * m + j]];
        if (out_sq) out_sq[i] = s;
        if (out_sqrt) out_sqrt[i] = sqrtf(s);
    }
}

/* Search a batch of queries. queries: float[nq][D] (or double when flags & ORC_F_QUERY_F64).
 * out_ids[nq][k] (ORC_PAD padded), out_dist[nq][k] (double; NaN padded), out_count[nq],
 * stats[nq][4] = {search_steps, nodes_visited, exact_distance_computations, pq_distance_computations}.
 * nthreads > 1 runs queries on OpenMP threads (used only by bench.py's cpu_baseline). Returns 0 or <0. */
int orc_search_batch(const float *vectors, const uint32_t *adj, const uint8_t *codes, const float *codebook,
                     uint64_t N, uint32_t D, uint32_t R, uint32_t m, uint32_t medoid,
                     const void *queries, uint32_t nq, uint32_t mode, uint32_t k, uint32_t L, uint32_t bw,
                     uint32_t policy, uint32_t flags, int nthreads,
                     uint32_t *out_ids, double *out_dist, uint32_t *out_count, uint32_t *stats)
{
    orc_index ix = { N, D, R, m, medoid, vectors, adj, codes, codebook };
    int rc_all = 0;
    if (mode < ORC_M1 || mode > ORC_PQB) return -1;
    if (mode == ORC_PQB && (flags & ORC_F_QUERY_F64)) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < (int64_t)nq; i++) {
        int rc;
        uint32_t *st = stats ? stats + (size_t)i * 4 : NULL;
        if (mode == ORC_PQB)
            rc = pqb_search_one(&ix, (const float *)queries + (size_t)i * D, k, L, bw, flags,
                                out_ids + (size_t)i * k, out_dist + (size_t)i * k, out_count + i, st);
        else if (flags & ORC_F_QUERY_F64)
            rc = search_one_f64(&ix, (const double *)queries + (size_t)i * D, (uint32_t)i, mode, k, L, bw, policy, flags,
                                out_ids + (size_t)i * k, out_dist + (size_t)i * k, out_count + i, st);
        else
            rc = search_one_f32(&ix, (const float *)queries + (size_t)i * D, (uint32_t)i, mode, k, L, bw, policy, flags,
                                out_ids + (size_t)i * k, out_dist + (size_t)i * k, out_count + i, st);
        if (rc) {
#ifdef _OPENMP
#pragma omp critical
#endif
            rc_all = rc;
        }
    }
    return rc_all;
}

/* brute-force ground truth (squared L2, numpy order), k smallest ids per query: used for recall */
void orc_bruteforce_topk(const float *vectors, uint64_t N, uint32_t D, const float *queries, uint32_t nq,
                         uint32_t k, int nthreads, uint32_t *out_ids)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t qi = 0; qi < (int64_t)nq; qi++) {
        const float *q = queries + (size_t)qi * D;
        float *bd = (float *)malloc(k * sizeof(float));
        uint32_t *bi = out_ids + (size_t)qi * k;
        uint32_t cnt = 0;
        for (uint64_t i = 0; i < N; i++) {
            float d = pw_sqdiff_f32(vectors + i * D, q, D);
            if (cnt < k || d < bd[cnt - 1]) {
                uint32_t p = cnt < k ? cnt++ : k - 1;
                while (p > 0 && bd[p - 1] > d) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; p--; }
                bd[p] = d; bi[p] = (uint32_t)i;
            }
        }
        for (uint32_t j = cnt; j < k; j++) bi[j] = ORC_PAD;
        free(bd);
    }
}

/* C8 scalar kernels (cython_utils.pyx:18-24, :53-70): one f32 accumulator in index order. */
float orc_l2_seq_f32(const float *x, const float *y, uint32_t n)
{
    float s = 0.0f;
    for (uint32_t i = 0; i < n; i++) s += (x[i] - y[i]) * (x[i] - y[i]);
    return s;
}
float orc_cosine_dist_f32(const float *x, const float *y, uint32_t n)
{
    float dot = 0.0f, nx = 0.0f, ny = 0.0f;
    for (uint32_t i = 0; i < n; i++) { dot += x[i] * y[i]; nx += x[i] * x[i]; ny += y[i] * y[i]; }
    if (nx == 0.0f || ny == 0.0f) return 0.0f;
    return (float)(1.0 - ((double)dot / (sqrt((double)nx) * sqrt((double)ny))));
}
