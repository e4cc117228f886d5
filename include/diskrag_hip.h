/* diskrag_hip.h -- C ABI of libdiskrag_hip.so, the MI355X (gfx950) Vamana beam-search engine.
 *
 * The reference (Jolara-ai/diskrag) has no FFI for search: its seam is Python method calls. Each entry point
 * below names the reference interface it stands behind (file:line into the reference tree). The Python host
 * layer (diskrag_amd/search_engine.py) binds these with ctypes and mirrors SearchEngineCorrect.
 *
 * Conventions: every function returns 0 on success or a negative DR_E_* code; dr_last_error() returns a
 * thread-local message for the last failure on the calling thread. The caller owns every host buffer; the
 * library copies index data to HBM and owns device memory until dr_index_close(). Calls on one handle may
 * come from several threads (the reference shares one engine across request threads, search_engine.py:879):
 * they are serialised internally. There is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef DISKRAG_HIP_H
#define DISKRAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DR_PAD 0xFFFFFFFFu /* empty neighbour slot (in-memory graphs) / empty output slot */

/* search variants */
#define DR_MODE_M1 1u /* SearchEngineCorrect._pq_accelerated_graph_search   search_engine.py:398-506 */
#define DR_MODE_M2 2u /* beam_search_from_disk                              pydiskann/vamana_graph.py:719-760 */
#define DR_MODE_M3 3u /* beam_search_with_pq                                pydiskann/vamana_graph.py:535-605 */
#define DR_MODE_M4 4u /* greedy_search / greedy_search_cython               vamana_graph.py:607-640, cython_utils.pyx:72-122 */
/* Engine mode with NO reference counterpart (intentional divergence): the loop of
 * _pq_accelerated_graph_search (search_engine.py:398-506: L-sized result list, stop rule :438, step cap :429, frontier
 * trim by heapq.nsmallest :477-479) with the squared ADC (fast_pq.py:320-328) as the ONLY distance -- what the
 * reference's PQ-only traversal beam_search_with_pq (DR_MODE_M3 + DR_F_USE_PQ) would be without its k-sized heap and
 * without the trim that pops the BEST candidates (vamana_graph.py:586-593). Serves PQ-only shards (config c5) and,
 * with DR_F_RERANK, config c3's "PQ traversal + full-precision rerank of the L list" (SURVEY.md 8d). Restated in
 * oracle/ as mode 5; the reference-faithful M3 stays as it is. */
#define DR_MODE_PQ 5u
/* The engine's PQ-only traversal restated as a BATCH per step (round 5; no reference counterpart either):
 * the same pieces -- the table of compute_distance_table (fast_pq.py:294-318), the squared ADC summed in strict sub-quantiser
 * order (:320-328), an L-sized list, the step cap min(10 L, N) (search_engine.py:429), heapq.nsmallest(beam_width) on the
 * frontier (:477-479) -- on a TOTAL order key = (distance bits, id): a step expands the `pops` smallest live list entries
 * (DR_F_POPS; default: the rows that fill 64 neighbour slots), scores every first-occurrence neighbour of their rows, and merges those whose key is below the
 * list's largest key (or any, while the list fills) and is not in the list already into the L smallest; newcomers are live;
 * the search stops when no live entry is left. No visited set (membership in the list decides, as with DR_F_NO_VISITED_SET),
 * no neighbour-by-neighbour walk to emulate: one third of DR_MODE_PQ's instruction stream. Results come back in
 * (distance, id) order (squared ADC); with DR_F_RERANK as DR_MODE_PQ's. stats.visited = stats.pq = code words scored,
 * stats.steps = nodes expanded. N < 2^31; pops * next_pow2(R) <= 256. Restated in oracle/diskrag_oracle.c (pqb_search_one) and
 * held to it bit for bit (tests/test_gpu_pqb.py). */
#define DR_MODE_PQB 6u

/* flags */
#define DR_F_USE_PQ 1u /* M3: use_pq=True (ADC-only traversal, vamana_graph.py:318-320) */
#define DR_F_SQDIST 2u /* M4: Cython twin metric, squared L2 (cython_utils.pyx:18-24) instead of the L2 norm */
#define DR_F_RERANK 4u /* DR_MODE_PQ: score the final result list with exact squared L2 (A1, search_engine.py:374-379) and
                          return the k best in (distance, id) order; needs the stored vectors */

#define DR_F_NO_VISITED_SET 16u /* DR_MODE_PQ only (round 4): no visited set. Every first-occurrence neighbour of an expansion is scored; one
                                 * that would enter the list is looked up in the list and dropped if it is there. Same ids, distances and
                                 * order as without the flag (a node scored before and not in the list now can never be accepted again),
                                 * no visited words (N/6 bytes per wavefront slot: 42 GB on a 1.25e8-point shard) and a third of the HBM
                                 * traffic, but nodes that left the list are scored again: stats.visited / stats.pq count EVALUATIONS
                                 * (1.7x with beam_width 8, 3x without trim -- measured slower there, profiles/r04/ab/). Off by default. */
#define DR_F_IP 32u /* with DR_F_RERANK (DR_MODE_PQ / DR_MODE_PQB): the INNER-PRODUCT metric of BASELINE configs c3 / c5 on unit-norm data. The reference
                       has no inner product (vamana_graph.py:294-299: 'l2' and 'cosine' only), so nothing is bit-compared: for unit vectors
                       1 - <q, v> = |q - v|^2 / 2 (SURVEY.md 8d's equivalence), the rerank orders by the exact squared L2 (A1) as always, and
                       out_dist = that distance x 0.5 (exact in float32) = 1 - <q, v>. Refused with DR_E_UNSUPPORTED when a stored vector's
                       squared norm differs from 1 by more than 1e-3 (checked once per index); a QUERY whose squared norm does is answered
                       with NaN distances and dr_stats.status bit 4 (16). */
#define DR_POLICY_COIN(seed0) (2u | (((uint32_t)(seed0) & 0xFFFFFFu) << 8)) /* band_policy: the reference's coin flip itself (see dr_search_batch) */
#define DR_F_POPS_SHIFT 8u
#define DR_F_POPS_MASK 0xF00u
#define DR_F_POPS(n) (((uint32_t)(n) & 15u) << DR_F_POPS_SHIFT) /* DR_MODE_PQB: frontier entries expanded per step (DiskANN's beam): narrow rows
                          (R = 32) fill the 64 lanes two at a time, and a query needs half as many DEPENDENT steps;
                          0 = max(1, 64 / next_pow2(R)): as many rows as fill 64 neighbour slots */
#define DR_F_RERANK_TOP_SHIFT 12u
#define DR_F_RERANK_TOP_MASK 0x3FF000u
#define DR_F_RERANK_TOP(n) (((uint32_t)(n) & 1023u) << DR_F_RERANK_TOP_SHIFT) /* DR_MODE_PQB | DR_F_RERANK (round 6): score only the n <= 1023 list
                          entries with the smallest squared ADC (the list is in (ADC, id) order) instead of all L; 0 = the whole list. The traversal's list
                          length L buys the ADC ranking its depth, the rerank's n what a 4 D-byte row read costs: at D = 1536 the rerank of an L = 250
                          list is a third of the call (config c3). Restated in oracle/ (ORC_F_RERANK_TOP); stats.exact counts the rows scored. */
#define DR_F_COSINE 8u /* M3 without DR_F_USE_PQ: the in-memory graph's distance_metric='cosine' -- compute_query_distance ->
                          cosine_similarity_cython (vamana_graph.py:324-329, cython_utils.pyx:53-70): 1 - cos, 0 when a norm is
                          0; out_dist = sqrt of it (vamana_graph.py:598). The reference sums in float32 under -ffast-math
                          (order unpinned; its own test holds 1e-5): ids equal up to near-ties, distances to 1e-5 */

/* error codes */
#define DR_OK 0
#define DR_E_ARG (-1)        /* bad argument (ValueError in the reference facade) */
#define DR_E_NODEVICE (-2)   /* no HIP device / HIP runtime failure */
#define DR_E_IO (-3)         /* index file missing or of the wrong size (search_engine.py:29-30) */
#define DR_E_NOPQ (-4)       /* mode needs PQ data but dr_index_set_pq was not called */
#define DR_E_OVERFLOW (-5)   /* a per-query work area overflowed (see dr_stats.status) */
#define DR_E_UNSUPPORTED (-6)
#define DR_E_REMOTE (-7)      /* dr_sharded_search: another rank of the exchange failed its local phase (every rank fails the call) */

typedef struct dr_index dr_index;

/* per-query counters: the keys of the stats dict search_engine.py:497-504 returns */
typedef struct {
    uint32_t steps;   /* search_steps */
    uint32_t visited; /* nodes_visited = len(visited) */
    uint32_t exact;   /* exact_distance_computations */
    uint32_t pq;      /* pq_distance_computations */
    uint32_t status;  /* 0 ok; bit0 visited-set overflow (the workgroup-per-query kernel's id set: a blocking call is then served again by the
                         batch kernels and never shows it), bit1 frontier overflow, bit2 insert-log overflow, bit3 internal guard, bit4 (DR_F_IP)
                         the query is not unit-norm */
    uint32_t inserts; /* accepted result-list inserts (engine counter, not in the reference) */
    uint32_t pq_evaluated; /* ADC sums actually computed: `pq` minus those whose outcome (rerank policy True) was
                              proven from a per-query upper bound without reading the code words (engine counter) */
    uint32_t adj_prefetch_hits; /* expansions whose adjacency row was already in LDS: the engine prefetches the row of
                                   the predicted next node (byte-query variants 13/14; 0 elsewhere; engine counter) */
} dr_stats;

/* timing of the last dr_search_batch, or of the dr_batch_run calls since the previous dr_batch_sync /
 * dr_batch_download, measured with HIP events on the engine's own stream. dr_batch_run queues its step and
 * returns; search_kernel_ms is the MEAN search-kernel duration of the steps waited for by that sync (bench.py's
 * roofline uses it), valid after the sync. */
typedef struct {
    float h2d_ms, search_kernel_ms, finalize_kernel_ms, d2h_ms, total_ms;
    uint32_t grid, block, lds_bytes, waves_per_cu;
    uint32_t variant; /* which search_kernel instantiation ran (csrc/variants.hpp) */
    float lut_kernel_ms; /* MEAN duration of the table-build kernel that precedes the search kernel of the per-query-table
                            variants (A2 for the whole batch, fast_pq.py:294-318); 0 for the other variants */
} dr_timing;

int dr_device_count(void);
const char *dr_last_error(void);

/* Opens an index from the reference's on-disk record file (T1, pydiskann/io/diskann_persist.py:17-24 writer,
 * :219-230 MMapNodeReader.get_node): N records of D float32 then R uint32, no header. Replaces
 * MMapNodeReader(index_path, dim, R) + meta["medoid_idx"] in SearchEngineCorrect.__init__
 * (search_engine.py:74-79). */
int dr_index_open(dr_index **out, const char *index_dat, uint64_t N, uint32_t D, uint32_t R, uint32_t medoid,
                  int device);

/* Same, from host arrays: vectors[N][D], adj[N][R] (DR_PAD slots are skipped; disk files pad with 0, which
 * is NOT skipped -- quirk Q3, diskann_persist.py:23). */
int dr_index_create(dr_index **out, const float *vectors, const uint32_t *adj, uint64_t N, uint32_t D,
                    uint32_t R, uint32_t medoid, int device);

/* The vector tier (SURVEY.md 8f N3's optional tier below the resident index; the reference's counterpart is
 * MMapNodeReader serving get_node from index.dat through the OS page cache, pydiskann/io/diskann_persist.py:201-234).
 * DR_TIER_HBM (what dr_index_open / dr_index_create / dr_index_create_empty use): everything lives in HBM.
 * DR_TIER_HOST: adjacency, code words, codebook and visited words stay in HBM; the N*D*4 bytes of full-precision rows
 * live in pinned host memory mapped into the device's address space and are read by the same kernels over PCIe / xGMI --
 * DiskANN's split (compressed vectors + graph in the fast tier, full vectors in the slow one) for an index whose rows
 * do not fit next to the graph. Every mode and entry point works unchanged and returns the same bits; what is meant to
 * run at speed is DR_MODE_PQ with DR_F_RERANK (the traversal never touches a row, the rerank reads L rows per query
 * at the link's rate) and M3 with DR_F_USE_PQ. Byte rows (the D = 128 integer-data copy) are not made for a host-tier
 * index. */
#define DR_TIER_HBM 0u
#define DR_TIER_HOST 1u
int dr_index_open_tiered(dr_index **out, const char *index_dat, uint64_t N, uint32_t D, uint32_t R, uint32_t medoid,
                         int device, uint32_t vector_tier);
int dr_index_create_tiered(dr_index **out, const float *vectors, const uint32_t *adj, uint64_t N, uint32_t D,
                           uint32_t R, uint32_t medoid, int device, uint32_t vector_tier);
int dr_index_create_empty_tiered(dr_index **out, const float *vectors /* NULL: filled by dr_index_write_rows */, uint64_t N,
                                 uint32_t D, uint32_t R, int device, uint32_t vector_tier);
/* Rows [row0, row0 + n) of the stored vectors from host memory: the streaming form of the upload (an index whose rows are
 * generated or read chunk by chunk and would not fit in host memory beside their own copy in the host tier). */
int dr_index_write_rows(dr_index *ix, const float *rows, uint64_t row0, uint64_t n);

/* Attaches PQ data: codebook[m][256][D/m] (kmeans_list[j].cluster_centers_, T3) and codes[N][m]
 * (pq_codes.bin, T2, diskann_persist.py:30-31,205-206). Replaces load_pq_codebook/load_pq_codes in
 * search_engine.py:55-59. */
int dr_index_set_pq(dr_index *ix, const float *codebook, const uint8_t *codes, uint32_t m);

/* Replaces the adjacency (same N, R) -- used for the in-memory graph variants whose neighbour order is the
 * Python set order (vamana_graph.py:581, :629). */
/* PQ-only shard (BASELINE config c5: 1e9 x 1536, the full vectors are never stored): adjacency + PQ codes + codebook,
 * no vectors. Serves dr_search_batch mode DR_MODE_M3 with DR_F_USE_PQ (beam_search_with_pq, vamana_graph.py:535-605,
 * the reference's only PQ-only traversal), dr_adc, dr_distance_table and dr_pq_scan; every entry point that needs a
 * stored vector returns DR_E_UNSUPPORTED. dr_index_drop_vectors turns a full index (built and encoded on the
 * device) into such a shard and frees its N*D*4 bytes. */
int dr_index_create_codes(dr_index **out, const uint32_t *adj, uint64_t N, uint32_t D, uint32_t R, uint32_t medoid,
                          const float *codebook, const uint8_t *codes, uint32_t m, int device);
int dr_index_drop_vectors(dr_index *ix);
/* Disk tier of the full-precision rows -- the counterpart of the reference's MMapNodeReader (io/diskann_persist.py:201-234: node records read from
 * index.dat on demand). A PQ-only index (dr_index_create_codes / dr_index_drop_vectors: graph + code words in HBM) is given the file its rows live
 * in: record i starts at vector_offset + i * record_bytes and begins with D float32 (record_bytes 0 = the reference's record, (D + R) * 4 bytes,
 * diskann_persist.py:17-24). DR_MODE_PQ / DR_MODE_PQB with DR_F_RERANK then traverse on the code words and read the rows of the final lists from the
 * file (O_DIRECT block reads by a pool of host threads where the file system allows it, buffered reads otherwise) for the exact rerank: the same
 * ids and distances as with the rows in HBM, at the file system's pace. DR_E_IO if the file is missing or too short. */
int dr_index_attach_row_file(dr_index *ix, const char *index_dat, uint64_t record_bytes, uint64_t vector_offset);

int dr_index_set_adjacency(dr_index *ix, const uint32_t *adj);

/* ---- graph-sharded search (BASELINE config c5; no reference counterpart) -----------------
 * The id space is cut into disjoint ranges; every range is an index of its own (its own Vamana sub-graph, PQ codes)
 * on one GPU; every query runs on every shard; the per-shard top-k lists (shard-local ids + the shard's base) are
 * merged in canonical (distance ascending, id ascending) order. Across processes the lists travel in ONE RCCL
 * all-gather of (nq*k + 1) * 8 bytes per rank (over xGMI inside a node) -- every entry one packed 64-bit key
 * (order-preserving map of the distance's bits << 32 | id), plus the rank's status word -- issued on the exchange stream
 * on the device-resident merged list, and are merged by a device kernel -- no host staging, no PyTorch.
 * Failure protocol: every rank reaches the collective whatever happened to its own shards; a rank whose local phase failed
 * sends an empty list and its error code in the status word, and the call then fails on EVERY rank (the local DR_E_* on the
 * rank it belongs to, DR_E_REMOTE on the others) instead of leaving the others blocked in the collective. Only a failure to
 * allocate the exchange buffers themselves cannot be told: the communicator is aborted (ncclCommAbort).
 *
 * dr_comm_unique_id: rank 0 creates the 128-byte RCCL id and hands it to the other ranks by whatever side channel the
 * host has (a file, a pipe, MPI); dr_comm_init is collective over the nranks processes (one process per GPU);
 * librccl.so is loaded on first use, the rest of the library does not depend on it.
 * dr_sharded_search: runs (mode, flags) on each of this rank's `nshards` indexes (all on `comm`'s device; without a
 * comm: on the first shard's device, one process), merges them on the device, all-gathers and merges across ranks;
 * every rank receives the full result. id_base[s] is added to shard s's local ids. Output as dr_search_batch
 * (DR_PAD / NaN padded); out_status[nq] (may be NULL) receives the OR of THIS rank's shards' dr_stats.status per query.
 * dr_sharded_submit / dr_sharded_wait: the same call in two halves, four exchanges in flight (per first shard): batch i+1 is uploaded
 * and searched while batch i is exchanged, merged and downloaded; the output buffers -- and a page-locked query buffer,
 * which the copy engine reads in place -- belong to the library until the ticket has been waited for (dr_sharded_wait takes
 * shards[0] of the submit). All ranks must submit -- and wait, when exchanges carry several submits (dr_sharded_set_group) -- in the
 * same order. */
typedef struct dr_comm dr_comm;
#define DR_COMM_ID_BYTES 128
int dr_comm_unique_id(void *out_id /*[DR_COMM_ID_BYTES]*/);
int dr_comm_init(dr_comm **out, const void *unique_id, int nranks, int rank, int device);
int dr_comm_rank(const dr_comm *c, int *out_rank, int *out_nranks);
void dr_comm_destroy(dr_comm *c);
int dr_sharded_search(dr_index *const *shards, const uint32_t *id_base, uint32_t nshards, dr_comm *comm,
                      const float *queries, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width, uint32_t mode,
                      uint32_t band_policy, uint32_t flags, uint32_t *out_ids, float *out_dist, uint32_t *out_status,
                      float *out_ms /*[3] search, all-gather, merge (may be NULL)*/);
int dr_sharded_submit(dr_index *const *shards, const uint32_t *id_base, uint32_t nshards, dr_comm *comm,
                      const float *queries, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width, uint32_t mode,
                      uint32_t band_policy, uint32_t flags, uint32_t *out_ids, float *out_dist, uint32_t *out_status,
                      float *out_ms, uint64_t *out_ticket);
int dr_sharded_wait(dr_index *first_shard, uint64_t ticket);
/* Submits per EXCHANGE (1 ... 16; default 1). With n > 1 the searches of n consecutive dr_sharded_submit calls (same shards, communicator and
 * parameters; <= 65536 queries together since round 6) run as ONE launch per shard and their lists travel in ONE all-gather: a 10 000-query launch of the
 * PQ-only traversal is 4.9 queries per wavefront slot and ends in a tail of idle slots (one 1.25e8-point shard: 1.44 -> 1.76 M QPS at 26.7 k
 * queries per launch). The rule is a COUNT, never a timing, so that every rank forms the same exchanges: an exchange is launched when it holds
 * n submits, when the next submit does not fit or differs, when one of its tickets is waited for, or by dr_sharded_flush; every ticket gets
 * the bits of a dr_sharded_search call of its own. Up to 4 exchanges are in flight per first shard (a further one first finishes the oldest);
 * if an exchange fails, every ticket that rode in it answers the error. */
int dr_sharded_set_group(dr_index *first_shard, uint32_t n);
int dr_sharded_flush(dr_index *first_shard);             /* launches the exchange that is still collecting submits */
/* The merge kernel alone, on host arrays (test seam): ids[S][nq][k] GLOBAL ids (DR_PAD = empty), dist[S][nq][k];
 * empty and NaN entries sort last and come out as DR_PAD / NaN. */
int dr_merge_topk(int device, const uint32_t *ids, const float *dist, uint32_t S, uint32_t nq, uint32_t k,
                  uint32_t k_out, uint32_t *out_ids, float *out_dist);

/* Searches a batch. queries[nq][D] float32 on the host. Outputs (host): out_ids[nq][k] (DR_PAD padded),
 * out_dist[nq][k] (NaN padded; M1: squared L2, M2/M4: L2 norm, M3: sqrt of the traversal metric, Q7),
 * out_count[nq] (results may be shorter than k), stats[nq] (may be NULL).
 *   L           result-list size (M1, M4); ignored by M2 (Q6) and M3
 *   beam_width  frontier trim (M1/M2: heapq.nsmallest, 0 = none; M3: pops the smallest, Q9); M2 list size
 *   band_policy Q2: the reference flips a coin (np.random.random() < 0.2) in the 0.8-1.2 band; 0 = always
 *               rerank, 1 = never (the two deterministic policies most golden vectors are generated with);
 *               DR_POLICY_COIN(seed0) = 2 | seed0 << 8 (round 5; M1, dr_search_batch and dr_search_batch_f64 only): the coin flip ITSELF, drawn
 *               from numpy's legacy MT19937 stream as if np.random.seed(seed0 + i) were called before query i of the call -- bit-exact against
 *               the reference run unpatched (tests/golden/gen_golden_coinflip.py). A sequential walk: served by the literal
 *               one-wavefront-per-query kernel (~1 ms per query), not by the batched engine
 * Stands behind _pq_accelerated_graph_search(q,k,L,beam_width) (search_engine.py:398) and
 * _exact_graph_search(q,k,L) (search_engine.py:508). */
int dr_search_batch(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width,
                    uint32_t mode, uint32_t band_policy, uint32_t flags, uint32_t *out_ids, float *out_dist,
                    uint32_t *out_count, dr_stats *stats);

/* The same seam for FLOAT64 queries: `diskrag search` builds its query with np.array(list_of_floats)
 * (diskrag.py:194), so the reference computes exact distances, table rows (before the float32 store), the worst
 * distance and the 0.8/1.2 products of the rerank policy in float64 and returns float64 distances (quirk Q8).
 * Modes DR_MODE_M1 and DR_MODE_M2 (the two searches the CLI reaches, search_engine.py:566-573); other arguments as
 * dr_search_batch; out_dist is double[nq][k] (NaN padded). Bit-exact against the reference on ids, distances and
 * counters for M1; M2 goes through np.linalg.norm (BLAS order, unpinned) and is held to 1e-4 like its f32 twin. */
int dr_search_batch_f64(dr_index *ix, const double *queries, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width,
                        uint32_t mode, uint32_t band_policy, uint32_t flags, uint32_t *out_ids, double *out_dist,
                        uint32_t *out_count, dr_stats *stats);

/* HBM-resident batches (bench.py: inputs already on the device when the timed region starts).
 * dr_batch_upload copies queries to the device. dr_batch_run launches one search step and returns when its
 * search kernel has finished; the tie-order pass of that step (finalize: replays the reference's heap for the
 * queries whose first k results hold equal distances) runs on a second stream and overlaps the NEXT step's
 * search kernel (outputs are double-buffered). dr_batch_sync waits for everything outstanding;
 * dr_batch_download syncs, then copies the last step's results back. */
int dr_batch_upload(dr_index *ix, const float *queries, uint32_t nq);
int dr_batch_run(dr_index *ix, uint32_t k, uint32_t L, uint32_t beam_width, uint32_t mode, uint32_t band_policy,
                 uint32_t flags);
int dr_batch_sync(dr_index *ix);
int dr_batch_download(dr_index *ix, uint32_t *out_ids, float *out_dist, uint32_t *out_count, dr_stats *stats);
int dr_get_timing(dr_index *ix, dr_timing *out);

/* Several batches can be resident at once (bench.py rotates distinct 10k-query batches): dr_batch_select picks the
 * batch that dr_batch_upload / dr_batch_run (and the kernel-level entry points below, which use it as scratch) refer
 * to. Slot 0 is selected when the handle is created. */
#define DR_MAX_RESIDENT 16u
int dr_batch_select(dr_index *ix, uint32_t slot);

/* Asynchronous, pipelined form of dr_search_batch (same arguments and results): dr_search_submit queues the upload
 * of the batch, its search, the tie-order pass and the download on separate HIP streams and returns a ticket;
 * dr_search_wait blocks until that batch's results are in the caller's output buffers. Up to DR_MAX_TICKETS tickets and
 * 4 LAUNCHES are in flight per handle (a further submit first finishes the oldest), so the copies of batch i+1 / i-1 overlap
 * the search kernel of batch i and the throughput of a stream of host-resident batches approaches the HBM-resident rate.
 * nq <= 32768 per submit. The query buffer may be reused as soon as dr_search_submit returns when it is pageable memory
 * (it is staged); memory from dr_host_alloc (pinned: the copy engine reads it directly, no staging copy) must stay
 * untouched until the ticket has been waited for. The output buffers belong to the library until then.
 *
 * Small submits are COALESCED (round 4; one query per request is the shape of the reference's API routes, app.py:84-130, and
 * nq/8 = 1250 queries is the per-GPU slice of SURVEY 8e's strong-scaling job): submits with equal (k, L, beam_width, mode,
 * band_policy, flags) that arrive while the search stream is busy are held and ride in ONE launch -- one ticket space over
 * their concatenated queries, one tie-order pass, one download; every ticket still gets exactly the bits a dr_search_batch
 * call of its own would return (queries are independent). A submit that finds fewer than two searches queued is launched at
 * once, so a lone request never waits; held submits are launched by the next submit that finds the stream running dry, by
 * any dr_search_wait (which keeps feeding the stream while it waits), by dr_search_flush, or when the group reaches
 * dr_set_coalesce's size or DR_MAX_TICKETS / 2 tickets (default 32768 queries; 0 = every submit is its own launch, the behaviour until round 3). FULL batches
 * are coalesced too: a 10 000-query launch is 2.4 queries per wavefront slot and ends in a tail of idle slots that the next batch's
 * kernel (same stream) cannot fill; launches of 20-25 k queries take 1.05 ms per 10 000 queries instead of 1.24. Round 6: a BULK group -- one that
 * already holds >= 8192 queries -- keeps collecting up to 30 000 while ONE search is still running (it is launched at once when none is): 29.9 k
 * instead of 19.8 k queries per launch on a stream of 10 000-query submits, 0.94 instead of 0.99 ms per 10 000 queries, +5.9 % host -> host.
 * If a launch fails, dr_search_wait of every ticket that rode in it answers the error. */
#define DR_MAX_TICKETS 128u
int dr_search_submit(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width,
                     uint32_t mode, uint32_t band_policy, uint32_t flags, uint32_t *out_ids, float *out_dist,
                     uint32_t *out_count, dr_stats *stats, uint64_t *out_ticket);
int dr_search_wait(dr_index *ix, uint64_t ticket);
int dr_search_flush(dr_index *ix);                       /* launches whatever dr_search_submit is holding back */
int dr_set_coalesce(dr_index *ix, uint32_t max_queries); /* queries per coalesced launch (<= 65536 since round 6; default 32768); 0: no coalescing */
int dr_pipeline_stats(dr_index *ix, uint64_t *out4);     /* [0] launches of the pipelined path, [1] tickets they carried, [2] most tickets in one launch, [3] queries */
int dr_debug_hold(dr_index *ix, int on);                 /* test hook: held submits launch only when full / flushed / waited for */
void *dr_host_alloc(uint64_t bytes); /* page-locked host memory (NULL on failure) */
void dr_host_free(void *p);

/* Kernel-level entry points (B5 seams: reader.get_node + np.sum, pq_model.compute_distance_table,
 * pq_model.asymmetric_distance; diskann_persist.py:219, fast_pq.py:294-333). They run the same device
 * functions as the search kernel. */
int dr_exact_distances(dr_index *ix, const float *queries, uint32_t nq, const uint32_t *node_ids, uint32_t n,
                       float *out /*[nq][n] squared L2*/);
int dr_distance_table(dr_index *ix, const float *queries, uint32_t nq, float *out /*[nq][m][256]*/);
int dr_adc(dr_index *ix, const float *queries, uint32_t nq, const uint32_t *node_ids, uint32_t n,
           float *out_sq /*[nq][n]*/, float *out_sqrt /*[nq][n]*/);
/* Flat PQ scan of all N codes against each query's table (bandwidth ceiling of the LUT-accumulate loop). */
int dr_pq_scan(dr_index *ix, const float *queries, uint32_t nq, float *out_sq /*[nq][N]*/, float *kernel_ms);
/* The same scan, also returning the nearest code word per query (smallest id among equal sums): the isolated
 * ADC kernel of the path, benched against the HBM roofline on a code table far larger than the caches
 * (scripts/bench_pq_scan.py). out_sq may be null. One query (and DR_PQ_SCAN_PER_QUERY=1), n_subvectors in {16, 32, 48, 64}:
 * the scan reads a second copy of the code words in its own order (N * m bytes of device memory, built by the first such call
 * and rebuilt after the code words change; without room for it, or with DR_PQ_SCAN_NO_SKEW=1, the scan reads the code words
 * themselves at 0.6 of the speed). Same results either way. */
int dr_pq_scan_best(dr_index *ix, const float *queries, uint32_t nq, float *out_sq, uint32_t *out_best_id,
                    float *out_best_sq, float *kernel_ms);
/* Brute-force ADC search: the k (<= 64) nearest code words per query by a flat scan of all N code words, in (distance, id)
 * order -- sums as asymmetric_distance_sq (pq/fast_pq.py:320-328), tables as compute_distance_table (:294-318). The ground
 * truth of the PQ-only traversals on shards whose vectors were never stored. kernel_ms (may be NULL):
 * summed duration of the scan launches. n_subvectors in {16, 32, 48, 64}. One query (the reference's request shape) runs on the skewed scan over the
 * scan-order copy of the code words (see dr_pq_scan_best): 64M code words of 32 bytes in 0.33 ms (2.9 ms until round 6); two or more share passes. */
int dr_pq_scan_topk(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t *out_ids /*[nq][k]*/,
                    float *out_sq /*[nq][k] or NULL*/, float *kernel_ms);
/* Brute-force exact top-k (recall ground truth), squared L2 in the A1 summation order. */
int dr_bruteforce_topk(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t *out_ids,
                       float *out_dist);

/* C8: the reference's scalar distance kernels on n row pairs x[i], y[i] of dimension D: squared L2
 * (l2_distance_fast_cython, pydiskann/cython_utils.pyx:18-24) and cosine distance 1 - cos (cosine_similarity_cython,
 * :53-70; 0.0 when either norm is 0). The reference compiles them -ffast-math (summation order unpinned) and tests them
 * at rtol 1e-5 (test_pydiskann_cython.sh:50-54): the same tolerance holds here. Either output may be NULL. The metric
 * 'cosine' exists only in the reference's in-memory M3 (vamana_graph.py:294-299); the disk paths are L2-only. */
int dr_scalar_kernels(int device, const float *x, const float *y, uint32_t n, uint32_t D, float *out_l2, float *out_cos);

/* Reads node i back from HBM in the reference's (vector, neighbours) form -- MMapNodeReader.get_node. */
int dr_get_node(dr_index *ix, uint64_t node_id, float *out_vec /*[D]*/, uint32_t *out_nbrs /*[R]*/);

/* ---- index construction on the device (offline in the reference) -------------------------
 * dr_index_create_empty uploads vectors only (all neighbour slots DR_PAD, medoid 0).
 * dr_build_vamana builds the graph in place: batched form of build_vamana_index_cython
 * (pydiskann/cython_utils.pyx:269-369: `passes` passes, alpha = 1 in the first, then `alpha`; greedy search with
 * list size L_build; robust prune to R; reverse edges with re-prune), medoid = stored vector nearest to the
 * centroid. pad_with_zero != 0 pads short rows with 0 exactly as DiskANNPersist.save_index does
 * (diskann_persist.py:23, quirk Q3), otherwise with DR_PAD. dr_get_adjacency reads the rows back (to write
 * index.dat with the reference's record layout). */
int dr_index_create_empty(dr_index **out, const float *vectors, uint64_t N, uint32_t D, uint32_t R, int device);
int dr_build_vamana(dr_index *ix, uint32_t L_build, float alpha, uint32_t passes, uint64_t seed,
                    uint32_t pad_with_zero, uint32_t max_batch, uint32_t *out_medoid, float *out_seconds);
int dr_get_adjacency(dr_index *ix, uint32_t *out /*[N][R]*/);

/* PQ-only shards built on the device (BASELINE config c5: 1e9 x 1536 -- the vectors are never stored).
 * dr_index_create_codes_empty allocates adjacency + code words for N points and attaches the codebook;
 * dr_pq_encode_rows encodes a streamed chunk of vectors (host, row-major) into rows [row0, row0 + rows) of the code table
 * and forgets the vectors (DiskANNPQ.encode, pq/fast_pq.py:245-267); dr_build_vamana_pq then builds the Vamana graph from
 * the code words alone: the batched builder of dr_build_vamana with every distance replaced by the symmetric PQ
 * distance d(a, b) = sum_j |C_j[code_a[j]] - C_j[code_b[j]]|^2 (the reference has it as pq_distance_fast_cython,
 * pydiskann/cython_utils.pyx:26-51; entries in the table's summation order, the sum in the ADC's), read from a
 * centroid-pair table [m][256][256]. The reference never builds from codes (vamana_graph.py:405 "always exact distances
 * at build time"): this is the engine's own construction, held to graph quality (recall of DR_MODE_PQ searches against
 * the brute-force ADC ranking), not to parity. Rows are DR_PAD padded. R <= 128 and L_build + R + 64 <= 320 (the prune's
 * candidate list: the construction list plus a row with its slack slots); DR_E_ARG otherwise. */
int dr_index_create_codes_empty(dr_index **out, uint64_t N, uint32_t D, uint32_t R, const float *codebook /*[m][256][D/m]*/,
                                uint32_t m, int device);
int dr_pq_encode_rows(dr_index *ix, const float *vectors /*[rows][D]*/, uint64_t row0, uint64_t rows);
int dr_build_vamana_pq(dr_index *ix, uint32_t L_build, float alpha, uint32_t passes, uint64_t seed, uint32_t max_batch,
                       uint32_t *out_medoid, float *out_seconds);
/* Copies the code table and codebook of `src` into `dst` (same N, D and device; device-to-device): a second shard handle
 * with another degree R over points whose vectors were streamed and forgotten. */
int dr_index_copy_codes(dr_index *dst, dr_index *src);

/* Test seam for the builder: the robust prune of ONE point over an explicit candidate list (n <= 448), run by the kernel
 * the builder launches; out_selected[R] receives the picked ids in pick order (DR_PAD padded). It is the textbook form
 * of robust_prune_fast_cython (pydiskann/cython_utils.pyx:435-492): candidates sorted by (distance, id), greedy picks,
 * every later candidate c with alpha * d(p*, c) <= d(p, c) dropped. The reference's loop additionally reads vector
 * slots its own erase() calls left behind (oracle/pybuild.py documents and reproduces that); the device builder does
 * not -- an intentional divergence. */
int dr_debug_prune(dr_index *ix, uint32_t point, const uint32_t *candidates, uint32_t n, float alpha, uint32_t R,
                   uint32_t *out_selected /*[R]*/, uint32_t *out_count);
/* The same seam for the PQ-only builder (dr_build_vamana_pq): the prune of ONE point over an explicit candidate list (n <= 320)
 * scored on code words alone -- d(a, b) = sum_j |C_j[code_a[j]] - C_j[code_b[j]]|^2, every term in A2's summation order, the sum
 * over j in A3's -- run by the kernel the builder launches (rows in registers for m = 16 / 32, in LDS otherwise). No reference
 * counterpart (the reference never builds from codes, vamana_graph.py:405); held to its numpy restatement in the tests. */
int dr_debug_prune_pq(dr_index *ix, uint32_t point, const uint32_t *candidates, uint32_t n, float alpha, uint32_t R,
                      uint32_t *out_selected, uint32_t *out_count);

/* PQ build on the device. dr_pq_train_ex: m independent k-means with 256 centroids on a sample of the
 * stored vectors, as DiskANNPQ.fit runs sklearn's KMeans (pq/fast_pq.py:188-243): greedy k-means++ seeding, n_init
 * restarts keeping the lowest inertia per sub-quantiser, Lloyd iterations until the total squared centre shift is
 * <= tol * mean feature variance or max_iter is reached -- seeding, assignment and centre update all on the device
 * (deterministic for a given seed: fixed reduction trees, fixed-point centre sums).
 * out_inertia (may be NULL) receives the summed quantisation error on the sample. Codebooks are not bit-comparable with
 * sklearn's (different random streams): the golden fixtures ship reference codebooks, and the trainer is held to the
 * reference's quantisation error (tests/test_gpu_round2.py). dr_pq_train = one restart, `iters` iterations, tol 1e-4.
 * dr_pq_encode: nearest-centroid codes for all N vectors (DiskANNPQ.encode, fast_pq.py:245-267), attached to
 * the index like dr_index_set_pq; out_codes may be NULL. */
int dr_pq_train(dr_index *ix, uint32_t m, uint32_t n_sample, uint32_t iters, uint64_t seed,
                float *out_codebook /*[m][256][D/m]*/);
int dr_pq_train_ex(dr_index *ix, uint32_t m, uint32_t n_sample, uint32_t max_iter, uint32_t n_init, float tol, uint64_t seed,
                   float *out_codebook /*[m][256][D/m]*/, double *out_inertia);
int dr_pq_encode(dr_index *ix, const float *codebook, uint32_t m, uint8_t *out_codes /*[N][m] or NULL*/);

/* Inline neighbour codes (off by default; enable != 0 turns them on): the PQ code words of every node's neighbours are
 * kept beside its adjacency row ([N][R][m] bytes of HBM, built on the device before the next search that evaluates ADC
 * sums and rebuilt whenever codes or adjacency change), so that an expansion of the rerank-policy-live M1
 * (search_engine.py:447-456: one ADC per new neighbour) or of a PQ-only traversal reads them as one coalesced block
 * instead of R scattered gathers behind the visited test -- the layout DiskANN uses for its on-disk nodes, here for the
 * HBM request rate. Results never depend on it. Needs n_subvectors % 4 == 0. */
int dr_index_inline_codes(dr_index *ix, int enable);

/* Diagnostic builds only (-DDR_PHASE_TIMING): shader-clock sums per phase of the last search, summed over queries:
 * 0 setup+table, 1 pop/stop, 2 adjacency, 3 visited set, 4 ADC, 5 exact distances, 6 decisions/inserts, 7 output. */
int dr_debug_phase_cycles(dr_index *ix, double *out8);

/* Diagnostic/test hook: pin the search-kernel variant (variants.hpp index) for the modes it serves; -1 restores the
 * engine's own choice. Every variant returns the same bits; the parity tests run all of them. Process-wide.
 * When ix and out_adc_live are non-null, *out_adc_live receives the handle's measured M1 regime: 1 = the rerank
 * policy A4 is live (ADC evaluated: unit-scale data), 0 = provably true almost always (SIFT-scale data, Q1),
 * -1 = not measured yet (the first M1 batch served by an index state is the measurement). */
int dr_debug_force_kind(dr_index *ix, int kind, int *out_adc_live);

void dr_index_close(dr_index *ix);

#ifdef __cplusplus
}
#endif
#endif
