// engine.hip -- host side of libdiskrag_hip.so: the C ABI declared in include/diskrag_hip.h.
//
// Index data lives in HBM for the life of the handle:
//   vecp   [N][D] f32, chain-major tiles (numerics.hpp)        from index.dat records (T1)
//   adj    [N][R] u32 + first-occurrence masks [N][ceil(R/64)] from index.dat records (T1)
//   codes  [N][m] u8, codebook [m][256][D/m] f32               pq_codes.bin (T2), cluster centres (T3)
// Scratch (per handle, grown on demand): device query buffers, per-workgroup visited tables, per-query result
// lists, insert logs, stats; one HIP stream per handle; calls on a handle are serialised by a mutex.
#include <hip/hip_runtime.h>

#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <map>
#include <atomic>
#include <cmath>
#include <mutex>
#include <thread>
#include <chrono>
#include <string>
#include <vector>

#include <rocprim/rocprim.hpp>      // device radix sort (the builder's reverse-edge pass)

#include "../../include/diskrag_hip.h"
#include "engine_kernels.hpp"
#include "search_f64.hpp"
#include "build_kernels.hpp"
#include "build_pq_kernels.hpp"
#include "variants.hpp"
#include "pqb_host.hpp"

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(DR_E_NODEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const DimKernels *dr_dim_kernels(int D)
{
    switch (D) {
    case 32: return dr_dim_kernels_32();
    case 64: return dr_dim_kernels_64();
    case 96: return dr_dim_kernels_96();
    case 128: return dr_dim_kernels_128();
    case 256: return dr_dim_kernels_256();
    case 768: return dr_dim_kernels_768();
    case 960: return dr_dim_kernels_960();
    case 1536: return dr_dim_kernels_1536();
    default: return nullptr;
    }
}

struct dr_index;
static int quiesce_locked(dr_index *ix);
static const uint32_t DR_MAX_GROUP = 65536;     // queries ONE launch of the pipelined path may hold (dr_set_coalesce; the default is 32768): a launch's scratch is sized by its queries

// A fill that has COMPLETED when the call returns. hipMemset on device memory queues a fill kernel on the null stream and may return before it has
// run; the handle's streams are created hipStreamNonBlocking and do not wait for the null stream -- on a GPU shared with other processes (eight
// bench ranks on one device) a kernel of the handle read or incremented a "zero-filled" counter BEFORE the fill landed: dr_build_vamana answered
// "adjacency holds 100 neighbour ids >= N" about a healthy graph and once built a different one (round 6, scripts/stress_concurrent_builds.py:
// 3 of 48 builds; none since).
static inline hipError_t dr_memset_sync(void *p, int v, size_t bytes)
{
    hipError_t e = hipMemsetAsync(p, v, bytes, nullptr);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(nullptr);
}

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    bool host = false;            // pinned host memory mapped into the device's address space (the host tier of the stored vectors)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }      // error paths (HIPCHK returns) free their temporaries
    int reserve(size_t want, bool zero = false)
    {
        if (want <= n) return 0;
        release();
        hipError_t e = host ? hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocMapped) : hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(DR_E_NODEVICE, "%s(%zu bytes) failed: %s", host ? "hipHostMalloc" : "hipMalloc", want * sizeof(T), hipGetErrorString(e));
        }
        n = want;
        if (zero) {
            if (host) memset(p, 0, want * sizeof(T));
            else { e = dr_memset_sync(p, 0, want * sizeof(T)); if (e != hipSuccess) return fail(DR_E_NODEVICE, "hipMemset failed"); }
        }
        return 0;
    }
    void release() { if (p) (void)(host ? hipHostFree(p) : hipFree(p)); p = nullptr; n = 0; }
};

// One resident query batch: queries in original and chain-major order, the per-query ADC bounds of M1, and what the
// host learnt about it at upload. DR_MAX_RESIDENT of them are addressable through dr_batch_select (bench.py rotates
// distinct batches); the pipelined dr_search_submit path owns DR_PIPE_DEPTH more.
struct QSlot {
    uint32_t nq = 0;
    DevBuf<float> q, qp, pq_ub, pq_max;
    bool qp_valid = false;       // qp holds the chain-major copy of q (dr_search_submit skips it at D <= 256: the register variants read q)
    bool pq_ub_valid = false;    // pq_ub matches these queries and the attached codebook
    bool q_u8 = false;           // every component is an integer in [0, 255] (byte-query variants 13/14)
    void release() { q.release(); qp.release(); pq_ub.release(); pq_max.release(); nq = 0; pq_ub_valid = false; q_u8 = false; qp_valid = false; }
};
#define DR_PIPE_DEPTH 4     // LAUNCHES of the pipelined path in flight (3 until round 3: a launch is finished only after its tie-order pass, which finds
                            // room in the TAIL of the next search kernel -- its latency, not its work, starved a 3-deep pipeline: profiles/r03/ab/ab_c2_companions_v3.log)
static_assert(DR_MAX_TICKETS == 128u, "header and engine agree on the tickets in flight");
#define DR_MAX_JOBS 128      // dr_search_submit tickets in flight (round 4: small submits are coalesced, several jobs ride in one launch)
#define DR_NUM_SETS (DR_PIPE_DEPTH + 1)

// One dr_search_submit ticket. Round 4: a job no longer owns a launch -- it is a run of q0 .. q0 + nq - 1 inside the launch of
// its PipeGroup; the host only touches it again in dr_search_wait (or when its ticket slot / its group's slot is needed).
struct PipeJob {
    bool active = false;
    uint64_t ticket = 0;
    int group = -1;                     // the PipeGroup it rides in
    uint32_t q0 = 0;                    // its first query inside the group's launch
    uint32_t nq = 0, k = 0;
    int rc = 0; std::string err;        // the group's launch failed: what dr_search_wait answers for this ticket
    void *pin_in = nullptr; size_t pin_in_bytes = 0;     // staging for pageable caller memory
    uint32_t *out_ids = nullptr; float *out_dist = nullptr; uint32_t *out_count = nullptr; dr_stats *stats = nullptr;
};

// One launch of the pipelined path: the queries of 1 .. n jobs with equal (k, L, beam_width, mode, band_policy, flags),
// concatenated in one resident batch (slots[DR_MAX_RESIDENT + g]) -- ONE ticket space, one search kernel, one tie-order
// pass, one download; every job copies its own rows out of the group's page-locked result slab. A submit that finds the
// search stream busy is HELD in the open group and rides with the submits that follow it (a 1250-query slice of a
// strong-scaling job occupies 79 of 256 CUs when launched alone: VERDICT r3 item 1); a submit that finds it idle is
// launched at once, so a lone request pays nothing.
struct PipeGroup {
    int state = 0;                      // 0 free, 1 open (collecting jobs, not launched), 2 launched
    uint32_t nq = 0, cap = 0, njobs = 0, live = 0;      // queries so far, room, jobs added, jobs not finished yet
    uint32_t k = 0, L = 0, bw = 0, mode = 0, policy = 0, flags = 0;
    bool q_u8 = true;                   // every job's queries are bytes (the byte-query variants)
    bool with_qp = false;               // jobs upload the chain-major copy too
    int lat_sc = -1;                    // launched with variant 18 (M1): its list-size class, else -1
    int set = -1;                       // the BatchSet holding its outputs once launched
    void *pin_out = nullptr; size_t pin_out_bytes = 0;
    hipEvent_t up_done = nullptr, down_done = nullptr;
};

// Persistent work areas of dr_sharded_submit (comm.inc), owned by the first shard of the call: staging arrays, events, the
// exchange stream and a page-locked slab for the results live across calls. Two of them: batch i+1 is searched while batch i
// is exchanged, merged and downloaded.
#define DR_SHARD_DEPTH 4
// one dr_sharded_submit riding in an exchange: its run of the exchange's queries and where its results go
struct ShardPart { uint64_t ticket; uint32_t q0, nq; uint32_t *out_ids; float *out_dist; uint32_t *out_status; float *out_ms; };
// ONE exchange: the submits it carries (dr_sharded_set_group: a fixed number per exchange, so that every rank forms the same
// exchanges whatever its timing), their queries concatenated on the device, the local lists, the packed keys, the results
struct ShardWork {
    DevBuf<uint32_t> loc_ids, fin_ids, status;
    DevBuf<float> loc_dist, fin_dist, q;
    DevBuf<u64> send_keys, all_keys, words;     // my list + status word; every rank's; the ranks' status words
    hipEvent_t e[5] = {};                       // start, local lists ready, exchange done, final merge done, download done
    hipEvent_t up = nullptr;                    // batch on the device
    hipEvent_t zeroed = nullptr; bool zeroed_valid = false;     // the status words were cleared again behind the last exchange of this work area
    void *pin = nullptr; size_t pin_bytes = 0;
    int state = 0;                              // 0 free, 1 collecting submits, 2 launched (its tickets only have to be collected)
    uint64_t gen = 0;                           // bumped whenever the work area is opened for a new exchange (a waiter that slept re-validates with it)
    std::vector<ShardPart> parts;
    uint32_t nq = 0, cap = 0, k = 0, L = 0, bw = 0, mode = 0, policy = 0, flags = 0; int nranks = 1;
    bool q_u8 = true;
    std::vector<dr_index *> shards; std::vector<uint32_t> id_base; dr_comm *comm = nullptr;
    int local_rc = 0; std::string local_msg;    // this rank's local phase failed: travels as the status word, answered at the wait
    int launch_rc = 0; std::string launch_msg;  // the exchange could not be queued at all: every ticket answers this
    ~ShardWork()
    {
        for (auto &x : e) if (x) (void)hipEventDestroy(x);
        if (up) (void)hipEventDestroy(up);
        if (zeroed) (void)hipEventDestroy(zeroed);
        if (pin) (void)hipHostFree(pin);
    }
};
struct ShardScratch {
    std::mutex mu;
    ShardWork w[DR_SHARD_DEPTH];
    uint64_t next_ticket = 1, next_work = 0;
    int open = -1;                        // the work that is collecting submits, if any
    uint32_t group = 1;                   // submits per exchange (dr_sharded_set_group)
    std::map<uint64_t, std::pair<int, std::string>> failed;   // tickets finished by a later submit with an error
    std::vector<hipEvent_t> shard_done;   // one per local shard
    hipStream_t xs = nullptr;             // merge / exchange / download stream (without a communicator)
    hipStream_t us = nullptr;             // the submits' batches go to the device on the first shard's upload stream (not owned): the exchange
                                          // stream is blocked behind the running searches (it waits for their lists), and the next exchange's
                                          // upload must not wait with it
    ~ShardScratch()
    {
        for (auto &x : shard_done) if (x) (void)hipEventDestroy(x);
        if (xs) (void)hipStreamDestroy(xs);
    }
};

struct dr_index {
    int device = 0;
    uint64_t N = 0;
    uint32_t D = 0, R = 0, medoid = 0, m = 0, sd = 0;
    const DimKernels *kern = nullptr;
    int num_cu = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {};
    // search-kernel launches are timed with a ring of event pairs and harvested at the next sync: dr_batch_run does
    // not wait for its kernel, consecutive steps queue back to back on the stream
    static constexpr int KEV = 32;
    std::array<hipEvent_t, 3> kev[KEV] = {};     // [0] search kernel start (= end of its table build), [1] its end, [2] table build start
    bool kev_lut[KEV] = {};
    int kev_pending = 0;
    double kms_sum = 0.0, lms_sum = 0.0; uint32_t kms_n = 0;
    std::mutex mu;

    DevBuf<float> vecp;
    bool has_vectors = true;      // false: a PQ-only shard (config c5): codes + adjacency, no stored vectors
    DevBuf<uint32_t> adj;
    DevBuf<u64> first;
    DevBuf<uint8_t> codes;
    DevBuf<uint8_t> nbcodes;      // [N][R][m] inline neighbour codes (dr_index_inline_codes), rebuilt before the next search when stale
    bool inline_codes = false, nbcodes_valid = false;
    DevBuf<uint8_t> scan_codes;   // the code words in the skewed flat scan's order (pq_scan_order_kernel), built by the first one-query dr_pq_scan_best
    bool scan_valid = false;
    uint32_t scan_streams = 0, scan_rows = 0, scan_interleaved = 1;      // the copy's layout: V streams of at most L rows, record t of stream k at t * V + k
    DevBuf<float> codebook;
    DevBuf<float> sdc;            // centroid-pair table [m][256][256] (PQ-only builder), built on first use FOR THE CODEBOOK IN PLACE:
    uint64_t codebook_gen = 0, sdc_gen = ~0ull;   // every codebook / m change bumps codebook_gen; ensure_sdc rebuilds a table of another generation
    DevBuf<uint32_t> perm;
    std::vector<uint32_t> h_perm;
    // bit order of the visited bitmaps (build_bit_order): rank[id] = bit position, adjr = rank of every adjacency slot
    DevBuf<float> vnorm2;         // squared norms of the stored vectors, built by the first cosine search (DR_F_COSINE)
    DevBuf<uint32_t> rank, adjr;
    // lossless byte copy of the vectors (integer-valued data, D = 128): 0 not checked yet, 1 present, -1 data does not qualify
    DevBuf<uint8_t> vec8;
    int vec8_state = 0;
    uint32_t vector_tier = DR_TIER_HBM;     // where vecp lives (DR_TIER_HOST: pinned host memory read over PCIe / xGMI)
    bool rank_valid = false, adjr_valid = false, use_adjr = false;
    uint32_t medoid_pos = 0;

    // batch scratch
    QSlot slots[DR_MAX_RESIDENT + DR_PIPE_DEPTH];
    QSlot *cs = &slots[0];        // the selected resident batch (dr_batch_select)
    PipeJob jobs[DR_MAX_JOBS];
    PipeGroup groups[DR_PIPE_DEPTH];
    uint64_t next_ticket = 1, next_group = 0;
    int open_group = -1;          // the group that is collecting jobs (state 1), or -1
    int waiters = 0;              // threads inside dr_search_wait's polling loop
    std::map<uint64_t, std::pair<int, std::string>> failed_tickets;   // tickets whose launch failed, until their dr_search_wait collects the error
    uint32_t coalesce_cap = 32768; // queries a group of small submits may grow to (dr_set_coalesce; 0: every submit is its own launch)
    bool hold_always = false;     // dr_debug_hold: submits are only launched when full / flushed / waited for (tests)
    uint64_t pipe_launches = 0, pipe_tickets = 0, pipe_max_tickets = 0, pipe_queries = 0;   // dr_pipeline_stats
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    // A second search lane (DR_TWO_LANES=1; A/B of round 6, VERDICT r5 item 2 ii): consecutive launches of the pipelined path alternate between
    // `stream` and `stream2`, each lane with its own visited words and table scratch, so that the tail of one launch -- persistent
    // wavefronts running out of tickets -- is filled by the head of the next. A launch takes lane 1 only when it repeats the parameters of
    // the lane-0 launch before it (nothing to prepare: byte rows, bit order, regime are settled) and is ordered behind that launch's
    // preparation by `prep_ev`, recorded on lane 0 right before its search kernel.
    hipStream_t stream2 = nullptr;
    hipEvent_t prep_ev = nullptr;
    bool two_lanes = false, prep_recorded = false;
    int lane_req = 0;             // lane of the launch being queued (set by launch_group_locked around run_locked)
    uint64_t lane0_sig = 0;       // parameters of the last launch queued on lane 0 (0: none)
    uint32_t lane_toggle = 0;
    DevBuf<uint32_t> vis2, vis_epoch2;
    DevBuf<float> lut2;
    uint32_t last_nq = 0;         // batch size of the last launch (dr_batch_download)
    DevBuf<uint32_t> vis, vis_epoch;   // visited words [slots][vis_words] + the per-slot query stamp (search_kernel.hpp)
    // Disk tier of the full-precision rows (dr_index_attach_row_file; the reference's MMapNodeReader, io/diskann_persist.py:201-234): a PQ-only index
    // whose rows stay in index.dat -- DR_F_RERANK reads the rows of a batch's final lists from the file (O_DIRECT where the file system allows it)
    int row_fd = -1; bool row_direct = false;
    uint64_t row_stride = 0, row_off = 0;       // bytes between records, offset of the first record's vector
    void *row_pin = nullptr; size_t row_pin_bytes = 0;      // page-locked staging of the fetched rows + ids
    DevBuf<float> row_raw, row_perm;            // the batch's rows on the device: file order, chain-major
    DevBuf<uint32_t> row_ids;                   // node id of each fetched row
    DevBuf<u64> row_keys;                       // the lists re-keyed by position in row_perm
    DevBuf<uint32_t> lat_spill;        // variant 18: [workgroups][2^bits] visited ids beyond the LDS table, all 0xFFFFFFFF between queries (latency_kernel.hpp)
    DevBuf<float> lut;            // [nq][m][256] per-query tables of the launch being queued (lut_build_kernel), rebuilt by every search that uses them:
                                  // ONE scratch per handle -- searches are serialised on the one search stream (round 3 kept one per resident batch:
                                  // 328 MB each at the bench shape, 20 of them)
    // per-step outputs are triple-buffered: the tie-order pass (finalize) of step i runs on its own stream while
    // the search kernels of steps i+1 and i+2 fill the other sets
    struct BatchSet {
        DevBuf<uint32_t> counter, res_n, tie, out_ids, out_count;
        DevBuf<u64> res_keys, log;
        DevBuf<KStats> stats;
        DevBuf<float> out_dist;
        hipEvent_t search_done = nullptr, fin_start = nullptr, fin_done = nullptr;
        bool fin_pending = false;
        bool counters_zeroed = false;
        int owner_group = -1;         // launched PipeGroup whose results live here (its jobs are finished before reuse)
        uint32_t ticket_base = 0;     // every launch draws exactly nq tickets from counter[0]: never reset
        void release() { counter.release(); res_n.release(); tie.release(); out_ids.release(); out_count.release();
                         res_keys.release(); log.release(); stats.release(); out_dist.release(); }
    } sets[DR_NUM_SETS];      // one set per in-flight pipelined job + the one being queued: the tie-order pass of step i only finds room in the TAILS of the next search kernels
                    // (its 19 VGPRs do not fit beside 3 x 168 per SIMD), so it gets two steps to finish, not one
    int parity = 0, last_set = 0;
    hipStream_t fstream = nullptr;
    DevBuf<uint32_t> fin_stat;    // [1] largest tie-list length since the last sync (finalize_kernel)
    std::map<std::pair<const void *, size_t>, int> occ_cache;
    void *pinned = nullptr; size_t pinned_bytes = 0;      // host slab for result downloads
    void *pin_q = nullptr; size_t pin_q_bytes = 0;        // page-locked staging of a blocking call's pageable query batch
    // A SMALL blocking call (dr_search_batch, <= DR_DIRECT_MAX queries): the kernels write ids / distances / counts / counters straight into the
    // page-locked result slab (no download copies), the tie-order pass runs only if a query was listed for it (a flag word in the slab), on
    // the search stream (no cross-stream hand-over) -- the one-query requests of the API routes pay launches and one synchronisation, nothing else
    int lat_adc_live[DR_NUM_SIZECLASS] = { -1, -1, -1, -1, -1 };     // variant 18's own regime per list-size class (its proof is sharper than search_kernel.hpp's): -1 not
                                  // measured (the scoring wavefronts compute every ADC), 0 the policy asks on < 5 % of the neighbours (they skip it), 1 live
    int lat_sc = 0;               // list-size class of the last variant-18 launch
    uint32_t lat_spill_bits = 0;  // table size lat_spill was last wiped for
    bool lat_skip = false;        // the re-run of a small call whose query outgrew variant 18's visited-id set: search_kernel.hpp serves it
    bool direct = false;          // set by dr_search_batch around run_locked
    bool direct_used = false;     // run_locked's answer: this launch wrote into the slab
    bool direct_fin = false;      // ... and a tie-order pass may be needed (direct_f says with what)
    struct FinalizeParams *direct_f = nullptr;
    bool h2d_pending = false;
    DevBuf<double> f64_q, f64_dist;                 // dr_search_batch_f64 scratch
    DevBuf<uint32_t> f64_ids, f64_cnt, f64_vis;
    DevBuf<KStats> f64_stats;
    uint32_t fin_hint = 0;        // tie-list length to size the tie-order launches for (0: not known yet -> full grid)
    float unit_norm_dev = -1.0f;  // largest | |v|^2 - 1 | over the stored vectors (DR_F_IP), -1 = not measured (reset when rows are written)
    int adc_live = -1;            // M1 on this index: does the rerank policy A4 really consult the ADC? -1 = not measured yet (the last class measured)
    int adc_live_sc[DR_NUM_SIZECLASS] = { -1, -1, -1, -1, -1 };   // ... per list-size class: with the API's L = 20 a 64-neighbour expansion can replace the whole
                                  // list, the skip cannot be proven and the ADC IS evaluated, while L = 100 on the same index never needs it (round 5: the
                                  // regime measured on a first L = 20 request used to pin the slow shared-codebook kernel on every later L = 100 batch)
    DevBuf<u64> phase;
    uint32_t last_k = 0;
    dr_timing timing = {};
    ShardScratch *shs = nullptr;  // dr_sharded_search's work area when this handle is the call's first shard
};

extern "C" int dr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *dr_last_error(void) { return g_err.c_str(); }

static int index_alloc_common(dr_index *ix, uint64_t N, uint32_t D, uint32_t R, uint32_t medoid, int device, bool with_vectors = true,
                              uint32_t vector_tier = DR_TIER_HBM)
{
    if (vector_tier != DR_TIER_HBM && vector_tier != DR_TIER_HOST) return fail(DR_E_ARG, "vector_tier must be DR_TIER_HBM or DR_TIER_HOST");
    if (N == 0 || D == 0 || R == 0) return fail(DR_E_ARG, "N, D and R must be positive");
    if (medoid >= N) return fail(DR_E_ARG, "medoid %u out of range (N=%llu)", medoid, (unsigned long long)N);
    if (N >= 0xFFFFFFFFull) return fail(DR_E_UNSUPPORTED, "N must fit in 32-bit ids");
    // Dimensions without compiled kernels (pydiskann's functions take any D; the facade's SUPPORTED_DIMENSIONS are all built): the index is
    // created with its rows in original element order and searched by the generic traversal (search_f64.hpp, D = 0: the pairwise tree
    // evaluated from D at run time) through dr_search_batch / dr_search_batch_f64 -- M1 ... M4; everything that needs a compiled
    // dimension (builder, brute force, the engine's PQ traversals, resident and pipelined batches) answers DR_E_UNSUPPORTED.
    ix->kern = dr_dim_kernels((int)D);
    if (!ix->kern && D > 32768) return fail(DR_E_UNSUPPORTED, "vector dimension %u: at most 32768", D);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DR_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(DR_E_ARG, "device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ix->num_cu = prop.multiProcessorCount;
    ix->device = device; ix->N = N; ix->D = D; ix->R = R; ix->medoid = medoid;
    HIPCHK(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    if (getenv("DR_COMPANION_PRIORITY") != nullptr) {
        // A/B (round 6): the companions of the search kernel (tie-order pass, bound kernels, copies) on high-priority streams, so that their few
        // wavefronts are dispatched ahead of a queued search kernel's workgroups when slots free up
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const int which = atoi(getenv("DR_COMPANION_PRIORITY"));       // 1: all three, 2: tie order + download, 3: tie order only, 4: upload only
        HIPCHK(hipStreamCreateWithPriority(&ix->fstream, hipStreamNonBlocking, which != 4 ? hi : lo));
        HIPCHK(hipStreamCreateWithPriority(&ix->up_stream, hipStreamNonBlocking, (which == 1 || which == 4) ? hi : lo));
        HIPCHK(hipStreamCreateWithPriority(&ix->down_stream, hipStreamNonBlocking, (which == 1 || which == 2) ? hi : lo));
    } else {
    HIPCHK(hipStreamCreateWithFlags(&ix->fstream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ix->up_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ix->down_stream, hipStreamNonBlocking));
    }
    ix->two_lanes = getenv("DR_TWO_LANES") != nullptr;
    if (const char *ec = getenv("DR_COALESCE_CAP")) ix->coalesce_cap = std::min<uint32_t>((uint32_t)atoi(ec), DR_MAX_GROUP);      // A/B: the default of dr_set_coalesce
    if (ix->two_lanes) { HIPCHK(hipStreamCreateWithFlags(&ix->stream2, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&ix->prep_ev, hipEventDisableTiming)); }
    for (auto &gr : ix->groups) { HIPCHK(hipEventCreateWithFlags(&gr.up_done, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&gr.down_done, hipEventDisableTiming)); }
    for (auto &e : ix->ev) HIPCHK(hipEventCreate(&e));
    for (auto &pr : ix->kev) { HIPCHK(hipEventCreate(&pr[0])); HIPCHK(hipEventCreate(&pr[1])); HIPCHK(hipEventCreate(&pr[2])); }
    for (auto &bs : ix->sets) { HIPCHK(hipEventCreate(&bs.search_done)); HIPCHK(hipEventCreate(&bs.fin_start)); HIPCHK(hipEventCreate(&bs.fin_done)); }
    ix->h_perm.resize(D);
    if (ix->kern) pw_build_perm_rec(0, D, ix->h_perm.data());
    else for (uint32_t e = 0; e < D; e++) ix->h_perm[e] = e;
    if (ix->perm.reserve(D)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpy(ix->perm.p, ix->h_perm.data(), D * sizeof(uint32_t), hipMemcpyHostToDevice));
    ix->has_vectors = with_vectors;
    ix->vector_tier = vector_tier;
    ix->vecp.host = (vector_tier == DR_TIER_HOST);
    if (with_vectors && ix->vecp.reserve((size_t)N * D)) return DR_E_NODEVICE;
    if (ix->adj.reserve((size_t)N * R)) return DR_E_NODEVICE;
    if (ix->first.reserve((size_t)N * ((R + 63) / 64))) return DR_E_NODEVICE;
    return 0;
}

static int need_built_dim(const dr_index *ix, const char *what)
{
    if (ix->kern) return 0;
    return fail(DR_E_UNSUPPORTED, "%s needs a compiled dimension (built: 32,64,96,128,256,768,960,1536); D = %u is served by the generic traversal of "
                "dr_search_batch / dr_search_batch_f64 only (modes M1 ... M4)", what, ix->D);
}

static int need_vectors(const dr_index *ix, const char *what)
{
    if (ix->has_vectors) return 0;
    return fail(DR_E_UNSUPPORTED, "%s needs the stored vectors; this index holds PQ codes only", what);
}

static int build_first_masks(dr_index *ix)
{
    ix->adjr_valid = false;       // the adjacency changed: its bit-position twin is rebuilt before the next search
    ix->nbcodes_valid = false;
    DevBuf<uint32_t> bad;
    if (bad.reserve(1, true)) return DR_E_NODEVICE;
    const uint64_t rows_per_block = 4;
    const uint64_t blocks = std::min<uint64_t>((ix->N + rows_per_block - 1) / rows_per_block, 1u << 20);   // grid-stride kernel
    hipLaunchKernelGGL(first_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, ix->stream, ix->adj.p, ix->N, ix->R,
                       ix->N, ix->first.p, bad.p);
    HIPCHK(hipGetLastError());
    uint32_t hbad = 0;
    HIPCHK(hipMemcpyAsync(&hbad, bad.p, 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    bad.release();
    if (hbad) return fail(DR_E_ARG, "adjacency holds %u neighbour ids >= N", hbad);
    return 0;
}

// uploads `rows` records starting at row0; src points at (vectors | raw records)
static int ingest_chunk(dr_index *ix, const void *src, uint64_t row0, uint64_t rows, uint32_t rec_words, bool with_adj,
                        DevBuf<uint32_t> &staging)
{
    if (staging.reserve((size_t)rows * rec_words)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpyAsync(staging.p, src, (size_t)rows * rec_words * 4, hipMemcpyHostToDevice, ix->stream));
    hipLaunchKernelGGL(ingest_records_kernel, dim3((unsigned)rows), dim3(64), 0, ix->stream, staging.p, rows, ix->D,
                       ix->R, rec_words, ix->perm.p, ix->vecp.p + (size_t)row0 * ix->D,
                       with_adj ? ix->adj.p + (size_t)row0 * ix->R : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ix->stream));
    return 0;
}

extern "C" int dr_index_create(dr_index **out, const float *vectors, const uint32_t *adj, uint64_t N, uint32_t D,
                               uint32_t R, uint32_t medoid, int device)
{
    return dr_index_create_tiered(out, vectors, adj, N, D, R, medoid, device, DR_TIER_HBM);
}

extern "C" int dr_index_create_tiered(dr_index **out, const float *vectors, const uint32_t *adj, uint64_t N, uint32_t D,
                                      uint32_t R, uint32_t medoid, int device, uint32_t vector_tier)
{
    if (!out || !vectors || !adj) return fail(DR_E_ARG, "null argument");
    dr_index *ix = new dr_index();
    int rc = index_alloc_common(ix, N, D, R, medoid, device, true, vector_tier);
    if (rc) { dr_index_close(ix); return rc; }
    DevBuf<uint32_t> staging;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / (D * 4));
    for (uint64_t r0 = 0; r0 < N && !rc; r0 += chunk) {
        const uint64_t rows = std::min(chunk, N - r0);
        rc = ingest_chunk(ix, vectors + (size_t)r0 * D, r0, rows, D, false, staging);
    }
    staging.release();
    if (!rc && hipMemcpy(ix->adj.p, adj, (size_t)N * R * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(DR_E_NODEVICE, "adjacency upload failed");
    if (!rc) rc = build_first_masks(ix);
    if (rc) { dr_index_close(ix); return rc; }
    *out = ix;
    return 0;
}

extern "C" int dr_index_open(dr_index **out, const char *index_dat, uint64_t N, uint32_t D, uint32_t R,
                             uint32_t medoid, int device)
{
    return dr_index_open_tiered(out, index_dat, N, D, R, medoid, device, DR_TIER_HBM);
}

extern "C" int dr_index_open_tiered(dr_index **out, const char *index_dat, uint64_t N, uint32_t D, uint32_t R,
                                    uint32_t medoid, int device, uint32_t vector_tier)
{
    if (!out || !index_dat) return fail(DR_E_ARG, "null argument");
    int fd = open(index_dat, O_RDONLY);
    if (fd < 0) return fail(DR_E_IO, "cannot open %s", index_dat);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return fail(DR_E_IO, "cannot stat %s", index_dat); }
    const uint64_t rec_bytes = 4ull * (D + R);
    if ((uint64_t)st.st_size != N * rec_bytes) {
        close(fd);
        return fail(DR_E_IO, "%s: size %lld != N*4*(D+R) = %llu (N=%llu D=%u R=%u)", index_dat, (long long)st.st_size,
                    (unsigned long long)(N * rec_bytes), (unsigned long long)N, D, R);
    }
    void *map = mmap(nullptr, st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) return fail(DR_E_IO, "mmap failed for %s", index_dat);
    dr_index *ix = new dr_index();
    int rc = index_alloc_common(ix, N, D, R, medoid, device, true, vector_tier);
    DevBuf<uint32_t> staging;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / rec_bytes);
    for (uint64_t r0 = 0; r0 < N && !rc; r0 += chunk) {
        const uint64_t rows = std::min(chunk, N - r0);
        rc = ingest_chunk(ix, (const char *)map + r0 * rec_bytes, r0, rows, D + R, true, staging);
    }
    staging.release();
    munmap(map, st.st_size);
    if (!rc) rc = build_first_masks(ix);
    if (rc) { dr_index_close(ix); return rc; }
    *out = ix;
    return 0;
}

extern "C" int dr_index_set_adjacency(dr_index *ix, const uint32_t *adj)
{
    if (!ix || !adj) return fail(DR_E_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }     // queued searches still read the old rows
    HIPCHK(hipMemcpy(ix->adj.p, adj, (size_t)ix->N * ix->R * 4, hipMemcpyHostToDevice));
    ix->adc_live = -1; ix->nbcodes_valid = false; ix->scan_valid = false;
    return build_first_masks(ix);
}

extern "C" int dr_index_set_pq(dr_index *ix, const float *codebook, const uint8_t *codes, uint32_t m)
{
    if (!ix || !codebook || !codes) return fail(DR_E_ARG, "null argument");
    if (m == 0 || ix->D % m) return fail(DR_E_ARG, "n_subvectors %u must divide D=%u", m, ix->D);
    // (the table kernels of the batched paths hold a centroid of <= 128 elements; the generic traversal of an index without compiled kernels
    //  walks numpy's tree at run time inside a table row too)
    if (ix->D / m > 128 && ix->kern) return fail(DR_E_UNSUPPORTED, "sub_dim %u > 128", ix->D / m);
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }     // queued searches still read the old codes
    if (ix->codes.reserve((size_t)ix->N * m)) return DR_E_NODEVICE;
    if (ix->codebook.reserve((size_t)256 * ix->D)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpy(ix->codes.p, codes, (size_t)ix->N * m, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->codebook.p, codebook, (size_t)256 * ix->D * 4, hipMemcpyHostToDevice));
    ix->m = m; ix->sd = ix->D / m; ix->codebook_gen++;
    for (auto &qs : ix->slots) qs.pq_ub_valid = false;
    ix->adc_live = -1; ix->nbcodes_valid = false; ix->scan_valid = false;
    return 0;
}

// PQ-only shard (config c5: the full vectors of 1e9 x 1536 points are never stored): adjacency + codes + codebook.
// Serves the PQ-only traversal (DR_MODE_M3 with DR_F_USE_PQ = beam_search_with_pq, vamana_graph.py:535-605) and the
// ADC entry points; everything that needs a stored vector answers DR_E_UNSUPPORTED.
extern "C" int dr_index_create_codes(dr_index **out, const uint32_t *adj, uint64_t N, uint32_t D, uint32_t R,
                                     uint32_t medoid, const float *codebook, const uint8_t *codes, uint32_t m, int device)
{
    if (!out || !adj || !codebook || !codes) return fail(DR_E_ARG, "null argument");
    dr_index *ix = new dr_index();
    int rc = index_alloc_common(ix, N, D, R, medoid, device, false);
    if (!rc && hipMemcpy(ix->adj.p, adj, (size_t)N * R * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(DR_E_NODEVICE, "adjacency upload failed");
    if (!rc) rc = build_first_masks(ix);
    if (!rc) rc = dr_index_set_pq(ix, codebook, codes, m);
    if (rc) { dr_index_close(ix); return rc; }
    *out = ix;
    return 0;
}

// Turns a full index (built and encoded on the device) into a PQ-only shard: frees the N*D*4 bytes of vectors.
extern "C" int dr_index_drop_vectors(dr_index *ix)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->m == 0) return fail(DR_E_NOPQ, "dropping the vectors of an index without PQ data would leave nothing to search");
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }
    ix->vecp.release();
    ix->vec8.release(); ix->vec8_state = -1;
    ix->has_vectors = false;
    return 0;
}

extern "C" void dr_index_close(dr_index *ix)
{
    if (ix && ix->row_fd >= 0) { close(ix->row_fd); ix->row_fd = -1; }
    if (ix && ix->row_pin) { (void)hipHostFree(ix->row_pin); ix->row_pin = nullptr; }
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    for (hipStream_t st : { ix->up_stream, ix->stream, ix->stream2, ix->fstream, ix->down_stream }) if (st) (void)hipStreamSynchronize(st);
    ix->lut.release(); ix->lut2.release(); ix->vis2.release(); ix->vis_epoch2.release();
    if (ix->prep_ev) (void)hipEventDestroy(ix->prep_ev);
    if (ix->stream2) (void)hipStreamDestroy(ix->stream2);
    ix->vecp.release(); ix->adj.release(); ix->first.release(); ix->codes.release(); ix->scan_codes.release(); ix->scan_valid = false; ix->codebook.release(); ix->nbcodes.release(); ix->sdc.release();
    ix->perm.release(); ix->vis.release(); ix->vis_epoch.release();
    for (auto &qs : ix->slots) qs.release();
    for (auto &jb : ix->jobs) if (jb.pin_in) (void)hipHostFree(jb.pin_in);
    for (auto &gr : ix->groups) {
        if (gr.pin_out) (void)hipHostFree(gr.pin_out);
        if (gr.up_done) (void)hipEventDestroy(gr.up_done);
        if (gr.down_done) (void)hipEventDestroy(gr.down_done);
    }
    for (auto &bs : ix->sets) {
        bs.release();
        if (bs.search_done) (void)hipEventDestroy(bs.search_done);
        if (bs.fin_start) (void)hipEventDestroy(bs.fin_start);
        if (bs.fin_done) (void)hipEventDestroy(bs.fin_done);
    }
    ix->phase.release(); ix->vnorm2.release(); ix->rank.release(); ix->adjr.release(); ix->fin_stat.release(); ix->vec8.release();
    ix->f64_q.release(); ix->f64_dist.release(); ix->f64_ids.release(); ix->f64_cnt.release(); ix->f64_vis.release(); ix->f64_stats.release();
    if (ix->pinned) (void)hipHostFree(ix->pinned);
    if (ix->pin_q) (void)hipHostFree(ix->pin_q);
    delete ix->direct_f;
    delete ix->shs;
    for (auto &e : ix->ev) if (e) (void)hipEventDestroy(e);
    for (auto &pr : ix->kev) for (auto &e : pr) if (e) (void)hipEventDestroy(e);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    if (ix->fstream) (void)hipStreamDestroy(ix->fstream);
    if (ix->up_stream) (void)hipStreamDestroy(ix->up_stream);
    if (ix->down_stream) (void)hipStreamDestroy(ix->down_stream);
    delete ix;
}

// ------------------------------------------------------------------------------------------------ batches

// every component an integer in [0, 255]? (byte-query variants) -- host_simd.cpp: AVX2 where the CPU has it
extern "C" bool dr_host_all_u8(const float *q, size_t n);
static bool queries_are_u8(const float *queries, size_t n) { return dr_host_all_u8(queries, n); }

// (q0, room: a job of a coalesced group lands behind the jobs before it in a slot sized for the whole group)
static int upload_slot_async(dr_index *ix, QSlot &qs, const float *src, uint32_t nq, hipStream_t st, bool with_qp = true, uint32_t q0 = 0, uint32_t room = 0)
{
    if (room < q0 + nq) room = q0 + nq;
    if (qs.q.reserve((size_t)room * ix->D) || qs.qp.reserve((size_t)room * ix->D)) return DR_E_NODEVICE;
    float *dq = qs.q.p + (size_t)q0 * ix->D;
    HIPCHK(hipMemcpyAsync(dq, src, (size_t)nq * ix->D * 4, hipMemcpyDefault, st));      // (host or device source)
    if (with_qp) {
        hipLaunchKernelGGL(permute_queries_kernel, dim3(nq), dim3(64), 0, st, dq, nq, ix->D, ix->perm.p, qs.qp.p + (size_t)q0 * ix->D);
        HIPCHK(hipGetLastError());
    }
    qs.qp_valid = with_qp;
    qs.nq = q0 + nq;
    qs.pq_ub_valid = false;
    return 0;
}

extern "C" bool dr_host_stage_u8(float *dst, const float *q, size_t n, bool want_check);
static int pin_reserve(void **p, size_t *have, size_t need);

static int upload_queries_locked(dr_index *ix, const float *queries, uint32_t nq, bool wait = true, bool with_qp = true)
{
    if (!queries || nq == 0) return fail(DR_E_ARG, "empty query batch");
    HIPCHK(hipSetDevice(ix->device));
    const bool want_u8 = (ix->vec8_state == 1 || (ix->vec8_state == 0 && ix->D == 128));
    // A blocking call from PAGEABLE memory (dr_search_batch, wait == false): the batch is staged into the handle's page-locked buffer here --
    // one pass that also answers "are all components bytes?" -- and the copy engine takes it from there while the call goes on queueing
    // its kernels (the runtime's own staging of a pageable source blocks the calling thread for the whole copy, and the byte check
    // was a second pass over the batch: 0.16 ms per 10 000 x 128 queries).
    bool staged = false, staged_u8 = false;
    if (!wait) {
        hipPointerAttribute_t at;
        const bool pinned = hipPointerGetAttributes(&at, queries) == hipSuccess && (at.type == hipMemoryTypeHost || at.type == hipMemoryTypeDevice);
        (void)hipGetLastError();
        if (!pinned) {
            const size_t bytes = (size_t)nq * ix->D * 4;
            // (the previous call's copy out of this buffer has completed: every blocking call ends with a synchronised download)
            const int rcp = pin_reserve(&ix->pin_q, &ix->pin_q_bytes, bytes);
            if (rcp) return rcp;
            staged = true;
        }
    }
    HIPCHK(hipEventRecord(ix->ev[0], ix->stream));
    int rc = 0;
    if (staged) {
        // in pieces: the copy engine moves piece i while the calling thread stages piece i + 1 (a 10 000 x 128 batch: 0.25 ms of staging and
        // 0.2 ms of copy that used to run one after the other)
        const uint32_t piece = nq >= 4096 ? (nq + 3) / 4 : nq;
        staged_u8 = want_u8;
        for (uint32_t q0 = 0; q0 < nq && !rc; q0 += piece) {
            const uint32_t n = std::min(piece, nq - q0);
            float *dst = static_cast<float *>(ix->pin_q) + (size_t)q0 * ix->D;
            const bool ok8 = dr_host_stage_u8(dst, queries + (size_t)q0 * ix->D, (size_t)n * ix->D, want_u8);
            staged_u8 = staged_u8 && ok8;
            rc = upload_slot_async(ix, *ix->cs, dst, n, ix->stream, with_qp, q0, nq);
        }
    } else rc = upload_slot_async(ix, *ix->cs, queries, nq, ix->stream, with_qp);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ix->ev[1], ix->stream));
    // dr_search_batch does not wait here: what consumes the queries is queued behind them on the same stream and the call
    // only returns after its download; the copy's duration is read at the next sync. An explicit dr_batch_upload waits,
    // so that the caller's buffer is free on return whatever kind of host memory it is.
    if (wait) HIPCHK(hipStreamSynchronize(ix->stream));
    ix->h2d_pending = true;
    // byte queries? (only asked when byte rows exist)
    ix->cs->q_u8 = staged ? staged_u8 : (want_u8 && queries_are_u8(queries, (size_t)nq * ix->D));
    return 0;
}

static uint32_t next_pow2(uint64_t v)
{
    uint64_t p = 1;
    while (p < v) p <<= 1;
    return (uint32_t)p;
}

// Bit order of the visited bitmaps. A query's visited set is spatially local, its ids are not: with bit = id the 64
// test-and-sets of one expansion touch 64 different cache lines of the slot's bitmap (measured: the whole benefit
// of locality-sorted ids -- 10 % of the kernel at beam_width 8, 15 % without trim -- comes from the bitmap, none
// from the vectors or adjacency rows). So bits are numbered by a coarse clustering instead (nearest of P pivot
// vectors, ids sorted by label), and every adjacency slot carries its neighbour's bit position in a second array
// read with the row (+4R bytes per expansion). Any bijection is correct; results never depend on it.
// Inline neighbour codes: see inline_codes_kernel (engine_kernels.hpp). N*R*m bytes; rebuilt when codes or adjacency change.
static int build_inline_codes(dr_index *ix)
{
    if (!ix->codes.p || ix->m == 0 || (ix->m & 3u)) return fail(DR_E_UNSUPPORTED, "inline neighbour codes need PQ codes with n_subvectors %% 4 == 0");
    if (ix->nbcodes.reserve((size_t)ix->N * ix->R * ix->m)) return DR_E_NODEVICE;
    const uint64_t total = (uint64_t)ix->N * ix->R * (ix->m / 4);
    const unsigned gx = (unsigned)std::min<uint64_t>((total + 255) / 256, (uint64_t)ix->num_cu * 64);
    hipLaunchKernelGGL(inline_codes_kernel, dim3(gx), dim3(256), 0, ix->stream, ix->adj.p, ix->codes.p, ix->N, ix->R, ix->m, ix->nbcodes.p);
    HIPCHK(hipGetLastError());
    ix->nbcodes_valid = true;
    return 0;
}

// rank[id] = the node's place in the locality order (a function of the vectors only). Leaves rank_valid false when the
// index is too small or holds no vectors.
static int build_rank(dr_index *ix)
{
    static const bool off = getenv("DR_NO_BITORDER") != nullptr;
    if (off || ix->N < 32768 || !ix->has_vectors) return 0;
    const uint64_t N = ix->N;
    const uint32_t D = ix->D;
    if (!ix->rank_valid) {
        uint64_t P = (uint64_t)(1.5e13 / ((double)N * D * 3.0));
        P = std::min<uint64_t>(std::min<uint64_t>(P, 4096), N / 64) & ~7ull;
        if (const char *ep = getenv("DR_BITORDER_P")) P = std::min<uint64_t>((uint64_t)atoll(ep), N / 64) & ~7ull;      // A/B: pivots (cells) of the locality order
        if (P < 64) return 0;
        std::vector<uint32_t> h(P);
        for (uint64_t i = 0; i < P; i++) h[i] = (uint32_t)(i * (N / P));
        DevBuf<uint32_t> pid, label;
        DevBuf<float> piv;
        if (pid.reserve(P) || label.reserve(N) || piv.reserve((size_t)P * D) || ix->rank.reserve(N)) return DR_E_NODEVICE;
        HIPCHK(hipMemcpyAsync(pid.p, h.data(), P * 4, hipMemcpyHostToDevice, ix->stream));
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)P), dim3(256), 0, ix->stream, ix->vecp.p, pid.p, (uint32_t)P, D, piv.p);
        HIPCHK(hipGetLastError());
        const float *vecp = ix->vecp.p; const float *pv = piv.p; uint64_t n64 = N; uint32_t p32 = (uint32_t)P; uint32_t *lab = label.p;
        void *args[] = { &vecp, &n64, &pv, &p32, &lab };
        const unsigned grid = (unsigned)std::min<uint64_t>(N, (uint64_t)ix->num_cu * 16);
        HIPCHK(hipLaunchKernel(ix->kern->nearest_pivot, dim3(grid), dim3(64), args, D > 256 ? (size_t)D * 4 : 0, ix->stream));
        std::vector<uint32_t> hl(N), hr(N);
        HIPCHK(hipMemcpyAsync(hl.data(), label.p, N * 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        // cells that are close in space get adjacent bit ranges: the cells are grouped by the nearest of S "super" pivots
        // (every (P/S)-th pivot), so the few cells that share a 128-byte bitmap line belong to one neighbourhood
        static const bool flat = getenv("DR_BITORDER_FLAT") != nullptr;
        std::vector<uint32_t> cell_order(P);
        for (uint64_t c = 0; c < P; c++) cell_order[c] = (uint32_t)c;
        uint64_t S = 64;
        if (const char *es = getenv("DR_BITORDER_S")) S = std::max<uint64_t>(8, (uint64_t)atoll(es));      // A/B: super cells the cells are grouped by
        if (!flat && P >= S * 8) {
            DevBuf<uint32_t> spid, slabel;
            DevBuf<float> spiv;
            if (spid.reserve(S) || slabel.reserve(P) || spiv.reserve((size_t)S * D)) return DR_E_NODEVICE;
            std::vector<uint32_t> hs(S), hsl(P);
            for (uint64_t i = 0; i < S; i++) hs[i] = h[i * (P / S)];
            HIPCHK(hipMemcpyAsync(spid.p, hs.data(), S * 4, hipMemcpyHostToDevice, ix->stream));
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)S), dim3(256), 0, ix->stream, ix->vecp.p, spid.p, (uint32_t)S, D, spiv.p);
            HIPCHK(hipGetLastError());
            const float *pv2 = piv.p; const float *sv = spiv.p; uint64_t np64 = P; uint32_t s32 = (uint32_t)S; uint32_t *slab = slabel.p;
            void *args2[] = { &pv2, &np64, &sv, &s32, &slab };
            HIPCHK(hipLaunchKernel(ix->kern->nearest_pivot, dim3((unsigned)std::min<uint64_t>(P, (uint64_t)ix->num_cu * 16)), dim3(64), args2,
                                   D > 256 ? (size_t)D * 4 : 0, ix->stream));
            HIPCHK(hipMemcpyAsync(hsl.data(), slabel.p, P * 4, hipMemcpyDeviceToHost, ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            for (uint64_t c = 0; c < P; c++) if (hsl[c] >= S) return fail(DR_E_NODEVICE, "bit order: bad super label");
            std::stable_sort(cell_order.begin(), cell_order.end(), [&](uint32_t a, uint32_t b) { return hsl[a] < hsl[b]; });
        }
        std::vector<uint64_t> cnt(P, 0), start(P, 0);
        for (uint64_t i = 0; i < N; i++) { if (hl[i] >= P) return fail(DR_E_NODEVICE, "bit order: bad label"); cnt[hl[i]]++; }
        { uint64_t pos = 0; for (uint64_t c = 0; c < P; c++) { start[cell_order[c]] = pos; pos += cnt[cell_order[c]]; } }
        for (uint64_t i = 0; i < N; i++) hr[i] = (uint32_t)start[hl[i]]++;      // stable: ids ascending inside a cell
        HIPCHK(hipMemcpy(ix->rank.p, hr.data(), N * 4, hipMemcpyHostToDevice));
        pid.release(); label.release(); piv.release();
        ix->rank_valid = true;
    }
    return 0;
}

static int build_bit_order(dr_index *ix)
{
    ix->use_adjr = false;
    ix->adjr_valid = true;
    { const int rcr = build_rank(ix); if (rcr) return rcr; }
    if (!ix->rank_valid) return 0;
    const uint64_t N = ix->N;
    const uint32_t R = ix->R;
    if (ix->adjr.reserve((size_t)N * R)) return DR_E_NODEVICE;
    hipLaunchKernelGGL(map_adjacency_kernel, dim3((unsigned)std::min<uint64_t>((N * R + 255) / 256, 65535)), dim3(256), 0, ix->stream,
                       ix->adj.p, N * R, N, ix->rank.p, ix->adjr.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&ix->medoid_pos, ix->rank.p + ix->medoid, 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->use_adjr = true;
    return 0;
}

// Byte rows (variant 10): SIFT-type descriptors are integers in [0, 255] held as float32. If EVERY component of the
// index is such a value, a second copy as bytes (N*D bytes) lets the landing variant move a quarter of the row bytes
// and take a whole expansion in one burst; v_cvt_f32_ubyte returns exactly the stored float, so distances are
// bit-identical. Data that does not qualify (embeddings, un-rounded descriptors) keeps the float rows.
static int build_byte_rows(dr_index *ix)
{
    static const bool off = getenv("DR_NO_BYTEROWS") != nullptr;
    ix->vec8_state = -1;
    if (off || ix->D != 128 || !ix->has_vectors || ix->vector_tier != DR_TIER_HBM) return 0;     // (a host-tier index keeps its HBM for graph and codes)
    DevBuf<uint32_t> bad;
    if (bad.reserve(1, true) || ix->vec8.reserve((size_t)ix->N * ix->D)) return DR_E_NODEVICE;
    hipLaunchKernelGGL(pack_u8_kernel, dim3((unsigned)std::min<uint64_t>((ix->N * ix->D + 255) / 256, 1u << 16)), dim3(256), 0, ix->stream,
                       ix->vecp.p, ix->N, ix->D, ix->perm.p, ix->vec8.p, bad.p);
    HIPCHK(hipGetLastError());
    uint32_t hb = 1;
    HIPCHK(hipMemcpyAsync(&hb, bad.p, 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    bad.release();
    if (hb) ix->vec8.release(); else ix->vec8_state = 1;
    return 0;
}

// Builder override: search over the under-construction rows (RX slots, degree array instead of first-masks),
// queries already resident in ix->q / ix->qp, no outputs besides res_keys / res_n.
struct BuildOverride { const uint32_t *adjb; const uint32_t *deg; uint32_t RX; uint32_t nq;
                       const float *sdc = nullptr; const uint32_t *pts = nullptr; };   // sdc/pts: PQ-only builder (codes are the queries)

static int g_force_kind = -1;   // test/diagnostic hook: DR_FORCE_KIND environment variable / dr_debug_force_kind
static bool g_force_kind_set = false;
static int sync_locked(dr_index *ix);

// the search kernels of the pending launches have finished (caller synchronised the stream): collect their durations;
// `publish` turns the sum since the last publication into timing.search_kernel_ms (mean per launch)
static void harvest_kernel_times(dr_index *ix, bool publish)
{
    // pairs are recorded in launch order on one stream: the finished ones are a prefix of the ring
    int done = 0;
    while (done < ix->kev_pending && hipEventQuery(ix->kev[done][1]) == hipSuccess) {
        float a = 0, l = 0;
        if (hipEventElapsedTime(&a, ix->kev[done][0], ix->kev[done][1]) == hipSuccess) { ix->kms_sum += a; ix->kms_n++; }
        if (ix->kev_lut[done] && hipEventElapsedTime(&l, ix->kev[done][2], ix->kev[done][0]) == hipSuccess) ix->lms_sum += l;
        done++;
    }
    (void)hipGetLastError();      // hipErrorNotReady from the query of an unfinished pair is not an error
    if (done) {
        std::rotate(&ix->kev[0], &ix->kev[done], &ix->kev[ix->kev_pending]);
        std::rotate(&ix->kev_lut[0], &ix->kev_lut[done], &ix->kev_lut[ix->kev_pending]);
        ix->kev_pending -= done;
    }
    if (publish && ix->kms_n) {
        ix->timing.search_kernel_ms = (float)(ix->kms_sum / ix->kms_n);
        ix->timing.lut_kernel_ms = (float)(ix->lms_sum / ix->kms_n);
        ix->kms_sum = 0.0; ix->lms_sum = 0.0; ix->kms_n = 0;
    }
}

static const uint32_t DR_DIRECT_MAX_BOUND = 256;
// sqrt-ADC upper bound per query (search_kernel.hpp "exact skip"): per-(query, sub-quantiser) maxima, then the ordered sum
static int launch_pq_bound(dr_index *ix, QSlot &qs, uint32_t nq, hipStream_t st, uint32_t q0 = 0, uint32_t room = 0)
{
    // (q0, room: the run of a coalesced group this call covers and the group's capacity; a whole batch: 0, nq)
    if (room < q0 + nq) room = q0 + nq;
    if (qs.pq_ub.reserve(room)) return DR_E_NODEVICE;
    const float *cbp = ix->codebook.p; const float *qp = qs.q.p + (size_t)q0 * ix->D; uint32_t nqv = nq, Dv = ix->D;
    const void *fn = nullptr;
    switch (ix->sd) {
    case 2: fn = reinterpret_cast<const void *>(&pq_bound_max_kernel<2>); break;
    case 3: fn = reinterpret_cast<const void *>(&pq_bound_max_kernel<3>); break;
    case 4: fn = reinterpret_cast<const void *>(&pq_bound_max_kernel<4>); break;
    case 6: fn = reinterpret_cast<const void *>(&pq_bound_max_kernel<6>); break;
    case 8: fn = reinterpret_cast<const void *>(&pq_bound_max_kernel<8>); break;
    case 12: fn = reinterpret_cast<const void *>(&pq_bound_max_kernel<12>); break;
    case 16: fn = reinterpret_cast<const void *>(&pq_bound_max_kernel<16>); break;
    default: break;
    }
    static const bool old_form = getenv("DR_PQ_BOUND_BLOCK") != nullptr;      // A/B: the block-per-query form
    if (fn && !old_form && nq > DR_DIRECT_MAX_BOUND) {      // (a handful of queries: the block-per-(query, sub-quantiser) form below -- the same bits)
        if (qs.pq_max.reserve((size_t)room * ix->m)) return DR_E_NODEVICE;
        float *mxp = qs.pq_max.p + (size_t)q0 * ix->m;
        void *args[] = { &cbp, &qp, &nqv, &Dv, &mxp };
        HIPCHK(hipLaunchKernel(fn, dim3((nq + 63) / 64, ix->m), dim3(64), args, 0, st));
        hipLaunchKernelGGL(pq_bound_sum_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, mxp, nq, ix->m, qs.pq_ub.p + q0);
        HIPCHK(hipGetLastError());
    } else if (!old_form && nq <= DR_DIRECT_MAX_BOUND) {
        // a handful of queries: a block per (query, sub-quantiser), then the ordered sums (DR_PQ_BOUND_BLOCK=1: the one-block-per-query form)
        if (qs.pq_max.reserve((size_t)room * ix->m)) return DR_E_NODEVICE;
        float *mxp = qs.pq_max.p + (size_t)q0 * ix->m;
        hipLaunchKernelGGL(pq_bound_rowmax_kernel, dim3(nq, ix->m), dim3(256), 0, st, cbp, qp, nqv, Dv, ix->sd, mxp);
        hipLaunchKernelGGL(pq_bound_sum_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, mxp, nq, ix->m, qs.pq_ub.p + q0);
        HIPCHK(hipGetLastError());
    } else {
        hipLaunchKernelGGL(pq_bound_kernel, dim3(nq), dim3(256), (size_t)ix->D * 4 + 16, st, ix->codebook.p, qp, ix->D, ix->m, ix->sd, qs.pq_ub.p + q0);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// the encode kernels are instantiated for the sub-vector lengths of the supported shapes (register-resident sub-vector);
// any other length runs the generic form
#define DR_ASSIGN_SD_CASES(F) \
    switch (sd) { \
    case 3: F(3); break; case 4: F(4); break; case 6: F(6); break; case 8: F(8); break; case 12: F(12); break; case 16: F(16); break; \
    case 24: F(24); break; case 30: F(30); break; case 32: F(32); break; case 48: F(48); break; case 64: F(64); break; case 96: F(96); break; \
    default: F(0); break; }

// A2 for a whole batch (engine_kernels.hpp lut_build_kernel): out[nq][m][256] on the engine's stream.
static int launch_lut_build(dr_index *ix, const float *d_queries, uint32_t nq, float *d_out, hipStream_t st = nullptr)
{
    if (!st) st = ix->stream;
    const float *cbp = ix->codebook.p; const float *qp0 = d_queries; uint32_t nqv = nq, Dv = ix->D, mv = ix->m, sdv = ix->sd; float *op = d_out;
    const void *lfn = nullptr;
    switch (ix->sd) {
    case 2: lfn = reinterpret_cast<const void *>(&lut_build_kernel<2>); break;
    case 3: lfn = reinterpret_cast<const void *>(&lut_build_kernel<3>); break;
    case 4: lfn = reinterpret_cast<const void *>(&lut_build_kernel<4>); break;
    case 6: lfn = reinterpret_cast<const void *>(&lut_build_kernel<6>); break;
    case 8: lfn = reinterpret_cast<const void *>(&lut_build_kernel<8>); break;
    case 10: lfn = reinterpret_cast<const void *>(&lut_build_kernel<10>); break;      // (D = 960: m = 96 / 64 / 48 / 32 / 24 / 16 -> 10 / 15 / 20 / 30 / 40 / 60)
    case 12: lfn = reinterpret_cast<const void *>(&lut_build_kernel<12>); break;
    case 15: lfn = reinterpret_cast<const void *>(&lut_build_kernel<15>); break;
    case 16: lfn = reinterpret_cast<const void *>(&lut_build_kernel<16>); break;
    case 20: lfn = reinterpret_cast<const void *>(&lut_build_kernel<20>); break;
    case 24: lfn = reinterpret_cast<const void *>(&lut_build_kernel<24>); break;
    case 30: lfn = reinterpret_cast<const void *>(&lut_build_kernel<30>); break;
    case 32: lfn = reinterpret_cast<const void *>(&lut_build_kernel<32>); break;
    case 40: lfn = reinterpret_cast<const void *>(&lut_build_kernel<40>); break;
    case 48: lfn = reinterpret_cast<const void *>(&lut_build_kernel<48>); break;
    case 60: lfn = reinterpret_cast<const void *>(&lut_build_kernel<60>); break;
    case 64: lfn = reinterpret_cast<const void *>(&lut_build_kernel<64>); break;
    case 96: lfn = reinterpret_cast<const void *>(&lut_build_kernel<96>); break;
    default: break;
    }
    // workgroups: m sub-quantisers x as many query strides as fill the chip a few times over
    const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nq, ((uint64_t)ix->num_cu * 16 + ix->m - 1) / ix->m));
    if (lfn) {
        void *largs[] = { &cbp, &qp0, &nqv, &Dv, &mv, &op };
        HIPCHK(hipLaunchKernel(lfn, dim3(gx, ix->m), dim3(256), largs, 0, st));
    } else {
        void *largs[] = { &cbp, &qp0, &nqv, &Dv, &mv, &sdv, &op };
        HIPCHK(hipLaunchKernel(reinterpret_cast<const void *>(&lut_build_generic_kernel), dim3(gx, ix->m), dim3(256), largs, 0, st));
    }
    return 0;
}

// ---- disk tier: rows of index.dat by node id (io/diskann_persist.py:17-31: record i = D float32 then R uint32) -------------------------------
static int pin_reserve(void **p, size_t *have, size_t need);
static int disk_fetch_rows(dr_index *ix, const uint32_t *ids, size_t n, float *dst)
{
    if (ix->row_fd < 0) return fail(DR_E_IO, "no row file attached");
    const size_t rowb = (size_t)ix->D * 4;
    const unsigned nth = (unsigned)std::max<size_t>(1, std::min<size_t>({ (size_t)16, (size_t)std::max(1u, std::thread::hardware_concurrency()), (n + 63) / 64 }));
    // (first failure wins: 1 = allocation, 2 = short read (end of file inside a row), 3 = pread error with the WORKER's errno kept beside the offset)
    std::atomic<int> bad{0};
    std::atomic<int> bad_errno{0};
    std::atomic<uint64_t> bad_off{0};
    auto flag = [&](int what, int en, uint64_t off) { int z = 0; if (bad.compare_exchange_strong(z, what)) { bad_errno = en; bad_off = off; } };
    auto work = [&](size_t lo, size_t hi) {
        void *blk = nullptr;
        const size_t blkb = ((rowb + 4095) & ~(size_t)4095) + 8192;
        if (ix->row_direct && posix_memalign(&blk, 4096, blkb) != 0) { flag(1, ENOMEM, 0); return; }
        for (size_t i = lo; i < hi && !bad; i++) {
            const uint64_t off = ix->row_off + (uint64_t)ids[i] * ix->row_stride;
            if (ix->row_direct) {
                const uint64_t a0 = off & ~(uint64_t)4095;
                const size_t len = (size_t)(((off + rowb + 4095) & ~(uint64_t)4095) - a0);
                const ssize_t got = pread(ix->row_fd, blk, len, (off_t)a0);
                if (got < (ssize_t)(off - a0 + rowb)) { flag(got < 0 ? 3 : 2, got < 0 ? errno : 0, off); break; }       // (the last block of the file may be short)
                memcpy(dst + i * ix->D, static_cast<unsigned char *>(blk) + (off - a0), rowb);
            } else {
                size_t done = 0;
                while (done < rowb) {
                    const ssize_t got = pread(ix->row_fd, reinterpret_cast<unsigned char *>(dst + i * ix->D) + done, rowb - done, (off_t)(off + done));
                    if (got <= 0) { flag(got < 0 ? 3 : 2, got < 0 ? errno : 0, off + done); break; }
                    done += (size_t)got;
                }
            }
        }
        free(blk);
    };
    if (nth <= 1) work(0, n);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nth; t++) th.emplace_back(work, n * t / nth, n * (t + 1) / nth);
        for (auto &x : th) x.join();
    }
    if (bad == 1) return fail(DR_E_IO, "reading rows from the index file failed: no memory for an aligned read block");
    if (bad == 2) return fail(DR_E_IO, "reading rows from the index file failed: short read at offset %llu (the file ends inside a row)", (unsigned long long)bad_off.load());
    if (bad) return fail(DR_E_IO, "reading rows from the index file failed at offset %llu: %s", (unsigned long long)bad_off.load(), strerror(bad_errno.load()));
    return 0;
}

extern "C" int dr_index_attach_row_file(dr_index *ix, const char *index_dat, uint64_t record_bytes, uint64_t vector_offset)
{
    if (!ix || !index_dat) return fail(DR_E_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->N == 0) return fail(DR_E_ARG, "an empty index has no rows to attach");
    if (ix->has_vectors) return fail(DR_E_ARG, "the index holds its rows already (HBM or host tier): the row file is the tier of a PQ-only index (dr_index_create_codes / dr_index_drop_vectors)");
    if (record_bytes == 0) record_bytes = ((uint64_t)ix->D + ix->R) * 4;        // the reference's record (diskann_persist.py:17-24)
    if (record_bytes < (uint64_t)ix->D * 4) return fail(DR_E_ARG, "a record of %llu bytes cannot hold a %u-dimensional vector", (unsigned long long)record_bytes, ix->D);
    struct stat stt;
    if (stat(index_dat, &stt) != 0) return fail(DR_E_IO, "cannot stat %s: %s", index_dat, strerror(errno));
    if ((uint64_t)stt.st_size < vector_offset + (ix->N - 1) * record_bytes + (uint64_t)ix->D * 4)
        return fail(DR_E_IO, "%s holds %lld bytes, %llu points of %llu-byte records need more", index_dat, (long long)stt.st_size, (unsigned long long)ix->N, (unsigned long long)record_bytes);
    int fd = -1; bool direct = false;
    if (getenv("DR_ROW_FILE_BUFFERED") == nullptr) { fd = open(index_dat, O_RDONLY | O_DIRECT); direct = fd >= 0; }
    if (fd < 0) fd = open(index_dat, O_RDONLY);          // (tmpfs and some network file systems refuse O_DIRECT: buffered reads)
    if (fd < 0) return fail(DR_E_IO, "cannot open %s: %s", index_dat, strerror(errno));
    if (ix->row_fd >= 0) close(ix->row_fd);
    ix->row_fd = fd; ix->row_direct = direct; ix->row_stride = record_bytes; ix->row_off = vector_offset;
    return 0;
}

static int finish_group_locked(dr_index *ix, int g);
// Per-query scratch (insert log, result keys) is sized by the batch: very large batches are processed in chunks.
static const uint32_t DR_MAX_CHUNK = 32768;
static const uint32_t DR_LAT_MAX_NQ = 256;      // launches of at most this many queries (a workgroup per CU): a workgroup per query (variant 18, latency_kernel.hpp) -- measured faster up to 512 at L = 20 (profiles/r05/latency_workgroup_per_query_nq.json)
static const uint32_t DR_DIRECT_MAX = 256;      // dr_search_batch calls of at most this many queries take the direct path (see dr_index::direct)

static int run_locked(dr_index *ix, uint32_t k, uint32_t L, uint32_t bw, uint32_t mode, uint32_t policy, uint32_t flags,
                      const BuildOverride *ov = nullptr)
{
    { const int rcd = need_built_dim(ix, ov ? "the builder" : "this search path (resident / pipelined batches, DR_MODE_PQ / DR_MODE_PQB)"); if (rcd) return rcd; }
    if (ov) ix->cs->nq = ov->nq;
    if (ix->cs->nq == 0) return fail(DR_E_ARG, "no queries uploaded");
    if (ix->cs->nq > 65536 && !ov) return fail(DR_E_UNSUPPORTED, "resident batches are limited to 65536 queries (dr_search_batch chunks larger ones)");
    if (mode < DR_MODE_M1 || mode > DR_MODE_PQB) return fail(DR_E_ARG, "unknown mode %u", mode);
    const bool pqb = (mode == DR_MODE_PQB);      // the batch-per-step PQ-only beam search (pqb_kernel.hpp)
    if ((flags & DR_F_POPS_MASK) && !pqb) return fail(DR_E_ARG, "DR_F_POPS goes with DR_MODE_PQB");
    if (pqb && ov) return fail(DR_E_ARG, "DR_MODE_PQB does not serve the builder");
    if (k == 0) return fail(DR_E_ARG, "k must be positive");
    if (!ov && (policy & 0xFFu) > 1u) return fail(DR_E_UNSUPPORTED, "band policy %u: the literal coin flip (2 | seed << 8) is served by dr_search_batch / dr_search_batch_f64 only "
                                                   "(a sequential walk); the batched paths take 0 (always rerank) or 1 (never)", policy & 0xFFu);
    const bool pq_only = (mode == DR_MODE_M3 && (flags & DR_F_USE_PQ)) || mode == DR_MODE_PQ || pqb || (ov && ov->sdc);   // ADC-only traversals
    const bool rerank = (mode == DR_MODE_PQ || pqb) && (flags & DR_F_RERANK);
    const bool use_pq = (mode == DR_MODE_M1) || pq_only;
    if ((flags & DR_F_IP) && !rerank) return fail(DR_E_ARG, "DR_F_IP goes with DR_F_RERANK (DR_MODE_PQ / DR_MODE_PQB)");
    const uint32_t rerank_top = (flags & DR_F_RERANK_TOP_MASK) >> DR_F_RERANK_TOP_SHIFT;
    if (rerank_top && !(pqb && rerank)) return fail(DR_E_ARG, "DR_F_RERANK_TOP goes with DR_MODE_PQB | DR_F_RERANK");
    if (use_pq && ix->m == 0) return fail(DR_E_NOPQ, "mode %u needs PQ data (dr_index_set_pq)", mode);
    const bool disk_rerank = rerank && !ix->has_vectors && ix->row_fd >= 0;      // the rows of the final lists come from index.dat
    if ((!pq_only || rerank) && !disk_rerank) { const int rcv = need_vectors(ix, rerank ? "DR_F_RERANK (stored vectors or dr_index_attach_row_file)" : "this search mode"); if (rcv) return rcv; }
    if (disk_rerank && (flags & DR_F_IP)) return fail(DR_E_UNSUPPORTED, "DR_F_IP needs the stored vectors' norms: not served from the disk tier");
    if (flags & DR_F_IP) {
        // the inner-product reading of the rerank is only defined on unit-norm rows: measured once per index (largest | |v|^2 - 1 |)
        if (ix->unit_norm_dev < 0.0f) {
            HIPCHK(hipSetDevice(ix->device));
            DevBuf<uint32_t> mx;
            if (mx.reserve(1, true)) return DR_E_NODEVICE;
            hipLaunchKernelGGL(unit_norm_dev_kernel, dim3((unsigned)std::min<uint64_t>((ix->N + 255) / 256, 1u << 16)), dim3(256), 0, ix->stream, ix->vecp.p, ix->N, ix->D, mx.p);
            HIPCHK(hipGetLastError());
            uint32_t hb = 0;
            HIPCHK(hipMemcpyAsync(&hb, mx.p, 4, hipMemcpyDeviceToHost, ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            memcpy(&ix->unit_norm_dev, &hb, 4);
        }
        if (!(ix->unit_norm_dev <= 1e-3f)) return fail(DR_E_UNSUPPORTED, "DR_F_IP needs unit-norm stored vectors: a squared norm differs from 1 by %g", (double)ix->unit_norm_dev);
    }
    // result-list capacity: M1/M4 (and the engine's PQ mode) L, M2 beam_width, M3 k (search_engine.py:468-474;
    // vamana_graph.py:746-750, :586-590)
    const uint32_t cap = (mode == DR_MODE_M2) ? bw : (mode == DR_MODE_M3) ? k : L;
    if (cap == 0) return fail(DR_E_ARG, "result-list capacity is zero (L / beam_width / k)");
    if (cap > DR_MAX_CAPACITY) return fail(DR_E_UNSUPPORTED, "result-list capacity %u > %u", cap, DR_MAX_CAPACITY);
    HIPCHK(hipSetDevice(ix->device));
    // ONE search stream per handle. (Round 3 tried a second one for small pipelined batches, with its own visited-set scratch:
    // slower -- profiles/r03/ab/ab_small_batches_two_search_lanes.json, and the kernel trace of it in profiles/r04/ -- and removed;
    // small submits are coalesced into one launch instead, dr_search_submit.)
    const int lane1 = (!ov && ix->two_lanes && ix->lane_req == 1) ? 1 : 0;
    hipStream_t st = lane1 ? ix->stream2 : ix->stream;
    DevBuf<uint32_t> &vis = lane1 ? ix->vis2 : ix->vis, &vis_epoch = lane1 ? ix->vis_epoch2 : ix->vis_epoch;
    if (lane1 && ix->prep_recorded) HIPCHK(hipStreamWaitEvent(st, ix->prep_ev, 0));

    const int sc = cap <= 64 ? 0 : cap <= 128 ? 1 : cap <= 256 ? 2 : cap <= 512 ? 3 : 4;
    // kernel variant (variants.hpp): the first available variant of the mode's preference list whose LDS footprint
    // fits. M1: byte rows with byte queries (13) > byte rows (11) > float rows landed in LDS (9) > codebook
    // shared in LDS (3) > per-query table (0); ADC traversal: 5 > 2; exact traversal: 14 > 12 > 8 > 1 (the builder
    // uses 1). The byte variants need integer-valued data / queries and are skipped otherwise.
    static const int NCHR_OF_SC[DR_NUM_SIZECLASS] = { 1, 2, 4, 8, 16 };
    auto lds_of = [&](int kd) -> size_t {
        const KindDesc &d = DR_KINDS[dr_kind_pos(kd)];
        const bool qorig_lds = d.pq && !d.lut && !d.rb;                          // search_kernel.hpp QORIG_LDS
        const size_t bloom = (d.rb || d.cb || ix->D <= 256) ? 512 : 0;           // search_kernel.hpp VB_BITS / 8
        const bool adc_only = (kd == 2 || kd == 5 || kd == 15);      // no exact distances: no chain-major query copy in LDS
        const size_t tab = d.lut ? (size_t)(ix->m > (uint32_t)d.treg ? ix->m - d.treg : 0) * 256 * 4 : 0;   // table rows in LDS (the rest in registers)
        const size_t pw = tab + (qorig_lds ? (size_t)ix->D * 4 : 0) +
                          ((ix->D > 256 && !adc_only) ? (size_t)ix->D * 4 : 0) + 512 + bloom + (d.qb ? 528 : 0) +     // (528: search_kernel.hpp ADJPRE)
                          (d.rb ? (size_t)d.rb * ix->D * (d.u8 ? 1 : 4) : (size_t)NCHR_OF_SC[sc] * 64 * 12);
        return (d.cb ? (size_t)256 * ix->D * 4 : 0) + (size_t)d.nw * pw;
    };
    if (!ov && ix->vec8_state == 0) { const int rcb8 = build_byte_rows(ix); if (rcb8) return rcb8; }
    auto usable = [&](int kd) {
        const int pos = dr_kind_pos(kd);
        if (pos < 0 || ix->kern->search[pos][sc] == nullptr || lds_of(kd) > 160 * 1024) return false;
        const KindDesc &d = DR_KINDS[pos];
        if (d.treg && ((ix->m & 15u) != 0 || ix->m < 32u || ix->m > 64u)) return false;     // (register rows: the last whole 16-byte code piece)
        return (!d.u8 || ix->vec8_state == 1) && (!d.qb || (ix->cs->q_u8 && !ov));
    };
    // ADC-only traversals: shared codebook (D <= 128) > table split between LDS and registers (15: twice the wavefronts
    // per CU of 2 at m = 32; DR_NO_TREG=1 switches it off for A/B) > table in LDS
    static const bool no_treg = getenv("DR_NO_TREG") != nullptr;
    static const int PREF_M1[] = { 13, 11, 9, 3, 0 }, PREF_ADC[] = { 5, 15, 2 }, PREF_ADC_NOTREG[] = { 5, 2, 2 }, PREF_EX[] = { 14, 12, 8, 1 }, PREF_BUILD[] = { 1, 8 };
    static const int PREF_M1_LIVE_LUT[] = { 0, 3, 13, 11, 9 }, PREF_M1_LIVE_CB[] = { 3, 0, 13, 11, 9 };
    const bool k_m1 = (mode == DR_MODE_M1), k_adc = pq_only;
    // M1 has two regimes. On SIFT-scale data the rerank policy A4 is provably true for almost every expansion (Q1),
    // the ADC is skipped and the kernel is a pure row gather: vectors landed in LDS, table never built (9, 6).
    // On unit-scale data A4 is live, every new neighbour's ADC is evaluated and the table wants to be in LDS: the
    // per-query table (0) when 8 of them fit a CU, else the shared codebook (3). Which regime an index is in is
    // MEASURED on the first M1 batch an index state serves (its counters are read once that launch has finished, see
    // the end of this function); until then the SIFT-scale preference applies. Results never depend on the variant.
    static const int PREF_BUILD_PQ[] = { 15, 2 }, PREF_BUILD_PQ_NOTREG[] = { 2, 2 };
    // ADC-only traversals at D <= 128 (round 4): the per-query table (2; built for the batch by lut_build_kernel since round 3) beats the
    // shared codebook (5) whenever six tables fit a CU -- c4 shape (D = 96, m = 16): L = 100 beam_width 8 3.99 -> 2.08 ms, L = 350 no trim
    // 21.1 -> 15.0 ms, same results (profiles/r04/ab/ab_c4_adc_table_vs_codebook_10M.jsonl); the codebook form stays for tables that
    // leave fewer than six wavefronts per CU (D = 128, m = 32)
    static const int PREF_ADC_TABLE[] = { 15, 2, 5 }, PREF_ADC_TABLE_NOTREG[] = { 2, 5, 5 };
    const bool adc_table_first = k_adc && !(ov && ov->sdc) && lds_of(2) * 6 <= 160 * 1024;
    const int *pref = k_m1 ? PREF_M1 : (ov && ov->sdc) ? (no_treg ? PREF_BUILD_PQ_NOTREG : PREF_BUILD_PQ)
                    : k_adc ? (adc_table_first ? (no_treg ? PREF_ADC_TABLE_NOTREG : PREF_ADC_TABLE) : (no_treg ? PREF_ADC_NOTREG : PREF_ADC)) : ov ? PREF_BUILD : PREF_EX;
    const int npref = k_m1 ? 5 : (ov && ov->sdc) ? 2 : k_adc ? 3 : ov ? 2 : 4;
    // (round 4: the per-query table wins with as few as five or six wavefronts per CU -- c4 shape, lists of 300-500 entries: 1.38x over
    // the shared codebook at eight, profiles/r04/ab/ab_c4_long_lists_table_vs_codebook.jsonl; it used to need eight to be preferred)
    if (ix->adc_live < 0) { for (int &v : ix->adc_live_sc) v = -1; for (int &v : ix->lat_adc_live) v = -1; }      // (codes / adjacency / rows changed: every class is measured again)
    // (a handful of queries -- the API's one-query requests at L = 20 -- stay on the row-landing kernels in 4-wavefront workgroups even when
    // the ADC is live: one query p50 0.436 -> 0.416 ms, 64 queries 0.664 -> 0.596 ms against the shared-codebook kernel, profiles/r05/latency_small.json)
    if (k_m1 && !ov && ix->adc_live_sc[sc] == 1 && ix->cs->nq > 256) pref = (lds_of(0) * 5 <= 160 * 1024) ? PREF_M1_LIVE_LUT : PREF_M1_LIVE_CB;
    int kind = -1;
    for (int i = 0; i < npref && kind < 0 && !pqb; i++) if (usable(pref[i])) kind = pref[i];
    {
        static bool env_read = false;
        if (!env_read) { const char *e = getenv("DR_FORCE_KIND"); if (e && !g_force_kind_set) g_force_kind = atoi(e); env_read = true; }
        const int g = g_force_kind;
        if (g >= 0 && g <= DR_MAX_KIND_ID && !pqb && usable(g) && !(ov && ov->sdc)) {
            const bool g_m1 = (g == 0 || g == 3 || g == 9 || g == 11 || g == 13 || g == 16 || g == 17), g_adc = (g == 2 || g == 5 || g == 15), g_ex = (g == 1 || g == 8 || g == 12 || g == 14);
            if ((g_m1 && k_m1) || (g_adc && k_adc) || (g_ex && !k_m1 && !k_adc)) kind = g;
        }
    }
    // DR_MODE_PQB: its own kernel family (pqb_kernel.hpp), independent of D: list chunks x passes per step x (m / 16, table rows in registers)
    PqbChoice pqc = {};
    uint32_t pqb_pops = 1, pqb_shift = 0;
    if (pqb) {
        pqb_pops = (flags & DR_F_POPS_MASK) >> DR_F_POPS_SHIFT;
        if (ix->N >= (1ull << 31)) return fail(DR_E_UNSUPPORTED, "DR_MODE_PQB: N must be below 2^31");
        while ((1u << pqb_shift) < ix->R) pqb_shift++;
        if (pqb_pops == 0) pqb_pops = pqb_shift >= 6 ? 1u : std::min(16u, 64u >> pqb_shift);      // default: the rows that fill 64 neighbour slots (R <= 2: 16, the size of the kernel's popped-id array)
        if (pqb_pops > 16) return fail(DR_E_UNSUPPORTED, "DR_MODE_PQB: at most 16 rows per step (asked: %u)", pqb_pops);
        const uint64_t passes = ((uint64_t)pqb_pops << pqb_shift) <= 64 ? 1 : (((uint64_t)pqb_pops << pqb_shift) + 63) / 64;
        if (passes > 4) return fail(DR_E_UNSUPPORTED, "DR_MODE_PQB: pops x next_pow2(R) = %u x %u exceeds 256 neighbour slots per step", pqb_pops, 1u << pqb_shift);
        const int nc = passes <= 1 ? 1 : passes <= 2 ? 2 : 4;
        int treg_pref = -1;
        if (const char *e = getenv("DR_PQB_TREG")) treg_pref = atoi(e);       // A/B: table rows held in registers
        // (round 5's visited filter + compaction for steps of several passes -- it scored the distinct nodes only and was 10-30 % slower,
        //  profiles/r05/ab/ab_pqb_*_v4.jsonl -- was removed in round 6)
        pqc = dr_pqb_choose(sc, nc, ix->m, treg_pref);
        if (!pqc.fn) return fail(DR_E_UNSUPPORTED, "DR_MODE_PQB: no kernel for m=%u, capacity %u", ix->m, cap);
        kind = 20;
    }
    if (kind < 0) return fail(DR_E_UNSUPPORTED, "no kernel variant fits in LDS (D=%u, m=%u, capacity %u)", ix->D, ix->m, cap);
    // A batch smaller than the chip's wavefront slots in 16-wavefront workgroups would fill ceil(nq / 16) CUs and leave the
    // rest idle: the same kernel in 4-wavefront workgroups spreads it over all of them (variants.hpp 16 / 17).
    {
        static const bool no_small = getenv("DR_NO_SMALL_WG") != nullptr;        // A/B
        const int tw = dr_small_twin(kind);
        if (!no_small && !ov && !pqb && tw >= 0 && kind != g_force_kind && usable(tw) && (uint64_t)ix->cs->nq < (uint64_t)ix->num_cu * 16) kind = tw;
    }
    // Variant 18 (latency_kernel.hpp): a workgroup of eight wavefronts per query for the handful of queries of a request -- M1 and the exact
    // traversals (M2, M4, M3 without PQ / cosine), launches of at most DR_LAT_MAX_NQ queries (blocking calls and the pipelined path's groups:
    // the facade's one-query requests ride in those), or forced (dr_debug_force_kind 18: any batch). DR_NO_LATENCY=1 switches it off (A/B, read
    // per call); ix->lat_skip is set for the one re-run of a blocking call whose query outgrew the visited-id set (LDS + its global continuation).
    // The engine's own choice for LONG rows only (D > 256; measured). At D = 128 what made it faster than search_kernel.hpp at the
    // API's L = 20 (one query 0.36 -> 0.22 ms) was its sharper proof that the rerank policy holds, and with the same proof in search_kernel.hpp
    // ("ask later") the one-wavefront kernels answer that query in 0.20 ms (variant 18: 0.23; 0.35 against 0.30 ms at L = 100). At D = 1536 a row
    // is 6 KiB and scoring a node's rows with eight wavefronts instead of one is worth more than the hand-over costs: unit-norm 200k x 1536, one
    // query at the API defaults 0.37 -> 0.27 ms, L = 100 0.66 -> 0.50 ms (16 queries 1.30 -> 0.93), the exact beam search 0.28 -> 0.17 ms
    // (profiles/r05/latency_embeddings*.json). DR_LAT_ALL=1 takes it wherever it is eligible, DR_NO_LATENCY=1 nowhere, dr_debug_force_kind 18
    // for any batch.
    // (lists of at most 256 entries: an M1 query then visits at most 10 L R <= 164 000 nodes, which the visited-id set and its continuation hold --
    //  only blocking calls have the re-run for a query that outgrows them)
    const bool lat_default = cap <= 256 && (k_m1 ? ix->D > 960 : ix->D > 256);      // (D = 768: M1 0.225 against 0.245 ms -- not taken; the exact beam search 0.162 -> 0.146 ms)
    uint32_t lat_vh_bits = 0;
    size_t lat_lds = 0;
    bool lat = false;
    // Only a blocking call on the direct path is served again when a query outgrows the visited-id set (status bit 0). Everywhere else --
    // pipelined groups (the facade's one-query requests), DR_NO_DIRECT, the sharded path -- variant 18 is taken only where NO query can:
    // an M1 query visits at most min(10 L, N) R nodes (R = 128, L = 256: 327 680), the exact traversals at most N; the set holds 3/4 of its
    // LDS slots + 3/4 of its continuation's (lat_capacity below, from the sizes this launch would get).
    const uint64_t lat_visit_bound = k_m1 ? std::min<uint64_t>((uint64_t)L * 10, ix->N) * ix->R : ix->N;
    if (!ov && !pqb && !k_adc && !(flags & DR_F_COSINE) && (k_m1 || mode == DR_MODE_M2 || mode == DR_MODE_M3 || mode == DR_MODE_M4) &&
        ix->kern->latency[k_m1 ? 1 : 0][sc] != nullptr && !ix->lat_skip &&
        (g_force_kind == 18 || (g_force_kind < 0 && ix->cs->nq <= DR_LAT_MAX_NQ && getenv("DR_NO_LATENCY") == nullptr &&
                                (lat_default || getenv("DR_LAT_ALL") != nullptr)))) {
        const size_t nwords = (ix->R + 63) / 64;
        const size_t slot_b = ((nwords * 64 * 4 * (k_m1 ? 3 : 2) + nwords * 16) + 15) & ~(size_t)15;
        const size_t fixed = (k_m1 ? (size_t)ix->m * 1024 : 0) + (ix->D > 256 ? (size_t)ix->D * 4 : 0) + slot_b * 8 + 8 * 512 + (size_t)NCHR_OF_SC[sc] * 64 * 12 + 768 + 1024;
        // visited-id set: 16 384 slots (12 288 ids) where they fit, 32 768 for the long lists; at least 4 096
        uint32_t bits = cap > 256 ? 15 : 14;
        if (const char *e = getenv("DR_LAT_VH_BITS")) bits = (uint32_t)atoi(e);      // tests: a set small enough to overflow
        while (bits > 6 && fixed + ((size_t)4 << bits) > 160 * 1024) bits--;
        uint32_t sbits_would = 18;
        if (const char *es = getenv("DR_LAT_SPILL_BITS")) sbits_would = std::min<uint32_t>((uint32_t)atoi(es), 20u);
        const uint64_t lat_capacity = (3ull << bits) / 4 + (sbits_would ? (3ull << sbits_would) / 4 - 64 : 0);
        const bool has_rerun = ix->direct;       // (dr_search_batch's direct path: the one place with the re-run)
        if (fixed + ((size_t)4 << bits) <= 160 * 1024 && (bits >= 12 || getenv("DR_LAT_VH_BITS") != nullptr) &&
            (has_rerun || g_force_kind == 18 || lat_visit_bound <= lat_capacity)) {
            lat = true; lat_vh_bits = bits; lat_lds = fixed + ((size_t)4 << bits); kind = 18;
        }
    }
    static const KindDesc PQB_DESC = { 20, 1, false, 0, true, true, false, false, 0 };
    static const KindDesc LAT_DESC_M1 = { 18, 8, false, 0, true, true, false, false, 0 }, LAT_DESC_EX = { 18, 8, false, 0, false, false, false, false, 0 };
    const KindDesc &kd_desc = pqb ? PQB_DESC : lat ? (k_m1 ? LAT_DESC_M1 : LAT_DESC_EX) : DR_KINDS[dr_kind_pos(kind)];
    const void *kfn = pqb ? pqc.fn : lat ? ix->kern->latency[k_m1 ? 1 : 0][sc] : ix->kern->search[dr_kind_pos(kind)][sc];
    const int NW = kd_desc.nw;
    const size_t lds = pqb ? pqb_lds_bytes(ix->m, pqc.treg, NCHR_OF_SC[sc], pqc.nc) : lat ? lat_lds : lds_of(kind);
    if (lds > 160 * 1024) return fail(DR_E_UNSUPPORTED, "LDS footprint %zu B exceeds 160 KiB", lds);
    // (kernel attribute + occupancy are asked once per (variant, LDS size): a single-query call is all overhead)
    int occ = 0;
    {
        auto it = ix->occ_cache.find(std::make_pair(kfn, lds));
        if (it != ix->occ_cache.end()) occ = it->second;
        else {
            HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 64 * NW, lds));
            if (occ < 1) occ = 1;
            ix->occ_cache[std::make_pair(kfn, lds)] = occ;
        }
    }
    const uint32_t nq = ix->cs->nq;
    const uint32_t grid = lat ? (uint32_t)std::min<uint64_t>(nq, (uint64_t)occ * ix->num_cu)
                              : (uint32_t)std::min<uint64_t>(((uint64_t)nq + NW - 1) / NW, (uint64_t)occ * ix->num_cu);
    const uint32_t slots = grid * NW;

    // M1 is capped at min(10L, N) expansions (search_engine.py:429); the other variants are bounded by N.
    const bool capped = (mode == DR_MODE_M1 || mode == DR_MODE_PQ || pqb);
    uint64_t max_steps = capped ? std::min<uint64_t>((uint64_t)L * 10, ix->N) : 0xFFFFFFFFull;
    // visited set: per wavefront slot one word per 24 bit positions (+ an 8-bit query stamp: nothing is cleared
    // between queries, search_kernel.hpp) and the slot's stamp counter
    const uint32_t vis_words = (uint32_t)(((ix->N + 23) / 24 + 3) & ~3ull);
    // DR_F_NO_VISITED_SET (DR_MODE_PQ): the traversal keeps NO visited set (SearchParams::novis) -- no visited words (20 MB per
    // wavefront slot on a 1.25e8-point shard), no bit-position twin of the adjacency; same results, more evaluations.
    if ((flags & DR_F_NO_VISITED_SET) && mode != DR_MODE_PQ) return fail(DR_E_ARG, "DR_F_NO_VISITED_SET goes with DR_MODE_PQ");
    // The PQ-only builder's searches run the same way on LARGE shards -- same lists, hence the same graph bit for bit (tests/test_gpu_round2.py),
    // and at the c5 shard's size 15 % off the whole build (1.25e8 points, R = 128: 420 -> 356 s, profiles/r04/scale_c5_shard_R128_build_no_visited_set.json):
    // there the visited words are tens of GB of random-access footprint. DR_BUILD_PQ_NO_VISITED_SET=0 / 1 overrides the size rule.
    bool build_novis = ix->N >= (1ull << 25);
    { const char *e = getenv("DR_BUILD_PQ_NO_VISITED_SET"); if (e) build_novis = e[0] == '1'; }
    const bool novis = (!ov && mode == DR_MODE_PQ && (flags & DR_F_NO_VISITED_SET) != 0) || pqb || (ov && ov->sdc && build_novis);
    if (!novis && !lat && ((size_t)slots * vis_words > vis.n || slots > vis_epoch.n)) {
        // (re)allocation: fresh words and stamps -- queued launches still use the old buffers
        if (vis.p) HIPCHK(hipStreamSynchronize(st));
        if (vis.reserve((size_t)slots * vis_words) || vis_epoch.reserve(slots)) return DR_E_NODEVICE;
        HIPCHK(hipMemsetAsync(vis.p, 0, vis.n * 4, st));
        HIPCHK(hipMemsetAsync(vis_epoch.p, 0, vis_epoch.n * 4, st));
    }
    // accepted-insert log per query (tie replay): 4096 entries cover L <= 256 with room to spare, deeper lists get more
    const uint32_t logcap = pqb ? 0u : std::max<uint32_t>(4096, 16 * cap);      // (DR_MODE_PQB returns a total order: no tie-order pass, no log)
    const int set = ov ? 0 : ix->parity;
    dr_index::BatchSet &bs = ix->sets[set];
    if (bs.owner_group >= 0) { const int rcj = finish_group_locked(ix, bs.owner_group); if (rcj) return rcj; }
    if (bs.fin_pending) {
        // this set's buffers may still be read by the tie-order pass of the step that used it last
        if (ov) HIPCHK(hipStreamSynchronize(ix->fstream));
        else HIPCHK(hipStreamWaitEvent(st, bs.fin_done, 0));
    }
    if (bs.counter.reserve(2) || bs.res_n.reserve(nq) || bs.tie.reserve(nq) || bs.stats.reserve(nq) ||
        bs.res_keys.reserve((size_t)nq * cap) || bs.log.reserve((size_t)nq * logcap) ||
        bs.out_ids.reserve((size_t)nq * std::max<uint32_t>(k, 64)) ||
        bs.out_dist.reserve((size_t)nq * std::max<uint32_t>(k, 64)) || bs.out_count.reserve(nq))
        return DR_E_NODEVICE;

    SearchParams p;
    memset(&p, 0, sizeof p);
    if (!ov && !novis && !pqb && !lat && !ix->adjr_valid) { const int rcb = build_bit_order(ix); if (rcb) return rcb; }      // (DR_MODE_PQB's filter is indexed by id)
    if (!ov && ix->inline_codes && !ix->nbcodes_valid && ix->codes.p && (mode == DR_MODE_M1 || pq_only)) { const int rci = build_inline_codes(ix); if (rci) return rci; }
    p.vecp = ix->vecp.p; p.adj = ix->adj.p; p.first = ix->first.p; p.codes = ix->codes.p; p.codebook = ix->codebook.p;
    p.adjr = (!ov && !novis && ix->use_adjr) ? ix->adjr.p : nullptr; p.medoid_pos = ix->medoid_pos;
    const bool rowpre = getenv("DR_PQ_ROW_PREFETCH") != nullptr;      // A/B (round 4): ids of the predicted next pop's row landed in LDS
    p.novis = novis ? (1u | (rowpre && !(ov && ov->sdc) ? 2u : 0u)) : 0u;
    if (const char *e = getenv("DR_REPREFETCH")) { if (e[0] == '1') p.novis |= 4u; }      // A/B (round 4): the adjacency prefetch with a second chance (1 % slower)
    // "ask later" (search_kernel.hpp, lists of at most 64 entries): every new row first, the rerank policy's ADC only if the sharper test that
    // the exact distances allow cannot prove it true. On while this list-size class is not known to keep the policy busy -- the launch that
    // measures the class runs with it, so "busy" means busy under the sharper test; DR_NO_ASK_LATER=1 switches it off (A/B, read per call).
    if (k_m1 && !ov && sc == 0 && ix->adc_live_sc[sc] != 1 && getenv("DR_NO_ASK_LATER") == nullptr) p.novis |= 8u;
    p.nbcodes = (!ov && ix->inline_codes && ix->nbcodes_valid) ? ix->nbcodes.p : nullptr;
    p.vec8 = ix->vec8_state == 1 ? ix->vec8.p : nullptr;
    // chain-major copy of the batch: the builder hands nothing else; a batch uploaded without it (dr_search_submit, D <= 256)
    // gets it here if this search needs it (large dimensions keep the query in LDS chain-major; the rerank pass reads it)
    const bool rerank_permutes = rerank && ix->cs->nq <= 64;      // (a handful of queries: rerank_kernel applies the permutation itself)
    if (!ov && !ix->cs->qp_valid && ((ix->D > 256 && !lat && !pqb && mode != DR_MODE_PQ) || (rerank && !rerank_permutes))) {      // (variant 18 permutes its query itself; the ADC-only traversals never read it)
        hipLaunchKernelGGL(permute_queries_kernel, dim3(nq), dim3(64), 0, st, ix->cs->q.p, nq, ix->D, ix->perm.p, ix->cs->qp.p);
        HIPCHK(hipGetLastError());
        ix->cs->qp_valid = true;
    }
    p.queries = ix->cs->q.p; p.queries_p = (ov || ix->cs->qp_valid) ? ix->cs->qp.p : nullptr;
    p.N = ix->N; p.D = ix->D; p.R = ix->R; p.m = ix->m; p.sd = ix->sd; p.medoid = ix->medoid; p.nq = nq;
    p.mode = mode; p.k = k; p.cap = cap; p.L = L; p.bw = bw; p.policy = policy; p.flags = flags;
    p.norm = (mode == DR_MODE_M2 || (mode == DR_MODE_M4 && !(flags & DR_F_SQDIST))) ? 1u : 0u;
    p.max_steps = (uint32_t)std::min<uint64_t>(max_steps, 0xFFFFFFFFull);
    p.vis = vis.p; p.vis_words = vis_words; p.vis_epoch = vis_epoch.p;
    // (bit 8: the scoring wavefronts skip the ADC -- this index / list-size class was measured: the rerank policy is proven true for
    //  nearly every row, the rest compute it inside the decisions; DR_LAT_EAGER_ADC=1 / DR_LAT_LAZY_ADC=1 pin it for A/B and tests)
    uint32_t lat_sbits = 0;
    if (lat) {
        // the continuation of the visited-id set in global memory: 2^18 slots per workgroup, 196 608 ids (an M1 query of the auto-selected shapes
        // visits at most min(10 L, N) R <= 163 840 nodes: it cannot run out); DR_LAT_SPILL_BITS overrides (0: none -- the overflow status and the re-run, tests)
        lat_sbits = 18;
        if (const char *es = getenv("DR_LAT_SPILL_BITS")) lat_sbits = (uint32_t)atoi(es);
        if (lat_sbits > 20) lat_sbits = 20;
        if (lat_sbits) {
            const size_t need = (size_t)grid << lat_sbits;
            if (need > ix->lat_spill.n) {
                if (ix->lat_spill.p) HIPCHK(hipStreamSynchronize(st));
                if (ix->lat_spill.reserve(need)) return DR_E_NODEVICE;
                HIPCHK(hipMemsetAsync(ix->lat_spill.p, 0xFF, ix->lat_spill.n * 4, st));
                ix->lat_spill_bits = lat_sbits;
            } else if (ix->lat_spill_bits != lat_sbits) {
                // (another table size than the tables were wiped for: the stride changed -- all of it again)
                HIPCHK(hipMemsetAsync(ix->lat_spill.p, 0xFF, ix->lat_spill.n * 4, st));
                ix->lat_spill_bits = lat_sbits;
            }
        }
    }
    {
        bool lazy = lat && k_m1 && ix->lat_adc_live[sc] == 0;
        if (getenv("DR_LAT_EAGER_ADC")) lazy = false;
        if (getenv("DR_LAT_LAZY_ADC")) lazy = lat && k_m1;
        uint32_t want = 0;
        if (const char *ew = getenv("DR_LAT_WANT")) want = (uint32_t)atoi(ew) & 15u;      // A/B: nodes scored per round
        p.vh_bits = lat_vh_bits | (lazy ? 256u : 0u) | (lat_sbits << 16) | (want << 24);
        if (lat) p.vis = ix->lat_spill.p;
    }
    p.counter = bs.counter.p;
    p.res_keys = bs.res_keys.p; p.res_n = bs.res_n.p; p.stats = bs.stats.p;
    p.tie_list = bs.tie.p; p.tie_count = bs.counter.p + 1;
    p.log = bs.log.p; p.logcap = logcap;
    if (getenv("DR_NO_LOG") != nullptr) p.logcap = 0;      // timing / counter experiment only (with DR_SKIP_FINALIZE=1): no insert log is written, the tie order of tied queries is then wrong
    p.out_ids = bs.out_ids.p; p.out_dist = bs.out_dist.p; p.out_count = bs.out_count.p;
    // (small blocking call: outputs land in the page-locked slab -- not on the one call per list-size class whose counters are read back below)
    const bool direct = ix->direct && !ov && !(mode == DR_MODE_M1 && ix->adc_live_sc[sc] < 0);
    ix->direct_fin = false; ix->direct_used = direct;
    if (direct) {
        const size_t b_ids = (size_t)nq * k * 4, b_cnt = (size_t)nq * 4, b_st = (size_t)nq * sizeof(KStats);
        const size_t need = 2 * b_ids + b_cnt + b_st + 16;
        if (ix->pinned_bytes < need) {
            if (ix->pinned) (void)hipHostFree(ix->pinned);
            ix->pinned = nullptr; ix->pinned_bytes = 0;
            if (hipHostMalloc(&ix->pinned, need, hipHostMallocDefault) != hipSuccess) return fail(DR_E_NODEVICE, "hipHostMalloc(%zu) failed", need);
            ix->pinned_bytes = need;
        }
        unsigned char *hp = static_cast<unsigned char *>(ix->pinned);
        p.out_ids = reinterpret_cast<uint32_t *>(hp); p.out_dist = reinterpret_cast<float *>(hp + b_ids);
        p.out_count = reinterpret_cast<uint32_t *>(hp + 2 * b_ids); p.stats = reinterpret_cast<KStats *>(hp + 2 * b_ids + b_cnt);
        p.tie_flag = reinterpret_cast<uint32_t *>(hp + 2 * b_ids + b_cnt + b_st);
        *p.tie_flag = 0u;
    }
    p.phase = nullptr;
    p.pq_ub = nullptr;
    p.vnorm2 = nullptr;
    if (flags & DR_F_COSINE) {
        if (mode != DR_MODE_M3 || (flags & DR_F_USE_PQ)) return fail(DR_E_ARG, "DR_F_COSINE goes with DR_MODE_M3 without DR_F_USE_PQ (the reference's only cosine traversal)");
        if (!ix->vnorm2.p) {
            if (ix->vnorm2.reserve(ix->N)) return DR_E_NODEVICE;
            hipLaunchKernelGGL(row_norm2_kernel, dim3((unsigned)std::min<uint64_t>((ix->N + 255) / 256, 1u << 16)), dim3(256), 0, st, ix->vecp.p, ix->N, ix->D, ix->vnorm2.p);
            HIPCHK(hipGetLastError());
        }
        p.vnorm2 = ix->vnorm2.p;
    }
    p.perm = ix->perm.p;
    if (mode == DR_MODE_M1 && lat && !ix->cs->pq_ub_valid) {
        p.pq_ub = nullptr;      // variant 18 sums the row maxima of the table it holds in LDS: no bound kernels in front of a small launch
    } else if (mode == DR_MODE_M1) {
        // per-query ADC upper bounds: a function of (queries, codebook) only, computed once per uploaded batch
        if (!ix->cs->pq_ub_valid) {
            { const int rcb = launch_pq_bound(ix, *ix->cs, nq, st); if (rcb) return rcb; }
            ix->cs->pq_ub_valid = true;
        }
        p.pq_ub = ix->cs->pq_ub.p;
    }
#ifdef DR_PHASE_TIMING
    if (!ov) { if (ix->phase.reserve((size_t)nq * 8, true)) return DR_E_NODEVICE; p.phase = ix->phase.p; }
#endif
#ifdef DR_TRACE_VIS
    if (!ov) { if (ix->phase.reserve((size_t)256 * 8192, true)) return DR_E_NODEVICE; p.phase = ix->phase.p; }
#endif
    if (ov) {
        p.sdc = ov->sdc; p.build_pts = ov->pts;
        p.adj = ov->adjb; p.first = nullptr; p.deg = ov->deg; p.R = ov->RX; p.logcap = 0;
        p.tie_list = nullptr; p.out_ids = nullptr; p.out_dist = nullptr; p.out_count = nullptr;
    }

    static const bool dbg = getenv("DR_DEBUG") != nullptr;
    if (dbg) { fprintf(stderr, "[dr] search kind=%d sc=%d NW=%d grid=%u lds=%zu occ=%d nq=%u cap=%u slots=%u vis_words=%u\n", kind, sc, NW, grid, lds, occ, nq, cap, slots, vis_words); fflush(stderr); }
    if (!bs.counters_zeroed) { HIPCHK(hipMemsetAsync(bs.counter.p, 0, 8, st)); bs.counters_zeroed = true; bs.ticket_base = 0; }
    // Ticket counter: slot s starts on query s and every later query is a ticket; the started slots draw
    // (nq - started) successful tickets plus one failing ticket each = exactly nq per launch, so the counter is
    // monotonic and the launch only needs its starting value (no per-step memset on the search stream). The tie-list
    // length (counter[1]) is zeroed on the tie-order stream after its consumer.
    p.ticket_base = bs.ticket_base;
    if (lat) ix->lat_sc = sc;
    if (!lat) bs.ticket_base += nq;      // (variant 18 walks its queries by workgroup index: no tickets drawn)
    if (!ov && ix->kev_pending == dr_index::KEV) harvest_kernel_times(ix, false);
    if (!ov && ix->kev_pending == dr_index::KEV) { HIPCHK(hipStreamSynchronize(ix->stream)); if (ix->stream2) HIPCHK(hipStreamSynchronize(ix->stream2)); harvest_kernel_times(ix, false); }
    p.lut_g = nullptr;
    const bool want_lut = kd_desc.lut && !(ov && ov->sdc);
    if (want_lut) {
        // The per-query tables T[q][j][c] (A2, fast_pq.py:294-318) of the whole batch, built at full occupancy right
        // before the search kernel that lands them in LDS (engine_kernels.hpp lut_build_kernel). Built by EVERY search --
        // a table is part of its query's search, not of the upload -- and timed separately (dr_timing.lut_kernel_ms).
        DevBuf<float> &lutb = lane1 ? ix->lut2 : ix->lut;
        if (lutb.reserve((size_t)nq * ix->m * 256)) return DR_E_NODEVICE;
        if (!ov) HIPCHK(hipEventRecord(ix->kev[ix->kev_pending][2], st));
        { const int rcl = launch_lut_build(ix, ix->cs->q.p, nq, lutb.p, st); if (rcl) return rcl; }
        p.lut_g = lutb.p;
    }
    PqbParams pp;
    if (pqb) {
        memset(&pp, 0, sizeof pp);
        pp.adj = ix->adj.p; pp.first = ix->first.p; pp.codes = ix->codes.p; pp.nbcodes = p.nbcodes; pp.lut_g = p.lut_g;
        pp.N = ix->N; pp.R = ix->R; pp.m = ix->m; pp.medoid = ix->medoid; pp.nq = nq; pp.k = k; pp.cap = cap; pp.bw = bw;
        pp.pops = pqb_pops; pp.max_steps = p.max_steps; pp.rs_shift = pqb_shift;
        pp.counter = p.counter; pp.ticket_base = p.ticket_base;
        pp.res_keys = p.res_keys; pp.res_n = p.res_n; pp.stats = p.stats; pp.out_ids = p.out_ids; pp.out_dist = p.out_dist; pp.out_count = p.out_count;
        pp.phase = p.phase;
    }
    void *args[] = { pqb ? (void *)&pp : (void *)&p };
    if (!ov && ix->two_lanes && !lane1) { HIPCHK(hipEventRecord(ix->prep_ev, st)); ix->prep_recorded = true; }      // (everything this index state needed has been queued on lane 0 by now)
    if (!ov) { ix->kev_lut[ix->kev_pending] = want_lut; HIPCHK(hipEventRecord(ix->kev[ix->kev_pending][0], st)); }
    {
        const hipError_t le = hipLaunchKernel(kfn, dim3(grid), dim3(64 * NW), args, lds, st);
        if (le != hipSuccess) { bs.counters_zeroed = false; return fail(DR_E_NODEVICE, "search kernel launch failed: %s", hipGetErrorString(le)); }
    }
    if (ov) return 0;   // the builder consumes res_keys / res_n directly on the stream
    HIPCHK(hipEventRecord(ix->kev[ix->kev_pending][1], st));
    ix->kev_pending++;
    if (rerank && disk_rerank) {
        // The disk tier: the traversal ran on the code words in HBM; the rows of the final lists are read from index.dat now -- the lists come to the
        // host, a pool of threads preads each row (aligned blocks under O_DIRECT), the rows go up, are permuted to the chain-major layout by the ingest
        // kernel and scored by the same rerank kernel (positions in the fetched buffer as keys, node ids through id_map). In query chunks of at most
        // 256 MiB of rows. Synchronous: this tier is bound by the file system, not by the GPU.
        HIPCHK(hipStreamSynchronize(st));
        std::vector<u64> hk((size_t)nq * cap); std::vector<uint32_t> hn(nq);
        HIPCHK(hipMemcpy(hk.data(), bs.res_keys.p, hk.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hn.data(), bs.res_n.p, hn.size() * 4, hipMemcpyDeviceToHost));
        const size_t rowb = (size_t)ix->D * 4;
        const uint32_t qchunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(nq, ((size_t)256 << 20) / (rowb * cap)));
        for (uint32_t q0 = 0; q0 < nq; q0 += qchunk) {
            const uint32_t nqc = std::min(qchunk, nq - q0);
            size_t total = 0;
            for (uint32_t q = 0; q < nqc; q++) total += std::min<uint32_t>(std::min<uint32_t>(hn[q0 + q], cap), rerank_top ? rerank_top : cap);
            const size_t need = total * rowb + total * 4 + (size_t)nqc * cap * 8 + 64;
            { const int rcp = pin_reserve(&ix->row_pin, &ix->row_pin_bytes, need); if (rcp) return rcp; }
            float *rows_h = static_cast<float *>(ix->row_pin);
            uint32_t *ids_h = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(ix->row_pin) + total * rowb);
            u64 *keys_h = reinterpret_cast<u64 *>(static_cast<unsigned char *>(ix->row_pin) + ((total * rowb + total * 4 + 7) & ~(size_t)7));
            size_t pos = 0;
            for (uint32_t q = 0; q < nqc; q++) {
                const uint32_t n = std::min<uint32_t>(std::min<uint32_t>(hn[q0 + q], cap), rerank_top ? rerank_top : cap);
                for (uint32_t i = 0; i < cap; i++) {
                    const u64 key = hk[(size_t)(q0 + q) * cap + i];
                    if (i < n) { ids_h[pos] = ~(uint32_t)key; keys_h[(size_t)q * cap + i] = (key & 0xFFFFFFFF00000000ull) | (uint32_t)(~(uint32_t)pos); pos++; }
                    else keys_h[(size_t)q * cap + i] = ~0ull;
                }
            }
            { const int rcf = disk_fetch_rows(ix, ids_h, total, rows_h); if (rcf) return rcf; }
            if (ix->row_raw.reserve(std::max<size_t>(total, 1) * ix->D) || ix->row_perm.reserve(std::max<size_t>(total, 1) * ix->D) ||
                ix->row_ids.reserve(std::max<size_t>(total, 1)) || ix->row_keys.reserve((size_t)nqc * cap)) return DR_E_NODEVICE;
            if (total) {
                HIPCHK(hipMemcpyAsync(ix->row_raw.p, rows_h, total * rowb, hipMemcpyHostToDevice, st));
                HIPCHK(hipMemcpyAsync(ix->row_ids.p, ids_h, total * 4, hipMemcpyHostToDevice, st));
                hipLaunchKernelGGL(ingest_records_kernel, dim3((unsigned)total), dim3(256), 0, st, reinterpret_cast<const u32 *>(ix->row_raw.p), (u64)total, ix->D, 0u, ix->D,
                                   ix->perm.p, ix->row_perm.p, (u32 *)nullptr);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipMemcpyAsync(ix->row_keys.p, keys_h, (size_t)nqc * cap * 8, hipMemcpyHostToDevice, st));
            const float *vecp = ix->row_perm.p; const float *qpp = ix->cs->qp_valid ? ix->cs->qp.p + (size_t)q0 * ix->D : nullptr; const u64 *rkp = ix->row_keys.p;
            const uint32_t *rnp = bs.res_n.p + q0; const float *qorig = ix->cs->q.p + (size_t)q0 * ix->D; const uint32_t *permp = ix->perm.p; const uint32_t *idm = ix->row_ids.p;
            uint32_t capv = cap, kv = k, nqv = nqc; uint32_t *oi = p.out_ids + (size_t)q0 * k; float *od = p.out_dist + (size_t)q0 * k; uint32_t *oc = p.out_count + q0;
            KStats *stp = p.stats + q0;
            uint32_t ipv = 0u;
            uint32_t topv = rerank_top;      // (the same cut as the host's above: only those rows were fetched)
            void *rargs[] = { &vecp, &qpp, &nqv, &rkp, &rnp, &capv, &kv, &oi, &od, &oc, &stp, &ipv, &qorig, &permp, &idm, &topv };
            const size_t rlds = (ix->D > 256 ? (size_t)ix->D * 4 : 0) + (size_t)cap * 8;
            const unsigned rblock = nqc <= (uint32_t)ix->num_cu * 2 ? 512u : 64u;
            HIPCHK(hipLaunchKernel(ix->kern->rerank, dim3(std::min<uint32_t>(nqc, (uint32_t)ix->num_cu * 16)), dim3(rblock), rargs, rlds, st));
            HIPCHK(hipStreamSynchronize(st));       // (the staging buffer is reused by the next chunk)
        }
    } else if (rerank) {
        // DR_MODE_PQ + DR_F_RERANK: exact squared L2 of the final list's entries, k best in (distance, id) order
        const float *vecp = ix->vecp.p; const float *qpp = ix->cs->qp_valid ? ix->cs->qp.p : nullptr; const u64 *rkp = bs.res_keys.p; const uint32_t *rnp = bs.res_n.p;
        const float *qorig = ix->cs->q.p; const uint32_t *permp = ix->perm.p; const uint32_t *idm = nullptr;
        uint32_t capv = cap, kv = k, nqv = nq; uint32_t *oi = p.out_ids; float *od = p.out_dist; uint32_t *oc = p.out_count;
        KStats *stp = p.stats;
        uint32_t ipv = (flags & DR_F_IP) ? 1u : 0u;
        uint32_t topv = rerank_top;
        void *rargs[] = { &vecp, &qpp, &nqv, &rkp, &rnp, &capv, &kv, &oi, &od, &oc, &stp, &ipv, &qorig, &permp, &idm, &topv };
        const size_t rlds = (ix->D > 256 ? (size_t)ix->D * 4 : 0) + (size_t)cap * 8;
        // (a handful of queries: eight wavefronts per query share its list's rows; a batch that fills the chip: one wavefront per query)
        const unsigned rblock = nq <= (uint32_t)ix->num_cu * 2 ? 512u : 64u;
        HIPCHK(hipLaunchKernel(ix->kern->rerank, dim3(std::min<uint32_t>(nq, (uint32_t)ix->num_cu * 16)), dim3(rblock), rargs, rlds, st));
    }

    // tie replay for the queries the search kernel listed: one wavefront per query, heap in registers. It runs on
    // its own stream so that it overlaps the NEXT step's search kernel (it needs 19 VGPRs and no LDS, so its
    // wavefronts fit beside the search kernel's); dr_batch_sync / dr_batch_download wait for it.
    FinalizeParams f;
    f.res_keys = bs.res_keys.p; f.res_n = bs.res_n.p; f.tie_list = bs.tie.p; f.tie_count = bs.counter.p + 1;
    f.log = bs.log.p; f.stats = p.stats;
    f.logcap = logcap; f.cap = cap; f.k = k; f.mode = mode;
    f.out_ids = p.out_ids; f.out_dist = p.out_dist;
    if (!ix->fin_stat.p) { if (ix->fin_stat.reserve(1, true)) return DR_E_NODEVICE; }
    f.ntie_stat = ix->fin_stat.p;
    f.serial = getenv("DR_FINALIZE_SERIAL") != nullptr ? 1u : 0u;
    if (direct) {
        // the caller (dr_search_batch) synchronises the search stream, looks at the flag word and runs the tie-order pass only if it is set
        if (!ix->direct_f) ix->direct_f = new FinalizeParams();
        *ix->direct_f = f;
        ix->direct_fin = !rerank && !pqb;
        ix->timing.grid = grid; ix->timing.block = 64 * NW; ix->timing.lds_bytes = (uint32_t)lds;
        ix->timing.waves_per_cu = (uint32_t)(occ * NW); ix->timing.variant = (uint32_t)kind;
        ix->last_k = k; ix->last_nq = nq; ix->last_set = set;
        ix->parity = (ix->parity + 1) % DR_NUM_SETS;
        return 0;
    }
    HIPCHK(hipEventRecord(bs.search_done, st));
    HIPCHK(hipStreamWaitEvent(ix->fstream, bs.search_done, 0));
    HIPCHK(hipEventRecord(bs.fin_start, ix->fstream));
    static const bool skip_fin = getenv("DR_SKIP_FINALIZE") != nullptr;   // timing experiment only: tie order is then wrong
    if (!skip_fin && !rerank && !pqb)      // (the rerank pass / DR_MODE_PQB have already written a total (distance, id) order)
    {
        // One wavefront per tied query, 4 per workgroup, spread over the chip (packing 16 per CU slowed each replay by a
        // third). Few queries tie (29 of 10 000 on the bench data), so the grid is sized for twice the largest tie list
        // seen at the last sync (waves loop if there are more) instead of one wave slot per query of the batch. One
        // replay takes ~0.6 ms of serial heap work: hidden behind the next search kernel, exposed once at the sync.
        unsigned fgrid = (unsigned)std::min<uint64_t>(((uint64_t)nq + 3) / 4, (uint64_t)ix->num_cu * 8);
        if (ix->fin_hint) fgrid = std::min<unsigned>(fgrid, (2 * ix->fin_hint + 64 + 3) / 4);
        hipLaunchKernelGGL(finalize_kernel, dim3(fgrid), dim3(256), 4 * ((size_t)cap + 2 + 64) * 8, ix->fstream, f);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(bs.counter.p + 1, 0, 4, ix->fstream));
    HIPCHK(hipEventRecord(bs.fin_done, ix->fstream));
    bs.fin_pending = true;
    // no host wait here: the next step may be queued right away (dr_batch_sync / dr_batch_download wait)
    ix->timing.grid = grid; ix->timing.block = 64 * NW; ix->timing.lds_bytes = (uint32_t)lds;
    ix->timing.waves_per_cu = (uint32_t)(occ * NW); ix->timing.variant = (uint32_t)kind;
    if (k_m1 && ix->adc_live_sc[sc] < 0) {
        // regime of this (graph, PQ) state: did the rerank policy really consult the ADC on this batch? (the one
        // launch per index state that is waited for on the host)
        HIPCHK(hipStreamSynchronize(st));
        std::vector<KStats> st(std::min<uint32_t>(nq, 1024));
        HIPCHK(hipMemcpy(st.data(), bs.stats.p, st.size() * sizeof(KStats), hipMemcpyDeviceToHost));
        uint64_t evald = 0, all = 0;
        for (const KStats &x : st) { evald += x.pq_evaluated; all += x.pq; }
        ix->adc_live = ix->adc_live_sc[sc] = (2 * evald > all) ? 1 : 0;
    }
    ix->last_k = k;
    ix->last_nq = nq;
    ix->last_set = set;
    ix->parity = (ix->parity + 1) % DR_NUM_SETS;
    return 0;
}

// waits for every outstanding kernel of the handle (the overlapped tie-order pass included)
static int sync_locked(dr_index *ix)
{
    HIPCHK(hipStreamSynchronize(ix->up_stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    if (ix->stream2) HIPCHK(hipStreamSynchronize(ix->stream2));
    HIPCHK(hipStreamSynchronize(ix->fstream));
    HIPCHK(hipStreamSynchronize(ix->down_stream));
    harvest_kernel_times(ix, true);
    if (ix->h2d_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ix->ev[0], ix->ev[1]) == hipSuccess) ix->timing.h2d_ms = ms;
        ix->h2d_pending = false;
    }
    if (ix->fin_stat.p && ix->last_nq >= 1024) {     // (small batches launch a small pass anyway: no blocking readback for them)
        uint32_t mx = 0;
        HIPCHK(hipMemcpy(&mx, ix->fin_stat.p, 4, hipMemcpyDeviceToHost));
        HIPCHK(dr_memset_sync(ix->fin_stat.p, 0, 4));
        ix->fin_hint = std::max<uint32_t>(mx, 16);
    }
    dr_index::BatchSet &bs = ix->sets[ix->last_set];
    if (bs.fin_pending) {
        float b = 0;
        if (hipEventElapsedTime(&b, bs.fin_start, bs.fin_done) == hipSuccess) ix->timing.finalize_kernel_ms = b;
    }
    for (auto &x : ix->sets) x.fin_pending = false;
    return 0;
}

static int download_locked(dr_index *ix, uint32_t *out_ids, float *out_dist, uint32_t *out_count, dr_stats *stats)
{
    if (ix->last_nq == 0 || ix->last_k == 0) return fail(DR_E_ARG, "nothing to download");
    HIPCHK(hipSetDevice(ix->device));
    int rc = sync_locked(ix);
    if (rc) return rc;
    dr_index::BatchSet &bs = ix->sets[ix->last_set];
    const uint32_t nq = ix->last_nq, k = ix->last_k;
    // results go through a pinned host slab: four asynchronous copies and one wait, then plain memcpys into the caller's
    // (pageable) arrays -- a pageable destination makes every hipMemcpyAsync a blocking staged copy of its own
    static_assert(sizeof(dr_stats) == sizeof(KStats), "stats layout");
    const size_t b_ids = (size_t)nq * k * 4, b_cnt = (size_t)nq * 4, b_st = (size_t)nq * sizeof(KStats);
    const size_t need = 2 * b_ids + b_cnt + b_st;
    if (ix->pinned_bytes < need) {
        if (ix->pinned) (void)hipHostFree(ix->pinned);
        ix->pinned = nullptr; ix->pinned_bytes = 0;
        if (hipHostMalloc(&ix->pinned, need, hipHostMallocDefault) != hipSuccess) return fail(DR_E_NODEVICE, "hipHostMalloc(%zu) failed", need);
        ix->pinned_bytes = need;
    }
    unsigned char *hp = static_cast<unsigned char *>(ix->pinned);
    HIPCHK(hipEventRecord(ix->ev[4], ix->stream));
    if (out_ids) HIPCHK(hipMemcpyAsync(hp, bs.out_ids.p, b_ids, hipMemcpyDeviceToHost, ix->stream));
    if (out_dist) HIPCHK(hipMemcpyAsync(hp + b_ids, bs.out_dist.p, b_ids, hipMemcpyDeviceToHost, ix->stream));
    if (out_count) HIPCHK(hipMemcpyAsync(hp + 2 * b_ids, bs.out_count.p, b_cnt, hipMemcpyDeviceToHost, ix->stream));
    if (stats) HIPCHK(hipMemcpyAsync(hp + 2 * b_ids + b_cnt, bs.stats.p, b_st, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipEventRecord(ix->ev[5], ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    if (out_ids) memcpy(out_ids, hp, b_ids);
    if (out_dist) memcpy(out_dist, hp + b_ids, b_ids);
    if (out_count) memcpy(out_count, hp + 2 * b_ids, b_cnt);
    if (stats) memcpy(stats, hp + 2 * b_ids + b_cnt, b_st);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ix->ev[4], ix->ev[5]);
    ix->timing.d2h_ms = ms;
    ix->timing.total_ms = ix->timing.h2d_ms + ix->timing.lut_kernel_ms + ix->timing.search_kernel_ms + ix->timing.finalize_kernel_ms + ms;
    return 0;
}

extern "C" int dr_batch_sync(dr_index *ix)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    return sync_locked(ix);
}

// waits for everything queued on the handle (pipelined jobs included: their results reach the callers' buffers)
static int finish_job_locked(dr_index *ix, int j);
static int quiesce_locked(dr_index *ix)
{
    int rc_first = 0; std::string msg;
    for (int j = 0; j < DR_MAX_JOBS; j++)
        if (ix->jobs[j].active) { const int rc = finish_job_locked(ix, j); if (rc && !rc_first) { rc_first = rc; msg = g_err; } }
    for (hipStream_t st : { ix->up_stream, ix->stream, ix->stream2, ix->fstream, ix->down_stream }) if (st) HIPCHK(hipStreamSynchronize(st));
    ix->lane0_sig = 0;
    if (rc_first) { g_err = msg; return rc_first; }
    return 0;
}

extern "C" int dr_batch_select(dr_index *ix, uint32_t slot)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    if (slot >= DR_MAX_RESIDENT) return fail(DR_E_ARG, "resident batch %u out of range (%u slots)", slot, DR_MAX_RESIDENT);
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->cs = &ix->slots[slot];
    return 0;
}

extern "C" void *dr_host_alloc(uint64_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

extern "C" void dr_host_free(void *p) { if (p) (void)hipHostFree(p); }

#include "pipeline.inc"

extern "C" int dr_batch_upload(dr_index *ix, const float *queries, uint32_t nq)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    return upload_queries_locked(ix, queries, nq);
}

extern "C" int dr_batch_run(dr_index *ix, uint32_t k, uint32_t L, uint32_t beam_width, uint32_t mode,
                            uint32_t band_policy, uint32_t flags)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    return run_locked(ix, k, L, beam_width, mode, band_policy, flags);
}

extern "C" int dr_batch_download(dr_index *ix, uint32_t *out_ids, float *out_dist, uint32_t *out_count, dr_stats *stats)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    return download_locked(ix, out_ids, out_dist, out_count, stats);
}

extern "C" int dr_get_timing(dr_index *ix, dr_timing *out)
{
    if (!ix || !out) return fail(DR_E_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    *out = ix->timing;
    return 0;
}


static int seq_search_locked(dr_index *ix, const void *queries, bool f64, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width, uint32_t mode,
                             uint32_t band_policy, uint32_t *out_ids, void *out_dist, uint32_t *out_count, dr_stats *stats, uint32_t flags = 0, bool generic = false);

extern "C" int dr_search_batch(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t L,
                               uint32_t beam_width, uint32_t mode, uint32_t band_policy, uint32_t flags,
                               uint32_t *out_ids, float *out_dist, uint32_t *out_count, dr_stats *stats)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    if (!out_ids || !out_dist || !out_count) return fail(DR_E_ARG, "null output buffer");
    if (!queries || nq == 0) return fail(DR_E_ARG, "empty query batch");
    std::lock_guard<std::mutex> lk(ix->mu);
    // a dimension without compiled kernels (or DR_FORCE_GENERIC=1 on a built one: the test that holds the run-time tree to the compiled trees):
    // the generic traversal, one wavefront per query
    if (!ix->kern || getenv("DR_FORCE_GENERIC") != nullptr) {
        if (mode < DR_MODE_M1 || mode > DR_MODE_M4) return fail(DR_E_UNSUPPORTED, "mode %u: the generic traversal (D = %u) serves M1 ... M4", mode, ix->D);
        return seq_search_locked(ix, queries, false, nq, k, L, beam_width, mode, band_policy, out_ids, out_dist, out_count, stats, flags, true);
    }
    if ((band_policy & 0xFFu) == 2u) {
        // the reference's coin flip itself (np.random.random() < 0.2 on numpy's MT19937 stream): a sequential walk, served by the literal kernel
        if (mode != DR_MODE_M1) return fail(DR_E_ARG, "band policy 2 (the literal coin flip) belongs to DR_MODE_M1");
        return seq_search_locked(ix, queries, false, nq, k, L, beam_width, mode, band_policy, out_ids, out_dist, out_count, stats);
    }
    const bool no_direct = getenv("DR_NO_DIRECT") != nullptr;      // A/B and tests: small calls through the general path (read per call)
    // (round 5: EVERY blocking call of up to one chunk takes the direct path -- results written straight into the page-locked slab, no download
    //  copies; DR_DIRECT_MAX=256 restores the old limit for A/B. Above a handful of queries some query ties almost surely: the tie-order pass is
    //  queued right behind the search instead of after a look at the flag word)
    uint32_t direct_max = DR_MAX_CHUNK;
    if (const char *edm = getenv("DR_DIRECT_MAX")) direct_max = (uint32_t)atoi(edm);
    if (nq <= direct_max && !no_direct) {
        // ---- a handful of queries (one per request is the shape of the reference's API routes, search_engine.py:530-614, app.py:84-130)
        HIPCHK(hipSetDevice(ix->device));
        {   // (pipelined jobs of other threads, or resident steps not waited for yet, own output sets and streams: finished first)
            bool busy = false;
            for (int j = 0; j < DR_MAX_JOBS; j++) busy = busy || ix->jobs[j].active;
            for (const auto &x : ix->sets) busy = busy || x.fin_pending;
            if (busy) { const int rcq = quiesce_locked(ix); if (rcq) return rcq; const int rcs = sync_locked(ix); if (rcs) return rcs; }
        }
        int rc = upload_queries_locked(ix, queries, nq, false, false);
        ix->direct = true;
        if (!rc) rc = run_locked(ix, k, L, beam_width, mode, band_policy, flags);
        ix->direct = false;
        if (rc) {
            // (the batch's copy out of the page-locked staging buffer may still be in flight: the next call restages into it)
            const std::string keep = g_err;
            (void)hipStreamSynchronize(ix->stream); (void)hipGetLastError();
            ix->h2d_pending = false;
            g_err = keep;
            return rc;
        }
        if (!ix->direct_used) {       // (the call that measures the index's regime went the general way)
            return download_locked(ix, out_ids, out_dist, out_count, stats);
        }
        const bool eager_fin = nq > DR_DIRECT_MAX && ix->direct_fin;
        bool fin_timed = false;
        if (eager_fin) {
            dr_index::BatchSet &bs = ix->sets[ix->last_set];
            const unsigned fgrid = (unsigned)std::min<uint64_t>(((uint64_t)nq + 3) / 4, (uint64_t)ix->num_cu * 8);
            HIPCHK(hipEventRecord(bs.fin_start, ix->stream));
            hipLaunchKernelGGL(finalize_kernel, dim3(fgrid), dim3(256), 4 * ((size_t)ix->direct_f->cap + 2 + 64) * 8, ix->stream, *ix->direct_f);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(bs.fin_done, ix->stream));
            HIPCHK(hipMemsetAsync(bs.counter.p + 1, 0, 4, ix->stream));
            fin_timed = true;
        }
        HIPCHK(hipStreamSynchronize(ix->stream));
        const size_t b_ids = (size_t)nq * k * 4, b_cnt = (size_t)nq * 4, b_st = (size_t)nq * sizeof(KStats);
        unsigned char *hp = static_cast<unsigned char *>(ix->pinned);
        if (ix->timing.variant == 18u) {
            // a query outgrew the visited-id set of the workgroup-per-query kernel (status bit 0): the whole call again through search_kernel.hpp
            const KStats *hs = reinterpret_cast<const KStats *>(hp + 2 * b_ids + b_cnt);
            bool over = false;
            uint64_t evald = 0, all = 0;
            for (uint32_t i = 0; i < nq; i++) { over = over || (hs[i].status & 1u) != 0u; evald += hs[i].pq_evaluated; all += hs[i].pq; }
            if (mode == DR_MODE_M1 && !over && all > 0) {
                int &lv = ix->lat_adc_live[ix->lat_sc];
                if (lv != 0) lv = (20 * evald > all) ? 1 : 0;           // (eager rows: asked on < 5 % of the neighbours -> lazy from now on)
                else if (5 * evald > all) lv = 1;                       // (lazy rows: back to eager when a fifth of them ask)
            }
            if (over) {
                dr_index::BatchSet &bs = ix->sets[ix->last_set];
                HIPCHK(hipMemsetAsync(bs.counter.p + 1, 0, 4, ix->stream));
                ix->direct = true; ix->lat_skip = true;
                rc = run_locked(ix, k, L, beam_width, mode, band_policy, flags);
                ix->direct = false; ix->lat_skip = false;
                if (rc) return rc;
                if (!ix->direct_used) return download_locked(ix, out_ids, out_dist, out_count, stats);
                if (eager_fin && ix->direct_fin) {      // (the tie-order pass queued above replayed the abandoned run's log: again, for this one)
                    const unsigned fgrid = (unsigned)std::min<uint64_t>(((uint64_t)nq + 3) / 4, (uint64_t)ix->num_cu * 8);
                    hipLaunchKernelGGL(finalize_kernel, dim3(fgrid), dim3(256), 4 * ((size_t)ix->direct_f->cap + 2 + 64) * 8, ix->stream, *ix->direct_f);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipMemsetAsync(ix->sets[ix->last_set].counter.p + 1, 0, 4, ix->stream));
                }
                HIPCHK(hipStreamSynchronize(ix->stream));
                hp = static_cast<unsigned char *>(ix->pinned);
            }
        }
        const uint32_t tied = *reinterpret_cast<const uint32_t *>(hp + 2 * b_ids + b_cnt + b_st);
        if (tied && !eager_fin) {
            dr_index::BatchSet &bs = ix->sets[ix->last_set];
            if (ix->direct_fin) hipLaunchKernelGGL(finalize_kernel, dim3((nq + 3) / 4), dim3(256), 4 * ((size_t)ix->direct_f->cap + 2 + 64) * 8, ix->stream, *ix->direct_f);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemsetAsync(bs.counter.p + 1, 0, 4, ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
        }
        harvest_kernel_times(ix, true);
        if (ix->h2d_pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ix->ev[0], ix->ev[1]) == hipSuccess) ix->timing.h2d_ms = ms;
            (void)hipGetLastError();
            ix->h2d_pending = false;
        }
        memcpy(out_ids, hp, b_ids);
        memcpy(out_dist, hp + b_ids, b_ids);
        memcpy(out_count, hp + 2 * b_ids, b_cnt);
        if (stats) memcpy(stats, hp + 2 * b_ids + b_cnt, b_st);
        // the pieces of a blocking call (dr_get_timing): the batch's copy, the search kernel (+ table kernel), the tie-order pass queued behind it; the
        // results were written into the page-locked slab by the kernels themselves (no download copies)
        ix->timing.d2h_ms = 0.0f; ix->timing.finalize_kernel_ms = 0.0f;
        if (fin_timed) {
            float fms = 0;
            const dr_index::BatchSet &bsf = ix->sets[ix->last_set];
            if (hipEventElapsedTime(&fms, bsf.fin_start, bsf.fin_done) == hipSuccess) ix->timing.finalize_kernel_ms = fms;
            (void)hipGetLastError();
        }
        ix->timing.total_ms = ix->timing.h2d_ms + ix->timing.lut_kernel_ms + ix->timing.search_kernel_ms + ix->timing.finalize_kernel_ms;
        return 0;
    }
    float h2d = 0, ker = 0, fin = 0, d2h = 0;
    for (uint32_t q0 = 0; q0 < nq; q0 += DR_MAX_CHUNK) {
        const uint32_t n = std::min(DR_MAX_CHUNK, nq - q0);
        int rc = upload_queries_locked(ix, queries + (size_t)q0 * ix->D, n, false);
        if (!rc) rc = run_locked(ix, k, L, beam_width, mode, band_policy, flags);
        if (!rc) rc = download_locked(ix, out_ids + (size_t)q0 * k, out_dist + (size_t)q0 * k, out_count + q0,
                                      stats ? stats + q0 : nullptr);
        if (rc) {
            const std::string keep = g_err;
            (void)hipStreamSynchronize(ix->stream); (void)hipGetLastError();      // (the staging buffer's copy: as above)
            ix->h2d_pending = false;
            g_err = keep;
            return rc;
        }
        h2d += ix->timing.h2d_ms; ker += ix->timing.search_kernel_ms; fin += ix->timing.finalize_kernel_ms; d2h += ix->timing.d2h_ms;
    }
    ix->timing.h2d_ms = h2d; ix->timing.search_kernel_ms = ker; ix->timing.finalize_kernel_ms = fin; ix->timing.d2h_ms = d2h;
    ix->timing.total_ms = h2d + ker + fin + d2h;
    return 0;
}

// The literal sequential kernel (search_f64.hpp): float64 queries (the CLI hands np.array(list): diskrag.py:194, quirk Q8) -- M1 and M2, the two
// searches the CLI reaches (search_engine.py:566-573) -- and, since round 5, float32 queries with the LITERAL coin flip of the rerank policy
// (band_policy 2 | seed0 << 8: numpy's MT19937 stream, seeded with seed0 + query index). One wavefront per query, at most 64 queries per launch.
// Round 6: `generic` = the D = 0 instantiation (the dimension a run-time parameter, numpy's tree walked at run time, M3 and M4 as well): what an
// index without compiled kernels is searched with.
static int seq_search_locked(dr_index *ix, const void *queries, bool f64, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width, uint32_t mode,
                             uint32_t band_policy, uint32_t *out_ids, void *out_dist, uint32_t *out_count, dr_stats *stats, uint32_t flags, bool generic)
{
    generic = generic || !ix->kern;
    const char *what = generic ? "the generic traversal" : f64 ? "float64 search" : "band policy 2 (the literal coin flip)";
    const bool m12 = mode == DR_MODE_M1 || mode == DR_MODE_M2;
    if (!m12 && !(generic && !f64 && (mode == DR_MODE_M3 || mode == DR_MODE_M4))) return fail(DR_E_UNSUPPORTED, "%s: modes M1 and M2 only", what);
    if (k == 0) return fail(DR_E_ARG, "k must be positive");
    if ((band_policy & 0xFFu) > 2u) return fail(DR_E_ARG, "band policy %u", band_policy & 0xFFu);
    if (generic && (flags & ~(DR_F_USE_PQ | DR_F_SQDIST))) return fail(DR_E_UNSUPPORTED, "%s takes DR_F_USE_PQ (M3) and DR_F_SQDIST (M4) only (flags %u)", what, flags);
    const bool adc_only = mode == DR_MODE_M3 && (flags & DR_F_USE_PQ);
    if (!adc_only) { const int rcv = need_vectors(ix, what); if (rcv) return rcv; }
    if ((mode == DR_MODE_M1 || adc_only) && ix->m == 0) return fail(DR_E_NOPQ, "mode %u needs PQ data (dr_index_set_pq)", mode);
    // result-heap capacity: M1 / M4 L (search_engine.py:468-474, vamana_graph.py:634-638), M2 beam_width (:746-750), M3 k (:586-590)
    const uint32_t cap = (mode == DR_MODE_M2) ? beam_width : (mode == DR_MODE_M3) ? k : L;
    if (cap == 0) return fail(DR_E_ARG, "result-list capacity is zero (L / beam_width)");
    if (cap > 512) return fail(DR_E_UNSUPPORTED, "result-list capacity %u > 512", cap);
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }
    const uint32_t D = ix->D, CH = 64;
    const size_t esz = f64 ? 8 : 4;
    const bool pq = (mode == DR_MODE_M1) || adc_only;        // (the table sits in LDS)
    // LDS: 2 queries, both heaps, per-expansion arrays, the table, the generator's state; the candidates heap takes what is left
    const size_t fixed = (size_t)2 * D * esz + (size_t)(cap + 2) * (esz + 4) + 64 * esz + 64 * 4 + 64 * 4 + (pq ? (size_t)ix->m * 256 * 4 : 0) + 626 * 4 + 64;
    if (fixed + 1024 * (esz + 4) > 160 * 1024) return fail(DR_E_UNSUPPORTED, "%s does not fit in LDS (D=%u, m=%u, capacity %u)", what, D, ix->m, cap);
    const uint32_t cand_cap = (uint32_t)std::min<size_t>((160 * 1024 - fixed) / (esz + 4), 8192) & ~1u;
    const size_t lds = fixed + (size_t)cand_cap * (esz + 4);
    const void *kfn = generic ? (f64 ? reinterpret_cast<const void *>(&search_seq_kernel<0, double>) : reinterpret_cast<const void *>(&search_seq_kernel<0, float>))
                              : f64 ? ix->kern->search_f64 : ix->kern->search_seq_f32;
    HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t vis_words = (uint32_t)((ix->N + 31) / 32);
    // scratch kept on the handle (the CLI asks one query per call: no allocation on its path after the first)
    DevBuf<double> &dq = ix->f64_q, &dd = ix->f64_dist; DevBuf<uint32_t> &dids = ix->f64_ids, &dcnt = ix->f64_cnt, &dvis = ix->f64_vis;
    DevBuf<KStats> &dst = ix->f64_stats;
    const uint32_t chq = std::min<uint32_t>(CH, nq);
    if (dq.reserve((size_t)chq * D) || dd.reserve((size_t)chq * k) || dids.reserve((size_t)chq * k) || dcnt.reserve(chq) ||
        dst.reserve(chq) || dvis.reserve((size_t)chq * vis_words)) return DR_E_NODEVICE;
    int rc = 0;
    for (uint32_t q0 = 0; q0 < nq && !rc; q0 += CH) {
        const uint32_t n = std::min(CH, nq - q0);
        F64Params p;
        memset(&p, 0, sizeof p);
        p.vecp = ix->vecp.p; p.adj = ix->adj.p; p.first = ix->first.p; p.deg = nullptr; p.codes = ix->codes.p;
        p.codebook = ix->codebook.p; p.perm = ix->perm.p; p.queries = dq.p;
        p.N = ix->N; p.D = D; p.R = ix->R; p.m = ix->m; p.sd = ix->sd; p.medoid = ix->medoid; p.nq = n;
        p.mode = mode; p.k = k; p.cap = cap; p.L = L; p.bw = beam_width; p.policy = band_policy; p.q0 = q0;
        p.max_steps = mode == DR_MODE_M1 ? (uint32_t)std::min<uint64_t>((uint64_t)L * 10, ix->N) : 0xFFFFFFFFu;
        p.flags = flags; p.chain_major = (generic && ix->kern) ? 1u : 0u;
        p.vis = dvis.p; p.vis_words = vis_words; p.cand_cap = cand_cap;
        p.out_ids = dids.p; p.out_dist = dd.p; p.out_count = dcnt.p; p.stats = dst.p;
        if (hipMemcpyAsync(dq.p, static_cast<const unsigned char *>(queries) + (size_t)q0 * D * esz, (size_t)n * D * esz, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
            hipMemsetAsync(dvis.p, 0, (size_t)n * vis_words * 4, ix->stream) != hipSuccess) { rc = fail(DR_E_NODEVICE, "%s: upload failed", what); break; }
        void *args[] = { &p };
        if (hipLaunchKernel(kfn, dim3(n), dim3(64), args, lds, ix->stream) != hipSuccess) { rc = fail(DR_E_NODEVICE, "%s: launch failed: %s", what, hipGetErrorString(hipGetLastError())); break; }
        // results through the pinned host slab (see download_locked)
        const size_t b_ids = (size_t)n * k * 4, b_d = (size_t)n * k * esz, b_cnt = (size_t)n * 4, b_st = (size_t)n * sizeof(KStats);
        const size_t need = b_d + b_ids + b_cnt + b_st + 16;
        if (ix->pinned_bytes < need) {
            if (ix->pinned) (void)hipHostFree(ix->pinned);
            ix->pinned = nullptr; ix->pinned_bytes = 0;
            if (hipHostMalloc(&ix->pinned, need, hipHostMallocDefault) != hipSuccess) { rc = fail(DR_E_NODEVICE, "hipHostMalloc(%zu) failed", need); break; }
            ix->pinned_bytes = need;
        }
        unsigned char *hp = static_cast<unsigned char *>(ix->pinned);
        if (hipMemcpyAsync(hp, dd.p, b_d, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
            hipMemcpyAsync(hp + b_d, dids.p, b_ids, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
            hipMemcpyAsync(hp + b_d + b_ids, dcnt.p, b_cnt, hipMemcpyDeviceToHost, ix->stream) != hipSuccess ||
            (stats && hipMemcpyAsync(hp + b_d + b_ids + b_cnt, dst.p, b_st, hipMemcpyDeviceToHost, ix->stream) != hipSuccess) ||
            hipStreamSynchronize(ix->stream) != hipSuccess) { rc = fail(DR_E_NODEVICE, "%s: %s", what, hipGetErrorString(hipGetLastError())); break; }
        memcpy(static_cast<unsigned char *>(out_dist) + (size_t)q0 * k * esz, hp, b_d);
        memcpy(out_ids + (size_t)q0 * k, hp + b_d, b_ids);
        memcpy(out_count + q0, hp + b_d + b_ids, b_cnt);
        if (stats) memcpy(stats + q0, hp + b_d + b_ids + b_cnt, b_st);
    }
    return rc;
}

extern "C" int dr_search_batch_f64(dr_index *ix, const double *queries, uint32_t nq, uint32_t k, uint32_t L,
                                   uint32_t beam_width, uint32_t mode, uint32_t band_policy, uint32_t flags,
                                   uint32_t *out_ids, double *out_dist, uint32_t *out_count, dr_stats *stats)
{
    (void)flags;
    if (!ix) return fail(DR_E_ARG, "null index");
    if (!out_ids || !out_dist || !out_count) return fail(DR_E_ARG, "null output buffer");
    if (!queries || nq == 0) return fail(DR_E_ARG, "empty query batch");
    std::lock_guard<std::mutex> lk(ix->mu);
    return seq_search_locked(ix, queries, true, nq, k, L, beam_width, mode, band_policy, out_ids, out_dist, out_count, stats, 0u,
                             !ix->kern || getenv("DR_FORCE_GENERIC") != nullptr);
}

#include "entry_points.inc"

#include "builder.inc"

#include "pq_build.inc"

// C8: the reference's scalar distance kernels on row pairs (no index needed).
extern "C" int dr_scalar_kernels(int device, const float *x, const float *y, uint32_t n, uint32_t D, float *out_l2, float *out_cos)
{
    if (!x || !y || n == 0 || D == 0 || (!out_l2 && !out_cos)) return fail(DR_E_ARG, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DR_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(DR_E_ARG, "device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    DevBuf<float> dx, dy, o1, o2;
    if (dx.reserve((size_t)n * D) || dy.reserve((size_t)n * D) || o1.reserve(n) || o2.reserve(n)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpy(dx.p, x, (size_t)n * D * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dy.p, y, (size_t)n * D * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scalar_pairs_kernel, dim3(std::min<uint32_t>(n, 65535)), dim3(64), 0, nullptr, dx.p, dy.p, n, D, o1.p, o2.p);
    HIPCHK(hipGetLastError());
    if (out_l2) HIPCHK(hipMemcpy(out_l2, o1.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (out_cos) HIPCHK(hipMemcpy(out_cos, o2.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

// Inline neighbour codes on / off (off by default: N*R*m bytes of HBM). The block is (re)built on the device before the
// next search that evaluates ADC sums; results never depend on it.
extern "C" int dr_index_inline_codes(dr_index *ix, int enable)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }
    ix->inline_codes = enable != 0;
    if (!ix->inline_codes) { ix->nbcodes.release(); ix->nbcodes_valid = false; }
    return 0;
}

extern "C" int dr_debug_force_kind(dr_index *ix, int kind, int *out_adc_live)
{
    if (kind != 18 && (kind > DR_MAX_KIND_ID || (kind >= 0 && dr_kind_pos(kind) < 0))) return fail(DR_E_ARG, "unknown kernel variant %d", kind);      // (18: latency_kernel.hpp)
    g_force_kind = kind < 0 ? -1 : kind;
    g_force_kind_set = true;
    if (ix && out_adc_live) { std::lock_guard<std::mutex> lk(ix->mu); *out_adc_live = ix->adc_live; }
    return 0;
}

// Diagnostic: per-phase shader-clock sums of the last search (library built with -DDR_PHASE_TIMING only).
extern "C" int dr_debug_phase_cycles(dr_index *ix, double *out8)
{
    if (!ix || !out8) return fail(DR_E_ARG, "null argument");
#ifdef DR_TRACE_VIS
    // diagnostic build: out8 is really a u32[256][16384] trace buffer (scripts/exp_vis_trace.py)
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!ix->phase.p) return fail(DR_E_ARG, "no traced search yet");
    HIPCHK(hipMemcpy(out8, ix->phase.p, (size_t)256 * 8192 * 8, hipMemcpyDeviceToHost));
    return 0;
#elif defined(DR_PHASE_TIMING)
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!ix->phase.p || ix->cs->nq == 0) return fail(DR_E_ARG, "no timed search yet");
    std::vector<u64> h((size_t)ix->cs->nq * 8);
    HIPCHK(hipMemcpy(h.data(), ix->phase.p, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; i++) out8[i] = 0;
    for (size_t q = 0; q < ix->cs->nq; q++) for (int i = 0; i < 8; i++) out8[i] += (double)h[q * 8 + i];
    return 0;
#else
    return fail(DR_E_UNSUPPORTED, "library was not built with -DDR_PHASE_TIMING");
#endif
}
#include "comm.inc"
