// engine.hip -- host side of libdiskrag_hip.so: the C ABI declared in include/diskrag_hip.h.
//
// Index data lives in HBM for the life of the handle:
//   vecp   [N][D] f32, chain-major tiles (numerics.hpp)        from index.dat records (T1)
//   adj    [N][R] u32 + first-occurrence masks [N][ceil(R/64)] from index.dat records (T1)
//   codes  [N][m] u8, codebook [m][256][D/m] f32               pq_codes.bin (T2), cluster centres (T3)
// Scratch (per handle, grown on demand): device query buffers, per-workgroup visited tables, per-query result
// lists, insert logs, stats; one HIP stream per handle; calls on a handle are serialised by a mutex.
#include <hip/hip_runtime.h>

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
            else { e = hipMemset(p, 0, want * sizeof(T)); if (e != hipSuccess) return fail(DR_E_NODEVICE, "hipMemset failed"); }
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
static_assert(DR_MAX_TICKETS == 32u, "header and engine agree on the tickets in flight");
#define DR_MAX_JOBS 32      // dr_search_submit tickets in flight (round 4: small submits are coalesced, several jobs ride in one launch)
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
    int set = -1;                       // the BatchSet holding its outputs once launched
    void *pin_out = nullptr; size_t pin_out_bytes = 0;
    hipEvent_t up_done = nullptr, down_done = nullptr;
};

// Persistent work areas of dr_sharded_submit (comm.inc), owned by the first shard of the call: staging arrays, events, the
// exchange stream and a page-locked slab for the results live across calls. Two of them: batch i+1 is searched while batch i
// is exchanged, merged and downloaded.
#define DR_SHARD_DEPTH 2
struct ShardWork {
    DevBuf<uint32_t> loc_ids, fin_ids, status;
    DevBuf<float> loc_dist, fin_dist, q;
    DevBuf<u64> send_keys, all_keys, words;     // my list + status word; every rank's; the ranks' status words
    hipEvent_t e[5] = {};                       // start, local lists ready, exchange done, final merge done, download done
    hipEvent_t up = nullptr;                    // batch on the device
    void *pin = nullptr; size_t pin_bytes = 0;
    bool active = false;
    uint64_t ticket = 0;
    uint32_t nq = 0, k = 0; int nranks = 1;
    int local_rc = 0; std::string local_msg;
    uint32_t *out_ids = nullptr; float *out_dist = nullptr; uint32_t *out_status = nullptr; float *out_ms = nullptr;
    ~ShardWork()
    {
        for (auto &x : e) if (x) (void)hipEventDestroy(x);
        if (up) (void)hipEventDestroy(up);
        if (pin) (void)hipHostFree(pin);
    }
};
struct ShardScratch {
    std::mutex mu;
    ShardWork w[DR_SHARD_DEPTH];
    uint64_t next_ticket = 1;
    std::map<uint64_t, std::pair<int, std::string>> failed;   // tickets finished by a later submit with an error
    std::vector<hipEvent_t> shard_done;   // one per local shard
    hipStream_t xs = nullptr;             // merge / exchange / download stream (without a communicator)
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
    std::map<uint64_t, std::pair<int, std::string>> failed_tickets;   // tickets whose launch failed, until their dr_search_wait collects the error
    uint32_t coalesce_cap = 10240; // queries a group of small submits may grow to (dr_set_coalesce; 0: every submit is its own launch)
    bool hold_always = false;     // dr_debug_hold: submits are only launched when full / flushed / waited for (tests)
    uint64_t pipe_launches = 0, pipe_tickets = 0, pipe_max_tickets = 0, pipe_queries = 0;   // dr_pipeline_stats
    hipStream_t up_stream = nullptr, down_stream = nullptr;
    uint32_t last_nq = 0;         // batch size of the last launch (dr_batch_download)
    DevBuf<uint32_t> vis, vis_epoch;   // visited words [slots][vis_words] + the per-slot query stamp (search_kernel.hpp)
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
    bool h2d_pending = false;
    DevBuf<double> f64_q, f64_dist;                 // dr_search_batch_f64 scratch
    DevBuf<uint32_t> f64_ids, f64_cnt, f64_vis;
    DevBuf<KStats> f64_stats;
    uint32_t fin_hint = 0;        // tie-list length to size the tie-order launches for (0: not known yet -> full grid)
    int adc_live = -1;            // M1 on this index: does the rerank policy A4 really consult the ADC? -1 = not measured yet
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
    ix->kern = dr_dim_kernels((int)D);
    if (!ix->kern) return fail(DR_E_UNSUPPORTED, "unsupported vector dimension %u (built: 32,64,96,128,256,768,960,1536)", D);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(DR_E_NODEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(DR_E_ARG, "device %d out of range (%d devices)", device, ndev);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    ix->num_cu = prop.multiProcessorCount;
    ix->device = device; ix->N = N; ix->D = D; ix->R = R; ix->medoid = medoid;
    HIPCHK(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ix->fstream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ix->up_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ix->down_stream, hipStreamNonBlocking));
    for (auto &gr : ix->groups) { HIPCHK(hipEventCreateWithFlags(&gr.up_done, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&gr.down_done, hipEventDisableTiming)); }
    for (auto &e : ix->ev) HIPCHK(hipEventCreate(&e));
    for (auto &pr : ix->kev) { HIPCHK(hipEventCreate(&pr[0])); HIPCHK(hipEventCreate(&pr[1])); HIPCHK(hipEventCreate(&pr[2])); }
    for (auto &bs : ix->sets) { HIPCHK(hipEventCreate(&bs.search_done)); HIPCHK(hipEventCreate(&bs.fin_start)); HIPCHK(hipEventCreate(&bs.fin_done)); }
    ix->h_perm.resize(D);
    pw_build_perm_rec(0, D, ix->h_perm.data());
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
    ix->adc_live = -1; ix->nbcodes_valid = false;
    return build_first_masks(ix);
}

extern "C" int dr_index_set_pq(dr_index *ix, const float *codebook, const uint8_t *codes, uint32_t m)
{
    if (!ix || !codebook || !codes) return fail(DR_E_ARG, "null argument");
    if (m == 0 || ix->D % m) return fail(DR_E_ARG, "n_subvectors %u must divide D=%u", m, ix->D);
    if (ix->D / m > 128) return fail(DR_E_UNSUPPORTED, "sub_dim %u > 128", ix->D / m);
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }     // queued searches still read the old codes
    if (ix->codes.reserve((size_t)ix->N * m)) return DR_E_NODEVICE;
    if (ix->codebook.reserve((size_t)256 * ix->D)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpy(ix->codes.p, codes, (size_t)ix->N * m, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->codebook.p, codebook, (size_t)256 * ix->D * 4, hipMemcpyHostToDevice));
    ix->m = m; ix->sd = ix->D / m; ix->codebook_gen++;
    for (auto &qs : ix->slots) qs.pq_ub_valid = false;
    ix->adc_live = -1; ix->nbcodes_valid = false;
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
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    for (hipStream_t st : { ix->up_stream, ix->stream, ix->fstream, ix->down_stream }) if (st) (void)hipStreamSynchronize(st);
    ix->lut.release();
    ix->vecp.release(); ix->adj.release(); ix->first.release(); ix->codes.release(); ix->codebook.release(); ix->nbcodes.release(); ix->sdc.release();
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

static int upload_queries_locked(dr_index *ix, const float *queries, uint32_t nq, bool wait = true)
{
    if (!queries || nq == 0) return fail(DR_E_ARG, "empty query batch");
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipEventRecord(ix->ev[0], ix->stream));
    const int rc = upload_slot_async(ix, *ix->cs, queries, nq, ix->stream);
    if (rc) return rc;
    HIPCHK(hipEventRecord(ix->ev[1], ix->stream));
    // dr_search_batch does not wait here: what consumes the queries is queued behind them on the same stream and the call
    // only returns after its download; the copy's duration is read at the next sync. An explicit dr_batch_upload waits,
    // so that the caller's buffer is free on return whatever kind of host memory it is.
    if (wait) HIPCHK(hipStreamSynchronize(ix->stream));
    ix->h2d_pending = true;
    // byte queries? (only asked when byte rows exist)
    ix->cs->q_u8 = (ix->vec8_state == 1 || (ix->vec8_state == 0 && ix->D == 128)) && queries_are_u8(queries, (size_t)nq * ix->D);
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
        const uint64_t S = 64;
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
    if (fn && !old_form) {
        if (qs.pq_max.reserve((size_t)room * ix->m)) return DR_E_NODEVICE;
        float *mxp = qs.pq_max.p + (size_t)q0 * ix->m;
        void *args[] = { &cbp, &qp, &nqv, &Dv, &mxp };
        HIPCHK(hipLaunchKernel(fn, dim3((nq + 63) / 64, ix->m), dim3(64), args, 0, st));
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

static int finish_group_locked(dr_index *ix, int g);
// Per-query scratch (insert log, result keys) is sized by the batch: very large batches are processed in chunks.
static const uint32_t DR_MAX_CHUNK = 32768;

static int run_locked(dr_index *ix, uint32_t k, uint32_t L, uint32_t bw, uint32_t mode, uint32_t policy, uint32_t flags,
                      const BuildOverride *ov = nullptr)
{
    if (ov) ix->cs->nq = ov->nq;
    if (ix->cs->nq == 0) return fail(DR_E_ARG, "no queries uploaded");
    if (ix->cs->nq > 65536 && !ov) return fail(DR_E_UNSUPPORTED, "resident batches are limited to 65536 queries (dr_search_batch chunks larger ones)");
    if (mode < DR_MODE_M1 || mode > DR_MODE_PQ) return fail(DR_E_ARG, "unknown mode %u", mode);
    if (k == 0) return fail(DR_E_ARG, "k must be positive");
    const bool pq_only = (mode == DR_MODE_M3 && (flags & DR_F_USE_PQ)) || mode == DR_MODE_PQ || (ov && ov->sdc);   // ADC-only traversals
    const bool rerank = (mode == DR_MODE_PQ) && (flags & DR_F_RERANK);
    const bool use_pq = (mode == DR_MODE_M1) || pq_only;
    if (use_pq && ix->m == 0) return fail(DR_E_NOPQ, "mode %u needs PQ data (dr_index_set_pq)", mode);
    if (!pq_only || rerank) { const int rcv = need_vectors(ix, rerank ? "DR_F_RERANK" : "this search mode"); if (rcv) return rcv; }
    // result-list capacity: M1/M4 (and the engine's PQ mode) L, M2 beam_width, M3 k (search_engine.py:468-474;
    // vamana_graph.py:746-750, :586-590)
    const uint32_t cap = (mode == DR_MODE_M2) ? bw : (mode == DR_MODE_M3) ? k : L;
    if (cap == 0) return fail(DR_E_ARG, "result-list capacity is zero (L / beam_width / k)");
    if (cap > DR_MAX_CAPACITY) return fail(DR_E_UNSUPPORTED, "result-list capacity %u > %u", cap, DR_MAX_CAPACITY);
    HIPCHK(hipSetDevice(ix->device));
    // ONE search stream per handle. (Round 3 tried a second one for small pipelined batches, with its own visited-set scratch:
    // slower -- profiles/r03/ab/ab_small_batches_two_search_lanes.json, and the kernel trace of it in profiles/r04/ -- and removed;
    // small submits are coalesced into one launch instead, dr_search_submit.)
    hipStream_t st = ix->stream;
    DevBuf<uint32_t> &vis = ix->vis, &vis_epoch = ix->vis_epoch;

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
    const int *pref = k_m1 ? PREF_M1 : (ov && ov->sdc) ? (no_treg ? PREF_BUILD_PQ_NOTREG : PREF_BUILD_PQ) : k_adc ? (no_treg ? PREF_ADC_NOTREG : PREF_ADC) : ov ? PREF_BUILD : PREF_EX;
    const int npref = k_m1 ? 5 : (ov && ov->sdc) ? 2 : k_adc ? 3 : ov ? 2 : 4;
    // (round 4: the per-query table wins with as few as five or six wavefronts per CU -- c4 shape, lists of 300-500 entries: 1.38x over
    // the shared codebook at eight, profiles/r04/ab/ab_c4_long_lists_table_vs_codebook.jsonl; it used to need eight to be preferred)
    if (k_m1 && !ov && ix->adc_live == 1) pref = (lds_of(0) * 5 <= 160 * 1024) ? PREF_M1_LIVE_LUT : PREF_M1_LIVE_CB;
    int kind = -1;
    for (int i = 0; i < npref && kind < 0; i++) if (usable(pref[i])) kind = pref[i];
    {
        static bool env_read = false;
        if (!env_read) { const char *e = getenv("DR_FORCE_KIND"); if (e && !g_force_kind_set) g_force_kind = atoi(e); env_read = true; }
        const int g = g_force_kind;
        if (g >= 0 && g <= DR_MAX_KIND_ID && usable(g) && !(ov && ov->sdc)) {
            const bool g_m1 = (g == 0 || g == 3 || g == 9 || g == 11 || g == 13 || g == 16 || g == 17), g_adc = (g == 2 || g == 5 || g == 15), g_ex = (g == 1 || g == 8 || g == 12 || g == 14);
            if ((g_m1 && k_m1) || (g_adc && k_adc) || (g_ex && !k_m1 && !k_adc)) kind = g;
        }
    }
    if (kind < 0) return fail(DR_E_UNSUPPORTED, "no kernel variant fits in LDS (D=%u, m=%u, capacity %u)", ix->D, ix->m, cap);
    // A batch smaller than the chip's wavefront slots in 16-wavefront workgroups would fill ceil(nq / 16) CUs and leave the
    // rest idle: the same kernel in 4-wavefront workgroups spreads it over all of them (variants.hpp 16 / 17).
    {
        static const bool no_small = getenv("DR_NO_SMALL_WG") != nullptr;        // A/B
        const int tw = dr_small_twin(kind);
        if (!no_small && !ov && tw >= 0 && kind != g_force_kind && usable(tw) && (uint64_t)ix->cs->nq < (uint64_t)ix->num_cu * 16) kind = tw;
    }
    const KindDesc &kd_desc = DR_KINDS[dr_kind_pos(kind)];
    const void *kfn = ix->kern->search[dr_kind_pos(kind)][sc];
    const int NW = kd_desc.nw;
    const size_t lds = lds_of(kind);
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
    const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)nq + NW - 1) / NW, (uint64_t)occ * ix->num_cu);
    const uint32_t slots = grid * NW;

    // M1 is capped at min(10L, N) expansions (search_engine.py:429); the other variants are bounded by N.
    const bool capped = (mode == DR_MODE_M1 || mode == DR_MODE_PQ);
    uint64_t max_steps = capped ? std::min<uint64_t>((uint64_t)L * 10, ix->N) : 0xFFFFFFFFull;
    // visited set: per wavefront slot one word per 24 bit positions (+ an 8-bit query stamp: nothing is cleared
    // between queries, search_kernel.hpp) and the slot's stamp counter
    const uint32_t vis_words = (uint32_t)(((ix->N + 23) / 24 + 3) & ~3ull);
    // DR_F_NO_VISITED_SET (DR_MODE_PQ): the traversal keeps NO visited set (SearchParams::novis) -- no visited words (20 MB per
    // wavefront slot on a 1.25e8-point shard), no bit-position twin of the adjacency; same results, more evaluations.
    if ((flags & DR_F_NO_VISITED_SET) && mode != DR_MODE_PQ) return fail(DR_E_ARG, "DR_F_NO_VISITED_SET goes with DR_MODE_PQ");
    const bool novis = !ov && mode == DR_MODE_PQ && (flags & DR_F_NO_VISITED_SET) != 0;
    if (!novis && ((size_t)slots * vis_words > vis.n || slots > vis_epoch.n)) {
        // (re)allocation: fresh words and stamps -- queued launches still use the old buffers
        if (vis.p) HIPCHK(hipStreamSynchronize(st));
        if (vis.reserve((size_t)slots * vis_words) || vis_epoch.reserve(slots)) return DR_E_NODEVICE;
        HIPCHK(hipMemsetAsync(vis.p, 0, vis.n * 4, st));
        HIPCHK(hipMemsetAsync(vis_epoch.p, 0, vis_epoch.n * 4, st));
    }
    // accepted-insert log per query (tie replay): 4096 entries cover L <= 256 with room to spare, deeper lists get more
    const uint32_t logcap = std::max<uint32_t>(4096, 16 * cap);
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
    if (!ov && !novis && !ix->adjr_valid) { const int rcb = build_bit_order(ix); if (rcb) return rcb; }
    if (!ov && ix->inline_codes && !ix->nbcodes_valid && ix->codes.p && (mode == DR_MODE_M1 || pq_only)) { const int rci = build_inline_codes(ix); if (rci) return rci; }
    p.vecp = ix->vecp.p; p.adj = ix->adj.p; p.first = ix->first.p; p.codes = ix->codes.p; p.codebook = ix->codebook.p;
    p.adjr = (!ov && !novis && ix->use_adjr) ? ix->adjr.p : nullptr; p.medoid_pos = ix->medoid_pos;
    const bool rowpre = getenv("DR_PQ_ROW_PREFETCH") != nullptr;      // A/B (round 4): ids of the predicted next pop's row landed in LDS
    p.novis = novis ? (1u | (rowpre && !(ov && ov->sdc) ? 2u : 0u)) : 0u;
    p.nbcodes = (!ov && ix->inline_codes && ix->nbcodes_valid) ? ix->nbcodes.p : nullptr;
    p.vec8 = ix->vec8_state == 1 ? ix->vec8.p : nullptr;
    // chain-major copy of the batch: the builder hands nothing else; a batch uploaded without it (dr_search_submit, D <= 256)
    // gets it here if this search needs it (large dimensions keep the query in LDS chain-major; the rerank pass reads it)
    if (!ov && !ix->cs->qp_valid && (ix->D > 256 || rerank)) {
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
    p.counter = bs.counter.p;
    p.res_keys = bs.res_keys.p; p.res_n = bs.res_n.p; p.stats = bs.stats.p;
    p.tie_list = bs.tie.p; p.tie_count = bs.counter.p + 1;
    p.log = bs.log.p; p.logcap = logcap;
    p.out_ids = bs.out_ids.p; p.out_dist = bs.out_dist.p; p.out_count = bs.out_count.p;
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
    if (mode == DR_MODE_M1) {
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
    bs.ticket_base += nq;
    if (!ov && ix->kev_pending == dr_index::KEV) harvest_kernel_times(ix, false);
    if (!ov && ix->kev_pending == dr_index::KEV) { HIPCHK(hipStreamSynchronize(ix->stream)); harvest_kernel_times(ix, false); }
    p.lut_g = nullptr;
    const bool want_lut = kd_desc.lut && !(ov && ov->sdc);
    if (want_lut) {
        // The per-query tables T[q][j][c] (A2, fast_pq.py:294-318) of the whole batch, built at full occupancy right
        // before the search kernel that lands them in LDS (engine_kernels.hpp lut_build_kernel). Built by EVERY search --
        // a table is part of its query's search, not of the upload -- and timed separately (dr_timing.lut_kernel_ms).
        if (ix->lut.reserve((size_t)nq * ix->m * 256)) return DR_E_NODEVICE;
        if (!ov) HIPCHK(hipEventRecord(ix->kev[ix->kev_pending][2], st));
        { const int rcl = launch_lut_build(ix, ix->cs->q.p, nq, ix->lut.p, st); if (rcl) return rcl; }
        p.lut_g = ix->lut.p;
    }
    void *args[] = { &p };
    if (!ov) { ix->kev_lut[ix->kev_pending] = want_lut; HIPCHK(hipEventRecord(ix->kev[ix->kev_pending][0], st)); }
    {
        const hipError_t le = hipLaunchKernel(kfn, dim3(grid), dim3(64 * NW), args, lds, st);
        if (le != hipSuccess) { bs.counters_zeroed = false; return fail(DR_E_NODEVICE, "search kernel launch failed: %s", hipGetErrorString(le)); }
    }
    if (ov) return 0;   // the builder consumes res_keys / res_n directly on the stream
    HIPCHK(hipEventRecord(ix->kev[ix->kev_pending][1], st));
    ix->kev_pending++;
    if (rerank) {
        // DR_MODE_PQ + DR_F_RERANK: exact squared L2 of the final list's entries, k best in (distance, id) order
        const float *vecp = ix->vecp.p; const float *qpp = ix->cs->qp.p; const u64 *rkp = bs.res_keys.p; const uint32_t *rnp = bs.res_n.p;
        uint32_t capv = cap, kv = k, nqv = nq; uint32_t *oi = bs.out_ids.p; float *od = bs.out_dist.p; uint32_t *oc = bs.out_count.p;
        KStats *stp = bs.stats.p;
        void *rargs[] = { &vecp, &qpp, &nqv, &rkp, &rnp, &capv, &kv, &oi, &od, &oc, &stp };
        const size_t rlds = (ix->D > 256 ? (size_t)ix->D * 4 : 0) + (size_t)cap * 8;
        HIPCHK(hipLaunchKernel(ix->kern->rerank, dim3(std::min<uint32_t>(nq, (uint32_t)ix->num_cu * 16)), dim3(64), rargs, rlds, st));
    }

    // tie replay for the queries the search kernel listed: one wavefront per query, heap in registers. It runs on
    // its own stream so that it overlaps the NEXT step's search kernel (it needs 19 VGPRs and no LDS, so its
    // wavefronts fit beside the search kernel's); dr_batch_sync / dr_batch_download wait for it.
    HIPCHK(hipEventRecord(bs.search_done, st));
    HIPCHK(hipStreamWaitEvent(ix->fstream, bs.search_done, 0));
    FinalizeParams f;
    f.res_keys = bs.res_keys.p; f.res_n = bs.res_n.p; f.tie_list = bs.tie.p; f.tie_count = bs.counter.p + 1;
    f.log = bs.log.p; f.stats = bs.stats.p;
    f.logcap = logcap; f.cap = cap; f.k = k; f.mode = mode;
    f.out_ids = bs.out_ids.p; f.out_dist = bs.out_dist.p;
    if (!ix->fin_stat.p) { if (ix->fin_stat.reserve(1, true)) return DR_E_NODEVICE; }
    f.ntie_stat = ix->fin_stat.p;
    HIPCHK(hipEventRecord(bs.fin_start, ix->fstream));
    static const bool skip_fin = getenv("DR_SKIP_FINALIZE") != nullptr;   // timing experiment only: tie order is then wrong
    if (!skip_fin && !rerank)      // (the rerank pass has already written a total (distance, id) order)
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
    if (k_m1 && ix->adc_live < 0) {
        // regime of this (graph, PQ) state: did the rerank policy really consult the ADC on this batch? (the one
        // launch per index state that is waited for on the host)
        HIPCHK(hipStreamSynchronize(st));
        std::vector<KStats> st(std::min<uint32_t>(nq, 1024));
        HIPCHK(hipMemcpy(st.data(), bs.stats.p, st.size() * sizeof(KStats), hipMemcpyDeviceToHost));
        uint64_t evald = 0, all = 0;
        for (const KStats &x : st) { evald += x.pq_evaluated; all += x.pq; }
        ix->adc_live = (2 * evald > all) ? 1 : 0;
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
        HIPCHK(hipMemset(ix->fin_stat.p, 0, 4));
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
    for (hipStream_t st : { ix->up_stream, ix->stream, ix->fstream, ix->down_stream }) HIPCHK(hipStreamSynchronize(st));
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

// ---- pipelined batches: upload | search | tie order | download on four streams ------------------------------------
static int pin_reserve(void **p, size_t *have, size_t need)
{
    if (*have >= need) return 0;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *have = 0;
    if (hipHostMalloc(p, need, hipHostMallocDefault) != hipSuccess) return fail(DR_E_NODEVICE, "hipHostMalloc(%zu) failed", need);
    *have = need;
    return 0;
}

// ---- coalesced launches (round 4) -------------------------------------------------------------------------------------
// search kernels of the pipelined path that are queued or running (their BatchSet's search_done has not fired)
static int searches_in_flight(dr_index *ix)
{
    int n = 0;
    for (const PipeGroup &gr : ix->groups)
        if (gr.state == 2 && gr.set >= 0 && hipEventQuery(ix->sets[gr.set].search_done) == hipErrorNotReady) n++;
    (void)hipGetLastError();
    return n;
}

// every job of a group answers `rc` from now on (its launch failed: nothing was or will be written to their buffers)
static void fail_group_locked(dr_index *ix, int g, int rc)
{
    PipeGroup &gr = ix->groups[g];
    const std::string msg = g_err;
    // copies that read the jobs' staging buffers / kernels that use the group's slot may be queued: drain before anything is reused
    for (hipStream_t st : { ix->up_stream, ix->stream, ix->fstream, ix->down_stream }) (void)hipStreamSynchronize(st);
    (void)hipGetLastError();
    for (PipeJob &jb : ix->jobs) if (jb.active && jb.group == g) { jb.rc = rc; jb.err = msg; }
    if (gr.set >= 0 && ix->sets[gr.set].owner_group == g) ix->sets[gr.set].owner_group = -1;
    gr.state = 2; gr.set = -1;        // "launched": its jobs only have to be collected
    if (ix->open_group == g) ix->open_group = -1;
    g_err = msg;
}

// queues the search, the tie-order pass and the download of an open group: ONE launch for all its jobs
static int launch_group_locked(dr_index *ix, int g)
{
    PipeGroup &gr = ix->groups[g];
    if (gr.state != 1) return 0;
    if (ix->open_group == g) ix->open_group = -1;
    QSlot &qs = ix->slots[DR_MAX_RESIDENT + g];
    const int rc = [&]() -> int {
        HIPCHK(hipEventRecord(gr.up_done, ix->up_stream));       // behind the upload (and bound kernels) of its last job
        HIPCHK(hipStreamWaitEvent(ix->stream, gr.up_done, 0));
        qs.nq = gr.nq; qs.q_u8 = gr.q_u8; qs.qp_valid = gr.with_qp;
        qs.pq_ub_valid = (gr.mode == DR_MODE_M1 && ix->m != 0);   // (every job computed the bounds of its run on the upload stream)
        QSlot *const keep = ix->cs;
        ix->cs = &qs;
        int r = run_locked(ix, gr.k, gr.L, gr.bw, gr.mode, gr.policy, gr.flags);
        ix->cs = keep;
        if (r) return r;
        gr.set = ix->last_set;
        dr_index::BatchSet &bs = ix->sets[gr.set];
        bs.owner_group = g;
        // download: behind the tie-order pass of this launch (which is behind its search kernel); jobs copy their rows out of the slab
        const size_t b_ids = (size_t)gr.nq * gr.k * 4, b_cnt = (size_t)gr.nq * 4, b_st = (size_t)gr.nq * sizeof(KStats);
        r = pin_reserve(&gr.pin_out, &gr.pin_out_bytes, 2 * b_ids + b_cnt + b_st + 4);
        if (r) return r;
        unsigned char *hp = static_cast<unsigned char *>(gr.pin_out);
        HIPCHK(hipStreamWaitEvent(ix->down_stream, bs.fin_done, 0));
        HIPCHK(hipMemcpyAsync(hp, bs.out_ids.p, b_ids, hipMemcpyDeviceToHost, ix->down_stream));
        HIPCHK(hipMemcpyAsync(hp + b_ids, bs.out_dist.p, b_ids, hipMemcpyDeviceToHost, ix->down_stream));
        HIPCHK(hipMemcpyAsync(hp + 2 * b_ids, bs.out_count.p, b_cnt, hipMemcpyDeviceToHost, ix->down_stream));
        HIPCHK(hipMemcpyAsync(hp + 2 * b_ids + b_cnt, bs.stats.p, b_st, hipMemcpyDeviceToHost, ix->down_stream));
        HIPCHK(hipMemcpyAsync(hp + 2 * b_ids + b_cnt + b_st, ix->fin_stat.p, 4, hipMemcpyDeviceToHost, ix->down_stream));
        HIPCHK(hipEventRecord(gr.down_done, ix->down_stream));
        return 0;
    }();
    if (rc) { fail_group_locked(ix, g, rc); return rc; }
    gr.state = 2;
    ix->pipe_launches++; ix->pipe_tickets += gr.njobs; ix->pipe_queries += gr.nq;
    ix->pipe_max_tickets = std::max<uint64_t>(ix->pipe_max_tickets, gr.njobs);
    return 0;
}

// Launch policy for the open group: at once while fewer than TWO search kernels of the pipeline are queued or running (one
// running + one queued behind it keeps the search stream fed; a lone request finds none and never waits), otherwise the
// group keeps collecting until it is full -- so under load a launch carries what arrived during one kernel. (A threshold
// of three launched nearly every submit alone: the host, blocked on the oldest launch's download, came back to find the
// stream nearly dry -- gpurun_out r04 call 2: 1.06 submits per launch.) Called by every submit and, while it waits, by
// dr_search_wait.
static int kick_locked(dr_index *ix, bool force = false)
{
    const int g = ix->open_group;
    if (g < 0) return 0;
    if (!force && (ix->hold_always || searches_in_flight(ix) >= 2)) return 0;
    return launch_group_locked(ix, g);
}

static int finish_job_locked(dr_index *ix, int j)
{
    PipeJob &jb = ix->jobs[j];
    if (!jb.active) return 0;
    HIPCHK(hipSetDevice(ix->device));
    PipeGroup &gr = ix->groups[jb.group];
    if (gr.state == 1) (void)launch_group_locked(ix, jb.group);     // (a failure is recorded in the group's jobs, this one included)
    auto retire = [&]() {
        jb.active = false;
        if (gr.live > 0 && --gr.live == 0) {
            if (gr.set >= 0 && ix->sets[gr.set].owner_group == jb.group) ix->sets[gr.set].owner_group = -1;
            gr.state = 0; gr.set = -1;
        }
    };
    if (jb.rc) {
        // its launch failed: the error belongs to the ticket (dr_search_wait answers it), not to whoever needs the slot now
        if (ix->failed_tickets.size() >= 1024) ix->failed_tickets.erase(ix->failed_tickets.begin());
        ix->failed_tickets[jb.ticket] = std::make_pair(jb.rc, jb.err);
        jb.rc = 0; retire();
        return 0;
    }
    // another group is still collecting: while this job's results are on their way the waiting thread keeps the search stream fed
    while (ix->open_group >= 0 && hipEventQuery(gr.down_done) == hipErrorNotReady) {
        (void)hipGetLastError();
        (void)kick_locked(ix);            // (a failure stays with that group's jobs)
        if (ix->open_group >= 0) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    (void)hipGetLastError();
    HIPCHK(hipEventSynchronize(gr.down_done));
    const size_t g_ids = (size_t)gr.nq * gr.k * 4, g_cnt = (size_t)gr.nq * 4, g_st = (size_t)gr.nq * sizeof(KStats);
    const unsigned char *hp = static_cast<const unsigned char *>(gr.pin_out);
    memcpy(jb.out_ids, hp + (size_t)jb.q0 * jb.k * 4, (size_t)jb.nq * jb.k * 4);
    memcpy(jb.out_dist, hp + g_ids + (size_t)jb.q0 * jb.k * 4, (size_t)jb.nq * jb.k * 4);
    memcpy(jb.out_count, hp + 2 * g_ids + (size_t)jb.q0 * 4, (size_t)jb.nq * 4);
    if (jb.stats) memcpy(jb.stats, hp + 2 * g_ids + g_cnt + (size_t)jb.q0 * sizeof(KStats), (size_t)jb.nq * sizeof(KStats));
    uint32_t ties = 0;
    memcpy(&ties, hp + 2 * g_ids + g_cnt + g_st, 4);
    if (gr.nq >= 1024) ix->fin_hint = std::max<uint32_t>(ties, 16);
    harvest_kernel_times(ix, false);
    retire();
    return 0;
}

// the results of every job of a launched group reach their callers' buffers (its BatchSet or its slot is needed)
static int finish_group_locked(dr_index *ix, int g)
{
    int rc_first = 0; std::string msg;
    for (int j = 0; j < DR_MAX_JOBS; j++)
        if (ix->jobs[j].active && ix->jobs[j].group == g) { const int rc = finish_job_locked(ix, j); if (rc && !rc_first) { rc_first = rc; msg = g_err; } }
    if (rc_first) g_err = msg;
    return rc_first;
}

extern "C" int dr_search_submit(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t L, uint32_t beam_width,
                                uint32_t mode, uint32_t band_policy, uint32_t flags, uint32_t *out_ids, float *out_dist,
                                uint32_t *out_count, dr_stats *stats, uint64_t *out_ticket)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    if (!out_ids || !out_dist || !out_count || !out_ticket) return fail(DR_E_ARG, "null output buffer");
    if (!queries || nq == 0) return fail(DR_E_ARG, "empty query batch");
    if (nq > DR_MAX_CHUNK) return fail(DR_E_UNSUPPORTED, "dr_search_submit takes at most %u queries per batch", DR_MAX_CHUNK);
    if (k == 0) return fail(DR_E_ARG, "k must be positive");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    const int j = (int)(ix->next_ticket % DR_MAX_JOBS);
    PipeJob &jb = ix->jobs[j];
    if (jb.active) { const int rc = finish_job_locked(ix, j); if (rc) return rc; }      // (the ticket DR_MAX_JOBS submits ago)
    // pinned source (dr_host_alloc / hipHostMalloc / hipHostRegister): the copy engine reads it in place; pageable: staged
    const float *src = queries;
    {
        hipPointerAttribute_t at;
        const bool pinned = hipPointerGetAttributes(&at, queries) == hipSuccess && at.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (!pinned) {
            const size_t bytes = (size_t)nq * ix->D * 4;
            const int rc = pin_reserve(&jb.pin_in, &jb.pin_in_bytes, bytes);
            if (rc) return rc;
            memcpy(jb.pin_in, queries, bytes);
            src = static_cast<const float *>(jb.pin_in);
        }
    }
    const bool q_u8 = (ix->vec8_state == 1 || (ix->vec8_state == 0 && ix->D == 128)) && queries_are_u8(src, (size_t)nq * ix->D);
    // ---- the group this job rides in: the open one if the parameters agree and it has room, else a new one
    int g = ix->open_group;
    if (g >= 0) {
        const PipeGroup &og = ix->groups[g];
        const bool same = og.k == k && og.L == L && og.bw == beam_width && og.mode == mode && og.policy == band_policy && og.flags == flags;
        if (!same || og.nq + nq > og.cap) { (void)launch_group_locked(ix, g); g = -1; }     // (a failure stays with that group's jobs)
    }
    if (g < 0) {
        g = (int)(ix->next_group % DR_PIPE_DEPTH);
        PipeGroup &ng = ix->groups[g];
        if (ng.state == 2) { const int rc = finish_group_locked(ix, g); if (rc) return rc; }   // (the launch DR_PIPE_DEPTH launches ago)
        ix->next_group++;
        ng.state = 1; ng.nq = 0; ng.njobs = 0; ng.live = 0; ng.set = -1;
        ng.cap = std::max<uint32_t>(nq, std::min<uint32_t>(ix->coalesce_cap, DR_MAX_CHUNK));
        ng.k = k; ng.L = L; ng.bw = beam_width; ng.mode = mode; ng.policy = band_policy; ng.flags = flags;
        ng.q_u8 = true;
        static const bool always_permute = getenv("DR_SUBMIT_PERMUTE") != nullptr;      // A/B: the round-2 upload (copy + permute kernel)
        ng.with_qp = ix->D > 256 || always_permute;
        ix->open_group = g;
    }
    PipeGroup &gr = ix->groups[g];
    QSlot &qs = ix->slots[DR_MAX_RESIDENT + g];
    // From here on copies that read jb.pin_in may be queued: a failure must not hand the ticket slot back (the next submit
    // would overwrite the staging buffer under a copy in flight) before those copies have drained -- fail_group_locked drains.
    const int rcq = [&]() -> int {
        int rc = upload_slot_async(ix, qs, src, nq, ix->up_stream, gr.with_qp, gr.nq, gr.cap);
        if (rc) return rc;
        if (mode == DR_MODE_M1 && ix->m) {
            // the per-query ADC bounds travel with the upload, off the search stream
            rc = launch_pq_bound(ix, qs, nq, ix->up_stream, gr.nq, gr.cap);
            if (rc) return rc;
        }
        return 0;
    }();
    if (rcq) {
        // this job never joined; the jobs already in the group lose their launch with it (their uploads share the slot)
        fail_group_locked(ix, g, rcq);
        return rcq;
    }
    jb.active = true; jb.ticket = ix->next_ticket++; jb.group = g; jb.q0 = gr.nq; jb.nq = nq; jb.k = k; jb.rc = 0;
    jb.out_ids = out_ids; jb.out_dist = out_dist; jb.out_count = out_count; jb.stats = stats;
    gr.nq += nq; gr.njobs++; gr.live++;
    gr.q_u8 = gr.q_u8 && q_u8;
    *out_ticket = jb.ticket;
    // full (a further job of this size would not fit), or the search stream is about to run dry: launch now; else it collects
    const bool full = gr.nq + nq > gr.cap || gr.njobs >= DR_MAX_JOBS / 2;
    const int rcl = kick_locked(ix, full);
    if (rcl) {      // this job's launch failed: the submit fails and its ticket is void
        const std::string msg = g_err;
        (void)finish_job_locked(ix, j); ix->failed_tickets.erase(jb.ticket);
        g_err = msg;
        return rcl;
    }
    return 0;
}

extern "C" int dr_search_wait(dr_index *ix, uint64_t ticket)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::unique_lock<std::mutex> lk(ix->mu);
    if (ticket == 0 || ticket >= ix->next_ticket) return fail(DR_E_ARG, "unknown ticket %llu", (unsigned long long)ticket);
    HIPCHK(hipSetDevice(ix->device));
    // The handle is NOT held while the ticket's results are on their way: other threads submit meanwhile (their requests ride
    // in the launch this one may still be waiting for -- a pool of request handlers, one query each, app.py:84-130), and the
    // waiting thread is the one that keeps the launch policy running.
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        int j = -1;
        for (int i = 0; i < DR_MAX_JOBS; i++) if (ix->jobs[i].active && ix->jobs[i].ticket == ticket) j = i;
        if (j < 0) break;                 // finished meanwhile (a submit or a sync needed its slot)
        PipeJob &jb = ix->jobs[j];
        PipeGroup &gr = ix->groups[jb.group];
        // still collecting: launched by the policy once fewer than two searches are queued (at once under the test hook)
        if (!jb.rc && gr.state == 1) (void)kick_locked(ix, ix->hold_always);
        else (void)kick_locked(ix);       // (another group may be collecting: keep the search stream fed)
        bool ready = jb.rc != 0;
        if (!ready && gr.state == 2) { ready = hipEventQuery(gr.down_done) != hipErrorNotReady; (void)hipGetLastError(); }
        if (ready) { const int rc = finish_job_locked(ix, j); if (rc) return rc; break; }
        lk.unlock();
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(300)) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
        lk.lock();
    }
    auto it = ix->failed_tickets.find(ticket);
    if (it != ix->failed_tickets.end()) {
        const int rc = it->second.first; g_err = it->second.second;
        ix->failed_tickets.erase(it);
        return rc;
    }
    return 0;     // finished (now, or earlier: a later submit or a sync needed its slot)
}

// launches whatever dr_search_submit is still holding back (a caller that submits and then goes away for a while)
extern "C" int dr_search_flush(dr_index *ix)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    return kick_locked(ix, true);
}

// test hook: with `on`, held submits are launched only when their group is full, flushed or waited for (never by the
// "search stream is running dry" rule, which depends on timing)
extern "C" int dr_debug_hold(dr_index *ix, int on)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->hold_always = on != 0;
    return 0;
}

// launches of the pipelined path since the handle was created: [0] launches, [1] tickets they carried, [2] most tickets in one
// launch, [3] queries
extern "C" int dr_pipeline_stats(dr_index *ix, uint64_t *out4)
{
    if (!ix || !out4) return fail(DR_E_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    out4[0] = ix->pipe_launches; out4[1] = ix->pipe_tickets; out4[2] = ix->pipe_max_tickets; out4[3] = ix->pipe_queries;
    return 0;
}

extern "C" int dr_set_coalesce(dr_index *ix, uint32_t max_queries)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (max_queries > DR_MAX_CHUNK) return fail(DR_E_ARG, "coalesced launches hold at most %u queries", DR_MAX_CHUNK);
    HIPCHK(hipSetDevice(ix->device));
    { const int rc = kick_locked(ix, true); if (rc) return rc; }
    ix->coalesce_cap = max_queries;
    return 0;
}

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


extern "C" int dr_search_batch(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t L,
                               uint32_t beam_width, uint32_t mode, uint32_t band_policy, uint32_t flags,
                               uint32_t *out_ids, float *out_dist, uint32_t *out_count, dr_stats *stats)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    if (!out_ids || !out_dist || !out_count) return fail(DR_E_ARG, "null output buffer");
    if (!queries || nq == 0) return fail(DR_E_ARG, "empty query batch");
    std::lock_guard<std::mutex> lk(ix->mu);
    float h2d = 0, ker = 0, fin = 0, d2h = 0;
    for (uint32_t q0 = 0; q0 < nq; q0 += DR_MAX_CHUNK) {
        const uint32_t n = std::min(DR_MAX_CHUNK, nq - q0);
        int rc = upload_queries_locked(ix, queries + (size_t)q0 * ix->D, n, false);
        if (!rc) rc = run_locked(ix, k, L, beam_width, mode, band_policy, flags);
        if (!rc) rc = download_locked(ix, out_ids + (size_t)q0 * k, out_dist + (size_t)q0 * k, out_count + q0,
                                      stats ? stats + q0 : nullptr);
        if (rc) return rc;
        h2d += ix->timing.h2d_ms; ker += ix->timing.search_kernel_ms; fin += ix->timing.finalize_kernel_ms; d2h += ix->timing.d2h_ms;
    }
    ix->timing.h2d_ms = h2d; ix->timing.search_kernel_ms = ker; ix->timing.finalize_kernel_ms = fin; ix->timing.d2h_ms = d2h;
    ix->timing.total_ms = h2d + ker + fin + d2h;
    return 0;
}

// Float64 queries (the CLI hands np.array(list): diskrag.py:194, quirk Q8). M1 and M2 only -- the two searches the
// CLI reaches (search_engine.py:566-573). One wavefront per query, at most 64 queries per launch; see search_f64.hpp.
extern "C" int dr_search_batch_f64(dr_index *ix, const double *queries, uint32_t nq, uint32_t k, uint32_t L,
                                   uint32_t beam_width, uint32_t mode, uint32_t band_policy, uint32_t flags,
                                   uint32_t *out_ids, double *out_dist, uint32_t *out_count, dr_stats *stats)
{
    (void)flags;
    if (!ix) return fail(DR_E_ARG, "null index");
    if (!out_ids || !out_dist || !out_count) return fail(DR_E_ARG, "null output buffer");
    if (!queries || nq == 0) return fail(DR_E_ARG, "empty query batch");
    if (mode != DR_MODE_M1 && mode != DR_MODE_M2) return fail(DR_E_UNSUPPORTED, "float64 queries: modes M1 and M2 only");
    if (k == 0) return fail(DR_E_ARG, "k must be positive");
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "float64 search"); if (rcv) return rcv; }
    if (mode == DR_MODE_M1 && ix->m == 0) return fail(DR_E_NOPQ, "mode %u needs PQ data (dr_index_set_pq)", mode);
    const uint32_t cap = (mode == DR_MODE_M2) ? beam_width : L;
    if (cap == 0) return fail(DR_E_ARG, "result-list capacity is zero (L / beam_width)");
    if (cap > 512) return fail(DR_E_UNSUPPORTED, "result-list capacity %u > 512", cap);
    HIPCHK(hipSetDevice(ix->device));
    const uint32_t D = ix->D, CH = 64;
    const bool pq = (mode == DR_MODE_M1);
    // LDS: 2 queries (f64), both heaps, per-expansion arrays, the table; the candidates heap takes what is left
    const size_t fixed = (size_t)2 * D * 8 + (size_t)(cap + 1) * 12 + 64 * 8 + 64 * 4 + 64 * 4 + (pq ? (size_t)ix->m * 256 * 4 : 0) + 64;
    if (fixed + 1024 * 12 > 160 * 1024) return fail(DR_E_UNSUPPORTED, "float64 search does not fit in LDS (D=%u, m=%u, capacity %u)", D, ix->m, cap);
    const uint32_t cand_cap = (uint32_t)std::min<size_t>((160 * 1024 - fixed) / 12, 8192) & ~1u;
    const size_t lds = fixed + (size_t)cand_cap * 12;
    const void *kfn = ix->kern->search_f64;
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
        p.mode = mode; p.k = k; p.cap = cap; p.L = L; p.bw = beam_width; p.policy = band_policy;
        p.max_steps = pq ? (uint32_t)std::min<uint64_t>((uint64_t)L * 10, ix->N) : 0xFFFFFFFFu;
        p.vis = dvis.p; p.vis_words = vis_words; p.cand_cap = cand_cap;
        p.out_ids = dids.p; p.out_dist = dd.p; p.out_count = dcnt.p; p.stats = dst.p;
        if (hipMemcpyAsync(dq.p, queries + (size_t)q0 * D, (size_t)n * D * 8, hipMemcpyHostToDevice, ix->stream) != hipSuccess ||
            hipMemsetAsync(dvis.p, 0, (size_t)n * vis_words * 4, ix->stream) != hipSuccess) { rc = fail(DR_E_NODEVICE, "float64 search: upload failed"); break; }
        void *args[] = { &p };
        if (hipLaunchKernel(kfn, dim3(n), dim3(64), args, lds, ix->stream) != hipSuccess) { rc = fail(DR_E_NODEVICE, "float64 search: launch failed: %s", hipGetErrorString(hipGetLastError())); break; }
        // results through the pinned host slab (see download_locked)
        const size_t b_ids = (size_t)n * k * 4, b_d = (size_t)n * k * 8, b_cnt = (size_t)n * 4, b_st = (size_t)n * sizeof(KStats);
        const size_t need = b_d + b_ids + b_cnt + b_st;
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
            hipStreamSynchronize(ix->stream) != hipSuccess) { rc = fail(DR_E_NODEVICE, "float64 search: %s", hipGetErrorString(hipGetLastError())); break; }
        memcpy(out_dist + (size_t)q0 * k, hp, b_d);
        memcpy(out_ids + (size_t)q0 * k, hp + b_d, b_ids);
        memcpy(out_count + q0, hp + b_d + b_ids, b_cnt);
        if (stats) memcpy(stats + q0, hp + b_d + b_ids + b_cnt, b_st);
    }
    return rc;
}

// --------------------------------------------------------------------------------- kernel-level entry points

extern "C" int dr_exact_distances(dr_index *ix, const float *queries, uint32_t nq, const uint32_t *node_ids, uint32_t n,
                                  float *out)
{
    if (!ix || !queries || !node_ids || !out || nq == 0 || n == 0) return fail(DR_E_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "dr_exact_distances"); if (rcv) return rcv; }
    for (uint32_t i = 0; i < n; i++) if (node_ids[i] >= ix->N) return fail(DR_E_ARG, "node id %u out of range", node_ids[i]);
    int rc = upload_queries_locked(ix, queries, nq);
    if (rc) return rc;
    DevBuf<uint32_t> ids; DevBuf<float> o;
    if (ids.reserve(n) || o.reserve((size_t)nq * n)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpyAsync(ids.p, node_ids, (size_t)n * 4, hipMemcpyHostToDevice, ix->stream));
    const float *vecp = ix->vecp.p; const float *qp = ix->cs->qp.p; const uint32_t *idp = ids.p; float *op = o.p;
    void *args[] = { &vecp, &qp, &nq, &idp, &n, &op };
    const unsigned gx = std::min<unsigned>((n + 7) / 8, 1024);
    HIPCHK(hipLaunchKernel(ix->kern->exact, dim3(gx, nq), dim3(64), args, (size_t)ix->D * 4, ix->stream));
    HIPCHK(hipMemcpyAsync(out, o.p, (size_t)nq * n * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ids.release(); o.release();
    return 0;
}

extern "C" int dr_distance_table(dr_index *ix, const float *queries, uint32_t nq, float *out)
{
    if (!ix || !queries || !out || nq == 0) return fail(DR_E_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->m == 0) return fail(DR_E_NOPQ, "no PQ data");
    int rc = upload_queries_locked(ix, queries, nq);
    if (rc) return rc;
    DevBuf<float> o;
    if (o.reserve((size_t)nq * ix->m * 256)) return DR_E_NODEVICE;
    { const int rcl = launch_lut_build(ix, ix->cs->q.p, nq, o.p); if (rcl) return rcl; }      // the kernel the searches use
    HIPCHK(hipMemcpyAsync(out, o.p, (size_t)nq * ix->m * 256 * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    o.release();
    return 0;
}

static int adc_common(dr_index *ix, const float *queries, uint32_t nq, const uint32_t *node_ids, uint64_t n,
                      float *out_sq, float *out_sqrt, float *kernel_ms)
{
    if (ix->m == 0) return fail(DR_E_NOPQ, "no PQ data");
    int rc = upload_queries_locked(ix, queries, nq);
    if (rc) return rc;
    DevBuf<uint32_t> ids; DevBuf<float> o1, o2;
    if (node_ids) {
        for (uint64_t i = 0; i < n; i++) if (node_ids[i] >= ix->N) return fail(DR_E_ARG, "node id out of range");
        if (ids.reserve(n)) return DR_E_NODEVICE;
        HIPCHK(hipMemcpyAsync(ids.p, node_ids, n * 4, hipMemcpyHostToDevice, ix->stream));
    }
    if (out_sq && o1.reserve((size_t)nq * n)) return DR_E_NODEVICE;
    if (out_sqrt && o2.reserve((size_t)nq * n)) return DR_E_NODEVICE;
    const size_t lds = (size_t)ix->D * 4 + (size_t)ix->m * 256 * 4;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&adc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned gx = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)ix->num_cu * 8);
    HIPCHK(hipEventRecord(ix->ev[2], ix->stream));
    hipLaunchKernelGGL(adc_kernel, dim3(gx, nq), dim3(256), lds, ix->stream, ix->codebook.p, ix->cs->q.p, ix->codes.p,
                       node_ids ? ids.p : nullptr, n, ix->D, ix->m, ix->sd, out_sq ? o1.p : nullptr, out_sqrt ? o2.p : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ix->ev[3], ix->stream));
    if (out_sq) HIPCHK(hipMemcpyAsync(out_sq, o1.p, (size_t)nq * n * 4, hipMemcpyDeviceToHost, ix->stream));
    if (out_sqrt) HIPCHK(hipMemcpyAsync(out_sqrt, o2.p, (size_t)nq * n * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ix->ev[2], ix->ev[3]);
    ids.release(); o1.release(); o2.release();
    return 0;
}

extern "C" int dr_adc(dr_index *ix, const float *queries, uint32_t nq, const uint32_t *node_ids, uint32_t n,
                      float *out_sq, float *out_sqrt)
{
    if (!ix || !queries || !node_ids || nq == 0 || n == 0) return fail(DR_E_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    return adc_common(ix, queries, nq, node_ids, n, out_sq, out_sqrt, nullptr);
}

extern "C" int dr_pq_scan(dr_index *ix, const float *queries, uint32_t nq, float *out_sq, float *kernel_ms)
{
    return dr_pq_scan_best(ix, queries, nq, out_sq, nullptr, nullptr, kernel_ms);
}

// Flat scan of all N code words per query (pq_scan_kernel): out_sq (optional) gets every squared ADC distance,
// out_best_id / out_best_sq (optional) the nearest code word per query (smallest id among equal sums).
static int pq_scan_best_locked(dr_index *ix, const float *queries, uint32_t nq, float *out_sq, uint32_t *out_best_id,
                               float *out_best_sq, float *kernel_ms);
extern "C" int dr_pq_scan_best(dr_index *ix, const float *queries, uint32_t nq, float *out_sq, uint32_t *out_best_id,
                               float *out_best_sq, float *kernel_ms)
{
    if (!ix || !queries || nq == 0) return fail(DR_E_ARG, "bad argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    return pq_scan_best_locked(ix, queries, nq, out_sq, out_best_id, out_best_sq, kernel_ms);
}
// (the caller holds ix->mu: the PQ-only builder scans for its medoid in the middle of a build)
static int pq_scan_best_locked(dr_index *ix, const float *queries, uint32_t nq, float *out_sq, uint32_t *out_best_id,
                               float *out_best_sq, float *kernel_ms)
{
    if (ix->m == 0) return fail(DR_E_NOPQ, "no PQ data");
    const uint32_t m = ix->m;
    if ((m & 15u) != 0 || m > 64) {
        // generic form (any m): all distances through adc_kernel, the nearest code word picked on the host
        std::vector<float> tmp;
        float *all = out_sq;
        if (!all && (out_best_id || out_best_sq)) { tmp.resize((size_t)nq * ix->N); all = tmp.data(); }
        int rcg = adc_common(ix, queries, nq, nullptr, ix->N, all, nullptr, kernel_ms);
        if (rcg || !all) return rcg;
        for (uint32_t qi = 0; qi < nq; qi++) {
            const float *row = all + (size_t)qi * ix->N;
            uint64_t bi = 0;
            for (uint64_t i = 1; i < ix->N; i++) if (row[i] < row[bi]) bi = i;
            if (out_best_id) out_best_id[qi] = (uint32_t)bi;
            if (out_best_sq) out_best_sq[qi] = row[bi];
        }
        return 0;
    }
    int rc = upload_queries_locked(ix, queries, nq);
    if (rc) return rc;
    const uint64_t n = ix->N;
    DevBuf<float> o1; DevBuf<u64> best;
    if (out_sq && o1.reserve((size_t)nq * n)) return DR_E_NODEVICE;
    if (best.reserve(nq)) return DR_E_NODEVICE;
    HIPCHK(hipMemsetAsync(best.p, 0xFF, (size_t)nq * 8, ix->stream));
    const size_t lds = (size_t)ix->D * 4 + (size_t)m * 256 * 4;
    const void *kfn = m == 16 ? reinterpret_cast<const void *>(&pq_scan_kernel<1>) : m == 32 ? reinterpret_cast<const void *>(&pq_scan_kernel<2>)
                    : m == 48 ? reinterpret_cast<const void *>(&pq_scan_kernel<3>) : reinterpret_cast<const void *>(&pq_scan_kernel<4>);
    HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 1;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 256, lds));
    if (occ < 1) occ = 1;
    // persistent blocks: the table is built once per block; with several queries in flight the rows share the chip
    const unsigned per_q = (unsigned)std::max<uint64_t>(1, (uint64_t)occ * ix->num_cu / std::min<uint32_t>(nq, (uint32_t)occ * ix->num_cu));
    const unsigned gx = (unsigned)std::min<uint64_t>((n + 255) / 256, per_q);
    const float *cbp = ix->codebook.p; const float *qp = ix->cs->q.p; const uint8_t *cdp = ix->codes.p; uint64_t nn = n;
    uint32_t D = ix->D, sd = ix->sd; float *op = out_sq ? o1.p : nullptr; u64 *bp = best.p;
    void *args[] = { &cbp, &qp, &cdp, &nn, &D, &sd, &op, &bp };
    HIPCHK(hipEventRecord(ix->ev[2], ix->stream));
    HIPCHK(hipLaunchKernel(kfn, dim3(gx, nq), dim3(256), args, lds, ix->stream));
    HIPCHK(hipEventRecord(ix->ev[3], ix->stream));
    std::vector<u64> hb(nq);
    if (out_sq) HIPCHK(hipMemcpyAsync(out_sq, o1.p, (size_t)nq * n * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipMemcpyAsync(hb.data(), best.p, (size_t)nq * 8, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    if (kernel_ms) (void)hipEventElapsedTime(kernel_ms, ix->ev[2], ix->ev[3]);
    for (uint32_t i = 0; i < nq; i++) {
        if (out_best_id) out_best_id[i] = (uint32_t)hb[i];
        if (out_best_sq) { const uint32_t b = (uint32_t)(hb[i] >> 32); memcpy(&out_best_sq[i], &b, 4); }
    }
    o1.release(); best.release();
    return 0;
}

// Brute-force ADC search: the k nearest code words per query by a flat scan (pq_scan_topk_kernel + topk_merge_kernel).
extern "C" int dr_pq_scan_topk(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t *out_ids, float *out_sq,
                               float *kernel_ms)
{
    if (!ix || !queries || !out_ids || nq == 0 || k == 0 || k > 64) return fail(DR_E_ARG, "bad argument (k <= 64)");
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->m == 0) return fail(DR_E_NOPQ, "no PQ data");
    const uint32_t m = ix->m;
    if ((m & 15u) != 0 || m > 64) return fail(DR_E_UNSUPPORTED, "dr_pq_scan_topk needs n_subvectors in {16, 32, 48, 64}");
    HIPCHK(hipSetDevice(ix->device));
    const size_t lds = (size_t)m * 1024 + 4 * 64 * 8;
    const void *kfn = m == 16 ? reinterpret_cast<const void *>(&pq_scan_topk_kernel<1>) : m == 32 ? reinterpret_cast<const void *>(&pq_scan_topk_kernel<2>)
                    : m == 48 ? reinterpret_cast<const void *>(&pq_scan_topk_kernel<3>) : reinterpret_cast<const void *>(&pq_scan_topk_kernel<4>);
    HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 1;
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, 256, lds));
    if (occ < 1) occ = 1;
    // queries are scanned in groups that fill the chip once: `per_q` blocks per query, `group` queries per launch
    const uint64_t n = ix->N;
    const uint32_t slots = (uint32_t)occ * ix->num_cu;
    const uint32_t per_q = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((n + 16383) / 16384, std::min<uint32_t>(slots, 256)));
    const uint32_t group = std::max<uint32_t>(1, std::min<uint32_t>(nq, 4 * slots / per_q));
    DevBuf<float> lut, osq; DevBuf<u64> part; DevBuf<uint32_t> oid;
    if (lut.reserve((size_t)group * m * 256) || part.reserve((size_t)group * per_q * 4 * k) || oid.reserve((size_t)nq * k) || osq.reserve((size_t)nq * k))
        return DR_E_NODEVICE;
    float ms_sum = 0;
    for (uint32_t q0 = 0; q0 < nq; q0 += group) {
        const uint32_t g = std::min(group, nq - q0);
        int rc = upload_queries_locked(ix, queries + (size_t)q0 * ix->D, g);
        if (rc) return rc;
        rc = launch_lut_build(ix, ix->cs->q.p, g, lut.p);
        if (rc) return rc;
        const float *lp = lut.p; const uint8_t *cdp = ix->codes.p; uint64_t nn = n; uint32_t kk = k; u64 *pp = part.p;
        void *args[] = { &lp, &cdp, &nn, &kk, &pp };
        HIPCHK(hipEventRecord(ix->ev[2], ix->stream));
        HIPCHK(hipLaunchKernel(kfn, dim3(per_q, g), dim3(256), args, lds, ix->stream));
        HIPCHK(hipEventRecord(ix->ev[3], ix->stream));
        hipLaunchKernelGGL(topk_merge_kernel, dim3(g), dim3(64), 0, ix->stream, part.p, per_q * 4, k, oid.p + (size_t)q0 * k, osq.p + (size_t)q0 * k);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ix->stream));
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ix->ev[2], ix->ev[3]);
        ms_sum += ms;
    }
    HIPCHK(hipMemcpy(out_ids, oid.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    if (out_sq) HIPCHK(hipMemcpy(out_sq, osq.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
    if (kernel_ms) *kernel_ms = ms_sum;
    return 0;
}

extern "C" int dr_bruteforce_topk(dr_index *ix, const float *queries, uint32_t nq, uint32_t k, uint32_t *out_ids,
                                  float *out_dist)
{
    if (!ix || !queries || !out_ids || nq == 0 || k == 0 || k > 64) return fail(DR_E_ARG, "bad argument (k <= 64)");
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "dr_bruteforce_topk"); if (rcv) return rcv; }
    int rc = upload_queries_locked(ix, queries, nq);
    if (rc) return rc;
    DevBuf<uint32_t> oi; DevBuf<float> od;
    if (oi.reserve((size_t)nq * k) || od.reserve((size_t)nq * k)) return DR_E_NODEVICE;
    const float *vecp = ix->vecp.p; const float *qp = ix->cs->qp.p; uint64_t N = ix->N; uint32_t *oip = oi.p; float *odp = od.p;
    // small batches: the rows are cut into S slices per query so that the launch still has a few thousand wavefronts
    const uint32_t S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(64, N / 4096 + 1), ((uint64_t)ix->num_cu * 16) / nq));
    DevBuf<u64> part;
    if (S > 1 && part.reserve((size_t)nq * S * k)) return DR_E_NODEVICE;
    u64 *pp = S > 1 ? part.p : nullptr;
    void *args[] = { &vecp, &N, &qp, &nq, &k, &oip, &odp, &pp };
    const size_t lds = (ix->D > 256 ? (size_t)ix->D * 4 : 0) + 64 * 8;
    HIPCHK(hipLaunchKernel(ix->kern->bruteforce, dim3(nq, S), dim3(64), args, lds, ix->stream));
    if (S > 1) { hipLaunchKernelGGL(topk_merge_kernel, dim3(nq), dim3(64), 0, ix->stream, part.p, S, k, oi.p, od.p); HIPCHK(hipGetLastError()); }
    HIPCHK(hipMemcpyAsync(out_ids, oi.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ix->stream));
    if (out_dist) HIPCHK(hipMemcpyAsync(out_dist, od.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    oi.release(); od.release();
    return 0;
}

extern "C" int dr_get_node(dr_index *ix, uint64_t node_id, float *out_vec, uint32_t *out_nbrs)
{
    if (!ix || !out_vec || !out_nbrs) return fail(DR_E_ARG, "null argument");
    if (node_id >= ix->N) return fail(DR_E_ARG, "node id out of range");
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "dr_get_node"); if (rcv) return rcv; }
    HIPCHK(hipSetDevice(ix->device));
    std::vector<float> tmp(ix->D);
    HIPCHK(hipMemcpy(tmp.data(), ix->vecp.p + node_id * ix->D, (size_t)ix->D * 4, hipMemcpyDefault));     // (HBM or the host tier)
    for (uint32_t e = 0; e < ix->D; e++) out_vec[e] = tmp[ix->h_perm[e]];
    HIPCHK(hipMemcpy(out_nbrs, ix->adj.p + node_id * ix->R, (size_t)ix->R * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ------------------------------------------------------------------------------------------------ builder

extern "C" int dr_index_create_empty(dr_index **out, const float *vectors, uint64_t N, uint32_t D, uint32_t R, int device)
{
    return dr_index_create_empty_tiered(out, vectors, N, D, R, device, DR_TIER_HBM);
}

extern "C" int dr_index_create_empty_tiered(dr_index **out, const float *vectors, uint64_t N, uint32_t D, uint32_t R, int device,
                                            uint32_t vector_tier)
{
    if (!out) return fail(DR_E_ARG, "null argument");
    dr_index *ix = new dr_index();
    int rc = index_alloc_common(ix, N, D, R, 0, device, true, vector_tier);
    if (rc) { dr_index_close(ix); return rc; }
    DevBuf<uint32_t> staging;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / (D * 4));
    // (vectors == NULL: the rows arrive later, chunk by chunk, through dr_index_write_rows -- an index whose rows do not fit in
    // host memory twice, or are generated / read as a stream)
    for (uint64_t r0 = 0; vectors && r0 < N && !rc; r0 += chunk) {
        const uint64_t rows = std::min(chunk, N - r0);
        rc = ingest_chunk(ix, vectors + (size_t)r0 * D, r0, rows, D, false, staging);
    }
    staging.release();
    if (!rc && hipMemset(ix->adj.p, 0xFF, (size_t)N * R * 4) != hipSuccess) rc = fail(DR_E_NODEVICE, "memset failed");
    if (!rc && hipMemset(ix->first.p, 0, (size_t)N * ((R + 63) / 64) * 8) != hipSuccess) rc = fail(DR_E_NODEVICE, "memset failed");
    if (rc) { dr_index_close(ix); return rc; }
    *out = ix;
    return 0;
}

// rows [row0, row0 + n) of the stored vectors, from host memory (any tier): the streaming form of dr_index_create_empty's upload
extern "C" int dr_index_write_rows(dr_index *ix, const float *rows, uint64_t row0, uint64_t n)
{
    if (!ix || !rows) return fail(DR_E_ARG, "null argument");
    if (row0 > ix->N || n > ix->N - row0) return fail(DR_E_ARG, "rows [%llu, %llu) outside the index (N=%llu)", (unsigned long long)row0, (unsigned long long)(row0 + n), (unsigned long long)ix->N);
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "dr_index_write_rows"); if (rcv) return rcv; }
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }
    DevBuf<uint32_t> staging;
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / (ix->D * 4));
    int rc = 0;
    for (uint64_t r0 = 0; r0 < n && !rc; r0 += chunk)
        rc = ingest_chunk(ix, rows + (size_t)r0 * ix->D, row0 + r0, std::min(chunk, n - r0), ix->D, false, staging);
    // what was derived from the old rows is stale
    ix->rank_valid = false; ix->adjr_valid = false; ix->vec8_state = (ix->vec8_state == -1 && ix->D != 128) ? -1 : 0; ix->vec8.release();
    ix->vnorm2.release(); ix->adc_live = -1;
    return rc;
}

extern "C" int dr_get_adjacency(dr_index *ix, uint32_t *out)
{
    if (!ix || !out) return fail(DR_E_ARG, "null argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    HIPCHK(hipMemcpy(out, ix->adj.p, (size_t)ix->N * ix->R * 4, hipMemcpyDeviceToHost));
    return 0;
}

static uint64_t splitmix64(uint64_t &x)
{
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// the exact prune's multi-pick form (build_kernels.hpp): dimensions with the split row form only
static uint32_t prune_multi_enabled(uint32_t D)
{
    static const bool off = getenv("DR_PRUNE_PLAIN") != nullptr;      // (A/B: one pass over the candidates' rows per pick)
    const DimKernels *k = dr_dim_kernels((int)D);
    return (!off && k && k->prune_multi) ? 1u : 0u;
}

// centroid-pair table S[m][256][256] (8 MB at m = 32): every distance of the PQ-only builder is a sum of its entries
static int ensure_sdc(dr_index *ix)
{
    // (the table used to outlive its codebook -- dr_index_set_pq / dr_pq_encode / a codes-empty upload replaced the
    // codebook or m and a later dr_build_vamana_pq / dr_debug_prune_pq scored with the OLD table, out of bounds if m grew)
    if (ix->sdc.p && ix->sdc_gen == ix->codebook_gen) return 0;
    ix->sdc.release();
    if (ix->sdc.reserve((size_t)ix->m * 65536)) return DR_E_NODEVICE;
    ix->sdc_gen = ix->codebook_gen;
    hipLaunchKernelGGL(sdc_table_kernel, dim3(ix->m * 256), dim3(256), 0, ix->stream, ix->codebook.p, ix->m, ix->sd, ix->sdc.p);
    HIPCHK(hipGetLastError());
    return 0;
}

// prune_pq_kernel for q.npoints points: rows in registers (8 wavefronts per CU) when m is 16 or 32, in LDS otherwise
// (DR_PQ_PRUNE_LDS=1 forces the LDS form: A/B, and the check that both build the same graph).
static int launch_prune_pq(dr_index *ix, const PrunePQParams &q)
{
    static const bool force_lds = getenv("DR_PQ_PRUNE_LDS") != nullptr;
    const int m16 = (!force_lds && (ix->m == 16 || ix->m == 32)) ? (int)(ix->m / 16) : 0;
    const size_t lds = (m16 ? 0 : (size_t)ix->m * 1024) + (size_t)DR_PRUNE_PQ_MAXC * 24 + 1024 + (size_t)DR_PRUNE_PQ_MAXC * ix->m;
    const dim3 grid(std::min<unsigned>(q.npoints, (unsigned)ix->num_cu * (m16 ? 8 : 4)));
    if (m16 == 2) hipLaunchKernelGGL(prune_pq_kernel<2>, grid, dim3(64), lds, ix->stream, q);
    else if (m16 == 1) hipLaunchKernelGGL(prune_pq_kernel<1>, grid, dim3(64), lds, ix->stream, q);
    else {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&prune_pq_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(prune_pq_kernel<0>, grid, dim3(64), lds, ix->stream, q);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static int build_vamana_common(dr_index *ix, uint32_t L_build, float alpha, uint32_t passes, uint64_t seed,
                               uint32_t pad_with_zero, uint32_t max_batch, uint32_t *out_medoid, float *out_seconds, bool pq)
{
    if (!ix) return fail(DR_E_ARG, "null index");
    if (L_build == 0 || L_build > 256) return fail(DR_E_ARG, "L_build must be in 1..256");
    if (ix->R > 128) return fail(DR_E_UNSUPPORTED, "builder supports R <= 128");
    // (the PQ-only prune holds its candidates' code words in LDS: the construction list plus a full row with its slack slots)
    if (pq && L_build + ix->R + 64 > DR_PRUNE_PQ_MAXC)
        return fail(DR_E_ARG, "dr_build_vamana_pq: L_build + R + 64 = %u exceeds the prune's %d candidates", L_build + ix->R + 64, DR_PRUNE_PQ_MAXC);
    if (passes == 0) passes = 2;
    std::lock_guard<std::mutex> lk(ix->mu);
    if (!pq) { const int rcv = need_vectors(ix, "dr_build_vamana"); if (rcv) return rcv; }
    if (pq && ix->m == 0) return fail(DR_E_NOPQ, "dr_build_vamana_pq needs the code words (dr_index_create_codes_empty + dr_pq_encode_rows)");
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }
    const uint64_t N = ix->N;
    const uint32_t D = ix->D, R = ix->R;
    const uint32_t RX = R + 64;                       // slack slots for reverse edges inside one batch
    if (max_batch == 0) max_batch = 32768;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    struct EvPair { hipEvent_t &a, &b; ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } evp{ t0, t1 };   // every early return frees them
    HIPCHK(hipEventCreate(&t0)); HIPCHK(hipEventCreate(&t1));
    HIPCHK(hipEventRecord(t0, ix->stream));

    // ---- medoid: stored vector nearest to the centroid (the reference samples, cython_utils.pyx:210-263)
    if (!pq) {
        DevBuf<double> acc;
        if (acc.reserve(D, true)) return DR_E_NODEVICE;
        hipLaunchKernelGGL(column_sum_kernel, dim3(1024), dim3(128), 0, ix->stream, ix->vecp.p, N, D, acc.p);
        HIPCHK(hipGetLastError());
        std::vector<double> hacc(D);
        HIPCHK(hipMemcpyAsync(hacc.data(), acc.p, D * sizeof(double), hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        acc.release();
        // acc is in chain-major positions: upload as an already-permuted query
        std::vector<float> cq(D);
        for (uint32_t e = 0; e < D; e++) cq[e] = (float)(hacc[e] / (double)N);
        if (ix->cs->q.reserve(D) || ix->cs->qp.reserve(D)) return DR_E_NODEVICE;
        HIPCHK(hipMemcpyAsync(ix->cs->qp.p, cq.data(), D * 4, hipMemcpyHostToDevice, ix->stream));
        DevBuf<uint32_t> oi; DevBuf<float> od;
        if (oi.reserve(1) || od.reserve(1)) return DR_E_NODEVICE;
        const float *vecp = ix->vecp.p; const float *qp = ix->cs->qp.p; uint64_t NN = N; uint32_t one = 1; uint32_t *oip = oi.p; float *odp = od.p;
        // (one query: the rows are cut into slices, one wavefront each, folded by topk_merge_kernel -- same winner as one
        // wavefront over all rows: smallest distance, then smallest id)
        const uint32_t S = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(1024, N / 4096));
        DevBuf<u64> part;
        if (S > 1 && part.reserve(S)) return DR_E_NODEVICE;
        u64 *pp = S > 1 ? part.p : nullptr;
        void *args[] = { &vecp, &NN, &qp, &one, &one, &oip, &odp, &pp };
        const size_t lds = (D > 256 ? (size_t)D * 4 : 0) + 64 * 8;
        HIPCHK(hipLaunchKernel(ix->kern->bruteforce, dim3(1, S), dim3(64), args, lds, ix->stream));
        if (S > 1) { hipLaunchKernelGGL(topk_merge_kernel, dim3(1), dim3(64), 0, ix->stream, part.p, S, 1u, oi.p, od.p); HIPCHK(hipGetLastError()); }
        uint32_t med = 0;
        HIPCHK(hipMemcpyAsync(&med, oi.p, 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        oi.release(); od.release();
        ix->medoid = med;
    } else {
        // PQ-only: the mean of the decoded vectors follows from the per-sub-quantiser histogram of the code words; the
        // medoid is the code word nearest to it (flat ADC scan)
        DevBuf<uint32_t> hist;
        if (hist.reserve((size_t)ix->m * 256, true)) return DR_E_NODEVICE;
        hipLaunchKernelGGL(code_histogram_kernel, dim3(4096), dim3(256), 0, ix->stream, ix->codes.p, N, ix->m, hist.p);
        HIPCHK(hipGetLastError());
        std::vector<uint32_t> hh((size_t)ix->m * 256);
        std::vector<float> hcb((size_t)256 * D), mean(D, 0.0f);
        HIPCHK(hipMemcpyAsync(hh.data(), hist.p, hh.size() * 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipMemcpyAsync(hcb.data(), ix->codebook.p, hcb.size() * 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        for (uint32_t jq = 0; jq < ix->m; jq++)
            for (uint32_t t = 0; t < ix->sd; t++) {
                double a = 0;
                for (uint32_t c = 0; c < 256; c++) a += (double)hh[jq * 256 + c] * hcb[((size_t)jq * 256 + c) * ix->sd + t];
                mean[jq * ix->sd + t] = (float)(a / (double)N);
            }
        // flat scan for the nearest code word (pq_scan_kernel through the regular entry point's machinery)
        uint32_t best = 0; float bsq = 0, kms = 0;
        const int rcs = pq_scan_best_locked(ix, mean.data(), 1, nullptr, &best, &bsq, &kms);
        if (rcs) return rcs;
        ix->medoid = best;
    }
    DevBuf<uint32_t> adjb, deg, order, fwd, fwd_n, ovf_list, ovf_count;
    if (adjb.reserve((size_t)N * RX) || deg.reserve(N + 2, true) || order.reserve(N) || fwd.reserve((size_t)max_batch * R) ||
        fwd_n.reserve(max_batch) || ovf_list.reserve(N) || ovf_count.reserve(1))
        return DR_E_NODEVICE;
    HIPCHK(hipMemsetAsync(adjb.p, 0xFF, (size_t)N * RX * 4, ix->stream));
    if (!pq && (ix->cs->q.reserve((size_t)max_batch * D) || ix->cs->qp.reserve((size_t)max_batch * D))) return DR_E_NODEVICE;
    // reverse-edge pass: sorted pairs (deterministic; DR_BUILD_ATOMIC_REV=1 selects the atomic append it replaced, for A/B)
    static const bool rev_atomic = getenv("DR_BUILD_ATOMIC_REV") != nullptr;
    DevBuf<u64> rkeys_a, rkeys_b;
    DevBuf<unsigned char> rtemp;
    if (!rev_atomic) {
        const size_t np = (size_t)max_batch * R;
        if (rkeys_a.reserve(np) || rkeys_b.reserve(np)) return DR_E_NODEVICE;
        size_t tb = 0;
        HIPCHK(rocprim::radix_sort_keys(nullptr, tb, rkeys_a.p, rkeys_b.p, (unsigned int)np, 0u, 64u, ix->stream));
        if (rtemp.reserve(tb + 16)) return DR_E_NODEVICE;
    }
    if (pq) { const int rcs = ensure_sdc(ix); if (rcs) return rcs; }

    // DR_PQ_BUILD_SLACK (diagnosis): how far a row of the PQ-only builder may exceed R before it is re-pruned (default: a quarter of the slack slots: 4x fewer re-prunes at 2 points of recall, profiles/r02/scale_c5_small_4M.json)
    static const char *slack_env = getenv("DR_PQ_BUILD_SLACK");
    const uint32_t pq_slack = slack_env ? std::min<uint32_t>((uint32_t)atoi(slack_env), RX - R - 1) : (RX - R) / 4;
    // the exact prune scores the next few likely picks in one pass over the candidates' rows (D <= 256; DR_PRUNE_PLAIN=1: one pass per pick)
    const uint32_t prune_multi = prune_multi_enabled(D);
    const size_t prune_lds = (D > 256 ? (size_t)D * 4 : 0) + (size_t)DR_PRUNE_MAXC * 24 + 1024 + (prune_multi ? (size_t)4 * DR_PRUNE_MAXC * 4 : 0);
    std::vector<uint32_t> horder(N);
    uint64_t rng = seed ? seed : 1;
    int rc = 0;
    for (uint32_t pass = 0; pass < passes && !rc; pass++) {
        for (uint64_t i = 0; i < N; i++) horder[i] = (uint32_t)i;
        for (uint64_t i = N - 1; i > 0; i--) { uint64_t jx = splitmix64(rng) % (i + 1); std::swap(horder[i], horder[jx]); }
        HIPCHK(hipMemcpyAsync(order.p, horder.data(), N * 4, hipMemcpyHostToDevice, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        const float a = (pass == 0 && passes > 1) ? 1.0f : alpha;      // cython_utils.pyx:312
        uint64_t done = 0;
        uint32_t bsz = 1;
        while (done < N && !rc) {
            // batch sizes double while the graph is small (pass 0); later passes run at the cap
            uint32_t b = (pass == 0) ? std::min<uint64_t>(bsz, std::max<uint64_t>(1, done / 16 + 1)) : max_batch;
            b = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(b, max_batch), N - done);
            const uint32_t *pts = order.p + done;
            // 1. queries = the batch points' own vectors (already chain-major); PQ-only: their code words
            if (!pq) hipLaunchKernelGGL(gather_rows_kernel, dim3(b), dim3(64), 0, ix->stream, ix->vecp.p, pts, b, D, ix->cs->qp.p);
            BuildOverride ov = { adjb.p, deg.p, RX, b };
            if (pq) { ov.sdc = ix->sdc.p; ov.pts = pts; }
            rc = run_locked(ix, 1, L_build, 0, DR_MODE_M4, 0, DR_F_SQDIST, &ov);
            if (rc) break;
            // 2. prune -> forward rows
            PruneParams pp;
            pp.vecp = ix->vecp.p; pp.adjb = adjb.p; pp.deg = deg.p; pp.RX = RX; pp.R = R; pp.alpha = a;
            pp.points = pts; pp.npoints = b; pp.res_keys = ix->sets[0].res_keys.p; pp.res_n = ix->sets[0].res_n.p; pp.cap = L_build;
            pp.fwd = fwd.p; pp.fwd_n = fwd_n.p; pp.multi = prune_multi;
            PrunePQParams pq_pp;
            pq_pp.codes = ix->codes.p; pq_pp.sdc = ix->sdc.p; pq_pp.m = ix->m; pq_pp.adjb = adjb.p; pq_pp.deg = deg.p; pq_pp.RX = RX; pq_pp.R = R;
            pq_pp.alpha = a; pq_pp.points = pts; pq_pp.npoints = b; pq_pp.res_keys = pp.res_keys; pq_pp.res_n = pp.res_n; pq_pp.cap = L_build;
            pq_pp.fwd = fwd.p; pq_pp.fwd_n = fwd_n.p;
            if (pq) {
                const int rcp = launch_prune_pq(ix, pq_pp);
                if (rcp) return rcp;
            } else {
                void *args[] = { &pp };
                const unsigned g = std::min<unsigned>(b, (unsigned)ix->num_cu * 16);
                HIPCHK(hipLaunchKernel(prune_multi ? ix->kern->prune_multi : ix->kern->prune, dim3(g), dim3(64), args, prune_lds, ix->stream));
            }
            // 3. reverse edges
            HIPCHK(hipMemsetAsync(ovf_count.p, 0, 4, ix->stream));
            {
                const uint64_t threads = (uint64_t)b * R;
                // (PQ-only builder: rows run half-way into their slack before they are re-pruned, see reverse_edges_kernel)
                if (rev_atomic) {
                    hipLaunchKernelGGL(reverse_edges_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ix->stream,
                                       adjb.p, deg.p, RX, R, pts, b, fwd.p, fwd_n.p, ovf_list.p, ovf_count.p, (uint32_t)N, pq ? R + pq_slack : R);
                    HIPCHK(hipGetLastError());
                } else {
                    // sorted (target, source) pairs, one thread per target's run: deterministic (engine_kernels.hpp)
                    hipLaunchKernelGGL(rev_pairs_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ix->stream, pts, b, R, fwd.p, fwd_n.p, rkeys_a.p);
                    HIPCHK(hipGetLastError());
                    size_t tb = rtemp.n;
                    HIPCHK(rocprim::radix_sort_keys((void *)rtemp.p, tb, rkeys_a.p, rkeys_b.p, (unsigned int)threads, 0u, 64u, ix->stream));
                    hipLaunchKernelGGL(rev_apply_sorted_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ix->stream, rkeys_b.p, (uint32_t)threads,
                                       adjb.p, deg.p, RX, ovf_list.p, ovf_count.p, (uint32_t)N, pq ? R + pq_slack : R);
                    HIPCHK(hipGetLastError());
                }
            }
            uint32_t novf = 0;
            HIPCHK(hipMemcpyAsync(&novf, ovf_count.p, 4, hipMemcpyDeviceToHost, ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            // 4. re-prune rows that grew past R
            if (novf) {
                novf = (uint32_t)std::min<uint64_t>(novf, N);
                PruneParams po = pp;
                po.points = ovf_list.p; po.npoints = novf; po.res_keys = nullptr; po.res_n = nullptr; po.cap = 0;
                po.fwd = nullptr; po.fwd_n = nullptr;
                if (pq) {
                    PrunePQParams qo = pq_pp;
                    qo.points = ovf_list.p; qo.npoints = novf; qo.res_keys = nullptr; qo.res_n = nullptr; qo.cap = 0; qo.fwd = nullptr; qo.fwd_n = nullptr;
                    const int rcp = launch_prune_pq(ix, qo);
                    if (rcp) return rcp;
                } else {
                void *args[] = { &po };
                const unsigned g = std::min<unsigned>(novf, (unsigned)ix->num_cu * 16);
                HIPCHK(hipLaunchKernel(prune_multi ? ix->kern->prune_multi : ix->kern->prune, dim3(g), dim3(64), args, prune_lds, ix->stream));
                }
            }
            done += b;
            if (bsz < max_batch) bsz *= 2;
        }
    }
    if (!rc && pq) {
        // what is still over R after the last batch is pruned now
        HIPCHK(hipMemsetAsync(ovf_count.p, 0, 4, ix->stream));
        hipLaunchKernelGGL(collect_over_kernel, dim3(4096), dim3(256), 0, ix->stream, deg.p, N, R, ovf_list.p, ovf_count.p);
        HIPCHK(hipGetLastError());
        uint32_t novf = 0;
        HIPCHK(hipMemcpyAsync(&novf, ovf_count.p, 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        if (novf) {
            PrunePQParams qo;
            qo.codes = ix->codes.p; qo.sdc = ix->sdc.p; qo.m = ix->m; qo.adjb = adjb.p; qo.deg = deg.p; qo.RX = RX; qo.R = R; qo.alpha = alpha;
            qo.points = ovf_list.p; qo.npoints = novf; qo.res_keys = nullptr; qo.res_n = nullptr; qo.cap = 0; qo.fwd = nullptr; qo.fwd_n = nullptr;
            const int rcp = launch_prune_pq(ix, qo);
            if (rcp) return rcp;
        }
    }
    if (!rc) {
        const uint64_t total = N * R;
        if (R <= 128) {     // rows in a canonical order: a device-built graph is reproducible bit for bit (engine_kernels.hpp)
            const int rcr = build_rank(ix);      // (locality order of the visited bits: neighbours that share a bitmap line sit in adjacent lanes)
            if (rcr) return rcr;
            hipLaunchKernelGGL(compact_adj_sorted_kernel, dim3((unsigned)std::min<uint64_t>((N + 3) / 4, (uint64_t)ix->num_cu * 32)), dim3(256), 0, ix->stream, adjb.p,
                               deg.p, N, RX, R, pad_with_zero ? 0u : 0xFFFFFFFFu, ix->rank_valid ? ix->rank.p : nullptr, ix->adj.p);
        }
        else
            hipLaunchKernelGGL(compact_adj_kernel, dim3((unsigned)std::min<uint64_t>((total + 255) / 256, 1u << 20)), dim3(256), 0, ix->stream, adjb.p,
                               deg.p, N, RX, R, pad_with_zero ? 0u : 0xFFFFFFFFu, ix->adj.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ix->stream));
        rc = build_first_masks(ix);
    }
    HIPCHK(hipEventRecord(t1, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, t0, t1);
    adjb.release(); deg.release(); order.release(); fwd.release(); fwd_n.release(); ovf_list.release(); ovf_count.release();
    ix->cs->nq = 0;
    if (out_medoid) *out_medoid = ix->medoid;
    if (out_seconds) *out_seconds = ms / 1000.0f;
    return rc;
}

extern "C" int dr_build_vamana(dr_index *ix, uint32_t L_build, float alpha, uint32_t passes, uint64_t seed,
                               uint32_t pad_with_zero, uint32_t max_batch, uint32_t *out_medoid, float *out_seconds)
{
    return build_vamana_common(ix, L_build, alpha, passes, seed, pad_with_zero, max_batch, out_medoid, out_seconds, false);
}

// ---- PQ-only shards built on the device (BASELINE config c5): code words streamed in, graph built from code words ------
extern "C" int dr_index_create_codes_empty(dr_index **out, uint64_t N, uint32_t D, uint32_t R, const float *codebook, uint32_t m, int device)
{
    if (!out || !codebook) return fail(DR_E_ARG, "null argument");
    if (m == 0 || D % m || D / m > 128) return fail(DR_E_ARG, "bad n_subvectors %u for D=%u", m, D);
    dr_index *ix = new dr_index();
    int rc = index_alloc_common(ix, N, D, R, 0, device, false);
    if (!rc && (ix->codes.reserve((size_t)N * m, true) || ix->codebook.reserve((size_t)256 * D))) rc = DR_E_NODEVICE;
    if (!rc && hipMemcpy(ix->codebook.p, codebook, (size_t)256 * D * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(DR_E_NODEVICE, "codebook upload failed");
    if (!rc && hipMemset(ix->adj.p, 0xFF, (size_t)N * R * 4) != hipSuccess) rc = fail(DR_E_NODEVICE, "memset failed");
    if (!rc && hipMemset(ix->first.p, 0, (size_t)N * ((R + 63) / 64) * 8) != hipSuccess) rc = fail(DR_E_NODEVICE, "memset failed");
    if (rc) { dr_index_close(ix); return rc; }
    ix->m = m; ix->sd = D / m; ix->codebook_gen++;
    *out = ix;
    return 0;
}

// the code table (and codebook) of `src` copied into `dst` on the device: a second shard handle over the same points with
// another degree R -- the streamed vectors are gone, only the code words can be reused
extern "C" int dr_index_copy_codes(dr_index *dst, dr_index *src)
{
    if (!dst || !src || dst == src) return fail(DR_E_ARG, "bad argument");
    std::lock(dst->mu, src->mu);
    std::lock_guard<std::mutex> l1(dst->mu, std::adopt_lock), l2(src->mu, std::adopt_lock);
    if (dst->N != src->N || dst->D != src->D || dst->device != src->device) return fail(DR_E_ARG, "the two handles differ in N, D or device");
    if (src->m == 0 || !src->codes.p) return fail(DR_E_NOPQ, "the source holds no code words");
    HIPCHK(hipSetDevice(dst->device));
    { const int rcq = quiesce_locked(dst); if (rcq) return rcq; }
    { const int rcq = quiesce_locked(src); if (rcq) return rcq; }
    if (dst->codes.reserve((size_t)src->N * src->m) || dst->codebook.reserve((size_t)256 * src->D)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpy(dst->codes.p, src->codes.p, (size_t)src->N * src->m, hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(dst->codebook.p, src->codebook.p, (size_t)256 * src->D * 4, hipMemcpyDeviceToDevice));
    dst->m = src->m; dst->sd = src->sd; dst->codebook_gen++;
    dst->sdc.release();
    for (auto &qs : dst->slots) qs.pq_ub_valid = false;
    dst->adc_live = -1; dst->nbcodes_valid = false;
    return 0;
}

extern "C" int dr_pq_encode_rows(dr_index *ix, const float *vectors, uint64_t row0, uint64_t rows)
{
    if (!ix || !vectors) return fail(DR_E_ARG, "null argument");
    if (ix->m == 0) return fail(DR_E_NOPQ, "no codebook attached");
    if (rows == 0 || row0 + rows > ix->N) return fail(DR_E_ARG, "rows [%llu, %llu) outside the index", (unsigned long long)row0, (unsigned long long)(row0 + rows));
    std::lock_guard<std::mutex> lk(ix->mu);
    HIPCHK(hipSetDevice(ix->device));
    const uint32_t D = ix->D, m = ix->m, sd = ix->sd;
    const uint64_t chunk = std::max<uint64_t>(1, (1ull << 30) / (D * 4));
    DevBuf<float> tmp;
    if (tmp.reserve((size_t)std::min(chunk, rows) * D)) return DR_E_NODEVICE;
    const size_t lds = (size_t)256 * sd * 4;
    const void *afn = nullptr;
#define DR_PICK(SDV) afn = reinterpret_cast<const void *>(&pq_assign_rows_kernel<SDV>)
    DR_ASSIGN_SD_CASES(DR_PICK)
#undef DR_PICK
    HIPCHK(hipFuncSetAttribute(afn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (uint64_t r0 = 0; r0 < rows; r0 += chunk) {
        const uint64_t n = std::min(chunk, rows - r0);
        HIPCHK(hipMemcpyAsync(tmp.p, vectors + (size_t)r0 * D, (size_t)n * D * 4, hipMemcpyHostToDevice, ix->stream));
        const unsigned gx = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)ix->num_cu * 4);
        const float *xin = tmp.p; uint64_t nn = n; uint32_t Dv = D, mv = m, sdv = sd; const float *cbk = ix->codebook.p;
        u8 *outp = ix->codes.p + (size_t)(row0 + r0) * m;
        void *args[] = { &xin, &nn, &Dv, &mv, &sdv, &cbk, &outp };
        HIPCHK(hipLaunchKernel(afn, dim3(gx, m), dim3(256), args, lds, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
    }
    ix->adc_live = -1; ix->nbcodes_valid = false;
    return 0;
}

extern "C" int dr_build_vamana_pq(dr_index *ix, uint32_t L_build, float alpha, uint32_t passes, uint64_t seed, uint32_t max_batch,
                                  uint32_t *out_medoid, float *out_seconds)
{
    return build_vamana_common(ix, L_build, alpha, passes, seed, 0, max_batch, out_medoid, out_seconds, true);
}

// Test seam for the builder: the robust prune of ONE point over an explicit candidate list, run by the
// same prune_kernel the builder launches. The tests compare it with the CPU restatement of the reference's
// robust_prune_fast_cython (cython_utils.pyx:435-492) without that function's stale vector reads.
extern "C" int dr_debug_prune(dr_index *ix, uint32_t point, const uint32_t *candidates, uint32_t n, float alpha, uint32_t R,
                              uint32_t *out_selected, uint32_t *out_count)
{
    if (!ix || !candidates || !out_selected || !out_count) return fail(DR_E_ARG, "null argument");
    if (point >= ix->N || n == 0 || n > DR_PRUNE_MAXC || R == 0 || R > 192) return fail(DR_E_ARG, "bad prune arguments (n <= %d, R <= 192)", DR_PRUNE_MAXC);
    for (uint32_t i = 0; i < n; i++) if (candidates[i] >= ix->N) return fail(DR_E_ARG, "candidate id %u out of range", candidates[i]);
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "dr_debug_prune"); if (rcv) return rcv; }
    HIPCHK(hipSetDevice(ix->device));
    // the candidates travel as a one-query "search result" (the kernel recomputes their distances); the row itself is empty
    DevBuf<u64> keys; DevBuf<uint32_t> resn, adjb, deg, pts, fwd, fwdn;
    if (keys.reserve(n) || resn.reserve(1) || adjb.reserve((size_t)ix->N * R) || deg.reserve(ix->N + 2, true) || pts.reserve(1) ||
        fwd.reserve(R) || fwdn.reserve(1)) return DR_E_NODEVICE;
    std::vector<u64> hk(n);
    for (uint32_t i = 0; i < n; i++) hk[i] = (u64)(uint32_t)(~candidates[i]);
    HIPCHK(hipMemcpyAsync(keys.p, hk.data(), (size_t)n * 8, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(resn.p, &n, 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(pts.p, &point, 4, hipMemcpyHostToDevice, ix->stream));
    PruneParams pp;
    pp.vecp = ix->vecp.p; pp.adjb = adjb.p; pp.deg = deg.p; pp.RX = R; pp.R = R; pp.alpha = alpha;
    pp.points = pts.p; pp.npoints = 1; pp.res_keys = keys.p; pp.res_n = resn.p; pp.cap = n; pp.fwd = fwd.p; pp.fwd_n = fwdn.p;
    pp.multi = prune_multi_enabled(ix->D);
    void *args[] = { &pp };
    const size_t prune_lds = (ix->D > 256 ? (size_t)ix->D * 4 : 0) + (size_t)DR_PRUNE_MAXC * 24 + 1024 + (pp.multi ? (size_t)4 * DR_PRUNE_MAXC * 4 : 0);
    HIPCHK(hipLaunchKernel(pp.multi ? ix->kern->prune_multi : ix->kern->prune, dim3(1), dim3(64), args, prune_lds, ix->stream));
    HIPCHK(hipMemcpyAsync(out_selected, fwd.p, (size_t)R * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipMemcpyAsync(out_count, fwdn.p, 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    return 0;
}

// The same seam for the PQ-only builder's prune (prune_pq_kernel, whichever form serves this m): candidates scored by sums of
// centroid-pair table entries (A3's order), sorted by (distance, id), greedy picks, alpha * d(p*, c) <= d(p, c) drops c.
extern "C" int dr_debug_prune_pq(dr_index *ix, uint32_t point, const uint32_t *candidates, uint32_t n, float alpha, uint32_t R,
                                 uint32_t *out_selected, uint32_t *out_count)
{
    if (!ix || !candidates || !out_selected || !out_count) return fail(DR_E_ARG, "null argument");
    if (point >= ix->N || n == 0 || n > DR_PRUNE_PQ_MAXC || R == 0 || R > 128)
        return fail(DR_E_ARG, "bad prune arguments (n <= %d, R <= 128)", DR_PRUNE_PQ_MAXC);
    for (uint32_t i = 0; i < n; i++) if (candidates[i] >= ix->N) return fail(DR_E_ARG, "candidate id %u out of range", candidates[i]);
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->m == 0) return fail(DR_E_NOPQ, "dr_debug_prune_pq needs the code words");
    HIPCHK(hipSetDevice(ix->device));
    { const int rcs = ensure_sdc(ix); if (rcs) return rcs; }
    DevBuf<u64> keys; DevBuf<uint32_t> resn, adjb, deg, pts, fwd, fwdn;
    if (keys.reserve(n) || resn.reserve(1) || adjb.reserve((size_t)ix->N * R) || deg.reserve(ix->N + 2, true) || pts.reserve(1) ||
        fwd.reserve(R) || fwdn.reserve(1)) return DR_E_NODEVICE;
    std::vector<u64> hk(n);
    for (uint32_t i = 0; i < n; i++) hk[i] = (u64)(uint32_t)(~candidates[i]);
    HIPCHK(hipMemcpyAsync(keys.p, hk.data(), (size_t)n * 8, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(resn.p, &n, 4, hipMemcpyHostToDevice, ix->stream));
    HIPCHK(hipMemcpyAsync(pts.p, &point, 4, hipMemcpyHostToDevice, ix->stream));
    PrunePQParams q;
    q.codes = ix->codes.p; q.sdc = ix->sdc.p; q.m = ix->m; q.adjb = adjb.p; q.deg = deg.p; q.RX = R; q.R = R; q.alpha = alpha;
    q.points = pts.p; q.npoints = 1; q.res_keys = keys.p; q.res_n = resn.p; q.cap = n; q.fwd = fwd.p; q.fwd_n = fwdn.p;
    { const int rcp = launch_prune_pq(ix, q); if (rcp) return rcp; }
    HIPCHK(hipMemcpyAsync(out_selected, fwd.p, (size_t)R * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipMemcpyAsync(out_count, fwdn.p, 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ PQ build

static int pq_assign(dr_index *ix, const uint32_t *d_ids, uint64_t n, uint32_t m, const float *d_codebook, uint8_t *d_out)
{
    const uint32_t sd = ix->D / m;
    const size_t lds = (size_t)256 * sd * 4;
    const void *afn = nullptr;
#define DR_PICK(SDV) afn = reinterpret_cast<const void *>(&pq_assign_kernel<SDV>)
    DR_ASSIGN_SD_CASES(DR_PICK)
#undef DR_PICK
    HIPCHK(hipFuncSetAttribute(afn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned gx = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)ix->num_cu * 4);
    const float *vp = ix->vecp.p; const u32 *pm = ix->perm.p; uint64_t nn = n; uint32_t Dv = ix->D, mv = m, sdv = sd;
    void *args[] = { &vp, &pm, &d_ids, &nn, &Dv, &mv, &sdv, &d_codebook, &d_out };
    HIPCHK(hipLaunchKernel(afn, dim3(gx, m), dim3(256), args, lds, ix->stream));
    return 0;
}

// DiskANNPQ.fit (pq/fast_pq.py:188-243) runs m independent sklearn KMeans(256, init='k-means++', n_init, max_iter, tol).
// Everything runs on the device (engine_kernels.hpp, "k-means on the device"): greedy k-means++ seeding with sklearn's
// 2 + log k local trials (one workgroup per sub-quantiser, uniform numbers from the host's splitmix64 stream), Lloyd
// iterations = nearest-centre assignment + fixed-point centre sums + centre update, sklearn's stopping rule (total squared
// centre shift <= tol * mean per-feature variance), n_init restarts keeping the lowest inertia per sub-quantiser. The host
// only reads m shifts per iteration and m inertias per restart. Deterministic for a given seed.
extern "C" int dr_pq_train_ex(dr_index *ix, uint32_t m, uint32_t n_sample, uint32_t max_iter, uint32_t n_init, float tol, uint64_t seed,
                              float *out_codebook, double *out_inertia)
{
    if (!ix || !out_codebook) return fail(DR_E_ARG, "null argument");
    if (m == 0 || ix->D % m || ix->D / m > 128) return fail(DR_E_ARG, "bad n_subvectors %u for D=%u", m, ix->D);
    if (ix->N < 256) return fail(DR_E_ARG, "need at least 256 vectors (fast_pq.py:212-213)");
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "dr_pq_train"); if (rcv) return rcv; }
    HIPCHK(hipSetDevice(ix->device));
    const uint32_t D = ix->D, sd = D / m;
    const uint32_t ns = (uint32_t)std::min<uint64_t>(n_sample ? n_sample : 100000, ix->N);
    if (max_iter == 0) max_iter = 10;
    if (n_init == 0) n_init = 1;
    // sample without replacement (partial Fisher-Yates over ids)
    std::vector<uint32_t> ids(ns);
    {
        uint64_t rng = seed ? seed : 42;
        if (ns == ix->N) { for (uint32_t i = 0; i < ns; i++) ids[i] = i; }
        else {
            std::vector<uint32_t> all(ix->N);
            for (uint64_t i = 0; i < ix->N; i++) all[i] = (uint32_t)i;
            for (uint32_t i = 0; i < ns; i++) { uint64_t jx = i + splitmix64(rng) % (ix->N - i); std::swap(all[i], all[jx]); ids[i] = all[i]; }
        }
    }
    DevBuf<uint32_t> d_ids, d_counts; DevBuf<float> d_x, d_xt, d_cb, d_d2, d_maxabs; DevBuf<uint8_t> d_assign;
    DevBuf<double> d_var, d_unif, d_shift, d_inertia; DevBuf<int> d_fix; DevBuf<unsigned long long> d_sums;
    if (d_ids.reserve(ns) || d_x.reserve((size_t)ns * D) || d_xt.reserve((size_t)ns * D) || d_cb.reserve((size_t)256 * D) || d_assign.reserve((size_t)ns * m) ||
        d_d2.reserve((size_t)ns * m) || d_maxabs.reserve(m) || d_var.reserve(m) || d_unif.reserve((size_t)m * 256 * 8) || d_shift.reserve(m) ||
        d_inertia.reserve(m) || d_fix.reserve(m) || d_sums.reserve((size_t)256 * D, true) || d_counts.reserve((size_t)m * 256, true))
        return DR_E_NODEVICE;
    HIPCHK(hipMemcpyAsync(d_ids.p, ids.data(), (size_t)ns * 4, hipMemcpyHostToDevice, ix->stream));
    hipLaunchKernelGGL(gather_subvectors_kernel, dim3(ns), dim3(64), 0, ix->stream, ix->vecp.p, ix->perm.p, d_ids.p, ns, D, d_x.p);
    HIPCHK(hipGetLastError());
    // the seeding kernel reads the sample by columns (km_transpose_kernel)
    hipLaunchKernelGGL(km_transpose_kernel, dim3((ns + 31) / 32, (D + 31) / 32), dim3(256), 0, ix->stream, d_x.p, ns, D, d_xt.p);
    HIPCHK(hipGetLastError());
    // sklearn's stopping rule: total squared centre shift <= tol * mean per-feature variance (KMeans tol, default 1e-4);
    // the fixed-point scale of the centre sums: the largest exponent that cannot overflow 63 bits over ns terms
    hipLaunchKernelGGL(km_stats_kernel, dim3(m), dim3(DR_KM_THREADS), 0, ix->stream, d_x.p, ns, D, sd, d_var.p, d_maxabs.p);
    HIPCHK(hipGetLastError());
    std::vector<double> var_mean(m);
    std::vector<float> max_abs(m);
    HIPCHK(hipMemcpyAsync(var_mean.data(), d_var.p, (size_t)m * 8, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipMemcpyAsync(max_abs.data(), d_maxabs.p, (size_t)m * 4, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    std::vector<double> tol_j(m);
    std::vector<int> fix(m);
    for (uint32_t jq = 0; jq < m; jq++) {
        tol_j[jq] = (double)tol * var_mean[jq];
        int e = 0;
        (void)std::frexp((double)max_abs[jq] * (double)ns + 1.0, &e);      // value < 2^e
        fix[jq] = 61 - e;
    }
    HIPCHK(hipMemcpyAsync(d_fix.p, fix.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, ix->stream));
    const size_t acc_lds_full = (size_t)256 * sd * 8 + 1024;
    const int use_lds = acc_lds_full <= 128 * 1024 ? 1 : 0;
    const size_t acc_lds = use_lds ? acc_lds_full : 1024;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&km_accumulate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)acc_lds));
    const unsigned acc_gx = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>((ns + 255) / 256, std::max<uint32_t>(1, (uint32_t)ix->num_cu / m)));

    std::vector<float> cb((size_t)m * 256 * sd), best_cb((size_t)m * 256 * sd);
    std::vector<double> best_inertia(m, -1.0), inertia(m), shift(m), unif((size_t)m * 256 * 8);
    for (uint32_t r = 0; r < n_init; r++) {
        // the restart's random numbers: per sub-quantiser its own splitmix64 stream (seed, restart, jq)
        for (uint32_t jq = 0; jq < m; jq++) {
            uint64_t rng = (seed ? seed : 42) * 0x9E3779B97F4A7C15ull + ((uint64_t)r << 32) + jq + 1;
            for (uint32_t e = 0; e < 256 * 8; e++) unif[(size_t)jq * 2048 + e] = (double)(splitmix64(rng) >> 11) * (1.0 / 9007199254740992.0);
        }
        HIPCHK(hipMemcpyAsync(d_unif.p, unif.data(), unif.size() * 8, hipMemcpyHostToDevice, ix->stream));
        hipLaunchKernelGGL(kmeanspp_kernel, dim3(m), dim3(DR_KM_THREADS), 0, ix->stream, d_xt.p, ns, D, sd, d_unif.p, d_d2.p, d_cb.p);
        HIPCHK(hipGetLastError());
        for (uint32_t it = 0; it < max_iter; it++) {
            int rc = pq_assign(ix, d_ids.p, ns, m, d_cb.p, d_assign.p);
            if (rc) return rc;
            hipLaunchKernelGGL(km_accumulate_kernel, dim3(acc_gx, m), dim3(256), acc_lds, ix->stream, d_x.p, d_assign.p, ns, D, m, sd, d_fix.p,
                               d_sums.p, d_counts.p, use_lds);
            hipLaunchKernelGGL(km_finalize_kernel, dim3(m), dim3(256), 0, ix->stream, d_cb.p, sd, d_fix.p, d_sums.p, d_counts.p, d_shift.p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(shift.data(), d_shift.p, (size_t)m * 8, hipMemcpyDeviceToHost, ix->stream));
            HIPCHK(hipStreamSynchronize(ix->stream));
            bool done = true;
            for (uint32_t jq = 0; jq < m; jq++) done = done && shift[jq] <= tol_j[jq];
            if (done) break;
        }
        // labels and inertia of the final centres
        const int rc = pq_assign(ix, d_ids.p, ns, m, d_cb.p, d_assign.p);
        if (rc) return rc;
        hipLaunchKernelGGL(km_inertia_kernel, dim3(m), dim3(DR_KM_THREADS), 0, ix->stream, d_x.p, d_assign.p, ns, D, m, sd, d_cb.p, d_inertia.p);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(inertia.data(), d_inertia.p, (size_t)m * 8, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipMemcpyAsync(cb.data(), d_cb.p, cb.size() * 4, hipMemcpyDeviceToHost, ix->stream));
        HIPCHK(hipStreamSynchronize(ix->stream));
        for (uint32_t jq = 0; jq < m; jq++)
            if (best_inertia[jq] < 0 || inertia[jq] < best_inertia[jq]) {
                best_inertia[jq] = inertia[jq];
                memcpy(&best_cb[(size_t)jq * 256 * sd], &cb[(size_t)jq * 256 * sd], (size_t)256 * sd * 4);
            }
    }
    memcpy(out_codebook, best_cb.data(), best_cb.size() * 4);
    if (out_inertia) { double tot = 0; for (double v : best_inertia) tot += v; *out_inertia = tot; }
    return 0;
}

extern "C" int dr_pq_train(dr_index *ix, uint32_t m, uint32_t n_sample, uint32_t iters, uint64_t seed, float *out_codebook)
{
    return dr_pq_train_ex(ix, m, n_sample, iters ? iters : 10, 1, 1e-4f, seed, out_codebook, nullptr);
}

extern "C" int dr_pq_encode(dr_index *ix, const float *codebook, uint32_t m, uint8_t *out_codes)
{
    if (!ix || !codebook) return fail(DR_E_ARG, "null argument");
    if (m == 0 || ix->D % m || ix->D / m > 128) return fail(DR_E_ARG, "bad n_subvectors %u for D=%u", m, ix->D);
    std::lock_guard<std::mutex> lk(ix->mu);
    { const int rcv = need_vectors(ix, "dr_pq_encode"); if (rcv) return rcv; }
    HIPCHK(hipSetDevice(ix->device));
    { const int rcq = quiesce_locked(ix); if (rcq) return rcq; }
    if (ix->codes.reserve((size_t)ix->N * m) || ix->codebook.reserve((size_t)256 * ix->D)) return DR_E_NODEVICE;
    HIPCHK(hipMemcpyAsync(ix->codebook.p, codebook, (size_t)256 * ix->D * 4, hipMemcpyHostToDevice, ix->stream));
    int rc = pq_assign(ix, nullptr, ix->N, m, ix->codebook.p, ix->codes.p);
    if (rc) return rc;
    if (out_codes) HIPCHK(hipMemcpyAsync(out_codes, ix->codes.p, (size_t)ix->N * m, hipMemcpyDeviceToHost, ix->stream));
    HIPCHK(hipStreamSynchronize(ix->stream));
    ix->m = m; ix->sd = ix->D / m; ix->codebook_gen++;
    for (auto &qs : ix->slots) qs.pq_ub_valid = false;
    ix->adc_live = -1; ix->nbcodes_valid = false;
    return 0;
}

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
    if (kind > DR_MAX_KIND_ID || (kind >= 0 && dr_kind_pos(kind) < 0)) return fail(DR_E_ARG, "unknown kernel variant %d", kind);
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
