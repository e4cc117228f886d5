// engine_kernels.hpp -- dimension-independent kernels: ingest (record split + chain-major permutation), adjacency
// masks, result finalisation (heap replay for the reference's tie order), PQ table / ADC entry points.
// Included by engine.hip only.
#pragma once
#include "search_kernel.hpp"

// ---- ingest ---------------------------------------------------------------------------------------------
// index.dat record i = D float32 then R uint32 (diskann_persist.py:17-24). rec_words = D + R when the source is
// a raw record buffer; vectors-only sources pass rec_words = D and adj_off = 0 with adj_out = nullptr.
__global__ void ingest_records_kernel(const u32 *__restrict__ rec, u64 n, u32 D, u32 R, u32 rec_words,
                                      const u32 *__restrict__ perm, float *__restrict__ vecp_out,
                                      u32 *__restrict__ adj_out)
{
    const u64 row = blockIdx.x;
    if (row >= n) return;
    const u32 *src = rec + row * rec_words;
    for (u32 e = threadIdx.x; e < D; e += blockDim.x) vecp_out[row * D + perm[e]] = __uint_as_float(src[e]);
    if (adj_out)
        for (u32 s = threadIdx.x; s < R; s += blockDim.x) adj_out[row * R + s] = src[D + s];
}

__global__ void permute_queries_kernel(const float *__restrict__ q, u32 nq, u32 D, const u32 *__restrict__ perm,
                                       float *__restrict__ qp)
{
    const u32 row = blockIdx.x;
    if (row >= nq) return;
    for (u32 e = threadIdx.x; e < D; e += blockDim.x) qp[(size_t)row * D + perm[e]] = q[(size_t)row * D + e];
}

// first[row][w] bit s: slot 64w+s holds a real id (not DR_PAD) that did not occur in an earlier slot of the row.
// The reference walks a row in stored order and skips ids already in `visited` (search_engine.py:444-448), so a
// repeated id (notably several 0 pads, Q3) is scored at its first position only. bad[0] counts ids >= N.
__global__ void first_mask_kernel(const u32 *__restrict__ adj, u64 n, u32 R, u64 N, u64 *__restrict__ first,
                                  u32 *__restrict__ bad)
{
    // grid-stride over the rows: a launch is limited to 2^32 - 1 work-items, N * 64 exceeds that from N = 6.7e7 on
    const u32 lane = threadIdx.x & 63;
    const u32 nw = (R + 63) / 64;
    for (u64 row = (u64)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); row < n; row += (u64)gridDim.x * (blockDim.x / 64)) {
    const u32 *a = adj + row * R;
    for (u32 w = 0; w < nw; w++) {
        const u32 s = w * 64 + lane;
        const u32 id = (s < R) ? a[s] : 0xFFFFFFFFu;
        bool ok = (s < R) && id != 0xFFFFFFFFu;
        if (ok && id >= N) { atomicAdd(bad, 1u); ok = false; }
        // earlier occurrence in previous words
        for (u32 t = 0; t < w * 64 && ok; t++) if (a[t] == id) ok = false;
        // earlier occurrence inside this word
        for (u32 t = 0; t < 64; t++) {
            const u32 other = __shfl(id, t);
            if (t < lane && other == id) ok = false;
        }
        const u64 m = __ballot(ok);
        if (lane == 0) first[row * nw + w] = m;
    }
    }
}

// ---- finalize -------------------------------------------------------------------------------------------
// The reference returns `results` (a heapq array of (-dist, id)) after a STABLE sort on distance only
// (search_engine.py:483-488; vamana_graph.py:596-598 for M3 with key sqrt(d); :640 for M4), so equal
// distances come out in heap-array order. The search kernel keeps the list sorted and writes the answer itself;
// only queries whose first k entries hold equal sort keys are listed for this kernel, which replays the heap
// from the accepted-insert log with CPython's exact sift rules (Lib/heapq.py) to recover that order.
// M2 sorts full (dist, id) tuples (vamana_graph.py:758): ids ascending inside a tie, no replay.
// One WAVEFRONT per listed query: the heap array lives in registers (element i in lane i%64 of chunk i/64), the
// sift indices are scalars, so every step is a v_readlane / masked move with no memory traffic.
struct FinalizeParams {
    const u64 *res_keys; const u32 *res_n; const u32 *tie_list; const u32 *tie_count; const u64 *log;
    const KStats *stats;
    u32 logcap, cap, k, mode;
    u32 *out_ids; float *out_dist;
    u32 *ntie_stat;     // largest tie-list length seen since the host last looked (sizes the next launches)
    u32 serial;         // 1: the one-lane replay at every capacity (DR_FINALIZE_SERIAL=1, A/B)
};

// python tuple (-d, id) "less than" on keys (dist bits << 32 | ~id): x < y  <=>  key(x) > key(y)
DEV bool py_lt(u64 x, u64 y) { return x > y; }

// Lib/heapq.py _siftdown / _siftup on a heap array in LDS, run by one lane (the array accesses of a sift step are
// independent LDS reads: ~1.4 k cycles per push + pop at capacity 100, against ~4.3 k for the same steps on a
// register-resident heap driven through v_readlane / masked moves)
DEV void lds_siftdown(u64 *H, int startpos, int pos)
{
    const u64 newitem = H[pos];
    while (pos > startpos) {
        const int parentpos = (pos - 1) >> 1;
        const u64 parent = H[parentpos];
        if (py_lt(newitem, parent)) { H[pos] = parent; pos = parentpos; continue; }
        break;
    }
    H[pos] = newitem;
}
DEV void lds_siftup(u64 *H, int n, int pos)
{
    const int endpos = n, startpos = pos;
    const u64 newitem = H[pos];
    int childpos = 2 * pos + 1;
    while (childpos < endpos) {
        const int rightpos = childpos + 1;
        const u64 cl = H[childpos], cr = H[min(rightpos, endpos - 1)];
        u64 cv = cl;
        if (rightpos < endpos && !py_lt(cl, cr)) { childpos = rightpos; cv = cr; }
        H[pos] = cv;
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    H[pos] = newitem;
    lds_siftdown(H, startpos, pos);
}

DEV float sort_key(u64 key, u32 mode)
{
    const float d = key_dist(key);
    return mode == 3u ? f_sqrt(d) : d;
}

// ---- the same two heap operations by the WHOLE wavefront (capacity <= 128: at most 64 internal nodes, 8 levels) --------------------------
// A sift is a chain of dependent LDS reads when one lane walks it (~1 000 cycles per push + pop at capacity 100, and a tied query replays
// ~750 of them: the 0.6 ms a blocking call waits for). The moves of a sift are decided by comparisons that do not depend on each other:
//   heappush  appends at index h and moves the new item up past every ancestor it is smaller than. The ancestors a_k = ((h + 1) >> k) - 1 are
//             known up front and ordered along the path (heap property), so "smaller than" holds for a PREFIX of them: lane k reads ancestor k,
//             one ballot counts the prefix t, lanes 1..t move their ancestor one step down the path, the item lands at a_t.
//   heappop   (heap[0] = the former last item, _siftup(0)) walks the hole down the smaller-child path to a leaf -- whatever the item is -- then
//             moves the item back up past the path elements it is smaller than. Lane i picks the smaller child of internal node i (all nodes at
//             once: one LDS round trip), the path is followed through those choices with v_readlane (no memory), lane k reads path element k,
//             and again a prefix count says how far the elements shift up and where the item lands.
// Same comparisons on the same values as Lib/heapq.py, so the same array after every operation (tests: every tied fixture query against the
// reference's order; DR_FINALIZE_SERIAL=1 selects the one-lane form for A/B).
// returns what index h holds afterwards (the item itself, or the parent it displaced): the entry a heappop that follows takes off the end
DEV u64 wave_heappush(u64 *H, int h, u64 item)
{
    const int lane = lane_id();
    const int a = (int)(((u32)h + 1u) >> min(lane, 31)) - 1;        // ancestor `lane` levels up (a < 0: past the root)
    const bool on = lane >= 1 && lane < 31 && a >= 0;
    const u64 anc = on ? H[a] : 0ull;
    const int t = __popcll(__ballot(on && py_lt(item, anc)));       // ancestors the item passes (a prefix of the path)
    const int below = (int)(((u32)h + 1u) >> min(max(lane - 1, 0), 31)) - 1;          // a_{lane - 1}
    if (on && lane <= t) H[below] = anc;
    if (lane == t) H[lane == 0 ? h : a] = item;
    WSYNC();
    return t == 0 ? item : readlane64(anc, 1);
}
// n = entries left after the last one was taken off (the caller read it: `last`); n >= 1
DEV void wave_heappop(u64 *H, int n, u64 last)
{
    const int lane = lane_id();
    int pref = -1;
    {
        const int l = 2 * lane + 1, r = 2 * lane + 2;
        if (l < n) {
            const u64 cl = H[l], cr = H[min(r, n - 1)];
            pref = (r < n && !py_lt(cl, cr)) ? r : l;
        }
    }
    int pcur = 0, depth = 0, myp = 0;
#pragma unroll
    for (int kk = 1; kk <= 7; kk++) {
        if (2 * pcur + 1 < n) {                                     // pcur is an internal node (index < 64 at capacity <= 128)
            pcur = __builtin_amdgcn_readlane(pref, __builtin_amdgcn_readfirstlane(pcur));
            depth = kk;
            myp = (lane == kk) ? pcur : myp;
        }
    }
    const bool on = lane >= 1 && lane <= depth;
    const u64 el = on ? H[myp] : 0ull;
    const int up = __popcll(__ballot(on && !py_lt(last, el)));      // path elements that move up one level (a prefix)
    const int above = (int)wave_shr1_u32((u32)myp, 0u);             // lane k: path node k - 1
    WSYNC();
    if (on && lane <= up) H[above] = el;
    if (lane == up) H[myp] = last;
    WSYNC();
}

// One wavefront per listed query; the heap array (cap + 1 entries) and a 64-entry slice of the insert log live in
// LDS. The whole wavefront replays the log (capacity <= 128; lane 0 alone above that), reads the log and looks keys up in the final array.
__global__ __launch_bounds__(256) void finalize_kernel(const FinalizeParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int lane = lane_id();
    const u32 hcap = p.cap + 2;
    u64 *H = reinterpret_cast<u64 *>(fsm) + (size_t)(threadIdx.x >> 6) * (hcap + 64);
    u64 *stage = H + hcap;
    const u32 ntie = *p.tie_count;
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.ntie_stat) atomicMax(p.ntie_stat, ntie);
    const u32 nwaves = gridDim.x * (blockDim.x >> 6);
    const u32 wave0 = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    for (u32 t = wave0; t < ntie; t += nwaves) {
        const u32 q = p.tie_list[t];
        const int n = (int)p.res_n[q];
        const int cnt = n < (int)p.k ? n : (int)p.k;
        const u64 *keys = p.res_keys + (size_t)q * p.cap;
        u32 *oid = p.out_ids + (size_t)q * p.k;
        float *od = p.out_dist + (size_t)q * p.k;
        int hn = 0;
        if (p.mode != 2u) {
            const u32 nins = p.stats[q].inserts;
            if (nins > p.logcap) continue;   // log overflowed (status bit is already set): order stays as is
            const u64 *lg = p.log + (size_t)q * p.logcap;
            WSYNC();
            for (u32 i0 = 0; i0 < nins; i0 += 64) {
                stage[lane] = (i0 + lane < nins) ? lg[i0 + lane] : 0ull;      // one coalesced read of 64 log entries
                WSYNC();
                const int lim = (int)min(64u, nins - i0);
                if (p.cap <= 128u && !p.serial) {
                    int h = (int)min(i0, p.cap);
                    for (int u = 0; u < lim; u++) {
                        const u64 e = stage[u];
                        const u64 last = wave_heappush(H, h, (e & 0xFFFFFFFF00000000ull) | (u32)(~(u32)e));
                        h++;
                        if (h > (int)p.cap) {
                            --h;
                            if (h) wave_heappop(H, h, last);
                        }
                    }
                } else if (lane == 0) {
                    int h = (int)min(i0, p.cap);
                    for (int u = 0; u < lim; u++) {
                        const u64 e = stage[u];
                        H[h] = (e & 0xFFFFFFFF00000000ull) | (u32)(~(u32)e);
                        h++;
                        lds_siftdown(H, 0, h - 1);                           // heappush
                        if (h > (int)p.cap) {                                // heappop
                            const u64 last = H[--h];
                            if (h) { H[0] = last; lds_siftup(H, h, 0); }
                        }
                    }
                }
                WSYNC();
            }
            hn = (int)min(nins, p.cap);
        }
        // walk equal-key groups that intersect the first cnt positions
        int i = 0;
        while (i < cnt) {
            const float ki = sort_key(keys[i], p.mode);
            int e = i + 1;
            while (e < n && sort_key(keys[e], p.mode) == ki) e++;
            if (e - i > 1) {
                if (p.mode == 2u) {
                    // full tuple order: id ascending; the list holds equal distances with id descending
                    if (lane == 0) for (int u = 0; i + u < cnt && u < e - i; u++) oid[i + u] = ~(u32)keys[e - 1 - u];
                } else {
                    // heap-array order: pick group members by increasing heap index
                    int last = -1;
                    for (int pos = i; pos < cnt && pos < e; pos++) {
                        int best = 0x7FFFFFFF; u64 bestkey = 0;
                        for (int g = i; g < e; g++) {
                            const u64 kg = keys[g];
                            int hi = -1;
                            for (int c0 = 0; c0 < hn && hi < 0; c0 += 64) {
                                const u64 m = __ballot(c0 + lane < hn && H[min(c0 + lane, hn - 1)] == kg);
                                if (m != 0ull) hi = c0 + __ffsll((long long)m) - 1;
                            }
                            if (hi > last && hi < best) { best = hi; bestkey = kg; }
                        }
                        if (best == 0x7FFFFFFF) break;
                        if (lane == 0) { oid[pos] = ~(u32)bestkey; od[pos] = sort_key(bestkey, p.mode); }
                        last = best;
                    }
                }
            }
            i = e;
        }
    }
}

// ---- A2 for a whole batch: T[q][j][c] = sum((C_j[c] - q_j)^2) in numpy's order (fast_pq.py:294-318) ----------------------
// The per-query-table search variants used to build their table inside the search kernel: one wavefront, 4-8 wavefronts
// per CU (the table itself caps them), 128 dependent round trips to the codebook at D = 1536 -- a quarter of the c3 kernel
// and 40 % of the PQ-only traversal. The table does not depend on the traversal, so it is built for the whole batch by
// this kernel at full occupancy: a workgroup owns one sub-quantiser j, thread c keeps centroid c of j in registers, and
// walks the batch's queries -- the query's sub-vector is wave-uniform (scalar loads, SGPR operands), the centroid never
// moves, every store is a coalesced KiB. Same operations in the same order as build_lut_sd (pw_run_regs): the same bits.
// 10 000 queries x 1536 (m = 32): 1.2e10 lane operations and 320 MB written.
template <int SD>
__global__ __launch_bounds__(256) void lut_build_kernel(const float *__restrict__ codebook, const float *__restrict__ queries,
                                                        u32 nq, u32 D, u32 m, float *__restrict__ out)
{
    const u32 jq = blockIdx.y, c = threadIdx.x;
    float cen[SD];
    const float *src = codebook + ((size_t)jq * 256 + c) * SD;
    if constexpr (SD % 4 == 0) {
#pragma unroll
        for (int i = 0; i < SD; i += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(src + i);
            cen[i] = v.x; cen[i + 1] = v.y; cen[i + 2] = v.z; cen[i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SD; i++) cen[i] = src[i];
    }
    for (u32 q = blockIdx.x; q < nq; q += gridDim.x) {
        const float *qs = queries + (size_t)q * D + jq * SD;       // uniform over the workgroup
        out[((size_t)q * m + jq) * 256 + c] = pw_run_regs<SD>(cen, qs);
    }
}
// any other sub_dim (<= 128): the centroid is read from global memory (L2) per query
__global__ __launch_bounds__(256) void lut_build_generic_kernel(const float *__restrict__ codebook, const float *__restrict__ queries,
                                                                u32 nq, u32 D, u32 m, u32 sd, float *__restrict__ out)
{
    const u32 jq = blockIdx.y, c = threadIdx.x;
    const float *src = codebook + ((size_t)jq * 256 + c) * sd;
    for (u32 q = blockIdx.x; q < nq; q += gridDim.x)
        out[((size_t)q * m + jq) * 256 + c] = pw_run_lane(src, queries + (size_t)q * D + jq * sd, (int)sd);
}

// ---- PQ entry points
// ADC for listed nodes (ids != nullptr) or a flat scan of all N codes (ids == nullptr): the table sits in LDS,
// each lane owns one code word and adds its m table entries in sub-quantiser order.
__global__ __launch_bounds__(256) void adc_kernel(const float *__restrict__ codebook, const float *__restrict__ queries,
        const u8 *__restrict__ codes, const u32 *__restrict__ ids, u64 n, u32 D, u32 m, u32 sd,
        float *__restrict__ out_sq, float *__restrict__ out_sqrt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *lut = q + D;
    const u32 qi = blockIdx.y;
    for (u32 i = threadIdx.x; i < D; i += blockDim.x) q[i] = queries[(size_t)qi * D + i];
    __syncthreads();
    for (u32 e = threadIdx.x; e < m * 256; e += blockDim.x)
        lut[e] = pw_run_lane(codebook + (size_t)e * sd, q + (e >> 8) * sd, (int)sd);
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 node = ids ? ids[i] : i;
        const float s = adc_lane<false>(lut, q, sd, codes + node * m, m);
        if (out_sq) out_sq[(size_t)qi * n + i] = s;
        if (out_sqrt) out_sqrt[(size_t)qi * n + i] = f_sqrt(s);
    }
}


// Flat PQ scan (the isolated ADC kernel of SURVEY 8d): every code word of the table against one query per block
// row, table in LDS, codes streamed from HBM with 16-byte loads (one code word per lane: a wavefront instruction
// covers 64 consecutive code words). The m table reads of a lane are issued together and added in sub-quantiser
// order (A3: strict sequential float sum, fast_pq.py:325-326), so the LDS pipeline -- not the latency of one read
// -- sets the pace. Optionally writes every distance; always folds (distance, id) into the query's best key, so the
// scan cannot be optimised away when the distances are not wanted.
template <int M16>
__global__ __launch_bounds__(256) void pq_scan_kernel(const float *__restrict__ codebook, const float *__restrict__ queries,
        const u8 *__restrict__ codes, u64 n, u32 D, u32 sd, float *__restrict__ out_sq, u64 *__restrict__ out_best)
{
    constexpr u32 m = 16u * M16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *lut = q + D;
    const u32 qi = blockIdx.y;
    for (u32 i = threadIdx.x; i < D; i += blockDim.x) q[i] = queries[(size_t)qi * D + i];
    __syncthreads();
    {
        // each of the 4 wavefronts builds the table rows of a quarter of the sub-quantisers
        const u32 w = threadIdx.x >> 6, j0 = w * (m / 4);
        build_lut_wave(lut + j0 * 256, codebook + (size_t)j0 * 256 * sd, q + j0 * sd, m / 4, sd);
    }
    __syncthreads();
    u64 best = ~0ull;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(codes);
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    uint4 cw[M16];
    if (i < n) {
#pragma unroll
        for (int w = 0; w < M16; w++) cw[w] = c4[i * M16 + w];
    }
    while (i < n) {
        uint4 nx[M16];
        const u64 inext = i + stride;
        if (inext < n) {      // next code word in flight while this one is looked up
#pragma unroll
            for (int w = 0; w < M16; w++) nx[w] = c4[inext * M16 + w];
        }
        float t[m];
#pragma unroll
        for (int w = 0; w < M16; w++) {
            const u32 words[4] = { cw[w].x, cw[w].y, cw[w].z, cw[w].w };
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int jj = w * 16 + u * 4 + b;
                    t[jj] = lut[jj * 256 + ((words[u] >> (8 * b)) & 255u)];
                }
        }
        float s = 0.0f;
#pragma unroll
        for (int jj = 0; jj < (int)m; jj++) s = f_add(s, t[jj]);
        if (out_sq) out_sq[(size_t)qi * n + i] = s;
        const u64 key = ((u64)__float_as_uint(s) << 32) | (u32)i;
        best = key < best ? key : best;
        i = inext;
#pragma unroll
        for (int w = 0; w < M16; w++) cw[w] = nx[w];
    }
    if (out_best) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const u64 other = __shfl_xor(best, o); best = other < best ? other : best; }
        if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(reinterpret_cast<unsigned long long *>(out_best + qi), (unsigned long long)best);
    }
}

// ---- the skewed flat scan (round 6; one query per block row). pq_scan_kernel looks up ONE table row per step in all 64 lanes at random code
// bytes: 3.15 LDS cycles per 32-lane group on average instead of 1 (68 % of the LDS cycles are bank conflicts), and keeps one code word ahead per
// lane. Here lane l looks up row (p + l) mod m at step p, from a table image stored entry-major: slot x = p + (l & 31) of code byte c sits at word
// c * S + x and holds row x mod m (S = 64 words for m <= 32, 128 above). The 32 lanes of a group are on 32 different slots, so on 32 different
// banks: every ds_read_b32 is conflict-free, and the address is one v_perm_b32 ((c << 8) | 4 (l & 31), the step in the offset field).
// A lane still meets the rows of a code word in the reference's order (A3: strict sequential float sum over j = 0 .. m - 1, fast_pq.py:325-326)
// but with the word boundary at its own step m - psi, psi = (l & 31) mod m: a record of m bytes holds the rows psi .. m - 1 of the word of one
// 64-point row (part A) and then the rows 0 .. psi - 1 of the word of the next row (part B). The bytes come in that order from a scan-order copy of
// the code words (pq_scan_order_kernel), stored by 16-byte quarters so that a wavefront load is 1 KiB contiguous.
// Two running sums per lane, sA for part A and sB for part B: fma(t, 1.0f, s) is the float add s + t and fma(t, 0.0f, s) is s for finite t, so two
// loop-invariant masks per step select the sum without a select instruction; a table image with a non-finite entry (every block checks its image as
// it lands) takes the body with selects instead, which is exact for any value. After a record sA is the finished sum of part A's word and sB carries
// on as the next record's sA. The same bits as pq_scan_kernel (tests/test_gpu_pq_scan_skew.py).
// The copy is cut into V streams of consecutive 64-point rows (stream k: rows G k / V .. G (k + 1) / V - 1, G = ceil(N / 64), plus one closing
// record: a stream of len rows is len + 1 records), one stream per wavefront of a launch that fills the device. il != 0: record t of stream k sits
// at position t * V + k -- the wavefronts of a launch read one moving window of the copy instead of V distant ones (1-4 % faster); il == 0
// (DR_PQ_SCAN_CONTIGUOUS_STREAMS, A/B): at k * (L + 1) + t, L = ceil(G / V).
template <int M16>
__global__ __launch_bounds__(256) void pq_scan_order_kernel(const u8 *__restrict__ codes, u64 n, u32 V, u32 L, u32 il, u8 *__restrict__ out)
{
    constexpr u32 m = 16u * M16;
    const u64 G = (n + 63) / 64, recs = (u64)(L + 1) * V * 64;
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < recs; r += (u64)gridDim.x * blockDim.x) {
        const u64 pos = r >> 6;
        const u32 l = (u32)(r & 63), psi = (l & 31u) % m;
        const u32 k = il ? (u32)(pos % V) : (u32)(pos / (L + 1)), t = il ? (u32)(pos / V) : (u32)(pos % (L + 1));
        const u64 r0 = G * k / V, len = G * (k + 1) / V - r0;
        const u64 pa = (r0 + t) * 64 + l - 64, pb = (r0 + t) * 64 + l;      // part A: the word of row r0 + t - 1, part B: of row r0 + t
        const bool va = t >= 1 && t <= len && pa < n, vb = t < len && pb < n;
        u32 w[m / 4];
#pragma unroll
        for (u32 d = 0; d < m / 4; d++) {
            u32 v = 0;
#pragma unroll
            for (u32 b = 0; b < 4; b++) {
                const u32 p = d * 4 + b;
                u32 byte = 0;
                if (p + psi < m) { if (va) byte = codes[pa * m + psi + p]; }
                else if (vb) byte = codes[pb * m + p + psi - m];
                v |= byte << (8 * b);
            }
            w[d] = v;
        }
        uint4 *o = reinterpret_cast<uint4 *>(out) + pos * (m / 16) * 64 + l;      // quarter d of the 64 lanes of a record: 1 KiB contiguous
#pragma unroll
        for (u32 d = 0; d < m / 16; d++) o[d * 64] = make_uint4(w[4 * d], w[4 * d + 1], w[4 * d + 2], w[4 * d + 3]);
    }
}

// the table of one query in the skewed scan's LDS image, straight from the codebook: out[q][c * S + x] = T_q[x mod m][c] for x < m + 31 (0 beyond),
// every entry by pw_run_lane -- numpy's pairwise order, the bits of lut_build_kernel (A2: fast_pq.py:294-318)
__global__ __launch_bounds__(256) void pq_scan_skew_table_kernel(const float *__restrict__ codebook, const float *__restrict__ queries, u32 D, u32 m, u32 sd, u32 S,
                                                                 float *__restrict__ out)
{
    const u32 q = blockIdx.y;
    float *dst = out + (size_t)q * 256 * S;
    for (u32 e = blockIdx.x * blockDim.x + threadIdx.x; e < 256u * S; e += gridDim.x * blockDim.x) {
        const u32 c = e / S, x = e % S, j = x % m;
        dst[e] = x < m + 31u ? pw_run_lane(codebook + ((size_t)j * 256 + c) * sd, queries + (size_t)q * D + j * sd, (int)sd) : 0.0f;
    }
}

// DEPTH records in flight per lane (registers; a record is refilled as soon as its last lookups are issued, never behind a branch: the loads past a
// stream's end repeat its closing record). Measured at 64M code words: 2 in flight at 16 (m <= 32) or 8 wavefronts per CU beat 1 (-5 %), 3, 4 and
// 6 (-4 .. -10 %: profiles/r06/ab/pq_scan_skew_*.jsonl); a read-only streaming kernel on the same device reaches 6.4-7.2 TB/s
// (scripts/micro/hbm_stream.hip), this one 6.4 at m = 32.
// (OUT: every distance is written. A store in the loop makes the compiler wait for ALL loads in flight once per DEPTH records -- loads and stores
// share gfx950's vmcnt and complete in no order relative to each other -- so the scan that wants the nearest code word only has none.)
// (TOPK: the brute-force ADC search of one query -- dr_pq_scan_topk. Every wavefront keeps an ascending list of k keys (distance bits << 32 | id) in
// LDS behind the table image and a wave-uniform threshold, as pq_scan_topk_kernel does; its list goes to part[query][wavefront of the launch][k] and
// topk_merge_kernel folds the lists.)
template <int M16, int THREADS, int DEPTH, bool OUT, bool TOPK = false>
__global__ __launch_bounds__(THREADS) void pq_scan_skew_kernel(const float *__restrict__ lut_skew, const u8 *__restrict__ scan_codes,
                                                              u64 n, u32 V, u32 L, u32 il, float *__restrict__ out_sq, u64 *__restrict__ out_best,
                                                              u32 k, u64 *__restrict__ part)
{
    constexpr u32 m = 16u * M16, S = m <= 32 ? 64u : 128u;
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 qi = blockIdx.y;
    bool safe;
    {
        const float4 *src = reinterpret_cast<const float4 *>(lut_skew + (size_t)qi * 256 * S);
        float4 *dst = reinterpret_cast<float4 *>(smem);
        bool bad = false;
        for (u32 e = threadIdx.x; e < 64u * S; e += THREADS) {
            const float4 v = src[e];
            dst[e] = v;
            bad |= !(fabsf(v.x) <= 3.402823466e38f && fabsf(v.y) <= 3.402823466e38f && fabsf(v.z) <= 3.402823466e38f && fabsf(v.w) <= 3.402823466e38f);
        }
        if ((u32)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();      // the lookups below address the LDS from 0
        // a non-finite entry anywhere: the body with selects. The vote goes through slot S - 1 of code byte 0, which no lookup reads (x <= m + 30) and
        // the image holds as 0 (__syncthreads_or would bring a static LDS word of its own and move the image off address 0)
        u32 *vote = reinterpret_cast<u32 *>(smem) + (S - 1);
        __syncthreads();
        if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0) atomicOr(vote, 1u);
        __syncthreads();
        safe = *vote != 0u;
    }
    const u32 lane = threadIdx.x & 63u, phi = lane & 31u, psi = phi % m;
    const u64 G = (n + 63) / 64;
    const u32 W = gridDim.x * (THREADS / 64), gw = blockIdx.x * (THREADS / 64) + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    auto uniform64 = [](u64 v) { return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)v); };
    const u32 n32 = (u32)(n > 0xFFFFFFFFull ? 0xFFFFFFFFull : n);
    u32 bs = ~0u, bi = ~0u;      // the lane's nearest code word: sum bits (sums are >= +0: their bits order as the values do, NaN last), id
    u64 *klist = reinterpret_cast<u64 *>(smem + (size_t)256 * S * 4) + (threadIdx.x >> 6) * 64;      // TOPK: the wavefront's k smallest keys, ascending
    u64 Wk = ~0ull;              // ... and its k-th smallest key so far (wave-uniform)
    int kn = 0;
    if constexpr (TOPK) { klist[lane] = ~0ull; WSYNC(); }
    float mA[m], mB[m];
#pragma unroll
    for (u32 p = 0; p < m; p++) {
        mA[p] = (p + psi < m) ? 1.0f : 0.0f;
        mB[p] = 1.0f - mA[p];
        asm volatile("" : "+v"(mB[p]));      // (kept in a register: re-derived from mA it is a third instruction per step)
    }
    const u32 lanebase = phi * 4u;
    const u32x4 *recs = reinterpret_cast<const u32x4 *>(scan_codes);      // record at position pos, quarter w, lane l: recs[(pos * M16 + w) * 64 + l]
    for (u32 ks = gw; ks < V; ks += W) {      // (the default launch: W == V, one stream per wavefront)
        // (scalar registers: the loop counter, the range tests and the record addresses are scalar code)
        const u64 r0 = uniform64(G * ks / V), len = uniform64(G * (ks + 1) / V) - r0;
        if (len == 0) continue;
        const u64 p0 = il ? (u64)ks : (u64)ks * (L + 1), pstep = il ? (u64)V : 1ull;      // record t at position p0 + t * pstep
        u32x4 buf[DEPTH][M16];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const u64 tl = (u64)d < len ? (u64)d : len;
#pragma unroll
            for (int w = 0; w < M16; w++) buf[d][w] = __builtin_nontemporal_load(recs + ((p0 + tl * pstep) * M16 + w) * 64 + lane);
        }
        float sA = 0.0f;
        for (u64 t0 = 0; t0 <= len; t0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; d++) {
                const u64 tt = t0 + d;      // (tt > len in a stream's last group: the repeated closing record is looked up and dropped -- an exit here
                                           //  makes the compiler wait for every load in flight at the loop's head)
                // the lookups of a 16-byte quarter are in flight while the sums take the quarter before it
                float t[2][16];
                auto look = [&](int w, float (&tq)[16]) {
                    const u32 words[4] = { buf[d][w].x, buf[d][w].y, buf[d][w].z, buf[d][w].w };
#pragma unroll
                    for (int u = 0; u < 4; u++)
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int p = w * 16 + u * 4 + b;
                            u32 addr;
                            if constexpr (S == 64) addr = __builtin_amdgcn_perm(words[u], lanebase, 0x0C0C0400u | ((u32)b << 8));      // (c << 8) | lanebase
                            else addr = (((words[u] >> (8 * b)) & 255u) << 9) + lanebase;
                            tq[u * 4 + b] = *(const __attribute__((address_space(3))) float *)(uintptr_t)(addr + 4 * p);      // (the image starts at LDS address 0: checked above)
                        }
                };
                look(0, t[0]);
                float sB = 0.0f;
#pragma unroll
                for (int w = 0; w < M16; w++) {
                    if (w + 1 < M16) look(w + 1, t[(w + 1) & 1]);
                    else {
                        const u64 tl = tt + DEPTH < len ? tt + DEPTH : len;
#pragma unroll
                        for (int w2 = 0; w2 < M16; w2++) buf[d][w2] = __builtin_nontemporal_load(recs + ((p0 + tl * pstep) * M16 + w2) * 64 + lane);
                    }
                    if (!safe) {
#pragma unroll
                        for (int e = 0; e < 16; e++) { sA = __builtin_fmaf(t[w & 1][e], mA[w * 16 + e], sA); sB = __builtin_fmaf(t[w & 1][e], mB[w * 16 + e], sB); }
                    } else {
#pragma unroll
                        for (int e = 0; e < 16; e++) { const bool inA = (u32)(w * 16 + e) + psi < m; sA = f_add(sA, inA ? t[w & 1][e] : 0.0f); sB = f_add(sB, inA ? 0.0f : t[w & 1][e]); }
                    }
                }
                if (tt >= 1 && tt <= len) {      // the word that ended in this record: point (r0 + tt - 1) * 64 + lane (ids ascend along a lane: "<" keeps the smallest id of equal sums)
                    const u32 i = (u32)(r0 + tt) * 64u - 64u + lane, sb = __float_as_uint(sA);
                    if constexpr (OUT) { if (i < n32) out_sq[(size_t)qi * n + i] = sA; }
                    if constexpr (TOPK) {
                        const u64 key = ((u64)sb << 32) | i;
                        u64 cmk = __ballot(i < n32 && key < Wk);
                        while (cmk) {      // (rare once the list has warmed up: the threshold is the k-th smallest of what this wavefront has seen)
                            const int f = __ffsll((long long)cmk) - 1;
                            cmk &= cmk - 1;
                            const u64 kf = readlane64(key, f);
                            if (kf < Wk) {
                                const u64 mine = ((int)lane < kn) ? klist[lane] : ~0ull;
                                const int pos = __popcll(__ballot((int)lane < kn && mine < kf));
                                WSYNC();
                                if ((int)lane < kn && (int)lane >= pos && lane + 1 < k) klist[lane + 1] = mine;
                                if (lane == 0) klist[pos] = kf;
                                WSYNC();
                                if (kn < (int)k) kn++;
                                if (kn == (int)k) Wk = klist[k - 1];
                            }
                        }
                    } else {
                        const bool better = sb < bs && i < n32;
                        bs = better ? sb : bs;
                        bi = better ? i : bi;
                    }
                }
                sA = sB;
            }
        }
    }
    if constexpr (TOPK) {
        WSYNC();
        if (lane < k) part[((size_t)qi * W + gw) * k + lane] = ((int)lane < kn) ? klist[lane] : ~0ull;
    } else if (out_best) {
        u64 best = ((u64)bs << 32) | bi;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const u64 other = __shfl_xor(best, o); best = other < best ? other : best; }
        if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(reinterpret_cast<unsigned long long *>(out_best + qi), (unsigned long long)best);
    }
}

// Flat PQ scan keeping the k nearest code words per query (brute-force ADC search: the ground truth of the PQ-only
// traversals, and what a PQ-only shard falls back to when no graph exists). Tables come from lut_build_kernel (global
// memory, landed in LDS once per block); every wavefront keeps its own ascending list of k keys (distance bits << 32 | id:
// the smaller id wins a tie) in LDS and a wave-uniform threshold, so after the first few thousand code words a candidate
// is rare and the loop runs at the pace of pq_scan_kernel. Lists go to part[query][block * 4 + wave][k]; topk_merge_kernel
// folds them.
template <int M16>
__global__ __launch_bounds__(256) void pq_scan_topk_kernel(const float *__restrict__ lut_g, const u8 *__restrict__ codes, u64 n, u32 k,
                                                           u64 *__restrict__ part)
{
    constexpr u32 m = 16u * M16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *lut = reinterpret_cast<float *>(smem);
    u64 *best = reinterpret_cast<u64 *>(smem + (size_t)m * 1024) + (threadIdx.x >> 6) * 64;
    const u32 qi = blockIdx.y, lane = threadIdx.x & 63;
    {
        const float4 *src = reinterpret_cast<const float4 *>(lut_g + (size_t)qi * m * 256);
        float4 *dst = reinterpret_cast<float4 *>(lut);
        for (u32 e = threadIdx.x; e < m * 64; e += 256) dst[e] = src[e];
    }
    best[lane] = ~0ull;
    __syncthreads();
    u64 Wk = ~0ull;       // the wave's k-th smallest key so far (wave-uniform)
    int bn = 0;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(codes);
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i0 = (u64)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += stride) {
        const u64 i = i0 + lane;
        float s = 0.0f;
        if (i < n) {
#pragma unroll
            for (int w = 0; w < M16; w++) s = adc_lut16(lut, c4[i * M16 + w], (u32)w * 16, s);
        }
        const u64 key = ((u64)__float_as_uint(s) << 32) | (u32)i;
        u64 cm = __ballot(i < n && key < Wk);
        while (cm) {
            const int f = __ffsll((long long)cm) - 1;
            cm &= cm - 1;
            const u64 kf = readlane64(key, f);
            if (kf < Wk) {
                const u64 mine = ((int)lane < bn) ? best[lane] : ~0ull;
                const int pos = __popcll(__ballot((int)lane < bn && mine < kf));
                WSYNC();
                if ((int)lane < bn && (int)lane >= pos && lane + 1 < k) best[lane + 1] = mine;
                if (lane == 0) best[pos] = kf;
                WSYNC();
                if (bn < (int)k) bn++;
                if (bn == (int)k) Wk = best[k - 1];
            }
        }
    }
    WSYNC();
    if (lane < k) part[((size_t)qi * gridDim.x * 4 + blockIdx.x * 4 + (threadIdx.x >> 6)) * k + lane] = ((int)lane < bn) ? best[lane] : ~0ull;
}
// one wavefront per query: the k smallest of its P partial lists of k keys
__global__ __launch_bounds__(64) void topk_merge_kernel(const u64 *__restrict__ part, u32 P, u32 k, u32 *__restrict__ out_ids, float *__restrict__ out_sq)
{
    __shared__ u64 best[64];
    const u32 qi = blockIdx.x, lane = threadIdx.x;
    best[lane] = ~0ull;
    WSYNC();
    u64 Wk = ~0ull;
    int bn = 0;
    const u64 *src = part + (size_t)qi * P * k;
    const u32 total = P * k;
    for (u32 i0 = 0; i0 < total; i0 += 64) {
        const u64 key = (i0 + lane < total) ? src[i0 + lane] : ~0ull;
        u64 cm = __ballot(key < Wk);
        while (cm) {
            const int f = __ffsll((long long)cm) - 1;
            cm &= cm - 1;
            const u64 kf = readlane64(key, f);
            if (kf < Wk) {
                const u64 mine = ((int)lane < bn) ? best[lane] : ~0ull;
                const int pos = __popcll(__ballot((int)lane < bn && mine < kf));
                WSYNC();
                if ((int)lane < bn && (int)lane >= pos && lane + 1 < k) best[lane + 1] = mine;
                if (lane == 0) best[pos] = kf;
                WSYNC();
                if (bn < (int)k) bn++;
                if (bn == (int)k) Wk = best[k - 1];
            }
        }
    }
    WSYNC();
    if (lane < k) {
        const u64 key = ((int)lane < bn) ? best[lane] : ~0ull;
        out_ids[(size_t)qi * k + lane] = ((int)lane < bn) ? (u32)key : 0xFFFFFFFFu;
        out_sq[(size_t)qi * k + lane] = ((int)lane < bn) ? key_dist(key) : __uint_as_float(0x7FC00000u);
    }
}

// ---- builder helpers (build_kernels.hpp has the per-dimension prune kernel) ---------------------------------
__global__ void gather_rows_kernel(const float *__restrict__ src, const u32 *__restrict__ ids, u32 n, u32 D,
                                   float *__restrict__ dst)
{
    const u32 row = blockIdx.x;
    if (row >= n) return;
    const float *s = src + (size_t)ids[row] * D;
    for (u32 e = threadIdx.x; e < D; e += blockDim.x) dst[(size_t)row * D + e] = s[e];
}

// Adds p to the row of each of its selected neighbours (cython_utils.pyx:338-348). Rows have RX >= R slots;
// the thread that takes slot `trigger` reports the row for re-pruning: trigger = R is the reference's rule (a row is
// pruned as soon as it exceeds R, :350-356); the PQ-only builder lets rows run into their slack first (R + slack/2:
// one re-prune per ~slack/2 reverse edges instead of one per edge) and prunes what is still over R at the end.
__global__ void reverse_edges_kernel(u32 *__restrict__ adjb, u32 *__restrict__ deg, u32 RX, u32 R,
                                     const u32 *__restrict__ points, u32 npoints, const u32 *__restrict__ fwd,
                                     const u32 *__restrict__ fwd_n, u32 *__restrict__ ovf_list,
                                     u32 *__restrict__ ovf_count, u32 ovf_cap, u32 trigger)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 pi = t / R, s = t % R;
    if (pi >= npoints || s >= fwd_n[pi]) return;
    const u32 pt = points[pi];
    const u32 n = fwd[(size_t)pi * R + s];
    if (n == 0xFFFFFFFFu || n == pt) return;
    u32 *row = adjb + (size_t)n * RX;
    const u32 dn = min(__hip_atomic_load(&deg[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), RX);
    for (u32 i = 0; i < dn; i++)
        if (__hip_atomic_load(&row[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == pt) return;
    const u32 slot = atomicAdd(&deg[n], 1u);
    if (slot < RX) __hip_atomic_store(&row[slot], pt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slot == trigger) {
        const u32 k = atomicAdd(ovf_count, 1u);
        if (k < ovf_cap) ovf_list[k] = n;
    }
}

// The deterministic form of the reverse-edge pass (the builder's default): the batch's (target, source) pairs are sorted
// (rocprim radix sort on target << 32 | source) and the thread that finds the head of a target's run appends the whole run
// in ascending source order. Which edges a full slack row keeps (the smallest sources) and the order inside every row no
// longer depend on the arrival order of atomics: two builds of one dataset are identical bit for bit.
__global__ void rev_pairs_kernel(const u32 *__restrict__ points, u32 npoints, u32 R, const u32 *__restrict__ fwd,
                                 const u32 *__restrict__ fwd_n, u64 *__restrict__ keys)
{
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= npoints * R) return;
    const u32 pi = t / R, s = t % R;
    u64 key = ~0ull;
    if (s < fwd_n[pi]) {
        const u32 pt = points[pi], n = fwd[(size_t)pi * R + s];
        if (n != 0xFFFFFFFFu && n != pt) key = ((u64)n << 32) | pt;
    }
    keys[t] = key;
}
__global__ void rev_apply_sorted_kernel(const u64 *__restrict__ keys, u32 total, u32 *__restrict__ adjb, u32 *__restrict__ deg, u32 RX,
                                        u32 *__restrict__ ovf_list, u32 *__restrict__ ovf_count, u32 ovf_cap, u32 trigger)
{
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const u64 k0 = keys[i];
    if (k0 == ~0ull) return;
    const u32 n = (u32)(k0 >> 32);
    if (i > 0 && (u32)(keys[i - 1] >> 32) == n) return;         // not the head of its target's run
    u32 *row = adjb + (size_t)n * RX;
    const u32 d0 = deg[n];                                       // this thread is the only writer of row n in this launch
    const u32 have = min(d0, RX);
    u32 d = d0;
    for (u32 j = i; j < total; j++) {
        const u64 k = keys[j];
        if ((u32)(k >> 32) != n) break;
        const u32 pt = (u32)k;
        bool dup = false;
        for (u32 e = 0; e < have; e++) dup = dup || row[e] == pt;   // (sources of one batch are distinct: only the old entries)
        if (dup) continue;
        if (d < RX) row[d] = pt;
        d++;
    }
    deg[n] = d;
    if (d0 <= trigger && d > trigger) {          // the row passed the re-prune threshold in this batch (slot == trigger)
        const u32 q = atomicAdd(ovf_count, 1u);
        if (q < ovf_cap) ovf_list[q] = n;
    }
}

// rows whose degree exceeds R (final prune of the slack-tolerant builder)
__global__ void collect_over_kernel(const u32 *__restrict__ deg, u64 N, u32 R, u32 *__restrict__ list, u32 *__restrict__ count)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x)
        if (deg[i] > R) list[atomicAdd(count, 1u)] = (u32)i;
}

// Final rows: the first min(deg, R) ids, then `padval` (0 reproduces the reference writer, diskann_persist.py:23).
__global__ void compact_adj_kernel(const u32 *__restrict__ adjb, const u32 *__restrict__ deg, u64 N, u32 RX, u32 R,
                                   u32 padval, u32 *__restrict__ adj)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N * R; i += (u64)gridDim.x * blockDim.x) {   // (2^32 work-item limit)
        const u64 row = i / R;
        const u32 s = (u32)(i % R);
        const u32 d = min(deg[row], R);
        adj[i] = s < d ? adjb[row * RX + s] : padval;
    }
}

// The same with every row's neighbours in a canonical order -- ascending place in the locality order of the visited bits
// (rank[id], a function of the vectors only: neighbours that share a bitmap line end up in adjacent lanes), or ascending
// id when there is no such order (small or PQ-only indexes); R <= 128, one wavefront per row, rank by counting. The
// batched builder appends reverse edges with atomics, so the ORDER inside a row depends on arrival order while the SET
// does not (two builds of one dataset: identical sorted rows, different raw rows -- and M1 / M3 walk a row in stored
// order). A canonical order makes device-built graphs reproducible bit for bit; the reference's own order is a Python
// set's iteration order (diskann_persist.py:17-24), i.e. arbitrary as well.
__global__ __launch_bounds__(256) void compact_adj_sorted_kernel(const u32 *__restrict__ adjb, const u32 *__restrict__ deg, u64 N, u32 RX, u32 R,
                                                                   u32 padval, const u32 *__restrict__ rank, u32 *__restrict__ adj)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 nwaves = (u64)gridDim.x * (blockDim.x >> 6);
    for (u64 row = (u64)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < N; row += nwaves) {
        const u32 d = min(deg[row], R);
        const u32 x0 = lane < d ? adjb[row * RX + lane] : 0xFFFFFFFFu;
        const u32 x1 = lane + 64u < d ? adjb[row * RX + 64u + lane] : 0xFFFFFFFFu;
        // sort key (distinct inside a row: ids are, and rank is a bijection)
        const u32 k0 = (rank && lane < d && (u64)x0 < N) ? rank[x0] : x0;
        const u32 k1 = (rank && lane + 64u < d && (u64)x1 < N) ? rank[x1] : x1;
        u32 r0 = 0u, r1 = 0u;
        for (u32 t = 0; t < d; t++) {
            const u32 v = t < 64u ? (u32)__builtin_amdgcn_readlane((int)k0, (int)t) : (u32)__builtin_amdgcn_readlane((int)k1, (int)(t - 64u));
            r0 += v < k0 ? 1u : 0u;
            r1 += v < k1 ? 1u : 0u;
        }
        if (lane < d) adj[row * R + r0] = x0;
        if (lane + 64u < d) adj[row * R + r1] = x1;
        for (u32 s = d + lane; s < R; s += 64u) adj[row * R + s] = padval;
    }
}

__global__ void column_sum_kernel(const float *__restrict__ vecp, u64 N, u32 D, double *__restrict__ acc)
{
    // one block per chunk of rows; thread e sums dimension e (+blockDim strides)
    const u64 rows_per_block = (N + gridDim.x - 1) / gridDim.x;
    const u64 r0 = (u64)blockIdx.x * rows_per_block, r1 = min(N, r0 + rows_per_block);
    for (u32 e = threadIdx.x; e < D; e += blockDim.x) {
        double s = 0.0;
        for (u64 r = r0; r < r1; r++) s += (double)vecp[r * D + e];
        atomicAdd(&acc[e], s);
    }
}

// ---- PQ training / encoding (DiskANNPQ.fit / encode, pq/fast_pq.py:197-267) -----------------
// Nearest centroid of a sub-vector among the 256 of cb[256][sd] (LDS): plain squared L2 in element order, ties to the
// lowest centroid index (argmin). SD > 0: the sub-vector is a register array and the loops are unrolled (SD = 0: any sd,
// the array is indexed dynamically and lives in scratch -- an order of magnitude slower).
template <int SD>
DEV u32 nearest_centroid(const float *x, const float *cb, u32 sd)
{
    float best = 3.4e38f;
    u32 bi = 0;
    if constexpr (SD > 0) {
        for (u32 c = 0; c < 256; c++) {
            float s = 0.0f;
#pragma unroll
            for (int t = 0; t < SD; t++) { const float d = x[t] - cb[c * SD + t]; s += d * d; }
            if (s < best) { best = s; bi = c; }
        }
    } else {
        for (u32 c = 0; c < 256; c++) {
            float s = 0.0f;
            for (u32 t = 0; t < sd; t++) { const float d = x[t] - cb[c * sd + t]; s += d * d; }
            if (s < best) { best = s; bi = c; }
        }
    }
    return bi;
}

// Nearest centroid of sub-vector j of stored vector ids[i] (or i when ids == nullptr); one thread per (i, j).
template <int SD>
__global__ void pq_assign_kernel(const float *__restrict__ vecp, const u32 *__restrict__ perm,
                                 const u32 *__restrict__ ids, u64 n, u32 D, u32 m, u32 sd,
                                 const float *__restrict__ codebook, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *cb = reinterpret_cast<float *>(smem);          // codebook of sub-quantiser blockIdx.y: [256][sd]
    const u32 jq = blockIdx.y;
    for (u32 e = threadIdx.x; e < 256 * sd; e += blockDim.x) cb[e] = codebook[(size_t)jq * 256 * sd + e];
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 node = ids ? ids[i] : i;
        float x[SD > 0 ? SD : 128];
        if constexpr (SD > 0) {
#pragma unroll
            for (int t = 0; t < SD; t++) x[t] = vecp[node * D + perm[jq * SD + t]];
        } else {
            for (u32 t = 0; t < sd; t++) x[t] = vecp[node * D + perm[jq * sd + t]];
        }
        out[i * m + jq] = (u8)nearest_centroid<SD>(x, cb, sd);
    }
}

// nearest-centroid codes for `rows` row-major vectors (a streamed chunk; DiskANNPQ.encode, fast_pq.py:245-267)
template <int SD>
__global__ void pq_assign_rows_kernel(const float *__restrict__ x, u64 rows, u32 D, u32 m, u32 sd,
                                      const float *__restrict__ codebook, u8 *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *cb = reinterpret_cast<float *>(smem);
    const u32 jq = blockIdx.y;
    for (u32 e = threadIdx.x; e < 256 * sd; e += blockDim.x) cb[e] = codebook[(size_t)jq * 256 * sd + e];
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (u64)gridDim.x * blockDim.x) {
        float v[SD > 0 ? SD : 128];
        if constexpr (SD > 0) {
#pragma unroll
            for (int t = 0; t < SD; t++) v[t] = x[i * D + jq * SD + t];
        } else {
            for (u32 t = 0; t < sd; t++) v[t] = x[i * D + jq * sd + t];
        }
        out[i * m + jq] = (u8)nearest_centroid<SD>(v, cb, sd);
    }
}

// ---- k-means on the device (round 3: seeding and centroid update used to run on host threads) ------------------------
// All kernels work on the gathered sample xs[ns][D] (original element order); one workgroup (or grid row) per
// sub-quantiser jq. Every reduction is a fixed tree or an integer sum: two trainings with one seed give the same bits.
#define DR_KM_THREADS 1024
#define DR_KM_TRIALS 7      // sklearn: 2 + int(log(n_clusters)) local trials per seeding step (k = 256)

// deterministic block sum of one double per thread (tree over LDS; every thread gets the total)
DEV double km_block_sum(double v, double *red)
{
    const u32 t = threadIdx.x;
    __syncthreads();
    red[t] = v;
    __syncthreads();
    for (u32 o = blockDim.x >> 1; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
    const double r = red[0];
    __syncthreads();
    return r;
}
DEV float km_dist2(const float *__restrict__ x, const float *c, u32 sd)
{
    float s2 = 0.0f;
    for (u32 t = 0; t < sd; t++) { const float d = x[t] - c[t]; s2 += d * d; }
    return s2;
}

// per sub-quantiser: mean per-feature variance (sklearn's tol scaling, _kmeans.py _tolerance) and the largest |x|
__global__ __launch_bounds__(DR_KM_THREADS) void km_stats_kernel(const float *__restrict__ xs, u32 ns, u32 D, u32 sd, double *__restrict__ var_mean,
                                                                  float *__restrict__ max_abs)
{
    __shared__ double red[DR_KM_THREADS];
    const u32 jq = blockIdx.x;
    double vsum = 0.0, mx = 0.0;
    for (u32 t = 0; t < sd; t++) {
        double s1 = 0.0, s2 = 0.0, m1 = 0.0;
        for (u32 i = threadIdx.x; i < ns; i += blockDim.x) { const double v = xs[(size_t)i * D + jq * sd + t]; s1 += v; s2 += v * v; m1 = fmax(m1, fabs(v)); }
        s1 = km_block_sum(s1, red); s2 = km_block_sum(s2, red);
        // max through the same tree (max is exact in any order)
        __syncthreads(); red[threadIdx.x] = m1; __syncthreads();
        for (u32 o = blockDim.x >> 1; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
        mx = fmax(mx, red[0]);
        const double mu = s1 / ns;
        vsum += s2 / ns - mu * mu;
    }
    if (threadIdx.x == 0) { var_mean[jq] = vsum / sd; max_abs[jq] = (float)mx; }
}

// xt[e][i] = xs[i][e]: the sample with one COLUMN per vector component, so that the threads of the seeding kernel (one sample
// each) read consecutive addresses. In row order every load instruction of a wavefront touched 64 rows 4*D bytes apart and the
// kernel ran at the texture addresser's line rate (2.35 s for 50 000 x 1536, m = 32); on the columns it is 0.25 s.
__global__ void km_transpose_kernel(const float *__restrict__ xs, u32 ns, u32 D, float *__restrict__ xt)
{
    __shared__ float tile[32][33];
    const u32 i0 = blockIdx.x * 32, e0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8 threads
    for (u32 r = ty; r < 32; r += 8) { const u32 i = i0 + r, e = e0 + tx; tile[r][tx] = (i < ns && e < D) ? xs[(size_t)i * D + e] : 0.0f; }
    __syncthreads();
    for (u32 r = ty; r < 32; r += 8) { const u32 e = e0 + r, i = i0 + tx; if (e < D && i < ns) xt[(size_t)e * ns + i] = tile[tx][r]; }
}

// Greedy k-means++ seeding (Arthur & Vassilvitskii 2007 with sklearn's local trials: sklearn.cluster._kmeans._kmeans_plusplus,
// what DiskANNPQ.fit reaches through init='k-means++', pq/fast_pq.py:231-238). One workgroup per sub-quantiser; the
// uniform random numbers come from the host (unif[jq][step][0..7): step 0 slot 0 picks the first centre, later steps use
// DR_KM_TRIALS slots), so the choice sequence is a function of the seed alone. d2[jq][i]: squared distance of sample i to its
// nearest chosen centre. xt: the sample by columns (km_transpose_kernel); every squared distance is summed over the
// components in ascending order, whatever the layout.
__global__ __launch_bounds__(DR_KM_THREADS) void kmeanspp_kernel(const float *__restrict__ xt, u32 ns, u32 D, u32 sd, const double *__restrict__ unif,
                                                                  float *__restrict__ d2_all, float *__restrict__ cb /*[m][256][sd]*/)
{
    __shared__ double red[DR_KM_THREADS];
    __shared__ float cand_v[DR_KM_TRIALS][128];
    __shared__ u32 cand_i[DR_KM_TRIALS];
    __shared__ double thr[DR_KM_TRIALS];
    const u32 jq = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    (void)D;
    const float *x0 = xt + (size_t)jq * sd * ns;          // component t of sample i: x0[t * ns + i]
    float *d2 = d2_all + (size_t)jq * ns;
    float *cbj = cb + (size_t)jq * 256 * sd;
    const double *uj = unif + (size_t)jq * 256 * 8;
    // first centre
    const u32 first = min((u32)(uj[0] * ns), ns - 1);
    for (u32 t = tid; t < sd; t += nt) { const float v = x0[(size_t)t * ns + first]; cand_v[0][t] = v; cbj[t] = v; }
    __syncthreads();
    double part = 0.0;
    for (u32 i = tid; i < ns; i += nt) {
        float s2 = 0.0f;
        for (u32 t = 0; t < sd; t++) { const float d = x0[(size_t)t * ns + i] - cand_v[0][t]; s2 += d * d; }
        d2[i] = s2; part += s2;
    }
    double pot = km_block_sum(part, red);
    const u32 cs = (ns + nt - 1) / nt;      // contiguous chunk per thread for the sampling pass
    for (u32 c = 1; c < 256; c++) {
        // (a) DR_KM_TRIALS candidates with probability proportional to d2: first index whose running sum exceeds r
        if (tid < DR_KM_TRIALS) { thr[tid] = uj[c * 8 + tid] * pot; cand_i[tid] = ns - 1; }
        const u32 lo = min(tid * cs, ns), hi = min(lo + cs, ns);
        double loc = 0.0;
        for (u32 i = lo; i < hi; i++) loc += d2[i];
        __syncthreads();
        red[tid] = loc;
        __syncthreads();
        // exclusive prefix of the chunk sums: a fixed serial pass by one thread (1024 adds) keeps the order simple and exact
        if (tid == 0) { double acc = 0.0; for (u32 u = 0; u < nt; u++) { const double v = red[u]; red[u] = acc; acc += v; } }
        __syncthreads();
        const double base = red[tid];
        for (int q = 0; q < DR_KM_TRIALS; q++) {
            const double r = thr[q];
            if (r >= base && r < base + loc) {
                double acc = base; u32 pick = hi - 1;
                for (u32 i = lo; i < hi; i++) { acc += d2[i]; if (r < acc) { pick = i; break; } }
                cand_i[q] = pick;
            }
        }
        __syncthreads();
        for (u32 e = tid; e < DR_KM_TRIALS * sd; e += nt) { const u32 q = e / sd, t = e - q * sd; cand_v[q][t] = x0[(size_t)t * ns + cand_i[q]]; }
        __syncthreads();
        // (b) potential of every candidate: one pass over the sample, the DR_KM_TRIALS sums side by side
        double np[DR_KM_TRIALS];
#pragma unroll
        for (int q = 0; q < DR_KM_TRIALS; q++) np[q] = 0.0;
        for (u32 i = tid; i < ns; i += nt) {
            float s2[DR_KM_TRIALS];
#pragma unroll
            for (int q = 0; q < DR_KM_TRIALS; q++) s2[q] = 0.0f;
            for (u32 t = 0; t < sd; t++) {
                const float xv = x0[(size_t)t * ns + i];
#pragma unroll
                for (int q = 0; q < DR_KM_TRIALS; q++) { const float d = xv - cand_v[q][t]; s2[q] += d * d; }
            }
            const float cur = d2[i];
#pragma unroll
            for (int q = 0; q < DR_KM_TRIALS; q++) np[q] += (double)fminf(cur, s2[q]);
        }
        int best = 0; double best_pot = 0.0;
#pragma unroll
        for (int q = 0; q < DR_KM_TRIALS; q++) {
            const double tot = km_block_sum(np[q], red);
            if (q == 0 || tot < best_pot) { best_pot = tot; best = q; }
        }
        // (c) the winner becomes centre c
        for (u32 i = tid; i < ns; i += nt) {
            float s2 = 0.0f;
            for (u32 t = 0; t < sd; t++) { const float d = x0[(size_t)t * ns + i] - cand_v[best][t]; s2 += d * d; }
            d2[i] = fminf(d2[i], s2);
        }
        for (u32 t = tid; t < sd; t += nt) cbj[(size_t)c * sd + t] = cand_v[best][t];
        pot = best_pot;
        __syncthreads();
    }
}

// Lloyd update, step 1: per-centre coordinate sums in FIXED POINT (x * 2^shift[jq] as a 64-bit integer: integer addition is
// associative, so the sums do not depend on the order the atomics arrive in) and member counts. LDS accumulators when
// they fit (256 * sd * 8 bytes), flushed with global atomics; grid (chunks, m).
__global__ __launch_bounds__(256) void km_accumulate_kernel(const float *__restrict__ xs, const u8 *__restrict__ assign, u32 ns, u32 D, u32 m, u32 sd,
                                                            const int *__restrict__ shift, unsigned long long *__restrict__ sums /*[m][256][sd]*/,
                                                            u32 *__restrict__ counts /*[m][256]*/, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long *ls = reinterpret_cast<unsigned long long *>(smem);
    u32 *lc = reinterpret_cast<u32 *>(ls + (use_lds ? (size_t)256 * sd : 0));
    const u32 jq = blockIdx.y;
    const double scale = ldexp(1.0, shift[jq]);
    unsigned long long *gs = sums + (size_t)jq * 256 * sd;
    u32 *gc = counts + (size_t)jq * 256;
    if (use_lds) { for (u32 e = threadIdx.x; e < 256 * sd; e += blockDim.x) ls[e] = 0ull; }
    for (u32 e = threadIdx.x; e < 256; e += blockDim.x) lc[e] = 0u;
    __syncthreads();
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < ns; i += gridDim.x * blockDim.x) {
        const u32 c = assign[(size_t)i * m + jq];
        atomicAdd(&lc[c], 1u);
        const float *xi = xs + (size_t)i * D + jq * sd;
        unsigned long long *dst = (use_lds ? ls : gs) + (size_t)c * sd;
        for (u32 t = 0; t < sd; t++) atomicAdd(&dst[t], (unsigned long long)(long long)llrint((double)xi[t] * scale));
    }
    __syncthreads();
    if (use_lds) { for (u32 e = threadIdx.x; e < 256 * sd; e += blockDim.x) if (ls[e]) atomicAdd(&gs[e], ls[e]); }
    for (u32 e = threadIdx.x; e < 256; e += blockDim.x) if (lc[e]) atomicAdd(&gc[e], lc[e]);
}
// step 2: centres = sums / counts (empty clusters keep their centre), total squared shift per sub-quantiser; clears the
// accumulators for the next iteration. One workgroup of 256 threads (thread c = centre c) per sub-quantiser.
__global__ __launch_bounds__(256) void km_finalize_kernel(float *__restrict__ cb, u32 sd, const int *__restrict__ shift, unsigned long long *__restrict__ sums,
                                                          u32 *__restrict__ counts, double *__restrict__ out_shift)
{
    __shared__ double red[256];
    const u32 jq = blockIdx.x, c = threadIdx.x;
    const double inv = ldexp(1.0, -shift[jq]);
    const u32 n = counts[(size_t)jq * 256 + c];
    double sh = 0.0;
    for (u32 t = 0; t < sd; t++) {
        const size_t e = ((size_t)jq * 256 + c) * sd + t;
        if (n) {
            const float nv = (float)((double)(long long)sums[e] * inv / (double)n);
            const double d = (double)nv - (double)cb[e];
            sh += d * d;
            cb[e] = nv;
        }
        sums[e] = 0ull;
    }
    counts[(size_t)jq * 256 + c] = 0u;
    const double tot = km_block_sum(sh, red);
    if (c == 0) out_shift[jq] = tot;
}
// quantisation error of the sample for the current centres and labels
__global__ __launch_bounds__(DR_KM_THREADS) void km_inertia_kernel(const float *__restrict__ xs, const u8 *__restrict__ assign, u32 ns, u32 D, u32 m, u32 sd,
                                                                   const float *__restrict__ cb, double *__restrict__ out)
{
    __shared__ double red[DR_KM_THREADS];
    const u32 jq = blockIdx.x;
    double s = 0.0;
    for (u32 i = threadIdx.x; i < ns; i += blockDim.x) {
        const float *c = cb + ((size_t)jq * 256 + assign[(size_t)i * m + jq]) * sd;
        const float *xi = xs + (size_t)i * D + jq * sd;
        for (u32 t = 0; t < sd; t++) { const double d = (double)xi[t] - (double)c[t]; s += d * d; }
    }
    const double tot = km_block_sum(s, red);
    if (threadIdx.x == 0) out[jq] = tot;
}

__global__ void gather_subvectors_kernel(const float *__restrict__ vecp, const u32 *__restrict__ perm,
                                         const u32 *__restrict__ ids, u32 n, u32 D, float *__restrict__ out)
{
    // original element order, for the host-side centroid update
    const u32 row = blockIdx.x;
    if (row >= n) return;
    for (u32 e = threadIdx.x; e < D; e += blockDim.x) out[(size_t)row * D + e] = vecp[(size_t)ids[row] * D + perm[e]];
}

// sqrt(sum_j max_c T[j][c]) per query, sum in A3's order (an upper bound of asymmetric_distance for any code word; its use is described in search_kernel.hpp): one block of
// 256 threads per query, thread c owns centroid c of every sub-quantiser (coalesced codebook reads).
__global__ __launch_bounds__(256) void pq_bound_kernel(const float *__restrict__ codebook, const float *__restrict__ queries,
                                                       u32 D, u32 m, u32 sd, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *q = reinterpret_cast<float *>(smem);
    float *wmax = q + D;     // [4] per-wave maxima
    const u32 qi = blockIdx.x, tid = threadIdx.x;
    for (u32 i = tid; i < D; i += 256) q[i] = queries[(size_t)qi * D + i];
    __syncthreads();
    float s = 0.0f;
    for (u32 jq = 0; jq < m; jq++) {
        const float t = pw_run_lane(codebook + ((size_t)jq * 256 + tid) * sd, q + jq * sd, (int)sd);
        float mx = t;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) wmax[tid >> 6] = mx;
        __syncthreads();
        const float all = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        s = f_add(s, all);
        __syncthreads();
    }
    if (tid == 0) out[qi] = f_sqrt(s);
}

// squared norm of every stored vector (cosine traversal): one thread per row, double accumulator (any element order)
__global__ void row_norm2_kernel(const float *__restrict__ vecp, u64 N, u32 D, float *__restrict__ out)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        double s = 0.0;
        const float *r = vecp + i * D;
        for (u32 t = 0; t < D; t++) s += (double)r[t] * (double)r[t];
        out[i] = (float)s;
    }
}

// largest | |v|^2 - 1 | over the stored vectors, as the bits of a non-negative float (atomicMax on the bits orders like the values): DR_F_IP
__global__ void unit_norm_dev_kernel(const float *__restrict__ vecp, u64 N, u32 D, u32 *__restrict__ out_bits)
{
    float worst = 0.0f;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (u64)gridDim.x * blockDim.x) {
        double s = 0.0;
        const float *r = vecp + i * D;
        for (u32 t = 0; t < D; t++) s += (double)r[t] * (double)r[t];
        const float dev = (float)__builtin_fabs(s - 1.0);
        worst = dev > worst ? dev : (dev == dev ? worst : __builtin_inff());
    }
    atomicMax(out_bits, __float_as_uint(worst));
}

// The same bound with one LANE per (query, sub-quantiser): wavefront (q-tile, j) holds 64 queries' sub-vectors of j in
// registers and walks j's 256 centroids, which are the same for every lane -- scalar loads, SGPR operands, no LDS, no
// barriers -- keeping max_c T[j][c] per lane; pq_bound_sum_kernel then adds the m maxima of a query in A3's order. A batch
// of 10 000 queries is 157 x m short wavefronts (3.4 k instructions each) instead of 40 000 of 1.5 k with barriers: the
// bound shares the upload stream with the search kernel's tails, where wavefront slots are what is scarce. (One lane per
// query with the loop over j inside was tried first: 157 wavefronts of 107 k dependent instructions took 2 ms in the
// search kernel's shadow and stalled the pipeline -- 7.3 -> 4.4 M QPS, profiles/r03/ab/ab_c2_companions_v2_slow_bound.log.)
// Same entries (pw_run_regs: A2's order), same max, same sum order as pq_bound_kernel: the same bits.
template <int SD>
__global__ __launch_bounds__(64) void pq_bound_max_kernel(const float *__restrict__ codebook, const float *__restrict__ queries, u32 nq,
                                                          u32 D, float *__restrict__ out_max /*[m][nq]*/)
{
    const u32 qi = blockIdx.x * 64 + threadIdx.x, jq = blockIdx.y;
    const float *qrow = queries + (size_t)min(qi, nq - 1) * D + jq * SD;
    float qv[SD];
#pragma unroll
    for (int t = 0; t < SD; t++) qv[t] = qrow[t];
    float mx = 0.0f;          // (entries are sums of squares: >= 0)
    const float *cj = codebook + (size_t)jq * 256 * SD;
    for (u32 c = 0; c < 256; c++) {
        float cen[SD];
#pragma unroll
        for (int t = 0; t < SD; t++) cen[t] = cj[c * SD + t];      // uniform: scalar loads
        mx = fmaxf(mx, pw_run_regs<SD>(cen, qv));
    }
    if (qi < nq) out_max[(size_t)jq * nq + qi] = mx;
}
// The same maxima for a HANDFUL of queries (round 5): one 256-thread block per (query, sub-quantiser) -- thread c holds T[j][c] (pw_run_lane: pq_bound_kernel's
// entry), the block keeps its maximum. pq_bound_kernel walks the m sub-quantisers of a query one after the other, a dependent codebook read each: 23 us at
// D = 128 and 53 us at D = 1536 (sub_dim 48) in front of every small M1 call; here they run side by side and pq_bound_sum_kernel adds them in A3's order.
__global__ __launch_bounds__(256) void pq_bound_rowmax_kernel(const float *__restrict__ codebook, const float *__restrict__ queries, u32 nq, u32 D, u32 sd,
                                                              float *__restrict__ out_max /*[m][nq]*/)
{
    __shared__ float wmax[4];
    const u32 qi = blockIdx.x, jq = blockIdx.y, tid = threadIdx.x;
    float mx = pw_run_lane(codebook + ((size_t)jq * 256 + tid) * sd, queries + (size_t)qi * D + jq * sd, (int)sd);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) wmax[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) out_max[(size_t)jq * nq + qi] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}
__global__ void pq_bound_sum_kernel(const float *__restrict__ mx, u32 nq, u32 m, float *__restrict__ out)
{
    const u32 qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    float s = 0.0f;
    for (u32 jq = 0; jq < m; jq++) s = f_add(s, mx[(size_t)jq * nq + qi]);
    out[qi] = f_sqrt(s);
}

// adjr[i][s] = bit position of neighbour adj[i][s] in the visited bitmap (pad slots: 0)
__global__ void map_adjacency_kernel(const u32 *__restrict__ adj, u64 total, u64 N, const u32 *__restrict__ rank,
                                     u32 *__restrict__ adjr)
{
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u32 nb = adj[i];
        adjr[i] = nb < N ? rank[nb] : 0u;
    }
}

// Lossless byte copy of the vectors for the byte-row search variant (D = 128): out[row][j*16 + t] = element 8t + j
// when that element is an integer in [0, 255]; any other value raises `bad` and the copy is discarded.
__global__ void pack_u8_kernel(const float *__restrict__ vecp, u64 N, u32 D, const u32 *__restrict__ perm,
                               u8 *__restrict__ out, u32 *__restrict__ bad)
{
    const u32 S = D / 8;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < N * D; i += (u64)gridDim.x * blockDim.x) {
        const u64 row = i / D;
        const u32 e = (u32)(i % D);                       // original element index
        const float v = vecp[row * D + perm[e]];
        const float r = __builtin_rintf(v);
        if (!(v == r && v >= 0.0f && v <= 255.0f)) { atomicOr(bad, 1u); continue; }
        out[row * D + (e & 7u) * S + (e >> 3)] = (u8)(u32)r;
    }
}

// Inline neighbour codes (dr_index_inline_codes): nbcodes[i][s][0..m) = codes[adj[i][s]][0..m) -- the code words of a
// node's neighbours stored beside its adjacency row, so that an expansion reads them as ONE contiguous block of R*m
// bytes (fully coalesced, issued together with the row) instead of R scattered m-byte gathers behind the visited test.
// Slots that hold no real id (DR_PAD, ids >= N) get zeros; they are never scored.
__global__ void inline_codes_kernel(const u32 *__restrict__ adj, const u8 *__restrict__ codes, u64 N, u32 R, u32 m,
                                    u8 *__restrict__ nbcodes)
{
    const u32 wpr = m / 4;                                    // 4-byte words per code word (m % 4 == 0)
    const u64 total = N * R * wpr;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 slot = t / wpr;
        const u32 w = (u32)(t - slot * wpr);
        const u32 id = adj[slot];
        u32 v = 0u;
        if ((u64)id < N) v = reinterpret_cast<const u32 *>(codes + (size_t)id * m)[w];
        reinterpret_cast<u32 *>(nbcodes)[t] = v;
    }
}
