// latency_kernel.hpp -- the beam search for a HANDFUL of queries: one WORKGROUP of eight wavefronts per query (variant 18).
//
// The reference serves one query per request (search_engine.py:530-614, app.py:84-130). search_kernel.hpp walks such a query with ONE
// wavefront: ~45 dependent expansions of ~1 500 instructions and three memory round trips each, 6 us apiece on an otherwise idle
// chip (profiles/r05/phase_shares_single_query_m1.txt). Here the expansion is cut in two halves that do not depend on each other:
//
//   SCORING (all eight wavefronts, one per SIMD pair): the adjacency row of a frontier node, the exact distances of its
//     neighbours that are not visited yet (and their ADC for the rerank policy) -- pure functions of (query, node). A node's
//     rows are split over the wavefronts that have no node of their own, the results land in one of eight SLOTS in LDS
//     (ids, distance bits, sqrt-ADC bits in stored order).
//   DECISIONS (wavefront 0, which owns the result list in registers exactly as search_kernel.hpp does): pop, stop rule, visited
//     test-and-set, the rerank policy A4 and the accept / merge / trim of search_engine.py:398-506 -- the same code path, fed from
//     a slot instead of from memory.
//
// A round scores every live frontier entry that is not in a slot yet (beam_width 8 = the whole frontier), then wavefront 0 consumes
// pops for as long as the next pop has its slot; a pop without one (a neighbour that was inserted a moment ago) starts the next
// round. Scoring ahead is speculation on pure functions: the decisions see the same numbers in the same order, so ids, distances,
// counters and the accepted-insert log are those of search_kernel.hpp bit for bit (tests/test_gpu_latency.py holds every golden
// M1 / M2 fixture to it with the variant forced). The visited set is a hash set of ids in LDS (one query owns the CU): no visited
// words in HBM, no third round trip; a query that outgrows it continues in a per-workgroup table in global memory, and one that
// outgrows that too sets DR_ST_VIS_OVERFLOW (a blocking call is then served again through search_kernel.hpp).
// What makes it faster than search_kernel.hpp where it is: with the exact distances known before the rerank policy is
// asked, "the policy is true for this whole row" can be proven from the neighbours that can still enter the list (see need_adc below)
// instead of from all new ones -- with the API's L = 20 that is nearly every row against none.
#pragma once
#include "search_kernel.hpp"

#define DR_ST_VIS_OVERFLOW 1u       // dr_stats.status bit 0 (include/diskrag_hip.h: "visited-set overflow")
// -DDR_PHASE_TIMING: wavefront 0's shader-clock sums per query -> SearchParams::phase[qi][8]:
//   0 setup  1 decisions (pops consumed)  2 scheduling a round  3 the round (barrier to barrier)  4 wavefront 0's own scoring inside it
//   5 rounds  6 pops consumed  7 output
#ifdef DR_PHASE_TIMING
#define LT_NOW(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory")
#define LT_DECL() u64 lt_acc[8] = {0,0,0,0,0,0,0,0}; u64 lt_a = 0, lt_b = 0; LT_NOW(lt_a)
#define LT(i) do { LT_NOW(lt_b); lt_acc[i] += lt_b - lt_a; lt_a = lt_b; } while (0)
#define LT_COUNT(i) do { lt_acc[i] += 1; } while (0)
#define LT_END(qi) do { if (wave == 0 && lane == 0 && p.phase) for (int i_ = 0; i_ < 8; i_++) p.phase[(size_t)(qi) * 8 + i_] = lt_acc[i_]; } while (0)
#ifdef DR_LAT_SUB2
// (second diagnostic: the decisions of wavefront 0 cut up -- 0 peek / stop / slot lookup / commit, 2 slot read + visited test-and-set, 6 rerank-policy
//  set-up up to the candidate mask, 7 accept + merge, 4 trim; 3 counts the pops that took the accept-all path)
#define LM_BEGIN() LT_NOW(lt_a)
#define LM(i) LT(i)
#else
#define LM_BEGIN() do {} while (0)
#define LM(i) do {} while (0)
#endif
#else
#define LM_BEGIN() do {} while (0)
#define LM(i) do {} while (0)
#define LT_DECL() do {} while (0)
#define LT(i) do {} while (0)
#define LT_COUNT(i) do {} while (0)
#define LT_END(qi) do {} while (0)
#endif
#define LAT_NW 8                    // wavefronts per workgroup = slots = tasks per round
#define LAT_NONE 0xFFFFFFFFu

// ---- visited ids: open addressing, linear probing, empty = 0xFFFFFFFF (DR_PAD is never a node)
DEV bool vh_contains(const u32 *vh, u32 vmask, u32 vshift, u32 id)
{
    u32 h = (id * 0x9E3779B1u) >> vshift;
    bool found = false;
    for (u32 it = 0; it <= vmask; it++) {
        const u32 v = vh[h];
        if (v == id) { found = true; break; }
        if (v == LAT_NONE) break;
        h = (h + 1u) & vmask;
    }
    return found;
}
// true: id was not in the set (and is now). The table is never allowed to fill (see the capacity check of the caller).
DEV bool vh_insert(u32 *vh, u32 vmask, u32 vshift, u32 id)
{
    u32 h = (id * 0x9E3779B1u) >> vshift;
    bool isnew = false;
    for (u32 it = 0; it <= vmask; it++) {
        const u32 old = atomicCAS(&vh[h], LAT_NONE, id);
        if (old == LAT_NONE) { isnew = true; break; }
        if (old == id) break;
        h = (h + 1u) & vmask;
    }
    return isnew;
}

// The adjacency row (and first-occurrence mask) of a node that has just entered the list, requested while the decisions go on: nobody waits
// for it (no register destination: the words land in a dump area of LDS), and when the node is popped a moment later its row is an L2
// hit instead of the first HBM round trip of its scoring. Called with the accepted lanes active.
DEV void lat_prefetch_row(const SearchParams &p, u32 id, u32 *dump)
{
    const u32 *g = p.adj + (size_t)id * p.R;
    for (u32 o = 0; o < p.R; o += 32)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + o), (__attribute__((address_space(3))) void *)dump, 4, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.first + (size_t)id * ((p.R + 63) / 64)),
                                     (__attribute__((address_space(3))) void *)(dump + 64), 4, 0, 0);
}

// The same set continued in global memory once the LDS table holds its share (three quarters of its slots): one table of 2^sbits ids per
// workgroup, all 0xFFFFFFFF between queries (the workgroup that spilled wipes it behind the query). Ids never move: a lookup asks LDS first,
// then -- if the query has spilled -- the global table, with loads served by the L2 (a line this CU cached during an earlier query must not
// answer); inserts are compare-and-swaps in the L2. (Wavefronts that score read it without the decisions' inserts in flight; a stale miss
// would only cost a row that is scored in vain -- the decisions test-and-set every neighbour again.)
DEV bool vg_contains(const u32 *vg, u32 gmask, u32 gshift, u32 id)
{
    u32 h = (id * 0x9E3779B1u) >> gshift;
    bool found = false;
    for (u32 it = 0; it <= gmask; it++) {
        const u32 v = __hip_atomic_load(&vg[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == id) { found = true; break; }
        if (v == LAT_NONE) break;
        h = (h + 1u) & gmask;
    }
    return found;
}
DEV bool vg_insert(u32 *vg, u32 gmask, u32 gshift, u32 id)
{
    u32 h = (id * 0x9E3779B1u) >> gshift;
    bool isnew = false;
    for (u32 it = 0; it <= gmask; it++) {
        const u32 old = atomicCAS(&vg[h], LAT_NONE, id);
        if (old == LAT_NONE) { isnew = true; break; }
        if (old == id) break;
        h = (h + 1u) & gmask;
    }
    return isnew;
}

// ---- scoring: part `part` of `nparts` of node `node` into a slot
// The slot's ids / scored mask are written by part 0; distance and ADC bits by the part that owns the neighbour's row pass
// (compacted rows r with (r / 8) % nparts == part). Every wavefront of a node sees the same visited set (nobody inserts while
// a round is scored), hence the same compaction.
template <int D, bool FILTER>
DEV void lat_score(const SearchParams &p, u32 node, u32 *sl_ids, u32 *sl_eb, u32 *sl_ab, u64 *sl_mask, u64 *sl_rowmask, u32 wpub, bool wfull, int part, int nparts,
                   const u32 *vh, u32 vmask, u32 vshift, const u32 *vg, u32 gmask, u32 gshift, const float *lut, const QueryRegs<D> &qreg,
                   const float *qperm, u32 *nb_id, u32 *nb_ln, u32 knorm, u64 *lt_sub = nullptr)
{
#ifdef DR_PHASE_TIMING
    u64 st0 = 0, st1 = 0;
#define LS_BEGIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st0) :: "memory")
#define LS(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st1) :: "memory"); if (lt_sub) lt_sub[i] += st1 - st0; st0 = st1; } while (0)
#else
#define LS_BEGIN() do {} while (0)
#define LS(i) do {} while (0)
#endif
    LS_BEGIN();
    constexpr bool QREG = (D <= 256);
    constexpr bool SPLIT = QREG && split_form_ok<D>();
    const int lane = lane_id();
    const int j = lane & 7, oct = lane >> 3;
    const u32 nwords = (p.R + 63) / 64;
    for (u32 cbase = 0; cbase < p.R; cbase += 64) {
        const u32 slot = cbase + lane;
        const u32 sl = min(slot, p.R - 1);
        const u32 nbid = p.adj[(size_t)node * p.R + sl];
        const u64 aux = p.first[(size_t)node * nwords + (cbase >> 6)];
        const bool active = slot < p.R && ((aux >> lane) & 1ull) != 0ull;
        LS(0);
        const bool seen = active && (vh_contains(vh, vmask, vshift, nbid) || (vg != nullptr && vg_contains(vg, gmask, gshift, nbid)));
        const bool tofetch = active && !seen;
        const u64 fm0 = __ballot(tofetch);
        if (part == 0) {
            sl_ids[slot] = active ? nbid : LAT_NONE;
            if (lane == 0) sl_mask[cbase >> 6] = fm0;
        }
        constexpr bool PREFILTER = FILTER && D > 256;
        // (the second mask exists for the kernels that filter rows only: written unconditionally it cost the D <= 256 kernels 57 spilled registers)
        if (fm0 == 0ull) { if constexpr (PREFILTER) { if (part == 0 && lane == 0) sl_rowmask[cbase >> 6] = 0ull; } continue; }
        uint4 cw0 = make_uint4(0, 0, 0, 0), cw1 = cw0, cw2 = cw0, cw3 = cw0;
        const u8 *mycode = p.codes + (size_t)nbid * p.m;
        const bool lazy_adc = (p.vh_bits & 256u) != 0u;      // the rerank policy is (measured) proven true on this index: the few rows that ask compute their ADC in the decisions
        // Long rows (D > 256: a 6-KiB row at D = 1536): the policy's own filter BEFORE the rows, as search_kernel.hpp has it -- a neighbour whose
        // threshold x (A4 passes iff W > x) is not below the worst distance W that was in effect when the round was scheduled can never pass while
        // the list stays full (W only shrinks), so its row is not fetched (the decisions never look at the distance of such a lane). Every
        // part evaluates every lane's ADC (the lanes are there anyway), so all parts agree on the rows; it costs a third dependent round trip
        // (row ids -> code words -> rows), which the short rows do not repay -- they fetch code words and rows side by side.
        bool rowl = tofetch;
        if constexpr (PREFILTER) {
            if (!lazy_adc) {
                float pqd = 0.0f;
                if (tofetch) {
                    adc_load_codes(cw0, cw1, cw2, cw3, mycode, p.m);
                    pqd = f_sqrt(adc_compute<false>(lut, nullptr, p.sd, cw0, cw1, cw2, cw3, mycode, p.m));
                    if (part == 0) sl_ab[slot] = __float_as_uint(pqd);
                }
                if (wfull) {
                    bool ok = true;
                    const u32 xb = a4_threshold_bits(pqd, p.policy == 0u ? 1.2f : 0.8f, ok);
                    rowl = tofetch && (xb < wpub || !ok);
                }
            }
        }
        const u64 fm = __ballot(rowl);
        const int nrow = __popcll(fm);
        const int myrow = __popcll(fm & lanemask_lt());
        if constexpr (PREFILTER) { if (part == 0 && lane == 0) sl_rowmask[cbase >> 6] = fm; }
        if (nrow == 0) continue;
        const bool mine = rowl && ((myrow >> 3) & (nparts - 1)) == part;      // (nparts is 1, 2, 4 or 8)
        if constexpr (FILTER && !PREFILTER) { if (mine && !lazy_adc) adc_load_codes(cw0, cw1, cw2, cw3, mycode, p.m); }
        if (rowl) { nb_id[myrow] = nbid; nb_ln[myrow] = (u32)lane; }
        WSYNC();
        LS(2);
        const int npass = (nrow + 7) >> 3;
        bool rows_done = false;
        if constexpr (D == 128) {
            // the lossless byte copy of integer-valued rows (search_kernel.hpp "Byte rows"): one 128-byte line per row instead of four,
            // v_cvt_f32_ubyte gives back the stored float, the same sums in the same order
            if (p.vec8 != nullptr) {
                uint4 wv[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int r = part + i * nparts;
                    if (r < npass) wv[i] = *reinterpret_cast<const uint4 *>(p.vec8 + (size_t)nb_id[min(r * 8 + oct, nrow - 1)] * D + j * 16);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int r = part + i * nparts;
                    if (r < npass) {
                        const u32 words[4] = { wv[i].x, wv[i].y, wv[i].z, wv[i].w };
                        float rs = 0.0f;
#pragma unroll
                        for (int t = 0; t < 16; t++) {
                            const float v = (float)((words[t >> 2] >> (8 * (t & 3))) & 255u);
                            const float sq = sqd(v, qreg.v[t]);
                            rs = (t == 0) ? sq : f_add(rs, sq);
                        }
                        float ev = octet_combine(rs);
                        if (knorm) ev = f_sqrt(ev);
                        if (j == 0 && r * 8 + oct < nrow) sl_eb[cbase + nb_ln[r * 8 + oct]] = __float_as_uint(ev);
                    }
                }
                rows_done = true;
            }
        }
        if (rows_done) {
        } else if constexpr (SPLIT) {
            constexpr int NP = (D >= 256) ? 2 : 4;
            RowRegs<D> rr[NP];
#pragma unroll
            for (int i = 0; i < NP; i++) {
                const int r = part + i * nparts;
                if (r < npass) row_load<0, D, D>(p.vecp + (size_t)nb_id[min(r * 8 + oct, nrow - 1)] * D, j, rr[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int r = part + i * nparts;
                if (r < npass) {
                    float ev = row_reduce<0, D, D>(rr[i % NP], qreg);
                    if (knorm) ev = f_sqrt(ev);
                    if (j == 0 && r * 8 + oct < nrow) sl_eb[cbase + nb_ln[r * 8 + oct]] = __float_as_uint(ev);
                    const int rn = part + (i + NP) * nparts;
                    if (rn < npass) row_load<0, D, D>(p.vecp + (size_t)nb_id[min(rn * 8 + oct, nrow - 1)] * D, j, rr[i % NP]);
                }
            }
        } else if constexpr (ChunkCfg<D>::ok && !QREG) {
            ChunkRegs<D> cbuf[ChunkCfg<D>::NBUF];
            if (part < npass) {
                const float *rp = p.vecp + (size_t)nb_id[min(part * 8 + oct, nrow - 1)] * D;
                chunk_prologue<D, 0>(rp, j, cbuf);
                float cres[ChunkCfg<D>::NC];
#pragma unroll 1
                for (int r = part; r < npass; r += nparts) {
                    if (r + nparts < npass) {
                        const float *rnext = p.vecp + (size_t)nb_id[min((r + nparts) * 8 + oct, nrow - 1)] * D;
                        chunk_pass<D, 0, true>(rp, rnext, j, cbuf, qperm, cres);
                        rp = rnext;
                    } else {
                        chunk_pass<D, 0, false>(rp, rp, j, cbuf, qperm, cres);
                    }
                    float ev = chunk_tree<0, ChunkCfg<D>::NC>(cres);
                    if (knorm) ev = f_sqrt(ev);
                    if (j == 0 && r * 8 + oct < nrow) sl_eb[cbase + nb_ln[r * 8 + oct]] = __float_as_uint(ev);
                }
            }
        } else {
            for (int r = part; r < npass; r += nparts) {
                float ev = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)nb_id[min(r * 8 + oct, nrow - 1)] * D, &qreg, qperm, j);
                if (knorm) ev = f_sqrt(ev);
                if (j == 0 && r * 8 + oct < nrow) sl_eb[cbase + nb_ln[r * 8 + oct]] = __float_as_uint(ev);
            }
        }
        LS(7);
        if constexpr (FILTER && !(FILTER && D > 256)) {
            // asymmetric_distance = sqrt(sum_j T[j, code_j]) (fast_pq.py:320-333), from the query's table in LDS
            if (mine && !lazy_adc) sl_ab[slot] = __float_as_uint(f_sqrt(adc_compute<false>(lut, nullptr, p.sd, cw0, cw1, cw2, cw3, mycode, p.m)));
        }
        WSYNC();        // nb_id / nb_ln are rewritten by the next 64 slots of the row
        LS(6);
    }
}

// LDS footprint (the host computes the same sum: engine.hip lat_lds_bytes)
DEV size_t lat_slot_bytes(u32 R, bool filter) { const size_t rs = (size_t)((R + 63) / 64) * 64; return rs * 4 * (filter ? 3 : 2) + (size_t)((R + 63) / 64) * 16; }

template <int D, bool FILTER, int NCHR>
__global__ __launch_bounds__(64 * LAT_NW) void lat_kernel(const SearchParams p)
{
    constexpr bool QREG = (D <= 256);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int j = lane & 7;
    const u32 kmode = FILTER ? 1u : p.mode;
    const u32 knorm = FILTER ? 0u : p.norm;
    const u32 nwords = (p.R + 63) / 64;
    const u32 RS = nwords * 64;
    const int cap = (int)p.cap;

    // ---- LDS carve-up (every offset a multiple of 16 bytes)
    size_t off = 0;
    float *lut = reinterpret_cast<float *>(smem);
    if constexpr (FILTER) off += (size_t)p.m * 256 * 4;
    float *qperm = reinterpret_cast<float *>(smem + off);
    if constexpr (!QREG) off += (size_t)D * 4;
    u32 *vh = reinterpret_cast<u32 *>(smem + off);
    const u32 vbits = p.vh_bits & 255u;
    const bool lazy_adc = (p.vh_bits & 256u) != 0u;
    const u32 vslots = 1u << vbits, vmask = vslots - 1u, vshift = 32u - vbits;
    const u32 sbits = (p.vh_bits >> 16) & 255u;      // the workgroup's share of the spill area (0: none)
    const u32 gslots = sbits ? (1u << sbits) : 0u, gmask = gslots - 1u, gshift = 32u - sbits;
    u32 *vgw = sbits ? p.vis + ((size_t)blockIdx.x << sbits) : nullptr;
    off += (size_t)vslots * 4;
    const size_t slot_bytes = (lat_slot_bytes(p.R, FILTER) + 15) & ~(size_t)15;
    unsigned char *slots = smem + off;
    off += slot_bytes * LAT_NW;
    u32 *nb_id = reinterpret_cast<u32 *>(smem + off + (size_t)wave * 512);
    u32 *nb_ln = nb_id + 64;
    off += (size_t)LAT_NW * 512;
    u64 *mk = reinterpret_cast<u64 *>(smem + off);
    u32 *mf = reinterpret_cast<u32 *>(mk + NCHR * 64);
    off += (size_t)NCHR * 64 * 12;
    u32 *ctl = reinterpret_cast<u32 *>(smem + off);      // [0] tasks of the round (0: the query is finished); [8..15] task node; [16..23] task slot | part << 8 | nparts << 16;
                                                          // [24..31] wanted ids, [32..39] node / [40..47] slot of the wanted that need scoring, [48..55] slot -> new tag
                                                          // [64..191] dump area of the adjacency prefetch, [192..447] row maxima of the query's table (fused ADC bound)
    auto slot_ids = [&](int s) { return reinterpret_cast<u32 *>(slots + (size_t)s * slot_bytes); };
    auto slot_eb = [&](int s) { return slot_ids(s) + RS; };
    auto slot_ab = [&](int s) { return slot_ids(s) + 2 * RS; };
    auto slot_mask = [&](int s) { return reinterpret_cast<u64 *>(slot_ids(s) + (FILTER ? 3 : 2) * RS); };      // [nwords] not visited when scored, then [nwords] rows scored
    // nodes wanted per round: with long rows (D > 256) the popped node and, with the rerank policy, one more -- a node scored by one wavefront
    // alone IS the round's length there (measured at D = 1536, profiles/r05/latency_embeddings_sweep.json: 2 beats 1 / 3 / 4 / 8 for M1, 1 beats 2
    // for the exact traversals); bits 24..27 of vh_bits override it (DR_LAT_WANT, A/B). (A deeper row pipeline for the one pass a wavefront
    // scores -- all eight chunks of a 6-KiB row in flight, DR_CHUNK_NBUF=8 -- spills 65 registers around the list and is 15-30 % slower.)
    const int WANT = ((p.vh_bits >> 24) & 15u) ? (int)min((p.vh_bits >> 24) & 15u, (u32)LAT_NW) : ((D > 256) ? (FILTER ? 2 : 1) : LAT_NW);

    for (u32 qi = blockIdx.x; qi < p.nq; qi += gridDim.x) {
        LT_DECL();
        // ---- per-query setup, all wavefronts
        {
            uint4 *v4 = reinterpret_cast<uint4 *>(vh);
            for (u32 i = threadIdx.x; i < vslots / 4; i += 64 * LAT_NW) v4[i] = make_uint4(LAT_NONE, LAT_NONE, LAT_NONE, LAT_NONE);
        }
        if constexpr (FILTER) {
            // the query's table T[j][c] (A2, built for the batch by lut_build_kernel): m KiB landed in LDS, 1 KiB per wave instruction
            const float *tg = p.lut_g + (size_t)qi * p.m * 256 + lane * 4;
            for (u32 e = (u32)wave * 256; e < p.m * 256; e += 256 * LAT_NW)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tg + e),
                    (__attribute__((address_space(3))) void *)(lut + e), 16, 0, 0);
        }
        QueryRegs<D> qreg;
        {
            const float *qg = p.queries + (size_t)qi * D;
            const float *qpg = p.queries_p + (size_t)qi * D;
            // (no chain-major copy of the batch: the element permutation is applied here -- one launch less in front of a one-query call)
            if constexpr (!QREG) {
                if (p.queries_p != nullptr) { for (int i = threadIdx.x; i < D; i += 64 * LAT_NW) qperm[i] = qpg[i]; }
                else { for (int i = threadIdx.x; i < D; i += 64 * LAT_NW) qperm[p.perm[i]] = qg[i]; }
            }
            if constexpr (QREG) {
                if (p.queries_p != nullptr) load_query_regs<0, D, D>(qpg, j, qreg);
                else load_query_regs_orig<0, D, D>(qg, j, qreg);
            }
        }
        if (threadIdx.x == 0) ctl[1] = 0u;      // 1: this query's visited ids continue in the global table
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- state of the search (wavefront 0)
        bool spilled = false;
        u32 nlds = 1;      // ids in the LDS table (the start node)
        float pq_ub = __uint_as_float(0x7F800000u);
        u32 npq_eval = 0, steps = 0, nvisited = 0, nexact = 0, npq = 0, status = 0, ninserts = 0, nhits = 0;
        int rn = 0, cnT = 0, tn = 0;
        u32 junk = 0;
        RegList<NCHR> rk;
        FlagList<NCHR> fl;
        RegList<1> tl;
        u32 tags = LAT_NONE;        // lane s < LAT_NW: the node whose scored row sits in slot s
        bool finished = false;
        u64 *qlog = p.log + (size_t)qi * p.logcap;
        // a set that is three quarters full is an overflow (probe sequences stay short, the table never fills)
        const u32 vlimit = vslots - (vslots >> 2);
        if constexpr (FILTER) {
            // sqrt(sum_j max_c T[j][c]), the sum in A3's order (pq_bound_kernel's value: the same table entries, the same additions) from the table
            // in LDS when the engine did not queue the bound kernels: row j's maximum by wavefront j mod 8, the ordered sum by one lane
            if (p.pq_ub == nullptr) {
                float *rowmax = reinterpret_cast<float *>(ctl + 192);
                for (u32 jq = (u32)wave; jq < p.m; jq += LAT_NW) {
                    const float4 t4 = reinterpret_cast<const float4 *>(lut + (size_t)jq * 256)[lane];
                    const float mx = wave_max(fmaxf(fmaxf(t4.x, t4.y), fmaxf(t4.z, t4.w)));
                    if (lane == 0) rowmax[jq] = mx;
                }
                __syncthreads();
                if (wave == 0) {
                    float sacc = 0.0f;
                    for (u32 jq = 0; jq < p.m; jq++) sacc = f_add(sacc, rowmax[jq]);
                    pq_ub = f_sqrt(sacc);
                }
            }
        }
        if (wave == 0) {
            if constexpr (FILTER) { if (p.pq_ub != nullptr) pq_ub = p.pq_ub[qi]; }
#pragma unroll
            for (int c = 0; c < NCHR; c++) { rk.v[c] = ~0ull; fl.v[c] = 0u; }
            tl.v[0] = ~0ull;
            // start node (search_engine.py:416-426)
            const u32 start = p.medoid;
            if (lane == 0) (void)vh_insert(vh, vmask, vshift, start);
            nvisited = 1;
            float d0 = pw_row_stream<0, D, D, QREG>(p.vecp + (size_t)start * D, &qreg, qperm, j);
            d0 = __uint_as_float(readlane32(__float_as_uint(d0), 0));
            if (knorm) d0 = f_sqrt(d0);
            nexact++;
            const u32 db = __float_as_uint(d0);
            u64 dr; u32 df; bool dd;
            rn = list_insert_f<NCHR>(rk, fl, 0, cap, ((u64)db << 32) | (u32)(~start), dr, df, dd);
            cnT = 1;
            if (lane == 0 && p.logcap > 0) qlog[0] = ((u64)db << 32) | start;
            ninserts = 1;
        }

        // (a round per pop at most; counted loops only -- search_kernel.hpp's hipcc note)
        const u32 rounds_max = (u32)min((u64)p.max_steps, p.N + 16ull) + 2u;
        LT(0);
        for (u32 round = 0; round < rounds_max; round++) {
            if (wave == 0) {
                u32 ntasks = 0;
                // ---- consume pops while the next one has its scored row
                for (u32 guard = 0; !finished; guard++) {
                    if (!((cnT + tn > 0 || junk > 0) && steps < p.max_steps)) { finished = true; break; }
                    if ((u64)steps > p.N + 8 || guard > p.max_steps) { status |= DR_ST_INTERNAL; finished = true; break; }
                    if (cnT + tn == 0) { steps++; finished = true; break; }   // only junk left: the reference pops it and stops (it is worse than W)
                    if (status & DR_ST_VIS_OVERFLOW) { finished = true; break; }
                    LM_BEGIN();
                    // heappop(candidates), not committed yet: the smaller of the first live result entry and the side-list head
                    const int ia = frontier_first<NCHR>(rk, fl, rn);
                    const u64 ka = (ia >= 0) ? fkey(list_get<NCHR>(rk, ia)) : ~0ull;
                    const u64 kb = (tn > 0) ? readlane64(tl.v[0], 0) : ~0ull;
                    const bool from_list = ka <= kb;
                    const u64 ckey = from_list ? ka : kb;
                    const float cd = key_dist(ckey);
                    const u32 cur = (u32)ckey;
                    {
                        const float W = key_dist(list_get<NCHR>(rk, rn - 1));
                        bool stop;
                        if (kmode == 3u) stop = (cd > W) && (rn == cap);
                        else if (kmode == 4u) stop = (cd > W);
                        else stop = (rn >= cap) && (cd > W);
                        if (stop) { steps++; finished = true; break; }
                    }
                    const u64 hm = __ballot(lane < LAT_NW && tags == cur);
                    if (hm == 0ull) {
                        LT(1);
                        // ---- no scored row for this pop: schedule a round. Wanted = this node + the first live entries of the list
                        // (the frontier after a trim IS at most beam_width entries); rows already in a slot stay, the others are scored.
                        int base = 0;
#pragma unroll
                        for (int c = 0; c < NCHR; c++) {
                            const bool live = (c * 64 + lane < rn) && fl.v[c] == 0u;
                            const u64 lm = __ballot(live);
                            const int rank = base + __popcll(lm & lanemask_lt());
                            if (live && rank < WANT) ctl[24 + rank] = ~(u32)rk.v[c];
                            base += __popcll(lm);
                        }
                        int nw = min(base, WANT);
                        if (lane == 0) { ctl[2] = (u32)(list_get<NCHR>(rk, rn - 1) >> 32); ctl[3] = (rn == cap) ? 1u : 0u; }      // the worst distance the round's policy filter may rely on
                        WSYNC();
                        u32 wid = lane < nw ? ctl[24 + min(lane, LAT_NW - 1)] : LAT_NONE;
                        if (__ballot(lane < nw && wid == cur) == 0ull) {
                            const int pos = min(nw, WANT - 1);
                            if (lane == pos) wid = cur;
                            nw = pos + 1;
                        }
                        const bool valid = lane < nw;
                        u32 keep = 0u;
                        u64 sat = 0ull;
#pragma unroll
                        for (int s = 0; s < LAT_NW; s++) {
                            const u32 ts = readlane32(tags, s);
                            const u64 mm = __ballot(valid && wid == ts);
                            if (ts != LAT_NONE && mm != 0ull) { keep |= 1u << s; sat |= mm; }
                        }
                        const u64 unsat = __ballot(valid) & ~sat;
                        const int nt = __popcll(unsat);
                        const bool isun = ((unsat >> lane) & 1ull) != 0ull;
                        const int myr = __popcll(unsat & lanemask_lt());
                        int myslot = -1;
                        {
                            int cnt = 0;
#pragma unroll
                            for (int s = 0; s < LAT_NW; s++)
                                if (((keep >> s) & 1u) == 0u) { if (cnt == myr && myslot < 0) myslot = s; cnt++; }
                        }
                        if (lane < LAT_NW) ctl[48 + lane] = LAT_NONE;
                        WSYNC();
                        if (isun && myslot >= 0) { ctl[32 + myr] = wid; ctl[40 + myr] = (u32)myslot; ctl[48 + myslot] = wid; }
                        WSYNC();
                        const int nparts = nt <= 1 ? 8 : nt == 2 ? 4 : nt <= 4 ? 2 : 1;
                        if (lane < LAT_NW) {
                            if (((keep >> lane) & 1u) == 0u) tags = ctl[48 + lane];
                            const int tr = lane / nparts, tq = lane % nparts;
                            const bool tv = tr < nt;
                            ctl[8 + lane] = tv ? ctl[32 + min(tr, LAT_NW - 1)] : LAT_NONE;
                            ctl[16 + lane] = tv ? (ctl[40 + min(tr, LAT_NW - 1)] | ((u32)tq << 8) | ((u32)nparts << 16)) : 0u;
                        }
                        ntasks = (u32)(nt * nparts);
                        if (nt < 1) { status |= DR_ST_INTERNAL; finished = true; }
                        LT(2); LT_COUNT(5);
                        break;
                    }
                    LT_COUNT(6);
                    const int s = __ffsll((long long)hm) - 1;
                    if (guard > 0) nhits++;
                    // ---- commit the pop
                    steps++;
                    if (from_list) { flag_or<NCHR>(fl, ia, 1u); cnT--; }
                    else { list_pop_front<1>(tl); tn--; }
                    tags = (lane == s) ? LAT_NONE : tags;
                    const u32 *s_ids = slot_ids(s), *s_eb = slot_eb(s), *s_ab = slot_ab(s);
                    const u64 *s_mask = slot_mask(s);
                    LM(0);

                    for (u32 cbase = 0; cbase < p.R; cbase += 64) {
                        const u32 myid = s_ids[cbase + lane];
                        const bool active = myid != LAT_NONE;
                        // visited test-and-set (search_engine.py:444-448), in stored order by construction: ids of a row are distinct
                        if (!spilled && nlds + 64u > vlimit) {
                            if (vgw == nullptr) { status |= DR_ST_VIS_OVERFLOW; break; }
                            spilled = true;
                            if (lane == 0) ctl[1] = 1u;
                        }
                        if (spilled && (nvisited - nlds) + 64u > gslots - (gslots >> 2)) { status |= DR_ST_VIS_OVERFLOW; break; }
                        // (a lane that was visited when the row was scored still is)
                        const bool totest = active && ((s_mask[cbase >> 6] >> lane) & 1ull) != 0ull;
                        bool isnew;
                        if (!spilled) isnew = totest && vh_insert(vh, vmask, vshift, myid);
                        else isnew = totest && !vh_contains(vh, vmask, vshift, myid) && vg_insert(vgw, gmask, gshift, myid);
                        const u64 newmask = __ballot(isnew);
                        const int nnew = __popcll(newmask);
                        LM(2);
                        if (nnew == 0) continue;
                        nvisited += nnew;
                        if (!spilled) nlds += nnew;
                        // (a new lane whose row the round's policy filter left out has no distance: the decisions never look at it -- its
                        //  threshold is not below the worst distance, so it is no candidate and the sharper proof below fails with it)
                        bool rowscored = true;      // (without the filter every lane that was not visited when the row was scored has its distance)
                        if constexpr (FILTER && D > 256) rowscored = ((s_mask[nwords + (cbase >> 6)] >> lane) & 1ull) != 0ull;
                        const float e = (isnew && rowscored) ? __uint_as_float(s_eb[cbase + lane]) : __builtin_inff();
                        // Is the ADC value of this expansion's neighbours needed at all? (search_kernel.hpp: A4 is provably True for all
                        // of them when the list cannot fill up during the expansion, or when pq_ub clears the threshold for the smallest
                        // worst distance the expansion can reach)
                        // Here the exact distances are known BEFORE the policy is asked, which sharpens the second test: while the list is full the
                        // worst distance only shrinks, so a neighbour is accepted only if its distance is below today's worst -- at most c' of them
                        // (those lanes), at most c' results are replaced and the worst distance stays >= list[cap - 1 - c'] throughout the row.
                        // (search_kernel.hpp has to assume nnew replacements: with the API's L = 20 it can prove nothing and evaluates the policy
                        // for every row; here those rows take the proven-true path as the L = 100 ones do.)
                        bool need_adc = FILTER;
                        if constexpr (FILTER) {
                            if (rn + nnew <= (int)p.L) need_adc = false;
                            else if (rn == cap) {
                                const u32 Wb = (u32)(list_get<NCHR>(rk, rn - 1) >> 32);
                                const int cpr = __popcll(__ballot(isnew && __float_as_uint(e) < Wb));
                                if (cap - 1 - cpr >= 0) {
                                    const float Wlow = key_dist(list_get<NCHR>(rk, cap - 1 - cpr));
                                    if (pq_ub < f_mul(Wlow, 0.8f)) need_adc = false;
                                }
                            }
                        }
                        const bool all_pass = !need_adc;
                        u32 xbits = 0u;
                        if constexpr (FILTER) {
                            if (need_adc) {
                                float pq_d = 0.0f;
                                if (!lazy_adc) pq_d = isnew ? __uint_as_float(s_ab[cbase + lane]) : 0.0f;
                                else if (isnew) {
                                    uint4 cw0 = make_uint4(0, 0, 0, 0), cw1 = cw0, cw2 = cw0, cw3 = cw0;
                                    const u8 *mycode = p.codes + (size_t)myid * p.m;
                                    adc_load_codes(cw0, cw1, cw2, cw3, mycode, p.m);
                                    pq_d = f_sqrt(adc_compute<false>(lut, nullptr, p.sd, cw0, cw1, cw2, cw3, mycode, p.m));
                                }
                                bool ok = true;
                                xbits = a4_threshold_bits(pq_d, p.policy == 0u ? 1.2f : 0.8f, ok);
                                if (__ballot(isnew && !ok) != 0ull) status |= DR_ST_INTERNAL;
                                npq_eval += nnew;
                            }
                            npq += nnew;
                        }

                        // ---- decisions (search_kernel.hpp "decisions": the reference's stored-order walk without a walk)
                        const bool count_pass = FILTER && !all_pass;
                        if (!count_pass) nexact += nnew;
                        const u32 ebits = __float_as_uint(e);
                        const u32 tbits = max(ebits, xbits);
                        const bool full0 = (rn == cap);
                        const u32 W0b = (u32)(list_get<NCHR>(rk, rn - 1) >> 32);
                        const u64 cm = __ballot(isnew && (!full0 || (count_pass ? xbits : ebits) < W0b));
                        const u64 mykey = ((u64)ebits << 32) | (u32)(~myid);
                        int na = 0;
                        u64 accmask = 0ull;
                        int rT = 0, rA = 0;
                        u32 sT[NCHR];
#pragma unroll
                        for (int ch = 0; ch < NCHR; ch++) sT[ch] = 0u;
                        // ---- the common case without the general machinery: the rerank policy is proven true for the whole row (or absent) and
                        // EVERY candidate is accepted whatever the order. With c candidates and d = rn + c - cap list entries to drop, a candidate i is
                        // accepted iff #(S_i <= e_i) < cap (search_kernel.hpp); when every candidate is below list[rn - d] at least d list entries are
                        // above it and at most c - 1 earlier candidates below: #(S_i <= e_i) <= (rn - d) + (c - 1) = cap - 1. If moreover list[rn - d]
                        // is strictly above list[rn - d - 1], no dropped entry ties with the new worst distance: the live ones among them only
                        // count (junk), the side list stays as it is. One loop over the candidates gives the merge ranks.
                        bool fast_done = false;
                        LM(6);
#ifdef DR_LAT_SUB2
                        { const int c_ = __popcll(cm); npq_eval += (c_ == 0) ? 1u : (c_ <= 2) ? (1u << 8) : (1u << 16); }
#endif
                        if (!count_pass && cm != 0ull && __popcll(cm) <= 24) {
                            const int c = __popcll(cm);
                            const int d = max(0, rn + c - cap);
                            const bool iscand = ((cm >> lane) & 1ull) != 0ull;
                            bool fast = true;
                            if (d > 0) {
                                const u32 emax = wave_max_u32(iscand ? ebits : 0u);
                                const u32 kd = (u32)(list_get<NCHR>(rk, rn - d) >> 32);
                                fast = emax < kd;
                                if (rn - d - 1 >= 0) fast = fast && (u32)(list_get<NCHR>(rk, rn - d - 1) >> 32) < kd;
                            }
                            if (fast) {
                                u32 lessc = 0u, rTc = 0u, sTc[NCHR];
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) sTc[ch] = 0u;
                                for (u64 mm = cm; mm != 0ull; mm &= mm - 1ull) {
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
                                {
                                    const u32 o = ninserts + (u32)__popcll(cm & lanemask_lt());
                                    if (iscand && o < p.logcap) qlog[o] = ((u64)ebits << 32) | myid;
                                    if (ninserts + (u32)c > p.logcap && p.logcap > 0) status |= DR_ST_LOG_OVERFLOW;
                                    ninserts += (u32)c;
                                }
                                const int keep = rn - d, rn2 = rn + c - d;
                                int nlive_out = 0;
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) {
                                    const int idx = ch * 64 + lane;
                                    if (idx < keep) { mk[idx + (int)sTc[ch]] = rk.v[ch]; mf[idx + (int)sTc[ch]] = fl.v[ch]; }
                                    if (d > 0) nlive_out += __popcll(__ballot(idx >= keep && idx < rn && fl.v[ch] == 0u));
                                }
                                if (iscand) { mk[rTc + lessc] = mykey; mf[rTc + lessc] = 0u; lat_prefetch_row(p, myid, ctl + 64); }
                                WSYNC();
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) {
                                    const int idx = ch * 64 + lane;
                                    rk.v[ch] = (idx < rn2) ? mk[idx] : ~0ull;
                                    fl.v[ch] = (idx < rn2) ? mf[idx] : 0u;
                                }
                                junk += (u32)nlive_out;
                                cnT += c - nlive_out;
                                rn = rn2;
                                WSYNC();
                                fast_done = true;

                            }
                        }
                        // ---- the policy proven true (or absent), but not every candidate surely accepted: without A4 the stored-order rule needs no
                        // fixed point -- i is accepted iff #(list <= e_i) + #(candidates j < i with e_j <= e_i) < cap (a candidate rejected earlier had
                        // e_j >= W_j >= W_i > e_i: search_kernel.hpp "decisions" (3)) -- one loop over the candidates for the verdicts, one over the
                        // accepted for the merge ranks; the merge below then handles evictions and ties as ever.
                        bool ranks_done = false;
                        if (!count_pass && cm != 0ull && !fast_done && __popcll(cm) > 8) {
                            // many candidates (a filling list, the descent towards the query): the same two counts as all-pairs loops over the
                            // candidate keys staged in LDS -- broadcast reads and independent compares instead of a readlane round per candidate --
                            // and the counts against the list by binary search over its staged copy
                            const int c = __popcll(cm);
                            const bool iscand = ((cm >> lane) & 1ull) != 0ull;
                            const int pc = __popcll(cm & lanemask_lt());
                            u64 *ck = reinterpret_cast<u64 *>(nb_id);      // wavefront 0's scoring scratch (512 bytes), idle during the decisions
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) if (ch * 64 + lane < rn) mk[ch * 64 + lane] = rk.v[ch];
                            if (iscand) ck[pc] = mykey;
                            WSYNC();
                            int lb_lo = 0, lb_hi = rn, ut_lo = 0, ut_hi = rn;
                            const u64 key_ut = ((u64)ebits << 32) | 0xFFFFFFFFull;
                            if (iscand) {
                                constexpr int ITER = (NCHR == 1) ? 7 : (NCHR == 2) ? 8 : (NCHR == 4) ? 9 : (NCHR == 8) ? 10 : 11;
#pragma unroll
                                for (int it = 0; it < ITER; it++) {
                                    const int m1 = (lb_lo + lb_hi) >> 1, m2 = (ut_lo + ut_hi) >> 1;
                                    const u64 v1 = mk[min(m1, rn - 1)], v2 = mk[min(m2, rn - 1)];
                                    if (lb_lo < lb_hi) { if (v1 < mykey) lb_lo = m1 + 1; else lb_hi = m1; }
                                    if (ut_lo < ut_hi) { if (v2 <= key_ut) ut_lo = m2 + 1; else ut_hi = m2; }
                                }
                            }
                            u32 before = 0u;
#pragma unroll 4
                            for (int t = 0; t < c; t++) { const u32 ef = (u32)(ck[t] >> 32); before += (t < pc && ef <= ebits) ? 1u : 0u; }
                            accmask = __ballot(iscand && (u32)ut_lo + before < (u32)cap);
                            na = __popcll(accmask);
                            WSYNC();
                            if (((accmask >> lane) & 1ull) != 0ull) ck[__popcll(accmask & lanemask_lt())] = mykey;
                            WSYNC();
                            u32 lessc = 0u;
#pragma unroll 4
                            for (int t = 0; t < na; t++) {
                                const u64 kf = ck[t];
                                lessc += (kf < mykey) ? 1u : 0u;
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) sT[ch] += (kf < rk.v[ch]) ? 1u : 0u;
                            }
                            rT = lb_lo; rA = (int)lessc;
                            ranks_done = true;
                            WSYNC();
                        }
                        if (!count_pass && cm != 0ull && !fast_done && !ranks_done) {
                            const bool iscand = ((cm >> lane) & 1ull) != 0ull;
                            u32 before = 0u, ut = 0u;
                            for (u64 mm = cm; mm != 0ull; mm &= mm - 1ull) {
                                const int f = __ffsll((long long)mm) - 1;
                                const u32 ef = readlane32(ebits, f);
                                before += (f < lane && ef <= ebits) ? 1u : 0u;
                                u32 cnt = 0u;
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) cnt += (u32)__popcll(__ballot((u32)(rk.v[ch] >> 32) <= ef && ch * 64 + lane < rn));
                                ut = (lane == f) ? cnt : ut;
                            }
                            accmask = __ballot(iscand && ut + before < (u32)cap);
                            na = __popcll(accmask);
                            u32 lessc = 0u, rTc = 0u;
                            for (u64 mm = accmask; mm != 0ull; mm &= mm - 1ull) {
                                const int f = __ffsll((long long)mm) - 1;
                                const u64 kf = readlane64(mykey, f);
                                lessc += (kf < mykey) ? 1u : 0u;
                                u32 cnt = (u32)NCHR * 64u;
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) {
                                    sT[ch] += (kf < rk.v[ch]) ? 1u : 0u;
                                    cnt -= (u32)__popcll(__ballot(kf < rk.v[ch]));      // (keys are distinct, unused slots hold ~0: list entries below kf)
                                }
                                rTc = (lane == f) ? cnt : rTc;
                            }
                            rT = (int)rTc; rA = (int)lessc;
                            ranks_done = true;
                        }
                        LM(7);
#ifdef DR_LAT_SUB2
                        if (cm != 0ull && !fast_done && !ranks_done) npq_eval += (1u << 24);
#endif
                        if (cm != 0ull && !fast_done && !ranks_done) {
                            // (more than eight candidates: the pair masks below are built from candidate keys staged in LDS -- broadcast reads and
                            //  independent compares, no readlane round per candidate -- and the counts against the list by binary search)
                            const bool lds_pairs = __popcll(cm) > 8;
                            const bool by_ballot = !lds_pairs && __popcll(cm) * (NCHR + 2) <= 48;
                            if (!by_ballot) {
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) if (ch * 64 + lane < rn) mk[ch * 64 + lane] = rk.v[ch];
                                WSYNC();
                            }
                            const bool iscand = ((cm >> lane) & 1ull) != 0ull;
                            int lb_lo = 0, lb_hi = rn, ut_lo = 0, ut_hi = rn, ux_lo = 0, ux_hi = rn;
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
                            }
                            u64 Mt = 0ull, Mx = 0ull, lessm = 0ull, oldm[NCHR];
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) oldm[ch] = 0ull;
                            if (lds_pairs) {
                                const int c = __popcll(cm);
                                u64 *ck = reinterpret_cast<u64 *>(nb_id);      // wavefront 0's scoring scratch: candidate keys in stored order
                                u32 *cl = mf;                                   // ... and their lanes (the merge flags' scratch is idle until the merge)
                                if (iscand) { const int pc = __popcll(cm & lanemask_lt()); ck[pc] = mykey; cl[pc] = (u32)lane; }
                                WSYNC();
#pragma unroll 4
                                for (int t = 0; t < c; t++) {
                                    const u64 kf = ck[t];
                                    const u32 lf = cl[t];
                                    const u32 ef = (u32)(kf >> 32);
                                    const u64 bit = 1ull << lf;
                                    const bool earlier = lf < (u32)lane;
                                    Mt |= (earlier && ef <= tbits) ? bit : 0ull;
                                    if (count_pass) Mx |= (earlier && ef <= xbits) ? bit : 0ull;
                                    lessm |= (kf < mykey) ? bit : 0ull;
#pragma unroll
                                    for (int ch = 0; ch < NCHR; ch++) oldm[ch] |= (kf < rk.v[ch]) ? bit : 0ull;
                                }
                                WSYNC();
                            } else
#pragma unroll
                            for (int half = 0; half < 2; half++) {
                                u32 mt = 0u, mx = 0u, ls = 0u, om[NCHR];
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) om[ch] = 0u;
                                for (u32 mm = (u32)(cm >> (32 * half)); mm != 0u; mm &= mm - 1u) {
                                    const int fl_ = __ffs((int)mm) - 1;
                                    const int f = fl_ + 32 * half;
                                    const u32 ef = readlane32(ebits, f);
                                    const u64 kf = readlane64(mykey, f);
                                    const u32 bit = 1u << fl_;
                                    mt |= (f < lane && ef <= tbits) ? bit : 0u;
                                    if (count_pass) mx |= (f < lane && ef <= xbits) ? bit : 0u;
                                    ls |= (kf < mykey) ? bit : 0u;
#pragma unroll
                                    for (int ch = 0; ch < NCHR; ch++) om[ch] |= (kf < rk.v[ch]) ? bit : 0u;
                                    if (by_ballot) {
                                        const u64 kut = ((u64)readlane32(tbits, f) << 32) | 0xFFFFFFFFull;
                                        const u64 kux = ((u64)readlane32(xbits, f) << 32) | 0xFFFFFFFFull;
                                        int c_lb = NCHR * 64, c_ut = 0, c_ux = 0;
#pragma unroll
                                        for (int ch = 0; ch < NCHR; ch++) {
                                            c_lb -= __popcll(__ballot(kf < rk.v[ch]));
                                            c_ut += __popcll(__ballot(rk.v[ch] <= kut));
                                            if (count_pass) c_ux += __popcll(__ballot(rk.v[ch] <= kux));
                                        }
                                        lb_lo = (lane == f) ? c_lb : lb_lo;
                                        ut_lo = (lane == f) ? c_ut : ut_lo;
                                        if (count_pass) ux_lo = (lane == f) ? c_ux : ux_lo;
                                    }
                                }
                                Mt |= (u64)mt << (32 * half); Mx |= (u64)mx << (32 * half); lessm |= (u64)ls << (32 * half);
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) oldm[ch] |= (u64)om[ch] << (32 * half);
                            }
                            const bool canacc = iscand && (!full0 || tbits < W0b);
                            accmask = __ballot(canacc);
                            for (int rnd = 0; rnd < 66; rnd++) {
                                const u64 nxt = __ballot(canacc && (u32)ut_lo + (u32)__popcll(Mt & accmask) < (u32)cap);
                                if (nxt == accmask) break;
                                accmask = nxt;
                                if (rnd == 65) status |= DR_ST_INTERNAL;
                            }
                            na = __popcll(accmask);
                            if (count_pass)
                                nexact += (u32)__popcll(__ballot(iscand && (u32)ux_lo + (u32)__popcll(Mx & accmask) < (u32)cap));
                            rT = lb_lo;
                            rA = __popcll(lessm & accmask);
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) sT[ch] = (u32)__popcll(oldm[ch] & accmask);
                            WSYNC();
                        }
                        if (na > 0) {
                            const bool isacc = ((accmask >> lane) & 1ull) != 0ull;
                            {
                                const u32 o = ninserts + (u32)__popcll(accmask & lanemask_lt());
                                if (isacc && o < p.logcap) qlog[o] = ((u64)ebits << 32) | myid;
                                if (ninserts + (u32)na > p.logcap && p.logcap > 0) status |= DR_ST_LOG_OVERFLOW;
                                ninserts += (u32)na;
                            }
                            const int rn2 = min(rn + na, cap);
                            int npT[NCHR];
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) {
                                const int idx = ch * 64 + lane;
                                npT[ch] = idx + (int)sT[ch];
                                if (idx < rn && npT[ch] < cap) { mk[npT[ch]] = rk.v[ch]; mf[npT[ch]] = fl.v[ch]; }
                            }
                            const int npA = rT + rA;
                            if (isacc && npA < cap) { mk[npA] = mykey; mf[npA] = 0u; lat_prefetch_row(p, myid, ctl + 64); }
                            WSYNC();
                            const u32 Wfb = (u32)(mk[rn2 - 1] >> 32);
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
#pragma unroll
                            for (int ch = 0; ch < NCHR; ch++) {
                                const int idx = ch * 64 + lane;
                                rk.v[ch] = (idx < rn2) ? mk[idx] : ~0ull;
                                fl.v[ch] = (idx < rn2) ? mf[idx] : 0u;
                            }
                            rn = rn2;
                            WSYNC();
                        }
                    }

                    LM(1);
                    // ---- frontier trim (search_kernel.hpp, unchanged)
                    if (kmode == 1u || kmode == 2u || kmode == 5u) {
                        if (p.bw != 0u && (u32)(cnT + tn) + junk > p.bw) {
                            u32 excess = (u32)(cnT + tn) + junk - p.bw;
                            const u32 rj = excess < junk ? excess : junk;
                            junk -= rj; excess -= rj;
                            bool done_trim = false;
                            if (tn == 0 && excess > 0 && (int)excess <= cnT) {
                                const int keepn = cnT - (int)excess;
                                int base = 0;
                                u32 kd[NCHR]; bool kill[NCHR];
                                u32 d_lastkept = 0xFFFFFFFFu, d_firstkill = 0xFFFFFFFEu;
#pragma unroll
                                for (int ch = 0; ch < NCHR; ch++) {
                                    const bool live = (ch * 64 + lane < rn) && fl.v[ch] == 0u;
                                    const u64 lm = __ballot(live);
                                    const int rank = base + __popcll(lm & lanemask_lt());
                                    kill[ch] = live && rank >= keepn;
                                    kd[ch] = (u32)(rk.v[ch] >> 32);
                                    const u64 m1 = __ballot(live && rank == keepn - 1), m2 = __ballot(live && rank == keepn);
                                    if (m1 != 0ull) d_lastkept = readlane32(kd[ch], __ffsll((long long)m1) - 1);
                                    if (m2 != 0ull) d_firstkill = readlane32(kd[ch], __ffsll((long long)m2) - 1);
                                    base += __popcll(lm);
                                }
                                if (keepn == 0 || d_lastkept != d_firstkill) {
#pragma unroll
                                    for (int ch = 0; ch < NCHR; ch++) fl.v[ch] |= kill[ch] ? 2u : 0u;
                                    cnT -= (int)excess;
                                    done_trim = true;
                                }
                            }
                            if (!done_trim)
                            for (u32 t = 0; t < excess; t++) {
                                const int ib = frontier_last<NCHR>(rk, fl, rn);
                                const u64 ka2 = (ib >= 0) ? fkey(list_get<NCHR>(rk, ib)) : 0ull;
                                const u64 kb2 = (tn > 0) ? readlane64(tl.v[0], tn - 1) : 0ull;
                                if (ib >= 0 && (tn == 0 || ka2 > kb2)) { flag_or<NCHR>(fl, ib, 2u); cnT--; }
                                else tn--;
                            }
                        }
                    } else if (kmode == 3u) {
                        if ((u32)(cnT + tn) + junk > p.bw) {
                            u32 excess = (u32)(cnT + tn) + junk - p.bw;
                            const u32 rl = excess < (u32)(cnT + tn) ? excess : (u32)(cnT + tn);
                            for (u32 t = 0; t < rl; t++) {
                                const int ib = frontier_first<NCHR>(rk, fl, rn);
                                const u64 ka2 = (ib >= 0) ? fkey(list_get<NCHR>(rk, ib)) : ~0ull;
                                const u64 kb2 = (tn > 0) ? readlane64(tl.v[0], 0) : ~0ull;
                                if (ka2 <= kb2) { flag_or<NCHR>(fl, ib, 2u); cnT--; }
                                else { list_pop_front<1>(tl); tn--; }
                            }
                            excess -= rl;
                            junk -= excess;
                        }
                    }
                    LM(4);
                }
                if (lane == 0) ctl[0] = finished ? 0u : ntasks;
                if (finished) LT(1);
            }
            __syncthreads();
            const u32 nt = (u32)__builtin_amdgcn_readfirstlane((int)ctl[0]);
            if (nt == 0u) break;
            // ---- scoring round
            if ((u32)wave < nt) {
                const u32 node = ctl[8 + wave];
                const u32 info = ctl[16 + wave];
                const int s = (int)(info & 255u), part = (int)((info >> 8) & 255u), nparts = (int)(info >> 16);
                if (node != LAT_NONE)
                    lat_score<D, FILTER>(p, node, slot_ids(s), slot_eb(s), slot_ab(s), slot_mask(s), slot_mask(s) + nwords, ctl[2], ctl[3] != 0u, part, nparts, vh, vmask, vshift,
                                         ctl[1] != 0u ? vgw : nullptr, gmask, gshift, lut, qreg, qperm, nb_id, nb_ln, knorm
#ifdef DR_LAT_SUBPHASES
                                         , wave == 0 ? lt_acc : nullptr
#endif
                                         );
            }
#ifdef DR_PHASE_TIMING
            if (wave == 0) { LT_NOW(lt_b); lt_acc[4] += lt_b - lt_a; }
#endif
            __syncthreads();
            LT(3);
        }

        if (wave == 0)
            write_results<NCHR>(p, qi, cap, kmode, true, true, rk, rn, steps, nvisited, nexact, npq, status, ninserts, npq_eval, nhits);
        LT(7);
        LT_END(qi);
        __syncthreads();        // the next query rewrites the table, the set and the slots
        if (ctl[1] != 0u) {
            // this query spilled: its workgroup's global table back to all-empty (sixteen bytes per store, the whole workgroup)
            uint4 *g4 = reinterpret_cast<uint4 *>(vgw);
            for (u32 i = threadIdx.x; i < gslots / 4; i += 64 * LAT_NW) g4[i] = make_uint4(LAT_NONE, LAT_NONE, LAT_NONE, LAT_NONE);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}
