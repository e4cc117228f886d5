// numerics.hpp -- float32 arithmetic in the reference's summation orders, for gfx950 wavefronts.
//
// The reference computes exact distances with np.sum(diff*diff) (search_engine.py:378-379) and PQ table rows
// with np.sum(diff*diff, axis=1) (pydiskann/pq/fast_pq.py:315-316). numpy sums a contiguous run with its
// pairwise routine: n < 8 sequential; n <= 128 eight interleaved accumulators r[j] += a[8t+j] combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); n > 128 split at n/2 rounded down to a multiple of 8, recursively.
// Returned neighbour ids depend on exact float comparisons, so the device code reproduces that order bit for
// bit: products are rounded before they are added (no FMA), chains are accumulated in t order, and the
// combine tree is fixed.
//
// Device layout ("chain-major tiles"): a 64-lane wave scores 8 stored vectors at a time, 8 lanes (an octet)
// per vector, lane j of the octet owning accumulator chain j. For that lane's elements 8t+j to arrive as
// 16-byte loads, each leaf of the pairwise tree is stored in HBM with groups of four consecutive steps
// transposed: position off + g*32 + j*4 + u holds original element off + 8*(4g+u) + j. One
// global_load_dwordx4 per lane then reads 128 contiguous bytes per vector and 1 KiB per wave instruction,
// fully coalesced, and the lane adds its four values in step order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define DEV __device__ __forceinline__

// IEEE round-to-nearest f32 operations, never contracted into FMA (the pragma covers this whole header's
// functions even if the build drops -ffp-contract=off). sqrt must be the correctly rounded one: HIP's
// __fsqrt_rn lowers to the approximate native sqrt, __builtin_sqrtf to the IEEE sequence
// (-fhip-fp32-correctly-rounded-divide-sqrt, on by default).
#pragma clang fp contract(off)
DEV float f_sub(float a, float b) { return a - b; }
DEV float f_mul(float a, float b) { return a * b; }
DEV float f_add(float a, float b) { return a + b; }
DEV float f_sqrt(float a) { return __builtin_sqrtf(a); }
DEV float sqd(float v, float q) { float d = f_sub(v, q); return f_mul(d, d); }

// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) across the 8 lanes of an octet; every lane ends with the sum
// (float addition is commutative, so both partners of each exchange compute identical bits).
// The exchanges are DPP operands of the adds (no LDS permute round trips): lane^1 and lane^2 as quad permutations;
// after those every lane of a quad holds its quad's sum, so the mirror of the 8-lane half row (lane -> 7 - lane)
// delivers the other quad's sum. All 64 lanes must be active (every caller is wave-uniform).
template <int CTRL> DEV unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
#define DPP_QUAD_XOR1 0xB1      /* quad_perm:[1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E      /* quad_perm:[2,3,0,1] */
#define DPP_HALF_MIRROR 0x141   /* row_half_mirror */
DEV float octet_combine(float r)
{
    r = f_add(r, __uint_as_float(dpp_u32<DPP_QUAD_XOR1>(__float_as_uint(r))));
    r = f_add(r, __uint_as_float(dpp_u32<DPP_QUAD_XOR2>(__float_as_uint(r))));
    r = f_add(r, __uint_as_float(dpp_u32<DPP_HALF_MIRROR>(__float_as_uint(r))));
    return r;
}
DEV int octet_combine_i32(int r)
{
    r += (int)dpp_u32<DPP_QUAD_XOR1>((unsigned)r);
    r += (int)dpp_u32<DPP_QUAD_XOR2>((unsigned)r);
    r += (int)dpp_u32<DPP_HALF_MIRROR>((unsigned)r);
    return r;
}

// ---- host/device description of the pairwise tree -------------------------------------------------------
// perm[e] = position of original element e in the chain-major layout (host side, used at ingest and for
// queries). D must be a multiple of 8.
static inline void pw_build_perm_rec(uint32_t off, uint32_t n, uint32_t *perm)
{
    if (n <= 128) {
        uint32_t S = n / 8, G = S / 4, rem = S % 4;
        for (uint32_t t = 0; t < S; t++)
            for (uint32_t j = 0; j < 8; j++) {
                uint32_t e = off + 8 * t + j, g = t / 4, u = t % 4;
                perm[e] = (g < G) ? off + g * 32 + j * 4 + u : off + G * 32 + j * rem + (t - 4 * G);
            }
    } else {
        uint32_t n2 = n / 2;
        n2 -= n2 % 8;
        pw_build_perm_rec(off, n2, perm);
        pw_build_perm_rec(off + n2, n - n2, perm);
    }
}

// ---- query accessors ----------------------------------------------------------------------------------
// The query lives in the same chain-major layout. Small D: the lane's D/8 elements sit in registers;
// large D: in LDS, read as ds_read_b128 (octets read the same addresses: broadcast).
template <int D> struct QueryRegs {
    float v[D / 8];   // lane j: its chain elements in load order (leaf by leaf, group by group)
};

template <int OFF, int N> struct LeafInfo {
    static constexpr int S = N / 8, G = S / 4, REM = S % 4;
};

// number of per-lane floats consumed before leaf OFF starts = OFF/8 (each leaf of n elements gives n/8 per lane)

// Loads this lane's D/8 query elements from a chain-major query in memory (global or LDS).
template <int OFF, int N, int D> DEV void load_query_regs(const float *qp, int j, QueryRegs<D> &q)
{
    if constexpr (N <= 128) {
        constexpr int S = N / 8, G = S / 4, REM = S % 4;
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int u = 0; u < 4; u++) q.v[OFF / 8 + g * 4 + u] = qp[OFF + g * 32 + j * 4 + u];
#pragma unroll
        for (int u = 0; u < REM; u++) q.v[OFF / 8 + G * 4 + u] = qp[OFF + G * 32 + j * REM + u];
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        load_query_regs<OFF, N2, D>(qp, j, q);
        load_query_regs<OFF + N2, N - N2, D>(qp, j, q);
    }
}

// The same registers from a query in its ORIGINAL element order: entry t of a leaf's chain j is element OFF + 8 t + j (what
// the chain-major layout stores at g*32 + j*4 + u for t = 4g + u). Sixteen 4-byte loads per lane at D = 128, the eight
// lanes of an octet contiguous: the pipelined submit path needs no permuted copy of the batch (permute_queries_kernel).
template <int OFF, int N, int D> DEV void load_query_regs_orig(const float *q, int j, QueryRegs<D> &qr)
{
    if constexpr (N <= 128) {
#pragma unroll
        for (int t = 0; t < N / 8; t++) qr.v[OFF / 8 + t] = q[OFF + 8 * t + j];
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        load_query_regs_orig<OFF, N2, D>(q, j, qr);
        load_query_regs_orig<OFF + N2, N - N2, D>(q, j, qr);
    }
}

// ---- streaming form: loads and arithmetic interleaved (any D % 8 == 0) ----------------------------------
// row: chain-major stored vector (global). qreg != nullptr -> registers, else qlds (chain-major, LDS).
template <int OFF, int N, int D, bool QREG>
DEV float pw_row_stream(const float *__restrict__ row, const QueryRegs<D> *qreg, const float *qlds, int j)
{
    if constexpr (N <= 128) {
        constexpr int S = N / 8, G = S / 4, REM = S % 4;
        float r = 0.0f;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float4 v = *reinterpret_cast<const float4 *>(row + OFF + g * 32 + j * 4);
            float4 qq;
            if constexpr (QREG) {
                qq.x = qreg->v[OFF / 8 + g * 4 + 0]; qq.y = qreg->v[OFF / 8 + g * 4 + 1];
                qq.z = qreg->v[OFF / 8 + g * 4 + 2]; qq.w = qreg->v[OFF / 8 + g * 4 + 3];
            } else {
                qq = *reinterpret_cast<const float4 *>(qlds + OFF + g * 32 + j * 4);
            }
            const float s0 = sqd(v.x, qq.x), s1 = sqd(v.y, qq.y), s2 = sqd(v.z, qq.z), s3 = sqd(v.w, qq.w);
            r = (g == 0) ? s0 : f_add(r, s0);   // numpy: r[j] = a[j], then r[j] += a[8t+j]
            r = f_add(r, s1);
            r = f_add(r, s2);
            r = f_add(r, s3);
        }
#pragma unroll
        for (int u = 0; u < REM; u++) {
            const float v = row[OFF + G * 32 + j * REM + u];
            float qq;
            if constexpr (QREG) qq = qreg->v[OFF / 8 + G * 4 + u];
            else qq = qlds[OFF + G * 32 + j * REM + u];
            const float s = sqd(v, qq);
            r = (G == 0 && u == 0) ? s : f_add(r, s);
        }
        return octet_combine(r);
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        const float a = pw_row_stream<OFF, N2, D, QREG>(row, qreg, qlds, j);
        const float b = pw_row_stream<OFF + N2, N - N2, D, QREG>(row, qreg, qlds, j);
        return f_add(a, b);
    }
}

// ---- split form for small D: all loads of several rows first, arithmetic after ---------------------------
// Valid when every leaf has S % 4 == 0 (D = 96, 128, 256, ...): a row is D/32 float4 per lane.
template <int D> struct RowRegs { float4 g[D / 32]; };

template <int OFF, int N, int D> DEV void row_load(const float *__restrict__ row, int j, RowRegs<D> &rr)
{
    if constexpr (N <= 128) {
        constexpr int G = (N / 8) / 4;
        static_assert((N / 8) % 4 == 0, "split form needs whole groups");
#pragma unroll
        for (int g = 0; g < G; g++) rr.g[OFF / 32 + g] = *reinterpret_cast<const float4 *>(row + OFF + g * 32 + j * 4);
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        row_load<OFF, N2, D>(row, j, rr);
        row_load<OFF + N2, N - N2, D>(row, j, rr);
    }
}

template <int OFF, int N, int D> DEV float row_reduce(const RowRegs<D> &rr, const QueryRegs<D> &q)
{
    if constexpr (N <= 128) {
        constexpr int G = (N / 8) / 4;
        float r = 0.0f;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float4 v = rr.g[OFF / 32 + g];
            const float s0 = sqd(v.x, q.v[OFF / 8 + g * 4 + 0]), s1 = sqd(v.y, q.v[OFF / 8 + g * 4 + 1]);
            const float s2 = sqd(v.z, q.v[OFF / 8 + g * 4 + 2]), s3 = sqd(v.w, q.v[OFF / 8 + g * 4 + 3]);
            r = (g == 0) ? s0 : f_add(r, s0);
            r = f_add(r, s1);
            r = f_add(r, s2);
            r = f_add(r, s3);
        }
        return octet_combine(r);
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        const float a = row_reduce<OFF, N2, D>(rr, q);
        const float b = row_reduce<OFF + N2, N - N2, D>(rr, q);
        return f_add(a, b);
    }
}

template <int D> constexpr bool split_form_ok()
{
    // every leaf of the pairwise tree must have a multiple of 32 elements (S % 4 == 0)
    if (D <= 128) return (D % 32) == 0;
    return false;
}
template <> constexpr bool split_form_ok<256>() { return true; }

// ---- chunked form for large D: a software pipeline over sub-trees ("chunks") of the pairwise tree ----------
// A 6-KiB row (D = 1536) scored by pw_row_stream leaves the order of loads and arithmetic to the compiler, which keeps
// 4-13 KiB per wavefront in flight and drains to zero at every pass boundary: with 4 wavefronts per CU (the per-query
// table fills the LDS) that is ~0.5 of the HBM rate. Here the row is cut at a level of the SAME tree (a chunk = a
// whole sub-tree, so the arithmetic and its order are unchanged), a chunk's loads go to their own registers, and the
// caller keeps NBUF - 1 chunks in flight ahead of the one being reduced, across row passes (search_kernel.hpp).
template <int D> struct ChunkCfg {
    static constexpr int leaf() { int n = D; while (n > 128) n = n / 2 - ((n / 2) % 8); return n; }
    static constexpr int LS = leaf();                       // leaf size (every leaf equal for the dimensions below)
    static constexpr int S = LS / 8, G = S / 4, REM = S % 4;
#ifndef DR_CHUNK_LPC
#define DR_CHUNK_LPC 2
#endif
#ifndef DR_CHUNK_NBUF
#define DR_CHUNK_NBUF 4
#endif
    static constexpr int LPC = DR_CHUNK_LPC;                // leaves per chunk
    static constexpr int CH = LS * LPC;                     // elements per chunk (D = 1536, 768: 192; 960: 240)
    static constexpr int NC = D / CH;                       // chunks per row
    static constexpr int NBUF = DR_CHUNK_NBUF;              // ring depth (NC % NBUF == 0: buffer index is static)
    static constexpr bool ok = (D > 256) && (D % CH == 0) && (NC % NBUF == 0) && ((NC & (NC - 1)) == 0);
};
template <int D> struct ChunkRegs {
    float4 g[ChunkCfg<D>::LPC * ChunkCfg<D>::G];
    float r[ChunkCfg<D>::LPC * ChunkCfg<D>::REM + 1];
};

template <int COFF, int OFF, int N, int D> DEV void chunk_load(const float *__restrict__ row, int j, ChunkRegs<D> &cr)
{
    if constexpr (N <= 128) {
        static_assert(N == ChunkCfg<D>::LS, "equal leaves");
        constexpr int G = ChunkCfg<D>::G, REM = ChunkCfg<D>::REM, LI = (OFF - COFF) / N;
#pragma unroll
        for (int g = 0; g < G; g++) cr.g[LI * G + g] = *reinterpret_cast<const float4 *>(row + OFF + g * 32 + j * 4);
#pragma unroll
        for (int u = 0; u < REM; u++) cr.r[LI * REM + u] = row[OFF + G * 32 + j * REM + u];
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        chunk_load<COFF, OFF, N2, D>(row, j, cr);
        chunk_load<COFF, OFF + N2, N - N2, D>(row, j, cr);
    }
}
// the sub-tree's sum, bit for bit what pw_row_stream<OFF, N> returns (query chain-major in LDS)
template <int COFF, int OFF, int N, int D> DEV float chunk_reduce(const ChunkRegs<D> &cr, const float *qlds, int j)
{
    if constexpr (N <= 128) {
        constexpr int G = ChunkCfg<D>::G, REM = ChunkCfg<D>::REM, LI = (OFF - COFF) / N;
        float r = 0.0f;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float4 v = cr.g[LI * G + g];
            const float4 qq = *reinterpret_cast<const float4 *>(qlds + OFF + g * 32 + j * 4);
            const float s0 = sqd(v.x, qq.x), s1 = sqd(v.y, qq.y), s2 = sqd(v.z, qq.z), s3 = sqd(v.w, qq.w);
            r = (g == 0) ? s0 : f_add(r, s0);
            r = f_add(r, s1);
            r = f_add(r, s2);
            r = f_add(r, s3);
        }
#pragma unroll
        for (int u = 0; u < REM; u++) {
            const float s = sqd(cr.r[LI * REM + u], qlds[OFF + G * 32 + j * REM + u]);
            r = (G == 0 && u == 0) ? s : f_add(r, s);
        }
        return octet_combine(r);
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        const float a = chunk_reduce<COFF, OFF, N2, D>(cr, qlds, j);
        const float b = chunk_reduce<COFF, OFF + N2, N - N2, D>(cr, qlds, j);
        return f_add(a, b);
    }
}
// the levels of the tree above the chunks: c[0..NC) are the chunk sums in row order
template <int LO, int N> DEV float chunk_tree(const float *c)
{
    if constexpr (N == 1) return c[LO];
    else return f_add(chunk_tree<LO, N / 2>(c), chunk_tree<LO + N / 2, N / 2>(c));
}

// One row pass of the pipeline: chunk C of the row `rp` is reduced while chunks C+1 .. C+NBUF-1 are in flight -- the
// last of them issued here, from this row or (HAS_NEXT) from the next pass's row `rnext`. HAS_NEXT is a template
// parameter, not a branch: behind a branch the compiler's wait counters must assume the loads were not issued and the
// pipeline drains at every pass boundary.
template <int D, int C, bool HAS_NEXT>
DEV void chunk_pass(const float *__restrict__ rp, const float *__restrict__ rnext, int j,
                    ChunkRegs<D> (&buf)[ChunkCfg<D>::NBUF], const float *qlds, float (&cres)[ChunkCfg<D>::NC])
{
    constexpr int CH = ChunkCfg<D>::CH, NC = ChunkCfg<D>::NC, NBUF = ChunkCfg<D>::NBUF;
    if constexpr (C < NC) {
        constexpr int CN = C + NBUF - 1;
        if constexpr (CN < NC) chunk_load<CN * CH, CN * CH, CH, D>(rp, j, buf[CN % NBUF]);
        else if constexpr (HAS_NEXT) chunk_load<(CN - NC) * CH, (CN - NC) * CH, CH, D>(rnext, j, buf[CN % NBUF]);
        // (scheduling barriers: a pass is one basic block, and without them the machine scheduler sinks the loads next
        // to their uses -- fewer live registers, no pipeline)
        __builtin_amdgcn_sched_barrier(0);
        cres[C] = chunk_reduce<C * CH, C * CH, CH, D>(buf[C % NBUF], qlds, j);
        __builtin_amdgcn_sched_barrier(0);
        chunk_pass<D, C + 1, HAS_NEXT>(rp, rnext, j, buf, qlds, cres);
    }
}
// the pipeline's prologue: chunks 0 .. NBUF-2 of the first row
template <int D, int C> DEV void chunk_prologue(const float *__restrict__ rp, int j, ChunkRegs<D> (&buf)[ChunkCfg<D>::NBUF])
{
    constexpr int CH = ChunkCfg<D>::CH, NBUF = ChunkCfg<D>::NBUF;
    if constexpr (C < NBUF - 1) {
        chunk_load<C * CH, C * CH, CH, D>(rp, j, buf[C]);
        __builtin_amdgcn_sched_barrier(0);
        chunk_prologue<D, C + 1>(rp, j, buf);
    }
}

// Same run with the centroid already in registers and a compile-time length (numpy order for n <= 128).
template <int N>
DEV float pw_run_regs(const float (&c)[N], const float *q)
{
    static_assert(N >= 1 && N <= 128, "sub_dim");
    if constexpr (N < 8) {
        float res = 0.0f;
#pragma unroll
        for (int i = 0; i < N; i++) res = f_add(res, sqd(c[i], q[i]));
        return res;
    } else {
        // the eight accumulator chains as four PAIRS (round 6): difference, square and add of two neighbouring chains are one packed instruction each
        // (v_pk_add_f32 with a negated operand, v_pk_mul_f32, v_pk_add_f32) -- the same IEEE operations on the same values in the same order per
        // chain, 72 instead of 96 instructions per table entry at sub_dim 48 (the compiler packed the squares and adds but left 48 scalar
        // subtractions: the query pair is a scalar-register operand)
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 r[4];
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const f2 cc = { c[2 * jj], c[2 * jj + 1] }, qq = { q[2 * jj], q[2 * jj + 1] };
            const f2 d = cc - qq;
            r[jj] = d * d;
        }
        constexpr int lim = N - (N % 8);
#pragma unroll
        for (int i = 8; i < lim; i += 8) {
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const f2 cc = { c[i + 2 * jj], c[i + 2 * jj + 1] }, qq = { q[i + 2 * jj], q[i + 2 * jj + 1] };
                const f2 d = cc - qq;
                const f2 sq = d * d;
                r[jj] = r[jj] + sq;
            }
        }
        float res = f_add(f_add(f_add(r[0].x, r[0].y), f_add(r[1].x, r[1].y)), f_add(f_add(r[2].x, r[2].y), f_add(r[3].x, r[3].y)));
#pragma unroll
        for (int i = lim; i < N; i++) res = f_add(res, sqd(c[i], q[i]));
        return res;
    }
}

// ---- one lane, one short contiguous run (PQ table rows, n = sub_dim <= 128), original element order -----
// A2: DiskANNPQ.compute_distance_table, fast_pq.py:294-318.
DEV float pw_run_lane(const float *__restrict__ c, const float *q, int n)
{
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; i++) res = f_add(res, sqd(c[i], q[i]));
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = sqd(c[j], q[j]);
    int i = 8;
    const int lim = n - (n % 8);
    for (; i < lim; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = f_add(r[j], sqd(c[i + j], q[i + j]));
    }
    float res = f_add(f_add(f_add(r[0], r[1]), f_add(r[2], r[3])), f_add(f_add(r[4], r[5]), f_add(r[6], r[7])));
    for (; i < n; i++) res = f_add(res, sqd(c[i], q[i]));
    return res;
}

// ---- float64 queries (the CLI path, diskrag.py:194: np.array(list) is float64) ------------------------------
// vec (f32) - q (f64) promotes to f64 and np.sum runs the same pairwise routine on doubles
// (search_engine.py:378-379 with a float64 query). Same chain-major layout, same tree, double arithmetic.
DEV double d_sub(double a, double b) { return a - b; }
DEV double d_mul(double a, double b) { return a * b; }
DEV double d_add(double a, double b) { return a + b; }
DEV double sqd64(float v, double q) { double d = d_sub((double)v, q); return d_mul(d, d); }

DEV double octet_combine64(double r)
{
    r = d_add(r, __shfl_xor(r, 1));
    r = d_add(r, __shfl_xor(r, 2));
    r = d_add(r, __shfl_xor(r, 4));
    return r;
}

// row: chain-major stored vector (global); qlds: chain-major float64 query (LDS). Lane j of the octet owns chain j.
template <int OFF, int N, int D>
DEV double pw_row_stream64(const float *__restrict__ row, const double *qlds, int j)
{
    if constexpr (N <= 128) {
        constexpr int S = N / 8, G = S / 4, REM = S % 4;
        double r = 0.0;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const float4 v = *reinterpret_cast<const float4 *>(row + OFF + g * 32 + j * 4);
            const double *qq = qlds + OFF + g * 32 + j * 4;
            const double s0 = sqd64(v.x, qq[0]), s1 = sqd64(v.y, qq[1]), s2 = sqd64(v.z, qq[2]), s3 = sqd64(v.w, qq[3]);
            r = (g == 0) ? s0 : d_add(r, s0);
            r = d_add(r, s1);
            r = d_add(r, s2);
            r = d_add(r, s3);
        }
#pragma unroll
        for (int u = 0; u < REM; u++) {
            const double s = sqd64(row[OFF + G * 32 + j * REM + u], qlds[OFF + G * 32 + j * REM + u]);
            r = (G == 0 && u == 0) ? s : d_add(r, s);
        }
        return octet_combine64(r);
    } else {
        constexpr int N2 = (N / 2) - ((N / 2) % 8);
        const double a = pw_row_stream64<OFF, N2, D>(row, qlds, j);
        const double b = pw_row_stream64<OFF + N2, N - N2, D>(row, qlds, j);
        return d_add(a, b);
    }
}

// One lane, one short contiguous run in original element order, float64 query (A2 with a float64 query:
// the table row is summed in f64 and stored as f32, fast_pq.py:307, 315-316).
DEV double pw_run_lane64(const float *__restrict__ c, const double *q, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res = d_add(res, sqd64(c[i], q[i]));
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = sqd64(c[j], q[j]);
    int i = 8;
    const int lim = n - (n % 8);
    for (; i < lim; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = d_add(r[j], sqd64(c[i + j], q[i + j]));
    }
    double res = d_add(d_add(d_add(r[0], r[1]), d_add(r[2], r[3])), d_add(d_add(r[4], r[5]), d_add(r[6], r[7])));
    for (; i < n; i++) res = d_add(res, sqd64(c[i], q[i]));
    return res;
}

// ---- any length, the tree evaluated from n AT RUN TIME (round 6: dimensions that have no compiled tree) ------------------------------
// numpy's pairwise routine literally (numpy/_core/src/umath/loops_utils.h.src @TYPE@_pairwise_sum): n < 8 sequential from 0; n <= 128 eight accumulators r[j] += a[8 t + j], combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the
// n % 8 leftovers added one by one; above 128 split at n / 2 rounded down to a multiple of 8 -- the recursion kept on an explicit stack
// (n <= 32768: nine frames; twelve are there). ONE LANE sums ONE run (the generic traversal scores 64 neighbours at a time, a lane each). `pos` (may be null):
// element i of the run is stored at c[pos[i]] -- the chain-major rows of a built dimension, so that this routine and the compiled trees can
// be held to each other on the same index (tests/test_gpu_shapes.py). The same bits as pw_row_stream<0, D, D> / pw_run_lane wherever those exist.
template <typename REAL> DEV REAL rt_sqd(float v, REAL q);
template <> DEV float rt_sqd<float>(float v, float q) { return sqd(v, q); }
template <> DEV double rt_sqd<double>(float v, double q) { return sqd64(v, q); }
DEV float rt_add(float a, float b) { return f_add(a, b); }
DEV double rt_add(double a, double b) { return d_add(a, b); }

template <typename REAL>
DEV REAL pw_leaf_rt(const float *__restrict__ c, const u32 *pos, const REAL *q, int off, int n)
{
#define DR_RT_EL(i) rt_sqd<REAL>(c[pos ? pos[off + (i)] : (u32)(off + (i))], q[off + (i)])
    if (n < 8) {
        REAL res = 0;
        for (int i = 0; i < n; i++) res = rt_add(res, DR_RT_EL(i));
        return res;
    }
    REAL r[8];
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = DR_RT_EL(j);
    int i = 8;
    const int lim = n - (n % 8);
    for (; i < lim; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = rt_add(r[j], DR_RT_EL(i + j));
    }
    REAL res = rt_add(rt_add(rt_add(r[0], r[1]), rt_add(r[2], r[3])), rt_add(rt_add(r[4], r[5]), rt_add(r[6], r[7])));
    for (; i < n; i++) res = rt_add(res, DR_RT_EL(i));
    return res;
#undef DR_RT_EL
}

template <typename REAL>
DEV REAL pw_run_rt(const float *__restrict__ c, const u32 *pos, const REAL *q, int n)
{
    if (n <= 128) return pw_leaf_rt<REAL>(c, pos, q, 0, n);
    // frames of the recursion: (off, len, stage 0 = enter / 1 = left half running / 2 = right half running, left half's sum)
    int f_off[12], f_len[12], f_stage[12];
    REAL f_left[12];
    int sp = 0;
    f_off[0] = 0; f_len[0] = n; f_stage[0] = 0;
    REAL ret = 0;
    for (;;) {
        if (f_len[sp] <= 128) {
            ret = pw_leaf_rt<REAL>(c, pos, q, f_off[sp], f_len[sp]);
            // return to the callers: a left half hands over to its right half, a right half adds and returns further up
            for (;;) {
                if (sp == 0) return ret;
                sp--;
                int n2 = f_len[sp] / 2; n2 -= n2 % 8;
                if (f_stage[sp] == 1) {
                    f_left[sp] = ret; f_stage[sp] = 2;
                    f_off[sp + 1] = f_off[sp] + n2; f_len[sp + 1] = f_len[sp] - n2; f_stage[sp + 1] = 0;
                    sp++;
                    break;
                }
                ret = rt_add(f_left[sp], ret);
            }
        } else {
            if (sp >= 10) return ret;       // (frames halve: n <= 32768 needs 9; the host refuses longer runs)
            int n2 = f_len[sp] / 2; n2 -= n2 % 8;
            f_stage[sp] = 1;
            f_off[sp + 1] = f_off[sp]; f_len[sp + 1] = n2; f_stage[sp + 1] = 0;
            sp++;
        }
    }
}
