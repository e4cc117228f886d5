// Can a gather of items SMALLER than a 128-byte line cost less than a line each? (GPU box; round 6.) Round 4's calibration (profiles/r04/tcc_calibration.json)
// found that every L2 miss of a cached load is a 128-byte fabric request: a 16- or 32-byte code word costs a whole line, and the ADC traversals of c4 / c5
// move 4-6x their own bytes. This probe gathers random ITEM-byte items (16 or 32) from a table far larger than the caches with other load kinds --
// plain, non-temporal, sc0 sc1 (system scope) -- and from memory allocated UNCACHED (hipDeviceMallocUncached), and prints items per second for each.
// build + run: hipcc --offload-arch=gfx950 -O3 scripts/micro/gather_small_items.hip -o /tmp/gather_small && /tmp/gather_small
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// eight items in flight per lane; KIND 0 plain, 1 non-temporal, 2 sc0 sc1, 3 sc1 (the last two as ONE asm block that ends in its own wait: the
// compiler does not know that an asm load is still in flight and would reuse its registers)
#define LD8(MOD) asm volatile( \
    "global_load_dwordx4 %0, %8, off " MOD "\n\tglobal_load_dwordx4 %1, %9, off " MOD "\n\tglobal_load_dwordx4 %2, %10, off " MOD "\n\tglobal_load_dwordx4 %3, %11, off " MOD "\n\t" \
    "global_load_dwordx4 %4, %12, off " MOD "\n\tglobal_load_dwordx4 %5, %13, off " MOD "\n\tglobal_load_dwordx4 %6, %14, off " MOD "\n\tglobal_load_dwordx4 %7, %15, off " MOD "\n\t" \
    "s_waitcnt vmcnt(0)" \
    : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]) \
    : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]) : "memory")

template <int KIND, int ITEM16>      // ITEM16: 16-byte pieces per item (1 or 2)
__global__ __launch_bounds__(1024) void gather(const unsigned char *table, uint64_t nitems, uint32_t iters, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63, gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t acc = 0, seed = gw * 0x9E3779B1u + 12345u;
    constexpr int NI = 8 / ITEM16;      // items per trip: eight 16-byte loads in flight per lane in every kind
    for (uint32_t it = 0; it < iters * ITEM16; it++) {
        const u32x4 *g[8];
        u32x4 v[8];
#pragma unroll
        for (int f = 0; f < NI; f++) {
            seed = mix(seed + f + 1);
            const uint64_t item = (uint64_t)mix(seed ^ (lane * 0x85ebca6bU)) % nitems;
#pragma unroll
            for (int w = 0; w < ITEM16; w++) g[f * ITEM16 + w] = reinterpret_cast<const u32x4 *>(table + item * (16 * ITEM16)) + w;
        }
        if constexpr (KIND == 0) {
#pragma unroll
            for (int f = 0; f < 8; f++) v[f] = *g[f];
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int f = 0; f < 8; f++) v[f] = __builtin_nontemporal_load(g[f]);
        } else if constexpr (KIND == 2) LD8("sc0 sc1");
        else LD8("sc1");
#pragma unroll
        for (int f = 0; f < 8; f++) acc += v[f].x + v[f].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int KIND, int ITEM16> static double run(const unsigned char *t, uint64_t bytes, uint32_t *sink, int cus)
{
    const uint32_t iters = 64;
    const uint64_t nitems = bytes / (16 * ITEM16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double best = 1e30;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((gather<KIND, ITEM16>), dim3(cus), dim3(1024), 0, 0, t, nitems, iters, sink);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms < best) best = ms;
    }
    { hipError_t e_ = hipDeviceSynchronize(); fprintf(stderr, "kind %d item %d: %.3f ms %s\n", KIND, 16 * ITEM16, best, hipGetErrorString(e_)); }
    return (double)cus * 1024 * iters * 8 / (best * 1e-3) / 1e9;      // G items / s
}

int main()
{
    const uint64_t bytes = 8ull << 30;
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    unsigned char *cached = nullptr, *unc = nullptr; uint32_t *sink;
    CHECK(hipMalloc((void **)&cached, bytes)); CHECK(hipMalloc((void **)&sink, 4));
    CHECK(hipMemset(cached, 0x11, bytes));
    const bool have_unc = hipExtMallocWithFlags((void **)&unc, bytes, hipDeviceMallocUncached) == hipSuccess;
    if (have_unc) CHECK(hipMemset(unc, 0x11, bytes)); else (void)hipGetLastError();
    fprintf(stderr, "allocated (uncached %d)\n", (int)have_unc);
    printf("{\"table_bytes\": %llu, \"cus\": %d, \"uncached_allocation\": %s, \"G_items_per_s\": {", (unsigned long long)bytes, cus, have_unc ? "true" : "false");
    printf("\"16B_plain\": %.2f, \"16B_nt\": %.2f, \"16B_sc0sc1\": %.2f, \"16B_sc1\": %.2f, ", run<0, 1>(cached, bytes, sink, cus), run<1, 1>(cached, bytes, sink, cus), run<2, 1>(cached, bytes, sink, cus), run<3, 1>(cached, bytes, sink, cus));
    printf("\"32B_plain\": %.2f, \"32B_nt\": %.2f, \"32B_sc0sc1\": %.2f, \"32B_sc1\": %.2f", run<0, 2>(cached, bytes, sink, cus), run<1, 2>(cached, bytes, sink, cus), run<2, 2>(cached, bytes, sink, cus), run<3, 2>(cached, bytes, sink, cus));
    if (have_unc) printf(", \"16B_plain_uncached_memory\": %.2f, \"16B_nt_uncached_memory\": %.2f, \"32B_plain_uncached_memory\": %.2f, \"32B_nt_uncached_memory\": %.2f",
                         run<0, 1>(unc, bytes, sink, cus), run<1, 1>(unc, bytes, sink, cus), run<0, 2>(unc, bytes, sink, cus), run<1, 2>(unc, bytes, sink, cus));
    printf("}}\n");
    return 0;
}
