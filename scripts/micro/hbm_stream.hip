// What a read-only streaming kernel reaches on this device (GPU box): 2 GiB read once per launch with 16-byte loads, UNROLL loads in flight per
// lane, xor-reduced; the best of a few grid shapes. The reference point for the flat PQ scans' roofline fractions.
// build + run: hipcc --offload-arch=gfx950 -O3 scripts/micro/hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT>
__global__ __launch_bounds__(512) void stream_kernel(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *__restrict__ out)
{
    u32x4 acc = { 0, 0, 0, 0 };
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= src[i];
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[0] = r;      // (keeps the loads alive)
}

template <int UNROLL, bool NT>
static double run(const u32x4 *src, uint64_t n16, uint32_t *out, int blocks, int threads)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    double best = 1e30;
    for (int rep = 0; rep < 8; rep++) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((stream_kernel<UNROLL, NT>), dim3(blocks), dim3(threads), 0, 0, src, n16, out);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        if (rep > 1 && ms < best) best = ms;
    }
    return best;
}

int main()
{
    const uint64_t bytes = 2ull << 30, n16 = bytes / 16;
    u32x4 *src; uint32_t *out;
    if (hipMalloc((void **)&src, bytes) != hipSuccess || hipMalloc((void **)&out, 4) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
    hipMemset(src, 0x5a, bytes);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    printf("{\"bytes\": %llu, \"cus\": %d, \"runs\": [", (unsigned long long)bytes, cus);
    bool first = true;
    double top = 0;
    for (int threads : { 256, 512 })
        for (int per_cu : { 1, 2, 4, 8 }) {
            if (threads * per_cu > 2048) continue;
            const int blocks = cus * per_cu;
            struct { const char *name; double ms; } r[] = {
                { "unroll1", run<1, false>(src, n16, out, blocks, threads) }, { "unroll2", run<2, false>(src, n16, out, blocks, threads) },
                { "unroll4", run<4, false>(src, n16, out, blocks, threads) }, { "unroll8", run<8, false>(src, n16, out, blocks, threads) },
                { "unroll4_nt", run<4, true>(src, n16, out, blocks, threads) }, { "unroll8_nt", run<8, true>(src, n16, out, blocks, threads) } };
            for (auto &x : r) {
                const double tbps = bytes / (x.ms * 1e-3) / 1e12;
                if (tbps > top) top = tbps;
                printf("%s{\"threads\": %d, \"blocks_per_cu\": %d, \"kind\": \"%s\", \"ms\": %.4f, \"TBps\": %.3f}", first ? "" : ", ", threads, per_cu, x.name, x.ms, tbps);
                first = false;
            }
        }
    printf("], \"best_TBps\": %.3f}\n", top);
    return 0;
}
