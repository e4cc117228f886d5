// gather_probe.hip -- measurement tool (not product): what rate of random fixed-size row reads can one MI355X sustain?
// The search kernel's expansion is a burst of random 128-byte (byte rows) / 512-byte (float rows) reads plus ~60
// random 4-byte bitmap accesses; this probe measures the chip's ceiling for exactly that access shape so that the
// kernel's achieved request rate can be priced against it.
//   usage: gather_probe <table_MiB> <row_bytes> <waves_per_cu> <rows_in_flight_per_wave> [mode]
//   mode 0: rows landed in LDS by global_load_lds (16 B per lane), mode 1: scattered 4-byte loads (one per lane),
//   mode 2: scattered 4-byte load + store of the same word (read-modify-write without atomics), mode 3: atomicOr
//   mode 4: scattered 32-byte rows, one per LANE (two 16-byte loads to registers: the code-word gather of the ADC traversals, m = 32)
// Every run prints "REQS <requests per launch> BYTES <useful bytes per launch>", so that a rocprofv3 --pmc pass over the
// probe calibrates FETCH_SIZE / TCC_EA0_RDREQ[_32B] for the access shape (profiles/r04/tcc_calibration.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const unsigned char *table, uint32_t *wtable, uint64_t nrows, uint32_t row_bytes,
                                              uint32_t iters, uint32_t inflight, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + wave;
    unsigned char *my = lds + (size_t)wave * inflight * 1024;
    uint32_t acc = 0, seed = gw * 0x9E3779B1u + 12345u;
    const uint32_t lanes_per_row = row_bytes / 16;           // MODE 0: 16 B per lane
    const uint32_t rows_per_instr = 64 / lanes_per_row;
    for (uint32_t it = 0; it < iters; it++) {
        if constexpr (MODE == 0) {
            for (uint32_t f = 0; f < inflight; f++) {
                seed = mix(seed + f + 1);
                const uint32_t r = mix(seed + lane / lanes_per_row);
                const uint64_t row = (uint64_t)r % nrows;
                const unsigned char *g = table + row * row_bytes + (lane % lanes_per_row) * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                    (__attribute__((address_space(3))) void *)(my + f * 1024), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += *reinterpret_cast<const uint32_t *>(my + lane * 4);
        } else if constexpr (MODE == 4) {
            uint4 v[16];
            for (uint32_t f = 0; f < inflight && f < 8; f++) {
                seed = mix(seed + f + 1);
                const uint64_t row = (uint64_t)mix(seed ^ (lane * 0x85ebca6bU)) % (nrows * (row_bytes / 32));
                const uint4 *g = reinterpret_cast<const uint4 *>(table + row * 32);
                v[2 * f] = g[0]; v[2 * f + 1] = g[1];
            }
            for (uint32_t f = 0; f < inflight && f < 8; f++) acc += v[2 * f].x + v[2 * f + 1].w;
        } else {
            uint32_t v[8];
            for (uint32_t f = 0; f < inflight && f < 8; f++) {
                seed = mix(seed + f + 1);
                const uint64_t w = (uint64_t)mix(seed ^ (lane * 0x85ebca6bU)) % (nrows * (row_bytes / 4));
                if constexpr (MODE == 1) v[f] = __hip_atomic_load(&wtable[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if constexpr (MODE == 2) { v[f] = __hip_atomic_load(&wtable[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else v[f] = atomicOr(&wtable[w], 0u);
                if constexpr (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wtable[w] = v[f]; }
            }
            for (uint32_t f = 0; f < inflight && f < 8; f++) acc += v[f];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
    (void)rows_per_instr;
}

int main(int argc, char **argv)
{
    const uint64_t mib = argc > 1 ? strtoull(argv[1], 0, 10) : 128;
    const uint32_t row_bytes = argc > 2 ? atoi(argv[2]) : 128;
    const uint32_t wpc = argc > 3 ? atoi(argv[3]) : 16;
    const uint32_t inflight = argc > 4 ? atoi(argv[4]) : 8;
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t cus = prop.multiProcessorCount;
    const uint64_t bytes = mib << 20, nrows = bytes / row_bytes;
    unsigned char *table; uint32_t *sink;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(table, 0, bytes));
    const uint32_t iters = 2000;
    const size_t lds = (size_t)wpc * inflight * 1024;
    auto k = mode == 0 ? probe<0> : mode == 1 ? probe<1> : mode == 2 ? probe<2> : mode == 4 ? probe<4> : probe<3>;
    CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(cus), dim3(64 * wpc), lds, 0, table, (uint32_t *)table, nrows, row_bytes, iters, inflight, sink);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        const double instr = (double)cus * wpc * iters * inflight;
        const double reqs = mode == 0 ? instr * (64.0 / (row_bytes / 16)) : instr * 64.0;
        const double gb = mode == 0 ? instr * 1024.0 / 1e9 : reqs * (mode == 4 ? 32 : 4) / 1e9;
        if (rep == 2) printf("REQS %.0f BYTES %.0f\n", reqs, gb * 1e9);
        if (rep == 2) printf("table %llu MiB row %u B waves/CU %u inflight %u mode %d: %.3f ms  %.2f G req/s  %.2f TB/s\n",
                             (unsigned long long)mib, row_bytes, wpc, inflight, mode, ms, reqs / ms / 1e6, gb / ms);
    }
    return 0;
}
