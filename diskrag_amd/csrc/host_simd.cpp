// host_simd.cpp -- host-side helpers of libdiskrag_hip.so that want the CPU's vector units (plain C++, no HIP).
//
// dr_host_all_u8: is every component of a query batch an integer in [0, 255]? Asked once per submitted batch (the byte-query
// kernel variants 13 / 14 / 17 need it; a single fractional, negative, too-large or NaN component sends the batch to the float-query
// variants). The portable loop costs 61 us per 1250 x 128 queries (0.5 ms per 10k-query batch) on the submitting thread -- a third
// of what a coalesced 1250-query submit costs the host; with AVX2 it is 20 us. -0.0 counts as 0 (it converts to the same byte).
#include <cstddef>
#include <cstdint>
#include <algorithm>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

static bool all_u8_portable(const float *q, size_t n)
{
    bool ok = true;
    for (size_t b = 0; b < n && ok; b += 4096) {
        const size_t e = std::min(n, b + 4096);
        unsigned bad = 0;     // branch-free so that it vectorises (out-of-range and NaN are clamped before the conversion)
        for (size_t i = b; i < e; i++) {
            const float v = q[i];
            const float c = (v >= 0.0f && v <= 255.0f) ? v : -1.0f;
            const int iv = (int)c;
            bad |= (unsigned)(iv < 0) | (unsigned)((float)iv != v);
        }
        ok = !bad;
    }
    return ok;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static bool all_u8_avx2(const float *q, size_t n)
{
    const __m256 lo = _mm256_set1_ps(0.0f), hi = _mm256_set1_ps(255.0f);
    const __m256 ones = _mm256_castsi256_ps(_mm256_set1_epi32(-1));
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        __m256 bad = _mm256_setzero_ps();
        for (int u = 0; u < 4; u++) {
            const __m256 v = _mm256_loadu_ps(q + i + 8 * u);
            const __m256 r = _mm256_round_ps(v, _MM_FROUND_TO_ZERO | _MM_FROUND_NO_EXC);
            // in range and integral; NaN fails every ordered compare
            const __m256 ok = _mm256_and_ps(_mm256_and_ps(_mm256_cmp_ps(v, lo, _CMP_GE_OQ), _mm256_cmp_ps(v, hi, _CMP_LE_OQ)), _mm256_cmp_ps(r, v, _CMP_EQ_OQ));
            bad = _mm256_or_ps(bad, _mm256_andnot_ps(ok, ones));
        }
        if (_mm256_movemask_ps(bad)) return false;
    }
    return all_u8_portable(q + i, n - i);
}
#endif

extern "C" bool dr_host_all_u8(const float *q, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return all_u8_avx2(q, n);
#endif
    return all_u8_portable(q, n);
}

// dr_host_stage_u8: the staging copy of a pageable query batch into page-locked memory AND the byte check, in one pass over the batch
// (a blocking dr_search_batch used to pay for both: the runtime's own staged copy, then 0.16 ms of scanning per 10 000 x 128 queries).
#if defined(__x86_64__)
__attribute__((target("avx2"))) static bool stage_u8_avx2(float *dst, const float *q, size_t n)
{
    const __m256 lo = _mm256_set1_ps(0.0f), hi = _mm256_set1_ps(255.0f);
    const __m256 ones = _mm256_castsi256_ps(_mm256_set1_epi32(-1));
    __m256 bad = _mm256_setzero_ps();
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(q + i);
        _mm256_storeu_ps(dst + i, v);
        const __m256 r = _mm256_round_ps(v, _MM_FROUND_TO_ZERO | _MM_FROUND_NO_EXC);
        const __m256 ok = _mm256_and_ps(_mm256_and_ps(_mm256_cmp_ps(v, lo, _CMP_GE_OQ), _mm256_cmp_ps(v, hi, _CMP_LE_OQ)), _mm256_cmp_ps(r, v, _CMP_EQ_OQ));
        bad = _mm256_or_ps(bad, _mm256_andnot_ps(ok, ones));
    }
    bool okall = _mm256_movemask_ps(bad) == 0;
    for (; i < n; i++) dst[i] = q[i];
    return okall && all_u8_portable(q + (n & ~(size_t)7), n & 7);
}
#endif
extern "C" bool dr_host_stage_u8(float *dst, const float *q, size_t n, bool want_check)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && want_check) return stage_u8_avx2(dst, q, n);
#endif
    for (size_t i = 0; i < n; i++) dst[i] = q[i];
    return want_check ? all_u8_portable(q, n) : false;
}
