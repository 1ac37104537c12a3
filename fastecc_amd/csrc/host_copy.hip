// host_copy.hip — the host side of the staging rings (host_stage.hip stage_transfer): rows of a column slab between the caller's pageable stripe
// (one piece of `width` bytes every `pitch` bytes: 512 bytes of every 4 KB block at 8 slabs) and the packed rows of a pinned slot.
// Host code only.  Plain memcpy moved 37-45 GB/s per direction with six threads on an EPYC 9575F: a piece per page defeats the hardware
// prefetchers (they stop at 4 KB boundaries), and ordinary stores read every destination line before overwriting it.  Here the source
// pieces of the rows ahead are prefetched in software and the destination is written with non-temporal stores (whole 64-byte lines, no
// read for ownership).
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#include <immintrin.h>
#define FASTECC_HOST_COPY_AVX2 1  // x86-64 hosts only; every other host takes the memcpy rows below
#endif

#include "internal.hpp"

namespace fastecc {

#ifndef __HIP_DEVICE_COMPILE__
#ifdef FASTECC_HOST_COPY_AVX2
namespace {

__attribute__((target("avx2"))) void copy_rows_avx2(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t width, size_t rows)
{
    constexpr size_t AHEAD = 8;  // rows: 8 x (width / 64) lines in flight per thread
    for (size_t r = 0; r < rows; r++) {
        const char* s = src + r * src_pitch;
        char* d = dst + r * dst_pitch;
        if (r + AHEAD < rows) {
            const char* p = src + (r + AHEAD) * src_pitch;
            for (size_t b = 0; b < width; b += 64) _mm_prefetch(p + b, _MM_HINT_NTA);
        }
        size_t b = 0;
        for (; b + 128 <= width; b += 128) {
            const __m256i v0 = _mm256_loadu_si256((const __m256i*)(s + b)), v1 = _mm256_loadu_si256((const __m256i*)(s + b + 32));
            const __m256i v2 = _mm256_loadu_si256((const __m256i*)(s + b + 64)), v3 = _mm256_loadu_si256((const __m256i*)(s + b + 96));
            _mm256_stream_si256((__m256i*)(d + b), v0);
            _mm256_stream_si256((__m256i*)(d + b + 32), v1);
            _mm256_stream_si256((__m256i*)(d + b + 64), v2);
            _mm256_stream_si256((__m256i*)(d + b + 96), v3);
        }
        for (; b + 32 <= width; b += 32) _mm256_stream_si256((__m256i*)(d + b), _mm256_loadu_si256((const __m256i*)(s + b)));
        if (b < width) memcpy(d + b, s + b, width - b);
    }
    _mm_sfence();
}

}  // namespace
#endif

void host_copy_rows(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t width, size_t rows)
{
#ifdef FASTECC_HOST_COPY_AVX2
    static const bool avx2 = __builtin_cpu_supports("avx2");
    // streaming stores want 32-byte aligned destinations in every row
    if (avx2 && width >= 64 && (((uintptr_t)dst | dst_pitch) & 31u) == 0) {
        copy_rows_avx2(dst, dst_pitch, src, src_pitch, width, rows);
        return;
    }
#endif
    for (size_t r = 0; r < rows; r++) memcpy(dst + r * dst_pitch, src + r * src_pitch, width);
}
#else
void host_copy_rows(char*, size_t, const char*, size_t, size_t, size_t) {}
#endif

}  // namespace fastecc
