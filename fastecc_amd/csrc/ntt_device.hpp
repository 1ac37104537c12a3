// ntt_device.hpp — device building blocks shared by the pass kernels (kernels.hip) and the LDS tile
// kernels (tile_kernels.hip): vector load/store, and runs of radix-2 butterfly levels held in VGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf.hpp"

namespace fastecc {

template <int V> __device__ __forceinline__ void load_vec(uint32_t (&dst)[V], const uint32_t* p)
{
    if constexpr (V == 1) {
        dst[0] = *p;
    } else if constexpr (V == 2) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        dst[0] = t.x; dst[1] = t.y;
    } else {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
    }
}

template <int V> __device__ __forceinline__ void store_vec(uint32_t* p, const uint32_t (&src)[V])
{
    if constexpr (V == 1) {
        *p = src[0];
    } else if constexpr (V == 2) {
        *reinterpret_cast<uint2*>(p) = make_uint2(src[0], src[1]);
    } else {
        *reinterpret_cast<uint4*>(p) = make_uint4(src[0], src[1], src[2], src[3]);
    }
}

__device__ __forceinline__ uint32_t bitrev(uint32_t v, int bits)
{
    return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
}

// A thread holds x[j] = block (base + j*2^s + lo) for j in [0, 2^LOGR): 2^LOGR blocks at stride 2^s, `lo`
// being the block offset below that stride (lo < 2^s).  LEVELS radix-2 levels are run on them with
// in-thread partner distance 2^t, t = LEVELS-1..0 (DIF) or 0..LEVELS-1 (DIT), i.e. global half-size
// h = 2^(s+t).  The twiddle of the butterfly whose lower block is p is (root of order 2h)^(p mod h)
// (ntt.cpp:254-283) = w_N^e with e = (p mod h) * N/(2h), p mod h = (j mod 2^t)*2^s + lo; tw[] holds w^e
// in Montgomery form for e < N/2.  All indices are wave-uniform, so the loads are scalar (s_load_dword).
// LO_ZERO (s == 0, lo == 0) lets butterflies with exponent 0 skip the multiply at compile time, as the
// reference does for its first butterfly of each group (ntt.cpp:259-267).
template <int LOGR, int V, bool LO_ZERO, int LEVELS = LOGR>
__device__ __forceinline__ void dif_levels(uint32_t (&x)[1 << LOGR][V], const uint32_t* __restrict__ tw, uint32_t lo, int s,
                                           int n)
{
    constexpr int R = 1 << LOGR;
#pragma unroll
    for (int t = LEVELS - 1; t >= 0; --t) {
        const int half = 1 << t;
#pragma unroll
        for (int m = 0; m < half; ++m) {
            const bool unit = LO_ZERO && m == 0;
            uint32_t w = 0;
            if (!unit) w = tw[(((uint32_t)m << s) + lo) << (n - 1 - s - t)];
#pragma unroll
            for (int j0 = 0; j0 < R; j0 += 2 * half) {
                const int ja = j0 + m, jb = ja + half;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const uint32_t a = x[ja][v], b = x[jb][v];
                    x[ja][v] = gf::add(a, b);
                    const uint32_t d = gf::sub(a, b);
                    x[jb][v] = unit ? d : gf::mul_mont(d, w);
                }
            }
        }
    }
}

template <int LOGR, int V, bool LO_ZERO, int LEVELS = LOGR>
__device__ __forceinline__ void dit_levels(uint32_t (&x)[1 << LOGR][V], const uint32_t* __restrict__ tw, uint32_t lo, int s,
                                           int n)
{
    constexpr int R = 1 << LOGR;
#pragma unroll
    for (int t = 0; t < LEVELS; ++t) {
        const int half = 1 << t;
#pragma unroll
        for (int m = 0; m < half; ++m) {
            const bool unit = LO_ZERO && m == 0;
            uint32_t w = 0;
            if (!unit) w = tw[(((uint32_t)m << s) + lo) << (n - 1 - s - t)];
#pragma unroll
            for (int j0 = 0; j0 < R; j0 += 2 * half) {
                const int ja = j0 + m, jb = ja + half;
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const uint32_t a = x[ja][v];
                    const uint32_t b = unit ? x[jb][v] : gf::mul_mont(x[jb][v], w);
                    x[ja][v] = gf::add(a, b);
                    x[jb][v] = gf::sub(a, b);
                }
            }
        }
    }
}

}  // namespace fastecc
