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

// A thread holds x[j] = block (base + j*2^sl + off) for j in [0, 2^LOGR): 2^LOGR blocks at stride 2^sl, `off`
// being the block offset below that stride (off < 2^sl).  LEVELS radix-2 levels are run on them with
// in-thread partner distance 2^t, t = LEVELS-1..0 (DIF) or 0..LEVELS-1 (DIT), i.e. global level
// l = sl + t with half-size h = 2^l.  The twiddle of the butterfly whose lower block is p is
// (root of order 2h)^(p mod h) (ntt.cpp:254-283), p mod h = (m << sl) + off with m = j mod 2^t.
//
// Twiddles come from a LEVEL-PACKED table built by the host for the plan in use (plan.hip,
// build_level_table): level l occupies entries [2^l, 2^(l+1)) and is stored in the order the kernel
// consumes it, entry 2^l + (off << t) + m — so the 2^t twiddles a wave needs at one level are contiguous
// and arrive with one or two wide scalar loads (s_load_dwordx2..x16) instead of 2^t separate ones.
// Values are in Montgomery form (gf.hpp).  LO_ZERO (sl == 0, off == 0) lets exponent-0 butterflies skip
// the multiply at compile time, as the reference does for the first butterfly of a group (ntt.cpp:259-267).
// Tables (twiddles, per-block factors) are immutable while a kernel runs.  Reading them through the
// CONSTANT address space tells the compiler so: a wave-uniform address then always becomes a scalar load
// (s_load_dword*), even inside a persistent loop whose stores could otherwise be assumed to clobber it.
using const_u32_ptr = const uint32_t __attribute__((address_space(4)))*;
__device__ __forceinline__ const_u32_ptr as_constant(const uint32_t* p)
{
    return (const_u32_ptr)(reinterpret_cast<uintptr_t>(p));
}

template <int T>
__device__ __forceinline__ void load_level(uint32_t (&w)[1 << T], const uint32_t* __restrict__ twl, uint32_t off, int sl)
{
    const_u32_ptr p = as_constant(twl) + (1u << (sl + T)) + (off << T);
#pragma unroll
    for (int m = 0; m < (1 << T); ++m) w[m] = p[m];
}

template <int LOGR, int V, bool LO_ZERO, int T>
__device__ __forceinline__ void dif_one_level(uint32_t (&x)[1 << LOGR][V], const uint32_t* __restrict__ twl, uint32_t off, int sl)
{
    constexpr int R = 1 << LOGR, half = 1 << T;
    uint32_t w[half];
    load_level<T>(w, twl, off, sl);
#pragma unroll
    for (int m = 0; m < half; ++m) {
        const bool unit = LO_ZERO && m == 0;
#pragma unroll
        for (int j0 = 0; j0 < R; j0 += 2 * half) {
            const int ja = j0 + m, jb = ja + half;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint32_t a = x[ja][v], b = x[jb][v];
                x[ja][v] = gf::add(a, b);
                const uint32_t d = gf::sub(a, b);
                x[jb][v] = unit ? d : gf::mul_mont(d, w[m]);
            }
        }
    }
}

template <int LOGR, int V, bool LO_ZERO, int T>
__device__ __forceinline__ void dit_one_level(uint32_t (&x)[1 << LOGR][V], const uint32_t* __restrict__ twl, uint32_t off, int sl)
{
    constexpr int R = 1 << LOGR, half = 1 << T;
    uint32_t w[half];
    load_level<T>(w, twl, off, sl);
#pragma unroll
    for (int m = 0; m < half; ++m) {
        const bool unit = LO_ZERO && m == 0;
#pragma unroll
        for (int j0 = 0; j0 < R; j0 += 2 * half) {
            const int ja = j0 + m, jb = ja + half;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint32_t a = x[ja][v];
                const uint32_t b = unit ? x[jb][v] : gf::mul_mont(x[jb][v], w[m]);
                x[ja][v] = gf::add(a, b);
                x[jb][v] = gf::sub(a, b);
            }
        }
    }
}

template <int LOGR, int V, bool LO_ZERO, int LEVELS = LOGR>
__device__ __forceinline__ void dif_levels(uint32_t (&x)[1 << LOGR][V], const uint32_t* __restrict__ twl, uint32_t off, int sl)
{
    if constexpr (LEVELS >= 5) dif_one_level<LOGR, V, LO_ZERO, 4>(x, twl, off, sl);
    if constexpr (LEVELS >= 4) dif_one_level<LOGR, V, LO_ZERO, 3>(x, twl, off, sl);
    if constexpr (LEVELS >= 3) dif_one_level<LOGR, V, LO_ZERO, 2>(x, twl, off, sl);
    if constexpr (LEVELS >= 2) dif_one_level<LOGR, V, LO_ZERO, 1>(x, twl, off, sl);
    if constexpr (LEVELS >= 1) dif_one_level<LOGR, V, LO_ZERO, 0>(x, twl, off, sl);
    static_assert(LEVELS <= 5, "at most 5 levels per run");
}

template <int LOGR, int V, bool LO_ZERO, int LEVELS = LOGR>
__device__ __forceinline__ void dit_levels(uint32_t (&x)[1 << LOGR][V], const uint32_t* __restrict__ twl, uint32_t off, int sl)
{
    if constexpr (LEVELS >= 1) dit_one_level<LOGR, V, LO_ZERO, 0>(x, twl, off, sl);
    if constexpr (LEVELS >= 2) dit_one_level<LOGR, V, LO_ZERO, 1>(x, twl, off, sl);
    if constexpr (LEVELS >= 3) dit_one_level<LOGR, V, LO_ZERO, 2>(x, twl, off, sl);
    if constexpr (LEVELS >= 4) dit_one_level<LOGR, V, LO_ZERO, 3>(x, twl, off, sl);
    if constexpr (LEVELS >= 5) dit_one_level<LOGR, V, LO_ZERO, 4>(x, twl, off, sl);
    static_assert(LEVELS <= 5, "at most 5 levels per run");
}

}  // namespace fastecc
