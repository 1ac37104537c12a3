// gf.hpp — GF(p), p = 0xFFF00001 = 2^32 - 2^20 + 1, for gfx950 device code and the host table builder.
//
// Semantics follow the reference (GF(p).cpp:37-48 add/sub, 110-127 mul, 254-297 pow/root/inv): every
// function returns the canonical representative in [0,p).  The algorithms are our own:
//
//   * Twiddle multiplies use a Montgomery step with R = 2^32.  A constant w is stored as
//     w~ = w * 2^32 mod p, and mont(x, w~) = x * w~ / 2^32 = x * w (mod p): the DATA never leaves the
//     ordinary representation, so results are bit-identical to the reference's Barrett form.
//     Because p = 1 - 2^20 (mod 2^32), p^-1 = 1 + 2^20 (mod 2^32), so the Montgomery quotient digit
//     is one shift-add (v_lshl_add_u32) instead of a multiply; the reduction is
//         t = hi(x*w~) - hi(m*p),  m = lo(x*w~) * (1 + 2^20),   t in (-p, p)  ->  +p if negative.
//     (The low words of x*w~ and m*p are equal by construction, so there is no borrow between words.)
//   * p > 2^31, so no lazy (unreduced) intermediates fit in 32 bits: add/sub normalise every time,
//     exactly like GF(p).cpp:37-48.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GF_HD __host__ __device__ __forceinline__
#define GF_D __device__ __forceinline__
#else
#define GF_HD inline
#endif

namespace gf {

constexpr uint32_t P = 0xFFF00001u;
constexpr uint32_t GENERATOR = 19u;         // GF(p).cpp:272
constexpr uint32_t MONT_ONE = 0x000FFFFFu;  // 2^32 mod p = 2^20 - 1

// ---- host-side exact arithmetic (table generation; not performance relevant) ----
inline uint32_t h_mul(uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)x * y) % P); }
inline uint32_t h_pow(uint32_t x, uint64_t e)
{
    uint32_t r = 1;
    for (; e; e >>= 1) {
        if (e & 1) r = h_mul(r, x);
        x = h_mul(x, x);
    }
    return r;
}
inline uint32_t h_root(uint32_t order) { return h_pow(GENERATOR, (P - 1u) / order); }  // GF(p).cpp:268-276
inline uint32_t h_inv(uint32_t x) { return h_pow(x, P - 2u); }                         // GF(p).cpp:293-297
inline uint32_t h_to_mont(uint32_t w) { return (uint32_t)((((uint64_t)w) << 32) % P); }
// The table builders walk a million powers per table: one 64-bit division per entry (h_mul, h_to_mont) was 45 ms per 2^19-entry table.
// Montgomery step on the host, division-free like the device's mul_mont: a * b / 2^32 mod p for a, b < p.  With both factors in
// Montgomery form the product is again in Montgomery form, so a running power w~ <- h_mont_mul(w~, root~) yields table entries directly.
inline uint32_t h_mont_mul(uint32_t a, uint32_t b)
{
    const uint64_t t = (uint64_t)a * b;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t m = lo + (lo << 20);                      // lo * p^-1 mod 2^32, p^-1 = 1 + 2^20
    const uint32_t q = (uint32_t)(((uint64_t)m * P) >> 32);  // lo(m * p) == lo: t - m*p = (hi - q) * 2^32
    return hi >= q ? hi - q : hi - q + P;
}
constexpr uint32_t MONT_R2 = 0x0FDFFF01u;  // 2^64 mod p: x * 2^32 mod p == h_mont_mul(x, MONT_R2)

#if defined(__HIPCC__)
// ---- device arithmetic ----
GF_D uint32_t sub(uint32_t x, uint32_t y)
{
    uint32_t d;
    const bool borrow = __builtin_usub_overflow(x, y, &d);  // v_sub_co_u32
    return borrow ? d + P : d;
}

// s = x + y; u = s - p (mod 2^32) = s + (2^20 - 1).  Take u when x + y overflowed 32 bits or s >= p; the
// second condition is exactly "s + (2^20-1) carries".  Three VALU ops + one scalar OR of the carry masks
// (measured 5 % faster than x - (p - y), tools/microbench.hip).
GF_D uint32_t add(uint32_t x, uint32_t y)
{
    uint32_t s, u;
    const bool c1 = __builtin_uadd_overflow(x, y, &s);
    const bool c2 = __builtin_uadd_overflow(s, MONT_ONE, &u);
    return (c1 | c2) ? u : s;
}

// x * w mod p for a constant held in Montgomery form (wm = w * 2^32 mod p).  x may be any uint32.
//   t = x * wm                     (v_mad_u64_u32: both halves in one instruction)   t <= (2^32-1)(p-1)
//   m = lo(t) * (1 + 2^20)         (v_lshl_add_u32)           m * p == lo(t)  (mod 2^32)
//   q = hi(m * p)                  (v_mul_hi_u32)             lo(m*p) == lo(t), so t - m*p = (hi(t) - q) * 2^32 exactly
//   r = hi(t) - q  in (-p, p)      (v_sub_co_u32, +p if it borrowed)
// Equivalent forms measured on MI355X (tools/microbench.hip "bfly", profiles/r01/microbench_bfly_variants.jsonl):
// a second v_mad_u64_u32 folding the reduction (u = t + m*(2^20-1), r = hi(u) - m), mul_lo+mul_hi+mul_hi, a
// shift-only hi(m*p), and a two-word twiddle without the 64-bit product are all within 0-5 % of this one.
GF_D uint32_t mul_mont(uint32_t x, uint32_t wm)
{
    const uint64_t t = (uint64_t)x * wm;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t m = lo + (lo << 20);
    const uint32_t q = __umulhi(m, P);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(hi, q, &r);
    return borrow ? r + P : r;
}

// General product of two canonical values, Barrett form with the 32-bit reciprocal 0x001000FF
// (same estimate as GF(p).cpp:110-127; used by the element-wise test kernel and to lift values into
// Montgomery form on the device).
GF_D uint32_t mul(uint32_t x, uint32_t y)
{
    uint64_t t = (uint64_t)x * y;
    const uint64_t q = (t + (t >> 32) * 0x001000FFull) >> 32;
    t -= q * P;
    return (uint32_t)(t >= P ? t - P : t);
}
#endif

}  // namespace gf
