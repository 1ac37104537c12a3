// gf61.hpp — GF(p^2), p = 2^61 - 1, for the 64-bit-field configuration (BASELINE.json configs[4]).
//
// The reference has no code for this field (RS.cpp:86 instantiates GF(0xFFF00001) only; README.md:178 and
// GF.md:30-31 merely name the idea), so the conventions are ours and are stated in include/fastecc.h:
// an element is (re, im) = two consecutive uint64 words, i^2 = -1, generator 4 + i, w_(2^62) = (4+i)^(2^60-1).
// GF(p) alone has no roots of unity of order > 2; the extension has order p^2 - 1 = 2^62 (2^60 - 1).
//
// Device arithmetic, built for a VALU whose widest multiply is v_mad_u64_u32 (32 x 32 + 64 -> 64):
//
//   * A data component x is kept LAZY in [0, 2^61 + 2^33): congruent to the value, not necessarily canonical.
//     It is split x = x1 * 2^31 + x0 with x0 < 2^31 and x1 <= 2^30 + 4 (x >> 31 of a value below 2^61 + 2^33).
//   * A twiddle (c, d) is canonical and wave-uniform: its limbs c0 < 2^31, c1 < 2^30 and the doubled high
//     limb 2*c1 live in SGPRs, as do those of e = p - d.
//   * re = a*c + b*e and im = a*d + b*c are each accumulated as TWO 64-bit sums of four v_mad_u64_u32:
//         acc = a0*c0 + b0*e0 + a1*(2 c1) + b1*(2 e1)      < 1.5 * 2^63     (2^62 = 2 mod p: the high product is doubled)
//         mid = a0*c1 + a1*c0 + b0*e1 + b1*e0              < 2^63           (weight 2^31)
//     and  value = acc + mid * 2^31  (mod p), with mid = mh * 2^30 + ml  =>  mid * 2^31 = mh + ml * 2^31 (mod p),
//     so   t = acc + ml * 2^31 + mh < 2^64 needs one more v_mad_u64_u32 and one 64-bit add, and
//          fold(t) = (t mod 2^61) + (t >> 61) < 2^61 + 8 is lazy again (rot30 results reach 2^61 + 2^32: the bound of the lazy range).
//          No division, no conditional.
//   * add/sub fold the same way; only the last pass of a transform makes values canonical (one conditional
//     subtract) so that the stripe in HBM is bit-identical to what exact arithmetic gives.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#endif

namespace gf61 {

constexpr uint64_t P = 0x1FFFFFFFFFFFFFFFull;  // 2^61 - 1

struct Elem {
    uint64_t re, im;
};

// ---- host-side exact arithmetic (table generation and the scalar entry points of the ABI) ----
inline uint64_t h_mulp(uint64_t x, uint64_t y)
{
    // any 64-bit x, y.  2^61 = 1 (mod p): the 128-bit product folds in 61-bit pieces — no division (the tables of a decoder are
    // millions of these products on the host)
    const unsigned __int128 t = (unsigned __int128)x * y;
    const unsigned __int128 hi = t >> 61;  // < 2^67
    uint64_t r = ((uint64_t)t & P) + ((uint64_t)hi & P) + (uint64_t)(hi >> 61);  // < 2^62 + 2^6
    r = (r & P) + (r >> 61);
    return r >= P ? r - P : r;
}
inline uint64_t h_addp(uint64_t x, uint64_t y)
{
    const uint64_t s = x + y;
    return s >= P ? s - P : s;
}
inline uint64_t h_subp(uint64_t x, uint64_t y) { return x >= y ? x - y : x + P - y; }
inline Elem h_mul(Elem x, Elem y)
{
    return Elem{h_subp(h_mulp(x.re, y.re), h_mulp(x.im, y.im)), h_addp(h_mulp(x.re, y.im), h_mulp(x.im, y.re))};
}
inline Elem h_pow(Elem x, uint64_t e)
{
    Elem r{1, 0};
    for (; e; e >>= 1) {
        if (e & 1) r = h_mul(r, x);
        x = h_mul(x, x);
    }
    return r;
}
inline Elem h_inv(Elem x)
{
    const uint64_t norm = h_addp(h_mulp(x.re, x.re), h_mulp(x.im, x.im));
    uint64_t inv = 1, b = norm;
    for (uint64_t e = P - 2; e; e >>= 1) {
        if (e & 1) inv = h_mulp(inv, b);
        b = h_mulp(b, b);
    }
    return Elem{h_mulp(x.re, inv), h_mulp(h_subp(0, x.im), inv)};
}
// root of unity of order `order` (a power of two <= 2^62); (0,0) if there is none
inline Elem h_root(uint64_t order)
{
    if (order == 0 || (order & (order - 1)) != 0 || order > (1ull << 62)) return Elem{0, 0};
    const Elem w62 = h_pow(Elem{4, 1}, (1ull << 60) - 1);
    return h_pow(w62, (1ull << 62) / order);
}

#if defined(__HIPCC__)
// ---- device arithmetic ----
// Shaped for what hipcc emits on gfx950 (checked in the ISA, tools/count_valu.py): 58-60 VALU instructions per
// radix-2 butterfly, 22 of them v_mad_u64_u32, no register moves.  The round-1 formulation compiled to 117 — the
// compiler strength-reduced "x * 2^31 + acc" into 64-bit shifts, masks and adds, and built every masked 64-bit value in
// a fresh register pair with v_mov copies.  Two constants that the compiler must not see through (Opaque) keep those
// multiply-adds as the single instruction they are.
#define GF61_D __device__ __forceinline__

struct Opaque {
    uint32_t k31;  // 2^31
    uint32_t k30;  // 2^30
    uint32_t one;  // 1
};
GF61_D Opaque make_opaque()
{
    Opaque k;
    asm("s_mov_b32 %0, 0x80000000" : "=s"(k.k31));
    asm("s_mov_b32 %0, 0x40000000" : "=s"(k.k30));
    asm("s_mov_b32 %0, 1" : "=s"(k.one));
    return k;
}

GF61_D uint64_t mad64(uint32_t a, uint32_t b, uint64_t acc) { return (uint64_t)a * b + acc; }  // v_mad_u64_u32
GF61_D uint64_t join(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// t < 2^64  ->  congruent value < 2^61 + 8:  (t mod 2^61) + (t >> 61), as  and / shift / one multiply-add by 1
GF61_D uint64_t fold(uint64_t t, const Opaque& k)
{
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    return mad64(hi >> 29, k.one, join(lo, hi & 0x1FFFFFFFu));
}

// lazy + lazy -> lazy
GF61_D uint64_t add(uint64_t x, uint64_t y, const Opaque& k) { return fold(x + y, k); }
// lazy - lazy, NOT folded: < 2^63.  y < 2^61 + 2^33 < 2p, so 2p - y does not wrap.  Feeds mul() directly.
GF61_D uint64_t sub_raw(uint64_t x, uint64_t y) { return x + (2 * P - y); }
// lazy - lazy -> lazy
GF61_D uint64_t sub(uint64_t x, uint64_t y, const Opaque& k) { return fold(sub_raw(x, y), k); }
// "Loose" values: the unfolded sums and differences of two lazy values, < 1.5 * 2^62 + 2^34.  They may be added once more
// (< 2^64) or subtracted with a 4p offset (y <= 4p = 2^63 - 4; the result stays below 3.5 * 2^62 + 2^35 < 2^64) before the
// next fold; mul_raw's limb split takes any 64-bit value.
GF61_D uint64_t sub_raw4(uint64_t x, uint64_t y) { return x + (4 * P - y); }

// 4p - y for y < 2^63: the negation of a sum of two lazy values, < 2^63
GF61_D uint64_t neg_raw(uint64_t y) { return 4 * P - y; }
// x < 2^63  ->  x * 2^30 (mod p), < 2^61 + 2^32:  x = xh 2^31 + xl  =>  x 2^30 = xh 2^61 + xl 2^30 = xh + xl 2^30.
// This is what makes w_8 = 2^30 (1 + i) cheap: two multiply-adds instead of a 61 x 61-bit product.
GF61_D uint64_t rot30(uint64_t x, const Opaque& k)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    return mad64(__builtin_amdgcn_alignbit(hi, lo, 31), k.one, mad64(lo & 0x7FFFFFFFu, k.k30, 0));
}

// lazy (< 2^61 + 2^33 <= 2p) -> canonical
GF61_D uint64_t canon(uint64_t x) { return x >= P ? x - P : x; }

// The wave-uniform half of a product: limbs of a canonical twiddle (c, d) and of e = p - d.
struct Twiddle {
    uint32_t c0, c1, c1d, d0, d1, d1d, e0, e1, e1d;
};

GF61_D Twiddle make_twiddle(uint64_t c, uint64_t d)
{
    const uint64_t e = P - d;  // d = 0 gives e = p = 0 (mod p): limbs stay in range
    Twiddle w;
    w.c0 = (uint32_t)c & 0x7FFFFFFFu;
    w.c1 = (uint32_t)(c >> 31);
    w.c1d = w.c1 << 1;
    w.d0 = (uint32_t)d & 0x7FFFFFFFu;
    w.d1 = (uint32_t)(d >> 31);
    w.d1d = w.d1 << 1;
    w.e0 = (uint32_t)e & 0x7FFFFFFFu;
    w.e1 = (uint32_t)(e >> 31);
    w.e1d = w.e1 << 1;
    return w;
}

// x < 2^63 -> limbs with x = x1 2^31 + x0 (mod p), x0 < 2^31 + 4, x1 < 2^30: bits 61, 62 (weight 2^61 = 1) join the low limb
GF61_D void split_raw(uint64_t x, uint32_t& x0, uint32_t& x1)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    x1 = __builtin_amdgcn_alignbit(hi, lo, 31) & 0x3FFFFFFFu;
    x0 = (lo & 0x7FFFFFFFu) + (hi >> 29);
}
// the same for a lazy x (< 2^61 + 2^33): x1 <= 2^30 + 4 needs no mask (the products below keep their bounds with 2^30 + 4)
GF61_D void split_lazy(uint64_t x, uint32_t& x0, uint32_t& x1)
{
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    x1 = __builtin_amdgcn_alignbit(hi, lo, 31);
    x0 = lo & 0x7FFFFFFFu;
}

// acc + mid 2^31 (mod p) -> lazy.  mid = mh 2^30 + ml  =>  mid 2^31 = mh + ml 2^31 (mod p); acc < 1.5 2^63 + 2^35, mid < 2^63 + 2^35
GF61_D uint64_t combine(uint64_t acc, uint64_t mid, const Opaque& k)
{
    const uint32_t ml = (uint32_t)mid & 0x3FFFFFFFu;
    uint64_t t = mad64(ml, k.k31, acc);
    t += mid >> 30;
    return fold(t, k);
}

// (a + b i)(c + d i) = (a c + b e) + (a d + b c) i,  e = -d; limbs a0, b0 < 2^31 + 4 and a1, b1 <= 2^30 + 4
GF61_D Elem mul_limbs(uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, const Twiddle& w, const Opaque& k)
{
    const uint64_t re_acc = mad64(b1, w.e1d, mad64(a1, w.c1d, mad64(b0, w.e0, mad64(a0, w.c0, 0))));
    const uint64_t re_mid = mad64(b1, w.e0, mad64(b0, w.e1, mad64(a1, w.c0, mad64(a0, w.c1, 0))));
    const uint64_t im_acc = mad64(b1, w.c1d, mad64(a1, w.d1d, mad64(b0, w.c0, mad64(a0, w.d0, 0))));
    const uint64_t im_mid = mad64(b1, w.c0, mad64(b0, w.c1, mad64(a1, w.d0, mad64(a0, w.d1, 0))));
    return Elem{combine(re_acc, re_mid, k), combine(im_acc, im_mid, k)};
}
// x lazy
GF61_D Elem mul(Elem x, const Twiddle& w, const Opaque& k)
{
    uint32_t a0, a1, b0, b1;
    split_lazy(x.re, a0, a1);
    split_lazy(x.im, b0, b1);
    return mul_limbs(a0, a1, b0, b1, w, k);
}
// x components < 2^63 (sub_raw results)
GF61_D Elem mul_raw(Elem x, const Twiddle& w, const Opaque& k)
{
    uint32_t a0, a1, b0, b1;
    split_raw(x.re, a0, a1);
    split_raw(x.im, b0, b1);
    return mul_limbs(a0, a1, b0, b1, w, k);
}

GF61_D Elem add(Elem x, Elem y, const Opaque& k) { return Elem{add(x.re, y.re, k), add(x.im, y.im, k)}; }
GF61_D Elem sub(Elem x, Elem y, const Opaque& k) { return Elem{sub(x.re, y.re, k), sub(x.im, y.im, k)}; }
GF61_D Elem sub_raw(Elem x, Elem y) { return Elem{sub_raw(x.re, y.re), sub_raw(x.im, y.im)}; }
GF61_D Elem sub_raw4(Elem x, Elem y) { return Elem{sub_raw4(x.re, y.re), sub_raw4(x.im, y.im)}; }
GF61_D Elem fold(Elem x, const Opaque& k) { return Elem{fold(x.re, k), fold(x.im, k)}; }
GF61_D Elem canon(Elem x) { return Elem{canon(x.re), canon(x.im)}; }
// both operands per lane (the twiddle's limbs then live in VGPRs): canonical x canonical -> canonical.  Table builders, not transforms.
GF61_D Elem mul_canon(Elem x, Elem y, const Opaque& k) { return canon(mul(x, make_twiddle(y.re, y.im), k)); }
GF61_D Elem pow_canon(Elem x, uint64_t e, const Opaque& k)
{
    Elem r{1, 0};
    for (; e; e >>= 1) {
        if (e & 1u) r = mul_canon(r, x, k);
        x = mul_canon(x, x, k);
    }
    return r;
}
#endif

}  // namespace gf61
