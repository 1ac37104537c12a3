// gf61_kernels.hip — the encode path over GF((2^61-1)^2): gfx950 kernels, pass plan and tables.
//
// Same operation as the 32-bit path (RS.cpp:40-63: unscaled inverse transform, block i *= w_2N^i / N, forward
// transform; butterfly ntt.cpp:16-22 / 251-284) over the field of gf61.hpp, for BASELINE.json configs[4]
// (64 KB blocks).  The reference has no code for this field, see gf61.hpp / include/fastecc.h.
//
// Data: a stripe is X[N][E] of 16-byte elements (re, im), block-major.  Mapping:
//
//      lane  <->  one element of a block               (a wave reads 64 * 16 = 1024 contiguous bytes per block)
//      wave  <->  R = 2^r blocks of one 64-element column chunk, held in VGPRs (4 per element)
//      twiddles are wave-uniform: fetched with scalar loads, their 31/30-bit limbs live in SGPRs (gf61.hpp)
//
// A pass runs r consecutive radix-2 levels in registers (DIF going down, DIT coming up, MID = the lowest
// levels of both with the per-block factor in between), exactly like the register passes of kernels.hip;
// encode = DIF passes over all levels, factor, DIT passes, and no permutation pass.  Values stay lazy
// (< 2^61 + 16) between passes; the last pass of a transform writes canonical words.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/fastecc.h"
#include "gf61.hpp"
#include "gf61_path.hpp"

namespace fastecc {
namespace p61 {

namespace {

enum { MODE_DIF = 0, MODE_DIT = 1, MODE_MID = 2, MODE_MID_FOLD = 3, MODE_DIF_GATHER = 4, MODE_DIT_SCATTER = 5, MODE_DIF_ROWS = 6, MODE_MID_ADD = 7, MODE_MID_UP = 8,
       MODE_MID_FOLD4 = 9 };  // MID_FOLD keeping every FOURTH output position (the decoder of the n = 4k codes)

// forward w_16^1, ^3, ^5, ^7 (re, im): the only general constants inside a run of levels; the inverse roots are their conjugates
struct SmallRoots {
    uint64_t w16[4][2];
};

struct PassArgs {
    const uint64_t* in;
    uint64_t* out;
    const uint64_t* tw_dif;  // level-packed twiddles for the DIF levels (16 bytes per entry)
    const uint64_t* tw_dit;  // ... for the DIT levels
    const uint64_t* dscale;  // position p -> w_2N^bitrev(p) / N
    uint32_t elems;          // elements per block covered by this launch (a column range may be narrower than a block)
    uint32_t pitch;          // elements between consecutive blocks in memory (the full block)
    uint32_t col_chunks;     // ceil(elems / 64)
    uint64_t items;          // (N >> r) * col_chunks
    int s;                   // log2 of the smallest stride of the pass
    SmallRoots sr;
    // the decoder's fused ends (MODE_DIF_GATHER, MODE_DIT_SCATTER): row u of the input is (u even ? in : in2)[u / 2] times side[u], read
    // only where side[u] != 0; row i of the output is stored, times side[i], only where side[i] != 0
    const uint64_t* in2;
    const uint64_t* side;
    const uint32_t* map;  // MODE_DIF_GATHER, optional: row u of the input is block map[u] & 0x7FFFFFFF of `in` (bit 31 set: of `in2`) instead of block u / 2 (n = 4k / 8k codes)
    uint64_t* out2;  // MODE_DIT_SCATTER, optional: the output rows are codeword positions — row i goes to (i even ? out : out2)[i / 2]
    // the split decoder (gf61_decode.hip, "even / odd split"):
    //   MODE_DIF_ROWS  block u of `in` times side[u * side_stride] on the way in (zero = block not in use, never read)
    //   MODE_MID_ADD   between the halves of MID, position p gets + addend[p >> addend_shift] * addend_factor[p]; addend is a stripe of
    //                  (N >> addend_shift) blocks in the position order a DIF leaves (block q = coefficient bitrev(q))
    //   keep           MODE_MID_ADD, optional: the tile as it is after MID's first half (before any factor) is also stored here, by position
    //   MODE_MID_UP    MID's second half alone on such a stored stripe (`in`), with the same "+ addend * factor": fastecc_repair's second chain
    uint32_t side_stride;
    const uint64_t* addend;
    const uint64_t* addend_factor;
    int addend_shift;
    uint64_t* keep;
};

using gf61::Elem;
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
using const_u64_ptr = const uint64_t __attribute__((address_space(4)))*;

__device__ __forceinline__ const_u64_ptr as_constant(const uint64_t* p) { return (const_u64_ptr)(reinterpret_cast<uintptr_t>(p)); }

__device__ __forceinline__ Elem load_elem(const uint64_t* p)
{
    // every element is read once and written once per pass: non-temporal keeps the stream from displacing itself in L2/MALL
    const u64x2 t = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(p));  // global_load_dwordx4 ... nt
    return Elem{t.x, t.y};
}

__device__ __forceinline__ void store_elem(uint64_t* p, Elem e)
{
    u64x2 t;
    t.x = e.re;
    t.y = e.im;
    __builtin_nontemporal_store(t, reinterpret_cast<u64x2*>(p));
}

// ------------------------------------------------------------------------------------------------
// Runs of radix-2 levels held in registers, in "radix-2^r" form (the reference's NTT2/NTT4 codelets, ntt.cpp:16-22 and
// 50-62, are the r = 1, 2 cases; its radix-4 butterfly multiplies by the fourth root of unity just like this one).
//
// A lane holds x[j] = block (.. + j*2^sl + off), and `LEVELS` levels pair registers at distance 2^t, t < LEVELS.  The
// twiddle of level t is (root of order 2^(sl+t+1))^((m << sl) + off), m = j mod 2^t (ntt.cpp:254-283).  It factors as
//        w_(2^(t+1))^m          a SMALL root of unity: depends on the register only
//      * w_(2^(sl+t+1))^off     the same for every butterfly of the level; commutes with all later (lower) levels
// so the second factors are collected into ONE multiplication per element, x[j] *= w_(2^(sl+LEVELS))^(off*bitrev(j)),
// after the levels (DIF) or before them (DIT), and the levels themselves only meet roots of order <= 16.  In this field
// those are cheap: w_4 = i is a swap of the components inside the add/sub that follows or precedes it, w_8 = 2^30 (1 + i)
// is two additions and two 30-bit rotations (gf61.hpp rot30); only the four odd powers of w_16 stay general.  Per 16
// elements and 4 levels: 19 general products instead of 32 (17 -> 4 in the runs next to the per-block factor, where off = 0
// and nothing is left to collect).  Exact arithmetic: every stored word is the same field element as before.
//
// Tables (k_run_table): the run of levels [sl, sl + r) owns entries [2^(sl+r), 2^(sl+r+1)); entry
// (off << r) + j is (root of order 2^(sl+r))^(off * bitrev_r(j)): the 2^r - 1 values a wave needs are contiguous.
enum { K_ONE, K_I, K_W8, K_W8I, K_GEN };
constexpr int small_kind(int t, int m)
{
    if (m == 0) return K_ONE;
    if (t >= 1 && m == (1 << (t - 1))) return K_I;
    if (t >= 2 && m == (1 << (t - 2))) return K_W8;
    if (t >= 2 && m == 3 * (1 << (t - 2))) return K_W8I;
    return K_GEN;
}

// (root of order 16)^m for odd m, in the direction of the transform (INV: conjugate)
template <bool INV>
__device__ __forceinline__ gf61::Twiddle w16_twiddle(const SmallRoots& sr, int m)
{
    const uint64_t c = sr.w16[m >> 1][0], d = sr.w16[m >> 1][1];
    return gf61::make_twiddle(c, INV ? gf61::P - d : d);  // d != 0 for these roots
}

// y = x * w_8 (KIND K_W8) or x * w_8^3 = x * w_8 * w_4 (K_W8I), x lazy -> y lazy.  Forward: w_8 = 2^30 (1 + i), w_4 = i;
// INV: the conjugates 2^30 (1 - i) and -i.
template <int KIND, bool INV>
__device__ __forceinline__ Elem mul_w8(Elem x, const gf61::Opaque& k)
{
    const uint64_t sum = x.re + x.im;  // < 2^62 + 2^34
    if constexpr (KIND == K_W8) {
        if constexpr (!INV) return Elem{gf61::rot30(gf61::sub_raw(x.re, x.im), k), gf61::rot30(sum, k)};
        else                return Elem{gf61::rot30(sum, k), gf61::rot30(gf61::sub_raw(x.im, x.re), k)};
    } else {
        // forward: (x i)(1 + i) = -(re + im) + (re - im) i;  INV: (x (-i))(1 - i) = (im - re) - (re + im) i
        if constexpr (!INV) return Elem{gf61::rot30(gf61::neg_raw(sum), k), gf61::rot30(gf61::sub_raw(x.re, x.im), k)};
        else                return Elem{gf61::rot30(gf61::sub_raw(x.im, x.re), k), gf61::rot30(gf61::neg_raw(sum), k)};
    }
}

// Folding only every other level.  A level whose outputs the NEXT level of the same run consumes may leave them "loose"
// (LOOSE_OUT: plain 64-bit sums / differences, gf61.hpp); the next level (LOOSE_IN) adds them once more or subtracts them with
// a 4p offset and folds then.  The last level of a run always folds: collected twiddles, the LDS exchange and HBM see lazy
// values only.  Saves 12 of the 20 add/sub instructions of a butterfly on every other level.
template <bool LOOSE_IN>
__device__ __forceinline__ uint64_t diff(uint64_t x, uint64_t y) { return LOOSE_IN ? gf61::sub_raw4(x, y) : gf61::sub_raw(x, y); }
template <bool LOOSE_OUT>
__device__ __forceinline__ uint64_t settle(uint64_t t, const gf61::Opaque& k) { return LOOSE_OUT ? t : gf61::fold(t, k); }

// Decimation in frequency: (a, b) -> (a + b, (a - b) * w), w the small root of `KIND`.
template <int KIND, bool INV, bool LOOSE_IN, bool LOOSE_OUT>
__device__ __forceinline__ void dif_bfly(Elem& xa, Elem& xb, const gf61::Twiddle& w, const gf61::Opaque& k)
{
    const Elem a = xa, b = xb;
    xa = Elem{settle<LOOSE_OUT>(a.re + b.re, k), settle<LOOSE_OUT>(a.im + b.im, k)};
    if constexpr (KIND == K_ONE) {
        xb = Elem{settle<LOOSE_OUT>(diff<LOOSE_IN>(a.re, b.re), k), settle<LOOSE_OUT>(diff<LOOSE_IN>(a.im, b.im), k)};
    } else if constexpr (KIND == K_I) {
        // (a - b) i = (b.im - a.im) + (a.re - b.re) i;  (a - b)(-i) = (a.im - b.im) + (b.re - a.re) i
        if constexpr (!INV) xb = Elem{settle<LOOSE_OUT>(diff<LOOSE_IN>(b.im, a.im), k), settle<LOOSE_OUT>(diff<LOOSE_IN>(a.re, b.re), k)};
        else                xb = Elem{settle<LOOSE_OUT>(diff<LOOSE_IN>(a.im, b.im), k), settle<LOOSE_OUT>(diff<LOOSE_IN>(b.re, a.re), k)};
    } else if constexpr (KIND == K_GEN) {
        xb = gf61::mul_raw(Elem{diff<LOOSE_IN>(a.re, b.re), diff<LOOSE_IN>(a.im, b.im)}, w, k);  // the limb split takes any 64-bit value
    } else {
        xb = mul_w8<KIND, INV>(Elem{gf61::fold(diff<LOOSE_IN>(a.re, b.re), k), gf61::fold(diff<LOOSE_IN>(a.im, b.im), k)}, k);
    }
}

// Decimation in time: (a, b) -> (a + b w, a - b w).
template <int KIND, bool INV, bool LOOSE_IN, bool LOOSE_OUT>
__device__ __forceinline__ void dit_bfly(Elem& xa, Elem& xb, const gf61::Twiddle& w, const gf61::Opaque& k)
{
    const Elem a = xa;
    if constexpr (KIND == K_I) {
        const Elem b = xb;
        // b i = -b.im + b.re i;  b (-i) = b.im - b.re i
        const Elem plus{settle<LOOSE_OUT>(a.re + b.im, k), settle<LOOSE_OUT>(a.im + b.re, k)};
        const Elem minus{settle<LOOSE_OUT>(diff<LOOSE_IN>(a.re, b.im), k), settle<LOOSE_OUT>(diff<LOOSE_IN>(a.im, b.re), k)};
        if constexpr (!INV) {
            xa = Elem{minus.re, plus.im};
            xb = Elem{plus.re, minus.im};
        } else {
            xa = Elem{plus.re, minus.im};
            xb = Elem{minus.re, plus.im};
        }
        return;
    }
    Elem b = xb;
    if constexpr (KIND == K_ONE) {
        xa = Elem{settle<LOOSE_OUT>(a.re + b.re, k), settle<LOOSE_OUT>(a.im + b.im, k)};
        xb = Elem{settle<LOOSE_OUT>(diff<LOOSE_IN>(a.re, b.re), k), settle<LOOSE_OUT>(diff<LOOSE_IN>(a.im, b.im), k)};
        return;
    }
    // the product is lazy whatever came in; a loose b goes through the limb split that takes any 64-bit value (K_GEN) or is
    // folded first (w_8: its rotations want values below 2^63)
    if constexpr (KIND == K_GEN) b = LOOSE_IN ? gf61::mul_raw(b, w, k) : gf61::mul(b, w, k);
    if constexpr (KIND == K_W8 || KIND == K_W8I) b = mul_w8<KIND, INV>(LOOSE_IN ? gf61::fold(b, k) : b, k);
    xa = Elem{settle<LOOSE_OUT>(a.re + b.re, k), settle<LOOSE_OUT>(a.im + b.im, k)};
    xb = Elem{settle<LOOSE_OUT>(gf61::sub_raw(a.re, b.re), k), settle<LOOSE_OUT>(gf61::sub_raw(a.im, b.im), k)};  // b is lazy here: 2p suffices
}

template <int LOGR, int T, bool INV, bool DIT, bool LOOSE_IN, bool LOOSE_OUT>
__device__ __forceinline__ void small_level(Elem (&x)[1 << LOGR], const gf61::Opaque& k, const SmallRoots& sr)
{
    static_assert(T <= 3, "roots of order <= 16 inside a run");
    constexpr int R = 1 << LOGR, half = 1 << T;
#pragma unroll
    for (int m = 0; m < half; ++m) {
        // compile-time dispatch on the kind of w_(2^(T+1))^m (the loop is fully unrolled, m is a constant in each copy)
        auto run = [&](auto kind_tag) {
            constexpr int KIND = decltype(kind_tag)::value;
            gf61::Twiddle w{};
            if constexpr (KIND == K_GEN) w = w16_twiddle<INV>(sr, m);
#pragma unroll
            for (int j0 = 0; j0 < R; j0 += 2 * half) {
                if constexpr (DIT) dit_bfly<KIND, INV, LOOSE_IN, LOOSE_OUT>(x[j0 + m], x[j0 + m + half], w, k);
                else               dif_bfly<KIND, INV, LOOSE_IN, LOOSE_OUT>(x[j0 + m], x[j0 + m + half], w, k);
            }
        };
        switch (small_kind(T, m)) {
        case K_ONE: run(std::integral_constant<int, K_ONE>{}); break;
        case K_I:   run(std::integral_constant<int, K_I>{}); break;
        case K_W8:  run(std::integral_constant<int, K_W8>{}); break;
        case K_W8I: run(std::integral_constant<int, K_W8I>{}); break;
        default:    run(std::integral_constant<int, K_GEN>{}); break;
        }
    }
}

// x[j] *= (root of order 2^(sl+LEVELS))^(off * bitrev(j mod 2^LEVELS)) for the registers with j mod 2^LEVELS != 0
// A twiddle in use is ~12 SGPRs of limbs.  Left alone, the scheduler fetches and splits all 2^LEVELS - 1 of them before the first
// product (while the tile's rows are still arriving), which for 4 levels is > 150 SGPRs: 91-117 SGPR spills and, with them, VGPR spills
// to scratch in the 7-level DIT / MID tiles of round 2.  So the raw words travel in groups of four, the next group requested before the
// current one is split and used, and scheduling barriers keep the groups apart.
template <int LOGR, int LEVELS>
__device__ __forceinline__ void collected_twiddles(Elem (&x)[1 << LOGR], const uint64_t* tw, uint32_t off, int sl, const gf61::Opaque& k)
{
    constexpr int R = 1 << LOGR, W = 1 << LEVELS;
    const_u64_ptr p = as_constant(tw) + 2 * (((size_t)1 << (sl + LEVELS)) + ((size_t)off << LEVELS));
    if constexpr (W <= 8) {
#pragma unroll
        for (int jl = 1; jl < W; ++jl) {
            const gf61::Twiddle w = gf61::make_twiddle(p[2 * jl], p[2 * jl + 1]);
#pragma unroll
            for (int j0 = 0; j0 < R; j0 += W) x[j0 + jl] = gf61::mul(x[j0 + jl], w, k);
        }
    } else {
        constexpr int GS = 4, NG = W / GS;
        uint64_t raw[2][GS][2];
#pragma unroll
        for (int i = 0; i < GS; ++i) raw[0][i][0] = p[2 * i], raw[0][i][1] = p[2 * i + 1];
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 1 < NG) {
#pragma unroll
                for (int i = 0; i < GS; ++i) raw[(grp + 1) & 1][i][0] = p[2 * ((grp + 1) * GS + i)], raw[(grp + 1) & 1][i][1] = p[2 * ((grp + 1) * GS + i) + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GS; ++i) {
                const int jl = grp * GS + i;
                if (jl == 0) continue;
                const gf61::Twiddle w = gf61::make_twiddle(raw[grp & 1][i][0], raw[grp & 1][i][1]);
#pragma unroll
                for (int j0 = 0; j0 < R; j0 += W) x[j0 + jl] = gf61::mul(x[j0 + jl], w, k);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// LEVELS < LOGR: only the low LEVELS register bits are butterfly levels (the lane holds 2^(LOGR-LEVELS) independent groups).
// LO_ZERO: off == 0 and sl == 0, nothing to collect (the first butterflies of a group in ntt.cpp:259-267).
// Which levels of a run leave their outputs loose: every other one, counted from the first level the run executes, never the
// last.  DIF runs execute T = LEVELS-1 .. 0, DIT runs T = 0 .. LEVELS-1; `step` is the position in that order.
constexpr bool loose_out(int step, int levels) { return (step % 2) == 0 && step + 1 < levels; }
constexpr bool loose_in(int step, int levels) { return step >= 1 && loose_out(step - 1, levels); }

template <int LOGR, bool LO_ZERO, bool INV, int LEVELS = LOGR>
__device__ __forceinline__ void dif_levels(Elem (&x)[1 << LOGR], const uint64_t* tw, uint32_t off, int sl, const gf61::Opaque& k,
                                           const SmallRoots& sr)
{
    static_assert(LEVELS <= 4, "at most 4 levels per run");
    // step s executes level T = LEVELS - 1 - s
    if constexpr (LEVELS >= 4) small_level<LOGR, 3, INV, false, loose_in(LEVELS - 4, LEVELS), loose_out(LEVELS - 4, LEVELS)>(x, k, sr);
    if constexpr (LEVELS >= 3) small_level<LOGR, 2, INV, false, loose_in(LEVELS - 3, LEVELS), loose_out(LEVELS - 3, LEVELS)>(x, k, sr);
    if constexpr (LEVELS >= 2) small_level<LOGR, 1, INV, false, loose_in(LEVELS - 2, LEVELS), loose_out(LEVELS - 2, LEVELS)>(x, k, sr);
    if constexpr (LEVELS >= 1) small_level<LOGR, 0, INV, false, loose_in(LEVELS - 1, LEVELS), loose_out(LEVELS - 1, LEVELS)>(x, k, sr);
    if constexpr (!LO_ZERO && LEVELS >= 1) collected_twiddles<LOGR, LEVELS>(x, tw, off, sl, k);
}

template <int LOGR, bool LO_ZERO, bool INV, int LEVELS = LOGR>
__device__ __forceinline__ void dit_levels(Elem (&x)[1 << LOGR], const uint64_t* tw, uint32_t off, int sl, const gf61::Opaque& k,
                                           const SmallRoots& sr)
{
    static_assert(LEVELS <= 4, "at most 4 levels per run");
    if constexpr (!LO_ZERO && LEVELS >= 1) collected_twiddles<LOGR, LEVELS>(x, tw, off, sl, k);
    // step s executes level T = s
    if constexpr (LEVELS >= 1) small_level<LOGR, 0, INV, true, loose_in(0, LEVELS), loose_out(0, LEVELS)>(x, k, sr);
    if constexpr (LEVELS >= 2) small_level<LOGR, 1, INV, true, loose_in(1, LEVELS), loose_out(1, LEVELS)>(x, k, sr);
    if constexpr (LEVELS >= 3) small_level<LOGR, 2, INV, true, loose_in(2, LEVELS), loose_out(2, LEVELS)>(x, k, sr);
    if constexpr (LEVELS >= 4) small_level<LOGR, 3, INV, true, loose_in(3, LEVELS), loose_out(3, LEVELS)>(x, k, sr);
}

// One register pass.  Work item = (block group g, column chunk cc); a wave owns one work item.
// INV: the DIF levels use the inverse roots (the encode's way down, and the inverse stand-alone transform); DIT levels
// always use the forward roots (the encode's way up).
template <int LOGR, int MODE, bool CANON, bool INV>
__global__ __launch_bounds__(256, 3) void p61_pass_kernel(const PassArgs a)
{
    constexpr int R = 1 << LOGR;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= a.items) return;  // wave-uniform
    const uint32_t cc = (uint32_t)(item % a.col_chunks);
    const uint32_t g = (uint32_t)(item / a.col_chunks);
    const uint32_t col = cc * 64u + lane;
    const bool live = col < a.elems;

    const int s = MODE == MODE_MID ? 0 : a.s;
    const uint32_t lo = g & ((1u << s) - 1u);
    const uint32_t hi = g >> s;
    const uint64_t base = ((uint64_t)hi << (s + LOGR)) + lo;  // first block of this group
    const uint64_t row_words = 2ull * a.pitch;

    const gf61::Opaque k = gf61::make_opaque();
    Elem x[R];
    if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = load_elem(a.in + (base + ((uint64_t)j << s)) * row_words + 2u * col);
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = Elem{0, 0};
    }

    if constexpr (MODE == MODE_DIF) {
        if (a.s == 0) dif_levels<LOGR, true, INV>(x, a.tw_dif, 0u, 0, k, a.sr);
        else          dif_levels<LOGR, false, INV>(x, a.tw_dif, lo, s, k, a.sr);
    } else if constexpr (MODE == MODE_DIT) {
        if (a.s == 0) dit_levels<LOGR, true, false>(x, a.tw_dit, 0u, 0, k, a.sr);
        else          dit_levels<LOGR, false, false>(x, a.tw_dit, lo, s, k, a.sr);
    } else {
        dif_levels<LOGR, true, true>(x, a.tw_dif, 0u, 0, k, a.sr);
        // position p = hi*R + j holds coefficient bitrev_n(p); dscale is stored in position order
        const_u64_ptr d = as_constant(a.dscale) + 2 * ((size_t)hi * R);
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = gf61::mul(x[j], gf61::make_twiddle(d[2 * j], d[2 * j + 1]), k);
        dit_levels<LOGR, true, false>(x, a.tw_dit, 0u, 0, k, a.sr);
    }

    if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j)
            store_elem(a.out + (base + ((uint64_t)j << s)) * row_words + 2u * col, CANON ? gf61::canon(x[j]) : x[j]);
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-tiled pass: LOGT = LOGR + L2 radix-2 levels per trip through HBM (register passes: at most 4).
//
// A workgroup owns T = 2^LOGT blocks x 64 element columns (1 KiB of every block row: contiguous in HBM).  A lane keeps
// R = 2^LOGR elements of ONE column in VGPRs; the workgroup's G = 2^L2 waves hold
//     layout A  blocks q = j*G + g   (j < R)  -> the high LOGR levels are in-thread, twiddles wave-uniform (off = g)
//     layout B  blocks q = g*R + k   (k < R)  -> the low L2 levels are in-thread
// and LDS is touched only to turn A into B or back: one ds_write_b128 / ds_read_b128 round trip per element, rows of
// 64 (or 32: SPLIT = 2) consecutive 16-byte elements, conflict-free.  An exchange never mixes columns, so SPLIT = 2 runs
// it one half of the columns at a time through a buffer half the size (64 KiB for the 128-block tile: two workgroups
// per CU).  Everything else — twiddle tables, level order, lazy values — is the register pass's.
//   DIF  load A -> LOGR levels -> A=>B -> L2 levels -> store B           (outer passes: s >= 1)
//   DIT  load B -> L2 levels -> B=>A -> LOGR levels -> store A
//   MID  s = 0: DIF half, multiply block p by D[bitrev(p)] (RS.cpp:51-59), DIT half: 2*LOGT levels per trip
// With k = 2^19 the encode is dif6@13, dif6@7, mid7@0, dit6@7, dit6@13: 5 trips instead of 9.
template <int LOGT, int LOGR, int MODE, bool CANON, int SPLIT, bool INV>
__global__ __launch_bounds__((1 << (LOGT - LOGR)) * 64, 4) void p61_tile_kernel(const PassArgs a)
{
    constexpr int R = 1 << LOGR, L2 = LOGT - LOGR, G = 1 << L2, WS = 64 / SPLIT;
    static_assert(L2 >= 1 && L2 <= LOGR, "tile shape");
    extern __shared__ __attribute__((aligned(16))) u64x2 lds[];  // T rows of WS elements

    const uint32_t g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x;
    const uint32_t cc = tile % a.col_chunks;
    const uint32_t grp = tile / a.col_chunks;
    // Lanes beyond a ragged block end work on the block's LAST element column as well: same loads, same arithmetic, and stores of the same
    // value to the same address from the same wave — so the kernel has no divergent region at all.  (With an "if (live)" around the stores
    // the compiler sinks the whole second half of the tile into that branch, where its scheduling barriers no longer apply.)
    const uint32_t col = min(cc * 64u + lane, a.elems - 1u);
    const int s = (MODE == MODE_MID || MODE == MODE_MID_FOLD || MODE == MODE_MID_FOLD4 || MODE == MODE_MID_ADD || MODE == MODE_MID_UP) ? 0 : a.s;
    const uint32_t lo = grp & ((1u << s) - 1u);
    const uint32_t hi = grp >> s;
    const uint64_t block0 = ((uint64_t)hi << (s + LOGT)) + lo;  // stripe block of tile row q: block0 + (q << s)
    const uint64_t row_words = 2ull * a.pitch;
    const gf61::Opaque k = gf61::make_opaque();
    const uint32_t my_round = lane / WS;
    u64x2* my_lds = lds + (lane % WS);

    auto row_a = [&](int j) { return (uint32_t)j * G + g; };
    auto row_b = [&](int j) { return g * R + (uint32_t)j; };
    Elem x[R];
    // Rows of a layout are equidistant: the word offset of the row is ONE running scalar whose updates go through an empty asm — left
    // alone, the 2 R row addresses (64 bits each) are all formed in the prologue and parked in spilled SGPRs (23-49 spills, round 3).
    auto load = [&](auto row_of) {
        uint64_t off = (block0 + ((uint64_t)row_of(0) << s)) * row_words;
        const uint64_t step = ((uint64_t)(row_of(1) - row_of(0)) << s) * row_words;
        asm volatile("" : "+s"(off));
#pragma unroll
        for (int j = 0; j < R; ++j) {
            x[j] = load_elem(a.in + off + 2u * col);
            off += step;
            asm volatile("" : "+s"(off));
        }
    };
    auto store = [&](auto row_of) {
        uint64_t off = (block0 + ((uint64_t)row_of(0) << s)) * row_words;
        const uint64_t step = ((uint64_t)(row_of(1) - row_of(0)) << s) * row_words;
        asm volatile("" : "+s"(off));
#pragma unroll
        for (int j = 0; j < R; ++j) {
            store_elem(a.out + off + 2u * col, CANON ? gf61::canon(x[j]) : x[j]);
            off += step;
            asm volatile("" : "+s"(off));
        }
    };
    // write the registers in one layout, read them back in the other; nobody may still be reading the buffer on entry
    auto exchange_of = [&](auto& v, auto wrow, auto rrow) {
        constexpr int CNT = (int)(sizeof(v) / sizeof(v[0]));
#pragma unroll
        for (int round = 0; round < SPLIT; ++round) {
            const bool mine = SPLIT == 1 || my_round == (uint32_t)round;
            if (mine) {
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    u64x2 t;
                    t.x = v[j].re;
                    t.y = v[j].im;
                    my_lds[wrow(j) * WS] = t;
                }
            }
            __syncthreads();
            if (mine) {
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    const u64x2 t = my_lds[rrow(j) * WS];
                    v[j] = Elem{t.x, t.y};
                }
            }
            if (round + 1 < SPLIT) __syncthreads();
        }
    };
    auto exchange = [&](auto wrow, auto rrow) { exchange_of(x, wrow, rrow); };

    const uint32_t off_a = (g << s) + lo;  // layout A: x[j] = block (.. + j * 2^(s+L2) + off_a)
    if constexpr (MODE == MODE_DIF || MODE == MODE_DIF_GATHER || MODE == MODE_DIF_ROWS) {
        if constexpr (MODE == MODE_DIF_ROWS) {
            // the split decoder's first pass over the data half: block u times its factor (wave-uniform); a block not in use (factor 0: lost)
            // is a zero row that is never read
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint64_t u = block0 + ((uint64_t)row_a(j) << s);
                const uint64_t fre = as_constant(a.side)[2 * u * a.side_stride], fim = as_constant(a.side)[2 * u * a.side_stride + 1];
                x[j] = Elem{0, 0};
                if ((fre | fim) != 0) x[j] = gf61::mul(load_elem(a.in + u * row_words + 2u * col), gf61::make_twiddle(fre, fim), k);
            }
        } else if constexpr (MODE == MODE_DIF_GATHER) {
            // the decoder's gather in the first pass of its transform: position u is data block u/2 or parity block u/2 times a per-position
            // factor (wave-uniform), and an erased position (factor 0) is a zero row that is never read
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint64_t u = block0 + ((uint64_t)row_a(j) << s);
                const uint64_t fre = as_constant(a.side)[2 * u], fim = as_constant(a.side)[2 * u + 1];
                x[j] = Elem{0, 0};
                if ((fre | fim) != 0) {
                    uint64_t from_second = u & 1u, block = u >> 1;
                    if (a.map) {  // (uniform) the position map of the codes whose parity sits at other positions than the odd ones
                        const uint32_t m = ((const uint32_t __attribute__((address_space(4)))*)(reinterpret_cast<uintptr_t>(a.map)))[u];
                        from_second = m >> 31;
                        block = m & 0x7FFFFFFFu;
                    }
                    x[j] = gf61::mul(load_elem((from_second ? a.in2 : a.in) + block * row_words + 2u * col), gf61::make_twiddle(fre, fim), k);
                }
            }
        } else {
            load(row_a);
        }
        dif_levels<LOGR, false, INV>(x, a.tw_dif, off_a, s + L2, k, a.sr);
        exchange(row_a, row_b);
        dif_levels<LOGR, false, INV, L2>(x, a.tw_dif, lo, s, k, a.sr);
        store(row_b);
    } else if constexpr (MODE == MODE_DIT || MODE == MODE_DIT_SCATTER) {
        load(row_b);
        dit_levels<LOGR, false, false, L2>(x, a.tw_dit, lo, s, k, a.sr);
        exchange(row_b, row_a);
        dit_levels<LOGR, false, false>(x, a.tw_dit, off_a, s + L2, k, a.sr);
        if constexpr (MODE == MODE_DIT_SCATTER) {
            // the decoder's scatter in the last pass: only the rows it is rebuilding leave, each times its factor, canonical
#pragma unroll
            for (int j = 0; j < R; ++j) asm volatile("" : "+v"(x[j].re), "+v"(x[j].im));  // (or the last level's butterflies sink into the branches below)
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint64_t i = block0 + ((uint64_t)row_a(j) << s);
                const uint64_t fre = as_constant(a.side)[2 * i], fim = as_constant(a.side)[2 * i + 1];
                uint64_t* dst = a.out2 ? ((i & 1u) ? a.out2 : a.out) + (i >> 1) * row_words : a.out + i * row_words;  // (uniform)
                if ((fre | fim) != 0) store_elem(dst + 2u * col, gf61::canon(gf61::mul(x[j], gf61::make_twiddle(fre, fim), k)));
                __builtin_amdgcn_sched_barrier(0);  // one row at a time: the products of all R rows at once do not fit the registers
            }
        } else {
            store(row_a);
        }
    } else if constexpr (MODE == MODE_MID_FOLD4) {
        // MID of a transform of which only every FOURTH output position is wanted (n = 4k codes: the data sit at the multiples of 4): two DIT
        // levels fold away, four consecutive positions into one, and what is left is the second half of the 5-level MID tile (8 registers, 4
        // waves) of the QUARTER-size transform: its low 2 levels here on the 4 folded values a lane holds, one exchange into that tile's layout
        // (waves 0-3 take 8 rows each, the others only keep the barriers), its high 3 levels with that transform's collected twiddles (a.tw_dit),
        // and the T/4 rows go to the quarter-size stripe `out`.
        static_assert(LOGT == 7 && LOGR == 4, "the folding tile is the 7-level one");
        constexpr int RQ = R / 4, GQ = 4, RS = 8;  // values per lane after the fold; the 5-level tile: 4 waves of 8 registers
        load(row_a);
        dif_levels<LOGR, false, true>(x, a.tw_dif, g, L2, k, a.sr);
        exchange(row_a, row_b);
        dif_levels<LOGR, true, true, L2>(x, a.tw_dif, 0u, 0, k, a.sr);
        const_u64_ptr d = as_constant(a.dscale) + 2 * (((size_t)hi << LOGT) + (size_t)g * R);
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = gf61::mul(x[j], gf61::make_twiddle(d[2 * j], d[2 * j + 1]), k);
        Elem y[RQ];
#pragma unroll
        for (int j = 0; j < RQ; ++j) {
            const Elem u0{gf61::add(x[4 * j].re, x[4 * j + 1].re, k), gf61::add(x[4 * j].im, x[4 * j + 1].im, k)};
            const Elem u1{gf61::add(x[4 * j + 2].re, x[4 * j + 3].re, k), gf61::add(x[4 * j + 2].im, x[4 * j + 3].im, k)};
            y[j] = Elem{gf61::add(u0.re, u1.re, k), gf61::add(u0.im, u1.im, k)};
        }
        dit_levels<2, true, false, 2>(y, a.tw_dit, 0u, 0, k, a.sr);  // folded rows g * 4 + j: the quarter-size tile's two low levels
        __syncthreads();  // every lane has finished reading the first exchange
        Elem z[RS];
        // write rows g * 4 + j (32 of them), read rows j * 4 + g in waves 0-3
#pragma unroll
        for (int round = 0; round < SPLIT; ++round) {
            const bool mine = SPLIT == 1 || my_round == (uint32_t)round;
            if (mine) {
#pragma unroll
                for (int j = 0; j < RQ; ++j) {
                    u64x2 t;
                    t.x = y[j].re;
                    t.y = y[j].im;
                    my_lds[(g * RQ + (uint32_t)j) * WS] = t;
                }
            }
            __syncthreads();
            if (mine && g < (uint32_t)GQ) {
#pragma unroll
                for (int j = 0; j < RS; ++j) {
                    const u64x2 t = my_lds[((uint32_t)j * GQ + g) * WS];
                    z[j] = Elem{t.x, t.y};
                }
            }
            if (round + 1 < SPLIT) __syncthreads();
        }
        if (g < (uint32_t)GQ) {  // (wave-uniform; no barrier below)
            dit_levels<3, false, false>(z, a.tw_dit, g, 2, k, a.sr);
            const uint64_t quarter0 = (uint64_t)hi << (LOGT - 2);
#pragma unroll
            for (int j = 0; j < RS; ++j) store_elem(a.out + (quarter0 + (uint32_t)j * GQ + g) * row_words + 2u * col, CANON ? gf61::canon(z[j]) : z[j]);
        }
    } else if constexpr (MODE == MODE_MID_FOLD) {
        // MID of a transform of which only the EVEN output positions are wanted (the decoder's x p'(x), gf61_decode.hip): at the first DIT
        // level an even position is a + b, and what is left is a DIT of half the size on the surviving positions — level l of this transform
        // on even positions is level l - 1 of the half-size one.  So after the per-block factor the R registers fold to R/2, and the rest
        // of the tile is the second half of a (LOGT-1)-level MID tile of the HALF-size transform: its low L2 levels in registers, one
        // exchange, its high LOGR-1 levels with that transform's collected twiddles (a.tw_dit = the half-size path's table), and the
        // T/2 rows go to the half-size stripe `out`.
        static_assert(LOGR - 1 >= L2, "the folded tile keeps its low levels in registers");
        constexpr int RH = R / 2;
        load(row_a);
        dif_levels<LOGR, false, true>(x, a.tw_dif, g, L2, k, a.sr);
        exchange(row_a, row_b);
        dif_levels<LOGR, true, true, L2>(x, a.tw_dif, 0u, 0, k, a.sr);
        const_u64_ptr d = as_constant(a.dscale) + 2 * (((size_t)hi << LOGT) + (size_t)g * R);
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = gf61::mul(x[j], gf61::make_twiddle(d[2 * j], d[2 * j + 1]), k);
        Elem y[RH];
#pragma unroll
        for (int j = 0; j < RH; ++j) y[j] = Elem{gf61::add(x[2 * j].re, x[2 * j + 1].re, k), gf61::add(x[2 * j].im, x[2 * j + 1].im, k)};
        dit_levels<LOGR - 1, true, false, L2>(y, a.tw_dit, 0u, 0, k, a.sr);
        __syncthreads();  // every lane has finished reading the first exchange
        exchange_of(y, [&](int j) { return g * RH + (uint32_t)j; }, [&](int j) { return (uint32_t)j * G + g; });
        dit_levels<LOGR - 1, false, false>(y, a.tw_dit, g, L2, k, a.sr);
        const uint64_t half0 = (uint64_t)hi << (LOGT - 1);
#pragma unroll
        for (int j = 0; j < RH; ++j) store_elem(a.out + (half0 + (uint32_t)j * G + g) * row_words + 2u * col, CANON ? gf61::canon(y[j]) : y[j]);
    } else {
        if constexpr (MODE == MODE_MID_UP) {
            load(row_b);  // the stored first half: position p = hi*T + g*R + j
        } else {
            load(row_a);
            dif_levels<LOGR, false, true>(x, a.tw_dif, g, L2, k, a.sr);
            exchange(row_a, row_b);
            dif_levels<LOGR, true, true, L2>(x, a.tw_dif, 0u, 0, k, a.sr);
            if constexpr (MODE == MODE_MID_ADD) {
                if (a.keep) {  // (uniform) the repair's second chain starts from these values
                    uint64_t off = (((uint64_t)hi << LOGT) + (uint64_t)g * R) * row_words;
#pragma unroll
                    for (int j = 0; j < R; ++j) {
                        store_elem(a.keep + off + 2u * col, gf61::canon(x[j]));
                        off += row_words;
                    }
                }
            }
        }
        // position p = hi*T + g*R + j holds coefficient bitrev_n(p); dscale is stored in position order
        const_u64_ptr d = as_constant(a.dscale) + 2 * (((size_t)hi << LOGT) + (size_t)g * R);
#pragma unroll
        for (int j = 0; j < R; ++j) x[j] = gf61::mul(x[j], gf61::make_twiddle(d[2 * j], d[2 * j + 1]), k);
        if constexpr (MODE == MODE_MID_ADD || MODE == MODE_MID_UP) {
            // + addend[p >> shift] * factor[p]: the other half of the split decoder's coefficient vector.  2^shift consecutive positions read
            // the same block of the (small) addend stripe: it is fetched once per run (p is wave-uniform, so is the branch).
            const uint64_t p0 = ((uint64_t)hi << LOGT) + (uint64_t)g * R;
            const_u64_ptr af = as_constant(a.addend_factor) + 2 * p0;
            if (a.addend_shift >= LOGR) {  // (uniform) the lane's R positions lie in ONE block of the addend stripe: p0 is a multiple of R
                const Elem ad = load_elem(a.addend + (p0 >> a.addend_shift) * row_words + 2u * col);
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    x[j] = gf61::add(x[j], gf61::mul(ad, gf61::make_twiddle(af[2 * j], af[2 * j + 1]), k), k);
                    __builtin_amdgcn_sched_barrier(0);  // one position at a time: the products of all R positions at once do not fit the registers
                }
            } else {
#pragma unroll
                for (int j = 0; j < R; ++j) {
                    const Elem ad = load_elem(a.addend + ((p0 + (uint64_t)j) >> a.addend_shift) * row_words + 2u * col);
                    x[j] = gf61::add(x[j], gf61::mul(ad, gf61::make_twiddle(af[2 * j], af[2 * j + 1]), k), k);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        dit_levels<LOGR, true, false, L2>(x, a.tw_dit, 0u, 0, k, a.sr);
        if constexpr (MODE != MODE_MID_UP) __syncthreads();  // every lane has finished reading the first exchange
        exchange(row_b, row_a);
        dit_levels<LOGR, false, false>(x, a.tw_dit, g, L2, k, a.sr);
        store(row_a);
    }
}

// Swap block j with block bitrev(j): only the stand-alone transform needs it (the reference permutes
// pointers instead, ntt.cpp:292-309).
__global__ __launch_bounds__(256) void p61_bitrev_rows_kernel(uint64_t* data, uint32_t elems, int n, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t j = (uint32_t)(item / col_chunks);
    const uint32_t rj = __brev(j) >> (32 - n);
    if (rj <= j) return;
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    uint64_t* pa = data + ((uint64_t)j * elems + col) * 2;
    uint64_t* pb = data + ((uint64_t)rj * elems + col) * 2;
    const Elem va = load_elem(pa), vb = load_elem(pb);
    store_elem(pa, vb);
    store_elem(pb, va);
}

__global__ __launch_bounds__(256) void p61_count_out_of_range_kernel(const uint64_t* data, uint64_t words, unsigned long long* counter)
{
    unsigned long long bad = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (uint64_t)gridDim.x * blockDim.x)
        bad += data[i] >= gf61::P;
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o);
    if ((threadIdx.x & 63u) == 0 && bad) atomicAdd(counter, bad);
}

template <int LOGR, int MODE, bool INV>
hipError_t launch_canon(bool canon, const PassArgs& a, dim3 grid, hipStream_t st)
{
    if (canon) hipLaunchKernelGGL((p61_pass_kernel<LOGR, MODE, true, INV>), grid, dim3(256), 0, st, a);
    else       hipLaunchKernelGGL((p61_pass_kernel<LOGR, MODE, false, INV>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// inverse_roots: only DIF passes run in both directions (the stand-alone transform); see p61_pass_kernel
template <int LOGR>
hipError_t launch_mode(int mode, bool canon, bool inverse_roots, const PassArgs& a, dim3 grid, hipStream_t st)
{
    switch (mode) {
    case MODE_DIF: return inverse_roots ? launch_canon<LOGR, MODE_DIF, true>(canon, a, grid, st) : launch_canon<LOGR, MODE_DIF, false>(canon, a, grid, st);
    case MODE_DIT: return launch_canon<LOGR, MODE_DIT, false>(canon, a, grid, st);
    default:       return launch_canon<LOGR, MODE_MID, true>(canon, a, grid, st);
    }
}

struct Pass {
    int mode, logr, s;
    bool canon;  // last pass of the transform: write canonical words
    bool tile = false;  // LDS-tiled kernel covering `logr` levels (p61_tile_kernel), else a register pass
};

// Tile shapes that are instantiated: 6 levels = 8 elements per lane, runs of 3 + 3 levels (8 waves, 64 KiB; no root of
// order 16 inside a run, i.e. no general product besides the collected twiddles); 7 levels = 16 elements per lane, runs
// of 4 + 3 (8 waves, 128 KiB or 2 x 64 KiB with the split exchange).
// MID alone also as 5 levels (runs of 3 + 2, 4 waves, 32 KiB): MID is the VALU-bound pass (12 levels + the factors at 6) while the outer
// passes move the stripe at 5.5 TB/s with arithmetic to spare, so k = 2^19 runs dif7, dif7, mid5, dit7, dit7 rather than dif7, dif6, mid6, ...
constexpr int tile_logr(int levels) { return levels <= 6 ? 3 : 4; }
bool tile_shape(int levels) { return levels == 6 || levels == 7; }
bool tile_shape_mid(int levels) { return levels == 5 || tile_shape(levels); }

template <int LOGT, int MODE, bool CANON, int SPLIT, bool INV>
hipError_t launch_tile_one(const PassArgs& a, unsigned tiles, hipStream_t st)
{
    constexpr int TILE_LOGR = tile_logr(LOGT);
    auto kern = p61_tile_kernel<LOGT, TILE_LOGR, MODE, CANON, SPLIT, INV>;
    constexpr int lds_bytes = (1 << LOGT) * (64 / SPLIT) * 16;
    static bool configured[64] = {};
    int dev = 0;
    if (lds_bytes > 64 * 1024 && hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (lds_bytes > 64 * 1024 && (dev < 0 || dev >= 64 || !configured[dev])) {  // > 64 KiB of dynamic LDS is opt-in per kernel and device (0.2 ms per call: not for the small tiles)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) configured[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles), dim3((1 << (LOGT - TILE_LOGR)) * 64), lds_bytes, st, a);
    return hipGetLastError();
}

template <int LOGT, int SPLIT>
hipError_t launch_tile_mode(int mode, bool canon, bool inverse_roots, const PassArgs& a, unsigned tiles, hipStream_t st)
{
    switch (mode) {  // a DIF tile is never the last pass of a transform (the plans end on MID, DIT or a register pass)
    case MODE_DIF: return inverse_roots ? launch_tile_one<LOGT, MODE_DIF, false, SPLIT, true>(a, tiles, st) : launch_tile_one<LOGT, MODE_DIF, false, SPLIT, false>(a, tiles, st);
    case MODE_DIT: return canon ? launch_tile_one<LOGT, MODE_DIT, true, SPLIT, false>(a, tiles, st) : launch_tile_one<LOGT, MODE_DIT, false, SPLIT, false>(a, tiles, st);
    case MODE_MID_FOLD:
        if constexpr (LOGT == 7) return launch_tile_one<LOGT, MODE_MID_FOLD, false, SPLIT, true>(a, tiles, st);
        else return hipErrorInvalidValue;
    case MODE_MID_FOLD4:
        if constexpr (LOGT == 7) return launch_tile_one<LOGT, MODE_MID_FOLD4, false, SPLIT, true>(a, tiles, st);
        else return hipErrorInvalidValue;
    case MODE_DIF_GATHER: return launch_tile_one<LOGT, MODE_DIF_GATHER, false, SPLIT, true>(a, tiles, st);
    case MODE_DIF_ROWS: return launch_tile_one<LOGT, MODE_DIF_ROWS, false, SPLIT, true>(a, tiles, st);
    case MODE_MID_ADD: return launch_tile_one<LOGT, MODE_MID_ADD, false, SPLIT, true>(a, tiles, st);
    case MODE_MID_UP: return launch_tile_one<LOGT, MODE_MID_UP, false, SPLIT, true>(a, tiles, st);
    case MODE_DIT_SCATTER: return launch_tile_one<LOGT, MODE_DIT_SCATTER, true, SPLIT, false>(a, tiles, st);
    default:       return canon ? launch_tile_one<LOGT, MODE_MID, true, SPLIT, true>(a, tiles, st) : launch_tile_one<LOGT, MODE_MID, false, SPLIT, true>(a, tiles, st);
    }
}

}  // namespace

struct Path {
    int n = 0;
    uint64_t N = 0, elems = 0;
    int levels = DEFAULT_LEVELS;  // levels per register pass
    bool tiles = true;            // LDS-tiled passes where a chunk of the plan has a tile shape
    int split = 2;                // 7-level tiles: exchange rounds (2 = 64 KiB of LDS, two workgroups per CU)
    int force_mid = 0;            // > 0: MID covers exactly that many levels (the decoder's folded transform pairs a 7-level MID with a 6-level one)
    std::vector<Pass> enc, fwd;  // encode plan; stand-alone transform plan (all DIF, then the block permutation)
    uint64_t* tw_fwd = nullptr;  // forward roots, level-packed for the encode plan
    uint64_t* tw_inv = nullptr;  // inverse roots, the same packing
    uint64_t* tw_ntt_fwd = nullptr;  // the two again, packed for the stand-alone transform's plan when its register runs
    uint64_t* tw_ntt_inv = nullptr;  // differ from the encode plan's (a MID tile there, register passes here); else null
    uint64_t* dscale = nullptr;  // per-block factors by position; `cosets` tables of N elements back to back
    int cosets = 1;              // 1: the (2k,k) code; 3 / 7: n = 4k / 8k (one table per coset of evaluation points, create_cosets)
    SmallRoots sr{};  // forward w_16^(1,3,5,7)
    std::string text;
};

namespace {

int fail(char* detail, size_t cap, hipError_t e, const char* what)
{
    if (detail && cap) snprintf(detail, cap, "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}

// The encode plan is [DIF chunks, top down][MID over the lowest m levels][DIT chunks, bottom up].  With tiles, a chunk of
// 6 or 7 levels is one LDS-tiled pass and anything else falls back to register passes of at most L levels; the split is
// the one with the fewest trips through HBM (ties: fewest register passes).  Without tiles: MID takes min(n, L) levels
// and the rest is cut into near-equal chunks of at most L levels, as in round 1.
struct Chunking {
    int mid = 0;
    std::vector<int> outer;  // top down
    int trips = 1 << 30, reg = 1 << 30;
};

int reg_passes(int levels, int L) { return (levels + L - 1) / L; }

Chunking choose_chunks(int n, int L, bool tiles, int force_mid = 0)
{
    Chunking best;
    auto consider = [&](int mid, const std::vector<int>& outer) {
        int trips = (tiles && tile_shape_mid(mid)) ? 1 : reg_passes(mid, L), reg = (tiles && tile_shape_mid(mid)) ? 0 : 1;
        for (int c : outer) {
            const bool t = tiles && tile_shape(c);
            trips += 2 * (t ? 1 : reg_passes(c, L));
            reg += t ? 0 : 2;
        }
        if (trips < best.trips || (trips == best.trips && reg < best.reg)) {
            best.mid = mid;
            best.outer = outer;
            best.trips = trips;
            best.reg = reg;
        }
    };
    const int mid_max = std::min(n, tiles ? 7 : L);
    for (int mid = 1; mid <= mid_max; mid++) {
        if (!tiles && mid != mid_max) continue;
        if (force_mid > 0 && mid != force_mid) continue;
        const int rest = n - mid;
        if (rest == 0) {
            consider(mid, {});
            continue;
        }
        const int cmax = tiles ? 7 : L;
        for (int q = (rest + cmax - 1) / cmax; q <= rest && q <= (rest + cmax - 1) / cmax + 1; q++) {
            // q near-equal chunks, largest first
            std::vector<int> outer;
            for (int i = 0; i < q; i++) outer.push_back(rest / q + (i < rest % q ? 1 : 0));
            consider(mid, outer);
        }
    }
    return best;
}

void push_chunk(std::vector<Pass>& plan, int mode, int levels, int s, int L, bool tiles)
{
    if (tiles && (mode == MODE_MID ? tile_shape_mid(levels) : tile_shape(levels) && s >= 1)) {
        Pass q{mode, levels, s, false};
        q.tile = true;
        plan.push_back(q);
        return;
    }
    // register passes of at most L levels: top down for DIF, bottom up for DIT
    const int q = reg_passes(levels, L);
    std::vector<int> parts;
    for (int i = 0; i < q; i++) parts.push_back(levels / q + (i < levels % q ? 1 : 0));
    if (mode == MODE_DIT) {
        int ss = s;
        for (size_t i = parts.size(); i-- > 0;) {
            plan.push_back(Pass{mode, parts[i], ss, false});
            ss += parts[i];
        }
    } else {
        int ss = s + levels;
        for (int r : parts) {
            ss -= r;
            plan.push_back(Pass{mode, r, ss, false});
        }
    }
}

void build_plans(Path* p)
{
    const int n = p->n, L = p->levels;
    Chunking ch = choose_chunks(n, L, p->tiles, p->force_mid);
    if (ch.trips == (1 << 30)) ch = choose_chunks(n, L, p->tiles);  // the forced size does not exist for this n
    // a register MID pass covers at most L levels: the rest of a longer MID chunk becomes DIF / DIT passes around it
    int mid = ch.mid;
    std::vector<int> outer = ch.outer;
    if (!(p->tiles && tile_shape_mid(mid)) && mid > L) {
        outer.push_back(mid - L);
        mid = L;
    }
    p->enc.clear();
    p->fwd.clear();
    int top = n;
    for (int c : outer) {
        push_chunk(p->enc, MODE_DIF, c, top - c, L, p->tiles);
        top -= c;
    }
    // the stand-alone transform: the same DIF chunks, then the lowest levels as DIF register passes (natural order out
    // needs the block permutation anyway)
    p->fwd = p->enc;
    push_chunk(p->fwd, MODE_DIF, mid, 0, L, false);
    p->fwd.back().canon = true;
    push_chunk(p->enc, MODE_MID, mid, 0, L, p->tiles);
    for (size_t i = outer.size(); i-- > 0;) {
        push_chunk(p->enc, MODE_DIT, outer[i], top, L, p->tiles);
        top += outer[i];
    }
    p->enc.back().canon = true;

    p->text.clear();
    char buf[32];
    for (const Pass& q : p->enc) {
        snprintf(buf, sizeof buf, "%s%s%s%d@%d", p->text.empty() ? "" : ",", q.tile ? "T64:" : "",
                 q.mode == MODE_DIF ? "dif" : q.mode == MODE_DIT ? "dit" : "mid", q.logr, q.s);
        p->text += buf;
    }
    p->text += " gf61^2";
}

// Level l is executed by a register run whose smallest stride is 2^sl[l] (see ntt_device.hpp).
// The register runs of a plan's DIF side (the DIT side mirrors it): levels [sl, sl + r) each.
struct Run {
    int sl, r;
    bool operator==(const Run& o) const { return sl == o.sl && r == o.r; }
};

std::vector<Run> runs_of(const std::vector<Pass>& plan)
{
    std::vector<Run> runs;
    for (const Pass& q : plan) {
        if (q.mode == MODE_DIT) continue;
        if (q.tile) {  // a tile runs its low l2 levels at stride 2^s and the high tile_logr ones at 2^(s + l2)
            const int l2 = q.logr - tile_logr(q.logr);
            runs.push_back(Run{q.s + l2, tile_logr(q.logr)});
            runs.push_back(Run{q.s, l2});
        } else {
            runs.push_back(Run{q.s, q.logr});
        }
    }
    return runs;
}

// Collected twiddles (see dif_levels): the run of levels [sl, sl + r) owns entries [2^(sl+r), 2^(sl+r+1)); entry
// (off << r) + j = (root of order 2^(sl+r))^(off * bitrev_r(j)).  2N entries of 16 bytes.  Replaces the roots[] array of
// ntt.cpp:397-402 and the running root_i *= root of ntt.cpp:270-281.  Built on the device: every entry is one power (a decoder owns some twenty
// paths; their tables on the host, 10 ns per product, were most of its first fastecc_decode_prepare).
__global__ __launch_bounds__(256) void k_run_table(uint64_t* __restrict__ tab, uint64_t root_re, uint64_t root_im, int sl, int r)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // (off << r) + j
    if (t >= (1ull << (sl + r))) return;
    const uint64_t off = t >> r, j = t & ((1ull << r) - 1u);
    const uint64_t i = r ? __brevll(j) >> (64 - r) : 0;  // i = bitrev_r(j)
    const gf61::Opaque k = gf61::make_opaque();
    const Elem w = gf61::pow_canon(Elem{root_re, root_im}, off * i, k);
    const uint64_t entry = (1ull << (sl + r)) + t;
    tab[2 * entry] = w.re;
    tab[2 * entry + 1] = w.im;
}

// per-block factors c * w^i (RS.cpp:51-54: c = 1/N, w of order 2N), stored by position: position q holds coefficient bitrev_n(q)
// (by_index: c * i instead — the derivative's factor in the decoder's x p'(x), gf61_decode.hip)
// (plus: + plus_re, a real constant — the split decoder's factor (2i + k) / 2k = i / k + 1 / 2, FACTOR_SPLIT)
__global__ __launch_bounds__(256) void k_block_factors(uint64_t* __restrict__ d, uint64_t c_re, uint64_t c_im, uint64_t w_re, uint64_t w_im, int n,
                                                       bool by_index, uint64_t plus_re = 0)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= (1ull << n)) return;
    const uint64_t i = n ? __brevll(q) >> (64 - n) : 0;
    const gf61::Opaque k = gf61::make_opaque();
    const Elem v = gf61::mul_canon(by_index ? Elem{i, 0} : gf61::pow_canon(Elem{w_re, w_im}, i, k), Elem{c_re, c_im}, k);
    d[2 * q] = gf61::canon(gf61::add(v.re, plus_re, k));
    d[2 * q + 1] = v.im;
}
// the split decoder's factor of the parity half's coefficients, by position: position q holds coefficient m = bitrev_n(q): -1/2 * w^(-m), w of order 2k
// (wi = w^-1)
__global__ __launch_bounds__(256) void k_split_addend_factors(uint64_t* __restrict__ d, uint64_t c_re, uint64_t c_im, uint64_t wi_re, uint64_t wi_im, int n)
{
    const uint64_t q = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (q >= (1ull << n)) return;
    const uint64_t m = n ? __brevll(q) >> (64 - n) : 0;
    const gf61::Opaque k = gf61::make_opaque();
    const Elem v = gf61::mul_canon(gf61::pow_canon(Elem{wi_re, wi_im}, m, k), Elem{c_re, c_im}, k);
    d[2 * q] = v.re;
    d[2 * q + 1] = v.im;
}

int device_run_table(uint64_t** dst, int n, gf61::Elem root_of_order_N, const std::vector<Run>& runs, char* detail, size_t cap)
{
    const size_t words = 4 * std::max<size_t>((size_t)1 << n, 2);
    hipError_t e = hipSuccess;
    if (!*dst) e = hipMalloc((void**)dst, words * 8);
    if (e != hipSuccess) return fail(detail, cap, e, "hipMalloc(gf61 table)");
    e = hipMemsetAsync(*dst, 0, words * 8, nullptr);
    if (e != hipSuccess) return fail(detail, cap, e, "hipMemsetAsync(gf61 table)");
    for (const Run& run : runs) {
        const int lev = run.sl + run.r;
        const gf61::Elem root = gf61::h_pow(root_of_order_N, 1ull << (n - lev));  // order 2^(sl + r)
        const uint64_t count = 1ull << lev;
        hipLaunchKernelGGL(k_run_table, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, nullptr, *dst, root.re, root.im, run.sl, run.r);
        e = hipGetLastError();
        if (e != hipSuccess) return fail(detail, cap, e, "gf61 twiddle table kernel");
    }
    return FASTECC_OK;
}

int upload(uint64_t** dst, const std::vector<uint64_t>& src, char* detail, size_t cap)
{
    hipError_t e = hipSuccess;
    if (!*dst) e = hipMalloc((void**)dst, src.size() * 8);
    if (e != hipSuccess) return fail(detail, cap, e, "hipMalloc(gf61 table)");
    e = hipMemcpy(*dst, src.data(), src.size() * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(detail, cap, e, "hipMemcpy(gf61 table)");
    return FASTECC_OK;
}

// (kernels on the null stream; the caller synchronises once all tables of a path are under way)
int upload_tables(Path* p, char* detail, size_t cap)
{
    const gf61::Elem wN = gf61::h_root(p->N), wNi = gf61::h_inv(wN);
    const std::vector<Run> sl = runs_of(p->enc), sl_ntt = runs_of(p->fwd);
    int rc = device_run_table(&p->tw_fwd, p->n, wN, sl, detail, cap);
    if (rc == FASTECC_OK) rc = device_run_table(&p->tw_inv, p->n, wNi, sl, detail, cap);
    if (sl_ntt == sl) {
        if (p->tw_ntt_fwd) (void)hipFree(p->tw_ntt_fwd);
        if (p->tw_ntt_inv) (void)hipFree(p->tw_ntt_inv);
        p->tw_ntt_fwd = p->tw_ntt_inv = nullptr;
    } else {
        if (rc == FASTECC_OK) rc = device_run_table(&p->tw_ntt_fwd, p->n, wN, sl_ntt, detail, cap);
        if (rc == FASTECC_OK) rc = device_run_table(&p->tw_ntt_inv, p->n, wNi, sl_ntt, detail, cap);
    }
    return rc;
}

struct Scope {
    const LaunchHooks* h;
    hipStream_t st;
    Scope(const LaunchHooks* h_, hipStream_t st_, const char* name, uint64_t bytes) : h(h_), st(st_)
    {
        if (h) h->begin(h->user, st, name, bytes);
    }
    ~Scope()
    {
        if (h) h->end(h->user, st);
    }
};

// The decoder's ends of a run of passes (encode_fold): the first pass gathers, the last one scatters (see PassArgs).
struct FusedEnds {
    const uint64_t* first_in2 = nullptr;
    const uint64_t* first_side = nullptr;  // set: plan[0] must be a DIF tile
    const uint32_t* first_map = nullptr;   // the gather's position map (PassArgs::map)
    const uint64_t* last_side = nullptr;   // set: plan.back() must be a canonical DIT tile; it then writes to last_out
    uint64_t* last_out = nullptr;
    uint64_t* last_out2 = nullptr;         // set: the rows are codeword positions, even ones go to last_out, odd ones here (PassArgs::out2)
    // the split decoder: the first pass takes block u of `in` times first_side[u * first_rows_stride] (MODE_DIF_ROWS instead of the gather);
    // MID adds mid_addend[p >> mid_shift] * mid_addend_factor[p] between its halves (MODE_MID_ADD; MID must be a tile that is not the last pass)
    uint32_t first_rows_stride = 0;
    const uint64_t* mid_addend = nullptr;
    const uint64_t* mid_addend_factor = nullptr;
    int mid_shift = 0;
    uint64_t* mid_keep = nullptr;  // MODE_MID_ADD also stores the tile after its first half here
    bool mid_up = false;           // MID runs its second half only, on such a stored stripe (MODE_MID_UP; MID is then the plan's first pass)
};

// inverse_roots: tw_dif holds inverse roots (selects the conjugate small roots inside the DIF runs)
int run_passes(Path* p, const std::vector<Pass>& plan, const uint64_t* in, uint64_t* out, const uint64_t* tw_dif,
               const uint64_t* tw_dit, bool inverse_roots, hipStream_t st, const LaunchHooks* hooks, uint64_t col0 = 0, uint64_t width = 0,
               const FusedEnds* ends = nullptr, const uint64_t* dscale = nullptr)
{
    if (width == 0) width = p->elems;
    in += 2 * col0;  // element columns [col0, col0 + width) of every block: independent transforms
    out += 2 * col0;
    const uint64_t* src = in;
    for (size_t qi = 0; qi < plan.size(); qi++) {
        const Pass& q = plan[qi];
        int mode = q.mode;
        PassArgs a{};
        a.in = src;
        a.out = out;
        if (ends && ends->first_side && qi == 0) {
            if (!q.tile || q.mode != MODE_DIF || !inverse_roots) return FASTECC_E_UNSUPPORTED;
            mode = ends->first_rows_stride ? MODE_DIF_ROWS : MODE_DIF_GATHER;
            a.in2 = ends->first_in2;
            a.side = ends->first_side;
            a.map = ends->first_rows_stride ? nullptr : ends->first_map;
            a.side_stride = ends->first_rows_stride;
        }
        if (ends && ends->mid_addend && q.mode == MODE_MID) {
            if (!q.tile || q.canon || qi + 1 == plan.size()) return FASTECC_E_UNSUPPORTED;
            mode = ends->mid_up ? MODE_MID_UP : MODE_MID_ADD;
            a.addend = ends->mid_addend + 2 * col0;
            a.addend_factor = ends->mid_addend_factor;
            a.addend_shift = ends->mid_shift;
            a.keep = ends->mid_keep && !ends->mid_up ? ends->mid_keep + 2 * col0 : nullptr;
        }
        if (ends && ends->last_side && qi + 1 == plan.size()) {
            if (!q.tile || q.mode != MODE_DIT || !q.canon || mode != q.mode) return FASTECC_E_UNSUPPORTED;
            mode = MODE_DIT_SCATTER;
            a.side = ends->last_side;
            a.out = ends->last_out;
            a.out2 = ends->last_out2;
        }
        a.tw_dif = tw_dif;
        a.tw_dit = tw_dit;
        a.dscale = dscale ? dscale : p->dscale;
        a.elems = (uint32_t)width;
        a.pitch = (uint32_t)p->elems;
        a.col_chunks = (uint32_t)((width + 63) / 64);
        a.items = (p->N >> q.logr) * a.col_chunks;
        a.s = q.s;
        a.sr = p->sr;
        const uint64_t blocks = q.tile ? a.items : (a.items + 3) / 4;  // a workgroup per tile / a wave per work item
        if (blocks > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
        const dim3 grid((unsigned)blocks);
        char name[32];
        snprintf(name, sizeof name, "p61_%s%s%d%s", q.tile ? "tile_" : "", q.mode == MODE_DIF ? "dif" : q.mode == MODE_DIT ? "dit" : "mid", q.logr,
                 mode == MODE_DIF_GATHER ? "_gather" : mode == MODE_DIT_SCATTER ? "_scatter" : mode == MODE_DIF_ROWS ? "_rows" : mode == MODE_MID_ADD ? "_add" : mode == MODE_MID_UP ? "_up" : "");
        Scope sc(hooks, st, name, 2ull * p->N * width * 16ull);
        hipError_t e;
        if (q.tile) {
            if (q.logr == 5) {  // MID only (tile_shape_mid)
                if (mode == MODE_MID_ADD) e = p->split == 2 ? launch_tile_one<5, MODE_MID_ADD, false, 2, true>(a, (unsigned)blocks, st)
                                                            : launch_tile_one<5, MODE_MID_ADD, false, 1, true>(a, (unsigned)blocks, st);
                else if (mode == MODE_MID_UP) e = p->split == 2 ? launch_tile_one<5, MODE_MID_UP, false, 2, true>(a, (unsigned)blocks, st)
                                                                : launch_tile_one<5, MODE_MID_UP, false, 1, true>(a, (unsigned)blocks, st);
                else if (mode != MODE_MID) return FASTECC_E_UNSUPPORTED;
                else if (p->split == 2) e = q.canon ? launch_tile_one<5, MODE_MID, true, 2, true>(a, (unsigned)blocks, st) : launch_tile_one<5, MODE_MID, false, 2, true>(a, (unsigned)blocks, st);
                else e = q.canon ? launch_tile_one<5, MODE_MID, true, 1, true>(a, (unsigned)blocks, st) : launch_tile_one<5, MODE_MID, false, 1, true>(a, (unsigned)blocks, st);
            } else if (q.logr == 6 && p->split == 2) e = launch_tile_mode<6, 2>(mode, q.canon, inverse_roots, a, (unsigned)blocks, st);
            else if (q.logr == 6) e = launch_tile_mode<6, 1>(mode, q.canon, inverse_roots, a, (unsigned)blocks, st);
            else if (p->split == 2) e = launch_tile_mode<7, 2>(mode, q.canon, inverse_roots, a, (unsigned)blocks, st);
            else e = launch_tile_mode<7, 1>(mode, q.canon, inverse_roots, a, (unsigned)blocks, st);
            if (e != hipSuccess) return fail(nullptr, 0, e, "p61 tile pass");
            src = out;
            continue;
        }
        switch (q.logr) {
        case 1: e = launch_mode<1>(q.mode, q.canon, inverse_roots, a, grid, st); break;
        case 2: e = launch_mode<2>(q.mode, q.canon, inverse_roots, a, grid, st); break;
        case 3: e = launch_mode<3>(q.mode, q.canon, inverse_roots, a, grid, st); break;
        case 4: e = launch_mode<4>(q.mode, q.canon, inverse_roots, a, grid, st); break;
        default: return FASTECC_E_UNSUPPORTED;
        }
        if (e != hipSuccess) return fail(nullptr, 0, e, "p61 pass");
        src = out;  // after the first pass everything is in place on `out`
    }
    return FASTECC_OK;
}

}  // namespace

int create(Path** out, int n, uint64_t elems, char* detail, size_t cap) { return create_transform(out, n, elems, FACTOR_ENCODE, detail, cap); }

// n = 4k / 8k (cosets = 3 / 7): the parity is f on the cosets g <w_k> of the data points inside the n-th roots of unity, in the nesting order of
// include/fastecc.h — g = w_2k; w_4k, w_4k^3; w_8k, w_8k^3, w_8k^5, w_8k^7 — block j of coset g being f(g w_k^j): RS.cpp:40-63 with g in place
// of root(2N).  One table of per-block factors g^m / N per coset; the DIF half of the encode is shared (encode_cosets).
int create_cosets(Path** out, int n, uint64_t elems, int cosets, char* detail, size_t cap)
{
    if (cosets != 3 && cosets != 7) return FASTECC_E_INVAL;
    int rc = create_transform(out, n, elems, FACTOR_ENCODE, detail, cap);
    if (rc != FASTECC_OK) return rc;
    Path* p = *out;
    uint64_t* all = nullptr;
    hipError_t e = hipMalloc((void**)&all, (size_t)cosets * 2 * p->N * 8);
    if (e == hipSuccess) {
        const gf61::Elem c = gf61::h_inv(gf61::Elem{p->N % gf61::P, 0});
        for (int t = 0; t < cosets && e == hipSuccess; t++) {
            // coset t: generator w_(2^j k)^odd with j = floor(log2(t + 1)) + 1 and the (t + 2 - 2^(j-1))-th odd exponent
            int j = 1;
            while ((1 << j) - 1 <= t) j++;
            const uint64_t odd = 2ull * (uint64_t)(t + 1 - (1 << (j - 1))) + 1ull;
            const gf61::Elem g = gf61::h_pow(gf61::h_root(p->N << j), odd);
            hipLaunchKernelGGL(k_block_factors, dim3((unsigned)((p->N + 255) / 256)), dim3(256), 0, nullptr, all + (size_t)t * 2 * p->N, c.re, c.im, g.re, g.im, n, false);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
    if (e != hipSuccess) {
        if (all) (void)hipFree(all);
        rc = fail(detail, cap, e, "gf61 coset factors");
        destroy(p);
        *out = nullptr;
        return rc;
    }
    (void)hipFree(p->dscale);
    p->dscale = all;
    p->cosets = cosets;
    return FASTECC_OK;
}
int cosets_of(const Path* p) { return p ? p->cosets : 1; }

int create_transform(Path** out, int n, uint64_t elems, int factor, char* detail, size_t cap) { return create_transform_mid(out, n, elems, factor, 0, detail, cap); }

// The decoder's folded transform (only the even output positions of a size-2^(n+1) transform are wanted): `big` is that transform with a
// 7-level MID tile, `half` a size-2^n path with a 6-level MID tile.  DIF passes of `big` (in -> work, then in place), the folding MID tile
// (work -> out, 2^n blocks), DIT passes of `half` above its MID, in place on `out`.  FASTECC_E_UNSUPPORTED when the two plans do not pair up.
namespace {
// the two paths pair up when big's MID is a 7-level tile and the small one's a 6-level tile (half the size) or a 5-level tile (a quarter of
// the size), same exchange split
bool fold_pairs(Path* big, Path* half, size_t* mb_out, size_t* mh_out)
{
    if (!big || !half || (big->n != half->n + 1 && big->n != half->n + 2) || big->elems != half->elems) return false;
    const int f = big->n - half->n;
    size_t mb = 0, mh = 0;
    while (mb < big->enc.size() && big->enc[mb].mode != MODE_MID) mb++;
    while (mh < half->enc.size() && half->enc[mh].mode != MODE_MID) mh++;
    if (mb == big->enc.size() || mh == half->enc.size()) return false;
    const Pass &qb = big->enc[mb], &qh = half->enc[mh];
    if (!qb.tile || qb.logr != 7 || !qh.tile || qh.logr != 7 - f || big->split != half->split) return false;
    *mb_out = mb;
    *mh_out = mh;
    return true;
}
}  // namespace

int fold_caps(Path* big, Path* half)
{
    size_t mb = 0, mh = 0;
    if (!fold_pairs(big, half, &mb, &mh)) return 0;
    int caps = FOLD_PAIRS;
    if (mb > 0 && big->enc[0].tile && big->enc[0].mode == MODE_DIF) caps |= FOLD_GATHERS;
    if (mh + 1 < half->enc.size() && half->enc.back().tile && half->enc.back().mode == MODE_DIT && half->enc.back().canon) caps |= FOLD_SCATTERS;
    return caps;
}

int encode_fold(Path* big, Path* half, const uint64_t* in, uint64_t* work, uint64_t* out, hipStream_t st, const LaunchHooks* hooks, const FoldEnds* ends)
{
    size_t mb = 0, mh = 0;
    if (!fold_pairs(big, half, &mb, &mh)) return FASTECC_E_UNSUPPORTED;
    const int caps = fold_caps(big, half);
    if (ends && ((ends->fin && !(caps & FOLD_GATHERS)) || (ends->gout && !(caps & FOLD_SCATTERS)))) return FASTECC_E_UNSUPPORTED;
    const std::vector<Pass> down(big->enc.begin(), big->enc.begin() + mb), up(half->enc.begin() + mh + 1, half->enc.end());
    const uint64_t* src = in;
    if (!down.empty()) {
        FusedEnds fe;
        if (ends && ends->fin) {
            fe.first_in2 = ends->parity;
            fe.first_side = ends->fin;
            fe.first_map = ends->map;
        }
        const int rc = run_passes(big, down, in, work, big->tw_inv, big->tw_fwd, true, st, hooks, 0, 0, &fe);
        if (rc != FASTECC_OK) return rc;
        src = work;
    }
    {
        PassArgs a{};
        a.in = src;
        a.out = out;
        a.tw_dif = big->tw_inv;
        a.tw_dit = half->tw_fwd;
        a.dscale = big->dscale;
        a.elems = (uint32_t)big->elems;
        a.pitch = (uint32_t)big->elems;
        a.col_chunks = (uint32_t)((big->elems + 63) / 64);
        a.items = (big->N >> 7) * a.col_chunks;
        a.s = 0;
        a.sr = big->sr;
        if (a.items > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
        const int fold_mode = big->n == half->n + 2 ? MODE_MID_FOLD4 : MODE_MID_FOLD;
        Scope sc(hooks, st, fold_mode == MODE_MID_FOLD4 ? "p61_tile_mid7_fold4" : "p61_tile_mid7_fold", (big->N + half->N) * big->elems * 16ull);
        const hipError_t e = big->split == 2 ? launch_tile_mode<7, 2>(fold_mode, false, true, a, (unsigned)a.items, st)
                                             : launch_tile_mode<7, 1>(fold_mode, false, true, a, (unsigned)a.items, st);
        if (e != hipSuccess) return fail(nullptr, 0, e, "p61 folding MID tile");
    }
    if (up.empty()) return FASTECC_OK;  // (the MID tile then wrote lazy values: the caller's next step accepts them)
    FusedEnds fe;
    if (ends && ends->gout) {
        fe.last_side = ends->gout;
        fe.last_out = ends->data_out;
    }
    return run_passes(half, up, out, out, half->tw_inv, half->tw_fwd, true, st, hooks, 0, 0, &fe);
}

// The whole transform of `p` with the decoder's ends fused in (fastecc_repair in one transform): input position u = (u even ? data : parity)[u / 2]
// times fin[u] (read where fin[u] != 0), output position u times gout_all[u] written to (u even ? data_out : parity_out)[u / 2] where that
// factor is not zero.  work: the path's 2^n-block stripe.  FASTECC_E_UNSUPPORTED when the plan does not start with a DIF tile and end on a DIT tile.
int encode_ends(Path* p, const uint64_t* data, const uint64_t* parity, const uint64_t* fin, uint64_t* work, const uint64_t* gout_all, uint64_t* data_out,
                uint64_t* parity_out, hipStream_t st, const LaunchHooks* hooks)
{
    if (!p || p->enc.size() < 2) return FASTECC_E_UNSUPPORTED;
    const Pass &first = p->enc.front(), &last = p->enc.back();
    if (!first.tile || first.mode != MODE_DIF || !last.tile || last.mode != MODE_DIT || !last.canon) return FASTECC_E_UNSUPPORTED;
    FusedEnds fe;
    fe.first_in2 = parity;
    fe.first_side = fin;
    fe.last_side = gout_all;
    fe.last_out = data_out;
    fe.last_out2 = parity_out;
    return run_passes(p, p->enc, data, work, p->tw_inv, p->tw_fwd, true, st, hooks, 0, 0, &fe);
}

int create_transform_mid(Path** out, int n, uint64_t elems, int factor, int force_mid, char* detail, size_t cap)
{
    *out = nullptr;
    if (n < 1 || n > MAX_LOG2_K || elems == 0 || elems > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
    Path* p = new (std::nothrow) Path();
    if (!p) return FASTECC_E_NOMEM;
    p->n = n;
    p->N = 1ull << n;
    p->elems = elems;
    p->force_mid = force_mid;
    {
        const gf61::Elem w16 = gf61::h_root(16);
        for (int i = 0; i < 4; i++) {
            const gf61::Elem w = gf61::h_pow(w16, 2 * i + 1);
            p->sr.w16[i][0] = w.re;
            p->sr.w16[i][1] = w.im;
        }
    }
    build_plans(p);

    // per-block factors w_2N^i / N (RS.cpp:51-54) — or i / N, the decoder's — stored by position: position q holds coefficient bitrev(q)
    int rc = upload_tables(p, detail, cap);
    if (rc == FASTECC_OK) {
        const gf61::Elem w2N = gf61::h_root(2 * p->N), c = gf61::h_inv(gf61::Elem{p->N % gf61::P, 0});
        hipError_t e = hipMalloc((void**)&p->dscale, 2 * p->N * 8);
        if (e == hipSuccess) {
            const uint64_t half = gf61::h_inv(gf61::Elem{2, 0}).re;  // FACTOR_SPLIT: (2 m + N) / 2N = m / N + 1 / 2
            hipLaunchKernelGGL(k_block_factors, dim3((unsigned)((p->N + 255) / 256)), dim3(256), 0, nullptr, p->dscale, c.re, c.im, w2N.re, w2N.im, n,
                               factor == FACTOR_INDEX || factor == FACTOR_SPLIT, factor == FACTOR_SPLIT ? half : 0ull);
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = fail(detail, cap, e, "gf61 block factors");
    }
    if (rc == FASTECC_OK) {
        const hipError_t e = hipStreamSynchronize(nullptr);  // the tables are complete before any stream may use the path
        if (e != hipSuccess) rc = fail(detail, cap, e, "gf61 tables");
    }
    if (rc != FASTECC_OK) {
        destroy(p);
        return rc;
    }
    *out = p;
    return FASTECC_OK;
}

void destroy(Path* p)
{
    if (!p) return;
    if (p->tw_fwd) (void)hipFree(p->tw_fwd);
    if (p->tw_inv) (void)hipFree(p->tw_inv);
    if (p->tw_ntt_fwd) (void)hipFree(p->tw_ntt_fwd);
    if (p->tw_ntt_inv) (void)hipFree(p->tw_ntt_inv);
    if (p->dscale) (void)hipFree(p->dscale);
    delete p;
}

int encode_columns(Path* p, const uint64_t* data, uint64_t* parity, uint64_t col0, uint64_t width, hipStream_t st, const LaunchHooks* hooks)
{
    if (width == 0 || col0 + width > p->elems) return FASTECC_E_INVAL;
    return run_passes(p, p->enc, data, parity, p->tw_inv, p->tw_fwd, true, st, hooks, col0, width);
}

int encode(Path* p, const uint64_t* data, uint64_t* parity, hipStream_t st, const LaunchHooks* hooks)
{
    // inverse roots on the way down (interpolate), forward roots on the way up (evaluate) — RS.cpp:41,63
    return run_passes(p, p->enc, data, parity, p->tw_inv, p->tw_fwd, true, st, hooks);
}

// n = 4k / 8k: the DIF passes once (data -> work, k blocks; with no DIF pass in the plan MID reads the data itself), then MID and the DIT passes
// once per coset with that coset's factors, coset t writing parity blocks [t k, (t + 1) k).
int encode_cosets(Path* p, const uint64_t* data, uint64_t* parity, uint64_t* work, hipStream_t st, const LaunchHooks* hooks, uint32_t coset_mask)
{
    size_t mid = 0;
    while (mid < p->enc.size() && p->enc[mid].mode != MODE_MID) mid++;
    if (mid >= p->enc.size()) return FASTECC_E_UNSUPPORTED;
    const std::vector<Pass> head(p->enc.begin(), p->enc.begin() + (long)mid), tail(p->enc.begin() + (long)mid, p->enc.end());
    const uint64_t* src = data;
    if (!head.empty()) {
        if (!work) return FASTECC_E_INVAL;
        const int rc = run_passes(p, head, data, work, p->tw_inv, p->tw_fwd, true, st, hooks);
        if (rc != FASTECC_OK) return rc;
        src = work;
    }
    for (int t = 0; t < p->cosets; t++) {
        if (!((coset_mask >> t) & 1u)) continue;  // (fastecc_repair: cosets that have lost no block are not computed again)
        const int rc = run_passes(p, tail, src, parity + (size_t)t * p->N * p->elems * 2, p->tw_inv, p->tw_fwd, true, st, hooks, 0, 0, nullptr,
                                  p->dscale + (size_t)t * 2 * p->N);
        if (rc != FASTECC_OK) return rc;
    }
    return FASTECC_OK;
}
bool encode_cosets_needs_work(const Path* p) { return p && !p->enc.empty() && p->enc[0].mode != MODE_MID; }

// The stand-alone transform's DIF passes WITHOUT the closing block permutation: block q of the result holds coefficient bitrev(q) — the
// order MID's first half leaves, which is what the split decoder's addend is read in.
int dif_only(Path* p, uint64_t* data, bool inverse, hipStream_t st, const LaunchHooks* hooks)
{
    const uint64_t* tw = inverse ? (p->tw_ntt_inv ? p->tw_ntt_inv : p->tw_inv) : (p->tw_ntt_fwd ? p->tw_ntt_fwd : p->tw_fwd);
    return run_passes(p, p->fwd, data, data, tw, tw, inverse, st, hooks);
}
// ... from `in` into another stripe `out` (the first pass reads `in`, the others run in place on `out`)
int dif_only_to(Path* p, const uint64_t* in, uint64_t* out, bool inverse, hipStream_t st, const LaunchHooks* hooks)
{
    const uint64_t* tw = inverse ? (p->tw_ntt_inv ? p->tw_ntt_inv : p->tw_inv) : (p->tw_ntt_fwd ? p->tw_ntt_fwd : p->tw_fwd);
    return run_passes(p, p->fwd, in, out, tw, tw, inverse, st, hooks);
}

// The split decoder's data chain on a FACTOR_SPLIT path of size k (gf61_decode.hip): DIF passes over data[i] * rows_factor[i * rows_stride] (into
// `work`, then in place), MID with + addend[p >> shift] * addend_factor[p] between its halves, DIT passes, the last of which stores only the rows
// with gout[i] != 0, times gout[i], into data_out.  FASTECC_E_UNSUPPORTED unless the plan is [DIF tile .. MID tile .. canonical DIT tile].
bool split_decode_supported(const Path* p)
{
    if (!p || p->enc.size() < 3) return false;
    const Pass &first = p->enc.front(), &last = p->enc.back();
    bool mid_ok = false;
    for (size_t i = 1; i + 1 < p->enc.size(); i++)
        if (p->enc[i].mode == MODE_MID) mid_ok = p->enc[i].tile && !p->enc[i].canon;
    return first.tile && first.mode == MODE_DIF && last.tile && last.mode == MODE_DIT && last.canon && mid_ok;
}
int split_decode(Path* p, const uint64_t* data, const uint64_t* rows_factor, uint32_t rows_stride, const uint64_t* addend, int addend_shift,
                 const uint64_t* addend_factor, uint64_t* work, const uint64_t* gout, uint64_t* data_out, hipStream_t st, const LaunchHooks* hooks,
                 uint64_t* keep)
{
    if (!split_decode_supported(p) || rows_stride == 0) return FASTECC_E_UNSUPPORTED;
    FusedEnds f;
    f.mid_keep = keep;
    f.first_side = rows_factor;
    f.first_rows_stride = rows_stride;
    f.mid_addend = addend;
    f.mid_addend_factor = addend_factor;
    f.mid_shift = addend_shift;
    f.last_side = gout;
    f.last_out = data_out;
    return run_passes(p, p->enc, data, work, p->tw_inv, p->tw_fwd, true, st, hooks, 0, 0, &f);
}
// fastecc_repair's second chain: x p'(x) at the ODD positions (the parity blocks) is the k-point transform of h[m] = -1/2 w^m q~[m] + (2m+k)/2k r~[m]
// (decode.hip's header) — MID's second half alone on the q~ that split_decode kept, the two factor tables exchanged (data_factor: -1/2 w^m by
// position; the path's own table now scales the addend), then the DIT passes, the last of which stores parity block j times gout_par[j] where
// that is not zero.  work: a k-block stripe (split_decode's may be reused once it has finished).
int split_repair_parity(Path* p, const uint64_t* keep, const uint64_t* data_factor, const uint64_t* addend, int addend_shift, uint64_t* work,
                        const uint64_t* gout_par, uint64_t* parity_out, hipStream_t st, const LaunchHooks* hooks)
{
    if (!split_decode_supported(p)) return FASTECC_E_UNSUPPORTED;
    size_t mid = 0;
    while (mid < p->enc.size() && p->enc[mid].mode != MODE_MID) mid++;
    const std::vector<Pass> tail(p->enc.begin() + (long)mid, p->enc.end());
    FusedEnds f;
    f.mid_addend = addend;
    f.mid_addend_factor = p->dscale;
    f.mid_shift = addend_shift;
    f.mid_up = true;
    f.last_side = gout_par;
    f.last_out = parity_out;
    return run_passes(p, tail, keep, work, p->tw_inv, p->tw_fwd, true, st, hooks, 0, 0, &f, data_factor);
}
// addend_factor table of the split decoder for a path of size 2^n: n-bit position q -> -1/2 w_2N^(-bitrev(q)) (2^n elements, device memory of the caller)
int split_addend_factors(uint64_t* table, int n, hipStream_t st, bool forward)
{
    const uint64_t N = 1ull << n;
    const gf61::Elem wi = forward ? gf61::h_root(2 * N) : gf61::h_inv(gf61::h_root(2 * N));  // forward: -1/2 w^(+bitrev(q)), the parity chain's factor of q~
    const gf61::Elem half = gf61::h_inv(gf61::Elem{2, 0});
    const gf61::Elem neg_half{gf61::h_subp(0, half.re), 0};
    hipLaunchKernelGGL(k_split_addend_factors, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, table, neg_half.re, neg_half.im, wi.re, wi.im, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? FASTECC_OK : fail(nullptr, 0, e, "p61 split addend factors");
}

int ntt(Path* p, uint64_t* data, bool inverse, hipStream_t st, const LaunchHooks* hooks)
{
    const uint64_t* tw = inverse ? (p->tw_ntt_inv ? p->tw_ntt_inv : p->tw_inv) : (p->tw_ntt_fwd ? p->tw_ntt_fwd : p->tw_fwd);
    const int rc = run_passes(p, p->fwd, data, data, tw, tw, inverse, st, hooks);
    if (rc != FASTECC_OK) return rc;
    if (p->n >= 2) {
        const uint32_t col_chunks = (uint32_t)((p->elems + 63) / 64);
        const uint64_t items = p->N * col_chunks;
        const uint64_t blocks = (items + 3) / 4;
        if (blocks > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
        Scope sc(hooks, st, "p61_bitrev_rows", 2ull * p->N * p->elems * 16ull);
        hipLaunchKernelGGL(p61_bitrev_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, data, (uint32_t)p->elems, p->n, col_chunks, items);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(nullptr, 0, e, "p61_bitrev_rows");
    }
    return FASTECC_OK;
}

int count_out_of_range(Path* p, const uint64_t* data, unsigned long long* counter, hipStream_t st)
{
    const uint64_t words = 2 * p->N * p->elems;
    const unsigned blocks = (unsigned)std::min<uint64_t>((words + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(p61_count_out_of_range_kernel, dim3(blocks), dim3(256), 0, st, data, words, counter);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, 0, e, "p61_count_out_of_range");
    return FASTECC_OK;
}

int set_plan(Path* p, int plan, char* detail, size_t cap)
{
    // 0 = default; 1..4 = register passes only, that many levels per pass; 10 + L = LDS tiles (64 KiB exchange buffer)
    // with register passes of at most L levels where no tile shape fits; 20 + L = the same with a 128 KiB buffer
    const int levels = plan == 0 ? DEFAULT_LEVELS : plan % 10, kind = plan / 10;
    if (plan < 0 || kind > 2 || levels < 1 || levels > 4) return FASTECC_E_INVAL;
    p->levels = levels;
    p->tiles = plan == 0 || kind >= 1;
    p->split = kind == 2 ? 1 : 2;
    build_plans(p);
    int rc = upload_tables(p, detail, cap);
    if (rc == FASTECC_OK) {
        const hipError_t e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) rc = fail(detail, cap, e, "gf61 tables");
    }
    return rc;
}

const char* plan_string(const Path* p) { return p->text.c_str(); }

}  // namespace p61
}  // namespace fastecc
