// microbench.hip — design-by-measurement probes for the encode path on gfx950 (not part of the library).
//
//   valu   : issue rate of the integer instructions the GF(p) butterfly is made of
//   bfly   : whole-butterfly rate for several modular-multiply formulations (all bit-exact)
//   copy   : HBM rate of the block-strided access patterns the pass kernels use
//            (row segment width x rows per wave x row stride)
//
// Prints one JSON object per line.  Usage: microbench [valu] [bfly] [copy]   (default: all)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "gf.hpp"

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

static float time_ms(hipStream_t st, int reps, const std::function<void()>& fn)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    fn();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return ms / reps;
}

// ------------------------------------------------------------------------------------------------
// VALU issue-rate probes: 8 independent chains per lane, ITER iterations, each iteration 8 instrs.
// ------------------------------------------------------------------------------------------------
enum { OP_ADD, OP_MUL_LO, OP_MUL_HI, OP_MAD64, OP_MUL24, OP_LSHL_ADD, OP_SUBCO_CND, OP_CNDMASK_ONLY, OP_COUNT };
static const char* OP_NAME[] = {"v_add_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24",
                                "v_lshl_add_u32", "v_sub_co+v_cndmask(2 instr)", "v_cndmask_b32"};

template <int OP>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t* out, int iters, uint32_t k)
{
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 8 + i + k;
    uint64_t y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = x[i];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (OP == OP_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
            if constexpr (OP == OP_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
            if constexpr (OP == OP_MUL_HI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(k));
            if constexpr (OP == OP_MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y[i]) : "v"(x[i]), "v"(k) : "vcc");
            if constexpr (OP == OP_MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(k));
            if constexpr (OP == OP_LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 20, %1" : "+v"(x[i]) : "v"(k));
            if constexpr (OP == OP_SUBCO_CND)
                asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(k) : "vcc");
            if constexpr (OP == OP_CNDMASK_ONLY) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(k) : "vcc");
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= x[i] ^ (uint32_t)y[i] ^ (uint32_t)(y[i] >> 32);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int OP>
static void run_valu(uint32_t* d_out, int blocks)
{
    const int iters = 4096;
    hipStream_t st = nullptr;
    float ms = time_ms(st, 5, [&] { hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(256), 0, st, d_out, iters, 3u); });
    const double lane_ops = (double)blocks * 256 * iters * 8;  // per "op" (a pair counts once)
    printf("{\"probe\":\"valu\",\"op\":\"%s\",\"ms\":%.4f,\"Tlaneops_per_s\":%.2f}\n", OP_NAME[OP], ms, lane_ops / ms / 1e9);
    fflush(stdout);
}

// ------------------------------------------------------------------------------------------------
// Butterfly-rate probes.  Each variant computes the same DIT butterfly (a,b) -> (a+b*w, a-b*w).
// ------------------------------------------------------------------------------------------------
namespace v {
using gf::P;

// A: as in gf.hpp today
__device__ __forceinline__ uint32_t mont_a(uint32_t x, uint32_t wm) { return gf::mul_mont(x, wm); }

// B: borrow taken from the 32-bit subtract itself (no 64-bit compare)
__device__ __forceinline__ uint32_t mont_b(uint32_t x, uint32_t wm)
{
    const uint32_t lo = x * wm, hi = __umulhi(x, wm);
    const uint32_t m = lo + (lo << 20);
    const uint32_t q = __umulhi(m, P);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(hi, q, &r);
    return borrow ? r + P : r;
}

// C: hi(m*p) by shifts: m*p = m*2^32 - (m*2^20 - m)  ->  hi = m - (m>>12) - ((m<<20) > m)
__device__ __forceinline__ uint32_t mont_c(uint32_t x, uint32_t wm)
{
    const uint64_t t = (uint64_t)x * wm;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t m = lo + (lo << 20);
    const uint32_t q = m - (m >> 12) - ((m << 20) > m ? 1u : 0u);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(hi, q, &r);
    return borrow ? r + P : r;
}

// D: second multiply-add folds the reduction: hi64(t + m*(2^20-1)) - m
__device__ __forceinline__ uint32_t mont_d(uint32_t x, uint32_t wm)
{
    const uint64_t t = (uint64_t)x * wm;
    const uint32_t lo = (uint32_t)t;
    const uint32_t m = lo + (lo << 20);
    const uint64_t u = t + (uint64_t)m * 0xFFFFFu;  // cannot overflow: see DESIGN.md
    const uint32_t h = (uint32_t)(u >> 32);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(h, m, &r);
    return borrow ? r + P : r;
}

// E: Barrett as in the reference (general operands)
__device__ __forceinline__ uint32_t mul_e(uint32_t x, uint32_t w) { return gf::mul(x, w); }

// F: like D but the second multiply is a plain high multiply: r = hi(t) - hi(m*p)
__device__ __forceinline__ uint32_t mont_f(uint32_t x, uint32_t wm)
{
    const uint64_t t = (uint64_t)x * wm;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t m = lo + (lo << 20);
    const uint32_t q = __umulhi(m, P);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(hi, q, &r);
    return borrow ? r + P : r;
}

// G: returns p - (x*w) (or 0 when the product is 0): the negated product, same cost as F
__device__ __forceinline__ uint32_t mont_neg(uint32_t x, uint32_t wm)
{
    const uint64_t t = (uint64_t)x * wm;
    const uint32_t lo = (uint32_t)t, hi = (uint32_t)(t >> 32);
    const uint32_t m = lo + (lo << 20);
    const uint32_t q = __umulhi(m, P);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(q, hi, &r);
    return borrow ? r + P : r;
}

// H: two-word twiddle (wm, wq = wm * p^-1 mod 2^32): no 64-bit product, three 32-bit multiplies
__device__ __forceinline__ uint32_t mont_h(uint32_t x, uint32_t wm, uint32_t wq)
{
    const uint32_t hi = __umulhi(x, wm);
    const uint32_t m = x * wq;
    const uint32_t q = __umulhi(m, P);
    uint32_t r;
    const bool borrow = __builtin_usub_overflow(hi, q, &r);
    return borrow ? r + P : r;
}

__device__ __forceinline__ uint32_t sub1(uint32_t a, uint32_t b)
{
    uint32_t d;
    const bool borrow = __builtin_usub_overflow(a, b, &d);
    return borrow ? d + P : d;
}

__device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b)
{
    // carry-based: s = a+b; u = s - p (mod 2^32); take u when a+b overflowed or s >= p
    uint32_t s, u;
    const bool c1 = __builtin_uadd_overflow(a, b, &s);
    const bool c2 = __builtin_uadd_overflow(s, 0xFFFFFu, &u);
    return (c1 | c2) ? u : s;
}
// Lazy representatives (round 2): any uint32 congruent to the value.  With the other operand canonical (a fresh product)
// one carry correction is exact: a + t - 2^32 = a + t - p - (2^20 - 1), and the sum stays below 2^32.
__device__ __forceinline__ uint32_t add_lazy(uint32_t a, uint32_t t)
{
    uint32_t s;
    const bool c = __builtin_uadd_overflow(a, t, &s);
    return s + (c ? 0xFFFFFu : 0u);
}
__device__ __forceinline__ uint32_t sub_lazy(uint32_t a, uint32_t t)
{
    uint32_t d;
    const bool b = __builtin_usub_overflow(a, t, &d);
    return d - (b ? 0xFFFFFu : 0u);
}
__device__ __forceinline__ uint32_t canon(uint32_t x) { return x >= gf::P ? x - gf::P : x; }
}  // namespace v

template <int VAR>
__global__ __launch_bounds__(256) void bfly_kernel(uint32_t* out, int iters, uint32_t w0)
{
    uint32_t a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = (threadIdx.x * 2654435761u + i * 40503u) % gf::P;
        b[i] = (threadIdx.x * 40503u + i * 2654435761u + 7u) % gf::P;
    }
    uint32_t w = w0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t t;
            if constexpr (VAR == 0) t = v::mont_a(b[i], w);
            if constexpr (VAR == 1) t = v::mont_b(b[i], w);
            if constexpr (VAR == 2) t = v::mont_c(b[i], w);
            if constexpr (VAR == 3) t = v::mont_d(b[i], w);
            if constexpr (VAR == 4) t = v::mul_e(b[i], w);
            if constexpr (VAR == 5) t = v::mont_b(b[i], w);
            if constexpr (VAR == 6) t = v::mont_f(b[i], w);
            if constexpr (VAR == 7) t = v::mont_f(b[i], w);
            if constexpr (VAR == 8) t = v::mont_neg(b[i], w);
            if constexpr (VAR == 9) t = v::mont_d(b[i], w);
            if constexpr (VAR == 10) t = v::mont_h(b[i], w, w * 0x00100001u);
            if constexpr (VAR == 11) t = v::mont_f(b[i], w);
            const uint32_t x = a[i];
            if constexpr (VAR == 5 || VAR == 6 || VAR == 10) {
                a[i] = v::add2(x, t);
                b[i] = v::sub1(x, t);
            } else if constexpr (VAR == 7 || VAR == 9) {
                a[i] = v::sub1(x, gf::P - t);  // add as subtract of the negation
                b[i] = v::sub1(x, t);
            } else if constexpr (VAR == 11) {
                a[i] = v::add_lazy(x, t);  // lazy outputs: canonicalised once at the end
                b[i] = v::sub_lazy(x, t);
            } else if constexpr (VAR == 8) {
                a[i] = v::sub1(x, t);          // t holds p - b*w: x + b*w
                b[i] = v::sub1(x, gf::P - t);  // x - b*w   (t == 0 -> p - 0 = p -> x - p wraps -> +p -> x)
            } else {
                a[i] = gf::add(x, t);
                b[i] = gf::sub(x, t);
            }
        }
        w = w * 3u + 1u;  // keep the twiddle scalar but changing (SALU)
        w = w >= gf::P ? w - gf::P : w;
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= VAR == 11 ? (v::canon(a[i]) ^ v::canon(b[i])) : (a[i] ^ b[i]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

static const char* BFLY_NAME[] = {"mont:mad64+mulhi (gf.hpp)", "mont:mullo+mulhi+mulhi,usub_overflow", "mont:mad64+shift-form hi(m*p)",
                                  "mont:2x mad64", "barrett (reference form)", "mont B + carry-form add",
                                  "mont F(mad64+mulhi) + carry-form add", "mont F + add-as-sub(p-t)", "mont NEG + sub-only",
                                  "mont D(2x mad64) + add-as-sub(p-t)", "mont H(two-word twiddle, 3 mul32) + carry-form add",
                                  "mont F + LAZY add/sub (any uint32 representative, one carry correction, canonical at the end)"};

template <int VAR>
static void run_bfly(uint32_t* d_out, int blocks, std::vector<uint32_t>* first)
{
    const int iters = 2048;
    hipStream_t st = nullptr;
    float ms = time_ms(st, 5, [&] { hipLaunchKernelGGL(bfly_kernel<VAR>, dim3(blocks), dim3(256), 0, st, d_out, iters, 12345u); });
    std::vector<uint32_t> h(256);
    CK(hipMemcpy(h.data(), d_out, 1024, hipMemcpyDeviceToHost));
    bool same = true;
    if (VAR == 4) same = true;  // Barrett multiplies by w, not w/2^32: different values by design
    else if (first->empty()) *first = h;
    else same = (h == *first);
    const double bf = (double)blocks * 256 * iters * 8;
    printf("{\"probe\":\"bfly\",\"variant\":\"%s\",\"ms\":%.4f,\"Gbfly_per_s\":%.1f,\"agrees\":%s}\n", BFLY_NAME[VAR], ms, bf / ms / 1e6,
           same ? "true" : "false");
    fflush(stdout);
}

// Radix-4 against two radix-2 levels (the reference's NTT4 codelet, ntt.cpp:50-62, is 2 x NTT2, one multiply by
// GF_Root(4), 2 x NTT2).  Four registers (x0..x3) at distance "1" and "2", DIF, outer twiddle w:
//   radix 2 x 2:  u0 = x0+x2, u2 = (x0-x2) w,    u1 = x1+x3, u3 = (x1-x3) (w w4);   y0 = u0+u1, y1 = (u0-u1) w^2, y2 = u2+u3, y3 = (u2-u3) w^2
//   radix 4    :  t0 = x0+x2, t2 = x0-x2, t1 = x1+x3, t3 = (x1-x3) w4;  y0 = t0+t1, y1 = (t0-t1) w^2, y2 = (t2+t3) w, y3 = (t2-t3) w^3
// Same values (tested), and the same 4 products + 8 additions per 4 words: w4 is a general element of this field (ord(2) = 2^18 * 117:
// no power of two is a small root of unity), so the radix-4 form saves nothing but one scalar twiddle.
template <int RADIX>
__global__ __launch_bounds__(256) void radix_kernel(uint32_t* out, int iters, uint32_t w0, uint32_t w4)
{
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = (threadIdx.x * 2654435761u + i * 40503u) % gf::P;
    uint32_t w = w0;
    for (int it = 0; it < iters; it++) {
        const uint32_t w2 = w * 5u + 3u, w3 = w * 7u + 1u, ww4 = w * 11u + 5u;  // stand-ins for w^2, w^3, w w4: scalar, changing, < 2^32
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            const uint32_t x0 = x[g], x1 = x[g + 1], x2 = x[g + 2], x3 = x[g + 3];
            if constexpr (RADIX == 2) {
                const uint32_t u0 = gf::add(x0, x2), u2 = gf::mul_mont(gf::sub(x0, x2), w);
                const uint32_t u1 = gf::add(x1, x3), u3 = gf::mul_mont(gf::sub(x1, x3), ww4);
                x[g] = gf::add(u0, u1);
                x[g + 1] = gf::mul_mont(gf::sub(u0, u1), w2);
                x[g + 2] = gf::add(u2, u3);
                x[g + 3] = gf::mul_mont(gf::sub(u2, u3), w2);
            } else {
                const uint32_t t0 = gf::add(x0, x2), t2 = gf::sub(x0, x2);
                const uint32_t t1 = gf::add(x1, x3), t3 = gf::mul_mont(gf::sub(x1, x3), w4);
                x[g] = gf::add(t0, t1);
                x[g + 1] = gf::mul_mont(gf::sub(t0, t1), w2);
                x[g + 2] = gf::mul_mont(gf::add(t2, t3), w);
                x[g + 3] = gf::mul_mont(gf::sub(t2, t3), w3);
            }
        }
        w = w * 3u + 1u;
        w = w >= gf::P ? w - gf::P : w;
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc ^= x[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// exactness of the identity above with true powers (one launch, real twiddles): both forms on the same inputs
__global__ void radix_check_kernel(uint32_t* out, uint32_t w, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t ww4)
{
    const uint32_t x0 = (threadIdx.x * 2654435761u) % gf::P, x1 = (threadIdx.x * 40503u + 1u) % gf::P, x2 = (threadIdx.x * 97u + 5u) % gf::P, x3 = gf::P - 1u - threadIdx.x;
    const uint32_t u0 = gf::add(x0, x2), u2 = gf::mul_mont(gf::sub(x0, x2), w), u1 = gf::add(x1, x3), u3 = gf::mul_mont(gf::sub(x1, x3), ww4);
    const uint32_t a0 = gf::add(u0, u1), a1 = gf::mul_mont(gf::sub(u0, u1), w2), a2 = gf::add(u2, u3), a3 = gf::mul_mont(gf::sub(u2, u3), w2);
    const uint32_t t0 = gf::add(x0, x2), t2 = gf::sub(x0, x2), t1 = gf::add(x1, x3), t3 = gf::mul_mont(gf::sub(x1, x3), w4);
    const uint32_t b0 = gf::add(t0, t1), b1 = gf::mul_mont(gf::sub(t0, t1), w2), b2 = gf::mul_mont(gf::add(t2, t3), w), b3 = gf::mul_mont(gf::sub(t2, t3), w3);
    out[threadIdx.x] = (a0 == b0 && a1 == b1 && a2 == b2 && a3 == b3) ? 1u : 0u;
}

template <int RADIX>
static void run_radix(uint32_t* d_out, int blocks)
{
    const int iters = 2048;
    hipStream_t st = nullptr;
    const uint32_t w4m = gf::h_to_mont(gf::h_root(4));
    float ms = time_ms(st, 5, [&] { hipLaunchKernelGGL(radix_kernel<RADIX>, dim3(blocks), dim3(256), 0, st, d_out, iters, 12345u, w4m); });
    const double bf = (double)blocks * 256 * iters * 16;  // 4 butterfly-equivalents per 4 words and 2 levels
    printf("{\"probe\":\"radix\",\"variant\":\"%s\",\"ms\":%.4f,\"Gbfly_per_s\":%.1f}\n", RADIX == 2 ? "two radix-2 levels" : "one radix-4 level (NTT4 form)", ms, bf / ms / 1e6);
    fflush(stdout);
}
static void run_radix_check(uint32_t* d_out)
{
    const uint32_t w = gf::h_root(1u << 12), w4 = gf::h_root(4);
    const uint32_t w2 = gf::h_mul(w, w), w3 = gf::h_mul(w2, w), ww4 = gf::h_mul(w, w4);
    hipLaunchKernelGGL(radix_check_kernel, dim3(1), dim3(256), 0, nullptr, d_out, gf::h_to_mont(w), gf::h_to_mont(w2), gf::h_to_mont(w3), gf::h_to_mont(w4), gf::h_to_mont(ww4));
    std::vector<uint32_t> h(256);
    CK(hipMemcpy(h.data(), d_out, 1024, hipMemcpyDeviceToHost));
    bool ok = true;
    for (uint32_t v : h) ok = ok && v == 1u;
    printf("{\"probe\":\"radix\",\"check\":\"radix-4 (w4 = 0x%08X) == two radix-2 levels on 256 random quadruples\",\"agrees\":%s}\n", w4, ok ? "true" : "false");
}

// The same butterfly loops run for seconds instead of milliseconds: the board's power cap (1400 W) then sets the clock, and variants
// that issue the same number of instructions may differ in what they sustain.  Power and clock are read with rocm-smi while a queue
// of launches keeps the GPU busy.
template <int VAR>
static void run_bfly_sustained(uint32_t* d_out, int blocks, double seconds)
{
    const int iters = 2048;
    hipStream_t st = nullptr;
    auto launch = [&](int n) {
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(bfly_kernel<VAR>, dim3(blocks), dim3(256), 0, st, d_out, iters, 12345u);
    };
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    while (elapsed() < seconds / 2) {  // heat up
        launch(50);
        CK(hipStreamSynchronize(st));
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    long launches = 0;
    CK(hipEventRecord(a, st));
    char smi[256] = "";
    bool sampled = false;
    while (elapsed() < seconds) {
        launch(200);
        launches += 200;
        if (!sampled) {  // the queue above keeps the GPU busy for ~0.5 s while rocm-smi runs
            FILE* f = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk' | sed -e 's/.*: //' | tr '\\n' ' '", "r");
            if (f) {
                if (!fgets(smi, sizeof smi, f)) smi[0] = 0;
                pclose(f);
            }
            sampled = true;
        }
        CK(hipStreamSynchronize(st));
    }
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    for (char* c = smi; *c; ++c)
        if (*c == '"' || *c == '\n') *c = ' ';
    const double bf = (double)blocks * 256 * iters * 8 * launches;
    printf("{\"probe\":\"bfly_sustained\",\"variant\":\"%s\",\"seconds\":%.1f,\"Gbfly_per_s\":%.1f,\"rocm_smi\":\"%s\"}\n", BFLY_NAME[VAR], ms / 1e3,
           bf / ms / 1e6, smi);
    fflush(stdout);
}

// ------------------------------------------------------------------------------------------------
// Access-pattern probes: a wave copies ROWS row segments of 64*V/RPL... see below.
//   V    words per lane (1,2,4)
//   LPR  lanes per row segment (64 = a wave spans one row; 32/16/8 = 2/4/8 rows per load instruction)
//   K    load instructions per lane (rows per wave = K * 64/LPR)
//   s    log2 of the row stride inside a wave's group (like a pass with lowest stride 2^s)
// The matrix is N x S words; every element is read once and written once per launch.
// ------------------------------------------------------------------------------------------------
template <int V, int LPR, int K>
__global__ __launch_bounds__(256) void copy_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t S, int n, int s,
                                                   uint32_t col_chunks, uint64_t items)
{
    constexpr int RPI = 64 / LPR;    // rows per load instruction
    constexpr int ROWS = K * RPI;    // rows per wave
    constexpr int W = LPR * V;       // words per row segment
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t g = (uint32_t)(item / col_chunks);
    const uint32_t lo = g & ((1u << s) - 1u), hi = g >> s;
    const uint32_t base = hi * (ROWS << s) + lo;
    const uint32_t col = cc * W + (lane % LPR) * V;
    const uint32_t sub = lane / LPR;
    uint32_t x[K][V];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const size_t row = base + ((size_t)(k * RPI + sub) << s);
        const uint32_t* p = in + row * S + col;
        if constexpr (V == 4) { uint4 t = *(const uint4*)p; x[k][0] = t.x; x[k][1] = t.y; x[k][2] = t.z; x[k][3] = t.w; }
        else if constexpr (V == 2) { uint2 t = *(const uint2*)p; x[k][0] = t.x; x[k][1] = t.y; }
        else x[k][0] = *p;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const size_t row = base + ((size_t)(k * RPI + sub) << s);
        uint32_t* p = out + row * S + col;
#pragma unroll
        for (int v2 = 0; v2 < V; v2++) x[k][v2] += 1u;
        if constexpr (V == 4) *(uint4*)p = make_uint4(x[k][0], x[k][1], x[k][2], x[k][3]);
        else if constexpr (V == 2) *(uint2*)p = make_uint2(x[k][0], x[k][1]);
        else *p = x[k][0];
    }
}

template <int V, int LPR, int K>
static void run_copy(const uint32_t* in, uint32_t* out, uint32_t S, int n, int s)
{
    constexpr int ROWS = K * (64 / LPR);
    constexpr int W = LPR * V;
    const uint32_t col_chunks = S / W;
    if (((uint64_t)ROWS << s) > (1ull << n)) return;  // group would not fit in the matrix
    const uint64_t items = ((uint64_t)col_chunks << n) / ROWS;
    const unsigned blocks = (unsigned)((items + 3) / 4);
    hipStream_t st = nullptr;
    float ms = time_ms(st, 5, [&] {
        hipLaunchKernelGGL((copy_kernel<V, LPR, K>), dim3(blocks), dim3(256), 0, st, in, out, S, n, s, col_chunks, items);
    });
    const double bytes = 2.0 * 4.0 * S * (double)(1ull << n);
    printf("{\"probe\":\"copy\",\"seg_bytes\":%d,\"V\":%d,\"rows_per_wave\":%d,\"log2_stride\":%d,\"inplace\":%s,\"ms\":%.4f,\"GBps\":%.0f}\n", W * 4, V,
           ROWS, s, in == out ? "true" : "false", ms, bytes / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    bool do_valu = argc == 1, do_bfly = argc == 1, do_copy = argc == 1, do_radix = argc == 1;
    double sustain = 0;
    for (int i = 1; i < argc; i++) {
        if (!strncmp(argv[i], "sustain=", 8)) sustain = atof(argv[i] + 8);
        if (!strcmp(argv[i], "valu")) do_valu = true;
        if (!strcmp(argv[i], "bfly")) do_bfly = true;
        if (!strcmp(argv[i], "radix")) do_radix = true;
        if (!strcmp(argv[i], "copy")) do_copy = true;
    }
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"probe\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_mhz\":%d,\"mem_clock_mhz\":%d,\"bus_bits\":%d,\"l2_bytes\":%d,\"hbm_gb\":%.0f}\n",
           prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.memoryBusWidth, prop.l2CacheSize,
           prop.totalGlobalMem / 1e9);
    const int blocks = prop.multiProcessorCount * 8;  // 8 x 256 threads = 32 waves per CU

    uint32_t* d_out;
    CK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    if (do_valu) {
        run_valu<OP_ADD>(d_out, blocks);
        run_valu<OP_MUL_LO>(d_out, blocks);
        run_valu<OP_MUL_HI>(d_out, blocks);
        run_valu<OP_MAD64>(d_out, blocks);
        run_valu<OP_MUL24>(d_out, blocks);
        run_valu<OP_LSHL_ADD>(d_out, blocks);
        run_valu<OP_SUBCO_CND>(d_out, blocks);
        run_valu<OP_CNDMASK_ONLY>(d_out, blocks);
    }
    if (sustain > 0) {
        run_bfly_sustained<0>(d_out, blocks, sustain);
        run_bfly_sustained<1>(d_out, blocks, sustain);
        run_bfly_sustained<3>(d_out, blocks, sustain);
        run_bfly_sustained<5>(d_out, blocks, sustain);
        run_bfly_sustained<6>(d_out, blocks, sustain);
        run_bfly_sustained<7>(d_out, blocks, sustain);
        run_bfly_sustained<8>(d_out, blocks, sustain);
        run_bfly_sustained<10>(d_out, blocks, sustain);
        run_bfly_sustained<11>(d_out, blocks, sustain);
    }
    if (do_radix) {
        run_radix_check(d_out);
        for (int r = 0; r < 2; r++) {
            run_radix<2>(d_out, blocks);
            run_radix<4>(d_out, blocks);
        }
    }
    if (do_bfly) {
        std::vector<uint32_t> first;
        run_bfly<0>(d_out, blocks, &first);
        run_bfly<1>(d_out, blocks, &first);
        run_bfly<2>(d_out, blocks, &first);
        run_bfly<3>(d_out, blocks, &first);
        run_bfly<4>(d_out, blocks, &first);
        run_bfly<5>(d_out, blocks, &first);
        run_bfly<6>(d_out, blocks, &first);
        run_bfly<7>(d_out, blocks, &first);
        run_bfly<8>(d_out, blocks, &first);
        run_bfly<9>(d_out, blocks, &first);
        run_bfly<10>(d_out, blocks, &first);
        run_bfly<11>(d_out, blocks, &first);
        run_bfly<6>(d_out, blocks, &first);
    }
    if (do_copy) {
        const int n = 19;
        const uint32_t S = 1024;
        uint32_t *a, *b;
        CK(hipMalloc(&a, (size_t)S * 4 << n));
        CK(hipMalloc(&b, (size_t)S * 4 << n));
        CK(hipMemset(a, 1, (size_t)S * 4 << n));
        CK(hipMemset(b, 2, (size_t)S * 4 << n));
        // plain streaming reference: 1 KiB segments, contiguous rows, out of place and in place
        run_copy<4, 64, 16>(a, b, S, n, 0);
        run_copy<4, 64, 16>(a, a, S, n, 0);
        // register-pass patterns (wave = one row segment)
        for (int s : {4, 10, 15}) {
            run_copy<4, 64, 16>(a, a, S, n, s);
            run_copy<4, 64, 8>(a, a, S, n, s);
            run_copy<4, 64, 32>(a, a, S, n, s);
            run_copy<2, 64, 32>(a, a, S, n, s);
            run_copy<1, 64, 32>(a, a, S, n, s);
        }
        // sub-wave row segments (LDS-tile candidates): 128 B and 64 B segments
        for (int s : {0, 9, 10}) {
            run_copy<4, 8, 8>(a, a, S, n, s);    // 128 B segments, 64 rows per wave
            run_copy<4, 4, 8>(a, a, S, n, s);    // 64 B segments, 128 rows per wave
            run_copy<1, 32, 32>(a, a, S, n, s);  // 128 B segments, dword lanes, 64 rows per wave
            run_copy<1, 32, 16>(a, a, S, n, s);  // 128 B segments, dword lanes, 32 rows per wave
            run_copy<1, 16, 16>(a, a, S, n, s);  // 64 B segments, dword lanes, 64 rows per wave
            run_copy<1, 64, 32>(a, a, S, n, s);  // 256 B segments, dword lanes, 32 rows per wave
            run_copy<2, 32, 16>(a, a, S, n, s);  // 256 B segments, dwordx2 lanes, 2 rows per instr
            run_copy<4, 16, 8>(a, a, S, n, s);   // 256 B segments, dwordx4 lanes, 4 rows per instr
        }
        CK(hipFree(a));
        CK(hipFree(b));
    }
    CK(hipFree(d_out));
    return 0;
}
