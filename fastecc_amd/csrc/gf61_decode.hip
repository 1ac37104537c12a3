// gf61_decode.hip — erasure decoding over GF((2^61-1)^2): the scheme of decode.hip (README.md:102-119 "Fastest",
// RS.md:42-79; documented by the reference, implemented nowhere upstream) for the 64-bit field of BASELINE configs[4].
//
// Codeword of the (2k,k) code = f on the 2k-th roots of unity: position u <-> w^u (w = w_2k), data block i at u = 2i, parity
// block j at u = 2j+1.  With E the erased positions and l = prod_{e in E} (x - w^e):  p = f l has degree < 2k and known values
// everywhere (c[u] l(w^u) on survivors, 0 on erasures), and f(w^e) = [x p'(x)](w^e) / (w^e l'(w^e)).  Data-parallel part:
//   gather   work[u] = codeword[u] * l(w^u)                              (erased positions: zeros, nothing is read)
//   x p'(x)  one pass of the encoder's own pipeline one size up: inverse transform of size 2k, the block holding coefficient
//            m times m / 2k, forward transform  (gf61_kernels.hip with the factor m / 2k: p61::create_transform, FACTOR_INDEX)
//   scatter  data[i] = work[2i] / (w^2i l'(w^2i))  for the erased data blocks
// Pattern-only part (decode_prepare), on the device: l by a full product tree over the zero-padded erasure list (a zero root is
// a factor x, undone by a table lookup per position), every level ONE batch of cyclic products — all polynomials of a level
// are the element columns of a stripe that the path's own stand-alone transform handles; the values of L and x L' by one
// transform of a two-column stripe; inverses as conj / norm with the norm inverted by a Fermat power in GF(p).
//
// Parity: nothing upstream to pin to (there is no code for the field at all); the checks are the size-independent round trip
// encode -> erase -> decode = original, and the oracle's O(N^2) Lagrange decoder (tests/test_gpu_p61.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <new>
#include <chrono>
#include <vector>

#include "../../include/fastecc.h"
#include "gf61.hpp"
#include "gf61_path.hpp"

namespace fastecc {
namespace p61 {

namespace {

using gf61::Elem;
using gf61::P;
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

enum : uint32_t { ST_LOST = 0, ST_HELD = 1 };
constexpr int LEAF_LOG = 4, LEAF = 1 << LEAF_LOG;
constexpr int TREE_LOW = 8;  // tall trees: the polynomials of 2^TREE_LOW roots come from one kernel (k_tree_low) instead of four more levels of six to eight launches

__device__ __forceinline__ Elem ld(const uint64_t* p)
{
    const u64x2 t = *reinterpret_cast<const u64x2*>(p);
    return Elem{t.x, t.y};
}
__device__ __forceinline__ void st(uint64_t* p, Elem e)
{
    u64x2 t;
    t.x = e.re;
    t.y = e.im;
    *reinterpret_cast<u64x2*>(p) = t;
}
// canonical x canonical -> canonical (both operands per lane: the twiddle limbs live in VGPRs here)
__device__ __forceinline__ Elem mulc(Elem x, Elem y, const gf61::Opaque& k) { return gf61::canon(gf61::mul(x, gf61::make_twiddle(y.re, y.im), k)); }
__device__ __forceinline__ uint64_t addc(uint64_t x, uint64_t y)
{
    const uint64_t s = x + y;
    return s >= P ? s - P : s;
}
__device__ __forceinline__ uint64_t subc(uint64_t x, uint64_t y) { return x >= y ? x - y : x + P - y; }
__device__ __forceinline__ Elem addc(Elem x, Elem y) { return Elem{addc(x.re, y.re), addc(x.im, y.im)}; }
__device__ __forceinline__ Elem powc(Elem x, uint64_t e, const gf61::Opaque& k)
{
    Elem r{1, 0};
    for (; e; e >>= 1) {
        if (e & 1u) r = mulc(r, x, k);
        x = mulc(x, x, k);
    }
    return r;
}
// 1 / x = conj(x) / (re^2 + im^2); the norm lies in GF(p) and is inverted by norm^(p-2)
__device__ __forceinline__ Elem invc(Elem x, const gf61::Opaque& k)
{
    const Elem re2 = mulc(Elem{x.re, 0}, Elem{x.re, 0}, k), im2 = mulc(Elem{x.im, 0}, Elem{x.im, 0}, k);
    const Elem ninv = powc(Elem{addc(re2.re, im2.re), 0}, P - 2, k);
    return mulc(Elem{x.re, subc(0, x.im)}, ninv, k);
}

__global__ __launch_bounds__(256) void k_wpow(uint64_t* __restrict__ wpow, uint64_t wre, uint64_t wim, uint32_t count)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= count) return;
    const gf61::Opaque k = gf61::make_opaque();
    st(wpow + 2ull * u, powc(Elem{wre, wim}, u, k));
}

// the lost positions listed on the device: sixteen positions per thread, one atomic per workgroup (the order of the list between workgroups is not
// fixed; the locator is a product over the list and does not depend on it)
__global__ __launch_bounds__(256) void k_erased_list(const uint8_t* __restrict__ state, uint32_t NC, uint32_t* __restrict__ erased, uint32_t* __restrict__ counter)
{
    __shared__ uint32_t wave_sum[4], block_base;
    const uint32_t u0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    uint32_t bits = 0;
    if (u0 + 16u <= NC) {
        const uint4 v = *reinterpret_cast<const uint4*>(state + u0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 16; ++i) bits |= (uint32_t)(((w[i >> 2] >> (8 * (i & 3))) & 0xFFu) == ST_LOST) << i;
    } else {
        for (uint32_t i = 0; i < 16u && u0 + i < NC; ++i) bits |= (uint32_t)(state[u0 + i] == ST_LOST) << i;
    }
    const uint32_t mine = (uint32_t)__builtin_popcount(bits), lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= (uint32_t)d) incl += t;
    }
    if (lane == 63u) wave_sum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
        block_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t at = block_base + incl - mine;
    for (uint32_t w = 0; w < wave; ++w) at += wave_sum[w];
    for (uint32_t b = bits; b; b &= b - 1u) erased[at++] = u0 + (uint32_t)__builtin_ctz(b);
}

// even / odd split: the parity blocks not at multiples of 2^h are not in use — roots of the locator like the lost ones (nothing of them is
// rebuilt from this state; state_real keeps the caller's flags)
__global__ __launch_bounds__(256) void k_mark_unused(uint8_t* __restrict__ state, uint32_t N, uint32_t mask)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < N && (j & mask) != 0) state[2u * j + 1u] = ST_LOST;
}

__global__ __launch_bounds__(256) void k_roots(uint64_t* __restrict__ roots, const uint32_t* __restrict__ erased, const uint64_t* __restrict__ wpow,
                                               uint32_t n_erased, uint32_t T)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T) return;
    st(roots + 2ull * i, i < n_erased ? ld(wpow + 2ull * erased[i]) : Elem{0, 0});
}

// ---- transforms of FEW columns (the locator tree's upper levels: 2, 4, .. polynomials side by side; the pattern transform: 2 columns) ----
// The stripe kernels give every wave 64 columns of one row: with E < 64 columns most lanes idle and a row is a 16 E-byte access.  Here the rows x E
// array is taken as N1 x (N2 E) with N2 E = 2^CHUNK_LOG: (1) an N1-point DIF over the upper row bits by the stripe kernels, the N2 E elements of a
// chunk being its columns (twiddles uniform per row, as those kernels want them); (2) this kernel, one workgroup per chunk c = bitrev(k1): times
// w_N^(i2 k1) (the four-step method's diagonal), then the N2-point DIF over i2 inside LDS, twiddles per lane from the w^u table.  Row k1 + N1 k2
// of the transform ends up at row bitrev(k1 + N1 k2), where the stripe kernels' DIF passes alone would leave it.
constexpr int CHUNK_LOG = 12, CHUNK = 1 << CHUNK_LOG;
constexpr uint64_t NARROW_COLUMNS = 8;  // tree levels of at most this many polynomials go this way
template <bool INV>
__global__ __launch_bounds__(256) void k_chunk_dif(uint64_t* __restrict__ data, const uint64_t* __restrict__ wpow, int logE, int logN1, uint32_t step_n,
                                                   uint32_t step_n2, uint32_t nc_mask)
{
    __shared__ u64x2 lds[CHUNK];
    const gf61::Opaque k = gf61::make_opaque();
    const uint32_t tid = threadIdx.x, k1 = __brev(blockIdx.x) >> (32 - logN1);
    uint64_t* base = data + 2ull * CHUNK * blockIdx.x;
    auto twiddle = [&](uint32_t ex) { return ld(wpow + 2ull * (INV ? (0u - ex) & nc_mask : ex)); };
    auto put = [&](uint32_t i, Elem v) {
        u64x2 t;
        t.x = v.re;
        t.y = v.im;
        lds[i] = t;
    };
    auto get = [&](uint32_t i) {
        const u64x2 t = lds[i];
        return Elem{t.x, t.y};
    };
    auto butterfly = [&](Elem& x, Elem& y, uint32_t ex) {
        const Elem s = addc(x, y), t{subc(x.re, y.re), subc(x.im, y.im)};
        x = s;
        y = mulc(t, twiddle(ex), k);
    };
    const int levels = CHUNK_LOG - logE;  // of the N2-point transform
    // the diagonal and the first level on the way in: a thread takes i and i + CHUNK / 2
#pragma unroll 2
    for (int r = 0; r < CHUNK / 512; ++r) {
        const uint32_t i = r * 256u + tid, j = i + CHUNK / 2;
        Elem x = mulc(ld(base + 2ull * i), twiddle((i >> logE) * k1 * step_n), k);
        Elem y = mulc(ld(base + 2ull * j), twiddle((j >> logE) * k1 * step_n), k);
        butterfly(x, y, (i >> logE) * step_n2);
        put(i, x);
        put(j, y);
    }
    __syncthreads();
    for (int l = 1; l < levels; ++l) {
        const int hb = CHUNK_LOG - 1 - l;
        const uint32_t low = (1u << hb) - 1u;
        const bool last = l + 1 == levels;
#pragma unroll 2
        for (int r = 0; r < CHUNK / 512; ++r) {
            const uint32_t b = r * 256u + tid, i = ((b >> hb) << (hb + 1)) | (b & low), j = i + low + 1u;
            Elem x = get(i), y = get(j);
            butterfly(x, y, (((i & low) >> logE) << l) * step_n2);
            if (last) {
                st(base + 2ull * i, x);
                st(base + 2ull * j, y);
            } else {
                put(i, x);
                put(j, y);
            }
        }
        if (!last) __syncthreads();
    }
}

// polynomial p = prod_{j < leaf} (x - roots[p*leaf + j]); coefficient i (< leaf; the monic one is implied) -> x[i*m + p]
__global__ __launch_bounds__(64) void k_leaves(const uint64_t* __restrict__ roots, uint64_t* __restrict__ x, uint32_t leaf, uint32_t m)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const gf61::Opaque k = gf61::make_opaque();
    Elem c[LEAF + 1];
#pragma unroll
    for (int i = 0; i <= LEAF; ++i) c[i] = Elem{i == 0 ? 1ull : 0ull, 0};
    for (uint32_t j = 0; j < leaf; ++j) {
        const Elem r = ld(roots + 2ull * (p * leaf + j));
#pragma unroll
        for (int i = LEAF; i >= 1; --i) {  // c <- c * (x - r)
            const Elem t = mulc(r, c[i], k);
            c[i] = Elem{subc(c[i - 1].re, t.re), subc(c[i - 1].im, t.im)};
        }
        const Elem t = mulc(r, c[0], k);
        c[0] = Elem{subc(0, t.re), subc(0, t.im)};
    }
    for (uint32_t i = 0; i < leaf; ++i) st(x + 2ull * ((uint64_t)i * m + p), c[i]);
}

// The lowest LOW levels of the product tree in one kernel (as decode.hip's tree_low_levels_kernel): a workgroup takes 2^LOW roots, multiplies the
// monic polynomials pairwise in LDS by the schoolbook rule, degree 1 -> 2 -> ... -> 2^LOW ((x^d + a)(x^d + b) = x^2d + x^d (a + b) + a b), one
// thread per coefficient, and writes coefficient i of polynomial p to x[i * m + p] (the upper half of the 2 * 2^LOW rows is zero).
template <int LOW>
__global__ __launch_bounds__(1 << LOW) void k_tree_low(const uint64_t* __restrict__ roots, uint64_t* __restrict__ x, uint32_t m)
{
    constexpr uint32_t R = 1u << LOW;
    __shared__ uint64_t bre[2][R], bim[2][R];
    const uint32_t p = blockIdx.x, o = threadIdx.x;
    const gf61::Opaque k = gf61::make_opaque();
    {
        const Elem r = ld(roots + 2ull * ((uint64_t)p * R + o));
        bre[0][o] = subc(0, r.re);  // x - r
        bim[0][o] = subc(0, r.im);
    }
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < R; d <<= 1) {
        const uint32_t t = o & (2u * d - 1u), a0 = o - t, b0 = a0 + d;
        const uint32_t lo_i = t >= d ? t - d + 1u : 0u, hi_i = t < d ? t : d - 1u;
        Elem acc{0, 0};
        if (t >= d) acc = addc(Elem{bre[cur][a0 + t - d], bim[cur][a0 + t - d]}, Elem{bre[cur][b0 + t - d], bim[cur][b0 + t - d]});
        for (uint32_t i = lo_i; i <= hi_i; ++i)  // (t = 2d - 1: lo_i > hi_i, no product)
            acc = addc(acc, mulc(Elem{bre[cur][a0 + i], bim[cur][a0 + i]}, Elem{bre[cur][b0 + t - i], bim[cur][b0 + t - i]}, k));
        bre[cur ^ 1][o] = acc.re;
        bim[cur ^ 1][o] = acc.im;
        __syncthreads();
        cur ^= 1;
    }
    st(x + 2ull * ((uint64_t)o * m + p), Elem{bre[cur][o], bim[cur][o]});
}

// y[i][q] = f[i'][2q] * f[i'][2q+1] * scale, q < m/2 (rows of m/2 elements; f: rows of m).  The transforms stay in the order their DIF passes leave
// (dif_only: no reordering pass, 29 launches of 10 us per pattern in round 5): the value for point i is in row i' = bitrev(i) of f; y is written
// in natural order, which is what the way back's DIF passes read.
__global__ __launch_bounds__(256) void k_pairs(const uint64_t* __restrict__ f, uint64_t* __restrict__ y, uint32_t m, uint64_t total, uint64_t sre,
                                               uint64_t sim, int lg)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint32_t half = m >> 1;
    const uint64_t i = t / half;
    const uint32_t q = (uint32_t)(t - i * half);
    const uint64_t ir = __brev((uint32_t)i) >> (32 - lg);
    st(y + 2 * (i * half + q), mulc(mulc(ld(f + 2 * (ir * m + 2 * q)), ld(f + 2 * (ir * m + 2 * q + 1)), k), Elem{sre, sim}, k));
}

// (x^d + a)(x^d + b) = x^2d + x^d (a + b) + a b: xnew [4d][m/2] (upper 2d rows zero) from the cyclic products y (rows of m/2) and xold [d][m]
// (y: the cyclic products as the way back's DIF passes leave them — coefficient i in row bitrev(i) of its 2d rows)
__global__ __launch_bounds__(256) void k_combine(const uint64_t* __restrict__ y, const uint64_t* __restrict__ xold, uint64_t* __restrict__ xnew,
                                                 uint32_t d, uint32_t m, uint64_t total, bool top, int lg)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const uint32_t half = m >> 1;
    const uint64_t i = t / half;
    const uint32_t q = (uint32_t)(t - i * half);
    Elem v{0, 0};
    if (i < 2ull * d) {
        v = ld(y + 2 * ((uint64_t)(__brev((uint32_t)i) >> (32 - lg)) * half + q));
        if (i >= d) v = addc(v, addc(ld(xold + 2 * ((i - d) * m + 2 * q)), ld(xold + 2 * ((i - d) * m + 2 * q + 1))));
    } else if (top) {
        return;
    }
    st(xnew + 2 * (i * half + q), v);
}

// lv[m][0] = c_m, lv[m][1] = m c_m for L = x^T + sum c_m x^m modulo x^NC - 1
__global__ __launch_bounds__(256) void k_locator_columns(const uint64_t* __restrict__ c, uint64_t* __restrict__ lv, uint32_t T, uint32_t NC)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= NC) return;
    const gf61::Opaque k = gf61::make_opaque();
    Elem v0 = m < T ? ld(c + 2ull * m) : Elem{0, 0};
    Elem v1 = mulc(v0, Elem{m, 0}, k);
    if (m == T % NC) {
        v0.re = addc(v0.re, 1);
        v1.re = addc(v1.re, T);
    }
    st(lv + 4ull * m, v0);
    st(lv + 4ull * m + 2, v1);
}

// fin[u] = l(w^u) on surviving positions (0 elsewhere); gout[i] = 1 / (w^2i l'(w^2i)) for erased data block i (0 elsewhere), gout_all[u] the
// same by position for every lost block (optional);
// l = L w^(-u pad) on the points (the padding), see decode.hip finish_tables_kernel
__global__ __launch_bounds__(256) void k_finish(const uint64_t* __restrict__ lv, const uint8_t* __restrict__ state, const uint64_t* __restrict__ wpow,
                                                uint64_t* __restrict__ fin, uint64_t* __restrict__ gout, uint32_t NC, uint32_t pad,
                                                uint64_t* __restrict__ gout_all, int e, int lg_nc)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= NC) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint64_t ur = lg_nc ? (__brev(u) >> (32 - lg_nc)) : 0;  // lv keeps the order its transform's DIF passes leave: the value for w^u in row bitrev(u)
    const uint32_t back = (uint32_t)(((uint64_t)u * pad) % NC);
    const bool held = state[u] == ST_HELD;
    st(fin + 2ull * u, held ? mulc(ld(lv + 4ull * ur), ld(wpow + 2ull * (back == 0 ? 0 : NC - back)), k) : Elem{0, 0});
    // the output factors: zero here, those of the lost positions from k_finish_lost (an inversion each: only the listed positions pay for one)
    if ((u & ((1u << e) - 1u)) == 0) st(gout + 2ull * (u >> e), Elem{0, 0});
    if (gout_all) st(gout_all + 2ull * u, Elem{0, 0});
}
__global__ __launch_bounds__(256) void k_finish_lost(const uint64_t* __restrict__ lv, const uint32_t* __restrict__ erased, uint32_t n_erased,
                                                     const uint64_t* __restrict__ wpow, uint64_t* __restrict__ gout, uint32_t NC, uint32_t pad,
                                                     uint64_t* __restrict__ gout_all, int e, int lg_nc)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_erased) return;
    const uint32_t u = erased[j];
    const bool data_pos = (u & ((1u << e) - 1u)) == 0;  // data block i sits at position i << e (e = 1: the (2k,k) code; 2, 3: n = 4k, 8k)
    if (!data_pos && !gout_all) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint64_t ur = lg_nc ? (__brev(u) >> (32 - lg_nc)) : 0;
    const uint32_t back = (uint32_t)(((uint64_t)u * pad) % NC);
    const Elem g = invc(mulc(ld(lv + 4ull * ur + 2), ld(wpow + 2ull * (back == 0 ? 0 : NC - back)), k), k);
    if (data_pos) st(gout + 2ull * (u >> e), g);
    if (gout_all) st(gout_all + 2ull * u, g);  // every lost position, parity too: fastecc_repair in one transform
}

// One wave per (row, 64-element column chunk); the row's factor is wave-uniform.
using const_u64_ptr = const uint64_t __attribute__((address_space(4)))*;
__device__ __forceinline__ const_u64_ptr as_constant(const uint64_t* p) { return (const_u64_ptr)(reinterpret_cast<uintptr_t>(p)); }

// work[u] = (u even ? data[u/2] : parity[u/2]) * fin[u]; zero rows where fin == 0 (nothing is read there)
__global__ __launch_bounds__(256) void k_gather(const uint64_t* __restrict__ data, const uint64_t* __restrict__ parity, uint64_t* __restrict__ work,
                                                const uint64_t* __restrict__ fin, uint32_t elems, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t u = (uint32_t)(item / col_chunks);
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint64_t fre = as_constant(fin)[2ull * u], fim = as_constant(fin)[2ull * u + 1];
    Elem v{0, 0};
    if ((fre | fim) != 0) {
        const uint64_t* src = ((u & 1u) ? parity : data) + ((uint64_t)(u >> 1) * elems + col) * 2;
        v = gf61::mul(ld(src), gf61::make_twiddle(fre, fim), k);  // lazy: the transform's first pass takes it
    }
    st(work + ((uint64_t)u * elems + col) * 2, v);
}

// even / odd split, small form of the parity half: row m of `small` = parity block m << h times fin[2 (m << h) + 1] (zero rows where that
// parity block is lost or not in use: nothing is read there)
__global__ __launch_bounds__(256) void k_split_small_gather(const uint64_t* __restrict__ parity, uint64_t* __restrict__ small, const uint64_t* __restrict__ fin,
                                                            uint32_t elems, int h, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t m = (uint32_t)(item / col_chunks);
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint64_t j = (uint64_t)m << h;
    const uint64_t fre = as_constant(fin)[2ull * (2 * j + 1)], fim = as_constant(fin)[2ull * (2 * j + 1) + 1];
    Elem v{0, 0};
    if ((fre | fim) != 0) v = gf61::mul(ld(parity + (j * elems + col) * 2), gf61::make_twiddle(fre, fim), k);  // lazy: the transform's first pass takes it
    st(small + ((uint64_t)m * elems + col) * 2, v);
}

// n = 4k / 8k: work[u] = block srcmap[u] (bit 31: of the parity stripe) times fin[u]; zero rows where fin == 0 (nothing is read there)
__global__ __launch_bounds__(256) void k_gather_map(const uint64_t* __restrict__ data, const uint64_t* __restrict__ parity, uint64_t* __restrict__ work,
                                                    const uint64_t* __restrict__ fin, const uint32_t* __restrict__ srcmap, uint32_t elems, uint32_t col_chunks,
                                                    uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t u = (uint32_t)(item / col_chunks);
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint64_t fre = as_constant(fin)[2ull * u], fim = as_constant(fin)[2ull * u + 1];
    Elem v{0, 0};
    if ((fre | fim) != 0) {
        const uint32_t m = srcmap[u];
        const uint64_t* src = ((m >> 31) ? parity : data) + ((uint64_t)(m & 0x7FFFFFFFu) * elems + col) * 2;
        v = gf61::mul(ld(src), gf61::make_twiddle(fre, fim), k);
    }
    st(work + ((uint64_t)u * elems + col) * 2, v);
}
// ... and parity[q] = again[q] for the lost parity blocks (lost[q] != 0)
__global__ __launch_bounds__(256) void k_restore_map(const uint64_t* __restrict__ again, uint64_t* __restrict__ parity, const uint8_t* __restrict__ lost,
                                                     uint32_t elems, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t q = (uint32_t)(item / col_chunks);
    if (!lost[q]) return;
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    st(parity + ((uint64_t)q * elems + col) * 2, ld(again + ((uint64_t)q * elems + col) * 2));
}

// split repair: gout_par[j] = gout_all[2j + 1] for the parity blocks the CALLER lost (state_real), zero for the rest — the locator's state counts the
// unused parity blocks as lost too, and those are not to be rewritten
__global__ __launch_bounds__(256) void k_gout_par(const uint64_t* __restrict__ gout_all, const uint8_t* __restrict__ state_real, uint64_t* __restrict__ gout_par,
                                                  uint32_t N)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    st(gout_par + 2ull * j, state_real[2u * j + 1u] == ST_HELD ? Elem{0, 0} : ld(gout_all + 2ull * (2u * j + 1u)));
}

// data[i] = work[stride * i] * gout[i] for the erased data blocks (gout != 0)
__global__ __launch_bounds__(256) void k_scatter(const uint64_t* __restrict__ work, uint64_t* __restrict__ data, const uint64_t* __restrict__ gout,
                                                 uint32_t elems, uint32_t col_chunks, uint64_t items, uint32_t stride)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t i = (uint32_t)(item / col_chunks);
    const uint64_t gre = as_constant(gout)[2ull * i], gim = as_constant(gout)[2ull * i + 1];
    if ((gre | gim) == 0) return;  // wave-uniform
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    const gf61::Opaque k = gf61::make_opaque();
    const Elem v = gf61::mul(ld(work + ((uint64_t)stride * i * elems + col) * 2), gf61::make_twiddle(gre, gim), k);  // stride 2: row 2i of the 2k outputs
    st(data + ((uint64_t)i * elems + col) * 2, gf61::canon(v));
}

// parity[j] = again[j] for the lost parity blocks
__global__ __launch_bounds__(256) void k_restore(const uint64_t* __restrict__ again, uint64_t* __restrict__ parity, const uint8_t* __restrict__ state,
                                                 uint32_t elems, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t j = (uint32_t)(item / col_chunks);
    if (state[2u * j + 1u] == ST_HELD) return;
    const uint32_t col = cc * 64u + lane;
    if (col >= elems) return;
    st(parity + ((uint64_t)j * elems + col) * 2, ld(again + ((uint64_t)j * elems + col) * 2));
}

// ---- few losses: every lost block is a fixed linear combination of the surviving ones (see decode.hip, "Few losses") ----
constexpr int DIRECT_MAX = DECODE_DIRECT_MAX;  // 32 (16 until the nodes became k blocks instead of all survivors); a power of two: the tables' capacity
constexpr uint32_t DIRECT_ROWS = 512, DIRECT_SEGS = 32;

// ---- lazy accumulation: no reduction per term ----
// A weight component c < 2^61 is wave-uniform.  With c' = c * 2^32 mod p (a rotation of its 61 bits) a data component a = a1 2^32 + a0 (ANY 64-bit
// word) gives a * c = a0 * c + a1 * c' (mod p), and with c = l0 + l1 2^21 + l2 2^42 (limbs of 21, 21, 19 bits) every partial product is below 2^53:
// a sum of 4 products per row over DIRECT_ROWS = 512 rows stays below 2^64 in a plain 64-bit accumulator — one v_mad_u64_u32 per product, no carry,
// no fold.  re = a c + b (p - d), im = a d + b c: 24 multiply-adds per term (the reducing form: 16 + limb splits + three folds = 58 instructions).
// The 18 limbs of a weight — of c, c', d, d', e = p - d, e' — are what the table holds (COEF_WORDS 32-bit words per entry, all zero for a zero weight):
// splitting them in the kernel costs 35 scalar instructions per term, and the scalar unit, shared by the four SIMDs of a CU, then limits the kernel
// (measured: 16 lost blocks 3.0 ms against 3.6 for the reducing form).  The three sums of a component are put together once per chunk.
constexpr int COEF_WORDS = 18;
struct Limbs {
    uint32_t l0, l1, l2;
};
__device__ __forceinline__ Limbs limbs_of(uint64_t v) { return Limbs{(uint32_t)v & 0x1FFFFFu, (uint32_t)(v >> 21) & 0x1FFFFFu, (uint32_t)(v >> 42)}; }
__device__ __forceinline__ uint64_t times_2_32(uint64_t v) { return ((v << 32) & gf61::P) | (v >> 29); }  // v <= p: v 2^32 mod p (or p for v = p)
struct Acc3 {
    uint64_t t0, t1, t2;
};
__device__ __forceinline__ void mac3(Acc3& s, uint32_t x, const Limbs& w)
{
    s.t0 += (uint64_t)x * w.l0;
    s.t1 += (uint64_t)x * w.l1;
    s.t2 += (uint64_t)x * w.l2;
}
// t0 + t1 2^21 + t2 2^42 (mod p) as a lazy value (< 2^61 + 4)
__device__ __forceinline__ uint64_t gather3(const Acc3& s)
{
    auto fold = [](uint64_t t) { return (t & gf61::P) + (t >> 61); };                                       // < 2^61 + 8
    auto shl = [](uint64_t y, int sh) { return ((y << sh) & gf61::P) + (y >> (61 - sh)); };                // y < 2^62: y 2^sh mod p, < 2^61 + 2^(sh + 1)
    return fold(fold(s.t0) + shl(fold(s.t1), 21) + shl(fold(s.t2), 42));                                    // the sum is below 2^63
}

// The nodes of the interpolation are k blocks: the data rows (a lost one gets zero weights) and as many surviving parity blocks y_a as data
// blocks are lost — half the read, and half the arithmetic, of a sum over all 2k - e survivors (round 6: 11.8 -> 6 ms for one lost block at
// k = 2^19 x 64 KB).  The formulas are decode.hip's / direct.hip's (interp_coef_kernel): with x_i = w^2i, A(x) = prod_a (x - y_a), R(x) = prod_r
// (x - x_r) over the lost data rows, R_r = R / (x - x_r), and y^k = -1 at every parity point:
//     data row i, lost data row r:     C_r x_i R_r(x_i) / A(x_i),                   C_r = -A(x_r) / (x_r R_r(x_r))
//     data row i, lost parity block t: c_t x_i R(x_i) / (A(x_i) (y_t - x_i)),       c_t = -2 A(y_t) / (k R(y_t))
//     parity node a: on the host (decode_prepare), packed by k_direct_pack
// params (elements): [0, MAX) the targets x_r then y_t, [MAX, 2 MAX) y_a, [2 MAX, 3 MAX) C_r then c_t
__device__ __forceinline__ Elem subc(Elem x, Elem y) { return Elem{subc(x.re, y.re), subc(x.im, y.im)}; }
__device__ __forceinline__ void store_weight(uint32_t* __restrict__ o, Elem v)
{
    if ((v.re | v.im) == 0) {
        for (int i = 0; i < COEF_WORDS; ++i) o[i] = 0;
        return;
    }
    const uint64_t e = gf61::P - v.im;
    const uint64_t parts[6] = {v.re, times_2_32(v.re), v.im, times_2_32(v.im), e, times_2_32(e)};
    for (int i = 0; i < 6; ++i) {
        const Limbs l = limbs_of(parts[i]);
        o[3 * i] = l.l0;
        o[3 * i + 1] = l.l1;
        o[3 * i + 2] = l.l2;
    }
}
__global__ __launch_bounds__(256) void k_direct_coef(uint32_t* __restrict__ coef, const uint64_t* __restrict__ wpow, const uint64_t* __restrict__ params, uint32_t N,
                                                     int ed, int ep, int pad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint64_t *z = params, *ya = params + 2 * DIRECT_MAX, *c = params + 4 * DIRECT_MAX;
    const Elem xi = ld(wpow + 4ull * i);  // w^2i
    Elem A{1, 0}, R{1, 0}, D{1, 0};
    for (int a = 0; a < ed; ++a) {
        A = mulc(A, subc(xi, ld(ya + 2 * a)), k);
        R = mulc(R, subc(xi, ld(z + 2 * a)), k);
    }
    for (int t = 0; t < ep; ++t) D = mulc(D, subc(ld(z + 2 * (ed + t)), xi), k);
    const bool node = (R.re | R.im) != 0;  // (a lost row is no node)
    const Elem inv = invc(mulc(A, D, k), k);  // one inversion for both denominators: A(x_i) and prod_t (y_t - x_i), neither is zero
    const Elem base = mulc(xi, mulc(inv, D, k), k), base_far = mulc(mulc(base, R, k), mulc(inv, A, k), k);
    for (int j = 0; j < pad; ++j) {
        Elem v{0, 0};
        if (node && j < ed) {
            Elem Rr{1, 0};
            for (int s = 0; s < ed; ++s)
                if (s != j) Rr = mulc(Rr, subc(xi, ld(z + 2 * s)), k);
            v = mulc(mulc(ld(c + 2 * j), base, k), Rr, k);
        } else if (node && j < ed + ep) {
            Elem Q{1, 0};
            for (int s = ed; s < ed + ep; ++s)
                if (s != j) Q = mulc(Q, subc(ld(z + 2 * s), xi), k);
            v = mulc(mulc(ld(c + 2 * j), base_far, k), Q, k);
        }
        store_weight(coef + ((uint64_t)i * pad + j) * COEF_WORDS, v);
    }
}
// the parity nodes' rows (k .. k + ed - 1 of the table) from their weights as elements
__global__ __launch_bounds__(256) void k_direct_pack(uint32_t* __restrict__ coef_rows, const uint64_t* __restrict__ weights, uint32_t count)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) store_weight(coef_rows + (uint64_t)t * COEF_WORDS, ld(weights + 2ull * t));
}

// partial[chunk][j][col] = sum over the chunk's rows of block(u)[col] * coef[u][j] (lazy values); rows u < n_data are the data blocks, the others the
// parity blocks nodes[u - n_data]; a wave owns (chunk, 64 element columns, a sweep of EB outputs: blockIdx.y)
template <int EB>
__global__ __launch_bounds__(256) void k_direct_accumulate(const uint64_t* __restrict__ data, const uint64_t* __restrict__ parity,
                                                           const uint32_t* __restrict__ coef, uint64_t* __restrict__ partial, uint32_t elems, uint32_t NC,
                                                           uint32_t col_chunks, uint64_t items, uint32_t pad, uint32_t n_data, const uint32_t* __restrict__ nodes)
{
    constexpr int U = EB >= 8 ? 2 : 4;      // rows per trip of the loop; the next trip's rows are requested before this trip's arithmetic
    constexpr int G = EB >= 4 ? 4 : EB;     // outputs and
    constexpr int RB = EB >= 4 ? 1 : 4 / EB;  // rows whose limbs come in one scalar fetch (72 SGPRs); rows are adjacent in the table when pad == EB (EB < 8)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t j0 = blockIdx.y * EB;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t chunk = (uint32_t)(item / col_chunks);
    const uint32_t col = min(cc * 64u + lane, elems - 1u);  // lanes past a ragged end repeat the last column (same loads, same stores)
    Acc3 re[EB], im[EB];
#pragma unroll
    for (int j = 0; j < EB; ++j) re[j] = im[j] = Acc3{0, 0, 0};
    const uint32_t u0 = chunk * DIRECT_ROWS, u1 = min(u0 + DIRECT_ROWS, NC);
    auto fetch = [&](Elem* x, uint32_t ub) {  // what stands in a lost block's place meets zero coefficients
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const uint32_t u = min(ub + i, u1 - 1u);
            const uint64_t* row = u < n_data ? data + (uint64_t)u * elems * 2 : parity + (uint64_t)nodes[u - n_data] * elems * 2;  // (wave-uniform)
            x[i] = ld(row + 2ull * col);
        }
    };
    Elem xn[U];
    fetch(xn, u0);
    for (uint32_t ub = u0; ub < u1; ub += U) {
        Elem x[U];
#pragma unroll
        for (int i = 0; i < U; ++i) x[i] = xn[i];
        fetch(xn, min(ub + U, u1 - 1u));
#pragma unroll
        for (int ib = 0; ib < U; ib += RB) {
#pragma unroll
            for (int g = 0; g < EB; g += G) {
                // (the table has four rows more than NC: a fetch that starts at a row below NC stays inside it)
                const __attribute__((address_space(4))) uint32_t* cf =
                    (const __attribute__((address_space(4))) uint32_t*)(coef + ((uint64_t)(ub + ib) * pad + j0 + g) * COEF_WORDS);
                uint32_t w[RB * G * COEF_WORDS];  // pinned, so that the loads are not sunk into each output's branch
#pragma unroll
                for (int t = 0; t < RB * G * COEF_WORDS; ++t) w[t] = cf[t];
#pragma unroll
                for (int t = 0; t < RB * G * COEF_WORDS; ++t) asm volatile("" : "+s"(w[t]));
#pragma unroll
                for (int ir = 0; ir < RB; ++ir) {
                    const int i = ib + ir;
                    if (ub + i >= u1) continue;
                    const uint32_t a0 = (uint32_t)x[i].re, a1 = (uint32_t)(x[i].re >> 32), b0 = (uint32_t)x[i].im, b1 = (uint32_t)(x[i].im >> 32);
#pragma unroll
                    for (int jj = 0; jj < G; ++jj) {
                        const uint32_t* q = w + (ir * G + jj) * COEF_WORDS;
                        if ((q[0] | q[1] | q[2] | q[6] | q[7] | q[8]) == 0) continue;  // wave-uniform: a lost position, or a padding output
                        const Limbs C{q[0], q[1], q[2]}, C2{q[3], q[4], q[5]}, D{q[6], q[7], q[8]}, D2{q[9], q[10], q[11]}, E{q[12], q[13], q[14]},
                            E2{q[15], q[16], q[17]};
                        const int j = g + jj;
                        mac3(re[j], a0, C);
                        mac3(re[j], a1, C2);
                        mac3(re[j], b0, E);
                        mac3(re[j], b1, E2);
                        mac3(im[j], a0, D);
                        mac3(im[j], a1, D2);
                        mac3(im[j], b0, C);
                        mac3(im[j], b1, C2);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < EB; ++j)
        st(partial + 2ull * (((uint64_t)chunk * pad + j0 + j) * elems + col), Elem{gather3(re[j]), gather3(im[j])});
}

__global__ __launch_bounds__(256) void k_direct_reduce1(const uint64_t* __restrict__ partial, uint64_t* __restrict__ stage, uint32_t elems, uint32_t chunks,
                                                        int pad, int e)
{
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const uint32_t seg = blockIdx.z;
    if (col >= elems || j >= e) return;
    const gf61::Opaque k = gf61::make_opaque();
    const uint32_t per = (chunks + DIRECT_SEGS - 1) / DIRECT_SEGS;
    const uint32_t c0 = seg * per, c1 = min(c0 + per, chunks);
    Elem v{0, 0};
#pragma unroll 4
    for (uint32_t c = c0; c < c1; ++c) v = gf61::add(v, ld(partial + 2ull * (((uint64_t)c * pad + j) * elems + col)), k);
    st(stage + 2ull * (((uint64_t)seg * pad + j) * elems + col), v);
}
__global__ __launch_bounds__(256) void k_direct_reduce2(const uint64_t* __restrict__ stage, const uint32_t* __restrict__ epos, uint64_t* __restrict__ data,
                                                        uint64_t* __restrict__ parity, uint32_t elems, int pad, int e, bool with_parity)
{
    const uint32_t col = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (col >= elems || j >= e) return;
    const uint32_t pos = epos[j];
    if ((pos & 1u) && !with_parity) return;
    const gf61::Opaque k = gf61::make_opaque();
    Elem v{0, 0};
    for (uint32_t g = 0; g < DIRECT_SEGS; ++g) v = gf61::add(v, ld(stage + 2ull * (((uint64_t)g * pad + j) * elems + col)), k);
    st(((pos & 1u) ? parity : data) + ((uint64_t)(pos >> 1) * elems + col) * 2, gf61::canon(v));
}

// FASTECC_TRACE_PREPARE=1: wall-clock of the phases of the first decode_prepare on stderr (as decode.hip does for the 32-bit field)
struct PhaseTimer {
    bool on = getenv("FASTECC_TRACE_PREPARE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[fastecc prepare p61] %-24s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

int fail(char* detail, size_t cap, hipError_t e, const char* what)
{
    if (detail && cap) snprintf(detail, cap, "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}

#define D61_TRY(expr)                                              \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) return fail(detail, cap, e_, #expr); \
    } while (0)

}  // namespace

struct Decoder {
    int log2k = 0;
    uint64_t N = 0, NC = 0, T = 0, elems = 0;
    Path* transform = nullptr;         // size 2k, factor m / 2k, `elems` columns: x p'(x) on a whole stripe
    Path* pattern = nullptr;           // size 2k, 2 columns: L and x L' on the points
    std::vector<Path*> tree;           // level k >= LEAF_LOG: size 2^(k+1), T >> k columns
    std::vector<Path*> tree_inv;       // the way back of level k: size 2^(k+1), T >> (k+1) columns (the products)
    Path* narrow_tree_inv = nullptr;   // ... of the few-column levels: T / CHUNK rows of CHUNK columns
    Path *narrow_tree = nullptr, *narrow_pattern = nullptr;  // the upper row bits of the few-column transforms (k_chunk_dif): 2T / CHUNK and 2 NC / CHUNK rows of CHUNK columns
    uint64_t *tree_x = nullptr, *tree_y = nullptr, *tree_f = nullptr;  // 2T elements each
    uint64_t *wpow = nullptr, *roots = nullptr, *lv = nullptr, *fin = nullptr, *gout = nullptr;
    uint64_t* gout_all = nullptr;      // 2k factors by position (lazy: patterns that lose data AND parity), valid for the current pattern if gout_all_valid
    bool gout_all_valid = false;
    uint32_t* erased = nullptr;
    uint8_t* state = nullptr;
    uint64_t* work = nullptr;          // 2k blocks (lazy)
    Path* half = nullptr;              // size k, 6-level MID: the DIT passes and tables of the folded transform (gf61_path.hpp: encode_fold)
    uint64_t* rec = nullptr;           // k blocks: x p'(x) at the data positions, written by the folded transform (lazy)
    uint64_t* again = nullptr;         // k parity blocks of the re-encode (lazy, repair only)
    uint64_t* stage = nullptr;         // data + parity stripes of a host-memory call (lazy)
    // even / odd split ((2k,k) codes, k >= 2^11; the scheme of decode.hip's header): the data chain runs on `splitp`, a size-k path with the factor
    // (2m + k) / 2k; of the parity half only the blocks at multiples of 2^split_shift are used (the others count as erased in the locator), so its
    // DIF is the DIF of k >> shift rows (`small[shift]`, its stripe `small_buf`) and MID reads block p >> shift of it.
    // n = 4k / 8k (e = 2 / 3): the same scheme on the (k << e)-th roots of unity — data block i at position i << e, block j of coset t at
    // c 2^(e-jt) + (j << e) (include/fastecc.h's nesting order) — with the unfolded transform of all k << e positions, a gather through `srcmap`
    // and up to n - k erasures (T = k << e roots in the locator's tree)
    int e = 1;
    uint64_t M = 0;                      // parity blocks: (2^e - 1) k
    uint32_t* srcmap = nullptr;          // e > 1: position -> block (bit 31: parity stripe)
    uint8_t* parity_lost = nullptr;      // e > 1: M flags, the caller's lost parity blocks
    uint64_t* cos_work = nullptr;        // e > 1, repair: the k-block work stripe of the re-encode
    Path* splitp = nullptr;
    bool split_unavailable = false;      // no such plan / no memory: the folded 2k-point transform serves
    Path* small[6] = {};
    uint64_t* small_buf = nullptr;       // small_rows blocks (grown on demand: k >> shift)
    uint64_t small_rows = 0;
    uint64_t* split_af = nullptr;        // k elements by position: -1/2 w^(-bitrev(p))
    uint64_t* split_work = nullptr;      // k blocks: the data chain's intermediate stripe
    uint8_t* state_real = nullptr;       // the caller's flags (the locator's `state` counts unused parity blocks as lost)
    uint32_t lost_coset_mask = 0;        // n = 4k / 8k: bit t set = coset t (parity blocks [t k, (t+1) k)) has lost a block: fastecc_repair re-encodes those cosets only
    // fastecc_repair through the split: a second MID + DIT chain over the same two halves gives x p'(x) at the odd positions — the lost parity blocks
    uint64_t* split_q2 = nullptr;        // k blocks: q~ as the data chain's MID leaves it after its first half (lazy)
    uint64_t* split_pos_odd = nullptr;   // k elements by position: -1/2 w^(+bitrev(p))
    uint64_t* gout_par = nullptr;        // k elements by parity block: 1 / (w^(2j+1) l'(w^(2j+1))) where the caller lost block j, else 0
    bool split_repair_ready = false;
    bool split_pos_odd_built = false;
    int split_shift = 0;
    bool split_ready = false;
    uint64_t erased_data = 0, erased_parity = 0;
    bool built = false;  // contexts, buffers and the w^u table exist
    // few losses: the direct path
    int direct = 0, direct_pad = 0;
    uint64_t direct_nc = 0;              // positions of the code the direct path works in (2k; NC unless e > 1)
    int direct_ed = 0;                   // ... of its outputs the first direct_ed are the lost data blocks
    uint64_t direct_rows = 0;            // ... and its rows: the k data blocks, then direct_ed parity blocks
    uint32_t* direct_coef = nullptr;     // [NC][pad][COEF_WORDS]: the limbs of each weight
    uint64_t direct_coef_elems = 0;
    uint64_t* direct_inv = nullptr;      // the table kernels' parameters (3 DIRECT_MAX elements) and the parity nodes' weights
    uint32_t* direct_pos = nullptr;      // [0, MAX) output positions, [MAX, 2 MAX) the parity nodes' rows
    uint64_t* direct_partial = nullptr;  // [chunks + DIRECT_SEGS][pad][elems] elements
    uint64_t direct_partial_elems = 0;
    uint64_t* direct_wpow = nullptr;     // the w^u table when only this path has been used (else d->wpow)
    bool ready = false;
};

bool decoder_ready(const Decoder* d) { return d && d->ready; }

void destroy_decoder(Decoder* d)
{
    if (!d) return;
    destroy(d->transform);
    destroy(d->half);
    destroy(d->pattern);
    destroy(d->narrow_tree);
    destroy(d->narrow_tree_inv);
    for (Path* t : d->tree_inv) destroy(t);
    destroy(d->narrow_pattern);
    destroy(d->splitp);
    for (Path* t : d->small) destroy(t);
    for (void* b : {(void*)d->small_buf, (void*)d->split_af, (void*)d->split_work, (void*)d->state_real, (void*)d->srcmap, (void*)d->parity_lost, (void*)d->cos_work,
                    (void*)d->split_q2, (void*)d->split_pos_odd, (void*)d->gout_par})
        if (b) (void)hipFree(b);
    for (Path* t : d->tree) destroy(t);
    for (void* b : {(void*)d->tree_x, (void*)d->tree_y, (void*)d->tree_f, (void*)d->wpow, (void*)d->roots, (void*)d->lv, (void*)d->fin,
                    (void*)d->gout, (void*)d->gout_all, (void*)d->erased, (void*)d->state, (void*)d->work, (void*)d->rec, (void*)d->again, (void*)d->stage, (void*)d->direct_coef,
                    (void*)d->direct_inv, (void*)d->direct_pos, (void*)d->direct_partial, (void*)d->direct_wpow})
        if (b) (void)hipFree(b);
    delete d;
}

int decode_prepare(Decoder** slot, int log2k, uint64_t elems, const uint8_t* data_present, const uint8_t* parity_present, int direct_max, char* detail,
                   size_t cap, int split, int e)
{
    if (e < 1 || e > 3 || (*slot && (*slot)->built && (*slot)->e != e)) return FASTECC_E_INVAL;
    PhaseTimer pt_call;
    const uint64_t N = 1ull << log2k, NC = N << e, M = NC - N;
    if (NC > 0x7FFFFFFFull) return FASTECC_E_UNSUPPORTED;
    std::vector<uint8_t> state(NC);
    std::vector<uint32_t> srcmap(e > 1 ? NC : 0);
    std::vector<uint8_t> plost(e > 1 ? M : 0);
    uint64_t erased_data = 0, erased_parity = 0;
    if (e == 1) {  // data block i at 2i, parity block j at 2j + 1: one interleaving pass
        for (uint64_t i = 0; i < N; i++) {
            const uint8_t a = data_present[i] != 0, b = parity_present[i] != 0;
            state[2 * i] = a ? ST_HELD : ST_LOST;
            state[2 * i + 1] = b ? ST_HELD : ST_LOST;
            erased_data += !a;
            erased_parity += !b;
        }
    } else {
        for (uint64_t i = 0; i < N; i++) {
            state[i << e] = data_present[i] ? ST_HELD : ST_LOST;
            erased_data += !data_present[i];
            srcmap[i << e] = (uint32_t)i;
        }
        for (uint64_t q = 0; q < M; q++) {
            // parity block q = block j of coset t: generator w_(2^jt k)^c, jt = floor(log2(t + 1)) + 1, c the (t + 2 - 2^(jt-1))-th odd number (include/fastecc.h)
            const uint64_t t = q >> log2k, j = q & (N - 1);
            int jt = 1;
            while ((1ull << jt) - 1 <= t) jt++;
            const uint64_t c = 2 * (t + 1 - (1ull << (jt - 1))) + 1;
            const uint64_t u = (c << (e - jt)) + (j << e);
            state[u] = parity_present[q] ? ST_HELD : ST_LOST;
            erased_parity += !parity_present[q];
            srcmap[u] = (uint32_t)q | 0x80000000u;
            plost[q] = !parity_present[q];
        }
    }
    const int direct_max_user = direct_max;
    if (e > 1) {
        split = 0;       // the even / odd split and the direct path are the (2k,k) code's (the latter: of the (2k,k) code inside, see below)
        direct_max = 0;
    }
    // ---- even / odd split: recovering e data blocks takes e parity blocks, so the others may count as erased too; take them at multiples of 2^h of
    // the parity half (largest h <= 5 that leaves enough survivors) and the parity half's transform shrinks to k >> h rows ----
    int split_shift = 0;
    uint64_t unused_held = 0;  // parity blocks the caller holds that the split leaves aside: they count as lost in the locator
    const bool few = erased_data + erased_parity != 0 && (int)(erased_data + erased_parity) <= std::min(direct_max, DECODE_DIRECT_MAX);
    if (split && log2k >= 11 && erased_data != 0 && !few && !(*slot && (*slot)->split_unavailable)) {
        // (a pattern that has lost parity blocks too may be REPAIRED: the split then runs a second MID + DIT chain for the odd positions — or, without
        //  the memory for its extra stripe, re-encodes — see decode())
        for (int h = 5; h >= 1 && split_shift == 0; h--) {  // (light patterns stop at h = 5: k / 32 flags read)
            uint64_t held = 0;
            for (uint64_t j = 0; j < N; j += 1ull << h) held += parity_present[j] != 0;
            if (held >= erased_data) {
                split_shift = h;
                unused_held = (N - erased_parity) - held;  // marked on the device (k_mark_unused): the host's `state` stays the caller's
            }
        }
    }
    // the positions themselves are listed on the device (k_erased_list); the host needs their number, and the list itself only for the few-loss path
    const uint64_t n_erased = erased_data + erased_parity + unused_held;
    if (n_erased > NC - N) return FASTECC_E_INVAL;  // fewer than k blocks survive
    std::vector<uint32_t> erased;
    if (n_erased != 0 && (int64_t)n_erased <= std::min(direct_max, DECODE_DIRECT_MAX))
        for (uint64_t u = 0; u < NC; u++)
            if (state[u] == ST_LOST) erased.push_back((uint32_t)u);
    // n = 4k / 8k, few losses: the data and the FIRST coset are a (2k,k) code of their own (generator w_2k, parity blocks 0 .. k-1), and that code's
    // direct path rebuilds the data from its 2k - few survivors — a read of 2k blocks instead of a transform over n; lost parity blocks of the other
    // cosets are re-encoded (fastecc_repair)
    std::vector<uint8_t> state_sub;
    std::vector<uint32_t> erased_sub;
    if (e > 1 && erased_data != 0 && (int64_t)erased_data <= std::min(direct_max_user, DECODE_DIRECT_MAX)) {
        uint64_t lost0 = 0;
        for (uint64_t j = 0; j < N; j++) lost0 += !parity_present[j];
        if ((int64_t)(erased_data + lost0) <= std::min(direct_max_user, DECODE_DIRECT_MAX) && erased_data + lost0 <= N) {  // (k of the inner code's 2k blocks survive)
            state_sub.resize(2 * N);
            for (uint64_t i = 0; i < N; i++) {
                state_sub[2 * i] = data_present[i] ? ST_HELD : ST_LOST;
                state_sub[2 * i + 1] = parity_present[i] ? ST_HELD : ST_LOST;
            }
            for (uint64_t u = 0; u < 2 * N; u++)
                if (state_sub[u] == ST_LOST) erased_sub.push_back((uint32_t)u);
        }
    }
    pt_call.mark("pattern scan (host)");

    if (!*slot) {
        *slot = new (std::nothrow) Decoder();
        if (!*slot) return FASTECC_E_NOMEM;
    }
    Decoder* d = *slot;
    d->ready = false;
    d->log2k = log2k;
    d->N = N;
    d->NC = NC;
    d->elems = elems;
    d->erased_data = erased_data;
    d->erased_parity = erased_parity;
    if (erased_parity == 0) {
        // fastecc_repair's re-encode stripes (n - k and, for n = 4k / 8k, k more blocks: up to 32 + 8 GiB at 2^17 x 64 KB) are only held while the
        // pattern has lost parity blocks; the caller has waited for the last call that used them
        for (uint64_t** b : {&d->again, &d->cos_work}) {
            if (*b) (void)hipFree(*b);
            *b = nullptr;
        }
    }
    d->split_ready = false;
    d->split_repair_ready = false;
    d->split_shift = 0;
    d->e = e;
    d->M = M;
    const uint64_t T = e == 1 ? N : NC;  // a power of two >= n - k, the most losses the code tolerates (3k -> 4k, 7k -> 8k: the roots beyond are padding)
    int lgT = e == 1 ? log2k : log2k + e;
    const int leaf_log = lgT >= TREE_LOW + 2 ? TREE_LOW : std::min(LEAF_LOG, lgT), leaf = 1 << leaf_log;
    const bool narrow_tree = lgT + 1 - CHUNK_LOG >= 1, narrow_pattern = log2k + e + 1 - CHUNK_LOG >= 1;  // at least two chunks
    const bool narrow_tree_inv = lgT - CHUNK_LOG >= 1;  // (the products are half as many columns)

    // e > 1: the caller's lost parity blocks by number and the cosets that have one (what fastecc_repair re-encodes)
    auto upload_parity_flags = [&]() -> int {
        if (!d->parity_lost) D61_TRY(hipMalloc((void**)&d->parity_lost, M));
        D61_TRY(hipMemcpyAsync(d->parity_lost, plost.data(), M, hipMemcpyHostToDevice, nullptr));
        d->lost_coset_mask = 0;
        for (uint64_t q = 0; q < M; q++)
            if (plost[q]) d->lost_coset_mask |= 1u << (q / N);
        return FASTECC_OK;
    };
    d->direct = 0;
    const bool direct_sub = !erased_sub.empty();
    const std::vector<uint32_t>& erased_all = erased;
    const std::vector<uint8_t>& state_all = state;
    const uint64_t NC_all = NC;
    if (direct_sub || (!erased.empty() && (int)erased.size() <= std::min(direct_max, DECODE_DIRECT_MAX))) {
        // few losses: a coefficient table, no locator tree and no transform contexts.  Out of memory for its tables is not an
        // error: the transform path below needs none of them.
        const int rc_direct = [&]() -> int {
            const std::vector<uint32_t>& erased = direct_sub ? erased_sub : erased_all;
            const std::vector<uint8_t>& state = direct_sub ? state_sub : state_all;
            const uint64_t NC = direct_sub ? 2 * N : NC_all;  // (the code the path works in)
            if (direct_sub) {
                const int rc = upload_parity_flags();
                if (rc != FASTECC_OK) return rc;
            }
            const int e = (int)erased.size();
            int pad = 1;
            while (pad < e) pad <<= 1;
            const gf61::Elem w = gf61::h_root(NC);
            // nodes: the data rows and the first |lost data| surviving parity blocks; targets: the lost data rows, then the lost parity blocks
            using gf61::Elem;
            const uint64_t Nd = NC / 2;
            std::vector<uint32_t> Rr, Pl, An;
            for (uint32_t u : erased) ((u & 1u) ? Pl : Rr).push_back(u >> 1);
            for (uint64_t q = 0; q < Nd && An.size() < Rr.size(); q++)
                if (state[2 * q + 1] == ST_HELD) An.push_back((uint32_t)q);
            if (An.size() != Rr.size()) return FASTECC_E_INVAL;
            const int ed = (int)Rr.size(), ep = (int)Pl.size();
            auto sub = [](Elem a, Elem b) { return Elem{gf61::h_subp(a.re, b.re), gf61::h_subp(a.im, b.im)}; };
            auto neg = [](Elem a) { return Elem{gf61::h_subp(0, a.re), gf61::h_subp(0, a.im)}; };
            auto mul = [](Elem a, Elem b) { return gf61::h_mul(a, b); };
            const Elem one{1, 0}, two{2, 0}, kk{Nd % P, 0};
            std::vector<Elem> z(e), ya(ed), cst(e), nodew((size_t)ed * pad, Elem{0, 0});
            for (int r = 0; r < ed; r++) z[r] = gf61::h_pow(w, 2ull * Rr[r]), ya[r] = gf61::h_pow(w, 2ull * An[r] + 1);
            for (int t = 0; t < ep; t++) z[ed + t] = gf61::h_pow(w, 2ull * Pl[t] + 1);
            auto A_at = [&](Elem x, int skip) { Elem v = one; for (int a = 0; a < ed; a++) if (a != skip) v = mul(v, sub(x, ya[a])); return v; };
            auto R_at = [&](Elem x, int skip) { Elem v = one; for (int r = 0; r < ed; r++) if (r != skip) v = mul(v, sub(x, z[r])); return v; };
            for (int r = 0; r < ed; r++) cst[r] = neg(mul(A_at(z[r], -1), gf61::h_inv(mul(z[r], R_at(z[r], r)))));  // C_r = -A(x_r) / (x_r R_r(x_r))
            for (int t = ed; t < e; t++) cst[t] = neg(mul(mul(two, A_at(z[t], -1)), gf61::h_inv(mul(kk, R_at(z[t], -1)))));  // c_t = -2 A(y_t) / (k R(y_t))
            for (int a = 0; a < ed; a++) {
                // parity node a (y_a^k - 1 = -2): target r: k A_a(x_r) R(y_a) / (-2 x_r R_r(x_r) A_a(y_a)); target t: A_a(y_t) R(y_a) / (R(y_t) A_a(y_a))
                const Elem Rya = R_at(ya[a], -1), Aaya = A_at(ya[a], a);
                for (int r = 0; r < ed; r++)
                    nodew[(size_t)a * pad + r] = mul(mul(mul(kk, A_at(z[r], a)), Rya), gf61::h_inv(neg(mul(mul(mul(two, z[r]), R_at(z[r], r)), Aaya))));
                for (int t = ed; t < e; t++) nodew[(size_t)a * pad + t] = mul(mul(A_at(z[t], a), Rya), gf61::h_inv(mul(R_at(z[t], -1), Aaya)));
            }
            std::vector<uint64_t> params(6 * DIRECT_MAX, 0);
            std::vector<uint32_t> lists(2 * DIRECT_MAX, 0xFFFFFFFFu);  // [0, MAX) output positions (row << 1 | parity), [MAX, 2 MAX) the parity nodes' rows
            for (int j = 0; j < e; j++) {
                params[2 * j] = z[j].re, params[2 * j + 1] = z[j].im;
                params[4 * DIRECT_MAX + 2 * j] = cst[j].re, params[4 * DIRECT_MAX + 2 * j + 1] = cst[j].im;
                lists[j] = j < ed ? 2u * Rr[j] : 2u * Pl[j - ed] + 1u;
            }
            for (int a = 0; a < ed; a++) params[2 * DIRECT_MAX + 2 * a] = ya[a].re, params[2 * DIRECT_MAX + 2 * a + 1] = ya[a].im, lists[DIRECT_MAX + a] = An[a];
            hipStream_t s0 = nullptr;
            uint64_t* wp = direct_sub ? d->direct_wpow : d->built ? d->wpow : d->direct_wpow;  // (the inner code's table is its own: w_2k, not w_n)
            if (!wp) {
                // the table becomes visible to later calls only once it has been filled
                uint64_t* fresh = nullptr;
                D61_TRY(hipMalloc((void**)&fresh, NC * 16));
                hipLaunchKernelGGL(k_wpow, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, s0, fresh, w.re, w.im, (uint32_t)NC);
                hipError_t e1 = hipGetLastError();
                if (e1 == hipSuccess) e1 = hipStreamSynchronize(s0);
                if (e1 != hipSuccess) {
                    (void)hipFree(fresh);
                    D61_TRY(e1);
                }
                d->direct_wpow = wp = fresh;
            }
            // [k + ed][pad] weights: sized for this pattern's pad (0.5 GiB instead of 16 GiB at k = 2^24 for one lost block), grown on demand
            const uint64_t rows = Nd + (uint64_t)ed;
            const uint64_t coef_elems = (rows + 4) * (uint64_t)pad;  // (k_direct_accumulate fetches the rows of a trip together)
            if (d->direct_coef_elems < coef_elems) {
                if (d->direct_coef) (void)hipFree(d->direct_coef);
                d->direct_coef = nullptr;
                d->direct_coef_elems = 0;
                D61_TRY(hipMalloc((void**)&d->direct_coef, coef_elems * COEF_WORDS * 4));
                d->direct_coef_elems = coef_elems;
            }
            if (!d->direct_inv) D61_TRY(hipMalloc((void**)&d->direct_inv, (6 * DIRECT_MAX + 2 * DIRECT_MAX * DIRECT_MAX) * 8));  // the parameters, the nodes' weights
            if (!d->direct_pos) D61_TRY(hipMalloc((void**)&d->direct_pos, 2 * DIRECT_MAX * 4));
            const uint64_t chunks = (rows + DIRECT_ROWS - 1) / DIRECT_ROWS;
            const uint64_t need = (chunks + DIRECT_SEGS) * pad * elems;
            if (d->direct_partial_elems < need) {
                if (d->direct_partial) (void)hipFree(d->direct_partial);
                d->direct_partial = nullptr;
                d->direct_partial_elems = 0;
                D61_TRY(hipMalloc((void**)&d->direct_partial, need * 16));
                d->direct_partial_elems = need;
            }
            D61_TRY(hipMemcpyAsync(d->direct_pos, lists.data(), 2 * DIRECT_MAX * 4, hipMemcpyHostToDevice, s0));
            D61_TRY(hipMemcpyAsync(d->direct_inv, params.data(), 6 * DIRECT_MAX * 8, hipMemcpyHostToDevice, s0));
            D61_TRY(hipMemsetAsync(d->direct_coef + rows * pad * COEF_WORDS, 0, 4ull * pad * COEF_WORDS * 4, s0));  // (the four spare rows)
            hipLaunchKernelGGL(k_direct_coef, dim3((unsigned)((Nd + 255) / 256)), dim3(256), 0, s0, d->direct_coef, wp, d->direct_inv, (uint32_t)Nd, ed, ep, pad);
            if (ed > 0) {
                D61_TRY(hipMemcpyAsync(d->direct_inv + 6 * DIRECT_MAX, nodew.data(), (size_t)ed * pad * 16, hipMemcpyHostToDevice, s0));
                hipLaunchKernelGGL(k_direct_pack, dim3((unsigned)((ed * pad + 255) / 256)), dim3(256), 0, s0, d->direct_coef + Nd * pad * COEF_WORDS, d->direct_inv + 6 * DIRECT_MAX,
                                   (uint32_t)(ed * pad));
            }
            D61_TRY(hipGetLastError());
            D61_TRY(hipStreamSynchronize(s0));
            d->direct_ed = ed;
            d->direct_rows = rows;
            d->direct = e;
            d->direct_pad = pad;
            d->direct_nc = NC;
            return FASTECC_OK;
        }();
        if (rc_direct == FASTECC_OK) {
            d->ready = true;
            return FASTECC_OK;
        }
        if (rc_direct != FASTECC_E_NOMEM) return rc_direct;
        (void)hipGetLastError();
        d->direct = 0;
    }
    // ---- built once; a failure half way leaves no decoder behind (the next call starts from scratch) ----
    auto build_once = [&]() -> int {
        PhaseTimer pt;
        // only the even (data) positions of this transform are wanted: a 7-level MID here pairs with the 6-level MID of a size-k path,
        // whose DIT passes finish the folded transform (encode_fold); where no such pair of plans exists all 2k outputs are computed
        // (n = 4k: every FOURTH position — the size-4k transform's 7-level MID pairs with the 5-level MID of a size-k path; n = 8k: every fourth
        //  position as well, with a size-2k path: 2k outputs instead of 8k, the data at the even ones)
        int rc = create_transform_mid(&d->transform, log2k + e, elems, FACTOR_INDEX, 7, detail, cap);
        pt.mark("transform path");
        if (rc == FASTECC_OK && log2k >= 6) rc = create_transform_mid(&d->half, e == 3 ? log2k + 1 : log2k, elems, FACTOR_ENCODE, e == 1 ? 6 : 5, detail, cap);
        if (rc == FASTECC_OK && e >= 2 && !(fold_caps(d->transform, d->half) & FOLD_PAIRS)) {
            // the two plans do not pair up at this size: the transform keeps its own choice of MID and computes all n outputs
            destroy(d->transform);
            d->transform = nullptr;
            destroy(d->half);
            d->half = nullptr;
            rc = create_transform_mid(&d->transform, log2k + e, elems, FACTOR_INDEX, 0, detail, cap);
        }
        pt.mark("half path");
        // (of these paths only the stand-alone transform is used; few columns: the upper row bits by a path of CHUNK columns, the rest by k_chunk_dif)
        if (rc == FASTECC_OK) rc = narrow_pattern ? create(&d->narrow_pattern, log2k + e + 1 - CHUNK_LOG, CHUNK, detail, cap) : create(&d->pattern, log2k + e, 2, detail, cap);
        pt.mark("pattern path");
        if (rc != FASTECC_OK) return rc;
        d->tree.assign(lgT, nullptr);
        d->tree_inv.assign(lgT, nullptr);
        for (int k = leaf_log; k < lgT; k++) {
            const bool few = (T >> k) <= NARROW_COLUMNS;
            if (!(narrow_tree && few)) rc = create(&d->tree[k], k + 1, T >> k, detail, cap);
            if (rc == FASTECC_OK && !(narrow_tree_inv && few)) rc = create(&d->tree_inv[k], k + 1, T >> (k + 1), detail, cap);
            if (rc != FASTECC_OK) return rc;
        }
        if (narrow_tree && lgT > leaf_log) {
            rc = create(&d->narrow_tree, lgT + 1 - CHUNK_LOG, CHUNK, detail, cap);
            if (rc == FASTECC_OK && narrow_tree_inv) rc = create(&d->narrow_tree_inv, lgT - CHUNK_LOG, CHUNK, detail, cap);
            if (rc != FASTECC_OK) return rc;
        }
        pt.mark("tree paths");
        d->T = T;
        D61_TRY(hipMalloc((void**)&d->tree_x, 2 * T * 16));
        D61_TRY(hipMalloc((void**)&d->tree_y, 2 * T * 16));
        D61_TRY(hipMalloc((void**)&d->tree_f, 2 * T * 16));
        D61_TRY(hipMalloc((void**)&d->wpow, NC * 16));
        D61_TRY(hipMalloc((void**)&d->roots, T * 16));
        D61_TRY(hipMalloc((void**)&d->lv, NC * 32));
        D61_TRY(hipMalloc((void**)&d->fin, NC * 16));
        D61_TRY(hipMalloc((void**)&d->gout, N * 16));
        D61_TRY(hipMalloc((void**)&d->erased, (T + 1) * 4));  // the list and its counter
        D61_TRY(hipMalloc((void**)&d->state, NC));
        if (e > 1) {
            D61_TRY(hipMalloc((void**)&d->srcmap, NC * 4));
        }
        const gf61::Elem w = gf61::h_root(NC);
        hipLaunchKernelGGL(k_wpow, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, nullptr, d->wpow, w.re, w.im, (uint32_t)NC);
        D61_TRY(hipGetLastError());
        d->built = true;
        pt.mark("buffers, w^u");
        return FASTECC_OK;
    };
    if (!d->built) {
        const int rc = build_once();
        if (rc != FASTECC_OK) {
            destroy_decoder(d);
            *slot = nullptr;
            return rc;
        }
    }
    hipStream_t s0 = nullptr;
    // (the caller has waited for the last decode that used the previous pattern)
    D61_TRY(hipMemcpyAsync(d->state, state.data(), NC, hipMemcpyHostToDevice, s0));
    if (!d->state_real) D61_TRY(hipMalloc((void**)&d->state_real, NC));
    D61_TRY(hipMemcpyAsync(d->state_real, d->state, NC, hipMemcpyDeviceToDevice, s0));
    if (split_shift != 0) {
        hipLaunchKernelGGL(k_mark_unused, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s0, d->state, (uint32_t)N, (1u << split_shift) - 1u);
        D61_TRY(hipGetLastError());
    }
    if (e > 1) {
        D61_TRY(hipMemcpyAsync(d->srcmap, srcmap.data(), NC * 4, hipMemcpyHostToDevice, s0));  // (fixed per code; cheap next to the locator)
        const int rc = upload_parity_flags();
        if (rc != FASTECC_OK) return rc;
    }
    if (split_shift != 0) {
        // the split transform's paths and tables (once; the small transform per shift).  Anything missing — no plan of the needed shape, no memory
        // — leaves the folded 2k-point transform in charge: it decodes the same pattern (the unused parity blocks are unused there as well).
        const int rc_split = [&]() -> int {
            if (!d->splitp) {
                int rc = FASTECC_E_UNSUPPORTED;
                for (int mid : {5, 6, 0}) {  // MID with the addend is leanest at 5 or 6 levels (78 VGPRs; the 7-level one spills)
                    if (d->splitp) destroy(d->splitp);
                    d->splitp = nullptr;
                    rc = create_transform_mid(&d->splitp, log2k, elems, FACTOR_SPLIT, mid, detail, cap);
                    if (rc != FASTECC_OK) return rc;
                    if (split_decode_supported(d->splitp)) break;
                }
                if (!split_decode_supported(d->splitp)) return FASTECC_E_UNSUPPORTED;
            }
            if (!d->split_af) {
                D61_TRY(hipMalloc((void**)&d->split_af, N * 16));
                const int rc = split_addend_factors(d->split_af, log2k, s0);
                if (rc != FASTECC_OK) return rc;
            }
            if (!d->split_work) D61_TRY(hipMalloc((void**)&d->split_work, N * elems * 16));
            if (d->small_rows < (N >> split_shift)) {
                if (d->small_buf) (void)hipFree(d->small_buf);
                d->small_buf = nullptr;
                d->small_rows = 0;
                D61_TRY(hipMalloc((void**)&d->small_buf, (N >> split_shift) * elems * 16));
                d->small_rows = N >> split_shift;
            }
            if (!d->small[split_shift]) {
                const int rc = create(&d->small[split_shift], log2k - split_shift, elems, detail, cap);  // only its stand-alone transform's DIF passes are used
                if (rc != FASTECC_OK) return rc;
            }
            return FASTECC_OK;
        }();
        if (rc_split != FASTECC_OK) {
            (void)hipGetLastError();
            if (rc_split != FASTECC_E_NOMEM && rc_split != FASTECC_E_UNSUPPORTED) return rc_split;
            destroy(d->splitp);
            d->splitp = nullptr;
            d->split_unavailable = true;  // (the pattern stays as it is: the folded transform decodes it)
        } else {
            d->split_shift = split_shift;
            d->split_ready = true;
        }
    }
    if (erased_data == 0 && erased_parity == 0) {
        D61_TRY(hipStreamSynchronize(s0));
        d->ready = true;
        return FASTECC_OK;
    }
    pt_call.mark("set-up, uploads");
    auto grid = [](uint64_t items) { return dim3((unsigned)((items + 255) / 256)); };
    // a transform of 2^log_rows rows of 2^logE columns (2^log_rows <= NC): the upper row bits by `top`, a path of CHUNK columns, the rest inside LDS
    auto narrow = [&](Path* top, int log_rows, int logE, const uint64_t* in, uint64_t* out, bool inverse) -> int {
        const int rc = in == out ? dif_only(top, out, inverse, s0, nullptr) : dif_only_to(top, in, out, inverse, s0, nullptr);
        if (rc != FASTECC_OK) return rc;
        const int logN1 = log_rows + logE - CHUNK_LOG;
        const uint32_t step_n = (uint32_t)(NC >> log_rows), step_n2 = (uint32_t)(NC >> (CHUNK_LOG - logE));
        hipLaunchKernelGGL(inverse ? k_chunk_dif<true> : k_chunk_dif<false>, dim3(1u << logN1), dim3(256), 0, s0, out, d->wpow, logE, logN1, step_n, step_n2,
                           (uint32_t)(NC - 1));
        const hipError_t launched = hipGetLastError();
        return launched == hipSuccess ? FASTECC_OK : fail(detail, cap, launched, "k_chunk_dif");
    };
    D61_TRY(hipMemsetAsync(d->erased + T, 0, 4, s0));
    hipLaunchKernelGGL(k_erased_list, grid((NC + 15) / 16), dim3(256), 0, s0, d->state, (uint32_t)NC, d->erased, d->erased + T);
    hipLaunchKernelGGL(k_roots, grid(T), dim3(256), 0, s0, d->roots, d->erased, d->wpow, (uint32_t)n_erased, (uint32_t)T);
    D61_TRY(hipMemsetAsync(d->tree_x, 0, 2 * T * 16, s0));
    if (leaf_log == TREE_LOW) hipLaunchKernelGGL(k_tree_low<TREE_LOW>, dim3((unsigned)(T >> leaf_log)), dim3(1 << TREE_LOW), 0, s0, d->roots, d->tree_x, (uint32_t)(T >> leaf_log));
    else hipLaunchKernelGGL(k_leaves, dim3((unsigned)(((T >> leaf_log) + 63) / 64)), dim3(64), 0, s0, d->roots, d->tree_x, (uint32_t)leaf, (uint32_t)(T >> leaf_log));
    D61_TRY(hipGetLastError());
    // three 2T-element buffers change roles level by level: a = this level's polynomials [2 deg][m] (rows deg.. zero),
    // b = their transforms and then the next level's polynomials, c = the pairwise products
    uint64_t *a = d->tree_x, *b = d->tree_f, *c = d->tree_y;
    for (int k = leaf_log; k < lgT; k++) {
        const uint64_t deg = 1ull << k, m = T >> k;
        const bool few = d->tree[k] == nullptr;  // few columns: narrow()
        int rc = few ? narrow(d->narrow_tree, k + 1, lgT - k, a, b, false)
                     : dif_only_to(d->tree[k], a, b, false, s0, nullptr);  // all m polynomials at once, a -> b (a survives for the combine step); no reordering pass
        if (rc != FASTECC_OK) return rc;
        const gf61::Elem scale = gf61::h_inv(gf61::Elem{(2 * deg) % P, 0});
        hipLaunchKernelGGL(k_pairs, grid(deg * m), dim3(256), 0, s0, b, c, (uint32_t)m, deg * m, scale.re, scale.im, k + 1);
        D61_TRY(hipGetLastError());
        rc = d->tree_inv[k] == nullptr ? narrow(d->narrow_tree_inv, k + 1, lgT - k - 1, c, c, true) : dif_only(d->tree_inv[k], c, true, s0, nullptr);  // the m/2 products, left in bit-reversed row order
        if (rc != FASTECC_OK) return rc;
        const bool top = k + 1 == lgT;
        const uint64_t rows = top ? 2 * deg : 4 * deg;
        hipLaunchKernelGGL(k_combine, grid(rows * (m / 2)), dim3(256), 0, s0, c, a, b, (uint32_t)deg, (uint32_t)m, rows * (m / 2), top, k + 1);
        D61_TRY(hipGetLastError());
        std::swap(a, b);
    }
    uint64_t* x = a;  // the T lower coefficients of L = x^pad l
    hipLaunchKernelGGL(k_locator_columns, grid(NC), dim3(256), 0, s0, x, d->lv, (uint32_t)T, (uint32_t)NC);
    D61_TRY(hipGetLastError());
    {
        const int rc = d->narrow_pattern ? narrow(d->narrow_pattern, log2k + e, 1, d->lv, d->lv, false)
                                         : dif_only(d->pattern, d->lv, false, s0, nullptr);  // (k_finish reads the values where the DIF passes leave them)
        if (rc != FASTECC_OK) return rc;
    }
    d->gout_all_valid = false;
    if (erased_data != 0 && erased_parity != 0 && e == 1) {  // fastecc_repair can then rebuild everything in one transform (no memory for the table: decode + encode)
        // (with the split the locator counts the unused parity blocks as lost: the table then only feeds gout_par below)
        if (!d->gout_all && hipMalloc((void**)&d->gout_all, NC * 16) != hipSuccess) {
            (void)hipGetLastError();
            d->gout_all = nullptr;
        }
        d->gout_all_valid = d->gout_all != nullptr;
    }
    hipLaunchKernelGGL(k_finish, grid(NC), dim3(256), 0, s0, d->lv, d->state, d->wpow, d->fin, d->gout, (uint32_t)NC, (uint32_t)(T - n_erased),
                       d->gout_all_valid ? d->gout_all : nullptr, e, log2k + e);
    // the inversions: one per position the CALLER lost.  The split's list also holds the parity blocks it leaves aside (their factors are never
    // read: k_gout_par takes those of state_real's lost blocks), so its pattern gets a list of its own, in a tree buffer (all three are free by now)
    const uint32_t* lost_list = d->erased;
    uint64_t lost_count = n_erased;
    if (split_shift != 0) {
        uint32_t* scratch = reinterpret_cast<uint32_t*>(d->tree_y);  // NC positions and the counter: 4 (NC + 1) <= 32 T bytes
        D61_TRY(hipMemsetAsync(scratch + NC, 0, 4, s0));
        hipLaunchKernelGGL(k_erased_list, grid((NC + 15) / 16), dim3(256), 0, s0, d->state_real, (uint32_t)NC, scratch, scratch + NC);
        lost_list = scratch;
        lost_count = erased_data + erased_parity;
    }
    hipLaunchKernelGGL(k_finish_lost, grid(lost_count), dim3(256), 0, s0, d->lv, lost_list, (uint32_t)lost_count, d->wpow, d->gout, (uint32_t)NC,
                       (uint32_t)(T - n_erased), d->gout_all_valid ? d->gout_all : nullptr, e, log2k + e);
    D61_TRY(hipGetLastError());
    if (d->split_ready && d->gout_all_valid) {
        // the second chain's tables: the factor of q~ by position (once) and the lost parity blocks' output factors (this pattern)
        const hipError_t e1 = d->split_pos_odd ? hipSuccess : hipMalloc((void**)&d->split_pos_odd, N * 16);
        const hipError_t e2 = e1 != hipSuccess ? e1 : d->gout_par ? hipSuccess : hipMalloc((void**)&d->gout_par, N * 16);
        if (e2 == hipSuccess) {
            if (!d->split_pos_odd_built) {
                const int rc = split_addend_factors(d->split_pos_odd, log2k, s0, true);
                if (rc != FASTECC_OK) return rc;
                d->split_pos_odd_built = true;
            }
            hipLaunchKernelGGL(k_gout_par, grid(N), dim3(256), 0, s0, d->gout_all, d->state_real, d->gout_par, (uint32_t)N);
            D61_TRY(hipGetLastError());
            d->split_repair_ready = true;
        } else {
            (void)hipGetLastError();  // no memory for the tables: the split decodes, the lost parity is re-encoded
        }
    }
    // (the one-transform repair is the unsplit pattern's.  Also when the split's paths or buffers could NOT be built: the locator's state then still
    //  counts the unused parity blocks as lost, so gout_all is non-zero there and that repair would rewrite up to k (1 - 2^-h) parity blocks the
    //  caller holds — decode + re-encode restores only what state_real says is lost)
    if (d->split_ready || split_shift != 0) d->gout_all_valid = false;
    D61_TRY(hipStreamSynchronize(s0));
    pt_call.mark("this pattern (device)");
    d->ready = true;
    return FASTECC_OK;
}

int decode(Decoder* d, uint64_t* data, uint64_t* parity, Path* rebuild_with, hipStream_t s0, const LaunchHooks* hooks)
{
    if (!d || !d->ready) return FASTECC_E_INVAL;
    char* detail = nullptr;
    const size_t cap = 0;
    const uint32_t elems = (uint32_t)d->elems, col_chunks = (elems + 63) / 64;
    const bool rebuild = rebuild_with != nullptr && d->erased_parity != 0;
    // n = 4k / 8k: the lost parity blocks from the encoder again, on the repaired data; only the cosets with a lost block, only the lost ones written
    auto rebuild_cosets = [&](uint32_t coset_mask, bool first_coset_done) -> int {
        if (cosets_of(rebuild_with) != (1 << d->e) - 1) return FASTECC_E_INVAL;
        if (!d->again) D61_TRY(hipMalloc((void**)&d->again, d->M * d->elems * 16));
        if (encode_cosets_needs_work(rebuild_with) && !d->cos_work) D61_TRY(hipMalloc((void**)&d->cos_work, d->N * d->elems * 16));
        const int rc = encode_cosets(rebuild_with, data, d->again, d->cos_work, s0, hooks, coset_mask);
        if (rc != FASTECC_OK) return rc;
        // (after the inner code's direct path the first coset's blocks are in place and `again` holds nothing for them)
        const uint64_t skip = first_coset_done ? d->N : 0, items = (d->M - skip) * col_chunks;
        hipLaunchKernelGGL(k_restore_map, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s0, d->again + 2 * skip * elems, parity + 2 * skip * elems,
                           d->parity_lost + skip, elems, col_chunks, items);
        D61_TRY(hipGetLastError());
        return FASTECC_OK;
    };
    if (d->direct > 0) {
        if (d->erased_data == 0 && !rebuild) return FASTECC_OK;
        // rows: the k data blocks and as many parity blocks as data blocks are lost; outputs: the lost data blocks first — fastecc_decode stops after them
        const uint32_t rows = (uint32_t)d->direct_rows, chunks = (rows + DIRECT_ROWS - 1) / DIRECT_ROWS;
        const int outputs = rebuild ? d->direct : d->direct_ed;
        const uint64_t items = (uint64_t)chunks * col_chunks;
        const dim3 grid((unsigned)((items + 3) / 4));
#define FASTECC_DIRECT61(EB)                                                                                                                                  \
    hipLaunchKernelGGL(k_direct_accumulate<EB>, dim3(grid.x, (unsigned)((std::min(outputs, d->direct_pad) + EB - 1) / EB)), dim3(256), 0, s0, data, parity,    \
                       d->direct_coef, d->direct_partial, elems, rows, col_chunks, items, (uint32_t)d->direct_pad, (uint32_t)(d->direct_nc / 2),               \
                       d->direct_pos + DIRECT_MAX)
        switch (d->direct_pad) {
            case 1: FASTECC_DIRECT61(1); break;
            case 2: FASTECC_DIRECT61(2); break;
            case 4: FASTECC_DIRECT61(4); break;
            default: FASTECC_DIRECT61(8); break;  // 8, 16, 32: sweeps of 8 outputs
        }
#undef FASTECC_DIRECT61
        D61_TRY(hipGetLastError());
        uint64_t* stage = d->direct_partial + 2ull * (uint64_t)chunks * d->direct_pad * elems;
        hipLaunchKernelGGL(k_direct_reduce1, dim3((elems + 255) / 256, (unsigned)outputs, DIRECT_SEGS), dim3(256), 0, s0, d->direct_partial, stage, elems, chunks,
                           d->direct_pad, outputs);
        hipLaunchKernelGGL(k_direct_reduce2, dim3((elems + 255) / 256, (unsigned)outputs), dim3(256), 0, s0, stage, d->direct_pos, data, parity, elems,
                           d->direct_pad, outputs, rebuild);
        D61_TRY(hipGetLastError());
        // (n = 4k / 8k: that was the (2k,k) code of the data and the first coset; the other cosets' lost blocks are re-encoded)
        if (d->e > 1 && rebuild && (d->lost_coset_mask & ~1u) != 0) return rebuild_cosets(d->lost_coset_mask & ~1u, true);
        return FASTECC_OK;
    }
    if (d->erased_data != 0 && rebuild && d->gout_all_valid) {
        // fastecc_repair in ONE transform: x p'(x) at all 2k positions, the gather in its first tile, the scatter — lost data AND lost parity
        // blocks, each times its factor — in its last; no fold, no second encode
        if (!d->work) D61_TRY(hipMalloc((void**)&d->work, d->NC * d->elems * 16));
        const int rc = encode_ends(d->transform, data, parity, d->fin, d->work, d->gout_all, data, parity, s0, hooks);
        if (rc == FASTECC_OK) return FASTECC_OK;
        if (rc != FASTECC_E_UNSUPPORTED) return rc;
    }
    bool data_done = false;
    if (d->e > 1) {
        // n = 4k / 8k: gather through the position map, x p'(x) on all k << e positions, the data positions (multiples of 2^e) scattered back
        if (d->erased_data != 0) {
            if (!d->work) D61_TRY(hipMalloc((void**)&d->work, d->NC * d->elems * 16));
            // n = 4k: x p'(x) at the data positions only — the way down on all 4k positions, the folding MID tile, the way up on k positions
            // (encode_fold), the gather through the position map in the first tile and the scatter in the last where the plans have such passes
            const int caps = d->half ? fold_caps(d->transform, d->half) : 0;
            // (n = 8k: the folded transform leaves 2k rows, the data at the even ones: k_scatter takes every second)
            const bool folded = (caps & FOLD_PAIRS) != 0, fused_gather = folded && (caps & FOLD_GATHERS), fused_scatter = folded && (caps & FOLD_SCATTERS) && d->e == 2;
            const int rec_shift = d->e - 2;
            uint64_t items = d->NC * col_chunks;
            if (!fused_gather) {
                hipLaunchKernelGGL(k_gather_map, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s0, data, parity, d->work, d->fin, d->srcmap, elems, col_chunks, items);
                D61_TRY(hipGetLastError());
            }
            if (folded) {
                if (!d->rec) D61_TRY(hipMalloc((void**)&d->rec, (d->N << rec_shift) * d->elems * 16));
                FoldEnds ends;
                if (fused_gather) {
                    ends.parity = parity;
                    ends.fin = d->fin;
                    ends.map = d->srcmap;
                }
                if (fused_scatter) {
                    ends.gout = d->gout;
                    ends.data_out = data;
                }
                const int rc = encode_fold(d->transform, d->half, fused_gather ? data : d->work, d->work, d->rec, s0, hooks, &ends);
                if (rc != FASTECC_OK) return rc;
            } else {
                const int rc = encode(d->transform, d->work, d->work, s0, hooks);
                if (rc != FASTECC_OK) return rc;
            }
            if (!fused_scatter) {
                items = d->N * col_chunks;
                hipLaunchKernelGGL(k_scatter, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s0, folded ? d->rec : d->work, data, d->gout, elems, col_chunks, items,
                                   folded ? 1u << rec_shift : 1u << d->e);
                D61_TRY(hipGetLastError());
            }
        }
        if (rebuild) {
            const int rc = rebuild_cosets(d->lost_coset_mask, false);
            if (rc != FASTECC_OK) return rc;
        }
        return FASTECC_OK;
    }
    if (d->erased_data != 0 && d->split_ready) {
        // even / odd split: r~ = DIF of the k >> h parity rows in use (times l), then the data chain — DIF of data * l, g = (2m+k)/2k q~ - 1/2 w^-m r~
        // between the halves of MID, DIT, and only the rebuilt blocks stored, times 1 / (w^2i l'(w^2i)), straight into the data stripe
        const int h = d->split_shift;
        const uint64_t rows = d->N >> h, items = rows * col_chunks;
        hipLaunchKernelGGL(k_split_small_gather, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s0, parity, d->small_buf, d->fin, elems, h, col_chunks, items);
        D61_TRY(hipGetLastError());
        int rc = dif_only(d->small[h], d->small_buf, true, s0, hooks);
        // (repair: the data chain also stores q~, and a second MID + DIT chain turns it and the same r~ into the lost parity blocks)
        bool second = rebuild && d->split_repair_ready;
        if (second && !d->split_q2 && hipMalloc((void**)&d->split_q2, d->N * d->elems * 16) != hipSuccess) {
            (void)hipGetLastError();
            d->split_q2 = nullptr;
            second = false;  // no room for the extra stripe: the lost parity is re-encoded below
        }
        if (rc == FASTECC_OK)
            rc = split_decode(d->splitp, data, d->fin, 2, d->small_buf, h, d->split_af, d->split_work, d->gout, data, s0, hooks, second ? d->split_q2 : nullptr);
        if (rc != FASTECC_OK && rc != FASTECC_E_UNSUPPORTED) return rc;
        data_done = rc == FASTECC_OK;
        if (data_done && second) {
            rc = split_repair_parity(d->splitp, d->split_q2, d->split_pos_odd, d->small_buf, h, d->split_work, d->gout_par, parity, s0, hooks);
            if (rc != FASTECC_OK && rc != FASTECC_E_UNSUPPORTED) return rc;
            if (rc == FASTECC_OK) return FASTECC_OK;  // data and parity are whole again
        }
    }
    if (d->erased_data != 0 && !data_done) {
        if (!d->work) D61_TRY(hipMalloc((void**)&d->work, d->NC * d->elems * 16));
        // x p'(x) at the even (data) positions only where the plans pair up (encode_fold) — then the gather rides in the first DIF tile and the
        // scatter in the last DIT tile where there are such passes — else on all 2k points
        const int caps = d->half ? fold_caps(d->transform, d->half) : 0;
        const bool fused_gather = (caps & FOLD_GATHERS) != 0, fused_scatter = (caps & FOLD_SCATTERS) != 0;
        if (!fused_gather) {
            const uint64_t items = d->NC * col_chunks;
            hipLaunchKernelGGL(k_gather, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s0, data, parity, d->work, d->fin, elems, col_chunks, items);
            D61_TRY(hipGetLastError());
        }
        int rc = FASTECC_E_UNSUPPORTED;
        if (caps & FOLD_PAIRS) {
            if (!d->rec) D61_TRY(hipMalloc((void**)&d->rec, d->N * d->elems * 16));
            FoldEnds ends;
            if (fused_gather) {
                ends.parity = parity;
                ends.fin = d->fin;
            }
            if (fused_scatter) {
                ends.gout = d->gout;
                ends.data_out = data;
            }
            rc = encode_fold(d->transform, d->half, fused_gather ? data : d->work, d->work, d->rec, s0, hooks, &ends);
            if (rc != FASTECC_OK) return rc;
        }
        const bool folded = rc == FASTECC_OK;
        if (!folded) {
            rc = encode(d->transform, d->work, d->work, s0, hooks);
            if (rc != FASTECC_OK) return rc;
        }
        if (!(folded && fused_scatter)) {
            const uint64_t items = d->N * col_chunks;
            hipLaunchKernelGGL(k_scatter, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s0, folded ? d->rec : d->work, data, d->gout, elems, col_chunks, items,
                               folded ? 1u : 2u);
            D61_TRY(hipGetLastError());
        }
    }
    if (rebuild) {
        if (!d->again) D61_TRY(hipMalloc((void**)&d->again, d->N * d->elems * 16));
        const int rc = encode(rebuild_with, data, d->again, s0, hooks);
        if (rc != FASTECC_OK) return rc;
        const uint64_t items = d->N * col_chunks;
        hipLaunchKernelGGL(k_restore, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s0, d->again, parity, d->state_real ? d->state_real : d->state, elems, col_chunks, items);
        D61_TRY(hipGetLastError());
    }
    return FASTECC_OK;
}

// The same for stripes in (pageable) host memory: staged through HBM, synchronous.
int decode_host(Decoder* d, void* data, void* parity, Path* rebuild_with, hipStream_t s0, const LaunchHooks* hooks)
{
    if (!d || !d->ready) return FASTECC_E_INVAL;
    char* detail = nullptr;
    const size_t cap = 0;
    const bool rebuild = rebuild_with != nullptr && d->erased_parity != 0;
    if (d->erased_data == 0 && !rebuild) return FASTECC_OK;
    const size_t stripe = d->N * d->elems * 16, pstripe = (d->e > 1 ? d->M : d->N) * d->elems * 16;
    if (!d->stage) D61_TRY(hipMalloc((void**)&d->stage, stripe + pstripe));
    uint64_t* ddata = d->stage;
    uint64_t* dpar = d->stage + stripe / 8;
    D61_TRY(hipMemcpyAsync(ddata, data, stripe, hipMemcpyHostToDevice, s0));
    D61_TRY(hipMemcpyAsync(dpar, parity, pstripe, hipMemcpyHostToDevice, s0));
    const int rc = decode(d, ddata, dpar, rebuild_with, s0, hooks);
    if (rc != FASTECC_OK) return rc;
    if (d->erased_data != 0) D61_TRY(hipMemcpyAsync(data, ddata, stripe, hipMemcpyDeviceToHost, s0));
    if (rebuild) D61_TRY(hipMemcpyAsync(parity, dpar, pstripe, hipMemcpyDeviceToHost, s0));
    D61_TRY(hipStreamSynchronize(s0));
    return FASTECC_OK;
}

}  // namespace p61
}  // namespace fastecc
