// kernels.hip — gfx950 kernels of the NTT Reed-Solomon encode path.
//
// Data: a stripe is X[N][S] uint32, block-major (RS.cpp:28-33).  The transform runs down the block
// index; the S word columns are independent (ntt.cpp:348-350).  Mapping used by every kernel here:
//
//      lane  <->  V adjacent words of one block          (coalesced: a wave reads 64*V*4 contiguous bytes)
//      wave  <->  one column chunk of R = 2^r blocks     (the butterfly operands live in VGPRs)
//      twiddles are the same for all 64 lanes            (fetched with scalar loads into SGPRs)
//
// A "pass" executes r consecutive radix-2 levels entirely in registers: the wave loads its R block
// segments, runs r butterfly levels, stores them back.  Three flavours:
//
//   DIF  decimation in frequency, strides 2^(s+r-1) .. 2^s  : (a,b) -> (a+b, (a-b)*w)       natural -> bit-reversed
//   DIT  decimation in time,      strides 2^s .. 2^(s+r-1)  : (a,b) -> (a+b*w, a-b*w)       bit-reversed -> natural
//   MID  s = 0: DIF levels, multiply block p by D[bitrev(p)] (D_i = w_2N^i/N, RS.cpp:51-59), DIT levels
//
// The reference's radix-2 butterfly is ntt.cpp:16-22 / 259-281; its bit-reversal (ntt.cpp:292-309) and
// pointer transposes (ntt.cpp:322-341) become index arithmetic: encode = DIF(inverse roots) over all
// levels, D, DIT(forward roots) over all levels, and no permutation pass exists at all.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf.hpp"
#include "kernels.hpp"
#include "ntt_device.hpp"

namespace fastecc {

// One register pass.  Work item = (block group g, column chunk cc); a wave owns one work item.
template <int LOGR, int V, int MODE>
__global__ __launch_bounds__(256) void ntt_pass_kernel(const PassArgs a)
{
    constexpr int R = 1 << LOGR;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= a.items) return;  // wave-uniform
    const uint32_t cc = (uint32_t)(item % a.col_chunks);
    const uint32_t g = (uint32_t)(item / a.col_chunks);
    const uint32_t col = (cc * 64u + lane) * V;
    const bool live = col < a.S;  // S % V == 0 is guaranteed by the launcher

    const int s = MODE == MODE_MID ? 0 : a.s;
    const uint32_t lo = g & ((1u << s) - 1u);
    const uint32_t hi = g >> s;
    const uint32_t base = (hi << (s + LOGR)) + lo;  // first block of this group

    uint32_t x[R][V];
    if (MODE == MODE_DIF && a.row_factor != nullptr) {
        // the decoder's gather fused into its first pass: codeword position u -> data or parity block u/2, times l(w^u)
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const uint32_t u = base + ((uint32_t)j << s);
            const uint32_t f = as_constant(a.row_factor)[u];
            if (live && f != 0) {
                load_vec<V>(x[j], ((u & 1u) ? a.in_odd : a.in) + (size_t)(u >> 1) * a.ld + col);
#pragma unroll
                for (int v = 0; v < V; ++v) x[j][v] = gf::mul_mont(x[j][v], f);
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) x[j][v] = 0;
            }
        }
    } else if (live && a.in_rows != 0) {
        // zero-extended stripe: blocks from in_rows on do not exist and read as zero
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const uint32_t blk = base + ((uint32_t)j << s);
            if (blk < a.in_rows) {
                load_vec<V>(x[j], a.in + (size_t)blk * a.ld + col);
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) x[j][v] = 0;
            }
        }
    } else if (live) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const uint32_t* row = a.in + (size_t)(base + ((uint32_t)j << s)) * a.ld;
            load_vec<V>(x[j], row + col);
        }
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int v = 0; v < V; ++v) x[j][v] = 0;
    }

    if constexpr (MODE == MODE_DIF) {
        if (a.s == 0) dif_levels<LOGR, V, true>(x, a.tw_dif, 0u, 0);
        else          dif_levels<LOGR, V, false>(x, a.tw_dif, lo, s);
    } else if constexpr (MODE == MODE_DIT) {
        if (a.s == 0) dit_levels<LOGR, V, true>(x, a.tw_dit, 0u, 0);
        else          dit_levels<LOGR, V, false>(x, a.tw_dit, lo, s);
    } else {
        dif_levels<LOGR, V, true>(x, a.tw_dif, 0u, 0);
        // position p = hi*R + j holds coefficient bitrev_n(p); a.dscale is stored in position order
        // the table repeats per stripe of a batch unless it covers the whole batch (mixed-radix transforms)
        const_u32_ptr d = as_constant(a.dscale) + (size_t)(a.dscale_whole ? hi : (hi & ((1u << (a.n - LOGR)) - 1u))) * R;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const uint32_t f = d[j];
#pragma unroll
            for (int v = 0; v < V; ++v) x[j][v] = gf::mul_mont(x[j][v], f);
        }
        dit_levels<LOGR, V, true>(x, a.tw_dit, 0u, 0);
    }

    if (live) {
        if (MODE == MODE_MID && a.fold > 0) {
            // fewer parity than data blocks: parity block j of the (N + N/2^fold, N) code is parity block j * 2^fold of
            // the (2N, N) code, i.e. the output positions that are multiples of 2^fold, stored compactly
            const uint32_t drop = (1u << a.fold) - 1u;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint32_t blk = (base + j) >> a.fold;
                if ((j & drop) == 0 && (a.out_rows == 0 || blk < a.out_rows)) store_vec<V>(a.out + (size_t)blk * a.ld + col, x[j]);
            }
            return;
        }
        if (a.out_rows != 0) {  // truncated result stripe
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const uint32_t blk = base + ((uint32_t)j << s);
                if (blk < a.out_rows) store_vec<V>(a.out + (size_t)blk * a.ld + col, x[j]);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            uint32_t* row = a.out + (size_t)(base + ((uint32_t)j << s)) * a.ld;
            store_vec<V>(row + col, x[j]);
        }
    }
}

// Swap block j with block bitrev(j) (the data movement the reference avoids by permuting pointers,
// ntt.cpp:292-309; only the stand-alone fastecc_ntt needs it).
template <int V>
__global__ __launch_bounds__(256) void bitrev_rows_kernel(uint32_t* data, uint32_t S, int n, uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t j = (uint32_t)(item / col_chunks);
    const uint32_t rj = bitrev(j, n);
    if (rj <= j) return;
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    uint32_t* pa = data + (size_t)j * S + col;
    uint32_t* pb = data + (size_t)rj * S + col;
    uint32_t va[V], vb[V];
    load_vec<V>(va, pa);
    load_vec<V>(vb, pb);
    store_vec<V>(pa, vb);
    store_vec<V>(pb, va);
}

// block i *= factor[i] (Montgomery-form factors) — RS.cpp:52-59 / ntt.cpp:421-431 as a stand-alone kernel.
template <int V>
__global__ __launch_bounds__(256) void scale_rows_kernel(uint32_t* data, const uint32_t* __restrict__ factor, uint32_t S,
                                                         uint32_t col_chunks, uint64_t items)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= items) return;
    const uint32_t cc = (uint32_t)(item % col_chunks);
    const uint32_t i = (uint32_t)(item / col_chunks);
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= S) return;
    const uint32_t f = factor[i];
    uint32_t* p = data + (size_t)i * S + col;
    uint32_t x[V];
    load_vec<V>(x, p);
#pragma unroll
    for (int v = 0; v < V; ++v) x[v] = gf::mul_mont(x[v], f);
    store_vec<V>(p, x);
}

__global__ __launch_bounds__(256) void gf_binary_kernel(int op, const uint32_t* __restrict__ x, const uint32_t* __restrict__ y,
                                                        uint32_t* __restrict__ out, uint64_t count)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = x[i], b = y[i];
        uint32_t r;
        if (op == 0) r = gf::add(a, b);
        else if (op == 1) r = gf::sub(a, b);
        else if (op == 2) r = gf::mul(a, b);
        else r = gf::mul_mont(a, gf::mul(b, gf::MONT_ONE));  // op 3: the Montgomery path, b lifted on the fly
        out[i] = r;
    }
}

// Count words that are not field elements (>= p).  The encode requires every input word < p
// (README.md:160-162 of the reference; GF_Add/GF_Sub are only defined on [0,p), GF(p).cpp:37-48).
__global__ __launch_bounds__(256) void count_out_of_range_kernel(const uint4* __restrict__ x4, const uint32_t* __restrict__ tail,
                                                                  uint64_t n4, uint32_t ntail, unsigned long long* __restrict__ bad)
{
    unsigned int local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = x4[i];
        local += (v.x >= gf::P) + (v.y >= gf::P) + (v.z >= gf::P) + (v.w >= gf::P);
    }
    if (blockIdx.x == 0 && threadIdx.x < ntail) local += tail[threadIdx.x] >= gf::P;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63u) == 0 && local) atomicAdd(bad, (unsigned long long)local);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

template <int LOGR, int V>
static hipError_t launch_pass_rv(int mode, const PassArgs& a, dim3 grid, hipStream_t st)
{
    switch (mode) {
        case MODE_DIF: hipLaunchKernelGGL((ntt_pass_kernel<LOGR, V, MODE_DIF>), grid, dim3(256), 0, st, a); break;
        case MODE_DIT: hipLaunchKernelGGL((ntt_pass_kernel<LOGR, V, MODE_DIT>), grid, dim3(256), 0, st, a); break;
        default:       hipLaunchKernelGGL((ntt_pass_kernel<LOGR, V, MODE_MID>), grid, dim3(256), 0, st, a); break;
    }
    return hipGetLastError();
}

template <int V>
static hipError_t launch_pass_v(int logr, int mode, const PassArgs& a, dim3 grid, hipStream_t st)
{
    switch (logr) {
        case 1: return launch_pass_rv<1, V>(mode, a, grid, st);
        case 2: return launch_pass_rv<2, V>(mode, a, grid, st);
        case 3: return launch_pass_rv<3, V>(mode, a, grid, st);
        case 4: return launch_pass_rv<4, V>(mode, a, grid, st);
        case 5: return launch_pass_rv<5, V>(mode, a, grid, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_pass(int logr, int vec, int mode, PassArgs a, hipStream_t st)
{
    a.col_chunks = (a.S + 64u * vec - 1u) / (64u * vec);
    a.items = ((uint64_t)a.col_chunks * (a.batch > 1 ? a.batch : 1u)) << (a.n - logr);
    const uint64_t blocks = (a.items + 3u) / 4u;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    switch (vec) {
        case 1: return launch_pass_v<1>(logr, mode, a, grid, st);
        case 2: return launch_pass_v<2>(logr, mode, a, grid, st);
        case 4: return launch_pass_v<4>(logr, mode, a, grid, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_bitrev_rows(uint32_t* data, uint32_t S, int n, int vec, hipStream_t st)
{
    const uint32_t col_chunks = (S + 64u * vec - 1u) / (64u * vec);
    const uint64_t items = (uint64_t)col_chunks << n;
    const dim3 grid((unsigned)((items + 3u) / 4u));
    switch (vec) {
        case 1: hipLaunchKernelGGL((bitrev_rows_kernel<1>), grid, dim3(256), 0, st, data, S, n, col_chunks, items); break;
        case 2: hipLaunchKernelGGL((bitrev_rows_kernel<2>), grid, dim3(256), 0, st, data, S, n, col_chunks, items); break;
        default: hipLaunchKernelGGL((bitrev_rows_kernel<4>), grid, dim3(256), 0, st, data, S, n, col_chunks, items); break;
    }
    return hipGetLastError();
}

hipError_t launch_scale_rows(uint32_t* data, const uint32_t* factor, uint32_t S, uint64_t rows, int vec, hipStream_t st)
{
    const uint32_t col_chunks = (S + 64u * vec - 1u) / (64u * vec);
    const uint64_t items = (uint64_t)col_chunks * rows;
    const dim3 grid((unsigned)((items + 3u) / 4u));
    switch (vec) {
        case 1: hipLaunchKernelGGL((scale_rows_kernel<1>), grid, dim3(256), 0, st, data, factor, S, col_chunks, items); break;
        case 2: hipLaunchKernelGGL((scale_rows_kernel<2>), grid, dim3(256), 0, st, data, factor, S, col_chunks, items); break;
        default: hipLaunchKernelGGL((scale_rows_kernel<4>), grid, dim3(256), 0, st, data, factor, S, col_chunks, items); break;
    }
    return hipGetLastError();
}

hipError_t launch_count_out_of_range(const uint32_t* x, uint64_t count, unsigned long long* bad, hipStream_t st)
{
    // x is 4-byte aligned; peel words up to a 16-byte boundary on the host side of the call (api.hip)
    const uint64_t n4 = count / 4;
    const uint32_t ntail = (uint32_t)(count % 4);
    uint64_t blocks = (n4 + 255u) / 256u;
    if (blocks > 4096u) blocks = 4096u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(count_out_of_range_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const uint4*>(x), x + n4 * 4, n4, ntail,
                       bad);
    return hipGetLastError();
}

hipError_t launch_gf_binary(int op, const uint32_t* x, const uint32_t* y, uint32_t* out, uint64_t count, hipStream_t st)
{
    uint64_t blocks = (count + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(gf_binary_kernel, dim3((unsigned)blocks), dim3(256), 0, st, op, x, y, out, count);
    return hipGetLastError();
}

void preload_pass_kernels()
{
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(ntt_pass_kernel<5, 1, MODE_DIF>)) != hipSuccess) (void)hipGetLastError();  // (speed only)
}

}  // namespace fastecc
