// mixed_kernels.hip — the odd-radix level of a transform of order q * 2^m, q in {3, 5, 7, 9, 13, 15}.
//
// The reference's roadmap for block counts that are not powers of two (NTT.md:43-46, README.md:175: "PFA NTT as well
// as NTT kernels of orders 3,5,7,9,13, since 0xFFF00000 = 2^20*3*3*5*7*13"; its codelets NTT3 / NTT9, ntt.cpp:25-146,
// are never reached by its drivers).  Here a transform of order N = q * M, M = 2^m, is Cooley-Tukey with the odd factor
// OUTERMOST, so that everything between the two odd-radix passes is the power-of-two pipeline this library already
// has, run on q stripes of M blocks stored back to back:
//
//   way down (inverse roots, decimation in frequency), block i = i1*M + i2:
//       y[j1*M + i2] = ( sum_i1 x[i1*M + i2] * w_q^-(i1*j1) ) * w_N^-(i2*j1)           <- radix_kernel<Q, false>
//       then q independent size-M transforms of the blocks [j1*M, (j1+1)*M)             <- ntt_tile_kernel / ntt_pass_kernel
//   per-block factor: position j1*M + r holds coefficient q*bitrev_m(r) + j1            <- the MID pass, table by position
//   way up (forward roots, decimation in time): the q size-M transforms, then
//       X[t1*M + t] = sum_j1 ( z[j1*M + t] * w_N^(t*j1) ) * w_q^(j1*t1)                 <- radix_kernel<Q, true>
//
// A wave owns one i2 (one row of each of the q stripes) and a 64*V-word column chunk: the q blocks are in VGPRs, the
// twiddles w_N^(i2*j1) and the q x q matrix w_q^(i*j) are wave-uniform scalars.  The odd-order DFT is the plain matrix
// product: (q-1)^2 products per q words — 6 VALU instructions each — stay below the HBM time of the pass for q <= 7
// and about match it for q = 9; q = 13 and 15 (the remaining small divisors of p - 1 = 2^20 3^2 5 7 13) are VALU-bound by
// about 2x and exist for completeness.  NTT3's (ntt.cpp:25-44) two-product form only trades products for additions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf.hpp"
#include "kernels.hpp"
#include "ntt_device.hpp"

namespace fastecc {

template <int Q, bool DIT, int V>
__global__ __launch_bounds__(256) void radix_kernel(const RadixArgs a)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t item = (uint64_t)blockIdx.x * 4u + wave;
    if (item >= a.items) return;  // wave-uniform
    const uint32_t cc = (uint32_t)(item % a.col_chunks);
    const uint32_t i2 = (uint32_t)(item / a.col_chunks);
    const uint32_t col = (cc * 64u + lane) * V;
    if (col >= a.S) return;
    const_u32_ptr tw = as_constant(a.tw) + (size_t)i2 * (Q - 1);  // w_N^(+-i2*j), j = 1..Q-1, Montgomery form
    const_u32_ptr dft = as_constant(a.dft);                       // w_q^(+-i*j), Q x Q, Montgomery form

    uint32_t x[Q][V];
#pragma unroll
    for (int i = 0; i < Q; ++i) {
        const uint32_t row = (uint32_t)i * a.M + i2;
        if (a.in_rows == 0 || row < a.in_rows) {
            load_vec<V>(x[i], a.in + (size_t)row * a.ld + col);
        } else {  // zero-extended data: blocks from in_rows on do not exist
#pragma unroll
            for (int v = 0; v < V; ++v) x[i][v] = 0;
        }
    }
    if constexpr (DIT) {
#pragma unroll
        for (int j = 1; j < Q; ++j) {
            const uint32_t w = tw[j - 1];
#pragma unroll
            for (int v = 0; v < V; ++v) x[j][v] = gf::mul_mont(x[j][v], w);
        }
    }
    uint32_t y[Q][V];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
#pragma unroll
        for (int v = 0; v < V; ++v) y[j][v] = x[0][v];
#pragma unroll
        for (int i = 1; i < Q; ++i) {
            if (j == 0) {
#pragma unroll
                for (int v = 0; v < V; ++v) y[j][v] = gf::add(y[j][v], x[i][v]);
            } else {
                const uint32_t w = dft[i * Q + j];
#pragma unroll
                for (int v = 0; v < V; ++v) y[j][v] = gf::add(y[j][v], gf::mul_mont(x[i][v], w));
            }
        }
    }
    if constexpr (!DIT) {
#pragma unroll
        for (int j = 1; j < Q; ++j) {
            const uint32_t w = tw[j - 1];
#pragma unroll
            for (int v = 0; v < V; ++v) y[j][v] = gf::mul_mont(y[j][v], w);
        }
    }
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const uint32_t row = (uint32_t)j * a.M + i2;
        if (a.out_rows == 0 || row < a.out_rows) store_vec<V>(a.out + (size_t)row * a.ld + col, y[j]);
    }
}

template <int Q, int V>
static hipError_t launch_q(bool dit, const RadixArgs& a, dim3 grid, hipStream_t st)
{
    if (dit) hipLaunchKernelGGL((radix_kernel<Q, true, V>), grid, dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((radix_kernel<Q, false, V>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

template <int V>
static hipError_t launch_v(int q, bool dit, const RadixArgs& a, dim3 grid, hipStream_t st)
{
    switch (q) {
        case 3: return launch_q<3, V>(dit, a, grid, st);
        case 5: return launch_q<5, V>(dit, a, grid, st);
        case 7: return launch_q<7, V>(dit, a, grid, st);
        case 9: return launch_q<9, V>(dit, a, grid, st);
        case 13:
            if constexpr (V == 1) return launch_q<13, 1>(dit, a, grid, st);
            else return hipErrorInvalidValue;
        case 15:
            if constexpr (V == 1) return launch_q<15, 1>(dit, a, grid, st);
            else return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

bool radix_supported(int q) { return q == 3 || q == 5 || q == 7 || q == 9 || q == 13 || q == 15; }

hipError_t launch_radix(int q, bool dit, int vec, RadixArgs a, hipStream_t st)
{
    if (!radix_supported(q) || a.M == 0) return hipErrorInvalidValue;
    if (q == 9 && vec == 4) vec = 2;  // 9 blocks of 4 words per lane (twice: in and out) do not fit the register budget
    if (q > 9) vec = 1;
    a.col_chunks = (a.S + 64u * vec - 1u) / (64u * vec);
    a.items = (uint64_t)a.col_chunks * a.M;
    const uint64_t blocks = (a.items + 3u) / 4u;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    switch (vec) {
        case 1: return launch_v<1>(q, dit, a, grid, st);
        case 2: return launch_v<2>(q, dit, a, grid, st);
        case 4: return launch_v<4>(q, dit, a, grid, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fastecc
