// A wave owns one i2 (one row of each of the q stripes) and a 64*V-word column chunk: the q blocks are in VGPRs, the
// twiddles w_N^(i2*j1) and the constants of the q-point transform are wave-uniform scalars.
//
// The q-point transform (small_dft) costs far fewer products than the q x q matrix it computes:
//   * any odd q: pairing x[i] with x[q-i] — u_i = x_i + x_(q-i), d_i = x_i - x_(q-i) — gives
//         X[j], X[q-j] = x_0 + sum_i u_i C_ij  +-  sum_i d_i S_ij,   C_ij = (w^ij + w^-ij)/2, S_ij = (w^ij - w^-ij)/2
//     i.e. (q-1)^2 / 2 products instead of (q-1)^2 (q = 3: the two-product form of the reference's NTT3, ntt.cpp:25-44);
//   * q = 9 = 3 * 3: Cooley-Tukey inside the registers, 3 + 3 three-point transforms and four twiddles (the structure of the
//     reference's NTT9, ntt.cpp:75-146): 16 products instead of 64;
//   * q = 15 = 3 * 5: the prime-factor map (NTT.md:43-46 "PFA"): input index 5 i1 + 3 i2, output index 10 j1 + 6 j2 (mod 15)
//     turn the transform into five 3-point and three 5-point transforms with NO twiddles in between: 34 products instead of 196.
// Products per q words including the q - 1 twiddles towards the power-of-two part: 4 / 12 / 24 / 24 / 84 / 48 for
// q = 3 / 5 / 7 / 9 / 13 / 15 (the matrix form: 6 / 20 / 42 / 72 / 156 / 210).  All index maps are compile-time constants of
// fully unrolled loops: "permutations" are register renaming.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "gf.hpp"
#include "kernels.hpp"
#include "ntt_device.hpp"

namespace fastecc {

namespace {

template <int V> __device__ __forceinline__ void vadd(uint32_t (&r)[V], const uint32_t (&a)[V], const uint32_t (&b)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) r[v] = gf::add(a[v], b[v]);
}
template <int V> __device__ __forceinline__ void vsub(uint32_t (&r)[V], const uint32_t (&a)[V], const uint32_t (&b)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) r[v] = gf::sub(a[v], b[v]);
}
template <int V> __device__ __forceinline__ void vmul(uint32_t (&r)[V], const uint32_t (&a)[V], uint32_t w)
{
#pragma unroll
    for (int v = 0; v < V; ++v) r[v] = gf::mul_mont(a[v], w);
}
template <int V> __device__ __forceinline__ void vmadd(uint32_t (&r)[V], const uint32_t (&a)[V], uint32_t w)  // r += a * w
{
#pragma unroll
    for (int v = 0; v < V; ++v) r[v] = gf::add(r[v], gf::mul_mont(a[v], w));
}

// In-place transform of the H2*2+1 = Q values *p[0..Q-1] (pointers to registers: the callers pick the slots), by the
// symmetric form above.  tab: C[i][j] at (i-1)*H2 + (j-1), S[i][j] at H2*H2 + the same, i, j = 1..H2.  X[j] lands in *p[j].
template <int Q, int V, typename Slots>
__device__ __forceinline__ void sym_dft(const Slots& p, const_u32_ptr tab)
{
    constexpr int H2 = (Q - 1) / 2;
    uint32_t u[H2][V], d[H2][V], x0[V];
#pragma unroll
    for (int v = 0; v < V; ++v) x0[v] = (*p[0])[v];
#pragma unroll
    for (int i = 0; i < H2; ++i) {
        vadd<V>(u[i], *p[i + 1], *p[Q - 1 - i]);
        vsub<V>(d[i], *p[i + 1], *p[Q - 1 - i]);
    }
#pragma unroll
    for (int i = 0; i < H2; ++i) vadd<V>(*p[0], *p[0], u[i]);
#pragma unroll
    for (int j = 0; j < H2; ++j) {
        uint32_t a[V], b[V];
        vmul<V>(b, d[0], tab[H2 * H2 + j]);
#pragma unroll
        for (int v = 0; v < V; ++v) a[v] = x0[v];
        vmadd<V>(a, u[0], tab[j]);
#pragma unroll
        for (int i = 1; i < H2; ++i) {
            vmadd<V>(a, u[i], tab[i * H2 + j]);
            vmadd<V>(b, d[i], tab[H2 * H2 + i * H2 + j]);
        }
        vadd<V>(*p[j + 1], a, b);
        vsub<V>(*p[Q - 1 - j], a, b);
    }
}

// slot of x[] that holds X[j] after small_dft
template <int Q> constexpr int out_slot(int j)
{
    if (Q == 9) return 3 * (j % 3) + j / 3;                 // j = j1 + 3 j2 sits in slot 3 j1 + j2
    if (Q == 15) return (5 * (j % 3) + 3 * (j % 5)) % 15;   // j = 10 j1 + 6 j2 (j1 = j mod 3, j2 = j mod 5) sits in slot 5 j1 + 3 j2
    return j;
}

template <int Q, int V>
__device__ __forceinline__ void small_dft(uint32_t (&x)[Q][V], const_u32_ptr tab)
{
    if constexpr (Q == 9) {
        // tab: C3, S3 (root w^3), then w^(i2*j1) for (i2, j1) = (1,1), (1,2), (2,1), (2,2)
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2) {
            uint32_t(*const p[3])[V] = {&x[i2], &x[3 + i2], &x[6 + i2]};
            sym_dft<3, V>(p, tab);
        }
        vmul<V>(x[3 + 1], x[3 + 1], tab[2]);
        vmul<V>(x[6 + 1], x[6 + 1], tab[3]);
        vmul<V>(x[3 + 2], x[3 + 2], tab[4]);
        vmul<V>(x[6 + 2], x[6 + 2], tab[5]);
#pragma unroll
        for (int j1 = 0; j1 < 3; ++j1) {
            uint32_t(*const p[3])[V] = {&x[3 * j1], &x[3 * j1 + 1], &x[3 * j1 + 2]};
            sym_dft<3, V>(p, tab);
        }
    } else if constexpr (Q == 15) {
        // tab: C3, S3 (root w^5), then the 2 x 2 C and S tables of the 5-point transform (root w^3)
#pragma unroll
        for (int i2 = 0; i2 < 5; ++i2) {
            uint32_t(*const p[3])[V] = {&x[(3 * i2) % 15], &x[(5 + 3 * i2) % 15], &x[(10 + 3 * i2) % 15]};
            sym_dft<3, V>(p, tab);
        }
#pragma unroll
        for (int j1 = 0; j1 < 3; ++j1) {
            uint32_t(*const p[5])[V] = {&x[(5 * j1) % 15], &x[(5 * j1 + 3) % 15], &x[(5 * j1 + 6) % 15], &x[(5 * j1 + 9) % 15], &x[(5 * j1 + 12) % 15]};
            sym_dft<5, V>(p, tab + 2);
        }
    } else {
        uint32_t(*p[Q])[V];
#pragma unroll
        for (int i = 0; i < Q; ++i) p[i] = &x[i];
        sym_dft<Q, V>(p, tab);
    }
}

}  // namespace

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
    const_u32_ptr dft = as_constant(a.dft);                       // constants of the Q-point transform (radix_dft_table)

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
        for (int j = 1; j < Q; ++j) vmul<V>(x[j], x[j], tw[j - 1]);
    }
    small_dft<Q, V>(x, dft);
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        uint32_t(&y)[V] = x[out_slot<Q>(j)];
        if constexpr (!DIT) {
            if (j > 0) vmul<V>(y, y, tw[j - 1]);
        }
        const uint32_t row = (uint32_t)j * a.M + i2;
        if (a.out_rows == 0 || row < a.out_rows) store_vec<V>(a.out + (size_t)row * a.ld + col, y);
    }
}

// Host: the constants small_dft<q> reads, for the primitive q-th root wq (forward or inverse), Montgomery form.
std::vector<uint32_t> radix_dft_table(int q, uint32_t wq)
{
    const uint32_t inv2 = (uint32_t)((gf::P + 1ull) / 2);
    auto sym = [&](std::vector<uint32_t>& t, int n, uint32_t w) {  // C then S tables of the n-point transform with root w
        const int h2 = (n - 1) / 2;
        const size_t base = t.size();
        t.resize(base + 2 * (size_t)h2 * h2);
        const uint32_t wi = gf::h_inv(w);
        for (int i = 1; i <= h2; i++)
            for (int j = 1; j <= h2; j++) {
                const uint32_t a = gf::h_pow(w, (uint64_t)i * j), b = gf::h_pow(wi, (uint64_t)i * j);
                const uint32_t sum = (uint32_t)(((uint64_t)a + b) % gf::P), dif = (uint32_t)(((uint64_t)a + gf::P - b) % gf::P);
                t[base + (size_t)(i - 1) * h2 + (j - 1)] = gf::h_to_mont(gf::h_mul(sum, inv2));
                t[base + (size_t)h2 * h2 + (size_t)(i - 1) * h2 + (j - 1)] = gf::h_to_mont(gf::h_mul(dif, inv2));
            }
    };
    std::vector<uint32_t> t;
    if (q == 9) {
        sym(t, 3, gf::h_pow(wq, 3));
        for (int i2 = 1; i2 <= 2; i2++)
            for (int j1 = 1; j1 <= 2; j1++) t.push_back(gf::h_to_mont(gf::h_pow(wq, (uint64_t)i2 * j1)));
        // order read by the kernel: slots 3+1 (i2=1,j1=1), 6+1 (i2=1,j1=2), 3+2 (i2=2,j1=1), 6+2 (i2=2,j1=2)
    } else if (q == 15) {
        sym(t, 3, gf::h_pow(wq, 5));
        sym(t, 5, gf::h_pow(wq, 3));
    } else {
        sym(t, q, wq);
    }
    return t;
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
