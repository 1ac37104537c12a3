// mixed_device.hpp — device code of the odd-radix level of a transform of order q * 2^m (included by mixed_kernels.hip: q <= 15, and
// mixed_kernels_pfa.hip: the composite q = 21 ... 117 of the prime-factor map).
//
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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <vector>

#include "gf.hpp"
#include "kernels.hpp"
#include "ntt_device.hpp"

namespace fastecc {

// f(integral_constant<int, I>) for I = BEGIN .. END-1, as straight-line code: the stripe loops of the fused kernel are too large for
// "#pragma unroll" (its size threshold) — a loop left rolled indexes the register array dynamically and sends it to scratch memory.
template <int BEGIN, int END, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (BEGIN < END) {
        f(std::integral_constant<int, BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

// Prime-factor splits (NTT.md:43-46 "PFA"): q = QA * QB with coprime factors, QA = pfa_first(q); 0 for the q that are not split this way.
// With input index QB i1 + QA i2 and output index j = (j mod QA, j mod QB) the q-point transform is QB transforms of QA points followed by
// QA transforms of QB points with NO twiddles in between (w_q^QB and w_q^QA are the roots of the two); the factors may be composite
// themselves (45 = 9 * 5 with the Cooley-Tukey 9, 105 = 15 * 7 with the prime-factor 15).
constexpr int pfa_first(int q)
{
    return q == 15 ? 3 : q == 21 ? 3 : q == 35 ? 5 : q == 39 ? 3 : q == 45 ? 9 : q == 63 ? 9 : q == 65 ? 5 : q == 91 ? 7 : q == 105 ? 15 : q == 117 ? 9 : 0;
}
// words of constants the q-point transform reads (radix_dft_table)
constexpr int dft_table_words(int q)
{
    if (q == 2) return 1;
    if (q == 9) return 2 + 4;
    if (pfa_first(q)) return dft_table_words(pfa_first(q)) + dft_table_words(q / pfa_first(q));
    return 2 * ((q - 1) / 2) * ((q - 1) / 2);
}
// slot of x[] that holds X[j] after the in-place q-point transform
constexpr int out_slot_of(int q, int j)
{
    if (q == 9) return 3 * (j % 3) + j / 3;  // j = j1 + 3 j2 sits in slot 3 j1 + j2
    if (pfa_first(q)) {
        const int qa = pfa_first(q), qb = q / qa;
        return (qb * out_slot_of(qa, j % qa) + qa * out_slot_of(qb, j % qb)) % q;  // 15: j = 10 j1 + 6 j2 sits in slot 5 j1 + 3 j2
    }
    return j;
}

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
    // C and S are symmetric (C_ij = C_ji), so output j reads ROW j of the two tables: H2 consecutive constants each, one wide scalar load.
    // Large q (13: 72 constants) fetch them row by row, two rows in flight: the row offset goes through an empty asm, which makes the
    // address opaque — the loads of row j + 1 can be neither hoisted above that point nor merged with the neighbouring rows' (the compiler
    // had fused the loads of C rows 0-2 into one early request and parked them in spilled SGPRs: 45 spills for q = 13) — and the set about
    // to be used is pinned BEFORE the next request (scalar loads return out of order: a wait after a request waits for it too).
    if constexpr (Q >= 11) {
        uint32_t cs[2][2 * H2];  // (a third row in flight measured the same and costs 12 SGPRs)
        auto fetch = [&](uint32_t (&t)[2 * H2], int j) {
            uint32_t o = (uint32_t)(j * H2);
            asm volatile("" : "+s"(o));
            const_u32_ptr row = tab + o;  // (tab[o + i] would be 2 H2 separate loads, each with its own 64-bit address: o + i may wrap in 32 bits)
#pragma unroll
            for (int i = 0; i < H2; ++i) t[i] = row[i], t[H2 + i] = row[H2 * H2 + i];
        };
        fetch(cs[0], 0);
#pragma unroll
        for (int j = 0; j < H2; ++j) {
            uint32_t(&c)[2 * H2] = cs[j & 1];
#pragma unroll
            for (int i = 0; i < 2 * H2; ++i) asm volatile("" : "+s"(c[i]));
            if (j + 1 < H2) fetch(cs[(j + 1) & 1], j + 1);
            __builtin_amdgcn_sched_barrier(0);
            uint32_t a[V], b[V];
            vmul<V>(b, d[0], c[H2]);
#pragma unroll
            for (int v = 0; v < V; ++v) a[v] = x0[v];
            vmadd<V>(a, u[0], c[0]);
#pragma unroll
            for (int i = 1; i < H2; ++i) {
                vmadd<V>(a, u[i], c[i]);
                vmadd<V>(b, d[i], c[H2 + i]);
            }
            vadd<V>(*p[j + 1], a, b);
            vsub<V>(*p[Q - 1 - j], a, b);
#pragma unroll
            for (int v = 0; v < V; ++v) asm volatile("" : "+v"((*p[j + 1])[v]), "+v"((*p[Q - 1 - j])[v]));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < H2; ++j) {
            uint32_t cj[H2], sj[H2];
#pragma unroll
            for (int i = 0; i < H2; ++i) cj[i] = tab[j * H2 + i], sj[i] = tab[H2 * H2 + j * H2 + i];
            uint32_t a[V], b[V];
            vmul<V>(b, d[0], sj[0]);
#pragma unroll
            for (int v = 0; v < V; ++v) a[v] = x0[v];
            vmadd<V>(a, u[0], cj[0]);
#pragma unroll
            for (int i = 1; i < H2; ++i) {
                vmadd<V>(a, u[i], cj[i]);
                vmadd<V>(b, d[i], sj[i]);
            }
            vadd<V>(*p[j + 1], a, b);
            vsub<V>(*p[Q - 1 - j], a, b);
        }
    }
}

// (as a table: a recursive function is never inlined, and a call with a loop variable — constant only after unrolling — would index the
//  register arrays at run time, i.e. send them to scratch memory)
template <int Q> struct SlotTable {
    int v[Q];
    constexpr SlotTable() : v{}
    {
        for (int j = 0; j < Q; ++j) v[j] = out_slot_of(Q, j);
    }
};
template <int Q> constexpr int out_slot(int j)
{
    constexpr SlotTable<Q> table{};
    return table.v[j];
}

// In-place Q-point transform of the values *p[0..Q-1] (p: anything indexable that yields pointers to uint32_t[V] registers);
// X[j] ends up in *p[out_slot<Q>(j)].
template <int Q, int V, typename Slots>
__device__ __forceinline__ void small_dft_at(const Slots& p, const_u32_ptr tab)
{
    using Reg = uint32_t(*)[V];
    if constexpr (Q == 2) {  // the ordinary butterfly (ntt.cpp:16-22): a power-of-two top level treated like an odd one
        uint32_t t[V];
        vsub<V>(t, *p[0], *p[1]);
        vadd<V>(*p[0], *p[0], *p[1]);
#pragma unroll
        for (int v = 0; v < V; ++v) (*p[1])[v] = t[v];
    } else if constexpr (Q == 9) {
        // tab: C3, S3 (root w^3), then w^(i2*j1) for (i2, j1) = (1,1), (1,2), (2,1), (2,2)
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2) {
            const Reg t[3] = {p[i2], p[3 + i2], p[6 + i2]};
            sym_dft<3, V>(t, tab);
        }
        vmul<V>(*p[3 + 1], *p[3 + 1], tab[2]);
        vmul<V>(*p[6 + 1], *p[6 + 1], tab[3]);
        vmul<V>(*p[3 + 2], *p[3 + 2], tab[4]);
        vmul<V>(*p[6 + 2], *p[6 + 2], tab[5]);
#pragma unroll
        for (int j1 = 0; j1 < 3; ++j1) {
            const Reg t[3] = {p[3 * j1], p[3 * j1 + 1], p[3 * j1 + 2]};
            sym_dft<3, V>(t, tab);
        }
    } else if constexpr (pfa_first(Q) != 0) {
        // tab: the constants of the QA-point transform (root w^QB), then those of the QB-point transform (root w^QA)
        constexpr int QA = pfa_first(Q), QB = Q / QA;
        static_for<0, QB>([&](auto I2) {
            Reg t[QA];
            static_for<0, QA>([&](auto I1) { t[I1.value] = p[(QB * I1.value + QA * I2.value) % Q]; });
            small_dft_at<QA, V>(t, tab);
        });
        static_for<0, QA>([&](auto J1) {
            Reg t[QB];
            static_for<0, QB>([&](auto I2) { t[I2.value] = p[(QB * out_slot<QA>(J1.value) + QA * I2.value) % Q]; });
            small_dft_at<QB, V>(t, tab + dft_table_words(QA));
        });
    } else {
        sym_dft<Q, V>(p, tab);
    }
}

template <int Q, int V>
__device__ __forceinline__ void small_dft(uint32_t (&x)[Q][V], const_u32_ptr tab)
{
    uint32_t(*p[Q])[V];
#pragma unroll
    for (int i = 0; i < Q; ++i) p[i] = &x[i];
    small_dft_at<Q, V>(p, tab);
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
    if constexpr (Q <= 15) {
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
    } else {
        // The composite q (20 ... 116 twiddles, as many row addresses): the word offset of the row is ONE running scalar pinned by an empty
        // asm (left alone, all Q 64-bit row addresses are formed in the prologue and parked in spilled SGPRs), and the twiddles come in
        // pieces of at most 16 through an opaque offset, the next piece requested while this one is multiplied in (as in the fused kernel).
        uint64_t off = (uint64_t)i2 * a.ld;  // wave-uniform: the lane's column is added at the access
        const uint64_t pitch = (uint64_t)a.M * a.ld;
        static_for<0, Q>([&](auto I) {
            const uint32_t row = (uint32_t)I.value * a.M + i2;
            if (a.in_rows == 0 || row < a.in_rows) {
                load_vec<V>(x[I.value], a.in + off + col);
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) x[I.value][v] = 0;
            }
            off += pitch;
            asm volatile("" : "+s"(off));
        });
        auto twiddles = [&]() {
            constexpr int NTW = Q - 1, NCH = (NTW + 15) / 16, CH = (NTW + NCH - 1) / NCH;
            uint32_t t[2][CH];
            auto fetch = [&](uint32_t (&w)[CH], auto C) {
                uint32_t o = (uint32_t)(C.value * CH);
                asm volatile("" : "+s"(o));
                const_u32_ptr piece = tw + o;
                static_for<0, CH>([&](auto I) {
                    if constexpr (C.value * CH + I.value < NTW) w[I.value] = piece[I.value];
                    else w[I.value] = 0;
                });
            };
            fetch(t[0], std::integral_constant<int, 0>{});
            static_for<0, NCH>([&](auto C) {
                constexpr int c = C.value;
#pragma unroll
                for (int i = 0; i < CH; ++i) asm volatile("" : "+s"(t[c & 1][i]));
                if constexpr (c + 1 < NCH) fetch(t[(c + 1) & 1], std::integral_constant<int, c + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, CH>([&](auto I) {
                    constexpr int idx = c * CH + I.value + 1;  // twiddle of stripe idx (way up) / of output idx (way down)
                    if constexpr (idx < Q) {
                        constexpr int slot = DIT ? idx : out_slot<Q>(idx);
                        vmul<V>(x[slot], x[slot], t[c & 1][I.value]);
#pragma unroll
                        for (int v = 0; v < V; ++v) asm volatile("" : "+v"(x[slot][v]));
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        if constexpr (DIT) twiddles();
        small_dft<Q, V>(x, dft);
        static_for<0, Q>([&](auto I) {
            uint32_t(&y)[V] = x[I.value];  // (a lambda does not capture a variable it names in asm operands only)
#pragma unroll
            for (int v = 0; v < V; ++v) asm volatile("" : "+v"(y[v]));
        });
        if constexpr (!DIT) twiddles();
        off = (uint64_t)i2 * a.ld;
        asm volatile("" : "+s"(off));
        static_for<0, Q>([&](auto J) {
            const uint32_t row = (uint32_t)J.value * a.M + i2;
            if (a.out_rows == 0 || row < a.out_rows) store_vec<V>(a.out + off + col, x[out_slot<Q>(J.value)]);
            off += pitch;
            asm volatile("" : "+s"(off));
        });
    }
}

// ------------------------------------------------------------------------------------------------
// The odd-radix level fused with the A outermost power-of-two levels: one trip through HBM instead of two.
//
// A workgroup owns, for one (hi, lo) and one 64-word column chunk, the 2^A rows i2 = (hi << (s+A)) + (r << s) + lo of ALL q
// stripes.  A lane is one word column; the G = 2^(A-RLOG) waves hold
//     layout A  rows r = j*G + g  (j < R = 2^RLOG): the q-point transforms (each needs the q blocks i1*M + i2 of one r) and the
//               high RLOG levels are in-thread, all twiddles wave-uniform scalars;
//     layout B  rows r = g*R + k: the low A - RLOG levels are in-thread,
// and LDS is touched only to turn A into B, one stripe at a time (2^A rows of 256 bytes, conflict-free).  Way down:
// load A -> q-point transforms + twiddles -> per stripe: high levels, A=>B, low levels -> store B.  Way up: the mirror image.
// Reads happen before the first barrier and writes after it, so the pass may run in place.
// ------------------------------------------------------------------------------------------------
// Workgroup barrier for the LDS exchanges: __syncthreads() also waits for global memory (vmcnt(0)), which would drain the tile's loads
// and stores at every exchange; the exchange only needs this wave's LDS traffic to have completed (as in tile_kernels.hip).
__device__ __forceinline__ void lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Waves per SIMD the register allocator should aim for: a lane holds Q * 2^RLOG values plus ~24 temporaries.  Without the upper bound a
// 512-lane workgroup is compiled for 8 waves per SIMD (64 VGPRs) and the 80 values of q = 5 go to scratch.
constexpr int fused_max_waves(int q, int rlog)
{
    const int regs = ((q * (1 << rlog) + 24 + 7) / 8) * 8;
    const int w = 512 / regs;
    return w < 1 ? 1 : w > 8 ? 8 : w;
}

template <int Q, int A, int RLOG, bool DIT>
__global__ __launch_bounds__(64 << (A - RLOG)) __attribute__((amdgpu_waves_per_eu(1, fused_max_waves(Q, RLOG)))) void fused_radix_kernel(const FusedArgs a)
{
    // The A levels run in NRUNS register runs, counted from the top: run p < NRUNS-1 covers the RLOG levels [A - (p+1) RLOG, A - p RLOG),
    // the last one the remaining LAST levels [0, LAST).  In the layout of run p a lane's register j is bits [B, B + RLOG) of the tile row
    // (B = the run's lowest level; 0 for the last run) and the wave number fills the other A - RLOG bits.
    constexpr int R = 1 << RLOG, NRUNS = (A + RLOG - 1) / RLOG, LAST = A - (NRUNS - 1) * RLOG;
    static_assert(A >= RLOG && A - RLOG <= 4, "at most 16 waves per workgroup");
    extern __shared__ uint32_t lds[];  // 2^A rows of 64 words (unused with a single run)
    const uint32_t g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x;
    const uint32_t cc = tile % a.col_chunks;
    const uint32_t grp = tile / a.col_chunks;
    // Lanes beyond a ragged block end work on the block's last column as well (same loads, same arithmetic, identical stores from the same
    // wave), so the kernel has no divergent region: an "if (live)" around the stores made the compiler sink the q-point transforms of the DIT
    // form into that branch and fetch all their constants at its top (up to 134 SGPR spills, round 2).
    const uint32_t col = min(cc * 64u + lane, a.S - 1u);
    const int s = a.s;
    const uint32_t lo = grp & ((1u << s) - 1u);
    const uint32_t hi = grp >> s;
    const uint32_t row0 = (hi << (s + A)) + lo;  // row of the tile's r = 0 inside a stripe
    const_u32_ptr dft = as_constant(a.dft);
    uint32_t* my_lds = lds + lane;

    auto run_base = [](int p) { return p == NRUNS - 1 ? 0 : A - (p + 1) * RLOG; };
    // tile row of register j in the layout of run p
    auto row_of = [&](int p, int j) -> uint32_t {
        const int B = run_base(p);
        return ((g >> B) << (B + RLOG)) | ((uint32_t)j << B) | (g & ((1u << B) - 1u));
    };
    // block offset below the run's lowest level: what its twiddles depend on
    auto off_of = [&](int p) -> uint32_t { return ((g & ((1u << run_base(p)) - 1u)) << s) + lo; };
    uint32_t x[Q][R][1];

    // one stripe's registers from the layout of run `from` to that of run `to`.  The LDS address of register j in a layout is
    // (one VGPR per layout: lane + the wave's share of the row number) + (j << B) rows — a compile-time immediate of the ds instruction
    // (< 64 KiB), not 2 R separately computed addresses that would stay in VGPRs across all q stripes.
    auto lds_of = [&](int p) -> uint32_t* {
        const int B = run_base(p);
        return my_lds + ((((g >> B) << (B + RLOG)) | (g & ((1u << B) - 1u))) * 64u);
    };
    auto exchange = [&](uint32_t (&y)[R][1], int from, int to) {
        uint32_t* const wr = lds_of(from);
        uint32_t* const rd = lds_of(to);
        const int Bf = run_base(from), Bt = run_base(to);
        lds_sync();  // the previous exchange's reads are done
#pragma unroll
        for (int j = 0; j < R; ++j) wr[((uint32_t)j << Bf) * 64u] = y[j][0];
        lds_sync();
#pragma unroll
        for (int j = 0; j < R; ++j) y[j][0] = rd[((uint32_t)j << Bt) * 64u];
    };
    // the q-point transforms of register row j: twiddles w_N^(+-i2*j1) (tw, wave-uniform) before (way up) or after (way down)
    auto radix = [&](int j, const uint32_t (&tw)[Q - 1]) {
        uint32_t(*p[Q])[1];
#pragma unroll
        for (int i = 0; i < Q; ++i) p[i] = &x[i][j];
        if constexpr (DIT) {
#pragma unroll
            for (int i = 1; i < Q; ++i) x[i][j][0] = gf::mul_mont(x[i][j][0], tw[i - 1]);
        }
        small_dft_at<Q, 1>(p, dft);
        if constexpr (!DIT) {
#pragma unroll
            for (int i = 1; i < Q; ++i) x[out_slot<Q>(i)][j][0] = gf::mul_mont(x[out_slot<Q>(i)][j][0], tw[i - 1]);
        }
    };
    // All R rows, one at a time, the Q - 1 twiddles of row j + 1 requested while row j is in the arithmetic: two small SGPR sets instead of
    // R (Q - 1) scalars fetched at the top (which spilled: up to 246 SGPRs in round 3) and instead of all rows' temporaries live at once.
    // The empty asm pins the set that is about to be used (its s_waitcnt lands there, BEFORE the next set is requested — scalar loads return
    // out of order, so a wait issued after the next request would wait for both).
    auto radix_rows = [&]() {
        if constexpr (Q <= 15) {
            uint32_t tw[2][Q - 1];
            auto fetch = [&](uint32_t (&t)[Q - 1], int j) {
                const uint32_t i2 = row0 + (row_of(0, j) << s);
                const_u32_ptr src = as_constant(a.tw) + (size_t)i2 * (Q - 1);
    #pragma unroll
                for (int i = 0; i < Q - 1; ++i) t[i] = src[i];
            };
            fetch(tw[0], 0);
    #pragma unroll
            for (int j = 0; j < R; ++j) {
    #pragma unroll
                for (int i = 0; i < Q - 1; ++i) asm volatile("" : "+s"(tw[j & 1][i]));
                if (j + 1 < R) fetch(tw[(j + 1) & 1], j + 1);
                __builtin_amdgcn_sched_barrier(0);
                radix(j, tw[j & 1]);
                // (a scheduling barrier alone orders nothing here: instruction selection places pure arithmetic wherever it likes inside the
                //  kernel's one basic block — it sank all R rows' transforms below the last barrier; pinning the results ties them to this point)
    #pragma unroll
                for (int i = 0; i < Q; ++i) asm volatile("" : "+v"(x[i][j][0]));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // The composite q (20 ... 62 twiddles per row): the same scheme in pieces of at most 16 twiddles — one wide scalar load each, the
            // next piece (of this row or the next) requested while this one is multiplied in; the q-point transform of a row stands before
            // (way down) or after (way up) its pieces.
            constexpr int NTW = Q - 1, NCH = (NTW + 15) / 16, CH = (NTW + NCH - 1) / NCH;
            uint32_t tw[2][CH];
            auto fetch = [&](uint32_t (&t)[CH], int j, int c) {
                const uint32_t i2 = row0 + (row_of(0, j) << s);
                const_u32_ptr src = as_constant(a.tw) + (size_t)i2 * NTW + c * CH;
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if (c * CH + i < NTW) t[i] = src[i];
                    else t[i] = 0;
            };
            auto transform = [&](int j) {
                uint32_t(*p[Q])[1];
#pragma unroll
                for (int i = 0; i < Q; ++i) p[i] = &x[i][j];
                small_dft_at<Q, 1>(p, dft);
#pragma unroll
                for (int i = 0; i < Q; ++i) asm volatile("" : "+v"(x[i][j][0]));
                __builtin_amdgcn_sched_barrier(0);
            };
            fetch(tw[0], 0, 0);
            static_for<0, R>([&](auto J) {
                constexpr int j = J.value;
                if constexpr (!DIT) transform(j);
                static_for<0, NCH>([&](auto C) {
                    constexpr int c = C.value, e = j * NCH + c;
#pragma unroll
                    for (int i = 0; i < CH; ++i) asm volatile("" : "+s"(tw[e & 1][i]));
                    if constexpr (e + 1 < R * NCH) fetch(tw[(e + 1) & 1], (e + 1) / NCH, (e + 1) % NCH);
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<0, CH>([&](auto I) {
                        constexpr int idx = c * CH + I.value + 1;  // twiddle of stripe idx (way up) / of output idx (way down)
                        if constexpr (idx < Q) {
                            constexpr int slot = DIT ? idx : out_slot<Q>(idx);
                            x[slot][j][0] = gf::mul_mont(x[slot][j][0], tw[e & 1][I.value]);
                            asm volatile("" : "+v"(x[slot][j][0]));
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (DIT) transform(j);
            });
        }
    };
    // the levels of run p on one stripe, one level at a time: without the scheduling barriers the whole kernel is ONE basic block, and the
    // scheduler interleaves the butterflies of several levels (and their products' temporaries) until nothing fits the registers
    auto levels = [&](uint32_t (&y)[R][1], int p) {
        const int B = run_base(p);
        const int n = p == NRUNS - 1 ? LAST : RLOG;
        const uint32_t off = off_of(p);
        auto pin = [&]() {
#pragma unroll
            for (int j = 0; j < R; ++j) asm volatile("" : "+v"(y[j][0]));
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (DIT) {
            if (n >= 1) { dit_one_level<RLOG, 1, false, 0>(y, a.twl, off, s + B); pin(); }
            if constexpr (RLOG >= 2) if (n >= 2) { dit_one_level<RLOG, 1, false, 1>(y, a.twl, off, s + B); pin(); }
            if constexpr (RLOG >= 3) if (n >= 3) { dit_one_level<RLOG, 1, false, 2>(y, a.twl, off, s + B); pin(); }
            if constexpr (RLOG >= 4) if (n >= 4) { dit_one_level<RLOG, 1, false, 3>(y, a.twl, off, s + B); pin(); }
        } else {
            if constexpr (RLOG >= 4) if (n >= 4) { dif_one_level<RLOG, 1, false, 3>(y, a.twl, off, s + B); pin(); }
            if constexpr (RLOG >= 3) if (n >= 3) { dif_one_level<RLOG, 1, false, 2>(y, a.twl, off, s + B); pin(); }
            if constexpr (RLOG >= 2) if (n >= 2) { dif_one_level<RLOG, 1, false, 1>(y, a.twl, off, s + B); pin(); }
            if (n >= 1) { dif_one_level<RLOG, 1, false, 0>(y, a.twl, off, s + B); pin(); }
        }
    };

    // Addresses: raw buffer ops through ONE descriptor for the whole batch of q * M blocks (the host plans this kernel only for batches
    // below 4 GiB: plan.hip), the lane's column in voffset, and the block as ONE running scalar byte offset whose updates are pinned by an
    // empty asm — left alone, the compiler materialises all Q * R row offsets up front (160 SGPRs for q = 5), and a pinned POINTER loses
    // its address space (flat_load instead of global_load).  The descriptor ends after the last block the batch really holds (in_rows /
    // out_rows: zero-extended data, truncated parity), so the hardware bounds check replaces the branches.
    // (Rounds 2-4 had a descriptor per stripe — a stripe, not the batch, had to stay below 4 GiB; with all Q stripes' loads issued at the
    //  top, Q descriptors were alive at once: 16-58 SGPRs spilled to VGPR lanes for q = 13, up to 670 for q = 63.  Structured addressing
    //  (stride = block pitch, block number as index) reaches any batch size with one descriptor as well, but measured 1.2-1.6x slower:
    //  fused3_dif7 0.97 ms against 0.60.)
    // Layout of run 0: register j <-> tile row j * G + g; last run: tile row g * R + j.
    constexpr uint32_t G = 1u << (A - RLOG);
    const uint32_t voff = col * 4u;
    const uint32_t step0 = G << s, step_last = 1u << s;                           // in blocks
    const uint32_t first0 = row0 + (g << s), first_last = row0 + ((g * (uint32_t)R) << s);  // register 0 of the two layouts, inside a stripe
    auto batch_desc = [&](const uint32_t* base, uint32_t rows) {
        const uint64_t v = reinterpret_cast<uint64_t>(base);
        const uint32_t lo32 = __builtin_amdgcn_readfirstlane((uint32_t)v), hi32 = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        const uint32_t nbytes = rows == 0 ? 0xFFFFFFFFu : rows * (a.ld * 4u);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi32 << 32) | lo32), 0, __builtin_amdgcn_readfirstlane(nbytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t d_in = batch_desc(a.in, a.in_rows), d_out = batch_desc(a.out, a.out_rows);
    const uint32_t row_bytes = a.ld * 4u;
    auto load_stripe = [&](uint32_t (&y)[R][1], uint32_t stripe, uint32_t first, uint32_t step) {
        asm volatile("" : "+s"(stripe));  // (as in store_stripe)
        uint32_t soff = (stripe * a.M + first) * row_bytes;
        asm volatile("" : "+s"(soff));
#pragma unroll
        for (int j = 0; j < R; ++j) {
            y[j][0] = __builtin_amdgcn_raw_buffer_load_b32(d_in, voff, soff, 2);  // non-temporal: every word is touched once per pass
            soff += step * row_bytes;
            asm volatile("" : "+s"(soff));
        }
    };
    auto store_stripe = [&](const uint32_t (&y)[R][1], uint32_t stripe, uint32_t first, uint32_t step) {
        // (the stripe number goes through an empty asm: otherwise the offsets of all Q stripes' stores are computed in the kernel's prologue
        //  and parked in spilled SGPRs until each stripe's turn)
        asm volatile("" : "+s"(stripe));
        uint32_t soff = (stripe * a.M + first) * row_bytes;
        asm volatile("" : "+s"(soff));
#pragma unroll
        for (int j = 0; j < R; ++j) {
            __builtin_amdgcn_raw_buffer_store_b32(y[j][0], d_out, voff, soff, 2);
            soff += step * row_bytes;
            asm volatile("" : "+s"(soff));
        }
    };
    if constexpr (!DIT) {
        static_for<0, Q>([&](auto I) { load_stripe(x[I.value], (uint32_t)I.value, first0, step0); });
        __builtin_amdgcn_sched_barrier(0);
        radix_rows();
        static_for<0, Q>([&](auto J1) {
            uint32_t(&y)[R][1] = x[out_slot<Q>(J1.value)];  // stripe j1
            static_for<0, NRUNS>([&](auto P) {
                if constexpr (P.value > 0) exchange(y, P.value - 1, P.value);
                levels(y, P.value);
            });
            store_stripe(y, (uint32_t)J1.value, NRUNS > 1 ? first_last : first0, NRUNS > 1 ? step_last : step0);
            __builtin_amdgcn_sched_barrier(0);  // one stripe at a time: interleaving the stripes' levels costs registers, not time
        });
    } else {
        // every load of the tile is in flight before the first barrier
        static_for<0, Q>([&](auto J1) { load_stripe(x[J1.value], (uint32_t)J1.value, NRUNS > 1 ? first_last : first0, NRUNS > 1 ? step_last : step0); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, Q>([&](auto J1) {
            uint32_t(&y)[R][1] = x[J1.value];
            static_for<0, NRUNS>([&](auto PP) {
                constexpr int p = NRUNS - 1 - PP.value;
                levels(y, p);
                if constexpr (p > 0) exchange(y, p, p - 1);
            });
            __builtin_amdgcn_sched_barrier(0);  // one stripe at a time
        });
        radix_rows();
        static_for<0, Q>([&](auto T) {
            store_stripe(x[out_slot<Q>(T.value)], (uint32_t)T.value, first0, step0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }
}

// Values per lane (q * 2^RLOG) against workgroup size (64 * 2^(A - RLOG) lanes, at most 1024): the register run per radix and level count.
constexpr int fused_shape_rlog(int q, int levels)
{
    if (levels < 1 || levels > 8) return 0;
    if (q > 15) {
        // the composite radices: 84 values per lane (q = 21), else two rows of the q stripes (70 ... 126 values: every level is an exchange
        // through LDS).  Past 4 levels (q = 63: 3) the pass of its own plus a plain outer tile measured faster
        // (profiles/r04/mixed_pfa_fused_or_not.jsonl), so those shapes are not built.
        if (q > 63) return 0;
        const int rlog = q == 21 ? 2 : 1;
        const int most = q == 21 ? 6 : q == 63 ? 3 : 4;
        return levels > most ? 0 : rlog > levels ? levels : rlog;
    }
    int rlog = q <= 3 ? 4 : q <= 5 ? (levels == 8 ? 0 : 4) : q <= 9 ? 3 : 2;  // 48 / 80 / 56-72 / 52-60 values per lane
    if (rlog == 0 || (q == 9 && levels == 7)) return 0;  // (9, 7): 72 values per lane in a 1024-lane workgroup spill
    if (rlog > levels) rlog = levels;
    return levels - rlog <= 4 ? rlog : 0;
}

template <int Q, int A, bool DIT>
static hipError_t launch_fused_shape(const FusedArgs& a, unsigned tiles, hipStream_t st)
{
    constexpr int RLOG = fused_shape_rlog(Q, A);
    if constexpr (RLOG == 0) {
        return hipErrorInvalidValue;
    } else {
        constexpr int lds_bytes = A > RLOG ? (1 << A) * 256 : 0;
        hipLaunchKernelGGL((fused_radix_kernel<Q, A, RLOG, DIT>), dim3(tiles), dim3(64 << (A - RLOG)), lds_bytes, st, a);
        return hipGetLastError();
    }
}

template <int Q, bool DIT>
static hipError_t launch_fused_q(int levels, const FusedArgs& a, unsigned tiles, hipStream_t st)
{
    switch (levels) {
        case 1: return launch_fused_shape<Q, 1, DIT>(a, tiles, st);
        case 2: return launch_fused_shape<Q, 2, DIT>(a, tiles, st);
        case 3: return launch_fused_shape<Q, 3, DIT>(a, tiles, st);
        case 4: return launch_fused_shape<Q, 4, DIT>(a, tiles, st);
        case 5: return launch_fused_shape<Q, 5, DIT>(a, tiles, st);
        case 6: return launch_fused_shape<Q, 6, DIT>(a, tiles, st);
        case 7: return launch_fused_shape<Q, 7, DIT>(a, tiles, st);
        case 8: return launch_fused_shape<Q, 8, DIT>(a, tiles, st);
        default: return hipErrorInvalidValue;
    }
}


template <int Q, int V>
static hipError_t launch_q(bool dit, const RadixArgs& a, dim3 grid, hipStream_t st)
{
    if (dit) hipLaunchKernelGGL((radix_kernel<Q, true, V>), grid, dim3(256), 0, st, a);
    else     hipLaunchKernelGGL((radix_kernel<Q, false, V>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// mixed_kernels_pfa.hip: q = 21 ... 117 (the fused kernels of q = 39, 45 and of q = 63 in mixed_kernels_pfa2.hip / _pfa3.hip: compile time)
hipError_t launch_fused_pfa(int q, int levels, bool dit, const FusedArgs& a, unsigned tiles, hipStream_t st);
hipError_t launch_fused_pfa2(int q, int levels, bool dit, const FusedArgs& a, unsigned tiles, hipStream_t st);
hipError_t launch_fused_pfa3(int q, int levels, bool dit, const FusedArgs& a, unsigned tiles, hipStream_t st);
template <int Q>
static hipError_t launch_fused_dir(int levels, bool dit, const FusedArgs& a, unsigned tiles, hipStream_t st)
{
    return dit ? launch_fused_q<Q, true>(levels, a, tiles, st) : launch_fused_q<Q, false>(levels, a, tiles, st);
}
hipError_t launch_radix_pfa(int q, bool dit, const RadixArgs& a, dim3 grid, hipStream_t st);

}  // namespace fastecc
