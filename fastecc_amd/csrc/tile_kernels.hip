// tile_kernels.hip — LDS-tiled passes: up to 10 radix-2 levels per trip through HBM.
//
// A workgroup owns a tile of T = 2^LOGT blocks (block stride 2^s) x W words.  Each lane keeps R = 2^LOGR
// words of ONE column in VGPRs and the tile visits LDS only to change which blocks a lane holds:
//
//   layout A  lane holds blocks q = [half*T/2 +] j*G + g   (j < R)  -> the high LOGR levels are in-thread
//   layout B  lane holds blocks q = g'*R + k                (k < R)  -> the low L2 = log2(G) levels are in-thread
//
// (G = waves per workgroup, g = wave id.)  One LDS round trip (ds_write_b32 / ds_read_b32, rows of W
// consecutive words -> conflict-free) turns A into B or back.  Twiddles depend only on the block index,
// never on the column, so they stay wave-uniform and are fetched with scalar loads.
//
// PAIR variant (W = 32): a wave covers two half-tiles, lanes 0-31 hold block q and lanes 32-63 block
// q + T/2 — butterfly partners at the top level.  That level is done across lanes with
// v_permlane32_swap: swapping registers (ja, jb) gives the low half-wave both operands of pair ja and
// the high half-wave both operands of pair jb, so every lane does one full butterfly per two registers
// (no redundant work) and all remaining levels again see wave-uniform twiddles.  This is what makes a
// 1024-block tile fit: 1024 x 32 words = 128 KiB of the CU's 160 KiB LDS, in 128-byte row segments.
//
//   DIF  load A -> [pair level] -> LOGR levels -> A=>B -> L2 levels -> store B
//   DIT  load B -> L2 levels -> B=>A -> LOGR levels -> [pair level] -> store A
//   MID  DIF half, multiply block p by D[bitrev(p)] (RS.cpp:51-59), DIT half: 2*LOGT levels per HBM trip
//
// With N = 2^19 the encode is 3 launches: DIF over levels 18..9, MID over 8..0 twice, DIT over 9..18
// (the reference needs 2 x (3 sweeps + twiddle sweep), ntt.cpp:412-446).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gf.hpp"
#include "kernels.hpp"
#include "ntt_device.hpp"

namespace fastecc {

template <int LOGT, int LOGR, bool PAIR>
struct TileCfg {
    static constexpr int T = 1 << LOGT;
    static constexpr int R = 1 << LOGR;
    static constexpr int L2 = LOGT - LOGR - (PAIR ? 1 : 0);  // levels done in layout B
    static constexpr int G = 1 << L2;                        // waves per workgroup
    static constexpr int W = PAIR ? 32 : 64;                 // words per tile row
    static constexpr int THREADS = G * 64;
    static constexpr int LDS_BYTES = T * W * 4;
    static_assert(L2 >= 1 && L2 <= LOGR, "tile shape");
    static_assert(THREADS <= 1024, "workgroup size");
};

// Top level of a PAIR tile, decimation in frequency: (a, b) -> (a + b, (a - b) * w).
template <int LOGR>
__device__ __forceinline__ void pair_level_dif(uint32_t (&x)[1 << LOGR][1], const uint32_t* __restrict__ tw, uint32_t g, int G,
                                               uint32_t lo, int s, int shift, bool upper)
{
    constexpr int R = 1 << LOGR;
#pragma unroll
    for (int ja = 0; ja < R; ja += 2) {
        const int jb = ja + 1;
        const uint32_t wa = tw[((((uint32_t)ja * G + g) << s) + lo) << shift];
        const uint32_t wb = tw[((((uint32_t)jb * G + g) << s) + lo) << shift];
        const auto r = __builtin_amdgcn_permlane32_swap(x[ja][0], x[jb][0], false, false);
        const uint32_t a = r[0], b = r[1];
        const uint32_t w = upper ? wb : wa;
        const uint32_t sum = gf::add(a, b);
        const uint32_t dif = gf::mul_mont(gf::sub(a, b), w);
        const auto o = __builtin_amdgcn_permlane32_swap(sum, dif, false, false);
        x[ja][0] = o[0];
        x[jb][0] = o[1];
    }
}

// Top level of a PAIR tile, decimation in time: (a, b) -> (a + b*w, a - b*w).
template <int LOGR>
__device__ __forceinline__ void pair_level_dit(uint32_t (&x)[1 << LOGR][1], const uint32_t* __restrict__ tw, uint32_t g, int G,
                                               uint32_t lo, int s, int shift, bool upper)
{
    constexpr int R = 1 << LOGR;
#pragma unroll
    for (int ja = 0; ja < R; ja += 2) {
        const int jb = ja + 1;
        const uint32_t wa = tw[((((uint32_t)ja * G + g) << s) + lo) << shift];
        const uint32_t wb = tw[((((uint32_t)jb * G + g) << s) + lo) << shift];
        const auto r = __builtin_amdgcn_permlane32_swap(x[ja][0], x[jb][0], false, false);
        const uint32_t a = r[0];
        const uint32_t b = gf::mul_mont(r[1], upper ? wb : wa);
        const auto o = __builtin_amdgcn_permlane32_swap(gf::add(a, b), gf::sub(a, b), false, false);
        x[ja][0] = o[0];
        x[jb][0] = o[1];
    }
}

template <int LOGT, int LOGR, bool PAIR, int MODE>
__global__ __launch_bounds__((TileCfg<LOGT, LOGR, PAIR>::THREADS)) void ntt_tile_kernel(const TileArgs a)
{
    using C = TileCfg<LOGT, LOGR, PAIR>;
    constexpr int R = C::R, G = C::G, W = C::W, L2 = C::L2, T = C::T;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];

    const uint32_t g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t c = PAIR ? (lane & 31u) : lane;
    const uint32_t half = PAIR ? (lane >> 5) : 0u;
    const bool upper = half != 0;

    const uint32_t cc = blockIdx.x % a.col_chunks;
    const uint32_t grp = blockIdx.x / a.col_chunks;
    const int s = MODE == MODE_MID ? 0 : a.s;
    const uint32_t lo = grp & ((1u << s) - 1u);
    const uint32_t hi = grp >> s;
    const uint32_t base_row = (hi << (s + LOGT)) + lo;  // tile block q is stripe block base_row + (q << s)
    const uint32_t col = cc * W + c;
    const bool live = col < a.S;

    // block held in register j (layout A) / k (layout B)
    const uint32_t qa0 = half * (T / 2) + g;                 // + j*G
    const uint32_t qb0 = (PAIR ? 2u * g + half : g) * R;     // + k
    // layout A as seen by dif_levels/dit_levels: stride 2^(s+L2), offset (g << s) + lo below it
    const int sl = s + L2;
    const uint32_t off = (g << s) + lo;
    const int pair_shift = a.n - s - LOGT;  // exponent scale of the pair level (half-size 2^(s+LOGT-1))

    uint32_t x[R][1];

    auto load_rows = [&](uint32_t q0, uint32_t qstep) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const size_t row = base_row + ((size_t)(q0 + j * qstep) << s);
            x[j][0] = live ? a.in[row * a.S + col] : 0u;
        }
    };
    auto store_rows = [&](uint32_t q0, uint32_t qstep) {
        if (!live) return;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const size_t row = base_row + ((size_t)(q0 + j * qstep) << s);
            a.out[row * a.S + col] = x[j][0];
        }
    };
    auto lds_write = [&](uint32_t q0, uint32_t qstep) {
#pragma unroll
        for (int j = 0; j < R; ++j) lds[(q0 + j * qstep) * W + c] = x[j][0];
    };
    auto lds_read = [&](uint32_t q0, uint32_t qstep) {
#pragma unroll
        for (int j = 0; j < R; ++j) x[j][0] = lds[(q0 + j * qstep) * W + c];
    };

    if constexpr (MODE == MODE_DIF || MODE == MODE_MID) {
        load_rows(qa0, G);
        if constexpr (PAIR) pair_level_dif<LOGR>(x, a.tw_dif, g, G, lo, s, pair_shift, upper);
        dif_levels<LOGR, 1, false>(x, a.tw_dif, off, sl, a.n);
        lds_write(qa0, G);
        __syncthreads();
        lds_read(qb0, 1);
        if constexpr (MODE == MODE_DIF) {
            if (s == 0) dif_levels<LOGR, 1, true, L2>(x, a.tw_dif, 0u, 0, a.n);
            else        dif_levels<LOGR, 1, false, L2>(x, a.tw_dif, lo, s, a.n);
            store_rows(qb0, 1);
        } else {
            dif_levels<LOGR, 1, true, L2>(x, a.tw_dif, 0u, 0, a.n);
            // position p = hi*T + q holds coefficient bitrev_n(p); dscale is stored in position order
            const uint32_t* __restrict__ d = a.dscale + ((size_t)hi << LOGT) + qb0;
#pragma unroll
            for (int k = 0; k < R; ++k) x[k][0] = gf::mul_mont(x[k][0], d[k]);
            dit_levels<LOGR, 1, true, L2>(x, a.tw_dit, 0u, 0, a.n);
            __syncthreads();  // every lane has finished reading the first exchange
            lds_write(qb0, 1);
            __syncthreads();
            lds_read(qa0, G);
            dit_levels<LOGR, 1, false>(x, a.tw_dit, off, sl, a.n);
            if constexpr (PAIR) pair_level_dit<LOGR>(x, a.tw_dit, g, G, lo, s, pair_shift, upper);
            store_rows(qa0, G);
        }
    } else {
        load_rows(qb0, 1);
        if (s == 0) dit_levels<LOGR, 1, true, L2>(x, a.tw_dit, 0u, 0, a.n);
        else        dit_levels<LOGR, 1, false, L2>(x, a.tw_dit, lo, s, a.n);
        lds_write(qb0, 1);
        __syncthreads();
        lds_read(qa0, G);
        dit_levels<LOGR, 1, false>(x, a.tw_dit, off, sl, a.n);
        if constexpr (PAIR) pair_level_dit<LOGR>(x, a.tw_dit, g, G, lo, s, pair_shift, upper);
        store_rows(qa0, G);
    }
}

// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
template <int LOGT, int LOGR, bool PAIR, int MODE>
static hipError_t launch_one(const TileArgs& a, hipStream_t st)
{
    using C = TileCfg<LOGT, LOGR, PAIR>;
    auto kern = ntt_tile_kernel<LOGT, LOGR, PAIR, MODE>;
    static bool configured = false;  // per instantiation; the attribute is idempotent
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) return e;
        configured = true;
    }
    TileArgs b = a;
    b.col_chunks = (a.S + C::W - 1) / C::W;
    const uint64_t blocks = ((uint64_t)1 << (a.n - LOGT)) * b.col_chunks;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::THREADS), C::LDS_BYTES, st, b);
    return hipGetLastError();
}

template <int LOGT, bool PAIR>
static hipError_t launch_mode(int mode, const TileArgs& a, hipStream_t st)
{
    switch (mode) {
        case MODE_DIF: return launch_one<LOGT, 5, PAIR, MODE_DIF>(a, st);
        case MODE_DIT: return launch_one<LOGT, 5, PAIR, MODE_DIT>(a, st);
        default:       return launch_one<LOGT, 5, PAIR, MODE_MID>(a, st);
    }
}

bool tile_supported(int logt, bool pair)
{
    return pair ? (logt >= 7 && logt <= 10) : (logt >= 6 && logt <= 9);
}

hipError_t launch_tile(int logt, bool pair, int mode, const TileArgs& a, hipStream_t st)
{
    if (!tile_supported(logt, pair) || a.n < logt) return hipErrorInvalidValue;
    if (pair) {
        switch (logt) {
            case 7: return launch_mode<7, true>(mode, a, st);
            case 8: return launch_mode<8, true>(mode, a, st);
            case 9: return launch_mode<9, true>(mode, a, st);
            default: return launch_mode<10, true>(mode, a, st);
        }
    }
    switch (logt) {
        case 6: return launch_mode<6, false>(mode, a, st);
        case 7: return launch_mode<7, false>(mode, a, st);
        case 8: return launch_mode<8, false>(mode, a, st);
        default: return launch_mode<9, false>(mode, a, st);
    }
}

}  // namespace fastecc
