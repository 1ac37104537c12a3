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

// Top level of a PAIR tile (level sl + LOGR as seen from layout A: partner distance "2^LOGR registers" is
// the other half-wave).  Its 2^LOGR twiddles, one per register, are contiguous in the level table; they
// are fetched 16 at a time to bound SGPR pressure.  The low half-wave needs w[ja], the high one w[jb]:
// selected with (wa ^ ((wa ^ wb) & upper_mask)) — one scalar xor, two vector ops — because writing it as a
// ?: on SGPR array elements makes the compiler build a 32-way compare/select chain.
template <int LOGR>
__device__ __forceinline__ uint32_t pair_twiddle(uint32_t wa, uint32_t wb, uint32_t upper_mask)
{
    return wa ^ ((wa ^ wb) & upper_mask);
}

// Decimation in frequency: (a, b) -> (a + b, (a - b) * w).
template <int LOGR>
__device__ __forceinline__ void pair_level_dif(uint32_t (&x)[1 << LOGR][1], const uint32_t* __restrict__ twl, uint32_t off, int sl,
                                               uint32_t upper_mask)
{
    constexpr int R = 1 << LOGR, CH = R < 16 ? R : 16;
    const uint32_t* __restrict__ p = twl + (1u << (sl + LOGR)) + (off << LOGR);
#pragma unroll
    for (int c0 = 0; c0 < R; c0 += CH) {
        uint32_t w[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) w[i] = p[c0 + i];
#pragma unroll
        for (int i = 0; i < CH; i += 2) {
            const int ja = c0 + i, jb = ja + 1;
            const auto r = __builtin_amdgcn_permlane32_swap(x[ja][0], x[jb][0], false, false);
            const uint32_t a = r[0], b = r[1];
            const uint32_t sum = gf::add(a, b);
            const uint32_t dif = gf::mul_mont(gf::sub(a, b), pair_twiddle<LOGR>(w[i], w[i + 1], upper_mask));
            const auto o = __builtin_amdgcn_permlane32_swap(sum, dif, false, false);
            x[ja][0] = o[0];
            x[jb][0] = o[1];
        }
    }
}

// Decimation in time: (a, b) -> (a + b*w, a - b*w).
template <int LOGR>
__device__ __forceinline__ void pair_level_dit(uint32_t (&x)[1 << LOGR][1], const uint32_t* __restrict__ twl, uint32_t off, int sl,
                                               uint32_t upper_mask)
{
    constexpr int R = 1 << LOGR, CH = R < 16 ? R : 16;
    const uint32_t* __restrict__ p = twl + (1u << (sl + LOGR)) + (off << LOGR);
#pragma unroll
    for (int c0 = 0; c0 < R; c0 += CH) {
        uint32_t w[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) w[i] = p[c0 + i];
#pragma unroll
        for (int i = 0; i < CH; i += 2) {
            const int ja = c0 + i, jb = ja + 1;
            const auto r = __builtin_amdgcn_permlane32_swap(x[ja][0], x[jb][0], false, false);
            const uint32_t a = r[0];
            const uint32_t b = gf::mul_mont(r[1], pair_twiddle<LOGR>(w[i], w[i + 1], upper_mask));
            const auto o = __builtin_amdgcn_permlane32_swap(gf::add(a, b), gf::sub(a, b), false, false);
            x[ja][0] = o[0];
            x[jb][0] = o[1];
        }
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
    const uint32_t upper_mask = 0u - half;  // all ones in the high half-wave of a PAIR tile

    const uint32_t cc = blockIdx.x % a.col_chunks;
    const uint32_t grp = blockIdx.x / a.col_chunks;
    const int s = MODE == MODE_MID ? 0 : a.s;
    const uint32_t lo = grp & ((1u << s) - 1u);
    const uint32_t hi = grp >> s;
    const uint32_t base_row = (hi << (s + LOGT)) + lo;  // tile block q is stripe block base_row + (q << s)
    const uint32_t col = cc * W + c;
    const bool live = col < a.S;

    // Block held in register j (layout A) / k (layout B), split into a wave-uniform part (SGPRs) and the
    // half-wave part of a PAIR tile, which is folded ONCE into per-lane base pointers / LDS offsets so that
    // every global and LDS address below is "lane base + uniform offset".
    const uint32_t qa_u = g;                                 // + j*G          (+ half*T/2 per lane)
    const uint32_t qb_u = (PAIR ? 2u * g : g) * R;           // + k            (+ half*R   per lane)
    const uint32_t qa_l = half * (T / 2), qb_l = half * R;
    const size_t lane_a = ((size_t)qa_l << s) * a.S + col, lane_b = ((size_t)qb_l << s) * a.S + col;
    const uint32_t* in_a = a.in + lane_a;
    const uint32_t* in_b = a.in + lane_b;
    uint32_t* out_a = a.out + lane_a;
    uint32_t* out_b = a.out + lane_b;
    uint32_t* lds_a = lds + qa_l * W + c;
    uint32_t* lds_b = lds + qb_l * W + c;
    const size_t row_elems = (size_t)a.S << s;               // distance between consecutive tile blocks
    const size_t tile_origin = (size_t)base_row * a.S;
    // layout A as seen by dif_levels/dit_levels: stride 2^(s+L2), offset (g << s) + lo below it
    const int sl = s + L2;
    const uint32_t off = (g << s) + lo;

    uint32_t x[R][1];

    auto load_rows = [&](const uint32_t* lane_base, uint32_t q0, uint32_t qstep) {
        if (live) {
#pragma unroll
            for (int j = 0; j < R; ++j) x[j][0] = lane_base[tile_origin + (size_t)(q0 + j * qstep) * row_elems];
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) x[j][0] = 0u;
        }
    };
    auto store_rows = [&](uint32_t* lane_base, uint32_t q0, uint32_t qstep) {
        if (!live) return;
#pragma unroll
        for (int j = 0; j < R; ++j) lane_base[tile_origin + (size_t)(q0 + j * qstep) * row_elems] = x[j][0];
    };
    auto lds_write = [&](uint32_t* lane_base, uint32_t q0, uint32_t qstep) {
#pragma unroll
        for (int j = 0; j < R; ++j) lane_base[(q0 + j * qstep) * W] = x[j][0];
    };
    auto lds_read = [&](const uint32_t* lane_base, uint32_t q0, uint32_t qstep) {
#pragma unroll
        for (int j = 0; j < R; ++j) x[j][0] = lane_base[(q0 + j * qstep) * W];
    };

    if constexpr (MODE == MODE_DIF || MODE == MODE_MID) {
        load_rows(in_a, qa_u, G);
        if constexpr (PAIR) pair_level_dif<LOGR>(x, a.tw_dif, off, sl, upper_mask);
        dif_levels<LOGR, 1, false>(x, a.tw_dif, off, sl);
        lds_write(lds_a, qa_u, G);
        __syncthreads();
        lds_read(lds_b, qb_u, 1);
        if constexpr (MODE == MODE_DIF) {
            if (s == 0) dif_levels<LOGR, 1, true, L2>(x, a.tw_dif, 0u, 0);
            else        dif_levels<LOGR, 1, false, L2>(x, a.tw_dif, lo, s);
            store_rows(out_b, qb_u, 1);
        } else {
            dif_levels<LOGR, 1, true, L2>(x, a.tw_dif, 0u, 0);
            // position p = hi*T + q holds coefficient bitrev_n(p); dscale is stored in position order, so the
            // R factors of a lane are contiguous.  In a PAIR tile the two half-waves hold different blocks:
            // both runs are fetched (scalar) and selected per lane like the pair-level twiddles.
            const uint32_t* __restrict__ d = a.dscale + ((size_t)hi << LOGT) + qb_u;
            constexpr int CH = 8;
#pragma unroll
            for (int k0 = 0; k0 < R; k0 += CH) {
                uint32_t dl[CH], dh[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) dl[i] = d[k0 + i];
                if constexpr (PAIR) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) dh[i] = d[R + k0 + i];
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const uint32_t f = PAIR ? pair_twiddle<LOGR>(dl[i], dh[i], upper_mask) : dl[i];
                    x[k0 + i][0] = gf::mul_mont(x[k0 + i][0], f);
                }
            }
            dit_levels<LOGR, 1, true, L2>(x, a.tw_dit, 0u, 0);
            __syncthreads();  // every lane has finished reading the first exchange
            lds_write(lds_b, qb_u, 1);
            __syncthreads();
            lds_read(lds_a, qa_u, G);
            dit_levels<LOGR, 1, false>(x, a.tw_dit, off, sl);
            if constexpr (PAIR) pair_level_dit<LOGR>(x, a.tw_dit, off, sl, upper_mask);
            store_rows(out_a, qa_u, G);
        }
    } else {
        load_rows(in_b, qb_u, 1);
        if (s == 0) dit_levels<LOGR, 1, true, L2>(x, a.tw_dit, 0u, 0);
        else        dit_levels<LOGR, 1, false, L2>(x, a.tw_dit, lo, s);
        lds_write(lds_b, qb_u, 1);
        __syncthreads();
        lds_read(lds_a, qa_u, G);
        dit_levels<LOGR, 1, false>(x, a.tw_dit, off, sl);
        if constexpr (PAIR) pair_level_dit<LOGR>(x, a.tw_dit, off, sl, upper_mask);
        store_rows(out_a, qa_u, G);
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
