// pack_kernels.hip — data packing of GF.md:72-104 ("Efficient data packing") on gfx950.
//
// RS.cpp only encodes words < p = 0xFFF00001 (README.md:160-162); arbitrary bytes have to be recoded first.
// The reference describes the recoding in prose and ships no code for it; the format implemented here is
// stated in include/fastecc.h (fastecc_pack_blocks).  A word is (digit << 20) | low20; the `words` 12-bit
// digits of a block are recoded from base 4096 to base 4095 (no digit 0xFFF survives, so every packed word
// is < 0xFFF00000 < p) and one flag word is appended: 4096-byte blocks become 4100-byte blocks.
//
// Mapping: one wave per block.  Lane l holds words l, l + 64, l + 128, ... (16 registers for 1024 words), so every
// load and store is a 256-byte contiguous row segment; the positions of the 0xFFF digits are found with
// wave ballots (scalar masks), ranks with mbcnt, and the recoded digit string is permuted through 2 KiB of LDS.
// Pure HBM streaming: 4 bytes read and 4 written per word.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace fastecc {

namespace {

constexpr int CHUNKS = 16;  // 16 * 64 = 1024 words per block at most

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// LDS traffic of one wave is ordered; this only stops the compiler from moving accesses across it.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void pack_blocks_kernel(const uint32_t* __restrict__ raw, uint32_t* __restrict__ packed,
                                                          uint32_t words, uint32_t ld, uint64_t blocks)
{
    __shared__ uint16_t lds[4][1024];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b = (uint64_t)blockIdx.x * 4u + wave;
    if (b >= blocks) return;  // wave-uniform
    const uint32_t* src = raw + b * words;
    uint32_t* dst = packed + b * ld;  // ld >= words + 1: the row pitch of the encoder's device stripes
    uint16_t* digits = lds[wave];

    uint32_t w[CHUNKS];
    uint64_t fff[CHUNKS];
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const uint32_t j = 64u * i + lane;
        w[i] = j < words ? src[j] : 0u;
        fff[i] = __ballot(j < words && (w[i] >> 20) == 0xFFFu);
        m += (uint32_t)__popcll(fff[i]);
    }

    if (m == 0) {  // the common case (78 % of random 4 KB blocks): nothing to recode
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const uint32_t j = 64u * i + lane;
            if (j < words) dst[j] = w[i];
        }
        if (lane == 0) dst[words] = 0u;
        return;
    }

    // digits = [position of every 0xFFF digit, 0x400 = another one follows][all other digits in order]
    uint32_t before = 0;
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const uint32_t j = 64u * i + lane;
        const uint32_t r = before + lanes_below(fff[i]);  // 0xFFF digits at positions < j
        if (j < words) {
            if ((fff[i] >> lane) & 1ull) digits[r] = (uint16_t)(j | (r + 1 < m ? 0x400u : 0u));
            else                         digits[m + (j - r)] = (uint16_t)(w[i] >> 20);
        }
        before += (uint32_t)__popcll(fff[i]);
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const uint32_t j = 64u * i + lane;
        if (j < words) dst[j] = ((uint32_t)digits[j] << 20) | (w[i] & 0xFFFFFu);
    }
    if (lane == 0) dst[words] = 1u;
}

__global__ __launch_bounds__(256) void unpack_blocks_kernel(const uint32_t* __restrict__ packed, uint32_t* __restrict__ raw,
                                                            uint32_t words, uint32_t ld, uint64_t blocks, unsigned long long* bad_blocks)
{
    __shared__ uint16_t lds[4][1024];
    __shared__ uint32_t marks[4][32];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b = (uint64_t)blockIdx.x * 4u + wave;
    if (b >= blocks) return;
    const uint32_t* src = packed + b * ld;
    uint32_t* dst = raw + b * words;
    uint16_t* digits = lds[wave];
    uint32_t* mark = marks[wave];

    uint32_t w[CHUNKS];
    uint64_t any_fff = 0;
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const uint32_t j = 64u * i + lane;
        w[i] = j < words ? src[j] : 0u;
        any_fff |= __ballot(j < words && (w[i] >> 20) == 0xFFFu);
    }
    const uint32_t flag = __builtin_amdgcn_readfirstlane(src[words]);

    bool bad = flag > 1u;
    if (flag == 0u) bad = any_fff != 0;
    if (flag == 1u) {
        // entry t is an index entry iff every entry before it carries the continuation bit 0x400
        uint32_t m = 0;
        bool open = true;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const uint32_t j = 64u * i + lane;
            digits[j] = (uint16_t)(w[i] >> 20);
            const uint64_t cont = __ballot(j < words && ((w[i] >> 20) & 0x400u));
            if (open) {
                if (cont == ~0ull) m += 64u;
                else {
                    m += (uint32_t)__builtin_ctzll(~cont) + 1u;
                    open = false;
                }
            }
        }
        if (lane < 32u) mark[lane] = 0u;
        wave_lds_fence();
        if (open || m > words) bad = true;  // the last entry of the block asks for another one
        m = m > words ? words : m;
        // index entries: 11 bits, positions strictly increasing and inside the block; they mark the 0xFFF digits
        bool wrong = false;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const uint32_t j = 64u * i + lane;
            const uint32_t d = w[i] >> 20;
            if (j < m) {
                const uint32_t idx = d & 0x3FFu;
                wrong |= (d & 0x800u) != 0 || idx >= words || (j > 0 && idx <= (digits[j - 1] & 0x3FFu));
                if (idx < words) atomicOr(&mark[idx >> 5], 1u << (idx & 31u));
            } else if (j < words) {
                wrong |= d == 0xFFFu;
            }
        }
        bad |= __ballot(wrong) != 0;
        wave_lds_fence();
        if (!bad) {
            uint32_t before = 0;
#pragma unroll
            for (int i = 0; i < CHUNKS; ++i) {
                const uint32_t j = 64u * i + lane;
                const uint64_t fff = (uint64_t)mark[2 * i] | ((uint64_t)mark[2 * i + 1] << 32);
                const uint32_t r = before + lanes_below(fff);
                if (j < words) {
                    const uint32_t d = ((fff >> lane) & 1ull) ? 0xFFFu : digits[m + (j - r)];
                    w[i] = (d << 20) | (w[i] & 0xFFFFFu);
                }
                before += (uint32_t)__popcll(fff);
            }
        }
    }
    // a block no packer produces is passed through unchanged and counted
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const uint32_t j = 64u * i + lane;
        if (j < words) dst[j] = w[i];
    }
    if (bad && lane == 0 && bad_blocks) atomicAdd(bad_blocks, 1ull);
}

}  // namespace

hipError_t launch_pack_blocks(const uint32_t* raw, uint32_t* packed, uint32_t words, uint32_t ld, uint64_t blocks, hipStream_t st)
{
    if (words == 0 || words > 64u * CHUNKS || ld <= words || blocks == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pack_blocks_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, st, raw, packed, words, ld, blocks);
    return hipGetLastError();
}

hipError_t launch_unpack_blocks(const uint32_t* packed, uint32_t* raw, uint32_t words, uint32_t ld, uint64_t blocks,
                                unsigned long long* bad_blocks, hipStream_t st)
{
    if (words == 0 || words > 64u * CHUNKS || ld <= words || blocks == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(unpack_blocks_kernel, dim3((unsigned)((blocks + 3) / 4)), dim3(256), 0, st, packed, raw, words, ld, blocks, bad_blocks);
    return hipGetLastError();
}

}  // namespace fastecc
