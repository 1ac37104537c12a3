// internal.hpp — what decode.hip needs from api.hip / encode.hip / host_stage.hip / create.hip (nothing here is part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/fastecc.h"

namespace fastecc {

struct DecodeState;                        // decode.hip: tables of one erasure pattern + the size-2k transform context
void destroy_decode_state(DecodeState*);   // decode.hip, called by fastecc_destroy

// ---- sharded.hip: one stripe in column slabs on several devices (fastecc_create_sharded) ----
struct Sharded;
void destroy_sharded(Sharded*);
int sharded_encode_stripe(fastecc_ctx* shell, const void* data, void* parity, int mem_kind, hipStream_t st);
enum { SH_PROFILE_ENABLE, SH_PROFILE_RESET, SH_SET_OPTION, SH_SET_PLAN };
int sharded_forward(fastecc_ctx* shell, int what, const char* name, int value);
int sharded_decode_prepare(fastecc_ctx* shell, const uint8_t* data_present, const uint8_t* parity_present);
int sharded_decode_stripe(fastecc_ctx* shell, void* data, void* parity, int mem_kind, bool repair, hipStream_t st);
fastecc_ctx* sharded_child(fastecc_ctx* shell, int g);

// ---- direct.hip: blocks as fixed linear combinations of other blocks (few lost blocks / few parity blocks) ----
struct DirectPass;                 // the weights of one pattern (or code) and the work buffers of its pass
DirectPass* direct_pass_new();
void direct_pass_free(DirectPass*);
int direct_cap();                  // most outputs (lost or parity blocks) of one pass
// Output t = f(y[t]) from ALL K data rows (row i at the point wd^i, wd of order N): weight c[t] x_i / (y[t] - x_i), c[t] = (y[t]^N - 1) / N;
// written to data row pos >> 1 (pos even) or parity row pos >> 1 (pos odd).  Synchronises `st`.
int direct_build_lagrange(DirectPass* p, uint32_t wd, uint32_t K, const std::vector<uint32_t>& y, const std::vector<uint32_t>& c, const std::vector<uint32_t>& out_pos,
                          hipStream_t st);
// The lost data rows (points lost_points) from the surviving data rows and as many surviving parity rows (node_rows at node_points).
// more_points / more_pos: further outputs on the same nodes — f at those points, written to row pos >> 1 of the parity (pos odd) or data stripe: the lost
// parity blocks, for fastecc_repair in one pass.
int direct_build_interp(DirectPass* p, uint32_t wd, uint64_t N, uint32_t K, const std::vector<uint32_t>& lost_rows, const std::vector<uint32_t>& lost_points,
                        const std::vector<uint32_t>& node_rows, const std::vector<uint32_t>& node_points, hipStream_t st,
                        const std::vector<uint32_t>* more_points = nullptr, const std::vector<uint32_t>* more_pos = nullptr);
// kernel: 0 = choose, 1 = VALU (96-bit lazy accumulation), 2 = MFMA (i8 digits) when the stripes allow it
int direct_run(DirectPass* p, const uint32_t* data, const uint32_t* parity, uint32_t* data_out, uint32_t* parity_out, uint32_t S, int kernel, hipStream_t st);
bool direct_mfma_applies(const void* data, const void* parity, uint64_t words);
// encoding straight from the Lagrange basis for codes with few parity blocks (n - k <= direct_encode_max())
struct DirectEncode;
int direct_encode_max();
int direct_encode_build(DirectEncode** out, uint64_t N, uint64_t K, uint64_t m, int fold, uint64_t words);  // current device = the context's
int direct_encode_run(DirectEncode* de, const uint32_t* data, uint32_t* parity, int kernel, hipStream_t st);
void direct_encode_destroy(DirectEncode* de);

// ---- api.hip, encode.hip, host_stage.hip, create.hip ----
struct CtxInfo {
    int device, field, fold, cosets, log2k;
    uint64_t k, words, pitch;  // blocks, words per block, words between device blocks
    bool zero_extended;        // k or n - k is not the power of two the transform works on
    uint64_t user_k, user_m;   // data / parity blocks of the caller's stripes (== k, k >> fold unless zero_extended)
    int q;                     // > 1: the transform order is q * k (mixed radix); k is its power-of-two part
    int direct_max;            // decoder: patterns with at most this many lost blocks take the direct path (option "decode_direct_max")
    int direct_kernel;         // 0 choose, 1 VALU, 2 MFMA (option "direct_kernel")
    int p61_stride;            // GF((2^61-1)^2) codes other than (2N,N): parity block j = block j * p61_stride of the (2N,N) parity
    int decode_split;          // option "decode_split"
};
CtxInfo info_of(const fastecc_ctx* c);
DecodeState*& decoder_of(fastecc_ctx* c);
Sharded*& sharded_of(fastecc_ctx* c);
namespace p61 {
struct Decoder;
struct Path;
struct LaunchHooks;
}
p61::Decoder*& decoder61_of(fastecc_ctx* c);  // the erasure decoder of a GF((2^61-1)^2) context (gf61_decode.hip)
p61::Path* p61_path_of(fastecc_ctx* c);       // its encoder
// fastecc_profile_* over the launches of a p61 call: nullptr unless the context is profiling; *keep goes to p61_profile_done after the call
// one record of fastecc_profile_* around a step of the decoder (its inner contexts are not the caller's): nullptr unless the context is profiling
void* profile_scope_begin(fastecc_ctx* c, hipStream_t st, const char* name, uint64_t bytes);
void profile_scope_end(void* scope);
const p61::LaunchHooks* p61_profile_hooks(fastecc_ctx* c, void** keep);
void p61_profile_done(void* keep);
std::mutex& mutex_of(fastecc_ctx* c);
// a context that owns nothing but its geometry: fastecc_create_sharded hangs the per-device contexts on it
fastecc_ctx* new_shell_ctx(int root_device, int field, uint64_t k, uint64_t m, uint64_t block_bytes);
void set_plan_text(fastecc_ctx* c, const std::string& t);
int columns_supported(const fastecc_ctx* c);  // fastecc_encode_columns works on this context
void set_error_detail(const char* what, hipError_t e);
void set_error_text(const char* text);  // this thread's fastecc_last_error_detail, verbatim (a worker thread's text republished on the caller's)
// host_copy.hip: `rows` pieces of `width` bytes between two pitched host buffers (software prefetch + streaming stores)
void host_copy_rows(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, size_t width, size_t rows);
// a rows x width rectangle between PAGEABLE host memory and the device through c's ring of pinned slots (host_stage.hip stage_transfer); c's call lock is
// the caller's business; threads = 0: the default number of helper threads
int stage_rect(fastecc_ctx* c, bool to_device, void* host, size_t host_pitch, void* dev, size_t dev_pitch, size_t width, size_t rows, hipStream_t st, int threads);
// device -> pageable host memory through a ring of pinned slots emptied by helper threads (host_stage.hip); synchronous; c's call lock held by the caller
int download_pageable(fastecc_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st);

// A transform context (GF(0xFFF00001)): DIF over all log2k levels with inverse roots, the block holding coefficient m
// multiplied by factor[m] (plain representatives, k entries), DIT back with forward roots keeping every 2^fold-th
// output block.  fastecc_encode(ctx, in, out, FASTECC_MEM_DEVICE, stream) runs it; k may be 2^20 (no root of order 2k
// is needed).  The encoder of RS.cpp:40-63 is the case factor[m] = w_2k^m / k.
int create_transform_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int fold, const uint32_t* factor, int device);
// create_transform_ctx with factor[m] = m * scale, the table written by a kernel (the decoder's x p'(x) transform: scale = 1 / 2^log2k)
int create_ramp_transform_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int fold, uint32_t scale, int device, uint32_t offset = 0);  // factor m * scale + offset
// A context for stand-alone transforms only (fastecc_ntt, transform_bitrev): no per-block factor table is built, fastecc_encode is unsupported.
int create_ntt_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int device);
// The same for a transform of order q * 2^log2m (q an odd radix of mixed_kernels.hip): factor has q << log2m entries by
// coefficient index; fastecc_encode(ctx, in, out, DEVICE, stream) maps all q << log2m blocks, in place if in == out.
int create_mixed_transform_ctx(fastecc_ctx** out, int q, int log2m, uint64_t block_bytes, const uint32_t* factor, int device);
// The way down of such a context alone (inverse roots, unscaled): block j1 * 2^log2m + r of `out` holds
// Y[q * bitrev(r) + j1],  Y[v] = sum_i in[i] w^(-i v),  w of order q << log2m.
int mixed_dif(fastecc_ctx* c, const uint32_t* in, uint32_t* out, hipStream_t st);
// Decoder: when the transform's first pass is a register DIF pass it can read the codeword straight from its two
// halves (PassArgs::in_odd / row_factor) and the separate gather pass disappears.  run_gathered does that and returns
// FASTECC_E_UNSUPPORTED (nothing enqueued) when the plan starts with another kind of pass.
int run_gathered(fastecc_ctx* c, const uint32_t* even_blocks, const uint32_t* odd_blocks, const uint32_t* row_factor, uint32_t* out,
                 hipStream_t st);
// Half a stand-alone transform of a transform context, on the word columns [0, width) of every block: dit = false runs the
// DIF passes (natural order in, bit-reversed order out), dit = true the mirrored DIT passes (bit-reversed in, natural out);
// unscaled, forward roots or (inverse_roots) their inverses.  A product of transforms does not care about the order, so
// polynomial products on the device need no permutation pass (decode.hip: the erasure locator's product tree).
int transform_bitrev(fastecc_ctx* c, const uint32_t* in, uint32_t* out, bool dit, bool inverse_roots, uint32_t width, hipStream_t st);
// The order in which that first pass wants row_factor: returns false when it is the natural order (register pass),
// else fills `order` with order[i] = codeword position whose factor is entry i of the table (two-window DIF tile:
// the factors of one wave are contiguous).
bool gather_tile_order(const fastecc_ctx* c, std::vector<uint32_t>& order);
bool gather_tile_order_device(const fastecc_ctx* c, uint32_t* order, hipStream_t st);  // the same N words, written on the device
bool same_tile_order(const fastecc_ctx* a, const fastecc_ctx* b);
// The decoder's split transform (decode.hip, "even / odd split") on a context of k blocks created by create_ramp_transform_ctx with the factor
// (2m + k) / 2k: data and parity stripes each through the plan's first DIF tile with per-block factors (tile order: gather_tile_order; a zero
// factor = block not used), the parity half only in its first `parity_groups` block groups (group g = blocks g + (t << s), t < group_rows; the
// rest of r1 must be zero and stays zero), its low levels as a DIF tile of their own (r1 -> r2), then MID on q with + r2[p] * parity_pos_factor[p]
// and the plan's DIT tile.  q, r1, r2: k blocks each; the result, x p'(x) at the data positions, is left in q.
bool split_decode_supported(const fastecc_ctx* c);
uint32_t split_decode_groups(const fastecc_ctx* c);
uint32_t split_decode_group_rows(const fastecc_ctx* c);
int split_impulse_max();  // IMPULSE_MAX of kernels.hpp
// odd != null (fastecc_repair, (2k,k) layout): x p'(x) at the odd positions as well — MID's second half and a DIT once more over the same two halves.
struct SplitRepair {
    uint32_t* q2;                      // k blocks: the data half after ALL its DIF levels, stored by the first chain's MID on its way; the second chain works there
    const uint32_t* data_pos_factor;   // k words by position: -w^m / 2 at position bitrev(m)
    const uint32_t* out_rows_factor;   // k words, tile order: 1 / (w^(2j+1) l'(w^(2j+1))) of the lost parity blocks, 0 elsewhere
    uint32_t* out;                     // the parity stripe
};
// out_rows_factor != null: the DIT tile stores only the blocks with a non-zero factor (tile order), times it, to out[] (the decoder's scatter).
int run_split_decode(fastecc_ctx* c, const uint32_t* data, const uint32_t* parity, const uint32_t* data_rows_factor, const uint32_t* parity_rows_factor,
                     uint32_t parity_groups, const uint32_t* parity_pos_factor, uint32_t* q, uint32_t* r1, uint32_t* r2, const uint32_t* out_rows_factor,
                     uint32_t* out, const uint32_t* impulse_table, uint32_t data_blocks, uint32_t parity_blocks, hipStream_t st,
                     const SplitRepair* odd = nullptr, const uint32_t* small_addend = nullptr, uint32_t addend_shift = 0);
// small_addend != null (1 <= addend_shift <= 5): the parity blocks in use sit at multiples of 2^shift of the parity half only.  The DIF of such a
// stripe is the DIF of its (k >> shift) non-zero rows with every result block repeated 2^shift times (the last `shift` levels pair a value with a
// zero under the twiddle 1), so the caller has transformed those rows alone — small_addend, (k >> shift) blocks in that transform's own bit-
// reversed order — and parity, parity_rows_factor, parity_groups, r1, r2 and impulse_table are not used.
// data_blocks / parity_blocks: the blocks the two stripes really hold (<= k: zero-extended codes; the rest counts as zero blocks)
// impulse_table (optional): [IMPULSE_MAX][16][64] words (Montgomery form), entry [t][g][c] = block g + 16 c of a 1024-block tile after the DIF
// levels with strides 512 ... 16 when block g + 16 t alone was 1 — with at most 16 IMPULSE_MAX parity groups in use those six of the parity
// half's low levels are a multiply-add per word and block in use (MODE_DIF_IMPULSE)
// The k-block work stripe of a fold > 0 / multi-coset context (allocated on first use); a caller may build its input
// there and pass it as `data` to fastecc_encode, which then runs the DIF half in place.
int scratch_of(fastecc_ctx* c, uint32_t** out);
// GF((2^61-1)^2) codes other than (2N,N) work on padded copies of the caller's stripes: the context's two N-block work stripes
int p61_work_stripes(fastecc_ctx* c, uint64_t** data_full, uint64_t** parity_full);
// fastecc_encode on DEVICE memory for a caller that already holds the context's call lock (decode.hip: fastecc_repair)
int encode_unlocked(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st);

// Calls on one context are serialised on the host, and work that uses the context's internal device buffers is ordered
// between streams (encode.hip: fastecc_ctx::mu / buf_event).  decode.hip's entry points take part through this scope:
// the constructor locks, begin() makes `st` wait for the previous user of the internal buffers, end() records this one.
class CallScope {
public:
    explicit CallScope(fastecc_ctx* c);
    ~CallScope();
    int begin(hipStream_t st);
    int end(hipStream_t st);
    int wait_idle();  // host-side wait for the last recorded use (instead of a device-wide synchronise)
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;
private:
    fastecc_ctx* c_;
};

}  // namespace fastecc
