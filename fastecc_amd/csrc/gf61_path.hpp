// gf61_path.hpp — the encode path over GF((2^61-1)^2) (gf61_kernels.hip), as seen by the C ABI (api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fastecc {
namespace p61 {

struct Path;  // tables + plan for one (k, elements per block) on one device

// api.hip brackets every kernel launch with these so that fastecc_profile_* covers this field too.
struct LaunchHooks {
    void* user;
    void (*begin)(void* user, hipStream_t st, const char* kernel, uint64_t algorithmic_bytes);
    void (*end)(void* user, hipStream_t st);
};

// n = log2 k (1..MAX_LOG2_K); elems = GF(p^2) elements (16 bytes) per block.  Returns a FASTECC_* code; on
// failure `detail` gets a short message.  The current device must already be the target device.
constexpr int MAX_LOG2_K = 24;   // 3 tables of 16 * k bytes
constexpr int DECODE_DIRECT_MAX = 32;  // the decoder's direct path: most lost blocks (option "decode_direct_max"): 1.84 ms each at k = 2^19 x 64 KB, the transform path 71 ms
constexpr int DEFAULT_LEVELS = 4;
int create(Path** out, int n, uint64_t elems, char* detail, size_t detail_cap);
// The same pipeline — DIF over all levels with the inverse roots, the block holding coefficient m times its factor, DIT back
// with the forward roots — with the factor m / 2^n (FACTOR_INDEX: the decoder's x p'(x) transform, gf61_decode.hip; no root of order
// 2^(n+1) is needed) instead of the encoder's w_2N^m / N (FACTOR_ENCODE).  All tables of a path are built by kernels.
enum { FACTOR_ENCODE = 0, FACTOR_INDEX = 1, FACTOR_SPLIT = 2 };  // FACTOR_SPLIT: (2 m + 2^n) / 2^(n+1), the split decoder's data chain
int create_transform(Path** out, int n, uint64_t elems, int factor, char* detail, size_t detail_cap);
// n = 4k / 8k (cosets = 3 / 7 further cosets of evaluation points, include/fastecc.h's nesting order): encode_cosets only
int create_cosets(Path** out, int n, uint64_t elems, int cosets, char* detail, size_t detail_cap);
int cosets_of(const Path* p);
// create_transform with the MID pass forced to `force_mid` levels (0 = the plan's own choice, also when no such plan exists)
int create_transform_mid(Path** out, int n, uint64_t elems, int factor, int force_mid, char* detail, size_t detail_cap);
// Only the EVEN output positions of `big`'s transform (size 2^(n+1), created with force_mid = 7): DIF passes of `big`, a MID tile that folds
// the first DIT level away, DIT passes of `half` (size 2^n, force_mid = 6).  With `big` of size 2^(n+2) and `half` created with force_mid = 5:
// every FOURTH output position (two levels fold away; the decoder of the n = 4k codes).  in: 2^(n+1) blocks; work: 2^(n+1) blocks (may be `in`); out: 2^n blocks.
// FASTECC_E_UNSUPPORTED if the plans of the two do not pair up.
// ends (optional): the decoder's gather and scatter run inside the first and the last pass —
//   fin  != null: `in` is the data stripe, `parity` the parity stripe; position u of the input is (u even ? data : parity)[u / 2] * fin[u], read
//                 only where fin[u] != 0 (needs FOLD_GATHERS);
//   gout != null: row i of the result is written to data_out[i], times gout[i], canonical, only where gout[i] != 0 (needs FOLD_SCATTERS);
//                 `out` then holds intermediate values only.
struct FoldEnds {
    const uint64_t* parity = nullptr;
    const uint64_t* fin = nullptr;
    const uint32_t* map = nullptr;  // with fin: position u of the input is block map[u] & 0x7FFFFFFF of `in` (bit 31: of `parity`) — the n = 4k layout
    const uint64_t* gout = nullptr;
    uint64_t* data_out = nullptr;
};
enum { FOLD_PAIRS = 1, FOLD_GATHERS = 2, FOLD_SCATTERS = 4 };
int fold_caps(Path* big, Path* half);  // what encode_fold can do with these two paths
int encode_fold(Path* big, Path* half, const uint64_t* in, uint64_t* work, uint64_t* out, hipStream_t st, const LaunchHooks* hooks,
                const FoldEnds* ends = nullptr);
// fastecc_repair in ONE transform (a path made by create_transform with FACTOR_INDEX, size 2k): see gf61_kernels.hip
int encode_ends(Path* p, const uint64_t* data, const uint64_t* parity, const uint64_t* fin, uint64_t* work, const uint64_t* gout_all, uint64_t* data_out,
                uint64_t* parity_out, hipStream_t st, const LaunchHooks* hooks);
void destroy(Path* p);

int encode(Path* p, const uint64_t* data, uint64_t* parity, hipStream_t st, const LaunchHooks* hooks);
// a create_cosets path: parity = cosets * k blocks; work = a k-block stripe of the caller's (needed when encode_cosets_needs_work)
// coset_mask: bit t set = coset t (k blocks of the parity) is computed and written; the others are left alone
int encode_cosets(Path* p, const uint64_t* data, uint64_t* parity, uint64_t* work, hipStream_t st, const LaunchHooks* hooks, uint32_t coset_mask = 0xFFFFFFFFu);
bool encode_cosets_needs_work(const Path* p);
// the element columns [col0, col0 + width) of every block only (data / parity are the stripes' base addresses)
int encode_columns(Path* p, const uint64_t* data, uint64_t* parity, uint64_t col0, uint64_t width, hipStream_t st, const LaunchHooks* hooks);
int ntt(Path* p, uint64_t* data, bool inverse, hipStream_t st, const LaunchHooks* hooks);
// the same without the closing block permutation: block q of the result holds coefficient bitrev(q)
int dif_only(Path* p, uint64_t* data, bool inverse, hipStream_t st, const LaunchHooks* hooks);
int dif_only_to(Path* p, const uint64_t* in, uint64_t* out, bool inverse, hipStream_t st, const LaunchHooks* hooks);  // the same, out of place
// ---- the decoder's even / odd split (gf61_decode.hip): the data chain on a path made with FACTOR_SPLIT ----
bool split_decode_supported(const Path* p);
int split_decode(Path* p, const uint64_t* data, const uint64_t* rows_factor, uint32_t rows_stride, const uint64_t* addend, int addend_shift,
                 const uint64_t* addend_factor, uint64_t* work, const uint64_t* gout, uint64_t* data_out, hipStream_t st, const LaunchHooks* hooks,
                 uint64_t* keep = nullptr);  // keep: a k-block stripe that receives q~ (the tiles after MID's first half) for split_repair_parity
int split_repair_parity(Path* p, const uint64_t* keep, const uint64_t* data_factor, const uint64_t* addend, int addend_shift, uint64_t* work,
                        const uint64_t* gout_par, uint64_t* parity_out, hipStream_t st, const LaunchHooks* hooks);
int split_addend_factors(uint64_t* table, int n, hipStream_t st, bool forward = false);
// words >= p among the 2 * elems * k words of a stripe; `counter` is a device uint64 the caller zeroed
int count_out_of_range(Path* p, const uint64_t* data, unsigned long long* counter, hipStream_t st);

// plan id (fastecc_set_plan): 0 = default (LDS tiles + 4-level register passes), 1..4 = register passes only with that many
// levels, 10 + L / 20 + L = tiles with a 64 / 128 KiB exchange buffer; rebuilds the tables.  The device must be idle.
int set_plan(Path* p, int plan, char* detail, size_t detail_cap);
const char* plan_string(const Path* p);

// ---- erasure decoder over this field (gf61_decode.hip): the same scheme as decode.hip, (2k,k) codes ----
struct Decoder;
void destroy_decoder(Decoder* d);
// data_present / parity_present: k flags each (non-zero = the block survives).  Synchronous.  The current device must be
// the target device.  *slot is created on first use and reused.
// direct_max: patterns with at most that many lost blocks (<= 16) get the direct one-pass path instead of locator + transform.
// split != 0: patterns that lose data blocks run the even / odd split where a plan of the needed shape exists (k >= 2^11; gf61_decode.hip)
// e = 1: the (2k,k) code; e = 2 / 3: n = 4k / 8k (parity_present then has (2^e - 1) k flags in the stripe's block order; no split, no direct path;
// a slot serves one e)
int decode_prepare(Decoder** slot, int log2k, uint64_t elems, const uint8_t* data_present, const uint8_t* parity_present, int direct_max, char* detail,
                   size_t detail_cap, int split = 1, int e = 1);
// Recover the erased data blocks in place (device pointers, enqueued on st); rebuild_with != null: also re-encode with that
// path (the context's encoder) and write the lost parity blocks into `parity`.
int decode(Decoder* d, uint64_t* data, uint64_t* parity, Path* rebuild_with, hipStream_t st, const LaunchHooks* hooks);
// host-memory stripes: staged through device buffers of the decoder, synchronous
int decode_host(Decoder* d, void* data, void* parity, Path* rebuild_with, hipStream_t st, const LaunchHooks* hooks);
bool decoder_ready(const Decoder* d);

}  // namespace p61
}  // namespace fastecc
