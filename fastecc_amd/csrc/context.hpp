// context.hpp — what the translation units behind the C ABI share (api.hip: entry points; encode.hip / host_stage.hip / create.hip: the drivers behind them, drivers.hpp; plan.hip: pass plans
// and twiddle tables; options.hip: tuning options and profiling).  Nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/fastecc.h"
#include "gf.hpp"
#include "gf61.hpp"
#include "gf61_path.hpp"
#include "internal.hpp"
#include "kernels.hpp"

namespace fastecc {

struct Pass {
    int mode;   // MODE_DIF / MODE_DIT / MODE_MID
    int logr;   // levels covered by the pass (register pass: held in VGPRs; tile pass: log2 of the tile rows)
    int s;      // log2 of the smallest stride
    bool tile;  // LDS-tiled kernel (tile_kernels.hip) instead of a register pass (kernels.hip)
    bool pair;  // tile only: 32-word rows with the cross-lane top level
    int rlog;   // tile only: log2 of the words a lane keeps in registers (5, or 4 for the slim outer tiles)
    int wide = 0;  // tile only: the tile's blocks span >= 2^32 bytes and are addressed through this many windows (2, 4, 8)
    int fused = 0;  // > 0: this outer tile also does the odd-radix level of a mixed-radix context (mixed_kernels.hip: fused_radix_kernel,
                    // launched by encode_mixed; run_passes skips it).  rlog = its register run, pair = false.
};

struct ProfileRec {
    std::string name;
    hipEvent_t start, stop;
    uint64_t bytes;  // algorithmic bytes of the launch (stripe or slab read once + written once)
};

// What one call adds to the plain "read `in`, write `out`" form of run_passes.
struct CallBounds {
    // zero-extended codes: the first pass reads a stripe of in_rows blocks (the rest is zero), the last pass writes the
    // first out_rows blocks of its result to final_out
    uint32_t in_rows = 0, out_rows = 0;
    uint32_t* final_out = nullptr;
    // decoder (run_gathered): see PassArgs::in_odd / row_factor
    const uint32_t* gather_odd = nullptr;
    const uint32_t* gather_factor = nullptr;
    bool dscale_whole = false;  // batch > 1: the per-block factor table covers the whole batch (mixed-radix transforms)
    // split decoder (run_split_decode): per-block factors of a DIF tile that starts the plan (tile order; MODE_DIF_ROWS), the number of
    // block groups of that pass to run (0 = all), and the MID tile's addend stripe with its per-position factors (MODE_MID_ADD)
    const uint32_t* rows_factor = nullptr;
    uint32_t groups = 0;
    const uint32_t* addend = nullptr;
    const uint32_t* addend_factor = nullptr;
    uint32_t addend_shift = 0;  // TileArgs::addend_shift
    const uint32_t* impulse_table = nullptr;    // a lone DIF tile at s = 0 whose input is zero from block impulse_rows on (MODE_DIF_IMPULSE)
    uint32_t impulse_rows = 0;
    uint32_t* keep = nullptr;                   // MODE_MID_ADD also stores its tiles as they are after the first half (TileArgs::keep)
    bool mid_up = false;                        // the plan's MID pass runs its second half only, on such stored tiles (MODE_MID_UP)
    const uint32_t* dscale_override = nullptr;  // MID's per-block factors from this table instead of the context's
    const uint32_t* rows_out_factor = nullptr;  // the plan's last pass, a DIT tile, stores only the blocks with a non-zero factor, times it, to final_out (MODE_DIT_ROWS)
};

}  // namespace fastecc

using namespace fastecc;  // private header of three translation units that all do this

struct fastecc_ctx {
    int device = 0;
    int field = FASTECC_FIELD_GF_FFF00001;
    // Calls on one context are serialised on the host (every entry point that enqueues work holds `mu`), and nothing
    // about a call is stored here: what varies per call travels in a CallBounds on the caller's stack.  Work that
    // uses the context's internal device buffers (scratch, parbuf, dbuf, ...) on a stream other than the previous one
    // first waits for the previous use (buf_event), so one context may be driven from several streams.
    std::mutex mu;
    hipEvent_t buf_event = nullptr;
    hipStream_t buf_stream = nullptr;
    bool buf_used = false;
    DecodeState* decoder = nullptr;  // fastecc_decode_prepare: erasure pattern tables (decode.hip)
    Sharded* sharded = nullptr;      // fastecc_create_sharded: the per-device contexts of the column slabs (sharded.hip); a
                                     // context that has it is only a shell around them
    p61::Decoder* decoder61 = nullptr;  // the same for FASTECC_FIELD_GF_P61_SQUARED (gf61_decode.hip)
    int p61_stride = 1;  // 64-bit field, codes other than (2N,N): parity block j of the code is block j * p61_stride of the (2N,N) parity (N = 2^n)
    p61::Path* p61 = nullptr;  // FASTECC_FIELD_GF_P61_SQUARED: tables and plan of gf61_kernels.hip (everything uint32 below is unused)
    uint64_t N = 0;   // k
    int n = 0;        // log2 k
    uint64_t S = 0;   // words per block
    uint64_t ld = 0;  // words between consecutive blocks in DEVICE stripes (row pitch, >= S; S unless "row_pitch_words" is set)
    size_t stripe_bytes = 0;
    // Fewer parity than data blocks: n - k = M = k >> fold.  Parity block j of that code is parity block j << fold of
    // the (2k, k) code (same polynomial, a sub-coset of the evaluation points), so the DIF half is unchanged, the
    // MID pass keeps every 2^fold-th output and the DIT passes above it run as a size-M transform on the compact buffer.
    int fold = 0;
    // Any (n,k) by zero extension (RS.md:23-33 steps 1-7): K data blocks are the first K of N = 2^ceil(log2 K) (the rest
    // are zero blocks that never exist in memory), and the Mu requested parity blocks are the first Mu of the M = N >> fold
    // computed ones.  K == N and Mu == M in the power-of-two configurations.
    uint64_t K = 0, Mu = 0;
    // More parity than data blocks: n = 2^e k, e = 2 or 3.  The n - k parity blocks are the values of the same
    // polynomial on the 2^e - 1 cosets g_t * <w_k> of the data points inside the n-th roots of unity, ordered so that
    // codes nest: coset 0 is the reference's w_2k (the (2k,k) parity), then w_4k, w_4k^3, then w_8k, w_8k^3, w_8k^5, w_8k^7.
    int cosets = 1;
    // Transform order q * N with an odd q (3, 5, 7, 9): the odd-radix level is the outermost one (mixed_kernels.hip) and
    // N = 2^n is what everything else in this structure describes — q stripes of N blocks back to back.  K <= q*N data
    // blocks (zero-extended), the first Mu <= q*N parity blocks are handed out.  q = 1: an ordinary context.
    int q = 1;
    uint32_t* q_tw_dif = nullptr;   // N x (q-1): w_(qN)^-(i2*j)
    uint32_t* q_tw_dit = nullptr;   // N x (q-1): w_(qN)^+(i2*j)
    uint32_t* q_dft_inv = nullptr;  // q x q: w_q^-(i*j)
    uint32_t* q_dft_fwd = nullptr;  // q x q: w_q^+(i*j)
    uint32_t* mixbuf = nullptr;     // q*N-block work stripe for callers whose parity buffer is shorter (lazy)
    uint64_t M = 0;             // parity blocks
    size_t parity_bytes = 0;    // M * block_bytes
    uint32_t* scratch = nullptr;      // fold > 0: k-block work stripe for the DIF half (lazy)
    uint32_t* tw_fold_dit = nullptr;  // fold > 0: forward roots of order M, level-packed for the DIT passes above MID

    // device tables (Montgomery form, see gf.hpp)
    // level-packed twiddle tables (ntt_device.hpp), N words each, rebuilt whenever the plan changes:
    uint32_t* tw_enc_dif = nullptr;  // inverse roots, ordered for the encode plan's DIF/MID passes
    uint32_t* tw_enc_dit = nullptr;  // forward roots, same ordering (the DIT passes mirror the DIF ones)
    uint32_t* tw_ntt_fwd = nullptr;  // forward roots, ordered for the stand-alone transform's passes
    uint32_t* tw_ntt_inv = nullptr;  // inverse roots, same ordering
    unsigned tw_ready = 0;           // which of the five tables hold the current plan's values (bit TW_*): each is built on the device at first use
    unsigned tw_pending = 0;         // built by a kernel that may still be running: tw_event[i] marks its end, tw_stream[i] the stream it ran on
    hipEvent_t tw_event[5] = {};
    hipStream_t tw_stream[5] = {};
    uint32_t* dscale = nullptr;  // position p -> w_2N^i / N with i = bitrev_n(p)     (RS.cpp:51-54)
    uint32_t* factor = nullptr;  // scratch for fastecc_scale_blocks, N words
    uint32_t* dbuf = nullptr;    // staging stripe for FASTECC_MEM_HOST calls (lazy)
    uint32_t* parbuf = nullptr;  // Mu < M: the M computed parity blocks, of which the first Mu are handed out (lazy)
    uint32_t* hostpar = nullptr; // device parity for FASTECC_MEM_HOST encodes with more parity than data blocks (lazy)
    uint32_t* rawbuf = nullptr;  // staging for the raw side of fastecc_pack_blocks / _unpack_blocks on host memory (lazy)
    // pageable host memory (FASTECC_MEM_HOST results, fastecc_encode_blocks) moves through rings of pinned slots served by helper threads (lazy)
    static constexpr int STAGE_SLOTS = 4;
    static constexpr size_t STAGE_SLOT_BYTES = (size_t)16 << 20;
    struct StageRing {  // pinned slots between pageable host memory and the copy engine, one ring per direction (host_stage.hip stage_transfer)
        char* slots = nullptr;
        hipEvent_t event[STAGE_SLOTS] = {};
    };
    StageRing stage_up, stage_down;

    int rmax = 5;            // levels per register pass
    int vec = 1;             // words per lane in register passes
    int tile_mid = 10;       // > 0: LDS-tiled plan, MID covers min(n, tile_mid) levels
    bool outer64 = false;    // plans 4000+: 9-level outer chunks as 64-word-row tiles with the two-round exchange
    bool plan_auto = true;   // plan id 0: the plain (2k,k) encoder may give MID fewer levels (plan.hip build_plans)
    bool classic_plan = false;  // contexts the decoder builds: MID keeps tile_mid levels (the split transform's shapes)
    bool tile_mid_wide = false;  // MID tile with 64-word rows instead of the 32-word pair form
    bool split2 = true;      // 1024-block tiles exchange through a 64 KiB LDS buffer in two column rounds
    int cache_policy = 15;   // tile kernels: bit 0/1 non-temporal loads/stores in the outer passes, bit 2/3 the same in MID
    int xcd_swizzle = 1;     // tile kernels: 1 = each XCD takes a contiguous run of column chunks, 2 = whole block groups
    int host_slabs = 8;      // column slabs of a FASTECC_MEM_HOST_PINNED encode (upload / kernels / download pipeline)
    int stage_threads = 0;   // helper threads per staging ring (0: min(6, hardware threads / 4))
    int host_pipeline = 0;   // 1: FASTECC_MEM_HOST encodes of large stripes run the column-slab pipeline through the staging rings (0: upload, encode, download in turn)
    DirectEncode* direct_enc = nullptr;  // n - k <= encode_direct_max: the parity straight from the Lagrange basis (direct.hip), built on first use
    int encode_direct_max = 160;  // ... with the MFMA kernel; stripes it cannot take (odd or misaligned rows) stop at 32
    int decode_split = 1;         // (2k,k) codes: 1 = the decoder's transform as two half-size ones (decode.hip, "even / odd split"), 0 = one of size 2k
    int decode_direct_max = 256;  // up to this many lost blocks are recomputed directly (direct.hip), 0 = always the transform; 96 without the MFMA kernel
    int direct_kernel = 0;        // 0 choose, 1 VALU, 2 MFMA
    int slab_mode = 0;       // how `slabs` > 1 are scheduled (fastecc_set_option "slab_mode")
    int slabs = 1;           // > 1: encode in this many column slabs on internal streams, staggered by one pass,
                             // so the VALU-bound MID of one slab runs beside the HBM-bound outer passes of others
    static constexpr int MAX_SLABS = 32;
    hipStream_t slab_stream[MAX_SLABS] = {};
    hipEvent_t slab_fork = nullptr, slab_first_done[MAX_SLABS] = {}, slab_done[MAX_SLABS] = {};
    bool slab_ready = false;
    int fuse_radix = 1;      // mixed-radix contexts: fuse the odd-radix level into the outermost tile where a shape exists (option "fuse_radix")
    bool slim_outer = true;  // outer 8/9-level tiles keep 16 words per lane instead of 32 (twice the waves per CU)
    bool persistent = true;  // tile kernels as persistent workgroups (one grid of resident workgroups)
    int cus = 256;           // compute units of the device (sizes the persistent grids)
    std::vector<Pass> encode_plan, ntt_plan;
    std::string plan_text;

    bool profiling = false;
    std::vector<ProfileRec> prof;
    size_t prof_used = 0;
};

namespace fastecc {

extern thread_local char g_detail[256];  // fastecc_last_error_detail
int hip_fail(hipError_t e, const char* what);
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) return hip_fail(e_, #expr); \
    } while (0)

int ilog2_exact(uint64_t v);
uint32_t bitrev_host(uint32_t v, int bits);

// plan.hip
void build_plans(fastecc_ctx* c);
const char* pass_name(const Pass& p, int vec, char* buf, size_t cap);
std::vector<int> level_strides(const std::vector<Pass>& plan, int n, bool up = false);
std::vector<uint32_t> build_level_table(int n, uint32_t root_of_order_N, const std::vector<int>& sl);
int upload_table(uint32_t** dst, const std::vector<uint32_t>& src);
int upload_twiddles(fastecc_ctx* c);  // the plans changed: every twiddle table is rebuilt at its next use; the device must be idle w.r.t. this context
// The table `which` for the current plans, built on the context's device (the current device) at first use: a context that only ever encodes
// never builds the stand-alone transform's tables and vice versa (the decoder's ~19 internal contexts each use one kind).  nullptr on failure.
enum { TW_ENC_DIF = 0, TW_ENC_DIT = 1, TW_NTT_FWD = 2, TW_NTT_INV = 3, TW_FOLD_DIT = 4 };
// The table is written by a kernel enqueued on `st` — the stream that is about to use it — with NO host synchronisation: the first call on a
// context stays asynchronous (and may run under stream capture).  A later use on another stream waits for the build's event on the device.
const uint32_t* twiddle_table(fastecc_ctx* c, int which, hipStream_t st);

using CallLock = std::lock_guard<std::mutex>;

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};


}  // namespace fastecc
