// plan.hip — how the levels of a transform are cut into GPU passes, and the twiddle tables those passes read.
//
// Where the reference picks an R x C (x L) split so that a sub-transform fits the CPU's L2 (ntt.cpp:385-394), the plan here picks how many
// radix-2 levels each GPU pass keeps in registers or in an LDS tile; the level-packed tables replace the roots[] array of ntt.cpp:397-402 and
// the running root_i *= root of ntt.cpp:270-281.  Host code only: fastecc_plan_describe / fastecc_plan_twiddles run without a device.
#include "context.hpp"

using namespace fastecc;

namespace fastecc {

namespace {

// Split `bits` levels into ceil(bits/rmax) passes of near-equal size, largest first.
std::vector<int> split_levels(int bits, int rmax)
{
    std::vector<int> r;
    if (bits <= 0) return r;
    const int q = (bits + rmax - 1) / rmax;
    for (int i = 0; i < q; i++) r.push_back(bits / q + (i < bits % q ? 1 : 0));
    return r;
}

// How a run of `bits` consecutive levels is executed: an LDS tile when one exists for that size,
// register passes otherwise.
// A tile pass addresses its tile with 32-bit offsets from a per-tile buffer descriptor (tile_kernels.hip).
bool tile_fits(const fastecc_ctx* c, int logt, int s)
{
    // block offsets (SGPR) and lane offsets (VGPR) are 32-bit and their sum must stay below num_records = 2^32-1
    return (((uint64_t)c->ld * 4) << (logt + s)) <= 0xFFFF0000ull;
}

void push_chunk(std::vector<Pass>& plan, int mode, int bits, int s, const fastecc_ctx* c)
{
    // with fold > 0 the DIT passes above MID run on the compact parity stripe: their strides are 2^fold smaller
    const int s_run = mode == MODE_DIT ? s - c->fold : s;
    const bool fits = tile_fits(c, bits, s_run);
    // larger spans: 2, 4 or 8 address windows per tile (tile_kernels.hip NWIN), only for the outer pair shapes that have them
    int windows = 0;
    for (int lw = 1; lw <= 4 && !fits && windows == 0; lw++)
        if (bits - lw >= 1 && tile_fits(c, bits - lw, s_run)) windows = 1 << lw;
    if (c->tile_mid > 0 && fits && c->outer64 && c->split2 && bits == 9 && tile_supported(bits, false)) plan.push_back({mode, bits, s, true, false, 5});  // 64-word rows, split buffer
    else if (c->tile_mid > 0 && windows && c->slim_outer && tile_supported(bits, true, 4) && tile_max_windows(bits, true, 4) >= windows)
        plan.push_back({mode, bits, s, true, true, 4, windows});
    else if (c->tile_mid > 0 && windows && tile_supported(bits, true) && tile_max_windows(bits, true) >= windows)
        plan.push_back({mode, bits, s, true, true, 5, windows});
    else if (c->tile_mid > 0 && fits && c->slim_outer && tile_supported(bits, true, 4)) plan.push_back({mode, bits, s, true, true, 4});
    else if (c->tile_mid > 0 && fits && bits >= 7 && tile_supported(bits, true)) plan.push_back({mode, bits, s, true, true, 5});
    else if (c->tile_mid > 0 && fits && bits == 6 && tile_supported(bits, false)) plan.push_back({mode, bits, s, true, false, 5});
    else if (mode == MODE_DIT) {
        int ss = s;
        const std::vector<int> parts = split_levels(bits, c->rmax);
        for (auto it = parts.rbegin(); it != parts.rend(); ++it) {
            plan.push_back({mode, *it, ss, false, false, 0});
            ss += *it;
        }
    } else {
        int ss = s + bits;
        for (int r : split_levels(bits, c->rmax)) {
            ss -= r;
            plan.push_back({mode, r, ss, false, false, 0});
        }
    }
}

}  // namespace

static void build_plans_with(fastecc_ctx* c, int tile_mid);

// The default plan (fastecc_set_plan 0) of the power-of-two codes gives MID fewer levels than it could take: MID is bound by VALU issue,
// the outer passes by HBM with arithmetic to spare, so at k = 2^19 the split dif10 / mid9 / dit10 runs in 3.33 ms where dif9 / mid10 / dit9
// takes 3.47 (profiles/r04/plan_sweep_mid_levels.jsonl: k = 2^16 ... 2^19; it needed the 1024-block outer tiles without scratch and the
// 512-block MID tile at four workgroups per CU).  Contexts the decoder builds (their passes are the split transform's: MID10 between slim
// outer tiles), the mixed-radix orders and explicit plan ids keep MID at `tile_mid` levels; so does any size whose outer chunk has no tile.
void build_plans(fastecc_ctx* c)
{
    // (codes with fewer parity blocks, n = 4k / 8k and zero extension gain as much or more: 2.86 -> 2.62 ms at 2^19 + 2^18, 2.25 -> 2.05 at 2^19 + 2^16,
    //  2.48 -> 2.22 at 400000 + 100000; the mixed-radix orders do not — their fused outer passes are VALU-bound themselves: 13 x 2^15 3.25 -> 3.87;
    //  blocks below 2 KB keep MID10 too: k = 2^19 x 1 KB measured 0.87 ms with it against 0.90; 2 KB, 2052 B, 4 KB, 4100 B gain 2-4 %)
    const bool plain = c->plan_auto && !c->classic_plan && c->tile_mid == 10 && !c->tile_mid_wide && c->split2 && c->slim_outer && c->q <= 1 && !c->p61 && c->S >= 512;
    const int shorter = c->n >= 17 ? 9 : c->n == 16 ? 8 : 0;
    if (plain && shorter) {
        // k = 2^18: both outer chunks have 9 levels, the shape that exists with 64-word rows (256-byte pieces of a block per request:
        // 0.435 ms per pass against 0.47 with 32-word rows; plan 4090) — at 2^19 the outer chunks need 10 levels, which only the 32-word tile has
        // (only 9-level chunks are affected: the encoder's at 2^18, the stand-alone transform's second chunk at 2^19)
        const bool outer64 = c->outer64;
        c->outer64 = true;
        build_plans_with(c, shorter);
        c->outer64 = outer64;
        bool tiles = c->encode_plan.size() == 3;
        for (const Pass& p : c->encode_plan) tiles = tiles && p.tile && p.wide == 0;  // (tiles of several address windows: not measured in this split)
        if (tiles) return;
    }
    // Narrow blocks at k = 2^19 (64 .. 511 words in whole 64-word rows: the sub-slabs of ONE stripe spread over 2, 4 or 8 GPUs):
    // 128- and 256-byte pieces of a block per request make the slim 32-word outer tiles the slow part, so the 9-level outer chunks run as tiles
    // of 64-word rows around MID10 (plan 4100's shape): 256 B blocks 4.38 -> 4.06 ms per 2 GiB-equivalent, 512 B 4.04 -> 3.94, 1 KB 3.79 -> 3.62
    // (profiles/r05/narrow_block_plans.jsonl; at 2 KB and above the rule above holds, at 128 B there are no 64-word rows).
    const bool narrow = c->plan_auto && !c->classic_plan && c->tile_mid == 10 && !c->tile_mid_wide && c->split2 && c->slim_outer && c->q <= 1 && !c->p61 &&
                        c->fold == 0 && c->cosets == 1 && c->n == 19 && c->S >= 64 && c->S < 512 && (c->S % 64) == 0 && c->ld == c->S;
    if (narrow) {
        const bool outer64 = c->outer64;
        c->outer64 = true;
        build_plans_with(c, c->tile_mid);
        c->outer64 = outer64;
        bool tiles = c->encode_plan.size() == 3;
        for (const Pass& p : c->encode_plan) tiles = tiles && p.tile && p.wide == 0;
        if (tiles) return;
    }
    build_plans_with(c, c->tile_mid);
}

static void build_plans_with(fastecc_ctx* c, int tile_mid)
{
    const int n = c->n;
    c->encode_plan.clear();
    c->ntt_plan.clear();
    // encode: DIF over the high levels, MID over the low levels, DIT back up (kernels.hip header)
    int mid = std::min(n, c->rmax);
    bool mid_tile = false, mid_pair = false;
    if (c->tile_mid > 0) {
        const int want = std::min(n, tile_mid);
        if (!tile_fits(c, want, 0)) {
            // blocks too large for 32-bit tile offsets: register passes handle the low levels
        } else if (tile_supported(want, !c->tile_mid_wide) && tile_max_fold(want, !c->tile_mid_wide) >= c->fold) {
            mid = want, mid_tile = true, mid_pair = !c->tile_mid_wide;
        } else if (tile_supported(want, c->tile_mid_wide) && tile_max_fold(want, c->tile_mid_wide) >= c->fold) {
            mid = want, mid_tile = true, mid_pair = c->tile_mid_wide;
        }
    }
    if (!mid_tile && mid < c->fold) mid = std::min(n, c->fold);  // a register MID pass drops blocks within its own 2^mid
    const int max_chunk = c->tile_mid > 0 ? 10 : c->rmax;
    const std::vector<int> outer = split_levels(n - mid, max_chunk);
    // mixed radix: one outer chunk and a fused shape for it -> the odd-radix level rides on that pass (3 trips instead of 5)
    // (the fused kernel addresses the batch of q * 2^n blocks through one buffer descriptor: 32-bit offsets)
    const bool stripe_fits = fused_batch_fits(c->ld, c->S, (uint64_t)std::max(c->q, 1) * c->N);
    const int fused_run = (c->q > 1 && c->fuse_radix && outer.size() == 1 && stripe_fits) ? fused_rlog(c->q, outer[0]) : 0;
    int s = n;
    for (int r : outer) {
        s -= r;
        if (fused_run) c->encode_plan.push_back({MODE_DIF, r, s, true, false, fused_run, 0, c->q});
        else push_chunk(c->encode_plan, MODE_DIF, r, s, c);
    }
    c->encode_plan.push_back({MODE_MID, mid, 0, mid_tile, mid_pair, 5});
    s = mid;
    for (auto it = outer.rbegin(); it != outer.rend(); ++it) {
        if (fused_run) c->encode_plan.push_back({MODE_DIT, *it, s, true, false, fused_run, 0, c->q});
        else push_chunk(c->encode_plan, MODE_DIT, *it, s, c);
        s += *it;
    }
    // stand-alone transform: DIF over all levels, then the block bit-reversal
    s = n;
    for (int r : split_levels(n, max_chunk)) {
        s -= r;
        push_chunk(c->ntt_plan, MODE_DIF, r, s, c);
    }
    char buf[64];
    c->plan_text.clear();
    for (const Pass& p : c->encode_plan) {
        if (p.fused) {  // R<q>+: the odd-radix level and these levels in one pass
            snprintf(buf, sizeof buf, "%sR%d+%s%d@%d", c->plan_text.empty() ? "" : ",", p.fused, p.mode == MODE_DIF ? "dif" : "dit", p.logr, p.s);
            c->plan_text += buf;
            continue;
        }
        snprintf(buf, sizeof buf, "%s%s%s%d@%d", c->plan_text.empty() ? "" : ",", p.tile ? (p.wide ? (p.rlog == 4 ? (p.wide == 2 ? "SW32:" : p.wide == 4 ? "SW4x32:" : p.wide == 8 ? "SW8x32:" : "SW16x32:") : "TW32:") : p.rlog == 4 ? "S32:" : p.pair ? "T32:" : "T64:") : "",
                 p.mode == MODE_DIF ? "dif" : p.mode == MODE_DIT ? "dit" : "mid", p.logr, p.s);
        c->plan_text += buf;
    }
    if (c->q > 1 && !fused_run) {  // the odd-radix level around the power-of-two pipeline (mixed_kernels.hip)
        snprintf(buf, sizeof buf, "R%d:dif1@%d,", c->q, c->n);
        c->plan_text = std::string(buf) + c->plan_text;
        snprintf(buf, sizeof buf, ",R%d:dit1@%d", c->q, c->n);
        c->plan_text += buf;
    }
    snprintf(buf, sizeof buf, " v%d", c->vec);
    c->plan_text += buf;
}

const char* pass_name(const Pass& p, int vec, char* buf, size_t cap)
{
    const char* m = p.mode == MODE_DIF ? "dif" : p.mode == MODE_DIT ? "dit" : "mid";
    if (p.tile) snprintf(buf, cap, "tile_%s%d_w%d%s", m, p.logr, p.pair ? 32 : 64, p.rlog == 4 ? "_r16" : "");
    else snprintf(buf, cap, "%s%dv%d", m, p.logr, vec);
    return buf;
}

// For every level l: the stride 2^sl of the register run that executes it (see ntt_device.hpp).
// `up` selects the side of an encode plan: false = the way down (DIF passes and MID), true = the way up (MID and DIT
// passes).  The two sides mirror each other level for level unless fold > 0 made a chunk tile-eligible on one side only.
std::vector<int> level_strides(const std::vector<Pass>& plan, int n, bool up)
{
    std::vector<int> sl(n, 0);
    for (const Pass& p : plan) {
        if (p.mode == (up ? MODE_DIF : MODE_DIT)) continue;
        if (!p.tile) {
            for (int l = p.s; l < p.s + p.logr; l++) sl[l] = p.s;
        } else if (p.fused) {
            // fused_radix_kernel: runs of p.rlog levels counted from the top of the pass, the rest in the last one
            const int runs = (p.logr + p.rlog - 1) / p.rlog;
            for (int l = p.s; l < p.s + p.logr; l++) {
                const int run = (p.s + p.logr - 1 - l) / p.rlog;
                sl[l] = run == runs - 1 ? p.s : p.s + p.logr - (run + 1) * p.rlog;
            }
        } else {
            const int l2 = p.logr - p.rlog - (p.pair ? 1 : 0);  // TileCfg::L2
            for (int l = p.s; l < p.s + l2; l++) sl[l] = p.s;
            for (int l = p.s + l2; l < p.s + p.logr; l++) sl[l] = p.s + l2;
        }
    }
    return sl;
}

// Level-packed table: entry 2^l + ((i mod 2^sl) << (l - sl)) + (i >> sl) = (root of order 2^(l+1))^i, i < 2^l,
// in Montgomery form.  Replaces the roots[] array of ntt.cpp:397-402 and the running root_i *= root of
// ntt.cpp:270-281: every twiddle of every level is tabulated once per context.
std::vector<uint32_t> build_level_table(int n, uint32_t root_of_order_N, const std::vector<int>& sl)
{
    std::vector<uint32_t> tab(std::max<size_t>((size_t)1 << n, 2), 0);
    for (int l = 0; l < n; l++) {
        const uint32_t h = 1u << l;
        const uint32_t root = gf::h_pow(root_of_order_N, (uint64_t)1 << (n - 1 - l));
        const int t = l - sl[l];
        const uint32_t lowmask = (1u << sl[l]) - 1u;
        const uint32_t root_m = gf::h_to_mont(root);
        uint32_t w = gf::MONT_ONE;  // running power in Montgomery form: no division per entry
        for (uint32_t i = 0; i < h; i++) {
            tab[h + (((i & lowmask) << t) | (i >> sl[l]))] = w;
            w = gf::h_mont_mul(w, root_m);
        }
    }
    return tab;
}

int upload_table(uint32_t** dst, const std::vector<uint32_t>& src)
{
    if (!*dst) HIP_TRY(hipMalloc((void**)dst, src.size() * 4));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * 4, hipMemcpyHostToDevice));
    return FASTECC_OK;
}

namespace {

// build_level_table on the device: thread (l, i) writes entry 2^l + ((i mod 2^sl) << (l - sl)) + (i >> sl) = (root of order 2^(l+1))^i =
// root_N^(i << (n-1-l)), Montgomery form.  A context needs four or five of these tables; walking 2^n powers per table on the host and
// uploading them was most of what creating a context cost (the decoder's first fastecc_decode_prepare builds ~19 contexts).
struct LevelStrides {
    int sl[32];
};
__global__ __launch_bounds__(256) void level_table_kernel(uint32_t* __restrict__ tab, int n, uint32_t root_N, LevelStrides st)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1u << n)) return;
    if (idx == 0) {
        tab[0] = 0;
        return;
    }
    const int l = 31 - __clz(idx);
    const uint32_t h = 1u << l, i = idx - h;
    const int s = st.sl[l], t = l - s;
    uint32_t r = 1, b = root_N;
    for (uint32_t e = i << (n - 1 - l); e; e >>= 1) {  // e < 2^(n-1)
        if (e & 1u) r = gf::mul(r, b);
        b = gf::mul(b, b);
    }
    tab[h + (((i & ((1u << s) - 1u)) << t) | (i >> s))] = gf::mul(r, gf::MONT_ONE);
}

int device_level_table(uint32_t** dst, int n, uint32_t root_of_order_N, const std::vector<int>& sl, hipStream_t stream)
{
    const size_t entries = std::max<size_t>((size_t)1 << n, 2);
    if (!*dst) HIP_TRY(hipMalloc((void**)dst, entries * 4));
    if (n < 1) {
        HIP_TRY(hipMemsetAsync(*dst, 0, entries * 4, stream));
        return FASTECC_OK;
    }
    LevelStrides st{};
    for (int l = 0; l < n && l < 32; l++) st.sl[l] = sl[l];
    hipLaunchKernelGGL(level_table_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, stream, *dst, n, root_of_order_N, st);
    HIP_TRY(hipGetLastError());
    return FASTECC_OK;
}

}  // namespace

// The plans changed (or the context is new): the tables are rebuilt when they are next used.  Builds that may still be running (and
// their readers, on the same streams) are waited for first, so the old tables are not overwritten under a kernel.
int upload_twiddles(fastecc_ctx* c)
{
    // (the build events cover the builds and the readers on the building stream; readers on other streams hold no event: every caller that
    //  changes a plan on a live context — fastecc_set_plan, fastecc_set_option — waits for the whole device under a DeviceGuard first)
    for (int i = 0; i < 5; i++)
        if ((c->tw_pending & (1u << i)) && c->tw_event[i]) (void)hipEventSynchronize(c->tw_event[i]);
    c->tw_pending = 0;
    c->tw_ready = 0;
    return FASTECC_OK;
}

const uint32_t* twiddle_table(fastecc_ctx* c, int which, hipStream_t st)
{
    uint32_t** slot = which == TW_ENC_DIF ? &c->tw_enc_dif : which == TW_ENC_DIT ? &c->tw_enc_dit : which == TW_NTT_FWD ? &c->tw_ntt_fwd
                    : which == TW_NTT_INV ? &c->tw_ntt_inv : &c->tw_fold_dit;
    const unsigned bit = 1u << which;
    if (c->tw_ready & bit) {
        // built earlier by a kernel on tw_stream: a use on another stream waits for that kernel on the device, until the event is seen complete
        if ((c->tw_pending & bit) && st != c->tw_stream[which]) {
            if (hipEventQuery(c->tw_event[which]) == hipSuccess) c->tw_pending &= ~bit;
            else if (hipStreamWaitEvent(st, c->tw_event[which], 0) != hipSuccess) return nullptr;
            (void)hipGetLastError();  // hipErrorNotReady of the query is not an error
        }
        return *slot;
    }
    // A call that is being captured into a graph must not leave "ready, built on st" behind: the build kernel and its event would be graph
    // nodes, i.e. the table would not exist before the first replay, an eager call on the same stream would read it unbuilt, and a call on
    // any other stream would wait for an event that never completes outside the graph.  The table is then built eagerly, on a stream of its
    // own outside the capture (relaxed capture mode for the allocation and the launch), and waited for here; the captured call only reads it.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) != hipSuccess) {
        (void)hipGetLastError();
        cap = hipStreamCaptureStatusNone;
    }
    const bool capturing = cap != hipStreamCaptureStatusNone;
    hipStream_t caller = st;
    hipStreamCaptureMode prev_mode = hipStreamCaptureModeRelaxed;
    if (capturing) {
        if (hipThreadExchangeStreamCaptureMode(&prev_mode) != hipSuccess) {
            (void)hip_fail(hipGetLastError(), "hipThreadExchangeStreamCaptureMode");
            return nullptr;
        }
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
            (void)hip_fail(hipGetLastError(), "hipStreamCreateWithFlags(twiddle table during capture)");
            (void)hipThreadExchangeStreamCaptureMode(&prev_mode);
            return nullptr;
        }
    }
    auto leave_capture = [&](bool ok) -> bool {
        if (!capturing) return ok;
        if (ok && hipStreamSynchronize(st) != hipSuccess) {
            (void)hip_fail(hipGetLastError(), "hipStreamSynchronize(twiddle table during capture)");
            ok = false;
        }
        (void)hipStreamDestroy(st);
        (void)hipThreadExchangeStreamCaptureMode(&prev_mode);
        st = caller;
        return ok;
    };
    const uint32_t wN = gf::h_root((uint32_t)c->N), wNi = gf::h_inv(wN);
    int rc = FASTECC_OK;
    switch (which) {
    case TW_ENC_DIF: rc = device_level_table(slot, c->n, wNi, level_strides(c->encode_plan, c->n), st); break;        // interpolate: inverse roots (RS.cpp:41)
    case TW_ENC_DIT: rc = device_level_table(slot, c->n, wN, level_strides(c->encode_plan, c->n, true), st); break;   // evaluate (RS.cpp:63)
    case TW_NTT_FWD: rc = device_level_table(slot, c->n, wN, level_strides(c->ntt_plan, c->n), st); break;
    case TW_NTT_INV: rc = device_level_table(slot, c->n, wNi, level_strides(c->ntt_plan, c->n), st); break;
    default: {
        // level l' of the size-M transform is level l' + fold of the size-k one, on positions >> fold
        const std::vector<int> enc_up = level_strides(c->encode_plan, c->n, true);
        const int nf = c->n - c->fold;
        std::vector<int> sl(std::max(nf, 0), 0);
        for (int l = 0; l < nf; l++) sl[l] = std::max(enc_up[l + c->fold] - c->fold, 0);
        rc = device_level_table(slot, nf, gf::h_root((uint32_t)c->M), sl, st);
    }
    }
    if (!leave_capture(rc == FASTECC_OK)) return nullptr;
    if (capturing) {  // complete and visible to every stream: nothing pending, nothing tied to the captured stream
        c->tw_pending &= ~bit;
        c->tw_ready |= bit;
        return *slot;
    }
    // no host synchronisation: the table's first reader follows on the same stream; other streams wait for this event (above)
    hipError_t e = hipSuccess;
    if (!c->tw_event[which]) e = hipEventCreateWithFlags(&c->tw_event[which], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(c->tw_event[which], st);
    if (e != hipSuccess) {  // no event to be had: fall back to waiting here
        (void)hipGetLastError();
        if (hipStreamSynchronize(st) != hipSuccess) {
            (void)hip_fail(hipGetLastError(), "hipStreamSynchronize(twiddle table)");
            return nullptr;
        }
        c->tw_pending &= ~bit;
    } else {
        c->tw_pending |= bit;
        c->tw_stream[which] = st;
    }
    c->tw_ready |= bit;
    return *slot;
}

}  // namespace fastecc

extern "C" {

const char* fastecc_plan_string(fastecc_ctx* c) { return c ? c->plan_text.c_str() : ""; }

// Plan ids:
//   0            default: 3100, or for the power-of-two codes at k >= 2^16 a shorter MID (build_plans: 3090 / 3080 / 4090 by size)
//   rv           register passes only: r levels per pass (1..5), v words per lane (1,2,4), e.g. 51
//   1000+10*a+f  LDS-tiled: MID covers a levels (6..10); f&1: MID tile uses 64-word rows;
//                f&4: never use persistent workgroups (f&2, the next-tile prefetch of rounds 1-2, is gone: such ids are rejected)
//   2000+..      as 1000+ with 16-word-per-lane outer tiles; 3000+..: also the two-round exchange; 4000+..: also 64-word rows for 9-level outer chunks
static int apply_plan(fastecc_ctx* c, int plan)
{
    int rmax = 5, vec = 1, tile_mid = 10;
    bool wide = false, persistent = true, slim = true;  // plan 0 == 2100
    bool split2 = true;  // plan 0 == 3100
    bool outer64 = false;
    if (plan >= 1000) {
        slim = plan >= 2000;  // 2000+10*a+f: as 1000+10*a+f with 16-word-per-lane outer tiles (8/9 levels)
        split2 = plan >= 3000;  // 3000+10*a+f: as 2000+... with the two-round (64 KiB) exchange in 1024-block tiles
        if (plan >= 5000) return FASTECC_E_INVAL;
        outer64 = plan >= 4000;  // 4000+10*a+f: as 3000+... with 9-level outer chunks as tiles of 64-word rows (256-byte pieces of a block) through a split buffer
        tile_mid = (plan % 1000) / 10;
        const int f = (plan % 1000) % 10;
        wide = f & 1;
        persistent = !(f & 4);
        if (f > 7 || (f & 2)) return FASTECC_E_INVAL;
        if (tile_mid < 6 || tile_mid > 10) return FASTECC_E_INVAL;
    } else if (plan != 0) {
        rmax = plan / 10;
        vec = plan % 10;
        tile_mid = 0;
    }
    if (plan < 0 || rmax < 1 || rmax > 5 || (vec != 1 && vec != 2 && vec != 4)) return FASTECC_E_INVAL;
    c->rmax = rmax;
    c->vec = vec;
    c->tile_mid = tile_mid;
    c->tile_mid_wide = wide;
    c->persistent = persistent;
    c->slim_outer = slim;
    c->split2 = split2;
    c->outer64 = outer64;
    c->plan_auto = plan == 0;
    build_plans(c);
    return FASTECC_OK;
}

int fastecc_set_plan(fastecc_ctx* c, int plan)
{
    if (!c) return FASTECC_E_INVAL;
    if (c->sharded) return sharded_forward(c, SH_SET_PLAN, nullptr, plan);
    CallLock lk(c->mu);
    if (c->p61) {
        // plan ids of this field: gf61_path.hpp
        DeviceGuard dg(c->device);
        if (!dg.ok) return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
        HIP_TRY(hipDeviceSynchronize());
        const int rc = p61::set_plan(c->p61, plan, g_detail, sizeof g_detail);
        c->plan_text = p61::plan_string(c->p61);
        return rc;
    }
    const int rc = apply_plan(c, plan);
    if (rc != FASTECC_OK) return rc;
    DeviceGuard dg(c->device);
    if (!dg.ok) return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
    HIP_TRY(hipDeviceSynchronize());  // kernels still reading the old tables
    return upload_twiddles(c);
}

// ---- host-only introspection: no device is touched, so the planning logic is testable anywhere ----
static int host_plan(fastecc_ctx* c, uint64_t k, uint64_t block_bytes, int plan)
{
    const int lg = ilog2_exact(k);
    if (k < 2 || lg < 0 || block_bytes == 0 || (block_bytes % 4) != 0) return FASTECC_E_INVAL;
    if (lg > 19) return FASTECC_E_UNSUPPORTED;
    c->N = k;
    c->n = lg;
    c->S = block_bytes / 4;
    c->ld = c->S;
    return apply_plan(c, plan);
}

int fastecc_plan_describe(uint64_t k, uint64_t block_bytes, int plan, char* buf, size_t cap)
{
    if (!buf || cap == 0) return FASTECC_E_INVAL;
    fastecc_ctx c;
    const int rc = host_plan(&c, k, block_bytes, plan);
    if (rc != FASTECC_OK) return rc;
    snprintf(buf, cap, "%s", c.plan_text.c_str());
    return FASTECC_OK;
}

int fastecc_plan_twiddles(uint64_t k, uint64_t block_bytes, int plan, int which, uint32_t* out, int32_t* level_stride)
{
    if (!out || which < 0 || which > 3) return FASTECC_E_INVAL;
    fastecc_ctx c;
    const int rc = host_plan(&c, k, block_bytes, plan);
    if (rc != FASTECC_OK) return rc;
    const uint32_t wN = gf::h_root((uint32_t)k), wNi = gf::h_inv(wN);
    const std::vector<int> sl = level_strides(which < 2 ? c.encode_plan : c.ntt_plan, c.n);
    const bool inverse_roots = (which == 0 || which == 3);
    const std::vector<uint32_t> tab = build_level_table(c.n, inverse_roots ? wNi : wN, sl);
    memcpy(out, tab.data(), (size_t)k * 4);
    if (level_stride)
        for (int l = 0; l < c.n; l++) level_stride[l] = sl[l];
    return FASTECC_OK;
}

}  // extern "C"
