// api.hip — the C ABI of libfastecc_hip.so (include/fastecc.h): contexts, the entry points that move data and the drivers behind them.
//
// Host side of the encode path.  It replaces the body of EncodeReedSolomon (RS.cpp:22-68) and the drivers MFA_NTT / Rec_NTT
// (ntt.cpp:349-447).  The pass plans and twiddle tables live in plan.hip, options and profiling in options.hip, the shared context in
// context.hpp.  No CPU compute fallback exists: every entry point that moves data runs HIP kernels or fails.
#include <condition_variable>
#include <thread>

#include "context.hpp"

using namespace fastecc;

namespace fastecc {


thread_local char g_detail[256] = "";

int hip_fail(hipError_t e, const char* what)
{
    snprintf(g_detail, sizeof g_detail, "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}


int ilog2_exact(uint64_t v)
{
    int l = 0;
    while ((1ull << l) < v) l++;
    return ((1ull << l) == v) ? l : -1;
}

uint32_t bitrev_host(uint32_t v, int bits)
{
    return bits <= 0 ? 0u : (__builtin_bitreverse32(v) >> (32 - bits));  // called once per table entry: no bit loop
}

}  // namespace fastecc

namespace {

// widest lane vector that the block size and the pointers allow
int pick_vec(const fastecc_ctx* c, const void* a, const void* b)
{
    int v = c->vec;
    const uintptr_t bits = (uintptr_t)a | (uintptr_t)b;
    while (v > 1 && ((c->S % v) != 0 || (c->ld % v) != 0 || (bits % (4u * v)) != 0)) v >>= 1;
    return v;
}

struct ProfScope {
    fastecc_ctx* c;
    hipStream_t st;
    ProfileRec* rec = nullptr;
    ProfScope(fastecc_ctx* c_, hipStream_t st_, const char* name, uint64_t bytes = 0) : c(c_), st(st_)
    {
        if (!c->profiling) return;
        if (c->prof_used == c->prof.size()) {
            ProfileRec r;
            if (hipEventCreate(&r.start) != hipSuccess) return;
            if (hipEventCreate(&r.stop) != hipSuccess) {
                (void)hipEventDestroy(r.start);
                return;
            }
            c->prof.push_back(r);
        }
        rec = &c->prof[c->prof_used++];
        rec->name = name;
        rec->bytes = bytes;
        (void)hipEventRecord(rec->start, st);
    }
    void finish()
    {
        if (rec) (void)hipEventRecord(rec->stop, st);
        rec = nullptr;
    }
    ~ProfScope() { finish(); }
};

// Runs the passes of `plan` on columns [col0, col0 + width) of every block (the whole block by default).
// first_done (optional) is recorded on `st` right after the first pass.
//
// The encode plan is [DIF passes][MID][DIT passes].  Normally the first pass reads `in`, writes `out`, and the rest
// runs in place on `out`.  Two variations share the DIF half on a k-block scratch stripe:
//   fold > 0   : MID keeps every 2^fold-th block (written compactly to `out`), the DIT passes above it are a size-M
//                transform in place on `out`;
//   cosets > 1 : [MID][DIT passes] run once per coset of evaluation points (its own per-block factor table), coset t
//                writing blocks [t*k, (t+1)*k) of `out`.
int run_passes(fastecc_ctx* c, const std::vector<Pass>& plan, const uint32_t* in, uint32_t* out, const uint32_t* tw_dif,
               const uint32_t* tw_dit, hipStream_t st, uint32_t col0 = 0, uint32_t width = 0, hipEvent_t first_done = nullptr,
               uint32_t batch = 1, const CallBounds& cb = CallBounds())
{
    if (!tw_dif || !tw_dit) return FASTECC_E_DEVICE;  // twiddle_table failed (detail recorded)
    if (&plan == &c->encode_plan && !c->dscale) return FASTECC_E_UNSUPPORTED;  // create_ntt_ctx: no per-block factors, no encode
    if (width == 0) width = (uint32_t)c->S;
    in += col0;
    out += col0;
    char name[32];
    const bool is_encode = &plan == &c->encode_plan;
    const bool folded = c->fold > 0 && is_encode;
    const int cosets = is_encode ? c->cosets : 1;
    const bool staged = folded || cosets > 1;
    if (staged && !c->scratch) HIP_TRY(hipMalloc((void**)&c->scratch, c->N * c->ld * 4));
    int vec = staged ? std::min(pick_vec(c, in, out), pick_vec(c, c->scratch, c->scratch)) : pick_vec(c, in, out);
    // the last pass may store to another buffer and the decoder's first pass reads a second one: they bound the lane vector too
    while (vec > 1 && (width % vec) != 0) vec >>= 1;
    if (cb.final_out) vec = std::min(vec, pick_vec(c, cb.final_out + col0, cb.final_out + col0));
    if (cb.gather_odd) vec = std::min(vec, pick_vec(c, cb.gather_odd + col0, cb.gather_odd + col0));

    auto run_one = [&](const Pass& p, const uint32_t* src, uint32_t* dst, const uint32_t* dscale, bool last = false) -> int {
        const uint32_t in_rows = src == in ? cb.in_rows : 0;
        uint32_t out_rows = 0;
        if (last && cb.final_out) {
            dst = cb.final_out + col0;
            out_rows = cb.out_rows;
        }
        const bool above_mid = folded && p.mode == MODE_DIT;
        const int n_eff = above_mid ? c->n - c->fold : c->n, s_eff = above_mid ? p.s - c->fold : p.s;
        const uint32_t* twd = above_mid ? twiddle_table(c, TW_FOLD_DIT, st) : tw_dit;
        if (!twd) return FASTECC_E_DEVICE;
        const uint64_t rows_moved = !folded ? 2 * c->N : p.mode == MODE_DIF ? 2 * c->N : p.mode == MODE_MID ? c->N + c->M : 2 * c->M;
        ProfScope ps(c, st, pass_name(p, vec, name, sizeof name), rows_moved * width * 4ull * batch);
        if (p.tile) {
            TileArgs a{};
            a.in = src;
            a.out = dst;
            a.tw_dif = tw_dif;
            a.tw_dit = twd;
            a.dscale = dscale;
            a.S = width;
            a.ld = (uint32_t)c->ld;
            a.n = n_eff;
            a.s = s_eff;
            a.fold = folded && p.mode == MODE_MID ? c->fold : 0;
            a.wide = p.wide;
            a.batch = batch;
            a.dscale_whole = cb.dscale_whole ? 1u : 0u;
            a.in_rows = in_rows;
            a.out_rows = out_rows;
            if (cb.gather_factor && src == in) {  // first pass of the decoder's transform
                a.in_odd = cb.gather_odd;
                a.row_factor = cb.gather_factor;
            }
            int mode = p.mode;
            if (cb.rows_factor && src == in && p.mode == MODE_DIF) {  // split decoder: blocks times their factors on the way in
                mode = MODE_DIF_ROWS;
                a.row_factor = cb.rows_factor;
                a.groups = cb.groups;
            }
            if (cb.impulse_table && src == in && p.mode == MODE_DIF && p.s == 0) {  // split decoder: the parity half's low levels, few groups in use
                mode = MODE_DIF_IMPULSE;
                a.row_factor = cb.impulse_table;
                a.impulse_rows = cb.impulse_rows;
            }
            if (cb.rows_out_factor && last && p.mode == MODE_DIT) {  // split decoder: its scatter
                mode = MODE_DIT_ROWS;
                a.row_factor = cb.rows_out_factor;
            }
            if (cb.addend && p.mode == MODE_MID) {
                mode = cb.mid_up ? MODE_MID_UP : MODE_MID_ADD;
                a.addend = cb.addend + col0;
                a.addend_factor = cb.addend_factor;
                a.addend_shift = cb.addend_shift;
                a.keep = cb.keep && !cb.mid_up ? cb.keep + col0 : nullptr;
            }
            a.persistent_cus = c->persistent ? c->cus : 0;
            a.split2 = c->split2;
            a.xcd_swizzle = c->xcd_swizzle;
            // Non-temporal streaming only pays when block rows are cache-line aligned: with e.g. 2052- or 4100-byte
            // blocks every 128-byte row segment straddles two lines that the neighbouring workgroup needs too,
            // and keeping them cacheable is worth 1.2-1.4x (profiles/r01/ablation_dif_tiles.md).
            const bool rows_aligned = ((c->ld * 4) % 128) == 0;
            a.cache_policy = !rows_aligned ? 0 : p.mode == MODE_MID ? (c->cache_policy >> 2) & 3 : c->cache_policy & 3;
            HIP_TRY(launch_tile(p.logr, p.pair, p.rlog, mode, a, st));
        } else {
            PassArgs a{};
            a.in = src;
            a.out = dst;
            a.tw_dif = tw_dif;
            a.tw_dit = twd;
            a.dscale = dscale;
            a.S = width;  // `in` / `out` already point at the first column of the range
            a.ld = (uint32_t)c->ld;
            a.n = n_eff;
            a.s = s_eff;
            a.fold = folded && p.mode == MODE_MID ? c->fold : 0;
            a.batch = batch;
            a.dscale_whole = cb.dscale_whole ? 1u : 0u;
            a.in_rows = in_rows;
            a.out_rows = out_rows;
            if (cb.gather_factor && src == in) {  // first pass of the decoder's transform
                a.in_odd = cb.gather_odd;
                a.row_factor = cb.gather_factor;
            }
            if ((cb.rows_factor && src == in) || (cb.addend && p.mode == MODE_MID) || (cb.rows_out_factor && last) || cb.impulse_table) return FASTECC_E_UNSUPPORTED;  // tile passes only
            HIP_TRY(launch_pass(p.logr, vec, p.mode, a, st));
        }
        return FASTECC_OK;
    };

    const uint32_t* src = in;
    if (!staged) {
        bool first = true;
        for (const Pass& p : plan) {
            if (p.fused) continue;  // encode_mixed launches it
            const int rc = run_one(p, src, out, cb.dscale_override ? cb.dscale_override : c->dscale, &p == &plan.back());
            if (rc != FASTECC_OK) return rc;
            src = out;  // after the first pass everything is in place on `out`
            if (first && first_done) HIP_TRY(hipEventRecord(first_done, st));
            first = false;
        }
        return FASTECC_OK;
    }
    size_t i = 0;
    for (; i < plan.size() && plan[i].mode == MODE_DIF; ++i) {
        const int rc = run_one(plan[i], src, c->scratch, c->dscale);
        if (rc != FASTECC_OK) return rc;
        src = c->scratch;
    }
    for (int t = 0; t < cosets; ++t) {
        uint32_t* o = out + (size_t)t * c->N * c->ld;
        const uint32_t* s2 = src;
        for (size_t j = i; j < plan.size(); ++j) {
            const int rc = run_one(plan[j], s2, o, c->dscale + (size_t)t * c->N, j + 1 == plan.size());
            if (rc != FASTECC_OK) return rc;
            s2 = o;
        }
    }
    return FASTECC_OK;
}

int ensure_slab_streams(fastecc_ctx* c)
{
    if (c->slab_ready) return FASTECC_OK;
    HIP_TRY(hipEventCreateWithFlags(&c->slab_fork, hipEventDisableTiming));
    for (int h = 0; h < fastecc_ctx::MAX_SLABS; h++) {
        HIP_TRY(hipStreamCreateWithFlags(&c->slab_stream[h], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->slab_first_done[h], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->slab_done[h], hipEventDisableTiming));
    }
    c->slab_ready = true;
    return FASTECC_OK;
}

bool plan_is_all_tiles(const std::vector<Pass>& plan)
{
    for (const Pass& p : plan)
        if (!p.tile) return false;
    return !plan.empty();
}

// fastecc_profile_* for the launches of gf61_kernels.hip: one ProfScope per launch (launches of a context are serial)
struct P61Hooks {
    fastecc_ctx* c;
    ProfScope* open = nullptr;
    p61::LaunchHooks h;
    explicit P61Hooks(fastecc_ctx* c_) : c(c_)
    {
        h.user = this;
        h.begin = [](void* u, hipStream_t st, const char* name, uint64_t bytes) {
            P61Hooks* self = (P61Hooks*)u;
            self->open = new (std::nothrow) ProfScope(self->c, st, name, bytes);
        };
        h.end = [](void* u, hipStream_t) {
            P61Hooks* self = (P61Hooks*)u;
            delete self->open;
            self->open = nullptr;
        };
    }
    ~P61Hooks() { delete open; }
};

int encode_pow2(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st, const CallBounds& cb = CallBounds());

// Transform order q * N: [radix-q pass down][the power-of-two pipeline on q stripes of N blocks][radix-q pass up].
// The first pass reads the K existing data blocks (the rest is zero), the last one writes the first Mu parity blocks.
int encode_mixed(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st)
{
    const uint64_t N1 = (uint64_t)c->q * c->N;
    uint32_t* work = parity;
    if (c->Mu != N1) {
        if (!c->mixbuf) HIP_TRY(hipMalloc((void**)&c->mixbuf, N1 * c->ld * 4));
        work = c->mixbuf;
    }
    // these passes are free to go as wide as the block size and the three pointers allow (the plan's `vec` is about its own passes)
    int vec = 4;
    const uintptr_t bits = (uintptr_t)data | (uintptr_t)work | (uintptr_t)parity;
    while (vec > 1 && ((c->S % vec) != 0 || (c->ld % vec) != 0 || (bits % (4u * vec)) != 0)) vec >>= 1;
    char name[32];
    if (!c->encode_plan.empty() && c->encode_plan.front().fused) {
        // [odd radix + outer DIF tile][MID on q stripes][outer DIT tile + odd radix]: three trips through HBM
        const Pass& pd = c->encode_plan.front();
        const Pass& pu = c->encode_plan.back();
        FusedArgs f{};
        f.S = (uint32_t)c->S;
        f.ld = (uint32_t)c->ld;
        f.M = (uint32_t)c->N;
        {
            f.in = data;
            f.out = work;
            f.dft = c->q_dft_inv;
            f.tw = c->q_tw_dif;
            f.twl = twiddle_table(c, TW_ENC_DIF, st);
            if (!f.twl) return FASTECC_E_DEVICE;
            f.s = pd.s;
            f.in_rows = c->K != N1 ? (uint32_t)c->K : 0;
            snprintf(name, sizeof name, "fused%d_dif%d", c->q, pd.logr);
            ProfScope ps(c, st, name, (c->K + N1) * c->S * 4ull);
            HIP_TRY(launch_fused(c->q, pd.logr, false, f, st));
        }
        CallBounds cbm;
        cbm.dscale_whole = true;
        const int rcm = run_passes(c, c->encode_plan, work, work, twiddle_table(c, TW_ENC_DIF, st), twiddle_table(c, TW_ENC_DIT, st), st, 0, 0, nullptr, (uint32_t)c->q, cbm);
        if (rcm != FASTECC_OK) return rcm;
        {
            f.in = work;
            f.out = parity;
            f.dft = c->q_dft_fwd;
            f.tw = c->q_tw_dit;
            f.twl = twiddle_table(c, TW_ENC_DIT, st);
            if (!f.twl) return FASTECC_E_DEVICE;
            f.s = pu.s;
            f.in_rows = 0;
            f.out_rows = c->Mu != N1 ? (uint32_t)c->Mu : 0;
            snprintf(name, sizeof name, "fused%d_dit%d", c->q, pu.logr);
            ProfScope ps(c, st, name, (N1 + c->Mu) * c->S * 4ull);
            HIP_TRY(launch_fused(c->q, pu.logr, true, f, st));
        }
        return FASTECC_OK;
    }
    RadixArgs a{};
    a.S = (uint32_t)c->S;
    a.ld = (uint32_t)c->ld;
    a.M = (uint32_t)c->N;
    {
        a.in = data;
        a.out = work;
        a.dft = c->q_dft_inv;
        a.tw = c->q_tw_dif;
        a.in_rows = c->K != N1 ? (uint32_t)c->K : 0;
        a.out_rows = 0;
        snprintf(name, sizeof name, "radix%d_dif", c->q);
        ProfScope ps(c, st, name, (c->K + N1) * c->S * 4ull);
        HIP_TRY(launch_radix(c->q, false, vec, a, st));
    }
    CallBounds cb;
    cb.dscale_whole = true;
    const int rc = run_passes(c, c->encode_plan, work, work, twiddle_table(c, TW_ENC_DIF, st), twiddle_table(c, TW_ENC_DIT, st), st, 0, 0, nullptr, (uint32_t)c->q, cb);
    if (rc != FASTECC_OK) return rc;
    {
        a.in = work;
        a.out = parity;
        a.dft = c->q_dft_fwd;
        a.tw = c->q_tw_dit;
        a.in_rows = 0;
        a.out_rows = c->Mu != N1 ? (uint32_t)c->Mu : 0;
        snprintf(name, sizeof name, "radix%d_dit", c->q);
        ProfScope ps(c, st, name, (N1 + c->Mu) * c->S * 4ull);
        HIP_TRY(launch_radix(c->q, true, vec, a, st));
    }
    return FASTECC_OK;
}

// codes with few parity blocks skip the transform pipeline: one read of the data (direct.hip: direct_encode_run).  The pipeline costs the
// same for any n - k <= N/16; the direct pass grows with n - k: on the matrix cores it wins up to ~128 parity blocks, on the VALU up to 16.
static bool direct_encode_applies(const fastecc_ctx* c, const void* data = nullptr, const void* parity = nullptr)
{
    if (c->p61 || c->cosets != 1 || c->ld != c->S || c->Mu < 1) return false;
    // measured at k = 2^19 x 4 KB (profiles/r03/direct_bench.jsonl): pipeline 2.4 ms; MFMA kernel 0.40 (n - k <= 16) ... 1.4 (128) ... 2.7 ms (256);
    // VALU kernel 0.9 ms per sweep of 16 outputs
    int limit = std::min(c->encode_direct_max, direct_encode_max());
    if (c->direct_kernel == 0 && !direct_mfma_applies(data, parity, c->S)) limit = std::min(limit, 32);
    return (int)std::min<uint64_t>(c->Mu, 100000) <= limit;
}

int encode_device(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st)
{
    if (direct_encode_applies(c, data, parity)) {
        // out of memory for the weight tables or the partial sums is not an error: the transform pipeline below needs neither
        int rc = FASTECC_OK;
        if (!c->direct_enc) rc = direct_encode_build(&c->direct_enc, (uint64_t)c->q * c->N, c->K, c->Mu, c->fold, c->S);  // q > 1: the mixed-radix order
        if (rc == FASTECC_OK) {
            ProfScope ps(c, st, "direct_encode", (c->K + c->Mu) * c->S * 4ull);
            rc = direct_encode_run(c->direct_enc, data, parity, c->direct_kernel, st);
        }
        if (rc != FASTECC_E_NOMEM) return rc;
        (void)hipGetLastError();
    }
    if (c->q > 1) return encode_mixed(c, data, parity, st);
    if (c->K == c->N && c->Mu == c->M) return encode_pow2(c, data, parity, st);
    if (c->p61) {
        // 64-bit field, any (n,k): the K data blocks extended with zero blocks to N (a copy), the (2N,N) encode, and parity block j picked
        // from block j * stride of its result (a strided copy).  K == N needs no data copy.
        const size_t row = (size_t)c->S * 4;
        const uint32_t* src = data;
        if (c->K != c->N) {
            if (!c->scratch) HIP_TRY(hipMalloc((void**)&c->scratch, c->N * row));
            HIP_TRY(hipMemcpyAsync(c->scratch, data, c->K * row, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemsetAsync((char*)c->scratch + c->K * row, 0, (c->N - c->K) * row, st));
            src = c->scratch;
        }
        if (!c->parbuf) HIP_TRY(hipMalloc((void**)&c->parbuf, c->N * row));
        const int rc = encode_pow2(c, src, c->parbuf, st);
        if (rc != FASTECC_OK) return rc;
        HIP_TRY(hipMemcpy2DAsync(parity, row, c->parbuf, (size_t)c->p61_stride * row, row, c->Mu, hipMemcpyDeviceToDevice, st));
        return FASTECC_OK;
    }
    // any (n,k): the first pass reads the K existing data blocks and takes the rest of the stripe as zero, the last pass
    // writes only the first Mu of the M parity blocks it computes — both through the kernels' bounds handling, no copies.
    // The passes in between need all M blocks somewhere: the caller's parity buffer when it is that large, else parbuf.
    const size_t row = (size_t)c->ld * 4;
    uint32_t* out = parity;
    CallBounds cb;
    if (c->Mu != c->M) {
        if (!c->parbuf) HIP_TRY(hipMalloc((void**)&c->parbuf, c->M * row));
        out = c->parbuf;
        cb.final_out = parity;
        cb.out_rows = (uint32_t)c->Mu;
    }
    cb.in_rows = c->K != c->N ? (uint32_t)c->K : 0;
    return encode_pow2(c, data, out, st, cb);
}

int encode_pow2(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st, const CallBounds& cb)
{
    if (c->p61) {
        P61Hooks hk(c);
        if (c->cosets > 1) {  // n = 4k / 8k: the DIF half once into a k-block work stripe, MID and the DIT half once per coset
            if (p61::encode_cosets_needs_work(c->p61) && !c->scratch) HIP_TRY(hipMalloc((void**)&c->scratch, c->N * (size_t)c->S * 4));
            return p61::encode_cosets(c->p61, (const uint64_t*)data, (uint64_t*)parity, (uint64_t*)c->scratch, st, c->profiling ? &hk.h : nullptr);
        }
        return p61::encode(c->p61, (const uint64_t*)data, (uint64_t*)parity, st, c->profiling ? &hk.h : nullptr);
    }
    // inverse roots on the way down (interpolate), forward roots on the way up (evaluate) — RS.cpp:41,63
    const int H = c->slabs;
    const bool slabbed = c->fold == 0 && c->cosets == 1 && H > 1 && H <= fastecc_ctx::MAX_SLABS && plan_is_all_tiles(c->encode_plan) && c->encode_plan.size() >= 2 &&
                         (c->S % (32u * H)) == 0;
    if (!slabbed) return run_passes(c, c->encode_plan, data, parity, twiddle_table(c, TW_ENC_DIF, st), twiddle_table(c, TW_ENC_DIT, st), st, 0, 0, nullptr, 1, cb);

    // Column slabs are independent transforms.  Slab h runs on its own stream and starts when slab h-1 has
    // finished its first pass, so that at any time the GPU holds one slab in each kind of pass: the
    // VALU-bound MID tiles and the HBM-bound outer tiles then share the CUs (both are 64 KiB / 16 waves).
    const uint32_t width = (uint32_t)(c->S / H);
    if (c->slab_mode == 1) {
        // one slab after the other on the caller's stream: a slab's three passes follow each other closely enough for the
        // second and third to find it in the memory-side cache (256 MB) when the slab is small enough
        for (int h = 0; h < H; h++) {
            const int rc1 = run_passes(c, c->encode_plan, data, parity, twiddle_table(c, TW_ENC_DIF, st), twiddle_table(c, TW_ENC_DIT, st), st, h * width, width, nullptr, 1, cb);
            if (rc1 != FASTECC_OK) return rc1;
        }
        return FASTECC_OK;
    }
    int rc = ensure_slab_streams(c);
    if (rc != FASTECC_OK) return rc;
    HIP_TRY(hipEventRecord(c->slab_fork, st));
    for (int h = 0; h < H; h++) {
        hipStream_t sh = c->slab_stream[h];
        HIP_TRY(hipStreamWaitEvent(sh, c->slab_fork, 0));
        if (h > 0) HIP_TRY(hipStreamWaitEvent(sh, c->slab_first_done[h - 1], 0));
        rc = run_passes(c, c->encode_plan, data, parity, twiddle_table(c, TW_ENC_DIF, sh), twiddle_table(c, TW_ENC_DIT, sh), sh, h * width, width, c->slab_first_done[h], 1, cb);
        if (rc != FASTECC_OK) return rc;
        HIP_TRY(hipEventRecord(c->slab_done[h], sh));
        HIP_TRY(hipStreamWaitEvent(st, c->slab_done[h], 0));
    }
    return FASTECC_OK;
}

int ensure_dbuf(fastecc_ctx* c);
int stage_download(fastecc_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st);

// FASTECC_MEM_HOST_PINNED: the stripe lives in pinned host memory.  Column slabs are independent transforms, so the call is a
// three-stage pipeline over the slabs — upload (a strided 2-D copy on a copy engine: full link rate from 512-byte rows up), encode in
// place in the device staging stripe, download — on THREE streams, one per stage: every direction of the link then has exactly one
// transfer in flight, in slab order, and the two directions and the kernels overlap.  (One stream per SLAB, as in rounds 1-3, let the
// runtime map eight streams onto its few hardware queues: the rocprofv3 copy trace showed upload 3 waiting behind downloads 1 and 2.)
int encode_host_pinned(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st)
{
    int rc = ensure_dbuf(c);
    if (rc != FASTECC_OK) return rc;
    int H = c->host_slabs;
    while (H > 1 && (c->S % (32u * H)) != 0) H >>= 1;
    if (!(plan_is_all_tiles(c->encode_plan) && c->encode_plan.size() >= 2)) H = 1;  // register passes work on whole blocks
    const size_t pitch = (size_t)c->S * 4;
    if (H == 1) {
        HIP_TRY(hipMemcpyAsync(c->dbuf, data, c->stripe_bytes, hipMemcpyHostToDevice, st));
        rc = run_passes(c, c->encode_plan, c->dbuf, c->dbuf, twiddle_table(c, TW_ENC_DIF, st), twiddle_table(c, TW_ENC_DIT, st), st);
        if (rc != FASTECC_OK) return rc;
        HIP_TRY(hipMemcpyAsync(parity, c->dbuf, c->stripe_bytes, hipMemcpyDeviceToHost, st));
        return FASTECC_OK;
    }
    rc = ensure_slab_streams(c);
    if (rc != FASTECC_OK) return rc;
    const uint32_t width = (uint32_t)(c->S / H);
    hipStream_t s_up = c->slab_stream[0], s_cp = c->slab_stream[1], s_dn = c->slab_stream[2];
    HIP_TRY(hipEventRecord(c->slab_fork, st));
    for (hipStream_t q : {s_up, s_cp, s_dn}) HIP_TRY(hipStreamWaitEvent(q, c->slab_fork, 0));
    for (int h = 0; h < H; h++) {
        HIP_TRY(hipMemcpy2DAsync(c->dbuf + (size_t)h * width, pitch, data + (size_t)h * width, pitch, (size_t)width * 4, c->N, hipMemcpyHostToDevice, s_up));
        HIP_TRY(hipEventRecord(c->slab_first_done[h], s_up));
        HIP_TRY(hipStreamWaitEvent(s_cp, c->slab_first_done[h], 0));
        rc = run_passes(c, c->encode_plan, c->dbuf, c->dbuf, twiddle_table(c, TW_ENC_DIF, s_cp), twiddle_table(c, TW_ENC_DIT, s_cp), s_cp, h * width, width);
        if (rc != FASTECC_OK) return rc;
        HIP_TRY(hipEventRecord(c->slab_done[h], s_cp));
        HIP_TRY(hipStreamWaitEvent(s_dn, c->slab_done[h], 0));
        HIP_TRY(hipMemcpy2DAsync(parity + (size_t)h * width, pitch, c->dbuf + (size_t)h * width, pitch, (size_t)width * 4, c->N, hipMemcpyDeviceToHost, s_dn));
    }
    // the call behaves as one operation on `st`: it ends with the last download (which follows everything else)
    HIP_TRY(hipEventRecord(c->slab_fork, s_dn));
    HIP_TRY(hipStreamWaitEvent(st, c->slab_fork, 0));
    return FASTECC_OK;
}

int ntt_device(fastecc_ctx* c, uint32_t* data, bool inverse, hipStream_t st)
{
    if (c->p61) {
        P61Hooks hk(c);
        return p61::ntt(c->p61, (uint64_t*)data, inverse, st, c->profiling ? &hk.h : nullptr);
    }
    const uint32_t* tw = inverse ? twiddle_table(c, TW_NTT_INV, st) : twiddle_table(c, TW_NTT_FWD, st);
    int rc = run_passes(c, c->ntt_plan, data, data, tw, tw, st);
    if (rc != FASTECC_OK) return rc;
    if (c->n >= 2) {
        ProfScope ps(c, st, "bitrev_rows");
        HIP_TRY(launch_bitrev_rows(data, (uint32_t)c->S, c->n, pick_vec(c, data, data), st));
    }
    return FASTECC_OK;
}

// PAGEABLE host memory (what RS.cpp's malloc'ed buffers are) <-> device.  The runtime's own pageable download stages through pinned memory
// with one copying host thread: 2 GiB took 89 ms (24 GB/s) on a link that moves them in 37 ms.  Here a ring of pinned slots sits between the
// two: the copy engine fills (or empties) a slot with one hipMemcpy2DAsync on `st`, an event per slot, and a few helper threads move the
// slot's rows from / to the caller's buffer side by side; a slot is reused once all of them (download) or the copy engine (upload) are done
// with it.  The transfer is a `rows x width` rectangle on both sides (pitches may differ: a column slab of a stripe), packed in the slots.
// Synchronous on the host: returns when the caller's memory is complete (download) or every copy is on the stream (upload).  Any failure to
// set this up — no pinned memory, no threads — falls back to the plain copy.
struct StageJob {
    bool to_device;
    char* host;           // pageable
    size_t host_pitch;
    char* dev;
    size_t dev_pitch;
    size_t width, rows;   // bytes per row, rows
    void* const* host_rows = nullptr;  // optional: row r lives at host_rows[r] (host, host_pitch unused): the reference's T** block table
};

int stage_plain(const StageJob& j, hipStream_t st)
{
    if (j.host_rows) {
        if (j.width * j.rows <= ((size_t)64 << 20)) {  // small stripes: packed in a host buffer, one copy (a copy per tiny block would cost ~10 us each)
            std::vector<char> packed;
            try {
                packed.resize(j.width * j.rows);
            } catch (const std::bad_alloc&) {
                return FASTECC_E_NOMEM;
            }
            if (j.to_device) {
                for (size_t r = 0; r < j.rows; r++) memcpy(packed.data() + r * j.width, j.host_rows[r], j.width);
                HIP_TRY(hipMemcpy2DAsync(j.dev, j.dev_pitch, packed.data(), j.width, j.width, j.rows, hipMemcpyHostToDevice, st));
                HIP_TRY(hipStreamSynchronize(st));
            } else {
                HIP_TRY(hipMemcpy2DAsync(packed.data(), j.width, j.dev, j.dev_pitch, j.width, j.rows, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                for (size_t r = 0; r < j.rows; r++) memcpy(j.host_rows[r], packed.data() + r * j.width, j.width);
            }
            return FASTECC_OK;
        }
        for (size_t r = 0; r < j.rows; r++) {  // blocks larger than a slot: a copy each
            if (j.to_device) HIP_TRY(hipMemcpyAsync(j.dev + r * j.dev_pitch, j.host_rows[r], j.width, hipMemcpyHostToDevice, st));
            else HIP_TRY(hipMemcpyAsync(j.host_rows[r], j.dev + r * j.dev_pitch, j.width, hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        return FASTECC_OK;
    }
    if (j.to_device) HIP_TRY(hipMemcpy2DAsync(j.dev, j.dev_pitch, j.host, j.host_pitch, j.width, j.rows, hipMemcpyHostToDevice, st));
    else HIP_TRY(hipMemcpy2DAsync(j.host, j.host_pitch, j.dev, j.dev_pitch, j.width, j.rows, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return FASTECC_OK;
}

bool ensure_stage_ring(fastecc_ctx::StageRing& r)
{
    constexpr int NSLOT = fastecc_ctx::STAGE_SLOTS;
    if (r.slots) return true;
    if (hipHostMalloc((void**)&r.slots, NSLOT * fastecc_ctx::STAGE_SLOT_BYTES, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        r.slots = nullptr;
        return false;
    }
    for (int i = 0; i < NSLOT; i++)
        if (hipEventCreateWithFlags(&r.event[i], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            for (int k = 0; k < i; k++) (void)hipEventDestroy(r.event[k]), r.event[k] = nullptr;
            (void)hipHostFree(r.slots);
            r.slots = nullptr;
            return false;
        }
    return true;
}

int stage_transfer(fastecc_ctx* c, const StageJob& j, hipStream_t st, int threads = 0)
{
    constexpr int NSLOT = fastecc_ctx::STAGE_SLOTS;
    constexpr size_t SLOT = fastecc_ctx::STAGE_SLOT_BYTES;
    if (j.width == 0 || j.rows == 0) return FASTECC_OK;
    fastecc_ctx::StageRing& ring = j.to_device ? c->stage_up : c->stage_down;
    if (j.width * j.rows < 2 * SLOT || j.width > SLOT || !ensure_stage_ring(ring)) return stage_plain(j, st);
    const size_t chunk_rows = SLOT / j.width, chunks = (j.rows + chunk_rows - 1) / chunk_rows;
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = threads > 0 ? threads : c->stage_threads > 0 ? c->stage_threads : (int)std::min<unsigned>(6u, std::max<unsigned>(2u, hw / 4u));
    std::mutex mu;
    std::condition_variable cv;
    long issued = -1;                   // chunks [0, issued] have their copy and event on the stream
    std::vector<int> done;              // helper threads finished with chunk i (download: emptied the slot; upload: filled it)
    try {
        done.assign(chunks, 0);
    } catch (const std::bad_alloc&) {
        return stage_plain(j, st);
    }
    bool failed = false;
    const int device = c->device;
    auto rows_of = [&](size_t i) { return std::min(chunk_rows, j.rows - i * chunk_rows); };
    auto move_rows = [&](size_t i, int t) {  // thread t's share of chunk i between the slot (packed rows) and the caller's buffer
        const size_t n = rows_of(i), r0 = i * chunk_rows;
        char* slot = ring.slots + (i % NSLOT) * SLOT;
        if (j.host_rows) {
            const size_t per = (n + T - 1) / T, lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
            for (size_t r = lo; r < hi; r++) {
                if (r + 4 < hi) __builtin_prefetch(j.host_rows[r0 + r + 4]);
                if (j.to_device) memcpy(slot + r * j.width, j.host_rows[r0 + r], j.width);
                else memcpy(j.host_rows[r0 + r], slot + r * j.width, j.width);
            }
            return;
        }
        if (j.width == j.host_pitch) {  // contiguous on the host: one piece per thread
            const size_t bytes = n * j.width, piece = ((bytes / T + 63) / 64) * 64;
            const size_t lo = std::min(bytes, (size_t)t * piece), hi = std::min(bytes, lo + piece);
            if (hi > lo) {
                if (j.to_device) memcpy(slot + lo, j.host + r0 * j.host_pitch + lo, hi - lo);
                else memcpy(j.host + r0 * j.host_pitch + lo, slot + lo, hi - lo);
            }
            return;
        }
        const size_t per = (n + T - 1) / T, lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
        if (hi <= lo) return;
        if (j.to_device) host_copy_rows(slot + lo * j.width, j.width, j.host + (r0 + lo) * j.host_pitch, j.host_pitch, j.width, hi - lo);
        else host_copy_rows(j.host + (r0 + lo) * j.host_pitch, j.host_pitch, slot + lo * j.width, j.width, j.width, hi - lo);
    };
    auto worker = [&](int t) {
        (void)hipSetDevice(device);
        for (size_t i = 0; i < chunks; i++) {
            bool ok = true;
            if (j.to_device) {
                // the slot's previous content has left for the device: chunk i - NSLOT of this call, or the tail of the previous call on this ring
                // (an upload returns with its copies on the stream, not completed)
                if (i >= (size_t)NSLOT) {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return issued >= (long)(i - NSLOT) || failed; });
                    if (failed) return;
                }
                ok = hipEventSynchronize(ring.event[i % NSLOT]) == hipSuccess;
                if (ok) move_rows(i, t);
            } else {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return issued >= (long)i || failed; });
                    if (failed) return;
                }
                ok = hipEventSynchronize(ring.event[i % NSLOT]) == hipSuccess;
                if (ok) move_rows(i, t);
            }
            std::lock_guard<std::mutex> lk(mu);
            if (!ok) failed = true;
            done[i]++;
            cv.notify_all();
            if (!ok) return;
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int t = 0; t < T; t++) pool.emplace_back(worker, t);
    } catch (...) {
        {
            std::lock_guard<std::mutex> lk(mu);
            failed = true;
        }
        cv.notify_all();
        for (std::thread& th : pool) th.join();
        return stage_plain(j, st);  // (helpers that had started have touched nothing the plain copy does not rewrite)
    }
    hipError_t err = hipSuccess;
    for (size_t i = 0; i < chunks && err == hipSuccess; i++) {
        {
            // download: the slot's previous content has been copied out by every helper; upload: every helper has filled its share
            std::unique_lock<std::mutex> lk(mu);
            if (j.to_device) cv.wait(lk, [&] { return done[i] == T || failed; });
            else if (i >= (size_t)NSLOT) cv.wait(lk, [&] { return done[i - NSLOT] == T || failed; });
            if (failed) break;
        }
        const size_t n = rows_of(i), r0 = i * chunk_rows;
        char* slot = ring.slots + (i % NSLOT) * SLOT;
        if (j.to_device) err = hipMemcpy2DAsync(j.dev + r0 * j.dev_pitch, j.dev_pitch, slot, j.width, j.width, n, hipMemcpyHostToDevice, st);
        else err = hipMemcpy2DAsync(slot, j.width, j.dev + r0 * j.dev_pitch, j.dev_pitch, j.width, n, hipMemcpyDeviceToHost, st);
        if (err == hipSuccess) err = hipEventRecord(ring.event[i % NSLOT], st);
        std::lock_guard<std::mutex> lk(mu);
        if (err != hipSuccess) failed = true;
        else issued = (long)i;
        cv.notify_all();
    }
    for (std::thread& th : pool) th.join();
    if (err != hipSuccess) return hip_fail(err, "stage_transfer");
    if (failed) return hip_fail(hipErrorUnknown, "stage_transfer (helper thread)");
    return FASTECC_OK;
}

int stage_download(fastecc_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return FASTECC_OK;
    // one "row" per slot-sized piece keeps the 2-D copies wide
    const size_t width = std::min<size_t>(bytes, (size_t)1 << 20);
    const size_t rows = bytes / width, rest = bytes - rows * width;
    int rc = stage_transfer(c, StageJob{false, (char*)dst, width, (char*)const_cast<void*>(src), width, width, rows}, st);
    if (rc == FASTECC_OK && rest) {
        HIP_TRY(hipMemcpyAsync((char*)dst + rows * width, (const char*)src + rows * width, rest, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return rc;
}

// FASTECC_MEM_HOST on a code the column-slab pipeline takes (as encode_host_pinned: n = 2k = 2^m, all-tile plan), large stripes: slab h
// goes up through the staging ring while slab h - 1 comes down through the other one — its own helper threads, driven by one more thread
// — and the kernels of a slab run in between on a third stream.  Both directions of the link and 2 x T host cores are busy at once:
// 2 + 2 GiB in 65-100 ms where upload, encode and download one after the other take a steady 81-86 (a 16-CPU quota of an EPYC 9575F
// shared with other jobs: the twelve copying threads move 8 GiB through the cores in that time, which is what bounds it, not the link) —
// hence an option ("host_pipeline"), off by default.  Returns FASTECC_E_UNSUPPORTED for what it does
// not take (the caller then runs the plain sequence).
int encode_host_pageable(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st)
{
    int H = c->host_slabs;
    while (H > 1 && (c->S % (32u * H)) != 0) H >>= 1;
    if (H <= 1 || c->stripe_bytes < ((size_t)256 << 20) || !(plan_is_all_tiles(c->encode_plan) && c->encode_plan.size() >= 2)) return FASTECC_E_UNSUPPORTED;
    const uint32_t width = (uint32_t)(c->S / H);
    const size_t pitch = (size_t)c->S * 4, wbytes = (size_t)width * 4;
    if (wbytes * c->N < 2 * fastecc_ctx::STAGE_SLOT_BYTES) return FASTECC_E_UNSUPPORTED;
    int rc = ensure_dbuf(c);
    if (rc == FASTECC_OK) rc = ensure_slab_streams(c);
    if (rc != FASTECC_OK) return rc;
    if (!ensure_stage_ring(c->stage_up) || !ensure_stage_ring(c->stage_down)) return FASTECC_E_UNSUPPORTED;
    hipStream_t s_up = c->slab_stream[0], s_cp = c->slab_stream[1], s_dn = c->slab_stream[2];
    HIP_TRY(hipEventRecord(c->slab_fork, st));
    for (hipStream_t q : {s_up, s_cp, s_dn}) HIP_TRY(hipStreamWaitEvent(q, c->slab_fork, 0));
    const int T = 0;  // stage_transfer's own rule: the "stage_threads" option when it is set, else min(6, hardware threads / 4)
    std::thread down;
    int down_rc = FASTECC_OK;
    char down_text[256] = "";
    auto join_down = [&]() -> int {
        if (down.joinable()) down.join();
        if (down_rc != FASTECC_OK) set_error_text(down_text);
        return down_rc;
    };
    for (int h = 0; h < H && rc == FASTECC_OK; h++) {
        rc = stage_transfer(c, StageJob{true, (char*)const_cast<uint32_t*>(data + (size_t)h * width), pitch, (char*)(c->dbuf + (size_t)h * width), pitch, wbytes, c->N}, s_up, T);
        if (rc != FASTECC_OK) break;
        hipError_t e = hipEventRecord(c->slab_first_done[h], s_up);
        if (e == hipSuccess) e = hipStreamWaitEvent(s_cp, c->slab_first_done[h], 0);
        if (e != hipSuccess) { rc = hip_fail(e, "encode_host_pageable"); break; }
        rc = run_passes(c, c->encode_plan, c->dbuf, c->dbuf, twiddle_table(c, TW_ENC_DIF, s_cp), twiddle_table(c, TW_ENC_DIT, s_cp), s_cp, h * width, width);
        if (rc != FASTECC_OK) break;
        e = hipEventRecord(c->slab_done[h], s_cp);
        if (e != hipSuccess) { rc = hip_fail(e, "encode_host_pageable"); break; }
        rc = join_down();  // slab h - 1 is home; its ring is free for slab h
        if (rc != FASTECC_OK) break;
        try {
            down = std::thread([c, h, width, pitch, wbytes, parity, s_dn, &down_rc, &down_text] {
                (void)hipSetDevice(c->device);
                hipError_t w = hipStreamWaitEvent(s_dn, c->slab_done[h], 0);
                down_rc = w != hipSuccess ? hip_fail(w, "encode_host_pageable")
                                          : stage_transfer(c, StageJob{false, (char*)(parity + (size_t)h * width), pitch, (char*)(c->dbuf + (size_t)h * width), pitch, wbytes, c->N}, s_dn, T);
                if (down_rc != FASTECC_OK) snprintf(down_text, sizeof down_text, "%s", fastecc_last_error_detail());
            });
        } catch (...) {  // no thread: this slab comes down on the calling thread (a failure here still falls through to the settling code below)
            const hipError_t w = hipStreamWaitEvent(s_dn, c->slab_done[h], 0);
            rc = w != hipSuccess ? hip_fail(w, "encode_host_pageable")
               : stage_transfer(c, StageJob{false, (char*)(parity + (size_t)h * width), pitch, (char*)(c->dbuf + (size_t)h * width), pitch, wbytes, c->N}, s_dn, T);
        }
    }
    const int rd = join_down();
    if (rc == FASTECC_OK) rc = rd;
    // settle the three streams on every path: the context's buffers are free when the call returns
    for (hipStream_t q : {s_up, s_cp, s_dn}) (void)hipStreamSynchronize(q);
    return rc;
}

int ensure_dbuf(fastecc_ctx* c)
{
    if (c->dbuf) return FASTECC_OK;
    HIP_TRY(hipMalloc((void**)&c->dbuf, c->stripe_bytes));
    return FASTECC_OK;
}

// Ordering of the context's internal device buffers (scratch, parbuf, dbuf, factor, ...) between streams: work that
// touches them waits for the previous such work when that ran on another stream.  The caller holds c->mu.
int order_internal_buffers(fastecc_ctx* c, hipStream_t st)
{
    if (c->buf_used && c->buf_stream != st) HIP_TRY(hipStreamWaitEvent(st, c->buf_event, 0));
    return FASTECC_OK;
}
int mark_internal_buffers(fastecc_ctx* c, hipStream_t st)
{
    if (!c->buf_event) HIP_TRY(hipEventCreateWithFlags(&c->buf_event, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->buf_event, st));
    c->buf_stream = st;
    c->buf_used = true;
    return FASTECC_OK;
}
// Runs `body` (which enqueues work on `st` that uses internal buffers) between the two.
template <class F> int with_internal_buffers(fastecc_ctx* c, hipStream_t st, F body)
{
    int rc = order_internal_buffers(c, st);
    if (rc != FASTECC_OK) return rc;
    rc = body();
    const int rc2 = mark_internal_buffers(c, st);  // also after a failure: part of the work may have been enqueued
    return rc != FASTECC_OK ? rc : rc2;
}
using CallLock = std::lock_guard<std::mutex>;

}  // namespace

static int create_impl(fastecc_ctx** out, uint64_t n, uint64_t k, int lg, uint64_t block_bytes, int field, int device, int fold,
                       int cosets, const uint32_t* custom_factor);
static const uint32_t* const NTT_ONLY = reinterpret_cast<const uint32_t*>(~(uintptr_t)0);  // custom_factor value: see create_ntt_ctx

// Turns a fresh (2N, N) context into the power-of-two core of a transform of order q * N: the per-block factors for all
// q stripes (position j1*N + r holds coefficient q*bitrev(r) + j1 -> w_(2qN)^coefficient / (qN), RS.cpp:51-54 with qN for
// N) and the tables of the two odd-radix passes (mixed_kernels.hip).
static int setup_mixed(fastecc_ctx* c, int q, uint64_t k_user, uint64_t m_user, const uint32_t* custom_factor = nullptr)
{
    DeviceGuard dg(c->device);
    if (!dg.ok) return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
    const uint64_t M = c->N, N1 = (uint64_t)q * M;
    const uint32_t wN1 = gf::h_root((uint32_t)N1), wN1i = gf::h_inv(wN1);
    const uint32_t inv = gf::h_inv((uint32_t)N1);
    std::vector<uint32_t> dsc(N1), twd((size_t)M * (q - 1)), twu((size_t)M * (q - 1)), dfi, dff;
    {
        std::vector<uint32_t> pw(N1);  // w_(2 N1)^j / N1 by coefficient index, or the caller's factors (a transform context)
        if (custom_factor) {
            for (uint64_t j = 0; j < N1; j++) pw[j] = gf::h_to_mont(custom_factor[j] % gf::P);
        } else {
            const uint32_t w2 = gf::h_root((uint32_t)(2 * N1));
            uint32_t d = inv;
            for (uint64_t j = 0; j < N1; j++) {
                pw[j] = gf::h_to_mont(d);
                d = gf::h_mul(d, w2);
            }
        }
        for (uint64_t j1 = 0; j1 < (uint64_t)q; j1++)
            for (uint64_t r = 0; r < M; r++) dsc[j1 * M + r] = pw[(uint64_t)q * bitrev_host((uint32_t)r, c->n) + j1];
    }
    for (uint64_t i2 = 0; i2 < M; i2++) {
        const uint32_t a = gf::h_pow(wN1, i2), b = gf::h_pow(wN1i, i2);
        uint32_t x = 1, y = 1;
        for (int j = 1; j < q; j++) {
            x = gf::h_mul(x, a);
            y = gf::h_mul(y, b);
            twu[i2 * (q - 1) + j - 1] = gf::h_to_mont(x);
            twd[i2 * (q - 1) + j - 1] = gf::h_to_mont(y);
        }
    }
    const uint32_t wq = gf::h_pow(wN1, M);
    dff = radix_dft_table(q, wq);
    dfi = radix_dft_table(q, gf::h_inv(wq));
    (void)hipFree(c->dscale);  // sized for one stripe by create_impl
    c->dscale = nullptr;
    int rc = upload_table(&c->dscale, dsc);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_tw_dif, twd);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_tw_dit, twu);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_dft_inv, dfi);
    if (rc == FASTECC_OK) rc = upload_table(&c->q_dft_fwd, dff);
    if (rc != FASTECC_OK) return rc;
    c->q = q;
    c->K = k_user;
    c->Mu = m_user;
    c->stripe_bytes = (size_t)N1 * c->S * 4;  // the staging stripe of the host-memory calls holds all q * N blocks
    c->parity_bytes = (size_t)m_user * c->S * 4;
    build_plans(c);  // the odd-radix level joins the plan: its own two passes, or fused into the outer tiles (other run lengths)
    return upload_twiddles(c);
}

extern "C" {

const char* fastecc_strerror(int code)
{
    switch (code) {
        case FASTECC_OK: return "ok";
        case FASTECC_E_INVAL: return "invalid argument";
        case FASTECC_E_NOMEM: return "out of memory";
        case FASTECC_E_DEVICE: return "HIP device or runtime error";
        case FASTECC_E_UNSUPPORTED: return "unsupported field or size";
        default: return "unknown error";
    }
}

int fastecc_version(void) { return FASTECC_VERSION; }

const char* fastecc_last_error_detail(void) { return g_detail; }

uint32_t fastecc_gf_mul(uint32_t x, uint32_t y) { return gf::h_mul(x % gf::P, y % gf::P); }
uint32_t fastecc_gf_pow(uint32_t x, uint32_t e) { return gf::h_pow(x % gf::P, e); }
uint32_t fastecc_gf_root(uint32_t order) { return (order == 0 || ((gf::P - 1u) % order) != 0) ? 0u : gf::h_root(order); }
uint32_t fastecc_gf_inv(uint32_t x) { return gf::h_inv(x % gf::P); }

// GF((2^61-1)^2) scalars: z[0] = re, z[1] = im; inputs are reduced mod p first
static gf61::Elem elem_of(const uint64_t z[2]) { return gf61::Elem{z[0] % gf61::P, z[1] % gf61::P}; }
int fastecc_gf61_mul(const uint64_t x[2], const uint64_t y[2], uint64_t out[2])
{
    if (!x || !y || !out) return FASTECC_E_INVAL;
    const gf61::Elem r = gf61::h_mul(elem_of(x), elem_of(y));
    out[0] = r.re;
    out[1] = r.im;
    return FASTECC_OK;
}
int fastecc_gf61_pow(const uint64_t x[2], uint64_t e, uint64_t out[2])
{
    if (!x || !out) return FASTECC_E_INVAL;
    const gf61::Elem r = gf61::h_pow(elem_of(x), e);
    out[0] = r.re;
    out[1] = r.im;
    return FASTECC_OK;
}
int fastecc_gf61_inv(const uint64_t x[2], uint64_t out[2])
{
    if (!x || !out) return FASTECC_E_INVAL;
    const gf61::Elem r = gf61::h_inv(elem_of(x));
    out[0] = r.re;
    out[1] = r.im;
    return FASTECC_OK;
}
int fastecc_gf61_root(uint64_t order, uint64_t out[2])
{
    if (!out) return FASTECC_E_INVAL;
    const gf61::Elem r = gf61::h_root(order);
    out[0] = r.re;
    out[1] = r.im;
    return (r.re | r.im) ? FASTECC_OK : FASTECC_E_INVAL;
}

int fastecc_create(fastecc_ctx** out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, int device)
{
    if (!out) return FASTECC_E_INVAL;
    *out = nullptr;
    const bool f61 = field == FASTECC_FIELD_GF_P61_SQUARED;
    if (field != FASTECC_FIELD_GF_FFF00001 && !f61) return FASTECC_E_UNSUPPORTED;
    if (k < 1 || n <= k || block_bytes == 0 || (block_bytes % (f61 ? 16 : 4)) != 0) return FASTECC_E_INVAL;
    // transform size: the next power of two (RS.md:23-27 "find N1 >= N ... extend input vector with zeroes")
    int lg = 1;
    while ((1ull << lg) < k && lg < 63) lg++;
    const uint64_t N1 = 1ull << lg, m = n - k;
    const bool pow2 = N1 == k;
    // parity blocks: k (the reference's configuration), 3k or 7k (further cosets), or any m <= N1: then the smallest
    // power-of-two count >= m (at least N1/16) is computed and the first m blocks are the parity
    int fold = 0, cosets = 1;
    if (pow2 && (n == 4 * k || n == 8 * k)) {
        cosets = (int)(n / k) - 1;
    } else {
        if (m > N1) return FASTECC_E_UNSUPPORTED;
        int lgm = 0;
        while ((1ull << lgm) < m) lgm++;
        fold = std::min(lg - lgm, 4);
    }
    // The 64-bit field always runs the (2 N1, N1) transform; other (n,k) of the rules above (zero extension, fewer parity blocks) work on
    // padded copies of the stripes, and parity block j is block j * 2^fold of the full parity — the same code definition as for
    // GF(0xFFF00001), without the kernels' bounds handling (RS.md:23-33 spells out exactly this: extend with zeroes, output some values).
    int p61_stride = 1;
    if (f61) {
        p61_stride = 1 << fold;
        fold = 0;
    }
    // root(2N) must exist: 2N | 2^20 (GF.md:20, RS.cpp:51); in GF(p61^2) 2N | 2^62, the bound is table memory
    if (lg > (f61 ? p61::MAX_LOG2_K : 19)) return FASTECC_E_UNSUPPORTED;
    if (!f61 && cosets > 1 && n > (1ull << 20)) return FASTECC_E_UNSUPPORTED;  // w_n must exist
    if (block_bytes / 4 > 0xFFFFFFFFull / 2) return FASTECC_E_UNSUPPORTED;
    const uint64_t n_internal = cosets > 1 ? n : N1 + (N1 >> fold);
    const int rc = create_impl(out, n_internal, N1, lg, block_bytes, field, device, fold, cosets, nullptr);
    if (rc == FASTECC_OK) {
        (*out)->K = k;
        (*out)->Mu = m;
        (*out)->p61_stride = p61_stride;
    }
    return rc;

}

int fastecc_create_ex(fastecc_ctx** out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, int device, unsigned flags)
{
    if (!out) return FASTECC_E_INVAL;
    *out = nullptr;
    if (flags & ~(unsigned)(FASTECC_CODE_MIXED_RADIX | FASTECC_CODE_TOP_RADIX2 | FASTECC_CODE_MIXED_RADIX_PFA)) return FASTECC_E_INVAL;
    if (flags & FASTECC_CODE_TOP_RADIX2) {
        if (flags != FASTECC_CODE_TOP_RADIX2 || field != FASTECC_FIELD_GF_FFF00001) return FASTECC_E_UNSUPPORTED;
        int lg = 0;
        while ((1ull << lg) < k) lg++;
        if (n != 2 * k || (1ull << lg) != k || lg < 12 || lg > 19 || block_bytes == 0 || (block_bytes % 4) != 0) return FASTECC_E_UNSUPPORTED;
        const uint64_t M = k / 2;
        int rc = create_impl(out, 2 * M, M, lg - 1, block_bytes, field, device, 0, 1, nullptr);
        if (rc != FASTECC_OK) return rc;
        rc = setup_mixed(*out, 2, k, k);
        if (rc != FASTECC_OK) {
            fastecc_destroy(*out);
            *out = nullptr;
        }
        return rc;
    }
    if (!(flags & (FASTECC_CODE_MIXED_RADIX | FASTECC_CODE_MIXED_RADIX_PFA))) return fastecc_create(out, n, k, block_bytes, field, device);
    if (field != FASTECC_FIELD_GF_FFF00001) return FASTECC_E_UNSUPPORTED;
    if (k < 1 || n <= k || block_bytes == 0 || (block_bytes % 4) != 0) return FASTECC_E_INVAL;
    // transform order: the smallest q * 2^m >= k with q in {1, 3, 5, 7, 9, 13, 15} — with FASTECC_CODE_MIXED_RADIX_PFA also the products
    // of coprime factors 21 ... 117 —, m >= 1 (NTT.md:43-46: "the next divider of 0xFFF00000 is only a few percents larger than N itself");
    // w_(2 q 2^m) must exist: 2^(m+1) | 2^20
    uint64_t best = 0;
    int bq = 1, bm = 0;
    for (int q : {1, 3, 5, 7, 9, 13, 15, 21, 35, 39, 45, 63, 65, 91, 105, 117}) {
        if (q > 15 && !(flags & FASTECC_CODE_MIXED_RADIX_PFA)) break;
        for (int m = 1; m <= 19; m++) {
            const uint64_t N1 = (uint64_t)q << m;
            if (N1 >= k && (best == 0 || N1 < best)) best = N1, bq = q, bm = m;
        }
    }
    if (best == 0 || n - k > best) return FASTECC_E_UNSUPPORTED;
    if (bq == 1) return fastecc_create(out, n, k, block_bytes, field, device);
    if (block_bytes / 4 > 0xFFFFFFFFull / 2) return FASTECC_E_UNSUPPORTED;
    const uint64_t M = 1ull << bm;
    int rc = create_impl(out, 2 * M, M, bm, block_bytes, field, device, 0, 1, nullptr);
    if (rc != FASTECC_OK) return rc;
    rc = setup_mixed(*out, bq, k, n - k);
    if (rc != FASTECC_OK) {
        fastecc_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

}  // extern "C"

// Everything after argument validation.  custom_factor (k plain values by coefficient index) replaces the encoder's
// w_2k^m / k table (create_transform_ctx).
static int create_impl(fastecc_ctx** out, uint64_t n, uint64_t k, int lg, uint64_t block_bytes, int field, int device, int fold,
                       int cosets, const uint32_t* custom_factor)
{
    const bool f61 = field == FASTECC_FIELD_GF_P61_SQUARED;
    int ndev = 0;
    {
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return hip_fail(e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount");
    }
    if (device < 0 || device >= ndev) return FASTECC_E_INVAL;

    fastecc_ctx* c = new (std::nothrow) fastecc_ctx();
    if (!c) return FASTECC_E_NOMEM;
    c->device = device;
    c->field = field;
    c->N = k;
    c->n = lg;
    c->S = block_bytes / 4;
    c->ld = c->S;
    c->stripe_bytes = (size_t)k * block_bytes;
    c->fold = fold;
    c->cosets = cosets;
    c->M = n - k;
    c->K = k;
    c->Mu = n - k;
    c->parity_bytes = (size_t)(n - k) * block_bytes;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->cus = cus;
        else (void)hipGetLastError();
    }
    c->classic_plan = custom_factor != nullptr;  // the decoder's transform contexts (and the stand-alone transform's): see build_plans
    if (!f61) build_plans(c);

    DeviceGuard dg(device);
    if (dg.ok && !f61) {  // the engine's code objects, loaded once per process and device (kernels.hpp)
        static std::mutex preload_mu;
        static bool preloaded[64] = {};
        std::lock_guard<std::mutex> lk(preload_mu);
        if (device < 64 && !preloaded[device]) {
            preload_pass_kernels();
            preload_tile_kernels();
            preloaded[device] = true;
        }
    }
    if (!dg.ok) {
        delete c;
        return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
    }
    if (f61) {
        int rc = cosets > 1 ? p61::create_cosets(&c->p61, lg, block_bytes / 16, cosets, g_detail, sizeof g_detail)
                            : p61::create(&c->p61, lg, block_bytes / 16, g_detail, sizeof g_detail);
        if (rc == FASTECC_OK) {
            const hipError_t e = hipMalloc((void**)&c->factor, 8);  // counter of fastecc_check_range
            if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(counter)");
        }
        if (rc != FASTECC_OK) {
            fastecc_destroy(c);
            return rc;
        }
        c->plan_text = p61::plan_string(c->p61);
        *out = c;
        return FASTECC_OK;
    }

    // ---- tables: per-level twiddles for the plan, and the per-block factors w_2N^i / N of RS.cpp:51-54 ----
    const uint64_t N = k;
    const bool ntt_only = custom_factor == NTT_ONLY;  // create_ntt_ctx: the stand-alone transform's passes only, no per-block factors
    if (ntt_only) custom_factor = nullptr;
    std::vector<uint32_t> dsc(ntt_only ? 0 : N * cosets);
    const uint32_t invN = gf::h_inv((uint32_t)N);
    if (custom_factor || ntt_only) c->encode_direct_max = 0;  // a transform context is not the encoder's polynomial evaluation: always the pipeline
    for (int t = 0; t < cosets && custom_factor; t++) {
        for (uint64_t i = 0; i < N; i++) {
            const uint32_t f = custom_factor[i] >= gf::P ? custom_factor[i] - gf::P : custom_factor[i];
            dsc[bitrev_host((uint32_t)i, lg)] = gf::h_mont_mul(f, gf::MONT_R2);
        }
    }
    for (int t = 0; t < cosets && !custom_factor && !ntt_only; t++) {
        // coset t: generator w_(2^j k)^c with j = floor(log2(t + 1)) + 1 and c the (t + 2 - 2^(j-1))-th odd number
        int j = 1;
        while ((1 << j) - 1 <= t) j++;
        const uint32_t cth_odd = 2u * (uint32_t)(t + 1 - (1 << (j - 1))) + 1u;
        const uint32_t gen = gf::h_pow(gf::h_root((uint32_t)(N << j)), cth_odd);
        const uint32_t gen_m = gf::h_to_mont(gen);
        uint32_t d = gf::h_to_mont(invN);  // Montgomery form throughout
        for (uint64_t i = 0; i < N; i++) {
            dsc[t * N + bitrev_host((uint32_t)i, lg)] = d;  // by position: position p holds coefficient bitrev(p)
            d = gf::h_mont_mul(d, gen_m);
        }
    }
    int rc = upload_twiddles(c);
    if (rc == FASTECC_OK && !ntt_only) rc = upload_table(&c->dscale, dsc);
    if (rc == FASTECC_OK) {
        const hipError_t e = hipMalloc((void**)&c->factor, (ntt_only ? 2 : N) * 4);
        if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(factor)");
    }
    if (rc != FASTECC_OK) {
        fastecc_destroy(c);
        return rc;
    }
    *out = c;
    return FASTECC_OK;
}

namespace fastecc {

CtxInfo info_of(const fastecc_ctx* c)
{
    return CtxInfo{c->device, c->field, c->fold, c->cosets, c->n, c->N, c->S, c->ld, c->K != c->N || c->Mu != c->M, c->K, c->Mu, c->q, c->decode_direct_max, c->direct_kernel, c->p61_stride, c->decode_split};
}
DecodeState*& decoder_of(fastecc_ctx* c) { return c->decoder; }
Sharded*& sharded_of(fastecc_ctx* c) { return c->sharded; }
p61::Decoder*& decoder61_of(fastecc_ctx* c) { return c->decoder61; }
p61::Path* p61_path_of(fastecc_ctx* c) { return c->p61; }
void* profile_scope_begin(fastecc_ctx* c, hipStream_t st, const char* name, uint64_t bytes)
{
    return c->profiling ? new (std::nothrow) ProfScope(c, st, name, bytes) : nullptr;
}
void profile_scope_end(void* scope) { delete (ProfScope*)scope; }
const p61::LaunchHooks* p61_profile_hooks(fastecc_ctx* c, void** keep)
{
    *keep = nullptr;
    if (!c->profiling) return nullptr;
    P61Hooks* hk = new (std::nothrow) P61Hooks(c);
    *keep = hk;
    return hk ? &hk->h : nullptr;
}
void p61_profile_done(void* keep) { delete (P61Hooks*)keep; }
std::mutex& mutex_of(fastecc_ctx* c) { return c->mu; }
fastecc_ctx* new_shell_ctx(int root_device, int field, uint64_t k, uint64_t m, uint64_t block_bytes)
{
    fastecc_ctx* c = new (std::nothrow) fastecc_ctx();
    if (!c) return nullptr;
    c->device = root_device;
    c->field = field;
    c->N = c->K = k;
    c->M = c->Mu = m;
    c->S = c->ld = block_bytes / 4;
    c->stripe_bytes = (size_t)k * block_bytes;
    c->parity_bytes = (size_t)m * block_bytes;
    return c;
}
void set_plan_text(fastecc_ctx* c, const std::string& t) { c->plan_text = t; }
int columns_supported(const fastecc_ctx* c)
{
    return !c->sharded && c->q == 1 && c->fold == 0 && c->cosets == 1 && c->K == c->N && c->Mu == c->M && c->slabs <= 1;
}
int download_pageable(fastecc_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st) { return stage_download(c, dst, src, bytes, st); }
int stage_rect(fastecc_ctx* c, bool to_device, void* host, size_t host_pitch, void* dev, size_t dev_pitch, size_t width, size_t rows, hipStream_t st, int threads)
{
    return stage_transfer(c, StageJob{to_device, (char*)host, host_pitch, (char*)dev, dev_pitch, width, rows}, st, threads);
}
void set_error_detail(const char* what, hipError_t e) { (void)hip_fail(e, what); }
void set_error_text(const char* text) { snprintf(g_detail, sizeof g_detail, "%s", text ? text : ""); }

int create_transform_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int fold, const uint32_t* factor, int device)
{
    if (!out || !factor || log2k < 1 || log2k > 20 || fold < 0 || fold > 4 || fold > log2k || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t k = 1ull << log2k;
    return create_impl(out, k + (k >> fold), k, log2k, block_bytes, FASTECC_FIELD_GF_FFF00001, device, fold, 1, factor);
}

int create_ntt_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int device)
{
    if (!out || log2k < 1 || log2k > 20 || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t k = 1ull << log2k;
    return create_impl(out, 2 * k, k, log2k, block_bytes, FASTECC_FIELD_GF_FFF00001, device, 0, 1, NTT_ONLY);
}

namespace {
// dscale[bitrev(m)] = m * scale + offset in Montgomery form; scale_mm = scale * 2^64 mod p (mul_mont divides by 2^32 once), offset_m = offset * 2^32
__global__ __launch_bounds__(256) void ramp_factor_kernel(uint32_t* __restrict__ dsc, uint32_t N, int lg, uint32_t scale_mm, uint32_t offset_m)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < N) dsc[__brev(m) >> (32 - lg)] = gf::add(gf::mul_mont(m, scale_mm), offset_m);
}
}  // namespace

int create_ramp_transform_ctx(fastecc_ctx** out, int log2k, uint64_t block_bytes, int fold, uint32_t scale, int device, uint32_t offset)
{
    if (!out || log2k < 1 || log2k > 20 || fold < 0 || fold > 4 || fold > log2k || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t k = 1ull << log2k;
    int rc = create_impl(out, k + (k >> fold), k, log2k, block_bytes, FASTECC_FIELD_GF_FFF00001, device, fold, 1, NTT_ONLY);
    if (rc != FASTECC_OK) return rc;
    fastecc_ctx* c = *out;
    DeviceGuard dg(device);
    hipError_t e = dg.ok ? hipMalloc((void**)&c->dscale, k * 4) : hipErrorInvalidDevice;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(ramp_factor_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, nullptr, c->dscale, (uint32_t)k, log2k,
                           gf::h_to_mont(gf::h_to_mont(scale % gf::P)), gf::h_to_mont(offset % gf::P));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        rc = hip_fail(e, "create_ramp_transform_ctx");
        fastecc_destroy(c);
        *out = nullptr;
    }
    return rc;
}

int create_mixed_transform_ctx(fastecc_ctx** out, int q, int log2m, uint64_t block_bytes, const uint32_t* factor, int device)
{
    if (!out || !factor || !radix_supported(q) || log2m < 1 || log2m > 20 || block_bytes == 0 || (block_bytes % 4)) return FASTECC_E_INVAL;
    *out = nullptr;
    const uint64_t M = 1ull << log2m;
    const std::vector<uint32_t> ones((size_t)M, 1u);  // replaced by setup_mixed below
    int rc = create_impl(out, 2 * M, M, log2m, block_bytes, FASTECC_FIELD_GF_FFF00001, device, 0, 1, ones.data());
    if (rc != FASTECC_OK) return rc;
    rc = setup_mixed(*out, q, (uint64_t)q * M, (uint64_t)q * M, factor);
    if (rc != FASTECC_OK) {
        fastecc_destroy(*out);
        *out = nullptr;
    }
    return rc;
}

int mixed_dif(fastecc_ctx* c, const uint32_t* in, uint32_t* out, hipStream_t st)
{
    if (c->q <= 1 || c->ntt_plan.empty()) return FASTECC_E_UNSUPPORTED;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    int vec = 4;
    const uintptr_t bits = (uintptr_t)in | (uintptr_t)out;
    while (vec > 1 && ((c->S % vec) != 0 || (c->ld % vec) != 0 || (bits % (4u * vec)) != 0)) vec >>= 1;
    RadixArgs a{};
    a.S = (uint32_t)c->S;
    a.ld = (uint32_t)c->ld;
    a.M = (uint32_t)c->N;
    a.in = in;
    a.out = out;
    a.dft = c->q_dft_inv;
    a.tw = c->q_tw_dif;
    HIP_TRY(launch_radix(c->q, false, vec, a, st));
    return run_passes(c, c->ntt_plan, out, out, twiddle_table(c, TW_NTT_INV, st), twiddle_table(c, TW_NTT_INV, st), st, 0, 0, nullptr, (uint32_t)c->q);
}

int transform_bitrev(fastecc_ctx* c, const uint32_t* in, uint32_t* out, bool dit, bool inverse_roots, uint32_t width, hipStream_t st)
{
    if (c->p61 || c->sharded || c->q > 1 || c->ntt_plan.empty() || width == 0 || width > c->S) return FASTECC_E_UNSUPPORTED;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    const uint32_t* tw = inverse_roots ? twiddle_table(c, TW_NTT_INV, st) : twiddle_table(c, TW_NTT_FWD, st);
    if (!dit) return run_passes(c, c->ntt_plan, in, out, tw, tw, st, 0, width);
    // the stand-alone plan mirrored: the same chunks bottom up as DIT passes; level for level the same register runs, so
    // the level-packed tables of the DIF plan serve both
    std::vector<Pass> up(c->ntt_plan.rbegin(), c->ntt_plan.rend());
    for (Pass& p : up) p.mode = MODE_DIT;
    return run_passes(c, up, in, out, tw, tw, st, 0, width);
}

int run_gathered(fastecc_ctx* c, const uint32_t* even_blocks, const uint32_t* odd_blocks, const uint32_t* row_factor, uint32_t* out,
                 hipStream_t st)
{
    // the first pass must be able to read the two half stripes itself: a register DIF pass or a two-window DIF tile
    if (c->encode_plan.empty() || c->p61) return FASTECC_E_UNSUPPORTED;
    const Pass& p0 = c->encode_plan[0];
    if (p0.mode != MODE_DIF || p0.s < 1 || (p0.tile && p0.wide != 2)) return FASTECC_E_UNSUPPORTED;
    // unstaged plans (fold 0: the first pass writes `out`, the rest runs in place on it) recognise the first pass by its source: `out` must be another buffer
    if (c->fold == 0 && c->cosets == 1 && (const uint32_t*)out == even_blocks) return FASTECC_E_UNSUPPORTED;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    CallBounds cb;
    cb.gather_odd = odd_blocks;
    cb.gather_factor = row_factor;
    return run_passes(c, c->encode_plan, even_blocks, out, twiddle_table(c, TW_ENC_DIF, st), twiddle_table(c, TW_ENC_DIT, st), st, 0, 0, nullptr, 1, cb);
}

bool gather_tile_order(const fastecc_ctx* c, std::vector<uint32_t>& order)
{
    order.clear();
    if (c->encode_plan.empty() || !c->encode_plan[0].tile) return false;
    const Pass& p = c->encode_plan[0];
    // layout of ntt_tile_kernel's paired load (PAIR tiles): tile (hi, lo), wave g, register pair i, +T/2, half-wave
    const int logt = p.logr, l2 = p.logr - p.rlog - 1, s = p.s;
    const uint32_t T = 1u << logt, G = 1u << l2, R = 1u << p.rlog;
    order.resize(c->N);
    size_t k = 0;
    for (uint64_t tile = 0; tile < (c->N >> logt); tile++) {
        const uint32_t lo = (uint32_t)(tile & ((1u << s) - 1u)), hi = (uint32_t)(tile >> s);
        const uint32_t pos0 = (hi << (s + logt)) + lo;
        for (uint32_t g = 0; g < G; g++)
            for (uint32_t i = 0; i < R / 2; i++)
                for (uint32_t far = 0; far < 2; far++)
                    for (uint32_t half = 0; half < 2; half++)
                        order[k++] = pos0 + ((g + 2 * i * G + half * G + far * (T / 2)) << s);
    }
    return true;
}

// The same order written by a kernel (the host loop above and the upload of its N words were 0.5 ms a piece in the first fastecc_decode_prepare).
namespace {
__global__ __launch_bounds__(256) void tile_order_kernel(uint32_t* __restrict__ order, uint32_t N, int logt, int l2, int rlog, int s)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    const uint32_t T = 1u << logt, G = 1u << l2;
    const uint32_t half = k & 1u, far = (k >> 1) & 1u, i = (k >> 2) & ((1u << (rlog - 1)) - 1u), g = (k >> (rlog + 1)) & (G - 1u), tile = k >> logt;
    const uint32_t lo = tile & ((1u << s) - 1u), hi = tile >> s;
    order[k] = ((hi << (s + logt)) + lo) + ((g + 2u * i * G + half * G + far * (T / 2u)) << s);
}
}  // namespace

bool gather_tile_order_device(const fastecc_ctx* c, uint32_t* order, hipStream_t st)
{
    if (c->encode_plan.empty() || !c->encode_plan[0].tile || c->N > 0x7FFFFFFFull) return false;
    const Pass& p = c->encode_plan[0];
    hipLaunchKernelGGL(tile_order_kernel, dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, st, order, (uint32_t)c->N, p.logr, p.logr - p.rlog - 1, p.rlog, p.s);
    return hipGetLastError() == hipSuccess;
}

// Two contexts whose first passes read their blocks in the same order (what comparing two gather_tile_order vectors decided)
bool same_tile_order(const fastecc_ctx* a, const fastecc_ctx* b)
{
    if (a->encode_plan.empty() || b->encode_plan.empty() || !a->encode_plan[0].tile || !b->encode_plan[0].tile) return false;
    const Pass &p = a->encode_plan[0], &q = b->encode_plan[0];
    return a->N == b->N && p.logr == q.logr && p.rlog == q.rlog && p.s == q.s;
}

// ---- the decoder's split transform (decode.hip, "even / odd split") on a context of k blocks whose per-block factors are (2m + k) / 2k ----
bool split_decode_supported(const fastecc_ctx* c)
{
    if (c->p61 || c->q > 1 || c->fold != 0 || c->cosets != 1 || c->encode_plan.size() != 3 || !c->dscale || c->ld != c->S) return false;
    const Pass &p0 = c->encode_plan[0], &p1 = c->encode_plan[1], &p2 = c->encode_plan[2];
    // a pair tile down (slim, or the 128-block one), the split 1024-block MID tile, the same tile up: k = 2^17, 2^18, 2^19 with the default plan
    return p0.mode == MODE_DIF && p0.tile && p0.pair && (p0.wide == 0 ? (p0.rlog == 4 || p0.logr == 7) : p0.rlog == 4) && p1.mode == MODE_MID && p1.tile && p1.pair && p1.logr == 10 && c->split2 &&
           p2.mode == MODE_DIT && p2.tile && p2.wide == p0.wide && p2.rlog == p0.rlog && p2.logr == p0.logr;
}

int split_impulse_max() { return IMPULSE_MAX; }
uint32_t split_decode_groups(const fastecc_ctx* c) { return 1u << c->encode_plan[0].s; }       // block groups of the first pass
uint32_t split_decode_group_rows(const fastecc_ctx* c) { return 1u << c->encode_plan[0].logr; }  // blocks per group: i = group + (t << s)

int run_split_decode(fastecc_ctx* c, const uint32_t* data, const uint32_t* parity, const uint32_t* data_rows_factor, const uint32_t* parity_rows_factor,
                     uint32_t parity_groups, const uint32_t* parity_pos_factor, uint32_t* q, uint32_t* r1, uint32_t* r2, const uint32_t* out_rows_factor,
                     uint32_t* out, const uint32_t* impulse_table, uint32_t data_blocks, uint32_t parity_blocks, hipStream_t st, const SplitRepair* odd,
                     const uint32_t* small_addend, uint32_t addend_shift)
{
    if (!split_decode_supported(c)) return FASTECC_E_UNSUPPORTED;
    if (small_addend ? (addend_shift < 1 || addend_shift > 5) : (parity_groups < 1 || parity_groups > split_decode_groups(c))) return FASTECC_E_UNSUPPORTED;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    const uint32_t *twd = twiddle_table(c, TW_ENC_DIF, st), *twu = twiddle_table(c, TW_ENC_DIT, st);
    const std::vector<Pass> first{c->encode_plan[0]};
    // the levels MID takes on its way down, as a DIF tile of their own (the level tables are packed by level: the same table serves)
    const std::vector<Pass> low{Pass{MODE_DIF, c->encode_plan[1].logr, 0, true, true, 5}};
    const std::vector<Pass> rest{c->encode_plan[1], c->encode_plan[2]};
    CallBounds cq, cr, cm;
    cq.rows_factor = data_rows_factor;
    cr.rows_factor = parity_rows_factor;
    // zero-extended codes: the stripes hold fewer than k blocks — the rest reads as zero (and a scattered block never lies beyond: its factor is 0)
    cq.in_rows = data_blocks < c->N ? data_blocks : 0;
    cr.in_rows = parity_blocks < c->N ? parity_blocks : 0;
    cr.groups = parity_groups;
    // r~ after all its DIF levels: the k-block stripe r2 — or, when the parity blocks in use sit at multiples of 2^shift only, the (k >> shift)-
    // block transform of those (the caller's small_addend), each of whose blocks stands for 2^shift consecutive positions
    cm.addend = small_addend ? small_addend : r2;
    cm.addend_shift = small_addend ? addend_shift : 0;
    cm.addend_factor = parity_pos_factor;
    if (out_rows_factor) {  // the last pass scatters: block i of the result, times its factor, goes to out[i] where that factor is not zero
        cm.rows_out_factor = out_rows_factor;
        cm.final_out = out;
        cm.out_rows = data_blocks < c->N ? data_blocks : 0;
    }
    int rc = run_passes(c, first, data, q, twd, twu, st, 0, 0, nullptr, 1, cq);            // q~ : top levels of the data half
    if (!small_addend) {
        if (rc == FASTECC_OK) rc = run_passes(c, first, parity, r1, twd, twu, st, 0, 0, nullptr, 1, cr);  // r~ : top levels, the groups that hold parity blocks in use
        CallBounds cl;
        if (impulse_table && parity_groups <= 16u * IMPULSE_MAX) {  // few groups: six of the ten low levels as a multiply-add per block in use (MODE_DIF_IMPULSE)
            cl.impulse_table = impulse_table;
            cl.impulse_rows = parity_groups;
        }
        if (rc == FASTECC_OK) rc = run_passes(c, low, r1, r2, twd, twu, st, 0, 0, nullptr, 1, cl);       // r~ : low levels (r1 is zero outside those groups)
    }
    // g = fq q~ + fr r~, and the transform back up.  With the odd positions wanted as well MID also stores q~ (its tiles after the first half).
    if (odd) cm.keep = odd->q2;
    if (rc == FASTECC_OK) rc = run_passes(c, rest, q, q, twd, twu, st, 0, 0, nullptr, 1, cm);
    if (rc == FASTECC_OK && odd) {
        // x p'(x) at the ODD positions (the parity blocks): the k-point transform of h[m] = w^m (m P[m] - (m+k) P[m+k]) = -1/2 w^m q~[m] +
        // (2m+k)/2k r~[m] — MID's second half alone on the stored q~, the factor tables exchanged (the context's own table now scales the addend)
        CallBounds ch;
        ch.addend = cm.addend;
        ch.addend_shift = cm.addend_shift;
        ch.addend_factor = c->dscale;
        ch.mid_up = true;
        ch.dscale_override = odd->data_pos_factor;
        ch.rows_out_factor = odd->out_rows_factor;
        ch.final_out = odd->out;
        ch.out_rows = parity_blocks < c->N ? parity_blocks : 0;  // (positions beyond a shorter parity stripe count as lost: nothing is stored there)
        rc = run_passes(c, rest, odd->q2, odd->q2, twd, twu, st, 0, 0, nullptr, 1, ch);
    }
    return rc;
}

CallScope::CallScope(fastecc_ctx* c) : c_(c) { c_->mu.lock(); }
CallScope::~CallScope() { c_->mu.unlock(); }
int CallScope::begin(hipStream_t st) { return order_internal_buffers(c_, st); }
int CallScope::end(hipStream_t st) { return mark_internal_buffers(c_, st); }
int CallScope::wait_idle()
{
    if (c_->buf_used) HIP_TRY(hipEventSynchronize(c_->buf_event));
    return FASTECC_OK;
}

int p61_work_stripes(fastecc_ctx* c, uint64_t** data_full, uint64_t** parity_full)
{
    const size_t row = (size_t)c->S * 4;
    if (!c->scratch) HIP_TRY(hipMalloc((void**)&c->scratch, c->N * row));
    if (!c->parbuf) HIP_TRY(hipMalloc((void**)&c->parbuf, c->N * row));
    *data_full = (uint64_t*)c->scratch;
    *parity_full = (uint64_t*)c->parbuf;
    return FASTECC_OK;
}

int encode_unlocked(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st)
{
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    return encode_device(c, data, parity, st);
}

int scratch_of(fastecc_ctx* c, uint32_t** out)
{
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    if (!c->scratch) HIP_TRY(hipMalloc((void**)&c->scratch, c->N * c->ld * 4));
    *out = c->scratch;
    return FASTECC_OK;
}

}  // namespace fastecc

extern "C" {

void fastecc_destroy(fastecc_ctx* c)
{
    if (!c) return;
    destroy_sharded(c->sharded);
    c->sharded = nullptr;
    destroy_decode_state(c->decoder);
    c->decoder = nullptr;
    DeviceGuard dg(c->device);
    direct_encode_destroy(c->direct_enc);
    c->direct_enc = nullptr;
    p61::destroy_decoder(c->decoder61);
    c->decoder61 = nullptr;
    for (ProfileRec& r : c->prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    if (c->slab_ready) {
        for (int h = 0; h < fastecc_ctx::MAX_SLABS; h++) {
            if (c->slab_stream[h]) (void)hipStreamSynchronize(c->slab_stream[h]);
            if (c->slab_first_done[h]) (void)hipEventDestroy(c->slab_first_done[h]);
            if (c->slab_done[h]) (void)hipEventDestroy(c->slab_done[h]);
            if (c->slab_stream[h]) (void)hipStreamDestroy(c->slab_stream[h]);
        }
        if (c->slab_fork) (void)hipEventDestroy(c->slab_fork);
    }
    if (c->buf_event) (void)hipEventDestroy(c->buf_event);
    p61::destroy(c->p61);
    for (hipEvent_t e : c->tw_event)
        if (e) (void)hipEventDestroy(e);
    if (c->tw_enc_dif) (void)hipFree(c->tw_enc_dif);
    if (c->tw_enc_dit) (void)hipFree(c->tw_enc_dit);
    if (c->tw_ntt_fwd) (void)hipFree(c->tw_ntt_fwd);
    if (c->tw_ntt_inv) (void)hipFree(c->tw_ntt_inv);
    if (c->dscale) (void)hipFree(c->dscale);
    if (c->factor) (void)hipFree(c->factor);
    if (c->dbuf) (void)hipFree(c->dbuf);
    if (c->rawbuf) (void)hipFree(c->rawbuf);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->parbuf) (void)hipFree(c->parbuf);
    if (c->hostpar) (void)hipFree(c->hostpar);
    if (c->tw_fold_dit) (void)hipFree(c->tw_fold_dit);
    for (uint32_t* t : {c->q_tw_dif, c->q_tw_dit, c->q_dft_inv, c->q_dft_fwd, c->mixbuf})
        if (t) (void)hipFree(t);
    for (fastecc_ctx::StageRing* r : {&c->stage_up, &c->stage_down}) {
        if (r->slots) (void)hipHostFree(r->slots);
        for (hipEvent_t e : r->event)
            if (e) (void)hipEventDestroy(e);
    }
    delete c;
}

int fastecc_encode(fastecc_ctx* c, const void* data, void* parity, int mem_kind, void* stream)
{
    if (!c || !data || !parity) return FASTECC_E_INVAL;
    if (((uintptr_t)data | (uintptr_t)parity) & (c->field == FASTECC_FIELD_GF_P61_SQUARED ? 15u : 3u)) return FASTECC_E_INVAL;
    if (c->sharded) return sharded_encode_stripe(c, data, parity, mem_kind, (hipStream_t)stream);
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    if (c->Mu > c->K && parity == data) return FASTECC_E_INVAL;  // the parity is larger than the data
    CallLock lk(c->mu);
    if (mem_kind == FASTECC_MEM_DEVICE) {
        // the reference's configuration touches nothing but the caller's buffers and the read-only tables: calls on
        // different streams may overlap on the device.  The other codes work through scratch stripes of the context.
        const bool internal = c->fold != 0 || c->cosets != 1 || c->Mu != c->M || c->slabs > 1 || c->q > 1 || (c->p61 && c->K != c->N) || direct_encode_applies(c);
        if (!internal) return encode_device(c, (const uint32_t*)data, (uint32_t*)parity, st);
        return with_internal_buffers(c, st, [&] { return encode_device(c, (const uint32_t*)data, (uint32_t*)parity, st); });
    }
    if (mem_kind == FASTECC_MEM_HOST_PINNED) {
        if (c->p61 || c->q > 1 || c->fold != 0 || c->cosets != 1 || c->ld != c->S || c->K != c->N || c->Mu != c->M) return FASTECC_E_UNSUPPORTED;
        return with_internal_buffers(c, st, [&] { return encode_host_pinned(c, (const uint32_t*)data, (uint32_t*)parity, st); });
    }
    if (mem_kind != FASTECC_MEM_HOST) return FASTECC_E_INVAL;
    if (c->ld != c->S) return FASTECC_E_UNSUPPORTED;  // host stripes are always contiguous
    return with_internal_buffers(c, st, [&]() -> int {
        if (!c->p61 && c->q <= 1 && c->fold == 0 && c->cosets == 1 && c->K == c->N && c->Mu == c->M && c->host_pipeline) {
            const int rp = encode_host_pageable(c, (const uint32_t*)data, (uint32_t*)parity, st);
            if (rp != FASTECC_E_UNSUPPORTED) return rp;
        }
        int rc = ensure_dbuf(c);
        if (rc != FASTECC_OK) return rc;
        const size_t block_bytes = (size_t)c->S * 4;
        HIP_TRY(hipMemcpyAsync(c->dbuf, data, c->K * block_bytes, hipMemcpyHostToDevice, st));
        uint32_t* dpar = c->dbuf;
        if (c->cosets > 1 || c->Mu > c->K) {  // more parity than the data stripe has room for
            if (!c->hostpar) HIP_TRY(hipMalloc((void**)&c->hostpar, c->Mu * block_bytes));
            dpar = c->hostpar;
        }
        rc = encode_device(c, c->dbuf, dpar, st);
        if (rc != FASTECC_OK) return rc;
        return stage_download(c, parity, dpar, c->Mu * block_bytes, st);
    });
}

int fastecc_encode_columns(fastecc_ctx* c, const void* data, void* parity, uint64_t col0, uint64_t width, void* stream)
{
    if (!c || !data || !parity || (((uintptr_t)data | (uintptr_t)parity) & 3u)) return FASTECC_E_INVAL;
    if (c->sharded) return FASTECC_E_UNSUPPORTED;
    if (width == 0 || col0 > c->S || width > c->S - col0) return FASTECC_E_INVAL;
    if (!columns_supported(c)) return FASTECC_E_UNSUPPORTED;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    CallLock lk(c->mu);
    if (c->p61) {  // the range is given in 4-byte words like everywhere in this ABI: whole 16-byte elements only
        if ((col0 % 4) != 0 || (width % 4) != 0 || ((((uintptr_t)data | (uintptr_t)parity)) & 15u)) return FASTECC_E_INVAL;
        P61Hooks hk(c);
        return p61::encode_columns(c->p61, (const uint64_t*)data, (uint64_t*)parity, col0 / 4, width / 4, (hipStream_t)stream,
                                   c->profiling ? &hk.h : nullptr);
    }
    return run_passes(c, c->encode_plan, (const uint32_t*)data, (uint32_t*)parity, twiddle_table(c, TW_ENC_DIF, (hipStream_t)stream), twiddle_table(c, TW_ENC_DIT, (hipStream_t)stream), (hipStream_t)stream,
                      (uint32_t)col0, (uint32_t)width);
}

int fastecc_encode_batch(fastecc_ctx* c, const void* data, void* parity, uint64_t count, void* stream)
{
    if (!c || !data || !parity || count == 0 || (((uintptr_t)data | (uintptr_t)parity) & 3u)) return FASTECC_E_INVAL;
    if (c->sharded) return FASTECC_E_UNSUPPORTED;
    if (c->p61 || c->q > 1 || c->fold != 0 || c->cosets != 1 || c->K != c->N || c->Mu != c->M) return FASTECC_E_UNSUPPORTED;  // n = 2k = 2^m
    if (count * c->N > 0x7FFFFFFFull || count > 0xFFFFFFFFull) return FASTECC_E_UNSUPPORTED;  // 32-bit block indices
    if (c->ld != c->S) return FASTECC_E_UNSUPPORTED;  // stripes of a batch are contiguous (b * k * block_bytes apart)
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    CallLock lk(c->mu);
    return run_passes(c, c->encode_plan, (const uint32_t*)data, (uint32_t*)parity, twiddle_table(c, TW_ENC_DIF, (hipStream_t)stream), twiddle_table(c, TW_ENC_DIT, (hipStream_t)stream), (hipStream_t)stream, 0, 0,
                      nullptr, (uint32_t)count);
}

int fastecc_encode_blocks(fastecc_ctx* c, void* const* blocks)
{
    if (!c || !blocks) return FASTECC_E_INVAL;
    if (c->sharded) return FASTECC_E_UNSUPPORTED;
    if (c->Mu > c->K) return FASTECC_E_UNSUPPORTED;  // the in-place form has room for at most k parity blocks
    if (c->ld != c->S) return FASTECC_E_UNSUPPORTED;  // the staging stripe is contiguous
    for (uint64_t i = 0; i < c->K; i++)
        if (!blocks[i] || ((uintptr_t)blocks[i] & (c->p61 ? 15u : 3u))) return FASTECC_E_INVAL;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    CallLock lk(c->mu);
    int rc = order_internal_buffers(c, nullptr);
    if (rc == FASTECC_OK) rc = ensure_dbuf(c);
    if (rc != FASTECC_OK) return rc;
    // The blocks go up and the parity comes back through the staging rings (stage_transfer: helper threads gather / scatter whole blocks
    // between the caller's table of pointers and pinned slots that the copy engine moves): 2 + 2 GiB in ~85 ms.  (Rounds 1-4: one thread
    // and one synchronous copy per 64 MiB, 325-365 ms.)
    const size_t bb = (size_t)c->S * 4;
    // RS.cpp's own table points into ONE buffer, block after block (RS.cpp:31-33): then the stripe goes up as one copy, like FASTECC_MEM_HOST
    // (the runtime's pageable upload runs at the link rate; it is its download that needs the ring)
    bool one_buffer = true;
    for (uint64_t i = 1; i < c->K && one_buffer; i++) one_buffer = (const char*)blocks[i] == (const char*)blocks[i - 1] + bb;
    if (one_buffer) HIP_TRY(hipMemcpyAsync(c->dbuf, blocks[0], c->K * bb, hipMemcpyHostToDevice, nullptr));
    else rc = stage_transfer(c, StageJob{true, nullptr, 0, (char*)c->dbuf, bb, bb, (size_t)c->K, blocks}, nullptr);
    if (rc != FASTECC_OK) return rc;
    rc = encode_device(c, c->dbuf, c->dbuf, nullptr);
    if (rc != FASTECC_OK) return rc;
    // the first n - k blocks receive the parity
    if (one_buffer) rc = stage_download(c, blocks[0], c->dbuf, c->Mu * bb, nullptr);
    else rc = stage_transfer(c, StageJob{false, nullptr, 0, (char*)c->dbuf, bb, bb, (size_t)c->Mu, blocks}, nullptr);
    if (rc != FASTECC_OK) return rc;
    return mark_internal_buffers(c, nullptr);
}

int fastecc_ntt(fastecc_ctx* c, void* data, int inverse, int mem_kind, void* stream)
{
    if (!c || !data || ((uintptr_t)data & (c->p61 ? 15u : 3u))) return FASTECC_E_INVAL;
    if (c->sharded) return FASTECC_E_UNSUPPORTED;
    if (c->ld != c->S) return FASTECC_E_UNSUPPORTED;  // a row pitch applies to fastecc_encode on device stripes only
    if (c->K != c->N || c->q > 1) return FASTECC_E_UNSUPPORTED;   // the stand-alone transform has power-of-two length
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    CallLock lk(c->mu);
    if (mem_kind == FASTECC_MEM_DEVICE) return ntt_device(c, (uint32_t*)data, inverse != 0, st);
    if (mem_kind != FASTECC_MEM_HOST) return FASTECC_E_INVAL;
    return with_internal_buffers(c, st, [&]() -> int {
        int rc = ensure_dbuf(c);
        if (rc != FASTECC_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->dbuf, data, c->stripe_bytes, hipMemcpyHostToDevice, st));
        rc = ntt_device(c, c->dbuf, inverse != 0, st);
        if (rc != FASTECC_OK) return rc;
        return stage_download(c, data, c->dbuf, c->stripe_bytes, st);
    });
}

int fastecc_scale_blocks(fastecc_ctx* c, void* data, uint32_t scale, uint32_t base, int mem_kind, void* stream)
{
    if (!c || !data || ((uintptr_t)data & 3u)) return FASTECC_E_INVAL;
    if (scale >= gf::P || base >= gf::P) return FASTECC_E_INVAL;
    if (c->p61 || c->sharded) return FASTECC_E_UNSUPPORTED;  // 32-bit scalars: GF(0xFFF00001) only
    if (c->ld != c->S || c->K != c->N || c->q > 1) return FASTECC_E_UNSUPPORTED;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    if (mem_kind != FASTECC_MEM_HOST && mem_kind != FASTECC_MEM_DEVICE) return FASTECC_E_INVAL;
    hipStream_t st = (hipStream_t)stream;
    std::vector<uint32_t> f(c->N);
    uint32_t cur = scale;
    for (uint64_t i = 0; i < c->N; i++) {
        f[i] = gf::h_to_mont(cur);
        cur = gf::h_mul(cur, base);
    }
    CallLock lk(c->mu);
    return with_internal_buffers(c, st, [&]() -> int {
        // the factor table is consumed by a kernel on `st`; a synchronous copy keeps the host vector's lifetime simple
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(c->factor, f.data(), c->N * 4, hipMemcpyHostToDevice));
        uint32_t* dev = (uint32_t*)data;
        if (mem_kind == FASTECC_MEM_HOST) {
            int rc = ensure_dbuf(c);
            if (rc != FASTECC_OK) return rc;
            HIP_TRY(hipMemcpyAsync(c->dbuf, data, c->stripe_bytes, hipMemcpyHostToDevice, st));
            dev = c->dbuf;
        }
        {
            ProfScope ps(c, st, "scale_rows");
            HIP_TRY(launch_scale_rows(dev, c->factor, (uint32_t)c->S, c->N, pick_vec(c, dev, dev), st));
        }
        if (mem_kind == FASTECC_MEM_HOST) {
            HIP_TRY(hipMemcpyAsync(data, c->dbuf, c->stripe_bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        return FASTECC_OK;
    });
}

int fastecc_gf_binary(fastecc_ctx* c, int op, const uint32_t* x, const uint32_t* y, uint32_t* out, uint64_t count, void* stream)
{
    if (!c || !x || !y || !out || op < 0 || op > 3) return FASTECC_E_INVAL;
    if (c->p61 || c->sharded) return FASTECC_E_UNSUPPORTED;  // 32-bit words: GF(0xFFF00001) only
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    CallLock lk(c->mu);
    ProfScope ps(c, (hipStream_t)stream, "gf_binary");
    HIP_TRY(launch_gf_binary(op, x, y, out, count, (hipStream_t)stream));
    return FASTECC_OK;
}

int fastecc_check_range(fastecc_ctx* c, const void* data, int mem_kind, void* stream, uint64_t* bad_words)
{
    if (!c || !data || !bad_words || ((uintptr_t)data & (c->p61 ? 15u : 3u))) return FASTECC_E_INVAL;  // as fastecc_encode
    if (mem_kind != FASTECC_MEM_HOST && mem_kind != FASTECC_MEM_DEVICE) return FASTECC_E_INVAL;
    if (c->sharded) return FASTECC_E_UNSUPPORTED;
    if (c->ld != c->S) return FASTECC_E_UNSUPPORTED;  // scans k * block_bytes contiguous bytes
    if (c->p61 && c->K != c->N) return FASTECC_E_UNSUPPORTED;  // the 64-bit field's scan covers the whole power-of-two stripe
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    CallLock lk(c->mu);
    return with_internal_buffers(c, st, [&]() -> int {
    const uint32_t* dev = (const uint32_t*)data;
    if (mem_kind == FASTECC_MEM_HOST) {
        int rc = ensure_dbuf(c);
        if (rc != FASTECC_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->dbuf, data, c->K * c->S * 4, hipMemcpyHostToDevice, st));
        dev = c->dbuf;
    }
    // c->factor (N >= 2 words of scratch) holds the 64-bit counter
    unsigned long long* counter = reinterpret_cast<unsigned long long*>(c->factor);
    HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
    if (c->p61) {
        ProfScope ps(c, st, "p61_count_out_of_range");
        const int rc = p61::count_out_of_range(c->p61, (const uint64_t*)dev, counter, st);
        if (rc != FASTECC_OK) return rc;
        ps.finish();
        unsigned long long found = 0;
        HIP_TRY(hipMemcpyAsync(&found, counter, sizeof found, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        *bad_words = found;
        return FASTECC_OK;
    }
    uint64_t words = c->K * c->S, head = 0;
    // the vector loop wants a 16-byte aligned start: count the few leading words on the host copy of them
    unsigned long long result = 0;
    while (((uintptr_t)(dev + head) & 15u) && head < words) head++;
    if (head) {
        uint32_t first[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyAsync(first, dev, head * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < head; i++) result += first[i] >= gf::P;
    }
    {
        ProfScope ps(c, st, "count_out_of_range");
        HIP_TRY(launch_count_out_of_range(dev + head, words - head, counter, st));
    }
    unsigned long long on_device = 0;
    HIP_TRY(hipMemcpyAsync(&on_device, counter, sizeof on_device, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *bad_words = result + on_device;
    return FASTECC_OK;
    });
}

// GF.md:72-104: W = S - 1 raw words per block <-> S packed words (pack_kernels.hip)
static int pack_args_ok(const fastecc_ctx* c, const void* a, const void* b)
{
    if (!c || !a || !b || (((uintptr_t)a | (uintptr_t)b) & 3u)) return FASTECC_E_INVAL;
    if (c->p61 || c->sharded) return FASTECC_E_UNSUPPORTED;         // the recoding is specific to p = 0xFFF00001
    if (c->S < 2 || c->S > 1025) return FASTECC_E_UNSUPPORTED;  // positions are 10-bit
    if (c->K != c->N || c->q > 1) return FASTECC_E_UNSUPPORTED;  // staging buffers are sized for power-of-two k
    return FASTECC_OK;
}

static int ensure_rawbuf(fastecc_ctx* c)
{
    if (c->rawbuf) return FASTECC_OK;
    HIP_TRY(hipMalloc((void**)&c->rawbuf, c->N * (c->S - 1) * 4));
    return FASTECC_OK;
}

int fastecc_pack_blocks(fastecc_ctx* c, const void* raw, void* packed, int mem_kind, void* stream)
{
    int rc = pack_args_ok(c, raw, packed);
    if (rc != FASTECC_OK) return rc;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t words = (uint32_t)c->S - 1;
    const uint64_t alg_bytes = c->N * (2ull * words + 1) * 4;
    CallLock lk(c->mu);
    if (mem_kind == FASTECC_MEM_DEVICE) {
        ProfScope ps(c, st, "pack_blocks", alg_bytes);
        HIP_TRY(launch_pack_blocks((const uint32_t*)raw, (uint32_t*)packed, words, (uint32_t)c->ld, c->N, st));
        return FASTECC_OK;
    }
    if (mem_kind != FASTECC_MEM_HOST) return FASTECC_E_INVAL;
    if (c->ld != c->S) return FASTECC_E_UNSUPPORTED;  // host stripes are always contiguous
    return with_internal_buffers(c, st, [&]() -> int {
        int rc2 = ensure_dbuf(c);
        if (rc2 == FASTECC_OK) rc2 = ensure_rawbuf(c);
        if (rc2 != FASTECC_OK) return rc2;
        HIP_TRY(hipMemcpyAsync(c->rawbuf, raw, c->N * words * 4, hipMemcpyHostToDevice, st));
        {
            ProfScope ps(c, st, "pack_blocks", alg_bytes);
            HIP_TRY(launch_pack_blocks(c->rawbuf, c->dbuf, words, (uint32_t)c->S, c->N, st));
        }
        HIP_TRY(hipMemcpyAsync(packed, c->dbuf, c->stripe_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        return FASTECC_OK;
    });
}

int fastecc_unpack_blocks(fastecc_ctx* c, const void* packed, void* raw, int mem_kind, void* stream, uint64_t* bad_blocks)
{
    int rc = pack_args_ok(c, packed, raw);
    if (rc != FASTECC_OK) return rc;
    DeviceGuard dg(c->device);
    if (!dg.ok) return FASTECC_E_DEVICE;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t words = (uint32_t)c->S - 1;
    const uint64_t alg_bytes = c->N * (2ull * words + 1) * 4;
    const uint32_t* src = (const uint32_t*)packed;
    uint32_t* dst = (uint32_t*)raw;
    if (mem_kind != FASTECC_MEM_HOST && mem_kind != FASTECC_MEM_DEVICE) return FASTECC_E_INVAL;
    if (mem_kind == FASTECC_MEM_HOST && c->ld != c->S) return FASTECC_E_UNSUPPORTED;
    CallLock lk(c->mu);
    return with_internal_buffers(c, st, [&]() -> int {  // the bad-block counter lives in c->factor
    int rc = FASTECC_OK;
    if (mem_kind == FASTECC_MEM_HOST) {
        rc = ensure_dbuf(c);
        if (rc == FASTECC_OK) rc = ensure_rawbuf(c);
        if (rc != FASTECC_OK) return rc;
        HIP_TRY(hipMemcpyAsync(c->dbuf, packed, c->stripe_bytes, hipMemcpyHostToDevice, st));
        src = c->dbuf;
        dst = c->rawbuf;
    }
    unsigned long long* counter = bad_blocks ? reinterpret_cast<unsigned long long*>(c->factor) : nullptr;  // N >= 2 words of scratch
    if (counter) HIP_TRY(hipMemsetAsync(counter, 0, sizeof(unsigned long long), st));
    {
        ProfScope ps(c, st, "unpack_blocks", alg_bytes);
        HIP_TRY(launch_unpack_blocks(src, dst, words, (uint32_t)c->ld, c->N, counter, st));
    }
    unsigned long long found = 0;
    if (counter) HIP_TRY(hipMemcpyAsync(&found, counter, sizeof found, hipMemcpyDeviceToHost, st));
    if (mem_kind == FASTECC_MEM_HOST) HIP_TRY(hipMemcpyAsync(raw, c->rawbuf, c->N * words * 4, hipMemcpyDeviceToHost, st));
    if (counter || mem_kind == FASTECC_MEM_HOST) HIP_TRY(hipStreamSynchronize(st));
    if (bad_blocks) *bad_blocks = found;
    return FASTECC_OK;
    });
}

}  // extern "C"
