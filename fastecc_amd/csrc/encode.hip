// encode.hip — the device drivers behind the entry points: runs of passes over a stripe, the encode of every code family, the stand-alone
// transform, the decoder's transform helpers (gathered first pass, split transform), ordering of the context's internal buffers.
// Split from api.hip in round 6 (no change of behaviour); declarations in drivers.hpp / internal.hpp.
#include "drivers.hpp"

using namespace fastecc;

namespace fastecc {

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
               const uint32_t* tw_dit, hipStream_t st, uint32_t col0, uint32_t width, hipEvent_t first_done,
               uint32_t batch, const CallBounds& cb)
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
bool direct_encode_applies(const fastecc_ctx* c, const void* data, const void* parity)
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
