// api.hip — the C ABI of libfastecc_hip.so (include/fastecc.h): the entry points that move data.  The drivers behind them: encode.hip (device
// paths), host_stage.hip (stripes in host memory), create.hip (contexts); shared declarations in drivers.hpp.
//
// Host side of the encode path.  It replaces the body of EncodeReedSolomon (RS.cpp:22-68) and the drivers MFA_NTT / Rec_NTT
// (ntt.cpp:349-447).  The pass plans and twiddle tables live in plan.hip, options and profiling in options.hip, the shared context in
// context.hpp.  No CPU compute fallback exists: every entry point that moves data runs HIP kernels or fails.
#include <condition_variable>
#include <thread>

#include "drivers.hpp"

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

}  // extern "C"

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
