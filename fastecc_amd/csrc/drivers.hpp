// drivers.hpp — what the translation units behind the C ABI share beyond context.hpp: the encode drivers (encode.hip), the host staging
// paths (host_stage.hip) and context creation (create.hip), called from the entry points in api.hip.  Nothing here is part of the ABI.
#pragma once
#include <condition_variable>
#include <thread>

#include "context.hpp"

namespace fastecc {

// widest lane vector that the block size and the pointers allow
inline int pick_vec(const fastecc_ctx* c, const void* a, const void* b)
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

// ---- encode.hip: the device drivers ----
int run_passes(fastecc_ctx* c, const std::vector<Pass>& plan, const uint32_t* in, uint32_t* out, const uint32_t* tw_dif,
               const uint32_t* tw_dit, hipStream_t st, uint32_t col0 = 0, uint32_t width = 0, hipEvent_t first_done = nullptr,
               uint32_t batch = 1, const CallBounds& cb = CallBounds());
int ensure_slab_streams(fastecc_ctx* c);
bool plan_is_all_tiles(const std::vector<Pass>& plan);
int encode_pow2(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st, const CallBounds& cb = CallBounds());
int encode_mixed(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st);
bool direct_encode_applies(const fastecc_ctx* c, const void* data = nullptr, const void* parity = nullptr);
int encode_device(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st);
int ntt_device(fastecc_ctx* c, uint32_t* data, bool inverse, hipStream_t st);
int order_internal_buffers(fastecc_ctx* c, hipStream_t st);
int mark_internal_buffers(fastecc_ctx* c, hipStream_t st);
// Runs `body` (which enqueues work on `st` that uses internal buffers) between the two.
template <class F> int with_internal_buffers(fastecc_ctx* c, hipStream_t st, F body)
{
    int rc = order_internal_buffers(c, st);
    if (rc != FASTECC_OK) return rc;
    rc = body();
    const int rc2 = mark_internal_buffers(c, st);  // also after a failure: part of the work may have been enqueued
    return rc != FASTECC_OK ? rc : rc2;
}
// ---- host_stage.hip: stripes in host memory ----
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
int stage_transfer(fastecc_ctx* c, const StageJob& j, hipStream_t st, int threads = 0);
int stage_download(fastecc_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st);
int ensure_dbuf(fastecc_ctx* c);
int encode_host_pinned(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st);
int encode_host_pageable(fastecc_ctx* c, const uint32_t* data, uint32_t* parity, hipStream_t st);

// ---- create.hip ----
int create_impl(fastecc_ctx** out, uint64_t n, uint64_t k, int lg, uint64_t block_bytes, int field, int device, int fold, int cosets,
                const uint32_t* custom_factor);
extern const uint32_t* const NTT_ONLY;  // custom_factor value: see create_ntt_ctx
int setup_mixed(fastecc_ctx* c, int q, uint64_t k_user, uint64_t m_user, const uint32_t* custom_factor = nullptr);

}  // namespace fastecc
