// host_stage.hip — stripes in host memory: the pinned-memory column-slab pipeline and the staging rings between pageable memory and the
// copy engine (what RS.cpp:25-38 measures end to end).  Split from api.hip in round 6 (no change of behaviour).
#include "drivers.hpp"

using namespace fastecc;

namespace fastecc {


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

int stage_transfer(fastecc_ctx* c, const StageJob& j, hipStream_t st, int threads)
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

}  // namespace fastecc
