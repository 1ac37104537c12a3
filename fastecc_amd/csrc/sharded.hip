// sharded.hip — one stripe on several GPUs: fastecc_create_sharded / fastecc_encode_sharded (include/fastecc.h).
//
// The reference has no multi-device code (SURVEY.md §2 "Distributed comm backend: none"); BASELINE.json configs[3]
// defines the mode: the blocks of ONE (n,k) stripe sharded over the GPUs of a node "with a final gather over xGMI".
// The transform runs down the block index and the word columns of a stripe never meet (ntt.cpp:348-350 loops over
// them independently), so the shard is a COLUMN SLAB: GPU g holds words [g*S/G, (g+1)*S/G) of every block and encodes
// them with an ordinary per-device context for block_bytes/G — no collective inside the transform, tables replicated.
// What is left is data movement, and that is all this file does:
//
//   upload    (only when the caller hands over a full stripe)  strided copy  stripe[:, slab g] -> GPU g's slab buffer
//   compute   the device context's passes on the slab, per column sub-slab
//   download  strided copy  GPU g's parity slab -> parity[:, slab g]   (the xGMI gather, or each GPU's own host link)
//
// Every GPU runs the three on its own streams, sub-slab h+1 computing while sub-slab h is being copied out; the root
// only waits.  Copies are issued by the GPU that owns the slab (it needs peer access to the root's memory, not the
// other way round), either on the copy engines (hipMemcpy2DAsync) or as a kernel that stores straight through the
// peer mapping ("gather_mode" 2).  All of it is wave-agnostic plumbing: no arithmetic happens here.
//
// A gather to one root is bound by the root's links ((G-1)/G of the stripe enters ONE GPU: 1.88 GB over 7 links at G = 8, four times
// the per-GPU compute).  fastecc_encode_sharded_blocks is the form that can scale: the result stays BLOCK-DISTRIBUTED — GPU g ends with
// parity blocks [g*M/G, (g+1)*M/G) whole (RS.md:13-33: the N data and M parity blocks are separately stored units) — so the exchange is an
// all-to-all in which every GPU sends 1/G of its slab to every peer: 1/G^2 of the stripe per link and direction, no hot spot.  The mirror
// transpose in front of the encode takes block-distributed DATA (GPU g holds data blocks [g*k/G, (g+1)*k/G) whole) to column slabs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "internal.hpp"

namespace fastecc {

namespace {

constexpr int MAX_SUB = 8;

struct Shard {
    int device = 0;
    fastecc_ctx* ctx = nullptr;  // (n, k, block_bytes / G) on `device`
    hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
    hipEvent_t ev_up[MAX_SUB] = {}, ev_comp[MAX_SUB] = {};
    hipEvent_t ev_all = nullptr;  // everything this shard did for the last call
    bool used = false;
    char* data_slab = nullptr;    // lazy: [k][slab_bytes], for callers that hand over full stripes
    char* parity_slab = nullptr;  // lazy: [n-k][slab_bytes], when the caller keeps no parity slabs
    std::vector<hipStream_t> s_peer;              // lazy, all-to-all on the copy engines: one stream per destination AND direction of the pipeline
                                                  // ([lane * G + d]; lane 0 = the data push in front, 1 = the parity transpose behind), so the G links
                                                  // work side by side and the push of sub-slab h + 1 is not queued behind the wait for compute h
    std::vector<hipEvent_t> ev_peer;
};

int fail(const char* what, hipError_t e)
{
    set_error_detail(what, e);
    return e == hipErrorOutOfMemory ? FASTECC_E_NOMEM : FASTECC_E_DEVICE;
}

#define SH_TRY(expr)                                  \
    do {                                              \
        hipError_t e_ = (expr);                       \
        if (e_ != hipSuccess) return fail(#expr, e_); \
    } while (0)

struct DeviceSwitch {
    int prev = -1;
    DeviceSwitch()
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceSwitch()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// dst[r][0..width) = src[r][0..width) for r < rows; pitches and width in units of T.  One thread per T, consecutive
// threads on consecutive addresses of a row segment (256-byte segments are 16 lanes of uint4).
template <typename T>
__global__ __launch_bounds__(256) void copy_window_kernel(const T* __restrict__ src, T* __restrict__ dst, uint32_t width, uint64_t spitch,
                                                          uint64_t dpitch, uint64_t total)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / width;
        const uint32_t c = (uint32_t)(i - r * width);
        dst[r * dpitch + c] = src[r * spitch + c];
    }
}

// One launch of an all-to-all: piece d (rows x width) of this GPU goes to destination d,
//     dst.p[d][r * dpitch + c] = src[d * sstride + r * spitch + c],
// stored through the peer mappings — the G - 1 remote destinations sit behind G - 1 different xGMI links, which therefore all carry
// traffic at once; consecutive threads = consecutive addresses of a row segment on both sides.
struct PeerPtrs {
    void* p[64];
};
template <typename T>
__global__ __launch_bounds__(256) void all_to_all_kernel(const T* __restrict__ src, PeerPtrs dst, uint32_t width, uint32_t rows, uint64_t spitch,
                                                         uint64_t sstride, uint64_t dpitch, uint64_t total)
{
    const uint64_t piece = (uint64_t)rows * width;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t d = (uint32_t)(i / piece);
        const uint64_t rem = i - d * piece;
        const uint64_t r = rem / width;
        const uint32_t c = (uint32_t)(rem - r * width);
        ((T*)dst.p[d])[r * dpitch + c] = src[d * sstride + r * spitch + c];
    }
}

}  // namespace

struct Sharded {
    std::vector<Shard> shards;
    int root = 0;
    int field = 0;
    uint64_t K = 0, M = 0;  // data / parity blocks of the caller's stripes
    uint64_t block_bytes = 0, slab_bytes = 0;
    int sub_slabs = 2;
    int gather_mode = 1;  // 1 = hipMemcpy2DAsync (copy engines), 2 = copy kernel on the slab's GPU
    bool copy_engine_refused = false;  // mode 2 was forced by an error return of hipMemcpy2DAsync
    int all_pairs = 0;   // fastecc_encode_sharded_blocks: 0 = not tried, 1 = every slab device can access every other one, -1 = it cannot
    hipEvent_t fork = nullptr;
    int stage_threads = 0;  // option "stage_threads" as last set through this context (0 = automatic)
    int fault_at = 0;    // tests only (option "inject_fault"): the fault_at-th slab step of the next call fails after its work has been enqueued
    int fault_step = 0;
    std::string text;
};

namespace {

void describe(fastecc_ctx* shell)
{
    Sharded* s = sharded_of(shell);
    char buf[128];
    snprintf(buf, sizeof buf, "%d slabs x %llu B/block, sub_slabs=%d, gather=%s | ", (int)s->shards.size(), (unsigned long long)s->slab_bytes,
             s->sub_slabs, s->gather_mode == 2 ? (s->copy_engine_refused ? "kernel (copy engine refused)" : "kernel") : "copy-engine");
    set_plan_text(shell, std::string(buf) + fastecc_plan_string(s->shards[0].ctx));
}

// A [rows][width bytes] window between two pitched buffers, on `st` of the current device.  Between device buffers the
// copy engines are the default; a runtime that refuses a pitched peer copy outright switches the context to the copy
// kernel for good (recorded in the plan string), so that the first multi-GPU box decides, not this file.
int copy_window(Sharded* s, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, bool device_both,
                hipStream_t st)
{
    if (rows == 0 || width == 0) return FASTECC_OK;
    if (s->gather_mode != 2 || !device_both) {
        const hipError_t e = hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDefault, st);
        if (e == hipSuccess) return FASTECC_OK;
        if (!device_both) return fail("hipMemcpy2DAsync", e);
        (void)hipGetLastError();
        s->gather_mode = 2;
        s->copy_engine_refused = true;
    }
    const bool v16 = ((width | dpitch | spitch | (uintptr_t)dst | (uintptr_t)src) & 15u) == 0;
    const size_t unit = v16 ? 16 : 4;
    const uint64_t total = (uint64_t)rows * (width / unit);
    const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 4096);
    if (v16)
        hipLaunchKernelGGL(copy_window_kernel<uint4>, dim3(blocks), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, (uint32_t)(width / 16),
                           (uint64_t)(spitch / 16), (uint64_t)(dpitch / 16), total);
    else
        hipLaunchKernelGGL(copy_window_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, (const uint32_t*)src, (uint32_t*)dst,
                           (uint32_t)(width / 4), (uint64_t)(spitch / 4), (uint64_t)(dpitch / 4), total);
    SH_TRY(hipGetLastError());
    return FASTECC_OK;
}

// An entry point that fails half way has already forked work onto some slabs' streams.  Before the error is returned that work is waited
// for (host side), so that neither the caller's buffers nor the slab buffers are still being read or written when the caller reacts, and
// the next call does not race with orphaned kernels and copies.  Errors of the wait itself are ignored: the first error is reported.
void settle_after_failure(Sharded* s)
{
    for (Shard& sh : s->shards) {
        if (hipSetDevice(sh.device) != hipSuccess) continue;
        for (hipStream_t q : {sh.s_up, sh.s_comp, sh.s_down})
            if (q) (void)hipStreamSynchronize(q);
        for (hipStream_t q : sh.s_peer)
            if (q) (void)hipStreamSynchronize(q);
        sh.used = false;  // nothing of this shard is in flight any more
    }
    (void)hipGetLastError();
}
// tests only: the "inject_fault"-th slab step of a call reports a device error AFTER its copies / kernels have been enqueued
bool injected_fault(Sharded* s)
{
    if (s->fault_at <= 0) return false;
    if (++s->fault_step != s->fault_at) return false;
    s->fault_at = 0;
    s->fault_step = 0;
    set_error_detail("injected fault (option inject_fault)", hipErrorUnknown);
    return true;
}

// The whole data path.  Exactly one of (data_slabs, data_stripe) and at least one of (parity_slabs, parity_stripe).
// stripe_on_host: the stripes are host memory (each GPU uses its own host link), else the root device's memory.
int run_body(fastecc_ctx* shell, const void* const* data_slabs, const void* data_stripe, void* const* parity_slabs, void* parity_stripe,
             bool stripe_on_host, hipStream_t st)
{
    Sharded* s = sharded_of(shell);
    const int G = (int)s->shards.size();
    const size_t slab = s->slab_bytes, full = s->block_bytes;
    DeviceSwitch restore;

    // sub-slabs only where the device context can encode a column range
    int H = std::max(1, std::min(s->sub_slabs, MAX_SUB));
    const uint64_t slab_words = slab / 4;
    while (H > 1 && (!columns_supported(s->shards[0].ctx) || (slab_words % (32u * H)) != 0)) H >>= 1;
    const size_t wbytes = slab / H;

    SH_TRY(hipSetDevice(s->root));
    SH_TRY(hipEventRecord(s->fork, st));

    for (int g = 0; g < G; g++) {
        Shard& sh = s->shards[g];
        SH_TRY(hipSetDevice(sh.device));
        const char* din = data_slabs ? (const char*)data_slabs[g] : nullptr;
        if (!din) {
            if (!sh.data_slab) SH_TRY(hipMalloc((void**)&sh.data_slab, s->K * slab));
            din = sh.data_slab;
        }
        char* pout = parity_slabs ? (char*)parity_slabs[g] : nullptr;
        if (!pout) {
            if (!sh.parity_slab) SH_TRY(hipMalloc((void**)&sh.parity_slab, s->M * slab));
            pout = sh.parity_slab;
        }
        // this call starts after prior work on the caller's stream and after the shard's previous call
        for (hipStream_t q : {sh.s_up, sh.s_comp}) {
            SH_TRY(hipStreamWaitEvent(q, s->fork, 0));
            if (sh.used) SH_TRY(hipStreamWaitEvent(q, sh.ev_all, 0));
        }
        for (int h = 0; h < H; h++) {
            const size_t col = (size_t)h * wbytes;
            if (data_stripe) {
                const int rc = copy_window(s, sh.data_slab + col, slab, (const char*)data_stripe + (size_t)g * slab + col, full, wbytes, s->K,
                                           !stripe_on_host, sh.s_up);
                if (rc != FASTECC_OK) return rc;
                SH_TRY(hipEventRecord(sh.ev_up[h], sh.s_up));
                SH_TRY(hipStreamWaitEvent(sh.s_comp, sh.ev_up[h], 0));
            }
            const int rc = H > 1 ? fastecc_encode_columns(sh.ctx, din, pout, col / 4, wbytes / 4, sh.s_comp)
                                 : fastecc_encode(sh.ctx, din, pout, FASTECC_MEM_DEVICE, sh.s_comp);
            if (rc != FASTECC_OK) return rc;
            if (injected_fault(s)) return FASTECC_E_DEVICE;
            SH_TRY(hipSetDevice(sh.device));  // the entry points restore the caller's device; keep ours explicit
            SH_TRY(hipEventRecord(sh.ev_comp[h], sh.s_comp));
            SH_TRY(hipStreamWaitEvent(sh.s_down, sh.ev_comp[h], 0));
            if (parity_stripe) {
                const int rc2 = copy_window(s, (char*)parity_stripe + (size_t)g * slab + col, full, pout + col, slab, wbytes, s->M,
                                            !stripe_on_host, sh.s_down);
                if (rc2 != FASTECC_OK) return rc2;
            }
        }
        SH_TRY(hipEventRecord(sh.ev_all, sh.s_down));
        sh.used = true;
    }
    SH_TRY(hipSetDevice(s->root));
    for (int g = 0; g < G; g++) SH_TRY(hipStreamWaitEvent(st, s->shards[g].ev_all, 0));
    if (s->copy_engine_refused) describe(shell);
    return FASTECC_OK;
}

// Stripes in PAGEABLE host memory (fastecc_encode with FASTECC_MEM_HOST on a sharded context): every slab has a host thread of its own that
// moves its columns through the slab context's rings of pinned slots (host_stage.hip stage_transfer: helper threads gather / scatter the rows, the
// copy engine of that GPU moves the slots over that GPU's host link), encodes, and brings the parity home — all slabs side by side.  The
// runtime's own pageable copies are synchronous and ran the slabs one after the other (8 slabs on one device: 330 ms for 2 + 2 GiB).
// Synchronous; the caller's stream has been waited for.
int run_host_pageable(fastecc_ctx* shell, const void* data_stripe, void* parity_stripe, hipStream_t st)
{
    Sharded* s = sharded_of(shell);
    const int G = (int)s->shards.size();
    const size_t slab = s->slab_bytes, full = s->block_bytes;
    DeviceSwitch restore;
    SH_TRY(hipSetDevice(s->root));
    SH_TRY(hipStreamSynchronize(st));
    for (int g = 0; g < G; g++) {  // allocations and the previous call's tail, on this thread
        Shard& sh = s->shards[g];
        SH_TRY(hipSetDevice(sh.device));
        if (!sh.data_slab) SH_TRY(hipMalloc((void**)&sh.data_slab, s->K * slab));
        if (!sh.parity_slab) SH_TRY(hipMalloc((void**)&sh.parity_slab, s->M * slab));
        if (sh.used) SH_TRY(hipEventSynchronize(sh.ev_all));
    }
    const unsigned hw = std::thread::hardware_concurrency();
    // helper threads per slab: the "stage_threads" option of the slab contexts where it is set, else a share of the host's threads
    const int set_by_option = s->stage_threads;
    const int T = set_by_option > 0 ? set_by_option : (int)std::min<unsigned>(4u, std::max<unsigned>(1u, hw / (2u * (unsigned)G)));
    std::vector<int> rcs((size_t)G, FASTECC_OK);
    std::vector<std::string> texts((size_t)G);
    std::mutex fault_mu;
    auto slab_job = [&](int g) {
        Shard& sh = s->shards[g];
        int rc = FASTECC_OK;
        try {
            if (hipSetDevice(sh.device) != hipSuccess) rc = FASTECC_E_DEVICE;
            if (rc == FASTECC_OK) rc = stage_rect(sh.ctx, true, const_cast<char*>((const char*)data_stripe) + (size_t)g * slab, full, sh.data_slab, slab, slab, s->K, sh.s_up, T);
            if (rc == FASTECC_OK) rc = fastecc_encode(sh.ctx, sh.data_slab, sh.parity_slab, FASTECC_MEM_DEVICE, sh.s_up);
            if (rc == FASTECC_OK) {
                std::lock_guard<std::mutex> lk(fault_mu);
                if (injected_fault(s)) rc = FASTECC_E_DEVICE;
            }
            if (rc == FASTECC_OK && hipSetDevice(sh.device) != hipSuccess) rc = FASTECC_E_DEVICE;
            if (rc == FASTECC_OK) rc = stage_rect(sh.ctx, false, (char*)parity_stripe + (size_t)g * slab, full, sh.parity_slab, slab, slab, s->M, sh.s_up, T);
        } catch (...) {
            rc = FASTECC_E_NOMEM;
        }
        if (rc != FASTECC_OK) texts[(size_t)g] = fastecc_last_error_detail();
        rcs[(size_t)g] = rc;
    };
    std::vector<std::thread> pool;
    try {
        for (int g = 1; g < G; g++) pool.emplace_back(slab_job, g);
    } catch (...) {
        for (std::thread& th : pool) th.join();
        return FASTECC_E_NOMEM;
    }
    slab_job(0);
    for (std::thread& th : pool) th.join();
    for (int g = 0; g < G; g++) {  // settle every slab's stream on every path; the slabs are free for the next call
        Shard& sh = s->shards[g];
        if (hipSetDevice(sh.device) == hipSuccess) (void)hipStreamSynchronize(sh.s_up);
        sh.used = false;
    }
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g] != FASTECC_OK) {
            set_error_text(texts[(size_t)g].c_str());
            return rcs[(size_t)g];
        }
    return FASTECC_OK;
}

int run(fastecc_ctx* shell, const void* const* data_slabs, const void* data_stripe, void* const* parity_slabs, void* parity_stripe, bool stripe_on_host,
        hipStream_t st)
{
    DeviceSwitch restore;
    const int rc = run_body(shell, data_slabs, data_stripe, parity_slabs, parity_stripe, stripe_on_host, st);
    if (rc != FASTECC_OK) settle_after_failure(sharded_of(shell));
    return rc;
}

// every slab device must reach every other one's memory (the gather only needs slab device -> root)
int enable_all_pairs(Sharded* s)
{
    if (s->all_pairs) return s->all_pairs > 0 ? FASTECC_OK : FASTECC_E_UNSUPPORTED;
    for (Shard& a : s->shards)
        for (Shard& b : s->shards) {
            if (a.device == b.device) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a.device, b.device) != hipSuccess || !can) {
                set_error_detail("hipDeviceCanAccessPeer(slab device -> slab device)", hipErrorPeerAccessUnsupported);
                s->all_pairs = -1;
                return FASTECC_E_UNSUPPORTED;
            }
            SH_TRY(hipSetDevice(a.device));
            const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail("hipDeviceEnablePeerAccess", e);
            (void)hipGetLastError();
        }
    s->all_pairs = 1;
    return FASTECC_OK;
}

// piece d = rows x width bytes at src + d * sstride (row pitch spitch) -> dst[d] (row pitch dpitch), all G pieces of the current device.
// gather_mode 2: ONE kernel storing through the peer mappings on `st`.  gather_mode 1: G pitched copies on the copy engines, each on the
// stream of its destination (after `after`, joined into `st` again), so the transfers to different peers run side by side.
int all_to_all_step(Sharded* s, Shard& sh, void* const* dst, size_t dpitch, const char* src, size_t spitch, size_t sstride, size_t width, size_t rows,
                    hipStream_t st, hipEvent_t after, int lane)
{
    const int G = (int)s->shards.size();
    if (rows == 0 || width == 0) return FASTECC_OK;
    if (s->gather_mode != 2) {
        if (sh.s_peer.empty()) {
            sh.s_peer.assign(2 * (size_t)G, nullptr);
            sh.ev_peer.assign(2 * (size_t)G, nullptr);
            for (int d = 0; d < 2 * G; d++) {
                SH_TRY(hipStreamCreateWithFlags(&sh.s_peer[d], hipStreamNonBlocking));
                SH_TRY(hipEventCreateWithFlags(&sh.ev_peer[d], hipEventDisableTiming));
            }
        }
        hipStream_t* const peer = sh.s_peer.data() + (size_t)(lane ? G : 0);
        hipEvent_t* const pev = sh.ev_peer.data() + (size_t)(lane ? G : 0);
        bool refused = false;
        for (int d = 0; d < G && !refused; d++) {
            SH_TRY(hipStreamWaitEvent(peer[d], after, 0));
            const hipError_t e = hipMemcpy2DAsync(dst[d], dpitch, src + (size_t)d * sstride, spitch, width, rows, hipMemcpyDefault, peer[d]);
            if (e != hipSuccess) {  // a runtime that refuses pitched peer copies: the copy kernel takes over for good (as in copy_window)
                (void)hipGetLastError();
                s->gather_mode = 2;
                s->copy_engine_refused = true;
                refused = true;
                for (int f = 0; f < d; f++) SH_TRY(hipStreamSynchronize(peer[f]));  // what was enqueued is simply repeated below
                break;
            }
            SH_TRY(hipEventRecord(pev[d], peer[d]));
            SH_TRY(hipStreamWaitEvent(st, pev[d], 0));
        }
        if (!refused) return FASTECC_OK;
    }
    PeerPtrs pp;
    uintptr_t align = width | dpitch | spitch | sstride | (uintptr_t)src;
    for (int d = 0; d < G; d++) {
        pp.p[d] = dst[d];
        align |= (uintptr_t)dst[d];
    }
    const bool v16 = (align & 15u) == 0;
    const size_t unit = v16 ? 16 : 4;
    const uint64_t total = (uint64_t)G * rows * (width / unit);
    const unsigned blocks = (unsigned)std::min<uint64_t>((total + 255) / 256, 8192);
    if (v16)
        hipLaunchKernelGGL(all_to_all_kernel<uint4>, dim3(blocks), dim3(256), 0, st, (const uint4*)src, pp, (uint32_t)(width / 16), (uint32_t)rows,
                           (uint64_t)(spitch / 16), (uint64_t)(sstride / 16), (uint64_t)(dpitch / 16), total);
    else
        hipLaunchKernelGGL(all_to_all_kernel<uint32_t>, dim3(blocks), dim3(256), 0, st, (const uint32_t*)src, pp, (uint32_t)(width / 4), (uint32_t)rows,
                           (uint64_t)(spitch / 4), (uint64_t)(sstride / 4), (uint64_t)(dpitch / 4), total);
    SH_TRY(hipGetLastError());
    return FASTECC_OK;
}

// fastecc_encode_sharded_blocks.  data_blocks: GPU g holds data blocks [g*K/G, (g+1)*K/G) whole, else data[g] is slab g.
// Per column sub-slab h (every GPU, its own streams):
//   s_up    (block-distributed data) push columns [d*slab + h*wb, +wb) of my K/G blocks into rows [g*K/G, ...) of GPU d's data slab, all d
//   s_comp  after the pushes of ALL GPUs for sub-slab h: encode the sub-slab
//   s_down  rows [d*M/G, (d+1)*M/G) of my parity sub-slab into columns [g*slab + h*wb, +wb) of GPU d's whole parity blocks, all d
// so sub-slab h+1 is transposed in and sub-slab h-1 transposed out while sub-slab h is in the kernels.
int run_blocks_body(fastecc_ctx* shell, const void* const* data, bool data_blocks, void* const* parity_blocks, hipStream_t st)
{
    Sharded* s = sharded_of(shell);
    const int G = (int)s->shards.size();
    const size_t slab = s->slab_bytes, full = s->block_bytes;
    const uint64_t kg = s->K / G, mg = s->M / G;
    DeviceSwitch restore;
    {
        const int rc = enable_all_pairs(s);
        if (rc != FASTECC_OK) return rc;
    }
    int H = std::max(1, std::min(s->sub_slabs, MAX_SUB));
    const uint64_t slab_words = slab / 4;
    while (H > 1 && (!columns_supported(s->shards[0].ctx) || (slab_words % (32u * H)) != 0)) H >>= 1;
    const size_t wbytes = slab / H;

    SH_TRY(hipSetDevice(s->root));
    SH_TRY(hipEventRecord(s->fork, st));
    // buffers, and the start of this call on every stream: after prior work on the caller's stream and after EVERY shard's previous call
    // (the pushes write into other shards' slab buffers, the transposes out into other shards' result blocks)
    for (int g = 0; g < G; g++) {
        Shard& sh = s->shards[g];
        SH_TRY(hipSetDevice(sh.device));
        if (data_blocks && !sh.data_slab) SH_TRY(hipMalloc((void**)&sh.data_slab, s->K * slab));
        if (!sh.parity_slab) SH_TRY(hipMalloc((void**)&sh.parity_slab, s->M * slab));
        for (hipStream_t q : {sh.s_up, sh.s_comp, sh.s_down}) {
            SH_TRY(hipStreamWaitEvent(q, s->fork, 0));
            for (int o = 0; o < G; o++)
                if (s->shards[o].used) SH_TRY(hipStreamWaitEvent(q, s->shards[o].ev_all, 0));
        }
    }
    std::vector<void*> dst(G);
    for (int h = 0; h < H; h++) {
        const size_t col = (size_t)h * wbytes;
        if (data_blocks) {
            for (int g = 0; g < G; g++) {  // GPU g pushes sub-slab h of its blocks to everyone
                Shard& sh = s->shards[g];
                SH_TRY(hipSetDevice(sh.device));
                for (int d = 0; d < G; d++) dst[d] = s->shards[d].data_slab + (size_t)g * kg * slab + col;
                SH_TRY(hipEventRecord(sh.ev_up[h], sh.s_up));  // the point the copy-engine form forks from; recorded again after the step
                const int rc = all_to_all_step(s, sh, dst.data(), slab, (const char*)data[g] + col, full, slab, wbytes, kg, sh.s_up, sh.ev_up[h], 0);
                if (rc != FASTECC_OK) return rc;
                SH_TRY(hipEventRecord(sh.ev_up[h], sh.s_up));
            }
        }
        for (int g = 0; g < G; g++) {
            Shard& sh = s->shards[g];
            SH_TRY(hipSetDevice(sh.device));
            if (data_blocks)
                for (int o = 0; o < G; o++) SH_TRY(hipStreamWaitEvent(sh.s_comp, s->shards[o].ev_up[h], 0));
            const char* din = data_blocks ? sh.data_slab : (const char*)data[g];
            const int rc = H > 1 ? fastecc_encode_columns(sh.ctx, din, sh.parity_slab, col / 4, wbytes / 4, sh.s_comp)
                                 : fastecc_encode(sh.ctx, din, sh.parity_slab, FASTECC_MEM_DEVICE, sh.s_comp);
            if (rc != FASTECC_OK) return rc;
            if (injected_fault(s)) return FASTECC_E_DEVICE;
            SH_TRY(hipSetDevice(sh.device));
            SH_TRY(hipEventRecord(sh.ev_comp[h], sh.s_comp));
            SH_TRY(hipStreamWaitEvent(sh.s_down, sh.ev_comp[h], 0));
            for (int d = 0; d < G; d++) dst[d] = (char*)parity_blocks[d] + (size_t)g * slab + col;
            const int rc2 = all_to_all_step(s, sh, dst.data(), full, sh.parity_slab + col, slab, (size_t)mg * slab, wbytes, mg, sh.s_down, sh.ev_comp[h], 1);
            if (rc2 != FASTECC_OK) return rc2;
        }
    }
    for (int g = 0; g < G; g++) {
        Shard& sh = s->shards[g];
        SH_TRY(hipSetDevice(sh.device));
        if (data_blocks) {  // the slab buffers are free again once the last push AND the last read of them are done
            SH_TRY(hipEventRecord(sh.ev_up[0], sh.s_up));
            SH_TRY(hipStreamWaitEvent(sh.s_down, sh.ev_up[0], 0));
        }
        SH_TRY(hipEventRecord(sh.ev_all, sh.s_down));
        sh.used = true;
    }
    SH_TRY(hipSetDevice(s->root));
    for (int g = 0; g < G; g++) SH_TRY(hipStreamWaitEvent(st, s->shards[g].ev_all, 0));
    if (s->copy_engine_refused) describe(shell);
    return FASTECC_OK;
}

// Decoding on a sharded context: erasures hit whole blocks, so every slab sees the same pattern and repairs its own columns
// with its device context's decoder.  Full stripes (root memory or host): slab g of data and parity is pulled to GPU g,
// repaired in place, and the data slab (with `repair`: the parity slab too) pushed back; only the columns move, no exchange
// between slabs.  data_slabs / parity_slabs given: the slabs are repaired where they live.
int run_decode_body(fastecc_ctx* shell, void* data_stripe, void* parity_stripe, void* const* data_slabs, void* const* parity_slabs, bool stripe_on_host,
                    bool repair, hipStream_t st)
{
    Sharded* s = sharded_of(shell);
    const int G = (int)s->shards.size();
    const size_t slab = s->slab_bytes, full = s->block_bytes;
    DeviceSwitch restore;
    SH_TRY(hipSetDevice(s->root));
    SH_TRY(hipEventRecord(s->fork, st));
    for (int g = 0; g < G; g++) {
        Shard& sh = s->shards[g];
        SH_TRY(hipSetDevice(sh.device));
        char* d = data_slabs ? (char*)data_slabs[g] : nullptr;
        char* p = parity_slabs ? (char*)parity_slabs[g] : nullptr;
        if (!d) {
            if (!sh.data_slab) SH_TRY(hipMalloc((void**)&sh.data_slab, s->K * slab));
            if (!sh.parity_slab) SH_TRY(hipMalloc((void**)&sh.parity_slab, s->M * slab));
            d = sh.data_slab;
            p = sh.parity_slab;
        }
        hipStream_t q = sh.s_comp;  // one stream per slab: pull, repair, push
        SH_TRY(hipStreamWaitEvent(q, s->fork, 0));
        if (sh.used) SH_TRY(hipStreamWaitEvent(q, sh.ev_all, 0));
        if (!data_slabs) {
            int rc = copy_window(s, d, slab, (const char*)data_stripe + (size_t)g * slab, full, slab, s->K, !stripe_on_host, q);
            if (rc == FASTECC_OK) rc = copy_window(s, p, slab, (const char*)parity_stripe + (size_t)g * slab, full, slab, s->M, !stripe_on_host, q);
            if (rc != FASTECC_OK) return rc;
        }
        const int rc = repair ? fastecc_repair(sh.ctx, d, p, FASTECC_MEM_DEVICE, q) : fastecc_decode(sh.ctx, d, p, FASTECC_MEM_DEVICE, q);
        if (rc != FASTECC_OK) return rc;
        if (injected_fault(s)) return FASTECC_E_DEVICE;
        SH_TRY(hipSetDevice(sh.device));
        if (!data_slabs) {
            int rc2 = copy_window(s, (char*)data_stripe + (size_t)g * slab, full, d, slab, slab, s->K, !stripe_on_host, q);
            if (rc2 == FASTECC_OK && repair) rc2 = copy_window(s, (char*)parity_stripe + (size_t)g * slab, full, p, slab, slab, s->M, !stripe_on_host, q);
            if (rc2 != FASTECC_OK) return rc2;
        }
        SH_TRY(hipEventRecord(sh.ev_all, q));
        sh.used = true;
    }
    SH_TRY(hipSetDevice(s->root));
    for (int g = 0; g < G; g++) SH_TRY(hipStreamWaitEvent(st, s->shards[g].ev_all, 0));
    if (s->copy_engine_refused) describe(shell);
    return FASTECC_OK;
}

int run_decode(fastecc_ctx* shell, void* data_stripe, void* parity_stripe, void* const* data_slabs, void* const* parity_slabs, bool stripe_on_host, bool repair,
               hipStream_t st)
{
    DeviceSwitch restore;
    const int rc = run_decode_body(shell, data_stripe, parity_stripe, data_slabs, parity_slabs, stripe_on_host, repair, st);
    if (rc != FASTECC_OK) settle_after_failure(sharded_of(shell));
    return rc;
}

}  // namespace

int sharded_decode_prepare(fastecc_ctx* shell, const uint8_t* data_present, const uint8_t* parity_present)
{
    Sharded* s = sharded_of(shell);
    std::lock_guard<std::mutex> lk(mutex_of(shell));
    // the same pattern for every slab (each context keeps its own tables on its own device): one host thread per slab, so the set-ups — a host
    // scan and a few dozen small kernels each, synchronous — run side by side instead of n_slabs times 2.3 ms one after the other.
    // The error detail is thread-local and a worker must not throw: each worker maps exceptions to a code and hands its detail text back,
    // and the first failing slab's text is republished on the calling thread.
    const size_t G = s->shards.size();
    std::vector<int> rcs(G, FASTECC_OK);
    std::vector<std::string> details(G);
    auto one = [&](size_t i) {
        try {
            rcs[i] = fastecc_decode_prepare(s->shards[i].ctx, data_present, parity_present);
            if (rcs[i] != FASTECC_OK) details[i] = fastecc_last_error_detail();
        } catch (const std::bad_alloc&) {
            rcs[i] = FASTECC_E_NOMEM;
        } catch (...) {
            rcs[i] = FASTECC_E_DEVICE;
        }
    };
    std::vector<std::thread> workers;
    std::vector<char> started(G, 0);
    for (size_t i = 1; i < G; i++) {
        try {
            workers.emplace_back(one, i);
            started[i] = 1;
        } catch (...) {  // no thread to be had: that slab is set up on this one below (no exception crosses the ABI)
        }
    }
    one(0);
    for (std::thread& t : workers) t.join();
    for (size_t i = 1; i < G; i++)
        if (!started[i]) one(i);
    for (size_t i = 0; i < G; i++)
        if (rcs[i] != FASTECC_OK) {
            set_error_text(details[i].c_str());
            return rcs[i];
        }
    return FASTECC_OK;
}

int sharded_decode_stripe(fastecc_ctx* shell, void* data, void* parity, int mem_kind, bool repair, hipStream_t st)
{
    if (mem_kind != FASTECC_MEM_DEVICE && mem_kind != FASTECC_MEM_HOST && mem_kind != FASTECC_MEM_HOST_PINNED) return FASTECC_E_INVAL;
    std::lock_guard<std::mutex> lk(mutex_of(shell));
    const int rc = run_decode(shell, data, parity, nullptr, nullptr, mem_kind != FASTECC_MEM_DEVICE, repair, st);
    // host stripes, pageable or pinned: synchronous, as include/fastecc.h says for fastecc_decode / fastecc_repair (the single-device path
    // stages both kinds the same way)
    if (rc != FASTECC_OK || mem_kind == FASTECC_MEM_DEVICE) return rc;
    DeviceSwitch restore;
    SH_TRY(hipSetDevice(sharded_of(shell)->root));
    SH_TRY(hipStreamSynchronize(st));
    return FASTECC_OK;
}

void destroy_sharded(Sharded* s)
{
    if (!s) return;
    DeviceSwitch restore;
    for (Shard& sh : s->shards) {
        if (hipSetDevice(sh.device) != hipSuccess) continue;
        for (hipStream_t q : {sh.s_up, sh.s_comp, sh.s_down})
            if (q) (void)hipStreamSynchronize(q);
        if (sh.ctx) fastecc_destroy(sh.ctx);
        for (int h = 0; h < MAX_SUB; h++) {
            if (sh.ev_up[h]) (void)hipEventDestroy(sh.ev_up[h]);
            if (sh.ev_comp[h]) (void)hipEventDestroy(sh.ev_comp[h]);
        }
        if (sh.ev_all) (void)hipEventDestroy(sh.ev_all);
        for (hipStream_t q : {sh.s_up, sh.s_comp, sh.s_down})
            if (q) (void)hipStreamDestroy(q);
        for (hipStream_t q : sh.s_peer)
            if (q) {
                (void)hipStreamSynchronize(q);
                (void)hipStreamDestroy(q);
            }
        for (hipEvent_t e : sh.ev_peer)
            if (e) (void)hipEventDestroy(e);
        if (sh.data_slab) (void)hipFree(sh.data_slab);
        if (sh.parity_slab) (void)hipFree(sh.parity_slab);
    }
    if (s->fork && hipSetDevice(s->root) == hipSuccess) (void)hipEventDestroy(s->fork);
    delete s;
}

int sharded_encode_stripe(fastecc_ctx* shell, const void* data, void* parity, int mem_kind, hipStream_t st)
{
    if (mem_kind != FASTECC_MEM_DEVICE && mem_kind != FASTECC_MEM_HOST && mem_kind != FASTECC_MEM_HOST_PINNED) return FASTECC_E_INVAL;
    if (parity == data && sharded_of(shell)->M > sharded_of(shell)->K) return FASTECC_E_INVAL;  // in place: the parity must fit the data stripe (as fastecc_encode on one device)
    std::lock_guard<std::mutex> lk(mutex_of(shell));
    if (mem_kind == FASTECC_MEM_HOST && parity != data) {
        DeviceSwitch restore;
        const int rp = run_host_pageable(shell, data, parity, st);
        if (rp != FASTECC_OK) settle_after_failure(sharded_of(shell));
        return rp;
    }
    const int rc = run(shell, nullptr, data, nullptr, parity, mem_kind != FASTECC_MEM_DEVICE, st);
    if (rc != FASTECC_OK || mem_kind != FASTECC_MEM_HOST) return rc;
    // pageable host memory: the call is synchronous, like fastecc_encode on one device
    DeviceSwitch restore;
    SH_TRY(hipSetDevice(sharded_of(shell)->root));
    SH_TRY(hipStreamSynchronize(st));
    return FASTECC_OK;
}

fastecc_ctx* sharded_child(fastecc_ctx* shell, int g)
{
    Sharded* s = sharded_of(shell);
    return (s && g >= 0 && g < (int)s->shards.size()) ? s->shards[g].ctx : nullptr;
}

int sharded_forward(fastecc_ctx* shell, int what, const char* name, int value)
{
    Sharded* s = sharded_of(shell);
    std::lock_guard<std::mutex> lk(mutex_of(shell));
    if (what == SH_SET_OPTION && !strcmp(name, "sub_slabs")) {
        if (value < 1 || value > MAX_SUB || (value & (value - 1))) return FASTECC_E_INVAL;
        s->sub_slabs = value;
        describe(shell);
        return FASTECC_OK;
    }
    if (what == SH_SET_OPTION && !strcmp(name, "inject_fault")) {  // tests only: see injected_fault
        if (value < 0) return FASTECC_E_INVAL;
        s->fault_at = value;
        s->fault_step = 0;
        return FASTECC_OK;
    }
    if (what == SH_SET_OPTION && !strcmp(name, "row_pitch_words")) return FASTECC_E_UNSUPPORTED;  // slabs are contiguous [k][block_bytes / G]
    if (what == SH_SET_OPTION && !strcmp(name, "gather_mode")) {
        if (value != 1 && value != 2) return FASTECC_E_INVAL;
        s->gather_mode = value;
        s->copy_engine_refused = false;
        describe(shell);
        return FASTECC_OK;
    }
    if (what == SH_SET_OPTION && !strcmp(name, "stage_threads") && value >= 0 && value <= 64) s->stage_threads = value;  // (also forwarded)
    for (Shard& sh : s->shards) {
        int rc = FASTECC_OK;
        switch (what) {
            case SH_PROFILE_ENABLE: rc = fastecc_profile_enable(sh.ctx, value); break;
            case SH_PROFILE_RESET: rc = fastecc_profile_reset(sh.ctx); break;
            case SH_SET_OPTION: rc = fastecc_set_option(sh.ctx, name, value); break;
            case SH_SET_PLAN: rc = fastecc_set_plan(sh.ctx, value); break;
            default: rc = FASTECC_E_INVAL;
        }
        if (rc != FASTECC_OK) return rc;
    }
    describe(shell);
    return FASTECC_OK;
}

}  // namespace fastecc

using namespace fastecc;

extern "C" {

int fastecc_create_sharded(fastecc_ctx** out, uint64_t n, uint64_t k, uint64_t block_bytes, int field, const int* gpu_ids, int n_gpus)
{
    if (!out) return FASTECC_E_INVAL;
    *out = nullptr;
    if (!gpu_ids || n_gpus < 1 || n_gpus > 64) return FASTECC_E_INVAL;
    const uint64_t unit = field == FASTECC_FIELD_GF_P61_SQUARED ? 16 : 4;
    if (k < 1 || n <= k || block_bytes == 0 || (block_bytes % (unit * n_gpus)) != 0) return FASTECC_E_INVAL;
    int ndev = 0;
    {
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) return fail("hipGetDeviceCount", e == hipSuccess ? hipErrorNoDevice : e);
    }
    for (int g = 0; g < n_gpus; g++)
        if (gpu_ids[g] < 0 || gpu_ids[g] >= ndev) return FASTECC_E_INVAL;

    fastecc_ctx* shell = new_shell_ctx(gpu_ids[0], field, k, n - k, block_bytes);
    Sharded* s = new (std::nothrow) Sharded();
    if (!shell || !s) {
        delete s;
        if (shell) fastecc_destroy(shell);
        return FASTECC_E_NOMEM;
    }
    sharded_of(shell) = s;  // from here on fastecc_destroy(shell) releases everything built so far
    s->root = gpu_ids[0];
    s->field = field;
    s->K = k;
    s->M = n - k;
    s->block_bytes = block_bytes;
    s->slab_bytes = block_bytes / n_gpus;
    s->shards.resize(n_gpus);
    DeviceSwitch restore;
    auto bail = [&](int rc) {
        fastecc_destroy(shell);
        return rc;
    };
    for (int g = 0; g < n_gpus; g++) {
        Shard& sh = s->shards[g];
        sh.device = gpu_ids[g];
        if (sh.device != s->root) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, sh.device, s->root) != hipSuccess || !can) {
                set_error_detail("hipDeviceCanAccessPeer(slab device -> root)", hipErrorPeerAccessUnsupported);
                return bail(FASTECC_E_UNSUPPORTED);
            }
        }
        hipError_t e = hipSetDevice(sh.device);
        if (e != hipSuccess) return bail(fail("hipSetDevice", e));
        if (sh.device != s->root) {
            e = hipDeviceEnablePeerAccess(s->root, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(fail("hipDeviceEnablePeerAccess", e));
            (void)hipGetLastError();
        }
        const int rc = fastecc_create(&sh.ctx, n, k, s->slab_bytes, field, sh.device);
        if (rc != FASTECC_OK) return bail(rc);
        for (hipStream_t* q : {&sh.s_up, &sh.s_comp, &sh.s_down}) {
            e = hipStreamCreateWithFlags(q, hipStreamNonBlocking);
            if (e != hipSuccess) return bail(fail("hipStreamCreateWithFlags", e));
        }
        for (int h = 0; h < MAX_SUB; h++) {
            e = hipEventCreateWithFlags(&sh.ev_up[h], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sh.ev_comp[h], hipEventDisableTiming);
            if (e != hipSuccess) return bail(fail("hipEventCreateWithFlags", e));
        }
        e = hipEventCreateWithFlags(&sh.ev_all, hipEventDisableTiming);
        if (e != hipSuccess) return bail(fail("hipEventCreateWithFlags", e));
    }
    hipError_t e = hipSetDevice(s->root);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&s->fork, hipEventDisableTiming);
    if (e != hipSuccess) return bail(fail("hipEventCreateWithFlags(fork)", e));
    describe(shell);
    *out = shell;
    return FASTECC_OK;
}

int fastecc_encode_sharded(fastecc_ctx* c, const void* const* data_slabs, void* const* parity_slabs, void* parity, void* stream)
{
    if (!c || !sharded_of(c) || !data_slabs || (!parity_slabs && !parity)) return FASTECC_E_INVAL;
    Sharded* s = sharded_of(c);
    const uintptr_t mask = s->field == FASTECC_FIELD_GF_P61_SQUARED ? 15u : 3u;
    if ((uintptr_t)parity & mask) return FASTECC_E_INVAL;
    for (size_t g = 0; g < s->shards.size(); g++) {
        if (!data_slabs[g] || ((uintptr_t)data_slabs[g] & mask)) return FASTECC_E_INVAL;
        if (parity_slabs && (!parity_slabs[g] || ((uintptr_t)parity_slabs[g] & mask))) return FASTECC_E_INVAL;
    }
    std::lock_guard<std::mutex> lk(mutex_of(c));
    return run(c, data_slabs, nullptr, parity_slabs, parity, false, (hipStream_t)stream);
}

int fastecc_encode_sharded_blocks(fastecc_ctx* c, const void* const* data, int data_layout, void* const* parity_blocks, void* stream)
{
    if (!c || !sharded_of(c) || !data || !parity_blocks) return FASTECC_E_INVAL;
    if (data_layout != FASTECC_SHARD_SLABS && data_layout != FASTECC_SHARD_BLOCKS) return FASTECC_E_INVAL;
    Sharded* s = sharded_of(c);
    const size_t G = s->shards.size();
    if (s->M % G || (data_layout == FASTECC_SHARD_BLOCKS && s->K % G)) return FASTECC_E_INVAL;  // whole blocks per GPU
    const uintptr_t mask = s->field == FASTECC_FIELD_GF_P61_SQUARED ? 15u : 3u;
    for (size_t g = 0; g < G; g++)
        if (!data[g] || ((uintptr_t)data[g] & mask) || !parity_blocks[g] || ((uintptr_t)parity_blocks[g] & mask)) return FASTECC_E_INVAL;
    std::lock_guard<std::mutex> lk(mutex_of(c));
    DeviceSwitch restore;
    const int rc = run_blocks_body(c, data, data_layout == FASTECC_SHARD_BLOCKS, parity_blocks, (hipStream_t)stream);
    if (rc != FASTECC_OK) settle_after_failure(s);
    return rc;
}

int fastecc_shard_info(const fastecc_ctx* c, int* n_slabs, uint64_t* slab_block_bytes, int* devices, int cap)
{
    Sharded* s = c ? sharded_of(const_cast<fastecc_ctx*>(c)) : nullptr;
    if (!s) return FASTECC_E_INVAL;
    if (n_slabs) *n_slabs = (int)s->shards.size();
    if (slab_block_bytes) *slab_block_bytes = s->slab_bytes;
    if (devices)
        for (int g = 0; g < cap && g < (int)s->shards.size(); g++) devices[g] = s->shards[g].device;
    return FASTECC_OK;
}

}  // extern "C"
