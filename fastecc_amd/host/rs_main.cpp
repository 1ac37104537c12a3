// rs_main.cpp — `rs`-compatible host driver over the C ABI (include/fastecc.h).
//
// Keeps the command line of the reference benchmark (RS.cpp:71-87, RS.md:4):
//     rs_hip [.][log2(N) [block_bytes [gpus=g0,g1,...]]]    defaults: N = 2^19, 2052-byte blocks, "." = quiet
// (gpus=...: one stripe in column slabs on several GPUs through fastecc_create_sharded — BASELINE configs[3]; the same
// fastecc_encode calls, g0 is the root.  Ids may repeat, e.g. gpus=0,0,0,0 on a one-GPU box.)
// fills the stripe with the reference's i % p pattern (RS.cpp:28-29), encodes it on the GPU through
// fastecc_encode, and reports time / MiB/s in the reference's convention (data + parity bytes per
// second, RS.cpp:38, wall_clock_timer.h:91).  Unlike the reference it also prints the parity checksum
// (main.cpp:202-212 formula) so a run can be compared with SURVEY.md Appendix B.
//
// The host stays C++ and knows nothing about HIP kernels: everything goes through the C ABI.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fastecc.h"

static uint32_t rolling_hash(const uint32_t* w, size_t count)
{
    uint32_t h = 314159253u;
    for (size_t i = 0; i < count; i++) h = (h + w[i]) * 123456791u + (h >> 17);
    return h;
}

static double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    bool verbose = true;
    int logn = 19;
    size_t block_bytes = 2052;
    int arg = 1;
    if (argc > arg && argv[arg][0] == '.') {
        verbose = false;
        if (argv[arg][1] == 0) arg++;
        else argv[arg]++;
    }
    if (argc > arg) logn = atoi(argv[arg++]);
    if (argc > arg) block_bytes = (size_t)atoll(argv[arg++]);
    std::vector<int> gpus;
    if (argc > arg && !strncmp(argv[arg], "gpus=", 5)) {
        for (const char* p = argv[arg] + 5; *p;) {
            gpus.push_back(atoi(p));
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        arg++;
    }
    const uint64_t N = 1ull << logn;
    const size_t words = block_bytes / 4;  // RS.cpp:86 truncates the same way
    const size_t total = N * words;

    fastecc_ctx* ctx = nullptr;
    int rc = gpus.empty() ? fastecc_create(&ctx, 2 * N, N, words * 4, FASTECC_FIELD_GF_FFF00001, 0)
                          : fastecc_create_sharded(&ctx, 2 * N, N, words * 4, FASTECC_FIELD_GF_FFF00001, gpus.data(), (int)gpus.size());
    if (rc != FASTECC_OK) {
        fprintf(stderr, "%s: %s (%s)\n", gpus.empty() ? "fastecc_create" : "fastecc_create_sharded", fastecc_strerror(rc), fastecc_last_error_detail());
        return 1;
    }
    if (!gpus.empty() && hipSetDevice(gpus[0]) != hipSuccess) return 1;  // stripes live on the root

    std::vector<uint32_t> host(total);
    for (size_t i = 0; i < total; i++) host[i] = (uint32_t)(i % 0xFFF00001ull);
    if (verbose) printf("Allocated %.0lf MiB\n", total * 4 / 1048576.0);

    void* dev = nullptr;
    if (hipMalloc(&dev, total * 4) != hipSuccess) {
        fprintf(stderr, "Can't alloc %.0lf MiB of device memory!\n", total * 4 / 1048576.0);
        return 1;
    }
    const double t_h2d0 = now_ms();
    if (hipMemcpy(dev, host.data(), total * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    const double t_h2d1 = now_ms();

    // one untimed encode of a scratch stripe first: module load and per-kernel LDS configuration happen on the
    // first launch of each kernel and are not part of the steady-state rate the reference's MiB/s line is compared to
    {
        void* scratch = nullptr;
        if (hipMalloc(&scratch, total * 4) == hipSuccess) {
            (void)hipMemset(scratch, 0, total * 4);
            (void)fastecc_encode(ctx, scratch, scratch, FASTECC_MEM_DEVICE, nullptr);
            (void)hipDeviceSynchronize();
            (void)hipFree(scratch);
        } else {
            (void)hipGetLastError();
        }
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, nullptr);
    rc = fastecc_encode(ctx, dev, dev, FASTECC_MEM_DEVICE, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    if (rc != FASTECC_OK) {
        fprintf(stderr, "fastecc_encode: %s\n", fastecc_strerror(rc));
        return 1;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);

    const double t_d2h0 = now_ms();
    if (hipMemcpy(host.data(), dev, total * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const double t_d2h1 = now_ms();

    const double bytes = 2.0 * N * words * 4;
    printf("Reed-Solomon encoding (2^%.0lf source blocks => 2^%.0lf ECC blocks, %.0lf bytes each): %.3lf ms = %.0lf MiB/s"
           "  [MI355X, HBM-resident; plan %s]\n",
           std::log2((double)N), std::log2((double)N), words * 4.0, ms, bytes / ms * 1000 / (1 << 20),
           fastecc_plan_string(ctx));
    if (verbose) {
        const double e2e = (t_h2d1 - t_h2d0) + ms + (t_d2h1 - t_d2h0);
        printf("  with PCIe copies: %.0lf ms = %.0lf MiB/s (h2d %.0lf ms, d2h %.0lf ms)\n", e2e, bytes / e2e * 1000 / (1 << 20),
               t_h2d1 - t_h2d0, t_d2h1 - t_d2h0);
        printf("  parity checksum: %u\n", rolling_hash(host.data(), total));
    }
    if (verbose && !gpus.empty()) {
        // the host-memory form on a sharded context: every GPU moves its own column slab over its own host link
        std::vector<uint32_t> in(total), out(total);
        for (size_t i = 0; i < total; i++) in[i] = (uint32_t)(i % 0xFFF00001ull);
        (void)fastecc_encode(ctx, in.data(), out.data(), FASTECC_MEM_HOST, nullptr);  // (first call: every slab's staging buffers)
        const double h0 = now_ms();
        rc = fastecc_encode(ctx, in.data(), out.data(), FASTECC_MEM_HOST, nullptr);
        const double h1 = now_ms();
        printf("  host stripe through %d slabs (each GPU its own host link, pageable memory): %.0lf ms = %.0lf MiB/s, parity %s\n",
               (int)gpus.size(), h1 - h0, bytes / (h1 - h0) * 1000 / (1 << 20), rc == FASTECC_OK && out == host ? "identical" : "MISMATCH");
    }
    if (verbose && gpus.empty()) {
        // the drop-in call itself: the caller's malloc'ed (pageable) buffers, as RS.cpp holds them — upload, encode, threaded download
        std::vector<uint32_t> in(total), out(total);
        for (size_t i = 0; i < total; i++) in[i] = (uint32_t)(i % 0xFFF00001ull);
        (void)fastecc_encode(ctx, in.data(), out.data(), FASTECC_MEM_HOST, nullptr);  // (first call: staging buffers)
        const double h0 = now_ms();
        rc = fastecc_encode(ctx, in.data(), out.data(), FASTECC_MEM_HOST, nullptr);
        const double h1 = now_ms();
        printf("  fastecc_encode(FASTECC_MEM_HOST) on malloc'ed buffers: %.0lf ms = %.0lf MiB/s, parity %s\n", h1 - h0, bytes / (h1 - h0) * 1000 / (1 << 20),
               rc == FASTECC_OK && out == host ? "identical" : "MISMATCH");
    }
    if (verbose && gpus.empty()) {
        // Beyond the reference (it documents decoding, README.md:83-119, and has no code for it): lose every third data
        // block and every fifth parity block of the codeword just produced, and repair the data on the GPU.
        uint32_t* ddata = nullptr;
        if (hipMalloc(&ddata, total * 4) == hipSuccess) {
            std::vector<uint32_t> data(total);
            for (size_t i = 0; i < total; i++) data[i] = (uint32_t)(i % 0xFFF00001ull);
            std::vector<uint8_t> dflag(N, 1), pflag(N, 1);
            uint64_t lost = 0;
            for (uint64_t i = 0; i < N; i += 3) dflag[i] = 0, lost++;
            for (uint64_t i = 0; i < N; i += 5) pflag[i] = 0, lost++;
            std::vector<uint32_t> damaged(data);
            for (uint64_t i = 0; i < N; i += 3) std::fill(damaged.begin() + i * words, damaged.begin() + (i + 1) * words, 0xFFFFFFFFu);
            (void)hipMemcpy(ddata, damaged.data(), total * 4, hipMemcpyHostToDevice);
            const double p0 = now_ms();
            rc = fastecc_decode_prepare(ctx, dflag.data(), pflag.data());
            const double p1 = now_ms();
            if (rc == FASTECC_OK) {
                (void)hipEventRecord(e0, nullptr);
                rc = fastecc_decode(ctx, ddata, dev /* the parity */, FASTECC_MEM_DEVICE, nullptr);
                (void)hipEventRecord(e1, nullptr);
                (void)hipEventSynchronize(e1);
            }
            if (rc == FASTECC_OK) {
                (void)hipEventElapsedTime(&ms, e0, e1);
                (void)hipMemcpy(damaged.data(), ddata, total * 4, hipMemcpyDeviceToHost);
                printf("  erasure decoding, %llu of %llu blocks lost: pattern set-up %.0lf ms, decode %.3lf ms = %.0lf MiB/s, data %s\n",
                       (unsigned long long)lost, (unsigned long long)(2 * N), p1 - p0, ms, bytes / ms * 1000 / (1 << 20),
                       damaged == data ? "restored bit for bit" : "MISMATCH");
            } else {
                printf("  erasure decoding: %s\n", fastecc_strerror(rc));
            }
            // the common failure: two lost blocks (one data, one parity) — the decoder's direct one-pass path, parity rebuilt as well
            std::fill(dflag.begin(), dflag.end(), 1);
            std::fill(pflag.begin(), pflag.end(), 1);
            dflag[N / 3] = 0;
            pflag[N / 7] = 0;
            (void)hipMemcpy(ddata, data.data(), total * 4, hipMemcpyHostToDevice);
            (void)hipMemset((char*)ddata + (N / 3) * words * 4, 0xEE, words * 4);
            (void)hipMemset((char*)dev + (N / 7) * words * 4, 0xDD, words * 4);
            const double q0 = now_ms();
            rc = fastecc_decode_prepare(ctx, dflag.data(), pflag.data());
            const double q1 = now_ms();
            if (rc == FASTECC_OK) {
                (void)hipEventRecord(e0, nullptr);
                rc = fastecc_repair(ctx, ddata, dev, FASTECC_MEM_DEVICE, nullptr);
                (void)hipEventRecord(e1, nullptr);
                (void)hipEventSynchronize(e1);
            }
            if (rc == FASTECC_OK) {
                (void)hipEventElapsedTime(&ms, e0, e1);
                std::vector<uint32_t> par_back(total);
                (void)hipMemcpy(damaged.data(), ddata, total * 4, hipMemcpyDeviceToHost);
                (void)hipMemcpy(par_back.data(), dev, total * 4, hipMemcpyDeviceToHost);
                printf("  repair of 2 lost blocks (1 data, 1 parity): pattern set-up %.1lf ms, repair %.3lf ms = %.0lf MiB/s, stripe %s\n", q1 - q0, ms,
                       bytes / ms * 1000 / (1 << 20), damaged == data && par_back == host ? "restored bit for bit" : "MISMATCH");
            } else {
                printf("  repair of 2 lost blocks: %s\n", fastecc_strerror(rc));
            }
            (void)hipFree(ddata);
        } else {
            (void)hipGetLastError();
        }
    }
    (void)hipFree(dev);
    fastecc_destroy(ctx);
    return 0;
}
