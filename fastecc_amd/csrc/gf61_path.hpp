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
constexpr int DEFAULT_LEVELS = 4;
int create(Path** out, int n, uint64_t elems, char* detail, size_t detail_cap);
void destroy(Path* p);

int encode(Path* p, const uint64_t* data, uint64_t* parity, hipStream_t st, const LaunchHooks* hooks);
int ntt(Path* p, uint64_t* data, bool inverse, hipStream_t st, const LaunchHooks* hooks);
// words >= p among the 2 * elems * k words of a stripe; `counter` is a device uint64 the caller zeroed
int count_out_of_range(Path* p, const uint64_t* data, unsigned long long* counter, hipStream_t st);

// plan id (fastecc_set_plan): 0 = default (LDS tiles + 4-level register passes), 1..4 = register passes only with that many
// levels, 10 + L / 20 + L = tiles with a 64 / 128 KiB exchange buffer; rebuilds the tables.  The device must be idle.
int set_plan(Path* p, int plan, char* detail, size_t detail_cap);
const char* plan_string(const Path* p);

}  // namespace p61
}  // namespace fastecc
