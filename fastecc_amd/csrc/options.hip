// options.hip — fastecc_set_option and the per-kernel profiling of a context (include/fastecc.h).
#include "context.hpp"

using namespace fastecc;

extern "C" {

int fastecc_profile_enable(fastecc_ctx* c, int on)
{
    if (!c) return FASTECC_E_INVAL;
    if (c->sharded) return sharded_forward(c, SH_PROFILE_ENABLE, nullptr, on);
    CallLock lk(c->mu);
    c->profiling = on != 0;
    return FASTECC_OK;
}

int fastecc_profile_reset(fastecc_ctx* c)
{
    if (!c) return FASTECC_E_INVAL;
    if (c->sharded) return sharded_forward(c, SH_PROFILE_RESET, nullptr, 0);
    DeviceGuard dg(c->device);
    CallLock lk(c->mu);
    (void)hipDeviceSynchronize();
    c->prof_used = 0;
    return FASTECC_OK;
}

int fastecc_profile_read_bytes(fastecc_ctx* c, const char** names, double* ms, uint64_t* launches, uint64_t* bytes, int cap);

int fastecc_profile_read(fastecc_ctx* c, const char** names, double* ms, uint64_t* launches, int cap)
{
    return fastecc_profile_read_bytes(c, names, ms, launches, nullptr, cap);
}

int fastecc_set_option(fastecc_ctx* c, const char* name, int value)
{
    if (!c || !name) return FASTECC_E_INVAL;
    if (c->sharded) return sharded_forward(c, SH_SET_OPTION, name, value);
    if (c->p61 && strcmp(name, "decode_direct_max") != 0 && strcmp(name, "decode_split") != 0) return FASTECC_E_UNSUPPORTED;  // the other options tune the GF(0xFFF00001) tile kernels
    CallLock lk(c->mu);
    if (!strcmp(name, "row_pitch_words")) {
        // DEVICE stripes passed to fastecc_encode are then [k][pitch] words with the first block_bytes/4 of each
        // row valid: a host that owns its HBM layout can pad e.g. 4100-byte blocks to 4224 bytes so that every
        // 128-byte row segment is cache-line aligned.  0 restores the contiguous layout.
        const uint64_t pitch = value == 0 ? c->S : (uint64_t)value;
        if (value < 0 || pitch < c->S) return FASTECC_E_INVAL;
        if (pitch != c->ld) {
            // everything sized or laid out for the old pitch goes: the work stripes, and the decoder's pattern state
            // (its transform context and tables assume the geometry they were built with)
            DeviceGuard dgs(c->device);
            (void)hipDeviceSynchronize();
            if (c->scratch) (void)hipFree(c->scratch);
            if (c->parbuf) (void)hipFree(c->parbuf);
            if (c->mixbuf) (void)hipFree(c->mixbuf);
            c->scratch = c->parbuf = c->mixbuf = nullptr;
            destroy_decode_state(c->decoder);
            c->decoder = nullptr;
        }
        c->ld = pitch;
        build_plans(c);  // tile eligibility depends on the pitch
        DeviceGuard dg(c->device);
        if (!dg.ok) return hip_fail(hipErrorInvalidDevice, "hipSetDevice");
        HIP_TRY(hipDeviceSynchronize());
        return upload_twiddles(c);
    }
    if (!strcmp(name, "cache_policy")) {
        if (value < 0 || value > 15) return FASTECC_E_INVAL;
        c->cache_policy = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "xcd_swizzle")) {
        if (value < 0 || value > 2) return FASTECC_E_INVAL;
        c->xcd_swizzle = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "host_slabs")) {
        if (value < 1 || value > fastecc_ctx::MAX_SLABS || (value & (value - 1))) return FASTECC_E_INVAL;
        c->host_slabs = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "stage_threads")) {  // helper threads that move pageable host memory to / from the pinned staging slots (0 = automatic)
        if (value < 0 || value > 64) return FASTECC_E_INVAL;
        c->stage_threads = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "host_pipeline")) {  // FASTECC_MEM_HOST encodes of large stripes: column-slab pipeline through the staging rings (1) or upload, encode, download in turn (0)
        if (value < 0 || value > 1) return FASTECC_E_INVAL;
        c->host_pipeline = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "encode_direct_max")) {  // codes with at most this many parity blocks are encoded without the transform (0 = never)
        if (value < 0 || value > direct_encode_max()) return FASTECC_E_INVAL;
        c->encode_direct_max = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "decode_direct_max")) {  // takes effect at the next fastecc_decode_prepare
        if (value < 0 || value > (c->p61 ? p61::DECODE_DIRECT_MAX : direct_cap())) return FASTECC_E_INVAL;
        c->decode_direct_max = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "decode_split")) {  // (2k,k) codes, from the next fastecc_decode_prepare: see context.hpp
        if (value < 0 || value > 2) return FASTECC_E_INVAL;  // 2: the split transform in its block-group form only (A/B against the small form)
        c->decode_split = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "direct_kernel")) {  // 0 = choose, 1 = VALU, 2 = MFMA where the stripes allow it; decoder: from the next decode_prepare
        if (value < 0 || value > 2) return FASTECC_E_INVAL;
        c->direct_kernel = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "fuse_radix")) {  // mixed-radix contexts: 1 = odd-radix level fused into the outer tiles (default), 0 = its own passes
        if (value < 0 || value > 1) return FASTECC_E_INVAL;
        if (c->fuse_radix == value) return FASTECC_OK;
        DeviceGuard dg(c->device);  // the tables of THIS context's device are rebuilt: wait for its work, not the caller's current device's
        if (!dg.ok) return FASTECC_E_DEVICE;
        c->fuse_radix = value;
        HIP_TRY(hipDeviceSynchronize());
        build_plans(c);
        return upload_twiddles(c);
    }
    if (!strcmp(name, "slab_mode")) {  // 0: slabs staggered on internal streams, 1: one after the other on the caller's stream
        if (value < 0 || value > 1) return FASTECC_E_INVAL;
        c->slab_mode = value;
        return FASTECC_OK;
    }
    if (!strcmp(name, "slabs")) {
        if (value < 1 || value > fastecc_ctx::MAX_SLABS) return FASTECC_E_INVAL;
        c->slabs = value;
        return FASTECC_OK;
    }
    return FASTECC_E_INVAL;
}

int fastecc_profile_read_bytes(fastecc_ctx* c, const char** names, double* ms, uint64_t* launches, uint64_t* bytes, int cap)
{
    if (!c || !names || !ms || !launches || cap <= 0) return FASTECC_E_INVAL;
    if (c->sharded) return fastecc_profile_read_bytes(sharded_child(c, 0), names, ms, launches, bytes, cap);
    DeviceGuard dg(c->device);
    CallLock lk(c->mu);
    HIP_TRY(hipDeviceSynchronize());
    // names returned point into the context's records (valid until the next reset/launch)
    std::map<std::string, int> slot;
    int used = 0;
    for (size_t i = 0; i < c->prof_used; i++) {
        ProfileRec& r = c->prof[i];
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.start, r.stop) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        auto it = slot.find(r.name);
        int idx;
        if (it == slot.end()) {
            if (used == cap) continue;
            idx = used++;
            slot[r.name] = idx;
            names[idx] = r.name.c_str();
            ms[idx] = 0.0;
            launches[idx] = 0;
            if (bytes) bytes[idx] = 0;
        } else {
            idx = it->second;
        }
        ms[idx] += t;
        launches[idx] += 1;
        if (bytes) bytes[idx] += r.bytes;
    }
    return used;
}

}  // extern "C"
