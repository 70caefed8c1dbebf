// The MoE forward for callers that cannot allocate: the workspace that the other entry points are handed by their
// caller lives in a file-scope device array instead, one copy per GPU at an address of its own.
//
// In host code the array's name is only a handle, not a device address: the address kernels can use comes from
// hipGetSymbolAddress, asked once per device with that device current (m355_init, or the first call on the device),
// and that address is what every launch is given, exactly as a caller's workspace pointer would be.
#include "common.h"

namespace m355 {

static __device__ __align__(256) unsigned char scratch[M355_STATIC_WORKSPACE_BYTES];
static void* scratch_ptr[64];   // per device; written under the PerDeviceOnce mutex before the device's flag is set

static int scratch_address(int dev, void** out) {
    static PerDeviceOnce once;
    const bool tracked = dev >= 0 && dev < 64;
    void* fresh = nullptr;
    if (int rc = once.run(dev, [&] {
            void* p = nullptr;
            if (int rc = hip_status(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)")) return rc;
            if (!p) return fail(-7, "m355: the static workspace has no address on device %d", dev);
            if (tracked) scratch_ptr[dev] = p;
            fresh = p;
            return 0;
        }))
        return rc;
    *out = tracked ? scratch_ptr[dev] : fresh;
    return *out ? 0 : fail(-7, "m355: the static workspace has no address on device %d", dev);
}

int moe_static_device_init(int dev) {
    void* p = nullptr;
    return scratch_address(dev, &p);
}

}  // namespace m355

using namespace m355;

extern "C" int m355_moe_nf4_forward_static(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                           const uint8_t* w13_q, const float* w13_absmax, const uint8_t* w2_q,
                                           const float* w2_absmax, int E_local, int e_first, int H, int FF, int act_mode,
                                           int flags, const void* b13_f16, const void* b2_f16, float alpha, float limit,
                                           void* out_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (T < 0 || k <= 0 || H <= 0 || FF <= 0 || E_local <= 0)
        return fail(-2, "m355_moe_nf4_forward_static: unsupported shape T=%d k=%d E_local=%d H=%d FF=%d", T, k, E_local, H, FF);
    const size_t need = m355_moe_workspace_bytes(T, k, H, FF, E_local);
    if (need == 0 || need > (size_t)M355_STATIC_WORKSPACE_BYTES)
        return fail(-6, "m355_moe_nf4_forward_static: T=%d k=%d H=%d FF=%d E_local=%d needs %zu bytes of workspace, the "
                        "static one holds %zu: use m355_moe_nf4_forward_ex with a caller-owned workspace",
                    T, k, H, FF, E_local, need, (size_t)M355_STATIC_WORKSPACE_BYTES);
    int dev = 0;
    if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
    void* ws = nullptr;
    if (int rc = scratch_address(dev, &ws)) return rc;
    // Every list the kernels read from the workspace is written by this call's own routing step first, and the h / y rows
    // are written by the GEMMs before anything reads them: nothing has to start at zero, whatever the last call left.
    return m355_moe_nf4_forward_ex(x_f16, T, ids, w, k, w13_q, w13_absmax, w2_q, w2_absmax, E_local, e_first, H, FF, act_mode,
                                   flags, b13_f16, b2_f16, alpha, limit, ws, (size_t)M355_STATIC_WORKSPACE_BYTES, out_f16,
                                   stream);
}
