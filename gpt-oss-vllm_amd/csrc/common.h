// Shared host/device helpers of libmoe355 (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>

#include "m355.h"

namespace m355 {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

// NF4 code book, patches/streaming_unsloth_bnb.py:101-106 of the reference.
#define M355_NF4_TABLE                                                                                   \
    {-1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f, -0.28444138169288635f,    \
     -0.18477343022823334f, -0.09105003625154495f, 0.0f, 0.07958029955625534f, 0.16093020141124725f,     \
     0.24611230194568634f, 0.33791524171829224f, 0.44070982933044434f, 0.5626170039176941f,              \
     0.7229568362236023f, 1.0f}

// ---- error reporting (thread-local message, integer status) ----
char* error_buffer();
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int hip_status(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    return fail((int)e, "%s: %s", what, hipGetErrorString(e));
}
#define M355_LAUNCH_CHECK(what)                                      \
    do {                                                             \
        hipError_t e__ = hipGetLastError();                          \
        if (e__ != hipSuccess) return m355::hip_status(e__, what);   \
    } while (0)

// Per-device set-up (kernel attributes and the like) that must run once on every device the library is used on.
// Host threads may call the library concurrently: the flags are read and written under a mutex.  A device index
// beyond the table runs the set-up on every call instead of sharing another device's flag.
struct PerDeviceOnce {
    std::mutex mu;
    bool done[64] = {false};
    template <typename F>
    int run(int dev, F setup) {
        std::lock_guard<std::mutex> lock(mu);
        const bool tracked = dev >= 0 && dev < 64;
        if (tracked && done[dev]) return 0;
        if (int rc = setup()) return rc;
        if (tracked) done[dev] = true;
        return 0;
    }
};

// Per-device set-up of each source file (kernel attributes, the static workspace's address), idempotent; the device is
// current.  m355_init runs them all so that nothing is left to a first call inside a graph capture.
int moe_device_init(int dev);
int dense_tile_device_init(int dev);
int moe_static_device_init(int dev);

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ---- device helpers ----
// Reinterpret a dword as two fp16.  Always go through this by-value helper: __builtin_bit_cast applied
// directly to an indexed vector element (v[j] with an unrolled loop index) was seen to read element 0
// for every j with hipcc 7.2.
__device__ __forceinline__ half2_t as_half2(uint32_t v) { return __builtin_bit_cast(half2_t, v); }
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_get(float v) {   // lanes outside ROW_MASK (or reading an invalid lane) get 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}
// Wave-wide sum with DPP only (no LDS crossbar round trips); every lane returns the total.
// Rotary embedding of one (first half, second half) pair of a head, as transformers' _apply_rotary_emb computes it on
// fp16 tensors: every product and every sum is rounded to fp16 (a fused multiply-add would skip a rounding).
__device__ __forceinline__ void rope_rotate(half_t a, half_t b, half_t c, half_t s, half_t& o1, half_t& o2) {
#pragma clang fp contract(off)
    const half_t ac = (half_t)((float)a * (float)c), bs = (half_t)((float)b * (float)s);
    const half_t bc = (half_t)((float)b * (float)c), as = (half_t)((float)a * (float)s);
    o1 = (half_t)((float)ac - (float)bs);
    o2 = (half_t)((float)bc + (float)as);
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_get<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_get<0x141>(v);         // row_half_mirror
    v += dpp_get<0x140>(v);         // row_mirror: every lane now holds its 16-lane row's sum
    v += dpp_get<0x142, 0xa>(v);    // row_bcast15 into rows 1 and 3
    v += dpp_get<0x143, 0xc>(v);    // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace m355
