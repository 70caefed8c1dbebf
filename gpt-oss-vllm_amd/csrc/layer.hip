// The small per-token operators that sit between the hot-path kernels in a gpt-oss decoder layer: residual add +
// RMSNorm, and rotary embedding + KV-cache write.  Both are a few KB per token: latency, not bandwidth.
// Arithmetic follows transformers' modeling_gpt_oss.py (GptOssRMSNorm.forward, _apply_rotary_emb) operation by
// operation, with fp16 tensors where that code has model-dtype tensors, so every intermediate rounds where theirs does.
#include "norm_shared.h"
#include "router_shared.h"

namespace m355 {

template <int DELTA>
__global__ __launch_bounds__(NORM_THREADS) void add_rmsnorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ delta,
                                                                   const half_t* __restrict__ weight, float eps, int H,
                                                                   half_t* __restrict__ resid_out, half_t* __restrict__ y,
                                                                   const half_t* __restrict__ cy, const int32_t* __restrict__ cinv,
                                                                   const float* __restrict__ cw, int k_top, int fp16_mode) {
    __shared__ float part[NORM_THREADS / 64];
    add_rmsnorm_row<DELTA>(true, blockIdx.x, threadIdx.x, x, delta, weight, eps, H, resid_out, y, cy, cinv, cw, k_top, fp16_mode,
                           part, nullptr);
}

// add + RMSNorm of a token's row, then the router on the normalised row (logits, top-k, softmax) by the same workgroup:
// the norm in front of the MoE block and the router that reads its output, one launch.  The norm is the first
// NORM_THREADS threads' work, exactly as add_rmsnorm_kernel does it; the router is router_topk_kernel's own code on the
// row kept in LDS (the fp16 values that also go to y): same bits as the two launches.
constexpr int NORM_ROUTER_THREADS = 512;
__global__ __launch_bounds__(NORM_ROUTER_THREADS) void add_rmsnorm_router_kernel(
    const half_t* __restrict__ x, const half_t* __restrict__ delta, const half_t* __restrict__ weight, float eps, int H,
    half_t* __restrict__ resid_out, half_t* __restrict__ y, const half_t* __restrict__ wr, const half_t* __restrict__ br, int E,
    int k, int32_t* __restrict__ ids, float* __restrict__ w) {
    extern __shared__ __attribute__((aligned(16))) half_t yrow[];   // [H]
    __shared__ float part[NORM_THREADS / 64];
    __shared__ float logit[ROUTER_MAX_E];
    const int t = blockIdx.x, tid = threadIdx.x;
    add_rmsnorm_row<0>(tid < NORM_THREADS, t, tid, x, delta, weight, eps, H, resid_out, y, nullptr, nullptr, nullptr, 0, 0, part, yrow);
    __syncthreads();
    router_logits_wg<NORM_ROUTER_THREADS / 64>(yrow, wr, br, H, E, logit, nullptr);
    __syncthreads();
    if (tid < 64) topk_softmax_wave(logit, E, k, ids + (int64_t)t * k, w + (int64_t)t * k);
}

// Rotate q and k (first half / second half pairing, every product and sum rounded to fp16), write q to q_out and
// k, v into the paged cache at the token's slot.  qkv row = [Hq q heads][Hkv k heads][Hkv v heads] x D.
// One workgroup per token; a work item is 8 columns of the first half of a head with their 8 partners.
constexpr int ROPE_THREADS = 256;
__global__ __launch_bounds__(ROPE_THREADS) void rope_kv_write_kernel(const half_t* __restrict__ qkv, const half_t* __restrict__ cos_t,
                                                                     const half_t* __restrict__ sin_t,
                                                                     const int32_t* __restrict__ positions,
                                                                     const int64_t* __restrict__ slots, int Hq, int Hkv, int D,
                                                                     int block_size, int max_pos, half_t* __restrict__ q_out,
                                                                     half_t* __restrict__ kcache, half_t* __restrict__ vcache) {
    const int t = blockIdx.x, tid = threadIdx.x;
    const int half_d = D / 2, groups = half_d / 8;           // 8-column groups per half head
    const int row_w = (Hq + 2 * Hkv) * D;
    const half_t* row = qkv + (int64_t)t * row_w;
    const int pos = min(max(positions[t], 0), max_pos - 1);   // the tables end at max_pos
    const int64_t slot = slots[t];
    const int64_t cache_row = slot >= 0 ? ((slot / block_size) * block_size + slot % block_size) * (int64_t)Hkv * D : -1;
    const half8_t* cr = reinterpret_cast<const half8_t*>(cos_t + (int64_t)pos * half_d);
    const half8_t* sr = reinterpret_cast<const half8_t*>(sin_t + (int64_t)pos * half_d);
    const int rot_items = (Hq + Hkv) * groups;
    for (int i = tid; i < rot_items; i += ROPE_THREADS) {
        const int head = i / groups, gi = i - head * groups;
        const half8_t a = reinterpret_cast<const half8_t*>(row + head * D)[gi];
        const half8_t b = reinterpret_cast<const half8_t*>(row + head * D + half_d)[gi];
        const half8_t c = cr[gi], s = sr[gi];
        half8_t o1, o2;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            half_t r1, r2;
            rope_rotate(a[j], b[j], c[j], s[j], r1, r2);
            o1[j] = r1;
            o2[j] = r2;
        }
        half_t* dst;
        if (head < Hq) {
            dst = q_out + ((int64_t)t * Hq + head) * D;
        } else {
            if (cache_row < 0) continue;
            dst = kcache + cache_row + (head - Hq) * D;
        }
        reinterpret_cast<half8_t*>(dst)[gi] = o1;
        reinterpret_cast<half8_t*>(dst + half_d)[gi] = o2;
    }
    if (cache_row >= 0) {
        const int v_items = Hkv * D / 8;
        const half8_t* vsrc = reinterpret_cast<const half8_t*>(row + (Hq + Hkv) * D);
        for (int i = tid; i < v_items; i += ROPE_THREADS) reinterpret_cast<half8_t*>(vcache + cache_row)[i] = vsrc[i];
    }
}

}  // namespace m355

using namespace m355;

extern "C" int m355_add_rmsnorm(const void* x_f16, const void* delta_f16, const void* weight_f16, float eps, int T, int H,
                                void* resid_out_f16, void* y_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !weight_f16 || !y_f16) return fail(-1, "m355_add_rmsnorm: null pointer");
    if (T < 0 || H <= 0 || (H % 8) || H > 8 * NORM_THREADS * 4)
        return fail(-2, "m355_add_rmsnorm: unsupported shape T=%d H=%d (need H %% 8 == 0, H <= %d)", T, H, 8 * NORM_THREADS * 4);
    if (!aligned(x_f16, 16) || !aligned(weight_f16, 16) || !aligned(y_f16, 16) || (delta_f16 && !aligned(delta_f16, 16)) ||
        (resid_out_f16 && !aligned(resid_out_f16, 16)))
        return fail(-3, "m355_add_rmsnorm: buffers must be 16-byte aligned");
    add_rmsnorm_kernel<0><<<dim3(T), dim3(NORM_THREADS), 0, (hipStream_t)stream>>>(
        (const half_t*)x_f16, (const half_t*)delta_f16, (const half_t*)weight_f16, eps, H, (half_t*)resid_out_f16, (half_t*)y_f16,
        nullptr, nullptr, nullptr, 0, 0);
    M355_LAUNCH_CHECK("add_rmsnorm_kernel");
    return 0;
}

extern "C" int m355_add_rmsnorm_router(const void* x_f16, const void* delta_f16, const void* weight_f16, float eps, int T, int H,
                                       void* resid_out_f16, void* y_f16, const void* wr_f16, const void* br_f16, int E, int k,
                                       int32_t* ids, float* w, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !weight_f16 || !y_f16 || !wr_f16 || !ids || !w) return fail(-1, "m355_add_rmsnorm_router: null pointer");
    if (T < 0 || H <= 0 || (H % 8) || H > 8 * NORM_THREADS * 4 || E <= 0 || E > ROUTER_MAX_E || k <= 0 || k > 32 || k > E)
        return fail(-2, "m355_add_rmsnorm_router: unsupported shape T=%d H=%d E=%d k=%d (need H %% 8 == 0, H <= %d, E <= %d, k <= min(32, E))",
                    T, H, E, k, 8 * NORM_THREADS * 4, ROUTER_MAX_E);
    if (!aligned(x_f16, 16) || !aligned(weight_f16, 16) || !aligned(y_f16, 16) || !aligned(wr_f16, 16) ||
        (delta_f16 && !aligned(delta_f16, 16)) || (resid_out_f16 && !aligned(resid_out_f16, 16)))
        return fail(-3, "m355_add_rmsnorm_router: buffers must be 16-byte aligned");
    if (router_uses_tile_form(T, E, H)) {   // many tokens: the router's 16-tokens-per-workgroup form takes over, as its own launch
        if (int rc = m355_add_rmsnorm(x_f16, delta_f16, weight_f16, eps, T, H, resid_out_f16, y_f16, stream)) return rc;
        return m355_router_topk(y_f16, wr_f16, br_f16, T, H, E, k, ids, w, nullptr, stream);
    }
    add_rmsnorm_router_kernel<<<dim3(T), dim3(NORM_ROUTER_THREADS), (size_t)H * 2, (hipStream_t)stream>>>(
        (const half_t*)x_f16, (const half_t*)delta_f16, (const half_t*)weight_f16, eps, H, (half_t*)resid_out_f16, (half_t*)y_f16,
        (const half_t*)wr_f16, (const half_t*)br_f16, E, k, ids, w);
    M355_LAUNCH_CHECK("add_rmsnorm_router_kernel");
    return 0;
}

extern "C" int m355_moe_combine_add_rmsnorm(const void* moe_workspace, size_t ws_bytes, int T, int k, int FF, int E_local,
                                            const float* w, int flags, const void* x_f16, const void* weight_f16, float eps,
                                            int H, void* resid_out_f16, void* y_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!moe_workspace || !w || !x_f16 || !weight_f16 || !y_f16) return fail(-1, "m355_moe_combine_add_rmsnorm: null pointer");
    if (T < 0 || k <= 0 || k > 32 || FF <= 0 || E_local <= 0 || H <= 0 || (H % 8) || H > 8 * NORM_THREADS * 4)
        return fail(-2, "m355_moe_combine_add_rmsnorm: unsupported shape T=%d k=%d H=%d FF=%d E_local=%d (need H %% 8 == 0, H <= %d)",
                    T, k, H, FF, E_local, 8 * NORM_THREADS * 4);
    if (!aligned(moe_workspace, 256) || !aligned(x_f16, 16) || !aligned(weight_f16, 16) || !aligned(y_f16, 16) ||
        (resid_out_f16 && !aligned(resid_out_f16, 16)))
        return fail(-3, "m355_moe_combine_add_rmsnorm: workspace 256-byte, other buffers 16-byte aligned");
    const MoeWorkspace ws = carve_workspace(const_cast<void*>(moe_workspace), T, k, H, FF, E_local);
    if (ws_bytes < ws.bytes)
        return fail(-6, "m355_moe_combine_add_rmsnorm: the workspace of this forward has %zu bytes, %zu given", ws.bytes, ws_bytes);
    add_rmsnorm_kernel<1><<<dim3(T), dim3(NORM_THREADS), 0, (hipStream_t)stream>>>(
        (const half_t*)x_f16, nullptr, (const half_t*)weight_f16, eps, H, (half_t*)resid_out_f16, (half_t*)y_f16, ws.y, ws.inv, w,
        k, (flags & M355_FLAG_FP16_COMBINE) ? 1 : 0);
    M355_LAUNCH_CHECK("add_rmsnorm_kernel");
    return 0;
}

extern "C" int m355_ep_reduce_add_rmsnorm(const void* recv_f16, int n_ranks, const void* x_f16, const void* weight_f16, float eps,
                                          int T, int H, void* resid_out_f16, void* y_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!recv_f16 || !x_f16 || !weight_f16 || !y_f16) return fail(-1, "m355_ep_reduce_add_rmsnorm: null pointer");
    if (T < 0 || n_ranks <= 0 || H <= 0 || (H % 8) || H > 8 * NORM_THREADS * 4)
        return fail(-2, "m355_ep_reduce_add_rmsnorm: unsupported shape n_ranks=%d T=%d H=%d (need H %% 8 == 0, H <= %d)", n_ranks, T, H,
                    8 * NORM_THREADS * 4);
    if (!aligned(recv_f16, 16) || !aligned(x_f16, 16) || !aligned(weight_f16, 16) || !aligned(y_f16, 16) ||
        (resid_out_f16 && !aligned(resid_out_f16, 16)))
        return fail(-3, "m355_ep_reduce_add_rmsnorm: buffers must be 16-byte aligned");
    add_rmsnorm_kernel<2><<<dim3(T), dim3(NORM_THREADS), 0, (hipStream_t)stream>>>(
        (const half_t*)x_f16, nullptr, (const half_t*)weight_f16, eps, H, (half_t*)resid_out_f16, (half_t*)y_f16,
        (const half_t*)recv_f16, nullptr, nullptr, n_ranks, 0);
    M355_LAUNCH_CHECK("add_rmsnorm_kernel");
    return 0;
}

extern "C" int m355_rope_kv_write(const void* qkv_f16, const void* cos_f16, const void* sin_f16, const int32_t* positions,
                                  const int64_t* slots, int T, int Hq, int Hkv, int D, int block_size, int max_pos,
                                  void* q_out_f16, void* kcache_f16, void* vcache_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!qkv_f16 || !cos_f16 || !sin_f16 || !positions || !slots || !q_out_f16 || !kcache_f16 || !vcache_f16)
        return fail(-1, "m355_rope_kv_write: null pointer");
    if (T < 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || (D % 16) || block_size <= 0 || max_pos <= 0)
        return fail(-2, "m355_rope_kv_write: unsupported shape T=%d Hq=%d Hkv=%d D=%d block_size=%d (need D %% 16 == 0)",
                    T, Hq, Hkv, D, block_size);
    if (!aligned(qkv_f16, 16) || !aligned(cos_f16, 16) || !aligned(sin_f16, 16) || !aligned(q_out_f16, 16) ||
        !aligned(kcache_f16, 16) || !aligned(vcache_f16, 16))
        return fail(-3, "m355_rope_kv_write: buffers must be 16-byte aligned");
    rope_kv_write_kernel<<<dim3(T), dim3(ROPE_THREADS), 0, (hipStream_t)stream>>>(
        (const half_t*)qkv_f16, (const half_t*)cos_f16, (const half_t*)sin_f16, positions, slots, Hq, Hkv, D, block_size, max_pos,
        (half_t*)q_out_f16, (half_t*)kcache_f16, (half_t*)vcache_f16);
    M355_LAUNCH_CHECK("rope_kv_write_kernel");
    return 0;
}
