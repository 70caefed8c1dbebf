// Device functions of the router shared by its own kernels (router.hip) and by the MoE GEMM launch that computes the
// routing itself for a handful of tokens (moe.hip): same code, so both give the same bits.
#pragma once
#include "common.h"

namespace m355 {

constexpr int ROUTER_MAX_E = 1024;

// Which of the two router kernels m355_router_topk runs: from ROUTER_TILE_MIN_T tokens on, 16 tokens per workgroup on the
// matrix cores; below (few 16-token workgroups are latency-bound), or for shapes that form does not take, one workgroup
// per token.  The two accumulate in different orders, so every caller that promises the bits of m355_router_topk (the
// fused norm + router launch, the router + pack launch) decides with this same function.
constexpr int ROUTER_TILE_MAX_E = 128;
constexpr int ROUTER_TILE_MIN_T = 1024;
__host__ __device__ inline bool router_uses_tile_form(int T, int E, int H) {
    return T >= ROUTER_TILE_MIN_T && E <= ROUTER_TILE_MAX_E && (H % 32) == 0;
}

// Wave-wide arg-max over (v, idx): larger v wins, on equal v the smaller idx wins.
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// Sort key of (logit, expert) for unsigned comparison: larger logit first, on equal logits the smaller expert id.  The
// logit is an exact fp16 number held as fp32 (or -inf), so the low 13 bits of its pattern carry nothing and six of them
// take 63 - expert; -0 counts as +0, as it does when floats are compared.
__device__ __forceinline__ uint32_t topk_sort_key(float v, int expert) {
    uint32_t b = __builtin_bit_cast(uint32_t, v);
    if (b == 0x80000000u) b = 0u;
    const uint32_t m = b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);   // monotone in v
    return (m & ~63u) | (uint32_t)(63 - expert);
}

// E <= 64: lane e holds logit e (an exact fp16 number); its rank is the number of logits that beat it (larger, or equal
// with a smaller index), counted on sort keys against one lane broadcast per rival; ranks < k are the picks, in order.
__device__ __forceinline__ void topk_softmax_wave64(float mine, int E, int k, int32_t* ids_out, float* w_out) {
    const int lane = threadIdx.x & 63;
    const bool valid = lane < E && !(mine != mine);
    const float v = valid ? mine : -INFINITY;
    const uint32_t key = lane < E ? topk_sort_key(v, lane) : 0u;   // (a lane past E beats nobody: every real key is > 0)
    int rank = 0;
#pragma unroll
    for (int e = 0; e < 32; e++) rank += ((uint32_t)__builtin_amdgcn_readlane((int)key, e) > key) ? 1 : 0;
    if (E > 32) {
#pragma unroll
        for (int e = 32; e < 64; e++) rank += ((uint32_t)__builtin_amdgcn_readlane((int)key, e) > key) ? 1 : 0;
    }
    if (lane >= E) rank = 64;
    const unsigned long long top = __ballot(rank == 0);
    const int l0 = top ? __builtin_ctzll(top) : 0;
    const float mx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l0));
    const float p = (rank < k) ? __expf(v - mx) : 0.f;
    const float sum = wave_sum(p);
    if (rank < k) {
        ids_out[rank] = lane;
        w_out[rank] = p / sum;
    }
}

// Selects top-k of logit[0..E) (LDS, fp32 values that are exact fp16 numbers); run by one whole wave.
__device__ __forceinline__ void topk_softmax_wave(float* logit, int E, int k, int32_t* ids_out, float* w_out) {
    const int lane = threadIdx.x & 63;
    if (E <= 64) {
        topk_softmax_wave64(lane < E ? logit[lane] : 0.f, E, k, ids_out, w_out);
        return;
    }
    float myv = 0.f;
    int myid = 0;
    for (int j = 0; j < k; j++) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int e = lane; e < E; e += 64) {
            const float v = logit[e];
            if (v > bv || (v == bv && e < bi)) { bv = v; bi = e; }   // NaN logits never win
        }
        wave_argmax(bv, bi);
        if (bi == 0x7fffffff) {   // every remaining logit is NaN: the lowest id that has not been picked yet
            for (int c = 0; c <= j; c++) {
                if (!__ballot(lane < j && myid == c)) { bi = c; break; }
            }
        }
        if (lane == j) { myv = bv; myid = bi; }
        // a picked entry becomes NaN, which never wins again (a -inf would tie with logits that are -inf themselves)
        if (lane == 0 && bi < E) logit[bi] = __builtin_nanf("");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const float mx = __shfl(myv, 0, 64);          // first pick is the largest
    float p = (lane < k) ? __expf(myv - mx) : 0.f;
    const float sum = wave_sum(p);
    if (lane < k) {
        ids_out[lane] = myid;
        w_out[lane] = p / sum;
    }
}

// The logits of experts e0 .. e0 + 3 of one token, by one wave: logit_out[e] (fp32 holding exact fp16 values; LDS or
// null) = fp16(x_row . wr[e] + br[e]), also stored to logits_global[e] (fp16) when given.  Each 16-byte piece of x is
// read once per 4 experts, six pieces per lane requested before any is used.
__device__ __forceinline__ void router_logits_wave(const half_t* __restrict__ x_row, const half_t* __restrict__ wr,
                                                   const half_t* __restrict__ br, int H, int E, int e0, float* logit_out,
                                                   half_t* __restrict__ logits_global) {
    const int lane = threadIdx.x & 63;
    const int chunks = H / 8;
    const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x_row);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 6;
    for (int cbase = 0; cbase < chunks; cbase += 64 * U) {
        u32x4_t xv[U], wv[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int c = min(cbase + u * 64 + lane, chunks - 1);
            xv[u] = xr[c];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int e = (e0 + i < E) ? e0 + i : E - 1;
                wv[u][i] = reinterpret_cast<const u32x4_t*>(wr + (int64_t)e * H)[c];
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (cbase + u * 64 + lane < chunks) {
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        acc[i] = __builtin_amdgcn_fdot2(as_half2(xv[u][j]), as_half2(wv[u][i][j]), acc[i], false);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float s = wave_sum(acc[i]);
        if (lane == 0 && e0 + i < E) {
            const float full = s + (br ? (float)br[e0 + i] : 0.f);
            const half_t l16 = (half_t)full;   // the reference's router output is an fp16 tensor
            if (logit_out) logit_out[e0 + i] = (float)l16;
            if (logits_global) logits_global[e0 + i] = l16;
        }
    }
}

// All E logits of one token by the NW waves of the calling workgroup, a wave taking 4 experts at a time.  The caller
// synchronises the workgroup before reading logit_out.
template <int NW>
__device__ __forceinline__ void router_logits_wg(const half_t* __restrict__ x_row, const half_t* __restrict__ wr,
                                                 const half_t* __restrict__ br, int H, int E, float* logit_out,
                                                 half_t* __restrict__ logits_global) {
    const int wave = threadIdx.x >> 6;
    for (int e0 = wave * 4; e0 < E; e0 += NW * 4) router_logits_wave(x_row, wr, br, H, E, e0, logit_out, logits_global);
}

}  // namespace m355
