// Residual add + RMSNorm of one token's row as a device function, shared by the norm kernels (layer.hip) and by the
// kernels that put the norm in front of their own work in the same launch (the fused norm + q/k/v projection of
// dense.hip): same code, so every form gives the same bits.
#pragma once
#include "moe_shared.h"

namespace m355 {

constexpr int NORM_THREADS = 256;

// r = fp16(x + delta) (or x); y = fp16(w * (float(r) * rsqrt(mean(float(r)^2) + eps))).  One workgroup per token.
// DELTA 1: delta is not read but formed here, as the MoE combine step forms it (combine_chunk: same bits), from the
// expert outputs cy / cinv / cw that a forward called with M355_FLAG_SKIP_COMBINE left in its workspace.
// DELTA 2: delta is formed as the expert-parallel reduce step forms it: cy holds the ranks' partial rows [k_top ranks]
// [tokens][H] fp16, summed in fp32 in rank order and rounded to fp16 (ep_reduce_kernel: same bits).
// The work of one token's row, by the first NORM_THREADS threads of the workgroup (`active`; the others only keep the
// barrier).  y (optional): where the normalised rows go in memory; y_copy (optional, LDS): the normalised row for a
// consumer in the same workgroup.  `part` (LDS, NORM_THREADS / 64 floats) must not be shared with a call that other
// waves may still be in.
template <int DELTA>
__device__ __forceinline__ void add_rmsnorm_row(bool active, int t, int tid, const half_t* __restrict__ x,
                                                const half_t* __restrict__ delta, const half_t* __restrict__ weight, float eps,
                                                int H, half_t* __restrict__ resid_out, half_t* __restrict__ y,
                                                const half_t* __restrict__ cy, const int32_t* __restrict__ cinv,
                                                const float* __restrict__ cw, int k_top, int fp16_mode, float* part,
                                                half_t* y_copy) {
    const int chunks = H / 8;
    const half8_t* xr = reinterpret_cast<const half8_t*>(x + (int64_t)t * H);
    const half8_t* dr = delta ? reinterpret_cast<const half8_t*>(delta + (int64_t)t * H) : nullptr;
    constexpr int MAXC = 4;   // chunks a thread keeps in registers: H <= 8 * 256 * 4 = 8192
    half8_t r[MAXC];
    float ss = 0.f;
#pragma unroll
    for (int u = 0; u < MAXC; u++) {
        const int c = tid + u * NORM_THREADS;
        if (active && c < chunks) {
            half8_t v = xr[c];
            if constexpr (DELTA == 1) {
                v = v + combine_chunk<4>(cy, cinv, cw, t, c, H, k_top, fp16_mode, k_top);
            } else if constexpr (DELTA == 2) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int rk = 0; rk < k_top; rk++) {
                    const half8_t pv = reinterpret_cast<const half8_t*>(cy + ((int64_t)rk * gridDim.x + t) * H)[c];
#pragma unroll
                    for (int j = 0; j < 8; j++) acc[j] += (float)pv[j];
                }
                half8_t d;
#pragma unroll
                for (int j = 0; j < 8; j++) d[j] = (half_t)acc[j];
                v = v + d;
            } else {
                if (dr) v = v + dr[c];   // fp16 add, rounded, as the residual tensors are
            }
            r[u] = v;
#pragma unroll
            for (int j = 0; j < 8; j++) ss += (float)v[j] * (float)v[j];
        }
    }
    ss = wave_sum(ss);
    if (active && (tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float tot = part[0] + part[1] + part[2] + part[3];
    const float rs = rsqrtf(tot / (float)H + eps);
#pragma unroll
    for (int u = 0; u < MAXC; u++) {
        const int c = tid + u * NORM_THREADS;
        if (active && c < chunks) {
            const half8_t wv = reinterpret_cast<const half8_t*>(weight)[c];
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float n = (float)r[u][j] * rs;     // fp32 product, then the weight: two fp32 roundings
                o[j] = (half_t)((float)wv[j] * n);
            }
            if (y) reinterpret_cast<half8_t*>(y + (int64_t)t * H)[c] = o;
            if (y_copy) reinterpret_cast<half8_t*>(y_copy)[c] = o;
            if (resid_out) reinterpret_cast<half8_t*>(resid_out + (int64_t)t * H)[c] = r[u];
        }
    }
}

}  // namespace m355
