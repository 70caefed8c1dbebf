// Paged-attention decode for grouped-query heads (8 query heads per KV head, head size 64).
// HBM-bound on the KV read: 2 * Hkv * D * 2 bytes per cached token per sequence.
//
// One workgroup (4 waves) serves one (sequence, KV head, 256-token partition): the 8 query heads of the
// group share every K and V row that is fetched.  A lane is (token slot ts = lane/8, 16-byte piece dc =
// lane%8), so a wave-load covers 8 whole 128-byte rows.  Pass 1 streams K and leaves the scaled scores
// in LDS, the softmax numerators are formed once per partition (no running rescale), pass 2 streams V.
// Longer contexts use several partitions plus a small merge kernel.
#include "common.h"

namespace m355 {

constexpr int ATT_THREADS = 256;
constexpr int ATT_PART = 256;   // tokens per partition
constexpr int ATT_G = 8;        // query heads per KV head
constexpr int ATT_D = 64;

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 8 lanes that share a token slot (lane%8 varies): quad xor 1, quad xor 2, half-row mirror
__device__ __forceinline__ float sum8(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    return v;
}

__global__ __launch_bounds__(ATT_THREADS) void paged_attn_decode_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ kc, const half_t* __restrict__ vc,
    const int32_t* __restrict__ block_tables, const int32_t* __restrict__ ctx_lens, int Hq, int Hkv, int block_size,
    int max_blocks, float scale, int nparts, float* __restrict__ part_o, float* __restrict__ part_ml,
    half_t* __restrict__ out, const half_t* __restrict__ sinks, int window) {
    __shared__ __attribute__((aligned(16))) float sc[ATT_PART * ATT_G];   // [token][head]
    __shared__ __attribute__((aligned(16))) float wred[4 * ATT_G * ATT_D];
    __shared__ float m_s[ATT_G], l_s[ATT_G];

    const int b = blockIdx.x, hk = blockIdx.y, part = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ts = lane >> 3, dc = lane & 7;
    const int L = ctx_lens[b];
    const int t0 = part * ATT_PART;
    // sliding window: the query (position L - 1) sees keys L - window .. L - 1; tokens below `first` are masked
    const int first = window > 0 ? max(L - window, 0) - t0 : 0;
    int npt = min(ATT_PART, L - t0);   // tokens of this partition below the context end (may be <= 0)
    if (first >= npt) npt = 0;         // the whole partition lies before the window
    const int64_t ob = ((int64_t)b * Hq + hk * ATT_G);

    if (npt <= 0) {
        if (nparts == 1) {
            for (int i = tid; i < ATT_G * ATT_D; i += ATT_THREADS) out[ob * ATT_D + i] = (half_t)0.f;
        } else {
            for (int i = tid; i < ATT_G * ATT_D; i += ATT_THREADS) {
                const int h = i / ATT_D, d = i % ATT_D;
                part_o[((ob + h) * nparts + part) * ATT_D + d] = 0.f;
            }
            if (tid < ATT_G) {
                part_ml[((ob + tid) * nparts + part) * 2 + 0] = -INFINITY;
                part_ml[((ob + tid) * nparts + part) * 2 + 1] = 0.f;
            }
        }
        return;
    }

    // this lane's 8 columns of the 8 query heads
    u32x4_t qv[ATT_G];
#pragma unroll
    for (int h = 0; h < ATT_G; h++) qv[h] = reinterpret_cast<const u32x4_t*>(q + (ob + h) * ATT_D)[dc];

    const int32_t* bt = block_tables + (int64_t)b * max_blocks;
    constexpr int ITERS = ATT_PART / 32;
    const int64_t row_stride = (int64_t)Hkv * ATT_D;

    // ---- pass 1: scores ----  (the V rows are requested together with the K rows: one latency, not two)
    u32x4_t vv[ITERS];
    {
        u32x4_t kv[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int tl = min(it * 32 + wave * 8 + ts, npt - 1);
            const int t = t0 + tl;
            const int blk = bt[t / block_size];
            const int64_t row = ((int64_t)blk * block_size + t % block_size) * row_stride + hk * ATT_D;
            kv[it] = reinterpret_cast<const u32x4_t*>(kc + row)[dc];
            vv[it] = reinterpret_cast<const u32x4_t*>(vc + row)[dc];
        }
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int tl = it * 32 + wave * 8 + ts;
            float mine = 0.f;
#pragma unroll
            for (int h = 0; h < ATT_G; h++) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 4; j++)
                    s = __builtin_amdgcn_fdot2(as_half2(qv[h][j]), as_half2(kv[it][j]), s, false);
                s = sum8(s);
                if (dc == h) mine = s;
            }
            sc[tl * ATT_G + dc] = (tl < npt && tl >= first) ? mine * scale : -INFINITY;
        }
    }
    __syncthreads();

    // ---- softmax numerators: 32 threads per head ----
    {
        const int h = tid >> 5, i = tid & 31;
        float v[ATT_PART / 32];
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < ATT_PART / 32; u++) {
            v[u] = sc[(u * 32 + i) * ATT_G + h];
            mx = fmaxf(mx, v[u]);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < ATT_PART / 32; u++) {
            const float pv = __expf(v[u] - mx);
            sum += pv;
            sc[(u * 32 + i) * ATT_G + h] = pv;
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (i == 0) { m_s[h] = mx; l_s[h] = sum; }
    }
    __syncthreads();

    // ---- pass 2: o[h][d] += p[h][t] * v[t][d] ----
    float o[ATT_G][8];
#pragma unroll
    for (int h = 0; h < ATT_G; h++)
#pragma unroll
        for (int j = 0; j < 8; j++) o[h][j] = 0.f;
    {
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const int tl = it * 32 + wave * 8 + ts;
            const float4_t p0 = *reinterpret_cast<const float4_t*>(&sc[tl * ATT_G]);       // 0 for masked tokens
            const float4_t p1 = *reinterpret_cast<const float4_t*>(&sc[tl * ATT_G + 4]);
            float vf[8];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const half2_t h2 = as_half2(vv[it][j]);
                vf[2 * j] = (float)h2[0];
                vf[2 * j + 1] = (float)h2[1];
            }
#pragma unroll
            for (int h = 0; h < ATT_G; h++) {
                const float ph = (h < 4) ? p0[h & 3] : p1[h & 3];
#pragma unroll
                for (int j = 0; j < 8; j++) o[h][j] += ph * vf[j];
            }
        }
    }
    // sum over the 8 token slots of the wave (lane bits 3,4,5), then over the 4 waves through LDS
#pragma unroll
    for (int h = 0; h < ATT_G; h++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float v = o[h][j];
            v += dpp_mov<0x128>(v);        // row_ror:8  (lane ^ 8 within the 16-lane row)
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            o[h][j] = v;
        }
    if (ts == 0) {
#pragma unroll
        for (int h = 0; h < ATT_G; h++) {
            float4_t a = {o[h][0], o[h][1], o[h][2], o[h][3]}, c = {o[h][4], o[h][5], o[h][6], o[h][7]};
            *reinterpret_cast<float4_t*>(&wred[(wave * ATT_G + h) * ATT_D + dc * 8]) = a;
            *reinterpret_cast<float4_t*>(&wred[(wave * ATT_G + h) * ATT_D + dc * 8 + 4]) = c;
        }
    }
    __syncthreads();
    for (int i = tid; i < ATT_G * ATT_D; i += ATT_THREADS) {
        const int h = i / ATT_D, d = i % ATT_D;
        const float tot = wred[(0 * ATT_G + h) * ATT_D + d] + wred[(1 * ATT_G + h) * ATT_D + d] +
                          wred[(2 * ATT_G + h) * ATT_D + d] + wred[(3 * ATT_G + h) * ATT_D + d];
        if (nparts == 1) {
            // a sink is one more logit in the softmax denominator (its value row is dropped)
            float num = tot, den = l_s[h];
            if (sinks) {
                const float sk = (float)sinks[hk * ATT_G + h], m2 = fmaxf(m_s[h], sk), f = __expf(m_s[h] - m2);
                num *= f;
                den = den * f + __expf(sk - m2);
            }
            out[(ob + h) * ATT_D + d] = (half_t)(num / den);
        } else {
            part_o[((ob + h) * nparts + part) * ATT_D + d] = tot;
            if (d == 0) {
                part_ml[((ob + h) * nparts + part) * 2 + 0] = m_s[h];
                part_ml[((ob + h) * nparts + part) * 2 + 1] = l_s[h];
            }
        }
    }
}

// Merge the partitions of one (sequence, query head): lane = output column.
__global__ __launch_bounds__(64) void paged_attn_merge_kernel(const float* __restrict__ part_o,
                                                              const float* __restrict__ part_ml, int nparts,
                                                              half_t* __restrict__ out, const half_t* __restrict__ sinks,
                                                              int Hq) {
    const int64_t bh = blockIdx.x;
    const int d = threadIdx.x;
    float mx = -INFINITY;
    for (int p = 0; p < nparts; p++) mx = fmaxf(mx, part_ml[(bh * nparts + p) * 2]);
    float num = 0.f, den = 0.f;
    if (sinks) {
        const float sk = (float)sinks[bh % Hq];
        mx = fmaxf(mx, sk);
        den = __expf(sk - mx);
    }
    for (int p = 0; p < nparts; p++) {
        const float m = part_ml[(bh * nparts + p) * 2], l = part_ml[(bh * nparts + p) * 2 + 1];
        if (l <= 0.f) continue;
        const float f = __expf(m - mx);
        num += f * part_o[(bh * nparts + p) * ATT_D + d];
        den += f * l;
    }
    out[bh * ATT_D + d] = (half_t)(den > 0.f ? num / den : 0.f);
}

}  // namespace m355

using namespace m355;

extern "C" size_t m355_paged_attn_workspace_bytes(int B, int Hq, int D, int max_ctx) {
    if (B <= 0 || Hq <= 0 || D <= 0 || max_ctx <= ATT_PART) return 0;
    const size_t nparts = ((size_t)max_ctx + ATT_PART - 1) / ATT_PART;
    return (size_t)B * Hq * nparts * (D + 2) * sizeof(float);
}

extern "C" int m355_paged_attn_decode(const void* q_f16, const void* kcache_f16, const void* vcache_f16,
                                      const int32_t* block_tables, const int32_t* ctx_lens, int B, int Hq, int Hkv, int D,
                                      int block_size, int max_blocks, int max_ctx, float scale, void* workspace,
                                      size_t ws_bytes, void* out_f16, m355_stream_t stream) {
    return m355_paged_attn_decode_ex(q_f16, kcache_f16, vcache_f16, block_tables, ctx_lens, B, Hq, Hkv, D, block_size,
                                     max_blocks, max_ctx, scale, nullptr, 0, workspace, ws_bytes, out_f16, stream);
}

extern "C" int m355_paged_attn_decode_ex(const void* q_f16, const void* kcache_f16, const void* vcache_f16,
                                         const int32_t* block_tables, const int32_t* ctx_lens, int B, int Hq, int Hkv, int D,
                                         int block_size, int max_blocks, int max_ctx, float scale, const void* sinks_f16,
                                         int window, void* workspace, size_t ws_bytes, void* out_f16,
                                         m355_stream_t stream) {
    if (B == 0) return 0;
    if (window < 0) return fail(-2, "m355_paged_attn_decode: negative window %d", window);
    if (!q_f16 || !kcache_f16 || !vcache_f16 || !block_tables || !ctx_lens || !out_f16)
        return fail(-1, "m355_paged_attn_decode: null pointer");
    if (B < 0 || Hkv <= 0 || Hq != Hkv * ATT_G || D != ATT_D || block_size <= 0 || max_blocks <= 0 || max_ctx <= 0)
        return fail(-2, "m355_paged_attn_decode: unsupported shape B=%d Hq=%d Hkv=%d D=%d block_size=%d (need D == 64, Hq == 8*Hkv)",
                    B, Hq, Hkv, D, block_size);
    if ((int64_t)max_blocks * block_size < max_ctx)
        return fail(-2, "m355_paged_attn_decode: max_ctx %d exceeds max_blocks*block_size %lld", max_ctx, (long long)max_blocks * block_size);
    if (!aligned(q_f16, 16) || !aligned(kcache_f16, 16) || !aligned(vcache_f16, 16))
        return fail(-3, "m355_paged_attn_decode: q and caches must be 16-byte aligned");
    if (B == 0) return 0;
    const int nparts = (max_ctx + ATT_PART - 1) / ATT_PART;
    const size_t need = m355_paged_attn_workspace_bytes(B, Hq, D, max_ctx);
    float *po = nullptr, *pml = nullptr;
    if (nparts > 1) {
        if (!workspace || ws_bytes < need) return fail(-6, "m355_paged_attn_decode: workspace of %zu bytes needed, %zu given", need, ws_bytes);
        po = reinterpret_cast<float*>(workspace);
        pml = po + (size_t)B * Hq * nparts * D;
    }
    hipStream_t st = (hipStream_t)stream;
    paged_attn_decode_kernel<<<dim3(B, Hkv, nparts), dim3(ATT_THREADS), 0, st>>>(
        (const half_t*)q_f16, (const half_t*)kcache_f16, (const half_t*)vcache_f16, block_tables, ctx_lens, Hq, Hkv,
        block_size, max_blocks, scale, nparts, po, pml, (half_t*)out_f16, (const half_t*)sinks_f16, window);
    M355_LAUNCH_CHECK("paged_attn_decode_kernel");
    if (nparts > 1) {
        paged_attn_merge_kernel<<<dim3(B * Hq), dim3(64), 0, st>>>(po, pml, nparts, (half_t*)out_f16, (const half_t*)sinks_f16, Hq);
        M355_LAUNCH_CHECK("paged_attn_merge_kernel");
    }
    return 0;
}
