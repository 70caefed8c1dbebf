// Router: fp16 linear (H -> E logits) + top-k + softmax over the selected logits, one workgroup per
// token.  The dot products are spread over the workgroup's waves (4 experts at a time per wave so
// each 16-byte piece of x is read once per 4 experts); wave 0 then selects the k largest logits by
// repeated wave-wide arg-max (value descending, expert id ascending on ties).
#include "common.h"

namespace m355 {

constexpr int ROUTER_THREADS = 512;
constexpr int ROUTER_MAX_E = 1024;

// Wave-wide arg-max over (v, idx): larger v wins, on equal v the smaller idx wins.
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// E <= 64: lane e holds logit e; its rank is the number of logits that beat it (larger, or equal with a
// smaller index), found with E scalar broadcasts; ranks < k are the picks, in order.
__device__ __forceinline__ void topk_softmax_wave64(float mine, int E, int k, int32_t* ids_out, float* w_out) {
    const int lane = threadIdx.x & 63;
    const bool valid = lane < E && !(mine != mine);
    const float v = valid ? mine : -INFINITY;
    int rank = 0;
    for (int e = 0; e < E; e++) {
        const float o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), e));
        rank += (o > v || (o == v && e < lane)) ? 1 : 0;
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
        if (bi == 0x7fffffff) bi = j;   // all remaining are NaN / -inf: fall back to a valid id
        if (lane == j) { myv = bv; myid = bi; }
        if (lane == 0 && bi < E) logit[bi] = -INFINITY;
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

__global__ __launch_bounds__(ROUTER_THREADS) void router_topk_kernel(const half_t* __restrict__ x,
                                                                     const half_t* __restrict__ wr,
                                                                     const half_t* __restrict__ br, int H, int E, int k,
                                                                     int32_t* __restrict__ ids, float* __restrict__ w,
                                                                     half_t* __restrict__ logits_out) {
    __shared__ float logit[ROUTER_MAX_E];
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = ROUTER_THREADS / 64;
    const int chunks = H / 8;
    const u32x4_t* xr = reinterpret_cast<const u32x4_t*>(x + (int64_t)t * H);
    for (int e0 = wave * 4; e0 < E; e0 += NW * 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // six 16-byte pieces of x and of four router rows per lane and round, all requested before any is used
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
                logit[e0 + i] = (float)l16;
                if (logits_out) logits_out[(int64_t)t * E + e0 + i] = l16;
            }
        }
    }
    __syncthreads();
    if (wave == 0) topk_softmax_wave(logit, E, k, ids + (int64_t)t * k, w + (int64_t)t * k);
}

// Many tokens (prefill): 16 tokens per workgroup on the matrix cores, so the router matrix is read from L2 once per
// 16 tokens instead of once per token.  logits[16 tokens][E] = x[16][H] Wr^T with v_mfma_f32_16x16x32_f16: the A
// operand is 16 token rows (lane (r, g) loads x[token r][32 s + 8 g ..] straight from global memory), the B operand
// 16 router rows per expert tile; the 4 waves take the 32-column steps s = wave, wave + 4, .. and their partial tiles
// are summed through LDS in wave order; then each wave selects for 4 of the tokens.
constexpr int ROUTER_TILE_MAX_E = 128;
constexpr int ROUTER_TILE_MIN_T = 1024;   // below, one workgroup per token is faster (few workgroups of 16 tokens are latency-bound)
template <int NJ>   // expert tiles of 16 (E <= 16 NJ)
__global__ __launch_bounds__(256) void router_topk_tile_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wr,
                                                               const half_t* __restrict__ br, int T, int H, int E, int k,
                                                               int32_t* __restrict__ ids, float* __restrict__ w,
                                                               half_t* __restrict__ logits_out) {
    extern __shared__ __attribute__((aligned(16))) float part[];   // [4 waves][16 tokens][E], then logits in [0][..][..]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.x * 16;
    const half_t* xrow = x + (int64_t)min(t0 + r, T - 1) * H + 8 * g;
    float4_t acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = float4_t{0.f, 0.f, 0.f, 0.f};
    const int steps = H / 32;
    // several 32-column steps per round, every load of the round requested before the first MFMA
    constexpr int U = NJ <= 2 ? 6 : (NJ <= 4 ? 4 : 2);
    for (int s0 = wave; s0 < steps; s0 += 4 * U) {
        half8_t a[U], b[U][NJ];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int s = min(s0 + 4 * u, steps - 1);
            a[u] = *reinterpret_cast<const half8_t*>(xrow + 32 * s);
#pragma unroll
            for (int j = 0; j < NJ; j++)
                b[u][j] = *reinterpret_cast<const half8_t*>(wr + (int64_t)min(16 * j + r, E - 1) * H + 32 * s + 8 * g);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (s0 + 4 * u < steps) {
#pragma unroll
                for (int j = 0; j < NJ; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[u][j], acc[j], 0, 0, 0);
            }
        }
    }
    // accumulator register i of lane (r, g): token 4 g + i, expert 16 j + r
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        if (16 * j + r < E) {
#pragma unroll
            for (int i = 0; i < 4; i++) part[((size_t)wave * 16 + 4 * g + i) * E + 16 * j + r] = acc[j][i];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * E; idx += 256) {
        const int e = idx % E, tok = idx / E;
        const float full = ((part[idx] + part[16 * E + idx]) + part[32 * E + idx]) + part[48 * E + idx] + (br ? (float)br[e] : 0.f);
        const half_t l16 = (half_t)full;   // the reference's router output is an fp16 tensor
        part[idx] = (float)l16;
        if (logits_out && t0 + tok < T) logits_out[(int64_t)(t0 + tok) * E + e] = l16;
    }
    __syncthreads();
    for (int i = 0; i < 4; i++) {
        const int t = t0 + 4 * wave + i;
        if (t < T) topk_softmax_wave(part + (size_t)(4 * wave + i) * E, E, k, ids + (int64_t)t * k, w + (int64_t)t * k);
    }
}

__global__ __launch_bounds__(256) void topk_softmax_kernel(const half_t* __restrict__ logits, int T, int E, int k,
                                                           int32_t* __restrict__ ids, float* __restrict__ w) {
    extern __shared__ __attribute__((aligned(16))) float lg[];   // [4][E]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= T) return;
    float* mine = lg + wave * E;
    for (int e = lane; e < E; e += 64) mine[e] = (float)logits[(int64_t)t * E + e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    topk_softmax_wave(mine, E, k, ids + (int64_t)t * k, w + (int64_t)t * k);
}

}  // namespace m355

using namespace m355;

extern "C" int m355_router_topk(const void* x_f16, const void* wr_f16, const void* br_f16, int T, int H, int E, int k,
                                int32_t* ids, float* w, void* logits_out_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !wr_f16 || !ids || !w) return fail(-1, "m355_router_topk: null pointer");
    if (T < 0 || H <= 0 || (H % 8) || E <= 0 || E > ROUTER_MAX_E || k <= 0 || k > 32 || k > E)
        return fail(-2, "m355_router_topk: unsupported shape T=%d H=%d E=%d k=%d (need H %% 8 == 0, E <= %d, k <= min(32,E))",
                    T, H, E, k, ROUTER_MAX_E);
    if (!aligned(x_f16, 16) || !aligned(wr_f16, 16)) return fail(-3, "m355_router_topk: x and wr must be 16-byte aligned");
    if (T == 0) return 0;
    if (T >= ROUTER_TILE_MIN_T && E <= ROUTER_TILE_MAX_E && (H % 32) == 0) {
        const dim3 grid((T + 15) / 16), block(256);
        const size_t lds = 4 * 16 * E * sizeof(float);
        hipStream_t st = (hipStream_t)stream;
#define M355_ROUTER_TILE(NJ)                                                                                          \
    router_topk_tile_kernel<NJ><<<grid, block, lds, st>>>((const half_t*)x_f16, (const half_t*)wr_f16, (const half_t*)br_f16, \
                                                          T, H, E, k, ids, w, (half_t*)logits_out_f16)
        if (E <= 16) M355_ROUTER_TILE(1);
        else if (E <= 32) M355_ROUTER_TILE(2);
        else if (E <= 64) M355_ROUTER_TILE(4);
        else M355_ROUTER_TILE(8);
#undef M355_ROUTER_TILE
        M355_LAUNCH_CHECK("m355_router_topk");
        return 0;
    }
    router_topk_kernel<<<dim3(T), dim3(ROUTER_THREADS), 0, (hipStream_t)stream>>>(
        (const half_t*)x_f16, (const half_t*)wr_f16, (const half_t*)br_f16, H, E, k, ids, w, (half_t*)logits_out_f16);
    M355_LAUNCH_CHECK("m355_router_topk");
    return 0;
}

extern "C" int m355_topk_softmax(const void* logits_f16, int T, int E, int k, int32_t* ids, float* w,
                                 m355_stream_t stream) {
    if (T == 0) return 0;
    if (!logits_f16 || !ids || !w) return fail(-1, "m355_topk_softmax: null pointer");
    if (T < 0 || E <= 0 || E > ROUTER_MAX_E || k <= 0 || k > 32 || k > E)
        return fail(-2, "m355_topk_softmax: unsupported shape T=%d E=%d k=%d", T, E, k);
    if (T == 0) return 0;
    topk_softmax_kernel<<<dim3((T + 3) / 4), dim3(256), 4 * E * sizeof(float), (hipStream_t)stream>>>(
        (const half_t*)logits_f16, T, E, k, ids, w);
    M355_LAUNCH_CHECK("m355_topk_softmax");
    return 0;
}
