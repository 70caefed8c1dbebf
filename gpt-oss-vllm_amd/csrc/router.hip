// Router: fp16 linear (H -> E logits) + top-k + softmax over the selected logits, one workgroup per
// token.  The dot products are spread over the workgroup's waves (4 experts at a time per wave so
// each 16-byte piece of x is read once per 4 experts); wave 0 then selects the k largest logits by
// repeated wave-wide arg-max (value descending, expert id ascending on ties).
#include "router_shared.h"

namespace m355 {

constexpr int ROUTER_THREADS = 512;

// What the expert-parallel dispatch needs from the router's launch: the rows of the send buffer, one per (destination
// rank, token), [x: H fp16][ids: k int32][w: k fp32] padded to 16 bytes (m355_ep_row_bytes); a token with no expert on a
// destination gets ids = -1 there and no payload -- what ep_pack_kernel writes, written here by the workgroup that has
// just routed the token.
struct PackArg {
    unsigned char* send;   // null: no packing
    int T, n_ranks, epr;
    size_t row_bytes;
};

__global__ __launch_bounds__(ROUTER_THREADS) void router_topk_kernel(const half_t* __restrict__ x,
                                                                     const half_t* __restrict__ wr,
                                                                     const half_t* __restrict__ br, int H, int E, int k,
                                                                     int32_t* __restrict__ ids, float* __restrict__ w,
                                                                     half_t* __restrict__ logits_out, PackArg pk) {
    __shared__ float logit[ROUTER_MAX_E];
    __shared__ int32_t pick_id[32];
    __shared__ float pick_w[32];
    const int t = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    router_logits_wg<ROUTER_THREADS / 64>(x + (int64_t)t * H, wr, br, H, E, logit, logits_out ? logits_out + (int64_t)t * E : nullptr);
    __syncthreads();
    if (wave == 0) {
        topk_softmax_wave(logit, E, k, pick_id, pick_w);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int lane = threadIdx.x & 63;
        if (lane < k) {
            ids[(int64_t)t * k + lane] = pick_id[lane];
            w[(int64_t)t * k + lane] = pick_w[lane];
        }
    }
    if (!pk.send) return;
    __syncthreads();
    for (int r = 0; r < pk.n_ranks; r++) {
        const int lo = r * pk.epr, hi = lo + pk.epr;
        bool hit = false;
        for (int kk = 0; kk < k; kk++) hit |= (pick_id[kk] >= lo && pick_id[kk] < hi);   // (the same for every thread)
        unsigned char* dst = pk.send + ((size_t)r * pk.T + t) * pk.row_bytes;
        int32_t* dids = reinterpret_cast<int32_t*>(dst + (size_t)2 * H);
        float* dw = reinterpret_cast<float*>(dst + (size_t)2 * H + 4 * (size_t)k);
        if (!hit) {
            if ((int)threadIdx.x < k) dids[threadIdx.x] = -1;
            continue;
        }
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(x + (size_t)t * H);
        for (int c = threadIdx.x; c < H / 8; c += ROUTER_THREADS) reinterpret_cast<u32x4_t*>(dst)[c] = src[c];
        if ((int)threadIdx.x < k) {
            dids[threadIdx.x] = pick_id[threadIdx.x];
            dw[threadIdx.x] = pick_w[threadIdx.x];
        }
    }
}

// Many tokens (prefill): 16 tokens per workgroup on the matrix cores, so the router matrix is read from L2 once per
// 16 tokens instead of once per token.  logits[16 tokens][E] = x[16][H] Wr^T with v_mfma_f32_16x16x32_f16: the A
// operand is 16 token rows (lane (r, g) loads x[token r][32 s + 8 g ..] straight from global memory), the B operand
// 16 router rows per expert tile; the 4 waves take the 32-column steps s = wave, wave + 4, .. and their partial tiles
// are summed through LDS in wave order; then each wave selects for 4 of the tokens.
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
    if (router_uses_tile_form(T, E, H)) {
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
        (const half_t*)x_f16, (const half_t*)wr_f16, (const half_t*)br_f16, H, E, k, ids, w, (half_t*)logits_out_f16,
        PackArg{nullptr, 0, 0, 0, 0});
    M355_LAUNCH_CHECK("m355_router_topk");
    return 0;
}

extern "C" int m355_router_topk_ep_pack(const void* x_f16, const void* wr_f16, const void* br_f16, int T, int H, int E, int k,
                                        int32_t* ids, float* w, int n_ranks, int experts_per_rank, void* send,
                                        m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !wr_f16 || !ids || !w || !send) return fail(-1, "m355_router_topk_ep_pack: null pointer");
    if (T < 0 || H <= 0 || (H % 8) || E <= 0 || E > ROUTER_MAX_E || k <= 0 || k > 32 || k > E || n_ranks <= 0 ||
        n_ranks > 65535 || experts_per_rank <= 0)
        return fail(-2, "m355_router_topk_ep_pack: unsupported shape T=%d H=%d E=%d k=%d n_ranks=%d experts_per_rank=%d", T, H, E, k,
                    n_ranks, experts_per_rank);
    if (!aligned(x_f16, 16) || !aligned(wr_f16, 16) || !aligned(send, 16))
        return fail(-3, "m355_router_topk_ep_pack: x, wr and send must be 16-byte aligned");
    if (router_uses_tile_form(T, E, H)) {   // the 16-tokens-per-workgroup router does not pack: two launches
        if (int rc = m355_router_topk(x_f16, wr_f16, br_f16, T, H, E, k, ids, w, nullptr, stream)) return rc;
        return m355_ep_pack(x_f16, ids, w, T, k, H, n_ranks, experts_per_rank, send, stream);
    }
    router_topk_kernel<<<dim3(T), dim3(ROUTER_THREADS), 0, (hipStream_t)stream>>>(
        (const half_t*)x_f16, (const half_t*)wr_f16, (const half_t*)br_f16, H, E, k, ids, w, nullptr,
        PackArg{(unsigned char*)send, T, n_ranks, experts_per_rank, m355_ep_row_bytes(H, k)});
    M355_LAUNCH_CHECK("m355_router_topk_ep_pack");
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
