// Expert-parallel token exchange helpers: the byte movers either side of the two all-to-alls.
// One fixed-size row per (destination rank, local token): [x: H fp16][ids: k int32][w: k fp32], padded to 16 B.
// A token that routes no expert to a destination is marked ids = -1 there and its activations are not copied;
// the receiving rank's routing kernel drops such rows because -1 is nobody's expert.
#include "common.h"

namespace m355 {

__host__ __device__ inline size_t ep_row_bytes(int H, int k) { return ((size_t)2 * H + 8 * (size_t)k + 15) / 16 * 16; }

__global__ __launch_bounds__(128) void ep_pack_kernel(const half_t* __restrict__ x, const int32_t* __restrict__ ids,
                                                      const float* __restrict__ w, int T, int k, int H, int epr,
                                                      unsigned char* __restrict__ send) {
    const int t = blockIdx.x, r = blockIdx.y;
    const size_t rowB = ep_row_bytes(H, k);
    unsigned char* dst = send + ((size_t)r * T + t) * rowB;
    const int lo = r * epr, hi = lo + epr;
    bool hit = false;
    for (int kk = 0; kk < k; kk++) {
        const int e = ids[t * k + kk];
        hit |= (e >= lo && e < hi);
    }
    int32_t* dids = reinterpret_cast<int32_t*>(dst + (size_t)2 * H);
    float* dw = reinterpret_cast<float*>(dst + (size_t)2 * H + 4 * (size_t)k);
    if (!hit) {
        for (int kk = threadIdx.x; kk < k; kk += blockDim.x) dids[kk] = -1;
        return;
    }
    const u32x4_t* src = reinterpret_cast<const u32x4_t*>(x + (size_t)t * H);
    for (int c = threadIdx.x; c < H / 8; c += blockDim.x) reinterpret_cast<u32x4_t*>(dst)[c] = src[c];
    for (int kk = threadIdx.x; kk < k; kk += blockDim.x) {
        dids[kk] = ids[t * k + kk];
        dw[kk] = w[t * k + kk];
    }
}

__global__ __launch_bounds__(128) void ep_unpack_kernel(const unsigned char* __restrict__ recv, int k, int H,
                                                        half_t* __restrict__ x_all, int32_t* __restrict__ ids_all,
                                                        float* __restrict__ w_all) {
    const int t = blockIdx.x;
    const size_t rowB = ep_row_bytes(H, k);
    const unsigned char* src = recv + (size_t)t * rowB;
    const int32_t* sids = reinterpret_cast<const int32_t*>(src + (size_t)2 * H);
    const float* sw = reinterpret_cast<const float*>(src + (size_t)2 * H + 4 * (size_t)k);
    const bool hit = sids[0] >= 0;
    for (int kk = threadIdx.x; kk < k; kk += blockDim.x) {
        ids_all[t * k + kk] = hit ? sids[kk] : -1;
        w_all[t * k + kk] = hit ? sw[kk] : 0.f;
    }
    if (!hit) return;
    u32x4_t* dst = reinterpret_cast<u32x4_t*>(x_all + (size_t)t * H);
    for (int c = threadIdx.x; c < H / 8; c += blockDim.x) dst[c] = reinterpret_cast<const u32x4_t*>(src)[c];
}

// out[t] = sum over source ranks of their partial rows, accumulated in fp32 in rank order.
__global__ __launch_bounds__(256) void ep_reduce_kernel(const half_t* __restrict__ recv, int n_ranks, int T, int H,
                                                        half_t* __restrict__ out) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < H / 8; c += blockDim.x) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < n_ranks; r++) {
            const half8_t v = reinterpret_cast<const half8_t*>(recv + ((size_t)r * T + t) * H)[c];
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += (float)v[j];
        }
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (half_t)acc[j];
        reinterpret_cast<half8_t*>(out + (size_t)t * H)[c] = o;
    }
}

}  // namespace m355

using namespace m355;

extern "C" size_t m355_ep_row_bytes(int H, int k) { return (H > 0 && k > 0) ? ep_row_bytes(H, k) : 0; }

extern "C" int m355_ep_pack(const void* x_f16, const int32_t* ids, const float* w, int T, int k, int H, int n_ranks,
                            int experts_per_rank, void* send, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !ids || !w || !send) return fail(-1, "m355_ep_pack: null pointer");
    if (T < 0 || k <= 0 || H <= 0 || (H % 8) || n_ranks <= 0 || n_ranks > 65535 || experts_per_rank <= 0)
        return fail(-2, "m355_ep_pack: unsupported shape T=%d k=%d H=%d n_ranks=%d experts_per_rank=%d", T, k, H, n_ranks, experts_per_rank);
    if (!aligned(x_f16, 16) || !aligned(send, 16)) return fail(-3, "m355_ep_pack: x and send must be 16-byte aligned");
    ep_pack_kernel<<<dim3(T, n_ranks), dim3(128), 0, (hipStream_t)stream>>>((const half_t*)x_f16, ids, w, T, k, H,
                                                                            experts_per_rank, (unsigned char*)send);
    M355_LAUNCH_CHECK("ep_pack_kernel");
    return 0;
}

extern "C" int m355_ep_unpack(const void* recv, int NT, int k, int H, void* x_all_f16, int32_t* ids_all, float* w_all,
                              m355_stream_t stream) {
    if (NT == 0) return 0;
    if (!recv || !x_all_f16 || !ids_all || !w_all) return fail(-1, "m355_ep_unpack: null pointer");
    if (NT < 0 || k <= 0 || H <= 0 || (H % 8)) return fail(-2, "m355_ep_unpack: unsupported shape NT=%d k=%d H=%d", NT, k, H);
    if (!aligned(recv, 16) || !aligned(x_all_f16, 16)) return fail(-3, "m355_ep_unpack: recv and x_all must be 16-byte aligned");
    ep_unpack_kernel<<<dim3(NT), dim3(128), 0, (hipStream_t)stream>>>((const unsigned char*)recv, k, H, (half_t*)x_all_f16,
                                                                      ids_all, w_all);
    M355_LAUNCH_CHECK("ep_unpack_kernel");
    return 0;
}

extern "C" int m355_ep_reduce(const void* recv_f16, int n_ranks, int T, int H, void* out_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!recv_f16 || !out_f16) return fail(-1, "m355_ep_reduce: null pointer");
    if (n_ranks <= 0 || T < 0 || H <= 0 || (H % 8)) return fail(-2, "m355_ep_reduce: unsupported shape n_ranks=%d T=%d H=%d", n_ranks, T, H);
    if (!aligned(recv_f16, 16) || !aligned(out_f16, 16)) return fail(-3, "m355_ep_reduce: buffers must be 16-byte aligned");
    ep_reduce_kernel<<<dim3(T), dim3(256), 0, (hipStream_t)stream>>>((const half_t*)recv_f16, n_ranks, T, H, (half_t*)out_f16);
    M355_LAUNCH_CHECK("ep_reduce_kernel");
    return 0;
}
