// Declarations shared by the two fused NF4 MoE GEMM kernels (moe.hip: few tokens per expert, HBM-bound;
// moe_tile.hip: many tokens per expert, MFMA-bound).
#pragma once
#include "common.h"

namespace m355 {

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct MoeWorkspace {
    int32_t* n_active;   // [1]
    int32_t* offsets;    // [E_local + 1]
    int32_t* active;     // [E_local]
    int32_t* sorted;     // [P]   sorted position -> pair index (t*k + kk)
    int32_t* inv;        // [P]   pair index -> sorted position, -1 if the expert is not local
    int32_t* n_tiles;    // [1]   token tiles (<= mt_rows tokens of one expert) over all active experts
    int32_t* tile_expert;  // [P + E_local]
    int32_t* tile_m0;      // [P + E_local]  first token (within the expert) of the tile
    half_t* h;           // [P][FF]
    half_t* y;           // [P][H]
    int32_t* pg_done;    // [SYNC_PG_COUNTERS]  finished token tiles per group of output columns of the down stage (the
                         //   gate_up launch clears them, the down launch counts: see CombineIO)
    size_t bytes;
    int max_pairs;       // P
};

constexpr int SYNC_PG_COUNTERS = 128;   // >= groups of output columns of the down stage (H <= 3072: at most 96)
static_assert(SYNC_PG_COUNTERS >= 96, "one counter per 32 output columns of the down stage (H <= 3072, single-pair items)");

__host__ inline MoeWorkspace carve_workspace(void* base, int T, int k, int H, int FF, int E_local) {
    const size_t P = (size_t)T * k;
    size_t off = 0;
    char* b = reinterpret_cast<char*>(base);
    MoeWorkspace w;
    w.pg_done = reinterpret_cast<int32_t*>(b + off);  off = align_up(off + 4 * (size_t)SYNC_PG_COUNTERS, 256);
    w.n_active = reinterpret_cast<int32_t*>(b + off); off = align_up(off + 4, 256);
    w.offsets = reinterpret_cast<int32_t*>(b + off);  off = align_up(off + 4 * (size_t)(E_local + 1), 256);
    w.active = reinterpret_cast<int32_t*>(b + off);   off = align_up(off + 4 * (size_t)E_local, 256);
    w.sorted = reinterpret_cast<int32_t*>(b + off);   off = align_up(off + 4 * P, 256);
    w.inv = reinterpret_cast<int32_t*>(b + off);      off = align_up(off + 4 * P, 256);
    w.n_tiles = reinterpret_cast<int32_t*>(b + off);  off = align_up(off + 4, 256);
    w.tile_expert = reinterpret_cast<int32_t*>(b + off); off = align_up(off + 4 * (P + (size_t)E_local), 256);
    w.tile_m0 = reinterpret_cast<int32_t*>(b + off);  off = align_up(off + 4 * (P + (size_t)E_local), 256);
    w.h = reinterpret_cast<half_t*>(b + off);         off = align_up(off + 2 * P * FF, 256);
    w.y = reinterpret_cast<half_t*>(b + off);         off = align_up(off + 2 * P * H, 256);
    w.bytes = off;
    w.max_pairs = (int)P;
    return w;
}

// Activation and bias of one GEMM stage.  bias: [E_local][R] fp16 in the kernel's row order (gate rows, then up rows),
// or null.
struct StageParams {
    int act_mode;            // M355_ACT_* (gate_up stage only)
    float alpha, limit;      // swiglu-oai
    const half_t* bias;
    int64_t x_stride;        // elements between the rows of the stage's input (the row length, unless the rows sit
                             // inside wider records, as in the receive buffer of the expert-parallel exchange)
};


// Device-scope accesses: the store is performed at memory and the load is served from there, whichever compute die's L2
// the address may sit in -- what workgroups of one launch need to hand data to each other across dies without writing
// back or invalidating whole caches.
__device__ __forceinline__ void store_device_scope(half_t* p, half2_t v) {   // p 4-byte aligned
    __hip_atomic_store(reinterpret_cast<unsigned int*>(p), __builtin_bit_cast(unsigned int, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ half8_t load_half8_device_scope(const half_t* p) {
    const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
    unsigned long long v[2];
    v[0] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[1] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(half8_t, v);
}

// The combine step folded into the down stage's launch (streaming kernel).  A work item of that launch is (token tile,
// group of output columns); when the last token tile of a column group has been written, every y value those columns of
// `out` depend on exists, and the workgroup that finished last combines them.  Completion is counted per column group in
// pg_done, which the gate_up launch of the same forward call has cleared (it precedes the down launch on the stream).
// out == null: not folded (the combine runs as its own launch, or the caller combines).
struct CombineIO {
    half_t* out;             // [T][H]
    const int32_t* inv;      // pair index -> row of y, -1: expert not local
    const float* w;          // routing weights, w_stride floats per token
    int w_stride, T, fp16_mode;
    int32_t* pg_done;        // gate_up: cleared when non-null; down: counted
};

// The combine step for 8 columns of one token: sum over the token's local pairs, in k order.  FP16 mode rounds as the
// reference's fp16 tensors do: w -> fp16, product -> fp16, running sum -> fp16.  The pairs are taken KB at a time: their
// positions and weights are requested together, then their y rows together, and only the arithmetic runs in k order, so a
// batch costs two memory round trips instead of two per pair.  Shared by the combine kernel and by the kernel that fuses
// the combine with the residual add and RMSNorm that follow it, so both give the same bits.
// THROUGH: y is read with device-scope loads (it was written by other workgroups of the same launch).
// The positions and routing weights of KB pairs of token t, from pair k0 on (-1 / 0 past the token's last pair).
template <int KB>
__device__ __forceinline__ void combine_fetch_pairs(const int32_t* __restrict__ inv, const float* __restrict__ w, int t, int k0,
                                                    int k_top, int w_stride, int (&pos)[KB], float (&wf)[KB]) {
#pragma unroll
    for (int i = 0; i < KB; i++) {
        const bool in = k0 + i < k_top;
        pos[i] = in ? inv[t * k_top + k0 + i] : -1;
        wf[i] = in ? w[(int64_t)t * w_stride + k0 + i] : 0.f;
    }
}

// acc += the weighted y rows of those KB pairs, columns 8c .. 8c + 7, in k order.
template <int KB, bool THROUGH>
__device__ __forceinline__ void combine_accumulate(const half_t* __restrict__ y, const int (&pos)[KB], const float (&wf)[KB],
                                                   int c, int H, int fp16_mode, float (&acc)[8]) {
    half8_t yv[KB];
#pragma unroll
    for (int i = 0; i < KB; i++)
        if (pos[i] >= 0) {
            if constexpr (THROUGH) yv[i] = load_half8_device_scope(y + (int64_t)pos[i] * H + 8 * c);
            else yv[i] = reinterpret_cast<const half8_t*>(y + (int64_t)pos[i] * H)[c];
        }
#pragma unroll
    for (int i = 0; i < KB; i++) {
        if (pos[i] < 0) continue;
        if (fp16_mode) {
            const float w16 = (float)(half_t)wf[i];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float prod = (float)(half_t)(w16 * (float)yv[i][j]);
                acc[j] = (float)(half_t)(acc[j] + prod);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) acc[j] += wf[i] * (float)yv[i][j];
        }
    }
}

template <int KB, bool THROUGH = false>
__device__ __forceinline__ half8_t combine_chunk(const half_t* __restrict__ y, const int32_t* __restrict__ inv,
                                                 const float* __restrict__ w, int t, int c, int H, int k_top, int fp16_mode,
                                                 int w_stride) {   // floats between the tokens' weight rows (k_top: dense)
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    for (int k0 = 0; k0 < k_top; k0 += KB) {
        int pos[KB];
        float wf[KB];
        combine_fetch_pairs<KB>(inv, w, t, k0, k_top, w_stride, pos, wf);
        combine_accumulate<KB, THROUGH>(y, pos, wf, c, H, fp16_mode, acc);
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (half_t)acc[j];
    return o;
}

// Token tiles of the large-M kernel: up to this many rows of one expert share one pass over the expert's weights.
int tile_gemm_rows();

// Large-M fused dequant GEMM (moe_tile.hip).  mode 0: h = act(W13 x) from x [T][K] gathered through `sorted`;
// mode 1: y = W2 h from h [P][K] in sorted order.  W is [R][K] per expert in tile order; absmax fp32 (wam) or code
// bytes + per-expert table (wam8, wtab).  The routing lists in `ws` must have been built with token tiles of
// tile_gemm_rows() rows; `max_tiles` bounds their number.
int launch_tile_gemm(int mode, const half_t* xin, const uint8_t* wq, const float* wam, const uint8_t* wam8,
                     const float* wtab, const MoeWorkspace& ws, int K, int R, int k_top, int max_tiles, half_t* outp,
                     const StageParams& sp, hipStream_t st);
int tile_gemm_device_setup();

}  // namespace m355
