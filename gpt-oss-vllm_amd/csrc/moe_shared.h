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
    size_t bytes;
    int max_pairs;       // P
};

__host__ inline MoeWorkspace carve_workspace(void* base, int T, int k, int H, int FF, int E_local) {
    const size_t P = (size_t)T * k;
    size_t off = 0;
    char* b = reinterpret_cast<char*>(base);
    MoeWorkspace w;
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
};

// Token tiles of the large-M kernel: up to this many rows of one expert share one pass over the expert's weights.
constexpr int TILE_ROWS = 256;

// Large-M fused dequant GEMM (moe_tile.hip).  mode 0: h = act(W13 x) from x [T][K] gathered through `sorted`;
// mode 1: y = W2 h from h [P][K] in sorted order.  W is [R][K] per expert in tile order; absmax fp32 (wam) or code
// bytes + per-expert table (wam8, wtab).  The routing lists in `ws` must have been built with token tiles of
// TILE_ROWS rows; `max_tiles` bounds their number.
int launch_tile_gemm(int mode, const half_t* xin, const uint8_t* wq, const float* wam, const uint8_t* wam8,
                     const float* wtab, const MoeWorkspace& ws, int K, int R, int k_top, int max_tiles, half_t* outp,
                     const StageParams& sp, hipStream_t st);
int tile_gemm_device_setup();

}  // namespace m355
