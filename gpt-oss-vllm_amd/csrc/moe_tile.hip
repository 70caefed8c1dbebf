// Fused NF4-dequant grouped GEMM for many tokens per expert (prefill, large decode batches): MFMA-bound.
//
// A workgroup owns up to 256 tokens of one expert (a token tile) times 256 weight rows (8 row-tile pairs of the tile
// order that m355_repack_nf4 writes, one pair = 16 + 16 rows per wave) and walks K in 64-column blocks, the NF4 absmax
// block.  Per block:
//   * each wave takes its own packed KiB (brought into LDS by DMA two blocks ahead) and turns it into the A fragments of
//     v_mfma_f32_16x16x32_f16 exactly as the tile order holds them -- lane (r, g) owns 16 codes of row r of the pair's
//     first tile and 16 of row r of its partner tile, columns 16g .. 16g + 15: two K-steps of 32 columns each, the
//     lane's share of a step being columns 16g + 8t .. + 7 -- and reuses them for every token of the tile: the weights
//     never pass through LDS as fp16 and are dequantised once per 256 tokens;
//   * the activations of the token tile (256 x 64 fp16 = 32 KiB) are shared by the 8 waves: LDS-DMA
//     (global_load_lds_dwordx4) in whole 128-byte lines, one block ahead, into one of two buffers, the 16-byte chunks
//     XOR-swizzled on the source side so that the ds_read_b128 of the B fragments (token r of a 16-token tile, the same
//     columns as the lane's A fragment) is bank-conflict free.
// The 16x16x32 shape is used although 32x32x16 needs half as many instructions for the same product: under sustained
// matrix load the chip holds a higher clock with it (measured here: +5 % on the whole kernel at equal work).
#include <type_traits>

#include "moe_shared.h"

namespace m355 {


#ifndef TG_WAVES_N
#define TG_WAVES_N 8
#endif
constexpr int TG_WAVES = TG_WAVES_N;             // 8: two waves per SIMD
constexpr int TG_THREADS = TG_WAVES * 64;
constexpr int TG_LUT_BYTES = 256 * 256;          // half2 per byte value, 32 replicas of 4 B in a 256-byte row
// The kernel is a template over RT, the row-tile pairs (32 weight rows each) a wave owns, and NTT, the 16-token tiles of
// a token tile: a workgroup covers 256 RT weight rows x 16 NTT tokens, and a wave's accumulators are 8 RT NTT registers.
//   <1, 16>: 256 rows x 256 tokens (the form in use).  A weight is dequantised once per 256 tokens; every wave reads the
//            whole token tile's B fragments from LDS for its 32 rows.
//   <2, 8>:  512 rows x 128 tokens (-DTG_FORM_RT=2).  A B fragment serves two row-tile pairs, which halves the LDS traffic
//            per MFMA, but a weight is dequantised once per 128 tokens.  Measured on one box, 8192 tokens: 32 experts
//            gate_up 1131 against 1077 us (0.385 against 0.404 of the MFMA peak), 128 experts (about 256 tokens per
//            expert, where its short last tiles are cheaper) 1291 against 1301 us: the dequantisation's instructions,
//            not the B reads, are what the loop is short of.
#ifndef TG_FORM_RT
#define TG_FORM_RT 1
#endif
constexpr int TG_DEFAULT_RT = TG_FORM_RT;
#ifndef TG_FORM_NTT
#define TG_FORM_NTT (16 / TG_FORM_RT)
#endif
constexpr int TG_DEFAULT_NTT = TG_FORM_NTT;
template <int RT, int NTT>
struct TileForm {
    static constexpr int ROWS = 16 * NTT;                        // tokens per token tile
    static constexpr int XBUF = ROWS * 128;                      // one activation buffer: ROWS tokens x 64 columns fp16
    static constexpr int XI = ROWS / 8 / TG_WAVES;               // activation DMA instructions (8 tokens each) per wave and block
    static constexpr int WBUF = TG_WAVES * RT * 1024;            // one packed-weight slot: a KiB per pair
    static constexpr int SBUF = TG_WAVES * RT * 256;             // one absmax slot: a dword per lane and pair
    static constexpr int X_OFF = TG_LUT_BYTES;
    static constexpr int W_OFF = X_OFF + 2 * XBUF;
    static constexpr int S_OFF = W_OFF + 2 * WBUF;
    static constexpr int TAB_OFF = S_OFF + 2 * SBUF;             // one expert's absmax table (byte absmax)
    static constexpr int LDS = TAB_OFF + 1024;
    static constexpr int PIECES = XI + 2 * RT;                   // DMA instructions per wave and iteration
    static constexpr int DW = 4 * RT;                            // packed dwords a wave dequantises per block
    static_assert(ROWS % (8 * TG_WAVES) == 0 && NTT % DW == 0 && PIECES <= NTT && LDS <= 160 * 1024, "tile form");
};
#ifndef TG_LA
#define TG_LA 1   // 16-token tiles of B fragments requested ahead of the MFMAs that use them
#endif

static __device__ const float nf4_codes[16] = M355_NF4_TABLE;

typedef const half8_t __attribute__((address_space(3))) lds_h8_t;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// 8 weights of one packed dword: four table lookups (address = one v_perm_b32: byte k of d above the lane's replica
// offset) fetch the two fp16 codes of each byte; then one packed fp16 multiply per byte by the block's absmax (rounded
// to fp16).  Two steps so that the lookups can be issued ahead of other LDS reads and the arithmetic placed where their
// results have arrived.  w = fp16(fp16(code) * fp16(absmax)): three fp16 roundings where the reference's dequantiser has
// one (fp16(code * absmax), patches/streaming_unsloth_bnb.py:192-222), each below 2^-11 relative; the fp32 product and
// its conversion cost twice the vector instructions and 5 % of the kernel.
typedef const uint32_t __attribute__((address_space(3))) lds_u32c_t;
__device__ __forceinline__ void lookup8(uint32_t lbase, uint32_t d, uint32_t* o) {
    uint32_t a0 = __builtin_amdgcn_perm(d, lbase, 0x03020400u), a1 = __builtin_amdgcn_perm(d, lbase, 0x03020500u);
    uint32_t a2 = __builtin_amdgcn_perm(d, lbase, 0x03020600u), a3 = __builtin_amdgcn_perm(d, lbase, 0x03020700u);
    o[0] = *reinterpret_cast<lds_u32c_t*>(a0);
    o[1] = *reinterpret_cast<lds_u32c_t*>(a1);
    o[2] = *reinterpret_cast<lds_u32c_t*>(a2);
    o[3] = *reinterpret_cast<lds_u32c_t*>(a3);
}
__device__ __forceinline__ half8_t scale8(const uint32_t* o, float s) {
    const half2_t hs = {(half_t)s, (half_t)s};
    const half2_t h0 = as_half2(o[0]) * hs, h1 = as_half2(o[1]) * hs, h2 = as_half2(o[2]) * hs, h3 = as_half2(o[3]) * hs;
    return half8_t{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
}
__device__ __forceinline__ half8_t dequant8(uint32_t lbase, uint32_t d, float s) {
    uint32_t f[4];
    lookup8(lbase, d, f);
    return scale8(f, s);
}

__device__ __forceinline__ float act_pair(float gate_acc, float up_acc, const StageParams& sp) {
    if (sp.act_mode == M355_ACT_SWIGLU_OAI) {
        const float gt = fminf((float)(half_t)gate_acc, sp.limit);   // gate and up leave the GEMM as fp16
        const float up = fminf(fmaxf((float)(half_t)up_acc, -sp.limit), sp.limit);
        const float glu = gt / (1.0f + expf(-sp.alpha * gt));
        return (up + 1.0f) * glu;
    }
    const float gt = (float)(half_t)gate_acc;
    const float up = (float)(half_t)up_acc;
    const half_t sg = (half_t)(gt / (1.0f + expf(-gt)));
    return (float)sg * up;
}

template <int MODE, int RT, int NTT>
__global__ __launch_bounds__(TG_THREADS) void moe_tile_gemm_kernel(
    const half_t* __restrict__ xin, const uint8_t* __restrict__ wq, const float* __restrict__ wam,
    const uint8_t* __restrict__ wam8, const float* __restrict__ wtab, const int32_t* __restrict__ n_tiles,
    const int32_t* __restrict__ tile_expert, const int32_t* __restrict__ tile_m0, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ sorted, int K, int R, int k_top, half_t* __restrict__ outp, StageParams sp, int n_rg) {
    using F = TileForm<RT, NTT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // blocks of one token tile share blockIdx % 8 (one XCD's L2 under round-robin placement serves the tile's
    // activations to all its row groups; placement affects speed only)
    const int slot = blockIdx.x >> 3;
    const int tt = (slot / n_rg) * 8 + (blockIdx.x & 7);
    const int rg = slot - (slot / n_rg) * n_rg;
    if (tt >= *n_tiles) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = K / 64, NP = R / 32;
    const int e = tile_expert[tt];
    const int pos0 = offsets[e] + tile_m0[tt];
    const int mt = min(F::ROWS, offsets[e + 1] - pos0);
    const int ntt = (mt + 15) >> 4;                         // live 16-token tiles
    const int p0 = (rg * TG_WAVES + wave) * RT;             // this wave's row-tile pairs: p0 .. p0 + RT - 1
    const int r = lane & 15, s = (lane >> 4) & 1, g = lane >> 4, n32 = lane & 31;
    // where a 16-byte chunk of a token's 128-byte line sits in LDS: position = chunk ^ key(token & 15), the key chosen so
    // that the 16 lanes a ds_read_b128 serves together (tokens 0..15 once each, column groups g or g ^ 1 by token) touch
    // every bank once
    auto xkey = [](int tok) { return (((tok >> 3) & 1) << 2) | ((tok >> 1) & 1); };

    // ---- everything that comes from global memory goes to LDS by DMA (global_load_lds), so no vector register waits
    // for memory and the only waits are the vmcnt(0) + barrier at the top of an iteration ----
    // activations: per wave and block XI instructions of 8 tokens x 128 bytes (whole lines); the 16-byte chunk that
    // lane (token, c') deposits at chunk position c' of the token's LDS line is source chunk c' ^ key(token)
    const half_t* xsrc[F::XI];
#pragma unroll
    for (int j = 0; j < F::XI; j++) {
        const int m = 8 * (F::XI * wave + j) + (lane >> 3);
        const int pos = pos0 + min(m, mt - 1);
        const int64_t src_row = (MODE == 0) ? (int64_t)(sorted[pos] / k_top) : (int64_t)pos;
        xsrc[j] = xin + src_row * sp.x_stride + (((lane & 7) ^ xkey(m)) << 3);
    }
    // B fragment of K-step t (32 columns: this lane's are 16g + 8t .. + 7), token r of a 16-token tile: chunk 2g + t
    uint32_t boff[2];
#pragma unroll
    for (int t = 0; t < 2; t++) boff[t] = F::X_OFF + r * 128 + (((2 * g + t) ^ xkey(r)) << 4);

    // weights: a packed KiB per pair and block; absmax: the dword that holds the lane's row (4 code bytes of rows
    // 4 (r >> 2).. for byte absmax, the row's fp32 otherwise).  Pairs past the end re-read the last one, results dropped.
    const size_t wq_expert = (size_t)R * K / 2, am_expert = (size_t)R * NB;
    const bool q8 = wam8 != nullptr;
    const int sstride = q8 ? 4 : 16;   // bytes per block
    const uint8_t* wsrc[RT];
    const uint8_t* ssrc[RT];
#pragma unroll
    for (int i = 0; i < RT; i++) {
        const int pc = min(p0 + i, NP - 1);
        wsrc[i] = wq + (size_t)e * wq_expert + (size_t)pc * NB * 1024 + lane * 16;
        const size_t am_row = ((size_t)(pc * 2 + s) * 4 + (r >> 2)) * NB * 4;
        ssrc[i] = q8 ? wam8 + (size_t)e * am_expert + am_row
                     : reinterpret_cast<const uint8_t*>(wam + (size_t)e * am_expert + am_row + (r & 3));
    }
    float* ltab = reinterpret_cast<float*>(smem + F::TAB_OFF);   // this expert's absmax table (byte absmax)
    const uint32_t wlds = F::W_OFF + wave * RT * 1024 + lane * 16;
    const uint32_t slds = F::S_OFF + wave * RT * 256 + r * 4 + (q8 ? (r & 3) : 0);   // row r of a pair's first tile; + 64: of its partner

    // DMA piece k of an iteration: pieces 0 .. XI - 1 bring activations of block bx (8 tokens each), then per pair
    // one piece for the packed weights and one for the absmax of block bw
    auto dma_piece = [&](auto full_tag, int k, int bx, int bw) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int bxc = min(bx, NB - 1), bwc = min(bw, NB - 1);
        if (k < F::XI) {
            if (FULL || 8 * (F::XI * wave + k) < mt)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(xsrc[k] + 64 * bxc),
                                                 (lds_void_t*)(smem + F::X_OFF + (bx & 1) * F::XBUF + (F::XI * wave + k) * 1024), 16, 0, 0);
        } else if (k < F::PIECES) {
            const int i = (k - F::XI) >> 1;
            if ((k - F::XI) & 1)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(ssrc[i] + (size_t)bwc * sstride),
                                                 (lds_void_t*)(smem + F::S_OFF + (bw & 1) * F::SBUF + (wave * RT + i) * 256), 4, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((glb_void_t*)(wsrc[i] + (size_t)bwc * 1024),
                                                 (lds_void_t*)(smem + F::W_OFF + (bw & 1) * F::WBUF + (wave * RT + i) * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](auto full_tag, int bx, int bw) {
#pragma unroll
        for (int k = 0; k < F::PIECES; k++) dma_piece(full_tag, k, bx, bw);
    };
    auto packed_of = [&](int bw, int i) -> u32x4_t {
        return *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(wlds + (uint32_t)((bw & 1) * F::WBUF + i * 1024));
    };
    auto scale_of = [&](int bw, int i, int half) -> float {
        const uint32_t a = slds + (uint32_t)((bw & 1) * F::SBUF + i * 256 + half * 64);
        if (q8) return ltab[*reinterpret_cast<const __attribute__((address_space(3))) uint8_t*>(a)];
        return *reinterpret_cast<const __attribute__((address_space(3))) float*>(a);
    };

    // code table: entry (byte v, replica) = (code[v & 15], code[v >> 4]) as two fp16
    for (int i = tid; i < 256 * 32; i += TG_THREADS) {
        const int v = i >> 5;
        const half2_t hv = {(half_t)nf4_codes[v & 15], (half_t)nf4_codes[v >> 4]};
        *reinterpret_cast<uint32_t*>(smem + v * 256 + (i & 31) * 4) = __builtin_bit_cast(uint32_t, hv);
    }
    if (q8 && tid < 256) ltab[tid] = wtab[(size_t)e * 256 + tid];
    const uint32_t lbase = (uint32_t)(n32 * 4);   // the table starts at LDS address 0 (no static LDS in this kernel)

    float4_t accA[RT][NTT], accB[RT][NTT];   // rows 4g .. 4g + 3 of pair i's first / partner row tile, token r of tile c
#pragma unroll
    for (int i = 0; i < RT; i++)
#pragma unroll
        for (int c = 0; c < NTT; c++) {
            accA[i][c] = float4_t{0.f, 0.f, 0.f, 0.f};
            accB[i][c] = float4_t{0.f, 0.f, 0.f, 0.f};
        }

    // Pipeline.  Iteration b multiplies block b (A fragments a_use[pair]: first tile K-steps 0, 1, partner tile K-steps
    // 0, 1; activations in buffer b & 1) and meanwhile turns packed block b + 1 (LDS slot (b + 1) & 1, brought in during
    // iteration b - 1) into a_fill, a dword at a time between the MFMAs of the token tiles, while the DMA brings
    // activations b + 1 and weights b + 2.  The B fragments of token tile c + 1 are requested before the MFMAs of tile c
    // are issued (two register sets).  Iterations come in pairs so that the two A-fragment sets keep static names.
    half8_t afA[RT][4], afB[RT][4];
    auto main_loop = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;   // all token tiles live: no per-tile tests
        stage(full_tag, 0, 0);
        stage(full_tag, 0, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tables complete, block 0 landed
#pragma unroll
        for (int i = 0; i < RT; i++) {
            const u32x4_t wv = packed_of(0, i);
            const float s0 = scale_of(0, i, 0), s1 = scale_of(0, i, 1);
            afA[i][0] = dequant8(lbase, wv[0], s0);
            afA[i][1] = dequant8(lbase, wv[1], s0);
            afA[i][2] = dequant8(lbase, wv[2], s1);
            afA[i][3] = dequant8(lbase, wv[3], s1);
        }
        auto iteration = [&](int b, half8_t (*a_use)[4], half8_t (*a_fill)[4]) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            u32x4_t wv[RT];
            float sc[RT][2];
#pragma unroll
            for (int i = 0; i < RT; i++) {
                wv[i] = packed_of(b + 1, i);
                sc[i][0] = scale_of(b + 1, i, 0);
                sc[i][1] = scale_of(b + 1, i, 1);
            }
            const uint32_t xb = (uint32_t)((b & 1) * F::XBUF);
            lds_h8_t* bp[2];
#pragma unroll
            for (int t = 0; t < 2; t++) bp[t] = reinterpret_cast<lds_h8_t*>(boff[t] + xb);
            uint32_t lf[4];   // table values of the dword being dequantised
            // B fragments of TG_LA token tiles ahead are requested before a tile's MFMAs are issued (TG_LA + 1 register sets)
            half8_t bf[TG_LA + 1][2];
            constexpr int EVERY = NTT / F::DW;              // a packed dword of the next block with every EVERY-th token tile
            constexpr int PHASE = EVERY > 1 ? 1 : 0;
#define TG_READ(C)                                                                                              \
    if constexpr ((C) < NTT) {                                                                                  \
        if (FULL || (C) < ntt) {                                                                                \
            _Pragma("unroll") for (int t = 0; t < 2; t++) bf[(C) % (TG_LA + 1)][t] = *(bp[t] + (C) * 128);      \
        }                                                                                                       \
    }
#define TG_MMA(C)                                                                                               \
    if constexpr ((C) < NTT) {                                                                                  \
        if (FULL || (C) < ntt) {                                                                                \
            _Pragma("unroll") for (int i = 0; i < RT; i++) {                                                    \
                accA[i][C] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_use[i][0], bf[(C) % (TG_LA + 1)][0], accA[i][C], 0, 0, 0); \
                accB[i][C] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_use[i][2], bf[(C) % (TG_LA + 1)][0], accB[i][C], 0, 0, 0); \
                accA[i][C] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_use[i][1], bf[(C) % (TG_LA + 1)][1], accA[i][C], 0, 0, 0); \
                accB[i][C] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_use[i][3], bf[(C) % (TG_LA + 1)][1], accB[i][C], 0, 0, 0); \
            }                                                                                                   \
        }                                                                                                       \
    }
            /* One dword (8 weights) of the next block with every EVERY-th token tile: its table lookups go out ahead */ \
            /* of the B reads, the arithmetic sits among the tile's MFMAs; one DMA piece follows each of the first tiles. */
#define TG_LUT(C)                                                                                               \
    if constexpr ((C) < NTT && (C) % EVERY == PHASE) lookup8(lbase, wv[((C) / EVERY) >> 2][((C) / EVERY) & 3], lf);
#define TG_MATH(C)                                                                                              \
    if constexpr ((C) < NTT && (C) % EVERY == PHASE)                                                            \
        a_fill[((C) / EVERY) >> 2][((C) / EVERY) & 3] = scale8(lf, sc[((C) / EVERY) >> 2][(((C) / EVERY) & 3) >> 1]);
#define TG_DMA(C) if constexpr ((C) < F::PIECES) dma_piece(full_tag, (C), b + 1, b + 2);
#define TG_SB __builtin_amdgcn_sched_barrier(0);
#define TG_STEP(C) TG_SB TG_LUT(C) TG_READ((C) + TG_LA) TG_SB TG_MMA(C) TG_MATH(C) TG_DMA(C)
            TG_SB TG_READ(0)
#if TG_LA >= 2
            TG_SB TG_READ(1)
#endif
#if TG_LA >= 3
            TG_SB TG_READ(2)
#endif
            TG_STEP(0) TG_STEP(1) TG_STEP(2) TG_STEP(3) TG_STEP(4) TG_STEP(5) TG_STEP(6) TG_STEP(7)
            TG_STEP(8) TG_STEP(9) TG_STEP(10) TG_STEP(11) TG_STEP(12) TG_STEP(13) TG_STEP(14) TG_STEP(15)
            TG_SB
#undef TG_STEP
#undef TG_LUT
#undef TG_MATH
#undef TG_DMA
#undef TG_SB
#undef TG_READ
#undef TG_MMA
        };
        int b = 0;
        for (; b + 1 < NB; b += 2) {
            iteration(b, afA, afB);
            iteration(b + 1, afB, afA);
        }
        if (b < NB) iteration(b, afA, afB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last iterations' DMA re-reads the last block)
    };
    // A wave whose pairs all lie past the last one (the last row group of R = 32 * NP rows when NP is no multiple of the
    // pairs per workgroup) brings its share of the activations and keeps the barriers, nothing else: the workgroup then
    // runs at the pace of its live waves.
    auto idle_loop = [&](auto full_tag) {
#pragma unroll
        for (int k = 0; k < F::XI; k++) dma_piece(full_tag, k, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int b = 0; b < NB; b++) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int k = 0; k < F::XI; k++) dma_piece(full_tag, k, b + 1, b + 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (p0 >= NP) {
        if (ntt == NTT) idle_loop(std::true_type{});
        else idle_loop(std::false_type{});
        return;
    }
    if (ntt == NTT) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    // ---- epilogue: register q of lane (r, g) is token r of the 16-token tile, row 4g + q of pair p's first tile
    // (accA: row 16p + 4g + q) and of its partner (accB: row 16 (p + NP) + 4g + q)
#pragma unroll
    for (int i = 0; i < RT; i++) {
        const int p = p0 + i;
        if (p >= NP) break;
        float ba[4], bb[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            ba[q] = sp.bias ? (float)sp.bias[(int64_t)e * R + 16 * p + 4 * g + q] : 0.f;
            bb[q] = sp.bias ? (float)sp.bias[(int64_t)e * R + 16 * (p + NP) + 4 * g + q] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < NTT; c++) {
            const int m = 16 * c + r;
            if (m < mt) {
                const int64_t orow = pos0 + m;
                half4_t va, vb;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float a = accA[i][c][q] + ba[q], bq = accB[i][c][q] + bb[q];
                    if (MODE == 0) {
                        va[q] = (half_t)act_pair(a, bq, sp);
                    } else {
                        va[q] = (half_t)a;
                        vb[q] = (half_t)bq;
                    }
                }
                if (MODE == 0) {
                    *reinterpret_cast<half4_t*>(outp + orow * (R / 2) + 16 * p + 4 * g) = va;
                } else {
                    *reinterpret_cast<half4_t*>(outp + orow * R + 16 * p + 4 * g) = va;
                    *reinterpret_cast<half4_t*>(outp + orow * R + 16 * (p + NP) + 4 * g) = vb;
                }
            }
        }
    }
}

using DefaultForm = TileForm<TG_DEFAULT_RT, TG_DEFAULT_NTT>;

int tile_gemm_rows() { return DefaultForm::ROWS; }

int tile_gemm_device_setup() {
    if (int rc = hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_tile_gemm_kernel<0, TG_DEFAULT_RT, TG_DEFAULT_NTT>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, DefaultForm::LDS), "hipFuncSetAttribute"))
        return rc;
    return hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(&moe_tile_gemm_kernel<1, TG_DEFAULT_RT, TG_DEFAULT_NTT>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, DefaultForm::LDS), "hipFuncSetAttribute");
}

int launch_tile_gemm(int mode, const half_t* xin, const uint8_t* wq, const float* wam, const uint8_t* wam8,
                     const float* wtab, const MoeWorkspace& ws, int K, int R, int k_top, int max_tiles, half_t* outp,
                     const StageParams& sp, hipStream_t st) {
    const int NP = R / 32;
    const int n_rg = (NP + TG_WAVES * TG_DEFAULT_RT - 1) / (TG_WAVES * TG_DEFAULT_RT);
    const int64_t blocks = (int64_t)8 * ((max_tiles + 7) / 8) * n_rg;
    if (blocks <= 0 || blocks > 0x7fffffff) return fail(-2, "m355_moe_nf4_forward: %lld workgroups", (long long)blocks);
    if (mode == 0)
        moe_tile_gemm_kernel<0, TG_DEFAULT_RT, TG_DEFAULT_NTT><<<dim3((unsigned)blocks), dim3(TG_THREADS), DefaultForm::LDS, st>>>(
            xin, wq, wam, wam8, wtab, ws.n_tiles, ws.tile_expert, ws.tile_m0, ws.offsets, ws.sorted, K, R, k_top, outp, sp, n_rg);
    else
        moe_tile_gemm_kernel<1, TG_DEFAULT_RT, TG_DEFAULT_NTT><<<dim3((unsigned)blocks), dim3(TG_THREADS), DefaultForm::LDS, st>>>(
            xin, wq, wam, wam8, wtab, ws.n_tiles, ws.tile_expert, ws.tile_m0, ws.offsets, ws.sorted, K, R, k_top, outp, sp, n_rg);
    M355_LAUNCH_CHECK("moe_tile_gemm_kernel");
    return 0;
}

}  // namespace m355
