// Dense projection y = x W^T + b (+ residual) for many tokens (prompts): MFMA-bound, fp16 in, fp32 accumulate.
//
// A workgroup owns 32 NTT tokens (NTT = 8 or 4 MFMA tiles of 32) times 256 rows of W (32 per wave) and walks K in
// 64-column blocks.  Both operands come into LDS by DMA (global_load_lds_dwordx4) in whole 128-byte lines, the
// 16-byte chunks XOR-swizzled on the source side so that the ds_read_b128 of a fragment is bank-conflict free:
//   * the activations of the token tile are shared by the 8 waves, one block ahead, two buffers;
//   * each wave brings its own 32 rows of W two blocks ahead and reads them (four ds_read_b128) into the A fragments of
//     the next block while it multiplies the current one; they are reused for every token of the tile.
// MFMA step t (v_mfma_f32_32x32x16_f16) takes columns 32 h + 8 t .. of the block from lane half h of both operands.
#include <type_traits>

#include "common.h"

namespace m355 {

typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int DT_WAVES = 8;
constexpr int DT_THREADS = DT_WAVES * 64;
constexpr int DT_WBUF = DT_WAVES * 32 * 128;   // one W slot: 32 rows x 64 columns fp16 per wave
constexpr int DT_WI = 4;                       // W DMA instructions (8 rows each) per wave and block
constexpr int dt_xbuf(int ntt) { return ntt * 32 * 128; }
constexpr int dt_lds(int ntt) { return 2 * dt_xbuf(ntt) + 2 * DT_WBUF; }

typedef const half8_t __attribute__((address_space(3))) dt_lds_h8_t;
typedef __attribute__((address_space(3))) void dt_lds_void_t;
typedef const __attribute__((address_space(1))) void dt_glb_void_t;

template <int NTT>
__global__ __launch_bounds__(DT_THREADS) void dense_tile_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                                const half_t* __restrict__ bias, const half_t* __restrict__ resid,
                                                                int T, int N, int K, half_t* __restrict__ out, int n_rg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int XBUF = dt_xbuf(NTT), W_OFF = 2 * XBUF;
    constexpr int XI = NTT * 32 / 8 / DT_WAVES;   // activation DMA instructions (8 tokens each) per wave and block
    constexpr int PIECES = XI + DT_WI;
    // workgroups of one token tile share blockIdx % 8 (one XCD's L2 serves the tile's activations to its row groups)
    const int slot = blockIdx.x >> 3;
    const int tt = (slot / n_rg) * 8 + (blockIdx.x & 7);
    const int rg = slot - (slot / n_rg) * n_rg;
    const int m0 = tt * 32 * NTT;
    if (m0 >= T) return;
    const int mt = min(32 * NTT, T - m0);
    const int ntt = (mt + 31) >> 5;   // live 32-token tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NB = K / 64;
    const int n32 = lane & 31, hh = lane >> 5;
    const int row0 = (rg * DT_WAVES + wave) * 32;   // this wave's rows of W (clamped to N - 1 when loading)

    // the lane that deposits chunk position c' of line m fetches source chunk c' ^ ((m >> 1) & 7)
    const half_t* xsrc[XI];
#pragma unroll
    for (int j = 0; j < XI; j++) {
        const int m = 8 * (XI * wave + j) + (lane >> 3);
        xsrc[j] = x + (int64_t)(m0 + min(m, mt - 1)) * K + (((lane & 7) ^ ((m >> 1) & 7)) << 3);
    }
    const half_t* wsrc[DT_WI];
#pragma unroll
    for (int j = 0; j < DT_WI; j++) {
        const int m = 8 * j + (lane >> 3);
        wsrc[j] = w + (int64_t)min(row0 + m, N - 1) * K + (((lane & 7) ^ ((m >> 1) & 7)) << 3);
    }
    // fragment of MFMA step t for line n32 (a token of a 32-token tile, or a row of the wave's 32): chunk 4 h + t
    uint32_t foff[4];
#pragma unroll
    for (int t = 0; t < 4; t++) foff[t] = n32 * 128 + (((4 * hh + t) ^ ((n32 >> 1) & 7)) << 4);

    auto dma_piece = [&](auto full_tag, int k, int bx, int bw) {
        constexpr bool FULL = decltype(full_tag)::value;
        const int bxc = min(bx, NB - 1), bwc = min(bw, NB - 1);
        if (k < XI) {
            if (FULL || 8 * (XI * wave + k) < mt)
                __builtin_amdgcn_global_load_lds((dt_glb_void_t*)(xsrc[k] + 64 * bxc),
                                                 (dt_lds_void_t*)(smem + (bx & 1) * XBUF + (XI * wave + k) * 1024), 16, 0, 0);
        } else if (k < PIECES) {
            __builtin_amdgcn_global_load_lds((dt_glb_void_t*)(wsrc[k - XI] + 64 * bwc),
                                             (dt_lds_void_t*)(smem + W_OFF + (bw & 1) * DT_WBUF + wave * 4096 + (k - XI) * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](auto full_tag, int bx, int bw) {
#pragma unroll
        for (int k = 0; k < PIECES; k++) dma_piece(full_tag, k, bx, bw);
    };

    float16_t acc[NTT];
#pragma unroll
    for (int c = 0; c < NTT; c++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[c][q] = 0.f;

    // Iteration b multiplies block b (A fragments a_use, activations in buffer b & 1), reads W block b + 1 (slot
    // (b + 1) & 1, brought in during iteration b - 1) into a_fill, and its DMA brings activations b + 1 and W block
    // b + 2.  The B fragments of token tile c + 1 are requested before the MFMAs of tile c are issued.
    half8_t afA[4], afB[4];
    auto main_loop = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        stage(full_tag, 0, 0);
        stage(full_tag, 0, 1);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; t++) afA[t] = *reinterpret_cast<dt_lds_h8_t*>(W_OFF + wave * 4096 + foff[t]);
        auto iteration = [&](int b, half8_t* a_use, half8_t* a_fill) {
            // every DMA of the last iteration has landed (this wave's: vmcnt; the others': barrier), and nobody still
            // reads what this iteration's DMA overwrites
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const uint32_t xb = (uint32_t)((b & 1) * XBUF);
            const uint32_t wa = (uint32_t)(W_OFF + ((b + 1) & 1) * DT_WBUF + wave * 4096);
            half8_t bf[2][4];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; t++) bf[0][t] = *reinterpret_cast<dt_lds_h8_t*>(xb + foff[t]);
#pragma unroll
            for (int c = 0; c < NTT; c++) {
                __builtin_amdgcn_sched_barrier(0);
                if (c < 4) a_fill[c] = *reinterpret_cast<dt_lds_h8_t*>(wa + foff[c]);
                if (c + 1 < NTT && (FULL || c + 1 < ntt)) {
#pragma unroll
                    for (int t = 0; t < 4; t++) bf[(c + 1) & 1][t] = *reinterpret_cast<dt_lds_h8_t*>(xb + (c + 1) * 4096 + foff[t]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (FULL || c < ntt) {
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_use[t], bf[c & 1][t], acc[c], 0, 0, 0);
                }
#pragma unroll
                for (int k = c; k < PIECES; k += NTT) dma_piece(full_tag, k, b + 1, b + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        int b = 0;
        for (; b + 1 < NB; b += 2) {
            iteration(b, afA, afB);
            iteration(b + 1, afB, afA);
        }
        if (b < NB) iteration(b, afA, afB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last iterations' DMA re-reads the last block)
    };
    if (ntt == NTT) main_loop(std::true_type{});
    else main_loop(std::false_type{});

    // accumulator register q of lane (n32, h): token n32 of the tile, row (q & 3) + 8 (q >> 2) + 4 h of the wave's 32.
    // y = fp16(acc + bias), then + residual in fp16: the roundings of the small-batch kernel.
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int col = row0 + 8 * g + 4 * hh;
        if (col >= N) continue;
        float bv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) bv[q] = bias ? (float)bias[col + q] : 0.f;
#pragma unroll
        for (int c = 0; c < NTT; c++) {
            const int m = 32 * c + n32;
            if (m < mt) {
                const int64_t o_at = (int64_t)(m0 + m) * N + col;
                half4_t o;
#pragma unroll
                for (int q = 0; q < 4; q++) o[q] = (half_t)(acc[c][4 * g + q] + bv[q]);
                if (resid) o = o + *reinterpret_cast<const half4_t*>(resid + o_at);
                *reinterpret_cast<half4_t*>(out + o_at) = o;
            }
        }
    }
}

static int dense_tile_device_setup_once(int dev) {
    static PerDeviceOnce once;
    return once.run(dev, [] {
        if (int rc = hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_tile_kernel<8>),
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, dt_lds(8)), "hipFuncSetAttribute"))
            return rc;
        return hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_tile_kernel<4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, dt_lds(4)), "hipFuncSetAttribute");
    });
}

int dense_tile_device_init(int dev) { return dense_tile_device_setup_once(dev); }

// Shapes were checked by the caller (K % 64 == 0, N % 16 == 0, 16-byte aligned x and w, 8-byte aligned bias/resid/out).
int launch_dense_tile(const half_t* x, const half_t* w, const half_t* bias, const half_t* resid, int T, int N, int K,
                      half_t* out, int cus, hipStream_t st) {
    int dev = 0;
    if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (int rc = dense_tile_device_setup_once(dev)) return rc;
    const int n_rg = (N + 255) / 256;
    // 256-token tiles reuse a row of W twice as often as 128-token tiles; take them when they fill the chip's rounds
    // at least as well (workgroups / CUs rounded up, times the tile's work)
    const int64_t wg8 = (int64_t)((T + 255) / 256) * n_rg, wg4 = (int64_t)((T + 127) / 128) * n_rg;
    const int64_t rounds8 = 2 * ((wg8 + cus - 1) / cus), rounds4 = (wg4 + cus - 1) / cus;
#ifdef DT_FORCE_NTT   /* diagnostic builds */
    const bool big = DT_FORCE_NTT == 8;
    (void)rounds8, (void)rounds4;
#else
    const bool big = rounds8 <= rounds4;
#endif
    const int64_t blocks = 8 * (((big ? (T + 255) / 256 : (T + 127) / 128) + 7) / 8) * (int64_t)n_rg;
    if (blocks > 0x7fffffff) return fail(-2, "m355_linear_f16: %lld workgroups", (long long)blocks);
    if (big)
        dense_tile_kernel<8><<<dim3((unsigned)blocks), dim3(DT_THREADS), dt_lds(8), st>>>(x, w, bias, resid, T, N, K, out, n_rg);
    else
        dense_tile_kernel<4><<<dim3((unsigned)blocks), dim3(DT_THREADS), dt_lds(4), st>>>(x, w, bias, resid, T, N, K, out, n_rg);
    M355_LAUNCH_CHECK("dense_tile_kernel");
    return 0;
}

}  // namespace m355
