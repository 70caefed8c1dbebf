// Fused NF4-dequant grouped GEMM for the MoE experts, decode shapes (few tokens per expert).
//
// HBM-bound: every call streams the packed 4-bit blocks of the experts that were hit exactly once
// (0.5 B per weight) plus their per-64 absmax; the arithmetic runs on the matrix cores with the
// tokens on the 16-wide N side of v_mfma_f32_16x16x32_f16, so 1..16 tokens per expert cost the same.
//
// Pipeline of one forward (all on the caller's stream):
//   moe_route_kernel      groups the (token, k) pairs by local expert (counts, offsets, sorted list)
//   moe_gemm_kernel<0>    h[pair] = silu(W13 x)[:FF] * (W13 x)[FF:]            (gate/up fused epilogue)
//   moe_gemm_kernel<1>    y[pair] = W2 h[pair]
//   moe_combine_kernel    out[t]  = sum_k w[t][k] * y[pair(t,k)]
//
// Weight tile order (written by m355_repack_nf4): for row-tile pair p and 64-column block b, one
// contiguous KiB; lane 16g+r owns 16 bytes = 8 bytes of row 16p+r (columns 64b+16g..+15) and 8 bytes of the
// partner row 16(p+P)+r, same columns.  One global_load_dwordx4 per lane therefore brings the A operands of
// four MFMAs (two K-steps for each of the two row tiles), every wave-load is 1 KiB contiguous, and because
// the two K-steps of a tile stay inside one 64-column block the block's absmax can be applied to the
// 16x16 partial product afterwards (4 FMAs per lane) instead of to every weight.
//
// Dequantisation: a byte holds two codes; a 256-entry table of packed half2 (code[lo], code[hi]) is
// replicated 32x in LDS (entry address = (byte*32 + lane%32)*4) so a wave's lookups never collide on a bank.
// The code values are fp16-rounded NF4 entries; scales stay fp32.
#include <hip/hip_ext.h>

#include "moe_shared.h"
#include "router_shared.h"

namespace m355 {

constexpr int MOE_THREADS = 512;  // 8 waves = 2 row-tile pairs x 4 K-parts
constexpr int MOE_PAIRS = 2;
constexpr int MOE_KPARTS = 4;
constexpr int MOE_CHUNK = 12;     // 64-column blocks (1 KiB wave-loads) a wave keeps in flight
constexpr int LUT_BYTES = 256 * 32 * 4;
constexpr int RED_BYTES = MOE_PAIRS * (MOE_KPARTS - 1) * 64 * 8 * 4;
constexpr int XS_PAD = 16;        // bytes of padding per staged activation row
constexpr int MOE_MAX_MT = 16;

#ifdef M355_EXP_STAMPS   /* diagnostic build: cycle stamps of waves 0 and 4 of every workgroup of the streaming kernel */
__device__ unsigned long long m355_dbg[512 * 2 * 128];
#define M355_STAMP()                                                                                          \
    if ((wave & 3) == 0 && dbg_n < 124) {                                                                     \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
        if (lane == 0) m355_dbg[((size_t)(blockIdx.x & 511) * 2 + (wave >> 2)) * 128 + 4 + dbg_n] = t_;          \
        dbg_n++;                                                                                              \
    }
#else
#define M355_STAMP()
#endif

// ------------------------------------------------------------------------------------------------
// Routing: one workgroup buckets the pairs by local expert.
// ------------------------------------------------------------------------------------------------
constexpr int ROUTE_THREADS = 1024;
constexpr int ROUTE_MAX_E = 1024;

__global__ __launch_bounds__(ROUTE_THREADS) void moe_route_kernel(const int32_t* __restrict__ ids, int ids_stride, int k_top,
                                                                  int P, int E_local,
                                                                  int e_first, int32_t* __restrict__ n_active,
                                                                  int32_t* __restrict__ offsets,
                                                                  int32_t* __restrict__ active,
                                                                  int32_t* __restrict__ sorted,
                                                                  int32_t* __restrict__ inv, int mt_rows,
                                                                  int32_t* __restrict__ n_tiles,
                                                                  int32_t* __restrict__ tile_expert,
                                                                  int32_t* __restrict__ tile_m0) {
    __shared__ int cnt[ROUTE_MAX_E];
    __shared__ int cur[ROUTE_MAX_E];
    const int tid = threadIdx.x;
    for (int e = tid; e < E_local; e += ROUTE_THREADS) cnt[e] = 0;
    __syncthreads();
    // pair p = (token p / k_top, pick p % k_top); a token's picks are ids_stride int32 apart from the next token's
    for (int p = tid; p < P; p += ROUTE_THREADS) {
        const int e = ids[(int64_t)(p / k_top) * ids_stride + p % k_top] - e_first;
        if (e >= 0 && e < E_local) atomicAdd(&cnt[e], 1);
    }
    __syncthreads();
    if (tid < 64) {
        // wave 0: lane l owns experts [l*C, (l+1)*C); exclusive wave scans give where its lists start
        const int C = (E_local + 63) / 64;
        const int ea = min(tid * C, E_local), eb = min(ea + C, E_local);
        int s_cnt = 0, s_act = 0, s_tiles = 0;
        for (int e = ea; e < eb; e++) {
            s_cnt += cnt[e];
            s_act += cnt[e] > 0;
            s_tiles += (cnt[e] + mt_rows - 1) / mt_rows;
        }
        int run = s_cnt, na = s_act, nt = s_tiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a1 = __shfl_up(run, o, 64), a2 = __shfl_up(na, o, 64), a3 = __shfl_up(nt, o, 64);
            if (tid >= o) { run += a1; na += a2; nt += a3; }
        }
        if (tid == 63) {
            offsets[E_local] = run;
            *n_active = na;
            *n_tiles = nt;
        }
        run -= s_cnt; na -= s_act; nt -= s_tiles;   // exclusive
        for (int e = ea; e < eb; e++) {
            offsets[e] = run;
            cur[e] = run;
            if (cnt[e] > 0) active[na++] = e;
            for (int m0 = 0; m0 < cnt[e]; m0 += mt_rows) {
                tile_expert[nt] = e;
                tile_m0[nt++] = m0;
            }
            run += cnt[e];
        }
    }
    __syncthreads();
    for (int p = tid; p < P; p += ROUTE_THREADS) {
        const int e = ids[(int64_t)(p / k_top) * ids_stride + p % k_top] - e_first;
        int pos = -1;
        if (e >= 0 && e < E_local) {
            pos = atomicAdd(&cur[e], 1);
            sorted[pos] = p;
        }
        inv[p] = pos;
    }
}

// The same for a prompt's tens of thousands of pairs, over ROUTE_WGS workgroups in two launches: every workgroup counts
// its contiguous share of the pairs per expert (moe_route_count_kernel: one row of `hist` each); then every workgroup
// reads all rows, from which the list offsets (column sums, scanned) and its own first position in every expert's list
// (the rows above its own) follow, and places its pairs (moe_route_place_kernel; workgroup 0 also writes the tables).
// As in the single-workgroup kernel, the order of an expert's pairs within its list is whatever the atomics give: no
// result depends on it.
constexpr int ROUTE_WGS = 32;
constexpr int ROUTE_PARALLEL_MIN_PAIRS = 4096;

__global__ __launch_bounds__(ROUTE_THREADS) void moe_route_count_kernel(const int32_t* __restrict__ ids, int ids_stride, int k_top,
                                                                        int P, int E_local, int e_first,
                                                                        int32_t* __restrict__ hist) {
    __shared__ int cnt[ROUTE_MAX_E];
    const int tid = threadIdx.x;
    const int chunk = (P + ROUTE_WGS - 1) / ROUTE_WGS;
    const int pa = blockIdx.x * chunk, pb = min(pa + chunk, P);
    for (int e = tid; e < E_local; e += ROUTE_THREADS) cnt[e] = 0;
    __syncthreads();
    for (int p = pa + tid; p < pb; p += ROUTE_THREADS) {
        const int e = ids[(int64_t)(p / k_top) * ids_stride + p % k_top] - e_first;
        if (e >= 0 && e < E_local) atomicAdd(&cnt[e], 1);
    }
    __syncthreads();
    for (int e = tid; e < E_local; e += ROUTE_THREADS) hist[blockIdx.x * E_local + e] = cnt[e];
}

__global__ __launch_bounds__(ROUTE_THREADS) void moe_route_place_kernel(const int32_t* __restrict__ ids, int ids_stride, int k_top,
                                                                        int P, int E_local, int e_first,
                                                                        const int32_t* __restrict__ hist,
                                                                        int32_t* __restrict__ n_active,
                                                                        int32_t* __restrict__ offsets,
                                                                        int32_t* __restrict__ active,
                                                                        int32_t* __restrict__ sorted,
                                                                        int32_t* __restrict__ inv, int mt_rows,
                                                                        int32_t* __restrict__ n_tiles,
                                                                        int32_t* __restrict__ tile_expert,
                                                                        int32_t* __restrict__ tile_m0) {
    __shared__ int cnt[ROUTE_MAX_E];    // pairs of the expert over all workgroups
    __shared__ int cur[ROUTE_MAX_E];    // first the pairs of the expert in the workgroups above this one, then the running position
    const int tid = threadIdx.x;
    const bool scribe = blockIdx.x == 0;
    for (int e = tid; e < E_local; e += ROUTE_THREADS) {
        int tot = 0, above = 0;
#pragma unroll 8
        for (int g = 0; g < ROUTE_WGS; g++) {
            const int c = hist[g * E_local + e];
            tot += c;
            above += g < (int)blockIdx.x ? c : 0;
        }
        cnt[e] = tot;
        cur[e] = above;
    }
    __syncthreads();
    if (tid < 64) {
        // wave 0: lane l owns experts [l*C, (l+1)*C); exclusive wave scans give where its lists start
        const int C = (E_local + 63) / 64;
        const int ea = min(tid * C, E_local), eb = min(ea + C, E_local);
        int s_cnt = 0, s_act = 0, s_tiles = 0;
        for (int e = ea; e < eb; e++) {
            s_cnt += cnt[e];
            s_act += cnt[e] > 0;
            s_tiles += (cnt[e] + mt_rows - 1) / mt_rows;
        }
        int run = s_cnt, na = s_act, nt = s_tiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a1 = __shfl_up(run, o, 64), a2 = __shfl_up(na, o, 64), a3 = __shfl_up(nt, o, 64);
            if (tid >= o) { run += a1; na += a2; nt += a3; }
        }
        if (tid == 63 && scribe) {
            offsets[E_local] = run;
            *n_active = na;
            *n_tiles = nt;
        }
        run -= s_cnt; na -= s_act; nt -= s_tiles;   // exclusive
        for (int e = ea; e < eb; e++) {
            cur[e] += run;
            if (scribe) {
                offsets[e] = run;
                if (cnt[e] > 0) active[na] = e;
                for (int m0 = 0; m0 < cnt[e]; m0 += mt_rows) {
                    tile_expert[nt] = e;
                    tile_m0[nt++] = m0;
                }
            }
            na += cnt[e] > 0;
            run += cnt[e];
        }
    }
    __syncthreads();
    const int chunk = (P + ROUTE_WGS - 1) / ROUTE_WGS;
    const int pa = blockIdx.x * chunk, pb = min(pa + chunk, P);
    for (int p = pa + tid; p < pb; p += ROUTE_THREADS) {
        const int e = ids[(int64_t)(p / k_top) * ids_stride + p % k_top] - e_first;
        int pos = -1;
        if (e >= 0 && e < E_local) {
            pos = atomicAdd(&cur[e], 1);
            sorted[pos] = p;
        }
        inv[p] = pos;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused dequant GEMM
// ------------------------------------------------------------------------------------------------
// Broadcast lane (16*row + I)'s value to the 16 lanes of its row.
template <int I>
__device__ __forceinline__ float row_bcast(float v) {
    const int iv = __builtin_bit_cast(int, v);   // every lane is written; the zero "old" value folds away
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, iv, 0x150 + I, 0xf, 0xf, true));
}

// acc += (lane 16*row + I's value of s, broadcast over the 16-lane row) * p, as one VOP2 DPP instruction.
template <int I>
__device__ __forceinline__ void fmac_row_bcast(float& acc, float s, float p) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(p), "n"(I));
}

// acc += (lane 16*row + I's value of s) * p like fmac_row_bcast, for use AFTER `ready` has been computed by an
// instruction the compiler knows: the compiler places the wait states an MFMA result needs before a vector
// instruction reads it only for its own instructions, never for inline assembly, so p (an MFMA result) is safe to
// read here only because `ready` was derived from the same MFMA's result and this statement depends on it.
template <int I>
__device__ __forceinline__ float fmac_row_bcast_after(float acc, float s, float p, float ready) {
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(s), "v"(p), "n"(I), "v"(ready));
    return acc;
}

// Four table lookups with one v_perm_b32 each for the address: byte k of d lands in bits 8..15 above the lane's
// replica offset (low byte of lbase), so the table has 256 bytes per byte value (LUTP_BYTES in all) and must start
// at an LDS address that is a multiple of 64 KiB; bits 16..31 of lbase carry that start.  The kernels that use it
// declare no static LDS, so their dynamic LDS (and the table) starts at 0.
constexpr int LUTP_BYTES = 256 * 256;
typedef const uint32_t __attribute__((address_space(3))) lds_u32_t;
__device__ __forceinline__ uint32_t lds_load_u32(uint32_t addr) { return *reinterpret_cast<lds_u32_t*>(addr); }
__device__ __forceinline__ uint32_t lutp_lane_base(const unsigned char* lut_bytes, int lane) {
    typedef const unsigned char __attribute__((address_space(3))) lds_u8_t;
    return (uint32_t)(size_t)(lds_u8_t*)lut_bytes + (lane & 31) * 4;
}
__device__ __forceinline__ half8_t lut4p(uint32_t lbase, uint32_t d) {
    // the four addresses are formed first, in four registers, and the four reads follow: alternating one address
    // with one read through a single temporary register makes every instruction wait for the one before it
    uint32_t a0 = __builtin_amdgcn_perm(d, lbase, 0x03020400u), a1 = __builtin_amdgcn_perm(d, lbase, 0x03020500u);
    uint32_t a2 = __builtin_amdgcn_perm(d, lbase, 0x03020600u), a3 = __builtin_amdgcn_perm(d, lbase, 0x03020700u);
    asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    u32x4_t r;
    r[0] = lds_load_u32(a0);
    r[1] = lds_load_u32(a1);
    r[2] = lds_load_u32(a2);
    r[3] = lds_load_u32(a3);
    return __builtin_bit_cast(half8_t, r);
}

__device__ __forceinline__ half8_t lut4(const uint32_t* lut_lane, uint32_t d) {
    u32x4_t r;
    r[0] = lut_lane[(d & 0xffu) * 32];
    r[1] = lut_lane[((d >> 8) & 0xffu) * 32];
    r[2] = lut_lane[((d >> 16) & 0xffu) * 32];
    r[3] = lut_lane[(d >> 24) * 32];
    return __builtin_bit_cast(half8_t, r);
}

// MODE 0: xin = x [T][K] gathered by token, out = h [P][R/2] with the silu(gate)*up epilogue.
// MODE 1: xin = h [P][K] rows taken in sorted order, out = y [P][R].
// W is [R][K] per expert, in tile order.  grid = (ceil(R/32 / 2), slots); slot s serves active[s].
template <int MODE>
__global__ __launch_bounds__(MOE_THREADS) void moe_gemm_kernel(const half_t* __restrict__ xin,
                                                               const uint8_t* __restrict__ wq,
                                                               const float* __restrict__ wam,
                                                               const int32_t* __restrict__ n_active,
                                                               const int32_t* __restrict__ offsets,
                                                               const int32_t* __restrict__ active,
                                                               const int32_t* __restrict__ sorted, int K, int R,
                                                               int k_top, int mt_rows, half_t* __restrict__ outp,
                                                               int64_t x_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = blockIdx.y;
    if (slot >= *n_active) return;
    const int e = active[slot];
    const int pos0 = offsets[e];
    const int M = offsets[e + 1] - pos0;

    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + LUT_BYTES);
    unsigned char* xs = smem + LUT_BYTES + RED_BYTES;
    const int XS = K * 2 + XS_PAD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = wave >> 2, q = wave & 3;
    const int NB = K / 64;
    const int NP = R / 32;
    const int p = blockIdx.x * MOE_PAIRS + pl;
    const bool live = p < NP;
    const int pc = live ? p : NP - 1;   // idle waves still run the loads on a valid tile (results dropped)
    const int BQ = (NB + MOE_KPARTS - 1) / MOE_KPARTS;
    const int bq0 = min(q * BQ, NB), bq1 = min(bq0 + BQ, NB);
    const int n = lane & 15, g = lane >> 4;

    const u32x4_t* wtile = reinterpret_cast<const u32x4_t*>(wq + (size_t)e * ((size_t)R * K / 2)) + (size_t)pc * NB * 64 + lane;
    const float4_t* amA = reinterpret_cast<const float4_t*>(wam + (size_t)e * ((size_t)R * NB)) + ((size_t)(pc * 2 + 0) * 4 + g) * NB;
    const float4_t* amB = reinterpret_cast<const float4_t*>(wam + (size_t)e * ((size_t)R * NB)) + ((size_t)(pc * 2 + 1) * 4 + g) * NB;

    u32x4_t wreg[MOE_CHUNK];
    float4_t sA, sB;
    auto load_chunk = [&](int cb) {
#pragma unroll
        for (int i = 0; i < MOE_CHUNK; i++) {
            const int b = min(cb + i, NB - 1);
            wreg[i] = __builtin_nontemporal_load(wtile + (size_t)b * 64);
        }
        const int bs = min(cb + n, NB - 1);
        sA = amA[bs];
        sB = amB[bs];
    };
    load_chunk(bq0);   // weights are on their way while the table and the activations are set up
    bool preloaded = true;

    // code table: thread t fills 16 of the 32 replicas of byte value t/2
    {
        const float tab[16] = M355_NF4_TABLE;
        const int bv = tid >> 1;
        const half2_t hv = {(half_t)tab[bv & 15], (half_t)tab[bv >> 4]};
        const uint32_t v = __builtin_bit_cast(uint32_t, hv);
#pragma unroll
        for (int i = 0; i < 16; i++) lut[bv * 32 + (tid & 1) * 16 + i] = v;
    }
    const uint32_t* lut_lane = lut + (lane & 31);

    for (int m0 = 0; m0 < M; m0 += mt_rows) {
        const int mt = min(mt_rows, M - m0);
        // stage the activation rows of this token tile
        {
            const int cpr = K / 8;   // 16-byte pieces per row
            for (int idx = tid; idx < mt * cpr; idx += MOE_THREADS) {
                const int row = idx / cpr, c = idx - row * cpr;
                const int pos = pos0 + m0 + row;
                const int64_t src_row = (MODE == 0) ? (int64_t)(sorted[pos] / k_top) : (int64_t)pos;
                const u32x4_t v = reinterpret_cast<const u32x4_t*>(xin + src_row * x_stride)[c];
                *reinterpret_cast<u32x4_t*>(xs + row * XS + c * 16) = v;
            }
        }
        __syncthreads();

        float4_t accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
        const unsigned char* xrow = xs + min(n, mt - 1) * XS + g * 32;
        for (int cb = bq0; cb < bq1; cb += MOE_CHUNK) {
            if (!preloaded) load_chunk(cb);
            preloaded = false;
#define M355_BLOCK(I)                                                                                       \
    if (cb + I < bq1) {                                                                                     \
        const unsigned char* xb = xrow + (cb + I) * 128;                                                    \
        const half8_t b0 = *reinterpret_cast<const half8_t*>(xb);                                           \
        const half8_t b1 = *reinterpret_cast<const half8_t*>(xb + 16);                                      \
        const u32x4_t wv = wreg[I];                                                                         \
        const float4_t z = {0.f, 0.f, 0.f, 0.f};                                                            \
        float4_t pA = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut4(lut_lane, wv[0]), b0, z, 0, 0, 0);        \
        float4_t pB = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut4(lut_lane, wv[2]), b0, z, 0, 0, 0);        \
        pA = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut4(lut_lane, wv[1]), b1, pA, 0, 0, 0);                \
        pB = __builtin_amdgcn_mfma_f32_16x16x32_f16(lut4(lut_lane, wv[3]), b1, pB, 0, 0, 0);                \
        _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                     \
            accA[r] += row_bcast<I>(sA[r]) * pA[r];                                                         \
            accB[r] += row_bcast<I>(sB[r]) * pB[r];                                                         \
        }                                                                                                   \
    }
            M355_BLOCK(0) M355_BLOCK(1) M355_BLOCK(2) M355_BLOCK(3) M355_BLOCK(4) M355_BLOCK(5)
            M355_BLOCK(6) M355_BLOCK(7) M355_BLOCK(8) M355_BLOCK(9) M355_BLOCK(10) M355_BLOCK(11)
#undef M355_BLOCK
#undef M355_FIRST_FETCH
        }

        // sum the K-parts: parts 1..3 park their tiles in LDS, part 0 adds them and writes the result
        if (q > 0) {
            float4_t* dst = reinterpret_cast<float4_t*>(red) + ((size_t)(pl * (MOE_KPARTS - 1) + (q - 1)) * 64 + lane) * 2;
            dst[0] = accA;
            dst[1] = accB;
        }
        __syncthreads();
        if (q == 0 && live) {
#pragma unroll
            for (int qq = 0; qq < MOE_KPARTS - 1; qq++) {
                const float4_t* src = reinterpret_cast<const float4_t*>(red) + ((size_t)(pl * (MOE_KPARTS - 1) + qq) * 64 + lane) * 2;
                accA += src[0];
                accB += src[1];
            }
            if (n < mt) {
                const int64_t orow = pos0 + m0 + n;
                if (MODE == 0) {
                    half4_t hv;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float gt = (float)(half_t)accA[r];   // gate and up leave the GEMM as fp16
                        const float up = (float)(half_t)accB[r];
                        const half_t s = (half_t)(gt / (1.0f + expf(-gt)));
                        hv[r] = (half_t)((float)s * up);
                    }
                    *reinterpret_cast<half4_t*>(outp + orow * (R / 2) + 16 * p + 4 * g) = hv;
                } else {
                    half4_t ya, yb;
#pragma unroll
                    for (int r = 0; r < 4; r++) { ya[r] = (half_t)accA[r]; yb[r] = (half_t)accB[r]; }
                    *reinterpret_cast<half4_t*>(outp + orow * R + 16 * p + 4 * g) = ya;
                    *reinterpret_cast<half4_t*>(outp + orow * R + 16 * (p + NP) + 4 * g) = yb;
                }
            }
        }
        __syncthreads();   // red and xs are reused by the next token tile
    }
}

// ------------------------------------------------------------------------------------------------
// Streaming form of the same GEMM for K <= 48 blocks (K <= 3072): a workgroup walks a contiguous range of
// (expert, row-tile-pair group) items; each wave holds the whole K-part of its item in registers (12 KiB
// per wave) and fetches the next item's bytes before it computes the current one, so a CU always has two
// items' worth of HBM reads in flight.  The code table is built once per workgroup and the activations are
// restaged only when the expert changes.  Barriers are raw s_barrier + lgkmcnt(0): a __syncthreads() would
// also wait for the prefetch.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Write one lane's 4 rows of both tiles for token row `orow`: h = act(gate, up) (MODE 0) or y (MODE 1), expert e.
template <int MODE>
__device__ __forceinline__ void gemm_epilogue(float4_t accA, float4_t accB, half_t* __restrict__ outp,
                                              int64_t orow, int R, int NP, int p, int g, int e, const StageParams& sp) {
    if (sp.bias) {   // rows 16p + 4g .. +3 of the first tile, the same rows of the partner tile
        const half4_t ba = *reinterpret_cast<const half4_t*>(sp.bias + (int64_t)e * R + 16 * p + 4 * g);
        const half4_t bb = *reinterpret_cast<const half4_t*>(sp.bias + (int64_t)e * R + 16 * (p + NP) + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; r++) { accA[r] += (float)ba[r]; accB[r] += (float)bb[r]; }
    }
    if (MODE == 0) {
        half4_t hv;
        if (sp.act_mode == M355_ACT_SWIGLU_OAI) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float gt = fminf((float)(half_t)accA[r], sp.limit);   // gate and up leave the GEMM as fp16
                const float up = fminf(fmaxf((float)(half_t)accB[r], -sp.limit), sp.limit);
                const float glu = gt / (1.0f + expf(-sp.alpha * gt));
                hv[r] = (half_t)((up + 1.0f) * glu);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float gt = (float)(half_t)accA[r];   // gate and up leave the GEMM as fp16
                const float up = (float)(half_t)accB[r];
                const half_t sg = (half_t)(gt / (1.0f + expf(-gt)));
                hv[r] = (half_t)((float)sg * up);
            }
        }
        *reinterpret_cast<half4_t*>(outp + orow * (R / 2) + 16 * p + 4 * g) = hv;
    } else {
        half4_t ya, yb;
#pragma unroll
        for (int r = 0; r < 4; r++) { ya[r] = (half_t)accA[r]; yb[r] = (half_t)accB[r]; }
        *reinterpret_cast<half4_t*>(outp + orow * R + 16 * p + 4 * g) = ya;
        *reinterpret_cast<half4_t*>(outp + orow * R + 16 * (p + NP) + 4 * g) = yb;
    }
}

// One output element of an item: row 16p + `row` of the first tile (sumA) and the same row of the partner tile (sumB) for
// token row `orow`: h = act(gate, up) (MODE 0) or the two y values (MODE 1), expert e.  Same arithmetic, element by
// element, as gemm_epilogue.
template <int MODE>
__device__ __forceinline__ void gemm_epilogue1(float sumA, float sumB, half_t* __restrict__ outp, int64_t orow, int R, int NP,
                                               int p, int row, int e, const StageParams& sp) {
    if (sp.bias) {
        sumA += (float)sp.bias[(int64_t)e * R + 16 * p + row];
        sumB += (float)sp.bias[(int64_t)e * R + 16 * (p + NP) + row];
    }
    if (MODE == 0) {
        half_t hv;
        if (sp.act_mode == M355_ACT_SWIGLU_OAI) {
            const float gt = fminf((float)(half_t)sumA, sp.limit);   // gate and up leave the GEMM as fp16
            const float up = fminf(fmaxf((float)(half_t)sumB, -sp.limit), sp.limit);
            const float glu = gt / (1.0f + expf(-sp.alpha * gt));
            hv = (half_t)((up + 1.0f) * glu);
        } else {
            const float gt = (float)(half_t)sumA;   // gate and up leave the GEMM as fp16
            const float up = (float)(half_t)sumB;
            const half_t sg = (half_t)(gt / (1.0f + expf(-gt)));
            hv = (half_t)((float)sg * up);
        }
        outp[orow * (R / 2) + 16 * p + row] = hv;
    } else {
        outp[orow * R + 16 * p + row] = (half_t)sumA;
        outp[orow * R + 16 * (p + NP) + row] = (half_t)sumB;
    }
}

// Down stage: rows `col` and `col + 1` (col even) of y for token row `orow` as one 4-byte store.  Same arithmetic,
// element by element, as gemm_epilogue.  `through`: the store goes through to memory (a device-scope store), for readers
// on other compute dies of the same launch.
__device__ __forceinline__ void gemm_epilogue_y2(float sum0, float sum1, half_t* __restrict__ outp, int64_t orow, int R,
                                                 int col, int e, const StageParams& sp, bool through) {
    if (sp.bias) {
        const half2_t b = *reinterpret_cast<const half2_t*>(sp.bias + (int64_t)e * R + col);
        sum0 += (float)b[0];
        sum1 += (float)b[1];
    }
    const half2_t v = {(half_t)sum0, (half_t)sum1};
    half_t* dst = outp + orow * R + col;
    if (through) store_device_scope(dst, v);
    else *reinterpret_cast<half2_t*>(dst) = v;
}

// Router logits of a decode batch on every CU: one wave per (token, 4 experts), so that no CU has to pull the whole
// router matrix through its own L1 (a workgroup per token does, and at a few dozen tokens that sets the router's time).
// The top-k of a token needs all its logits: the gate_up launch that follows picks it (INLINE == 2).
__global__ __launch_bounds__(64) void router_logits_kernel(const half_t* __restrict__ x, int64_t x_stride,
                                                           const half_t* __restrict__ wr, const half_t* __restrict__ br,
                                                           int H, int E, half_t* __restrict__ logits) {
    const int t = blockIdx.y;
    router_logits_wave(x + (int64_t)t * x_stride, wr, br, H, E, 4 * (int)blockIdx.x, nullptr, logits + (int64_t)t * E);
}

// What a gate_up launch needs to bucket the (token, k) pairs by expert itself (INLINE), instead of reading the lists
// a routing kernel left in the workspace: the raw expert ids, and where workgroup 0 leaves the lists for the kernels
// that follow (down GEMM, combine).
struct RouteIO {
    const int32_t* ids;
    int ids_stride;   // int32 between the picks of one token and the next (k: dense)
    int P, E_local, e_first;
    int32_t *n_active, *offsets, *active, *sorted, *inv, *n_tiles, *tile_expert, *tile_m0;
    // INLINE == 2: the logits [T][r_E] (fp16) were left by router_logits_kernel; every workgroup picks the top-k of r_k
    // itself, and workgroup 0 leaves ids and weights in ids_out / w_out for the combine and the caller
    const half_t* r_logits;
    int r_E, r_k;
    int32_t* ids_out;
    float* w_out;
};
constexpr int SPLIT_ROUTER_MAX_T = 24;    // tokens up to which gate_up picks the top-k from finished logits (beyond, the picking
                                          // in front of every workgroup's first load costs more than the router kernel's launch)
constexpr int SPLIT_ROUTER_MAX_E = 256;   // global experts: a row of logits per wave must fit the reduction buffers
constexpr int INLINE_MAX_P = 256;      // one pair per thread of the smallest workgroup
constexpr int INLINE_MAX_E = 128;      // two experts per lane of the scanning wave
constexpr int INLINE_MAX_TILES = 160;  // <= min(P, E) when T <= 16, else <= E + P / 16
struct InlineTables {                  // lives in LDS behind the reduction buffers for the whole kernel
    int32_t off[INLINE_MAX_E + 1];
    int32_t tile_e[INLINE_MAX_TILES];
    int32_t tile_m0[INLINE_MAX_TILES];
    int32_t tok[INLINE_MAX_P];         // sorted position -> pair index
    int32_t nt;
};

// Every workgroup buckets the pairs the same way (the result must not depend on timing: all of them address the
// same h rows): per expert a bit mask of its pairs, built with LDS atomic ORs; a pair's place within its expert
// is the number of lower-numbered pairs in the mask.  `masks` is scratch ([E_local][8] words).
__device__ __forceinline__ void inline_route(const RouteIO& rio, int my_e, int mt_rows, uint32_t* masks, InlineTables* rt,
                                             int tid, int threads) {
    const int E = rio.E_local;
    const bool scribe = blockIdx.x == 0;
    for (int i = tid; i < E * 8; i += threads) masks[i] = 0;
    __syncthreads();
    if (my_e >= 0) atomicOr(&masks[my_e * 8 + (tid >> 5)], 1u << (tid & 31));
    __syncthreads();
    if (tid < 64) {
        // wave 0: lane l owns experts 2l and 2l+1; exclusive wave scans give where their lists start
        int c[2], s_cnt = 0, s_act = 0, s_tiles = 0;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int e = 2 * tid + j;
            c[j] = 0;
            if (e < E) {
#pragma unroll
                for (int w = 0; w < 8; w++) c[j] += __builtin_popcount(masks[e * 8 + w]);
            }
            s_cnt += c[j];
            s_act += c[j] > 0;
            s_tiles += (c[j] + mt_rows - 1) / mt_rows;
        }
        int run = s_cnt, na = s_act, nt = s_tiles;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a1 = __shfl_up(run, o, 64), a2 = __shfl_up(na, o, 64), a3 = __shfl_up(nt, o, 64);
            if (tid >= o) { run += a1; na += a2; nt += a3; }
        }
        if (tid == 63) {
            rt->off[E] = run;
            rt->nt = nt;
            if (scribe) { rio.offsets[E] = run; *rio.n_active = na; *rio.n_tiles = nt; }
        }
        run -= s_cnt; na -= s_act; nt -= s_tiles;   // exclusive
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int e = 2 * tid + j;
            if (e < E) {
                rt->off[e] = run;
                if (scribe) { rio.offsets[e] = run; if (c[j] > 0) rio.active[na] = e; }
                na += c[j] > 0;
                for (int m0 = 0; m0 < c[j]; m0 += mt_rows) {
                    rt->tile_e[nt] = e;
                    rt->tile_m0[nt] = m0;
                    if (scribe) { rio.tile_expert[nt] = e; rio.tile_m0[nt] = m0; }
                    nt++;
                }
                run += c[j];
            }
        }
    }
    __syncthreads();
    if (tid < rio.P) {
        int pos = -1;
        if (my_e >= 0) {
            const int chunk = tid >> 5;
            int rank = __builtin_popcount(masks[my_e * 8 + chunk] & ((1u << (tid & 31)) - 1u));
            for (int w = 0; w < chunk; w++) rank += __builtin_popcount(masks[my_e * 8 + w]);
            pos = rt->off[my_e] + rank;
            rt->tok[pos] = tid;
            if (scribe) rio.sorted[pos] = tid;
        }
        if (scribe) rio.inv[tid] = pos;
    }
    __syncthreads();   // the lists are complete, and the mask area is free again
}

// The lists of a single token's k pairs, derived by ONE wave from registers: lane j < k holds the token's j-th pick
// (global expert id).  An expert holds at most one pair, so a pair's place is the number of picked local experts with a
// smaller id, and the tables are what inline_route() would leave.  Every wave of the workgroup calls this: all write the
// same values to the same LDS words and each reads only behind its own writes, so no barrier is needed; wave 0 of
// workgroup 0 also leaves the lists in the workspace for the launches that follow.
__device__ __forceinline__ void single_token_route(const RouteIO& rio, int id_j, InlineTables* rt, int lane, int wave) {
    const int k = rio.P;
    const bool pair = lane < k;
    const int e_j = id_j - rio.e_first;
    const bool local = pair && e_j >= 0 && e_j < rio.E_local;
    constexpr int OC = (INLINE_MAX_E + 1 + 63) / 64;
    int pos = 0, off[OC];
#pragma unroll
    for (int c = 0; c < OC; c++) off[c] = 0;
    for (int i = 0; i < k; i++) {
        const int e_i = __builtin_amdgcn_readlane(e_j, i);
        const bool local_i = __builtin_amdgcn_readlane((int)local, i) != 0;
        pos += (local_i && e_i < e_j) ? 1 : 0;
#pragma unroll
        for (int c = 0; c < OC; c++) off[c] += (local_i && e_i < lane + 64 * c) ? 1 : 0;
    }
    const int n_local = __builtin_popcountll(__ballot(local));
    const bool scribe = blockIdx.x == 0 && wave == 0;
    if (local) {
        rt->tile_e[pos] = e_j;
        rt->tile_m0[pos] = 0;
        rt->tok[pos] = lane;
        if (scribe) { rio.tile_expert[pos] = e_j; rio.tile_m0[pos] = 0; rio.sorted[pos] = lane; rio.active[pos] = e_j; }
    }
#pragma unroll
    for (int c = 0; c < OC; c++) {
        const int e = lane + 64 * c;
        if (e <= rio.E_local) {
            rt->off[e] = off[c];
            if (scribe) rio.offsets[e] = off[c];
        }
    }
    if (lane == 0) {
        rt->nt = n_local;
        if (scribe) { *rio.n_active = n_local; *rio.n_tiles = n_local; }
    }
    if (scribe && pair) rio.inv[lane] = local ? pos : -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 256 x 32 code table with 256-byte rows: entry (byte, replica) = half2(code[byte & 15], code[byte >> 4]).  Thread t
// writes replica t & 31 of the byte values whose low nibble is fixed by t >> 5, so its low code is chosen once (a select
// chain over immediates) and the high code of each store is a compile-time constant: stores only, no reads, one barrier.
template <int THREADS>
__device__ __forceinline__ void build_lutp(uint32_t* lut, int tid) {
    constexpr float tab[16] = M355_NF4_TABLE;
    constexpr int ROWS = THREADS / 32;          // byte values covered by one pass of the workgroup: 8 or 16
    static_assert(ROWS == 8 || ROWS == 16, "256 or 512 threads");
    const int row = tid >> 5;
    uint32_t* dst = lut + row * 64 + (tid & 31);
    float lo_a = 0.f, lo_b = 0.f;               // low codes of byte values row (+ 8 in every other pass when ROWS == 8)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        lo_a = (row == j) ? tab[j] : lo_a;
        lo_b = (row + 8 == j) ? tab[j] : lo_b;
    }
#pragma unroll
    for (int pass = 0; pass < 256 / ROWS; pass++) {
        const int bv0 = pass * ROWS;             // byte value = bv0 + row
        const float lo = (ROWS == 16 || (pass & 1) == 0) ? lo_a : lo_b;
        const half2_t hv = {(half_t)lo, (half_t)tab[bv0 >> 4]};
        dst[bv0 * 64] = __builtin_bit_cast(uint32_t, hv);
    }
    lds_barrier();
}

// INLINE 0: routing lists from the workspace, 1: pairs bucketed here, 2: top-k picked from finished logits, then bucketed.  KP: K-parts per row-tile pair (the
// waves that share an item's rows; each holds 48 / KP blocks of the item in registers).  NT: 16-token sets per token tile
// (a tile of 16 NT tokens shares one pass over the weights; NT = 2 needs the shorter K-parts of KP = 8 to keep both
// sets' activations in registers).
template <int MODE, int PAIRS, int INLINE, int KP = MOE_KPARTS, int NT = 1>
__global__ __launch_bounds__(PAIRS * KP * 64) void moe_gemm_stream_kernel(
    const half_t* __restrict__ xin, const uint8_t* __restrict__ wq, const float* __restrict__ wam,
    const int32_t* __restrict__ n_tiles, const int32_t* __restrict__ tile_expert, const int32_t* __restrict__ tile_m0,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ sorted, int K, int R, int k_top, int mt_rows,
    half_t* __restrict__ outp, int table_len, RouteIO rio, StageParams sp, const uint8_t* __restrict__ wam8,
    const float* __restrict__ wtab, CombineIO cio) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (MODE == 0) {
        // the column-group counters of the down launch that follows on the stream
        if (cio.pg_done && blockIdx.x == 0 && (int)threadIdx.x < SYNC_PG_COUNTERS) cio.pg_done[threadIdx.x] = 0;
    }
#ifdef M355_EXP_STAMPS
    const unsigned long long dbg_entry = __builtin_amdgcn_s_memtime();
    unsigned long long dbg_pro[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define M355_PRO_STAMP(K) dbg_pro[K] = __builtin_amdgcn_s_memtime();
#else
#define M355_PRO_STAMP(K)
#endif
    constexpr int THREADS = PAIRS * KP * 64;
    constexpr int CH = MOE_KPARTS * MOE_CHUNK / KP;                   // blocks per wave and item
    static_assert(CH == 12 || CH == 6, "the item body is written for 12 or 6 blocks per wave");
    static_assert(INLINE == 0 || (KP == MOE_KPARTS && NT == 1), "inline routing is built for the 8-wave, 16-token form");
    static_assert(KP == 4 * NT, "wave q finishes accumulator element q & 3 of token set q >> 2");
    constexpr int RED1 = PAIRS * KP * 64 * 8 * NT * 4;                // one reduction buffer: every wave's partial sums
    constexpr int RED_ALL = 2 * RED1;                                 // both
    uint32_t* lut = reinterpret_cast<uint32_t*>(smem);
    unsigned char* red = smem + LUTP_BYTES;
    InlineTables* rt = reinterpret_cast<InlineTables*>(red + RED_ALL);
    static_assert(INLINE != 2 || INLINE_MAX_P * 8 + PAIRS * KP * SPLIT_ROUTER_MAX_E * 4 <= RED_ALL, "picks and a row of logits per wave");
    int te_lane = 0, tm_lane = 0, of_lane = 0;
    const int off_len = min(64, rio.E_local + 1);
    if constexpr (INLINE) {
        // the ids are requested first; the code table is built later, while the first weights are on their way
        int my_e = -1;
        if constexpr (INLINE == 2) {
            // The logits come from router_logits_kernel (every CU, a wave per token and 4 experts).  Every workgroup picks
            // the top-k of all tokens itself, a wave taking every (PAIRS * KP)-th token, with the router kernel's own code
            // (same bits): up to 64 experts from a lane per expert, beyond from a row of LDS per wave.  The picks live in
            // the reduction buffers until the pairs are bucketed.
            const int Tn = rio.P / rio.r_k;
            float* wl = reinterpret_cast<float*>(red);
            int32_t* il = reinterpret_cast<int32_t*>(wl + rio.P);
            const int wv_ = (int)threadIdx.x >> 6, ln_ = (int)threadIdx.x & 63;
            constexpr int NW = PAIRS * KP;
            if (Tn == 1 && rio.r_E <= 64) {
                // A single token (the whole batch of single-sequence decode): every wave picks AND buckets for itself
                // (single_token_route): no barrier in front of the first load.
                const float mine = ln_ < rio.r_E ? (float)rio.r_logits[ln_] : 0.f;
                topk_softmax_wave64(mine, rio.r_E, rio.r_k, il, wl);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const bool pair = ln_ < rio.r_k;
                const int id_j = pair ? il[ln_] : 0;
                single_token_route(rio, id_j, rt, ln_, wv_);
                if (blockIdx.x == 0 && wv_ == 0 && pair) {
                    rio.ids_out[ln_] = id_j;
                    rio.w_out[ln_] = wl[ln_];
                }
            } else {
            if (rio.r_E <= 64) {
                constexpr int PER = 4;   // a wave's tokens four at a time, their logits requested together
                for (int t0 = wv_; t0 < Tn; t0 += PER * NW) {
                    float mine[PER];
#pragma unroll
                    for (int j = 0; j < PER; j++) {
                        const int t = t0 + j * NW;
                        mine[j] = (t < Tn && ln_ < rio.r_E) ? (float)rio.r_logits[(int64_t)t * rio.r_E + ln_] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < PER; j++) {
                        const int t = t0 + j * NW;
                        if (t < Tn) topk_softmax_wave64(mine[j], rio.r_E, rio.r_k, il + t * rio.r_k, wl + t * rio.r_k);
                    }
                }
            } else {
                float* lg = reinterpret_cast<float*>(il + rio.P) + wv_ * rio.r_E;   // (SPLIT_ROUTER_MAX_E floats per wave fit)
                for (int t = wv_; t < Tn; t += NW) {
                    for (int e = ln_; e < rio.r_E; e += 64) lg[e] = (float)rio.r_logits[(int64_t)t * rio.r_E + e];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    topk_softmax_wave(lg, rio.r_E, rio.r_k, il + t * rio.r_k, wl + t * rio.r_k);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            M355_PRO_STAMP(3)   /* picks made */
            __syncthreads();
            if ((int)threadIdx.x < rio.P) {
                my_e = il[threadIdx.x] - rio.e_first;
                if (blockIdx.x == 0) {
                    rio.ids_out[threadIdx.x] = il[threadIdx.x];
                    rio.w_out[threadIdx.x] = wl[threadIdx.x];
                }
            }
            // il / wl live in the buffer that inline_route() clears first (its per-expert masks): every read above
            // must have returned before any wave starts clearing
            __syncthreads();
            }
        } else if (rio.P == k_top) {
            // a single token with its picks given: every wave buckets them for itself, as above
            const int ln_ = (int)threadIdx.x & 63;
            single_token_route(rio, ln_ < k_top ? rio.ids[ln_] : 0, rt, ln_, (int)threadIdx.x >> 6);
        } else {
            if ((int)threadIdx.x < rio.P)
                my_e = rio.ids[(int64_t)((int)threadIdx.x / k_top) * rio.ids_stride + (int)threadIdx.x % k_top] - rio.e_first;
        }
        if (my_e >= rio.E_local) my_e = -1;
        // (a single token has bucketed its pairs above)
        if (!(INLINE == 2 ? (rio.P == rio.r_k && rio.r_E <= 64) : rio.P == k_top))
            inline_route(rio, my_e, mt_rows, reinterpret_cast<uint32_t*>(red), rt, threadIdx.x, THREADS);
    } else {
        // The first 64 entries of the tile -> expert table are requested right away, one per lane, together with
        // the tile count, so the first weight address is one memory round trip away instead of two; later lookups
        // of those entries are lane reads.
        te_lane = ((int)(threadIdx.x & 63) < table_len) ? tile_expert[threadIdx.x & 63] : 0;
        // likewise the first 64 tiles' first rows and the first 64 experts' list offsets (the table holds E_local + 1 of
        // them, at least table_len): the activations of the first item are then one trip behind this one, not two
        tm_lane = ((int)(threadIdx.x & 63) < table_len) ? tile_m0[threadIdx.x & 63] : 0;
        of_lane = ((int)(threadIdx.x & 63) < off_len) ? offsets[threadIdx.x & 63] : 0;
    }
    M355_PRO_STAMP(0)   /* routing done (or the tile table requested) */
#define M355_TILE_EXPERT(TT) (INLINE ? __builtin_amdgcn_readfirstlane(rt->tile_e[(TT)]) \
                                     : (((TT) < 64) ? __builtin_amdgcn_readlane(te_lane, (TT)) : tile_expert[(TT)]))
#define M355_TILE_M0(TT) (INLINE ? __builtin_amdgcn_readfirstlane(rt->tile_m0[(TT)]) \
                                 : (((TT) < 64) ? __builtin_amdgcn_readlane(tm_lane, (TT)) : tile_m0[(TT)]))
#define M355_OFFSET(E) (INLINE ? __builtin_amdgcn_readfirstlane(rt->off[(E)]) \
                               : (((E) < off_len) ? __builtin_amdgcn_readlane(of_lane, (E)) : offsets[(E)]))
#define M355_PAIR_AT(POS) (INLINE ? rt->tok[(POS)] : sorted[(POS)])
    const int NB = K / 64, NP = R / 32;
    const int NPG = (NP + PAIRS - 1) / PAIRS;
    // work item = (token tile of one expert, group of PAIRS row-tile pairs); a workgroup takes a contiguous range
    const int total = (INLINE ? __builtin_amdgcn_readfirstlane(rt->nt) : *n_tiles) * NPG;
    // (one 32-bit division: the first `total % grid` workgroups take one item more)
    const int per_wg = (int)((uint32_t)total / gridDim.x), extra_wg = total - per_wg * (int)gridDim.x;
    const int it0 = (int)blockIdx.x * per_wg + min((int)blockIdx.x, extra_wg);
    const int it1 = it0 + per_wg + ((int)blockIdx.x < extra_wg ? 1 : 0);
    if constexpr (MODE == 1) {
        if (cio.out && total == 0) {   // no token for any local expert: the combined rows are zero
            const u32x4_t zero = {0u, 0u, 0u, 0u};
            for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < cio.T * (R / 8); i += (int)(gridDim.x * blockDim.x))
                reinterpret_cast<u32x4_t*>(cio.out)[i] = zero;
        }
    }
    if (it0 >= it1) return;
    M355_PRO_STAMP(4)   /* item range known */

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl = wave / KP, q = wave % KP;
    const int bq0 = q * NB / KP, bq1 = (q + 1) * NB / KP;   // this wave's blocks: 0..CH of them
    const int blast = max(bq1 - 1, bq0);   // block that out-of-part slots re-read (their scale is zero)
    const int n = lane & 15, g = lane >> 4;
    const size_t wq_expert = (size_t)R * K / 2, am_expert = (size_t)R * NB;

    // Register ring: W[i] holds block i of the current item until its table lookups are issued, then takes the
    // same block of the NEXT item straight away, so one item's worth of HBM reads (CH KiB per wave) is always in
    // flight behind the compute.  A load is one instruction: wave-uniform base (SGPR pair, set once per item) plus a
    // per-lane byte offset that never changes (voff[i]: lane * 16 + 1 KiB per block; slots past the end of the
    // wave's K-part re-read its last block and carry a zero scale).
    u32x4_t W[CH];
    float4_t sA, sB, sAn, sBn;
    const unsigned char* wbase_next;
    uint32_t voff[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) voff[i] = (uint32_t)lane * 16u + (uint32_t)(min(bq0 + i, blast) - bq0) * 1024u;

    // Absmax of an item: lane (n, g) needs the scales of rows 4g..4g+3 of both tiles at block bq0 + n.  Either fp32 per
    // block (wam), or, as the checkpoint stores it, one code byte per block (wam8, same tile order, 4 bytes per lane) and
    // a 256-entry fp32 table per expert (wtab).  The tables of the experts a workgroup is working on sit in LDS (four
    // slots, slot = token tile & 3: consecutive items lie in the same or in the next tile), so turning codes into scales
    // costs LDS reads, not a dependent trip to memory: the codes of item IT + 1 are requested while item IT - 1 runs,
    // looked up at the start of item IT, and the table of a new tile is fetched two items ahead of its first use and
    // stored just before an item's barrier.
    float* ltab = reinterpret_cast<float*>(red + RED_ALL + (INLINE ? (int)align_up(sizeof(InlineTables), 16) : 0));
    uint32_t cAn = 0, cBn = 0;
    const int bs_lane = min(bq0 + n, blast);
    const float keep_lane = (bq0 + n < bq1) ? 1.f : 0.f;   // blocks past the K-part count for nothing
    // An item's place is kept as (token tile, pair group) and stepped from item to item: the kernel divides by the
    // number of pair groups once.
#define M355_CODES_OF(POS, CA, CB)                                                                            \
    {                                                                                                         \
        const int pg_ = (POS).pg;                                                                             \
        const int e_ = (POS).e;                                                                               \
        const int pc_ = min(pg_ * PAIRS + pl, NP - 1);                                                        \
        const uint32_t* cm_ = reinterpret_cast<const uint32_t*>(wam8 + (size_t)e_ * am_expert);               \
        CA = cm_[((size_t)(pc_ * 2 + 0) * 4 + g) * NB + bs_lane];                                             \
        CB = cm_[((size_t)(pc_ * 2 + 1) * 4 + g) * NB + bs_lane];                                             \
    }
    // aim the ring loads at item IT
#define M355_ADDR_OF(POS)                                                                                     \
    {                                                                                                         \
        const int pg_ = (POS).pg;                                                                             \
        const int e_ = (POS).e;                                                                               \
        const int pc_ = min(pg_ * PAIRS + pl, NP - 1);                                                        \
        wbase_next = wq + (size_t)e_ * wq_expert + ((size_t)pc_ * NB + bq0) * 1024;                           \
    }
    // the scales of the item at POS into sAn / sBn (byte form: from the codes in cAn / cBn, which then receive those of
    // the item at NEXT_POS)
#define M355_SCALES_OF(POS, NEXT_POS, REQUEST_NEXT)                                                           \
    {                                                                                                         \
        const int tt_ = (POS).tt;                                                                             \
        if (wam8) {                                                                                           \
            const float* tab_ = ltab + (tt_ & 3) * 256;                                                       \
            sAn = float4_t{tab_[cAn & 0xffu], tab_[(cAn >> 8) & 0xffu], tab_[(cAn >> 16) & 0xffu], tab_[cAn >> 24]} * keep_lane; \
            sBn = float4_t{tab_[cBn & 0xffu], tab_[(cBn >> 8) & 0xffu], tab_[(cBn >> 16) & 0xffu], tab_[cBn >> 24]} * keep_lane; \
            if (REQUEST_NEXT) M355_CODES_OF(NEXT_POS, cAn, cBn)                                               \
        } else {                                                                                              \
            const int pg_ = (POS).pg;                                                                         \
            const int e_ = (POS).e;                                                                           \
            const int pc_ = min(pg_ * PAIRS + pl, NP - 1);                                                    \
            const float4_t* am_ = reinterpret_cast<const float4_t*>(wam + (size_t)e_ * am_expert);            \
            sAn = am_[((size_t)(pc_ * 2 + 0) * 4 + g) * NB + bs_lane] * keep_lane;                            \
            sBn = am_[((size_t)(pc_ * 2 + 1) * 4 + g) * NB + bs_lane] * keep_lane;                            \
        }                                                                                                     \
    }
    // (the empty asm keeps the offset a 32-bit register at the point of use, which is what lets the compiler pick
    // the SGPR-base + VGPR-offset addressing form instead of widening the offsets to 64 bits outside the loop)
#define M355_LOAD(I)                                                                                          \
    {                                                                                                         \
        asm("" : "+v"(voff[I]));                                                                              \
        W[I] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wbase_next + voff[I]));            \
    }
#define M355_NO_LOAD(I)

    // Start-up: everything the first item needs from memory is requested at once -- its absmax codes, the absmax tables
    // of the first two tiles, its weights -- in that order (loads return in order: the small ones first), and the code
    // table is built in LDS while they are on their way.
    constexpr int TAB_PER_THREAD = 512 / THREADS;   // two tables, 256 entries each, over the workgroup
    float tab_first[TAB_PER_THREAD];
    struct ItemPos { int tt, pg, e; };   // token tile, pair group, the tile's expert (looked up when the tile changes)
    auto step = [&](ItemPos p_) {
        p_.pg++;
        if (p_.pg == NPG) { p_.pg = 0; p_.tt++; p_.e = M355_TILE_EXPERT(p_.tt); }
        return p_;
    };
    ItemPos pos0, pos1, pos2;   // of items it, it + 1, it + 2 (it = the item being worked on); places past the range are clamped to its last item
    pos0.tt = (int)((uint32_t)it0 / (uint32_t)NPG);
    pos0.pg = it0 - pos0.tt * NPG;
    pos0.e = M355_TILE_EXPERT(pos0.tt);
    pos1 = (it0 + 1 < it1) ? step(pos0) : pos0;
    pos2 = (it0 + 2 < it1) ? step(pos1) : pos1;
    const int tt_first = pos0.tt, tt_second = pos1.tt;
    M355_PRO_STAMP(5)   /* first item places known */
    uint32_t cA1 = 0, cB1 = 0;
    M355_ADDR_OF(pos0)
    if (wam8) {
        M355_CODES_OF(pos0, cAn, cBn)
        M355_CODES_OF(pos1, cA1, cB1)
#pragma unroll
        for (int j = 0; j < TAB_PER_THREAD; j++) {
            const int i = tid + j * THREADS;
            const int e_ = i < 256 ? pos0.e : pos1.e;
            tab_first[j] = wtab[(size_t)e_ * 256 + (i & 255)];
        }
    } else {
        M355_SCALES_OF(pos0, pos1, false)
    }
    M355_PRO_STAMP(6)   /* codes and tables requested */
#pragma unroll
    for (int i = 0; i < CH; i++) M355_LOAD(i)
    M355_PRO_STAMP(1)   /* first loads issued */

    build_lutp<THREADS>(lut, tid);
    M355_PRO_STAMP(2)   /* code table built */
    const uint32_t lbase = lutp_lane_base(smem, lane);
    if (wam8) {
#pragma unroll
        for (int j = 0; j < TAB_PER_THREAD; j++) {
            const int i = tid + j * THREADS;
            ltab[((i < 256 ? tt_first : tt_second) & 3) * 256 + (i & 255)] = tab_first[j];
        }
        lds_barrier();
        M355_SCALES_OF(pos0, pos1, false)
        cAn = cA1;   // the second item's codes, requested ahead of the first item's weights
        cBn = cB1;
    }
    sA = sAn;
    sB = sBn;

    // Every wave runs CH blocks per item without branches.  Per block: M355_FETCH (16 table lookups = four A
    // fragments, then the ring load that refills the block's registers), M355_MFMA (four MFMAs per token set against
    // the register-resident B fragments), M355_SCALE (absmax FMAs on the 16x16 partial products).  The compiler
    // interleaves neighbouring blocks.
#define M355_FETCH(I, LOAD)                                                                                 \
    {                                                                                                       \
        fa[(I) & 1][0] = lut4p(lbase, W[I][0]);                                                             \
        fa[(I) & 1][2] = lut4p(lbase, W[I][2]);                                                             \
        fa[(I) & 1][1] = lut4p(lbase, W[I][1]);                                                             \
        fa[(I) & 1][3] = lut4p(lbase, W[I][3]);                                                             \
        LOAD(I)                                                                                             \
    }
#define M355_MFMA(I)                                                                                        \
    _Pragma("unroll") for (int s_ = 0; s_ < NT; s_++) {                                                     \
        const float4_t z = {0.f, 0.f, 0.f, 0.f};                                                            \
        pa[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[(I) & 1][0], xb[s_][I][0], z, 0, 0, 0);          \
        pb[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[(I) & 1][2], xb[s_][I][0], z, 0, 0, 0);          \
        pa[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[(I) & 1][1], xb[s_][I][1], pa[s_], 0, 0, 0);     \
        pb[s_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[(I) & 1][3], xb[s_][I][1], pb[s_], 0, 0, 0);     \
    }
    // element 0 of each tile goes through instructions the compiler knows (so it places the MFMA-result wait
    // states); the other elements follow as single v_fmac_f32_dpp instructions that depend on element 0's result
#define M355_SCALE(I)                                                                                       \
    _Pragma("unroll") for (int s_ = 0; s_ < NT; s_++) {                                                     \
        accA[s_][0] += row_bcast<I>(sA[0]) * pa[s_][0];                                                     \
        accB[s_][0] += row_bcast<I>(sB[0]) * pb[s_][0];                                                     \
        _Pragma("unroll") for (int r = 1; r < 4; r++) {                                                     \
            accA[s_][r] = fmac_row_bcast_after<I>(accA[s_][r], sA[r], pa[s_][r], accA[s_][0]);              \
            accB[s_][r] = fmac_row_bcast_after<I>(accB[s_][r], sB[r], pb[s_][r], accB[s_][0]);              \
        }                                                                                                   \
    }
#ifdef M355_EXP_NOCOMPUTE   /* experiment: keep the loads, drop table lookups and MFMAs */
#undef M355_FETCH
#undef M355_MFMA
#undef M355_SCALE
#define M355_FETCH(I, LOAD) asm volatile("" ::"v"(W[I])); LOAD(I)
#define M355_MFMA(I)
#define M355_SCALE(I) accA[0][0] += sA[0]; accB[0][0] += sB[0];
#endif
    // The table lookups of block I + 1 are issued before the MFMAs of block I (two sets of A fragments), so a wave's LDS
    // latency runs behind its own matrix work instead of in front of it.
#ifdef M355_EXP_NO_SWPIPE
#define M355_BLOCK(I, NEXT, LOAD) M355_FETCH(I, LOAD) M355_MFMA(I) M355_SCALE(I)
#define M355_FIRST_FETCH(LOAD)
#else
#define M355_BLOCK(I, NEXT, LOAD) NEXT M355_MFMA(I) M355_SCALE(I)
#define M355_FIRST_FETCH(LOAD) M355_FETCH(0, LOAD)
#endif

    // One work item.  LOAD is M355_LOAD while another item follows in this workgroup's range (the ring then refills
    // with it) and M355_NO_LOAD for the last one.
#define M355_ITEM(IT, LOAD, MORE)                                                                           \
    {                                                                                                       \
        const int it = (IT);                                                                                \
        const int tt = pos0.tt, pg = pos0.pg, e = pos0.e;                                                   \
        const int p = pg * PAIRS + pl;                                                                      \
        const bool live = p < NP;                                                                           \
        if (tt != staged_tile) {                                                                            \
            row0 = M355_OFFSET(e) + M355_TILE_M0(tt);                                                       \
            mt = min(mt_rows, M355_OFFSET(e + 1) - row0);                                                   \
            /* this lane's B fragments of the whole K-part: token rows n, n + 16, .. (clamped), columns */  \
            /* 64b + 16g .. + 15 of each block, from global memory into registers, where they stay until */ \
            /* the token tile changes */                                                                    \
            _Pragma("unroll") for (int s_ = 0; s_ < NT; s_++) {                                             \
                const int pos = row0 + min(n + 16 * s_, mt - 1);                                            \
                const int64_t src_row = (MODE == 0) ? (int64_t)(M355_PAIR_AT(pos) / k_top) : (int64_t)pos;  \
                const u32x4_t* src = reinterpret_cast<const u32x4_t*>(xin + src_row * sp.x_stride) + 2 * g; \
                _Pragma("unroll") for (int i = 0; i < CH; i++) {                                            \
                    const int bb = min(bq0 + i, blast);                                                     \
                    xb[s_][i][0] = __builtin_bit_cast(half8_t, src[8 * bb]);                                \
                    xb[s_][i][1] = __builtin_bit_cast(half8_t, src[8 * bb + 1]);                            \
                }                                                                                           \
            }                                                                                               \
            staged_tile = tt;                                                                               \
        }                                                                                                   \
        /* the table of a tile that starts two items from here: requested now, stored before this item's barrier */ \
        float tab_new = 0.f;                                                                                \
        int tab_slot = -1;                                                                                  \
        if (MORE) {                                                                                         \
            M355_ADDR_OF(pos1)                                                                              \
            M355_SCALES_OF(pos1, pos2, true)                                                                \
            if (wam8 && it + 2 < it1) {                                                                     \
                const int t2_ = pos2.tt;                                                                    \
                if (t2_ != pos1.tt) {                                                                       \
                    tab_slot = t2_ & 3;                                                                     \
                    if (THREADS == 256 || tid < 256) tab_new = wtab[(size_t)pos2.e * 256 + (tid & 255)];    \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
        float4_t accA[NT], accB[NT];                                                                        \
        _Pragma("unroll") for (int s_ = 0; s_ < NT; s_++) {                                                 \
            accA[s_] = float4_t{0.f, 0.f, 0.f, 0.f};                                                        \
            accB[s_] = float4_t{0.f, 0.f, 0.f, 0.f};                                                        \
        }                                                                                                   \
        half8_t fa[2][4];                                                                                   \
        float4_t pa[NT], pb[NT];                                                                            \
        M355_STAMP()                                                                                        \
        M355_FIRST_FETCH(LOAD)                                                                              \
        M355_BLOCK(0, M355_FETCH(1, LOAD), LOAD) M355_BLOCK(1, M355_FETCH(2, LOAD), LOAD)                   \
        M355_BLOCK(2, M355_FETCH(3, LOAD), LOAD) M355_BLOCK(3, M355_FETCH(4, LOAD), LOAD)                   \
        M355_BLOCK(4, M355_FETCH(5, LOAD), LOAD)                                                            \
        if constexpr (CH > 6) {                                                                             \
            M355_BLOCK(5, M355_FETCH(6, LOAD), LOAD) M355_BLOCK(6, M355_FETCH(7, LOAD), LOAD)               \
            M355_BLOCK(7, M355_FETCH(8, LOAD), LOAD) M355_BLOCK(8, M355_FETCH(9, LOAD), LOAD)               \
            M355_BLOCK(9, M355_FETCH(10, LOAD), LOAD) M355_BLOCK(10, M355_FETCH(11, LOAD), LOAD)            \
            M355_BLOCK(11, , LOAD)                                                                          \
        } else {                                                                                            \
            M355_BLOCK(5, , LOAD)                                                                           \
        }                                                                                                   \
        M355_STAMP()                                                                                        \
        sA = sAn;                                                                                           \
        sB = sBn;                                                                                           \
        /* Sum of the K-parts and epilogue, shared out: every wave leaves its partial sums in LDS (element-major, so */ \
        /* that both the stores and the reads below touch every bank once), and after the item's one barrier (two  */ \
        /* buffers in turn) wave q finishes accumulator element q & 3 of token set q >> 2 for all lanes: the four  */ \
        /* (or eight) partial values are added in K-part order, then bias, activation and one 2-byte store.        */ \
        float* rbuf = reinterpret_cast<float*>(red + (round & 1) * RED1) + (size_t)pl * (KP * NT * 8 * 64);   \
        round++;                                                                                            \
        _Pragma("unroll") for (int s_ = 0; s_ < NT; s_++)                                                   \
            _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                 \
                rbuf[((q * NT + s_) * 8 + r) * 64 + lane] = accA[s_][r];                                    \
                rbuf[((q * NT + s_) * 8 + 4 + r) * 64 + lane] = accB[s_][r];                                \
            }                                                                                               \
        if (tab_slot >= 0 && (THREADS == 256 || tid < 256)) ltab[tab_slot * 256 + (tid & 255)] = tab_new;   \
        lds_barrier();                                                                                      \
        M355_STAMP()                                                                                        \
        if constexpr (MODE == 0) {                                                                          \
            const int s_ = q >> 2, r = q & 3;                                                               \
            float sumA = 0.f, sumB = 0.f;                                                                   \
            _Pragma("unroll") for (int qq = 0; qq < KP; qq++) {                                             \
                sumA += rbuf[((qq * NT + s_) * 8 + r) * 64 + lane];                                         \
                sumB += rbuf[((qq * NT + s_) * 8 + 4 + r) * 64 + lane];                                     \
            }                                                                                               \
            if (live && n + 16 * s_ < mt) gemm_epilogue1<MODE>(sumA, sumB, outp, row0 + n + 16 * s_, R, NP, p, 4 * g + r, e, sp); \
        } else {                                                                                            \
            /* y needs no partner row: wave q takes rows 2 (q >> 1 & 1), + 1 of tile q & 1 -> one 4-byte store */ \
            const int s_ = q >> 2, tile = q & 1, r = q & 2;                                                 \
            float sum0 = 0.f, sum1 = 0.f;                                                                   \
            _Pragma("unroll") for (int qq = 0; qq < KP; qq++) {                                             \
                sum0 += rbuf[((qq * NT + s_) * 8 + 4 * tile + r) * 64 + lane];                              \
                sum1 += rbuf[((qq * NT + s_) * 8 + 4 * tile + r + 1) * 64 + lane];                          \
            }                                                                                               \
            if (live && n + 16 * s_ < mt)                                                                   \
                gemm_epilogue_y2(sum0, sum1, outp, row0 + n + 16 * s_, R, 16 * (p + tile * NP) + 4 * g + r, e, sp, cio.out != nullptr); \
        }                                                                                                   \
        pos0 = pos1;                                                                                        \
        pos1 = pos2;                                                                                        \
        if (it + 3 < it1) pos2 = step(pos2);                                                                \
        M355_STAMP()                                                                                        \
    }

#ifdef M355_EXP_STAMPS
    int dbg_n = 0;
    if ((wave & 3) == 0) {   // kernel entry (cycles, 100 MHz wall clock), start of the item loop
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime(), r0_ = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long* d_ = m355_dbg + ((size_t)(blockIdx.x & 511) * 2 + (wave >> 2)) * 128;
            d_[0] = dbg_entry; d_[1] = r0_; d_[2] = (unsigned long long)(it1 - it0); d_[3] = 0; d_[4] = t0_;
            d_[124] = dbg_pro[0]; d_[125] = dbg_pro[1]; d_[126] = dbg_pro[2]; d_[127] = dbg_pro[3];
            d_[121] = dbg_pro[4]; d_[122] = dbg_pro[5]; d_[123] = dbg_pro[6];
        }
        dbg_n = 1;
    }
#endif
    int staged_tile = -1, row0 = 0, mt = 0;   // the token tile whose activations are in registers: its first row and size
    int round = 0;
    half8_t xb[NT][CH][2];
    for (int itx = it0; itx < it1 - 1; itx++) M355_ITEM(itx, M355_LOAD, true)
    M355_ITEM(it1 - 1, M355_NO_LOAD, false)
#ifdef M355_EXP_STAMPS
    if ((wave & 3) == 0) {
        const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) m355_dbg[((size_t)(blockIdx.x & 511) * 2 + (wave >> 2)) * 128 + 3] = r1_;
    }
#endif
    if constexpr (MODE == 1) {
        if (cio.out) {
            // The combine of this launch's own output.  The y values were stored through to memory (4-byte device-scope
            // stores).  Every wave waits until its own stores have been acknowledged (the explicit vmcnt(0): a
            // workgroup-scope fence emits no such wait), the workgroup meets, and only then wave 0 counts the
            // workgroup's finished items on their column groups (device-scope atomics that return the old value).
            // Whoever brings a counter to the number of token tiles knows that every y value of those columns is in
            // memory.  That workgroup then makes one device-scope acquire (its CU's L1 is invalidated; the wait behind
            // it holds the barrier until the invalidate has completed) and combines the columns for every token,
            // reading y with device-scope loads.  Nobody waits for another workgroup, and no L2 is written back.
            constexpr int JOBS = PAIRS * 4;   // columns of pair p: [16p, 16p + 16) and [16(p + NP), ..): two 8-column chunks each
            // A decode batch gives a thread at most one (token, chunk) job per finished column group and a token at most
            // four pairs: their positions and routing weights (left by earlier launches) are requested up front, so that
            // a workgroup that turns out to finish a column group is one trip to memory away from its y rows.
            const bool few = k_top <= 4 && cio.T * JOBS <= THREADS;
            const int my_t = tid / JOBS, my_wi = tid % JOBS;
            int pos4[4] = {-1, -1, -1, -1};
            float wf4[4] = {0.f, 0.f, 0.f, 0.f};
            if (few && my_t < cio.T) combine_fetch_pairs<4>(cio.inv, cio.w, my_t, 0, k_top, cio.w_stride, pos4, wf4);
            // [0]: count, [1..]: column groups this workgroup completed (in the absmax table slots, which no wave reads
            // after the last item's barrier; the reduction buffers may still be in use by a wave's epilogue)
            int32_t* fin = reinterpret_cast<int32_t*>(ltab);
            if (tid == 0) fin[0] = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int n_tt = (int)((uint32_t)total / (uint32_t)NPG);
            if (wave == 0) {
                for (int i = it0 + lane; i < it1; i += 64) {
                    const int pg_i = i % NPG;
                    const int before = __hip_atomic_fetch_add(cio.pg_done + pg_i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (before == n_tt - 1) fin[1 + atomicAdd(&fin[0], 1)] = pg_i;
                }
            }
            __syncthreads();
            const int n_fin = fin[0];
            if (n_fin > 0) {   // (uniform over the workgroup)
                if (wave == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                for (int j = 0; j < n_fin; j++) {
                    const int pg_c = fin[1 + j];
                    if (few) {
                        const int p = pg_c * PAIRS + my_wi / 4;
                        if (my_t < cio.T && p < NP) {
                            const int c = (my_wi & 2) ? 2 * (p + NP) + (my_wi & 1) : 2 * p + (my_wi & 1);
                            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            combine_accumulate<4, true>(outp, pos4, wf4, c, R, cio.fp16_mode, acc);
                            half8_t o;
#pragma unroll
                            for (int jj = 0; jj < 8; jj++) o[jj] = (half_t)acc[jj];
                            reinterpret_cast<half8_t*>(cio.out + (int64_t)my_t * R)[c] = o;
                        }
                        continue;
                    }
                    for (int job = tid; job < cio.T * JOBS; job += THREADS) {
                        const int t = job / JOBS, wi = job % JOBS;
                        const int p = pg_c * PAIRS + wi / 4;
                        if (p >= NP) continue;
                        const int c = (wi & 2) ? 2 * (p + NP) + (wi & 1) : 2 * p + (wi & 1);
                        reinterpret_cast<half8_t*>(cio.out + (int64_t)t * R)[c] =
                            combine_chunk<4, true>(outp, cio.inv, cio.w, t, c, R, k_top, cio.fp16_mode, cio.w_stride);
                    }
                }
            }
        }
    }
#undef M355_ITEM
#undef M355_TILE_EXPERT
#undef M355_TILE_M0
#undef M355_OFFSET
#undef M355_PAIR_AT
#undef M355_BLOCK
#undef M355_FIRST_FETCH
#undef M355_SCALE
#undef M355_MFMA
#undef M355_FETCH
#undef M355_NO_LOAD
#undef M355_LOAD
#undef M355_SCALES_OF
#undef M355_ADDR_OF
#undef M355_CODES_OF
}

// out[t][c] = sum over the token's local pairs, in k order (combine_chunk, shared with the fused combine + norm kernel).
template <int KB>
__global__ __launch_bounds__(256) void moe_combine_kernel(const half_t* __restrict__ y, const int32_t* __restrict__ inv,
                                                          const float* __restrict__ w, int H, int k_top, int fp16_mode,
                                                          half_t* __restrict__ out, int w_stride) {
    const int t = blockIdx.x;
    const int chunks = H / 8;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x)
        reinterpret_cast<half8_t*>(out + (int64_t)t * H)[c] = combine_chunk<KB>(y, inv, w, t, c, H, k_top, fp16_mode, w_stride);
}

static int gemm_lds_bytes(int K, int mt_rows) { return LUT_BYTES + RED_BYTES + mt_rows * (K * 2 + XS_PAD); }


// The streaming kernel's form for 17 .. 32 tokens per expert and up: token tiles of 32 (two 16-token sets share one pass
// over the weights), one row-tile pair per item, 8 K-parts of 6 blocks per wave.
constexpr int WIDE_KP = 8;
constexpr int WIDE_MT = 32;
constexpr int WIDE_MIN_MEAN_PAIRS = 16;   // mean (token, k) pairs per local expert from which the 32-token tiles pay

template <typename Kern>
static int allow_big_lds(Kern kern) {
    return hip_status(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                      "hipFuncSetAttribute");
}

// Once per device: raise the dynamic-LDS limit of the GEMM kernels; the CU count is read on every call (a cached
// attribute query).
static int device_setup(int dev, int* cus) {
    static PerDeviceOnce once;
    if (int rc = once.run(dev, [] {
            if (int rc = allow_big_lds(&moe_gemm_kernel<0>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_kernel<1>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<0, 1, 0>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<0, 2, 0>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<0, 1, 1>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<0, 2, 1>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<0, 1, 2>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<0, 2, 2>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<1, 1, 0>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<1, 2, 0>)) return rc;
            if (int rc = allow_big_lds(&moe_gemm_stream_kernel<0, 1, 0, WIDE_KP, 2>)) return rc;
            return allow_big_lds(&moe_gemm_stream_kernel<1, 1, 0, WIDE_KP, 2>);
        }))
        return rc;
    int v = 0;
    if (int rc = hip_status(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return rc;
    *cus = v > 0 ? v : 256;
    return 0;
}

static int tile_gemm_device_setup_once(int dev) {
    static PerDeviceOnce once;
    return once.run(dev, [] { return tile_gemm_device_setup(); });
}

// Everything the MoE entry points set up once per device, for m355_init (the device is current).
int moe_device_init(int dev) {
    int cus = 0;
    if (int rc = device_setup(dev, &cus)) return rc;
    return tile_gemm_device_setup_once(dev);
}

#ifndef M355_FOLD_MAX_PAIRS
#define M355_FOLD_MAX_PAIRS 12
#endif
constexpr int FOLD_COMBINE_MAX_PAIRS = M355_FOLD_MAX_PAIRS;   // (token, k) pairs up to which the down launch also combines

// mean (token, k) pairs per local expert from which the tile kernel takes over
constexpr int TILE_MIN_MEAN_PAIRS = 80;   // measured crossover against the 32-token streaming form on gpt-oss-20b shapes (32 experts: 640 tokens)

constexpr int ABSMAX_TABLE_SLOTS_BYTES = 4 * 256 * 4;   // four experts' absmax tables (byte form)
static int stream_lds_bytes(int pairs, int inline_route, int kparts = MOE_KPARTS, int token_sets = 1) {   // (activations live in registers)
    return LUTP_BYTES + 2 * pairs * kparts * 64 * 8 * token_sets * 4 + (inline_route ? (int)align_up(sizeof(InlineTables), 16) : 0) +
           ABSMAX_TABLE_SLOTS_BYTES;
}

// A gate_up launch can bucket the pairs itself when a pair per thread and two experts per lane cover them.
static bool can_route_inline(int P, int E_local, int K) {
    return K / 64 <= MOE_KPARTS * MOE_CHUNK && P <= INLINE_MAX_P && E_local <= INLINE_MAX_E;
}

// Measuring aid: after m355_time_next_launch() the next streaming-kernel launch made from this thread carries start / stop
// events of its own (hipExtLaunchKernelGGL), which bracket the kernel alone -- the duration a kernel trace reports, with
// neither the gap to a neighbouring launch nor the cost of event markers around it.
static thread_local hipEvent_t tl_time_start = nullptr, tl_time_stop = nullptr;
static thread_local bool tl_time_armed = false, tl_time_taken = false;
template <typename... KArgs, typename... Args>
static void launch_stream_kernel(void (*kern)(KArgs...), dim3 grid, dim3 block, int lds, hipStream_t st, Args... args) {
    if (tl_time_armed) {
        tl_time_armed = false;
        tl_time_taken = true;
        hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, st, tl_time_start, tl_time_stop, 0, static_cast<KArgs>(args)...);
    } else {
        kern<<<grid, block, lds, st>>>(static_cast<KArgs>(args)...);
    }
}

// W [R][K] per expert.  K <= 48 blocks: streaming kernel, one or two workgroups per CU; else the general kernel.
// `rio` non-null (gate_up only): the kernel buckets the pairs itself and leaves the lists in the workspace.
template <int MODE, int INLINE>
static int launch_gemm(const half_t* xin, const uint8_t* wq, const float* wam, const MoeWorkspace& ws, int K, int R,
                       int k_top, int mt_rows, int slots, half_t* outp, int cus, hipStream_t st, const RouteIO& rio,
                       const StageParams& sp, const uint8_t* wam8, const float* wtab, const CombineIO& cio) {
    const int NP = R / 32, NB = K / 64;
    if (NB <= MOE_KPARTS * MOE_CHUNK) {
        // upper bound of sum_e ceil(M_e / mt_rows): with T <= 16 a token tile holds a whole expert (M_e <= T = mt_rows)
        const int max_tiles = (ws.max_pairs / k_top <= MOE_MAX_MT) ? slots : slots + ws.max_pairs / mt_rows;
        const int items2 = max_tiles * ((NP + 1) / 2);
        const int table_len = ws.max_pairs + slots < 64 ? ws.max_pairs + slots : 64;   // the table holds P + E_local entries
        if constexpr (INLINE == 0) {
            if (mt_rows == WIDE_MT) {   // 32-token tiles: one pair per item, 8 K-parts of 6 blocks, two token sets
                const int lds = stream_lds_bytes(1, 0, WIDE_KP, 2);
                const int items1 = max_tiles * NP;
                const int grid = items1 < cus ? items1 : cus;
                launch_stream_kernel(moe_gemm_stream_kernel<MODE, 1, 0, WIDE_KP, 2>, dim3(grid), dim3(WIDE_KP * 64), lds, st,
                    xin, wq, wam, ws.n_tiles, ws.tile_expert, ws.tile_m0, ws.offsets, ws.sorted, K, R, k_top, mt_rows, outp, table_len, rio, sp, wam8, wtab, cio);
                M355_LAUNCH_CHECK("moe_gemm_stream_kernel");
                return 0;
            }
        }
        if (2 * items2 >= cus) {   // 8-wave workgroups unless even those would leave most CUs idle
            const int lds = stream_lds_bytes(2, INLINE);
            const int grid = items2 < cus ? items2 : cus;
            launch_stream_kernel(moe_gemm_stream_kernel<MODE, 2, INLINE>, dim3(grid), dim3(512), lds, st,
                xin, wq, wam, ws.n_tiles, ws.tile_expert, ws.tile_m0, ws.offsets, ws.sorted, K, R, k_top, mt_rows, outp, table_len, rio, sp, wam8, wtab, cio);
        } else {
            const int lds = stream_lds_bytes(1, INLINE);
            const int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
            const int items1 = max_tiles * NP;
            const int grid = items1 < per_cu * cus ? items1 : per_cu * cus;
            launch_stream_kernel(moe_gemm_stream_kernel<MODE, 1, INLINE>, dim3(grid), dim3(256), lds, st,
                xin, wq, wam, ws.n_tiles, ws.tile_expert, ws.tile_m0, ws.offsets, ws.sorted, K, R, k_top, mt_rows, outp, table_len, rio, sp, wam8, wtab, cio);
        }
        M355_LAUNCH_CHECK("moe_gemm_stream_kernel");
        return 0;
    }
    if (sp.bias || sp.act_mode != M355_ACT_SILU_CONCAT || wam8)
        return fail(-5, "m355_moe_nf4_forward: biases, swiglu-oai and byte absmax need K <= %d (K = %d)", 64 * MOE_KPARTS * MOE_CHUNK, K);
    // this kernel walks an expert's tokens by itself (it reads the offsets, not the token-tile lists), so it may take fewer
    // tokens per pass than the routing step's tiles: as many activation rows as fit in LDS beside the tables
    const int fit = (160 * 1024 - LUT_BYTES - RED_BYTES) / (K * 2 + XS_PAD);
    const int mt_gen = mt_rows < fit ? mt_rows : fit;
    if (mt_gen < 1) return fail(-2, "m355_moe_nf4_forward: K = %d is too large for the LDS activation tile", K);
    moe_gemm_kernel<MODE><<<dim3((NP + MOE_PAIRS - 1) / MOE_PAIRS, slots), dim3(MOE_THREADS), gemm_lds_bytes(K, mt_gen), st>>>(
        xin, wq, wam, ws.n_active, ws.offsets, ws.active, ws.sorted, K, R, k_top, mt_gen, outp, sp.x_stride);
    M355_LAUNCH_CHECK("moe_gemm_kernel");
    return 0;
}

}  // namespace m355

using namespace m355;

#ifdef M355_EXP_STAMPS
extern "C" __attribute__((visibility("default"))) int m355_dbg_read(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(m355_dbg), bytes < sizeof(m355_dbg) ? bytes : sizeof(m355_dbg));
}
#endif

extern "C" int m355_time_next_launch(void) {
    if (!tl_time_start) {
        if (int rc = hip_status(hipEventCreate(&tl_time_start), "hipEventCreate")) return rc;
        if (int rc = hip_status(hipEventCreate(&tl_time_stop), "hipEventCreate")) return rc;
    }
    tl_time_armed = true;
    tl_time_taken = false;
    return 0;
}

extern "C" int m355_timed_launch_us(float* us) {
    if (!us) return fail(-1, "m355_timed_launch_us: null pointer");
    if (!tl_time_taken) return fail(-2, "m355_timed_launch_us: no streaming-kernel launch since m355_time_next_launch");
    if (int rc = hip_status(hipEventSynchronize(tl_time_stop), "hipEventSynchronize")) return rc;
    float ms = 0.f;
    if (int rc = hip_status(hipEventElapsedTime(&ms, tl_time_start, tl_time_stop), "hipEventElapsedTime")) return rc;
    *us = ms * 1e3f;
    return 0;
}

extern "C" size_t m355_moe_workspace_bytes(int T, int k, int H, int FF, int E_local) {
    if (T < 0 || k <= 0 || H <= 0 || FF <= 0 || E_local <= 0) return 0;
    return carve_workspace(nullptr, T, k, H, FF, E_local).bytes;
}

extern "C" int m355_moe_nf4_forward(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                    const uint8_t* w13_q, const float* w13_absmax, const uint8_t* w2_q,
                                    const float* w2_absmax, int E_local, int e_first, int H, int FF, int act_mode,
                                    int flags, void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream) {
    return m355_moe_nf4_forward_ex(x_f16, T, ids, w, k, w13_q, w13_absmax, w2_q, w2_absmax, E_local, e_first, H, FF, act_mode,
                                   flags, nullptr, nullptr, 1.702f, 7.0f, workspace, ws_bytes, out_f16, stream);
}

// absmax of one weight matrix: fp32 per block, or the checkpoint's code byte per block + a 256-entry table per expert
struct AbsmaxArg {
    const float* f32;
    const uint8_t* u8;
    const float* table;
    bool ok() const { return f32 ? true : (u8 && table); }
    const void* base() const { return f32 ? (const void*)f32 : (const void*)u8; }
};

// Where the rows of x, ids and w lie: dense arrays (strides H, k, k), or inside wider records.
struct RowsArg {
    int64_t x_stride;   // fp16 elements
    int ids_stride;     // int32
    int w_stride;       // floats
};

// The router of the same tokens, when the call is to compute it too: router matrix [E_global][H], bias or null; then
// `ids` / `w` of moe_forward_impl are outputs.
struct RouterArg {
    const void* wr;
    const void* br;
    int E_global;
    const void* logits;   // instead of wr / br: finished logits [T][E_global] fp16 (what a FusedMoE layer is handed)
};

static int moe_forward_impl(const void* x_f16, int T, const int32_t* ids, const float* w, int k, const uint8_t* w13_q,
                            AbsmaxArg a13, const uint8_t* w2_q, AbsmaxArg a2, int E_local, int e_first, int H, int FF,
                            int act_mode, int flags, const void* b13_f16, const void* b2_f16, float alpha, float limit,
                            void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream,
                            RouterArg router = RouterArg{nullptr, nullptr, 0, nullptr}, RowsArg rows = RowsArg{0, 0, 0}) {
    if (T == 0) return 0;   // an empty batch has no buffers to check
    if (!x_f16 || !ids || !w || !w13_q || !a13.ok() || !w2_q || !a2.ok() || !out_f16)
        return fail(-1, "m355_moe_nf4_forward: null pointer");
    const float* w13_absmax = a13.f32;
    const float* w2_absmax = a2.f32;
    if (T < 0 || k <= 0 || k > 32 || E_local <= 0 || E_local > ROUTE_MAX_E || H <= 0 || FF <= 0 || (H % 64) || (FF % 64))
        return fail(-2, "m355_moe_nf4_forward: unsupported shape T=%d k=%d E_local=%d H=%d FF=%d (need H,FF %% 64 == 0, k <= 32, E_local <= %d)",
                    T, k, E_local, H, FF, ROUTE_MAX_E);
    if (act_mode != M355_ACT_SILU_CONCAT && act_mode != M355_ACT_SWIGLU_OAI)
        return fail(-5, "m355_moe_nf4_forward: act_mode %d not supported", act_mode);
    if ((b13_f16 && !aligned(b13_f16, 8)) || (b2_f16 && !aligned(b2_f16, 8)))
        return fail(-3, "m355_moe_nf4_forward: biases must be 8-byte aligned");
    if (rows.x_stride == 0) rows = RowsArg{H, k, k};
    const StageParams sp13 = {act_mode, alpha, limit, (const half_t*)b13_f16, rows.x_stride};
    const StageParams sp2 = {M355_ACT_SILU_CONCAT, 0.f, 0.f, (const half_t*)b2_f16, FF};
    if ((int64_t)T * k > (int64_t)1 << 24) return fail(-2, "m355_moe_nf4_forward: T*k too large");
    const size_t need = m355_moe_workspace_bytes(T, k, H, FF, E_local);
    if (!workspace || ws_bytes < need) return fail(-6, "m355_moe_nf4_forward: workspace of %zu bytes needed, %zu given", need, ws_bytes);
    if (!aligned(workspace, 256) || !aligned(x_f16, 16) || !aligned(out_f16, 16) || !aligned(w13_q, 16) || !aligned(w2_q, 16) ||
        !aligned(a13.base(), 16) || !aligned(a2.base(), 16) || (a13.table && !aligned(a13.table, 16)) ||
        (a2.table && !aligned(a2.table, 16)))
        return fail(-3, "m355_moe_nf4_forward: workspace 256-byte, other buffers 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const MoeWorkspace ws = carve_workspace(workspace, T, k, H, FF, E_local);
    const int P = T * k;
    // Two kernels.  Few tokens per expert: the streaming kernel (HBM-bound; a token tile holds <= 16 tokens and every
    // tile streams its expert's weights).  Many: the tile kernel (256-token tiles, weights dequantised once per tile).
    // The choice is made on the host from the mean pairs per local expert, so that no routing result has to be read
    // back; the flags force either.
    bool tiled = T > MOE_MAX_MT && (int64_t)P >= (int64_t)TILE_MIN_MEAN_PAIRS * E_local;
    if (flags & M355_FLAG_FORCE_TILE) tiled = true;
    if (flags & M355_FLAG_FORCE_STREAM) tiled = false;
    // between the two: from WIDE_MIN_MEAN_PAIRS pairs per expert the streaming kernel takes token tiles of 32
    bool wide = !tiled && T > MOE_MAX_MT && H / 64 <= MOE_KPARTS * MOE_CHUNK && FF / 64 <= MOE_KPARTS * MOE_CHUNK;
    if (wide && !(flags & M355_FLAG_STREAM_WIDE))
        wide = !(flags & M355_FLAG_FORCE_STREAM) && (int64_t)P >= (int64_t)WIDE_MIN_MEAN_PAIRS * E_local;
    const int tile_rows = tile_gemm_rows();
    const int mt_rows = tiled ? tile_rows : wide ? WIDE_MT : (T < MOE_MAX_MT ? T : MOE_MAX_MT);
    const int slots = P < E_local ? P : E_local;

    int dev = 0, cus = 0;
    if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (int rc = device_setup(dev, &cus)) return rc;

    // gate_up buckets the pairs itself when it can; the routing kernel runs when it cannot, or when gate_up is skipped
    const bool inline_route = !tiled && !wide && can_route_inline(P, E_local, H) && !(flags & M355_FLAG_SKIP_GATE_UP);
    // The router of a small batch: the logits as a launch of their own that uses every CU, the top-k inside gate_up (a
    // workgroup per token, as m355_router_topk runs it, pulls the whole router matrix through one CU's L1).  The logits lie
    // in the y rows, which nothing touches before the down stage.
    const bool split_router = (router.wr || router.logits) && inline_route && T <= SPLIT_ROUTER_MAX_T &&
                              router.E_global <= SPLIT_ROUTER_MAX_E && (int64_t)T * router.E_global <= (int64_t)P * H &&
                              !(flags & (M355_FLAG_SKIP_ROUTE | M355_FLAG_ROUTER_LAUNCH));
    if (split_router) {
        if (!router.logits && !(flags & M355_FLAG_LOGITS_READY)) {
            router_logits_kernel<<<dim3((router.E_global + 3) / 4, T), dim3(64), 0, st>>>(
                (const half_t*)x_f16, rows.x_stride, (const half_t*)router.wr, (const half_t*)router.br, H, router.E_global, ws.y);
            M355_LAUNCH_CHECK("router_logits_kernel");
        }
    } else if (router.logits && !(flags & M355_FLAG_SKIP_ROUTE)) {
        if (int rc = m355_topk_softmax(router.logits, T, router.E_global, k, const_cast<int32_t*>(ids), const_cast<float*>(w), stream))
            return rc;
    } else if (router.wr && !(flags & M355_FLAG_SKIP_ROUTE)) {
        if (int rc = m355_router_topk(x_f16, router.wr, router.br, T, H, router.E_global, k, const_cast<int32_t*>(ids),
                                      const_cast<float*>(w), nullptr, stream))
            return rc;
    }
    const RouteIO rio = {ids, rows.ids_stride, P, E_local, e_first, ws.n_active, ws.offsets, ws.active, ws.sorted, ws.inv, ws.n_tiles, ws.tile_expert, ws.tile_m0,
                         router.logits ? (const half_t*)router.logits : ws.y, router.E_global, k, const_cast<int32_t*>(ids), const_cast<float*>(w)};
    if (!(flags & M355_FLAG_SKIP_ROUTE) && !inline_route) {
        // (the per-workgroup counts of the parallel form lie in the h rows, which nothing touches before the gate_up stage)
        if (P >= ROUTE_PARALLEL_MIN_PAIRS && (size_t)ROUTE_WGS * E_local * 4 <= (size_t)2 * P * FF && !(flags & M355_FLAG_SKIP_GATE_UP)) {
            int32_t* hist = reinterpret_cast<int32_t*>(ws.h);
            moe_route_count_kernel<<<dim3(ROUTE_WGS), dim3(ROUTE_THREADS), 0, st>>>(ids, rows.ids_stride, k, P, E_local, e_first, hist);
            M355_LAUNCH_CHECK("moe_route_count_kernel");
            moe_route_place_kernel<<<dim3(ROUTE_WGS), dim3(ROUTE_THREADS), 0, st>>>(ids, rows.ids_stride, k, P, E_local, e_first, hist,
                                                                                  ws.n_active, ws.offsets, ws.active, ws.sorted,
                                                                                  ws.inv, mt_rows, ws.n_tiles, ws.tile_expert,
                                                                                  ws.tile_m0);
            M355_LAUNCH_CHECK("moe_route_place_kernel");
        } else {
            moe_route_kernel<<<dim3(1), dim3(ROUTE_THREADS), 0, st>>>(ids, rows.ids_stride, k, P, E_local, e_first, ws.n_active, ws.offsets,
                                                                      ws.active, ws.sorted, ws.inv, mt_rows, ws.n_tiles,
                                                                      ws.tile_expert, ws.tile_m0);
            M355_LAUNCH_CHECK("moe_route_kernel");
        }
    }
    if (tiled) {
        if (int rc = tile_gemm_device_setup_once(dev)) return rc;
        const int max_tiles = (slots + P / tile_rows < P) ? slots + P / tile_rows : P;   // >= sum_e ceil(M_e / tile_rows)
        if (!(flags & M355_FLAG_SKIP_GATE_UP))
            if (int rc = launch_tile_gemm(0, (const half_t*)x_f16, w13_q, w13_absmax, a13.u8, a13.table, ws, H, 2 * FF, k,
                                          max_tiles, ws.h, sp13, st))
                return rc;
        if (!(flags & M355_FLAG_SKIP_DOWN))
            if (int rc = launch_tile_gemm(1, ws.h, w2_q, w2_absmax, a2.u8, a2.table, ws, FF, H, k, max_tiles, ws.y, sp2, st))
                return rc;
    }
    // The combine step rides in the down launch when both GEMM stages run here as streaming-kernel launches (gate_up clears
    // the counters the down launch counts on) and the batch is small: the y values then travel through memory as
    // device-scope 4-byte stores that every wave waits out, and the workgroups that finish a column group last make an
    // acquire and one more trip to memory, which costs more than the saved launch from 16 (token, k) pairs on (measured:
    // 1 us per layer gained at 4 pairs, 0.25 at 12; 0.7 lost at 16, 1.7 at 32, 6 at 128).  M355_FLAG_FOLD_COMBINE folds
    // at any size.
    const bool fold_combine = !tiled && (P <= FOLD_COMBINE_MAX_PAIRS || (flags & M355_FLAG_FOLD_COMBINE)) && H / 64 <= MOE_KPARTS * MOE_CHUNK && FF / 64 <= MOE_KPARTS * MOE_CHUNK && H / 32 <= SYNC_PG_COUNTERS &&
                              !(flags & (M355_FLAG_SKIP_GATE_UP | M355_FLAG_SKIP_DOWN | M355_FLAG_SKIP_COMBINE | M355_FLAG_SEPARATE_COMBINE));
    const CombineIO cio_gu = {nullptr, nullptr, nullptr, 0, 0, 0, fold_combine ? ws.pg_done : nullptr};
    const CombineIO cio_dn = {fold_combine ? (half_t*)out_f16 : nullptr, ws.inv, w, rows.w_stride, T,
                              (flags & M355_FLAG_FP16_COMBINE) ? 1 : 0, ws.pg_done};
    if (!tiled && !(flags & M355_FLAG_SKIP_GATE_UP)) {
        if (int rc = split_router   ? launch_gemm<0, 2>((const half_t*)x_f16, w13_q, w13_absmax, ws, H, 2 * FF, k, mt_rows, slots, ws.h, cus, st, rio, sp13, a13.u8, a13.table, cio_gu)
                     : inline_route ? launch_gemm<0, 1>((const half_t*)x_f16, w13_q, w13_absmax, ws, H, 2 * FF, k, mt_rows, slots, ws.h, cus, st, rio, sp13, a13.u8, a13.table, cio_gu)
                                    : launch_gemm<0, 0>((const half_t*)x_f16, w13_q, w13_absmax, ws, H, 2 * FF, k, mt_rows, slots, ws.h, cus, st, rio, sp13, a13.u8, a13.table, cio_gu))
            return rc;
    }
    if (!tiled && !(flags & M355_FLAG_SKIP_DOWN)) {
        if (int rc = launch_gemm<1, 0>(ws.h, w2_q, w2_absmax, ws, FF, H, k, mt_rows, slots, ws.y, cus, st, rio, sp2, a2.u8, a2.table, cio_dn)) return rc;
    }
    if (fold_combine) return 0;
    if (!(flags & M355_FLAG_SKIP_COMBINE)) {
        moe_combine_kernel<4><<<dim3(T), dim3(256), 0, st>>>(ws.y, ws.inv, w, H, k, (flags & M355_FLAG_FP16_COMBINE) ? 1 : 0,
                                                          (half_t*)out_f16, rows.w_stride);
        M355_LAUNCH_CHECK("moe_combine_kernel");
    }
    return 0;
}

extern "C" int m355_moe_nf4_forward_ex(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                       const uint8_t* w13_q, const float* w13_absmax, const uint8_t* w2_q,
                                       const float* w2_absmax, int E_local, int e_first, int H, int FF, int act_mode,
                                       int flags, const void* b13_f16, const void* b2_f16, float alpha, float limit,
                                       void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream) {
    return moe_forward_impl(x_f16, T, ids, w, k, w13_q, AbsmaxArg{w13_absmax, nullptr, nullptr}, w2_q,
                            AbsmaxArg{w2_absmax, nullptr, nullptr}, E_local, e_first, H, FF, act_mode, flags, b13_f16, b2_f16,
                            alpha, limit, workspace, ws_bytes, out_f16, stream);
}

extern "C" int m355_moe_nf4_forward_q8(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                       const uint8_t* w13_q, const uint8_t* w13_absmax_u8, const float* w13_table,
                                       const uint8_t* w2_q, const uint8_t* w2_absmax_u8, const float* w2_table, int E_local,
                                       int e_first, int H, int FF, int act_mode, int flags, const void* b13_f16,
                                       const void* b2_f16, float alpha, float limit, void* workspace, size_t ws_bytes,
                                       void* out_f16, m355_stream_t stream) {
    if (T != 0 && (!w13_absmax_u8 || !w13_table || !w2_absmax_u8 || !w2_table)) return fail(-1, "m355_moe_nf4_forward_q8: null pointer");
    return moe_forward_impl(x_f16, T, ids, w, k, w13_q, AbsmaxArg{nullptr, w13_absmax_u8, w13_table}, w2_q,
                            AbsmaxArg{nullptr, w2_absmax_u8, w2_table}, E_local, e_first, H, FF, act_mode, flags, b13_f16, b2_f16,
                            alpha, limit, workspace, ws_bytes, out_f16, stream);
}

extern "C" int m355_moe_nf4_forward_routed(const void* x_f16, int T, const void* wr_f16, const void* br_f16, int E_global,
                                           int k, int32_t* ids_out, float* w_out, const uint8_t* w13_q,
                                           const float* w13_absmax, const uint8_t* w13_absmax_u8, const float* w13_table,
                                           const uint8_t* w2_q, const float* w2_absmax, const uint8_t* w2_absmax_u8,
                                           const float* w2_table, int E_local, int e_first, int H, int FF, int act_mode,
                                           int flags, const void* b13_f16, const void* b2_f16, float alpha, float limit,
                                           void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream) {
    if (T != 0 && (!wr_f16 || !ids_out || !w_out)) return fail(-1, "m355_moe_nf4_forward_routed: null pointer");
    if (T != 0 && (E_global <= 0 || E_global > ROUTER_MAX_E || k > E_global || (H % 8) || !aligned(wr_f16, 16)))
        return fail(-2, "m355_moe_nf4_forward_routed: unsupported router shape E=%d k=%d H=%d (wr 16-byte aligned, E <= %d)",
                    E_global, k, H, ROUTER_MAX_E);
    const bool bytes = w13_absmax == nullptr;
    return moe_forward_impl(x_f16, T, ids_out, w_out, k, w13_q,
                            bytes ? AbsmaxArg{nullptr, w13_absmax_u8, w13_table} : AbsmaxArg{w13_absmax, nullptr, nullptr}, w2_q,
                            bytes ? AbsmaxArg{nullptr, w2_absmax_u8, w2_table} : AbsmaxArg{w2_absmax, nullptr, nullptr},
                            E_local, e_first, H, FF, act_mode, flags, b13_f16, b2_f16, alpha, limit, workspace, ws_bytes,
                            out_f16, stream, RouterArg{wr_f16, br_f16, E_global, nullptr});
}

extern "C" int m355_moe_nf4_forward_logits(const void* x_f16, int T, const void* logits_f16, int E_global, int k,
                                           int32_t* ids_out, float* w_out, const uint8_t* w13_q, const float* w13_absmax,
                                           const uint8_t* w13_absmax_u8, const float* w13_table, const uint8_t* w2_q,
                                           const float* w2_absmax, const uint8_t* w2_absmax_u8, const float* w2_table,
                                           int E_local, int e_first, int H, int FF, int act_mode, int flags,
                                           const void* b13_f16, const void* b2_f16, float alpha, float limit, void* workspace,
                                           size_t ws_bytes, void* out_f16, m355_stream_t stream) {
    if (T != 0 && (!logits_f16 || !ids_out || !w_out)) return fail(-1, "m355_moe_nf4_forward_logits: null pointer");
    if (T != 0 && (E_global <= 0 || E_global > ROUTER_MAX_E || k > E_global))
        return fail(-2, "m355_moe_nf4_forward_logits: unsupported router shape E=%d k=%d (E <= %d)", E_global, k, ROUTER_MAX_E);
    const bool bytes = w13_absmax == nullptr;
    return moe_forward_impl(x_f16, T, ids_out, w_out, k, w13_q,
                            bytes ? AbsmaxArg{nullptr, w13_absmax_u8, w13_table} : AbsmaxArg{w13_absmax, nullptr, nullptr}, w2_q,
                            bytes ? AbsmaxArg{nullptr, w2_absmax_u8, w2_table} : AbsmaxArg{w2_absmax, nullptr, nullptr},
                            E_local, e_first, H, FF, act_mode, flags, b13_f16, b2_f16, alpha, limit, workspace, ws_bytes,
                            out_f16, stream, RouterArg{nullptr, nullptr, E_global, logits_f16});
}

extern "C" int m355_moe_nf4_forward_ep_rows(const void* rows, int NT, int k, const uint8_t* w13_q, const float* w13_absmax,
                                            const uint8_t* w13_absmax_u8, const float* w13_table, const uint8_t* w2_q,
                                            const float* w2_absmax, const uint8_t* w2_absmax_u8, const float* w2_table,
                                            int E_local, int e_first, int H, int FF, int act_mode, int flags,
                                            const void* b13_f16, const void* b2_f16, float alpha, float limit, void* workspace,
                                            size_t ws_bytes, void* out_f16, m355_stream_t stream) {
    if (NT == 0) return 0;
    if (!rows) return fail(-1, "m355_moe_nf4_forward_ep_rows: null pointer");
    if (k <= 0 || H <= 0 || (H % 8)) return fail(-2, "m355_moe_nf4_forward_ep_rows: unsupported shape k=%d H=%d", k, H);
    if (!aligned(rows, 16)) return fail(-3, "m355_moe_nf4_forward_ep_rows: rows must be 16-byte aligned");
    const size_t rb = m355_ep_row_bytes(H, k);   // [x: H fp16][ids: k int32][w: k fp32], padded to 16 bytes
    const unsigned char* base = (const unsigned char*)rows;
    const bool bytes = w13_absmax == nullptr;
    return moe_forward_impl(base, NT, (const int32_t*)(base + (size_t)2 * H), (const float*)(base + (size_t)2 * H + 4 * (size_t)k), k, w13_q,
                            bytes ? AbsmaxArg{nullptr, w13_absmax_u8, w13_table} : AbsmaxArg{w13_absmax, nullptr, nullptr}, w2_q,
                            bytes ? AbsmaxArg{nullptr, w2_absmax_u8, w2_table} : AbsmaxArg{w2_absmax, nullptr, nullptr},
                            E_local, e_first, H, FF, act_mode, flags, b13_f16, b2_f16, alpha, limit, workspace, ws_bytes, out_f16,
                            stream, RouterArg{nullptr, nullptr, 0}, RowsArg{(int64_t)(rb / 2), (int)(rb / 4), (int)(rb / 4)});
}
