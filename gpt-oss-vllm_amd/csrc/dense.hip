// Dense projection y = x W^T + b (+ residual): fp16 in, fp32 accumulate, fp16 out.  This file: the entry point and the
// small-batch (decode) kernel; from 128 tokens on the entry point hands over to the tiled kernel of dense_tile.hip.
// At <= 32 tokens the product is a stream of W through the chip once (2 bytes per weight): HBM-bound, so the kernel
// is built like the MoE weight streamer, not like a tiled GEMM.  Tokens sit on the N side of v_mfma_f32_16x16x32_f16.
//
// A workgroup is 8 waves = 8 K-parts; a work item is a 16-row tile of W; a persistent workgroup walks a contiguous
// range of row tiles.  The activations of a wave's K-part (its B fragments, <= 2 token tiles of 16) are loaded once
// and stay in registers for the whole kernel.  The A fragments of the next item are requested as soon as the current
// item's are consumed (register ring), a row of W being read in whole 128-byte lines (four lanes x 32 bytes).
#include "common.h"
#include "norm_shared.h"

namespace m355 {

constexpr int DENSE_THREADS = 512;
constexpr int DENSE_KPARTS = 8;

__device__ __forceinline__ void dense_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// STEPS = 64-column blocks per wave and item (K <= 64 * 8 * STEPS); NT = token tiles (T <= 16 * NT); ROWS = rows of W per
// item, 16 or 8.  With 8 the lanes of rows 8..15 load rows 0..7 again (the same lines: no further HBM traffic) and their
// results are dropped: half-size items for shapes whose 16-row tiles do not spread evenly over the CUs (5120 rows are 320
// tiles for 256 CUs: two rounds for 1.25 rounds of work, against three half rounds with 8-row items).
template <int STEPS, int NT, int ROWS = 16>
__global__ __launch_bounds__(DENSE_THREADS) void dense_stream_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                                     const half_t* __restrict__ bias,
                                                                     const half_t* __restrict__ resid, int T, int N, int K,
                                                                     half_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float red[2][DENSE_KPARTS - 1][NT][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int NB = K / 64, tiles = N / ROWS;
    const int it0 = (int)((int64_t)blockIdx.x * tiles / gridDim.x), it1 = (int)((int64_t)(blockIdx.x + 1) * tiles / gridDim.x);
    if (it0 >= it1) return;
    const int b0 = q * NB / DENSE_KPARTS, b1 = (q + 1) * NB / DENSE_KPARTS;   // this wave's blocks

    // per-lane byte offsets into a 16-row tile of W: row (lane & 15), columns 64 * block + 16 * g (+ 8)
    uint32_t voff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; s++) voff[s] = (uint32_t)(n & (ROWS - 1)) * (uint32_t)K * 2u + (uint32_t)(min(b0 + s, NB - 1) * 64 + 16 * g) * 2u;

    // B fragments: token (16 * tile + n, clamped), the same columns; zero where the slot is outside this wave's part
    half8_t xb[NT][STEPS][2];
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
        const int tok = min(16 * tt + n, T - 1);
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const half8_t* src = reinterpret_cast<const half8_t*>(x + (int64_t)tok * K + min(b0 + s, NB - 1) * 64 + 16 * g);
            const bool live = b0 + s < b1;
            const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            xb[tt][s][0] = live ? src[0] : z;
            xb[tt][s][1] = live ? src[1] : z;
        }
    }

    u32x4_t A[STEPS][2];
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(w);
#define DENSE_LOAD(S, BASE)                                                                                       \
    {                                                                                                             \
        asm("" : "+v"(voff[S]));   /* keeps the offset 32-bit at the use: SGPR base + VGPR offset addressing */    \
        A[S][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>((BASE) + voff[S]));                \
        A[S][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>((BASE) + voff[S] + 16));           \
    }
    {
        const unsigned char* base = wb + (size_t)it0 * ROWS * K * 2;
#pragma unroll
        for (int s = 0; s < STEPS; s++) DENSE_LOAD(s, base)
    }
#define DENSE_NO_LOAD(S, BASE)
    // One row tile.  LOAD is DENSE_LOAD while another tile follows in this workgroup's range (the ring then refills
    // with it) and DENSE_NO_LOAD for the last one.
#define DENSE_ITEM(IT, LOAD)                                                                                      \
    {                                                                                                             \
        const int it = (IT);                                                                                      \
        const unsigned char* next = wb + (size_t)(it + 1) * ROWS * K * 2;                                         \
        (void)next;                                                                                               \
        float4_t acc[NT];                                                                                         \
        _Pragma("unroll") for (int tt = 0; tt < NT; tt++) acc[tt] = float4_t{0.f, 0.f, 0.f, 0.f};                 \
        _Pragma("unroll") for (int s = 0; s < STEPS; s++) {                                                       \
            const half8_t a0 = __builtin_bit_cast(half8_t, A[s][0]), a1 = __builtin_bit_cast(half8_t, A[s][1]);   \
            _Pragma("unroll") for (int tt = 0; tt < NT; tt++) {                                                   \
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xb[tt][s][0], acc[tt], 0, 0, 0);             \
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xb[tt][s][1], acc[tt], 0, 0, 0);             \
            }                                                                                                     \
            LOAD(s, next)                                                                                         \
        }                                                                                                         \
        /* sum the K-parts through LDS (two buffers in turn: one barrier per item) */                             \
        const int buf = round & 1;                                                                                \
        round++;                                                                                                  \
        if (q > 0) {                                                                                              \
            _Pragma("unroll") for (int tt = 0; tt < NT; tt++)                                                     \
                *reinterpret_cast<float4_t*>(red[buf][q - 1][tt][lane]) = acc[tt];                                \
        }                                                                                                         \
        dense_barrier();                                                                                          \
        if (q == 0) {                                                                                             \
            _Pragma("unroll") for (int tt = 0; tt < NT; tt++) {                                                   \
                _Pragma("unroll") for (int qq = 0; qq < DENSE_KPARTS - 1; qq++)                                   \
                    acc[tt] += *reinterpret_cast<const float4_t*>(red[buf][qq][tt][lane]);                        \
                const int tok = 16 * tt + n;                                                                      \
                if (tok < T && 4 * g < ROWS) {                                                                    \
                    const int col = ROWS * it + 4 * g; /* accumulator rows 4g..4g+3 of the tile = output columns */ \
                    half4_t o;                                                                                    \
                    const half4_t bv = bias ? *reinterpret_cast<const half4_t*>(bias + col) : half4_t{0, 0, 0, 0}; \
                    _Pragma("unroll") for (int r = 0; r < 4; r++) o[r] = (half_t)(acc[tt][r] + (float)bv[r]);     \
                    if (resid) o = o + *reinterpret_cast<const half4_t*>(resid + (int64_t)tok * N + col);         \
                    *reinterpret_cast<half4_t*>(out + (int64_t)tok * N + col) = o;                                \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
    }
    int round = 0;
    for (int itx = it0; itx < it1 - 1; itx++) DENSE_ITEM(itx, DENSE_LOAD)
    DENSE_ITEM(it1 - 1, DENSE_NO_LOAD)
#undef DENSE_ITEM
#undef DENSE_NO_LOAD
#undef DENSE_LOAD
}

// The same product when a workgroup's share of W is at most two row tiles (N <= 32 rows per CU: every projection of a
// gpt-oss layer): the rows are dealt out four at a time, so every CU streams the same number of bytes whatever N is (5120
// rows are 20 per CU instead of one or two whole 16-row tiles, 2880 rows 8 or 12 instead of 16 on 180 of the CUs), and
// ALL of a workgroup's weights are requested at once, ahead of the activations: one trip to memory per launch instead
// of one per item.  Lanes whose row lies past the workgroup's share load nothing.  Same K-parts, same order of sums:
// the bits of dense_stream_kernel.
template <int STEPS, int NT, int TILES>
__global__ __launch_bounds__(DENSE_THREADS) void dense_rows_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                                   const half_t* __restrict__ bias,
                                                                   const half_t* __restrict__ resid, int T, int N, int K,
                                                                   half_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float red[TILES][DENSE_KPARTS][NT][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int NB = K / 64, units = N / 4;
    const int r0 = 4 * (int)((int64_t)blockIdx.x * units / gridDim.x), r1 = 4 * (int)((int64_t)(blockIdx.x + 1) * units / gridDim.x);
    if (r0 >= r1) return;
    const int b0 = q * NB / DENSE_KPARTS, b1 = (q + 1) * NB / DENSE_KPARTS;   // this wave's blocks

    // A fragments of both tiles: row r0 + 16 tile + (lane & 15), columns 64 * block + 16 * g (+ 8)
    u32x4_t A[TILES][STEPS][2];
#pragma unroll
    for (int tile = 0; tile < TILES; tile++) {
        const int row = r0 + 16 * tile + n;
        const unsigned char* rp = reinterpret_cast<const unsigned char*>(w) + (size_t)min(row, r1 - 1) * K * 2 + 32 * g;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            A[tile][s][0] = z;
            A[tile][s][1] = z;
            if (row < r1) {
                const u32x4_t* src = reinterpret_cast<const u32x4_t*>(rp + (size_t)min(b0 + s, NB - 1) * 128);
                A[tile][s][0] = __builtin_nontemporal_load(src);
                A[tile][s][1] = __builtin_nontemporal_load(src + 1);
            }
        }
    }
    // B fragments: token (16 * tile + n, clamped), the same columns; zero where the slot is outside this wave's part
    half8_t xb[NT][STEPS][2];
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
        const int tok = min(16 * tt + n, T - 1);
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const half8_t* src = reinterpret_cast<const half8_t*>(x + (int64_t)tok * K + min(b0 + s, NB - 1) * 64 + 16 * g);
            const bool live = b0 + s < b1;
            const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            xb[tt][s][0] = live ? src[0] : z;
            xb[tt][s][1] = live ? src[1] : z;
        }
    }
#pragma unroll
    for (int tile = 0; tile < TILES; tile++) {
        if (r0 + 16 * tile >= r1) break;
        float4_t acc[NT];
#pragma unroll
        for (int tt = 0; tt < NT; tt++) acc[tt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const half8_t a0 = __builtin_bit_cast(half8_t, A[tile][s][0]), a1 = __builtin_bit_cast(half8_t, A[tile][s][1]);
#pragma unroll
            for (int tt = 0; tt < NT; tt++) {
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xb[tt][s][0], acc[tt], 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xb[tt][s][1], acc[tt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tt = 0; tt < NT; tt++) *reinterpret_cast<float4_t*>(red[tile][q][tt][lane]) = acc[tt];
    }
    dense_barrier();
    // wave `tile` finishes tile `tile`: the K-parts summed in part order, then bias, rounding, residual
    if (q < TILES && r0 + 16 * q < r1) {
#pragma unroll
        for (int tt = 0; tt < NT; tt++) {
            float4_t acc = *reinterpret_cast<const float4_t*>(red[q][0][tt][lane]);
#pragma unroll
            for (int qq = 1; qq < DENSE_KPARTS; qq++) acc += *reinterpret_cast<const float4_t*>(red[q][qq][tt][lane]);
            const int tok = 16 * tt + n;
            const int col = r0 + 16 * q + 4 * g;   // accumulator rows 4g..4g+3 of the tile = output columns
            if (tok < T && col < r1) {
                half4_t o;
                const half4_t bv = bias ? *reinterpret_cast<const half4_t*>(bias + col) : half4_t{0, 0, 0, 0};
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] = (half_t)(acc[r] + (float)bv[r]);
                if (resid) o = o + *reinterpret_cast<const half4_t*>(resid + (int64_t)tok * N + col);
                *reinterpret_cast<half4_t*>(out + (int64_t)tok * N + col) = o;
            }
        }
    }
}

// A handful of tokens (T <= 4: single-sequence decode).  The 16-row tiles of the kernel above are too coarse a unit
// there: 5120 rows are 320 tiles for 256 CUs, two rounds for 1.25 rounds of work, and a CU cannot stream faster than its
// own share of the HBM bandwidth.  Here the unit is half a row of W: a workgroup owns a contiguous run of rows (N / grid,
// so every CU streams the same number of bytes), its 8 waves take (row, K-half) units round-robin -- a wave keeps one
// K-half, so its slice of x stays in registers --, every lane multiplies 16-byte pieces with v_dot2_f32_f16, a wave-wide
// sum finishes a unit, and the two halves of a row meet in LDS.  All of a wave's loads of a batch of rows are requested
// before the first is used.  Same roundings as above: fp16(acc + bias), the residual added in fp16.
constexpr int GEMV_MAX_T = 4;
constexpr int GEMV_ROWS = 24;   // rows of a batch: 48 units, 6 per wave
constexpr int GEMV_UNITS = GEMV_ROWS * 2 / DENSE_KPARTS;

template <int TN, int J>   // tokens; 16-byte pieces per lane and K-half (K / 2 <= 512 J elements)
__global__ __launch_bounds__(DENSE_THREADS) void dense_gemv_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                                   const half_t* __restrict__ bias,
                                                                   const half_t* __restrict__ resid, int N, int K,
                                                                   half_t* __restrict__ out) {
    __shared__ float part[GEMV_ROWS][2][TN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = (int)((int64_t)blockIdx.x * N / gridDim.x), r1 = (int)((int64_t)(blockIdx.x + 1) * N / gridDim.x);
    const int kh = wv & 1;                  // this wave's K-half
    const int C = K / 16;                   // 16-byte pieces per half row
    const half_t* xh = x + kh * (K / 2);
    const half_t* wh = w + kh * (K / 2);
    // the wave's slice of x: piece lane + 64 j of its K-half, every token
    u32x4_t xr[TN][J];
#pragma unroll
    for (int t = 0; t < TN; t++)
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            xr[t][j] = c < C ? reinterpret_cast<const u32x4_t*>(xh + (int64_t)t * K)[c] : u32x4_t{0u, 0u, 0u, 0u};
        }
    for (int rb = r0; rb < r1; rb += GEMV_ROWS) {
        const int rows = min(GEMV_ROWS, r1 - rb);
        u32x4_t a[GEMV_UNITS][J];
#pragma unroll
        for (int u = 0; u < GEMV_UNITS; u++) {
            const int lr = (wv >> 1) + (DENSE_KPARTS / 2) * u;   // local row of the wave's u-th unit
            if (lr < rows) {                                     // (uniform over the wave)
                const half_t* row = wh + (int64_t)(rb + lr) * K;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c = min(lane + 64 * j, C - 1);     // (pieces past the half row meet a zero in xr)
                    a[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(row) + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < GEMV_UNITS; u++) {
            const int lr = (wv >> 1) + (DENSE_KPARTS / 2) * u;
            if (lr >= rows) break;
            float acc[TN];
#pragma unroll
            for (int t = 0; t < TN; t++) {
                acc[t] = 0.f;
#pragma unroll
                for (int j = 0; j < J; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        acc[t] = __builtin_amdgcn_fdot2(as_half2(a[u][j][i]), as_half2(xr[t][j][i]), acc[t], false);
                acc[t] = wave_sum(acc[t]);
            }
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < TN; t++) part[lr][kh][t] = acc[t];
            }
        }
        __syncthreads();
        if (tid < rows * TN) {
            const int lr = tid / TN, t = tid - lr * TN;
            const int col = rb + lr;
            half_t o = (half_t)(part[lr][0][t] + part[lr][1][t] + (bias ? (float)bias[col] : 0.f));
            if (resid) o = o + resid[(int64_t)t * N + col];
            out[(int64_t)t * N + col] = o;
        }
        __syncthreads();
    }
}

// The q/k/v projection of up to 4 tokens with what follows it in a decoder layer -- rotary embedding of q and k, q to
// its own buffer, k and v into the paged cache -- in the projection's launch.  The half-row kernel above, with the rows
// taken as pairs (d, d + D/2) of a head (the two values a rotation mixes; v rows pair up the same way and pass through),
// so the thread that adds a pair's K-halves has both values at hand.  Same sums, same roundings: the bits of
// m355_linear_f16 followed by m355_rope_kv_write.
struct RopeKvArgs {
    const half_t *cos_t, *sin_t;
    const int32_t* positions;
    const int64_t* slots;
    int Hq, Hkv, D, block_size, max_pos;
    half_t *q_out, *kcache, *vcache;
};

// NORM 1 / 2: x is the residual stream and the projection's input is its RMSNorm, formed here -- after delta has been
// added (1), or the expert outputs that a forward called with M355_FLAG_SKIP_COMBINE left in its workspace (2), as
// m355_add_rmsnorm / m355_moe_combine_add_rmsnorm do, by the same device code: every workgroup normalises the few rows
// for itself into LDS while its first weights are on their way, and workgroup 0 writes the new residual stream.
struct NormArgs {
    const half_t *delta, *weight;
    float eps;
    half_t* resid_out;
    const half_t* cy;
    const int32_t* cinv;
    const float* cw;
    int k_top, fp16_mode;
};

template <int TN, int J, int NORM>
__global__ __launch_bounds__(DENSE_THREADS) void qkv_rope_gemv_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                                      const half_t* __restrict__ bias, int N, int K,
                                                                      RopeKvArgs ra, NormArgs na) {
    extern __shared__ __attribute__((aligned(16))) half_t yrows[];   // NORM: [TN][K] normalised rows
    __shared__ float part[GEMV_ROWS][2][TN];
    __shared__ float norm_part[TN][NORM_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half_d = ra.D / 2, n_pairs = N / 2;
    const int p0 = (int)((int64_t)blockIdx.x * n_pairs / gridDim.x), p1 = (int)((int64_t)(blockIdx.x + 1) * n_pairs / gridDim.x);
    const int kh = wv & 1;
    const int C = K / 16;
    const half_t* xh = x + kh * (K / 2);
    const half_t* wh = w + kh * (K / 2);
    // local row lr of a batch that starts at pair pb: pair pb + lr / 2, its first (even lr) or second (odd lr) row
    auto row_of = [&](int pb, int lr) {
        const int p = pb + (lr >> 1);
        const int head = p / half_d, d = p - head * half_d;
        return head * ra.D + d + (lr & 1) * half_d;
    };
    u32x4_t xr[TN][J];
    u32x4_t a[GEMV_UNITS][J];
    auto request_rows = [&](int pb) {   // the weights of the batch of pairs that starts at pb
        const int rows = 2 * min(GEMV_ROWS / 2, p1 - pb);
#pragma unroll
        for (int u = 0; u < GEMV_UNITS; u++) {
            const int lr = (wv >> 1) + (DENSE_KPARTS / 2) * u;
            if (lr < rows) {
                const half_t* row = wh + (int64_t)row_of(pb, lr) * K;
#pragma unroll
                for (int j = 0; j < J; j++) {
                    const int c = min(lane + 64 * j, C - 1);
                    a[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(row) + c);
                }
            }
        }
    };
    if constexpr (NORM != 0) {
        if (p0 < p1) request_rows(p0);
#pragma unroll
        for (int t = 0; t < TN; t++)
            add_rmsnorm_row<NORM == 2 ? 1 : 0>(tid < NORM_THREADS, t, tid, x, na.delta, na.weight, na.eps, K,
                                               blockIdx.x == 0 ? na.resid_out : nullptr, nullptr, na.cy, na.cinv, na.cw, na.k_top,
                                               na.fp16_mode, norm_part[t], yrows + (size_t)t * K);
        __syncthreads();
        xh = yrows + kh * (K / 2);
    }
#pragma unroll
    for (int t = 0; t < TN; t++)
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int c = lane + 64 * j;
            xr[t][j] = c < C ? reinterpret_cast<const u32x4_t*>(xh + (int64_t)t * K)[c] : u32x4_t{0u, 0u, 0u, 0u};
        }
    for (int pb = p0; pb < p1; pb += GEMV_ROWS / 2) {
        const int pairs = min(GEMV_ROWS / 2, p1 - pb), rows = 2 * pairs;
        if (NORM == 0 || pb != p0) request_rows(pb);
#pragma unroll
        for (int u = 0; u < GEMV_UNITS; u++) {
            const int lr = (wv >> 1) + (DENSE_KPARTS / 2) * u;
            if (lr >= rows) break;
            float acc[TN];
#pragma unroll
            for (int t = 0; t < TN; t++) {
                acc[t] = 0.f;
#pragma unroll
                for (int j = 0; j < J; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        acc[t] = __builtin_amdgcn_fdot2(as_half2(a[u][j][i]), as_half2(xr[t][j][i]), acc[t], false);
                acc[t] = wave_sum(acc[t]);
            }
            if (lane == 0) {
#pragma unroll
                for (int t = 0; t < TN; t++) part[lr][kh][t] = acc[t];
            }
        }
        __syncthreads();
        if (tid < pairs * TN) {
            const int lp = tid / TN, t = tid - lp * TN;
            const int ra_row = row_of(pb, 2 * lp), rb_row = ra_row + half_d;
            const half_t va = (half_t)(part[2 * lp][0][t] + part[2 * lp][1][t] + (bias ? (float)bias[ra_row] : 0.f));
            const half_t vb = (half_t)(part[2 * lp + 1][0][t] + part[2 * lp + 1][1][t] + (bias ? (float)bias[rb_row] : 0.f));
            const int head = ra_row / ra.D, d = ra_row - head * ra.D;
            const int64_t slot = ra.slots[t];
            const int64_t cache_row = slot >= 0 ? ((slot / ra.block_size) * ra.block_size + slot % ra.block_size) * (int64_t)ra.Hkv * ra.D : -1;
            if (head < ra.Hq + ra.Hkv) {
                const int pos = min(max(ra.positions[t], 0), ra.max_pos - 1);
                half_t o1, o2;
                rope_rotate(va, vb, ra.cos_t[(int64_t)pos * half_d + d], ra.sin_t[(int64_t)pos * half_d + d], o1, o2);
                half_t* dst = nullptr;
                if (head < ra.Hq) dst = ra.q_out + ((int64_t)t * ra.Hq + head) * ra.D;
                else if (cache_row >= 0) dst = ra.kcache + cache_row + (head - ra.Hq) * ra.D;
                if (dst) { dst[d] = o1; dst[d + half_d] = o2; }
            } else if (cache_row >= 0) {
                half_t* dst = ra.vcache + cache_row + (head - ra.Hq - ra.Hkv) * ra.D;
                dst[d] = va;
                dst[d + half_d] = vb;
            }
        }
        __syncthreads();
    }
}

// The q/k/v projection of 5 to 32 tokens with rotary embedding and cache write in the projection's launch: the kernel
// above with the rows of a tile taken as 8 pairs (d, d + D/2) of a head -- tile rows 0..7 the first halves, 8..15 their
// partners -- so the lane that finishes rows 4g..4g+3 of a tile (g < 2) finds the partners' sums under lane + 32 of the
// same reduction buffers.  The pairs are dealt out four at a time (a lane's four consecutive d share one 8-byte store);
// a workgroup takes at most 16 pairs = two tiles.  Same K-parts, same order of sums, same roundings: the bits of
// m355_linear_f16 followed by m355_rope_kv_write.
template <int STEPS, int NT>
__global__ __launch_bounds__(DENSE_THREADS) void qkv_rope_rows_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                                      const half_t* __restrict__ bias, int T, int N, int K,
                                                                      RopeKvArgs ra) {
    constexpr int TILES = 2;
    __shared__ __attribute__((aligned(16))) float red[TILES][DENSE_KPARTS][NT][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int NB = K / 64, half_d = ra.D / 2, units = N / 8;   // a unit = 4 pairs
    const int p0 = 4 * (int)((int64_t)blockIdx.x * units / gridDim.x), p1 = 4 * (int)((int64_t)(blockIdx.x + 1) * units / gridDim.x);
    if (p0 >= p1) return;
    const int b0 = q * NB / DENSE_KPARTS, b1 = (q + 1) * NB / DENSE_KPARTS;
    // row of W behind tile row n of tile `tile`: pair p0 + 8 tile + (n & 7), its first (n < 8) or second half
    auto pair_row = [&](int p, int second) {
        const int head = p / half_d, d = p - head * half_d;
        return head * ra.D + d + second * half_d;
    };
    u32x4_t A[TILES][STEPS][2];
#pragma unroll
    for (int tile = 0; tile < TILES; tile++) {
        const int p = p0 + 8 * tile + (n & 7);
        const unsigned char* rp = reinterpret_cast<const unsigned char*>(w) + (size_t)pair_row(min(p, p1 - 1), n >> 3) * K * 2 + 32 * g;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const u32x4_t z = {0u, 0u, 0u, 0u};
            A[tile][s][0] = z;
            A[tile][s][1] = z;
            if (p < p1) {
                const u32x4_t* src = reinterpret_cast<const u32x4_t*>(rp + (size_t)min(b0 + s, NB - 1) * 128);
                A[tile][s][0] = __builtin_nontemporal_load(src);
                A[tile][s][1] = __builtin_nontemporal_load(src + 1);
            }
        }
    }
    half8_t xb[NT][STEPS][2];
#pragma unroll
    for (int tt = 0; tt < NT; tt++) {
        const int tok = min(16 * tt + n, T - 1);
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const half8_t* src = reinterpret_cast<const half8_t*>(x + (int64_t)tok * K + min(b0 + s, NB - 1) * 64 + 16 * g);
            const bool live = b0 + s < b1;
            const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            xb[tt][s][0] = live ? src[0] : z;
            xb[tt][s][1] = live ? src[1] : z;
        }
    }
#pragma unroll
    for (int tile = 0; tile < TILES; tile++) {
        if (p0 + 8 * tile >= p1) break;
        float4_t acc[NT];
#pragma unroll
        for (int tt = 0; tt < NT; tt++) acc[tt] = float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            const half8_t a0 = __builtin_bit_cast(half8_t, A[tile][s][0]), a1 = __builtin_bit_cast(half8_t, A[tile][s][1]);
#pragma unroll
            for (int tt = 0; tt < NT; tt++) {
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xb[tt][s][0], acc[tt], 0, 0, 0);
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xb[tt][s][1], acc[tt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int tt = 0; tt < NT; tt++) *reinterpret_cast<float4_t*>(red[tile][q][tt][lane]) = acc[tt];
    }
    dense_barrier();
    // wave (tile, tt) = (q & 1, q >> 1) finishes: lanes g < 2 hold pairs 4g .. 4g + 3 of the tile, first halves under
    // their own lane, partners under lane + 32
    const int tile = q & 1, tt = q >> 1;
    if (tt < NT && g < 2 && p0 + 8 * tile + 4 * g < p1) {
        float4_t fa = *reinterpret_cast<const float4_t*>(red[tile][0][tt][lane]);
        float4_t fb = *reinterpret_cast<const float4_t*>(red[tile][0][tt][lane + 32]);
#pragma unroll
        for (int qq = 1; qq < DENSE_KPARTS; qq++) {
            fa += *reinterpret_cast<const float4_t*>(red[tile][qq][tt][lane]);
            fb += *reinterpret_cast<const float4_t*>(red[tile][qq][tt][lane + 32]);
        }
        const int tok = 16 * tt + n;
        if (tok < T) {
            const int p = p0 + 8 * tile + 4 * g;                 // first of the lane's four pairs (one head: half_d % 4 == 0)
            const int ra_row = pair_row(p, 0), rb_row = ra_row + half_d;
            const int head = ra_row / ra.D, d = ra_row - head * ra.D;
            half4_t va, vb;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                va[r] = (half_t)(fa[r] + (bias ? (float)bias[ra_row + r] : 0.f));
                vb[r] = (half_t)(fb[r] + (bias ? (float)bias[rb_row + r] : 0.f));
            }
            const int64_t slot = ra.slots[tok];
            const int64_t cache_row = slot >= 0 ? ((slot / ra.block_size) * ra.block_size + slot % ra.block_size) * (int64_t)ra.Hkv * ra.D : -1;
            half_t* dst = nullptr;
            if (head < ra.Hq + ra.Hkv) {
                const int pos = min(max(ra.positions[tok], 0), ra.max_pos - 1);
                const half4_t cs = *reinterpret_cast<const half4_t*>(ra.cos_t + (int64_t)pos * half_d + d);
                const half4_t sn = *reinterpret_cast<const half4_t*>(ra.sin_t + (int64_t)pos * half_d + d);
                half4_t o1, o2;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    half_t u1, u2;
                    rope_rotate(va[r], vb[r], cs[r], sn[r], u1, u2);
                    o1[r] = u1;
                    o2[r] = u2;
                }
                va = o1;
                vb = o2;
                if (head < ra.Hq) dst = ra.q_out + ((int64_t)tok * ra.Hq + head) * ra.D;
                else if (cache_row >= 0) dst = ra.kcache + cache_row + (head - ra.Hq) * ra.D;
            } else if (cache_row >= 0) {
                dst = ra.vcache + cache_row + (head - ra.Hq - ra.Hkv) * ra.D;
            }
            if (dst) {
                *reinterpret_cast<half4_t*>(dst + d) = va;
                *reinterpret_cast<half4_t*>(dst + d + half_d) = vb;
            }
        }
    }
}

template <int STEPS, int NT>
static int launch_qkv_rope_rows(const half_t* x, const half_t* w, const half_t* bias, int T, int N, int K, const RopeKvArgs& ra,
                                int cus, hipStream_t st) {
    const int units = N / 8;
    const int grid = units < cus ? units : cus;
    qkv_rope_rows_kernel<STEPS, NT><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, T, N, K, ra);
    M355_LAUNCH_CHECK("qkv_rope_rows_kernel");
    return 0;
}

template <int TN, int NORM>
static int launch_qkv_rope(const half_t* x, const half_t* w, const half_t* bias, int N, int K, const RopeKvArgs& ra,
                           const NormArgs& na, int cus, hipStream_t st) {
    const int grid = N / 2 < cus ? N / 2 : cus;
    const int J = (K / 16 + 63) / 64;
    const size_t lds = NORM ? (size_t)TN * K * 2 : 0;
    switch (J) {
    case 1: qkv_rope_gemv_kernel<TN, 1, NORM><<<dim3(grid), dim3(DENSE_THREADS), lds, st>>>(x, w, bias, N, K, ra, na); break;
    case 2: qkv_rope_gemv_kernel<TN, 2, NORM><<<dim3(grid), dim3(DENSE_THREADS), lds, st>>>(x, w, bias, N, K, ra, na); break;
    case 3: qkv_rope_gemv_kernel<TN, 3, NORM><<<dim3(grid), dim3(DENSE_THREADS), lds, st>>>(x, w, bias, N, K, ra, na); break;
    default: qkv_rope_gemv_kernel<TN, 4, NORM><<<dim3(grid), dim3(DENSE_THREADS), lds, st>>>(x, w, bias, N, K, ra, na); break;
    }
    M355_LAUNCH_CHECK("qkv_rope_gemv_kernel");
    return 0;
}
template <int NORM>
static int launch_qkv_rope_t(int T, const half_t* x, const half_t* w, const half_t* bias, int N, int K, const RopeKvArgs& ra,
                             const NormArgs& na, int cus, hipStream_t st) {
    switch (T) {
    case 1: return launch_qkv_rope<1, NORM>(x, w, bias, N, K, ra, na, cus, st);
    case 2: return launch_qkv_rope<2, NORM>(x, w, bias, N, K, ra, na, cus, st);
    case 3: return launch_qkv_rope<3, NORM>(x, w, bias, N, K, ra, na, cus, st);
    default: return launch_qkv_rope<4, NORM>(x, w, bias, N, K, ra, na, cus, st);
    }
}

template <int TN>
static int launch_gemv(const half_t* x, const half_t* w, const half_t* bias, const half_t* resid, int N, int K, half_t* out,
                       int cus, hipStream_t st) {
    const int grid = N < cus ? N : cus;
    const int J = (K / 16 + 63) / 64;
    switch (J) {
    case 1: dense_gemv_kernel<TN, 1><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, resid, N, K, out); break;
    case 2: dense_gemv_kernel<TN, 2><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, resid, N, K, out); break;
    case 3: dense_gemv_kernel<TN, 3><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, resid, N, K, out); break;
    default: dense_gemv_kernel<TN, 4><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, resid, N, K, out); break;
    }
    M355_LAUNCH_CHECK("dense_gemv_kernel");
    return 0;
}

template <int STEPS, int NT>
static int launch_dense(const half_t* x, const half_t* w, const half_t* bias, const half_t* resid, int T, int N, int K,
                        half_t* out, int cus, hipStream_t st) {
    const int tiles = N / 16;
    // (8-row items -- ROWS = 8 -- even out shapes like 5120 rows = 320 tiles on 256 CUs, and measure slower: 16.3 against
    // 14.7 us at 32 tokens, 11.8 against 10.4 at 5: an item's fixed cost outweighs the shorter last round)
    const int grid = tiles < cus ? tiles : cus;
    dense_stream_kernel<STEPS, NT, 16><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, resid, T, N, K, out);
    M355_LAUNCH_CHECK("dense_stream_kernel");
    return 0;
}

template <int STEPS, int NT>
static int launch_dense_rows(const half_t* x, const half_t* w, const half_t* bias, const half_t* resid, int T, int N, int K,
                             half_t* out, int cus, hipStream_t st) {
    const int units = N / 4;
    const int grid = units < cus ? units : cus;
    const int max_rows = 4 * ((units + grid - 1) / grid);
    if (max_rows <= 16) {
        dense_rows_kernel<STEPS, NT, 1><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, resid, T, N, K, out);
    } else if constexpr (STEPS == 8 && NT == 2) {
        return fail(-2, "m355_linear_f16: two row tiles of K > 3072 for more than 16 tokens do not fit the registers");   // (the caller does not send this shape here)
    } else {
        dense_rows_kernel<STEPS, NT, 2><<<dim3(grid), dim3(DENSE_THREADS), 0, st>>>(x, w, bias, resid, T, N, K, out);
    }
    M355_LAUNCH_CHECK("dense_rows_kernel");
    return 0;
}

// many tokens: the tiled kernel of dense_tile.hip
int launch_dense_tile(const half_t* x, const half_t* w, const half_t* bias, const half_t* resid, int T, int N, int K,
                      half_t* out, int cus, hipStream_t st);
constexpr int DENSE_TILE_MIN_T = 128;   // from here on re-streaming W once per 32 tokens costs more than tiling

}  // namespace m355

using namespace m355;

extern "C" int m355_linear_f16(const void* x_f16, int T, const void* w_f16, const void* bias_f16, int N, int K,
                               const void* resid_f16, void* out_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !w_f16 || !out_f16) return fail(-1, "m355_linear_f16: null pointer");
    if (T < 0 || N <= 0 || K <= 0 || (N % 16) || (K % 64) || K > 64 * DENSE_KPARTS * 8)
        return fail(-2, "m355_linear_f16: unsupported shape T=%d N=%d K=%d (need N %% 16 == 0, K %% 64 == 0, K <= %d)", T, N, K,
                    64 * DENSE_KPARTS * 8);
    if (!aligned(x_f16, 16) || !aligned(w_f16, 16) || !aligned(out_f16, 8) || (bias_f16 && !aligned(bias_f16, 8)) ||
        (resid_f16 && !aligned(resid_f16, 8)))
        return fail(-3, "m355_linear_f16: x and w must be 16-byte aligned, bias / resid / out 8-byte aligned");
    int dev = 0, cus = 0;
    if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
    if (int rc = hip_status(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return rc;
    if (cus <= 0) cus = 256;
    hipStream_t st = (hipStream_t)stream;
    const half_t *x = (const half_t*)x_f16, *w = (const half_t*)w_f16, *b = (const half_t*)bias_f16, *r = (const half_t*)resid_f16;
    half_t* out = (half_t*)out_f16;
    if (T >= DENSE_TILE_MIN_T) return launch_dense_tile(x, w, b, r, T, N, K, out, cus, st);
    if (T <= GEMV_MAX_T && K <= 64 * 64 * 4) {   // at most 4 pieces of 8 elements per lane and K-half
        switch (T) {
        case 1: return launch_gemv<1>(x, w, b, r, N, K, out, cus, st);
        case 2: return launch_gemv<2>(x, w, b, r, N, K, out, cus, st);
        case 3: return launch_gemv<3>(x, w, b, r, N, K, out, cus, st);
        default: return launch_gemv<4>(x, w, b, r, N, K, out, cus, st);
        }
    }
    const bool small_k = K <= 64 * DENSE_KPARTS * 6;
#ifndef M355_EXP_NO_DENSE_ROWS
    // at most 32 rows of W per CU: every CU takes the same share of the rows and requests all of it at once
    if (T <= 32 && N <= 32 * cus && (small_k || N <= 16 * cus || T <= 16)) {
        if (T <= 16) return small_k ? launch_dense_rows<6, 1>(x, w, b, r, T, N, K, out, cus, st) : launch_dense_rows<8, 1>(x, w, b, r, T, N, K, out, cus, st);
        return small_k ? launch_dense_rows<6, 2>(x, w, b, r, T, N, K, out, cus, st) : launch_dense_rows<8, 2>(x, w, b, r, T, N, K, out, cus, st);
    }
#endif
    // 32 tokens per pass; more tokens re-stream W once per pass (decode batches above 32 are the exception here)
    for (int t0 = 0; t0 < T; t0 += 32) {
        const int tn = T - t0 < 32 ? T - t0 : 32;
        const half_t* xr = x + (int64_t)t0 * K;
        const half_t* rr = r ? r + (int64_t)t0 * N : nullptr;
        half_t* orow = out + (int64_t)t0 * N;
        int rc;
        if (tn <= 16) rc = small_k ? launch_dense<6, 1>(xr, w, b, rr, tn, N, K, orow, cus, st) : launch_dense<8, 1>(xr, w, b, rr, tn, N, K, orow, cus, st);
        else rc = small_k ? launch_dense<6, 2>(xr, w, b, rr, tn, N, K, orow, cus, st) : launch_dense<8, 2>(xr, w, b, rr, tn, N, K, orow, cus, st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int m355_qkv_rope_kv_write(const void* x_f16, int T, const void* wqkv_f16, const void* bqkv_f16, int K,
                                      const void* cos_f16, const void* sin_f16, const int32_t* positions, const int64_t* slots,
                                      int Hq, int Hkv, int D, int block_size, int max_pos, void* qkv_scratch_f16,
                                      void* q_out_f16, void* kcache_f16, void* vcache_f16, m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !wqkv_f16 || !cos_f16 || !sin_f16 || !positions || !slots || !q_out_f16 || !kcache_f16 || !vcache_f16)
        return fail(-1, "m355_qkv_rope_kv_write: null pointer");
    if (T < 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || (D % 16) || block_size <= 0 || max_pos <= 0 || K <= 0 || (K % 64))
        return fail(-2, "m355_qkv_rope_kv_write: unsupported shape T=%d Hq=%d Hkv=%d D=%d K=%d block_size=%d (need D %% 16 == 0, K %% 64 == 0)",
                    T, Hq, Hkv, D, K, block_size);
    const int N = (Hq + 2 * Hkv) * D;
    if (T <= GEMV_MAX_T && K <= 64 * 64) {
        if (!aligned(x_f16, 16) || !aligned(wqkv_f16, 16) || (bqkv_f16 && !aligned(bqkv_f16, 2)) || !aligned(cos_f16, 2) ||
            !aligned(sin_f16, 2) || !aligned(q_out_f16, 2) || !aligned(kcache_f16, 2) || !aligned(vcache_f16, 2))
            return fail(-3, "m355_qkv_rope_kv_write: x and wqkv must be 16-byte aligned");
        int dev = 0, cus = 0;
        if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
        if (int rc = hip_status(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return rc;
        if (cus <= 0) cus = 256;
        const RopeKvArgs ra = {(const half_t*)cos_f16, (const half_t*)sin_f16, positions, slots, Hq, Hkv, D, block_size, max_pos,
                               (half_t*)q_out_f16, (half_t*)kcache_f16, (half_t*)vcache_f16};
        const half_t *x = (const half_t*)x_f16, *w = (const half_t*)wqkv_f16, *b = (const half_t*)bqkv_f16;
        hipStream_t st = (hipStream_t)stream;
        return launch_qkv_rope_t<0>(T, x, w, b, N, K, ra, NormArgs{}, cus, st);
    }
    // up to 32 tokens, at most 16 (d, d + D/2) pairs of W per CU: still one launch
    {
        int dev = 0, cus = 0;
        if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
        if (int rc = hip_status(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return rc;
        if (cus <= 0) cus = 256;
        const int units = N / 8, grid = units < cus ? units : cus;
        const bool small_k = K <= 64 * DENSE_KPARTS * 6;
        if (T <= 32 && (D % 8) == 0 && K <= 64 * DENSE_KPARTS * 8 && (units + grid - 1) / grid <= 4 && (small_k || T <= 16) &&
            aligned(x_f16, 16) && aligned(wqkv_f16, 16) && (!bqkv_f16 || aligned(bqkv_f16, 2)) && aligned(cos_f16, 8) &&
            aligned(sin_f16, 8) && aligned(q_out_f16, 8) && aligned(kcache_f16, 8) && aligned(vcache_f16, 8)) {
            const RopeKvArgs ra = {(const half_t*)cos_f16, (const half_t*)sin_f16, positions, slots, Hq, Hkv, D, block_size, max_pos,
                                   (half_t*)q_out_f16, (half_t*)kcache_f16, (half_t*)vcache_f16};
            const half_t *x = (const half_t*)x_f16, *w = (const half_t*)wqkv_f16, *b = (const half_t*)bqkv_f16;
            hipStream_t st = (hipStream_t)stream;
            if (T <= 16) return small_k ? launch_qkv_rope_rows<6, 1>(x, w, b, T, N, K, ra, cus, st) : launch_qkv_rope_rows<8, 1>(x, w, b, T, N, K, ra, cus, st);
            return launch_qkv_rope_rows<6, 2>(x, w, b, T, N, K, ra, cus, st);
        }
    }
    // more tokens: the two launches, through the caller's qkv buffer
    if (!qkv_scratch_f16) return fail(-1, "m355_qkv_rope_kv_write: this many tokens (or this shape) need the qkv scratch buffer");
    if (int rc = m355_linear_f16(x_f16, T, wqkv_f16, bqkv_f16, N, K, nullptr, qkv_scratch_f16, stream)) return rc;
    return m355_rope_kv_write(qkv_scratch_f16, cos_f16, sin_f16, positions, slots, T, Hq, Hkv, D, block_size, max_pos, q_out_f16,
                              kcache_f16, vcache_f16, stream);
}

extern "C" int m355_norm_qkv_rope_kv_write(const void* x_f16, const void* delta_f16, const void* moe_workspace, size_t ws_bytes,
                                           int k, int FF, int E_local, const float* combine_w, int flags,
                                           const void* ln_weight_f16, float eps, int T, int K, const void* wqkv_f16,
                                           const void* bqkv_f16, const void* cos_f16, const void* sin_f16,
                                           const int32_t* positions, const int64_t* slots, int Hq, int Hkv, int D,
                                           int block_size, int max_pos, void* resid_out_f16, void* h_scratch_f16,
                                           void* qkv_scratch_f16, void* q_out_f16, void* kcache_f16, void* vcache_f16,
                                           m355_stream_t stream) {
    if (T == 0) return 0;
    if (!x_f16 || !ln_weight_f16 || !wqkv_f16 || !cos_f16 || !sin_f16 || !positions || !slots || !q_out_f16 || !kcache_f16 ||
        !vcache_f16 || (moe_workspace && !combine_w))
        return fail(-1, "m355_norm_qkv_rope_kv_write: null pointer");
    if (moe_workspace && delta_f16) return fail(-2, "m355_norm_qkv_rope_kv_write: delta and a pending combine exclude each other");
    if (T < 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || (D % 16) || block_size <= 0 || max_pos <= 0 || K <= 0 || (K % 64) ||
        K > 8 * NORM_THREADS * 4 || (moe_workspace && (k <= 0 || k > 32 || FF <= 0 || E_local <= 0)))
        return fail(-2, "m355_norm_qkv_rope_kv_write: unsupported shape T=%d Hq=%d Hkv=%d D=%d K=%d block_size=%d k=%d", T, Hq, Hkv, D,
                    K, block_size, k);
    const int N = (Hq + 2 * Hkv) * D;
    if (T <= GEMV_MAX_T && K <= 64 * 64) {
        if (!aligned(x_f16, 16) || !aligned(wqkv_f16, 16) || !aligned(ln_weight_f16, 16) || (delta_f16 && !aligned(delta_f16, 16)) ||
            (resid_out_f16 && !aligned(resid_out_f16, 16)) || (moe_workspace && !aligned(moe_workspace, 256)))
            return fail(-3, "m355_norm_qkv_rope_kv_write: x, delta, the norm weight, wqkv and resid_out must be 16-byte aligned, the workspace 256-byte");
        int dev = 0, cus = 0;
        if (int rc = hip_status(hipGetDevice(&dev), "hipGetDevice")) return rc;
        if (int rc = hip_status(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev), "hipDeviceGetAttribute")) return rc;
        if (cus <= 0) cus = 256;
        const RopeKvArgs ra = {(const half_t*)cos_f16, (const half_t*)sin_f16, positions, slots, Hq, Hkv, D, block_size, max_pos,
                               (half_t*)q_out_f16, (half_t*)kcache_f16, (half_t*)vcache_f16};
        NormArgs na = {(const half_t*)delta_f16, (const half_t*)ln_weight_f16, eps, (half_t*)resid_out_f16, nullptr, nullptr, nullptr, 0, 0};
        const half_t *x = (const half_t*)x_f16, *w = (const half_t*)wqkv_f16, *b = (const half_t*)bqkv_f16;
        hipStream_t st = (hipStream_t)stream;
        if (moe_workspace) {
            const MoeWorkspace ws = carve_workspace(const_cast<void*>(moe_workspace), T, k, K, FF, E_local);
            if (ws_bytes < ws.bytes)
                return fail(-6, "m355_norm_qkv_rope_kv_write: the workspace of that forward has %zu bytes, %zu given", ws.bytes, ws_bytes);
            na.cy = ws.y; na.cinv = ws.inv; na.cw = combine_w; na.k_top = k; na.fp16_mode = (flags & M355_FLAG_FP16_COMBINE) ? 1 : 0;
            return launch_qkv_rope_t<2>(T, x, w, b, N, K, ra, na, cus, st);
        }
        return launch_qkv_rope_t<1>(T, x, w, b, N, K, ra, na, cus, st);
    }
    // more tokens: the norm as its own launch into the caller's buffer, then the projection
    if (!h_scratch_f16) return fail(-1, "m355_norm_qkv_rope_kv_write: more than %d tokens need the buffer for the normalised rows", GEMV_MAX_T);
    if (int rc = moe_workspace ? m355_moe_combine_add_rmsnorm(moe_workspace, ws_bytes, T, k, FF, E_local, combine_w, flags, x_f16,
                                                              ln_weight_f16, eps, K, resid_out_f16, h_scratch_f16, stream)
                               : m355_add_rmsnorm(x_f16, delta_f16, ln_weight_f16, eps, T, K, resid_out_f16, h_scratch_f16, stream))
        return rc;
    return m355_qkv_rope_kv_write(h_scratch_f16, T, wqkv_f16, bqkv_f16, K, cos_f16, sin_f16, positions, slots, Hq, Hkv, D, block_size,
                                  max_pos, qkv_scratch_f16, q_out_f16, kcache_f16, vcache_f16, stream);
}
