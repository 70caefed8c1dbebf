// Paged-attention decode for grouped-query heads (8 query heads per KV head, head size 64).
// HBM-bound on the KV read: 2 * Hkv * D * 2 bytes per cached token per sequence.
//
// One workgroup (4 waves) serves one (sequence, KV head, 256-token partition); a wave owns 64 of its tokens.  Both
// products run on the matrix cores (v_mfma_f32_16x16x32_f16) with the 8 query heads on the 16-wide N side, so the 8
// heads of the group share every K and V row that is fetched:
//   S^T[token][head] = K Q^T    A = K rows straight from global memory: lane (r, g) loads the 32 contiguous bytes
//                               K[token r][16 g .. 16 g + 15] (a wave-load covers 16 whole 128-byte rows); the first
//                               MFMA takes the lower 8 columns, the second the upper 8, and Q is laid out the same way
//                               (a dot product does not care in which order its terms are summed).
//   O^T[d][head] += V^T P^T     the scores of a lane (head n, tokens 4 g .. 4 g + 3 of each 16-token tile), turned into
//                               softmax numerators and rounded to fp16 where they sit, ARE the B operand (tokens of two
//                               tiles in the order the lanes hold them); V^T in the same token order comes from the
//                               wave's V rows in LDS through ds_read_b64_tr_b16 (hardware transpose read).
// Each wave normalises against its own maximum; the four waves (and, for longer contexts, the partitions, in a small
// merge kernel) are merged with the usual exp(m_w - m) factors.
#include "common.h"

namespace m355 {

constexpr int ATT_THREADS = 256;
constexpr int ATT_WAVES = ATT_THREADS / 64;
constexpr int ATT_PART = 256;                    // tokens per partition
constexpr int ATT_WTOK = ATT_PART / ATT_WAVES;   // tokens per wave
constexpr int ATT_TILES = ATT_WTOK / 16;         // 16-token MFMA tiles per wave
constexpr int ATT_G = 8;                         // query heads per KV head
constexpr int ATT_D = 64;
static_assert(ATT_TILES % 2 == 0, "the second product takes the tiles in pairs");

typedef short att_short4_t __attribute__((ext_vector_type(4)));
typedef short att_short8_t __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) att_short4_t att_lds_s4_t;

template <bool MULTI>   // MULTI: several rounds per workgroup (else the round loop folds away)
__device__ __forceinline__ void paged_attn_decode_body(
    const half_t* __restrict__ q, const half_t* __restrict__ kc, const half_t* __restrict__ vc,
    const int32_t* __restrict__ block_tables, const int32_t* __restrict__ ctx_lens, int Hq, int Hkv, int block_size,
    int max_blocks, float scale, int nparts, int rounds_arg, float* __restrict__ part_o, float* __restrict__ part_ml,
    half_t* __restrict__ out, const half_t* __restrict__ sinks, int window) {
    const int rounds = MULTI ? rounds_arg : 1;
    // V rows of one round, 128 bytes each; the 32-byte piece p of row t sits at piece position p ^ ((t >> 1) & 3),
    // so the 8 rows x 32 bytes that half a wave reads transposed cover every bank once
    __shared__ __attribute__((aligned(16))) unsigned char vs[ATT_PART * 128];
    // a wave's unnormalised output [head][column] takes the place of its V rows once it has read them (LDS serves a
    // wave's instructions in order)
    auto po = [&](int wv, int h) -> float* { return reinterpret_cast<float*>(vs + wv * ATT_WTOK * 128) + h * ATT_D; };
    __shared__ float pm[ATT_WAVES][ATT_G], pl[ATT_WAVES][ATT_G];                 // the waves' maxima and denominators

    const int b = blockIdx.x, hk = blockIdx.y, part = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    // (a length beyond what the block table covers would index past the table: such tokens are not read)
    const int L = min(ctx_lens[b], max_blocks * block_size);
    // sliding window: the query (position L - 1) sees keys L - window .. L - 1
    const int lo = window > 0 ? max(L - window, 0) : 0;
    const int64_t ob = ((int64_t)b * Hq + hk * ATT_G);
    const int32_t* bt = block_tables + (int64_t)b * max_blocks;
    const int64_t row_stride = (int64_t)Hkv * ATT_D;

    // Q as the B operand: lane (n, g) holds Q[head n][16 g .. 16 g + 15]; columns 8..15 of the product are padding
    u32x4_t qa = {0u, 0u, 0u, 0u}, qb = {0u, 0u, 0u, 0u};
    if (r < ATT_G) {
        const u32x4_t* qp = reinterpret_cast<const u32x4_t*>(q + (ob + r) * ATT_D + 16 * g);
        qa = qp[0];
        qb = qp[1];
    }

    // A partition is `rounds` consecutive rounds of ATT_PART tokens; a wave keeps a running maximum, denominator and
    // output over its 64 tokens of every round (flash-style), so contexts of a few rounds need no second kernel.
    // K and V rows of a round: lane (r, g) takes bytes 32 g .. 32 g + 31 of token r of each of the wave's tiles (tokens
    // past the end re-read the last one and are masked below).  K and V are requested together: one latency, not two.
    u32x4_t ka[ATT_TILES], kb[ATT_TILES], va[ATT_TILES], vb[ATT_TILES];
    auto round_span = [&](int rd, int& t0, int& npt, int& first) {
        t0 = (part * rounds + rd) * ATT_PART;
        first = lo - t0;                      // tokens of the round below `first` are masked
        npt = min(ATT_PART, L - t0);          // tokens of the round below the context end (may be <= 0)
        if (first >= npt) npt = 0;            // the whole round lies before the window
    };
    auto request = [&](int t0, int npt) {
#pragma unroll
        for (int j = 0; j < ATT_TILES; j++) {
            // tokens below the window are masked later; their lanes re-read the window's first row instead, so block
            // table entries and cache rows of tokens that have left the window are never dereferenced
            const int t = max(t0 + min(wave * ATT_WTOK + j * 16 + r, npt - 1), lo);
            const int blk = bt[t / block_size];
            const int64_t row = ((int64_t)blk * block_size + t % block_size) * row_stride + hk * ATT_D + 16 * g;
            ka[j] = reinterpret_cast<const u32x4_t*>(kc + row)[0];
            kb[j] = reinterpret_cast<const u32x4_t*>(kc + row)[1];
            va[j] = reinterpret_cast<const u32x4_t*>(vc + row)[0];
            vb[j] = reinterpret_cast<const u32x4_t*>(vc + row)[1];
        }
    };

    float m_run = -INFINITY, l_run = 0.f;   // of head r over this wave's tokens so far
    float4_t o[ATT_D / 16];                 // O^T: lane (n, g) holds head n, columns 16 dt + 4 g .. + 3
#pragma unroll
    for (int dt = 0; dt < ATT_D / 16; dt++) o[dt] = float4_t{0.f, 0.f, 0.f, 0.f};
    const int ta = r >> 2, tc = r & 3;

    for (int rd = 0; rd < rounds; rd++) {
        int t0, npt_now, first_now;
        round_span(rd, t0, npt_now, first_now);
        if (npt_now <= 0) continue;   // (uniform over the workgroup)
        // (requesting the next round's rows under this round's softmax and second product was tried: it doubles the
        // registers and halves the workgroups per CU)
        request(t0, npt_now);
        // ---- scores: lane (n, g) gets head n, tokens 4 g .. 4 g + 3 of each tile ----
        float4_t sc[ATT_TILES];
#pragma unroll
        for (int j = 0; j < ATT_TILES; j++) {
            const float4_t z = {0.f, 0.f, 0.f, 0.f};
            sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, ka[j]), __builtin_bit_cast(half8_t, qa), z, 0, 0, 0);
            sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, kb[j]), __builtin_bit_cast(half8_t, qb), sc[j], 0, 0, 0);
        }
        // ---- the wave's V rows go to LDS (each wave reads back only its own rows, and LDS serves a wave's
        // instructions in order, so the previous round's transposed reads are done) ----
#pragma unroll
        for (int j = 0; j < ATT_TILES; j++) {
            const int lrow = wave * ATT_WTOK + j * 16 + r;
            unsigned char* dst = vs + lrow * 128 + ((g ^ ((lrow >> 1) & 3)) << 5);
            *reinterpret_cast<u32x4_t*>(dst) = va[j];
            *reinterpret_cast<u32x4_t*>(dst + 16) = vb[j];
        }

        // ---- softmax numerators against the wave's running maximum ----
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < ATT_TILES; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int tl = wave * ATT_WTOK + j * 16 + 4 * g + i;
                sc[j][i] = (tl < npt_now && tl >= first_now) ? sc[j][i] * scale : -INFINITY;
                mx = fmaxf(mx, sc[j][i]);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = MULTI ? fmaxf(m_run, mx) : mx;
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;   // nothing visible to this wave so far: all terms are 0
        const float alpha = MULTI ? __expf(m_run - m_use) : 0.f;  // 0 on the first visible round
        float sum = 0.f;
        half8_t pf[ATT_TILES / 2];   // element e of pair c: token 4 g + (e & 3) of tile 2 c + (e >> 2)
#pragma unroll
        for (int j = 0; j < ATT_TILES; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float pv = __expf(sc[j][i] - m_use);
                sum += pv;
                pf[j >> 1][(j & 1) * 4 + i] = (half_t)pv;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l_run = MULTI ? l_run * alpha + sum : sum;
        m_run = m_new;
        if constexpr (MULTI) {
#pragma unroll
            for (int dt = 0; dt < ATT_D / 16; dt++) o[dt] *= alpha;
        }
        __syncthreads();   // the V rows are in LDS

        // ---- O^T[d][head] += V^T P^T ----
        // transposed read: within a 16-lane group, lane 4 a + c supplies row a, columns 4 c .. 4 c + 3 of a block of 4
        // tokens x 16 columns and lane i receives column i of the 4 tokens
#pragma unroll
        for (int c = 0; c < ATT_TILES / 2; c++) {
            const int row0 = wave * ATT_WTOK + 2 * c * 16 + 4 * g + ta;   // its partner of the next tile is row0 + 16
            const int swz = (row0 >> 1) & 3;                              // (the same for row0 + 16)
#pragma unroll
            for (int dt = 0; dt < ATT_D / 16; dt++) {
                unsigned char* src = vs + row0 * 128 + ((dt ^ swz) << 5) + 8 * tc;
                const att_short4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4_t*)(src));
                const att_short4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((att_lds_s4_t*)(src + 16 * 128));
                const att_short8_t av = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, av), pf[c], o[dt], 0, 0, 0);
            }
        }
        if constexpr (MULTI) __syncthreads();   // (keeps the compiler from moving the next round's LDS writes above these reads)
    }
    __syncthreads();   // every wave is done with the V rows of the last round
    if (r < ATT_G) {
#pragma unroll
        for (int dt = 0; dt < ATT_D / 16; dt++) *reinterpret_cast<float4_t*>(po(wave, r) + 16 * dt + 4 * g) = o[dt];
        if (g == 0) { pm[wave][r] = m_run; pl[wave][r] = l_run; }
    }
    __syncthreads();

    // ---- merge the waves ----
    for (int i = tid; i < ATT_G * ATT_D; i += ATT_THREADS) {
        const int h = i / ATT_D, d = i % ATT_D;
        float m = -INFINITY;
#pragma unroll
        for (int wv = 0; wv < ATT_WAVES; wv++) m = fmaxf(m, pm[wv][h]);
        float tot = 0.f, l = 0.f;
        if (m > -INFINITY) {   // else nothing of the partition is visible
#pragma unroll
            for (int wv = 0; wv < ATT_WAVES; wv++) {
                const float f = __expf(pm[wv][h] - m);   // 0 for a wave with nothing visible
                tot += f * po(wv, h)[d];
                l += f * pl[wv][h];
            }
        }
        if (nparts == 1) {
            // a sink is one more logit in the softmax denominator (its value row is dropped)
            float num = tot, den = l;
            if (sinks) {
                const float sk = (float)sinks[hk * ATT_G + h], m2 = fmaxf(m, sk), f = __expf(m - m2);
                num *= f;
                den = den * f + __expf(sk - m2);
            }
            out[(ob + h) * ATT_D + d] = (half_t)(den > 0.f ? num / den : 0.f);
        } else {
            part_o[((ob + h) * nparts + part) * ATT_D + d] = tot;
            if (d == 0) {
                part_ml[((ob + h) * nparts + part) * 2 + 0] = m;
                part_ml[((ob + h) * nparts + part) * 2 + 1] = l;
            }
        }
    }
}

#define M355_ATT_PARAMS                                                                                              \
    const half_t *__restrict__ q, const half_t *__restrict__ kc, const half_t *__restrict__ vc,                          \
        const int32_t *__restrict__ block_tables, const int32_t *__restrict__ ctx_lens, int Hq, int Hkv, int block_size, \
        int max_blocks, float scale, int nparts, int rounds, float *__restrict__ part_o, float *__restrict__ part_ml,   \
        half_t *__restrict__ out, const half_t *__restrict__ sinks, int window
#define M355_ATT_ARGS q, kc, vc, block_tables, ctx_lens, Hq, Hkv, block_size, max_blocks, scale, nparts, rounds, part_o, part_ml, out, sinks, window
// one round per workgroup (a context of up to ATT_PART tokens, or partitions that a second kernel merges)
__global__ __launch_bounds__(ATT_THREADS) void paged_attn_decode_kernel(M355_ATT_PARAMS) {
    paged_attn_decode_body<false>(M355_ATT_ARGS);
}
// several rounds per workgroup; registers held to what four workgroups per CU allow (as LDS does)
__global__ __launch_bounds__(ATT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void paged_attn_decode_rounds_kernel(M355_ATT_PARAMS) {
    paged_attn_decode_body<true>(M355_ATT_ARGS);
}
#undef M355_ATT_ARGS
#undef M355_ATT_PARAMS

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
    // Rounds of ATT_PART tokens per workgroup.  Several rounds when that lets one workgroup take the whole context, so
    // that no merge launch is needed: always up to two rounds (a second round costs less than the launch), up to four
    // when the batch alone fills the chip.  Longer contexts keep one round per workgroup and merge (measured: four
    // rounds at 32 x 2048 tokens are 4 % slower than 2048 single-round workgroups).
    const int base_parts = (max_ctx + ATT_PART - 1) / ATT_PART;
    int rounds = 1;
    if (base_parts <= 2 || (base_parts <= 4 && (int64_t)B * Hkv >= 256)) rounds = base_parts;
    const int nparts = (base_parts + rounds - 1) / rounds;
    const size_t need = m355_paged_attn_workspace_bytes(B, Hq, D, max_ctx);   // (sized for one round per partition)
    float *po = nullptr, *pml = nullptr;
    if (nparts > 1) {
        if (!workspace || ws_bytes < need) return fail(-6, "m355_paged_attn_decode: workspace of %zu bytes needed, %zu given", need, ws_bytes);
        po = reinterpret_cast<float*>(workspace);
        pml = po + (size_t)B * Hq * nparts * D;
    }
    hipStream_t st = (hipStream_t)stream;
    if (rounds > 1)
        paged_attn_decode_rounds_kernel<<<dim3(B, Hkv, nparts), dim3(ATT_THREADS), 0, st>>>(
            (const half_t*)q_f16, (const half_t*)kcache_f16, (const half_t*)vcache_f16, block_tables, ctx_lens, Hq, Hkv,
            block_size, max_blocks, scale, nparts, rounds, po, pml, (half_t*)out_f16, (const half_t*)sinks_f16, window);
    else
        paged_attn_decode_kernel<<<dim3(B, Hkv, nparts), dim3(ATT_THREADS), 0, st>>>(
            (const half_t*)q_f16, (const half_t*)kcache_f16, (const half_t*)vcache_f16, block_tables, ctx_lens, Hq, Hkv,
            block_size, max_blocks, scale, nparts, 1, po, pml, (half_t*)out_f16, (const half_t*)sinks_f16, window);
    M355_LAUNCH_CHECK("paged_attn_decode_kernel");
    if (nparts > 1) {
        paged_attn_merge_kernel<<<dim3(B * Hq), dim3(64), 0, st>>>(po, pml, nparts, (half_t*)out_f16, (const half_t*)sinks_f16, Hq);
        M355_LAUNCH_CHECK("paged_attn_merge_kernel");
    }
    return 0;
}
