// Causal self-attention over whole prompts (prefill) for grouped-query heads: 8 query heads per KV head, head size 64.
// Flash-style: the score matrix is never stored; per 32-query block the keys are walked in tiles of 64 with a running
// maximum and denominator per query.
//
// A workgroup = one (sequence, KV head, block of 32 queries); its 8 waves are the 8 query heads of the group, so every
// K and V tile staged in LDS (LDS-DMA, two buffers) serves 8 heads.  Per wave and 32-key half tile:
//   S^T[key][q] = K Q^T      4 x v_mfma_f32_32x32x16_f16 (A = K rows from LDS, B = the wave's Q fragments in registers).
//                            The result has the query on the lane and 16 of the 32 keys in the lane's registers (the
//                            other 16 in lane ^ 32), so the row maximum and sum of the softmax are register loops plus
//                            one cross-half exchange.
//   O^T[d][q] += V^T P^T     the probabilities, rounded to fp16 where they sit, ARE the B operand of the second product
//                            (it sums over the accumulator's row index: no data movement); the A operand V^T comes from
//                            the row-major V tile through ds_read_b64_tr_b16 (hardware transpose read).
// Optional attention sinks (one more logit in the denominator) and sliding window, as the decode kernel has them.
#include "common.h"

namespace m355 {

typedef float float16_t __attribute__((ext_vector_type(16)));
typedef short short4_t __attribute__((ext_vector_type(4)));

constexpr int AP_THREADS = 512;
constexpr int AP_QBLK = 32;
constexpr int AP_KVBLK = 64;
constexpr int AP_D = 64;
constexpr int AP_G = 8;
constexpr int AP_TILE_BYTES = AP_KVBLK * AP_D * 2;   // 8 KiB: one K (or V) tile
constexpr int AP_LDS = 4 * AP_TILE_BYTES;            // K and V, two buffers each
constexpr float AP_LAZY = 6.0f;                      // log2 units the running maximum may exceed the exponents' reference point

typedef const half8_t __attribute__((address_space(3))) ap_lds_h8_t;
typedef __attribute__((address_space(3))) short4_t ap_lds_s4_t;
typedef __attribute__((address_space(3))) void ap_lds_void_t;
typedef const __attribute__((address_space(1))) void ap_glb_void_t;

// Combine a value with the one lane ^ 32 holds: one v_permlane32_swap (the upper half of the first operand trades places
// with the lower half of the second) leaves both halves' values on every lane, no LDS round trip.
__device__ __forceinline__ float cross_half_max(float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    const auto sw = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
}
__device__ __forceinline__ float cross_half_sum(float v) {
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    const auto sw = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __builtin_bit_cast(float, (uint32_t)sw[0]) + __builtin_bit_cast(float, (uint32_t)sw[1]);
}

__global__ __launch_bounds__(AP_THREADS) void attn_prefill_kernel(
    const half_t* __restrict__ q, const half_t* __restrict__ k, const half_t* __restrict__ v,
    const int32_t* __restrict__ seq_starts, int Hq, int64_t q_stride, int64_t kv_stride, int64_t o_stride, float scale,
    const half_t* __restrict__ sinks, int window, half_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // late query blocks see the most keys: they go first, so the launch does not end on its longest workgroups
    const int b = blockIdx.z, hk = blockIdx.y, q0 = (gridDim.x - 1 - blockIdx.x) * AP_QBLK;
    const int s0 = seq_starts[b];
    const int L = seq_starts[b + 1] - s0;
    if (q0 >= L) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hq = hk * AP_G + wave;
    const int qi = lane & 31, hh = lane >> 5;
    const int qabs = q0 + qi;   // this lane's query (rows past the end compute on the last row and are not stored)

    // Q fragments: B operand of step s holds Q[q][16 s + 8 h ..], scaled once by scale * log2(e) (rounded to fp16 again), so
    // that the first product delivers the scores in the units of the exponent (exp2 below) and no per-score multiply is left
    const float scale2 = scale * 1.44269504088896341f;
    half8_t qf[4];
    {
        const half_t* qrow = q + (int64_t)(s0 + min(qabs, L - 1)) * q_stride + (int64_t)hq * AP_D + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const half8_t v = *reinterpret_cast<const half8_t*>(qrow + 16 * s);
#pragma unroll
            for (int j = 0; j < 8; j++) qf[s][j] = (half_t)((float)v[j] * scale2);
        }
    }
    // keys this block of queries can see: up to its last query; from the window's start of its first query
    const int k_hi = min(q0 + AP_QBLK, L);                              // exclusive
    const int k_lo = window > 0 ? max(q0 - window + 1, 0) : 0;
    const int t_lo = k_lo / AP_KVBLK, t_hi = (k_hi + AP_KVBLK - 1) / AP_KVBLK;

    // staging: one DMA instruction per wave and tile for K and one for V, 8 keys x 128 bytes each.  K chunks are
    // XOR-swizzled on the source side (conflict-free ds_read_b128 of the A fragments); V rows 2, 3 (mod 4) have their
    // 64-byte halves swapped, so the 4 rows x 64 bytes that a 32-lane half reads transposed cover every bank once.
    const int skey = 8 * wave + (lane >> 3);   // key of the tile this lane stages
    const int kchunk = (lane & 7) ^ ((skey >> 1) & 7);
    auto stage = [&](int t, int buf) {
        const int key = min(t * AP_KVBLK + skey, L - 1);
        const half_t* krow = k + (int64_t)(s0 + key) * kv_stride + (int64_t)hk * AP_D;
        const half_t* vrow = v + (int64_t)(s0 + key) * kv_stride + (int64_t)hk * AP_D;
        __builtin_amdgcn_global_load_lds((ap_glb_void_t*)(krow + 8 * kchunk),
                                         (ap_lds_void_t*)(smem + buf * 2 * AP_TILE_BYTES + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((ap_glb_void_t*)(vrow + 8 * ((lane & 7) ^ (((skey >> 1) & 1) << 2))),
                                         (ap_lds_void_t*)(smem + buf * 2 * AP_TILE_BYTES + AP_TILE_BYTES + wave * 1024), 16, 0, 0);
    };
    // A fragment of S^T step s: K[key qi of the half tile][16 s + 8 h ..] = chunk 2 s + h of the key's line
    uint32_t koff[4];
#pragma unroll
    for (int s = 0; s < 4; s++) koff[s] = qi * 128 + (((2 * s + hh) ^ ((qi >> 1) & 7)) << 4);
    // transposed V reads: within a 16-lane group, lane 4 a + c supplies row a, columns 4 c .. 4 c + 3 of a block of
    // 4 keys x 16 columns and lane i receives column i of the 4 keys.  Group (h, dg): keys 4 h + .., columns 16 dg + ..
    const int dg = (lane >> 4) & 1, ga = (lane & 15) >> 2, gc = lane & 3;
    const uint32_t voff = AP_TILE_BYTES + (4 * hh + ga) * 128 + (16 * dg + 4 * gc) * 2;
    const uint32_t vswz = (uint32_t)(((ga >> 1) & 1) * 64);   // which 64-byte half holds columns 0..31 of this lane's row

    float m_run = -INFINITY, l_run = 0.f;   // reference point of the exponents (log2 units) and denominator of this lane's query
    // minus the reference point in every element: the accumulator the first product starts from, so it delivers
    // score - reference and the exponent needs no subtraction either (rewritten only when the reference moves)
    float16_t nref;
#pragma unroll
    for (int i = 0; i < 16; i++) nref[i] = 0.f;
    float16_t o[2];
#pragma unroll
    for (int i = 0; i < 16; i++) { o[0][i] = 0.f; o[1][i] = 0.f; }

    if (t_lo < t_hi) stage(t_lo, t_lo & 1);
    for (int t = t_lo; t < t_hi; t++) {
        // this tile has landed (every wave's DMA), and nobody reads the other buffer any more.  (Requesting tiles two or
        // three ahead into three or four buffers was measured: 2-3 % slower, the DMA latency is not what the loop waits for.)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (t + 1 < t_hi) stage(t + 1, (t + 1) & 1);
        const uint32_t base = (uint32_t)((t & 1) * 2 * AP_TILE_BYTES);
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            const int kbase = t * AP_KVBLK + sub * 32;
            if (kbase >= k_hi) break;   // (uniform) nothing of this half tile is visible
            float16_t x = nref;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const half8_t a = *reinterpret_cast<ap_lds_h8_t*>(base + sub * 4096 + koff[s]);
                x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], x, 0, 0, 0);
            }
            // register i of lane (q, h) is key kbase + (i & 3) + 8 (i >> 2) + 4 h; x = score (log2 units) - reference.
            // only half tiles on the diagonal or on the window's edge need the per-element test
            const bool edge = kbase + 31 > q0 || (window > 0 && kbase <= q0 + AP_QBLK - 1 - window);
            if (edge) {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int kabs = kbase + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    const bool ok = kabs <= qabs && (window <= 0 || kabs > qabs - window);
                    x[i] = ok ? x[i] : -INFINITY;
                }
            }
            float mx = fmaxf(fmaxf(x[0], x[1]), x[2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, x[i]), x[i + 1]);
            mx = fmaxf(mx, x[15]);
            mx = cross_half_max(mx);   // how far the largest score of this half tile lies above the reference
            // The reference point follows the running maximum lazily: it moves only when a score exceeds it by more than
            // AP_LAZY (log2 units), so the probabilities are at most 2^AP_LAZY instead of 1 (fp16 holds them; the
            // denominator carries the same factor) and the accumulator is rescaled a few times per query instead of
            // whenever any of the 32 queries sees a new maximum.  Until a query has seen a key its reference is 0 and the
            // first visible key moves it.
            const bool first = m_run == -INFINITY;
            const bool move = first ? mx > -INFINITY : mx > AP_LAZY;
            const float delta = move ? mx : 0.f;
            const float alpha = move ? (first ? 0.f : __builtin_amdgcn_exp2f(-delta)) : 1.0f;
            if (__any(move)) {
#pragma unroll
                for (int i = 0; i < 16; i++) x[i] -= delta;
                m_run = move ? (first ? mx : m_run + mx) : m_run;
                const float nr = (m_run == -INFINITY) ? 0.f : -m_run;
#pragma unroll
                for (int i = 0; i < 16; i++) nref[i] = nr;
#pragma unroll
                for (int i = 0; i < 16; i++) { o[0][i] *= alpha; o[1][i] *= alpha; }
            }
            float rs = 0.f;
            half8_t pf[2];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const float p0 = __builtin_amdgcn_exp2f(x[i]);
                const float p1 = __builtin_amdgcn_exp2f(x[i + 1]);
                const half2_t ph = {(half_t)p0, (half_t)p1};
                pf[i >> 3][i & 7] = ph[0];
                pf[i >> 3][(i & 7) + 1] = ph[1];
                // the denominator sums the probabilities as the second product sees them (rounded to fp16), two per instruction
                rs = __builtin_amdgcn_fdot2(ph, half2_t{(half_t)1.0f, (half_t)1.0f}, rs, false);
            }
            rs = cross_half_sum(rs);
            l_run = l_run * alpha + rs;
            // O^T[d][q] += V^T[d][key] P^T[key][q]: element j of step s2 is key 16 s2 + 8 (j >> 2) + 4 h + (j & 3)
#pragma unroll
            for (int dt = 0; dt < 2; dt++)
#pragma unroll
                for (int s2 = 0; s2 < 2; s2++) {
                    const uint32_t va = base + voff + (uint32_t)((sub * 32 + 16 * s2) * 128) + ((uint32_t)(dt * 64) ^ vswz);
                    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<ap_lds_s4_t*>(va));
                    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<ap_lds_s4_t*>(va + 8 * 128));
                    typedef short short8_t __attribute__((ext_vector_type(8)));
                    const short8_t av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, av), pf[s2], o[dt], 0, 0, 0);
                }
        }
    }
    if (sinks) {   // a sink is one more logit in the softmax denominator (its value row is dropped)
        const float sk = (float)sinks[hq] * 1.44269504088896341f;
        const float m2 = fmaxf(m_run, sk);
        const float f = __builtin_amdgcn_exp2f(m_run - m2);
        l_run = l_run * f + __builtin_amdgcn_exp2f(sk - m2);
#pragma unroll
        for (int i = 0; i < 16; i++) { o[0][i] *= f; o[1][i] *= f; }
    }
    if (qabs < L) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        half_t* orow = out + (int64_t)(s0 + qabs) * o_stride + (int64_t)hq * AP_D;
        // register i of tile dt is column d = 32 dt + (i & 3) + 8 (i >> 2) + 4 h: four consecutive d per group of 4
#pragma unroll
        for (int dt = 0; dt < 2; dt++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                half4_t hv;
#pragma unroll
                for (int i = 0; i < 4; i++) hv[i] = (half_t)(o[dt][4 * g4 + i] * inv);
                *reinterpret_cast<half4_t*>(orow + 32 * dt + 8 * g4 + 4 * hh) = hv;
            }
    }
}

}  // namespace m355

using namespace m355;

extern "C" int m355_attn_prefill(const void* q_f16, const void* k_f16, const void* v_f16, const int32_t* seq_starts, int B,
                                 int max_seq, int Hq, int Hkv, int D, int64_t q_stride, int64_t kv_stride, int64_t o_stride,
                                 float scale, const void* sinks_f16, int window, void* out_f16, m355_stream_t stream) {
    if (B == 0 || max_seq == 0) return 0;
    if (!q_f16 || !k_f16 || !v_f16 || !seq_starts || !out_f16) return fail(-1, "m355_attn_prefill: null pointer");
    if (B < 0 || max_seq < 0 || D != AP_D || Hkv <= 0 || Hq != AP_G * Hkv || window < 0)
        return fail(-2, "m355_attn_prefill: unsupported shape B=%d max_seq=%d Hq=%d Hkv=%d D=%d (need D == 64, Hq == 8 Hkv)",
                    B, max_seq, Hq, Hkv, D);
    if (q_stride < (int64_t)Hq * D || kv_stride < (int64_t)Hkv * D || o_stride < (int64_t)Hq * D || (q_stride % 8) ||
        (kv_stride % 8) || (o_stride % 4))
        return fail(-2, "m355_attn_prefill: token strides must cover the heads and be multiples of 8 elements (output: 4)");
    if (!aligned(q_f16, 16) || !aligned(k_f16, 16) || !aligned(v_f16, 16) || !aligned(out_f16, 8))
        return fail(-3, "m355_attn_prefill: q, k, v must be 16-byte aligned, out 8-byte aligned");
    const dim3 grid((max_seq + AP_QBLK - 1) / AP_QBLK, Hkv, B);
    attn_prefill_kernel<<<grid, dim3(AP_THREADS), AP_LDS, (hipStream_t)stream>>>(
        (const half_t*)q_f16, (const half_t*)k_f16, (const half_t*)v_f16, seq_starts, Hq, q_stride, kv_stride, o_stride, scale,
        (const half_t*)sinks_f16, window, (half_t*)out_f16);
    M355_LAUNCH_CHECK("m355_attn_prefill");
    return 0;
}
