/*
 * CPU restatement of the reference's NF4 MoE decode path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * are the only callers.  Nothing in the shipped GPU path links, loads or calls it.
 *
 * Pinning: every function below is checked against outputs of the reference's own Python
 * (patches/streaming_unsloth_bnb.py, patches/unsloth_to_vllm_converter.py) stored under
 * tests/golden/ (made by tests/golden/make_golden.py, which imports the reference).
 *
 * What follows which reference lines:
 *   m355_ref_resolve_absmax   streaming_unsloth_bnb.py:148-173   (double-quant absmax rules)
 *   m355_ref_nf4_dequant      streaming_unsloth_bnb.py:101-106,175-222 (unpack, codebook, scale, fp16)
 *   m355_ref_router_topk      unsloth_to_vllm_converter.py:291-292 + monkey_patch_router_fp16.py:11-14
 *   m355_ref_moe_nf4_forward  unsloth_to_vllm_converter.py:298-323 + streaming_unsloth_bnb.py:269-270
 *   m355_ref_paged_attn_decode  plain fp32 softmax attention (the reference only constrains the
 *                             upstream FA2 call: patch_3_disable_sinks.py:36-38) -- parity unpinned,
 *                             no reference output exists for it.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define M355_REF_API __attribute__((visibility("default")))

static __thread char g_err[256];

M355_REF_API const char* m355_ref_last_error(void) { return g_err; }
M355_REF_API int m355_ref_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- fp16 <-> fp32, round-to-nearest-even, subnormals kept (what torch .to(float16) does) ---- */
static inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x0200u | ((ax >> 13) & 0x3ffu)) : 0u));
    }
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to >= 65520 -> inf */
    if (ax < 0x38800000u) {                                   /* subnormal or zero in fp16 */
        if (ax < 0x33000000u) return (uint16_t)sign;          /* < 2^-25 -> 0 (2^-25 ties to even 0) */
        uint32_t e = ax >> 23;                                /* biased exponent, 102..112 */
        uint32_t m = (ax & 0x7fffffu) | 0x800000u;
        uint32_t shift = 126u - e;                            /* 14..24 */
        uint32_t r = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1u);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ax - 0x38000000u; /* rebias 127 -> 15 */
    uint32_t rem = r & 0x1fffu;
    r >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}

static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            int s = 0;
            while (!(m & 0x400u)) { m <<= 1; s++; }
            m &= 0x3ffu;
            x = sign | ((uint32_t)(113 - s) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7f800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &x, 4);
    return f;
}

M355_REF_API uint16_t m355_ref_f32_to_f16(float f) { return f32_to_f16(f); }
M355_REF_API float m355_ref_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

/* streaming_unsloth_bnb.py:101-106 */
static const float NF4_QUANT_MAP[16] = {
    -1.0f, -0.6961928009986877f, -0.5250730514526367f, -0.39491748809814453f,
    -0.28444138169288635f, -0.18477343022823334f, -0.09105003625154495f, 0.0f,
    0.07958029955625534f, 0.16093020141124725f, 0.24611230194568634f, 0.33791524171829224f,
    0.44070982933044434f, 0.5626170039176941f, 0.7229568362236023f, 1.0f};

M355_REF_API const float* m355_ref_nf4_table(void) { return NF4_QUANT_MAP; }

/*
 * Resolve the per-block fp32 absmax the way streaming_unsloth_bnb.py:148-173 does.
 *   nested present (nested_map != NULL): a = nested_map[absmax_u8[i]] then
 *       n_nested == 1          -> a *= nested_absmax[0]                 (:158-160)
 *       n_nested == n_absmax   -> a[i] *= nested_absmax[i]              (:161-163)
 *       otherwise              -> a *= mean(nested_absmax)              (:165-171; the broadcast of two
 *                                 1-D tensors of different length raises, so the fallback always runs)
 *   nested absent: a = float(absmax) -- either u8 codes taken as numbers or fp32 (:173)
 * The mean is an fp32 value.  torch sums fp32 in a cascade whose order this code does not copy; it
 * sums in double and rounds once.  The two agree bit for bit whenever the fp32 sum is exact
 * (the golden inputs are built that way); `mean_override` lets a caller pass torch's own value.
 * Only 1-D absmax / nested_absmax are handled (the shapes bitsandbytes writes).
 */
M355_REF_API int m355_ref_resolve_absmax(const uint8_t* absmax_u8, const float* absmax_f32,
                                         int64_t n_absmax, const float* nested_map,
                                         const float* nested_absmax, int64_t n_nested,
                                         const float* mean_override, float* out) {
    if (n_absmax < 1 || !out) { snprintf(g_err, sizeof g_err, "resolve_absmax: bad size"); return -1; }
    if (nested_map) {
        if (!absmax_u8 || !nested_absmax || n_nested < 1) {
            snprintf(g_err, sizeof g_err, "resolve_absmax: nested needs u8 absmax + nested_absmax");
            return -1;
        }
        if (n_nested == 1) {
            const float s = nested_absmax[0];
            for (int64_t i = 0; i < n_absmax; i++) out[i] = nested_map[absmax_u8[i]] * s;
        } else if (n_nested == n_absmax) {
            for (int64_t i = 0; i < n_absmax; i++) out[i] = nested_map[absmax_u8[i]] * nested_absmax[i];
        } else {
            float mean;
            if (mean_override) {
                mean = *mean_override;
            } else {
                double acc = 0.0;
                for (int64_t i = 0; i < n_nested; i++) acc += (double)nested_absmax[i];
                const float sum32 = (float)acc;
                mean = sum32 / (float)n_nested;
            }
            for (int64_t i = 0; i < n_absmax; i++) out[i] = nested_map[absmax_u8[i]] * mean;
        }
    } else if (absmax_f32) {
        for (int64_t i = 0; i < n_absmax; i++) out[i] = absmax_f32[i];
    } else if (absmax_u8) {
        for (int64_t i = 0; i < n_absmax; i++) out[i] = (float)absmax_u8[i];
    } else {
        if (n_absmax != 1) { snprintf(g_err, sizeof g_err, "resolve_absmax: no absmax given"); return -1; }
        out[0] = 1.0f; /* torch.ones(1), :173 */
    }
    return 0;
}

/*
 * NF4 packed bytes -> fp16, streaming_unsloth_bnb.py:175-222.
 *   element 2i = low nibble of byte i, element 2i+1 = high nibble (:180-187)
 *   value = NF4_QUANT_MAP[nibble] in fp32 (:193)
 *   n_absmax > 1, n_params % n_absmax == 0: blocksize = n_params / n_absmax, value *= absmax[block] (:199-202)
 *   n_absmax > 1 otherwise: blocksize 64 if ceil(n/64) <= n_absmax, else 128 if ceil(n/128) <= n_absmax,
 *       else 64; only min(ceil(n/bs), n_absmax) blocks are scaled, the tail keeps the bare code (:204-215)
 *   n_absmax == 1: value *= absmax[0] (:216-218)
 *   cast fp32 -> fp16 round-to-nearest-even (:222)
 */
M355_REF_API int m355_ref_nf4_dequant(const uint8_t* q, int64_t n_params, const float* absmax,
                                      int64_t n_absmax, uint16_t* out) {
    if (!q || !absmax || !out || n_params < 2 || (n_params & 1) || n_absmax < 1) {
        snprintf(g_err, sizeof g_err, "nf4_dequant: bad argument");
        return -1;
    }
    int64_t blocksize, nscaled;
    if (n_absmax == 1) {
        blocksize = n_params;
        nscaled = 1;
    } else if (n_params % n_absmax == 0) {
        blocksize = n_params / n_absmax;
        nscaled = n_absmax;
    } else {
        blocksize = 64;
        if ((n_params + 63) / 64 <= n_absmax) blocksize = 64;
        else if ((n_params + 127) / 128 <= n_absmax) blocksize = 128;
        nscaled = (n_params + blocksize - 1) / blocksize;
        if (nscaled > n_absmax) nscaled = n_absmax;
    }
    const int64_t scaled_end = (nscaled * blocksize < n_params) ? nscaled * blocksize : n_params;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_params / 2; i++) {
        const uint8_t b = q[i];
        for (int h = 0; h < 2; h++) {
            const int64_t idx = 2 * i + h;
            float v = NF4_QUANT_MAP[h ? (b >> 4) : (b & 0x0f)];
            if (idx < scaled_end) v = v * absmax[idx / blocksize];
            out[idx] = f32_to_f16(v);
        }
    }
    return 0;
}

/*
 * Router: logits = fp16(x . Wr^T + br) (an fp16 nn.Linear, monkey_patch_router_fp16.py:11-14), then
 * torch.topk(logits, k) and softmax over the k selected values (unsloth_to_vllm_converter.py:291-292).
 * Ties between equal fp16 logits go to the lower expert index (torch leaves the order unspecified).
 * `w` receives the fp32 softmax; the fp16 rounding the reference applies (its logits tensor is fp16, so
 * softmax returns fp16) is done where the weight is used.  `logits_out` (optional) gets the fp16 logits.
 */
M355_REF_API int m355_ref_router_topk(const uint16_t* x, const uint16_t* wr, const uint16_t* br, int T,
                                      int H, int E, int k, int32_t* ids, float* w, uint16_t* logits_out) {
    if (!x || !wr || !ids || !w || T < 0 || H < 1 || E < 1 || k < 1 || k > E || E > 4096) {
        snprintf(g_err, sizeof g_err, "router_topk: bad argument");
        return -1;
    }
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T; t++) {
        float lg[4096];
        uint8_t taken[4096];
        for (int e = 0; e < E; e++) {
            double acc = 0.0;
            for (int h = 0; h < H; h++)
                acc += (double)f16_to_f32(x[(int64_t)t * H + h]) * (double)f16_to_f32(wr[(int64_t)e * H + h]);
            if (br) acc += (double)f16_to_f32(br[e]);
            const uint16_t l16 = f32_to_f16((float)acc);
            if (logits_out) logits_out[(int64_t)t * E + e] = l16;
            lg[e] = f16_to_f32(l16);
            taken[e] = 0;
        }
        float sel[64];
        float mx = -INFINITY;
        for (int j = 0; j < k; j++) {
            int best = -1;
            for (int e = 0; e < E; e++)
                if (!taken[e] && (best < 0 || lg[e] > lg[best])) best = e;
            taken[best] = 1;
            ids[(int64_t)t * k + j] = best;
            if (j < 64) sel[j] = lg[best];
            if (lg[best] > mx) mx = lg[best];
        }
        float sum = 0.f;
        for (int j = 0; j < k; j++) { sel[j] = expf(sel[j] - mx); sum += sel[j]; }
        for (int j = 0; j < k; j++) w[(int64_t)t * k + j] = sel[j] / sum;
    }
    return 0;
}

/* y[r] = fp16( sum_c W[r][c] * v[c] ), fp32 accumulation in 8 interleaved partial sums. */
static void gemv_f16(const uint16_t* W, int rows, int cols, const float* v, uint16_t* y) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; r++) {
        const uint16_t* wr = W + (int64_t)r * cols;
        float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int c = 0;
        for (; c + 8 <= cols; c += 8)
            for (int u = 0; u < 8; u++) p[u] += f16_to_f32(wr[c + u]) * v[c + u];
        float acc = ((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7]));
        for (; c < cols; c++) acc += f16_to_f32(wr[c]) * v[c];
        y[r] = f32_to_f16(acc);
    }
}

#define M355_FLAG_FP16_COMBINE 1 /* round routing weights to fp16, accumulate over k in fp16 */

/*
 * One MoE layer, the reference's way (unsloth_to_vllm_converter.py:298-323):
 *   for each token, for kk = 0..k-1 in order: gu = W13_e x (fp16 out); gate = gu[:FF], up = gu[FF:]
 *   (concatenated halves, streaming_unsloth_bnb.py:269-270); h = silu(gate) * up (each op rounds to
 *   fp16); y = W2_e h (fp16 out); out += w * y.
 * With M355_FLAG_FP16_COMBINE the product and the running sum are rounded to fp16 at every step, as
 * the reference's fp16 tensors do; without it the sum over k is kept in fp32 and rounded once.
 * Weights come either dequantized (w13_f16 / w2_f16, [E_local][2FF][H] and [E_local][H][FF]) or
 * packed in checkpoint order with resolved fp32 absmax (one per `blocksize` elements); packed
 * experts are dequantized on the fly with m355_ref_nf4_dequant ("streaming" mode).
 * Pairs whose expert is outside [e_first, e_first + E_local) are skipped (another rank owns them).
 */
M355_REF_API int m355_ref_moe_nf4_forward(const uint16_t* x, int T, const int32_t* ids, const float* w,
                                          int k, const uint8_t* w13_q, const float* w13_absmax,
                                          const uint8_t* w2_q, const float* w2_absmax,
                                          const uint16_t* w13_f16, const uint16_t* w2_f16, int E_local,
                                          int e_first, int H, int FF, int blocksize, int flags,
                                          uint16_t* out) {
    if (!x || !ids || !w || !out || T < 0 || k < 1 || H < 1 || FF < 1 || E_local < 1) {
        snprintf(g_err, sizeof g_err, "moe_forward: bad argument");
        return -1;
    }
    const int have_f16 = (w13_f16 && w2_f16);
    if (!have_f16 && !(w13_q && w13_absmax && w2_q && w2_absmax && blocksize > 0)) {
        snprintf(g_err, sizeof g_err, "moe_forward: need fp16 or packed weights");
        return -1;
    }
    const int64_t n13 = (int64_t)2 * FF * H, n2 = (int64_t)H * FF;
    uint16_t* W13 = NULL;
    uint16_t* W2 = NULL;
    if (!have_f16) {
        if (n13 % blocksize || n2 % blocksize) { snprintf(g_err, sizeof g_err, "moe_forward: blocksize"); return -1; }
        W13 = (uint16_t*)malloc(n13 * 2);
        W2 = (uint16_t*)malloc(n2 * 2);
    }
    float* xv = (float*)malloc(sizeof(float) * (H > FF ? H : FF));
    uint16_t* gu = (uint16_t*)malloc(2 * (size_t)2 * FF);
    uint16_t* y = (uint16_t*)malloc(2 * (size_t)H);
    float* acc32 = (float*)malloc(sizeof(float) * H);
    int cached = -1; /* expert whose dequantized weights sit in W13/W2 */
    for (int t = 0; t < T; t++) {
        uint16_t* o = out + (int64_t)t * H;
        for (int c = 0; c < H; c++) { o[c] = 0; acc32[c] = 0.f; }
        for (int kk = 0; kk < k; kk++) {
            const int e = ids[(int64_t)t * k + kk] - e_first;
            if (e < 0 || e >= E_local) continue;
            const uint16_t *A13, *A2;
            if (have_f16) {
                A13 = w13_f16 + (int64_t)e * n13;
                A2 = w2_f16 + (int64_t)e * n2;
            } else {
                if (cached != e) {
                    m355_ref_nf4_dequant(w13_q + (int64_t)e * (n13 / 2), n13, w13_absmax + (int64_t)e * (n13 / blocksize), n13 / blocksize, W13);
                    m355_ref_nf4_dequant(w2_q + (int64_t)e * (n2 / 2), n2, w2_absmax + (int64_t)e * (n2 / blocksize), n2 / blocksize, W2);
                    cached = e;
                }
                A13 = W13;
                A2 = W2;
            }
            for (int h = 0; h < H; h++) xv[h] = f16_to_f32(x[(int64_t)t * H + h]);
            gemv_f16(A13, 2 * FF, H, xv, gu);
            for (int f = 0; f < FF; f++) {
                const float g = f16_to_f32(gu[f]);
                const uint16_t s16 = f32_to_f16(g / (1.0f + expf(-g)));
                xv[f] = f16_to_f32(f32_to_f16(f16_to_f32(s16) * f16_to_f32(gu[FF + f])));
            }
            gemv_f16(A2, H, FF, xv, y);
            if (flags & M355_FLAG_FP16_COMBINE) {
                const float w16 = f16_to_f32(f32_to_f16(w[(int64_t)t * k + kk]));
                for (int c = 0; c < H; c++) {
                    const uint16_t prod = f32_to_f16(w16 * f16_to_f32(y[c]));
                    o[c] = f32_to_f16(f16_to_f32(o[c]) + f16_to_f32(prod));
                }
            } else {
                const float wf = w[(int64_t)t * k + kk];
                for (int c = 0; c < H; c++) acc32[c] += wf * f16_to_f32(y[c]);
            }
        }
        if (!(flags & M355_FLAG_FP16_COMBINE))
            for (int c = 0; c < H; c++) o[c] = f32_to_f16(acc32[c]);
    }
    free(W13); free(W2); free(xv); free(gu); free(y); free(acc32);
    return 0;
}

/*
 * Decode attention over a paged KV cache, one query token per sequence, causal by construction
 * (the query is the newest token), grouped-query heads, no sinks, no sliding window
 * (patch_3_disable_sinks.py:36-38).  fp32 softmax, double accumulation.
 *   q [B][Hq][D], kcache/vcache [num_blocks][block_size][Hkv][D], block_tables [B][max_blocks],
 *   ctx_lens [B] (tokens in cache incl. the current one), out [B][Hq][D].
 */
M355_REF_API int m355_ref_paged_attn_decode(const uint16_t* q, const uint16_t* kcache, const uint16_t* vcache,
                                            const int32_t* block_tables, const int32_t* ctx_lens, int B, int Hq,
                                            int Hkv, int D, int block_size, int max_blocks, float scale,
                                            uint16_t* out) {
    if (!q || !kcache || !vcache || !block_tables || !ctx_lens || !out || Hkv < 1 || Hq % Hkv || D < 1 || block_size < 1) {
        snprintf(g_err, sizeof g_err, "paged_attn: bad argument");
        return -1;
    }
    const int G = Hq / Hkv;
#pragma omp parallel for schedule(static)
    for (int bh = 0; bh < B * Hq; bh++) {
        const int b = bh / Hq, h = bh % Hq, hk = h / G;
        const int L = ctx_lens[b];
        uint16_t* o = out + ((int64_t)b * Hq + h) * D;
        if (L <= 0) { for (int d = 0; d < D; d++) o[d] = 0; continue; }
        float* s = (float*)malloc(sizeof(float) * L);
        float mx = -INFINITY;
        for (int t = 0; t < L; t++) {
            const int blk = block_tables[(int64_t)b * max_blocks + t / block_size];
            const uint16_t* kr = kcache + (((int64_t)blk * block_size + t % block_size) * Hkv + hk) * D;
            double acc = 0.0;
            for (int d = 0; d < D; d++) acc += (double)f16_to_f32(q[((int64_t)b * Hq + h) * D + d]) * (double)f16_to_f32(kr[d]);
            s[t] = (float)acc * scale;
            if (s[t] > mx) mx = s[t];
        }
        double den = 0.0;
        for (int t = 0; t < L; t++) { s[t] = expf(s[t] - mx); den += s[t]; }
        for (int d = 0; d < D; d++) {
            double acc = 0.0;
            for (int t = 0; t < L; t++) {
                const int blk = block_tables[(int64_t)b * max_blocks + t / block_size];
                const uint16_t* vr = vcache + (((int64_t)blk * block_size + t % block_size) * Hkv + hk) * D;
                acc += (double)s[t] * (double)f16_to_f32(vr[d]);
            }
            o[d] = f32_to_f16((float)(acc / den));
        }
        free(s);
    }
    return 0;
}

/* Self-contained generator for synthetic inputs (splitmix64); tests and fixtures share it so the GPU
 * box rebuilds the same bytes without torch's generator.  Fills n bytes from the stream `seed`. */
M355_REF_API void m355_ref_fill_bytes(uint64_t seed, uint8_t* dst, int64_t n) {
    uint64_t s = seed;
    int64_t i = 0;
    while (i < n) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        for (int b = 0; b < 8 && i < n; b++, i++) dst[i] = (uint8_t)(z >> (8 * b));
    }
}
