/*
 * libmoe355 -- C ABI of the MI355X (gfx950) NF4 mixture-of-experts decode path.
 *
 * Every entry point enqueues HIP kernels on the caller's stream and returns without synchronising.
 * All buffers are caller-owned DEVICE pointers unless a parameter says "host"; the library keeps no
 * per-call state, allocates nothing and frees nothing.  Return value: 0 = ok, < 0 = bad argument or
 * unsupported shape, > 0 = a hipError_t.  m355_last_error() gives the message for the calling thread.
 * No C++ exception crosses this boundary.
 *
 * Each declaration names the interface of the reference (toddllm/gpt-oss-vllm, paths relative to
 * its root) whose work it takes over.
 */
#ifndef M355_H
#define M355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M355_API __attribute__((visibility("default")))

typedef void* m355_stream_t; /* a hipStream_t (NULL = the default stream) */

/* act_mode of m355_moe_nf4_forward */
#define M355_ACT_SILU_CONCAT 0 /* h = silu(gate) * up, gate = rows [0,FF), up = rows [FF,2FF) of W13:
                                  patches/unsloth_to_vllm_converter.py:312-315,
                                  patches/streaming_unsloth_bnb.py:269-270 */
#define M355_ACT_SWIGLU_OAI 1  /* gpt-oss's own activation: gate = min(gate, limit), up = clamp(up, -limit, limit),
                                  h = (up + 1) * gate * sigmoid(alpha * gate) (GptOssExperts._apply_gate,
                                  transformers models/gpt_oss/modeling_gpt_oss.py:79-85); the reference replaces it
                                  with plain silu (patches/unsloth_to_vllm_converter.py:312-315) */
/* flags of m355_moe_nf4_forward */
#define M355_FLAG_FP16_COMBINE 1 /* routing weights rounded to fp16 and out accumulated over k in fp16, in
                                    k order, as the reference's fp16 tensors do
                                    (patches/unsloth_to_vllm_converter.py:301,306,321) */
/* Profiling aid: leave out stages of the forward; the workspace must hold what a full call left there. */
#define M355_FLAG_SKIP_ROUTE 0x100
#define M355_FLAG_SKIP_GATE_UP 0x200
#define M355_FLAG_SKIP_DOWN 0x400
#define M355_FLAG_SKIP_COMBINE 0x800
/* Kernel choice, normally made from the mean number of (token, k) pairs per local expert: FORCE_STREAM = the
   HBM-bound kernel for few tokens per expert (token tiles of <= 16), FORCE_TILE = the MFMA-bound kernel for many
   (token tiles of 256, weights dequantised once per tile).  Both compute the same layer. */
#define M355_FLAG_FORCE_STREAM 0x1000
#define M355_FLAG_FORCE_TILE 0x2000
/* With FORCE_STREAM: token tiles of 32 instead of 16 (two 16-token sets share one pass over an expert's weights), the
   form the library itself picks from 16 pairs per local expert on.  A rank that holds a shard of the experts sees the
   token slots of all ranks but only its share of the pairs, so there the caller, who knows the global expert count,
   makes the choice. */
#define M355_FLAG_STREAM_WIDE 0x4000
/* The combine step (out[t] = sum over k of w * y) rides in the down stage's launch when the streaming kernel runs both
   stages on a small batch (up to 12 (token, k) pairs); SEPARATE_COMBINE makes it its own launch there too, as it is for
   larger batches and after the tile kernel; FOLD_COMBINE folds it at any batch size the streaming kernel takes.  Same
   bits every way. */
#define M355_FLAG_SEPARATE_COMBINE 0x8000
#define M355_FLAG_FOLD_COMBINE 0x10000
/* m355_moe_nf4_forward_routed up to 24 tokens computes the router logits as a launch of its own that uses every CU (a
   wave per token and 4 experts) and picks the top-k inside the gate_up launch, instead of running m355_router_topk (a
   workgroup per token) in front.  ROUTER_LAUNCH keeps m355_router_topk; same bits either way.  LOGITS_READY (profiling
   aid): the logits of these tokens are still in the workspace from an earlier call that ran with M355_FLAG_SKIP_DOWN;
   the logits launch is left out. */
#define M355_FLAG_ROUTER_LAUNCH 0x20000
#define M355_FLAG_LOGITS_READY 0x40000

M355_API int m355_version(void);
M355_API const char* m355_last_error(void);
/*
 * Per-device set-up ahead of time: the kernel attributes (dynamic-LDS limits) that the entry points otherwise set on
 * their first call on a device.  Those calls cannot be captured, so a host that captures HIP graphs calls this once per
 * device before its first capture; calling it again is free.  No counterpart in the reference, which runs vLLM with
 * --enforce-eager (patches/gpt_oss_server.py:45).
 */
M355_API int m355_init(int device);

/*
 * NF4 packed bytes -> fp16.  q holds n/2 bytes in checkpoint order (element 2i = low nibble of byte
 * i), absmax one fp32 per `blocksize` elements, out n fp16.  out[i] = fp16_rne(NF4[code_i] * absmax[i /
 * blocksize]) with the product taken in fp32: bit-identical to
 * patches/streaming_unsloth_bnb.py:175-222 (dequant_unsloth_expert_cpu, the divisible-blocksize
 * branch :199-202, which is the one real checkpoints take).
 * Requires n % blocksize == 0, blocksize % 8 == 0, q/out 16-byte aligned.
 */
M355_API int m355_nf4_dequant(const uint8_t* q, const float* absmax, int64_t n, int blocksize,
                              void* out_f16, m355_stream_t stream);

/*
 * Repack one expert matrix from checkpoint order into the tile order the fused kernels stream.
 * W is [rows][cols] NF4, row-major, two elements per byte (low nibble first), absmax [rows*cols/64]
 * fp32 (already un-double-quantized by the rules of patches/streaming_unsloth_bnb.py:148-173).
 * Output sizes equal the input sizes: q_tiled rows*cols/2 bytes, absmax_tiled rows*cols/64 floats.
 * Requires rows % 32 == 0, cols % 64 == 0.  Replaces the re-quantisation the reference performs when it
 * moves each expert to the GPU (patches/streaming_unsloth_bnb.py:48-65): the original blocks are kept.
 * Tile order (p = row-tile pair, b = 64-column block, lane = 16*g + r):
 *   q_tiled[((p*NB + b)*64 + lane)*16 + 0..7 ] = 8 bytes of row 16*p + r,           columns 64b+16g .. +15
 *   q_tiled[((p*NB + b)*64 + lane)*16 + 8..15] = 8 bytes of row 16*(p + rows/32) + r, same columns
 *   absmax_tiled[(((p*2 + half)*4 + g)*NB + b)*4 + i] = absmax of row 16*(p + half*rows/32) + 4g + i, block b
 */
M355_API int m355_repack_nf4(const uint8_t* q, const float* absmax, int rows, int cols, uint8_t* q_tiled,
                             float* absmax_tiled, m355_stream_t stream);
/* row_order 1: the checkpoint interleaves gate and up rows (gate = rows 0, 2, 4, ..., up = rows 1, 3, 5, ..., as
 * GptOssExperts stores gate_up_proj, modeling_gpt_oss.py:80); the tiled image still holds gate rows then up rows.
 * row_order 0 is m355_repack_nf4. */
M355_API int m355_repack_nf4_ex(const uint8_t* q, const float* absmax, int rows, int cols, int row_order,
                                uint8_t* q_tiled, float* absmax_tiled, m355_stream_t stream);

/*
 * Router: logits = fp16(x . wr^T + br), top-k by value (ties -> lower expert id), softmax over the k
 * selected logits.  x [T][H] fp16, wr [E][H] fp16, br [E] fp16 or NULL, ids [T][k] int32, w [T][k] fp32.
 * Replaces the fp16 router nn.Linear that patches/monkey_patch_router_fp16.py:6-17 forces, plus
 * torch.topk + softmax of patches/unsloth_to_vllm_converter.py:291-292.  E <= 1024, k <= 32, H % 8 == 0.
 * logits_out: optional [T][E] fp16.
 */
M355_API int m355_router_topk(const void* x_f16, const void* wr_f16, const void* br_f16, int T, int H, int E,
                              int k, int32_t* ids, float* w, void* logits_out_f16, m355_stream_t stream);

/* Same selection on logits the caller already has ([T][E] fp16): the FusedMoE.apply(router_logits=...) form. */
M355_API int m355_topk_softmax(const void* logits_f16, int T, int E, int k, int32_t* ids, float* w,
                               m355_stream_t stream);

/*
 * One MoE layer on this rank's experts: out[t] = sum_k w[t][k] * W2_e( act(W13_e x[t]) ), e = ids[t][k],
 * over the pairs whose expert lies in [e_first, e_first + E_local); other pairs contribute nothing
 * (another rank owns them).  x [T][H] fp16, ids [T][k] int32 (global expert ids), w [T][k] fp32,
 * w13_q [E_local][2FF*H/2] and w2_q [E_local][H*FF/2] in m355_repack_nf4 tile order (W13 is [2FF][H], W2 is
 * [H][FF]), their absmax in tile order, out [T][H] fp16.  blocksize is fixed at 64.
 * This is FusedMoE's quant_method.apply for the plug-in that patches/bnb_moe_4bit.py:22-48 declares, with
 * the forward semantics of patches/unsloth_to_vllm_converter.py:282-325.
 * Requires H % 64 == 0, FF % 64 == 0, k <= 32; workspace >= m355_moe_workspace_bytes(...), 256-byte aligned.
 */
M355_API int m355_moe_nf4_forward(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                  const uint8_t* w13_q, const float* w13_absmax, const uint8_t* w2_q,
                                  const float* w2_absmax, int E_local, int e_first, int H, int FF, int act_mode,
                                  int flags, void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream);
/* The same with per-expert biases and the activation's constants: b13_f16 [E_local][2*FF] (gate entries, then up
 * entries) is added to the gate_up product before the activation, b2_f16 [E_local][H] to the down product before
 * the routing weight (GptOssExperts.forward, modeling_gpt_oss.py:109-112); either may be NULL.  alpha and limit
 * are read by M355_ACT_SWIGLU_OAI only (gpt-oss: 1.702 and 7.0). */
M355_API int m355_moe_nf4_forward_ex(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                     const uint8_t* w13_q, const float* w13_absmax, const uint8_t* w2_q,
                                     const float* w2_absmax, int E_local, int e_first, int H, int FF, int act_mode,
                                     int flags, const void* b13_f16, const void* b2_f16, float alpha, float limit,
                                     void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream);
/*
 * m355_moe_nf4_forward_ex for callers that cannot allocate and have no workspace to pass: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  The workspace lives in a
 * device array inside the library, M355_STATIC_WORKSPACE_BYTES per GPU, held from the library's load to the end of the
 * process; the call allocates and frees nothing.  Same arguments and results (bit for bit) as m355_moe_nf4_forward_ex
 * without `workspace` / `ws_bytes`.  Shapes whose m355_moe_workspace_bytes(T, k, H, FF, E_local) exceeds the array are
 * refused with -6 (gpt-oss shapes: up to 181 tokens at top-4).  All calls on one GPU share that GPU's array: two calls that
 * overlap on one GPU (different streams, or different host threads) race; calls on one stream are ordered.  The array's
 * address is looked up once per device, by m355_init or by the first call on the device (which therefore must not sit
 * inside a graph capture unless m355_init ran).
 */
#define M355_STATIC_WORKSPACE_BYTES (8u << 20)
M355_API int m355_moe_nf4_forward_static(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                         const uint8_t* w13_q, const float* w13_absmax, const uint8_t* w2_q,
                                         const float* w2_absmax, int E_local, int e_first, int H, int FF, int act_mode,
                                         int flags, const void* b13_f16, const void* b2_f16, float alpha, float limit,
                                         void* out_f16, m355_stream_t stream);
/* The same with the absmax kept as the checkpoint stores it: one code byte per 64-weight block (tile order of
 * m355_repack_absmax_u8) and, per expert, a 256-entry fp32 table that maps a code to its absmax -- the
 * `nested_quant_map[code] * scale` of patches/streaming_unsloth_bnb.py:150-171 with the scalar (the single
 * nested_absmax, or its mean in the fallback real checkpoints take) multiplied in on the host, entry by entry, so
 * table[code] is bit for bit the fp32 absmax the other entry points are handed.  0.52 instead of 0.56 bytes of HBM
 * traffic per weight.  w13_table [E_local][256], w2_table [E_local][256]. */
M355_API int m355_moe_nf4_forward_q8(const void* x_f16, int T, const int32_t* ids, const float* w, int k,
                                     const uint8_t* w13_q, const uint8_t* w13_absmax_u8, const float* w13_table,
                                     const uint8_t* w2_q, const uint8_t* w2_absmax_u8, const float* w2_table, int E_local,
                                     int e_first, int H, int FF, int act_mode, int flags, const void* b13_f16,
                                     const void* b2_f16, float alpha, float limit, void* workspace, size_t ws_bytes,
                                     void* out_f16, m355_stream_t stream);
/* Router and experts in one call: ids_out [T][k] / w_out [T][k] receive what m355_router_topk(x, wr, br) gives (same
 * bits), then the layer runs as m355_moe_nf4_forward_ex / _q8 would (absmax: pass either the fp32 pointers or, with
 * w13_absmax == NULL, the byte form).  Up to 24 tokens (and 256 global experts) the logits are computed by a launch that
 * spreads them over every CU (a wave per token and 4 experts) and the gate_up launch picks the top-k itself, every
 * workgroup with the router kernel's own code; beyond that m355_router_topk runs in front.  Takes over the router nn.Linear that
 * patches/monkey_patch_router_fp16.py:6-17 casts to fp16 and the top-k + softmax of
 * patches/unsloth_to_vllm_converter.py:291-292, together with the forward of :282-325. */
M355_API int m355_moe_nf4_forward_routed(const void* x_f16, int T, const void* wr_f16, const void* br_f16, int E_global,
                                         int k, int32_t* ids_out, float* w_out, const uint8_t* w13_q,
                                         const float* w13_absmax, const uint8_t* w13_absmax_u8, const float* w13_table,
                                         const uint8_t* w2_q, const float* w2_absmax, const uint8_t* w2_absmax_u8,
                                         const float* w2_table, int E_local, int e_first, int H, int FF, int act_mode,
                                         int flags, const void* b13_f16, const void* b2_f16, float alpha, float limit,
                                         void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream);
/* The same for a caller that holds finished router logits [T][E_global] fp16, which is what vLLM's FusedMoE layer hands
 * its quantisation method (router_logits of FusedMoEMethodBase.apply; the reference's apply is
 * patches/unsloth_to_vllm_converter.py:282-325, its top-k + softmax :291-292): ids_out / w_out receive what
 * m355_topk_softmax(logits) gives (same bits), then the layer runs.  Up to 24 tokens the gate_up launch picks the top-k
 * itself, so the call is two launches (three with a separate combine); beyond, m355_topk_softmax runs in front. */
M355_API int m355_moe_nf4_forward_logits(const void* x_f16, int T, const void* logits_f16, int E_global, int k,
                                         int32_t* ids_out, float* w_out, const uint8_t* w13_q, const float* w13_absmax,
                                         const uint8_t* w13_absmax_u8, const float* w13_table, const uint8_t* w2_q,
                                         const float* w2_absmax, const uint8_t* w2_absmax_u8, const float* w2_table,
                                         int E_local, int e_first, int H, int FF, int act_mode, int flags,
                                         const void* b13_f16, const void* b2_f16, float alpha, float limit, void* workspace,
                                         size_t ws_bytes, void* out_f16, m355_stream_t stream);
/*
 * Causal self-attention over whole prompts (prefill), the companion of m355_paged_attn_decode for the tokens that
 * arrive together: q [tokens][Hq][D], k / v [tokens][Hkv][D] fp16 with token strides in elements (so slices of a fused
 * qkv tensor can be passed), sequences packed one after the other, seq_starts [B + 1] (device) their first tokens;
 * query i of a sequence sees keys 0..i of the same sequence (window > 0: the last `window` of them; sinks [Hq] fp16 or
 * NULL: one more logit per head in the denominator).  max_seq = the longest sequence (host).  D == 64, Hq == 8 Hkv.
 * Takes over the flash-attn 2 varlen call the reference reaches through vLLM's FA2 backend for prompts
 * (patches/patch_1..3, VLLM_FLASH_ATTN_VERSION=2 in patches/gpt_oss_server.py:26).
 */
M355_API int m355_attn_prefill(const void* q_f16, const void* k_f16, const void* v_f16, const int32_t* seq_starts, int B,
                               int max_seq, int Hq, int Hkv, int D, int64_t q_stride, int64_t kv_stride, int64_t o_stride,
                               float scale, const void* sinks_f16, int window, void* out_f16, m355_stream_t stream);
/* absmax codes [rows*cols/64] bytes, checkpoint order -> the tile order of absmax_tiled in m355_repack_nf4. */
M355_API int m355_repack_absmax_u8(const uint8_t* absmax_u8, int rows, int cols, int row_order, uint8_t* absmax_tiled,
                                   m355_stream_t stream);
M355_API size_t m355_moe_workspace_bytes(int T, int k, int H, int FF, int E_local);
/* Measuring aid, no counterpart in the reference: m355_time_next_launch() arms the calling thread so that the next launch
 * of the streaming MoE GEMM kernel it makes (the gate_up or the down stage of a small-batch m355_moe_nf4_forward*, not under
 * graph capture) carries start / stop events of its own; m355_timed_launch_us() waits for that launch and returns the
 * kernel's own duration in microseconds -- what a kernel trace reports for it, without the gap to neighbouring launches
 * and without the cost of event markers around the call.  The first call on a thread creates the two events. */
M355_API int m355_time_next_launch(void);
M355_API int m355_timed_launch_us(float* us);

/*
 * Decode attention over a paged KV cache: one query token per sequence, grouped-query heads, scores in
 * fp32, no sinks and no sliding window -- the attention the reference is left with after
 * patches/patch_3_disable_sinks.py:36-38 and patches/patch_2_force_fa2.py (FlashAttention-2 paged decode).
 * q [B][Hq][D] fp16, kcache/vcache [num_blocks][block_size][Hkv][D] fp16, block_tables [B][max_blocks] int32,
 * ctx_lens [B] int32 (cached tokens including the current one), out [B][Hq][D] fp16.
 * Requires D == 64, Hq / Hkv == 8.  workspace >= m355_paged_attn_workspace_bytes(...) (may be NULL when that is 0).
 */
M355_API int m355_paged_attn_decode(const void* q_f16, const void* kcache_f16, const void* vcache_f16,
                                    const int32_t* block_tables, const int32_t* ctx_lens, int B, int Hq, int Hkv,
                                    int D, int block_size, int max_blocks, int max_ctx, float scale,
                                    void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream);
M355_API size_t m355_paged_attn_workspace_bytes(int B, int Hq, int D, int max_ctx);
/*
 * The same with gpt-oss's two attention extras, which the reference switches off because its GPU's FA2 lacks them
 * (patches/patch_3_disable_sinks.py:29-40): sinks_f16 [Hq] (nullable) is one extra logit per query head in the
 * softmax denominator (eager_attention_forward, transformers models/gpt_oss/modeling_gpt_oss.py:243-251); window > 0
 * limits a query to the last `window` cached tokens (config.sliding_window on "sliding_attention" layers), 0 = all.
 */
M355_API int m355_paged_attn_decode_ex(const void* q_f16, const void* kcache_f16, const void* vcache_f16,
                                       const int32_t* block_tables, const int32_t* ctx_lens, int B, int Hq, int Hkv, int D,
                                       int block_size, int max_blocks, int max_ctx, float scale, const void* sinks_f16,
                                       int window, void* workspace, size_t ws_bytes, void* out_f16, m355_stream_t stream);

/*
 * Expert-parallel exchange helpers.  No counterpart in the reference (it is single-GPU, tensor_parallel_size=1:
 * patches/test_gptoss_bnb.py:30); they are the byte movers either side of the two all-to-alls that shard the
 * experts of one layer over the GPUs of a node (expert e lives on rank e / experts_per_rank).
 *   m355_ep_pack    builds send [n_ranks][T][row]: row = [x: H fp16][ids: k int32][w: k fp32] padded to 16 B
 *                   (m355_ep_row_bytes); a token with no expert on a destination gets ids = -1 there, no payload.
 *   m355_ep_unpack  splits received rows [NT][row] into x_all [NT][H], ids_all [NT][k], w_all [NT][k].
 *   m355_ep_reduce  out[t] = sum_r recv[r][t] (fp16 partial rows, fp32 sum in rank order), recv [n_ranks][T][H].
 */
/*   m355_router_topk_ep_pack   m355_router_topk and m355_ep_pack in one launch: the workgroup that routes a token also
 *                   writes its rows of the send buffer (same ids, weights and bytes as the two calls).
 *   m355_moe_nf4_forward_ep_rows   m355_moe_nf4_forward_ex / _q8 (absmax: pass either the fp32 pointers or, with
 *                   w13_absmax == NULL, the byte form) on the token slots of a receive buffer as they lie there, rows
 *                   [NT][row]: activations, expert ids and routing weights are read in place (no m355_ep_unpack); a slot
 *                   whose ids are -1 yields a zero row.
 *   m355_ep_reduce_add_rmsnorm   m355_ep_reduce followed by m355_add_rmsnorm(x, delta = the reduced rows) in one launch
 *                   (same bits): the sum over ranks lands in the residual add + RMSNorm that follows the MoE block.
 */
M355_API size_t m355_ep_row_bytes(int H, int k);
M355_API int m355_router_topk_ep_pack(const void* x_f16, const void* wr_f16, const void* br_f16, int T, int H, int E, int k,
                                      int32_t* ids, float* w, int n_ranks, int experts_per_rank, void* send,
                                      m355_stream_t stream);
M355_API int m355_moe_nf4_forward_ep_rows(const void* rows, int NT, int k, const uint8_t* w13_q, const float* w13_absmax,
                                          const uint8_t* w13_absmax_u8, const float* w13_table, const uint8_t* w2_q,
                                          const float* w2_absmax, const uint8_t* w2_absmax_u8, const float* w2_table,
                                          int E_local, int e_first, int H, int FF, int act_mode, int flags,
                                          const void* b13_f16, const void* b2_f16, float alpha, float limit, void* workspace,
                                          size_t ws_bytes, void* out_f16, m355_stream_t stream);
M355_API int m355_ep_reduce_add_rmsnorm(const void* recv_f16, int n_ranks, const void* x_f16, const void* weight_f16, float eps,
                                        int T, int H, void* resid_out_f16, void* y_f16, m355_stream_t stream);
M355_API int m355_ep_pack(const void* x_f16, const int32_t* ids, const float* w, int T, int k, int H, int n_ranks,
                          int experts_per_rank, void* send, m355_stream_t stream);
M355_API int m355_ep_unpack(const void* recv, int NT, int k, int H, void* x_all_f16, int32_t* ids_all, float* w_all,
                            m355_stream_t stream);
M355_API int m355_ep_reduce(const void* recv_f16, int n_ranks, int T, int H, void* out_f16, m355_stream_t stream);

/*
 * The per-token operators around the hot path in a gpt-oss decoder layer.  The reference leaves them to
 * upstream vLLM (it patches only the MoE, the router dtype and the attention backend choice); the arithmetic
 * here follows transformers' models/gpt_oss/modeling_gpt_oss.py operation by operation, on fp16 tensors.
 *
 * m355_add_rmsnorm: r = fp16(x + delta) (delta NULL: r = x); y = fp16(weight * (float(r) * rsqrt(mean(float(r)^2) +
 *   eps))); resid_out (nullable) receives r.  GptOssRMSNorm.forward (modeling_gpt_oss.py:57-62) with the residual
 *   add of GptOssDecoderLayer.forward in front.  x, delta, y, resid_out [T][H] fp16; weight [H] fp16; H % 8 == 0.
 *
 * m355_rope_kv_write: qkv [T][(Hq + 2*Hkv)*D] fp16 (q heads, k heads, v heads); cos/sin [max_pos][D/2] fp16 (YaRN
 *   scaling already folded in, as GptOssRotaryEmbedding.forward returns them); positions [T] int32; slots [T] int64
 *   (index of the token's row in the paged cache, block = slot / block_size; negative = do not write).  q and k are
 *   rotated as _apply_rotary_emb does (first half / second half of a head, every product and sum rounded to fp16);
 *   q goes to q_out [T][Hq][D], k and v into kcache / vcache [blocks][block_size][Hkv][D].  D % 16 == 0.
 */
/*
 * m355_linear_f16: out[T][N] = fp16(x[T][K] . w[N][K]^T + bias[N]) (+ resid[T][N], added as an fp16 tensor add), fp32
 *   accumulation: the q/k/v and o projections of GptOssAttention (modeling_gpt_oss.py:262-273), weights in
 *   nn.Linear layout.  Decode-sized: W is streamed once per 32 tokens.  N % 16 == 0, K % 64 == 0, K <= 4096;
 *   bias and resid may be NULL.
 */
M355_API int m355_linear_f16(const void* x_f16, int T, const void* w_f16, const void* bias_f16, int N, int K,
                             const void* resid_f16, void* out_f16, m355_stream_t stream);
M355_API int m355_add_rmsnorm(const void* x_f16, const void* delta_f16, const void* weight_f16, float eps, int T, int H,
                              void* resid_out_f16, void* y_f16, m355_stream_t stream);
/*
 * m355_add_rmsnorm followed by m355_router_topk on its output y, in one launch (below 1024 tokens): the norm in front of
 * the MoE block (GptOssDecoderLayer.forward, transformers models/gpt_oss/modeling_gpt_oss.py:372-374)
 * and the router that reads it (GptOssTopKRouter.forward, :126-130; the reference casts it to fp16,
 * patches/monkey_patch_router_fp16.py:1-19).  wr [E][H] fp16, br [E] fp16 or null; ids [T][k], w [T][k] as
 * m355_router_topk writes them.  Same bits as the two calls.
 */
M355_API int m355_add_rmsnorm_router(const void* x_f16, const void* delta_f16, const void* weight_f16, float eps, int T, int H,
                                     void* resid_out_f16, void* y_f16, const void* wr_f16, const void* br_f16, int E, int k,
                                     int32_t* ids, float* w, m355_stream_t stream);
/* m355_add_rmsnorm with the MoE combine step in front: the delta is formed here, exactly as the combine forms it, from
 * what a forward of the same T, k, H, FF, E_local called with M355_FLAG_SKIP_COMBINE left in `moe_workspace` (w: that
 * forward's routing weights, flags: its M355_FLAG_FP16_COMBINE), so a decoder stack spends one launch less per layer:
 * expert outputs -> residual add -> the next layer's RMSNorm (GptOssDecoderLayer.forward, modeling_gpt_oss.py, the
 * `residual + hidden_states` after the MLP followed by the next layer's input_layernorm). */
M355_API int m355_moe_combine_add_rmsnorm(const void* moe_workspace, size_t ws_bytes, int T, int k, int FF, int E_local,
                                          const float* w, int flags, const void* x_f16, const void* weight_f16, float eps,
                                          int H, void* resid_out_f16, void* y_f16, m355_stream_t stream);
M355_API int m355_rope_kv_write(const void* qkv_f16, const void* cos_f16, const void* sin_f16, const int32_t* positions,
                                const int64_t* slots, int T, int Hq, int Hkv, int D, int block_size, int max_pos,
                                void* q_out_f16, void* kcache_f16, void* vcache_f16, m355_stream_t stream);
/*
 * m355_qkv_rope_kv_write: the q/k/v projection (m355_linear_f16 with wqkv [(Hq + 2*Hkv)*D][K], bias bqkv or NULL) and
 *   m355_rope_kv_write on its result in one call: GptOssAttention.forward up to the attention itself
 *   (modeling_gpt_oss.py:262-273 with _apply_rotary_emb, :295-320 for the cache update).  Up to 32 tokens it is one launch
 *   (the projection's kernel rotates and stores the pairs it has just summed: a dot-product form up to 4 tokens, a
 *   matrix-core form beyond, for shapes with at most 16 (d, d + D/2) pairs of rows per CU; qkv_scratch may be NULL
 *   there); otherwise the two launches through qkv_scratch [T][(Hq + 2*Hkv)*D].  The same bits either way.
 */
M355_API int m355_qkv_rope_kv_write(const void* x_f16, int T, const void* wqkv_f16, const void* bqkv_f16, int K,
                                    const void* cos_f16, const void* sin_f16, const int32_t* positions, const int64_t* slots,
                                    int Hq, int Hkv, int D, int block_size, int max_pos, void* qkv_scratch_f16,
                                    void* q_out_f16, void* kcache_f16, void* vcache_f16, m355_stream_t stream);
/*
 * m355_norm_qkv_rope_kv_write: the residual add + input RMSNorm of a decoder layer in front of m355_qkv_rope_kv_write, in
 *   the same call: GptOssDecoderLayer.forward from `residual + hidden_states` of the layer before through
 *   input_layernorm to the attention's q/k/v, rotary embedding and cache update (modeling_gpt_oss.py, as cited above).
 *   x [T][K]: the residual stream; delta [T][K] (or NULL) is added first -- or, with moe_workspace non-NULL, the expert
 *   outputs that a forward of the same T, k, K (= H), FF, E_local called with M355_FLAG_SKIP_COMBINE left there
 *   (combine_w: its routing weights, flags: its M355_FLAG_FP16_COMBINE), as m355_moe_combine_add_rmsnorm forms them.
 *   resid_out [T][K] (or NULL) receives the sum.  Up to 4 tokens it is ONE launch: every workgroup of the projection
 *   normalises the rows for itself while its first weights are on their way (h_scratch, qkv_scratch may be NULL);
 *   beyond, the norm's launch into h_scratch [T][K] followed by m355_qkv_rope_kv_write.  The same bits either way.
 */
M355_API int m355_norm_qkv_rope_kv_write(const void* x_f16, const void* delta_f16, const void* moe_workspace, size_t ws_bytes,
                                         int k, int FF, int E_local, const float* combine_w, int flags,
                                         const void* ln_weight_f16, float eps, int T, int K, const void* wqkv_f16,
                                         const void* bqkv_f16, const void* cos_f16, const void* sin_f16,
                                         const int32_t* positions, const int64_t* slots, int Hq, int Hkv, int D,
                                         int block_size, int max_pos, void* resid_out_f16, void* h_scratch_f16,
                                         void* qkv_scratch_f16, void* q_out_f16, void* kcache_f16, void* vcache_f16,
                                         m355_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* M355_H */
