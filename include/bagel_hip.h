/* libbagel_hip.so -- C ABI of the MI355X (gfx950) kernels behind BAGEL's unified multimodal forward path.
 *
 * The reference (ByteDance-Seed/Bagel) is pure Python; its only by-name native seam is
 *   flash_attn.flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal)
 * (modeling/bagel/qwen2_navit.py:24,361-370,579-588; modeling/bagel/siglip_navit.py:18,232-241); every other
 * native op is reached through torch (cuBLAS F.linear, eager elementwise / index kernels, cuDNN).  Each entry
 * point below names the reference call site(s) it replaces.  Conventions:
 *   - plain device pointers + explicit sizes/strides (element units) + hipStream_t; no torch types;
 *   - bf16 tensors are raw uint16 bit patterns; "ld*" are row strides in ELEMENTS;
 *   - return 0 on success, <0 on error (bagel_hip_last_error() gives the thread-local message);
 *   - ops never allocate, never synchronise, hold no global mutable state; async on `stream`;
 *   - entry points may be called from several host threads (each on its own stream and workspaces).  The BAGEL_* environment
 *     variables the launchers consult are A/B tuning knobs for measurements: each is READ ONCE PER PROCESS at the first launch that
 *     consults it (thread-safe), later changes of the environment have no effect, and the defaults are what bench.py measures.
 */
#ifndef BAGEL_HIP_H
#define BAGEL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* bagel_stream_t; /* == hipStream_t */

int bagel_hip_version(void);
const char* bagel_hip_last_error(void);
const char* bagel_hip_arch(void);

/* epilogues of bagel_gemm_bf16 */
#define BAGEL_EPI_NONE 0       /* C = bf16(A W^T + bias)                                              */
#define BAGEL_EPI_GELU_TANH 1  /* C = bf16(gelu_tanh(bf16(A W^T + bias)))   siglip_navit.py:257, modeling_utils.py:122 */
#define BAGEL_EPI_SILU 2       /* C = bf16(silu(bf16(...)))                 modeling_utils.py:82 (TimestepEmbedder)    */
#define BAGEL_EPI_SWIGLU16 3   /* W rows interleaved [16 gate | 16 up]...;  C[:, N/2] = bf16(bf16(silu(g)) * u)       */
                               /*                                           modeling_qwen2.py:201 (Qwen2MLP)           */

/* C[M,N] = A[M,K] W[N,K]^T (+bias)(act)(+R), bf16 in/out, fp32 MFMA accumulate.  Up to two row groups, each with
 * its own weights/bias and optional gather (a_rows) / scatter (c_rows, also used for R) index lists: the MoT
 * expert routing of qwen2_navit.py:526-548,593-594,812-820 without gather/scatter kernels.
 * Replaces F.linear at qwen2_navit.py:515-517,529-536,591-594; modeling_qwen2.py:200-201; bagel.py:803,832,978;
 * modeling_utils.py:107-110,120-124; siglip_navit.py:190,216-218,243,256-258.
 * variant: 0 = 128x128 tile/256 threads, 1 = 256x256/512, 2 = 256x128/256, 3 = 256x256 two-group ping-pong (K % 64 == 0,
 * else falls back to 1), 4 = its persistent form (one workgroup per CU walks the tile list), 5 = variant 4 with SGPR-base LDS-DMA
 * addresses (one address register per DMA instruction: the kernel is DMA-issue bound) -- by passing 5 the CALLER PROMISES that every
 * A row and every W row the launch touches lies within 4 GiB of the A / W base pointer (dense rows are checked and fall back to 4;
 * gathered rows cannot be seen from this side of the ABI).  Variants 3, 4 and 5 give bit-identical results.  K % 8 == 0, N % 8 == 0. */
int bagel_gemm_bf16(const void* A, int64_t lda,
                    const void* W0, const void* bias0, const int32_t* a_rows0, const int32_t* c_rows0, int32_t M0,
                    const void* W1, const void* bias1, const int32_t* a_rows1, const int32_t* c_rows1, int32_t M1,
                    int64_t ldw, const void* R, int64_t ldr, void* C, int64_t ldc,
                    int32_t N, int32_t K, int32_t epilogue, int32_t variant, bagel_stream_t stream);

/* bagel_gemm_bf16 with a caller-owned fp32 workspace (16-byte aligned; touched only during the call): when the tile count leaves the
 * last round of the persistent kernel (variant 4) nearly empty -- the LLM prefill of an understanding request, M = 4936: o / down have
 * 280 tiles on 256 CUs -- the leftover tiles are cut along K into parts that fill the chip once more, each part leaves a 256 KB fp32
 * partial tile in the workspace and a small pass sums them and applies the epilogue (bias or residual, same rounding points).  Same
 * result as bagel_gemm_bf16 up to the fp32 summation order of those tiles; with workspace == NULL it IS bagel_gemm_bf16. */
int bagel_gemm_bf16_ws(const void* A, int64_t lda,
                       const void* W0, const void* bias0, const int32_t* a_rows0, const int32_t* c_rows0, int32_t M0,
                       const void* W1, const void* bias1, const int32_t* a_rows1, const int32_t* c_rows1, int32_t M1,
                       int64_t ldw, const void* R, int64_t ldr, void* C, int64_t ldc,
                       int32_t N, int32_t K, int32_t epilogue, int32_t variant, void* workspace, int64_t workspace_bytes,
                       bagel_stream_t stream);

/* Qwen2RMSNorm (modeling_qwen2.py:54-59); expert_of_row (nullable) selects w1 for rows flagged 1
 * (input_layernorm_moe_gen etc., qwen2_navit.py:784-787,812-815,1079-1082). */
int bagel_rmsnorm_bf16(const void* x, int64_t ldx, const void* w0, const void* w1, const int32_t* expert_of_row,
                       void* y, int64_t ldy, int32_t rows, int32_t cols, float eps, bagel_stream_t stream);

/* nn.LayerNorm with affine (siglip_navit.py:266-269,342). */
int bagel_layernorm_bf16(const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy,
                         int32_t rows, int32_t cols, float eps, bagel_stream_t stream);

/* Qwen2RotaryEmbedding.forward (modeling_qwen2.py:130-150): cos/sin(pos * inv_freq) rounded to bf16, [rows, half]. */
int bagel_rope_table(const int64_t* position_ids, const float* inv_freq, void* cos_out, void* sin_out,
                     int32_t rows, int32_t half_dim, bagel_stream_t stream);

/* q_norm/k_norm + apply_rotary_pos_emb + bf16 casts of PackedAttentionMoT.forward_inference
 * (qwen2_navit.py:518-557), in place on the fused [q|k|v] projection rows. gen_mode selects the fp32 pipeline. */
int bagel_qknorm_rope_bf16(void* qkv, int64_t ld, const void* cos_tab, const void* sin_tab, const void* q_w0,
                           const void* k_w0, const void* q_w1, const void* k_w1, const int32_t* expert_of_row,
                           int32_t rows, int32_t nq, int32_t nkv, int32_t head_dim, int32_t head_dim_padded,
                           float eps, int32_t gen_mode, int32_t use_norm, bagel_stream_t stream);

/* SigLIP 2-D RoPE (siglip_navit.py:102-142,224-230; config.rope): in place on the first `nheads` heads of every row of the
 * fused projection buffer (q heads then k heads): first half of a head rotated with (cos_h, sin_h)[pos], second half with
 * (cos_w, sin_w)[pos]; tables [positions, head_dim/2] bf16. */
int bagel_rope2d_bf16(void* qkv, int64_t ld, const void* cos_h, const void* sin_h, const void* cos_w, const void* sin_w,
                      const int64_t* pos_ids, int64_t rows, int32_t nheads, int32_t head_dim, int32_t head_dim_padded,
                      bagel_stream_t stream);

/* flash_attn_varlen_func replacement (qwen2_navit.py:579-588 / 361-370, siglip_navit.py:232-241).
 * Keys/values of sample b = [context rows cu_ctx[b]:cu_ctx[b+1] of (k_ctx, vt_ctx)] ++ [new rows cu_q[b]:cu_q[b+1] of
 * (k_new, vt_new)]  -- the merged layout of qwen2_navit.py:563-570 without the copy.  V is passed transposed:
 * vt[(g*head_dim + d) * ldvt + col_start[b] + key].  causal = bottom-right aligned (flash-attn >= 2.1). */
int bagel_attn_varlen_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                           int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx, int64_t ldvt_ctx,
                           void* out, int64_t ldo, const int32_t* cu_q, const int32_t* cu_ctx,
                           const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch, int32_t max_lq,
                           int32_t nq, int32_t nkv, int32_t head_dim, int32_t causal, float softmax_scale,
                           bagel_stream_t stream);

/* bagel_attn_varlen_bf16 with explicit [start, end) row ranges per sequence: the ranges need not tile the buffers and
 * the context ranges may overlap (prefixes of one key stream).  This is how the causal/full/noise block mask of the
 * training forward (data/data_utils.py:72-103, bagel.py:155-166, SDPA/flex call qwen2_navit.py:452-487) runs on the
 * same kernel: every split is a sequence whose context is the prefix of its sample's non-noise keys. */
int bagel_attn_varlen_ranges_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                                  int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx,
                                  int64_t ldvt_ctx, void* out, int64_t ldo, const int32_t* q_start,
                                  const int32_t* q_end, const int32_t* ctx_start, const int32_t* ctx_end,
                                  const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch,
                                  int32_t max_lq, int32_t nq, int32_t nkv, int32_t head_dim, int32_t causal,
                                  float softmax_scale, bagel_stream_t stream);

/* The same attention for a caller that holds the sequence lengths on the HOST (the engine's ForwardPlan): a host planner + a persistent
 * kernel (csrc/attention2.hip).  bagel_attn_plan is pure host code (no device is touched): from per-sample HOST arrays -- first
 * query / new-key row and length, first context row and length, V^T column starts (the layout of bagel_attn_varlen_bf16) -- it
 * writes the work list of `n_workers` persistent workgroups (a multiple of 8; the CU count) into `plan` (int32[plan_ints], host
 * memory):  [0] n_workers [1] n_items [2] n_comb [3] n_slots [4] item table offset [5] combine table offset [6] makespan
 * [7] total (tile steps), then worker_off[n_workers + 1], items[n_items][16], comb[n_comb][8].  Query tiles of <= 32 rows run in
 * head-per-wave form (one item serves the whole GQA group); items of a partial last round are split along the key axis when they
 * span >= split_min_tiles 64-key tiles (0 = default 16) and merged by a combine pass through `n_slots` partial slots of
 * 256 * (head_dim + 2) floats.  The caller copies the plan to the device once per forward SHAPE (it is the same for every layer)
 * and launches bagel_attn_planned_bf16 with the device copy; `partials` may be null when n_comb == 0.
 * Replaces flash_attn_varlen_func at qwen2_navit.py:579-588 / siglip_navit.py:232-241 like bagel_attn_varlen_bf16; items that are
 * not key-split produce bit-identical results to it.  A plan with MORE workers than the device has CUs selects the split-ring form of
 * the kernel (2 K + 2 V^T slots = 64 KB of LDS per workgroup, one tile step of DMA flight; csrc/attention2.hip): a round-6 experiment,
 * bit-identical, measured slower than the default (profiles/r06_attn_split_ring.log) -- plan with the CU count. */
int bagel_attn_plan(const int32_t* q_start, const int32_t* q_len, const int32_t* ctx_start, const int32_t* ctx_len,
                    const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch, int32_t nq, int32_t nkv,
                    int32_t causal, int32_t n_workers, int32_t split_min_tiles, int32_t* plan, int64_t plan_ints);
int bagel_attn_planned_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new,
                            int64_t ldvt_new, const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx, int64_t ldvt_ctx,
                            void* out, int64_t ldo, const int32_t* plan_dev, int32_t n_workers, int32_t n_comb,
                            int32_t off_items, int32_t off_comb, void* partials, int32_t head_dim, float softmax_scale,
                            bagel_stream_t stream);
/* TEST / TOOLING HOOK (not for integrators): resident workgroups per CU of the planned attention kernel -- split = 0 the unified 4-slot ring, 1 the split
 * 2 + 2 ring -- by the runtime's occupancy calculator; >= 0, or a negative error code. */
int bagel_debug_attn_occupancy(int32_t head_dim, int32_t split);

/* V[rows][nkv][D] -> V^T[nkv][D][cols] per sample (layout consumed by bagel_attn_varlen_bf16). */
int bagel_v_transpose_bf16(const void* v, int64_t ld_src, void* vt, int64_t ld_dst, const int32_t* cu_rows,
                           const int32_t* col_start, int32_t batch, int32_t max_len, int32_t nkv, int32_t head_dim,
                           bagel_stream_t stream);

/* Row gather/scatter copy: embed_tokens (bagel.py:277,377,508,796), KV-cache merge/append (qwen2_navit.py:563-570). */
int bagel_copy_rows_bf16(const void* src, int64_t ld_src, const int32_t* src_rows, void* dst, int64_t ld_dst,
                         const int32_t* dst_rows, int32_t n, int32_t cols, bagel_stream_t stream);

/* fp32 -> bf16 with strides; dst columns [cols, cols_padded) are zero-filled (autocast input cast, bagel.py:803). */
int bagel_f32_to_bf16(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, int32_t rows, int32_t cols,
                      int32_t cols_padded, bagel_stream_t stream);

/* TimestepEmbedder.timestep_embedding for one t (modeling_utils.py:88-104). */
int bagel_timestep_sinusoid(float t, const float* freqs, void* out, int32_t half, bagel_stream_t stream);

/* seq[rows[i]] = bf16(bf16(seq[rows[i]] + temb) + pos_table[pos_ids[i]])   (bagel.py:523,803-806). */
int bagel_flow_add_bf16(void* seq, int64_t ld, const int32_t* rows, const void* temb, const void* pos_table,
                        int64_t ld_pos, const int64_t* pos_ids, int32_t n, int32_t cols, bagel_stream_t stream);

/* x[i] = bf16(x[i] + table[ids[i]])   (siglip_navit.py:192; bagel.py:391-392). */
int bagel_add_table_rows_bf16(void* x, int64_t ld, const void* table, int64_t ld_table, const int64_t* ids, int32_t n,
                              int32_t cols, bagel_stream_t stream);

/* CFG combine + renorm (bagel.py:873-905).  mode 0 global / 1 channel / 2 text_channel.  Stage 1 writes the
 * (un)scaled velocity to tmp and, for mode 0, per-block partial sums; stage 2 applies the global scale (mode 0)
 * and the Euler update x_t -= bf16(v_t * dt) (bagel.py:746). */
int bagel_cfg_stage1(const void* v, const void* v_cfg_text, const void* v_cfg_img, void* tmp, float* partials,
                     int32_t max_partials, int32_t n_rows, int32_t cols, float text_scale, float img_scale,
                     float renorm_min, int32_t mode, int32_t* nparts_out, bagel_stream_t stream);
int bagel_cfg_stage2_euler(float* x_t, const void* v_or_tmp, const float* partials, int32_t nparts, float renorm_min,
                           float dt, int64_t n_elems, int32_t use_global_scale, bagel_stream_t stream);

/* torch.argmax(logits, -1) (bagel.py:984). */
int bagel_argmax_bf16(const void* logits, int64_t ld, int64_t* out, int32_t rows, int32_t cols, bagel_stream_t stream);

/* Sampled next token on the device (bagel.py:980-983: multinomial(softmax(logits / temperature))): Gumbel-max over Philox4x32-10 uniforms keyed by
 * `seed` with the counter (column / 4, row, *step_ctr, 0) -- the same categorical distribution as torch.multinomial, drawn inside the captured decode
 * step (step_ctr = the session's device-side step counter; NULL = step 0).  Not torch's RNG stream; oracle/sampling.py restates it. */
int bagel_sample_gumbel_bf16(const void* logits, int64_t ld, int64_t* out, int32_t rows, int32_t cols, float temperature, int64_t seed,
                             const int32_t* step_ctr, bagel_stream_t stream);
/* TEST HOOK (not for integrators): the sampler's uniform -> Gumbel map on caller-chosen 32-bit draws, g[i] = -log(-log(((x[i] >> 9) + 0.5) * 2^-23)):
 * finite for every x, including the edge draws 0 and 0xFFFFFFFF that no seed can be steered to. */
int bagel_debug_gumbel_of_u32(const uint32_t* x, float* g, int32_t n, bagel_stream_t stream);

/* ---- autoregressive text decode (Bagel.generate_text, bagel.py:930-1000) ------------------------------------ */
/* Skinny GEMM for M <= a few rows (HBM-bound weight streaming): C[M,N] = A[M,K] W[N,K]^T with the epilogues and
 * roundings of bagel_gemm_bf16, plus an optional fused Qwen2RMSNorm of the A rows (norm_w != NULL:
 * A <- w * bf16(A * rsqrt(mean(A^2) + eps)), modeling_qwen2.py:54-59) so a decode layer is 2 launches shorter.
 * Replaces F.linear at Lq = 1: qwen2_navit.py:515-517,591; modeling_qwen2.py:200-201; lm_head bagel.py:978;
 * TimestepEmbedder modeling_utils.py:107-110.  K % 8 == 0, N even; in-place residual (R == C) is allowed. */
int bagel_gemv_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* R,
                    int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps, int32_t M, int32_t N,
                    int32_t K, int32_t epilogue, bagel_stream_t stream);

/* Skinny MFMA GEMM for 2..64 rows (batched decode steps, short-prompt prefill): same contract and roundings as
 * bagel_gemm_bf16 on dense operands, but every wave streams its own 16 weight rows straight from HBM into the MFMA
 * (no LDS tile), so the whole chip pulls on W even when N/128 would give a few dozen workgroups.
 * K % 32 == 0, N % 16 == 0 (SwiGLU16: N % 32 == 0); in-place residual (R == C) allowed. */
int bagel_gemm_skinny_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* R,
                           int64_t ldr, void* C, int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                           bagel_stream_t stream);

/* Batched-decode projection: C[M <= 32, N] = norm(A) W^T with the epilogues and rounding points of bagel_gemm_bf16 (bias, activation,
 * SwiGLU16 pairing, residual; R == C allowed) and an optional fused Qwen2RMSNorm of the A rows (norm_w != NULL).  Replaces, for 2..32
 * concurrent requests, the F.linear call sites of a decode step (modeling/bagel/qwen2_navit.py:515-517,591-594;
 * modeling/qwen2/modeling_qwen2.py:54-59,200-201; modeling/bagel/bagel.py:978), serves the und marker rows of a denoise forward and the
 * 17..32 rows of a short text prefill (modeling/bagel/bagel.py:267-297).
 * A pure weight stream through the MFMA: the 8 waves of a persistent workgroup (one per CU) partition K and keep their activation
 * fragments in registers (one block of 16 rows, or two: every weight fragment then feeds two MFMAs), every weight byte is loaded once,
 * straight into the matrix-core operand layout.  K % 32 == 0, N % 16 == 0 (SwiGLU16: N % 32 == 0, no bias / residual).  Rows longer than
 * 4864 elements (M <= 16) resp. 3584 elements (M > 16: the second block of activation fragments takes the registers of the longest
 * instantiation) run as K slices over workgroups through a caller-owned fp32 workspace (bagel_gemv_mb_workspace_bytes: an upper bound for
 * every M; 16-byte aligned; only touched by this call) and a second small launch; such rows take neither the fused norm nor SwiGLU16
 * (BAGEL_ERR_UNSUPPORTED).  Deterministic: fixed summation order, no atomics; a row's result does not depend on M within one block
 * count (M <= 16 / M > 16) -- across the two the K partition may differ (fp32 summation order), like the K-split tiles of bagel_gemm_bf16. */
int bagel_gemv_mb_workspace_bytes(int32_t N, int32_t K, int64_t* bytes);
int bagel_gemv_mb_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, const void* bias, const void* R, int64_t ldr,
                       void* C, int64_t ldc, const void* norm_w, float eps, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                       void* workspace, int64_t workspace_bytes, bagel_stream_t stream);

/* Weight-only INT8 for the decode path (MI355X analogue of the reference's quantised inference modes, app.py:114-131,
 * bitsandbytes NF4 / LLM.int8).  An OPTION that changes results; the bf16 path is the default.
 * quantize: q[n,k] = round(W[n,k] / scale[n]) + 128 (u8), scale[n] = max_k |W[n,k]| / 127 (row-wise absmax). */
int bagel_quantize_rows_i8(const void* w, int64_t ldw, void* q, int64_t ldq, float* scale, int32_t rows, int32_t cols,
                           bagel_stream_t stream);
/* bagel_gemv_bf16 on u8 weights: y_n = scale[n] * (sum_k q[n,k] x_k - 128 sum_k x_k), fp32 accumulate, activations bf16,
 * same epilogues / fused RMSNorm / roundings.  K % 16 == 0, ldw in bytes % 16 == 0. */
int bagel_gemv_w8_bf16(const void* A, int64_t lda, const void* Wq, int64_t ldw, const float* scale, const void* bias,
                       const void* R, int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps, int32_t M,
                       int32_t N, int32_t K, int32_t epilogue, bagel_stream_t stream);

/* NF4 -- the 4-bit load mode the reference itself ships (app.py:114-125: bitsandbytes quant_type "nf4", blocksize 64, fp32 absmax, no
 * double quantisation, bf16 compute; restated in oracle/nf4.py from the library's published algorithm).  quantize: per block of 64
 * consecutive weights absmax[n, k/64] = max |w| (fp32), code = index of the NF4 code-book entry nearest to w * (1 / absmax) (the
 * library's midpoint decision tree), two codes per byte with the EVEN element in the HIGH nibble; cols % 64 == 0.  gemv: bagel_gemv_bf16
 * on the de-quantised weights code_book[code] * absmax ("W4A16": activations bf16, fp32 accumulation, same epilogues / fused RMSNorm /
 * roundings).  An OPTION that changes results (generate_text(weight_quant="nf4")); ldw in bytes % 16 == 0. */
int bagel_quantize_nf4(const void* w, int64_t ldw, void* q, int64_t ldq_bytes, float* absmax, int32_t rows, int32_t cols,
                       bagel_stream_t stream);

/* Whole-model 4-/8-bit load modes (app.py:114-131: bitsandbytes quantises EVERY nn.Linear of the language model): the engines keep the codes resident and
 * materialise one decoder layer's bf16 matrices at a time -- what bitsandbytes' matmul_4bit does in front of F.linear for more than one activation
 * row: w = bf16(code_book[code] * absmax[block]) (NF4) resp. bf16((q - 128) * scale[row]) (row-wise INT8); cols % 64 (NF4) / % 8 (INT8). */
int bagel_dequantize_nf4_bf16(const void* q, int64_t ldq_bytes, const float* absmax, void* out, int64_t ld_out, int32_t rows,
                              int32_t cols, bagel_stream_t stream);
int bagel_dequantize_rows_i8_bf16(const void* q, int64_t ldq, const float* scale, void* out, int64_t ld_out, int32_t rows, int32_t cols,
                                  bagel_stream_t stream);
int bagel_gemv_nf4_bf16(const void* A, int64_t lda, const void* Wq, int64_t ldw_bytes, const float* absmax, const void* bias,
                        const void* R, int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps, int32_t M, int32_t N,
                        int32_t K, int32_t epilogue, bagel_stream_t stream);

/* MXFP4 weight-only projections for the decode path: the 4-bit counterpart of the reference's bitsandbytes NF4 load mode
 * (app.py:114-125) on the chip's own block-scaled MFMA operand -- an OPTION that changes results.  OCP-MX FP4: codes E2M1, two per
 * byte (element 2i in the low nibble of byte i), one E8M0 scale byte per 32 elements along K (2^(b-127), b = max(exp(max|w|) - 2, 0)),
 * stored per row in groups of 16 bytes = 4 k-steps of 128 elements: byte 4q + j = block q of step j (oracle/mxfp4.py permute_scales).
 * cols % 128 == 0; ldq_bytes % 16 == 0 and >= cols/2; lds_bytes % 4 == 0 and >= 16 * ceil(cols / 512). */
int bagel_quantize_rows_mxfp4(const void* W, int64_t ldw, void* q, int64_t ldq_bytes, void* scales, int64_t lds_bytes, int32_t rows,
                              int32_t cols, bagel_stream_t stream);
/* C[M <= 4, N] = epilogue(A W^T) on those weights: the A rows are (optionally RMS-normalised, modeling_qwen2.py:54-59, and) quantised
 * to FP8 e4m3 with one scale per row inside the kernel, the product runs on v_mfma_scale_f32_16x16x128_f8f6f4, epilogues none / bias /
 * residual / SwiGLU16 (3) with the rounding points of bagel_gemm_bf16.  Replaces the F.linear sites of bagel_gemv_bf16 at Lq = 1. */
int bagel_gemv_w4_bf16(const void* A, int64_t lda, const void* Wq, int64_t ldq_bytes, const void* Ws, int64_t lds_bytes, const void* bias,
                       const void* R, int64_t ldr, void* C, int64_t ldc, const void* norm_w, float eps, int32_t M, int32_t N,
                       int32_t K, int32_t epilogue, bagel_stream_t stream);

/* FP8 (OCP e4m3) path for the gen-expert GEMMs of the denoise forward (the MI355X-native counterpart of the reference's quantised
 * load modes, app.py:114-131 -- an OPTION that changes results; bf16 is the default).
 * quantize: q[r,k] = e4m3_rne(x[r,k] / scale[r]), scale[r] = max_k |x[r,k]| / 448.  cols % 8 == 0; ldq in bytes. */
int bagel_quantize_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq_bytes, float* scale, int32_t rows, int32_t cols,
                            bagel_stream_t stream);
/* Qwen2RMSNorm (modeling_qwen2.py:54-59) with the FP8 quantiser fused behind it: q, scale = quantize_rows_fp8(rmsnorm(x, w)),
 * bit-identical to the two calls, 3 instead of 7 bytes of traffic per element. */
int bagel_rmsnorm_fp8(const void* x, int64_t ldx, const void* w, void* q, int64_t ldq_bytes, float* scale, int32_t rows,
                      int32_t cols, float eps, bagel_stream_t stream);
/* C[c_rows[i], :] = epilogue((sa[a_rows[i]] * sw[n]) * sum_k Aq[a_rows[i], k] * Wq[n, k]) on v_mfma_scale_f32_16x16x128_f8f6f4 with
 * fp32 accumulation; epilogues EPI_NONE (+bias | +residual) and EPI_SWIGLU16 with the roundings of bagel_gemm_bf16.
 * Replaces the *_moe_gen F.linear calls of qwen2_navit.py:529-536,593-594 and modeling_qwen2.py:200-201 when the model is built
 * with gen_weight_quant="fp8".  K % 128 == 0; leading dimensions of Aq / Wq in bytes. */
int bagel_gemm_fp8_bf16(const void* Aq, int64_t lda_bytes, const float* sa, const void* Wq, int64_t ldw_bytes, const float* sw,
                        const void* bias, const int32_t* a_rows, const int32_t* c_rows, int32_t M, const void* R, int64_t ldr,
                        void* C, int64_t ldc, int32_t N, int32_t K, int32_t epilogue, bagel_stream_t stream);
/* The FP8 gen expert's gate/up projection with the SwiGLU result written as e4m3 bytes -- no bf16 round trip and no quantiser pass in front of the down
 * projection (modeling_qwen2.py:200-201 on the gen expert, option gen_weight_quant = "fp8"): Cq[c_rows[i], n] = e4m3(clamp(swiglu(..)[i, n] / cs[c_rows[i]], +-448)),
 * n < N / 2, and cmax[c_rows[i]] = max(cmax[c_rows[i]], max_n |swiglu(..)[i, n]|) as fp32 bit patterns (atomicMax on uint32: non-negative floats order like
 * integers).  DELAYED scaling: the scale of a row is chosen BEFORE its values exist -- bagel_fp8_delayed_scales derives it from the row maxima the previous
 * denoise step collected (scale = margin * amax / 448; 1.0 where nothing was seen) and clears the maxima for the step that follows; the first step of a
 * request has no history and takes the exact path (bagel_gemm_fp8_bf16 + bagel_quantize_rows_fp8).  oracle/fp8.py restates the scheme. */
int bagel_gemm_fp8_swiglu_q8(const void* Aq, int64_t lda_bytes, const float* sa, const void* Wq, int64_t ldw_bytes, const float* sw,
                             const int32_t* a_rows, const int32_t* c_rows, int32_t M, void* Cq, int64_t ldcq_bytes, const float* cs,
                             void* cmax, int32_t N, int32_t K, bagel_stream_t stream);
int bagel_fp8_delayed_scales(void* amax, float* scale, const int32_t* rows, int32_t n, float margin, bagel_stream_t stream);

/* Paged KV cache (64-token pages; token j of sample b at pool row block_table[b*bt_stride + j/64]*64 + j%64).
 * Appends this step's K/V row of every sample at slot kv_len[b] (device memory) -- the in-place form of the
 * per-layer cache rebuild at qwen2_navit.py:563-575. */
int bagel_kv_append_paged_bf16(const void* k_new, const void* v_new, int64_t ld_new, void* kpool, void* vpool,
                               int64_t ld_pool, const int32_t* block_table, int32_t bt_stride,
                               const int32_t* kv_len, int32_t batch, int32_t width, bagel_stream_t stream);

/* Decode-step epilogue of the fused QKV projection (qwen2_navit.py:518-557,563-575 at Lq = 1): q_norm/k_norm + RoPE
 * with the und cast points of bagel_qknorm_rope_bf16, q rewritten in place, the finished K row and the V row stored
 * into their page slot kv_len[b] (pad lanes zeroed).  One launch for qknorm_rope + kv_append. */
int bagel_decode_qkv_post_bf16(void* qkv, int64_t ld, const void* cos_tab, const void* sin_tab, const void* q_w,
                               const void* k_w, void* kpool, void* vpool, int64_t ld_pool, const int32_t* block_table,
                               int32_t bt_stride, const int32_t* kv_len, int32_t batch, int32_t nq, int32_t nkv,
                               int32_t head_dim, int32_t head_dim_padded, float eps, int32_t use_norm,
                               bagel_stream_t stream);

/* flash_attn_varlen_func at Lq = 1 (qwen2_navit.py:579-588) over the paged cache: keys [0, kv_len[b] + len_add) of
 * sample b, split into 128-key chunks (grid sized for max_len), GQA heads share each K/V read; fp32 softmax.
 * part_o: batch*nq*ceil(max_len/128)*head_dim floats, part_ml: batch*nq*ceil(max_len/128)*2 floats (workspace). */
int bagel_attn_decode_paged_bf16(const void* q, int64_t ldq, const void* kpool, const void* vpool, int64_t ld_pool,
                                 const int32_t* block_table, int32_t bt_stride, const int32_t* kv_len,
                                 int32_t len_add, int32_t max_len, float* part_o, float* part_ml, void* out,
                                 int64_t ldo, int32_t batch, int32_t nq, int32_t nkv, int32_t head_dim,
                                 float softmax_scale, bagel_stream_t stream);

/* bagel_decode_qkv_post_bf16 + bagel_attn_decode_paged_bf16 in one launch (plus the combine): qkv = the RAW fused projection
 * rows; q/k norm + RoPE happen inside the attention workgroups, the new K/V row goes to page slot kv_len[b] and is attended
 * to (keys [0, kv_len[b]]).  Bit-identical to the two-kernel form. */
int bagel_attn_decode_fused_bf16(const void* qkv, int64_t ld, const void* cos_tab, const void* sin_tab, const void* q_w,
                                 const void* k_w, void* kpool, void* vpool, int64_t ld_pool, const int32_t* block_table,
                                 int32_t bt_stride, const int32_t* kv_len, int32_t max_len, float* part_o, float* part_ml,
                                 void* out, int64_t ldo, int32_t batch, int32_t nq, int32_t nkv, int32_t head_dim,
                                 int32_t head_dim_padded, float eps, int32_t use_norm, float softmax_scale,
                                 bagel_stream_t stream);

/* EXPERIMENTAL -- do NOT bind these four entry points in an integration: the persistent decode engine is a parity-tested
 * NEGATIVE result (94.4 us per layer against 83.8 us for the four gemv launches it replaces, profiles/r05_decode_engine.log), off by default
 * (opt-in BAGEL_DECODE_ENGINE=1), kept for the record; the signatures may change or disappear.
 * A CHAIN of 1..4 dependent batch-1 projections as ONE persistent launch (csrc/engine.hip): phase i computes what
 * bagel_gemv_bf16(A_i, W_i, bias_i, R_i, C_i, norm_w_i, eps, M = 1, N_i, K_i, epilogue_i) computes, bit for bit, with A_0 written by an
 * earlier launch and A_i == C_{i-1} for i > 0 -- for a decoder layer at Lq = 1 (bagel.py:930-1000): o_proj(+residual)
 * (qwen2_navit.py:591-594) -> post-attention RMSNorm + gate/up with SwiGLU -> down(+residual) (modeling_qwen2.py:200-201, 54-59) -> the
 * NEXT layer's input RMSNorm + qkv (qwen2_navit.py:515-517), or the final norm + lm_head (bagel.py:978).  One workgroup per CU: a loader
 * wave streams the workgroup's share of every phase's weight rows through an LDS ring with non-temporal LDS-DMA and never waits for an
 * activation (the next phase's weights arrive while the consumers hand over), the other waves run the lane-FMA body of the gemv kernel;
 * hand-offs between phases are write-through stores + one flag word per workgroup and phase.
 *   ptrs: HOST array [n_phases][6] = {A, W, bias | NULL, norm_w | NULL, R | NULL, C} (device pointers; R may alias C);
 *   dims: HOST array [n_phases][4] = {N, K, ldw, epilogue};
 *   sync_ws: bagel_decode_engine_sync_bytes(n_phases) bytes of device memory that are ZERO when the launch starts (the caller clears it,
 *            e.g. one memset per token over every layer's slice); status: 4 device uint32, written only on failure (a bounded spin
 *            gave up: code | workgroup << 8) -- the caller clears it once and checks it when it next synchronises.
 * The launch occupies EVERY CU (grid = bagel_decode_engine_workgroups()) and its workgroups wait for each other: nothing else may be
 * running on the device's other streams that could keep a CU from it for long.  K % 8 == 0, N even; a fused RMSNorm needs K <= 4096;
 * un-normalised rows of K >= 8192 take the 4-way K split of bagel_gemv_bf16 (the down projection). */
int bagel_decode_engine_workgroups(void);
int bagel_decode_engine_sync_bytes(int32_t n_phases);
int bagel_decode_engine_bf16(const void* const* ptrs, const int64_t* dims, int32_t n_phases, float eps, void* sync_ws, void* status,
                             bagel_stream_t stream);
/* Diagnostic form: the same launch, and every workgroup leaves its event times (100 MHz ticks of s_memrealtime) in
 * trace[workgroup][4 phases][16] uint64: 0 loader first issue, 1 last issue, 2 ticks blocked on a free ring slot, 3 ticks blocked in counted DMA
 * waits, 4 consumer 0 hand-off begin, 5 flags seen, 6 activation staged, 7..9 consumer c's last unit done, 10..12 ticks consumer c waited for
 * full slots, 13 the workgroup's flag store (tools/decode_engine_probe.py --trace). */
int bagel_decode_engine_traced_bf16(const void* const* ptrs, const int64_t* dims, int32_t n_phases, float eps, void* sync_ws, void* status,
                                    void* trace, bagel_stream_t stream);

/* Device-side bookkeeping of one decode step (bagel.py:984-994): cur_tok32 <- next_tok, tokens_out[step+1] <- next_tok,
 * pos += 1, kv_len += 1, step += 1.  Keeps the host out of the token loop so one captured step can be replayed. */
int bagel_decode_advance(const int64_t* next_tok, int32_t* cur_tok32, int64_t* tokens_out, int64_t* pos,
                         int32_t* kv_len, int32_t* step, int32_t batch, int32_t max_steps, bagel_stream_t stream);

/* hipGraph capture of a launch sequence on a (non-default) stream; replaces nothing in the reference (its decode loop
 * is ~5k eager launches + host syncs per token). */
int bagel_graph_begin(bagel_stream_t stream);
int bagel_graph_end(bagel_stream_t stream, void** exec_out);
int bagel_graph_launch(void* exec, bagel_stream_t stream);
int bagel_graph_destroy(void* exec);

/* ---- TaylorSeer step skipping (modeling/cache_utils/taylorseer.py; generate_image(enable_taylorseer=True)) ---- */
/* derivative_approximation (taylorseer.py:11-30) on the last decoder layer's output: factors[0] <- feature,
 * factors[i+1] <- bf16(bf16(new_i - old_i) / distance) for i < n_diff, in place over n_diff+1 [rows, cols] buffers. */
int bagel_taylor_update_bf16(const void* feature, int64_t ld_feature, void* const* factors, int32_t n_diff,
                             int32_t distance, int64_t rows, int32_t cols, bagel_stream_t stream);
/* taylor_formula (taylorseer.py:32-46): out = sum_{i<n} bf16(bf16(factors[i] / i!) * x^i), bf16 running sum. */
int bagel_taylor_eval_bf16(void* const* factors, int32_t n, int32_t x, void* out, int64_t ld_out, int64_t rows,
                           int32_t cols, bagel_stream_t stream);

/* ---- training forward glue (Bagel.forward, bagel.py:101-229) ---------------------------------------------------- */
/* out = bf16((1 - t[row]) * clean + t[row] * noise)  -- the noised latent of bagel.py:187, cast as autocast does at :190. */
int bagel_flow_mix_bf16(const float* clean, const float* noise, const float* t, void* out, int64_t n_rows,
                        int32_t cols, bagel_stream_t stream);
/* bagel_flow_add_bf16 with one timestep-embedding row per latent token group: seq[rows[i]] = bf16(bf16(seq[rows[i]] +
 * temb[temb_ids[i]]) + pos_table[pos_ids[i]])  (bagel.py:188-191). */
int bagel_flow_add_rows_bf16(void* seq, int64_t ld, const int32_t* rows, const void* temb, int64_t ld_temb,
                             const int32_t* temb_ids, const void* pos_table, int64_t ld_pos, const int64_t* pos_ids,
                             int32_t n, int32_t cols, bagel_stream_t stream);
/* out[i][c] = (pred[i][c] - (noise[src_rows[i]][c] - clean[src_rows[i]][c]))^2, fp32  (bagel.py:214-217). */
int bagel_mse_rows_f32(const void* pred, int64_t ld_pred, const float* noise, const float* clean,
                       const int32_t* src_rows, float* out, int64_t n_rows, int32_t cols, bagel_stream_t stream);
/* F.cross_entropy(logits.float(), labels, reduction="none") on bf16 logits (bagel.py:222). */
int bagel_cross_entropy_bf16(const void* logits, int64_t ld, const int64_t* labels, float* out, int32_t rows,
                             int32_t cols, bagel_stream_t stream);

/* ---- image pre/post-processing (data/transforms.py:15-115, inferencer.py:174-185) ---------------------------- */
/* One separable pass of Pillow's 8-bit bicubic/antialias resample (what torchvision's resize of a PIL image runs for
 * data/transforms.py:88).  bounds[o] = (first tap, tap count), kk[o][ksize] = 22-bit fixed-point weights, both computed
 * by the host as Resample.c's precompute_coeffs/normalize_coeffs_8bpc do.  vertical = 0: n_lines rows, out_len output
 * pixels x channels per row; vertical = 1: n_lines = bytes per row, out_len = output rows.  Strides in bytes. */
int bagel_resample_u8(const void* in, int64_t in_stride, void* out, int64_t out_stride, int32_t n_lines,
                      int32_t out_len, int32_t channels, const int32_t* bounds, const int32_t* kk, int32_t ksize,
                      int32_t vertical, bagel_stream_t stream);
/* ToTensor + Normalize (data/transforms.py:109-115): out[c][y][x] = ((in[y][x][c] / 255) - mean[c]) / std[c], fp32.
 * mean/std are HOST arrays of C floats (passed by value to the kernel). */
int bagel_u8_to_chw_f32(const void* in, int64_t in_stride, float* out, int32_t H, int32_t W, int32_t C,
                        const float* mean, const float* stdv, bagel_stream_t stream);
/* decode_image (inferencer.py:182-183): out[y][x][c] = uint8(trunc(clamp(in[c][y][x] * 0.5 + 0.5, 0, 1) * 255)). */
int bagel_chw_f32_to_u8(const float* in, int64_t chan_stride, int64_t row_stride, void* out, int64_t out_stride,
                        int32_t H, int32_t W, int32_t C, bagel_stream_t stream);

/* The same conversion for the bf16 decoder output of the inferencer's autocast region (inferencer.py:233 -> :182-183): every elementwise
 * op rounds to bf16 before the truncating uint8 cast.  Element strides of the source are free (an NHWC buffer viewed as CHW). */
int bagel_chw_bf16_to_u8(const void* in, int64_t chan_stride, int64_t row_stride, int64_t col_stride, void* out, int64_t out_stride,
                         int32_t H, int32_t W, int32_t C, bagel_stream_t stream);

/* ---- training backward (loss.backward() of train/pretrain_unified_navit.py:683-735 over Bagel.forward, bagel.py:101-229; the
 *      reference gets it from torch autograd, these are the hand-written reverse kernels the product chains) ---------------------- */
/* bagel_attn_varlen_ranges_bf16 that also leaves the row statistics the attention reverse needs: lse[h * ld_lse + row] = log2 of the
 * softmax denominator of (row, q head h) in the scaled base-2 domain (P = exp2(scale * log2(e) * s - lse)), fp32.  Every query row must
 * belong to exactly one range (the split decomposition of forward_train does that). */
int bagel_attn_varlen_ranges_lse_bf16(const void* q, int64_t ldq, const void* k_new, int64_t ldk_new, const void* vt_new, int64_t ldvt_new,
                                      const void* k_ctx, int64_t ldk_ctx, const void* vt_ctx, int64_t ldvt_ctx, void* out, int64_t ldo,
                                      const int32_t* q_start, const int32_t* q_end, const int32_t* ctx_start, const int32_t* ctx_end,
                                      const int32_t* vt_new_col, const int32_t* vt_ctx_col, int32_t batch, int32_t max_lq, int32_t nq,
                                      int32_t nkv, int32_t head_dim, int32_t causal, float softmax_scale, float* lse, int64_t ld_lse,
                                      bagel_stream_t stream);
/* dst[c][j] = src[src_rows ? src_rows[j] : j][c] for j < rows, 0 for rows <= j < rows_padded: the K-contiguous operand images of
 * the weight-gradient and input-gradient GEMMs (dW = dY^T X and dX = dY W are NT products of transposed images on bagel_gemm_bf16;
 * src_rows = one MoT expert's row list).  cols % 8 == 0, ld_dst >= rows_padded. */
int bagel_transpose_bf16(const void* src, int64_t ld_src, const int32_t* src_rows, int32_t rows, int32_t cols, void* dst,
                         int64_t ld_dst, int32_t rows_padded, bagel_stream_t stream);
/* Reverse of bagel_rmsnorm_bf16 (Qwen2RMSNorm, modeling_qwen2.py:54-59): with xh = x * rsqrt(mean(x^2) + eps), w = the row's expert
 * weight:  dx = rsqrt(..) * (dy w - xh * mean(dy w xh));  g = bf16((accumulate ? g : 0) + bf16(dx));  dw_e[c] = sum over the rows of
 * expert e of dy * bf16(xh).  fp32 arithmetic; dw0 / dw1 bf16 [cols] (dw1 NULL without a second expert); partial_ws fp32, at least
 * BAGEL_COLSUM_WS_FLOATS(rows, 2 * cols) + rows floats (the rows' rsqrt values are kept behind the partial sums). */
#define BAGEL_COLSUM_WS_FLOATS(rows, cols) ((((int64_t)(rows) + 63) / 64) * (int64_t)(cols))   /* one partial row per 64 input rows */
int bagel_rmsnorm_bwd_bf16(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* w0, const void* w1,
                           const int32_t* expert_of_row, void* g, int64_t ldg, int32_t accumulate, void* dw0, void* dw1,
                           float* partial_ws, int32_t rows, int32_t cols, float eps, bagel_stream_t stream);
/* Reverse of bagel_layernorm_bf16 (nn.LayerNorm of the SigLIP tower, siglip_navit.py:266-269,342): g = bf16((accumulate ? g : 0) +
 * bf16(dx)), dw[c] = sum dy * xh, db[c] = sum dy (bf16 [cols]); partial_ws fp32, BAGEL_COLSUM_WS_FLOATS(rows, 2 * cols) + 2 * rows floats. */
int bagel_layernorm_bwd_bf16(const void* x, int64_t ldx, const void* dy, int64_t lddy, const void* w, void* g, int64_t ldg,
                             int32_t accumulate, void* dw, void* db, float* partial_ws, int32_t rows, int32_t cols, float eps,
                             bagel_stream_t stream);
/* Reverse of bagel_qknorm_rope_bf16 in its training form (gen_mode 0; PackedAttentionMoT.forward_train, qwen2_navit.py:430-455): in
 * place on the gradient of the rotated [q | k | v] rows -> gradient of the raw projection (v columns untouched); qkv_raw = the
 * projection output the forward normalised.  dqw / dkw: bf16 [head_dim] per expert (NULL when use_norm == 0 or no second expert);
 * partial_ws fp32, at least BAGEL_COLSUM_WS_FLOATS(rows, 4 * head_dim) floats. */
int bagel_qknorm_rope_bwd_bf16(void* dqkv, int64_t ld, const void* qkv_raw, int64_t ld_raw, const void* cos_tab, const void* sin_tab,
                               const void* q_w0, const void* k_w0, const void* q_w1, const void* k_w1, const int32_t* expert_of_row,
                               void* dqw0, void* dkw0, void* dqw1, void* dkw1, float* partial_ws, int32_t rows, int32_t nq,
                               int32_t nkv, int32_t head_dim, int32_t head_dim_padded, float eps, int32_t use_norm,
                               bagel_stream_t stream);
/* Reverse of the SwiGLU16 epilogue (Qwen2MLP, modeling_qwen2.py:201): gu = the un-activated gate/up projection in the interleaved
 * [16 gate | 16 up] column layout, overwritten with its gradient: d_up = d_act * bf16(silu(g)), d_gate = d_act * u * silu'(g). */
int bagel_swiglu_bwd_bf16(void* gu, int64_t ld, const void* d_act, int64_t ld_d, int64_t rows, int32_t inter, bagel_stream_t stream);
/* The SwiGLU16 epilogue as a kernel of its own, act = bf16(bf16(silu(g)) * u) from the stored bf16 projection: bit-identical to the fused
 * epilogue; used when the training tape keeps the un-activated gate/up projection instead of recomputing it in the backward. */
int bagel_swiglu_fwd_bf16(const void* gu, int64_t ld, void* act, int64_t ld_act, int64_t rows, int32_t inter, bagel_stream_t stream);
/* Reverse of the GELU-tanh (kind 1) / SiLU (kind 2) epilogues: pre = the un-activated projection, overwritten with d_out * act'(pre). */
int bagel_act_bwd_bf16(void* pre, int64_t ld, const void* d_out, int64_t ld_d, int64_t rows, int32_t cols, int32_t kind,
                       bagel_stream_t stream);
/* Reverse of bagel_cross_entropy_bf16: logits <- bf16((softmax(logits) - onehot(label)) * d_loss[row]) (zero row for an ignored label). */
int bagel_cross_entropy_bwd_bf16(void* logits, int64_t ld, const int64_t* labels, const float* d_loss, int32_t rows, int32_t cols,
                                 bagel_stream_t stream);
/* Reverse of bagel_mse_rows_f32: d_pred[i][c] = bf16(2 (pred - (noise - clean)[src_rows[i]]) d_loss[i][c]). */
int bagel_mse_rows_bwd_bf16(const void* pred, int64_t ld_pred, const float* noise, const float* clean, const int32_t* src_rows,
                            const float* d_loss, void* d_pred, int64_t ld_d, int64_t n_rows, int32_t cols, bagel_stream_t stream);
/* dst[dst_rows[s]] = bf16(sum_{i in [seg_off[s], seg_off[s+1])} src[order[i]]) in fp32, deterministic: the gradient of an embedding
 * gather (nn.Embedding rows that several tokens share, bagel.py:148; the per-image timestep embedding, bagel.py:188). */
int bagel_rows_segment_sum_bf16(const void* src, int64_t ld_src, const int32_t* order, const int32_t* seg_off, const int32_t* dst_rows,
                                void* dst, int64_t ld_dst, int32_t n_seg, int32_t cols, bagel_stream_t stream);
/* out[c] = bf16(sum_j src[rows ? rows[j] : j][c]) (bias gradients); partial_ws fp32, BAGEL_COLSUM_WS_FLOATS(n_rows, cols) floats. */
int bagel_colsum_bf16(const void* src, int64_t ld, const int32_t* rows, int32_t n_rows, int32_t cols, float* partial_ws, void* out,
                      bagel_stream_t stream);
/* Reverse of the block-masked packed attention of forward_train (the causal / full / noise split mask of data/data_utils.py:72-103
 * that bagel_attn_varlen_ranges_bf16 runs as per-split sequences): dq, dk, dv from q, k, v (rotated, [rows, heads * D]), the forward
 * output o and its gradient d_o; qt / dot / kt = bagel_transpose_bf16 images of q / d_o / k ([heads * D, ld_t]).
 * Two deterministic kernels, no atomics: (1) per 128-query item and q head: row log-sum-exp and delta = rowsum(d_o * o), then
 * dQ = scale * sum_keys dS K;  (2) per 128-key item and kv head, over the group's q heads: dV = P^T dO, dK = scale * dS^T Q.
 * q_items [n_q_items][8] = {row0, nrows, sample_start, split_start, split_end, causal, first 64-key tile, end tile};
 * k_items [n_k_items][8] = {key0, nkeys, first visible query row, end visible query row, split_end, causal, 0, 0};
 * noise_bits[t] bit j = key 64 t + j belongs to a noise split (hidden from every later split);
 * lse_delta: fp32 workspace [2][nq][rows]; with lse_from_forward != 0 its first half already holds the rows' log2 softmax denominators
 * written by bagel_attn_varlen_ranges_lse_bf16 and the first kernel skips its statistics pass.  D = 64 or 128. */
int bagel_attn_bwd_blockmask_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                                  int64_t ldo, const void* d_o, int64_t lddo, const void* qt, const void* dot, const void* kt,
                                  int64_t ld_t, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv,
                                  const int32_t* q_items, int32_t n_q_items, const int32_t* k_items, int32_t n_k_items,
                                  const uint64_t* noise_bits, float* lse_delta, int32_t lse_from_forward, int32_t rows, int32_t nq,
                                  int32_t nkv, int32_t head_dim, float softmax_scale, bagel_stream_t stream);

/* ---- VAE (fp32, NHWC) ------------------------------------------------------------------------------------- */
/* Implicit-GEMM convolution / plain GEMM on the exact-fp32 MFMA.  mode 0: out[M,Cout] = in[M,Cin] w[Cout,Cin]^T
 * (1x1 conv, attention products; M = B*Hout*Wout); 1: 3x3 stride 1 pad 1; 2: 3x3 stride 2 with the (0,1,0,1) pad of
 * autoencoder.py:104-107; 3: nearest-2x upsample fused with the following 3x3 conv (autoencoder.py:116-118).
 * w is [Cout, taps*Cin] (tap-major), +bias, +residual (ResnetBlock / AttnBlock skip adds, autoencoder.py:65,95).
 * Replaces F.conv2d at autoencoder.py:76-78,102,114,139,170,221,248 and the SDPA matmuls at :60. */
int bagel_conv_gemm_f32(const float* in, int64_t ld_in, const float* w, int64_t ld_w, const float* bias,
                        const float* residual, float* out, int64_t ld_out, int32_t B, int32_t Hin, int32_t Win,
                        int32_t Cin, int32_t Hout, int32_t Wout, int32_t Cout, int32_t mode, bagel_stream_t stream);

/* GroupNorm(groups, eps, affine) (+swish) over NHWC fp32 (autoencoder.py:43,75,77,169,247; swish :34-35).
 * partial_ws: B*groups*(64*2 + 2) floats. */
int bagel_groupnorm_f32(const float* x, float* y, float* partial_ws, const float* gamma, const float* beta, int32_t B,
                        int32_t HW, int32_t C, int32_t groups, float eps, int32_t swish, bagel_stream_t stream);

/* x = softmax(scale * x) over rows, in place (single-head AttnBlock, autoencoder.py:60). */
int bagel_softmax_rows_f32(float* x, int64_t ld, int32_t rows, int32_t cols, float scale, bagel_stream_t stream);

/* The VAE under torch.autocast("cuda", bfloat16) -- how the reference's InterleaveInferencer runs it (inferencer.py:233 -> decode_image
 * :174-185; the VAE-encode of an edit request, modeling/bagel/bagel.py:491-550): conv2d and the AttnBlock's attention in bf16 (inputs,
 * weights and bias cast to bf16, fp32 accumulation, bf16 result), group_norm in fp32 on the bf16 input, residual adds in bf16
 * (oracle/bagel_oracle.py VAE_AUTOCAST = "cuda"; tests/golden/vae_full_bf16.pt).
 * bagel_conv_gemm_bf16: NHWC bf16 implicit-GEMM convolution / plain GEMM on mfma_f32_16x16x32_bf16, modes as bagel_conv_gemm_f32 (0 rows,
 * 1 conv3 s1 p1, 2 conv3 s2 pad(0,1,0,1), 3 nearest-2x upsample + conv3); out = bf16(acc + bias) [then bf16(that + residual)], or with
 * out_f32 != 0 the raw fp32 accumulators (attention scores; no bias / residual).  Cin % 8 == 0 (mode 0: K = Cin).
 * bagel_groupnorm_bf16: GroupNorm (+ swish) of a bf16 NHWC tensor, fp32 statistics and arithmetic, ONE rounding to bf16 on the way out;
 *                       partial_ws: fp32 workspace of B * groups * 2050 + B * C * 2 elements.
 * bagel_softmax_rows_bf16: y = bf16(softmax(scale * x)) per fp32 score row. */
int bagel_conv_gemm_bf16(const void* in, int64_t ld_in, const void* w, int64_t ld_w, const void* bias, const void* residual,
                         void* out, int64_t ld_out, int32_t out_f32, int32_t B, int32_t Hin, int32_t Win, int32_t Cin, int32_t Hout,
                         int32_t Wout, int32_t Cout, int32_t mode, bagel_stream_t stream);
int bagel_groupnorm_bf16(const void* x, void* y, float* partial_ws, const float* gamma, const float* beta, int32_t B, int32_t HW,
                         int32_t C, int32_t groups, float eps, int32_t swish, bagel_stream_t stream);
int bagel_softmax_rows_bf16(const float* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, float scale, bagel_stream_t stream);
/* z = bf16(scale * bf16(bf16(mean + bf16(std * noise)) - shift)), std = bf16(exp(bf16(0.5 * logvar))): DiagonalGaussian.sample + the
 * latent scale / shift on bf16 moments [n_pix, ld_moments >= 2 zc] (mean | logvar) with eager-bf16 rounding points (autoencoder.py:280-287,
 * 315-318 under autocast); noise [n_pix, zc] bf16 or NULL (mode of the distribution). */
int bagel_vae_reparam_bf16(const void* moments, int64_t ld_moments, const void* noise, void* z, int64_t n_pix, int32_t z_channels,
                           float scale, float shift, bagel_stream_t stream);

/* z = scale * ((mean + exp(0.5 logvar) * noise) - shift)  (autoencoder.py:280-287,315-318); moments [n_pix, 2*zc]. */
int bagel_vae_reparam_f32(const float* moments, const float* noise, float* z, int64_t n_pix, int32_t z_channels,
                          float scale, float shift, bagel_stream_t stream);

/* out = z / scale + shift (autoencoder.py:321). */
int bagel_vae_unscale_f32(const float* z, float* out, int64_t n, float scale, float shift, bagel_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
