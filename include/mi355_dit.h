/*
 * mi355_dit.h -- C ABI of the MI355X (gfx950) kernels behind SGLang-Diffusion's
 * per-denoise-step DiT hot path.
 *
 * Conventions (all entry points):
 *   - plain pointers to DEVICE memory + sizes/strides; no torch types.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it
 *     and never synchronise (the reference ops enqueue on the current stream,
 *     sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:251).
 *   - return 0 on success, a negative MI355_E* code otherwise; the text of the
 *     last failure on the calling thread is at mi355_last_error().  The torch
 *     binding turns a non-zero return into RuntimeError, mirroring TORCH_CHECK
 *     (sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1457-1487).
 *   - caller owns every buffer; outputs are caller-allocated.
 *   - strides ("ld*") are in ELEMENTS of the tensor they describe.
 *   - fp8 means OCP float8_e4m3fn (max 448).  bf16 is the only 16-bit float.
 */
#ifndef MI355_DIT_H
#define MI355_DIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK 0
#define MI355_EINVAL (-1)   /* bad argument / unsupported shape or alignment */
#define MI355_EHIP (-2)     /* HIP launch failure                            */
#define MI355_ENOSUP (-3)   /* feature the reference op has, not built here  */

/* activation fused in a linear epilogue */
#define MI355_ACT_NONE 0
#define MI355_ACT_GELU_TANH 1

const char* mi355_last_error(void);
/* 2: mi355_linear_args gained f16_begin/f16_end; mi355_flash_attn_fwd_f16v and
 *    mi355_bf16_to_f16 were added.
 * 3: mi355_linear_args gained block_scale and the four scale strides;
 *    mi355_layernorm_modulate_group_quant and mi355_merge_attn_states were added.
 * 4: mi355_flash_attn_fwd_ex was added; mi355_qk_norm_rope accepts q == NULL.
 * 5: mi355_cfg_normalize and mi355_guidance_rescale were added.
 * 6: mi355_scale_residual_layernorm_modulate, mi355_layernorm_affine and mi355_token_mean were
 *    added.
 * 7: mi355_f16_range_status was added (every bf16 -> half conversion is range-checked).
 * 8: mi355_guidance_rescale_static was added.
 * 9: mi355_linear_args gained w_shuffled; mi355_linear_grouped, mi355_shuffle_weight and
 *    mi355_shuffled_weight_bytes were added.
 * 10: mi355_linear_args gained out_f16.
 * 11: mi355_linear_args gained kn_weight, kn_cos, kn_sin, kn_eps, kn_seq, kn_begin, kn_end.
 * 12: the entry points and fields of measured-slower variants were withdrawn: mi355_linear_args
 *     lost w_shuffled and the seven kn_* fields (out_f16 is now its last field);
 *     mi355_linear_grouped, mi355_shuffle_weight and mi355_shuffled_weight_bytes are gone.
 *     mi355_reload_env and mi355_f16_range_snapshot were added. */
int mi355_abi_version(void);

/* The library reads its few MI355_* environment switches ONCE, when it is loaded (today:
 * MI355_ATTN_ASMV=0 selects the compiler-scheduled LDS reads of the attention kernel instead of
 * the hand-counted inline-asm ones; same results bit for bit).  This call reads them again, for
 * tests that compare the two forms inside one process.  No counterpart in the reference. */
void mi355_reload_env(void);

/* ---------------------------------------------------------------- quant --
 * sgl_per_token_quant_fp8(Tensor input, Tensor! output_q, Tensor! output_s)
 *   replaces sgl-kernel/csrc/gemm/per_token_quant_fp8.cu:242-314
 *   (schema sgl-kernel/csrc/common_extension.cc:157).
 * x bf16 [M,K] (row stride ldx), q fp8 [M,K] contiguous, s fp32 [M].
 * s = absmax/448 ; q = sat(x * (1/s)), (1/s := 0 when s == 0).
 * K % 8 == 0. */
int mi355_per_token_quant_fp8(const void* x, int64_t M, int64_t K, int64_t ldx,
                              void* q, float* s, void* stream);

/* sgl_per_token_group_quant_8bit(input, output_q!, output_s!, group_size, eps,
 *                                fp8_min, fp8_max, scale_ue8m0)
 *   replaces sgl-kernel/csrc/gemm/per_token_group_quant_8bit.cu:116-217
 *   (schema common_extension.cc:145).
 * s[m,g] = max(absmax_g, eps)/fp8_max (rounded up to a power of two when
 * scale_ue8m0), q = clamp(x / s).  Scale element (m,g) is written at
 * s[m*s_stride_m + g*s_stride_g] so row-major ([M,G]: G,1) and column-major
 * ([G,M] storage: 1,M) layouts both work.  K % group_size == 0,
 * group_size % 8 == 0. */
int mi355_per_token_group_quant_fp8(const void* x, int64_t M, int64_t K, int64_t ldx,
                                    void* q, float* s, int64_t s_stride_m,
                                    int64_t s_stride_g, int group_size, float eps,
                                    float fp8_min, float fp8_max, int scale_ue8m0,
                                    void* stream);

/* sgl_per_tensor_quant_fp8(input, output_q!, output_s!, is_static)
 *   replaces sgl-kernel/csrc/gemm/per_tensor_quant_fp8.cu (schema :154).
 * is_static == 0: s[0] must be zeroed by the caller (as the reference wrapper
 * does); the call computes s[0] = absmax/448 then q = sat(x / s[0]). */
int mi355_per_tensor_quant_fp8(const void* x, int64_t numel, void* q, float* s,
                               int is_static, void* stream);

/* ----------------------------------------------------------------- GEMM --
 * fp8_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype, bias?) -> Tensor
 *   replaces sgl-kernel/csrc/gemm/fp8_gemm_kernel.cu:1450-1532
 *   (schema common_extension.cc:135).
 * out[M,N] = bf16( (A[M,K] . B[K,N]) * sa[m] * sb[n] + bias[n] ), fp32 accumulate.
 * a: fp8 row-major, row stride lda.  b: fp8 COLUMN-major [K,N], i.e. the
 * weight W[N,K] with row stride ldb.  sa fp32 [M], sb fp32 [N], bias bf16 [N]
 * or NULL.  K % 128 == 0, lda/ldb % 16 == 0, N % 8 == 0, ldo % 8 == 0.
 * out_dtype: this entry point writes bf16 (the DiT path).  The reference op also takes
 * float16 (sgl-kernel/tests/test_fp8_gemm.py:17-35 runs both): that is mi355_linear_ex with
 * fp8 = 1 and out_f16 = 1 (same operands, half bias), which the registry wrapper calls for
 * out_dtype == float16; other dtypes raise, as in the reference (fp8_gemm_kernel.cu:1477). */
int mi355_fp8_scaled_mm(const void* a, const void* b, const float* sa, const float* sb,
                        const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                        int64_t lda, int64_t ldb, int64_t ldo, void* stream);

/* fp8_blockwise_scaled_mm(mat_a, mat_b, scales_a, scales_b, out_dtype) -> Tensor
 *   replaces sgl-kernel/csrc/gemm/fp8_blockwise_gemm_kernel.cu:368-465
 *   (schema common_extension.cc:140).
 * acc[m,n] += dot_k128(a,b) * sa[m,kb] * sb[kb,nb] per 128-wide K block.
 * Scale element strides are explicit: sa(m,kb) at sa[m*sa_sm + kb*sa_sk],
 * sb(kb,nb) at sb[kb*sb_sk + nb*sb_sn] (nb = n/128). */
int mi355_fp8_blockwise_scaled_mm(const void* a, const void* b, const float* sa,
                                  const float* sb, void* out, int64_t M, int64_t N,
                                  int64_t K, int64_t lda, int64_t ldb, int64_t ldo,
                                  int64_t sa_sm, int64_t sa_sk, int64_t sb_sk,
                                  int64_t sb_sn, void* stream);

/* y = act(x W^T + b) in bf16 with fp32 accumulate: the unquantised linear,
 *   replaces UnquantizedLinearMethod.apply -> F.linear
 *   (python/sglang/multimodal_gen/runtime/layers/linear.py:151-159) and, with
 *   act = GELU_TANH, diffusers' GELU(approximate="tanh") that follows it in
 *   FeedForward (python/sglang/multimodal_gen/runtime/layers/mlp.py:94-113):
 *   the linear result is rounded to bf16 before the activation, as in eager.
 * x bf16 [M,K] (ldx), w bf16 [N,K] (ldw), bias bf16 [N] or NULL, out bf16 (ldo).
 * K % 64 == 0 on the MFMA path (M > 16); any K % 8 == 0 when M <= 16. */
int mi355_linear_bf16(const void* x, const void* w, const void* bias, void* out,
                      int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                      int64_t ldo, int act, void* stream);

/* Row remapping for the batched joint-sequence layout [B, S_total, C]: logical
 * row m of a [B*rows_per_batch, C] operand lives at physical row
 *   (m / rows_per_batch) * batch_rows + row_offset + (m % rows_per_batch).
 * rows_per_batch == 0 means identity.  Lets the q/k/v projections write
 * straight into the concatenated text|image buffer that
 * FluxAttention.forward builds with torch.cat (models/dits/flux.py:379-382)
 * and lets the output projections read their slice of the attention output
 * (flux.py:401-413) without a copy. */
typedef struct {
  int64_t rows_per_batch;
  int64_t batch_rows;
  int64_t row_offset;
} mi355_rowmap;

/* Same maths as mi355_fp8_scaled_mm / mi355_linear_bf16, plus row maps on the
 * activation operand and on the output, and an optional fused
 *   out = residual + gate[b,:] * y      (gate bf16 [B, ldg], b = m / gate_rows)
 * epilogue reproducing "gate.unsqueeze(1) * y; hidden + ..."
 * (models/dits/flux.py:651-652,547-550) with its two bf16 roundings.
 * residual may alias out.  fp8 != 0 selects the W8A8 path (a fp8, sa/sb given). */
typedef struct {
  const void* a;        /* activations: fp8 (fp8=1) or bf16 */
  const void* w;        /* weight [N,K]: fp8 or bf16 */
  const float* sa;      /* per-token scales (fp8 only), indexed by the MAPPED activation row */
  const float* sb;      /* [N] per-channel scales (fp8 only) */
  const void* bias;     /* bf16 [N] or NULL */
  void* out;            /* bf16 */
  const void* residual; /* bf16, same layout as out, or NULL */
  const void* gate;     /* bf16 [B, ldg] or NULL */
  int64_t M, N, K;
  int64_t lda, ldw, ldo, ldg;
  int64_t gate_rows;    /* logical rows per gate row (tokens per sample) */
  mi355_rowmap a_map, o_map;
  int fp8;
  int act;
  /* output columns [f16_begin, f16_end) are stored as IEEE half instead of bf16: the value is
   * first rounded to bf16 (the linear's result, as always) and that bf16 is then written in
   * half format, which is exact for 2^-14 <= |y| <= 65504 (larger saturates, smaller goes
   * subnormal).  Values outside that range never pass silently: see mi355_f16_range_status.
   * Used for the V third of a fused q|k|v projection that feeds mi355_flash_attn_fwd_f16v.
   * Both multiples of 32; begin == end disables. */
  int64_t f16_begin, f16_end;
  /* block_scale != 0 (with fp8 != 0): the arithmetic of mi355_fp8_blockwise_scaled_mm instead
   * of per-token x per-channel scales -- sa is [M, K/128] (strides sa_sm, sa_sk; rows indexed
   * by the MAPPED activation row), sb is [N/128, K/128] (strides sb_sn, sb_sk), the product is
   * rounded to bf16 BEFORE the bias is added (fp8_utils.py:546-548), and everything after
   * that (activation, gate/residual, row maps, half columns) works as in the other modes. */
  int block_scale;
  int64_t sa_sm, sa_sk, sb_sk, sb_sn;
  /* Nonzero: out (and bias) are IEEE half instead of bf16 -- out_dtype = float16 of
   * fp8_scaled_mm / fp8_blockwise_scaled_mm (the reference's kernel test runs both dtypes,
   * sgl-kernel/tests/test_fp8_gemm.py:17-35).  fp32 accumulator x scales (+ bias) rounded once to
   * half (block scales: product rounded to half, then the half bias added).  fp8 modes only, no
   * activation / gate / residual / half-column range; the DiT itself is bf16 and never sets it. */
  int out_f16;
} mi355_linear_args;
int mi355_linear_ex(const mi355_linear_args* args, void* stream);

/* ------------------------------------------------------------ attention --
 * Dense non-causal (or causal) softmax(Q K^T * scale) V, bf16 in/out, fp32
 * softmax and accumulation; what FlashAttentionImpl/AITerImpl/SDPAImpl.forward
 * compute for the DiT
 *   (python/sglang/multimodal_gen/runtime/layers/attention/backends/
 *    flash_attn.py:436-565, aiter.py:63-93, sdpa.py:55-78; op schema
 *    sgl-kernel/csrc/flash_extension.cc:25-63).
 * q [B,Sq,H,D], k/v [B,Sk,Hk,D], out [B,Sq,H,D]; last dim contiguous, the
 * other strides explicit (elements).  lse fp32 [B,H,Sq] or NULL
 * (natural-log-sum-exp of the scaled scores, as FA returns it).
 * D == 128 (Flux/HunyuanVideo head size) or D == 64; H % Hk == 0. */
int mi355_flash_attn_fwd(const void* q, const void* k, const void* v, void* out,
                         float* lse, int64_t B, int64_t Sq, int64_t Sk, int64_t H,
                         int64_t Hk, int64_t D, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                         int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb,
                         int64_t v_ss, int64_t v_sh, int64_t o_sb, int64_t o_ss,
                         int64_t o_sh, float softmax_scale, int causal, void* stream);

/* Same computation with V given in IEEE half and the probabilities rounded to half (11
 * significant bits) instead of bf16 (8) before the P.V product: the result is closer to the
 * fp32-softmax reference (SDPAImpl on CPU) than the bf16-P form every flash kernel uses.  q, k
 * and out stay bf16.  v comes from mi355_bf16_to_f16 or from mi355_linear_ex's f16 columns. */
int mi355_flash_attn_fwd_f16v(const void* q, const void* k, const void* v, void* out,
                              float* lse, int64_t B, int64_t Sq, int64_t Sk, int64_t H,
                              int64_t Hk, int64_t D, int64_t q_sb, int64_t q_ss, int64_t q_sh,
                              int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t v_sb,
                              int64_t v_ss, int64_t v_sh, int64_t o_sb, int64_t o_ss,
                              int64_t o_sh, float softmax_scale, int causal, void* stream);

/* y[r][c] = half(x[r][c]) for a bf16 matrix with row strides ldx / ldy (elements);
 * cols % 8 == 0.  Exact inside half's normal range, saturating at +-65504 (counted, see
 * mi355_f16_range_status). */
int mi355_bf16_to_f16(const void* x, void* y, int64_t rows, int64_t cols, int64_t ldx,
                      int64_t ldy, void* stream);

/* The value bound of the half-format V operand, and how it is enforced.  No counterpart in the
 * reference (its attention backends keep V in bf16); it exists because the P.V product here
 * runs in half to reach the reference's eager result within 1e-3.
 * Half holds every bf16 value with 2^-14 <= |v| <= 65504 exactly.  Each conversion (the half
 * columns of mi355_linear_ex, mi355_bf16_to_f16) looks at the largest magnitude of every block
 * it converts (32 rows x 32 channels in the GEMM epilogue, 128 consecutive values in
 * mi355_bf16_to_f16) and counts, per device,
 *   overflow_blocks: blocks holding a value beyond +-65504 (stored saturated: WRONG results);
 *   tiny_blocks:     blocks whose largest magnitude is nonzero but below 2^-13, i.e. that lie
 *                    entirely in half's subnormal range and lose relative precision (single
 *                    small values among normal ones are harmless: their absolute error is
 *                    below 2^-25).
 * This call copies the current device's two counters to the host after the work queued on
 * `stream` has finished (it SYNCHRONISES the stream; do not call it under graph capture) and
 * zeroes them when reset != 0.  With both output pointers NULL and reset != 0 it only queues
 * the zeroing on `stream` and does not synchronise.  A caller that sees a non-zero count must redo the work with
 * bf16 V (f16_begin == f16_end, mi355_flash_attn_fwd): the denoise stage does exactly that. */
int mi355_f16_range_status(int64_t* overflow_blocks, int64_t* tiny_blocks, int reset,
                           void* stream);

/* The same two counters without a synchronisation: a copy of {overflow_blocks, tiny_blocks} into
 * host_counts[0..1] (PINNED host memory that the caller keeps alive until the copy has run) is
 * queued on `stream` behind the work already there, followed by the zeroing when reset != 0.
 * The caller reads host_counts after an event it records behind this call -- or, when the call
 * was captured into a HIP graph with the work it guards, after a replay of that graph has
 * finished.  No counterpart in the reference (see mi355_f16_range_status). */
int mi355_f16_range_snapshot(int64_t* host_counts, int reset, void* stream);

/* ------------------------------------------------------- norms and rope --
 * rmsnorm(Tensor! output, Tensor input, Tensor weight, float eps, bool pdl)
 *   (schema common_extension.cc:66; semantics RMSNorm.forward_native,
 *   python/sglang/multimodal_gen/runtime/layers/layernorm.py:106-143):
 *   out = bf16( x.float() * rsqrt(mean(x^2)+eps) * w ), w fp32 or bf16. */
int mi355_rmsnorm(void* out, const void* x, const void* w, int w_is_f32, int64_t rows,
                  int64_t dim, int64_t ldx, int64_t ldo, float eps, void* stream);

/* fused_add_rmsnorm(Tensor! input, Tensor! residual, Tensor weight, eps, pdl)
 *   (schema common_extension.cc:69; layernorm.py:115-117):
 *   residual = bf16(x + residual); x = rmsnorm(residual). */
int mi355_fused_add_rmsnorm(void* x, void* residual, const void* w, int w_is_f32,
                            int64_t rows, int64_t dim, float eps, void* stream);

/* In-place per-head RMSNorm of q and k followed by interleaved (GPT-J style,
 * is_neox=False) or neox RoPE, one pass over q and k:
 *   apply_qk_norm (layernorm.py:531-568; fused_inplace_qknorm,
 *   python/sglang/jit_kernel/norm.py:66-90) then
 *   apply_flashinfer_rope_qk_inplace (layers/rotary_embedding/utils.py:42-121;
 *   rotation formula python/sglang/jit_kernel/diffusion/triton/rotary.py:43-64).
 * q,k [B,S,H,D] with strides (sb, ss, sh), D contiguous, D == 128 or 64.
 * Tokens s < s_split use weights (wq0,wk0) (the text stream's
 * norm_added_q/k), the rest (wq1,wk1); pass the same pointers for a single
 * stream.  Weights fp32 [D] or NULL to skip the norm.  cos,sin fp32 [S, D/2]
 * or NULL to skip the rotation; position of token s is s.  q may be NULL: only k is processed
 * then (mi355_flash_attn_fwd_ex can do the query side while it loads Q). */
int mi355_qk_norm_rope(void* q, void* k, int64_t B, int64_t S, int64_t H, int64_t D,
                       int64_t sb, int64_t ss, int64_t sh, const float* wq0,
                       const float* wk0, const float* wq1, const float* wk1,
                       int64_t s_split, float eps, const float* cos, const float* sin,
                       int interleaved, void* stream);

/* rotary_embedding(positions, query!, key!?, head_size, cos_sin_cache, is_neox)
 *   (schema common_extension.cc:94; python wrapper
 *   sgl-kernel/python/sgl_kernel/elementwise.py:270-366): in-place RoPE from a
 *   [max_pos, rot_dim] cache holding cos | sin halves; cache fp32 or bf16. */
int mi355_rotary_embedding(const int64_t* positions, void* q, void* k, int64_t tokens,
                           int64_t Hq, int64_t Hk, int64_t head_size, int64_t rot_dim,
                           int64_t q_stride, int64_t k_stride, const void* cos_sin_cache,
                           int cache_is_f32, int is_neox, void* stream);

/* LayerNorm(no affine, eps) then x*(1+scale)+shift, the body of diffusers'
 * AdaLayerNormZero / ZeroSingle / Continuous and of norm2 + modulation
 *   (models/dits/flux.py:628-634,653-661,511,912), each eager bf16 rounding
 *   kept: ln -> bf16, (1+scale) -> bf16, product -> bf16, sum -> bf16.
 * x [B*T, dim] (ldx); shift,scale bf16 rows [B] with stride ldmod; row r uses
 * batch r / T.  Row maps as for the GEMM. */
int mi355_layernorm_modulate(void* out, const void* x, const void* shift,
                             const void* scale, int64_t B, int64_t T, int64_t dim,
                             int64_t ldx, int64_t ldo, int64_t ldmod, float eps,
                             const mi355_rowmap* x_map, const mi355_rowmap* o_map,
                             void* stream);

/* ScaleResidualLayerNormScaleShift (layers/layernorm.py:332-443, its native form):
 *   res_out = bf16(residual + bf16(x * gate[b]))   (gate NULL: res_out = bf16(residual + x))
 *   out     = the mi355_layernorm_modulate arithmetic on res_out
 * in one pass over the rows: x, residual, res_out, out are [B*T, dim] with their own row
 * strides; gate, shift, scale bf16 rows per sample (strides ldg, ldmod).  res_out may alias
 * residual or x. */
int mi355_scale_residual_layernorm_modulate(
    void* out, void* res_out, const void* residual, const void* x, const void* gate,
    const void* shift, const void* scale, int64_t B, int64_t T, int64_t dim, int64_t ldx,
    int64_t ldr, int64_t ldo, int64_t ldro, int64_t ldg, int64_t ldmod, float eps, void* stream);

/* torch.nn.LayerNorm with elementwise affine parameters over the last dimension (the token
 * refiner of the video DiT, models/dits/hunyuanvideo.py:880-905): fp32 statistics,
 * out = bf16((x - mean) * rstd * weight + bias), one rounding.  weight, bias bf16 [dim]. */
int mi355_layernorm_affine(void* out, const void* x, const void* weight, const void* bias,
                           int64_t rows, int64_t dim, int64_t ldx, int64_t ldo, float eps,
                           void* stream);

/* out[b, :] = bf16(mean over tokens of x[b, t, :]) (torch.mean(x, dim=1),
 * models/dits/hunyuanvideo.py:849); x [B, T, dim] with strides (x_batch_stride, ldx, 1),
 * out [B, dim] dense. */
int mi355_token_mean(void* out, const void* x, int64_t B, int64_t T, int64_t dim, int64_t ldx,
                     int64_t x_batch_stride, void* stream);

/* Same, then dynamic per-token fp8 quantisation of the bf16 result
 * (mi355_per_token_quant_fp8 semantics): feeds the fp8 linears directly. */
int mi355_layernorm_modulate_quant(void* q, float* s, const void* x, const void* shift,
                                   const void* scale, int64_t B, int64_t T, int64_t dim,
                                   int64_t ldx, int64_t ldmod, float eps,
                                   const mi355_rowmap* x_map, void* stream);

/* Same, then dynamic quantisation per (token, 128-channel group)
 * (mi355_per_token_group_quant_fp8 semantics with group_size 128, eps 1e-10, +-448):
 * feeds the block-scaled fp8 linears.  s is dense [B*T, dim/128]; dim % 128 == 0.
 * What Fp8LinearMethod.apply does per layer on a block-quantised checkpoint
 * (python/sglang/srt/layers/quantization/fp8_utils.py:518-548), fused behind the norm. */
int mi355_layernorm_modulate_group_quant(void* q, float* s, const void* x, const void* shift,
                                         const void* scale, int64_t B, int64_t T, int64_t dim,
                                         int64_t ldx, int64_t ldmod, float eps,
                                         const mi355_rowmap* x_map, void* stream);

/* out = residual + gate[b,:] * y with both bf16 roundings (flux.py:651-652,
 * 664-666, 671-688, 547-550).  y/residual/out [B*T, dim]. */
int mi355_gate_residual(void* out, const void* residual, const void* gate, const void* y,
                        int64_t B, int64_t T, int64_t dim, int64_t ldy, int64_t ldr,
                        int64_t ldo, int64_t ldg, const mi355_rowmap* y_map,
                        const mi355_rowmap* r_map, void* stream);

/* elementwise bf16 -> bf16: act 0 = silu, 1 = gelu(tanh). */
int mi355_activation(void* out, const void* x, int64_t rows, int64_t dim, int64_t ldx,
                     int64_t ldo, int act, void* stream);

/* Sinusoidal timestep embedding, fp32 math (diffusers get_timestep_embedding;
 *   torch restatement python/sglang/jit_kernel/tests/test_timestep_embedding.py:17-47;
 *   CUDA twin python/sglang/jit_kernel/csrc/diffusion/timestep_embedding.cuh:19-150).
 * t fp32 [B]; out [B,dim] fp32 (out_bf16=0) or bf16. */
int mi355_timestep_embedding(const float* t, void* out, int64_t B, int64_t dim,
                             int flip_sin_to_cos, float downscale_freq_shift, float scale,
                             float max_period, int out_bf16, void* stream);

/* Flow-match Euler update, FlowMatchEulerDiscreteScheduler.step
 *   (models/schedulers/scheduling_flow_match_euler_discrete.py:495-531):
 *   out = bf16( sample.float() + float(bf16(bf16(dt) * v)) ), dt = sigma_next - sigma
 *   (a 0-dim fp32 tensor times a bf16 tensor is a bf16 product in eager). */
int mi355_euler_step(void* out, const void* sample, const void* v, float sigma,
                     float sigma_next, int64_t n, void* stream);

/* Serial classifier-free guidance combine (pipelines_core/stages/denoising.py:1486-1488):
 *   out = bf16(u + bf16(s * bf16(c - u))). */
int mi355_cfg_combine(void* out, const void* cond, const void* uncond, float scale,
                      int64_t n, void* stream);

/* The two optional steps the reference applies to the combined prediction
 *   (pipelines_core/stages/denoising.py:1490-1506, rescale_noise_cfg :1164-1191), in place on
 *   `noise`; both read whole-tensor moments first, gathered through `scratch` (device, fp32,
 *   16-byte aligned, MI355_CFG_SCRATCH_FLOATS(batch) elements; contents need no initialisation).
 * cfg_normalize: if ||noise||_2 > factor * ||cond||_2 (norms over the WHOLE tensor, fp32)
 *   noise = bf16(noise * bf16(factor * ||cond|| / ||noise||)).  factor <= 0: no-op.  No host
 *   synchronisation: the comparison happens on the device.
 * guidance_rescale: per sample, with unbiased stds rounded to bf16 as eager does,
 *   noise = bf16( bf16(g * bf16(noise * bf16(std(cond) / std(noise)))) + bf16((1 - g) * noise) ).
 *   g <= 0: no-op.  n and per_sample must be multiples of 8. */
#define MI355_CFG_SCRATCH_FLOATS(batch) (256 * (int64_t)(batch))
int mi355_cfg_normalize(void* noise, const void* cond, float factor, int64_t n, float* scratch,
                        void* stream);
int mi355_guidance_rescale(void* noise, const void* cond, float guidance_rescale, int64_t batch,
                           int64_t per_sample, float* scratch, void* stream);

/* mi355_guidance_rescale without the scratch argument.  No counterpart in the reference: it is
 * for callers that cannot allocate device memory -- code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  The per-sample partial moments live in a
 * 64-KiB array that is part of the library's device image (every GPU the library runs on
 * holds one copy until the process ends); the call itself allocates and frees nothing.
 * Same arithmetic, same results as mi355_guidance_rescale.  batch <= 64
 * (MI355_STATIC_RESCALE_MAX_BATCH), larger batches are rejected with MI355_EINVAL.  All calls
 * on one GPU share that GPU's array: calls that overlap in time on one GPU (two streams) race;
 * calls on one stream, or otherwise ordered, are safe, and nothing needs clearing between
 * them. */
#define MI355_STATIC_RESCALE_MAX_BATCH 64
int mi355_guidance_rescale_static(void* noise, const void* cond, float guidance_rescale,
                                  int64_t batch, int64_t per_sample, void* stream);

/* Every option of the attention kernel in one argument block; the two entry points above are
 * this call with v_is_f16 = 0 / 1 and no query transform.
 * Query transform (optional): per-head RMSNorm and interleaved RoPE applied to each Q row as the
 * kernel loads it, with exactly the arithmetic (and summation order) of mi355_qk_norm_rope on
 * q -- rows s < q_split use q_norm_w0, the rest q_norm_w1 (fp32 [D], both or neither);
 * q_cos / q_sin fp32 [Sq, D/2] (both or neither); position of row s is s.  The q tensor in
 * memory is left untouched.  Saves the read-modify-write pass over Q
 * (apply_qk_norm + apply_flashinfer_rope_qk_inplace, models/dits/flux.py:356-395). */
typedef struct {
  const void* q;
  const void* k;
  const void* v;
  void* out;
  float* lse; /* fp32 [B,H,Sq] or NULL */
  int64_t B, Sq, Sk, H, Hk, D;
  int64_t q_sb, q_ss, q_sh, k_sb, k_ss, k_sh, v_sb, v_ss, v_sh, o_sb, o_ss, o_sh;
  float softmax_scale;
  int causal;
  int v_is_f16;
  const float* q_norm_w0;
  const float* q_norm_w1;
  int64_t q_split;
  float q_eps;
  const float* q_cos;
  const float* q_sin;
} mi355_attn_args;
int mi355_flash_attn_fwd_ex(const mi355_attn_args* args, void* stream);

/* Merge of two partial attention results over disjoint key sets (ring / context-parallel
 * attention): out = (a e^(la-m) + b e^(lb-m)) / (e^(la-m) + e^(lb-m)), lse = log(sum) + m,
 * m = max(la, lb); an infinite log-sum-exp marks an empty side.
 *   merge_state_v2(Tensor v_a, Tensor s_a, Tensor v_b, Tensor s_b, Tensor! v_merged,
 *                  Tensor! s_merged)   (sgl-kernel/csrc/common_extension.cc:55-56,
 *                  sgl-kernel/csrc/attention/merge_attn_states.cu:31-109,180-204)
 * out/a/b bf16 [B, S, H, D] and the fp32 log-sum-exps [B, S, H], each addressed through three
 * element strides {batch, token, head} (last dimension of out/a/b contiguous), so both the
 * op's [tokens, heads] layout and mi355_flash_attn_fwd's [B, H, S] layout fit.  out may alias
 * a or b; out_lse (and out_lse_strides) may be NULL.  D % 8 == 0. */
int mi355_merge_attn_states(void* out, float* out_lse, const void* a, const float* a_lse,
                            const void* b, const float* b_lse, int64_t B, int64_t S, int64_t H,
                            int64_t D, const int64_t* out_strides, const int64_t* a_strides,
                            const int64_t* b_strides, const int64_t* out_lse_strides,
                            const int64_t* a_lse_strides, const int64_t* b_lse_strides,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_DIT_H */
