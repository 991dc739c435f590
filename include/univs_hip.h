/*
 * univs_hip.h -- C ABI of libunivs_hip.so: the MI355X (gfx950) native operators of the UniVS
 * per-clip inference hot path.
 *
 * Conventions (the same for every entry point; they mirror what the reference's native operator
 * enforces at mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_attn_cuda.cu:33-59):
 *   - all data pointers are DEVICE pointers to dense, contiguous, row-major buffers;
 *     shape tables (`spatial_shapes`, `level_start_index`) are HOST pointers (they parameterise the
 *     launch; the reference keeps them in device memory and re-reads them per thread,
 *     ms_deform_im2col_cuda.cuh:277-280);
 *   - `stream` is a hipStream_t (NULL = the null stream); launches are asynchronous, never
 *     synchronise the host, and are legal inside hipGraph stream capture;
 *   - outputs are caller-allocated and fully overwritten (no zero-fill needed beforehand);
 *   - return value: UNIVS_OK (0) or a negative UNIVS_ERR_* code; univs_last_error() returns a
 *     thread-local human-readable message for the last failure.  Launch errors are RETURNED
 *     (the reference only printf's them, ms_deform_im2col_cuda.cuh:953-957).
 *   - no torch / ATen types anywhere in this ABI.
 */
#ifndef UNIVS_HIP_H
#define UNIVS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNIVS_OK 0
#define UNIVS_ERR_INVALID_ARGUMENT (-1)
#define UNIVS_ERR_NOT_IMPLEMENTED (-2)
#define UNIVS_ERR_LAUNCH (-3)

#define UNIVS_MAX_LEVELS 8

/* Library identification: "univs_hip <semver> gfx950". */
const char* univs_version(void);
/* Message for the last non-OK return on this thread ("" if none). */
const char* univs_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale deformable attention, forward.
 * Replaces: ms_deform_attn_forward  (ops/src/vision.cpp:19, ops/src/ms_deform_attn.h:25-44)
 *           -> ms_deform_attn_cuda_forward (ops/src/cuda/ms_deform_attn_cuda.cu:25-85)
 *           -> ms_deformable_im2col_gpu_kernel (ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304).
 *
 *   value          [N, S, M, D]        S = sum_l H_l*W_l
 *   spatial_shapes [L, 2] (H_l, W_l)   HOST int64
 *   level_start    [L]                 HOST int64
 *   sampling_loc   [N, Lq, M, L, P, 2] (x, y) normalised to [0,1] per level
 *   attn_weight    [N, Lq, M, L, P]
 *   out            [N, Lq, M*D]
 *   out[n,q,m,:] = sum_{l,p} attn[n,q,m,l,p] * bilinear(value_l[n,:,m,:], x*W_l-0.5, y*H_l-0.5)
 *   with zero padding outside the level (== grid_sample(align_corners=False, padding='zeros')).
 * The reference's `im2col_step` batching argument has no effect on results and is not part of this
 * ABI (one launch covers the whole batch); the Python shim still validates it like the reference
 * (batch % min(batch, im2col_step) == 0, ms_deform_attn_cuda.cu:55-57).
 * ------------------------------------------------------------------------------------------- */
int univs_msda_forward_f32(const float* value, const int64_t* spatial_shapes,
                           const int64_t* level_start, const float* sampling_loc,
                           const float* attn_weight, int N, int S, int M, int D, int L, int Lq,
                           int P, float* out, void* stream);
int univs_msda_forward_f64(const double* value, const int64_t* spatial_shapes,
                           const int64_t* level_start, const double* sampling_loc,
                           const double* attn_weight, int N, int S, int M, int D, int L, int Lq,
                           int P, double* out, void* stream);

/* Backward of the above (ops/src/ms_deform_attn.h:46-66, cuda/ms_deform_attn_cuda.cu:88-153):
 *   grad_output        [N, Lq, M*D]
 *   grad_value         [N, S, M, D]       (zero-filled here, then accumulated with float atomics)
 *   grad_sampling_loc  [N, Lq, M, L, P, 2]
 *   grad_attn_weight   [N, Lq, M, L, P]
 * Completes the operator boundary; training is not on the inference hot path, so this kernel is
 * correctness-first (one thread per sample). */
int univs_msda_backward_f32(const float* value, const int64_t* spatial_shapes,
                            const int64_t* level_start, const float* sampling_loc,
                            const float* attn_weight, const float* grad_output, int N, int S, int M,
                            int D, int L, int Lq, int P, float* grad_value, float* grad_sampling_loc,
                            float* grad_attn_weight, void* stream);

/* ---------------------------------------------------------------------------------------------
 * MSDeformAttn input preparation: sampling locations and attention weights from the raw projections.
 * Replaces: the elementwise tail of MSDeformAttn.forward (ops/modules/ms_deform_attn.py:100-113):
 *           softmax over the L*P logits of each (query, head); reference point + offset / (W_l, H_l).
 *   proj        [N*Lq, row_stride] rows; columns [0, M*L*P*2) = sampling offsets (m, l, p, xy),
 *               columns [n_off, n_off + M*L*P) = attention logits (m, l, p)   (one merged projection,
 *               or two tensors seen as one row with a stride)
 *   ref_points  [N or 1, Lq, L, 2] normalised reference points (ref_batch_stride = 0 broadcasts over N)
 *   spatial_shapes  HOST int64 [L, 2] as in univs_msda_forward_f32
 *   loc   [N, Lq, M, L, P, 2],  attn  [N, Lq, M, L, P]     (the operands of univs_msda_forward_f32)
 *   P == 4 and L <= 4, else UNIVS_ERR_NOT_IMPLEMENTED.
 * ------------------------------------------------------------------------------------------- */
int univs_msda_prepare_f32(const float* proj, int row_stride, int n_off, const float* ref_points,
                           long long ref_batch_stride, const int64_t* spatial_shapes, int N, int Lq, int M,
                           int L, int P, float* loc, float* attn, void* stream);

/* MSDeformAttn core fed with the RAW projections -- the fusion of univs_msda_prepare_f32 and univs_msda_forward_f32:
 *   out = ms_deform_attn_forward(value, ..., loc, attn)   with
 *   loc  = ref_points + offsets / (W_l, H_l)                                  (ms_deform_attn.py:106-109)
 *   attn = softmax(logits.view(N, Lq, M, L*P), -1)                            (ms_deform_attn.py:103)
 * the [N, Lq, M, L, P, 2] / [N, Lq, M, L, P] tensors are never materialised -- on HEAD-MAJOR operands, half a head per
 * workgroup (csrc/msda_strips.hip: resident
 * row-circular windows, a lane owns a sample, two workgroups per CU):
 *   value_hm [N][M][2][S][16]   value_proj's output in blocks of 16 channels: univs_linear_blocked_f32(..., S, 16);
 *   proj_hm  [N][M][S][P][3 L]  per (query, head, point): L offset pairs (x, y) then L attention logits, levels ordered by
 *                               size, largest first (ties: lower index first) -- the merged sampling_offsets /
 *                               attention_weights Linear with its weight rows permuted, univs_linear_blocked_f32(..., S, 3 L P);
 *   ref_points [N or 1][S][2]   ONE reference point per query, shared by all levels (the encoder's normalised pixel
 *                               centres, msdeformattn.py:143-158 with valid_ratio == 1); ref_batch_stride = 2 S or 0;
 *   out [N][S][M * 32]          the standard layout output_proj consumes.
 * Covered: D == 32, P == 4, 1 <= L <= 4, Lq == S; otherwise (or when the generic kernel is forced with
 * univs_msda_set_impl(1)) UNIVS_ERR_NOT_IMPLEMENTED and the caller runs the standard-layout operators. */
int univs_msda_forward_strips_f32(const float* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                  const float* proj_hm, const float* ref_points, long long ref_batch_stride, int N, int S,
                                  int M, int D, int L, int Lq, int P, float* out, void* stream);

/* The same operator one generation later (csrc/msda_heads.hip; what MSDeformAttn.forward runs since round 6): a lane owns a
 * sample of a FULL head, one 8-wave workgroup per CU with the windows of a 12 x 8 tile resident (<= 160 KB of LDS), the
 * workgroups of an XCD walk adjacent tile columns in lockstep so that shared halo columns come from HBM once.  Operands as
 * univs_msda_forward_strips_f32 except
 *   value_hm [N][M][S][32]      value_proj's output in blocks of 32 channels (one head): univs_linear_blocked_f32(..., S, 32).
 * Covered: D == 32, P == 4, 1 <= L <= 4, Lq == S; otherwise UNIVS_ERR_NOT_IMPLEMENTED. */
int univs_msda_forward_heads_f32(const float* value_hm, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                 const float* proj_hm, const float* ref_points, long long ref_batch_stride, int N, int S,
                                 int M, int D, int L, int Lq, int P, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Process-wide settings.  The library reads NO environment variable: everything that selects an implementation or a
 * tuning parameter is set here (0 / negative = the default).  univs_configure(NULL) restores the defaults.  Settings
 * apply to calls that start afterwards; they are read once per call (a consistent snapshot), so changing them from
 * another thread never tears a call -- but it does change what concurrent callers run, so tests and benchmarks that
 * select kernels do it from one thread.
 * ------------------------------------------------------------------------------------------- */
typedef struct UnivsConfig {
  int size;               /* sizeof(UnivsConfig) of the caller (versioning) */
  int msda_impl;          /* 0 auto, 1 generic direct-gather kernel, 2 LDS-tiled kernels where they apply */
  int msda_strip_w;       /* msda_strips / msda_heads: tile width in pixels of the finest level (default 12 / 16) */
  int msda_strip_h;       /* msda_strips / msda_heads: tile height (default 8 / 6; smaller tilings are tried until the windows fit the LDS) */
  int msda_halo;          /* LDS-tiled kernels: sampling offsets covered by the windows, pixels (default 6) */
  int msda_grid;          /* LDS-tiled kernels: workgroups launched (default: 4 x CUs for msda_strips, 1 x CUs for msda_heads / msda_tiled2) */
  int mask_decode_impl;   /* 0 by size, 1 exact-f32 MFMA kernel, 2 split-bf16 kernel wherever its preconditions hold */
  int mask_decode_ct;     /* split-bf16 mask decode: 4 = the 64-column kernel (default 2: 32 columns) */
  int mask_decode_ablate; /* timing experiments: 1 memory side only, 2 compute side only (results are then meaningless) */
  int window_attn_v1;     /* 1: the first 7x7 window-attention kernel (kernel benchmarks) */
  int linear_terms;       /* fp32 Linears on the matrix cores (W-stationary kernel): 0 / 3 = two row-scaled fp16 parts per operand,
                             three products (linear_f16x3.hip; the default), 6 = three bf16 parts per operand, six products
                             (linear_split.hip) */
  int linear_ablate;      /* timing experiments on linear_f16x3 (results then only valid for inputs already in fp16's range):
                             1 = no row-maximum pass over x (scale 1); 2 / 3 / 4 = univs_mlp_presplit_f32 (encoder FFN and Swin stage-1
                             shapes) without its MFMAs / without the LDS reads of the weight fragments / without the activation
                             and split of the hidden activations: results are then WRONG, kernel benchmarks only;
                             6 = univs_linear_presplit_f32 keeps the row-range x pass kernel (gemm_f16x3_stream) where it would take the
                             two-dimensional tiling (gemm_f16x3_tile; bit-identical results: A / B runs); 7 / 8 / 9 = that kernel with 2 / 3 / 4
                             k-steps of loads in flight where K allows (kernel benchmarks; with linear_grid_x = 3..5 as its CT and
                             linear_rows_per_pass = 128 / 192 / 256 as its feature-tile width); 10 = univs_mlp_presplit_f32 at C = 128 / 192 / 256
                             runs its phase-shifted form (the two waves of a SIMD half a chunk apart: mlp_f16x3_ps; same results, A / B runs) */
  int mask_decode_chunked;/* 1: the exact-f32 mask kernel always in its chunked form (kernel benchmarks; default 0: small maps with
                             C == 256 request every row of their columns at once, skinny_gemm_f32_oneshot) */
  int mask_decode_wave_tiles; /* split-bf16 mask decode: column tiles a wave should get before a workgroup is added (default 1;
                             the split of A is per workgroup, larger values trade balance for fewer splits) */
  int linear_rows_per_pass; /* kernel benchmarks: output features per pass of the three-product Linears (a multiple of 16, <= 128;
                             0 = as many as fit: 128 for the streamed kernel, the LDS capacity for the W-resident one) */
  int linear_grid_x;      /* kernel benchmarks: workgroups along the rows of the three-product Linears (0 = by shape) */
  int xattn_segments;     /* kernel benchmarks: key segments per (batch entry, head, query chunk) of univs_cross_attention_f32 (0 = by shape) */
  int msda_sched;         /* msda_heads: 0 / 2 = the workgroups of an XCD walk adjacent tile columns in lockstep rounds (default),
                             1 = one contiguous range of the (plane, column, row) sequence per workgroup (generation 5's rule; A / B runs) */
  int reserved[2];
} UnivsConfig;
int univs_configure(const UnivsConfig* cfg);
int univs_get_config(UnivsConfig* out);

/* Shorthand for UnivsConfig.msda_impl: 0 = auto (default), 1 = generic direct-gather kernel, 2 = LDS-tiled encoder kernel
 * (falls back to generic when its preconditions do not hold). */
int univs_msda_set_impl(int impl);
/* Which implementation the last univs_msda_forward_f32 call on this thread launched: 1 generic,
 * 2 LDS-tiled, 0 none yet.  Lets tests assert that a fast path really ran (no silent fallback). */
int univs_msda_last_impl(void);
/* Generation of the LDS-tiled kernel the last MSDA forward call on this thread launched: 6 = a full head per lane-sample
 * (msda_heads.hip, head-major operands), 5 = strips at half a head per
 * workgroup (msda_strips.hip, head-major operands), 2 = producer / consumer waves (msda_tiled2.hip, standard layouts),
 * 0 = none (generic kernel). */
int univs_msda_last_tiled_generation(void);

/* y[M, N] = act(x[M, K] * W[N, K]^T + bias[N]) (+ residual): torch.nn.functional.linear for contiguous float32 operands --
 * the token projections of MSDeformAttn (mask2former/modeling/pixel_decoder/ops/modules/ms_deform_attn.py:95-113: value_proj,
 * sampling_offsets + attention_weights, output_proj), the encoder FFN, and the Swin block's MLP and qkv / proj projections
 * (mask2former/modeling/backbone/swin.py:35-58 Mlp.forward: fc1 -> nn.GELU() -> fc2, :291-293 `x = shortcut + self.mlp(...)`,
 * :137-141, :163).  fp32 emulated on the fp16 matrix cores: both operands are scaled per row by a power of two into fp16's
 * range and split into two fp16 parts, three of the four part products are accumulated in fp32 (error <= 2^-21.7 per
 * product, below the rounding error of an fp32 FMA chain over K terms; linear_f16x3.hip).  UnivsConfig.linear_terms = 6
 * selects the older six-product split into three bf16 parts (error <= 3 * 2^-24 per product; linear_split.hip).
 *   act = 0: none, 1: ReLU, 2: GELU  x * 0.5 * (1 + erf(x / sqrt 2))  (nn.GELU(approximate='none'); erf by Abramowitz & Stegun
 *         7.1.26, branch-free: within 4.7e-7 absolute of the exact value in fp32, ATen's fp32 GELU is within 1.2e-6);
 *   residual (NULL or [M, N], contiguous): y = x W^T + bias + residual.  act != 0 together with a residual is rejected.
 * Covered: K % 128 == 0 or K % 96 == 0, K <= 768 (this entry splits W inside every workgroup: wider K goes through
 * univs_presplit_weights_f32 + univs_linear_presplit_f32), N % 4 == 0, M >= 2048, 16-byte aligned pointers,
 * M * max(N, K) * 4 < 2^31; anything else returns UNIVS_ERR_NOT_IMPLEMENTED without touching y (the caller keeps its library
 * GEMM).  bias may be NULL.  Inf / NaN inputs make the results of their own row Inf / NaN (as in a GEMM; with
 * linear_terms = 6 an infinite operand yields NaN where an fp32 GEMM yields Inf). */
int univs_linear_fused_f32(const float* x, const float* weight, const float* bias, const float* residual, long long M, int N,
                           int K, int act, float* y, void* stream);

/* The same Linear (bias only) with a COLUMN-BLOCKED output per batch element:
 *   y[M / rows_per_batch][N / col_block][rows_per_batch][col_block],  y[b][c][s][i] = (x W^T + bias)[b * rows_per_batch + s][c * col_block + i]
 * -- the head-major operand layouts of univs_msda_forward_strips_f32, written by the producing Linear's epilogue at no
 * extra cost (ms_deform_attn.py:95-102: value_proj with col_block = 16, the merged offset / logit projection with
 * col_block = 3 L P).  Covered: N % col_block == 0, col_block % 4 == 0, M % rows_per_batch == 0 and the
 * coverage rules of univs_linear_fused_f32; otherwise UNIVS_ERR_NOT_IMPLEMENTED. */
int univs_linear_blocked_f32(const float* x, const float* weight, const float* bias, long long M, int N, int K,
                             int rows_per_batch, int col_block, float* y, void* stream);

/* out[b][c][r] = x[b][r][c]: contiguous float32 [B, R, C] -> [B, C, R].  The layout changes at the edges of the Swin
 * backbone: stage outputs tokens [B, H*W, C] -> NCHW (mask2former/modeling/backbone/swin.py:676-683
 * `permute(0, 3, 1, 2).contiguous()`), PatchEmbed's NCHW -> tokens (:331-336 `flatten(2).transpose(1, 2)`).
 * Covered: R % 4 == 0, C % 4 == 0, B <= 65535, 16-byte aligned pointers; otherwise UNIVS_ERR_NOT_IMPLEMENTED (the caller
 * keeps its own permuted copy). */
int univs_transpose_f32(const float* x, long long B, int R, int C, float* out, void* stream);
/* the same with `in_batch_stride` floats (a multiple of 4, >= R * C; 0 = dense) between consecutive input matrices: a row range
 * x[:, r0:r0 + R, :] of a wider [B, S, C] tensor without a copy of its own -- the per-level split of the pixel decoder's encoder output
 * (mask2former/modeling/pixel_decoder/msdeformattn.py:336-344: `torch.split(y, ...)`, `z.transpose(1, 2).view(bs, -1, H_l, W_l)`). */
int univs_transpose_strided_f32(const float* x, long long B, int R, int C, long long in_batch_stride, float* out, void* stream);
/* the full form.  out_batch_stride (0 = dense): out may be a row range of a wider [B, S, R] tensor -- the levels of the pixel decoder's
 * encoder input written straight into `src_flatten` (msdeformattn.py:168-188: `torch.cat(src_flatten, 1)`).  row_affine [B * R][2] | NULL:
 * input row (b, r) is read as x * scale + bias -- `input_proj`'s GroupNorm (univs_group_norm_affine_f32 on the raw convolution output)
 * applied on the way through (:205-212).  addend [C][R] + out2 | NULL: a second output out2 = out + addend in out's layout -- the first
 * encoder layer's `with_pos_embed(src, pos)` (:61-63) with addend = the level's rows of lvl_pos_embed_flatten. */
int univs_transpose_ex_f32(const float* x, long long B, int R, int C, long long in_batch_stride, const float* row_affine, float* out,
                           long long out_batch_stride, const float* addend, float* out2, void* stream);

/* Shorthand for UnivsConfig.mask_decode_impl (univs_mask_decode_f32 / univs_mask_decode_attn_f32):
 * 0 = by size (default: large feature maps take the split-bf16 kernel), 1 = exact-f32 MFMA kernel
 * (v_mfma_f32_32x32x2_f32, bit-identical to a k-ordered fp32 fmaf chain), 2 = fp32 emulated on the bf16 matrix
 * cores from an exact 3-way split of both operands ("bf16 x 6", error <= 3 * 2^-24 per product) wherever its
 * preconditions hold (C % 64 == 0, HW % 4 == 0, 16-byte aligned pointers), the f32 kernel elsewhere. */
int univs_mask_decode_set_impl(int impl);
/* 1 / 2: which of the two kernels the last mask-decode call on this thread launched (0: none yet). */
int univs_mask_decode_last_impl(void);

/* ---------------------------------------------------------------------------------------------
 * Mask decode: per-frame contraction of mask embeddings with per-pixel features.
 * Replaces: torch.einsum("btqc,btchw->btqhw", mask_embed, mask_features).transpose(1, 2)
 *           (univs/modeling/transformer_decoder/video_mask2former_transformer_decoder_univs.py:527-528)
 *   mask_embed    [T, Q, C]
 *   mask_features [T, C, HW]      (NCHW feature map, HW flattened)
 *   out           [Q, T, HW]      (the [B=1, Q, T, H, W] layout callers consume)
 * fp32 in / fp32 accumulate on the f32 MFMA path (exact fmaf chain, no reduced precision).
 * ------------------------------------------------------------------------------------------- */
int univs_mask_decode_f32(const float* mask_embed, const float* mask_features, int T, int Q, int C,
                          int HW, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused attention-mask generation for the next decoder layer.
 * Replaces: the mask einsum above followed by F.interpolate(..., size=(h,w), mode="bilinear",
 *           align_corners=False), sigmoid() < 0.5, and the all-masked-row reset
 *           (...decoder_univs.py:527, :555-566 and :390) WITHOUT materialising full-resolution logits:
 *           bilinear resampling commutes with the channel contraction, so the caller passes mask
 *           features already resampled to the target level size.
 *   mask_embed   [T, Q, C]
 *   feat_lowres  [T, C, hw]        mask_features bilinearly resampled to (h, w)
 *   attn_mask    [T, Q, hw] uint8  1 = key masked out (logit < 0), 0 = key visible; a row that
 *                                  would be fully masked is written as all 0 (":390" rule)
 *   row_any_ws   [T*Q] uint32      caller-provided scratch (no allocation inside the library, so the
 *                                  call is legal under hipGraph capture); contents are clobbered
 *   The reference repeats the mask over the attention heads; consumers broadcast instead.
 * ------------------------------------------------------------------------------------------- */
int univs_mask_decode_attn_f32(const float* mask_embed, const float* feat_lowres, int T, int Q,
                               int C, int hw, uint8_t* attn_mask, uint32_t* row_any_ws,
                               void* stream);

/* The same contraction with the all-masked-row rule DEFERRED to the consumer (no flag memset, no second pass over the mask): row r of
 * attn_mask is to be read as "every key visible" wherever row_flags[r] != generation.  `generation`: any non-zero value the caller has
 * not used on this row_flags buffer before (a counter); row_flags [T * Q] keeps older generations' values and needs no initialisation
 * beyond not containing the first generation used.  univs_cross_attention_flagged_f32 consumes (attn_mask, row_flags, generation);
 * univs_attn_mask_rows_reset turns attn_mask into the tensor univs_mask_decode_attn_f32 would have written. */
int univs_mask_decode_attn_deferred_f32(const float* mask_embed, const float* feat_lowres, int T, int Q, int C, int hw, uint8_t* attn_mask,
                                        uint32_t* row_flags, uint32_t generation, void* stream);
int univs_attn_mask_rows_reset(uint8_t* attn_mask, const uint32_t* row_flags, uint32_t generation, long long rows, long long hw,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pre-split weights for the three-product fp16 GEMM kernels whose W streams through LDS (gemm_f16x3_stream.hip): wide-K
 * Linears (K >= 768) and the 3 x 3 convolution.  The split of W -- row maxima, power-of-two row scales, two fp16 parts in
 * the kernels' LDS order -- is done ONCE per weight tensor; the caller keeps the result (4 bytes per element + N floats)
 * for as long as the weights do not change.
 *   w      [N, K] fp32 (conv = 0), or a convolution weight [N, Cin, 3, 3] with conv = 1 (then K = 9 * Cin, tap-major), or
 *          [N, K] fp32 with conv = 2: the SECOND Linear of univs_mlp_presplit_f32 (inside every 32-wide k-step the k-order is
 *          (4 g + e, 16 + 4 g + e), g = 0..3, e = 0..3: the order in which that kernel holds its hidden activations)
 *   wp     N * K * 4 bytes, 16-byte aligned (out);  winv [N] fp32 (out)
 * univs_linear_presplit_f32 : y = act(x W^T + bias) (+ residual), arguments as univs_linear_fused_f32 with (wp, winv) in place
 *   of w; K % 128 == 0 or K % 96 == 0, N % 4 == 0, M >= 2048.  UNIVS_ERR_NOT_IMPLEMENTED when the shape is not covered.
 *   Two kernels behind it with bit-identical results: for K >= 384, K % 64 == 0, N >= 128 the two-dimensional tiling
 *   (gemm_f16x3_tile.hip: x is split once per workgroup and shared by its waves through LDS), else the row-range x pass kernel
 *   (gemm_f16x3_stream.hip); UnivsConfig.linear_ablate = 6 forces the latter.
 * univs_conv3x3_presplit_f32: y = conv2d(x, w, bias=None, stride=1, padding=1) for a 3 x 3 kernel on contiguous float32 NCHW
 *   tensors, x [T, Cin, H, W] -> y [T, Cout, H, W], as a GEMM with tap addressing of x (the FPN output convolution of the
 *   pixel decoder, mask2former/modeling/pixel_decoder/msdeformattn.py:227-232, :352; its GroupNorm + ReLU stay separate).
 *   Covered: Cin % 128 == 0, Cout % 16 == 0, T*H*W >= 4096, 16-byte aligned pointers, tensors < 2^31 bytes; otherwise
 *   UNIVS_ERR_NOT_IMPLEMENTED (the caller keeps its library convolution).
 * Replaces: the Linear call sites of univs_linear_fused_f32 with K >= 768 (swin.py:35-58, msdeformattn.py:87-91) and
 *   F.conv2d at msdeformattn.py:227-232.
 * ------------------------------------------------------------------------------------------- */
int univs_presplit_weights_f32(const float* w, int N, int K, int conv, void* wp, float* winv, void* stream);
int univs_linear_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, const float* residual,
                              long long M, int N, int K, int act, float* y, void* stream);
int univs_conv3x3_presplit_f32(const float* x, const void* wp, const float* winv, int T, int Cin, int Cout, int H, int W,
                               float* y, void* stream);
/* The W-RESIDENT kernel of univs_linear_fused_f32 / univs_linear_blocked_f32 (K <= 768; linear_f16x3.hip) on the pre-split image
 * (mode 0 of univs_presplit_weights_f32): staging a workgroup's slab of W in LDS becomes a copy.  Splitting the slab inside every
 * workgroup -- what the raw-weight entries do -- measured 13 - 15 us at the head of every launch (profiles/r05_gemm_phase_trace_v1.txt).
 * Same arguments, coverage and results (bit for bit) as the raw-weight entries with (wp, winv) in place of `weight`; replaces the
 * same call sites (ms_deform_attn.py:95-113, swin.py:137-141, :163, :35-58). */
int univs_linear_resident_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, const float* residual,
                                       long long M, int N, int K, int act, float* y, void* stream);
int univs_linear_blocked_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, long long M, int N, int K,
                                      int rows_per_batch, int col_block, float* y, void* stream);
/* univs_conv3x3_nhwc_presplit_f32: the same convolution on a CHANNELS-LAST operand x [T, H, W, Cin] (y stays NCHW [T, Cout, H, W]); same
 *   weights image, same k order, bit-identical results: a lane's 8 input channels of a tap are 32 contiguous bytes (two 16-byte loads)
 *   where the NCHW operand takes eight 4-byte loads a plane apart.  For producers that can write channels last (ops.upsample2x_add). */
int univs_conv3x3_nhwc_presplit_f32(const float* x, const void* wp, const float* winv, int T, int Cin, int Cout, int H, int W,
                                    float* y, void* stream);
/* univs_conv1x1_presplit_f32: y = conv2d(x, w [Cout, Cin, 1, 1], bias) (stride 1, no padding) on contiguous float32 NCHW tensors
 *   through the same kernel (tap addressing with the centre tap alone); (wp, winv) = univs_presplit_weights_f32(w, Cout, Cin, 0),
 *   bias [Cout] or NULL rides in the epilogue.  Covered: Cin % 96 == 0 or Cin % 128 == 0, Cout % 16 == 0, T*H*W >= 4096.
 *   Replaces: the 1 x 1 convolutions of the pixel decoder -- lateral convolutions, `mask_features`, `input_proj`
 *   (mask2former/modeling/pixel_decoder/msdeformattn.py:205-232, :262-283, :331-355). */
int univs_conv1x1_presplit_f32(const float* x, const void* wp, const float* winv, const float* bias, int T, int Cin, int Cout, int H,
                               int W, float* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * y[M, N] = act((x [+ x_add]) W[f_off : f_off + N]^T + bias) [+ residual] [-> LayerNorm] for FEW rows (small_linear.hip): the per-token
 * Linears of the UniVS decoder on its Q' T query tokens with the elementwise steps around them in the same launch.
 * Replaces (univs/modeling/transformer_decoder/transformer_layers.py):
 *   `q = in_proj_q(tgt + query_pos)`                                  (:30-46, :95-115; x_add = query_pos, f_off selects the rows
 *                                                                      of in_proj_weight: q 0, k E, v 2 E)
 *   `tgt = norm(tgt + out_proj(attn))`, `tgt = norm(tgt + linear2(h))` (:42-46, :106-110, :160-166; residual = tgt, ln_* = norm)
 *   `h = relu(linear1(tgt))`, the mask-embedding MLP                   (:150-166, :205-217; relu = 1)
 * Three-product fp16 arithmetic of univs_linear_fused_f32 (fp32-accurate); LayerNorm with exact two-pass statistics.
 *   wp, winv   univs_presplit_weights_f32(W [n_w, K], n_w, K, 0, ...) of the WHOLE weight matrix; bias [n_w] | NULL
 *   x, x_add | NULL [M, K]; residual | NULL [M, N]; ln_weight | NULL, ln_bias | NULL [N] (with ln_weight: N == 256); y [M, N]
 *   add_features: x_add enters the first add_features output features only (a multiple of 32; 0 = all) -- a self-attention's
 *                 q, k = in_proj(tgt + query_pos) and v = in_proj(tgt) as ONE launch over the packed in-projection
 *   out_T > 0 (no residual / LayerNorm): the M rows are (q, t) pairs, q-major, t < out_T; the result of row (q, t) is stored at row
 *                 (t, q): `mask_embed(decoder_output)` [Q', T, C] handed on as [T, Q', C] without a transposed copy
 *   K % 32 == 0, K <= 256, N % 16 == 0, f_off % 4 == 0, 16-byte aligned pointers; otherwise UNIVS_ERR_NOT_IMPLEMENTED.
 * ------------------------------------------------------------------------------------------- */
int univs_small_linear_presplit_f32(const float* x, const float* x_add, const void* wp, const float* winv, const float* bias, int n_w,
                                    int f_off, const float* residual, const float* ln_weight, const float* ln_bias, float ln_eps,
                                    long long M, int N, int K, int relu, int add_features, int out_T, float* y, void* stream);
/* y = L_n(act(... L_1(LN?(x)))) for up to three 256 -> 256 Linears on FEW rows in one launch: the mask-embedding MLP of a prediction head
 * (univs/modeling/transformer_decoder/transformer_layers.py:205-217, called at ...decoder_univs.py:520 on `decoder_norm(output)`, :513).
 *   wp / winv / bias / relu: `stages` entries each (univs_presplit_weights_f32 images of [256, 256] weights; bias entries may be NULL;
 *   relu[s] != 0: ReLU behind stage s).  Between stages the rows stay in LDS: results are bit-identical to `stages` calls of
 *   univs_small_linear_presplit_f32.  in_ln_weight != NULL: nn.LayerNorm(256) on the input rows first (two-pass statistics); x_normed
 *   (NULL or [M, 256]) receives the normalised rows.  out_T as in univs_small_linear_presplit_f32.  UNIVS_ERR_NOT_IMPLEMENTED when not covered. */
int univs_small_mlp_presplit_f32(const float* x, int stages, const void* const* wp, const float* const* winv, const float* const* bias,
                                 const int* relu, const float* in_ln_weight, const float* in_ln_bias, float in_ln_eps, float* x_normed,
                                 long long M, int out_T, float* y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * y[M, C] = act(LN(x)[M, C] W1^T + b1) W2^T + b2 (+ residual): a two-Linear MLP in one kernel (mlp_f16x3.hip), three-product fp16
 * arithmetic of univs_linear_fused_f32 in both products; the [M, Hd] hidden activations stay in registers (they are neither
 * written to nor read from memory).  LN: with ln_weight != NULL the rows of x first go through nn.LayerNorm(C) (weight, bias or
 * NULL, eps; two-pass statistics in registers) -- the pre-norm block `x + mlp(norm2(x))` of the Swin stages is then ONE launch
 * with residual == x.  POST-norm: with post_ln_weight != NULL the finished rows (bias and residual added) go through a second
 * nn.LayerNorm(C) before they are stored -- `norm2(src + ffn(src))` of the MSDeformAttn encoder layer; y2 (optional) additionally
 * receives y + post_add[row % post_add_rows] (the next layer's `with_pos_embed(src, pos)`, post_add [post_add_rows, C]).
 *   w1p, w1inv   univs_presplit_weights_f32(W1 [Hd, C], Hd, C, 0, ...)
 *   w2p, w2inv   univs_presplit_weights_f32(W2 [C, Hd], C, Hd, 2, ...)     (mode 2: the MLP k-order)
 *   b1 [Hd], b2 [C], residual [M, C]: optional (NULL);  act: 1 ReLU, 2 GELU (erf form, as univs_linear_fused_f32)
 * Covered: C in {96, 128, 192, 256, 384}, Hd % 32 == 0 (2 Hd + 134 C floats of LDS <= 160 KB; C = 384: 2 Hd + 70 C, one weight image
 * instead of two), M >= 2048, M * C * 4 < 2^31, 16-byte
 * aligned pointers; otherwise UNIVS_ERR_NOT_IMPLEMENTED (the caller keeps two univs_linear_* calls).
 * Replaces: linear2(dropout(activation(linear1(src)))) of the MSDeformAttn encoder layer
 *   (mask2former/modeling/pixel_decoder/msdeformattn.py:87-91) and Mlp.forward + the block's shortcut add of the Swin stages
 *   with C <= 256 (mask2former/modeling/backbone/swin.py:35-58, :291-293; with LN also norm2 of :289-293); with the post-norm also
 *   `src = norm2(src + ...)` and the next layer's `with_pos_embed` (msdeformattn.py:61-63, :91-95).
 * ------------------------------------------------------------------------------------------- */
int univs_mlp_presplit_f32(const float* x, const void* w1p, const float* w1inv, const float* b1, const void* w2p, const float* w2inv,
                           const float* b2, const float* residual, const float* ln_weight, const float* ln_bias, float ln_eps,
                           const float* post_ln_weight, const float* post_ln_bias, float post_ln_eps, const float* post_add,
                           long long post_add_rows, float* y2, long long M, int C, int Hd, int act, float* y, void* stream);
/* the same with `flags`:
 *   UNIVS_MLP_RESIDUAL_IS_NORMED_X (residual NULL, ln_weight given, y distinct from x): the residual is LN(x) itself -- the post-norm
 *     chain `x1 = norm1(x); y = norm2(x1 + ffn(x1))` of an MSDeformAttn encoder layer (msdeformattn.py:124-133) with x = src +
 *     output_proj(...) as the producing Linear's epilogue left it: norm1 costs no launch and no pass over memory of its own;
 *   UNIVS_MLP_DUAL_OUTPUT (post_ln_weight and y2 given, no post_add): y receives the finished rows UN-normalised and y2 their post-LN --
 *     a Swin block's output together with the next block's `norm1` of it, or the stage's output norm (swin.py:286-293, :236, :664-672). */
#define UNIVS_MLP_RESIDUAL_IS_NORMED_X 1
#define UNIVS_MLP_DUAL_OUTPUT 2
int univs_mlp_presplit_v2_f32(const float* x, const void* w1p, const float* w1inv, const float* b1, const void* w2p, const float* w2inv,
                              const float* b2, const float* residual, int flags, const float* ln_weight,
                              const float* ln_bias, float ln_eps, const float* post_ln_weight, const float* post_ln_bias, float post_ln_eps,
                              const float* post_add, long long post_add_rows, float* y2, long long M, int C, int Hd, int act, float* y,
                              void* stream);


/* ---------------------------------------------------------------------------------------------
 * Swin PatchEmbed in one pass (transpose.hip): out[t, (y/4) * (W/4) + x/4, :] = LayerNorm(conv4x4_stride4(x)[t, :, y/4, x/4] + bias)
 * Replaces: PatchEmbed.forward (mask2former/modeling/backbone/swin.py:307-339: `self.proj(x)`, `x.flatten(2).transpose(1, 2)`,
 *           `self.norm(x)`) for in_chans = 3, patch_size = 4 -- tokens come out in the layout the blocks consume.
 *   x [T, 3, H, W] (H, W multiples of 4: the caller pads as PatchEmbed does), weight [E, 3, 4, 4], bias [E] or NULL,
 *   ln_weight / ln_bias [E] or NULL (no norm), out [T, H/4 * W/4, E].  E in {96, 128, 192}.  Plain fp32 FMAs.
 * ------------------------------------------------------------------------------------------- */
int univs_patch_embed4_f32(const float* x, const float* weight, const float* bias, const float* ln_weight, const float* ln_bias, float ln_eps,
                           int T, int H, int W, int E, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Swin PatchMerging's gather + LayerNorm in one pass (layer_norm.hip):
 *   out[b, y2 * W2 + x2, :] = LayerNorm_{4C}( concat( x[b, 2 y2, 2 x2], x[b, 2 y2 + 1, 2 x2], x[b, 2 y2, 2 x2 + 1], x[b, 2 y2 + 1, 2 x2 + 1] ) )
 * with H2 = ceil(H / 2), W2 = ceil(W / 2) and pixels beyond H / W read as zeros (the reference pads to even sizes first).
 * Replaces: PatchMerging.forward up to `self.reduction` (mask2former/modeling/backbone/swin.py:341-386: F.pad, the four strided slices,
 *           torch.cat(..., -1), self.norm).
 *   x [B, H, W, C] (tokens of a stage in image order), gamma / beta [4 C], out [B, H2 * W2, 4 C].  C % 4 == 0, C <= 768.
 * ------------------------------------------------------------------------------------------- */
int univs_patch_merge_norm_f32(const float* x, const float* gamma, const float* beta, int B, int H, int W, int C, float eps, float* out,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * The decoder's cross-attention memory of one feature level from the NCHW feature map, one pass (transpose.hip):
 *   memory[hw][t][c] = x[t][c][hw] + level_embed[c]
 *   key[hw][t][c]    = memory[hw][t][c] + (pos_yx[hw][c] + pos_t[t][c])
 * Replaces: `src = input_proj(x).flatten(2) + level_embed[..., None]` permuted to [hw, bt, C] and `with_pos_embed(memory, pos)`
 *           with the 3-D sine embedding pos = yx + t (univs/modeling/transformer_decoder/...decoder_univs.py:350-355, :400-405;
 *           position_encoding.py:100-160) -- a broadcast add, two permuted copies and a strided add in ATen.
 *   x [T, C, HW]; level_embed [C]; pos_yx [HW, C]; pos_t [T, C]; memory / key [HW, T, C].  C % 4 == 0, HW % 4 == 0.
 * ------------------------------------------------------------------------------------------- */
int univs_decoder_memory_f32(const float* x, const float* level_embed, const float* pos_yx, const float* pos_t, int T, int C, int HW,
                             float* memory, float* key, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Masked multi-head attention core in one pass over the keys (cross_attn.hip):
 *   out[l, n, h, :] = softmax_s(scale * q[l, n, h, :] . k[s, n, h, :]  masked)  v[s, n, h, :]
 * Replaces: inside nn.MultiheadAttention.forward as CrossAttentionLayer calls it
 *           (univs/modeling/transformer_decoder/transformer_layers.py:95-115; ...decoder_univs.py:400-405), between the in- and
 *           out-projections: the scaled score GEMM, masked_fill(attn_mask, -inf), softmax over the keys, and the product with
 *           V -- the [N * H, L, S] scores are never written.
 *   q [L, N, H * head_dim], k / v [S, N, H * head_dim]  sequence-first fp32 (the Linears' outputs as they stand); ldq / ldk / ldv:
 *        floats between consecutive batch entries (0 = H * head_dim, dense) -- a tensor may be a column slice of a wider
 *        projection (the K or V of several decoder layers that attend to the same level, computed by one Linear)
 *   mask [N, L, S] uint8 / bool, non-zero = key masked out for that query, shared by the heads; or NULL
 *        (rows in which every key is masked yield NaN, as nn.MultiheadAttention; the decoder resets such rows beforehand,
 *        ...decoder_univs.py:390 -- univs_mask_decode_attn_f32 does).  When S % 4 != 0 every mask row is PADDED to the next
 *        multiple of four bytes ([N, L, (S + 3) & ~3]; the padding bytes are ignored): the kernel reads a row in aligned dwords
 *   workspace  univs_cross_attention_workspace(L, S, N, H) floats (per-segment partial results; no allocation inside)
 * Arithmetic: both products as three products of two-part fp16 splits with fp32 accumulation (the class of
 * univs_linear_fused_f32), softmax in fp32.  Covered: head_dim == 32, S >= 32, N * H <= 65535 (a row-flagged mask: S % 4 == 0);
 * otherwise UNIVS_ERR_NOT_IMPLEMENTED (the caller keeps GEMM + univs_masked_softmax_f32 + GEMM).
 * ------------------------------------------------------------------------------------------- */
long long univs_cross_attention_workspace(int L, int S, int N, int H);
int univs_cross_attention_f32(const float* q, const float* k, const float* v, const uint8_t* mask, int L, int S, int N, int H, int head_dim,
                              int ldq, int ldk, int ldv, float scale, float* workspace, float* out, void* stream);
/* with the deferred all-masked-row rule of univs_mask_decode_attn_deferred_f32: mask row (n, l) counts only where
 * mask_row_flags[n * L + l] == mask_generation (mask_row_flags NULL: every row counts, = univs_cross_attention_f32) */
int univs_cross_attention_flagged_f32(const float* q, const float* k, const float* v, const uint8_t* mask, const uint32_t* mask_row_flags,
                                      uint32_t mask_generation, int L, int S, int N, int H, int head_dim, int ldq, int ldk, int ldv,
                                      float scale, float* workspace, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Swin window attention core.
 * Replaces: WindowAttention.forward between the qkv and proj linears
 *           (mask2former/modeling/backbone/swin.py:137-168):
 *           softmax((q*scale) @ k^T + rel_pos_bias[h] (+ shift_mask[window % nW])) @ v
 *   qkv        [B_, Ntok, 3, nH, hd]   output of the qkv Linear, B_ = batch * windows (window fastest)
 *   bias       [nH, Ntok, Ntok]        relative position bias already gathered per head (swin.py:148-155)
 *   shift_mask [nW, Ntok, Ntok] or NULL (0 / -100 entries, swin.py:437-440)
 *   out        [B_, Ntok, nH*hd]
 *   hd must be 32 (Swin-T/B/L); Ntok <= 144 (window 12).
 * ------------------------------------------------------------------------------------------- */
int univs_window_attention_f32(const float* qkv, const float* bias, const float* shift_mask,
                               int B_, int nW, int Ntok, int nH, int hd, float scale, float* out,
                               void* stream);

/* Image-mode variant of the window attention core: qkv / out stay in TOKEN order and the kernel performs the
 * reference's pad -> roll(-shift) -> window_partition before and window_reverse -> roll(+shift) -> crop after
 * the attention (mask2former/modeling/backbone/swin.py:252-284) by index arithmetic.
 *   qkv        [B, H*W, 3, nH, hd]   the qkv Linear applied to the (un-padded) tokens
 *   qkv_bias   [3 * nH * hd] or NULL q/k/v of the zero-padded pixels (= the qkv Linear's bias)
 *   bias       [nH, ws*ws, ws*ws]    relative position bias per head
 *   shift_mask [nW, ws*ws, ws*ws] or NULL   (nW = ceil(H/ws) * ceil(W/ws), used when shift > 0)
 *   out        [B, H*W, nH*hd]
 * ------------------------------------------------------------------------------------------- */
int univs_window_attention_image_f32(const float* qkv, const float* qkv_bias, const float* bias,
                                     const float* shift_mask, int B, int H, int W, int ws, int shift,
                                     int nH, int hd, float scale, float* out, void* stream);

/* The same operator with the operand precision of its two matrix products chosen by the caller (BASELINE config 5: the
 * reference runs the backbone under autocast, train_net.py:334, so swin.py:141-168 multiplies in fp16).
 *   UNIVS_MMA_F32  exact fp32 products (v_mfma_f32_16x16x4_f32): identical to univs_window_attention_image_f32
 *   UNIVS_MMA_F16  q*scale, k, v and the un-normalised probabilities exp(s - max) are rounded to fp16 (RNE) as MFMA
 *                  operands (v_mfma_f32_16x16x32_f16 / 16x16x16_f16); accumulation, bias, shift mask, softmax and the
 *                  1/sum normalisation stay fp32, as do the input and output tensors.  ws <= 12, hd = 32,
 *                  B*H*W*3*nH*hd < 2^31.  Tolerance against the fp32 result: tests/test_ops_gpu.py
 *                  (test_window_attention_fp16_operands).
 *   UNIVS_MMA_F16X3  fp32-accurate on the fp16 matrix cores: every operand (q*scale, k, the probabilities, v) as TWO fp16 parts,
 *                  three of the four part products (error <= 2^-21.7 per product, the class of univs_linear_fused_f32), at
 *                  3/16 of the exact-f32 MFMA time.  No scaling is applied: |q*scale*log2(e)|, |k|, |v| < 65504.  Windows up to
 *                  9 x 9; larger windows run the UNIVS_MMA_F32 kernel.
 * Any other value of `mma` returns UNIVS_ERR_INVALID_ARGUMENT. */
#define UNIVS_MMA_F32 0
#define UNIVS_MMA_F16 1
#define UNIVS_MMA_F16X3 2
int univs_window_attention_image_mma(const float* qkv, const float* qkv_bias, const float* bias,
                                     const float* shift_mask, int B, int H, int W, int ws, int shift,
                                     int nH, int hd, float scale, int mma, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The FPN top-down step for an exact 2x up-sampling (resample.hip):
 *   out = (addend [* scale_p + bias_p]) + F.interpolate(in, scale 2, mode="bilinear", align_corners=False)
 * Replaces: `y = cur_fpn + F.interpolate(y, size=cur_fpn.shape[-2:], mode="bilinear", align_corners=False)`
 *           (mask2former/modeling/pixel_decoder/msdeformattn.py:350-351) and, with addend_affine, the GroupNorm of the lateral
 *           convolution in front of it (`cur_fpn = lateral_conv(x)`, a Conv2d with norm = GroupNorm(32), :214-226, :349): addend is
 *           then the raw convolution output and addend_affine [planes][2] the (scale, bias) pairs of univs_group_norm_affine_f32 --
 *           the normalised lateral tensor is never written.  Same taps, weights and expression order as
 *           univs_bilinear_resample_f32 (bit-identical results).
 *   in [planes, Hin, Win], addend / out [planes, 2 Hin, 2 Win]; Win even; in 8-byte, addend / out 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
int univs_upsample2x_add_f32(const float* in, const float* addend, const float* addend_affine, float* out, long long planes, int Hin,
                             int Win, void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm statistics in affine form (group_norm.hip): affine[(n * C + c) * 2 + {0, 1}] = (gamma_c / sqrt(var + eps),
 * beta_c - mean * gamma_c / sqrt(var + eps)) of x [N, C, HW] with `groups` groups -- F.group_norm(x) == x * scale + bias, the values
 * and the expression univs_group_norm_f32 applies.  For consumers that normalise while they read (univs_upsample2x_add_f32).
 *   ws: N * C * 2 * ceil(HW / 8192) floats of scratch (as univs_group_norm_f32).
 * ------------------------------------------------------------------------------------------- */
int univs_group_norm_affine_f32(const float* x, const float* gamma, const float* beta, int N, int C, long long HW, int groups, float eps,
                                float* ws, long long ws_floats, float* affine, void* stream);

/* out[t, c, :H, :W] = (x[t, c] - mean[c]) / std[c], zeros in rows H..Hp-1 and columns W..Wp-1: the caller's pre-step of a clip
 * (univs/inference/inference_video_entity.py:246-250: `self.normalizer(frame)` per frame, then `ImageList.from_tensors(images_norm,
 * self.size_divisibility)`; detectron2 pads after normalising, so the padding is 0 in the normalised domain) as one pass instead of
 * ATen's subtraction, division, fill and strided copy.  x [T, C, H, W] contiguous float32 (pixel values 0..255), mean / std [C]
 * device pointers; arithmetic as ATen's (fp32 subtraction, then a true division): bit-identical to `F.pad((x - mean) / std, ...)`.
 * UNIVS_ERR_NOT_IMPLEMENTED for more than 65535 planes. */
int univs_normalize_pad_f32(const float* x, const float* mean, const float* std, long long T, int C, int H, int W, int Hp, int Wp, float* out,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bilinear resampling of image planes, align_corners = false (PyTorch semantics).
 * Replaces: F.interpolate(x, size=(Hout, Wout), mode="bilinear", align_corners=False) on the path of the
 *           attention-mask heads (univs/modeling/transformer_decoder/
 *           video_mask2former_transformer_decoder_univs.py:555-558; this build resamples the mask
 *           features once per level instead of the mask logits of every layer).
 *           and the FPN top-down step `lateral + F.interpolate(coarser, size=lateral.shape[-2:])`
 *           (mask2former/modeling/pixel_decoder/msdeformattn.py:350-351) when `addend` is given.
 *   in      [planes, Hin, Win]    (planes = T * C, contiguous)
 *   addend  [planes, Hout, Wout] or NULL
 *   out     [planes, Hout, Wout]  = (addend +) resample(in)
 * ------------------------------------------------------------------------------------------- */
int univs_bilinear_resample_f32(const float* in, const float* addend, float* out, long long planes,
                                int Hin, int Win, int Hout, int Wout, void* stream);

/* The three attention-mask resolutions of the decoder in one pass: out2 / out4 / out8 = univs_bilinear_resample_f32(in) to
 * (H/2, W/2), (H/4, W/4), (H/8, W/8) -- bit-identical to the three separate calls, the input is read once
 * (...decoder_univs.py:555-558 resizes to the sizes of the three feature levels, strides 8 / 16 / 32 against mask features
 * at stride 4).  H and W multiples of 8, 16-byte aligned pointers; otherwise UNIVS_ERR_NOT_IMPLEMENTED. */
int univs_bilinear_pyramid3_f32(const float* in, long long planes, int H, int W, float* out2, float* out4, float* out8,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row LayerNorm with an optional fused residual add.
 * Replaces: nn.LayerNorm(C)(x) and nn.LayerNorm(C)(x + residual) on the token tensors of the path --
 *           Swin blocks (mask2former/modeling/backbone/swin.py:236-262: norm1, the residual, norm2),
 *           MSDeformAttn encoder layers (mask2former/modeling/pixel_decoder/msdeformattn.py:61-95:
 *           `src = norm1(src + src2)`, `src = norm2(src + ffn)`), decoder layers (transformer_layers.py).
 *   x, residual (or NULL)   [rows, C]
 *   gamma, beta             [C]
 *   sum_out (or NULL)       [rows, C]  receives x + residual (the un-normalised stream of pre-norm blocks)
 *   out                     [rows, C]  = (s - mean(s)) / sqrt(var(s) + eps) * gamma + beta,  s = x (+ residual)
 *   C % 4 == 0 and C <= 3072, else UNIVS_ERR_NOT_IMPLEMENTED.
 * ------------------------------------------------------------------------------------------- */
int univs_layer_norm_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                         long long rows, int C, float eps, float* sum_out, float* out, void* stream);

/* The same with a second output  out2 = LayerNorm(x + residual) + addend  ([rows, C]; addend [addend_rows, C] is repeated
 * every addend_rows rows, rows % addend_rows == 0: position embeddings [1, S, C] against tokens [N, S, C]): the encoder layer's
 * `src = norm2(src + ffn(src))` followed by the next layer's `with_pos_embed(src, pos)`
 * (mask2former/modeling/pixel_decoder/msdeformattn.py:61-63, :85-95) in one pass.  addend / out2 come together (or both
 * NULL: plain univs_layer_norm_f32), need a residual and exclude sum_out. */
int univs_layer_norm_add_f32(const float* x, const float* residual, const float* gamma, const float* beta, const float* addend,
                             long long addend_rows, long long rows, int C, float eps, float* sum_out, float* out, float* out2,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm (+ optional ReLU) on NCHW tensors.
 * Replaces: detectron2 Conv2d(..., norm=GroupNorm(32, C)[, activation=F.relu]) epilogues of the pixel
 *           decoder (mask2former/modeling/pixel_decoder/msdeformattn.py:214-232 input_proj,
 *           :262-283 lateral_convs / output_convs).
 *   x, out       [N, C, HW]   contiguous
 *   gamma, beta  [C]
 *   ws           workspace of ws_floats floats, at least N * C * 2 * ceil(HW / 8192)
 *   biased variance, y = (x - mean) / sqrt(var + eps) * gamma + beta, then max(y, 0) if relu != 0
 * ------------------------------------------------------------------------------------------- */
int univs_group_norm_f32(const float* x, const float* gamma, const float* beta, int N, int C,
                         long long HW, int groups, float eps, int relu, float* ws, long long ws_floats,
                         float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Masked softmax over the last dimension of attention scores, in place.
 * Replaces: `attn.masked_fill(attn_mask, -inf)` + `softmax(dim=-1)` inside nn.MultiheadAttention as used
 *           by the decoder's cross-attention (univs/modeling/transformer_decoder/transformer_layers.py:
 *           101-105, mask from ...decoder_univs.py:390-405).
 *   scores  [N, h, L, S]
 *   mask    [N, L, S] bytes, non-zero = masked out (broadcast over the h heads), or NULL
 * ------------------------------------------------------------------------------------------- */
int univs_masked_softmax_f32(float* scores, const uint8_t* mask, int N, int h, int L, int S, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ProCA attention: every prompt query attends only to its own prompt tokens (batch = Q_p * T, query length 1, key length 1 + L).
 * Replaces: inside forward_transformer_prompt_self_attention_layer (...decoder_univs.py:456-496 -> CrossAttentionLayer.forward_post,
 *           transformer_layers.py:95-115 -> nn.MultiheadAttention) the concatenation / transposition of the query state and the
 *           dense prompt tokens into `memory` (and of their position embeddings), q k^T, the softmax and p v.
 *   qkv0 [Q_p * T, 3 E]   query, first key and first value of every batch entry b = qp * T + t, already projected
 *                         (q and k0 from state + position, v0 from the state: one univs_small_linear launch)
 *   kd, vd [Q_p, L, T, E] the dense tokens' key / value projections, in the layout the tokens have in the memory pool
 *   out [Q_p * T, E]      softmax(scale q [k0; kd]^T) [v0; vd], heads concatenated (the input of out_proj)
 * Covered: head_dim == 32, (1 + L) * 4 bytes of LDS <= 64 KB; otherwise UNIVS_ERR_NOT_IMPLEMENTED.
 * ------------------------------------------------------------------------------------------- */
int univs_proca_attention_f32(const float* qkv0, const float* kd, const float* vd, int Qp, int L, int T, int heads, int head_dim,
                              float scale, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Visual-prompt sampler of a prompted clip (csrc/prompt_sampler.hip): the device part of VisualPromptEncoder.get_mask_prompt for the
 * F key frames x n entities of a clip at once (entity-frame i = f * n + e).
 * Replaces: univs/modeling/prompt_encoder/prompt_encoder.py:168-263 (get_mask_prompt), :362-442 (select_points_from_box_mask, mask
 *           branch), :445-497 (get_dense_features) and univs/utils/comm.py:6-39 (convert_box_to_mask) -- per key frame ~110 ATen
 *           launches in the reference (boolean indexing, nonzero, randperm on the host, gathers).
 *
 * univs_prompt_prefix_f32: everything that depends on the annotations only.
 *   masks [F n, h, w] float32, boxes [F n, 4] normalised xyxy, scale = h / h_img (the masks' stride over the 1/8 feature map)
 *   stats [2 F n + F] uint32, ZERO on entry (scratch: ordered-float maxima)
 *   -> feat_masks [F n, h_img, w_img] (nearest), sel [F n, h, w] bytes (candidate pixels of the point draw: the central half of the
 *      box where the mask reaches min(max, 0.75); without such a pixel the pixels >= min(max, 0.95)), rowcnt [F n, h] int32 (their
 *      count per image row), fmb [F n, h_img w_img] bytes (feat_masks >= min(max over the frame, mask_thresh)), counts [F, 2 n] int32
 *      (candidates per entity, then feature pixels per entity: the sizes of the reference's two randperm draws), valid / visible
 *      [F n] bytes (max > mask_thresh / max > 0).
 * univs_prompt_draw: the draws -> pixels.  Either (u [F n], keys [F n, HW]) uniform numbers in [0, 1) -- the point's rank =
 *   floor(u count), the R dense pixels = the R largest keys among the mask's feature pixels in decreasing order when it has >= R,
 *   its pixels cyclically when it has fewer -- or tab [F n, R + 2] int64 explicit ranks (R dense ranks, an "empty" flag, the point's
 *   rank: the reference's randperm values); the other one NULL.
 *   -> point_idx [F n] int64 (y w + x), point_coords [F n, 2] ((x + .5) / w, (y + .5) / h), dense_idx [F n, R] int64 (feature-map
 *      pixels), empty [F n] bytes.  UNIVS_ERR_NOT_IMPLEMENTED when the keys of one entity exceed the LDS (HW > ~36 000).
 * univs_prompt_tokens_f32: the dense tokens fd / pd [F n, R, T, C] = features / position embeddings at the sampled pixels of the key
 *   frame's maps (feats / pos [F, C, HW] addressed through element strides {frame, channel, pixel}); the pooled token qfeat / qpe
 *   [F n, C] for an empty mask; zeros for invalid entities; replicated over the T frames -- and the cross-attention masks attn
 *   [F, T, 1, n, HW] bytes: at frame kf[f] everything outside the box of a valid entity, zero elsewhere.
 * ------------------------------------------------------------------------------------------- */
int univs_prompt_prefix_f32(const float* masks, const float* boxes, int F, int n, int h, int w, int scale, float mask_thresh,
                            float* feat_masks, uint32_t* stats, uint8_t* sel, int32_t* rowcnt, uint8_t* fmb, int32_t* counts,
                            uint8_t* valid, uint8_t* visible, void* stream);
int univs_prompt_draw(const uint8_t* sel, const int32_t* rowcnt, const uint8_t* fmb, const int32_t* counts, const float* u,
                      const float* keys, const int64_t* tab, int F, int n, int h, int w, int HW, int R, int64_t* point_idx,
                      int64_t* dense_idx, uint8_t* empty, float* point_coords, void* stream);
int univs_prompt_tokens_f32(const float* feats, const int64_t* feats_strides, const float* pos, const int64_t* pos_strides,
                            const float* qfeat, const float* qpe, const int64_t* dense_idx, const uint8_t* empty, const uint8_t* valid,
                            const float* boxes, const int64_t* kf, int F, int n, int R, int T, int C, int h_img, int w_img, float* fd,
                            float* pd, uint8_t* attn, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Position tokens of sampled points (csrc/prompt_sampler.hip): out[i] = cat(sincos(y_i scale / dim_t), sincos(x_i scale / dim_t)) +
 * sincos(z[i / n] / dim_tz), sincos = (sin on even channels, cos on odd ones).
 * Replaces: PositionEmbeddingSine3D*.forward_points_with_size (univs/modeling/transformer_decoder/position_encoding.py:88-110, :170-236)
 *           as get_mask_prompt calls it (prompt_encoder.py:209-212): ~25 ATen launches per call, the same bits.
 *   xy [F n, 2] normalised (x, y); z [F] the frames' scaled temporal coordinate; dim_t [Fq], dim_tz [2 Fq] the frequency vectors
 *   (temperature^(2 floor(k / 2) / len)); scale = 2 pi; out [F n, 2 Fq].
 * ------------------------------------------------------------------------------------------- */
int univs_prompt_point_pe_f32(const float* xy, const float* z, const float* dim_t, const float* dim_tz, float scale, int F, int n, int Fq,
                              float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-plane statistics of mask logits in one pass: x [planes, H, W] -> out [planes, 8] int32 =
 *   {|{x > t_hi}|, |{x > t_lo}|, left, top, right, bottom of {x > t_box} (inclusive pixel indices; zeros when empty), non-empty, 0}
 * over the valid region rows [0, h_valid) x columns [0, w_valid) of every plane (h_valid <= H, w_valid <= W).
 * Replaces: univs/utils/comm.py:104-112 (calculate_mask_quality_scores: two compares + two sums over the clip's mask logits) and
 *           univs/utils/comm.py:10-38 (convert_mask_to_box: compare, two `any`, four where / min / max passes, stack, product) as the
 *           clip loop calls them on [Q, T, h, w] (univs/inference/inference_video_entity.py:452-470, :566-590): ~25 launches, 5 passes.
 * NaN exceeds no threshold (as in ATen).  planes <= 65535; otherwise UNIVS_ERR_NOT_IMPLEMENTED.
 * ------------------------------------------------------------------------------------------- */
int univs_mask_stats_f32(const float* x, long long planes, int H, int W, int h_valid, int w_valid, float t_hi, float t_lo, float t_box,
                         int32_t* out, void* stream);
/* The same over an [outer, inner, H, W] VIEW with two plane strides (in floats): plane (o, i) starts at x + o * stride_outer + i * stride_inner
 * -- the last T frames of the per-video history `mask_logits[:, -T:]` (inference_video_entity.py:466-468) without a copy of them.  out
 * [outer, inner, 8]. */
int univs_mask_stats_strided_f32(const float* x, long long outer, int inner, long long stride_outer, long long stride_inner, int H, int W,
                                 int h_valid, int w_valid, float t_hi, float t_lo, float t_box, int32_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Mean over the non-blank tokens: x [n, L, T, C] -> out [n, T, C] = sum_l x[:, l] / max(1, #{l : x[., l, ., :] is not all zero}) (+ add [C]).
 * Replaces: univs/modeling/transformer_decoder/video_mask2former_transformer_decoder_univs.py:640-650 (the initial prompt query / its
 *           position: compare, all, not, sum, clamp, sum, divide, add -- eight launches per tensor).  fp32, summed token by token.
 * Covered: C <= 1024, L <= 15360; otherwise UNIVS_ERR_NOT_IMPLEMENTED.
 * ------------------------------------------------------------------------------------------- */
int univs_token_mean_f32(const float* x, const float* add, int n, int L, int T, int C, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIVS_HIP_H */
