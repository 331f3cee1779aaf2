/*
 * combo_avs.h — C ABI of libcombo_avs_hip.so: the MI355X (gfx950) kernels behind COMBO-AVS's
 * fusion + mask-decoding hot path.
 *
 * Drop-in boundary.  The only native interface the reference has is the pybind module
 * `MultiScaleDeformableAttention` (reference: models/modeling/pixel_decoder/ops/src/vision.cpp:18-21,
 * ops/src/ms_deform_attn.h:25-66) whose two functions dispatch to the launchers
 * `ms_deformable_im2col_cuda` / `ms_deformable_col2im_cuda`
 * (ops/src/cuda/ms_deform_im2col_cuda.cuh:928-959, 961-1331).  `combo_msda_*` below replace exactly
 * those launchers.  All other entry points replace chains of ATen ops inside the reference's Python
 * modules; each cites the Python lines it stands for.  INTEGRATION.md shows the ctypes stub that
 * binds them from the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in `_host`; tensors are contiguous,
 *     row-major, in the layout written next to the argument;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous
 *     and never synchronise;
 *   - the return value is a hipError_t as int (0 = hipSuccess); argument errors return
 *     COMBO_EINVAL (= hipErrorInvalidValue, 1).  Unlike the reference, which only printf's launch
 *     failures (ms_deform_im2col_cuda.cuh:953-957), launch errors are returned;
 *   - outputs are caller-allocated; whether they must be zero-filled is stated per function.
 */
#ifndef COMBO_AVS_H_
#define COMBO_AVS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COMBO_EINVAL 1
/* bits of the device LSAP's status word (combo_lsap_small_f32) */
#define COMBO_LSAP_NONFINITE_COST 1
#define COMBO_LSAP_TOO_MANY_TARGETS 2
typedef void* combo_stream_t;

/* Library / device introspection (host only, no GPU needed). */
int combo_abi_version(void);
/* CU budget of the calling host thread's persistent GEMM launches (0 = whole device); returns the previous value.  For callers that run
 * two independent launch chains on two streams (the reference runs its Siam pair of backbones back to back: maskformer_model.py:333-341). */
int combo_set_cu_limit(int n);

/* Measurement aid (bench.py): device-side timing of the instrumented kernels.  HIP refuses event records inside a
 * captured hipGraph on ROCm 7, so the kernels take wall-clock timestamps themselves: `buf` = slots x 256 uint64 on the
 * device (16 sub-slots of 16 words: word 0 = earliest workgroup start, initialised to ~0, word 1 = latest end, initialised
 * to 0; everything else 0); every following instrumented launch takes the next slot (a graph node keeps its slot over all
 * replays); combo_timing_fold (call it once after every step / graph replay, on the stream of the launches) adds
 * end - start in ticks to slot[2] and 1 to slot[3] for every slot that ran and re-arms it.  combo_timing_slot_info returns what the host recorded for a slot: kind (0 MSDeformAttn forward core,
 * 1 fp32-MFMA GEMM, 2 3xbf16 forward/dX GEMM, 3 weight-gradient GEMM, 4/5 decoder attention forward/backward,
 * 6 MSDeformAttn backward) and its algorithmic work per launch (bytes for the HBM-bound kinds 0 and 6, flops
 * 2*M*N*K for the others).  buf == NULL switches timing off.  The reference has no counterpart (it times whole
 * iterations with detectron2's IterationTimer, models/evaluation/evaluator.py:149-228). */
int combo_timing_set_buffer(void* buf, int slots);
int combo_timing_slots_used(void);
int combo_timing_rewind(void);    /* eager (un-captured) steps: before every step, so its i-th launch takes slot i again */
int combo_timing_truncated(void); /* 1: a launch found no free slot - the figures cover the slotted launches only */
int combo_timing_fold(combo_stream_t stream);
int combo_timing_slot_info(int slot, int* kind, double* work);
int combo_timing_slot_bytes(int slot, double* bytes); /* algorithmic HBM bytes per launch of the slot (GEMM kinds; else 0) */
int combo_wall_clock_khz(void);
const char* combo_build_arch(void); /* "gfx950" */

/* Timing events on the launch stream (measurement plumbing for bench.py; the reference has no counterpart - it times
 * whole iterations with detectron2's IterationTimer).  external != 0 records an event-record NODE when `stream` is being
 * captured into a hipGraph, so the pair can be read after any replay.  elapsed_us needs both events completed. */
int combo_event_create(void** event);
int combo_event_record(void* event, combo_stream_t stream, int external);
int combo_event_elapsed_us(void* start, void* stop, float* us);
int combo_event_destroy(void* event);

/* ------------------------------------------------------------------------------------------------
 * Multi-tensor scale-per-output-channel + cast: dst[e] = cast(src[e] * scale[e / inner]) (scale NULL: plain cast) for
 * many contiguous tensors in one launch (tables of 56 in the kernel arguments).  Folds FrozenBN (d2 FrozenBatchNorm2d
 * [d2]) into all convolution weights of a ResNet and casts them to bf16, maps the gradients back, and makes the per-step
 * bf16 parameter copies of PVTv2 - what autocast / torch._foreach_* do with one kernel per tensor.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* src; void* dst; const float* scale;
  long long numel;
  int inner, src_bf16, dst_bf16;
} combo_fold_problem;
int combo_fold_cast_grouped(const combo_fold_problem* problems, int count, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Channel sums out[c] = sum_{a,l} x[a,c,l] of a contiguous [A, C, L] tensor (bf16 or fp32; L == 1: column sums of a
 * [A, C] matrix, C % 4 == 0): the bias gradients of the backbones' linears / convolutions (what autograd's
 * grad_output.sum(...) computes for nn.Linear / nn.Conv2d in pvtv2.py) and the level-embedding gradients (AVFuse.py:104-106,
 * mask2former_transformer_decoder.py:399).  Deterministic (partials per slice, added in a fixed order by a second launch)
 * and free of memset nodes, which a replayed hipGraph does not execute reliably on this stack (csrc/colsum.hip).
 *   combo_colsum_slices: rows of `partial` ([slices, C] fp32 scratch; may be NULL when 1), < 0 on invalid shapes
 * ---------------------------------------------------------------------------------------------- */
int combo_colsum_slices(long long A, int C, long long L);
int combo_colsum(const void* x, long long A, int C, long long L, int in_bf16, void* out, int out_bf16, float* partial,
                 combo_stream_t stream);
/* Many column sums (L == 1) in one launch + one finish launch: the bias gradients of a whole backbone's nn.Linear layers,
 * queued during the backward pass (ops/colsum.py).  partial: [combo_colsum_grouped_slices(rows, C), C] fp32 per problem. */
typedef struct {
  const void* x; void* out; float* partial;
  long long rows;
  int C, in_bf16, out_bf16;
} combo_colsum_problem;
int combo_colsum_grouped_slices(long long rows, int C);
int combo_colsum_grouped(const combo_colsum_problem* problems, int count, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Pre-norm residual step of a PVTv2 block (backbone/pvtv2.py:162-175: x = x + drop_path(branch(norm(x)))), one pass per
 * LayerNorm.  forward: z = x + scale[row / rows_per_sample] * r (x, z fp32; r bf16 or NULL: z = x, not written; scale NULL:
 * 1), y = LN(z) as bf16 (y_bf16 != 0) or fp32, mean / rstd per row.  backward: d = LN'(dy + dy2) + dz (each may be NULL; dy2
 * is the gradient of a second consumer of y),
 * dx = d, dr = bf16(scale * d) (NULL: no branch), dy32 = float(dy) (optional, for the deferred parameter gradients).
 * C in {64, 128, 256, 320, 512}.
 * ---------------------------------------------------------------------------------------------- */
int combo_prenorm_forward(const float* x, const void* r_bf16, const float* scale, long long rows_per_sample, const float* w,
                          const float* b, float eps, long long rows, int C, float* z, void* y, int y_bf16, float* mean,
                          float* rstd, combo_stream_t stream);
int combo_prenorm_backward(const void* dy, const void* dy2, int dy_bf16, const float* dz, const float* z, const float* mean, const float* rstd,
                           const float* w, const float* scale, long long rows_per_sample, long long rows, int C, float* dx,
                           void* dr_bf16, float* dy32, combo_stream_t stream);

/* Channel bias + LayerNorm on bf16 rows: y = LN(x + xb[c]) as bf16; backward dx = LN'(dy) as bf16 (+ dy32 = float(dy) and
 * z32 = x + xb in fp32, both or neither: the operands of the deferred parameter-gradient launch).
 * The key / value path of PVTv2's spatial-reduction attention (backbone/pvtv2.py:104-108, `self.norm(self.sr(x_))` with the
 * convolution's bias folded in here; its gradient is combo_colsum of dx).  xb: bf16 (xb_bf16 != 0) or fp32, may be NULL. */
int combo_bias_ln_bf16_forward(const void* x, const void* xb, int xb_bf16, const float* w, const float* b, float eps,
                               long long rows, int C, void* y, float* mean, float* rstd, combo_stream_t stream);
int combo_bias_ln_bf16_backward(const void* dy, const void* x, const void* xb, int xb_bf16, const float* mean, const float* rstd,
                                const float* w, long long rows, int C, void* dx, float* dy32, float* z32, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backbone epilogue (host-PyTorch ResNets, bf16 NHWC): y <- relu(y + bias[c] (+ residual)) in place, one pass (MIOpen
 * runs conv, bias and ReLU as three kernels); backward dx = dy * (y > 0).  Replaces d2 BottleneckBlock's
 * FrozenBN-affine + relu + residual add ([d2]; FrozenBN folded as in backbone.py).  y/residual/dy/dx: bf16, C % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
int combo_bias_act_bf16(void* y, const float* bias, const void* residual, long long tokens, int C, int relu,
                        combo_stream_t stream);
int combo_relu_grad_bf16(const void* dy, const void* y, long long n, void* dx, combo_stream_t stream);
/* fp32 activations (the reference's S4 recipe: SOLVER.AMP.ENABLED False, configs/avs_s4/R50-AVSS4-SemanticSegmentation.yaml:44-45); C % 4 == 0 */
int combo_bias_act_f32(float* y, const float* bias, const float* residual, long long tokens, int C, int relu, combo_stream_t stream);
/* fp32 variant for the head's Linear+ReLU layers (MLP.forward transformer_decoder.py:216-219, FFN :178-182, encoder FFN). */
int combo_relu_grad_f32(const float* dy, const float* y, long long n, float* dx, combo_stream_t stream);
/*   dx = (dy1 + dy2) * (y > 0): a block output with two consumers that hand their gradients over separately (backbone.py) */
int combo_relu_grad2_f32(const float* dy1, const float* dy2, const float* y, long long n, float* dx, combo_stream_t stream);
/* ... of a block output with three consumers (the last block of a ResNet stage also feeds the head): dx = (dy1 + dy2 + dy3) . [y > 0] */
int combo_relu_grad3_f32(const float* dy1, const float* dy2, const float* dy3, const float* y, long long n, float* dx,
                         combo_stream_t stream);
/* dst [B, H, W, C] = src [B, ceil(H/2), ceil(W/2), C] at the even pixels, zero elsewhere: the input gradient of a 1x1 stride-2
 * convolution (d2 BottleneckBlock's shortcut, built at models/maskformer_model.py:138,145) from the GEMM dY . W over the output
 * tokens; every element of dst is written. */
int combo_expand_stride2_f32(const float* src, float* dst, int B, int H, int W, int C, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * PVTv2 depth-wise 3x3 convolution on token-major bf16 activations (models/modeling/backbone/pvtv2.py:377-388, DWConv:
 * Conv2d(dim, dim, 3, 1, 1, groups=dim) on the [B,H,W,C] token grid), fp32 weights in tap-major layout [9][C].
 *   forward:        y = bias + sum_tap w[tap] * x[. + off(tap)]            (flip = 0)
 *   backward-data:  dx = sum_tap w[8 - tap] * dy[. + off(tap)]             (flip = 1, bias = NULL)
 *   weight/bias gradient: partials [slices][10][C] (taps 0..8, row 9 = bias gradient); the caller sums the slices
 *   (combo_splitk_reduce_f32).  slices from combo_dwconv3x3_wgrad_slices.  C % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
int combo_dwconv3x3_bf16(const void* x, const float* w_tap_major, const float* bias, int B, int H, int W, int C, int flip,
                         void* y, combo_stream_t stream);
/* the finish of the weight gradient: dw[c][tap] = sum_s partials[s][tap][c] (nn.Conv2d's own [C,1,3,3] layout),
 * db[c] = sum_s partials[s][9][c] (db may be NULL) in one launch, fixed order */
int combo_dwconv3x3_wgrad_finish_f32(const float* partials, int slices, int C, float* dw, float* db, combo_stream_t stream);
/*   host-side switch (A/B): 1 = the strip-form weight-gradient kernel (round 5, default), 0 = round 3's per-token form; returns the
 *   previous value; combo_dwconv3x3_wgrad_slices plans for the form that is on */
int combo_dwconv3x3_wgrad_strips(int on);
int combo_dwconv3x3_wgrad_slices(int B, int H, int W, int C);
int combo_dwconv3x3_wgrad_bf16(const void* x, const void* dy, int B, int H, int W, int C, int slices, float* partials,
                               combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a6  MSDeformAttn core op
 *   replaces ms_deform_attn_cuda_forward / _backward (ops/src/cuda/ms_deform_attn_cuda.cu:25-85, 88-157)
 *   out[b,q,m,:] = sum_{l<L,p<P} w[b,q,m,l,p] * bilinear(value_l[b,:,m,:], loc*(W_l,H_l) - 0.5), zero padding.
 *
 *   value          [B,S,M,D]        spatial_shapes [L,2] int64 (H,W)    level_start_index [L] int64
 *   sampling_loc   [B,Lq,M,L,P,2]   attn_weight    [B,Lq,M,L,P]         out [B,Lq,M*D] (fully overwritten)
 *   algo: 0 = auto, 1 = generic (gather from L2), 2 = LDS-staged value slab (D==32, slab must fit LDS)
 *   im2col_step of the reference has no meaning here (the batch is never chunked) and is not a parameter.
 * ---------------------------------------------------------------------------------------------- */
/* 1: the gradients must be zero-filled by the caller (generic path accumulates with atomics, like the reference's
 * at::zeros at ms_deform_attn_cuda.cu:126-128); 0: the LDS kernels overwrite every element. */
int combo_msda_backward_needs_zero(int S, int D, int L, int P, int elem_bytes, int algo);
int combo_msda_forward_f32(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const float* sampling_loc, const float* attn_weight,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           float* out, int algo, combo_stream_t stream);
int combo_msda_forward_f64(const double* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                           const double* sampling_loc, const double* attn_weight,
                           int B, int S, int M, int D, int L, int Lq, int P,
                           double* out, int algo, combo_stream_t stream);

/*   grad_out [B,Lq,M*D];  grad_value [B,S,M,D], grad_sampling_loc [B,Lq,M,L,P,2], grad_attn_weight [B,Lq,M,L,P].
 *   All three gradient buffers MUST be zero-filled by the caller (as the reference does,
 *   ms_deform_attn_cuda.cu:126-128).  LDS path (D == 32, slab fits): grad_value is accumulated in 32-bit fixed point
 *   (bitwise deterministic, ~1e-7 relative to max|grad_out|), no floating-point atomics anywhere.  Generic path:
 *   global float atomics, accumulation order not deterministic, as in the reference. */
int combo_msda_backward_f32(const float* grad_out, const float* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index, const float* sampling_loc, const float* attn_weight,
                            int B, int S, int M, int D, int L, int Lq, int P,
                            float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                            int algo, combo_stream_t stream);
int combo_msda_backward_f64(const double* grad_out, const double* value, const int64_t* spatial_shapes,
                            const int64_t* level_start_index, const double* sampling_loc, const double* attn_weight,
                            int B, int S, int M, int D, int L, int Lq, int P,
                            double* grad_value, double* grad_sampling_loc, double* grad_attn_weight,
                            int algo, combo_stream_t stream);

/*   Fused, windowed backward (csrc/msda_bwd.hip; D == 32, P == 4, fp32): the three gradients of
 *   ms_deform_attn_cuda_backward (ms_deform_attn_cuda.cu:88-157) from ONE launch that needs value, grad_out, sampling_loc
 *   and attn_weight once; workgroup = (frame, head, band of image rows of one level), all 32 channels, grad_value in 2 x 32-bit
 *   fixed point per 64-bit LDS word (bitwise deterministic), every output element written (no zero-fill).  The level
 *   geometry is taken ON THE HOST (host_shapes [L,2] = (H,W) ints, host_start [L]): the band table and the LDS budget depend
 *   on it, and the reference's launcher has it as device tensors only (ms_deform_attn_cuda.cu:72-73).
 *   combo_msda_backward_win_ok: 1 when this geometry is taken (else use combo_msda_backward_f32). */
int combo_msda_backward_win_ok(const int* host_shapes, int L, int P, int D, int elem_bytes);
int combo_msda_backward_win_f32(const float* grad_out, const float* value, const int* host_shapes, const int* host_start,
                                const float* sampling_loc, const float* attn_weight, int B, int S, int M, int D, int L, int Lq,
                                int P, float* grad_value, float* grad_sampling_loc, float* grad_attn_weight,
                                combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a2  GroupNorm (+ ReLU) on channels_last / token-major fp32 maps [B, HW, C] (the norm + activation of detectron2's Conv2d
 *   wrapper [d2] as used at msdeformattn.py:215-224 (input_proj) and :271-286 (FPN lateral / output convolutions)).
 *   forward:  y = relu?(GN(x)); mean, rstd [B,G] are saved for backward; part_ws [B*slices*C*2] floats with
 *             slices = combo_groupnorm_nhwc_slices(HW).
 *   backward: dx, dgamma, dbeta from dy (masked by y > 0 when relu); part_ws as above, s12_ws [B*G*2 + B*C*2].  C <= 1024.
 * ---------------------------------------------------------------------------------------------- */
int combo_groupnorm_nhwc_slices(int HW);
int combo_groupnorm_nhwc_forward_f32(const float* x, const float* gamma, const float* beta, int B, int HW, int C, int G,
                                     float eps, int relu, float* part_ws, float* mean, float* rstd, float* y,
                                     combo_stream_t stream);
int combo_groupnorm_nhwc_backward_f32(const float* dy, const float* x, const float* y, const float* mean, const float* rstd,
                                      const float* gamma, int B, int HW, int C, int G, int relu, float* part_ws, float* s12_ws,
                                      float* dx, float* dgamma, float* dbeta, combo_stream_t stream);

/* a2  FPN top-down step: bilinear 2x upsampling, align_corners = False, channels_last fp32 [B,H,W,C] -> [B,2H,2W,C]
 *   (msdeformattn.py:349-350); backward as a gather (H, W are the INPUT sizes in both calls).  C % 4 == 0. */
/*   x_batch_stride / dx_batch_stride (floats): the small map may be a per-level row block of the encoder memory [B, S, C]. */
int combo_upsample2x_bilinear_nhwc_f32(const float* x, long long x_batch_stride, int B, int H, int W, int C, float* y,
                                       combo_stream_t stream);
int combo_upsample2x_bilinear_nhwc_backward_f32(const float* dy, int B, int H, int W, int C, float* dx,
                                                long long dx_batch_stride, combo_stream_t stream);
/*   the whole step `y = cur_fpn + F.interpolate(out[-1], ...)` (msdeformattn.py:350) in one pass: add [B,2H,2W,C] (nullable),
 *   bitwise the two-pass result (the interpolated value is rounded to fp32, then added) */
int combo_upsample2x_bilinear_add_nhwc_f32(const float* x, long long x_batch_stride, const float* add, int B, int H, int W, int C,
                                           float* y, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a5  MSDeformAttn prologue (ops/modules/ms_deform_attn.py:101-118)
 *   proj [tokens, M*L*P*3] = [sampling offsets (M,L,P,2) | attention logits (M,L*P)] (the two nn.Linear outputs, merged),
 *   ref [B or 1, Lq, L, 2] reference points (ref_batch_stride = Lq*L*2 or 0), normalizer [L,2] = (W_l, H_l)
 *   -> sampling_loc [tokens,M,L,P,2] = ref + offset / normalizer, attn_weight [tokens,M,L,P] = softmax over L*P.
 *   backward: d_proj from (d_loc, d_attn, attn).  tokens = B*Lq, L*P <= 16.
 * ---------------------------------------------------------------------------------------------- */
int combo_msda_prep_forward_f32(const float* proj, const float* ref, const float* normalizer, long long tokens, int Lq,
                                int M, int L, int P, int ref_batch_stride, float* loc, float* attn, combo_stream_t stream);
int combo_msda_prep_backward_f32(const float* dloc, const float* dattn, const float* attn, const float* normalizer,
                                 long long tokens, int M, int L, int P, float* dproj, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a8-a9  bilateral audio-visual fusion, token stage (C == 256 channels, 8 heads)
 *   replaces, for one fused level, LayerNorm_v + BiMultiHeadAttention's visual-side chain + the layer-scale residual
 *   (fusion_module/utils/fuse_helper.py:155-237 and :320-332) in its algebraically collapsed form:
 *     xn = LN(x); s[h,i] = (xn_i+pos_i).u[b,h]+c[b,h] (clamped +-5e4); p = softmax_i(s);
 *     y_i = xn_i + gamma_v*(sum_h p[h,i]*dropv[h,i]*z[b,h] + b_ov); pooled[b,h] = sum_i p[h,i]*dropa[h,i]*xn_i; spa = sum_i p*dropa
 *   x,y [B,N,C] token-major; pos [N,C]; u,z [B,8,C]; c [B,8]; scores [B,8,N] and stat [B,8,2] are saved for backward;
 *   drop_v/drop_a: optional injected multipliers [B,8,N] (NULL -> in-kernel Philox with (p_drop, seed), p_drop = 0: none).
 *   seed_step: optional DEVICE pointer to a 64-bit step counter mixed into the Philox key inside the kernels, so that a
 *   captured hipGraph (kernel arguments frozen) still draws fresh dropout masks on every replay; NULL -> key = seed.
 *   Workspaces, with chunks = combo_bifuse_chunks(B,N): part_ws [B,chunks,8,2], pooled_part [B,chunks,8,C],
 *   spa_part [B,chunks,8] (the caller sums the *_part buffers over `chunks`).
 *   backward1 -> dp [B,8,N], r_part [B,chunks,8], dz_part [B,chunks,8,C], dgb_part [B,chunks,2,C] (d gamma_v, d b_ov)
 *   backward2 (needs rtot[B,8] = sum_chunks r_part) -> dx [B,N,C], du_part [B,chunks,8,C], dc_part [B,chunks,8],
 *   dln_part [B,chunks,2,C] (d ln_weight, d ln_bias).
 * ---------------------------------------------------------------------------------------------- */
int combo_bifuse_chunks(int B, int N);
int combo_bifuse_forward_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* pos,
                             const float* u, const float* c, const float* z, const float* b_ov, const float* gamma_v,
                             const float* drop_v, const float* drop_a, float p_drop, unsigned long long seed, const unsigned long long* seed_step,
                             int B, int N, int C, int heads, float* y, float* scores, float* stat, float* part_ws,
                             float* pooled_part, float* spa_part, combo_stream_t stream);
int combo_bifuse_backward1_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* scores,
                               const float* stat, const float* z, const float* b_ov, const float* gamma_v,
                               const float* drop_v, const float* drop_a, float p_drop, unsigned long long seed, const unsigned long long* seed_step,
                               const float* dy, const float* dpooled, const float* dspa, int B, int N, int C, int heads,
                               float* dp, float* r_part, float* dz_part, float* dgb_part, combo_stream_t stream);
int combo_bifuse_backward2_f32(const float* x, const float* ln_w, const float* ln_b, float eps, const float* pos,
                               const float* scores, const float* stat, const float* u, const float* drop_a,
                               float p_drop, unsigned long long seed, const unsigned long long* seed_step, const float* dy, const float* dpooled,
                               const float* dp, const float* rtot, int B, int N, int C, int heads, float* dx,
                               float* du_part, float* dc_part, float* dln_part, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a4/a5/a11/a12  dense layers of the head
 *   replace the nn.Linear calls of the encoder / decoder (pixel_decoder/msdeformattn.py:119-134,
 *   ops/modules/ms_deform_attn.py:102-108,128, transformer_decoder.py:99-118,50-58,178-182,216-219), the 1x1 / 3x3
 *   convolutions of the pixel decoder (msdeformattn.py:215-224, 271-286) and their backward GEMMs:
 *   forward in exact fp32 (combo_gemm_nt_f32 ...), gradients with the 3-product bf16 split (combo_gemm_nt_x3_* for dX,
 *   combo_gemm_tn_x3_* for dW).
 * ---------------------------------------------------------------------------------------------- */
/*   FORWARD dense layers in exact fp32 on the matrix cores (csrc/gemm_f32.hip, v_mfma_f32_32x32x2_f32: f32 in, f32
 *   accumulate, one rounding per product): C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (+ ReLU), both operands K-contiguous
 *   (A = tokens, B = the nn.Linear weight as stored).  Every forward nn.Linear / 1x1 convolution of the head runs here
 *   (same reference lines as the block above): their outputs end in the decoder's `sigmoid(logit) < 0.5` attention masks
 *   (transformer_decoder.py:502-507), where the 2^-17 relative error of the 3-product split flips near-zero cells.
 *   K % 16 == 0, lda/ldb % 4 == 0, A and B 16-byte aligned, C addressed with 32-bit byte offsets.
 *   _batched: `batch` problems of one shape, operand b at base + b*stride (elements): the mask-logit contraction
 *   einsum("bqc,bchw->bqhw") of every prediction head (transformer_decoder.py:498-500), A = mask_embed [BT,Q,C],
 *   B = token-major mask features [BT,HW,C].
 *   combo_conv3x3_nhwc_f32: the FPN output convolution (msdeformattn.py:281-286, 349-352) as an implicit GEMM on the
 *   same loop (layout contract: see combo_conv3x3_nhwc_x3_pre_f32 below). */
int combo_gemm_nt_f32(const float* A, long long lda, const float* B, long long ldb, const float* bias, float* C,
                      long long ldc, int M, int N, int K, int relu, combo_stream_t stream);
int combo_gemm_nt_batched_f32(const float* A, long long lda, long long sA, const float* B, long long ldb, long long sB,
                              float* C, long long ldc, long long sC, int M, int N, int K, int batch, int relu,
                              combo_stream_t stream);
/*   Split-K for long reductions with few output tiles (decoder FFN linear2 4000 x 2048 -> 256, the res5 / res4 input
 *   projections): combo_gemm_nt_splitk_plan -> number of K slices (1: do not split); combo_gemm_nt_splitk_f32 runs the slices
 *   as the batch entries of one launch into workspace [splits, M, N] and finishes with a fixed-order sum + bias + ReLU. */
/*   bf16 products per fp32 multiply-add of the combo_gemm_nt_x3_* / combo_conv3x3_nhwc_x3_* launches that follow: 3 (default)
 *   = the fp32-accurate split, 1 = plain bf16 inputs, fp32 accumulation (the head's bf16 throughput mode), 19 = the 3-product
 *   split on fp16 hi / lo pieces (22 mantissa bits instead of 16: the fp32-grade FORWARD mode; operands |x| < 65 504; the weight
 *   image must come from a pre-split issued under combo_presplit_pieces(1)).  Returns the previous value. */
int combo_gemm_nt2_products(int products);
/*   Piece type written by the combo_presplit_bf16x2_* launches that follow: 0 = bf16 (default), 1 = fp16.  Returns the previous value. */
int combo_presplit_pieces(int f16);
int combo_gemm_nt_x3_splitk_plan(int M, int N, int K);  /* K slices of a few-tile, long-K input-gradient GEMM (1 = none) */
int combo_gemm_nt_x3_splitk_f32(const float* A, long long lda, const float* Bimg, const float* mask, float* C, long long ldc,
                                int M, int N, int K, int splits, float* workspace /* [splits, M, N] */, combo_stream_t stream);
int combo_gemm_nt_x3_tile(int cfg);  /* 0 planner (default), 1 / 2 / 3 / 4: force 256x128 / 128x128 / 64x64 / 256x64 tiles (tests, tools); returns the previous value */
int combo_gemm_nt_x3_prof_buffer(unsigned long long* buf); /* COMBO_NT3_DBG=128: device buffer (256 x 8 x 4 u64) for the per-wave cycle sums of a stage's segments (wait + barrier | phase 0 | phase 1 | stages); NULL = off (tools/prof_nt3_stage.py) */
int combo_gemm_nt_splitk_plan(int M, int N, int K);
int combo_gemm_nt_splitk_f32(const float* A, long long lda, const float* B, long long ldb, const float* bias, float* C,
                             long long ldc, int M, int N, int K, int relu, int splits, float* workspace, combo_stream_t stream);
int combo_conv3x3_nhwc_f32(const float* X, long long ldx, const float* Wm, const float* bias, float* Y, long long ldy,
                           int B, int H, int W, int Cin, int Cout, int relu, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a13  prediction heads: `mask_embed @ pixel_embed` -> sigmoid < 0.5 -> attention mask, FUSED (csrc/maskbits.hip)
 *   replaces, per prediction head (transformer_decoder.py:498-507 + the row reset at :458): the full-resolution einsum, its
 *   bilinear down-interpolation to the next layer's memory size, sigmoid < 0.5, the 8x head replication and the
 *   nonzero()/index_put row reset.  Interpolation and contraction commute: the mask is computed from the DOWNSAMPLED pixel
 *   embedding (combo_downsample_tokens_f32, once per forward and level) by combo_mask_bits_f32, whose fp32-MFMA result tile
 *   is turned into the bit-packed mask words by wave ballots - the scores are never written.  The full-resolution logits
 *   (needed by the losses only) come from ONE launch for all heads, combo_mask_logits_all_f32.
 *     x [B, H*W, C] tokens -> out [B, h*w, C] (bilinear, align_corners = False); C % 4 == 0
 *     mask_embed [B, Q, 256], mfd [B, hw, 256] -> bits [B, Q, wpitch] (bit k of word j = cell 32 j + k blocked; cells >= hw
 *       blocked; fully blocked rows un-blocked when reset_full_rows), bytes (nullable) [B, Q, pitch]
 *     mask_embed [heads, B, Q, C], mask_features [B, HW, C] -> out [heads, B, Q, HW]
 * ---------------------------------------------------------------------------------------------- */
int combo_downsample_tokens_f32(const float* x, int B, int H, int W, int h, int w, int C, float* out, combo_stream_t stream);
int combo_mask_bits_f32(const float* mask_embed, const float* mfd, int B, int Q, int hw, int C, int reset_full_rows, int wpitch,
                        unsigned* bits, int pitch, unsigned char* bytes, combo_stream_t stream);
int combo_mask_logits_all_f32(const float* mask_embed, const float* mask_features, float* out, int heads, int B, int Q, int HW,
                              int C, combo_stream_t stream);

/*   a10  audio_mlp (models/modeling/misc/audio_transformation.py:5-14: 128 -> 4096 -> 4096 -> 256 with ReLU on the BT fused
 *   audio tokens): weight-streaming forward GEMM for M <= 64 rows, exact fp32 (csrc/gemm_smallm.hip).  Y[M,N] = X[M,K] .
 *   W[N,K]^T (+ bias) (+ ReLU); splits = combo_gemm_smallm_splits(M, N, K) K-splits (partial_ws: [splits, M, N] floats when
 *   splits > 1, finished in a fixed order by a second small launch).  K % (64 * splits) == 0, ldx / ldw % 4 == 0. */
int combo_gemm_smallm_splits(int M, int N, int K);
int combo_gemm_smallm_f32(const float* X, long long ldx, const float* W, long long ldw, const float* bias, float* Y, long long ldy,
                          float* partial_ws, int splits, int M, int N, int K, int relu, combo_stream_t stream);
/*   C[M,N] = A[M,K] . W[K,N] for K <= 16, N % 4 == 0 (rows of W and C 16-byte aligned): the input gradient of class_embed
 *   (transformer_decoder.py:495, K = classes + 1), exact fp32. */
int combo_gemm_smallk_f32(const float* A, long long lda, const float* W, long long ldw, float* C, long long ldc, long long M,
                          int N, int K, combo_stream_t stream);
/*   dW[N,K] = dY[M,N]^T . X[M,K] for N <= 16 (class_embed's weight gradient), exact fp32: split-K partials
 *   [combo_gemm_tn_smalln_slices(M)][N][K] (+ db_partials [slices][N] = per-slice column sums of dY, or NULL), to be summed by
 *   combo_splitk_reduce_f32. */
int combo_gemm_tn_smalln_slices(long long M);
int combo_gemm_tn_smalln_f32(const float* dY, long long ldy, const float* X, long long ldx, long long M, int N, int K,
                             float* partials, float* db_partials, combo_stream_t stream);

/*   Input-gradient GEMM C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (+ ReLU) with the 3-product bf16 split (x.w ~ hi.hi +
 *   hi.lo + lo.hi, hi = rne_bf16, ~2^-17 relative per product) on the bf16 matrix cores (csrc/gemm_nt3.hip): persistent
 *   workgroups with the next tile's first stages in flight under the epilogue stores, LDS-DMA ring with a source-side chunk
 *   swizzle, and the weight operand PRE-SPLIT into bf16 hi/lo groups by combo_presplit_bf16x2_f32 (element (n, k) =
 *   src[n*ld_row + k*ld_col], so for dX = dY . W the image of W^T needs no transpose copy; the image has N rows of K
 *   floats, 16-byte aligned, K % 8 == 0).  K % 16 == 0, lda % 4 == 0, A and the image 16-byte aligned. */
int combo_presplit_bf16x2_f32(const float* src, long long ld_row, long long ld_col, int N, int K, float* img,
                              combo_stream_t stream);
int combo_gemm_nt_x3_pre_f32(const float* A, long long lda, const float* Bimg, const float* bias, float* C, long long ldc,
                             int M, int N, int K, int relu, combo_stream_t stream);
/*   ... with split-K (splits from combo_gemm_nt_x3_splitk_plan; workspace [splits, M, N]): long reductions with few output tiles in the
 *   3-product forward modes (the decoder FFN's linear2, transformer_decoder.py:178-182; the res5 / res4 input projections) */
int combo_gemm_nt_x3_pre_splitk_f32(const float* A, long long lda, const float* Bimg, const float* bias, float* C, long long ldc,
                                    int M, int N, int K, int relu, int splits, float* workspace, combo_stream_t stream);
/*   combo_gemm_nt_x3_pre_f32 with the ReLU backward of the consumer folded into the epilogue: C = mask > 0 ? A.B^T : 0,
 *   mask of the shape and row pitch of C.  The input-gradient GEMM of an FFN's second layer, dH = (dY . W2) o [H > 0]
 *   (pixel_decoder/msdeformattn.py:125-134, transformer_decoder.py:178-182): the ReLU-gradient pass over the 1024- /
 *   2048-wide hidden tensor disappears. */
int combo_gemm_nt_x3_pre_masked_f32(const float* A, long long lda, const float* Bimg, const float* mask, float* C,
                                    long long ldc, int M, int N, int K, combo_stream_t stream);
/*   Batched forms: `batch` problems of one shape, operand b at base + b*stride (elements).  They carry the mask-logit
 *   contraction `einsum("bqc,bchw->bqhw", mask_embed, mask_features)` of every prediction head
 *   (transformer_decoder/transformer_decoder.py:498-500: A = mask_embed [BT,Q,C], B image = token-major mask features
 *   [BT,HW,C]) and its gradient w.r.t. mask_embed (A = dLogits [BT,heads*Q,HW], B image = mask features transposed per
 *   frame, made by the batched pre-split from a strided view).  img of the pre-split: [batch][N][K] floats. */
int combo_presplit_bf16x2_batched_f32(const float* src, long long ld_row, long long ld_col, long long batch_stride, int N,
                                      int K, int batch, float* img, combo_stream_t stream);
/*   Grouped pre-split: element (n, k) of problem i = src[n*ld_row + k*ld_col] lands in image row n (img_ld floats per row,
 *   >= K, a multiple of 8; img 32-byte aligned - an offset of k0 floats into a wider image concatenates sources along k).
 *   One launch per 56 problems: all weights whose input-gradient GEMMs the backward pass of a step runs. */
/*   taps > 1 (9: a 3x3 convolution weight): element (n, k, tap) = src[n*ld_row + k*ld_col + tap] (the taps of one (n, k) pair
 *   are contiguous: an NCHW-ordered [Cout, Cin, 3, 3] tensor read as n = cout, k = cin for the forward image or n = cin, k = cout
 *   for the input-gradient image) lands in image row n at column (flip ? taps - 1 - tap : tap) * K + k; img_ld >= taps * K.
 *   One thread reads the taps x 8 values of a (row, 8-k group): contiguous runs of the source either way. */
typedef struct {
  const float* src; float* img;
  long long ld_row, ld_col, img_ld;
  int N, K;
  int taps, flip;  /* 0 / 1 taps: a plain matrix */
} combo_presplit_problem;
int combo_presplit_bf16x2_grouped_f32(const combo_presplit_problem* problems, int count, combo_stream_t stream);
int combo_gemm_nt_x3_pre_batched_f32(const float* A, long long lda, long long sA, const float* Bimg, long long sB, float* C,
                                     long long ldc, long long sC, int M, int N, int K, int batch, int relu,
                                     combo_stream_t stream);
/*   3x3 / stride 1 / pad 1 convolution on an NHWC fp32 map as an implicit GEMM (the FPN output convolution `layer_1` of
 *   the reference's pixel decoder, pixel_decoder/msdeformattn.py:281-286,349-352, which the reference runs through cuDNN):
 *   X = [B*H*W tokens, Cin] (row stride ldx), weight as [Cout, 3, 3, Cin] (K = 9*Cin contiguous), Y = [B*H*W, Cout] (row
 *   stride ldy) (+ bias[Cout]) (+ ReLU).  combo_conv3x3_nhwc_f32 (above): forward, exact fp32, Wm = that weight matrix.
 *   combo_conv3x3_nhwc_x3_pre_f32: the input gradient - the same call on dY with the weight as [Cin, 3', 3', Cout] (taps
 *   flipped), 3-product split, Wimg = combo_presplit_bf16x2_f32 of that matrix.  Cin % 16 == 0, ldx % 4 == 0. */
int combo_conv3x3_nhwc_x3_pre_f32(const float* X, long long ldx, const float* Wimg, const float* bias, float* Y,
                                  long long ldy, int B, int H, int W, int Cin, int Cout, int relu, combo_stream_t stream);

/*   The same two kernels with the bottleneck-block epilogue: the FORWARD pass of the ResNet-50 backbones' stride-1 convolutions
 *   (detectron2 BottleneckBlock / FrozenBatchNorm2d - not under /root/reference, built at models/maskformer_model.py:138,145;
 *   SURVEY section 8 row f2), FrozenBN folded into the weights: v = acc (+ bias[n]) (+ aux[m, n] when aux_mode == 1: the
 *   identity / shortcut branch) -> ReLU when relu -> v = aux[m, n] > 0 ? v : 0 when aux_mode == 2 (input gradients: the ReLU
 *   gradient of the layer that produced the operand); aux has the output's pitch; aux_mode 0: aux must be NULL.
 *   splits > 1: K slices (a convolution: kernel rows / taps) as the batch entries of one launch into workspace [splits, M, N],
 *   the epilogue rides in the fixed-order finishing sum.  Plans: combo_gemm_nt_x3_splitk_plan, combo_conv3x3_x3_splitk_plan
 *   (1, 3 or 9). */
int combo_gemm_nt_x3_epi_f32(const float* A, long long lda, const float* Bimg, const float* bias, const float* aux, int aux_mode,
                             float* C, long long ldc, int M, int N, int K, int relu, int splits, float* workspace,
                             combo_stream_t stream);
/*   ... with both auxiliary tensors: v = acc (+ bias) (+ add[m, n]) -> ReLU -> mask[m, n] > 0 ? v : 0 (either may be NULL): the input
 *   gradient of a bottleneck block's first convolution takes the identity branch's gradient (add) and the ReLU gradient of the
 *   block input (mask) in its epilogue - no separate add + ReLU-gradient pass over the block input's gradient. */
int combo_gemm_nt_x3_epi2_f32(const float* A, long long lda, const float* Bimg, const float* bias, const float* add, const float* mask,
                              float* C, long long ldc, int M, int N, int K, int relu, int splits, float* workspace,
                              combo_stream_t stream);
int combo_conv3x3_x3_splitk_plan(long long M, int Cout, int Cin);
int combo_conv3x3_nhwc_x3_epi_f32(const float* X, long long ldx, const float* Wimg, const float* bias, const float* aux,
                                  int aux_mode, float* Y, long long ldy, int B, int H, int W, int Cin, int Cout, int relu,
                                  int splits, float* workspace, combo_stream_t stream);
/*   General form: ksize 3 (zero padding 1) or 1 (padding 0), stride 1 or 2 (output map ceil(H / 2) x ceil(W / 2); Y / aux rows
 *   and the split plan count OUTPUT tokens): the forward pass of the stride-2 3x3 and shortcut convolutions of a ResNet stage's
 *   first block (the library's kernels for them accumulate with atomics: not reproducible from run to run). */
int combo_conv_nhwc_x3_epi_f32(const float* X, long long ldx, const float* Wimg, const float* bias, const float* aux, int aux_mode,
                               float* Y, long long ldy, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                               int splits, float* workspace, combo_stream_t stream);

/*   Weight gradient of that convolution, split-K over the tokens like combo_gemm_tn_x3_f32: partial z is written at
 *   out_partials + z*Cout*9*Cin in [Cout, 3, 3, Cin] order; finish with combo_splitk_reduce_f32.  `splits` as for
 *   combo_gemm_tn_x3_f32 with (M, N, K) = (B*H*W, Cout, 9*Cin).  Cin % 128 == 0, Cout % 4 == 0, Cout >= 64,
 *   B*H*W*max(H,W) < 2^32. */
int combo_conv3x3_wgrad_x3_f32(const float* dY, long long ldy, const float* X, long long ldx, float* out_partials, int B,
                               int H, int W, int Cin, int Cout, int splits, combo_stream_t stream);
/* (round 5) the same implicit TN GEMM for kernel size 1 or 3 (padding ksize / 2), stride 1 or 2 (output map ceil(Hin / 2) x ceil(Win / 2)) and
 * any Cin % 4 == 0: the weight gradients of the ResNet backbones' 64-channel 3x3 layers and of the stride-2 3x3 / shortcut layers
 * (d2 BottleneckBlock, built at models/maskformer_model.py:138,145).  dY rows = output tokens, X rows = input tokens. */
int combo_conv_wgrad_x3_f32(const float* dY, long long ldy, const float* X, long long ldx, float* out_partials, int B, int Hin, int Win,
                            int Cin, int Cout, int ksize, int stride, int splits, combo_stream_t stream);

/*   Weight gradient dW[N,K] = dY[M,N]^T . X[M,K] (reduction over the M tokens), same 3-way bf16 split, fragments loaded
 *   straight from global memory (no LDS), split-K over M: partial z is written at out_partials + z*N*K and the caller
 *   sums the partials.  `splits` must be a value for which ceil(M / roundup16(ceil(M/splits))) == splits
 *   (combo_gemm_tn_splits returns a suitable first guess; the Python binding fixes the rounding). */
int combo_gemm_tn_splits(int M, int N, int K);
/*   db_partials (optional, [splits,N]): per-split column sums of dY = the bias gradient, fused into the same pass. */
int combo_gemm_tn_x3_f32(const float* dY, long long ldy, const float* X, long long ldx, float* out_partials,
                         float* db_partials, int M, int N, int K, int splits, combo_stream_t stream);
/*   Grouped variants: many independent problems in one launch (the problem tables travel in the kernel arguments, in
 *   chunks of 40).  Used for the decoder's weight gradients, which are latency-bound one by one and off the backward
 *   critical path (ops/linear.py defers them to the end of the backward pass).  Every problem must satisfy the conditions
 *   of the LDS-DMA kernel (N, K multiples of 4 and >= 64, M >= 256, 16-byte aligned rows); `splits` as for
 *   combo_gemm_tn_x3_f32, partials [splits][N][K], db_partials [splits][N] or NULL. */
typedef struct {
  const float* dY; const float* X; float* partials; float* db_partials;
  long long ldy, ldx;
  int M, N, K, splits;
} combo_gemm_tn_problem;
typedef struct {
  const float* partials; float* out; const float* db_partials; float* db;
  long long n;
  int splits, nb;
} combo_reduce_problem;
int combo_gemm_tn_x3_grouped_f32(const combo_gemm_tn_problem* problems, int count, combo_stream_t stream);
int combo_splitk_reduce_grouped_f32(const combo_reduce_problem* problems, int count, combo_stream_t stream);

/*   Grouped LayerNorm parameter gradients (deferred like the weight gradients above): per problem
 *   partials[slice][0][c] = sum_t dy*(x-mean)*rstd, partials[slice][1][c] = sum_t dy over the tokens of the slice
 *   (slices = ceil(tokens / tokens_per_slice)); the caller sums the slices (combo_splitk_reduce_grouped_f32, n = 2*C).
 *   Replaces ATen's per-layer cuComputePartGradGammaBeta + cuComputeGradGammaBeta for nn.LayerNorm
 *   (transformer_decoder.py / msdeformattn.py post-norm layers). */
typedef struct {
  const float* dy; const float* x; const float* mean; const float* rstd; float* partials;
  long long tokens;
  int C, tokens_per_slice;
} combo_ln_grad_problem;
int combo_ln_param_grad_grouped_f32(const combo_ln_grad_problem* problems, int count, combo_stream_t stream);

/*   Residual add + LayerNorm in one pass (csrc/layernorm.hip): z = x + r (r NULL: z = x, not written when z is NULL),
 *   y = (z - mean) * rstd * w + b over the last dimension C in {64, 128, 256, 320, 512}; mean / rstd [rows] are saved for the backward
 *   pass, which returns dz = d(loss)/dz (the gradient of BOTH x and r).  Replaces `self.norm(tgt + tgt2)` of the post-norm
 *   layers (transformer_decoder/transformer_decoder.py:99-118, 50-58, 178-182; pixel_decoder/msdeformattn.py:119-134) and
 *   decoder_norm (:494).  All tensors contiguous [rows, C], 16-byte aligned.
 *   Fan-out (round 3): pos [pos_rows, C] + yp (both or neither): yp = y + pos[row % pos_rows] is written as well (the next
 *   block's query `src + pos` / `tgt + query_pos`); backward: dy2 / dy3 / dy4 (nullable) are further gradients of the SAME
 *   output (its consumers hold aliases), summed on the fly; dy_sum (nullable) receives the sum (for the parameter gradients). */
int combo_add_layernorm_forward_f32(const float* x, const float* r, const float* w, const float* b, float eps, long long rows,
                                    int C, float* z, float* y, float* mean, float* rstd, const float* pos, long long pos_rows,
                                    float* yp, combo_stream_t stream);
int combo_layernorm_backward_f32(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                                 long long rows, int C, float* dz, const float* dy2, const float* dy3, const float* dy4,
                                 float* dy_sum, combo_stream_t stream);

/*   Finishes a split-K result in ONE launch: out[i] = sum_z partials[z*n + i] (n % 4 == 0, 16-byte aligned) and, when
 *   nb > 0, db[j] = sum_z db_partials[z*nb + j]; fixed summation order.  `out` may be a row block of a larger matrix
 *   (nn.MultiheadAttention's packed in_proj_weight gradient). */
int combo_splitk_reduce_f32(const float* partials, int splits, long long n, float* out, const float* db_partials, int nb,
                            float* db, combo_stream_t stream);
/*   ... of a 3x3 convolution's weight gradient (combo_conv3x3_wgrad_x3_f32 partials, [splits][Cout][taps][Cin]) written in the
 *   parameter's own NCHW order [Cout][Cin][taps]: no permute copy between the reduce and autograd. */
int combo_splitk_reduce_nchw_f32(const float* partials, int splits, int Cout, int taps, int Cin, float* out, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * f2  Spatial-reduction attention of the PVTv2 backbone (models/modeling/backbone/pvtv2.py:60-132, linear = False):
 *   softmax(q k^T * scale) v per (frame, head), head dimension 64, FEW keys (the token grid reduced by the strided sr x sr
 *   convolution: 49 keys at 224 x 224, 256 at 512 x 512; <= 256 here), bf16 operands, fp32 accumulation, no mask, no dropout
 *   (attn_drop = 0 in every shipped config).  Replaces F.scaled_dot_product_attention of the host-PyTorch backbone, reading the
 *   projections' outputs where they lie:
 *     q   [B, N, H*64] bf16    kv [B, Nk, 2, H, 64] bf16 (k = [:, :, 0], v = [:, :, 1])    out [B, N, H*64] bf16
 *     lse2 [B, H, Npad] fp32 (Npad = N rounded up to 32): log2 of the softmax denominator of the scaled scores (base 2)
 *   backward: dq [B, N, H*64], dkv [B, Nk, 2, H, 64] (the kv projection's gradient in one tensor); delta [B, H, Npad] and
 *   part [combo_sra_attention_backward_workspace floats] are workspaces.  No atomics: bitwise reproducible.
 *   combo_sra_attention_ok: 1 when the geometry is taken (else the caller keeps its own path).
 * ---------------------------------------------------------------------------------------------- */
int combo_sra_attention_ok(int N, int Nk, int head_dim);
long long combo_sra_attention_backward_workspace(int B, int N, int Nk, int H);
int combo_sra_attention_forward_bf16(const void* q, const void* kv, void* out, float* lse2, int B, int N, int Nk, int H, float scale,
                                     combo_stream_t stream);
int combo_sra_attention_backward_bf16(const void* q, const void* kv, const void* out, const void* dout, const float* lse2, float* delta,
                                      float* part, void* dq, void* dkv, int B, int N, int Nk, int H, float scale, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a12  masked multi-head attention of the decoder layers, head_dim = 32 (csrc/attention.hip)
 *   replaces nn.MultiheadAttention's core `softmax(q k^T * scale + mask) v` as called by CrossAttentionLayer / SelfAttentionLayer
 *   (transformer_decoder/transformer_decoder.py:99-118, 50-58; the packed in_proj / out_proj GEMMs are combo_gemm_nt_f32).
 *   q [B, Lq, >= H*32] (row stride ldq), k / v [B, Lk, .] (ldk / ldv), head h at column h*32; blocked: bytes [B, Lq, pitch]
 *   (1 = masked out, shared by the H heads; NULL = no mask), pitch % 4 == 0; scale = head_dim^-0.5 is applied to q.
 *   Both also take the mask bit-packed (blocked_bits [B, Lq, wpitch] words from combo_mask_bits_f32, NULL = use the
 *   bytes): one word per (query, 32-key tile) instead of 32 byte reads.
 *   forward:  out [B, Lq, H*32], lse [B, H, Lq] (log-sum-exp of the scaled masked scores, saved for backward), exact fp32 MFMA.
 *   backward: dq [B, Lq, H*32], dk / dv [B, Lk, H*32] from dout; delta_ws: [B, H, Lq] workspace.  A query whose keys are all
 *   blocked yields zeros (the reference would produce NaN; the decoder never passes such a row: :458 resets it).
 * ---------------------------------------------------------------------------------------------- */
int combo_attention_forward_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv,
                                const unsigned char* blocked, int pitch, const unsigned* blocked_bits, int wpitch, int B, int H,
                                int Lq, int Lk, float scale, float* out, float* lse, combo_stream_t stream);
int combo_attention_backward_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv,
                                 const unsigned char* blocked, int pitch, const unsigned* blocked_bits, int wpitch, int B, int H,
                                 int Lq, int Lk, float scale, const float* out, const float* lse, const float* dout,
                                 float* delta_ws, float* dq, float* dk, float* dv, combo_stream_t stream);
/*   _ld: dk / dv written with row pitches lddk / lddv (floats; multiples of 4, >= H*32): column blocks of a wider buffer - the K / V
 *   gradients of the decoder layers that share a memory level (ops/linear.py memory_kv). */
int combo_attention_backward_ld_f32(const float* q, long long ldq, const float* k, long long ldk, const float* v, long long ldv,
                                    const unsigned char* blocked, int pitch, const unsigned* blocked_bits, int wpitch, int B, int H,
                                    int Lq, int Lk, float scale, const float* out, const float* lse, const float* dout,
                                    float* delta_ws, float* dq, float* dk, long long lddk, float* dv, long long lddv,
                                    combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a14  Hungarian-matcher cost matrices, all (decoder output x frame) problems at once
 *   replaces HungarianMatcher.memory_efficient_forward's cost computation (models/modeling/matcher.py:93-131:
 *   point_sample of predictions and targets at P shared random points, batch_sigmoid_ce_loss :34-52,
 *   batch_dice_loss :13-28, class cost :97) - everything up to, not including, the LSAP solve.
 *   logits [N,Q,K1], masks [N,Q,h,w] (mask logits), labels [N,G] int64, gt [N,G,H,W] fp32 0/1, points [N,P,2] in [0,1],
 *   t_ws [N,G,P] workspace (sampled targets), cost [N,Q,G].  G <= 8.
 * ---------------------------------------------------------------------------------------------- */
/*   mask_base (optional, [N] int64): index of query 0's map of problem n inside a larger stack of maps (the decoder's
 *   [heads, BT, Q] logits buffer), so the ground-truth frames need no gathered copy; NULL: maps packed [N, Q]. */
int combo_matcher_cost_f32(const float* logits, const float* masks, const long long* mask_base, const long long* labels,
                           const float* gt, const float* points, int N, int Q, int K1, int G, int h, int w, int H, int W, int P,
                           float w_class, float w_mask, float w_dice, float* t_ws, float* cost, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a1  Siam-Encoder-Module mix on channels-last activations (bf16 or fp32 in, fp32 out)
 *   replaces channel_weighted_block's global average pool (models/utils/misc.py:125) and the mix
 *   `features[key] + scale * pre_sam_features[key]` (models/maskformer_model.py:352) and their backward.
 *   op 0: acc[B,C] += sum_i p[b,i,c]                 (a = p; o1 = acc, zero-filled by the caller)
 *   op 1: out[b,i,c] = f[b,i,c] + s[b,c]*p[b,i,c]    (a = f, b = p; o1 = out fp32)
 *   op 2: ds[B,C] += sum_i dout[b,i,c]*p[b,i,c]      (a = p, b = dout fp32; o1 = ds, zero-filled)
 *   op 3: df = dout, dp = dout*s + dgap[b,c]         (a = dout fp32, g = dgap (already / HW); o1 = df, o2 = dp)
 *   C % 8 == 0; f, p, df, dp are bf16 when is_bf16 != 0, else fp32.
 * ---------------------------------------------------------------------------------------------- */
int combo_sem_mix(int op, int is_bf16, const void* a, const void* b, const float* s, const float* g, int B, int HW, int C,
                  void* o1, void* o2, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a17  inference tail: out[f,k,Y,X] = sum_q cls_prob[f,q,k] * sigmoid(bilinear_up(masks[f,q])[Y,X])
 *   replaces F.interpolate(pred_masks, (H,W), bilinear, align_corners=False) + semantic_inference
 *   (models/maskformer_model.py:397-402, 460-464).  cls_prob [F,Q,K] = softmax(logits)[..., :-1]; K <= 8 per call.
 * ---------------------------------------------------------------------------------------------- */
int combo_semantic_inference_f32(const float* cls_prob, const float* masks, int F, int Q, int K, int h, int w, int H,
                                 int W, float* out, combo_stream_t stream);

/*   Exact linear sum assignment on the device for G <= 6 targets per problem (replaces the per-frame
 *   scipy.optimize.linear_sum_assignment host call + .cpu() sync, matcher.py:132-134).
 *   cost [N,Q,Gpad], gcount [N] int32 (real number of targets, <= Gpad <= 6) -> row_for_col [N,Gpad] int64 (-1 = padding).
 *   status (nullable device int): bits are OR-ed in, never cleared - COMBO_LSAP_NONFINITE_COST when a NaN / Inf cost was read
 *   (scipy raises ValueError there, matcher.py:133; the kernel reads it as the largest finite cost and still returns valid,
 *   distinct rows), COMBO_LSAP_TOO_MANY_TARGETS when gcount[n] > min(Gpad, 6).  Gpad > 6 returns COMBO_EINVAL. */
int combo_lsap_small_f32(const float* cost, const int* gcount, int N, int Q, int Gpad, long long* row_for_col,
                         int* status, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * a15  mask losses of the criterion (models/modeling/criterion.py:137-186, :19-62, :70-84)
 *   Pair n (n < NM) addresses its prediction map masks + mask_index[n]*h*w (stacked [L*F*Q,h,w] logits) and its target
 *   gt + gt_index[n]*H*W (padded [F*Gmax,H,W] fp32 0/1).
 *   combo_uncertain_points_f32 : get_uncertain_point_coords_with_randomness with uncertainty -|logit|:
 *       coords_out[n,0:k] = the k of the NS `over_points[n]` with the smallest |bilinear logit| (radix select, index
 *       order, ties by index), coords_out[n,k:k+NR] = extra_points[n].
 *   combo_mask_loss_forward_f32: stats[n] = (sum_p BCEwithlogits(x_p,t_p), sum_p s_p t_p, sum_p s_p, sum_p t_p) over the
 *       P points coords[n] (x = point_sample(prediction), t = point_sample(target), s = sigmoid(x)).
 *   combo_mask_loss_backward_f32: grad_masks[mask_index[n]] = d/dlogits (g_bce[n]*mean_p BCE + g_dice[n]*dice);
 *       grad_masks must be zero-filled (only matched maps are written).
 * ---------------------------------------------------------------------------------------------- */
int combo_uncertain_points_f32(const float* masks, const long long* mask_index, int NM, int h, int w, const float* over_points,
                               int NS, const float* extra_points, int NR, int k, float* coords_out, combo_stream_t stream);
int combo_mask_loss_forward_f32(const float* masks, const long long* mask_index, int NM, int h, int w, const float* gt,
                                const long long* gt_index, int H, int W, const float* coords, int P, float* stats,
                                combo_stream_t stream);
int combo_mask_loss_backward_f32(const float* masks, const long long* mask_index, int NM, int h, int w, const float* gt,
                                 const long long* gt_index, int H, int W, const float* coords, int P, const float* stats,
                                 const float* g_bce, const float* g_dice, float* grad_masks, int accumulate,
                                 const long long* grad_index, combo_stream_t stream);
/* accumulate != 0: grad_masks[...] += (another term is already there); grad_index (nullable): the gradient of pair n is
 * written at grad_masks + grad_index[n]*h*w instead of + mask_index[n]*h*w (a gradient stack in another layout) */

/*   Frame-to-frame cosine loss (criterion.py:208-231): x [rows,E], rows = heads*BT, clips = n_frame consecutive rows.
 *   stats: nrm[r] += |x_r|^2, dot[r] += x_r . x_{r+1} (same clip; both zero-filled by the caller);
 *   grad : grad[r] = 2 gnrm[r] x_r + gdot[r] x_{r+1} + gdot[r-1] x_{r-1} (neighbours inside the clip); perm_outer > 0: row
 *          r = (head, frame) of [rows / perm_inner, perm_inner] is written at row frame * perm_outer + head (the gradient stack
 *          transposed to [BT, heads, E]: the layout the mask-logit gradient GEMMs read, no 0.5 GB re-layout copy). */
int combo_cosine_stats_f32(const float* x, long long rows, long long E, int n_frame, float* dot, float* nrm, combo_stream_t stream);
int combo_cosine_grad_f32(const float* x, long long rows, long long E, int n_frame, const float* gdot, const float* gnrm,
                          float* grad, long long perm_inner, long long perm_outer, combo_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * (f)1  optimiser: fused gradient clip + AdamW on a flat fp32 segment
 *   replaces FullModelGradientClippingOptimizer.step (train_net.py:205-221): p *= 1-lr*wd; m,v EMAs of
 *   clip_coef*g; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps).  clip_coef: device scalar (NULL = 1).
 *   All four arrays are 16-byte aligned, length n.
 * ---------------------------------------------------------------------------------------------- */
int combo_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n,
                    const float* clip_coef, float lr, float weight_decay, float beta1, float beta2, float eps,
                    float bias_correction1, float bias_correction2, combo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* COMBO_AVS_H_ */
