/*
 * llie.h -- C ABI of libllie_hip.so: the MI355X (gfx950) engine for the LCM denoising hot path of
 * zamazincode/cv-diffusion-model.
 *
 * The reference has no FFI (it is pure PyTorch), so this header *defines* the native boundary that
 * replaces its hot path.  Each entry point cites the reference interface it stands in for
 * (paths relative to the reference root).  Conventions:
 *   - plain pointers and sizes only; no torch types.  All tensor pointers are DEVICE pointers.
 *   - public I/O tensors are fp32 NCHW contiguous, exactly what the reference's callers hold
 *     (scripts/inference.py:137-145, scripts/benchmark.py:61-79); NHWC / reduced precision are
 *     internal to the engine.
 *   - the caller owns every activation / IO buffer and the workspace (size from
 *     llie_workspace_bytes); the handle owns only the repacked weights.
 *   - every function returns 0 on success, a negative llie_status on bad arguments, or a positive
 *     hipError_t passed through.  Nothing throws.  llie_last_error() gives a message.
 *   - launches are asynchronous on the caller's stream; no hidden synchronisation, no host reads
 *     of device data.  A handle is not re-entrant (one stream at a time), like the reference module
 *     (lcm_scheduler.py:163-164,245 mutate scheduler state inside enhance()).
 */
#ifndef LLIE_H_
#define LLIE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct llie_ctx llie_ctx;
typedef void* llie_stream; /* hipStream_t */

enum llie_status {
  LLIE_OK = 0,
  LLIE_ERR_ARG = -1,       /* null pointer / bad enum / bad size */
  LLIE_ERR_SHAPE = -2,     /* shape not supported by the engine (e.g. image side not a multiple of 8, or below 64) */
  LLIE_ERR_CONFIG = -3,    /* topology the reference itself cannot construct (GroupNorm divisibility) */
  LLIE_ERR_KEY = -4,       /* unknown state_dict key or wrong element count */
  LLIE_ERR_NOT_LOADED = -5,/* forward called before every parameter was loaded */
  LLIE_ERR_WORKSPACE = -6, /* workspace too small */
  LLIE_ERR_NO_DEVICE = -7
};

enum llie_dtype { LLIE_F32 = 0, LLIE_F16 = 1, LLIE_BF16 = 2 };

/* Which module tree a handle holds.  LLIE_UNET is the product; the others expose single reference
 * operators (same parameter names as the reference classes) so that parity tests can be written
 * per operator, the way the reference's modules are organised. */
enum llie_kind {
  LLIE_UNET = 0,   /* EfficientUNet                      efficient_unet.py:387-606 */
  LLIE_IRB = 1,    /* InvertedResidualBlock              efficient_unet.py:134-236 */
  LLIE_ATTN = 2,   /* LinearAttention                    efficient_unet.py:239-308 */
  LLIE_DOWN = 3,   /* Downsample (3x3 stride-2 conv)     efficient_unet.py:360-372 */
  LLIE_UP = 4,     /* Upsample (bilinear x2 + 3x3 conv)  efficient_unet.py:375-384 */
  LLIE_SE = 5      /* SqueezeExcitation (in_channels = C) efficient_unet.py:79-100; forward only */
};

/* Mirrors EfficientUNetConfig (efficient_unet.py:24-57) for LLIE_UNET; for the single-operator kinds
 * only the fields named in the comments are read. */
typedef struct llie_config {
  int kind;               /* llie_kind */
  int compute_dtype;      /* llie_dtype: storage type of activations / MFMA operand type (accumulation is fp32) */
  int in_channels;        /* UNET: 6 (concat conditioning, low_light_diffusion.py:77); IRB/ATTN/DOWN/UP: C_in */
  int out_channels;       /* UNET: 3; IRB: C_out */
  int base_channels;
  int channel_multipliers[4];
  int num_res_blocks;
  int expansion_ratio;
  int time_embed_dim;     /* UNET, IRB */
  int num_attention_heads;/* UNET, ATTN */
  int image_size;         /* UNET: decides attention placement (efficient_unet.py:426,447,509) */
  int attention_resolutions[2];
  int allow_unpinned;     /* 0: topologies whose nn.GroupNorm(min(32,C), C) the reference cannot construct (tiny, base)
                             fail with LLIE_ERR_CONFIG like the reference's ValueError.  1: build them with the
                             documented deviation groups = largest divisor of C that is <= 32 (no reference output
                             exists to pin this: parity-unpinned; inference only) */
} llie_config;

/* LCM scheduler coefficients for one step (host values; computed by the host-side scheduler from the
 * fp32 alpha-bar table, lcm_scheduler.py:208-242). */
typedef struct llie_step_coef {
  float sqrt_alpha_t, sqrt_beta_t;       /* alpha_prod_t**0.5, (1-alpha_prod_t)**0.5 */
  float sqrt_alpha_prev, sqrt_beta_prev; /* for prev_t; ignored when is_last */
  int   is_last;                         /* prev_t == 0 -> prev_sample = x0 (lcm_scheduler.py:228-229) */
  int   v_prediction;                    /* 0 epsilon (:217), 1 v_prediction (:220) */
  int   clamp_x0;                        /* 1: x0 = clip(x0, -1, 1) before re-noising -- the deployment loop's
                                            semantics (src/export/android_pipeline.py:250-252); the LCMScheduler
                                            has that clamp commented out (lcm_scheduler.py:224-225) -> 0 */
} llie_step_coef;

const char* llie_last_error(void);
const char* llie_version(void);

/* create_efficient_unet / EfficientUNet.__init__ (efficient_unet.py:403-530, 631-692).  Fails with
 * LLIE_ERR_CONFIG for topologies whose GroupNorm the reference rejects (tiny, base). */
int llie_create(const llie_config* cfg, llie_ctx** out);
void llie_destroy(llie_ctx* ctx);

/* state_dict introspection: number of parameters and (key, element count, shape) of each, in the
 * reference's registration order, keys without the "unet." prefix.  `shape4` receives
 * the tensor's shape in the reference's state_dict (ndim entries valid). */
int llie_num_params(const llie_ctx* ctx);
int llie_param_info(const llie_ctx* ctx, int index, char* key_buf, size_t key_cap, int64_t* numel, int* ndim,
                    int64_t* shape4);

/* nn.Module.load_state_dict, one tensor at a time (scripts/inference.py:78-79, scripts/benchmark.py:56):
 * `src` is a DEVICE pointer to the fp32 tensor in the reference's own layout (OIHW conv weights,
 * [out,in] Linear weights); the engine repacks it on `stream`. */
int llie_load_param(llie_ctx* ctx, const char* key, const float* src, int64_t numel, llie_stream stream);
/* Reload all parameters at once -- what `optimizer.step()` implies for a module whose weights live outside PyTorch
 * (src/training/trainer.py:300-318) and what `load_state_dict` does (scripts/inference.py:78-84): srcs[i] = device fp32 tensor of parameter i in
 * llie_param_info order (host array of n = llie_num_params pointers).  One kernel for all plain / 1x1 / 3x3 /
 * depthwise tensors; equivalent to n llie_load_param calls. */
int llie_load_all(llie_ctx* ctx, const float* const* srcs, int n, llie_stream stream);
/* Same arguments; reloads only if the parameters' CONTENT differs from the last load.  The comparison (a 64-bit
 * position-weighted hash of the fp32 bits) and the decision both happen on the device, so the call stays asynchronous:
 * it catches in-place writes that bypass PyTorch's version counters -- `param.data.copy_(...)`, which is how the
 * reference's EMA swaps weights in and out (src/training/trainer.py:104-117, low_light_diffusion.py:317-323). */
int llie_refresh_params(llie_ctx* ctx, const float* const* srcs, int n, llie_stream stream);
int llie_params_loaded(const llie_ctx* ctx); /* 1 when every key has been loaded */

/* Bytes of scratch the forward needs for a batch (UNET: spatial size = image_size; single
 * operators: H x W given). */
int64_t llie_workspace_bytes(llie_ctx* ctx, int batch, int height, int width);
/* Workspace for llie_enhance including the staging area its hipGraph path needs (inputs / outputs of
 * up to `max_steps` steps).  With only llie_workspace_bytes() the loop runs as plain launches. */
int64_t llie_enhance_workspace_bytes(llie_ctx* ctx, int batch, int max_steps);

/* EfficientUNet.forward(x, timestep) (efficient_unet.py:532-606) with x given as its two concat
 * halves (low_light_diffusion.py:222: cat([latents, low_light], 1)), both fp32 NCHW [B,3,S,S].
 * `timesteps`: device int64[B].  uniform_t != 0 promises all B timesteps are equal (true inside
 * enhance(), low_light_diffusion.py:218) and lets the engine evaluate the time MLPs once.
 * `eps_out`: fp32 NCHW [B,3,S,S]. */
int llie_unet_forward(llie_ctx* ctx, const float* latents, const float* cond, const int64_t* timesteps,
                      int uniform_t, float* eps_out, int batch, void* workspace, int64_t workspace_bytes,
                      llie_stream stream);

/* Single-operator forward for kinds IRB / ATTN / DOWN / UP: x fp32 NCHW [B,C,H,W] -> y fp32 NCHW.
 * `temb` (IRB only): device fp32 [B, time_embed_dim] time embedding (efficient_unet.py:203). */
int llie_module_forward(llie_ctx* ctx, const float* x, const float* temb, float* y, int batch, int height,
                        int width, void* workspace, int64_t workspace_bytes, llie_stream stream);

/* ---- Training step (callers: src/training/trainer.py:269-338 via LowLightDiffusion.compute_loss,
 * low_light_diffusion.py:140-171,250-277).  The forward pass keeps its activations in the workspace; the
 * backward pass walks them in reverse and writes every parameter's gradient, fp32 in the reference's
 * state_dict layout, into one flat buffer: parameter i (llie_param_info order) at llie_param_grad_offset(i),
 * llie_grad_numel() floats in total.  Timesteps are per sample (int64[B]).  All reductions run in a fixed
 * order, so gradients are bitwise reproducible.
 *   llie_unet_train_forward:  eps = unet(cat[lat, cond], t), activations retained in `workspace`
 *   llie_unet_backward:       given d(loss)/d(eps) (fp32 NCHW) -> grads; `workspace` must be the untouched
 *                             buffer of the preceding train_forward; lat / cond / t must still be alive
 *   llie_module_backward:     single operator (IRB / ATTN / DOWN / UP handles): forward + backward in one call;
 *                             dx fp32 NCHW like x, dtemb [B][T] (IRB only) */
int64_t llie_grad_numel(const llie_ctx* ctx);
int64_t llie_param_grad_offset(const llie_ctx* ctx, int index);
int64_t llie_train_workspace_bytes(llie_ctx* ctx, int batch, int height, int width);
int llie_unet_train_forward(llie_ctx* ctx, const float* latents, const float* cond, const int64_t* timesteps, float* eps,
                            int batch, void* workspace, int64_t workspace_bytes, llie_stream stream);
int llie_unet_backward(llie_ctx* ctx, const float* d_eps, float* grads, int batch, void* workspace, int64_t workspace_bytes,
                       llie_stream stream);
int llie_module_backward(llie_ctx* ctx, const float* x, const float* temb, const float* dy, float* dx, float* dtemb,
                         float* grads, int batch, int H, int W, void* workspace, int64_t workspace_bytes, llie_stream stream);

/* ---- Optimiser step of the training loop (replaces, in the trainer's inner loop src/training/trainer.py:296-324, the eager
 * sequence  scaler.unscale_ -> torch.nn.utils.clip_grad_norm_(params, gradient_clip) -> optimizer.step() [torch.optim.AdamW,
 * trainer.py:163-168] -> EMAModel.update (trainer.py:98-104)  by three launches over every parameter tensor at once; same
 * arithmetic operation for operation: decoupled decay p *= 1 - lr wd; m = m + (g - m)(1 - b1); v = b2 v + (1 - b2) g g;
 * p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps); shadow = decay shadow + (1 - decay) p).
 * A tensor's gradient is read at grad_base + grad_offset (the flat buffer llie_unet_backward fills: grad_offset =
 * llie_param_grad_offset(i)); `ema` may be NULL.  All pointers are device fp32 and must stay valid for the optimiser's life.
 * llie_optimizer_step: every gradient is first multiplied by grad_scale (1 / loss scale, 1 / world size); the total norm of the
 * scaled gradients is clipped to max_grad_norm (<= 0: no clipping) as clip_grad_norm_ does (factor min(1, max / (norm + 1e-6)));
 * ema_decay < 0: shadows untouched; skip_nonfinite != 0: a non-finite norm leaves parameters, moments and shadows unchanged
 * (GradScaler.step).  `step` is the 1-based count of this update.  stats3 (device, 3 floats) receives {gradient norm, factor
 * applied to the gradients, 1 if the step was skipped else 0}.  The norm is a fixed-order sum: bitwise reproducible. */
typedef struct llie_opt_tensor {
  float* param;
  float* exp_avg;
  float* exp_avg_sq;
  float* ema;
  int64_t grad_offset; /* elements */
  int64_t numel;
} llie_opt_tensor;
typedef struct llie_opt_hyper {
  double lr, beta1, beta2, eps, weight_decay;
  double max_grad_norm, ema_decay, grad_scale;
  int64_t step;
  int32_t skip_nonfinite;
} llie_opt_hyper;
typedef struct llie_optimizer llie_optimizer;
int llie_optimizer_create(const llie_opt_tensor* tensors /* host array */, int count, llie_optimizer** out);
void llie_optimizer_destroy(llie_optimizer* opt);
int64_t llie_optimizer_numel(const llie_optimizer* opt);
int llie_optimizer_step(llie_optimizer* opt, const float* grad_base, const llie_opt_hyper* hyper, float* stats3, llie_stream stream);

/* LCMScheduler.step (lcm_scheduler.py:176-253), elementwise on fp32 [n]:
 *   x0 = (sample - sqrt_beta_t*model_output)/sqrt_alpha_t      (epsilon)
 *   prev = is_last ? x0 : sqrt_alpha_prev*x0 + sqrt_beta_prev*noise
 * `noise` may be null when is_last.  `x0_out` and `clamped_out` (prev.clamp(-1,1),
 * low_light_diffusion.py:240) are optional (null to skip). */
int llie_lcm_step(const float* model_output, const float* sample, const float* noise, float* prev_out,
                  float* x0_out, float* clamped_out, int64_t n, const llie_step_coef* coef, llie_stream stream);

/* LCMScheduler.add_noise / get_velocity (lcm_scheduler.py:255-305): per-sample timesteps (device
 * int64[B]) index a device fp32 alpha-bar table [num_train_timesteps]; tensors fp32 [B, per_sample]. */
int llie_add_noise(const float* x0, const float* noise, const int64_t* timesteps, const float* alphas_cumprod,
                   int table_len, float* out, int batch, int64_t per_sample, int velocity, llie_stream stream);
/* A timestep outside [0, table_len) -- an IndexError in the reference -- is never used as an index: that sample's
 * output is NaN (the call is asynchronous and cannot raise; the Python layer validates host-side timesteps). */

/* Whole denoising loop of LowLightDiffusion.enhance (low_light_diffusion.py:204-240) on one stream:
 * `noise` is fp32 [steps,B,3,S,S] in the reference's draw order (initial latents first, then one draw
 * per non-final step); `timesteps_dev` device int64 [steps,B]; `coefs` host array [steps].
 * Outputs: `enhanced` [B,3,S,S] (clamped); optional `intermediates` [steps,B,3,S,S] (post-step,
 * pre-clamp latents, :236-237) and `noise_preds` [steps,B,3,S,S].
 * When the workspace has llie_enhance_workspace_bytes() the launch sequence is captured into a hipGraph
 * on its second use for a given (batch, schedule, workspace, stream) and replayed afterwards; user
 * tensors are staged through the workspace so the graph's pointers never change
 * (LLIE_NO_GRAPH=1 in the environment disables this).  The cache of captured graphs holds at most 16 entries per
 * context; the least recently used one is destroyed when a 17th key appears (llie_graph_cache_entries reads the count). */
int llie_enhance(llie_ctx* ctx, const float* low_light, const float* noise, const int64_t* timesteps_dev,
                 const llie_step_coef* coefs, int steps, float* enhanced, float* intermediates,
                 float* noise_preds, int batch, void* workspace, int64_t workspace_bytes, llie_stream stream);

int llie_graph_cache_entries(const llie_ctx* ctx);

/* Byte-level I/O either side of the path (scripts/inference.py:99-134), on the device:
 *   llie_preprocess_u8:  uint8 HWC RGB [B][H0][W0][3] -> resize to SxS (cv2.INTER_LINEAR geometry, round half
 *                        up to uint8) -> x/127.5 - 1 -> fp32 NCHW [B][3][S][S]
 *   llie_postprocess_u8: fp32 NCHW [B][3][S][S] -> (x+1)*127.5, clip [0,255], truncate to uint8 -> resize to
 *                        H0xW0 -> uint8 HWC RGB [B][H0][W0][3]
 * Bit-exact with the host implementation in hostio.py (fp32 arithmetic without fused multiply-adds). */
int llie_preprocess_u8(const uint8_t* img, int batch, int H0, int W0, float* out, int S, llie_stream stream);
int llie_postprocess_u8(const float* x, int batch, int S, uint8_t* img, int H0, int W0, llie_stream stream);

/* ---- Kernel-level entry points (unit tests and tuning; SURVEY.md 8b "per-kernel entry points").
 * Activations are NHWC rows in the compute dtype T (llie_dtype); see DESIGN.md section 3.
 *
 * llie_pw_gemm: out[M][N] = sum over K-segments act(A_seg * scale + bias) . W[N][K]^T (+bias[N]) (+residual)
 *   -- the 1x1 convolutions efficient_unet.py:174,186,199,265-267 with their fused prologue/epilogue.
 *   scale/bias of a segment are fp32 [M/P][affine_ld] tables (null = identity); act: 0 none, 1 ReLU6.
 *   stats (optional): fp32 slab [M/P][P/tile_rows][2][N] of per-channel (sum, sum of squares).
 * llie_dwconv3x3: out = depthwise3x3(relu6(in*scale+bias)), weights fp32 [9][C] tap-major
 *   (efficient_unet.py:212-220); pool (optional): fp32 [B][tiles][C] partial sums for the SE average pool. */
typedef struct llie_gemm_seg {
  const void* ptr; int channels; const float* scale; const float* bias; int affine_ld; int act;
} llie_gemm_seg;
int llie_pw_gemm(int dtype, const llie_gemm_seg* segs, int nseg, const void* w, const float* bias, const void* residual,
                 void* out, float* stats, int M, int N, int P, llie_stream stream);
int llie_pw_gemm_tile_rows(int P);
/* llie_pw_expand: the expanding 1x1 conv of the wide InvertedResidualBlocks (efficient_unet.py:174 with norm1 + ReLU6
 *   :207-208 in the prologue, norm2's statistics :212 in the epilogue) in its activation-stationary form (2-byte dtypes;
 *   K in {128, 192, 256, 384, 512}, P a multiple of 128, every segment with act 3 = clamp01 tables already divided by 6):
 *   out[M][N] = sum_seg clamp01(A_seg * scale + bias) . (6 W)^T.  w32 = fp32 [N][K] as saved by the reference; wpack = N*K
 *   elements of T that receive the fragment-ordered copy the engine keeps per layer (w32 == NULL: wpack already holds it,
 *   from an earlier call with the same weights -- the GEMM alone).  stats: as llie_pw_gemm. */
int llie_pw_expand(int dtype, const llie_gemm_seg* segs, int nseg, const float* w32, void* wpack, void* out, float* stats,
                   int M, int N, int P, llie_stream stream);
/* The remaining per-kernel entry points of the hot path.  All tensors are device pointers; activations are NHWC of
 * the compute type `dtype` (0 fp32, 1 fp16, 2 bf16), tables and statistics fp32.
 *
 * llie_groupnorm_finalize: nn.GroupNorm(groups, C) statistics (efficient_unet.py:170-171,263,268,528) from per-tile
 *   (sum, sum of squares) slabs [batch][ntiles][2][ch] of up to two channel segments (a virtual concat) to the per-(image,
 *   channel) affine `scale`, `shift` [batch][C] that consumers apply on load; `film` (or NULL): [rows][2 C] FiLM (1 + scale, shift
 *   folded in, :215-217) with row stride film_stride (0 = one row for all images); post_scale 0 = none.
 * llie_conv3x3: Downsample (:367, mode 0: stride 2, pad 1) / Upsample (:383-384, mode 1: bilinear x2 then 3x3 pad 1) as implicit
 *   GEMM; w [9][Cout][Cin] of the compute type (tap-major), stats (or NULL) [batch][llie_conv3x3_tiles(Ho, Wo)][2][Cout].
 * llie_linattn: LinearAttention core (:288-302) on qkv [batch][N][3 * 32 heads] (q | k | v, head-major channels, dim_head 32):
 *   phi = elu + 1 on q and k, kv = sum_n phi(k) v^T, out = phi(q) kv / (phi(q) . sum_n phi(k) + 1e-6) -> [batch][N][32 heads];
 *   kv_scratch: llie_linattn_splits(N) * batch * heads * 32 * 33 floats.
 * llie_se_mlp: SqueezeExcitation MLP (:96-100): gate = sigmoid(W2 relu6(W1 mean + b1) + b2), mean = pool_sums / pixels;
 *   w1 [Cs][C], w2 [C][Cs] of the compute type; scratch: mean [batch][C], hidden [batch][Cs]; gate out [batch][C].
 * llie_film: all FiLM projections of a network in one launch (:189-192,215): film[r][f] = bf[f] + Wf[f][:] . silu_temb[r][:]. */
int llie_groupnorm_finalize(const float* slab0, int ntiles0, int ch0, const float* slab1, int ntiles1, int ch1, int groups, int pixels,
                            const float* gamma, const float* beta, const float* film, int64_t film_stride, float eps, float post_scale,
                            int batch, float* scale_out, float* shift_out, llie_stream stream);
int llie_conv3x3(int dtype, int mode, const void* in, const void* w, const float* bias, void* out, float* stats, int batch, int Hi, int Wi,
                 int Cin, int Cout, llie_stream stream);
int llie_conv3x3_tiles(int Ho, int Wo);
int llie_linattn(int dtype, const void* qkv, float* kv_scratch, void* out, int batch, int N, int heads, llie_stream stream);
int llie_linattn_splits(int N);
int llie_se_mlp(int dtype, const float* pool_sums, int pixels, const void* w1, const float* b1, const void* w2, const float* b2, float* mean_scratch,
                float* hidden_scratch, float* gate, int batch, int C, int Cs, llie_stream stream);
int llie_film(const float* silu_temb, const float* wf, const float* bf, float* film, int rows, int T, int F, llie_stream stream);

/* Statistics pass of the recompute form of InvertedResidualBlock (efficient_unet.py:207-212) on its own: Gram matrix
 * G = sum_px a' a'^T and column sums m = sum_px a' of a' = clamp01(x * scale + bias) rounded to the compute type, per image
 * (gram.hip).  x0 / x1: NHWC [batch][pixels][c0 / c1] of the compute type (x1 may be NULL with c1 = 0), c0 + c1 in {32, 64, 96},
 * c0 % 8 == 0, pixels % 512 == 0; scale / bias [batch][c0 + c1] fp32; part: scratch of batch * llie_gram_part_floats floats;
 * gtot out: [batch][K * K + K] fp32 (G row-major, then m); tickets: [batch] uint32, zero on entry (left zero). */
int llie_gram_stats(int dtype, const void* x0, int c0, const void* x1, int c1, const float* scale, const float* bias, int batch,
                    int pixels, float* part, float* gtot, unsigned int* tickets, llie_stream stream);
int64_t llie_gram_part_floats(int K, int pixels);
/* The finalize behind it (gram.hip: gram_finalize_kernel; replaces nn.GroupNorm(32, 4K) statistics of h1 = W1 . relu6(norm1(x)),
 * efficient_unet.py:174,212, + the FiLM fold :215-217): sum h1[c] = 6 w_c . m, sum h1[c]^2 = 36 w_c^T G w_c (row products and sums
 * in fp64), then scale[b][c] = rstd gamma[c] (1 + fs), shift[b][c] = (beta[c] - mean rstd gamma[c]) (1 + fs) + fh like
 * llie_groupnorm_finalize.  gram_totals: llie_gram_stats' gtot; w_expand: [4K][K] of the compute type; K in {32, 64, 96}. */
int llie_gram_finalize(int dtype, const float* gram_totals, const void* w_expand, int K, int pixels, const float* gamma, const float* beta,
                       const float* film, int64_t film_stride, float eps, float post_scale, int batch, float* scale_out, float* shift_out,
                       llie_stream stream);
int llie_dwconv3x3(int dtype, const void* in, void* out, const float* scale, const float* bias, const float* w9c,
                   float* pool, int B, int H, int W, int C, llie_stream stream);
int llie_dwconv3x3_tiles(int H, int W);
/* dst[0:bytes] = src[0:bytes] with 16-byte lane accesses: the on-box HBM copy-bandwidth probe behind bench.py's
 * `peak_measured` ("a copy-kernel bandwidth probe"; 2 x bytes move per call). */
int llie_copy_probe(const void* src, void* dst, int64_t bytes, llie_stream stream);
/* Streaming probe with a chosen read : write mix: `units` steps, each reading `reads` and writing `writes` 16 KB blocks
 * ((reads, writes) in {(1,0),(0,1),(1,1),(1,2),(1,4),(2,1),(4,1)}; src holds units*reads, dst units*writes blocks).
 * nontemporal: 0 plain, 1 non-temporal loads and stores, 2 write-through (sc1) stores, 3 sc0 sc1 stores.  reads == -1: a
 * trivial dependent launch (boundary cost studies).
 * The ceiling the write-dominated 4x expansions (efficient_unet.py:174) are compared with. */
int llie_rw_probe(const void* src, void* dst, int64_t units, int reads, int writes, int nontemporal, llie_stream stream);
/* Engine knobs (process-wide; every call starts a new epoch of the hipGraph cache).  Production defaults in brackets.
 *   "enhance_split"  [2]    concurrent batch branches of the captured enhance graph (1 = one chain; env LLIE_ENHANCE_SPLIT)
 *   "irbx"           [1]    recompute form of the inverted-residual front half (0 = expand GEMM + depthwise kernel)
 *   "irbx_dbuf" [0], "irbx_tiles" [4], "irbx_mask" [7], "irbx_dwv" [1]   variants of the recompute kernels (A/B runs)
 *   "ztot"           [1]    SE pool as fixed-point totals + fused gate kernel (0 = slab + pool / fc1 / fc2 launches)
 *   "gemm_bk" [0 = auto], "gemm_bk128" [1024 = largest grid that takes 128-wide K chunks], "dw_swap" [0],
 *   "bwd_async" [1], "wgrad_target" [1024], "pwx" [1] (activation-stationary expand GEMM, pwx.hip; 0 = tile kernel),
 *   "gram" [1] (norm2 statistics of the recompute form from the Gram matrix of the block input, gram.hip; 0 = expand_stats),
 *   "pwx_nbw" [0 = per-K default] (32-channel blocks per weight buffer), "pwx_ablate" 6 / 7 (stores straight from registers / through the LDS tile)
 * Diagnostics whose results are WRONG or slow (timing studies only): "skip_small", "gemm_ablate", "dw_ablate",
 * "irbx_ablate", "gemm_stamp", "irbx_stamp", "conv_stamp", "pwx_ablate", "pwx_stamp".
 * Threading: the knobs are plain process-wide variables read by every forward; call llie_tune only while no other
 * thread is inside an llie_* compute call (same rule as the handle itself: SURVEY.md 8b, one stream at a time). */
int llie_tune(const char* knob, int value);
int llie_debug_irbx_stamps(double* out10); /* diagnostic builds: 9 per-wave cycle sums of expand_dw (irbx.hip: STAMP) + waves averaged */
int llie_debug_conv_stamps(double* out8); /* diagnostic builds: 7 per-wave cycle sums of the up-sampling conv (conv.hip: STAMP) + waves averaged */
int llie_debug_gemm_stamps(double* out3); /* diagnostic builds: see gemm.hip (STAMP) */
int llie_debug_pwx_stamps(double* out4);  /* diagnostic builds: see pwx.hip (STAMP): {A phase, channel loop, of which waiting for the weight DMA} cycles per wave, waves */

/* Per-kernel-class timing with HIP events recorded on the launch stream (what bench.py's `roofline`
 * object is computed from).  llie_profile_begin arms recording for the classes in `class_mask`;
 * every subsequent launch of those kernels is bracketed by an event pair (at most 8192 pairs).
 * llie_profile_end disarms, synchronises on the recorded events and returns, for `kernel_class`, the
 * summed device time, the launch count and the summed ALGORITHMIC bytes of those launches
 * (activation elements the kernel must read and write once, plus its weights). */
enum llie_kernel_class {
  LLIE_K_GEMM = 1,  /* pw_gemm_kernel: 1x1 convs with fused prologue / epilogue */
  LLIE_K_DW = 2,    /* dwconv3x3_kernel */
  LLIE_K_CONV3 = 4, /* conv3x3_kernel (down / up sampling convs) */
  LLIE_K_SE = 8,    /* squeeze-excitation MLP launches */
  LLIE_K_OTHER = 16 /* everything else on the forward path (norm finalize, attention core, input / output convs, time MLPs) */
};
int llie_profile_begin(llie_ctx* ctx, int class_mask);
int llie_profile_end(llie_ctx* ctx, int kernel_class, double* total_ms, int64_t* launches, int64_t* algorithmic_bytes);
/* Same data aggregated per kernel NAME (template arguments included, the granularity of
 * `rocprofv3 --kernel-trace --stats`): writes lines "name\tms\tlaunches\talgorithmic_bytes\n" into buf. */
int llie_profile_report(llie_ctx* ctx, char* buf, size_t cap);
/* Every recorded launch in launch order: lines "class\tkernel\toperator tag\tms\talgorithmic_bytes\n". */
int llie_profile_dump(llie_ctx* ctx, char* buf, size_t cap);

/* SinusoidalPosEmb + time_mlp of a LLIE_UNET handle on their own (efficient_unet.py:60-76, 412-417, and the SiLU
 * that opens every block's FiLM projection, :189-192): emb [rows][base_channels] ([cos | sin]), temb [rows][T],
 * silu_temb [rows][T]; timesteps device int64 [rows].  emb may be null. */
int llie_time_embed(llie_ctx* ctx, const int64_t* timesteps, int rows, float* emb, float* temb, float* silu_temb,
                    llie_stream stream);

/* llie_algorithmic_bytes returns the roofline numerator of SURVEY.md 8d for one
 * UNet forward of `batch` images at the handle's dtype (activation traffic + weights once). */
int64_t llie_algorithmic_bytes(llie_ctx* ctx, int batch);
int64_t llie_flops(llie_ctx* ctx, int batch);
/* Same model with the blocks the engine runs in the recompute form (irbx.hip: h1 is never stored) counted as
 * (3Cin + 2Chid + Cout) P -- the bytes the engine's own kernel selection has to move. */
int64_t llie_path_bytes(llie_ctx* ctx, int batch);

#ifdef __cplusplus
}
#endif
#endif /* LLIE_H_ */
