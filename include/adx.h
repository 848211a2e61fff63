/*
 * adx.h -- C ABI of libadx.so: MI355X (gfx950) kernels and sampling engines for the
 * trajectory-diffusion refinement path.
 *
 * The reference implements this path as Python/PyTorch modules; it has no FFI of its own.
 * Each entry point below therefore names the reference *Python* interface it stands in for
 * (paths relative to the reference tree).  A binding is a ctypes stub (autodiffusion_amd/_lib.py is one).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless its name ends in _host;
 *   - tensors are contiguous row-major float32 unless stated; timesteps are int64;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *   - every function returns 0 (ADX_OK) or a negative/positive error code and records a message
 *     retrievable with adx_last_error() (thread-local); no C++ exception crosses the boundary;
 *   - work is enqueued on `stream` and NOT synchronised;
 *   - an engine is bound to the device that was current when it was created and is not
 *     re-entrant (the reference guards generate() with a mutex: run_simulation.py:736-748).
 */
#ifndef ADX_H_
#define ADX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADX_ABI_VERSION 1

enum { ADX_OK = 0, ADX_EINVAL = 1, ADX_ESTATE = 2, ADX_EHIP = 3, ADX_ENOMEM = 4, ADX_ENOTFOUND = 5 };

enum { ADX_ACT_NONE = 0, ADX_ACT_RELU = 1, ADX_ACT_GELU_ERF = 2, ADX_ACT_GELU_TANH = 3, ADX_ACT_LEAKY = 4, ADX_ACT_SILU = 5,
       ADX_ACT_TANH = 6 /* stand-alone operator adx_op_act only (action / key-point embeddings of the raster encoder) */ };
/* GEMM operand type; accumulation is always f32.  BF16X3 (engines only): bf16 MFMA on operands split into
 * high + low bf16 halves (x_lo w_hi + x_hi w_lo + x_hi w_hi), ~16 mantissa bits per operand. */
enum { ADX_PREC_F32 = 0, ADX_PREC_BF16 = 1, ADX_PREC_BF16X3 = 2,
       /* DiT engine and the operand-plane product only: operand planes hold fp16 halves -- activations split into high + low
        * fp16 halves, weights rounded once to fp16 (2^-12 relative) -- and a product is x_hi w + x_lo w, two matrix instructions
        * instead of three; every other kernel of such an engine runs as ADX_PREC_BF16X3.  Values beyond the fp16 range
        * saturate at +-65504. */
       ADX_PREC_F16X2 = 3,
       /* the same with the activations rounded once to fp16 as well: one plane per operand, one matrix instruction per product,
        * 2^-12 relative per operand.  Every other kernel of such an engine still runs as ADX_PREC_BF16X3.
        * Also a precision of the refiner engine (ADX_VARIANT_A): every product of its two per-step kernels that walk all
        * conditioning rows -- the encoder row pipeline (attention out-projection, both FFN products) and the last encoder attention
        * (scores, values, weighted sums) -- multiplies operands rounded once to fp16, one matrix instruction per product, and the
        * tensors those kernels exchange through memory (cached attention outputs, gates, LayerNorm rows) are stored as fp16; the
        * per-chunk caches, the tail and the scheduler run as under ADX_PREC_BF16X3. */
       ADX_PREC_F16 = 4 };
enum { ADX_VARIANT_A = 0, ADX_VARIANT_DIT = 1, ADX_VARIANT_MMDIT = 2 };

/* sampling flags */
#define ADX_FLAG_NORMALIZE 1u   /* auto_normalize_data: utils.py:56-80 applied to prior/transition in, trajectory out */
#define ADX_FLAG_RESIDUAL 2u    /* add the raw prior to the un-normalised sample (TrajRefiner.py:382-384) */
#define ADX_FLAG_INIT_ZERO 4u   /* x_N = 0 instead of noise[0] (determin=True, diffusion_decoder.py:390-393) */
#define ADX_FLAG_NO_GRAPH 8u    /* launch kernels eagerly instead of replaying a captured hipGraph */
#define ADX_FLAG_KEEP_MAPS 16u  /* adx_sample_a: maps_info is unchanged since the previous adx_sample_a call with this batch
                                   size (the chunks of one trajectory, ExplicitDiT.py:164-190): keep its embedding and caches */

int adx_abi_version(void);
const char* adx_last_error(void);

/* --------------------------------------------------------------------------------------------
 * Operator-level entry points (stateless).
 * ------------------------------------------------------------------------------------------ */

/* SinusoidalPosEmb.forward -- models/diffusion_loss/positional_embedding.py:10-17, libs/embeds.py:13-20.
 * t[n] int64 -> out[n][dim] = [sin(t w), cos(t w)]. */
int adx_op_timestep_embedding(const int64_t* t, int n, int dim, float* out, void* stream);

/* torch.nn.Linear (+ fused activation): y[M][N] = act(x[M][K] w[N][K]^T + bias).  With
 * ADX_PREC_BF16, `w` holds bf16 bit patterns (see adx_op_to_bf16) and x is rounded to bf16 as it
 * is staged; with ADX_PREC_BF16X3, `w` holds N*K high parts followed by N*K low parts
 * (bf16(w) and bf16(w - bf16(w))) and x is split the same way on the fly.  Stands in for every nn.Linear on the path (transformer.py:50,60-61,133;
 * libs/diffusionTransformer.py:38-47,63-76,134-165,189-193; libs/mlp.py:28,50). */
int adx_op_linear(const float* x, const void* w, const float* bias, float* y, int M, int N, int K, int act,
                  int prec, void* stream);
/* Same product with a caller-provided scratch buffer (device memory, 16-byte aligned, at least
 * adx_linear_stage_bytes(M, N, K, prec) bytes; bf16 precisions only): x and w are re-laid once as K-blocked bf16
 * planes and large problems are tiled by LDS-DMA copies of whole cache lines instead of being converted again by every
 * column tile.  Same reference
 * call sites as adx_op_linear; results agree with it to the last bit (same products, same accumulation order). */
int adx_op_linear_staged(const float* x, const void* w, const float* bias, float* y, int M, int N, int K, int act,
                         int prec, void* workspace, size_t workspace_bytes, void* stream);
size_t adx_linear_stage_bytes(int M, int N, int K, int prec);
/* The staged product with operands kept in its own layout across calls (same reference call sites as adx_op_linear; the
 * layers of libs/diffusionTransformer.py:63-76,134-165 multiply the same activations by several weights and the same weights
 * at every sampling step).  A plane set is K-blocked bf16, [K/16][rows][16], adx_linear_planes_bytes(rows, K, prec) bytes,
 * 16-byte aligned: the high halves, followed under ADX_PREC_BF16X3 by the low halves.  adx_op_stage_planes writes the
 * planes of fp32 x[M][K]; adx_op_block_weight those of a weight in adx_op_linear's `w` form; adx_op_linear_planes multiplies
 * two plane sets (M >= 256, N >= 64 and a multiple of 4, K >= 512 and a multiple of 64, at least 320 tiles of 256 x 128;
 * anything else is ADX_EINVAL).  Results equal adx_op_linear_staged's bit for bit.
 * Under ADX_PREC_F16X2 the planes hold fp16: x as high + low halves (two planes), the weight -- passed to adx_op_block_weight as
 * N*K fp16 bit patterns, see adx_op_to_f16 -- as ONE plane, and the product takes two matrix instructions per tile step
 * (x_hi w + x_lo w); the error is that of rounding w to fp16, 2^-12 relative per weight. */
size_t adx_linear_planes_bytes(int rows, int K, int prec);
int adx_op_stage_planes(const float* x, int M, int K, int prec, void* planes, void* stream);
int adx_op_block_weight(const void* w, int N, int K, int prec, void* blocked, void* stream);
int adx_op_linear_planes(const void* x_planes, const void* w_blocked, const float* bias, float* y, int M, int N, int K, int act,
                         int prec, void* stream);
/* No counterpart in the reference.  For callers that cannot allocate device memory (code under hipGraph capture, tight serving
 * loops, harnesses that forbid hipMalloc): adx_op_linear_staged with the staging workspace taken from a fixed device array inside
 * the library (adx_noalloc_scratch_bytes() = 64 MiB per GPU, held until the process ends) -- the call allocates nothing.  Always
 * runs the staged kernel, so it needs M >= 256, N >= 64, K a multiple of 64, a bf16 precision, and
 * adx_linear_stage_bytes(M, N, K, prec) <= adx_noalloc_scratch_bytes(); anything else is ADX_EINVAL.  Results equal
 * adx_op_linear_staged's bit for bit.  Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU race. */
int adx_op_linear_noalloc(const float* x, const void* w, const float* bias, float* y, int M, int N, int K, int act, int prec,
                          void* stream);
/* Per-device set-up of the entry point above (no counterpart in the reference either): call once per device, with that device
 * current and outside any stream capture -- it resolves the fixed array's address (which loads the code object) and sets kernel
 * attributes, neither of which may happen while a stream is being captured.  adx_op_linear_noalloc does it by itself on its first
 * call on a device; callers that record it into a hipGraph call this (or one eager adx_op_linear_noalloc) beforehand. */
int adx_noalloc_init(void);
size_t adx_noalloc_scratch_bytes(void);
int adx_op_to_bf16(const float* x, uint16_t* y, int64_t n, void* stream);
int adx_op_to_f16(const float* x, uint16_t* y, int64_t n, void* stream);     /* fp16 bit patterns, saturating at +-65504 */

/* select_k_from_mc (models/utils.py:120-132 -> utils/modify_traj_utils.py: farthest_point_sample_PointCloud :8-30,
 * run_EM_algorithm / _compute_EM_algorithm :124-219): reduce the M sampled trajectories traj[B][M][L][D] of every
 * scenario to k modes -- farthest-point sampling on the end points starting at sample 0, then em_iter (reference: 25)
 * EM iterations of a diagonal Gaussian mixture with uniform sample weights; modes[B][k][L][D] ordered by weights[B][k],
 * descending. */
/* compute_metrics (trainer.py:184-305), the per-scenario part: out[B][18] = ade at 1/3/5/8 s, fde at 1/3/5/8 s, heading
 * error of point 0, mean heading error over 3/5/8 s, heading error at points 29/49/79, max displacement over 3/5/8 s
 * (heading entries are 0 when C == 2).  pred, label: [B][S][C] with S dividing 80 (horizons scale by 80 / S). */
int adx_op_trajectory_metrics(const float* pred, const float* label, int B, int S, int C, float* out, void* stream);

int adx_op_select_modes(const float* traj, int B, int M, int L, int D, int k, int em_iter, float* modes, float* weights,
                        void* stream);

/* LayerNorm over the last axis, optionally affine (gamma/beta), optionally followed by the adaLN
 * `modulate` x*(1+scale)+shift with per-row shift/scale of row stride ld_mod, optionally ReLU.
 * nn.LayerNorm uses in transformer.py:132 and torch TransformerEncoder/DecoderLayer; custom LayerNorm
 * libs/mlp.py:5-20; modulate + LayerNorm(elementwise_affine=False): libs/diffusionTransformer.py:19-20,30,35. */
int adx_op_layernorm(const float* x, const float* gamma, const float* beta, const float* shift, const float* scale,
                     int64_t ld_mod, float eps, int relu, int64_t rows, int d, float* y, void* stream);

/* Multi-head softmax attention on contiguous q[B][Sq][H*dh], k[B][Skv][H*dh], v[B][Skv][H*dv] ->
 * o[B][Sq][H*dv].  mask_off < 0: unmasked; otherwise key j is visible to query i iff j <= i + mask_off
 * (0 = the causal tgt_mask, 1 = the memory_mask of transformer.py:106-124).
 * torch.nn.MultiheadAttention core / DiT_Attention.forward (libs/diffusionTransformer.py:99-117). */
int adx_op_attention(const float* q, const float* k, const float* v, float* o, int B, int H, int Sq, int Skv, int dh,
                     int dv, float scale, int mask_off, void* stream);
/* The same with the engine's operand precision: under ADX_PREC_BF16 / ADX_PREC_BF16X3 unmasked attention with Sq == Skv <= 128
 * and head widths that are multiples of 32 (every DiT / MM-DiT block) runs on the bf16 matrix cores, scores and probabilities
 * rounded to bf16 (split into high + low halves under BF16X3); other shapes and ADX_PREC_F32 are adx_op_attention. */
int adx_op_attention_prec(const float* q, const float* k, const float* v, float* o, int B, int H, int Sq, int Skv, int dh,
                          int dv, float scale, int mask_off, int precision, void* stream);

/* normalize / denormalize -- models/utils.py:56-80 (C = 4 or 2 channels). `residual` may be NULL. */
int adx_op_normalize(const float* x, float* y, int64_t n_points, int C, void* stream);
int adx_op_denormalize(const float* y, const float* residual, float* x, int64_t n_points, int C, void* stream);

/* One scheduler update over n elements.  All reference samplers reduce to this form:
 *     x0   = clamp(a0*x + a1*model_out, -clip, +clip)
 *     eps  = eps_mode ? (b0*x - x0) * b1 : model_out
 *     x'   = c0*x0 + c1*x + c2*eps + c3*z
 * DDPM p_sample (TrajRefiner.py:296-313; diffusion_decoder.py:434-479), DDIM ddim_sample
 * (TrajRefiner.py:349-371), and the diffusers-style DDPM/DDIM steps of diffusion_transition.py:72-75 /
 * traj_diffusion.py:76-78 differ only in the coefficients, which the host computes.
 * `t` is the diffusion timestep fed to the denoiser; `z_index` selects noise[1 + z_index] (-1: none). */
typedef struct adx_step_coef {
    float t, a0, a1, clip, eps_mode, b0, b1, c0, c1, c2, c3, z_index;
} adx_step_coef;
int adx_op_sched_step(const float* x, const float* model_out, const float* z, float* x_next, float* x0_out, int64_t n,
                      const adx_step_coef* coef_dev, void* stream);

/* --------------------------------------------------------------------------------------------
 * Engines: a denoiser (weights + workspace) plus the sampling loop, captured as one hipGraph.
 * ------------------------------------------------------------------------------------------ */
typedef struct adx_engine adx_engine;

typedef struct adx_model_config {
    int variant;        /* ADX_VARIANT_* */
    int precision;      /* ADX_PREC_* */
    int elide;          /* 0: every op of the reference; 1: skip work whose result the masks make unobservable;
                           2: also reuse, within a chunk, the encoder layer-0 attention of all step-invariant rows
                              (both exact in real arithmetic; 2 re-associates one softmax sum);
                           3: as 2, with the per-row encoder pipeline fused into one kernel (d_model 32, 8 heads, 2 layers) */
    int d_model;        /* n_emb (A) / feat_dim = n_embd (DiT) */
    int n_head;
    int n_enc_layers;   /* A: n_cond_layers */
    int n_layers;       /* A: decoder layers; DiT: blocks */
    int horizon;        /* A: T tokens per chunk; DiT: predict_range (80) */
    int in_dim;         /* A: channels per token (4, or fs*4 when flattened); DiT: out_features (4) */
    int out_dim;
    int n_prior_tokens; /* A: P (= frame_stack, or 1 when flattened) */
    int prior_dim;      /* A */
    int n_maps_token;   /* A: 788 */
    int maps_dim;       /* A: 256 */
    int map_cond;       /* A */
    int n_inner;        /* DiT: hidden width of state_encoder[0] */
    int connect_dim;    /* DiT: width between the two state_encoder stages */
} adx_model_config;

int adx_engine_create(const adx_model_config* cfg, adx_engine** out);
int adx_engine_destroy(adx_engine* e);

/* Copy one parameter (float32, device memory) into the engine under its reference state_dict name,
 * e.g. "encoder.layers.0.self_attn.in_proj_weight" (variant A: names of
 * models/diffusion_loss/transformer.py::Transformer) or "blocks.3.adaLN_modulation.1.weight"
 * (DiT: names of models/diffusion_loss/diffusion_decoder.py::Transformer_TrajDiffusionModel). */
int adx_engine_set_tensor(adx_engine* e, const char* name, const float* data, const int64_t* shape_host, int ndim);
/* Verify that every parameter the configured model needs is present with the right shape. */
int adx_engine_finalize(adx_engine* e);
/* Call order: adx_engine_create, adx_engine_set_tensor for every parameter, adx_engine_finalize, adx_engine_set_schedule (sampling
 * calls only), then any number of denoise / sample / generate calls.  Parameters may be replaced at any time with
 * adx_engine_set_tensor followed by adx_engine_finalize: finalize rebuilds every table derived from the weights (the schedule
 * set earlier stays in force) and drops the captured graphs. */

/* (Re)allocate the activation workspace for batches up to max_batch rows. */
int adx_engine_reserve(adx_engine* e, int max_batch);
int64_t adx_engine_workspace_bytes(const adx_engine* e);

/* Transformer.forward(noisy_x, timestep, condition) -- models/diffusion_loss/transformer.py:189-273.
 * x[B][T][in_dim], t[B] int64, maps[B][n_maps][maps_dim], transition/prior[B][P][prior_dim] (already
 * normalised, as the denoiser sees them) -> out[B][T][out_dim] (never squeezed). */
int adx_denoise_a(adx_engine* e, int B, const float* x, const int64_t* t, const float* maps, const float* transition,
                  const float* prior, float* out, void* stream);

/* Transformer_TrajDiffusionModel.forward(x, t, state) -- models/diffusion_loss/diffusion_decoder.py:251-277.
 * x[B][S][4], t[B] int64, state[B][S][d_model] -> out[B][S][4]. */
int adx_denoise_dit(adx_engine* e, int B, const float* x, const int64_t* t, const float* state, float* out,
                    void* stream);

/* No counterpart in the reference (its samplers call torch.randn / randn_like: TrajRefiner.py:319,311,367): seed and global index
 * of batch row 0 for sampling calls that pass noise == NULL.  Such calls draw their noise on the device, inside the captured
 * graph: Philox4x32-10 keyed by `seed`, counter = (row0 + batch row, chunk * (1 + draws) + draw, element / 4), Box-Muller.  A row's
 * noise therefore does not depend on which shard of the batch it runs in.  Only the draws the schedule reads are generated. */
int adx_engine_set_noise_seed(adx_engine* e, uint64_t seed, int64_t row0);
/* The same generator as an operator: out[n_lead][B][row_len]; ctl_scratch = 16 bytes of device memory. */
int adx_op_randn(float* out, int n_lead, int B, int row_len, uint64_t seed, int64_t row0, uint64_t* ctl_scratch, void* stream);

/* No counterpart in the reference: number of hipGraph captures this engine has made.  A sampling call captures once per
 * (shape, flags, schedule); callers whose buffer addresses change from call to call are served from engine-owned staging
 * buffers (inputs copied in, result copied out on `stream`), so they do not capture again either. */
int64_t adx_engine_capture_count(const adx_engine* e);

/* DiffusionWrapper.q_sample / predict_v (TrajRefiner.py:250-254, 408-415): xt = sqrt_acp[t] x0 + sqrt_1m_acp[t] noise with one
 * timestep per row of row_len values; v (nullable) = sqrt_acp[t] noise - sqrt_1m_acp[t] x0. */
int adx_op_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1m_acp,
                    float* xt, float* v, int n_rows, int row_len, void* stream);
/* The loss of DiffusionWrapper.p_losses (TrajRefiner.py:448-452): loss_rows[b] = weight[t[b]] * mean((out[b] - target[b])^2),
 * loss_mean[0] (nullable) = mean over rows. */
int adx_op_weighted_mse(const float* out, const float* target, const int64_t* t, const float* weight, float* loss_rows,
                        float* loss_mean, int n_rows, int row_len, void* stream);

/* The default branch of TrajectoryDecoder.compute_traj_loss (models/decoder/base.py:9-35, 238-241): logits[B][T][C], label[B][T][C_label];
 * out[0] = mean over (b, t, first two channels) of the squared error (smooth_l1 = 0) or SmoothL1 with beta 1; out[1] = mean over
 * (b, t) of the squared difference of the LAST channels as angles, each angle and the difference wrapped by atan2(sin, cos).
 * part[B][2] receives the per-row sums. */
int adx_op_traj_loss(const float* logits, const float* label, int B, int T, int C, int C_label, int smooth_l1, float* part, float* out,
                     void* stream);

/* --------------------------------------------------------------------------------------------
 * Conditioning producer, first stage (phase 1 of generate(): NuplanRasterizeEncoder.forward,
 * models/encoder/nuplan_raster_encoder.py:178-187, with cat_raster_seq, models/utils.py:14-41, and the Conv2d patch embedding
 * of the ViT it feeds).  raster[B][H][W][C] is the channels-last raster as the data loader hands it over, C = route + 20 road +
 * 4 traffic-light + 8 agent channels per context frame (agent channel a of frame f at route + 24 + f + a * framenum).
 * out[(b * framenum + f) * (W/16) * (H/16) + py * (H/16) + px][(c * 16 + ky) * 16 + kx] = raster[b][px*16 + kx][py*16 + ky][chan(c, f)]
 * with c over the route + 20 (+ 4 when `traffic`) shared channels followed by frame f's 8 agent channels: the rows of the patch
 * matrix whose product with the flattened Conv2d filter [hidden][c * 256] (adx_op_linear) is the reference's patch embedding of
 * raster.permute(0, 3, 2, 1) split into frames.  H, W multiples of 16; C <= 149. */
int adx_op_raster_patches(const float* raster, int B, int H, int W, int C, int framenum, int traffic, float* out, void* stream);

/* --------------------------------------------------------------------------------------------
 * Conditioning producer, middle stage: the sparse mixture-of-experts decoder that STR.embedding_to_hidden runs
 * (models/backbone/str_base.py:379-387 on the MixtralModel of models/backbone/mixtral.py:17-19; the decoder layer itself is
 * third-party `transformers`, pinned 4.42.3 in the reference's requirements.txt:212).  Its products and its causal attention are
 * adx_op_linear and adx_op_attention (mask_off = number of cached positions); these are the operators it needs beyond them.
 * ------------------------------------------------------------------------------------------ */
/* y[r] = weight * (x[r] * rsqrt(mean(x[r]^2) + eps)) over rows of d floats (MixtralRMSNorm). */
int adx_op_rmsnorm(const float* x, const float* weight, float eps, int64_t rows, int d, float* y, void* stream);
/* Rotate-half rotary embedding, in place: x[B][S][H * dh], token s of every item at position pos0 + s;
 * (x1, x2) -> (x1 cos - x2 sin, x2 cos + x1 sin) with x1 / x2 the two halves of a head and cos / sin tables [n_pos][dh / 2]
 * (angle(p, j) = p * rope_theta^(-2 j / dh)). */
int adx_op_rotary(float* x, const float* cos_table, const float* sin_table, int B, int S, int H, int dh, int pos0, int n_pos, void* stream);
/* The expert block: y[t] = res[t] + sum_j w_j E_{e_j}(x[t]),  E_e(x) = W2_e (silu(W1_e x) * W3_e x),  (e_j, w_j) = the top_k largest
 * of softmax(router_w x[t]) renormalised to sum one (ties: lowest expert index).  x, res (nullable), y: [T][d] fp32; router_w [E][d] fp32;
 * w13 [E][2 * inter][d] (rows 0 .. inter-1 of an expert = W1, the gate; the rest = W3) and w2 [E][d][inter] in the form
 * adx_op_linear reads under `precision` (fp32 values; bf16 bits; under ADX_PREC_BF16X3 the high halves of all experts followed by
 * the low halves).  The router always computes in fp32.  No host synchronisation, no atomics: tokens are bucketed per expert on
 * the device, both products run over the 64-row-padded buckets, results are combined in slot order.  workspace: at least
 * adx_moe_workspace_bytes(T, d, inter, E, top_k) bytes, 16-byte aligned; after the call its first T * top_k int32 hold the chosen
 * experts and, from the next multiple of 256 bytes on, T * top_k floats hold their weights.  d and inter multiples of 32, E <= 64,
 * top_k <= 8. */
size_t adx_moe_workspace_bytes(int64_t T, int d, int inter, int E, int top_k);
int adx_op_moe_block(const float* x, const float* res, const float* router_w, const void* w13, const void* w2, int64_t T, int d, int inter,
                     int E, int top_k, int precision, void* workspace, size_t workspace_bytes, float* y, void* stream);

/* --------------------------------------------------------------------------------------------
 * Operators of the training paths (fp32): the refiner, the flattened wrapper and the decoder twin.  The reference differentiates
 * its losses with torch autograd (loss.backward() on TrajRefiner.py:417-452 over transformer.py:189-273; diffusion_decoder.py:346-365
 * over libs/diffusionTransformer.py:24-199); the entry points below are the backward of the forward operators above, plus the few
 * forward operators a training forward needs on their own (activations, products, attention with kept statistics).  dX of a Linear is adx_op_linear on the transposed weight (adx_op_transpose).  Reductions over rows are
 * summed per row chunk and combined in a fixed order: results do not depend on launch timing.
 * ------------------------------------------------------------------------------------------ */
int adx_op_add(const float* a, const float* b, float* out, int64_t n, void* stream);              /* residual adds, gradient sums */
int adx_op_transpose(const float* x, float* y, int rows, int cols, void* stream);                 /* y[cols][rows] = x^T */
int adx_op_mul(const float* a, const float* b, float* out, int64_t n, void* stream);              /* gates, modulation: a * b */
/* The activations of the decoder twin (ADX_ACT_* codes: ReLU of nn.TransformerEncoderLayer and DecoderResCat, tanh-GELU of
 * the block MLPs, SiLU of adaLN, LeakyReLU(0.01) of the time encoder; libs/diffusionTransformer.py:24-53, libs/mlp.py:43-55,
 * diffusion_decoder.py:126-145) as stand-alone operators, and dx = dy * act'(x) with x the pre-activation. */
int adx_op_act(const float* x, float* y, int64_t n, int act, void* stream);
int adx_op_act_bwd(const float* x, const float* dy, float* dx, int64_t n, int act, void* stream);
/* out[g][c] = sum over r of x[g][r][c] (x is [G][R][N], G <= 65535): the gradient of a per-item vector that the forward
 * broadcast over the R tokens of an item (the time embedding added to every token's condition, diffusion_decoder.py:256). */
int adx_op_group_colsum(const float* x, int G, int64_t R, int N, float* out, void* stream);
/* F.gelu (erf form; activation of nn.TransformerEncoder/DecoderLayer, transformer.py:69-102) and its derivative:
 * dx = dy (Phi(x) + x phi(x)) with x the pre-activation. */
int adx_op_gelu(const float* x, float* y, int64_t n, void* stream);
int adx_op_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* Scratch bytes of the row reductions below: adx_reduce_ws_bytes(M, N, K) for adx_op_linear_wgrad, (M, N, 0) for adx_op_colsum. */
size_t adx_reduce_ws_bytes(int64_t M, int N, int K);
/* nn.Linear backward, parameter side: dw[N][K] = dy[M][N]^T x[M][K], db[N] (nullable) = column sums of dy; added to the
 * existing contents when `accumulate` is non-zero (torch accumulates into .grad). */
int adx_op_linear_wgrad(const float* dy, const float* x, int64_t M, int N, int K, float* dw, float* db, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream);
/* out[N] (+)= column sums of x[M][N]: LayerNorm gain / bias gradients, gradients of the learned position tables (sum over
 * the batch), bias gradients. */
int adx_op_colsum(const float* x, int64_t M, int N, float* out, int accumulate, void* workspace, size_t workspace_bytes,
                  void* stream);
/* nn.LayerNorm backward: dx, and dy_xhat (nullable) = dy * (x - mean) * rstd whose column sums are the gain gradient (the
 * bias gradient is the column sum of dy).  gamma may be NULL (no affine). */
int adx_op_layernorm_bwd(const float* x, const float* gamma, const float* dy, int64_t rows, int d, float eps, float* dx,
                         float* dy_xhat, void* stream);
/* Backward of adx_op_attention (same layouts and mask rule; dv == dh) for head widths 4, 8, 16, 32 at any sequence length and
 * for wider heads (multiples of 4; the 128 ... 512-wide joint attention of MM_DiTBlock, libs/diffusionTransformer.py:99-117)
 * when two Sq x Skv matrices fit in 64 KB of LDS (about 88 tokens on either side): o is the forward output,
 * dout its gradient; stats holds 2*B*H*Sq floats (per query and head: the log-sum-exp of the scores, base 2, and dout.o).
 * With stats_have_lse != 0 the log-sum-exps were written by adx_op_attention_stats and the pass that finds them is skipped;
 * otherwise stats is pure scratch.  Probabilities are recomputed, never stored. */
int adx_op_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout, int B, int H,
                         int Sq, int Skv, int dh, float scale, int mask_off, float* dq, float* dk, float* dv, float* stats,
                         int stats_have_lse, void* stream);
/* adx_op_attention for a forward whose backward will follow (same shapes as adx_op_attention_bwd, fp32): writes o and, in
 * stats[((b*H + h)*Sq + i)*2], the log-sum-exp of query i's scores. */
int adx_op_attention_stats(const float* q, const float* k, const float* v, float* o, float* stats, int B, int H, int Sq,
                           int Skv, int dh, float scale, int mask_off, void* stream);

/* Dropout of the training mode.  The reference trains with nn.Dropout(p_drop_emb) on the embedded sequences (transformer.py:52,
 * 248, 258) and with torch's TransformerEncoderLayer / TransformerDecoderLayer dropout (p_drop_attn there; torch's default 0.1
 * inside every DiTBlock, libs/diffusionTransformer.py:28) on attention probabilities, attention outputs, the FFN hidden layer and
 * the FFN output.  Every decision is a pure function of (seed, site, position) -- an integer hash, see csrc/backward.hip -- so
 * the backward pass regenerates its masks: adx_op_dropout applied to dy with the forward's (seed, site) IS the backward of
 * adx_op_dropout.  y = x * m, m = 1 / (1 - p) with probability 1 - p, else 0; x == NULL writes m itself (the mask, for
 * inspection or replay).  `site` numbers the dropout layers of one forward pass. */
int adx_op_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, int site, void* stream);
/* adx_op_attention_stats / adx_op_attention_bwd with dropout on the attention probabilities (after the softmax, before the
 * product with V, as torch's scaled_dot_product_attention applies dropout_p); head widths 4, 8, 16, 32.  host_mask (nullable):
 * multipliers [B][H][Sq][Skv] (0 or 1 / (1 - p)) that replace the hash -- how a test replays the masks a reference run drew;
 * adx_op_attention_dropout_mask writes the multipliers the hash gives. */
int adx_op_attention_stats_dropout(const float* q, const float* k, const float* v, float* o, float* stats, int B, int H, int Sq,
                                   int Skv, int dh, float scale, int mask_off, float p, uint64_t seed, int site,
                                   const float* host_mask, void* stream);
int adx_op_attention_bwd_dropout(const float* q, const float* k, const float* v, const float* o, const float* dout, int B, int H,
                                 int Sq, int Skv, int dh, float scale, int mask_off, float* dq, float* dk, float* dv, float* stats,
                                 int stats_have_lse, float p, uint64_t seed, int site, const float* host_mask, void* stream);
int adx_op_attention_dropout_mask(float* out, int B, int H, int Sq, int Skv, float p, uint64_t seed, int site, void* stream);
/* Gradient of adx_op_weighted_mse's loss_mean with respect to `out`, times upstream[0] (device scalar; NULL = 1):
 * dout[b][e] = upstream * 2 weight[t[b]] (out - target) / (n_rows * row_len)   (TrajRefiner.py:448-452 under autograd). */
int adx_op_weighted_mse_bwd(const float* out, const float* target, const int64_t* t, const float* weight,
                            const float* upstream, float* dout, int n_rows, int row_len, void* stream);

/* Per-step coefficient table of the sampler (host memory), in execution order. */
int adx_engine_set_schedule(adx_engine* e, const adx_step_coef* steps_host, int n_steps);

/* DiffusionWrapper.sample for one chunk -- TrajRefiner.py:315-386 (p_sample_loop / ddim_sample / sample).
 * transition/prior are RAW (un-normalised) [B][P][prior_dim]; noise[(1+n_draws)][B][T][in_dim] (NULL: drawn on the device, see
 * adx_engine_set_noise_seed); out[B][T][in_dim].  trace_x / trace_out (nullable) receive x_t and the model output of every step,
 * [n_steps][B][T][in_dim] each. */
int adx_sample_a(adx_engine* e, int B, const float* maps, const float* transition, const float* prior,
                 const float* noise, int n_noise, float* out, float* trace_x, float* trace_out, unsigned flags,
                 void* stream);

/* Chunk loop of the generate() wrappers -- models/backbone/DiffusionAugmentation.py:137-195 (token layout)
 * and models/backbone/ExplicitDiT.py:139-198 (flattened layout, config in_dim = frame_stack*4):
 * prior_full[B][S][4] is cut into S/frame_stack chunks, each sampled with the previous chunk's output as
 * transition_info (init_x[frame_stack][4] for the first), results re-assembled into out[B][S][4].
 * noise[n_chunks][n_noise][B][T][in_dim] (NULL: drawn on the device, see adx_engine_set_noise_seed). */
int adx_generate_a(adx_engine* e, int B, int S, int frame_stack, const float* maps, const float* prior_full,
                   const float* init_x, const float* noise, int n_noise, float* out, unsigned flags, void* stream);

/* Latent-carrying chunk loop (StrDiffDiT.generate, models/backbone/StrDiffDiT.py:202-235, around DiffusionTransitionModel.
 * conditional_sample, models/diffusion_loss/diffusion_transition.py:43-78) on a variant-A engine whose head is out_dim = z_dim wide:
 * per chunk of frame_stack points the sample starts from the chunk of prior_full[B][S][4] (first frame of the trajectory zeroed),
 * every step calls the denoiser with transition = z_emb(latent), maps x_from_z(denoiser latent) to the model output and applies
 * the step's coefficient row (adx_engine_set_schedule; z_index counts from 0: there is no initial draw); the latent of a chunk's
 * last step conditions the next chunk; out[B][S][4] = sample * std + mean with mean_std = {mean[4], std[4]} on the device
 * (nullable: no affine map).  init_z[frame_stack][z_dim] is the first chunk's latent for every item.  The tensors
 * "z_emb.weight" [4][z_dim], "z_emb.bias", "x_from_z.weight" [4][z_dim], "x_from_z.bias" are set with adx_engine_set_tensor
 * beside the denoiser's.  noise[n_chunks][n_noise][B][frame_stack * 4].  The reference's denoiser class for this wrapper is
 * missing from its tree: how the latent enters the network (z_emb into the transition slot) is this library's choice. */
int adx_generate_transition(adx_engine* e, int B, int S, int frame_stack, const float* maps, const float* prior_full,
                            const float* init_z, const float* noise, int n_noise, const float* mean_std, float* out,
                            unsigned flags, void* stream);

/* No counterpart in the reference: per-kernel timing of real calls.  While enabled, sampling calls on this engine run
 * without graph capture and bracket their marked launches with events on the caller's stream; stages of variant A:
 * 1 fused encoder rows, 2 last-layer attention, 3 fused tail, 4 per-chunk cache build, 5 per-call map cache build.
 * adx_engine_profile_read
 * synchronises the device and returns the summed duration and the number of launches of one stage; stage -1 clears
 * what was collected. */
int adx_engine_profile(adx_engine* e, int enable);
int adx_engine_profile_read(adx_engine* e, int stage, double* total_ms, int* count);

/* No counterpart in the reference: launches ONE kernel stage of a variant-A sampling step `iters` times on the
 * engine's current workspace (after a generate/sample call with the same B), so a benchmark can time that kernel
 * alone with events on `stream`.  stage: 0 time-token attention, 1 fused encoder rows, 2 last-layer attention,
 * 3 fused tail, 4 per-chunk attention-state build, 5 per-call map-token cache build. */
int adx_debug_stage_a(adx_engine* e, int B, int stage, int iters, void* stream);

/* Two stacked Linears of a 256-wide model in one kernel, the hidden layer never written to memory:
 *   y = res + gate * act2(act1(x w1^T + b1) w2^T + b2)        (gate, res nullable: [M][256]; b1, b2 nullable)
 * x[M][256], w1 = adx_op_block_weight form of a prepared [N1][256] weight, w2 likewise of a [256][N1] weight, N1 a
 * multiple of 64 (at most 4096), M >= 128, bf16 precisions.  The shape of nn.TransformerEncoderLayer's linear1/linear2 and of
 * DiTBlock.mlp (libs/diffusionTransformer.py:24-53) at hidden size 256. */
int adx_op_mlp256(const float* x, const void* w1_blocked, const float* b1, int act1, const void* w2_blocked, const float* b2,
                  int act2, const float* gate, const float* res, float* y, int M, int N1, int precision, void* stream);

/* DiffusionWrapper.sample_forward / p_sample_loop -- models/diffusion_loss/diffusion_decoder.py:377-446.
 * state[B][S][d_model]; noise[n_noise][B][S][4] (NULL: none with ADX_FLAG_INIT_ZERO, otherwise drawn on the device, see
 * adx_engine_set_noise_seed); out[B][S][4] is de-normalised when ADX_FLAG_NORMALIZE is set. */
int adx_sample_dit(adx_engine* e, int B, const float* state, const float* noise, int n_noise, float* out,
                   unsigned flags, void* stream);

/* The same loop for mc_num samples per scenario -- diffusion_decoder.py:408-416 repeats every scenario's state mc_num times
 * (repeat_interleave) before sampling.  Here the B sample rows come in groups of `group` consecutive rows that share one
 * state item: state[B / group][S][d_model], everything else as adx_sample_dit (noise and out have B rows).  The state
 * encoder and every adaLN modulation run once per scenario instead of once per sample (MM-DiT: its condition stream
 * evolves with the sample and stays per sample; the modulation tables of both streams are shared); results are those of
 * adx_sample_dit on the repeated state.  B must be a multiple of group. */
int adx_sample_dit_grouped(adx_engine* e, int B, int group, const float* state, const float* noise, int n_noise,
                           float* out, unsigned flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADX_H_ */
