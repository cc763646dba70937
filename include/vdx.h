/* vdx.h -- C ABI of the MI355X-native video-diffusion hot path (libvdx.so).
 *
 * The reference (maxsonate/video-diffusion-nnx) has NO FFI: its device boundary is the XLA runtime
 * under jax/flax.  This header is therefore the boundary the new path defines underneath the
 * reference's Python surface (Unet3D / GaussianDiffusion / Trainer); each entry cites the reference
 * function whose device work it replaces.  The reference-side binding is the ctypes layer in
 * video_diffusion_nnx_amd/_lib.py.
 *
 * Conventions
 *   - every function returns int: 0 = VDX_OK, negative = vdx_status; vdx_last_error() gives the text
 *     (thread-local);
 *   - ALL tensor memory is caller-owned device memory (the PyTorch allocator in the Python host);
 *     the library owns only the handle and never frees or allocates caller tensors;
 *   - pointers are device pointers on the current HIP device, contiguous, 16-byte aligned, fp32 unless
 *     stated; layouts are documented per call;
 *   - no hidden synchronisation: work is enqueued on the caller's stream (a hipStream_t passed as
 *     void*), the calls are graph-capturable (no malloc/free/sync inside);
 *   - a handle is not thread-safe: one host thread per rank / GPU.
 *
 * Tensor layouts: external video tensors are [B, C, F, H, W] (as the reference's public API);
 * internal activations are channel-last [B, F, H, W, C] (as the reference after unet3d.py:280).
 */
#ifndef VDX_H_
#define VDX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    VDX_OK = 0,
    VDX_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    VDX_ERR_HIP = -2,         /* a HIP runtime call failed */
    VDX_ERR_NOMEM = -3,       /* workspace too small */
    VDX_ERR_STATE = -4        /* call sequence error (e.g. backward without forward) */
} vdx_status;

/* Arithmetic of the MFMA contractions (activations and master weights are fp32 in HBM either way):
 *   VDX_MODE_F32  exact fp32 products (v_mfma_f32_16x16x4_f32)  -- the parity mode
 *   VDX_MODE_BF16 operands rounded to bf16 at LDS staging, fp32 accumulate (v_mfma_f32_16x16x32_bf16)
 *   VDX_MODE_F16  operands rounded to IEEE fp16 at LDS staging, fp32 accumulate (v_mfma_f32_16x16x32_f16): the "fp16" of
 *                 BASELINE.json configs[3]; generic kernels only (every tensor stays fp32 in HBM, no bf16 activation storage) */
typedef enum { VDX_MODE_F32 = 0, VDX_MODE_BF16 = 1, VDX_MODE_F16 = 2 } vdx_mode;

const char* vdx_last_error(void);
int vdx_version(void);

/* ------------------------------------------------------------------------------------------------
 * Operator-level entry points (used by the per-block parity tests; the network entry points below
 * are built from the same launchers).
 * ---------------------------------------------------------------------------------------------- */

/* Packs a Flax kernel [taps][Cin][Cout] (fp32) into the MFMA staging layout [taps][Cout][CinPad]
 * (fp32 or bf16 per mode, zero padded).  vdx_packed_conv_bytes gives the destination size. */
size_t vdx_packed_conv_bytes(int mode, int taps, int cin, int cout);
int vdx_pack_conv_weights(int mode, const float* kernel, void* packed, int taps, int cin, int cout, void* stream);

/* Size in bytes of one GroupNorm statistics slab for `batch` samples and `groups` groups. */
size_t vdx_gn_stats_bytes(int batch, int groups);

typedef struct {
    /* input = concat(x0 [B*F,H,W,c0], x1 [B*F,H,W,c1]) on channels (x1 may be NULL with c1 = 0) */
    const float* x0; const float* x1; int c0, c1;
    const void* packed_w;           /* from vdx_pack_conv_weights */
    const float* bias;              /* [cout] or NULL */
    float* y;                       /* [B*F, Ho, Wo, cout] */
    int cout;
    int batch, frames, h, w;
    int kind;                       /* 0: Conv(1,kh,kw) SAME, stride (1,s,s); 1: ConvTranspose(1,4,4)/(1,2,2) SAME */
    int kh, kw, stride;
    /* optional prologue on x0 (c1 must be 0): SiLU(GroupNorm(x0; in_stats, gamma, beta) * (scale+1) + shift) */
    const double* in_stats; const float* gamma; const float* beta; int groups;
    const float* scale_shift; int scale_shift_stride;   /* rows [scale(c0) | shift(c0)] per sample, or NULL */
    /* optional epilogue: accumulate GroupNorm partial statistics of y into out_stats (pre-zeroed) */
    double* out_stats; int out_groups;
    /* storage of the activation tensors (VDX_MODE_BF16 only): nonzero = the tensor holds bf16 elements instead of fp32 */
    int x_bf16;                     /* x0 and x1 */
    int y_bf16;
    /* optional residual added to the output, y = conv + bias + res ([.., cout] like y; the attention / SLA `to_out` projections of
     * the wide levels run as 1x1 convs with the block input as res: modules.py:312-326, unet3d.py:86-96 Residual) */
    const void* res; int res_bf16;
} vdx_conv_desc;

/* nnx.Conv / nnx.ConvTranspose on channel-last video (reference: modules.py:162-179 Block.proj + norm
 * + scale/shift + SiLU; utils.py:103-125 Up/Downsample; modules.py:219-222 res_conv). */
int vdx_conv_forward(int mode, const vdx_conv_desc* d, void* stream);

/* Instrumentation (bench.py's roofline leg): a process-global hook the library calls immediately before (phase 0) and after
 * (phase 1) EVERY kernel launch of the forward / sampling path -- inside vdx_unet_forward / vdx_p_sample_loop too -- so that a caller
 * can bracket each launch with its own events on `stream` while the real step runs.  `kernel` = the __global__ function's name
 * without template arguments (a substring of the name rocprofv3 prints); `shape` = its template arguments + the operand shape as
 * text (the key bench.py groups launches by: one symbol serves several levels of the network); flops / bytes = the ALGORITHMIC work
 * of this launch (SURVEY 8d op-level definition: 2*M*N*K per contraction; input + output tensors in their storage type + weights).
 * The strings are only valid during the call.  NULL disables.  Not for use while a stream is being captured into a graph (run the
 * loop with use_graph = 0). */
typedef struct {
    const char* kernel;
    const char* shape;
    double flops, bytes;
} vdx_launch_info;
typedef void (*vdx_launch_hook)(void* user, int phase, const vdx_launch_info* info, void* stream);
void vdx_set_launch_hook(vdx_launch_hook hook, void* user);

/* ResnetBlock tail: out = SiLU(GroupNorm(y2; stats, gn_gamma, gn_beta)) + LayerNorm_C(r; ln_gamma, ln_beta)
 * (reference: modules.py:173-179 for Block 2 and :240-243 `h + norm_2(res_conv(x))`).  r is res_conv(x), or x itself
 * when the block has no res_conv.  All tensors channel-last [batch, pix_per_sample, c]. */
int vdx_resblock_tail(const float* y2, const float* r, float* out, const double* stats, const float* gn_gamma,
                      const float* gn_beta, int groups, const float* ln_gamma, const float* ln_beta, int c, int batch,
                      long pix_per_sample, void* stream);

/* ResnetBlock tail with the block's 1x1 res_conv inside (reference: modules.py:219-222 `res_conv` + :240-243
 * `h + norm_2(res_conv(x))`), bf16 tensors (bf16 activation storage of a bf16-mode network):
 * out = SiLU(GroupNorm(y2)) + LayerNorm_C(concat(x0[..,c0], x1[..,c1]) . W + rc_bias).  y2, x0, x1, out: bf16 channel-last;
 * rc_w_packed: bf16 [c][c0 + c1] (row = output channel, K-contiguous: the packing vdx_pack_params produces for this conv);
 * x1 may be NULL with c1 = 0.  Shapes served: (c0 + c1, c) in {(128,64), (64,128), (256,64), (128,256)}, c1 = 0 or c1 = c0,
 * pix_per_sample a multiple of 16; anything else returns VDX_ERR_INVALID. */
int vdx_resblock_tail_rc_bf16(const void* y2, const void* x0, const void* x1, int c0, int c1, const void* rc_w_packed,
                              const float* rc_bias, void* out, const double* stats, const float* gn_gamma, const float* gn_beta,
                              int groups, const float* ln_gamma, const float* ln_beta, int c, int batch, long pix_per_sample,
                              void* stream);

/* Block prologue as its own pass, in place on a bf16 tensor (reference: modules.py:171-179, Block.__call__: GroupNorm, the
 * time-embedding `x * (scale + 1) + shift`, SiLU): y <- SiLU(GroupNorm(y) * (scale + 1) + shift).  y bf16 channel-last
 * [batch, pix_per_sample, c]; stats as the producing conv's epilogue leaves them (vdx_resblock_tail); scale_shift NULL (Block 2) or
 * fp32 [batch][ss_stride] rows holding scale[c] | shift[c].  c a multiple of 8, c <= 1024.  The sampling forward runs it in front of
 * the wide (c >= 256) second convs of bf16-storage networks. */
int vdx_gn_silu_apply_bf16(void* y, const double* stats, const float* gn_gamma, const float* gn_beta, const float* scale_shift,
                           int ss_stride, int groups, int c, int batch, long pix_per_sample, void* stream);

/* init_conv (reference: unet3d.py:110-115,282): x EXTERNAL layout [B,Cin,F,H,W], Flax kernel (1,k,k,Cin,Cout) fp32,
 * y channel-last [B,F,H,W,Cout]. */
int vdx_init_conv(const float* x, const float* kernel, const float* bias, float* y, int batch, int cin, int frames,
                  int h, int w, int cout, int k, void* stream);

/* final 1x1 conv (reference: unet3d.py:251): x [npix, d] channel-last, Flax kernel (1, d, cout), y [npix, cout]. */
int vdx_final_conv(const float* x, const float* kernel, const float* bias, float* y, long npix, int d, int cout, void* stream);

/* time_mlp (reference: modules.py:30-45 SinusoidalPosEmb; unet3d.py:128-133,288 Linear-GELU-Linear; :291-298 cond mix).
 * temb [batch, time_dim + cond_dim].  cond/null_cond_emb/cond_mask may be NULL when cond_dim == 0;
 * cond_mask (bytes, 1 = use null_cond_emb) overrides null_all. */
int vdx_time_mlp(const int* time, const float* w1, const float* b1, const float* w2, const float* b2, int dim,
                 const float* cond, const float* null_cond_emb, const unsigned char* cond_mask, int null_all, int cond_dim,
                 float* temb, int batch, void* stream);

/* Multi-head self-attention + residual (reference: modules.py:247-326 inside Residual(PreNorm(EinopsToAndFrom(..)))
 * where PreNorm is a no-op, unet3d.py:86-96 temporal / :196-208 bottleneck spatial).  x, y channel-last [B,F,H,W,C].
 * temporal != 0: sequences over F per (b,h,w); else sequences over (h w) per (b,f).  dim_head must be 32.
 * wqkv_packed: vdx_pack_conv_weights of the [C, 3*heads*32] matrix (q|k|v column blocks); bqkv [3*heads*32];
 * wo_packed: packed [heads*32, C] matrix; bo [C]. */
int vdx_attention_forward(int mode, const float* x, float* y, const void* wqkv_packed, const float* bqkv,
                          const void* wo_packed, const float* bo, int batch, int frames, int h, int w, int c, int heads,
                          int temporal, void* stream);
/* Same; fp8_core != 0 (VDX_MODE_BF16 only): QK^T and PV of sequences of <= 16 tokens on e4m3 operands (see vdx_set_attention_fp8;
 * longer sequences ignore the flag). */
int vdx_attention_forward_ex(int mode, const float* x, float* y, const void* wqkv_packed, const float* bqkv,
                             const void* wo_packed, const float* bo, int batch, int frames, int h, int w, int c, int heads,
                             int temporal, int fp8_core, void* stream);

/* Same block on bf16 TENSORS (the form the network runs under bf16 activation storage, vdx_set_activation_storage): x and y are
 * channel-last bf16 [batch, frames, h, w, c]; VDX_MODE_BF16 operands.  The level-0 shape of the N config (c = 64, 8 heads, temporal,
 * 16 frames) runs attention_w_kernel (one wave per group of 4 sequences), other shapes the kernels of vdx_attention_forward. */
int vdx_attention_forward_bf16(const void* x_bf16, void* y_bf16, const void* wqkv_packed, const float* bqkv,
                               const void* wo_packed, const float* bo, int batch, int frames, int h, int w, int c, int heads,
                               int temporal, int fp8_core, void* stream);

/* SpatialLinearAttention + residual (reference: modules.py:64-129 inside Residual(PreNorm(..)), unet3d.py:170-178).
 * heads must be 8, head dim 32.  wq/wk/wv_packed: packed [C,256]; wo_packed: packed [256,C].
 * workspace: vdx_sla_workspace_bytes(mode, batch*frames, h*w, heads). */
size_t vdx_sla_workspace_bytes(int mode, int nframes, int npix, int heads);
int vdx_sla_forward(int mode, const float* x, float* y, const void* wq_packed, const void* wk_packed, const void* wv_packed,
                    const void* wo_packed, void* workspace, int batch, int frames, int h, int w, int c, int heads, void* stream);

/* The same block on bf16 channel-last tensors (bf16 activation storage; VDX_MODE_BF16 operands).  c = 64 with >= 128 frames runs the
 * second half on sla_out_w_kernel (one wave per 64 pixels), other shapes the kernels of vdx_sla_forward.  workspace:
 * vdx_sla_workspace_bytes(VDX_MODE_BF16, batch*frames, h*w, heads). */
int vdx_sla_forward_bf16(const void* x_bf16, void* y_bf16, const void* wq_packed, const void* wk_packed, const void* wv_packed,
                         const void* wo_packed, void* workspace, int batch, int frames, int h, int w, int c, int heads, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Network-level entry points.
 * ---------------------------------------------------------------------------------------------- */

/* Mirror of the reference constructors' arguments that shape the device work:
 * Unet3D(dim, rngs, dim_mults, cond_dim, out_dim, channels, attn_heads, attn_dim_head, use_bert_text_cond,
 *        init_dim, init_kernel_size, use_sparse_linear_attn, block_type, resnet_groups)   unet3d.py:58-75
 * GaussianDiffusion(image_size, num_frames, ...)                                          gaussian_diffusion.py:53-65
 * (use_bert_text_cond is resolved by the host to cond_dim = 768; 0 means "no conditioning"). */
typedef struct {
    int dim;
    int n_mults; int dim_mults[8];
    int channels;
    int out_dim;                 /* 0 = channels */
    int cond_dim;                /* 0 = none */
    int attn_heads, attn_dim_head;
    int init_dim;                /* 0 = dim */
    int init_kernel_size;
    int use_sparse_linear_attn;
    int resnet_groups;
    int image_size, num_frames;
    int mode;                    /* vdx_mode */
} vdx_config;

typedef struct vdx_handle vdx_handle;

int vdx_create(const vdx_config* cfg, vdx_handle** out);
void vdx_destroy(vdx_handle* h);

/* Activation storage of vdx_unet_forward / vdx_p_sample_loop.  0 (default): every inter-kernel activation is fp32 in the
 * workspace (required by vdx_unet_backward, and what vdx_slot_info describes).  1 (VDX_MODE_BF16 handles only): they are
 * stored as bf16 -- half the HBM traffic of the bandwidth-bound levels; arithmetic stays fp32-accumulate.  The reference
 * has no such knob (XLA picks its own buffer types); this is the "bf16" of BASELINE.json's sampling configuration. */
int vdx_set_activation_storage(vdx_handle* h, int bf16);
int vdx_get_activation_storage(const vdx_handle* h);

/* fp8 attention (BASELINE.json configs[4]: "fp8 attention QK^T / PV on CDNA4"; no reference counterpart -- XLA picks its own types).
 * VDX_MODE_BF16 handles only, forward / sampling only.  1: the QK^T and PV products of every attention block over <= 16 tokens (all
 * temporal attention blocks of the configured shapes) round their operands (q, k, v, softmax probabilities) to OCP e4m3 and run on
 * v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulate and fp32 softmax; projections stay bf16.  Longer sequences (the 64-token spatial block
 * of the bottleneck) keep bf16 operands.  Tolerance: tests/test_gpu_blocks.py (attention block 4e-2 of the attention branch). */
int vdx_set_attention_fp8(vdx_handle* h, int on);
int vdx_get_attention_fp8(const vdx_handle* h);

/* Flat fp32 parameter buffer layout (names = nnx state-tree paths, shapes = Flax shapes). */
int vdx_param_count(const vdx_handle* h);
long vdx_param_total(const vdx_handle* h);                       /* floats in the flat buffer */
int vdx_param_info(const vdx_handle* h, int index, char* name, int name_cap, int* ndim, long shape[6], long* offset);

/* Derived (packed) weights in the MFMA staging layout; re-run after every parameter update. */
size_t vdx_packed_bytes(const vdx_handle* h);
int vdx_pack_params(const vdx_handle* h, const float* params, void* packed, void* stream);

/* Activation workspace for a batch; every intermediate keeps its own slot (inspectable for parity tests). */
size_t vdx_workspace_bytes(const vdx_handle* h, int batch);
int vdx_slot_count(const vdx_handle* h);
int vdx_slot_info(const vdx_handle* h, int index, char* name, int name_cap, long* floats_per_sample, long* float_offset_per_sample);

/* Unet3D.__call__ (reference: unet3d.py:262-387).  x [B,C,F,H,W]; time [B] int32 (device); cond [B,cond_dim] or NULL;
 * cond_mask [B] bytes (1 = replace by null_cond_emb) or NULL, in which case null_all selects all/none
 * (null_cond_prob 1 / 0); out channel-LAST [B,F,H,W,out_dim] (unet3d.py:387). */
int vdx_unet_forward(const vdx_handle* h, const float* params, const void* packed, const float* x, const int* time,
                     const float* cond, const unsigned char* cond_mask, int null_all, float* out, void* workspace,
                     size_t workspace_bytes, int batch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GaussianDiffusion device work.  External tensors [B,C,F,H,W]; eps_hat is the UNet output, channel-last.
 * ---------------------------------------------------------------------------------------------- */

/* N(0,1) stream replacing jax.random.normal (gaussian_diffusion.py:254,309,416,445): Philox4x32-10 keyed by `seed`,
 * draw number = offset (+ *dev_offset when given, a device counter advanced by the sampling loop). */
int vdx_randn(float* out, long n, uint64_t seed, uint64_t offset, const uint64_t* dev_offset, void* stream);

/* q_sample (gaussian_diffusion.py:401-420) with normalize_img folded in: x_t = sqrt_ac[t] (x0*pre_scale+pre_shift)
 * + sqrt_1m_ac[t] noise.  t device int32 [B]; tables are device fp32 [T]. */
int vdx_q_sample(const float* x_start, const int* t, const float* noise, float* out, const float* sqrt_ac,
                 const float* sqrt_one_minus_ac, int batch, long per_sample, float pre_scale, float pre_shift, void* stream);

/* p_sample (gaussian_diffusion.py:231-261 incl. predict_start_from_noise :120-136, clip :203-220, q_posterior :139-159).
 * tables: device fp32 [5][T] = sqrt_recip_ac | sqrt_recipm1_ac | posterior_mean_coef1 | posterior_mean_coef2 |
 * posterior_log_variance_clipped.  noise NULL -> Philox(seed, offset + *dev_offset) (per_sample % 4 == 0 required).
 * thres: per-sample dynamic-threshold s [B] or NULL.  x and out may alias. */
int vdx_p_sample_step(const float* x, const float* eps_hat, float* out, const int* t, const float* tables, int timesteps,
                      const float* noise, uint64_t seed, uint64_t offset, const uint64_t* dev_offset, const float* thres,
                      int clip_denoised, int batch, int channels, long per_sample, void* stream);

/* sum |eps_hat - noise| (l2 == 0) or (eps_hat - noise)^2 (l2 != 0) accumulated into *acc (device double, pre-zeroed);
 * the mean of gaussian_diffusion.py:463-466 is acc / (batch*channels*fhw). */
int vdx_loss_sum(const float* eps_hat, const float* noise, double* acc, int batch, int channels, long fhw, int l2, void* stream);

/* y = a x + b (normalize_img / unnormalize_img, utils.py:259-280). */
int vdx_affine(const float* x, float* y, long n, float a, float b, void* stream);

/* p_sample_loop (gaussian_diffusion.py:264-320): img holds x_T on entry and x_0 (still in [-1,1]) on return.
 * t_dev [B] int32 must hold T-1, *step_dev (device uint64) must be 0 on entry; eps_buf [B,F,H,W,out_dim] scratch.
 * One step = Unet3D forward + p_sample + (t -= 1); with use_graph != 0 the step is captured once into a hipGraph on
 * `stream` (which must not be the legacy default stream) and replayed, so the loop issues no per-step host work.
 * Draw k of the noise stream is Philox(seed, 1 + k); the caller draws x_T itself (e.g. vdx_randn with offset 0).
 * nsteps (<= timesteps) steps are enqueued, continuing from the state in t_dev / step_dev, so a loop may be issued in
 * pieces.  The instantiated graph is cached in the handle and reused while every argument but nsteps is unchanged. */
int vdx_p_sample_loop(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                      uint64_t* step_dev, const float* tables, int timesteps, int nsteps, const float* cond, uint64_t seed,
                      int clip_denoised, void* workspace, size_t workspace_bytes, int batch, int use_graph, void* stream);

/* Dynamic thresholding of p_mean_variance (gaussian_diffusion.py:205-217): s[b] = max(quantile(|x0_hat_b|, percentile), 1) with
 * x0_hat = predict_start_from_noise(x, t, eps_hat) and the linearly interpolated quantile (jnp.quantile default).  tables as in
 * vdx_p_sample_step (rows 0 and 1 are used).  thres_out [B]. */
int vdx_dynamic_threshold(const float* x, const float* eps_hat, const int* t, const float* tables, int timesteps, float percentile,
                          float* thres_out, int batch, int channels, long per_sample, void* stream);

/* vdx_p_sample_loop with use_dynamic_thres (gaussian_diffusion.py:205-217) inside the captured step: thres_buf [B] floats scratch,
 * percentile in (0, 1].  percentile <= 0 is exactly vdx_p_sample_loop. */
int vdx_p_sample_loop_dyn(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                          uint64_t* step_dev, const float* tables, int timesteps, int nsteps, const float* cond, uint64_t seed,
                          int clip_denoised, float percentile, float* thres_buf, void* workspace, size_t workspace_bytes, int batch,
                          int use_graph, void* stream);

/* DDIM sampling, eta = 0 (EXTENSION: the reference has ancestral DDPM sampling only; BASELINE.json configs[3] names "DDIM-100").
 * One step: x0 = (x - sqrt(1-ac_t) eps) / sqrt(ac_t), clipped like p_sample; eps' re-derived from the clipped x0;
 * out = sqrt(ac_next) x0 + sqrt(1-ac_next) eps'.  seq: device int32 [n + 1] = the time sequence t_0 > t_1 > ... > t_{n-1}, -1
 * (-1 = the data, alpha_bar 1); step k uses (seq[k], seq[k+1]) with k = *step_dev (device uint64) or 0 when step_dev is NULL.
 * alphas_cumprod: device fp32 [T].  x and out may alias. */
int vdx_ddim_step(const float* x, const float* eps_hat, float* out, const float* alphas_cumprod, const int* seq,
                  const uint64_t* step_dev, const float* thres, int clip_denoised, int batch, int channels, long per_sample, void* stream);

/* The DDIM loop: nsteps x { Unet3D forward at t = seq[k] ; vdx_ddim_step ; t = max(seq[k+1], 0), k += 1 }, captured once in a
 * hipGraph like vdx_p_sample_loop.  t_dev [B] must hold seq[0] and *step_dev 0 on entry; seq_len = n (seq has n + 1 entries). */
int vdx_ddim_sample_loop(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                         uint64_t* step_dev, const float* alphas_cumprod, const int* seq, int seq_len, int nsteps, const float* cond,
                         int clip_denoised, void* workspace, size_t workspace_bytes, int batch, int use_graph, void* stream);

/* vdx_ddim_sample_loop with use_dynamic_thres inside the captured step (the threshold of gaussian_diffusion.py:205-217 applied to DDIM's
 * x0): tables = the [5][T] tables of vdx_p_sample_step (rows 0, 1 are read), thres_buf [B] floats scratch, percentile in (0, 1].
 * percentile <= 0 is exactly vdx_ddim_sample_loop. */
int vdx_ddim_sample_loop_dyn(vdx_handle* h, const float* params, const void* packed, float* img, float* eps_buf, int* t_dev,
                             uint64_t* step_dev, const float* alphas_cumprod, const int* seq, int seq_len, int nsteps, const float* cond,
                             int clip_denoised, const float* tables, int timesteps, float percentile, float* thres_buf, void* workspace,
                             size_t workspace_bytes, int batch, int use_graph, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward building blocks (autodiff of the forward operators; reference trainer.py:361 jax.value_and_grad).
 * ---------------------------------------------------------------------------------------------- */

/* Transposed, tap-reversed packing [taps][Cin][CoutPad]: vdx_conv_forward(dy, this packing, cout = Cin) is the data
 * gradient of a stride-1 conv; with kind swapped (Conv 4x4/s2 <-> ConvTranspose) it is the data gradient of
 * Downsample / Upsample.  Size: vdx_packed_conv_bytes(mode, taps, cout, cin). */
int vdx_pack_conv_weights_t(int mode, const float* kernel, void* packed, int taps, int cin, int cout, void* stream);

typedef struct {
    const float* x0; const float* x1; int c0, c1;      /* the conv's input (concat on channels) */
    const float* dy; int cout;                         /* gradient of the conv's output [B*F, Ho, Wo, cout] */
    float* dw;                                         /* Flax layout [taps][c0+c1][cout] fp32, ACCUMULATED (atomics) */
    int batch, frames, h, w;                           /* input geometry */
    int kind, kh, kw, stride;                          /* as vdx_conv_desc */
    const double* in_stats; const float* gamma; const float* beta; int groups;     /* optional fused prologue of the forward */
    const float* scale_shift; int scale_shift_stride;
    int bf16_operands;                                 /* 0: exact-f32 MFMA; 1: operands rounded to bf16, fp32 accumulate (what a
                                                          VDX_MODE_BF16 handle's backward uses) */
} vdx_wgrad_desc;

/* dW += Xhat^T (*) dY on exact-f32 MFMA (both arithmetic modes use it).  This block-level entry point has no workspace and accumulates with
 * float atomics (the result is reproducible to rounding); inside vdx_unet_backward the same kernels store per-workgroup partial tiles
 * into the backward workspace and a second pass adds them in a fixed order (bit-reproducible gradients, round 3). */
int vdx_conv_backward_weights(const vdx_wgrad_desc* d, void* stream);

/* Backward of act = SiLU((gamma*GroupNorm(y)+beta)*(1+s)+sh) and, when r != NULL, of out = act + LayerNorm_C(r)
 * (reference forward: modules.py:171-179,233-243).  dact/y/dy/r/dr channel-last [batch, pix_per_sample, c];
 * stats = the forward's GroupNorm statistics slab; scale_shift rows [s(c)|sh(c)] or NULL; d_gamma/d_beta/d_ln_* are
 * ACCUMULATED; dss [batch][2c] (ds|dsh) written when non-NULL; scratch >= vdx_norm_act_backward_scratch_floats(c, batch,
 * pix_per_sample) floats, uninitialised (per-workgroup partial sums of the reduction pass + the per-group terms). */
size_t vdx_norm_act_backward_scratch_floats(int c, int batch, long pix_per_sample);
int vdx_norm_act_backward(const float* dact, const float* y, float* dy, const double* stats, const float* gamma, const float* beta,
                          int groups, const float* scale_shift, int scale_shift_stride, float* d_gamma, float* d_beta, float* dss,
                          const float* r, const float* ln_gamma, float* dr, float* d_ln_gamma, float* d_ln_beta, float* scratch,
                          int c, int batch, long pix_per_sample, void* stream);

/* Attention core backward (autodiff of modules.py:294-323 per sequence and head).  qkv [npix][3*heads*32] = x Wqkv + b
 * (q unscaled), d_o [npix][heads*32] = dy Wo^T; writes o (attention output before the out projection), dq, dk, dv
 * [npix][heads*32].  temporal != 0: sequences over F per (b,h,w), else over (h w) per (b,f). */
int vdx_attention_core_backward(const float* qkv, const float* d_o, float* o, float* dq, float* dk, float* dv, int batch, int frames,
                                int h, int w, int heads, int temporal, void* stream);
/* Same with a choice of arithmetic: bf16_operands != 0 = the form a VDX_MODE_BF16 handle's backward uses for sequences of <= 16
 * tokens (q, k, v, d_o rounded to bf16, every product on MFMA, fp32 accumulate); 0 = exact fp32. */
int vdx_attention_core_backward_ex(const float* qkv, const float* d_o, float* o, float* dq, float* dk, float* dv, int batch, int frames,
                                   int h, int w, int heads, int temporal, int bf16_operands, void* stream);

/* Whole backward of a temporal attention block y = MHA(x) + x (modules.py:271-327 under Residual, unet3d.py:284,308,365) except its
 * weight gradients, in one pass over x and dy: the shape a VDX_MODE_BF16 handle's backward runs at the widest level (C = 64 channels,
 * 8 heads x 32, sequences over <= 16 frames per (b, h, w); bf16 operands, fp32 accumulate).  x, dy, dx: fp32 [B][F][h][w][64];
 * packed_wqkv = vdx_pack_conv_weights(BF16) of the [64][q|k|v = 768] kernel, bqkv [768]; packed_wo_t =
 * vdx_pack_conv_weights_t(BF16) of the [256][64] out kernel.  Writes dx = dy + d(q|k|v) Wqkv^T and, as bf16 tensors for the
 * weight-gradient kernels, o [rows][256] (attention output before the out projection) and dqkv [rows][768]. */
int vdx_temporal_attention_backward_fused(const float* x, const float* dy, const void* packed_wqkv, const float* bqkv, const void* packed_wo_t,
                                          void* o_bf16, void* dqkv_bf16, float* dx, int batch, int frames, int h, int w, void* stream);

/* SpatialLinearAttention core backward (autodiff of modules.py:105-118 per frame and head; heads = 8, D = 32).
 * q, k, v, d_out [B*F*h*w][256]; writes o (pre to_out), dq, dk, dv.  scratch >= vdx_sla_backward_scratch_floats floats. */
size_t vdx_sla_backward_scratch_floats(int nframes, int heads);
int vdx_sla_core_backward(const float* q, const float* k, const float* v, const float* d_out, float* o, float* dq, float* dk, float* dv,
                          float* scratch, int nframes, int npix, int heads, void* stream);
/* Same with a choice of arithmetic for the per-(frame, head) reductions (softmax-over-pixels statistics, ctx, dctx):
 * bf16_operands != 0 = bf16 MFMA with fp32 accumulate (what a VDX_MODE_BF16 handle's backward uses); 0 = exact fp32. */
int vdx_sla_core_backward_ex(const float* q, const float* k, const float* v, const float* d_out, float* o, float* dq, float* dk, float* dv,
                             float* scratch, int nframes, int npix, int heads, int bf16_operands, void* stream);

/* out[c] += sum_rows x[row][c]  (bias gradients). */
int vdx_colsum(const float* x, float* out, long rows, int c, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Train step (reference trainer.py:322-392).
 * ---------------------------------------------------------------------------------------------- */

/* Unet3D backward (jax.value_and_grad of unet3d.py:262-387, reference trainer.py:361).  Must follow a vdx_unet_forward of the
 * same inputs on the same fwd_workspace (every intermediate is read back from its slot).  The reverse walk is cut in
 * vdx_num_stages() stages (head = num_stages-1, ups, mid, downs, stem = 0) so that the caller can all-reduce finished
 * gradient buckets while earlier stages still run: call with descending, contiguous [stage_hi .. stage_lo] ranges, starting
 * at the head (which zeroes `grads`).  grads: flat fp32, same layout as params (vdx_param_info).
 * packed_t: vdx_pack_params_bwd (transposed packing for the data gradients).
 * The gradients are bit-reproducible run to run (round 3): no sum of the pass depends on the arrival order of workgroups or of the
 * two streams it runs on (per-workgroup partial slots in bwd_workspace + a fixed-order second pass; vdx_bwd_workspace_bytes includes
 * 2 x 48 MB for them). */
int vdx_num_stages(const vdx_handle* h);
size_t vdx_packed_bwd_bytes(const vdx_handle* h);
int vdx_pack_params_bwd(const vdx_handle* h, const float* params, void* packed_t, void* stream);
size_t vdx_bwd_workspace_bytes(const vdx_handle* h, int batch);
int vdx_unet_backward(vdx_handle* h, const float* params, const void* packed, const void* packed_t, const float* x, const int* time,
                      const float* cond, const unsigned char* cond_mask, int null_all, const float* d_out, void* fwd_workspace,
                      void* bwd_workspace, size_t bwd_workspace_bytes, float* grads, int stage_hi, int stage_lo, int batch, void* stream);

/* d(mean loss)/d(eps_hat) for the l1 / l2 loss of gaussian_diffusion.py:463-466, written channel-last like eps_hat. */
int vdx_loss_grad(const float* eps_hat, const float* noise, float* d_eps_hat, int batch, int channels, long fhw, int l2, void* stream);

/* optax.adam + EMA on flat fp32 buffers (trainer.py:367-382): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
 * p -= lr * (m / (1-b1^t)) / (sqrt(v / (1-b2^t)) + eps), t = step_count + 1; g is read as grad * grad_scale
 * (1/world_size after a sum all-reduce).  If do_ema: ema = decay * ema + (1 - decay) * p_new. */
int vdx_adam_ema_step(float* params, const float* grads, float* m, float* v, float* ema, long n, float lr, float b1, float b2,
                      float eps, long step_count, float grad_scale, int do_ema, float ema_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel communicator (reference trainer.py:161-177, 307-320: the batch is sharded over a 'data' mesh axis and XLA inserts
 * the all-reduce of every gradient behind jax.value_and_grad, trainer.py:361 -- SURVEY section 2, collective C1).  Here: one
 * process per GPU, one RCCL communicator per handle, the flat gradient buffer reduced bucket by bucket as the staged backward
 * finishes them.  RCCL is loaded at run time on the first call (no link-time dependency).
 * ---------------------------------------------------------------------------------------------- */
#define VDX_UNIQUE_ID_BYTES 128

/* rank 0 creates the rendezvous id (ncclGetUniqueId) and hands its 128 bytes to the other ranks through any channel it has (a
 * torch.distributed broadcast, a file, MPI); host memory. */
int vdx_comm_unique_id(void* unique_id_out);
/* joins the communicator of `world` ranks as `rank` on the CURRENT device (collective: every rank calls it). */
int vdx_comm_init(vdx_handle* h, int rank, int world, const void* unique_id);
/* in-place SUM all-reduce of count floats (a bucket of the flat gradient buffer) enqueued on `stream`; the mean's 1/world is folded
 * into vdx_adam_ema_step's grad_scale. */
int vdx_allreduce_bucket(vdx_handle* h, float* ptr, size_t count, void* stream);
int vdx_comm_world(const vdx_handle* h);         /* ranks of the handle's communicator, 1 when none */
int vdx_comm_destroy(vdx_handle* h);             /* also done by vdx_destroy */

#ifdef __cplusplus
}
#endif
#endif /* VDX_H_ */
