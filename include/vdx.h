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
 *   VDX_MODE_BF16 operands rounded to bf16 at LDS staging, fp32 accumulate (v_mfma_f32_16x16x32_bf16) */
typedef enum { VDX_MODE_F32 = 0, VDX_MODE_BF16 = 1 } vdx_mode;

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
} vdx_conv_desc;

/* nnx.Conv / nnx.ConvTranspose on channel-last video (reference: modules.py:162-179 Block.proj + norm
 * + scale/shift + SiLU; utils.py:103-125 Up/Downsample; modules.py:219-222 res_conv). */
int vdx_conv_forward(int mode, const vdx_conv_desc* d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VDX_H_ */
