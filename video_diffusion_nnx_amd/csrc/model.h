// Host-side model description (non-ABI).
#pragma once
#include <string>
#include <vector>
#include "vdx_internal.h"

namespace vdx {

struct ParamInfo { std::string name; int ndim; long shape[6]; long offset; long numel; };
struct Slot { std::string name; long floats_per_sample; long offset_per_sample; };

// one tensor of vdx_pack_params / vdx_pack_params_bwd: kind 0 = [taps][Cin][Cout] -> [taps][Cout][Pad] (MFMA element type),
// 1 = transposed + tap-reversed [taps][Cin (rows)][Pad], 2 = plain fp32 copy of n floats; src in floats, dst in bytes
// 3 = three [Cin][Cout] projection kernels transposed and concatenated along K: [Cin rows][3*Cout] (src, src1, src2)
struct PackJob { long src, dst, n; int taps, Cin, Cout, Pad, kind, pad_; long src1, src2; };

struct ResP {
    std::string name;
    int cin = 0, cout = 0; bool has_mlp = false, has_res = false; int ss_index = -1;
    long mlp_w = 0, mlp_b = 0, n1_s, n1_b, b1_w, b1_b, b1_gs, b1_gb, b2_w, b2_b, b2_gs, b2_gb, rc_w, rc_b, n2_s, n2_b;   // float offsets
    size_t pk_b1, pk_b2, pk_rc;                                                                               // packed byte offsets
    size_t pt_b1 = 0, pt_b2 = 0, pt_rc = 0;                                                                   // transposed packing (backward)
    int s_y1, s_y2, s_rc, s_out, st1, st2;                                                                    // workspace slots / stats slabs
};
struct AttnP { std::string name; int C; long norm_s, norm_b, w[3], b[3], o_w, o_b; size_t pk_qkv, pk_bqkv, pk_o; size_t pt_w[3], pt_o, pt_qkv; };
struct SlaP { std::string name; int C; long norm_s, norm_b, w[3], o_w; size_t pk[3], pk_o; size_t pt_w[3], pt_o, pt_qkv; };
struct Level {
    int cin, cout, lvl; ResP res0, res1; bool has_sla; SlaP sla; AttnP attn; bool has_resample; long rs_w, rs_b; size_t pk_rs; size_t pt_rs = 0;
    int s_sla, s_attn, s_rs;
};

struct Model {
    vdx_config cfg;
    int mode, init_dim, out_dim, time_dim, temb_dim;
    int attn_fp8 = 0;                    // bf16 mode only: QK^T / PV of the <= 16-token attention cores on fp8 operands (forward only)
    int act16 = 0;                       // bf16 mode only: store every inter-kernel activation as bf16.  1 = inference (every fusion on); 2 = training forward
                                         // (every slot vdx_unet_backward reads is materialised: the 1x1 res_conv output is not folded into the block tail)
    std::vector<ParamInfo> params; long param_total = 0;
    size_t packed_bytes = 0;
    size_t packed_t_bytes = 0;
    std::vector<Slot> slots; long act_floats_per_sample = 0;
    std::vector<SsLayer> ss_layers; long ss_floats_per_sample = 0; SsLayer* d_ss_layers = nullptr;
    std::vector<PackJob> pack_jobs, pack_t_jobs; PackJob* d_pack_jobs = nullptr; PackJob* d_pack_t_jobs = nullptr;   // device copies owned by the handle
    int n_stats = 0;
    size_t sla_ws_bytes_per_sample = 0;
    long rel_pos_emb, init_w, init_b, t_w1, t_b1, t_w2, t_b2, null_cond, fin_w, fin_b;
    AttnP init_attn, mid_sattn, mid_tattn;
    ResP mid1, mid2, fin;
    std::vector<Level> downs, ups;
    int s_init, s_init_attn, s_mid_sattn, s_mid_tattn;
};

int model_build(Model* m);
void model_build_pack_tables(Model* m);        // fills pack_jobs / pack_t_jobs (host); the handle uploads them
size_t model_workspace_bytes(const Model* m, int B);
hipError_t model_pack(const Model* m, const float* params, void* packed, hipStream_t st);
size_t model_bwd_workspace_bytes(const Model* m, int B);
hipError_t model_pack_t(const Model* m, const float* params, void* packed_t, hipStream_t st);
// Reverse pass over stages [stage_lo, stage_hi] (descending; 2n+2 = head ... 0 = stem).  A full pass calls the stages in
// order; `state` carries the running activation gradient between calls.
// `side`: the handle's second stream for the weight-gradient kernels (leaves of the reverse graph: they run beside the data-gradient
// chain of the caller's stream; model_bwd.hip) + a ring of events for the fork / join edges.  Created on first use, freed by bwd_state_free.
constexpr int BWD_EVENTS = 64;
struct BwdState {
    float* g = nullptr; int next_stage = -1;
    hipStream_t side = nullptr; hipEvent_t ev[BWD_EVENTS] = {}; int ev_next = 0;
};
void bwd_state_free(BwdState* s);
int model_backward(const Model* m, BwdState* state, const float* params, const void* packed, const void* packed_t, const float* x,
                   const int* time, const float* cond, const unsigned char* cond_mask, int null_all, const float* d_out,
                   void* fwd_workspace, void* bwd_workspace, size_t bwd_workspace_bytes, float* grads, int stage_hi, int stage_lo,
                   int B, hipStream_t st);
int model_forward(const Model* m, const float* params, const void* packed, const float* x, const int* time,
                  const float* cond, const unsigned char* cond_mask, int null_all, float* out, void* workspace,
                  size_t workspace_bytes, int B, hipStream_t st);

}  // namespace vdx
