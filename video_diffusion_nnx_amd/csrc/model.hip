// Host runtime of the Unet3D hot path: parameter layout, weight packing, workspace plan and the
// forward launch sequence (one C call = the whole network enqueued on the caller's stream).
//
// Network structure follows /root/reference/unet3d.py:58-252 (constructor) and :262-387 (forward);
// parameter names follow the nnx state tree so reference checkpoints can be mapped 1:1.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "vdx_common.h"
#include "vdx_internal.h"
#include "model.h"
#include <stdlib.h>

namespace vdx {

static long add_param(Model* m, const std::string& name, std::initializer_list<long> shape) {
    ParamInfo p;
    p.name = name;
    p.ndim = (int)shape.size();
    long n = 1;
    int i = 0;
    for (long s : shape) { p.shape[i++] = s; n *= s; }
    p.offset = m->param_total;
    p.numel = n;
    m->param_total += (n + 3) / 4 * 4;               // keep every tensor 16-byte aligned in the flat buffer
    m->params.push_back(p);
    return p.offset;
}

static size_t add_packed_t(Model* m, size_t bytes) {
    const size_t off = m->packed_t_bytes;
    m->packed_t_bytes += (bytes + 255) / 256 * 256;
    return off;
}

static size_t add_packed(Model* m, size_t bytes) {
    const size_t off = m->packed_bytes;
    m->packed_bytes += (bytes + 255) / 256 * 256;
    return off;
}

static ResP make_res(Model* m, const std::string& pre, int cin, int cout, bool has_mlp) {
    ResP r = ResP();
    r.cin = cin; r.cout = cout; r.has_mlp = has_mlp; r.has_res = cin != cout;
    if (has_mlp) {
        r.mlp_w = add_param(m, pre + ".mlp.layers.1.kernel", {m->temb_dim, 2L * cout});
        r.mlp_b = add_param(m, pre + ".mlp.layers.1.bias", {2L * cout});
    }
    r.n1_s = add_param(m, pre + ".norm_1.scale", {2L * cout});
    r.n1_b = add_param(m, pre + ".norm_1.bias", {2L * cout});
    r.b1_w = add_param(m, pre + ".block_1.proj.kernel", {1, 3, 3, cin, cout});
    r.b1_b = add_param(m, pre + ".block_1.proj.bias", {cout});
    r.b1_gs = add_param(m, pre + ".block_1.norm.scale", {cout});
    r.b1_gb = add_param(m, pre + ".block_1.norm.bias", {cout});
    r.b2_w = add_param(m, pre + ".block_2.proj.kernel", {1, 3, 3, cout, cout});
    r.b2_b = add_param(m, pre + ".block_2.proj.bias", {cout});
    r.b2_gs = add_param(m, pre + ".block_2.norm.scale", {cout});
    r.b2_gb = add_param(m, pre + ".block_2.norm.bias", {cout});
    if (r.has_res) {
        r.rc_w = add_param(m, pre + ".res_conv.kernel", {1, cin, cout});
        r.rc_b = add_param(m, pre + ".res_conv.bias", {cout});
    }
    r.n2_s = add_param(m, pre + ".norm_2.scale", {cout});
    r.n2_b = add_param(m, pre + ".norm_2.bias", {cout});
    r.pk_b1 = add_packed(m, conv_packed_bytes(m->mode, 9, cin, cout));
    r.pk_b2 = add_packed(m, conv_packed_bytes(m->mode, 9, cout, cout));
    if (r.has_res) r.pk_rc = add_packed(m, conv_packed_bytes(m->mode, 1, cin, cout));
    r.pt_b1 = add_packed_t(m, conv_packed_bytes(m->mode, 9, cout, cin));
    r.pt_b2 = add_packed_t(m, conv_packed_bytes(m->mode, 9, cout, cout));
    if (r.has_res) r.pt_rc = add_packed_t(m, conv_packed_bytes(m->mode, 1, cout, cin));
    if (has_mlp) {
        r.ss_index = (int)m->ss_layers.size();
        SsLayer L;
        memset(&L, 0, sizeof(L));
        L.w_off = r.mlp_w; L.b_off = r.mlp_b; L.g_off = r.n1_s; L.be_off = r.n1_b; L.n = 2 * cout;
        L.out_off = m->ss_floats_per_sample;            // scaled by the batch at launch time (see forward)
        m->ss_floats_per_sample += 2L * cout;
        m->ss_layers.push_back(L);
    } else r.ss_index = -1;
    r.name = pre;
    return r;
}

static AttnP make_attn(Model* m, const std::string& pre, int C) {
    AttnP a = AttnP();
    const int H = m->cfg.attn_heads, D = m->cfg.attn_dim_head;
    a.C = C;
    a.norm_s = add_param(m, pre + ".fn.norm.scale", {C});
    a.norm_b = add_param(m, pre + ".fn.norm.bias", {C});
    const char* nm[3] = {"q", "k", "v"};
    for (int i = 0; i < 3; ++i) {
        a.w[i] = add_param(m, pre + ".fn.fn.fn." + nm[i] + ".kernel", {C, H, D});
        a.b[i] = add_param(m, pre + ".fn.fn.fn." + nm[i] + ".bias", {H, D});
    }
    a.o_w = add_param(m, pre + ".fn.fn.fn.out.kernel", {H, D, C});
    a.o_b = add_param(m, pre + ".fn.fn.fn.out.bias", {C});
    a.pk_qkv = add_packed(m, 3 * conv_packed_bytes(m->mode, 1, C, H * D));
    a.pk_bqkv = add_packed(m, (size_t)3 * H * D * 4);
    a.pk_o = add_packed(m, conv_packed_bytes(m->mode, 1, H * D, C));
    for (int i = 0; i < 3; ++i) a.pt_w[i] = add_packed_t(m, conv_packed_bytes(m->mode, 1, H * D, C));      // [C rows][HD]
    a.pt_qkv = add_packed_t(m, conv_packed_bytes(m->mode, 1, 3 * H * D, C));                                // [C rows][q|k|v]
    a.pt_o = add_packed_t(m, conv_packed_bytes(m->mode, 1, C, H * D));                                       // [HD rows][C]
    a.name = pre;
    return a;
}

static SlaP make_sla(Model* m, const std::string& pre, int C) {
    SlaP s = SlaP();
    const int HD = m->cfg.attn_heads * 32;
    s.C = C;
    s.norm_s = add_param(m, pre + ".fn.norm.scale", {C});
    s.norm_b = add_param(m, pre + ".fn.norm.bias", {C});
    const char* nm[3] = {"q", "k", "v"};
    for (int i = 0; i < 3; ++i) {
        s.w[i] = add_param(m, pre + ".fn.fn." + nm[i] + ".kernel", {1, C, HD});
        s.pk[i] = add_packed(m, conv_packed_bytes(m->mode, 1, C, HD));
    }
    s.o_w = add_param(m, pre + ".fn.fn.to_out.kernel", {1, HD, C});
    s.pk_o = add_packed(m, conv_packed_bytes(m->mode, 1, HD, C));
    for (int i = 0; i < 3; ++i) s.pt_w[i] = add_packed_t(m, conv_packed_bytes(m->mode, 1, HD, C));
    s.pt_qkv = add_packed_t(m, conv_packed_bytes(m->mode, 1, 3 * HD, C));
    s.pt_o = add_packed_t(m, conv_packed_bytes(m->mode, 1, C, HD));
    s.name = pre;
    return s;
}

static int add_slot(Model* m, const std::string& name, long floats_per_sample) {
    Slot s;
    s.name = name;
    s.floats_per_sample = floats_per_sample;
    s.offset_per_sample = m->act_floats_per_sample;
    m->act_floats_per_sample += (floats_per_sample + 63) / 64 * 64;
    m->slots.push_back(s);
    return (int)m->slots.size() - 1;
}

static void res_slots(Model* m, ResP& r, long pix) {
    r.s_y1 = add_slot(m, r.name + "#y1", pix * r.cout);
    r.s_y2 = add_slot(m, r.name + "#y2", pix * r.cout);
    r.s_rc = r.has_res ? add_slot(m, r.name + "#rc", pix * r.cout) : -1;
    r.s_out = add_slot(m, r.name, pix * r.cout);
    r.st1 = m->n_stats++;
    r.st2 = m->n_stats++;
}

int model_build(Model* m) {
    const vdx_config& c = m->cfg;
    if (c.dim < 8 || c.dim % 8 || c.n_mults < 1 || c.n_mults > 8) return vdx_set_error(VDX_ERR_INVALID, "config: dim must be a positive multiple of 8, 1..8 dim_mults", __FILE__, __LINE__);
    if (c.attn_dim_head != 32) return vdx_set_error(VDX_ERR_INVALID, "config: attn_dim_head must be 32", __FILE__, __LINE__);
    if (c.use_sparse_linear_attn && c.attn_heads != 8) return vdx_set_error(VDX_ERR_INVALID, "config: SpatialLinearAttention needs attn_heads == 8", __FILE__, __LINE__);
    if (c.attn_heads < 1 || c.attn_heads > 32) return vdx_set_error(VDX_ERR_INVALID, "config: attn_heads must be in 1..32", __FILE__, __LINE__);      // (odd counts: forward only, see vdx_unet_backward)
    if (c.resnet_groups < 1 || c.resnet_groups > 32) return vdx_set_error(VDX_ERR_INVALID, "config: resnet_groups must be in 1..32", __FILE__, __LINE__);
    if (!(c.init_kernel_size & 1)) return vdx_set_error(VDX_ERR_INVALID, "config: init_kernel_size must be odd", __FILE__, __LINE__);
    const int down = 1 << (c.n_mults - 1);
    if (c.image_size % down) return vdx_set_error(VDX_ERR_INVALID, "config: image_size must be divisible by 2^(levels-1)", __FILE__, __LINE__);
    m->mode = c.mode;
    m->init_dim = c.init_dim > 0 ? c.init_dim : c.dim;
    m->out_dim = c.out_dim > 0 ? c.out_dim : c.channels;
    m->time_dim = 4 * c.dim;
    m->temb_dim = m->time_dim + c.cond_dim;
    std::vector<int> dims;
    dims.push_back(m->init_dim);
    for (int i = 0; i < c.n_mults; ++i) dims.push_back(c.dim * c.dim_mults[i]);
    for (int d : dims) if (d % c.resnet_groups || d % 8 || d > 1024) return vdx_set_error(VDX_ERR_INVALID, "config: every level width must be a multiple of 8 and of resnet_groups, <= 1024", __FILE__, __LINE__);
    const int nl = c.n_mults;
    const int H = c.attn_heads;

    // ---- parameters, in the reference's construction order (oracle/unet3d_ref.py::param_spec) ----
    m->rel_pos_emb = add_param(m, "time_rel_pos_bias.relative_attention_bias.embedding", {32, H});
    m->init_w = add_param(m, "init_conv.kernel", {1, c.init_kernel_size, c.init_kernel_size, c.channels, m->init_dim});
    m->init_b = add_param(m, "init_conv.bias", {m->init_dim});
    m->init_attn = make_attn(m, "init_temporal_attn", m->init_dim);
    m->t_w1 = add_param(m, "time_mlp.layers.1.kernel", {c.dim, m->time_dim});
    m->t_b1 = add_param(m, "time_mlp.layers.1.bias", {m->time_dim});
    m->t_w2 = add_param(m, "time_mlp.layers.3.kernel", {m->time_dim, m->time_dim});
    m->t_b2 = add_param(m, "time_mlp.layers.3.bias", {m->time_dim});
    m->null_cond = c.cond_dim ? add_param(m, "null_cond_emb", {1, c.cond_dim}) : -1;
    m->downs.resize(nl); m->ups.resize(nl);
    for (int i = 0; i < nl; ++i) {
        Level& L = m->downs[i];
        const std::string pre = "downs." + std::to_string(i);
        L.cin = dims[i]; L.cout = dims[i + 1];
        L.res0 = make_res(m, pre + ".0", dims[i], dims[i + 1], true);
        L.res1 = make_res(m, pre + ".1", dims[i + 1], dims[i + 1], true);
        L.has_sla = c.use_sparse_linear_attn != 0;
        if (L.has_sla) L.sla = make_sla(m, pre + ".2", dims[i + 1]);
        L.attn = make_attn(m, pre + ".3", dims[i + 1]);
        L.has_resample = i < nl - 1;
        if (L.has_resample) {
            L.rs_w = add_param(m, pre + ".4.kernel", {1, 4, 4, dims[i + 1], dims[i + 1]});
            L.rs_b = add_param(m, pre + ".4.bias", {dims[i + 1]});
            L.pk_rs = add_packed(m, conv_packed_bytes(m->mode, 16, dims[i + 1], dims[i + 1]));
            L.pt_rs = add_packed_t(m, conv_packed_bytes(m->mode, 16, dims[i + 1], dims[i + 1]));
        }
    }
    const int mid = dims[nl];
    m->mid1 = make_res(m, "mid_block1", mid, mid, true);
    m->mid_sattn = make_attn(m, "mid_spatial_attn", mid);
    m->mid_tattn = make_attn(m, "mid_temporal_attn", mid);
    m->mid2 = make_res(m, "mid_block2", mid, mid, true);
    for (int i = 0; i < nl; ++i) {
        Level& L = m->ups[i];
        const std::string pre = "ups." + std::to_string(i);
        const int din = dims[nl - 1 - i], dout = dims[nl - i];   // reversed(in_out)[i]
        L.cin = 2 * dout; L.cout = din;
        L.res0 = make_res(m, pre + ".0", 2 * dout, din, true);
        L.res1 = make_res(m, pre + ".1", din, din, true);
        L.has_sla = c.use_sparse_linear_attn != 0;
        if (L.has_sla) L.sla = make_sla(m, pre + ".2", din);
        L.attn = make_attn(m, pre + ".3", din);
        L.has_resample = i < nl - 1;
        if (L.has_resample) {
            L.rs_w = add_param(m, pre + ".4.kernel", {1, 4, 4, din, din});
            L.rs_b = add_param(m, pre + ".4.bias", {din});
            L.pk_rs = add_packed(m, conv_packed_bytes(m->mode, 16, din, din));
            L.pt_rs = add_packed_t(m, conv_packed_bytes(m->mode, 16, din, din));
        }
    }
    m->fin = make_res(m, "final_conv.layers.0", 2 * c.dim, c.dim, false);
    m->fin_w = add_param(m, "final_conv.layers.1.kernel", {1, c.dim, m->out_dim});
    m->fin_b = add_param(m, "final_conv.layers.1.bias", {m->out_dim});
    if (m->init_dim != c.dim) return vdx_set_error(VDX_ERR_INVALID, "config: init_dim must equal dim (the final concat feeds ResnetBlock(2*dim))", __FILE__, __LINE__);

    // ---- workspace plan: one slot per intermediate (nothing is recycled: 288 GB of HBM) ----
    const long F = c.num_frames;
    auto pixels = [&](int lvl) { const long s = c.image_size >> lvl; return F * s * s; };
    m->s_init = add_slot(m, "init_conv", pixels(0) * m->init_dim);
    m->s_init_attn = add_slot(m, "init_temporal_attn", pixels(0) * m->init_dim);
    for (int i = 0; i < nl; ++i) {
        Level& L = m->downs[i];
        L.lvl = i;
        res_slots(m, L.res0, pixels(i));
        res_slots(m, L.res1, pixels(i));
        L.s_sla = L.has_sla ? add_slot(m, "downs." + std::to_string(i) + ".2", pixels(i) * L.cout) : -1;
        L.s_attn = add_slot(m, "downs." + std::to_string(i) + ".3", pixels(i) * L.cout);
        L.s_rs = L.has_resample ? add_slot(m, "downs." + std::to_string(i) + ".4", pixels(i + 1) * L.cout) : -1;
    }
    res_slots(m, m->mid1, pixels(nl - 1));
    m->s_mid_sattn = add_slot(m, "mid_spatial_attn", pixels(nl - 1) * mid);
    m->s_mid_tattn = add_slot(m, "mid_temporal_attn", pixels(nl - 1) * mid);
    res_slots(m, m->mid2, pixels(nl - 1));
    for (int i = 0; i < nl; ++i) {
        Level& L = m->ups[i];
        L.lvl = nl - 1 - i;
        res_slots(m, L.res0, pixels(L.lvl));
        res_slots(m, L.res1, pixels(L.lvl));
        L.s_sla = L.has_sla ? add_slot(m, "ups." + std::to_string(i) + ".2", pixels(L.lvl) * L.cout) : -1;
        L.s_attn = add_slot(m, "ups." + std::to_string(i) + ".3", pixels(L.lvl) * L.cout);
        L.s_rs = L.has_resample ? add_slot(m, "ups." + std::to_string(i) + ".4", pixels(L.lvl - 1) * L.cout) : -1;
    }
    res_slots(m, m->fin, pixels(0));
    // SLA scratch: sized for the largest level
    m->sla_ws_bytes_per_sample = 0;
    for (int l = 0; l < nl; ++l) {
        const long s = c.image_size >> l;
        if (c.use_sparse_linear_attn)
            m->sla_ws_bytes_per_sample = std::max(m->sla_ws_bytes_per_sample, sla_workspace_bytes(m->mode, (int)F, (int)(s * s), H));
        // the same scratch holds the per-head attention output [F * s * s rows][heads * 32] bf16 of launch_attention_heads (wide levels)
        if (c.dim * c.dim_mults[l] >= 256)
            m->sla_ws_bytes_per_sample = std::max(m->sla_ws_bytes_per_sample, (size_t)F * s * s * c.attn_heads * 32 * 2);
    }
    {   // bottleneck spatial attention over more than 64 tokens (frames larger than 64 x 64): materialised qkv + o, fp32
        const long s = c.image_size >> (nl - 1);
        if (s * s > 64) m->sla_ws_bytes_per_sample = std::max(m->sla_ws_bytes_per_sample, (size_t)F * s * s * c.attn_heads * 32 * 4 * 4);
    }
    return VDX_OK;
}

size_t model_workspace_bytes(const Model* m, int B) {
    size_t b = 0;
    b += ((size_t)m->act_floats_per_sample * B * 4 + 255) / 256 * 256;
    b += ((size_t)m->temb_dim * B * 4 + 255) / 256 * 256;
    b += 2 * (((size_t)m->ss_floats_per_sample * B * 4 + 255) / 256 * 256);       // scale/shift + pre-LayerNorm values
    b += ((size_t)m->n_stats * B * GN_SLOTS * m->cfg.resnet_groups * 2 * 8 + 255) / 256 * 256;
    b += (m->sla_ws_bytes_per_sample * B + 255) / 256 * 256;
    return b;
}

// ---- packing ------------------------------------------------------------------------------------------------

void model_build_pack_tables(Model* m) {
    const int HD = m->cfg.attn_heads * 32;
    auto pad = [&](int k) { return conv_cin_pad(m->mode, k); };
    auto fwd = [&](long src, size_t dst, int taps, int cin, int cout) {
        PackJob j{}; j.src = src; j.dst = (long)dst; j.taps = taps; j.Cin = cin; j.Cout = cout; j.Pad = pad(cin); j.kind = 0;
        j.n = (long)taps * cout * j.Pad; m->pack_jobs.push_back(j);
    };
    auto cpy = [&](long src, size_t dst, long n) { PackJob j{}; j.src = src; j.dst = (long)dst; j.n = n; j.kind = 2; j.Pad = 1; m->pack_jobs.push_back(j); };
    auto bwd = [&](long src, size_t dst, int taps, int cin, int cout) {
        PackJob j{}; j.src = src; j.dst = (long)dst; j.taps = taps; j.Cin = cin; j.Cout = cout; j.Pad = pad(cout); j.kind = 1;
        j.n = (long)taps * cin * j.Pad; m->pack_t_jobs.push_back(j);
    };
    auto cat3 = [&](const long (&w)[3], size_t dst, int c) {      // [c rows][3 * HD], K = (tensor, head channel)
        PackJob j{}; j.src = w[0]; j.src1 = w[1]; j.src2 = w[2]; j.dst = (long)dst; j.taps = 1; j.Cin = c; j.Cout = HD; j.Pad = pad(3 * HD); j.kind = 3;
        j.n = (long)c * j.Pad; m->pack_t_jobs.push_back(j);
    };
    auto res = [&](const ResP& r) {
        fwd(r.b1_w, r.pk_b1, 9, r.cin, r.cout); fwd(r.b2_w, r.pk_b2, 9, r.cout, r.cout);
        bwd(r.b1_w, r.pt_b1, 9, r.cin, r.cout); bwd(r.b2_w, r.pt_b2, 9, r.cout, r.cout);
        if (r.has_res) { fwd(r.rc_w, r.pk_rc, 1, r.cin, r.cout); bwd(r.rc_w, r.pt_rc, 1, r.cin, r.cout); }
    };
    auto attn = [&](const AttnP& a) {
        const size_t one = conv_packed_bytes(m->mode, 1, a.C, HD);
        for (int i = 0; i < 3; ++i) {
            fwd(a.w[i], a.pk_qkv + i * one, 1, a.C, HD);
            cpy(a.b[i], a.pk_bqkv + (size_t)i * HD * 4, HD);
            bwd(a.w[i], a.pt_w[i], 1, a.C, HD);
        }
        fwd(a.o_w, a.pk_o, 1, HD, a.C); bwd(a.o_w, a.pt_o, 1, HD, a.C);
        cat3(a.w, a.pt_qkv, a.C);
    };
    auto sla = [&](const SlaP& s) {
        for (int i = 0; i < 3; ++i) { fwd(s.w[i], s.pk[i], 1, s.C, HD); bwd(s.w[i], s.pt_w[i], 1, s.C, HD); }
        fwd(s.o_w, s.pk_o, 1, HD, s.C); bwd(s.o_w, s.pt_o, 1, HD, s.C);
        cat3(s.w, s.pt_qkv, s.C);
    };
    m->pack_jobs.clear(); m->pack_t_jobs.clear();
    attn(m->init_attn);
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<Level>& lv = pass ? m->ups : m->downs;
        for (const Level& L : lv) {
            res(L.res0); res(L.res1);
            if (L.has_sla) sla(L.sla);
            attn(L.attn);
            if (L.has_resample) { fwd(L.rs_w, L.pk_rs, 16, L.cout, L.cout); bwd(L.rs_w, L.pt_rs, 16, L.cout, L.cout); }
        }
    }
    res(m->mid1); attn(m->mid_sattn); attn(m->mid_tattn); res(m->mid2); res(m->fin);
}

hipError_t model_pack(const Model* m, const float* p, void* packed, hipStream_t st) {
    if (!m->d_pack_jobs) return hipErrorInvalidValue;
    return launch_pack_jobs(m->mode, p, packed, m->d_pack_jobs, (int)m->pack_jobs.size(), st);
}

// ---- forward -------------------------------------------------------------------------------------------------

struct Fwd {
    const Model* m; const float* p; const char* pk; int B; hipStream_t st;
    int a16;                                                  // every inter-kernel activation tensor is stored as bf16 (Model::act16)
    float* act; float* temb; float* ss; float* ss_lin; double* stats; char* sla_ws;
    float* slot(int s) const { return act + (size_t)m->slots[s].offset_per_sample * B; }
    double* stat(int i) const { return stats + (size_t)i * B * GN_SLOTS * m->cfg.resnet_groups * 2; }
};

#ifndef VDX_FUSE_HEAD
#define VDX_FUSE_HEAD 1                                      // final 1x1 conv (one output channel) inside the last block's tail (inference storage)
#endif
#ifndef VDX_PREPASS_MIN_C
#define VDX_PREPASS_MIN_C 256                                 // Block prologue as its own pass from this width on (sampling forward)
#endif

// head_out: (the network's last block) write head_out[pix] = out[pix][:] . fin_w + fin_b instead of the block's output when the tail form
// that can do it runs (*head_done = true), else the caller launches the final conv as before
static hipError_t run_res(const Fwd& f, const ResP& r, const float* x0, int c0, const float* x1, int c1, int lvl, float* head_out = nullptr, bool* head_done = nullptr) {
    const Model* m = f.m;
    const int G = m->cfg.resnet_groups;
    const int S = m->cfg.image_size >> lvl;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = x0; a.x1 = x1; a.C0 = c0; a.C1 = c1;
    a.wp = f.pk + r.pk_b1; a.bias = f.p + r.b1_b; a.y = f.slot(r.s_y1); a.Cout = r.cout;
    a.NF = f.B * m->cfg.num_frames; a.F = m->cfg.num_frames; a.H = S; a.W = S;
    a.kind = 0; a.kh = a.kw = 3; a.stride = 1; a.pad = 1;
    a.out_stats = f.stat(r.st1); a.out_groups = G;
    const int half = (m->mode == MODE_BF16);                  // bf16 mode stores the two intra-block tensors y1, y2 as bf16
    a.y_bf16 = half; a.x0_bf16 = f.a16; a.x1_bf16 = f.a16;
    hipError_t e = launch_conv(m->mode, a, f.st);
    if (e != hipSuccess) return e;
    ConvArgs b;
    memset(&b, 0, sizeof(b));
    b.x0 = f.slot(r.s_y1); b.C0 = r.cout;
    b.wp = f.pk + r.pk_b2; b.bias = f.p + r.b2_b; b.y = f.slot(r.s_y2); b.Cout = r.cout;
    b.NF = a.NF; b.F = a.F; b.H = S; b.W = S; b.kind = 0; b.kh = b.kw = 3; b.stride = 1; b.pad = 1;
    b.x0_bf16 = half; b.y_bf16 = half;
    b.pro = 1; b.in_stats = f.stat(r.st1); b.gamma = f.p + r.b1_gs; b.beta = f.p + r.b1_gb; b.groups = G;
    if (r.has_mlp) { b.ss = f.ss + (size_t)m->ss_layers[r.ss_index].out_off * f.B; b.ss_stride = 2 * r.cout; }
    b.out_stats = f.stat(r.st2); b.out_groups = G;
    // sampling forward, wide blocks: the prologue as one in-place pass over y1 (nothing reads y1 after conv2 in inference storage), and the plain
    // form of the weight-stationary conv -- its fused-prologue form repeats the GroupNorm / SiLU arithmetic in each of its Cout / 128 workgroups
    if (half && m->act16 == 1 && r.cout >= VDX_PREPASS_MIN_C) {
        e = launch_gn_silu_apply16(f.slot(r.s_y1), b.in_stats, b.gamma, b.beta, b.ss, b.ss_stride, G, r.cout, f.B, (long)m->cfg.num_frames * S * S, f.st);
        if (e != hipSuccess) return e;
        b.pro = 0; b.in_stats = nullptr; b.gamma = b.beta = nullptr; b.ss = nullptr; b.groups = 0;
    }
    e = launch_conv(m->mode, b, f.st);
    if (e != hipSuccess) return e;
    const float* rsrc = x0;
    // bf16 activation storage: the 1x1 res_conv runs inside the tail (resblock_tail_rc16_kernel) for the shapes it is built for
    // (act16 == 2, the training forward: r = res_conv(x) must exist in its slot, the backward's LayerNorm branch reads it)
    const bool fuse_rc = r.has_res && f.a16 && m->act16 == 1 && m->mode == MODE_BF16 &&
                         tail_rc16_supported(c0 + c1, c0, r.cout, (long)m->cfg.num_frames * S * S);
    if (r.has_res && !fuse_rc) {
        ConvArgs c;
        memset(&c, 0, sizeof(c));
        c.x0 = x0; c.x1 = x1; c.C0 = c0; c.C1 = c1;
        c.wp = f.pk + r.pk_rc; c.bias = f.p + r.rc_b; c.y = f.slot(r.s_rc); c.Cout = r.cout;
        c.NF = a.NF; c.F = a.F; c.H = S; c.W = S; c.kind = 0; c.kh = c.kw = 1; c.stride = 1; c.pad = 0;
        c.x0_bf16 = f.a16; c.x1_bf16 = f.a16; c.y_bf16 = f.a16;
        e = launch_conv(m->mode, c, f.st);
        if (e != hipSuccess) return e;
        rsrc = f.slot(r.s_rc);
    }
    TailArgs t;
    memset(&t, 0, sizeof(t));
    t.y2 = f.slot(r.s_y2); t.y2_bf16 = half; t.r = rsrc; t.out = f.slot(r.s_out); t.r_bf16 = f.a16; t.out_bf16 = f.a16;
    t.stats = f.stat(r.st2); t.gn_gamma = f.p + r.b2_gs; t.gn_beta = f.p + r.b2_gb; t.groups = G;
    t.ln_gamma = f.p + r.n2_s; t.ln_beta = f.p + r.n2_b;
    t.C = r.cout; t.batch = f.B; t.pix_per_sample = (long)m->cfg.num_frames * S * S;
    if (fuse_rc) { t.x0 = x0; t.x1 = x1; t.C0 = c0; t.C1 = c1; t.rc_w = f.pk + r.pk_rc; t.rc_b = f.p + r.rc_b; }
    if (VDX_FUSE_HEAD && head_out && fuse_rc && c1 == c0 && m->out_dim == 1 && ((c0 + c1 == 128 && r.cout == 64) || (c0 + c1 == 64 && r.cout == 32))) {
        t.fin_w = f.p + m->fin_w; t.fin_b = f.p + m->fin_b; t.fin_out = head_out;
        *head_done = true;
    }
    return launch_resblock_tail(t, f.st);
}

static hipError_t run_attn(const Fwd& f, const AttnP& ap, const float* x, float* y, int lvl, bool temporal) {
    const Model* m = f.m;
    const long S = m->cfg.image_size >> lvl, hw = S * S, Fr = m->cfg.num_frames;
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.wqkv = f.pk + ap.pk_qkv; a.bqkv = reinterpret_cast<const float*>(f.pk + ap.pk_bqkv);
    a.wo = f.pk + ap.pk_o; a.bo = f.p + ap.o_b; a.C = ap.C; a.heads = m->cfg.attn_heads;
    a.scale = 1.0f / sqrtf((float)m->cfg.attn_dim_head);
    if (temporal) { a.L = (int)Fr; a.nseq = f.B * hw; a.inner = hw; a.inner_stride = ap.C; a.outer_stride = Fr * hw * ap.C; a.tok_stride = hw * ap.C; }
    else { a.L = (int)hw; a.nseq = f.B * Fr; a.inner = 1; a.inner_stride = 0; a.outer_stride = hw * ap.C; a.tok_stride = ap.C; }
    a.io_bf16 = f.a16;
    a.fp8_core = (m->attn_fp8 && m->mode == MODE_BF16) ? 1 : 0;
    if (a.L > 64) {
        // long sequences (spatial attention of a bottleneck larger than 8 x 8): q|k|v projection and out-projection (+ bias + residual)
        // as 1x1 convs around the fp32 core of attention.hip; token-major rows = the channel-last tensor as it is
        if (temporal) return hipErrorInvalidValue;                       // (a temporal axis of more than 64 frames is not served)
        const int HD = a.heads * 32;
        float* qkv = reinterpret_cast<float*>(f.sla_ws);
        float* o = qkv + (size_t)f.B * Fr * hw * 3 * HD;
        ConvArgs p;
        memset(&p, 0, sizeof(p));
        p.x0 = x; p.C0 = ap.C; p.x0_bf16 = f.a16; p.wp = f.pk + ap.pk_qkv; p.bias = reinterpret_cast<const float*>(f.pk + ap.pk_bqkv);
        p.y = qkv; p.Cout = 3 * HD;
        p.NF = f.B * (int)Fr; p.F = (int)Fr; p.H = (int)S; p.W = (int)S; p.kind = 0; p.kh = p.kw = 1; p.stride = 1; p.pad = 0;
        hipError_t e = launch_conv(m->mode, p, f.st);
        if (e != hipSuccess) return e;
        e = launch_attention_long_core(qkv, o, a.nseq, a.L, a.heads, a.scale, f.st);
        if (e != hipSuccess) return e;
        ConvArgs c;
        memset(&c, 0, sizeof(c));
        c.x0 = o; c.C0 = HD; c.wp = f.pk + ap.pk_o; c.bias = f.p + ap.o_b; c.y = y; c.Cout = ap.C; c.y_bf16 = f.a16;
        c.res = x; c.res_bf16 = f.a16;
        c.NF = f.B * (int)Fr; c.F = (int)Fr; c.H = (int)S; c.W = (int)S; c.kind = 0; c.kh = c.kw = 1; c.stride = 1; c.pad = 0;
        return launch_conv(m->mode, c, f.st);
    }
    // wide levels in bf16 mode: per-head kernel (weights resident in LDS) + the out-projection as a 1x1 conv
    const int use_heads = 1;
    if (use_heads && m->mode == MODE_BF16 && (temporal ? a.L <= 16 : a.L <= 64) && a.heads == 8 && ap.C >= 256 && ap.C % 128 == 0 &&
        (size_t)96 * (ap.C * 2 + 32) <= 160 * 1024 && (size_t)Fr * S * S * a.heads * 64 <= m->sla_ws_bytes_per_sample) {
        a.oscratch = f.sla_ws;
        hipError_t e = launch_attention_heads(a, f.st);
        if (e != hipSuccess) return e;
        ConvArgs c;
        memset(&c, 0, sizeof(c));
        c.x0 = reinterpret_cast<const float*>(f.sla_ws); c.C0 = a.heads * 32; c.x0_bf16 = 1;
        c.wp = f.pk + ap.pk_o; c.bias = f.p + ap.o_b; c.y = y; c.Cout = ap.C; c.y_bf16 = f.a16;
        c.res = x; c.res_bf16 = f.a16;
        c.NF = f.B * (int)Fr; c.F = (int)Fr; c.H = (int)S; c.W = (int)S; c.kind = 0; c.kh = c.kw = 1; c.stride = 1; c.pad = 0;
        return launch_conv(m->mode, c, f.st);
    }
    return launch_attention(m->mode, a, f.st);
}

static hipError_t run_sla(const Fwd& f, const SlaP& sp, const float* x, float* y, int lvl) {
    const Model* m = f.m;
    const int S = m->cfg.image_size >> lvl;
    SlaArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.wq = f.pk + sp.pk[0]; a.wk = f.pk + sp.pk[1]; a.wv = f.pk + sp.pk[2]; a.wo = f.pk + sp.pk_o;
    a.workspace = f.sla_ws; a.C = sp.C; a.heads = m->cfg.attn_heads; a.NF = f.B * m->cfg.num_frames; a.N = S * S;
    a.io_bf16 = f.a16;
    // wide levels in bf16 mode: per-head kernel (weights resident in LDS) + to_out as a 1x1 conv
    const int use_heads = 1;
    if (use_heads && m->mode == MODE_BF16 && a.heads == 8 && sp.C >= 256 && sp.C % 128 == 0 && a.N % 16 == 0 &&
        (size_t)96 * (sp.C * 2 + 32) <= 160 * 1024 && (size_t)m->cfg.num_frames * a.N * a.heads * 64 <= m->sla_ws_bytes_per_sample) {
        hipError_t e = launch_sla_heads(a, f.sla_ws, f.st);
        if (e != hipSuccess) return e;
        ConvArgs c;
        memset(&c, 0, sizeof(c));
        c.x0 = reinterpret_cast<const float*>(f.sla_ws); c.C0 = a.heads * 32; c.x0_bf16 = 1;
        c.wp = f.pk + sp.pk_o; c.y = y; c.Cout = sp.C; c.y_bf16 = f.a16;
        c.res = x; c.res_bf16 = f.a16;
        c.NF = a.NF; c.F = m->cfg.num_frames; c.H = S; c.W = S; c.kind = 0; c.kh = c.kw = 1; c.stride = 1; c.pad = 0;
        return launch_conv(m->mode, c, f.st);
    }
    return launch_sla(m->mode, a, f.st);
}

static hipError_t run_resample(const Fwd& f, const Level& L, const float* x, float* y, int lvl_in, bool up) {
    const Model* m = f.m;
    const int S = m->cfg.image_size >> lvl_in;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x0 = x; a.C0 = L.cout; a.wp = f.pk + L.pk_rs; a.bias = f.p + L.rs_b; a.y = y; a.Cout = L.cout;
    a.NF = f.B * m->cfg.num_frames; a.F = m->cfg.num_frames; a.H = S; a.W = S;
    if (up) { a.kind = 1; a.kh = a.kw = 4; a.stride = 1; a.pad = 0; }
    else { a.kind = 0; a.kh = a.kw = 4; a.stride = 2; a.pad = 1; }
    a.x0_bf16 = f.a16; a.y_bf16 = f.a16;
    return launch_conv(m->mode, a, f.st);
}

int model_forward(const Model* m, const float* params, const void* packed, const float* x, const int* time,
                  const float* cond, const unsigned char* cond_mask, int null_all, float* out, void* workspace,
                  size_t workspace_bytes, int B, hipStream_t st) {
    if (B < 1) return vdx_set_error(VDX_ERR_INVALID, "forward: batch < 1", __FILE__, __LINE__);
    if (workspace_bytes < model_workspace_bytes(m, B)) return vdx_set_error(VDX_ERR_NOMEM, "forward: workspace too small", __FILE__, __LINE__);
    if (m->cfg.cond_dim && !cond) return vdx_set_error(VDX_ERR_INVALID, "cond must be passed in if cond_dim specified", __FILE__, __LINE__);
    const vdx_config& c = m->cfg;
    Fwd f;
    f.m = m; f.p = params; f.pk = reinterpret_cast<const char*>(packed); f.B = B; f.st = st;
    f.a16 = (m->act16 && m->mode == MODE_BF16) ? 1 : 0;
    char* w = reinterpret_cast<char*>(workspace);
    f.act = reinterpret_cast<float*>(w); w += ((size_t)m->act_floats_per_sample * B * 4 + 255) / 256 * 256;
    f.temb = reinterpret_cast<float*>(w); w += ((size_t)m->temb_dim * B * 4 + 255) / 256 * 256;
    f.ss = reinterpret_cast<float*>(w); w += ((size_t)m->ss_floats_per_sample * B * 4 + 255) / 256 * 256;
    f.ss_lin = reinterpret_cast<float*>(w); w += ((size_t)m->ss_floats_per_sample * B * 4 + 255) / 256 * 256;
    f.stats = reinterpret_cast<double*>(w);
    const size_t stats_bytes = (size_t)m->n_stats * B * GN_SLOTS * c.resnet_groups * 2 * 8;
    w += (stats_bytes + 255) / 256 * 256;
    f.sla_ws = w;
    hipError_t e;
#define VDX_E(x) do { e = (x); if (e != hipSuccess) return vdx_set_error(VDX_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__); } while (0)
    {
        LaunchScope ls(st, "fillBufferAligned", 0.0, (double)stats_bytes, "GroupNorm statistics slabs %zu B", stats_bytes);      // (hipMemsetAsync's kernel)
        VDX_E(hipMemsetAsync(f.stats, 0, stats_bytes, st));
    }
    const int S0 = c.image_size, Fr = c.num_frames;
    // time embedding + every ResnetBlock's (scale, shift)   (unet3d.py:288-298, modules.py:233-238)
    {
        TimeMlpArgs t;
        memset(&t, 0, sizeof(t));
        t.time = time; t.w1 = params + m->t_w1; t.b1 = params + m->t_b1; t.w2 = params + m->t_w2; t.b2 = params + m->t_b2;
        t.dim = c.dim; t.time_dim = m->time_dim; t.cond = cond; t.null_cond_emb = c.cond_dim ? params + m->null_cond : nullptr;
        t.cond_mask = cond_mask; t.null_all = null_all; t.cond_dim = c.cond_dim; t.temb = f.temb; t.temb_dim = m->temb_dim;
        VDX_E(launch_time_mlp(t, B, st));
        int max_n = 0;
        for (const SsLayer& l : m->ss_layers) max_n = std::max(max_n, l.n);
        VDX_E(launch_resblock_ss(params, f.temb, m->d_ss_layers, (int)m->ss_layers.size(), f.ss, f.ss_lin, m->temb_dim, B, max_n, st));
    }
    // init conv + init temporal attention   (unet3d.py:280-286)
    VDX_E(launch_init_conv_mode(m->mode, x, params + m->init_w, params + m->init_b, f.slot(m->s_init), B, c.channels, Fr, S0, S0, m->init_dim, c.init_kernel_size, f.a16, st));
    VDX_E(run_attn(f, m->init_attn, f.slot(m->s_init), f.slot(m->s_init_attn), 0, true));
    const float* cur = f.slot(m->s_init_attn);
    int cur_c = m->init_dim;
    const int nl = c.n_mults;
    for (int i = 0; i < nl; ++i) {                           // unet3d.py:303-314
        const Level& L = m->downs[i];
        VDX_E(run_res(f, L.res0, cur, cur_c, nullptr, 0, i));
        VDX_E(run_res(f, L.res1, f.slot(L.res0.s_out), L.cout, nullptr, 0, i));
        cur = f.slot(L.res1.s_out);
        if (L.has_sla) { VDX_E(run_sla(f, L.sla, cur, f.slot(L.s_sla), i)); cur = f.slot(L.s_sla); }
        VDX_E(run_attn(f, L.attn, cur, f.slot(L.s_attn), i, true));
        cur = f.slot(L.s_attn); cur_c = L.cout;
        if (L.has_resample) { VDX_E(run_resample(f, L, cur, f.slot(L.s_rs), i, false)); cur = f.slot(L.s_rs); }
    }
    VDX_E(run_res(f, m->mid1, cur, cur_c, nullptr, 0, nl - 1));                                   // unet3d.py:320-334
    VDX_E(run_attn(f, m->mid_sattn, f.slot(m->mid1.s_out), f.slot(m->s_mid_sattn), nl - 1, false));
    VDX_E(run_attn(f, m->mid_tattn, f.slot(m->s_mid_sattn), f.slot(m->s_mid_tattn), nl - 1, true));
    VDX_E(run_res(f, m->mid2, f.slot(m->s_mid_tattn), cur_c, nullptr, 0, nl - 1));
    cur = f.slot(m->mid2.s_out);
    for (int i = 0; i < nl; ++i) {                           // unet3d.py:337-370
        const Level& L = m->ups[i];
        const Level& D = m->downs[nl - 1 - i];               // skip = h.pop()
        VDX_E(run_res(f, L.res0, cur, cur_c, f.slot(D.s_attn), D.cout, L.lvl));
        VDX_E(run_res(f, L.res1, f.slot(L.res0.s_out), L.cout, nullptr, 0, L.lvl));
        cur = f.slot(L.res1.s_out);
        if (L.has_sla) { VDX_E(run_sla(f, L.sla, cur, f.slot(L.s_sla), L.lvl)); cur = f.slot(L.s_sla); }
        VDX_E(run_attn(f, L.attn, cur, f.slot(L.s_attn), L.lvl, true));
        cur = f.slot(L.s_attn); cur_c = L.cout;
        if (L.has_resample) { VDX_E(run_resample(f, L, cur, f.slot(L.s_rs), L.lvl, true)); cur = f.slot(L.s_rs); }
    }
    bool head_done = false;                                  // (sampling forward: the 1-channel head inside the last block's tail)
    VDX_E(run_res(f, m->fin, cur, cur_c, f.slot(m->s_init_attn), m->init_dim, 0, out, &head_done));                // unet3d.py:377-382
    if (!head_done) VDX_E(launch_final_conv(f.slot(m->fin.s_out), params + m->fin_w, params + m->fin_b, out, (long)B * Fr * S0 * S0, c.dim, m->out_dim, f.a16, st));
#undef VDX_E
    return VDX_OK;
}

}  // namespace vdx
