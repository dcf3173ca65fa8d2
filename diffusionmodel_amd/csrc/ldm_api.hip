// Host side of libldm_hip.so: parameter staging, BatchNorm folding and weight packing, the UNet
// forward as a sequence of kernel launches, the sampling loop with its hipGraph, and the C ABI of
// include/ldm_hip.h.  Reference: Unet / ResidualBlock / Attention / TimeAddition / DiffusionModel in
// src/main.rs:237-335, 336-527, 553-583, 631-633, 681-777.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <vector>

#include "common.h"

namespace ldm {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int round_up_i(int v, int a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// device allocations owned by a module / sampler
// ---------------------------------------------------------------------------------------------
struct DevPool {
    std::vector<void*> ptrs;
    ~DevPool() { release(); }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
    int alloc(void** out, size_t bytes) {
        void* p = nullptr;
        LDM_HIP(hipMalloc(&p, bytes ? bytes : 16));
        ptrs.push_back(p);
        *out = p;
        return LDM_OK;
    }
    int upload(void** out, const void* host, size_t bytes) {
        LDM_TRY(alloc(out, bytes));
        LDM_HIP(hipMemcpy(*out, host, bytes, hipMemcpyHostToDevice));
        return LDM_OK;
    }
};

// ---------------------------------------------------------------------------------------------
// schedule, f32, operation order of src/main.rs:553-583 and :712-747
// ---------------------------------------------------------------------------------------------
static void host_betas(int T, std::vector<float>& b) {
    b.resize(T);
    const float b0 = 1e-4f, b1 = 0.02f;
    const int dm = T - 1 > 1 ? T - 1 : 1;
    const float den = (float)dm;
    for (int i = 0; i < T; ++i) {
        const float ratio = (float)i / den;
        float beta = b0 + (b1 - b0) * ratio;
        if (beta < 1e-6f) beta = 1e-6f;
        if (beta > 0.999f) beta = 0.999f;
        b[i] = beta;
    }
}
static void host_alpha_bars(int T, std::vector<float>& ab) {
    std::vector<float> b;
    host_betas(T, b);
    ab.resize(T);
    float c = 1.0f;
    for (int i = 0; i < T; ++i) {
        const float a = 1.0f - b[i];
        c = c * a;
        ab[i] = c;
    }
}
static void host_ddpm_coef(int T, std::vector<float>& tab) {   // [T][5]
    std::vector<float> b, ab;
    host_betas(T, b);
    host_alpha_bars(T, ab);
    tab.assign((size_t)T * 5, 0.f);
    for (int i = 0; i < T; ++i) {
        const float a_t = ab[i], b_t = b[i];
        float* r = &tab[(size_t)i * 5];
        r[0] = sqrtf(a_t);
        const float om_t = a_t * -1.0f + 1.0f;
        r[1] = sqrtf(om_t);
        if (i > 0) {
            const float a_p = ab[i - 1];
            const float om_p = a_p * -1.0f + 1.0f;
            const float den = om_t + 1e-8f;
            const float var = (om_p / den) * b_t;
            r[4] = sqrtf(var);
            r[2] = (sqrtf(a_p) * b_t) / den;
            r[3] = (sqrtf(1.0f - b_t) * om_p) / den;
        } else {
            r[2] = 1.0f; r[3] = 0.0f; r[4] = 0.0f;
        }
    }
}
// add_noise's per-timestep scalars, src/main.rs:603-624: the gathered alpha-bar is replaced by 1e-6 when it is
// not a positive finite number and clamped to [1e-8, 0.9999] otherwise; sqrt(alpha) is clamped to [1e-8, 1],
// 1 - alpha to [1e-8, 1] before its square root.
static void host_add_noise_coef(int T, std::vector<float>& tab) {   // [T][2]
    std::vector<float> ab;
    host_alpha_bars(T, ab);
    tab.assign((size_t)T * 2, 0.f);
    for (int i = 0; i < T; ++i) {
        float a = ab[i];
        if (std::isnan(a) || std::isinf(a) || a <= 0.0f) a = 1e-6f;
        else a = std::fmin(std::fmax(a, 1e-8f), 0.9999f);
        const float sa = std::fmin(std::fmax(sqrtf(a), 1e-8f), 1.0f);
        const float om = std::fmin(std::fmax(1.0f - a, 1e-8f), 1.0f);
        tab[2 * i] = sa;
        tab[2 * i + 1] = sqrtf(om);
    }
}
static int host_ddim_ts(int T, int steps, std::vector<int>& ts) {
    LDM_REQUIRE(steps >= 1 && steps <= T, "ddim steps must be in [1, num_time]");
    const int stride = T / steps;
    ts.resize(steps);
    for (int k = 0; k < steps; ++k) ts[k] = (steps - 1 - k) * stride;
    return LDM_OK;
}
static int host_ddim_coef(int T, int steps, std::vector<float>& tab) {   // [steps][5]
    std::vector<float> ab;
    host_alpha_bars(T, ab);
    std::vector<int> ts;
    LDM_TRY(host_ddim_ts(T, steps, ts));
    const int stride = T / steps;
    tab.assign((size_t)steps * 5, 0.f);
    for (int k = 0; k < steps; ++k) {
        const int t = ts[k];
        const float a_t = ab[t];
        const float a_p = t - stride >= 0 ? ab[t - stride] : 1.0f;
        float* r = &tab[(size_t)k * 5];
        r[0] = sqrtf(a_t);
        r[1] = sqrtf(a_t * -1.0f + 1.0f);
        r[2] = sqrtf(a_p);
        r[3] = sqrtf(a_p * -1.0f + 1.0f);
    }
    return LDM_OK;
}

// ---------------------------------------------------------------------------------------------
// weight packing for the tap-convolution kernel
// ---------------------------------------------------------------------------------------------
struct PackSpec {
    int kind = 0, dtype = 0;
    int cout = 0;                 // real output channels
    std::vector<int> src;         // real channels of each source (1 or 2 sources)
    int src0_padto = 0;           // pad source 0 to this many channels (0: to the kernel's channel chunk)
    int ntaps1 = 0;               // taps of the second source's chunks (0: the same as the first source's)
    std::function<float(int phase, int tap, int co, int ci)> w;   // ci over the concatenated real channels
    std::function<float(int co)> bias;
};

static int pack_conv(const PackSpec& sp, DevPool& pool, PackedConv* pc, int* src_pad /*[2]*/) {
    const int es = sp.dtype == LDM_BF16 ? 2 : 4;
    const int ck = 128 / es, epp = 16 / es;
    LDM_REQUIRE(sp.src.size() == 1 || sp.src.size() == 2, "one or two sources");
    int pad[2] = {0, 0}, off[2] = {0, 0}, ch0[2] = {0, 0};
    int nchunks = 0, real_off = 0;
    for (size_t s = 0; s < sp.src.size(); ++s) {
        pad[s] = round_up_i(sp.src[s], ck);
        if (s == 0 && sp.src0_padto > 0) {
            LDM_REQUIRE(sp.src0_padto % ck == 0 && sp.src0_padto >= sp.src[0], "bad channel padding");
            pad[0] = sp.src0_padto;
        }
        off[s] = real_off;
        ch0[s] = nchunks;
        real_off += sp.src[s];
        nchunks += pad[s] / ck;
    }
    pc->kind = sp.kind;
    pc->dtype = sp.dtype;
    pc->cin = pad[0] + pad[1];
    pc->cout = round_up_i(sp.cout, 64);
    pc->nphase = sp.kind == CONVT4x4_S2 ? 4 : 1;
    pc->ntaps = (sp.kind == CONV3x3_S1 || sp.kind == CONV3x3_S2) ? 9 : (sp.kind == CONVT4x4_S2 ? 4 : 1);
    pc->ntaps1 = sp.ntaps1 > 0 ? sp.ntaps1 : pc->ntaps;
    const int nch_src0 = sp.src.size() == 2 ? ch0[1] : nchunks;
    const size_t slabs_per_phase = (size_t)nch_src0 * pc->ntaps + (size_t)(nchunks - nch_src0) * pc->ntaps1;
    const size_t nel = (size_t)pc->nphase * slabs_per_phase * pc->cout * ck;
    std::vector<float> wf(sp.dtype == LDM_F32 ? nel : 0);
    std::vector<bf16_t> wh(sp.dtype == LDM_BF16 ? nel : 0);
    for (int ph = 0; ph < pc->nphase; ++ph)
        for (int c = 0; c < nchunks; ++c) {
            const int s = (sp.src.size() == 2 && c >= ch0[1]) ? 1 : 0;
            const int nt = s == 0 ? pc->ntaps : pc->ntaps1;
            const size_t slab0 = c < nch_src0 ? (size_t)c * pc->ntaps : (size_t)nch_src0 * pc->ntaps + (size_t)(c - nch_src0) * pc->ntaps1;
            for (int t = 0; t < nt; ++t)
                for (int co = 0; co < pc->cout; ++co) {
                    // slab = 4 pair planes; plane pp holds the 16-byte pieces 2pp and 2pp+1 of every row
                    const size_t slab = ((size_t)ph * slabs_per_phase + slab0 + t) * pc->cout * ck;
                    for (int e = 0; e < ck; ++e) {
                        const int cs = (c - ch0[s]) * ck + e;
                        float v = 0.f;
                        if (co < sp.cout && cs < sp.src[s]) v = sp.w(ph, t, co, off[s] + cs);
                        const int piece = e / epp;
                        const size_t pos = slab + (((size_t)(piece >> 1) * pc->cout + co) * 2 + (piece & 1)) * epp + e % epp;
                        if (sp.dtype == LDM_BF16) wh[pos] = host_f2bf(v);
                        else wf[pos] = v;
                    }
                }
        }
    std::vector<float> bias(pc->cout, 0.f);
    for (int co = 0; co < sp.cout; ++co) bias[co] = sp.bias(co);
    pc->w_bytes = nel * es;
    LDM_TRY(pool.upload(&pc->w, sp.dtype == LDM_BF16 ? (const void*)wh.data() : (const void*)wf.data(), pc->w_bytes));
    LDM_TRY(pool.upload((void**)&pc->bias, bias.data(), bias.size() * 4));
    if (src_pad) { src_pad[0] = pad[0]; src_pad[1] = pad[1]; }
    return LDM_OK;
}

}  // namespace ldm

using namespace ldm;

// ---------------------------------------------------------------------------------------------
// modules
// ---------------------------------------------------------------------------------------------
struct ResBlockP {
    PackedConv c1, c2;
    int C = 0;
    int tv_off = 0;   // offset of this block's time vector inside a row of the time-vector table
    // training mode (batch statistics): the two convolutions WITHOUT their BatchNorm folded in, and the affine
    // parameters of bn1 / bn2 as device arrays [C]; packed on the first training-mode forward
    PackedConv c1_raw, c2_raw;
    float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
};
struct AttnP {
    PackedConv qkv, out;
    int C = 0;
    // bf16, C = 256 / 512: the projections folded (launch_attention_folded): q' | v' as one 1x1 convolution C -> 2C with
    // q' = x (Wq Wk^T) + Wk bq, v' = x (Wv Wo), and the constant bv Wo + bo that joins the residual
    PackedConv qv;
    float* fold_bias = nullptr;
};

struct UnetBufs {   // NHWC activation buffers of one forward at fixed (B, H, W)
    void *xin = nullptr, *eps = nullptr;
    void *a1 = nullptr, *b1 = nullptr, *s1 = nullptr;
    void *a2 = nullptr, *b2 = nullptr, *s2 = nullptr;
    void *a3 = nullptr, *b3 = nullptr, *c3 = nullptr, *qkv3 = nullptr, *att3 = nullptr;
    void *qkv2 = nullptr, *att2 = nullptr;
    float *sinemb = nullptr, *h1 = nullptr, *temb = nullptr, *tvec = nullptr;   // time path, `rows` rows
    float* bn_ws = nullptr;   // training mode: workspace of the batch-statistics kernels (largest level)
    size_t bytes = 0;
};

// The UNet at 4x4 latents (the size the reference's VAE produces, src/main.rs:84-92) with every spatial layer
// written as one matrix between whole images: row (input pixel, channel) -> column (output pixel, channel).
// A 4x4, 2x2 or 1x1 image fills a sixteenth or less of a 4x16-pixel tile of the tap kernel; as matrices the
// layers are 1x1 convolutions over a batch of B "pixels" (one per image) with 16*C, 4*C or C channels, and the
// tiles are full.  The per-pixel layers (attention projections, up_conv1/2) keep their packed weights and run
// on the same buffers viewed as B*P pixels.
struct Dense4 {
    bool ready = false;
    PackedConv down1, down2, down3, up1, up2, out_conv;
    PackedConv rb[7][2];
    // compositions of consecutive linear layers, for batches small enough that a launch costs more than its FLOPs:
    // up1 . up_conv1 and up2 . up_conv2 over the sources {x, skip}, and the attention block at one token per image
    // (soft-max weight 1): out(val(x)) + x as one matrix
    PackedConv up1c, up2c, att1;
    int tv_off[7] = {0, 0, 0, 0, 0, 0, 0};   // per block: its time vector repeated for each of its pixels
    int tv_total = 0;
    float *twt = nullptr, *tbt = nullptr;    // time projections with the columns repeated accordingly
};

struct ldm_module {
    int kind = 0, dtype = 0, in_ch = 0, te = 0;
    int ch[3] = {64, 128, 256};
    bool scaled = false, finalized = false;
    bool train_ready = false;   // the ResidualBlocks' unfolded convolutions are packed (first training-mode forward)
    std::map<std::string, std::vector<float>> params;   // after finalize: only what a training-mode forward still needs
    std::map<std::string, std::pair<std::vector<float>, std::vector<float>>> composed;   // compose_up results, until finalize ends
    DevPool pool;
    // packed network
    int clp = 0;   // latent channels padded to the kernel's channel chunk
    PackedConv down1, down2, down3, up1, up_conv1, up2, up_conv2, out_conv;
    PackedConv up1_fused, up2_fused;   // up1 / up2 composed with up_conv1 / up_conv2 (bf16), see pack_up_fused
    bool have_up_fused = false;
    ResBlockP rb[7];
    AttnP mid_atten, up_atten;
    float *tw1 = nullptr, *tb1 = nullptr, *tw2 = nullptr, *tb2 = nullptr, *twt = nullptr, *tbt = nullptr;
    int tv_total = 0;
    // VAE decoder (kind LDM_MOD_VAE_DECODER): in_ch = latent_dimen, te = image channels
    Dense4 d4;
    PackedConv rev1, rev2, rev3;
    void* lin_w = nullptr;        // transfer layer: bf16 MFMA fragments (bf16) or f32 [K][N] (f32), columns NHWC-ordered
    float* lin_b = nullptr;
    void* rev3_w = nullptr;       // last layer for up to 4 image channels (bf16): MFMA fragments [9][64][8]
    float* rev3_b = nullptr;
    // grow-on-demand workspace for the module-level forwards
    void* ws = nullptr;
    size_t ws_bytes = 0;
    ~ldm_module() { if (ws) (void)hipFree(ws); }
};

struct ldm_conv {
    PackedConv pc;
    DevPool pool;
    int cin0 = 0, cin1 = 0;
};

static const char* kResNames[7] = {"down_res1", "down_res2", "down_res3", "mid_res1", "mid_res2", "up_res1", "up_res2"};

static int es_of(int dtype) { return dtype == LDM_BF16 ? 2 : 4; }

static int get_param(ldm_module* m, const std::string& name, size_t numel, const std::vector<float>** out) {
    auto it = m->params.find(name);
    if (it == m->params.end()) LDM_FAIL(LDM_ERR_STATE, "parameter '" + name + "' was not set");
    if (it->second.size() != numel)
        LDM_FAIL(LDM_ERR_INVALID, "parameter '" + name + "' has " + std::to_string(it->second.size()) + " elements, expected " + std::to_string(numel));
    *out = &it->second;
    return LDM_OK;
}

// conv (+ optional BatchNorm) -> PackedConv.  prefix "" means the names are used without a leading dot.
static std::string pjoin(const std::string& prefix, const std::string& leaf) { return prefix.empty() ? leaf : prefix + "." + leaf; }

static int pack_conv3x3(ldm_module* m, const std::string& conv, const std::string& bn, int cin, int cout, int stride,
                        PackedConv* pc, int cin_padto = 0) {
    const std::vector<float>*w, *b, *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
    LDM_TRY(get_param(m, pjoin(conv, "weight"), (size_t)cout * cin * 9, &w));
    LDM_TRY(get_param(m, pjoin(conv, "bias"), cout, &b));
    std::vector<float> scale(cout, 1.f), shift(cout, 0.f);
    if (!bn.empty()) {
        LDM_TRY(get_param(m, pjoin(bn, "gamma"), cout, &g));
        LDM_TRY(get_param(m, pjoin(bn, "beta"), cout, &be));
        LDM_TRY(get_param(m, pjoin(bn, "running_mean"), cout, &mu));
        LDM_TRY(get_param(m, pjoin(bn, "running_var"), cout, &var));
        for (int c = 0; c < cout; ++c) {
            // y = (conv + b - mean) / sqrt(var + eps) * gamma + beta
            const double s = (double)(*g)[c] / std::sqrt((double)(*var)[c] + 1e-5);
            scale[c] = (float)s;
            shift[c] = (float)(((double)(*b)[c] - (double)(*mu)[c]) * s + (double)(*be)[c]);
        }
    } else {
        for (int c = 0; c < cout; ++c) shift[c] = (*b)[c];
    }
    PackSpec sp;
    sp.kind = stride == 2 ? CONV3x3_S2 : CONV3x3_S1;
    sp.dtype = m->dtype;
    sp.cout = cout;
    sp.src = {cin};
    sp.src0_padto = cin_padto;
    sp.w = [&](int, int tap, int co, int ci) { return (*w)[((size_t)co * cin + ci) * 9 + tap] * scale[co]; };
    sp.bias = [&](int co) { return shift[co]; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

// y = (x - mean) / sqrt(var + eps) * gamma + beta after a layer with bias b, as a scale of the layer's weights
// and a new bias (inference BatchNorm, eps 1e-5); scale 1 / shift b without a BatchNorm
static int bn_fold(ldm_module* m, const std::string& bn, const std::vector<float>& b, int cout, std::vector<float>* scale,
                   std::vector<float>* shift) {
    scale->assign(cout, 1.f);
    shift->assign(b.begin(), b.begin() + cout);
    if (bn.empty()) return LDM_OK;
    const std::vector<float>*g, *be, *mu, *var;
    LDM_TRY(get_param(m, pjoin(bn, "gamma"), cout, &g));
    LDM_TRY(get_param(m, pjoin(bn, "beta"), cout, &be));
    LDM_TRY(get_param(m, pjoin(bn, "running_mean"), cout, &mu));
    LDM_TRY(get_param(m, pjoin(bn, "running_var"), cout, &var));
    for (int c = 0; c < cout; ++c) {
        const double sc = (double)(*g)[c] / std::sqrt((double)(*var)[c] + 1e-5);
        (*scale)[c] = (float)sc;
        (*shift)[c] = (float)(((double)b[c] - (double)(*mu)[c]) * sc + (double)(*be)[c]);
    }
    return LDM_OK;
}

static int pack_convT(ldm_module* m, const std::string& name, int cin, int cout, PackedConv* pc, const std::string& bn = "") {
    const std::vector<float>*w, *b;
    LDM_TRY(get_param(m, pjoin(name, "weight"), (size_t)cin * cout * 16, &w));
    LDM_TRY(get_param(m, pjoin(name, "bias"), cout, &b));
    std::vector<float> scale, shift;
    LDM_TRY(bn_fold(m, bn, *b, cout, &scale, &shift));
    PackSpec sp;
    sp.kind = CONVT4x4_S2;
    sp.dtype = m->dtype;
    sp.cout = cout;
    sp.src = {cin};
    // out[2i+py, 2j+px] += in[i - 1 + py + ta, j - 1 + px + tb] * W[ci][co][3 - py - 2ta][3 - px - 2tb]
    sp.w = [&](int ph, int tap, int co, int ci) {
        const int py = ph >> 1, px = ph & 1, ta = tap >> 1, tb = tap & 1;
        const int ky = 3 - py - 2 * ta, kx = 3 - px - 2 * tb;
        return (*w)[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx] * scale[co];
    };
    sp.bias = [&](int co) { return shift[co]; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

// ConvTranspose2d 3x3, stride 1, padding 1 (weights [cin,cout,3,3]) is a 3x3 convolution with the kernel
// flipped: out[y,x] = sum in[y + 1 - ky, x + 1 - kx] W[ci][co][ky][kx], i.e. tap (dy,dx) = (2-ky, 2-kx).
static int pack_convT3x3_s1(ldm_module* m, const std::string& name, int cin, int cout, int cin_padto, PackedConv* pc) {
    const std::vector<float>*w, *b;
    LDM_TRY(get_param(m, pjoin(name, "weight"), (size_t)cin * cout * 9, &w));
    LDM_TRY(get_param(m, pjoin(name, "bias"), cout, &b));
    PackSpec sp;
    sp.kind = CONV3x3_S1;
    sp.dtype = m->dtype;
    sp.cout = cout;
    sp.src = {cin};
    sp.src0_padto = cin_padto;
    sp.w = [&](int, int tap, int co, int ci) {
        const int ky = 2 - tap / 3, kx = 2 - tap % 3;
        return (*w)[(((size_t)ci * cout + co) * 3 + ky) * 3 + kx];
    };
    sp.bias = [&](int co) { return (*b)[co]; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

static int pack_conv1x1_cat(ldm_module* m, const std::string& name, int cin0, int cin1, int cout, PackedConv* pc) {
    const std::vector<float>*w, *b;
    const int cin = cin0 + cin1;
    LDM_TRY(get_param(m, pjoin(name, "weight"), (size_t)cout * cin, &w));
    LDM_TRY(get_param(m, pjoin(name, "bias"), cout, &b));
    PackSpec sp;
    sp.kind = CONV1x1;
    sp.dtype = m->dtype;
    sp.cout = cout;
    sp.src = {cin0, cin1};
    sp.w = [&](int, int, int co, int ci) { return (*w)[(size_t)co * cin + ci]; };
    sp.bias = [&](int co) { return (*b)[co]; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

// up (convT 4x4/s2, cin -> cmid) followed by the 1x1 conv over cat[up, skip] (cmid + cskip -> cout), composed in
// double precision: wc [cin][cout][4][4] = W1a . WT, bc = b1 + W1a . bT, and w1 (the 1x1 weights, [cout][cmid+cskip])
static int compose_up(ldm_module* m, const std::string& up, const std::string& conv, int cin, int cmid, int cskip,
                      int cout, std::vector<float>* wc, std::vector<float>* bc, const std::vector<float>** w1_out) {
    const std::vector<float>*wt, *bt, *w1, *b1;
    LDM_TRY(get_param(m, pjoin(up, "weight"), (size_t)cin * cmid * 16, &wt));
    LDM_TRY(get_param(m, pjoin(up, "bias"), cmid, &bt));
    LDM_TRY(get_param(m, pjoin(conv, "weight"), (size_t)cout * (cmid + cskip), &w1));
    LDM_TRY(get_param(m, pjoin(conv, "bias"), cout, &b1));
    const int c1 = cmid + cskip;
    *w1_out = w1;
    auto hit = m->composed.find(up);
    if (hit != m->composed.end()) { *wc = hit->second.first; *bc = hit->second.second; return LDM_OK; }
    wc->assign((size_t)cin * cout * 16, 0.f);
    for (int ci = 0; ci < cin; ++ci)
        for (int co = 0; co < cout; ++co)
            for (int k = 0; k < 16; ++k) {
                double acc = 0.0;
                for (int mm = 0; mm < cmid; ++mm)
                    acc += (double)(*w1)[(size_t)co * c1 + mm] * (double)(*wt)[((size_t)ci * cmid + mm) * 16 + k];
                (*wc)[((size_t)ci * cout + co) * 16 + k] = (float)acc;
            }
    bc->assign(cout, 0.f);
    for (int co = 0; co < cout; ++co) {
        double acc = (*b1)[co];
        for (int mm = 0; mm < cmid; ++mm) acc += (double)(*w1)[(size_t)co * c1 + mm] * (double)(*bt)[mm];
        (*bc)[co] = (float)acc;
    }
    m->composed[up] = std::make_pair(*wc, *bc);
    return LDM_OK;
}

// up (convT 4x4/s2, cin -> cmid) followed by the 1x1 conv over cat[up, skip] (cmid + cskip -> cout), as one
// transposed conv over the sources {x, skip}: the 1x1 conv is linear, so
//   out = (W1a . WT) (*) x + W1b . skip + (b1 + W1a . bT)
// with W1 = [W1a | W1b] split at cmid.  Composed in double precision.  The skip source is sampled at
// the output pixel itself: its chunks have a single tap.
static int pack_up_fused(ldm_module* m, const std::string& up, const std::string& conv, int cin, int cmid, int cskip,
                         int cout, PackedConv* pc) {
    std::vector<float> wc, bc;
    const std::vector<float>* w1;
    LDM_TRY(compose_up(m, up, conv, cin, cmid, cskip, cout, &wc, &bc, &w1));
    const int c1 = cmid + cskip;
    PackSpec sp;
    sp.kind = CONVT4x4_S2;
    sp.dtype = m->dtype;
    sp.cout = cout;
    sp.src = {cin, cskip};
    sp.ntaps1 = 1;
    sp.w = [&](int ph, int tap, int co, int ci) {
        if (ci >= cin) return tap == 0 ? (*w1)[(size_t)co * c1 + cmid + (ci - cin)] : 0.f;
        const int py = ph >> 1, px = ph & 1, ta = tap >> 1, tb = tap & 1;
        const int ky = 3 - py - 2 * ta, kx = 3 - px - 2 * tb;
        return wc[((size_t)ci * cout + co) * 16 + ky * 4 + kx];
    };
    sp.bias = [&](int co) { return bc[co]; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

static int pack_attention(ldm_module* m, const std::string& prefix, int C, AttnP* ap) {
    const std::vector<float>*wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;
    LDM_TRY(get_param(m, pjoin(prefix, "question.weight"), (size_t)C * C, &wq));
    LDM_TRY(get_param(m, pjoin(prefix, "question.bias"), C, &bq));
    LDM_TRY(get_param(m, pjoin(prefix, "key.weight"), (size_t)C * C, &wk));
    LDM_TRY(get_param(m, pjoin(prefix, "key.bias"), C, &bk));
    LDM_TRY(get_param(m, pjoin(prefix, "val.weight"), (size_t)C * C, &wv));
    LDM_TRY(get_param(m, pjoin(prefix, "val.bias"), C, &bv));
    LDM_TRY(get_param(m, pjoin(prefix, "out.weight"), (size_t)C * C, &wo));
    LDM_TRY(get_param(m, pjoin(prefix, "out.bias"), C, &bo));
    ap->C = C;
    PackSpec sp;
    sp.kind = CONV1x1;
    sp.dtype = m->dtype;
    sp.cout = 3 * C;
    sp.src = {C};
    // Linear weights are [d_in, d_out]
    sp.w = [&](int, int, int co, int ci) {
        const std::vector<float>* ws[3] = {wq, wk, wv};
        return (*ws[co / C])[(size_t)ci * C + co % C];
    };
    sp.bias = [&](int co) {
        const std::vector<float>* bs[3] = {bq, bk, bv};
        return (*bs[co / C])[co % C];
    };
    LDM_TRY(pack_conv(sp, m->pool, &ap->qkv, nullptr));
    PackSpec so;
    so.kind = CONV1x1;
    so.dtype = m->dtype;
    so.cout = C;
    so.src = {C};
    so.w = [&](int, int, int co, int ci) { return (*wo)[(size_t)ci * C + co]; };
    so.bias = [&](int co) { return (*bo)[co]; };
    LDM_TRY(pack_conv(so, m->pool, &ap->out, nullptr));
    if (m->dtype == LDM_BF16 && (C == 256 || C == 512)) {
        // Linear weights are [d_in, d_out].  Composed in double precision:
        //   q_i . k_j = (x_i Wq + bq) . (x_j Wk + bk) = x_i (Wq Wk^T) x_j^T + (Wk bq) . x_j + (terms without j)
        //   (P (x Wv + 1 bv^T)) Wo + bo = P (x (Wv Wo)) + (bv Wo + bo)          (the rows of P sum to one)
        std::vector<double> a((size_t)C * C), vo((size_t)C * C), aq(C), bf(C);
        for (int ci = 0; ci < C; ++ci)
            for (int c = 0; c < C; ++c) {
                double sa = 0.0, sv = 0.0;
                for (int mm = 0; mm < C; ++mm) {
                    sa += (double)(*wq)[(size_t)ci * C + mm] * (double)(*wk)[(size_t)c * C + mm];
                    sv += (double)(*wv)[(size_t)ci * C + mm] * (double)(*wo)[(size_t)mm * C + c];
                }
                a[(size_t)ci * C + c] = sa;
                vo[(size_t)ci * C + c] = sv;
            }
        for (int c = 0; c < C; ++c) {
            double sq = 0.0, sb = (*bo)[c];
            for (int mm = 0; mm < C; ++mm) {
                sq += (double)(*bq)[mm] * (double)(*wk)[(size_t)c * C + mm];
                sb += (double)(*bv)[mm] * (double)(*wo)[(size_t)mm * C + c];
            }
            aq[c] = sq;
            bf[c] = sb;
        }
        PackSpec sf;
        sf.kind = CONV1x1;
        sf.dtype = m->dtype;
        sf.cout = 2 * C;
        sf.src = {C};
        sf.w = [&](int, int, int co, int ci) { return (float)(co < C ? a[(size_t)ci * C + co] : vo[(size_t)ci * C + co - C]); };
        sf.bias = [&](int co) { return co < C ? (float)aq[co] : 0.f; };
        LDM_TRY(pack_conv(sf, m->pool, &ap->qv, nullptr));
        std::vector<float> bff(C);
        for (int c = 0; c < C; ++c) bff[c] = (float)bf[c];
        LDM_TRY(m->pool.upload((void**)&ap->fold_bias, bff.data(), bff.size() * 4));
    }
    return LDM_OK;
}

// One spatial layer between whole Hin x Win images as a matrix (see Dense4).  kind: CONV3x3_S1 / CONV3x3_S2 with
// weights [cout,cin,3,3] (padding 1), CONVT4x4_S2 with weights [cin,cout,4,4] (padding 1).  Channels are padded
// to cin_pad / cout_pad per pixel, the layout of the NHWC buffers.
static int pack_dense_layer(ldm_module* m, int kind, const std::string& name, const std::string& bn, int Hin, int Win,
                            int cin, int cin_pad, int cout, int cout_pad, PackedConv* pc) {
    const bool tr = kind == CONVT4x4_S2;
    const int kk = tr ? 16 : 9, kw = tr ? 4 : 3, st = kind == CONV3x3_S1 ? 1 : 2;
    const int Hout = tr ? Hin * 2 : Hin / st, Wout = tr ? Win * 2 : Win / st;
    const std::vector<float>*w, *b;
    LDM_TRY(get_param(m, pjoin(name, "weight"), (size_t)cin * cout * kk, &w));
    LDM_TRY(get_param(m, pjoin(name, "bias"), cout, &b));
    std::vector<float> scale, shift;
    LDM_TRY(bn_fold(m, bn, *b, cout, &scale, &shift));
    PackSpec sp;
    sp.kind = CONV1x1;
    sp.dtype = m->dtype;
    sp.cout = Hout * Wout * cout_pad;
    sp.src = {Hin * Win * cin_pad};
    sp.w = [&](int, int, int n, int k) {
        const int pi = k / cin_pad, ci = k % cin_pad, po = n / cout_pad, co = n % cout_pad;
        if (ci >= cin || co >= cout) return 0.f;
        const int yi = pi / Win, xi = pi % Win, yo = po / Wout, xo = po % Wout;
        // conv: input row yi = yo*stride + ky - 1; transposed conv: output row yo = 2*yi + ky - 1
        const int ky = tr ? yo - 2 * yi + 1 : yi - yo * st + 1, kx = tr ? xo - 2 * xi + 1 : xi - xo * st + 1;
        if (ky < 0 || ky >= kw || kx < 0 || kx >= kw) return 0.f;
        const size_t idx = tr ? (((size_t)ci * cout + co) * 4 + ky) * 4 + kx : (((size_t)co * cin + ci) * 3 + ky) * 3 + kx;
        return (*w)[idx] * scale[co];
    };
    sp.bias = [&](int n) { const int co = n % cout_pad; return co < cout ? shift[co] : 0.f; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

// The same composition between whole images (see Dense4): x is a Hin x Hin image, skip and the output 2Hin x 2Hin.
// Row (pixel, channel) of x, then of skip -> column (output pixel, channel); the skip part is block diagonal.
static int pack_dense_up_fused(ldm_module* m, const std::string& up, const std::string& conv, int Hin, int cin, int cmid,
                               int cskip, int cout, PackedConv* pc) {
    std::vector<float> wc, bc;
    const std::vector<float>* w1;
    LDM_TRY(compose_up(m, up, conv, cin, cmid, cskip, cout, &wc, &bc, &w1));
    const int c1 = cmid + cskip, Win = Hin, Wout = 2 * Hin, Pin = Hin * Hin, Pout = 4 * Pin;
    PackSpec sp;
    sp.kind = CONV1x1;
    sp.dtype = m->dtype;
    sp.cout = Pout * cout;
    sp.src = {Pin * cin, Pout * cskip};
    sp.w = [&](int, int, int n, int k) {
        const int po = n / cout, co = n % cout;
        if (k >= Pin * cin) {
            const int ps = (k - Pin * cin) / cskip, cs = (k - Pin * cin) % cskip;
            return ps == po ? (*w1)[(size_t)co * c1 + cmid + cs] : 0.f;
        }
        const int pi = k / cin, ci = k % cin;
        const int ky = po / Wout - 2 * (pi / Win) + 1, kx = po % Wout - 2 * (pi % Win) + 1;
        if (ky < 0 || ky >= 4 || kx < 0 || kx >= 4) return 0.f;
        return wc[((size_t)ci * cout + co) * 16 + ky * 4 + kx];
    };
    sp.bias = [&](int n) { return bc[n % cout]; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

// Attention with one token per image: soft-max over a single key is 1, so Attention::forward (src/main.rs:256-276)
// is out(val(x)) + x; the two Linears composed in double precision (the residual stays in the epilogue).
static int pack_attention_one_token(ldm_module* m, const std::string& prefix, int C, PackedConv* pc) {
    const std::vector<float>*wv, *bv, *wo, *bo;
    LDM_TRY(get_param(m, pjoin(prefix, "val.weight"), (size_t)C * C, &wv));
    LDM_TRY(get_param(m, pjoin(prefix, "val.bias"), C, &bv));
    LDM_TRY(get_param(m, pjoin(prefix, "out.weight"), (size_t)C * C, &wo));
    LDM_TRY(get_param(m, pjoin(prefix, "out.bias"), C, &bo));
    std::vector<float> w((size_t)C * C), b(C);     // Linear weights are [d_in, d_out]
    for (int ci = 0; ci < C; ++ci)
        for (int co = 0; co < C; ++co) {
            double acc = 0.0;
            for (int mm = 0; mm < C; ++mm) acc += (double)(*wv)[(size_t)ci * C + mm] * (double)(*wo)[(size_t)mm * C + co];
            w[(size_t)ci * C + co] = (float)acc;
        }
    for (int co = 0; co < C; ++co) {
        double acc = (*bo)[co];
        for (int mm = 0; mm < C; ++mm) acc += (double)(*bv)[mm] * (double)(*wo)[(size_t)mm * C + co];
        b[co] = (float)acc;
    }
    PackSpec sp;
    sp.kind = CONV1x1;
    sp.dtype = m->dtype;
    sp.cout = C;
    sp.src = {C};
    sp.w = [&](int, int, int co, int ci) { return w[(size_t)ci * C + co]; };
    sp.bias = [&](int co) { return b[co]; };
    return pack_conv(sp, m->pool, pc, nullptr);
}

// wt [te][tv] / bt [tv]: the per-block time projections side by side (block i at column m->rb[i].tv_off)
static int pack_dense4(ldm_module* m, const std::vector<float>& wt, const std::vector<float>& bt) {
    Dense4& d = m->d4;
    const int c1 = m->ch[0], c2 = m->ch[1], c3 = m->ch[2], cl = m->in_ch, te = m->te, tv = m->tv_total;
    const int rc[7] = {c1, c2, c3, c3, c3, c2, c1};
    const int rs[7] = {4, 2, 1, 1, 1, 2, 4};        // image side at each block
    int off = 0;
    for (int i = 0; i < 7; ++i) { d.tv_off[i] = off; off += rs[i] * rs[i] * rc[i]; }
    d.tv_total = off;
    std::vector<float> wd((size_t)te * off), bd(off);
    for (int i = 0; i < 7; ++i)
        for (int p = 0; p < rs[i] * rs[i]; ++p)
            for (int c = 0; c < rc[i]; ++c) {
                const int dst = d.tv_off[i] + p * rc[i] + c, src = m->rb[i].tv_off + c;
                bd[dst] = bt[src];
                for (int k = 0; k < te; ++k) wd[(size_t)k * off + dst] = wt[(size_t)k * tv + src];
            }
    LDM_TRY(m->pool.upload((void**)&d.twt, wd.data(), wd.size() * 4));
    LDM_TRY(m->pool.upload((void**)&d.tbt, bd.data(), bd.size() * 4));
    for (int i = 0; i < 7; ++i) {
        const std::string pre = kResNames[i];
        LDM_TRY(pack_dense_layer(m, CONV3x3_S1, pre + ".conv1", pre + ".bn1", rs[i], rs[i], rc[i], rc[i], rc[i], rc[i], &d.rb[i][0]));
        LDM_TRY(pack_dense_layer(m, CONV3x3_S1, pre + ".conv2", pre + ".bn2", rs[i], rs[i], rc[i], rc[i], rc[i], rc[i], &d.rb[i][1]));
    }
    LDM_TRY(pack_dense_layer(m, CONV3x3_S1, "down1", "", 4, 4, cl, m->clp, c1, c1, &d.down1));
    LDM_TRY(pack_dense_layer(m, CONV3x3_S2, "down2", "", 4, 4, c1, c1, c2, c2, &d.down2));
    LDM_TRY(pack_dense_layer(m, CONV3x3_S2, "down3", "", 2, 2, c2, c2, c3, c3, &d.down3));
    LDM_TRY(pack_dense_layer(m, CONVT4x4_S2, "up1", "", 1, 1, c3, c3, c2, c2, &d.up1));
    LDM_TRY(pack_dense_layer(m, CONVT4x4_S2, "up2", "", 2, 2, c2, c2, c1, c1, &d.up2));
    LDM_TRY(pack_dense_layer(m, CONV3x3_S1, "out_conv", "", 4, 4, c1, c1, cl, m->clp, &d.out_conv));
    LDM_TRY(pack_dense_up_fused(m, "up1", "up_conv1", 1, c3, c2, c2, c2, &d.up1c));
    LDM_TRY(pack_dense_up_fused(m, "up2", "up_conv2", 2, c2, c1, c1, c1, &d.up2c));
    LDM_TRY(pack_attention_one_token(m, "mid_atten", c3, &d.att1));
    d.ready = true;
    return LDM_OK;
}

static int pack_resblock(ldm_module* m, const std::string& prefix, int C, ResBlockP* rp) {
    rp->C = C;
    LDM_TRY(pack_conv3x3(m, pjoin(prefix, "conv1"), pjoin(prefix, "bn1"), C, C, 1, &rp->c1));
    LDM_TRY(pack_conv3x3(m, pjoin(prefix, "conv2"), pjoin(prefix, "bn2"), C, C, 1, &rp->c2));
    return LDM_OK;
}

static int finalize_module(ldm_module* m) {
    LDM_REQUIRE(!m->finalized, "module already finalized");
    const int es = es_of(m->dtype), ck = 128 / es;
    const int te = m->te;
    if (m->kind == LDM_MOD_TIME_ADDITION) {
        const int d = m->in_ch;
        const std::vector<float>*w1, *b1, *w2, *b2;
        LDM_TRY(get_param(m, "linear1.weight", (size_t)d * d * 8, &w1));
        LDM_TRY(get_param(m, "linear1.bias", (size_t)d * 8, &b1));
        LDM_TRY(get_param(m, "linear2.weight", (size_t)d * 8 * d, &w2));
        LDM_TRY(get_param(m, "linear2.bias", d, &b2));
        LDM_TRY(m->pool.upload((void**)&m->tw1, w1->data(), w1->size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tb1, b1->data(), b1->size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tw2, w2->data(), w2->size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tb2, b2->data(), b2->size() * 4));
    } else if (m->kind == LDM_MOD_VAE_DECODER) {
        // SimpleDecoder::new, src/main.rs:108-147.  The transfer layer's output [B, 128*16*16] is reshaped to
        // NCHW [B,128,16,16] (:157): column c*256 + y*16 + x.  Its columns are re-ordered here to
        // (y*16 + x)*128 + c so that the Linear writes the NHWC tensor the transposed convs read.
        const int K = m->in_ch * 16, N = 128 * 16 * 16;
        const std::vector<float>*w, *b;
        LDM_TRY(get_param(m, "transfer.weight", (size_t)K * N, &w));
        LDM_TRY(get_param(m, "transfer.bias", N, &b));
        auto col = [](int n) { return (n & 127) * 256 + (n >> 7); };   // NHWC-ordered column -> reference column
        std::vector<float> bp(N);
        for (int n = 0; n < N; ++n) bp[n] = (*b)[col(n)];
        LDM_TRY(m->pool.upload((void**)&m->lin_b, bp.data(), bp.size() * 4));
        if (m->dtype == LDM_BF16) {
            LDM_REQUIRE(K % 128 == 0 && K <= 1024, "latent_dimen must be a multiple of 8 up to 64 for the bf16 decoder");
            const int KS = K / 32;
            std::vector<bf16_t> wp((size_t)K * N);
            for (int nt = 0; nt < N / 16; ++nt)
                for (int ks = 0; ks < KS; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int n = nt * 16 + (lane & 15), k0 = ks * 32 + (lane >> 4) * 8;
                        const int cn = col(n);
                        bf16_t* dst = &wp[(((size_t)nt * KS + ks) * 64 + lane) * 8];
                        for (int e = 0; e < 8; ++e) dst[e] = host_f2bf((*w)[(size_t)(k0 + e) * N + cn]);
                    }
            LDM_TRY(m->pool.upload(&m->lin_w, wp.data(), wp.size() * 2));
        } else {
            std::vector<float> wp((size_t)K * N);
            for (int k = 0; k < K; ++k)
                for (int n = 0; n < N; ++n) wp[(size_t)k * N + n] = (*w)[(size_t)k * N + col(n)];
            LDM_TRY(m->pool.upload(&m->lin_w, wp.data(), wp.size() * 4));
        }
        LDM_TRY(pack_convT(m, "reverse1", 128, 64, &m->rev1, "bn_1"));
        LDM_TRY(pack_convT(m, "reverse2", 64, 32, &m->rev2, "bn_2"));
        LDM_TRY(pack_convT3x3_s1(m, "reverse3", 32, m->te, 64, &m->rev3));
        if (m->te <= 4 && m->dtype == LDM_BF16) {
            // the same flipped kernel as MFMA A fragments for the direct few-channel kernel: tap (dy,dx) = (2-ky, 2-kx)
            const int CO = m->te;
            const std::vector<float>*w3, *b3;
            LDM_TRY(get_param(m, "reverse3.weight", (size_t)32 * CO * 9, &w3));
            LDM_TRY(get_param(m, "reverse3.bias", CO, &b3));
            std::vector<bf16_t> wt((size_t)9 * 64 * 8, host_f2bf(0.f));
            for (int tap = 0; tap < 9; ++tap)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = lane & 15, k0 = (lane >> 4) * 8;
                    if (co >= CO) continue;
                    for (int e = 0; e < 8; ++e)
                        wt[((size_t)tap * 64 + lane) * 8 + e] =
                            host_f2bf((*w3)[(((size_t)(k0 + e) * CO + co) * 3 + (2 - tap / 3)) * 3 + (2 - tap % 3)]);
                }
            LDM_TRY(m->pool.upload(&m->rev3_w, wt.data(), wt.size() * 2));
            LDM_TRY(m->pool.upload((void**)&m->rev3_b, b3->data(), b3->size() * 4));
        }
    } else if (m->kind == LDM_MOD_ATTENTION) {
        LDM_TRY(pack_attention(m, "", m->in_ch, &m->mid_atten));
    } else if (m->kind == LDM_MOD_RESBLOCK) {
        const int C = m->in_ch;
        LDM_REQUIRE(C % 64 == 0, "ResidualBlock channels must be a multiple of 64");
        LDM_TRY(pack_resblock(m, "", C, &m->rb[0]));
        const std::vector<float>*wt, *bt;
        LDM_TRY(get_param(m, "time.weight", (size_t)te * C, &wt));
        LDM_TRY(get_param(m, "time.bias", C, &bt));
        LDM_TRY(m->pool.upload((void**)&m->twt, wt->data(), wt->size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tbt, bt->data(), bt->size() * 4));
        m->rb[0].tv_off = 0;
        m->tv_total = C;
    } else {
        const int c1 = m->ch[0], c2 = m->ch[1], c3 = m->ch[2], cl = m->in_ch;
        LDM_REQUIRE(c1 % 64 == 0 && c2 % 64 == 0 && c3 % 64 == 0, "UNet widths must be multiples of 64");
        m->clp = round_up_i(cl, 64);   // = out_conv's padded width, so z and eps share one NHWC layout
        (void)ck;
        // time MLP and the per-block time projections (all f32)
        const std::vector<float>*w1, *b1, *w2, *b2;
        LDM_TRY(get_param(m, "time_emb.linear1.weight", (size_t)te * te * 8, &w1));
        LDM_TRY(get_param(m, "time_emb.linear1.bias", (size_t)te * 8, &b1));
        LDM_TRY(get_param(m, "time_emb.linear2.weight", (size_t)te * 8 * te, &w2));
        LDM_TRY(get_param(m, "time_emb.linear2.bias", te, &b2));
        LDM_TRY(m->pool.upload((void**)&m->tw1, w1->data(), w1->size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tb1, b1->data(), b1->size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tw2, w2->data(), w2->size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tb2, b2->data(), b2->size() * 4));
        const int rc[7] = {c1, c2, c3, c3, c3, c2, c1};
        int tv = 0;
        for (int i = 0; i < 7; ++i) { m->rb[i].tv_off = tv; tv += rc[i]; }
        m->tv_total = tv;
        std::vector<float> wt((size_t)te * tv), bt(tv);
        for (int i = 0; i < 7; ++i) {
            const std::vector<float>*w, *b;
            LDM_TRY(get_param(m, std::string(kResNames[i]) + ".time.weight", (size_t)te * rc[i], &w));
            LDM_TRY(get_param(m, std::string(kResNames[i]) + ".time.bias", rc[i], &b));
            for (int k = 0; k < te; ++k)
                for (int c = 0; c < rc[i]; ++c) wt[(size_t)k * tv + m->rb[i].tv_off + c] = (*w)[(size_t)k * rc[i] + c];
            for (int c = 0; c < rc[i]; ++c) bt[m->rb[i].tv_off + c] = (*b)[c];
            LDM_TRY(pack_resblock(m, kResNames[i], rc[i], &m->rb[i]));
        }
        LDM_TRY(m->pool.upload((void**)&m->twt, wt.data(), wt.size() * 4));
        LDM_TRY(m->pool.upload((void**)&m->tbt, bt.data(), bt.size() * 4));
        LDM_TRY(pack_conv3x3(m, "down1", "", cl, c1, 1, &m->down1, m->clp));
        LDM_TRY(pack_conv3x3(m, "down2", "", c1, c2, 2, &m->down2));
        LDM_TRY(pack_conv3x3(m, "down3", "", c2, c3, 2, &m->down3));
        LDM_TRY(pack_attention(m, "mid_atten", c3, &m->mid_atten));
        LDM_TRY(pack_convT(m, "up1", c3, c2, &m->up1));
        LDM_TRY(pack_conv1x1_cat(m, "up_conv1", c2, c2, c2, &m->up_conv1));
        if (m->scaled) LDM_TRY(pack_attention(m, "up_atten", c2, &m->up_atten));
        LDM_TRY(pack_convT(m, "up2", c2, c1, &m->up2));
        LDM_TRY(pack_conv1x1_cat(m, "up_conv2", c1, c1, c1, &m->up_conv2));
        if (m->dtype == LDM_BF16) {
            LDM_TRY(pack_up_fused(m, "up1", "up_conv1", c3, c2, c2, c2, &m->up1_fused));
            LDM_TRY(pack_up_fused(m, "up2", "up_conv2", c2, c1, c1, c1, &m->up2_fused));
            m->have_up_fused = true;
        }
        LDM_TRY(pack_conv3x3(m, "out_conv", "", c1, cl, 1, &m->out_conv));
        LDM_REQUIRE(m->out_conv.cout == m->clp, "internal: padded output and padded latent widths differ");
        LDM_TRY(pack_dense4(m, wt, bt));
    }
    // A training-mode forward (batch statistics) needs the ResidualBlocks' convolutions without the BatchNorm folded
    // in and the affine parameters: those host arrays stay until that forward has packed them; all others go now.
    for (auto it = m->params.begin(); it != m->params.end();) {
        const std::string& k = it->first;
        const bool keep = (m->kind == LDM_MOD_UNET || m->kind == LDM_MOD_RESBLOCK) &&
                          (k.find("conv1.") != std::string::npos || k.find("conv2.") != std::string::npos ||
                           k.find("bn1.gamma") != std::string::npos || k.find("bn1.beta") != std::string::npos ||
                           k.find("bn2.gamma") != std::string::npos || k.find("bn2.beta") != std::string::npos) &&
                          k.find("up_conv") == std::string::npos;
        if (keep) ++it;
        else it = m->params.erase(it);
    }
    m->composed.clear();
    m->finalized = true;
    return LDM_OK;
}

// The ResidualBlocks as a training-mode forward runs them: conv1 / conv2 with their own bias only, gamma / beta of
// bn1 / bn2 on the device.  Runs once per module, on the first training-mode forward (host packing + uploads).
static int ensure_train_pack(ldm_module* m) {
    if (m->train_ready) return LDM_OK;
    LDM_REQUIRE(m->kind == LDM_MOD_UNET || m->kind == LDM_MOD_RESBLOCK, "training mode needs a UNet or ResidualBlock module");
    const int nb = m->kind == LDM_MOD_UNET ? 7 : 1;
    for (int i = 0; i < nb; ++i) {
        ResBlockP& rp = m->rb[i];
        const std::string pre = m->kind == LDM_MOD_UNET ? kResNames[i] : "";
        LDM_TRY(pack_conv3x3(m, pjoin(pre, "conv1"), "", rp.C, rp.C, 1, &rp.c1_raw));
        LDM_TRY(pack_conv3x3(m, pjoin(pre, "conv2"), "", rp.C, rp.C, 1, &rp.c2_raw));
        const std::vector<float>*g1, *b1, *g2, *b2;
        LDM_TRY(get_param(m, pjoin(pre, "bn1.gamma"), rp.C, &g1));
        LDM_TRY(get_param(m, pjoin(pre, "bn1.beta"), rp.C, &b1));
        LDM_TRY(get_param(m, pjoin(pre, "bn2.gamma"), rp.C, &g2));
        LDM_TRY(get_param(m, pjoin(pre, "bn2.beta"), rp.C, &b2));
        LDM_TRY(m->pool.upload((void**)&rp.g1, g1->data(), g1->size() * 4));
        LDM_TRY(m->pool.upload((void**)&rp.be1, b1->data(), b1->size() * 4));
        LDM_TRY(m->pool.upload((void**)&rp.g2, g2->data(), g2->size() * 4));
        LDM_TRY(m->pool.upload((void**)&rp.be2, b2->data(), b2->size() * 4));
    }
    m->params.clear();
    m->train_ready = true;
    return LDM_OK;
}

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------
struct TimeVec {   // where the per-block time vectors come from
    const float* table = nullptr;
    int img_stride = 0, step_stride = 0;
    int pix_stride = 0;   // whole-image matrices: the batch is one image of B pixels, each with its own row
    const int* step_ptr = nullptr;
};

static int run_conv(const PackedConv& pc, const void* in0, int cin0, const void* in1, int cin1, void* out, int B, int H,
                    int W, int act, const void* res, const TimeVec* tv, int tv_off, int out_f32, hipStream_t st) {
    ConvCall cc;
    cc.in0 = in0; cc.cin0 = cin0; cc.in1 = in1; cc.cin1 = cin1;
    cc.out = out; cc.B = B; cc.Hin = H; cc.Win = W;
    cc.act = act; cc.res = res; cc.out_f32 = out_f32;
    if (tv) {
        cc.tvec = tv->table + tv_off;
        cc.t_img_stride = tv->img_stride;
        cc.t_step_stride = tv->step_stride;
        cc.t_pix_stride = tv->pix_stride;
        cc.step_ptr = tv->step_ptr;
    }
    return launch_tapconv(pc, cc, st);
}

// ResidualBlock::forward (src/main.rs:372-387): tmp = silu(bn1(conv1(x))) + t ; out = silu(bn2(conv2(tmp)) + x)
static int run_resblock(const ResBlockP& rp, const void* x, void* tmp, void* out, int B, int H, int W, const TimeVec& tv,
                        hipStream_t st) {
    // LDM_RESFUSE=0 (developer switch, read once): keep the two launches where the fused kernel would run
    static const bool fuse = [] { const char* e = getenv("LDM_RESFUSE"); return !(e && e[0] == '0'); }();
    if (fuse && tv.pix_stride == 0 && resfuse_shape_ok(rp.c1.dtype, rp.C, B, H, W))
        return launch_resfuse(rp.c1, rp.c2, x, out, B, H, W, tv.table + rp.tv_off, tv.img_stride, tv.step_stride, tv.step_ptr, st);
    LDM_TRY(run_conv(rp.c1, x, rp.C, nullptr, 0, tmp, B, H, W, ACT_SILU, nullptr, &tv, rp.tv_off, 0, st));
    LDM_TRY(run_conv(rp.c2, tmp, rp.C, nullptr, 0, out, B, H, W, ACT_SILU, x, nullptr, 0, 0, st));
    return LDM_OK;
}

// Training-mode forward (batch statistics, active dropout): what DiffusionModel::forward runs on the autodiff
// backend (src/main.rs:635-679, :1318).
struct TrainCtx {
    float* bn_ws = nullptr;          // workspace of the statistics kernels
    float* bn_mean = nullptr;        // optional exports [2 * tv_total]: per block bn1 then bn2
    float* bn_var = nullptr;
    float keep_prob = 1.0f;          // mid_drop: 1 - p (1: no dropout)
    unsigned long long seed = 0;
};
constexpr float kBnEps = 1e-5f;      // burn's BatchNormConfig default

// ResidualBlock::forward with bn1 / bn2 on the statistics of the batch (src/main.rs:372-387):
//   tmp = conv1(x); tmp <- silu(bn1(tmp)) + t;   out = conv2(tmp); out <- dropout(silu(bn2(out) + x))
static int run_resblock_train(const ResBlockP& rp, int dtype, const void* x, void* tmp, void* out, int B, int H, int W,
                              const TimeVec& tv, const TrainCtx& tc, float keep_prob, hipStream_t st) {
    const long HW = (long)H * W;
    float* m1 = tc.bn_mean ? tc.bn_mean + 2 * rp.tv_off : nullptr;
    float* v1 = tc.bn_var ? tc.bn_var + 2 * rp.tv_off : nullptr;
    LDM_TRY(run_conv(rp.c1_raw, x, rp.C, nullptr, 0, tmp, B, H, W, ACT_NONE, nullptr, nullptr, 0, 0, st));
    LDM_TRY(launch_bn_stats(dtype, tmp, B, HW, rp.C, rp.g1, rp.be1, kBnEps, tc.bn_ws, m1, v1, st));
    LDM_TRY(launch_bn_apply(dtype, tmp, nullptr, tmp, B, HW, rp.C, tc.bn_ws, ACT_SILU, tv.table + rp.tv_off, tv.img_stride,
                            1.0f, 0, st));
    LDM_TRY(run_conv(rp.c2_raw, tmp, rp.C, nullptr, 0, out, B, H, W, ACT_NONE, nullptr, nullptr, 0, 0, st));
    LDM_TRY(launch_bn_stats(dtype, out, B, HW, rp.C, rp.g2, rp.be2, kBnEps, tc.bn_ws, m1 ? m1 + rp.C : nullptr,
                            v1 ? v1 + rp.C : nullptr, st));
    LDM_TRY(launch_bn_apply(dtype, out, x, out, B, HW, rp.C, tc.bn_ws, ACT_SILU, nullptr, 0, keep_prob, tc.seed, st));
    return LDM_OK;
}

// Attention::forward (src/main.rs:256-276): qkv projection, soft-max attention, out projection + x
static int run_attention(const AttnP& ap, int dtype, const void* x, void* qkv, void* att, void* out, int B, int H, int W,
                         hipStream_t st) {
    if (ap.fold_bias && attention_can_fold(dtype, B, H * W, ap.C)) {
        LDM_TRY(run_conv(ap.qv, x, ap.C, nullptr, 0, qkv, B, H, W, ACT_NONE, nullptr, nullptr, 0, 0, st));
        return launch_attention_folded(qkv, x, ap.fold_bias, out, B, H * W, ap.C, st);
    }
    LDM_TRY(run_conv(ap.qkv, x, ap.C, nullptr, 0, qkv, B, H, W, ACT_NONE, nullptr, nullptr, 0, 0, st));
    LDM_TRY(launch_attention(dtype, qkv, att, B, H * W, ap.C, st));
    LDM_TRY(run_conv(ap.out, att, ap.C, nullptr, 0, out, B, H, W, ACT_NONE, x, nullptr, 0, 0, st));
    return LDM_OK;
}

static size_t plan_unet_bufs(const ldm_module* m, int B, int H, int W, int time_rows, char* base, UnetBufs* ub,
                             int tv_total = -1, bool train = false) {
    if (tv_total < 0) tv_total = m->tv_total;
    const size_t es = es_of(m->dtype);
    const size_t p1 = (size_t)B * H * W, p2 = p1 / 4, p3 = p1 / 16;
    const int c1 = m->ch[0], c2 = m->ch[1], c3 = m->ch[2];
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* p = base ? base + off : nullptr;
        off += align_up(bytes, 256);
        return p;
    };
    ub->xin = take(p1 * m->clp * es);
    ub->eps = take(p1 * m->out_conv.cout * 4);
    ub->a1 = take(p1 * c1 * es); ub->b1 = take(p1 * c1 * es); ub->s1 = take(p1 * c1 * es);
    ub->a2 = take(p2 * c2 * es); ub->b2 = take(p2 * c2 * es); ub->s2 = take(p2 * c2 * es);
    ub->a3 = take(p3 * c3 * es); ub->b3 = take(p3 * c3 * es); ub->c3 = take(p3 * c3 * es);
    ub->qkv3 = take(p3 * 3 * c3 * es); ub->att3 = take(p3 * c3 * es);
    if (m->scaled) { ub->qkv2 = take(p2 * 3 * c2 * es); ub->att2 = take(p2 * c2 * es); }
    ub->sinemb = (float*)take((size_t)time_rows * m->te * 4);
    ub->h1 = (float*)take((size_t)time_rows * m->te * 8 * 4);
    ub->temb = (float*)take((size_t)time_rows * m->te * 4);
    ub->tvec = (float*)take((size_t)time_rows * tv_total * 4);
    if (train) {
        long w = 0;
        const long ps[3] = {(long)p1, (long)p2, (long)p3};
        for (int l = 0; l < 3; ++l) {
            const long f = bn_ws_floats(m->dtype, ps[l], m->ch[l]);
            if (f > w) w = f;
        }
        ub->bn_ws = (float*)take((size_t)(w > 0 ? w : 4) * 4);
    }
    ub->bytes = off;
    return off;
}

// time path: rows of sinusoidal embeddings -> per-block time vectors (src/main.rs:302-306, :377-378)
static int run_time_path(const ldm_module* m, const float* sinemb, int rows, UnetBufs& ub, hipStream_t st,
                         bool dense4 = false) {
    LDM_TRY(launch_linear_f32(sinemb, m->tw1, m->tb1, ub.h1, rows, m->te, m->te * 8, ACT_RELU, st));
    LDM_TRY(launch_linear_f32(ub.h1, m->tw2, m->tb2, ub.temb, rows, m->te * 8, m->te, ACT_NONE, st));
    if (dense4) return launch_linear_f32(ub.temb, m->d4.twt, m->d4.tbt, ub.tvec, rows, m->te, m->d4.tv_total, ACT_SILU, st);
    LDM_TRY(launch_linear_f32(ub.temb, m->twt, m->tbt, ub.tvec, rows, m->te, m->tv_total, ACT_SILU, st));
    return LDM_OK;
}

// Unet::forward body on NHWC buffers (src/main.rs:499-525; pools skipped, dropout = identity).
// xin [B,H,W,clp]; eps_out [B,H,W,out_conv.cout] f32.
// With `ddim` the last convolution applies the deterministic DDIM update to the sampler state instead
// of writing eps_out (only where tapconv_can_fuse_ddim allows it).
struct DdimFuse {
    float* z = nullptr;          // f32 state, updated in place
    void* zl = nullptr;          // bf16 copy of the new state
    const float* coef = nullptr; // [steps][8]
    int* ctl = nullptr;          // {step counter, seed low, seed high, chain0, ticket}
    // Chained steps (launch_ddim_down1): the update reads z and writes z_next (another buffer), and the same kernel runs
    // down1 of the NEXT step into the buffer this body's own down1 writes; a body with have_down1 finds it there.
    float* z_next = nullptr;
    bool have_down1 = false;
};

static int run_unet_body(const ldm_module* m, UnetBufs& ub, const void* xin, void* eps_out, int B, int H, int W,
                         const TimeVec& tv, hipStream_t st, const DdimFuse* ddim = nullptr, const TrainCtx* train = nullptr) {
    const int c1 = m->ch[0], c2 = m->ch[1], c3 = m->ch[2];
    const int H2 = H / 2, W2 = W / 2, H3 = H / 4, W3 = W / 4;
    // one ResidualBlock, in the mode of this forward; `drop` = this block's output goes through mid_drop
    auto run_resblock = [&](const ResBlockP& rp, const void* x, void* tmp, void* out, int b, int h, int w, const TimeVec& t,
                            hipStream_t s2, bool drop = false) {
        if (train) return run_resblock_train(rp, m->dtype, x, tmp, out, b, h, w, t, *train, drop ? train->keep_prob : 1.0f, s2);
        return ::run_resblock(rp, x, tmp, out, b, h, w, t, s2);
    };
    if (!(ddim && ddim->have_down1))
        LDM_TRY(run_conv(m->down1, xin, m->clp, nullptr, 0, ub.a1, B, H, W, ACT_NONE, nullptr, nullptr, 0, 0, st));
    LDM_TRY(run_resblock(m->rb[0], ub.a1, ub.b1, ub.s1, B, H, W, tv, st));                 // down_1 = s1
    LDM_TRY(run_conv(m->down2, ub.s1, c1, nullptr, 0, ub.a2, B, H, W, ACT_NONE, nullptr, nullptr, 0, 0, st));
    LDM_TRY(run_resblock(m->rb[1], ub.a2, ub.b2, ub.s2, B, H2, W2, tv, st));               // down_2 = s2
    LDM_TRY(run_conv(m->down3, ub.s2, c2, nullptr, 0, ub.a3, B, H2, W2, ACT_NONE, nullptr, nullptr, 0, 0, st));
    LDM_TRY(run_resblock(m->rb[2], ub.a3, ub.b3, ub.c3, B, H3, W3, tv, st));               // down_3 = c3
    LDM_TRY(run_resblock(m->rb[3], ub.c3, ub.b3, ub.a3, B, H3, W3, tv, st));               // mid_res1 -> a3
    LDM_TRY(run_attention(m->mid_atten, m->dtype, ub.a3, ub.qkv3, ub.att3, ub.b3, B, H3, W3, st));   // -> b3
    LDM_TRY(run_resblock(m->rb[4], ub.b3, ub.c3, ub.a3, B, H3, W3, tv, st, true));         // mid_res2 (+ mid_drop) -> a3
    if (m->have_up_fused) {
        // up1 + cat + up_conv1 as one transposed conv over {a3, skip s2} -> b2
        LDM_TRY(run_conv(m->up1_fused, ub.a3, c3, ub.s2, c2, ub.b2, B, H3, W3, ACT_NONE, nullptr, nullptr, 0, 0, st));
    } else {
        LDM_TRY(run_conv(m->up1, ub.a3, c3, nullptr, 0, ub.a2, B, H3, W3, ACT_NONE, nullptr, nullptr, 0, 0, st));
        LDM_TRY(run_conv(m->up_conv1, ub.a2, c2, ub.s2, c2, ub.b2, B, H2, W2, ACT_NONE, nullptr, nullptr, 0, 0, st));
    }
    LDM_TRY(run_resblock(m->rb[5], ub.b2, ub.a2, ub.s2, B, H2, W2, tv, st));               // up_res1 -> s2
    const void* u1 = ub.s2;
    if (m->scaled) {
        LDM_TRY(run_attention(m->up_atten, m->dtype, ub.s2, ub.qkv2, ub.att2, ub.b2, B, H2, W2, st));
        u1 = ub.b2;
    }
    if (m->have_up_fused && upfuse_shape_ok(m->dtype, c2, c1, c1, B, H2, W2)) {
        LDM_TRY(launch_upfuse(m->up2_fused, u1, ub.s1, ub.b1, B, H2, W2, st));      // up2 + cat + up_conv2 -> b1
    } else if (m->have_up_fused) {
        LDM_TRY(run_conv(m->up2_fused, u1, c2, ub.s1, c1, ub.b1, B, H2, W2, ACT_NONE, nullptr, nullptr, 0, 0, st));
    } else {
        LDM_TRY(run_conv(m->up2, u1, c2, nullptr, 0, ub.a1, B, H2, W2, ACT_NONE, nullptr, nullptr, 0, 0, st));
        LDM_TRY(run_conv(m->up_conv2, ub.a1, c1, ub.s1, c1, ub.b1, B, H, W, ACT_NONE, nullptr, nullptr, 0, 0, st));
    }
    LDM_TRY(run_resblock(m->rb[6], ub.b1, ub.a1, ub.s1, B, H, W, tv, st));                 // up_res2 -> s1
    if (ddim && ddim->z_next)
        return launch_ddim_down1(m->out_conv, m->down1, ub.s1, ddim->z, ddim->z_next, ub.a1, B, H, W, ddim->coef, ddim->ctl, st);
    if (ddim) {
        ConvCall cc;
        cc.in0 = ub.s1; cc.cin0 = c1; cc.out = ddim->zl; cc.B = B; cc.Hin = H; cc.Win = W;
        cc.step_ptr = ddim->ctl;
        cc.ddim_z = ddim->z; cc.ddim_coef = ddim->coef; cc.ddim_ticket = ddim->ctl + 4;
        return launch_tapconv(m->out_conv, cc, st);
    }
    LDM_TRY(run_conv(m->out_conv, ub.s1, c1, nullptr, 0, eps_out, B, H, W, ACT_NONE, nullptr, nullptr, 0, 1, st));
    return LDM_OK;
}

// Unet::forward for Bp = 16*R images of 4x4 latents through the matrices of Dense4: every buffer [Bp, P*C] is
// handed to the tap kernel as ONE image of R x 16 pixels with P*C channels (spatial layers) or of R*P x 16 pixels
// with C channels (per-pixel layers).  tv.pix_stride = 0: all images share the time vector; otherwise one row per image.
static int run_unet_dense4(const ldm_module* m, UnetBufs& ub, const void* xin, void* eps_out, int Bp, const TimeVec& tv,
                           hipStream_t st) {
    const Dense4& d = m->d4;
    const int c1 = m->ch[0], c2 = m->ch[1], c3 = m->ch[2];
    const int R = Bp / 16;
    auto mat = [&](const PackedConv& pc, const void* in, void* out, int act, const void* res, int tv_off, int out_f32) {
        return run_conv(pc, in, pc.cin, nullptr, 0, out, 1, R, 16, act, res, tv_off >= 0 ? &tv : nullptr,
                        tv_off >= 0 ? tv_off : 0, out_f32, st);
    };
    auto block = [&](int i, const void* x, void* tmp, void* out) {
        LDM_TRY(mat(d.rb[i][0], x, tmp, ACT_SILU, nullptr, d.tv_off[i], 0));
        return mat(d.rb[i][1], tmp, out, ACT_SILU, x, -1, 0);
    };
    auto atten = [&](const AttnP& ap, int P, const void* x, void* qkv, void* att, void* out) {
        LDM_TRY(run_conv(ap.qkv, x, ap.C, nullptr, 0, qkv, 1, R * P, 16, ACT_NONE, nullptr, nullptr, 0, 0, st));
        if (P == 1) {
            // one token: its soft-max weight is exp(0) / exp(0) = 1 exactly, the attention output is the value row
            // itself; the out projection reads it in place (channels 2C .. 3C of every q|k|v row)
            ConvCall cc;
            cc.in0 = (const char*)qkv + (size_t)2 * ap.C * es_of(m->dtype);
            cc.cin0 = ap.C; cc.in0_stride = 3 * ap.C;
            cc.out = out; cc.res = x; cc.B = 1; cc.Hin = R; cc.Win = 16;
            return launch_tapconv(ap.out, cc, st);
        }
        LDM_TRY(launch_attention(m->dtype, qkv, att, Bp, P, ap.C, st));
        return run_conv(ap.out, att, ap.C, nullptr, 0, out, 1, R * P, 16, ACT_NONE, x, nullptr, 0, 0, st);
    };
    LDM_TRY(mat(d.down1, xin, ub.a1, ACT_NONE, nullptr, -1, 0));
    LDM_TRY(block(0, ub.a1, ub.b1, ub.s1));
    LDM_TRY(mat(d.down2, ub.s1, ub.a2, ACT_NONE, nullptr, -1, 0));
    LDM_TRY(block(1, ub.a2, ub.b2, ub.s2));
    LDM_TRY(mat(d.down3, ub.s2, ub.a3, ACT_NONE, nullptr, -1, 0));
    LDM_TRY(block(2, ub.a3, ub.b3, ub.c3));
    LDM_TRY(block(3, ub.c3, ub.b3, ub.a3));
    // Up to this many images a launch costs more than its FLOPs: consecutive linear layers run composed (the block-
    // diagonal skip part of the composed up-sampling matrices multiplies 4 / 16 times the FLOPs of up_conv1/2).
    const bool composed = Bp <= 256;
    if (composed) LDM_TRY(mat(d.att1, ub.a3, ub.b3, ACT_NONE, ub.a3, -1, 0));
    else LDM_TRY(atten(m->mid_atten, 1, ub.a3, ub.qkv3, ub.att3, ub.b3));
    LDM_TRY(block(4, ub.b3, ub.c3, ub.a3));
    if (composed) {
        LDM_TRY(run_conv(d.up1c, ub.a3, c3, ub.s2, 4 * c2, ub.b2, 1, R, 16, ACT_NONE, nullptr, nullptr, 0, 0, st));
    } else {
        LDM_TRY(mat(d.up1, ub.a3, ub.a2, ACT_NONE, nullptr, -1, 0));
        LDM_TRY(run_conv(m->up_conv1, ub.a2, c2, ub.s2, c2, ub.b2, 1, R * 4, 16, ACT_NONE, nullptr, nullptr, 0, 0, st));
    }
    LDM_TRY(block(5, ub.b2, ub.a2, ub.s2));
    const void* u1 = ub.s2;
    if (m->scaled) {
        LDM_TRY(atten(m->up_atten, 4, ub.s2, ub.qkv2, ub.att2, ub.b2));
        u1 = ub.b2;
    }
    if (composed) {
        LDM_TRY(run_conv(d.up2c, u1, 4 * c2, ub.s1, 16 * c1, ub.b1, 1, R, 16, ACT_NONE, nullptr, nullptr, 0, 0, st));
    } else {
        LDM_TRY(mat(d.up2, u1, ub.a1, ACT_NONE, nullptr, -1, 0));
        LDM_TRY(run_conv(m->up_conv2, ub.a1, c1, ub.s1, c1, ub.b1, 1, R * 16, 16, ACT_NONE, nullptr, nullptr, 0, 0, st));
    }
    LDM_TRY(block(6, ub.b1, ub.a1, ub.s1));
    (void)c3;
    return mat(d.out_conv, ub.s1, eps_out, ACT_NONE, nullptr, -1, 1);
}

// 4x4 latents run as whole-image matrices unless LDM_DENSE4=0 asks for the tap kernels (comparison runs, tests)
static bool use_dense4(const ldm_module* m, int H, int W) {
    const char* e = getenv("LDM_DENSE4");
    return H == 4 && W == 4 && m->d4.ready && !(e && e[0] == '0');
}

static int ensure_ws(ldm_module* m, size_t bytes) {
    if (bytes <= m->ws_bytes) return LDM_OK;
    if (m->ws) {
        LDM_HIP(hipDeviceSynchronize());
        LDM_HIP(hipFree(m->ws));
        m->ws = nullptr;
        m->ws_bytes = 0;
    }
    LDM_HIP(hipMalloc(&m->ws, bytes));
    m->ws_bytes = bytes;
    return LDM_OK;
}

// ---------------------------------------------------------------------------------------------
// sampler
// ---------------------------------------------------------------------------------------------
struct ldm_sampler {
    ldm_module* unet = nullptr;
    int T = 0, B = 0, H = 0, W = 0;
    bool dense4 = false;      // 4x4 latents: the layers run as matrices over whole images (Dense4)
    int Bp = 0;               // images the buffers hold: B, or B rounded up to 16 for dense4 (the extra rows are never read out)
    DevPool pool;
    UnetBufs ub;
    float* z = nullptr;       // NHWC f32 state [B,H,W,clp]
    void* zl = nullptr;       // bf16 copy (bf16 mode)
    // Chained DDIM steps (bf16, the shapes of ddim_down1_shape_ok; LDM_DDIMCHAIN=0 at creation switches them off): step k
    // reads the state from zbuf[k & 1] and writes zbuf[(k + 1) & 1] in the kernel that also starts step k + 1; the last
    // step updates its buffer in place.  zbuf[0] = z.
    float* z2 = nullptr;
    bool chain_ok = false;
    // One plan per (mode, steps) a caller has asked for: its timestep and coefficient tables in device memory and
    // the graphs captured over them (without / with injected noise).  A plan is built on first use and kept, so a
    // caller that alternates samplers or step counts neither re-uploads nor re-captures anything.
    struct Plan {
        float* coef = nullptr;    // [steps][8]
        int* d_ts = nullptr;      // [steps]
        hipGraphExec_t exec[2] = {nullptr, nullptr};
    };
    std::map<std::pair<int, int>, Plan> plans;
    float* coef = nullptr;    // the tables of the plan of the current call
    int* d_ts = nullptr;
    int* step = nullptr;      // device control words: {step counter, seed low, seed high, chain0, ticket}
    float* noise = nullptr;  // NHWC f32 [steps][B,H,W,clp], allocated on first use
    size_t noise_bytes = 0;
    hipStream_t cap_stream = nullptr;
    void drop_graphs() {
        for (auto& kv : plans)
            for (hipGraphExec_t& e : kv.second.exec)
                if (e) { (void)hipGraphExecDestroy(e); e = nullptr; }
    }
    void drop_plans() {
        drop_graphs();
        for (auto& kv : plans) {
            if (kv.second.coef) (void)hipFree(kv.second.coef);
            if (kv.second.d_ts) (void)hipFree(kv.second.d_ts);
        }
        plans.clear();
    }
    ~ldm_sampler() {
        drop_plans();
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        if (noise) (void)hipFree(noise);
    }
};

static bool sampler_chains(const ldm_sampler* s, int mode, bool tracing) {
    return s->chain_ok && mode == LDM_DDIM && !tracing;
}

static int sampler_step(ldm_sampler* s, int mode, bool have_noise, unsigned long long seed, unsigned int chain0,
                        float* z_trace, float* eps_trace, int k, int steps, hipStream_t st) {
    ldm_module* m = s->unet;
    const int B = s->B, H = s->H, W = s->W;
    const size_t img = (size_t)B * m->in_ch * H * W;
    if (z_trace) LDM_TRY(launch_nhwc_to_nchw(s->z, LDM_F32, z_trace + (size_t)k * img, B, m->in_ch, m->clp, H, W, st));
    TimeVec tv;
    tv.table = s->ub.tvec; tv.img_stride = 0; tv.step_stride = m->tv_total; tv.step_ptr = s->step;
    const void* xin = m->dtype == LDM_BF16 ? s->zl : (const void*)s->z;
    if (s->dense4) {
        tv.step_stride = m->d4.tv_total;
        LDM_TRY(run_unet_dense4(m, s->ub, xin, s->ub.eps, s->Bp, tv, st));
    } else if (sampler_chains(s, mode, z_trace || eps_trace)) {
        // the scheduler update, the step advance and the next step's first convolution happen in one kernel
        float* zb[2] = {s->z, s->z2};
        DdimFuse f;
        f.z = zb[k & 1]; f.zl = s->zl; f.coef = s->coef; f.ctl = s->step;
        f.z_next = k + 1 < steps ? zb[(k + 1) & 1] : nullptr;
        f.have_down1 = k > 0;
        return run_unet_body(m, s->ub, xin, nullptr, B, H, W, tv, st, &f);
    } else if (mode == LDM_DDIM && !eps_trace && m->clp == 64 && tapconv_can_fuse_ddim(m->out_conv, B, H, W)) {
        // the scheduler update and the step advance happen in the epilogue of the last convolution
        DdimFuse f;
        f.z = s->z; f.zl = s->zl; f.coef = s->coef; f.ctl = s->step;
        return run_unet_body(m, s->ub, xin, nullptr, B, H, W, tv, st, &f);
    } else {
        LDM_TRY(run_unet_body(m, s->ub, xin, s->ub.eps, B, H, W, tv, st));
    }
    if (eps_trace)
        LDM_TRY(launch_nhwc_to_nchw(s->ub.eps, LDM_F32, eps_trace + (size_t)k * img, B, m->in_ch, m->out_conv.cout, H, W, st));
    SchedArgs a{};
    a.z = s->z; a.eps = s->ub.eps; a.eps_dtype = LDM_F32;
    a.noise = have_noise ? s->noise : nullptr;
    a.z_lowp = m->dtype == LDM_BF16 ? s->zl : nullptr;
    a.coef = s->coef; a.step_ptr = s->step;
    a.n = (long)B * H * W * m->clp;
    a.pix_per_chain = (long)H * W;
    a.c4 = m->in_ch / 4; a.cpad4 = m->clp / 4;
    a.seed = seed; a.chain0 = chain0;
    a.rng = (const unsigned int*)(s->step + 1);   // device copy, so a captured graph follows later seeds
    LDM_TRY(launch_sched_step(a, st));
    LDM_TRY(launch_step_advance(s->step, st));
    return LDM_OK;
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* ldm_last_error(void) { return g_err.c_str(); }
int ldm_version(void) { return 1; }

int ldm_get_betas(int num_time, float* out) {
    LDM_REQUIRE(num_time > 0 && out, "bad arguments");
    std::vector<float> b;
    host_betas(num_time, b);
    memcpy(out, b.data(), b.size() * 4);
    return LDM_OK;
}
int ldm_get_alphas(int num_time, float* out) {
    LDM_REQUIRE(num_time > 0 && out, "bad arguments");
    std::vector<float> a;
    host_alpha_bars(num_time, a);
    memcpy(out, a.data(), a.size() * 4);
    return LDM_OK;
}
int ldm_ddpm_coefficients(int num_time, float* out) {
    LDM_REQUIRE(num_time > 0 && out, "bad arguments");
    std::vector<float> t;
    host_ddpm_coef(num_time, t);
    memcpy(out, t.data(), t.size() * 4);
    return LDM_OK;
}
int ldm_ddim_timesteps(int num_time, int steps, int* out) {
    LDM_REQUIRE(num_time > 0 && out, "bad arguments");
    std::vector<int> ts;
    LDM_TRY(host_ddim_ts(num_time, steps, ts));
    memcpy(out, ts.data(), ts.size() * 4);
    return LDM_OK;
}
int ldm_ddim_coefficients(int num_time, int steps, float* out) {
    LDM_REQUIRE(num_time > 0 && out, "bad arguments");
    std::vector<float> t;
    LDM_TRY(host_ddim_coef(num_time, steps, t));
    memcpy(out, t.data(), t.size() * 4);
    return LDM_OK;
}

int ldm_add_noise_coefficients(int num_time, float* out) {
    LDM_REQUIRE(num_time > 0 && out, "bad arguments");
    std::vector<float> t;
    host_add_noise_coef(num_time, t);
    std::memcpy(out, t.data(), t.size() * 4);
    return LDM_OK;
}

int ldm_add_noise(const float* x, const float* noise, const int* t, const float* coef, int num_time, float* out, int B,
                  long per_sample, void* stream) {
    return launch_add_noise(x, noise, t, coef, num_time, out, B, per_sample, (hipStream_t)stream);
}

int ldm_mse(const float* a, const float* b, long n, float* ws1024, float* out, void* stream) {
    return launch_mse(a, b, n, ws1024, out, (hipStream_t)stream);
}

int ldm_mse_noalloc(const float* a, const float* b, long n, float* out, void* stream) {
    return launch_mse_noalloc(a, b, n, out, (hipStream_t)stream);
}

int ldm_mse_noalloc_prepare(void) { return mse_noalloc_prepare(); }

int ldm_latent_is_finite(const float* z, long n, int* flag, void* stream) {
    return launch_finite_check(z, n, flag, (hipStream_t)stream);
}

const char* ldm_last_kernel(void) { return tapconv_last_kernel(); }

int ldm_sin_time_addition(const int* time_dev, int batch, int dim, float* out_dev, void* stream) {
    return launch_sin_embed(time_dev, batch, dim, out_dev, (hipStream_t)stream);
}

int ldm_module_create(ldm_module** out, int kind, int dtype, int in_channels, int time_embed_dim, const int* channels3,
                      int scaled_attention) {
    LDM_REQUIRE(out, "null output");
    LDM_REQUIRE(kind >= LDM_MOD_UNET && kind <= LDM_MOD_TIME_ADDITION, "unknown module kind (the VAE decoder has its own constructor)");
    LDM_REQUIRE(dtype == LDM_F32 || dtype == LDM_BF16, "unknown dtype");
    LDM_REQUIRE(in_channels > 0, "channels must be positive");
    if (kind == LDM_MOD_UNET || kind == LDM_MOD_RESBLOCK)
        LDM_REQUIRE(time_embed_dim > 0 && time_embed_dim % 4 == 0, "time_embed_dim must be a positive multiple of 4");
    if (kind == LDM_MOD_UNET) LDM_REQUIRE(in_channels % 4 == 0, "latent channels must be a multiple of 4");
    int ndev = 0;
    LDM_HIP(hipGetDeviceCount(&ndev));
    LDM_REQUIRE(ndev > 0, "no HIP device");
    std::unique_ptr<ldm_module> m(new ldm_module());
    m->kind = kind; m->dtype = dtype; m->in_ch = in_channels; m->te = time_embed_dim;
    if (channels3) for (int i = 0; i < 3; ++i) m->ch[i] = channels3[i];
    m->scaled = scaled_attention != 0;
    *out = m.release();
    return LDM_OK;
}

int ldm_module_set_param(ldm_module* m, const char* name, const float* host_data, long numel) {
    LDM_REQUIRE(m && name && host_data && numel > 0, "bad arguments");
    LDM_REQUIRE(!m->finalized, "module already finalized");
    m->params[name] = std::vector<float>(host_data, host_data + numel);
    return LDM_OK;
}

int ldm_module_finalize(ldm_module* m) {
    LDM_REQUIRE(m, "null module");
    return finalize_module(m);
}

void ldm_module_destroy(ldm_module* m) { delete m; }

int ldm_unet_forward(ldm_module* m, const float* x, const float* time_emb, float* eps, int B, int H, int W, void* stream) {
    LDM_REQUIRE(m && m->kind == LDM_MOD_UNET && m->finalized, "needs a finalized UNet module");
    LDM_REQUIRE(x && time_emb && eps, "null tensor");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "H and W must be positive multiples of 4");
    hipStream_t st = (hipStream_t)stream;
    UnetBufs ub;
    if (use_dense4(m, H, W)) {
        // rows B .. Bp-1 of every buffer are padding: computed from whatever they hold, never read out
        const int Bp = round_up_i(B, 16);
        const size_t need = plan_unet_bufs(m, Bp, H, W, Bp, nullptr, &ub, m->d4.tv_total);
        LDM_TRY(ensure_ws(m, need));
        plan_unet_bufs(m, Bp, H, W, Bp, (char*)m->ws, &ub, m->d4.tv_total);
        LDM_TRY(run_time_path(m, time_emb, B, ub, st, true));
        LDM_TRY(launch_nchw_to_nhwc(x, ub.xin, m->dtype, B, m->in_ch, m->clp, H, W, st));
        TimeVec tv;
        tv.table = ub.tvec; tv.pix_stride = m->d4.tv_total;
        LDM_TRY(run_unet_dense4(m, ub, ub.xin, ub.eps, Bp, tv, st));
        return launch_nhwc_to_nchw(ub.eps, LDM_F32, eps, B, m->in_ch, m->out_conv.cout, H, W, st);
    }
    const size_t need = plan_unet_bufs(m, B, H, W, B, nullptr, &ub);
    LDM_TRY(ensure_ws(m, need));
    plan_unet_bufs(m, B, H, W, B, (char*)m->ws, &ub);
    LDM_TRY(run_time_path(m, time_emb, B, ub, st));
    LDM_TRY(launch_nchw_to_nhwc(x, ub.xin, m->dtype, B, m->in_ch, m->clp, H, W, st));
    TimeVec tv;
    tv.table = ub.tvec; tv.img_stride = m->tv_total;
    LDM_TRY(run_unet_body(m, ub, ub.xin, ub.eps, B, H, W, tv, st));
    LDM_TRY(launch_nhwc_to_nchw(ub.eps, LDM_F32, eps, B, m->in_ch, m->out_conv.cout, H, W, st));
    return LDM_OK;
}

// Unet::forward as DiffusionModel::forward runs it (src/main.rs:655 on the autodiff backend, :1318): BatchNorm on the
// statistics of the batch, mid_drop active.  Always the tiled kernels (the whole-image matrices of 4x4 latents have
// the running statistics folded in).
int ldm_unet_forward_train(ldm_module* m, const float* x, const float* time_emb, float* eps, int B, int H, int W,
                           float dropout_p, unsigned long long dropout_seed, float* bn_mean, float* bn_var, void* stream) {
    LDM_REQUIRE(m && m->kind == LDM_MOD_UNET && m->finalized, "needs a finalized UNet module");
    LDM_REQUIRE(x && time_emb && eps, "null tensor");
    LDM_REQUIRE(B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "H and W must be positive multiples of 4");
    LDM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout probability must be in [0, 1)");
    hipStream_t st = (hipStream_t)stream;
    LDM_TRY(ensure_train_pack(m));
    UnetBufs ub;
    const size_t need = plan_unet_bufs(m, B, H, W, B, nullptr, &ub, -1, true);
    LDM_TRY(ensure_ws(m, need));
    plan_unet_bufs(m, B, H, W, B, (char*)m->ws, &ub, -1, true);
    LDM_TRY(run_time_path(m, time_emb, B, ub, st));
    LDM_TRY(launch_nchw_to_nhwc(x, ub.xin, m->dtype, B, m->in_ch, m->clp, H, W, st));
    TimeVec tv;
    tv.table = ub.tvec; tv.img_stride = m->tv_total;
    TrainCtx tc;
    tc.bn_ws = ub.bn_ws; tc.bn_mean = bn_mean; tc.bn_var = bn_var;
    tc.keep_prob = 1.0f - dropout_p; tc.seed = dropout_seed;
    LDM_TRY(run_unet_body(m, ub, ub.xin, ub.eps, B, H, W, tv, st, nullptr, &tc));
    LDM_TRY(launch_nhwc_to_nchw(ub.eps, LDM_F32, eps, B, m->in_ch, m->out_conv.cout, H, W, st));
    return LDM_OK;
}

int ldm_resblock_forward_train(ldm_module* m, const float* x, const float* t_emb, float* out, int B, int H, int W,
                               float* bn_mean, float* bn_var, void* stream) {
    LDM_REQUIRE(m && m->kind == LDM_MOD_RESBLOCK && m->finalized, "needs a finalized ResidualBlock module");
    LDM_REQUIRE(x && t_emb && out && B > 0 && H > 0 && W > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    LDM_TRY(ensure_train_pack(m));
    const size_t es = es_of(m->dtype), C = m->in_ch;
    const size_t act = align_up((size_t)B * H * W * C * es, 256);
    const long wsf = bn_ws_floats(m->dtype, (long)B * H * W, (int)C);
    LDM_REQUIRE(wsf > 0, "channel count not served by the batch-statistics kernels");
    const size_t tvb = align_up((size_t)B * C * 4, 256);
    LDM_TRY(ensure_ws(m, 3 * act + tvb + (size_t)wsf * 4));
    char* base = (char*)m->ws;
    void *xin = base, *tmp = base + act, *o = base + 2 * act;
    float* tvec = (float*)(base + 3 * act);
    TrainCtx tc;
    tc.bn_ws = (float*)(base + 3 * act + tvb); tc.bn_mean = bn_mean; tc.bn_var = bn_var;
    LDM_TRY(launch_linear_f32(t_emb, m->twt, m->tbt, tvec, B, m->te, (int)C, ACT_SILU, st));
    LDM_TRY(launch_nchw_to_nhwc(x, xin, m->dtype, B, (int)C, (int)C, H, W, st));
    TimeVec tv;
    tv.table = tvec; tv.img_stride = (int)C;
    LDM_TRY(run_resblock_train(m->rb[0], m->dtype, xin, tmp, o, B, H, W, tv, tc, 1.0f, st));
    LDM_TRY(launch_nhwc_to_nchw(o, m->dtype, out, B, (int)C, (int)C, H, W, st));
    return LDM_OK;
}

long ldm_batchnorm_ws_floats(int dtype, long pixels, int channels) { return bn_ws_floats(dtype, pixels, channels); }

int ldm_batchnorm_act(int dtype, const void* h, const void* res, void* out, int B, long HW, int C, const float* gamma,
                      const float* beta, float eps, int act, const float* tvec, float dropout_p, unsigned long long seed,
                      float* ws, float* mean_out, float* var_out, void* stream) {
    LDM_REQUIRE(dtype == LDM_F32 || dtype == LDM_BF16, "unknown dtype");
    LDM_REQUIRE(act >= ACT_NONE && act <= ACT_RELU, "unknown activation");
    LDM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "dropout probability must be in [0, 1)");
    hipStream_t st = (hipStream_t)stream;
    LDM_TRY(launch_bn_stats(dtype, h, B, HW, C, gamma, beta, eps, ws, mean_out, var_out, st));
    if (!out) return LDM_OK;    // statistics only
    return launch_bn_apply(dtype, h, res, out, B, HW, C, ws, act, tvec, C, 1.0f - dropout_p, seed, st);
}

int ldm_resblock_forward(ldm_module* m, const float* x, const float* t_emb, float* out, int B, int H, int W, void* stream) {
    LDM_REQUIRE(m && m->kind == LDM_MOD_RESBLOCK && m->finalized, "needs a finalized ResidualBlock module");
    LDM_REQUIRE(x && t_emb && out && B > 0 && H > 0 && W > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t es = es_of(m->dtype), C = m->in_ch;
    const size_t act = align_up((size_t)B * H * W * C * es, 256);
    const size_t need = 3 * act + align_up((size_t)B * C * 4, 256);
    LDM_TRY(ensure_ws(m, need));
    char* base = (char*)m->ws;
    void *xin = base, *tmp = base + act, *o = base + 2 * act;
    float* tvec = (float*)(base + 3 * act);
    LDM_TRY(launch_linear_f32(t_emb, m->twt, m->tbt, tvec, B, m->te, (int)C, ACT_SILU, st));
    LDM_TRY(launch_nchw_to_nhwc(x, xin, m->dtype, B, (int)C, (int)C, H, W, st));
    TimeVec tv;
    tv.table = tvec; tv.img_stride = (int)C;
    LDM_TRY(run_resblock(m->rb[0], xin, tmp, o, B, H, W, tv, st));
    LDM_TRY(launch_nhwc_to_nchw(o, m->dtype, out, B, (int)C, (int)C, H, W, st));
    return LDM_OK;
}

int ldm_attention_forward(ldm_module* m, const float* x, float* out, int B, int H, int W, void* stream) {
    LDM_REQUIRE(m && m->kind == LDM_MOD_ATTENTION && m->finalized, "needs a finalized Attention module");
    LDM_REQUIRE(x && out && B > 0 && H > 0 && W > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t es = es_of(m->dtype), C = m->in_ch;
    const size_t act = align_up((size_t)B * H * W * C * es, 256);
    LDM_TRY(ensure_ws(m, 6 * act));
    char* base = (char*)m->ws;
    void *xin = base, *qkv = base + act, *att = base + 4 * act, *o = base + 5 * act;
    LDM_TRY(launch_nchw_to_nhwc(x, xin, m->dtype, B, (int)C, (int)C, H, W, st));
    LDM_TRY(run_attention(m->mid_atten, m->dtype, xin, qkv, att, o, B, H, W, st));
    LDM_TRY(launch_nhwc_to_nchw(o, m->dtype, out, B, (int)C, (int)C, H, W, st));
    return LDM_OK;
}

int ldm_time_addition_forward(ldm_module* m, const float* t, float* out, int B, void* stream) {
    LDM_REQUIRE(m && m->kind == LDM_MOD_TIME_ADDITION && m->finalized, "needs a finalized TimeAddition module");
    LDM_REQUIRE(t && out && B > 0, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int d = m->in_ch;
    LDM_TRY(ensure_ws(m, (size_t)B * d * 8 * 4));
    LDM_TRY(launch_linear_f32(t, m->tw1, m->tb1, (float*)m->ws, B, d, d * 8, ACT_RELU, st));
    LDM_TRY(launch_linear_f32((const float*)m->ws, m->tw2, m->tb2, out, B, d * 8, d, ACT_NONE, st));
    return LDM_OK;
}

int ldm_vae_decoder_create(ldm_module** out, int dtype, int output_channels, int latent_dimen) {
    LDM_REQUIRE(out, "null output");
    LDM_REQUIRE(dtype == LDM_F32 || dtype == LDM_BF16, "unknown dtype");
    LDM_REQUIRE(output_channels > 0 && output_channels <= 64, "image channels must be between 1 and 64");
    LDM_REQUIRE(latent_dimen > 0 && latent_dimen % 8 == 0 && latent_dimen <= 64, "latent_dimen must be a multiple of 8 up to 64");
    int ndev = 0;
    LDM_HIP(hipGetDeviceCount(&ndev));
    LDM_REQUIRE(ndev > 0, "no HIP device");
    std::unique_ptr<ldm_module> m(new ldm_module());
    m->kind = LDM_MOD_VAE_DECODER; m->dtype = dtype; m->in_ch = latent_dimen; m->te = output_channels;
    *out = m.release();
    return LDM_OK;
}

// SimpleDecoder::forward, src/main.rs:149-169 (Dropout is the identity at inference)
int ldm_vae_decode(ldm_module* m, const float* z, float* img, int B, void* stream) {
    LDM_REQUIRE(m && m->kind == LDM_MOD_VAE_DECODER && m->finalized, "needs a finalized VAE decoder module");
    LDM_REQUIRE(z && img && B > 0 && B <= 65535, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t es = es_of(m->dtype);
    const int K = m->in_ch * 16, N = 128 * 16 * 16;
    const size_t y0b = align_up((size_t)B * N * es, 256);                 // [B,16,16,128]
    const size_t y1b = align_up((size_t)B * 32 * 32 * 64 * es, 256);     // [B,32,32,64]
    const size_t y2b = align_up((size_t)B * 64 * 64 * 64 * es, 256);     // [B,64,64,64], channels >= 32 zero
    const size_t y3b = align_up((size_t)B * 64 * 64 * 64 * 4, 256);      // [B,64,64,64] f32, channels >= image channels zero
    LDM_TRY(ensure_ws(m, y0b + y1b + y2b + y3b));
    char* base = (char*)m->ws;
    void *y0 = base, *y1 = base + y0b, *y2 = base + y0b + y1b, *y3 = base + y0b + y1b + y2b;
    if (m->dtype == LDM_BF16) LDM_TRY(launch_linear_bf16(z, m->lin_w, m->lin_b, y0, B, K, N, ACT_RELU, st));
    else LDM_TRY(launch_linear_f32(z, (const float*)m->lin_w, m->lin_b, (float*)y0, B, K, N, ACT_RELU, st));
    LDM_TRY(run_conv(m->rev1, y0, 128, nullptr, 0, y1, B, 16, 16, ACT_RELU, nullptr, nullptr, 0, 0, st));
    LDM_TRY(run_conv(m->rev2, y1, 64, nullptr, 0, y2, B, 32, 32, ACT_RELU, nullptr, nullptr, 0, 0, st));
    if (m->rev3_w) {
        LDM_TRY(launch_conv3x3_fewout(y2, m->rev3_w, m->rev3_b, img, B, 64, 64, 64, m->te, ACT_TANH, st));
    } else {
        LDM_TRY(run_conv(m->rev3, y2, 64, nullptr, 0, y3, B, 64, 64, ACT_TANH, nullptr, nullptr, 0, 1, st));
        LDM_TRY(launch_nhwc_to_nchw(y3, LDM_F32, img, B, m->te, m->rev3.cout, 64, 64, st));
    }
    return LDM_OK;
}

int ldm_to_rgb8(const float* img, unsigned char* out, int B, int C, int H, int W, void* stream) {
    return launch_to_rgb8(img, out, B, C, H, W, (hipStream_t)stream);
}

int ldm_conv_create(ldm_conv** out, int kind, int dtype, int cin0, int cin1, int cout, const float* w, const float* b) {
    LDM_REQUIRE(out && w && b, "null argument");
    LDM_REQUIRE(kind >= CONV3x3_S1 && kind <= CONV1x1, "unknown conv kind");
    LDM_REQUIRE(dtype == LDM_F32 || dtype == LDM_BF16, "unknown dtype");
    LDM_REQUIRE(cin0 > 0 && cin0 % 64 == 0 && cin1 >= 0 && cin1 % 64 == 0 && cout > 0 && cout % 64 == 0,
                "channel counts must be multiples of 64");
    LDM_REQUIRE(cin1 == 0 || kind == CONV1x1, "only conv1x1 takes a second source");
    int ndev = 0;
    LDM_HIP(hipGetDeviceCount(&ndev));
    LDM_REQUIRE(ndev > 0, "no HIP device");
    std::unique_ptr<ldm_conv> c(new ldm_conv());
    c->cin0 = cin0; c->cin1 = cin1;
    const int cin = cin0 + cin1;
    PackSpec sp;
    sp.kind = kind; sp.dtype = dtype; sp.cout = cout;
    sp.src = {cin0};
    if (cin1) sp.src.push_back(cin1);
    if (kind == CONV3x3_S1 || kind == CONV3x3_S2)
        sp.w = [=](int, int tap, int co, int ci) { return w[((size_t)co * cin + ci) * 9 + tap]; };
    else if (kind == CONVT4x4_S2)
        sp.w = [=](int ph, int tap, int co, int ci) {
            const int ky = 3 - (ph >> 1) - 2 * (tap >> 1), kx = 3 - (ph & 1) - 2 * (tap & 1);
            return w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
        };
    else
        sp.w = [=](int, int, int co, int ci) { return w[(size_t)co * cin + ci]; };
    sp.bias = [=](int co) { return b[co]; };
    LDM_TRY(pack_conv(sp, c->pool, &c->pc, nullptr));
    *out = c.release();
    return LDM_OK;
}

int ldm_conv_forward(ldm_conv* c, const void* in0, const void* in1, const void* res, void* out, int B, int Hin, int Win,
                     int act, void* stream) {
    LDM_REQUIRE(c && in0 && out, "null argument");
    LDM_REQUIRE(act >= ACT_NONE && act <= ACT_RELU, "unknown activation");
    ConvCall cc;
    cc.in0 = in0; cc.cin0 = c->cin0; cc.in1 = in1; cc.cin1 = c->cin1;
    cc.res = res; cc.out = out; cc.B = B; cc.Hin = Hin; cc.Win = Win; cc.act = act;
    return launch_tapconv(c->pc, cc, (hipStream_t)stream);
}

void ldm_conv_destroy(ldm_conv* c) { delete c; }

int ldm_sched_step(float* z, const float* eps, const float* noise, long n, const float* coef5, int mode, int last,
                   void* stream) {
    LDM_REQUIRE(z && eps && coef5 && n > 0 && n % 4 == 0, "bad arguments (n must be a multiple of 4)");
    LDM_REQUIRE(mode == LDM_DDPM || mode == LDM_DDIM, "unknown mode");
    LDM_REQUIRE(noise || mode == LDM_DDIM || last, "a DDPM step before the last needs noise");
    SchedArgs a{};
    a.z = z; a.eps = eps; a.eps_dtype = LDM_F32; a.noise = noise;
    for (int i = 0; i < 5; ++i) a.cval[i] = coef5[i];
    a.cval[5] = (float)mode; a.cval[6] = last ? 1.f : 0.f;
    a.n = n; a.pix_per_chain = n / 4; a.c4 = 1; a.cpad4 = 1;
    return launch_sched_step(a, (hipStream_t)stream);
}

int ldm_philox_normal(float* out, int batch, int C, int H, int W, unsigned long long seed, unsigned int chain0,
                      unsigned int step, void* stream) {
    return launch_philox_nchw(out, batch, C, H, W, seed, chain0, step, (hipStream_t)stream);
}

int ldm_sampler_create(ldm_sampler** out, ldm_module* unet, int num_time, int batch, int H, int W) {
    LDM_REQUIRE(out, "null output");
    LDM_REQUIRE(unet && unet->kind == LDM_MOD_UNET && unet->finalized, "needs a finalized UNet module");
    LDM_REQUIRE(num_time > 0 && batch > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "H and W must be positive multiples of 4");
    std::unique_ptr<ldm_sampler> s(new ldm_sampler());
    s->unet = unet; s->T = num_time; s->B = batch; s->H = H; s->W = W;
    s->dense4 = use_dense4(unet, H, W);
    s->Bp = s->dense4 ? round_up_i(batch, 16) : batch;
    const int tvt = s->dense4 ? unet->d4.tv_total : unet->tv_total;
    UnetBufs tmp;
    const size_t need = plan_unet_bufs(unet, s->Bp, H, W, num_time, nullptr, &tmp, tvt);
    void* arena = nullptr;
    LDM_TRY(s->pool.alloc(&arena, need));
    LDM_HIP(hipMemset(arena, 0, need));
    plan_unet_bufs(unet, s->Bp, H, W, num_time, (char*)arena, &s->ub, tvt);
    const size_t nz = (size_t)s->Bp * H * W * unet->clp;
    LDM_TRY(s->pool.alloc((void**)&s->z, nz * 4));
    LDM_HIP(hipMemset(s->z, 0, nz * 4));
    if (unet->dtype == LDM_BF16) {
        LDM_TRY(s->pool.alloc(&s->zl, nz * 2));
        LDM_HIP(hipMemset(s->zl, 0, nz * 2));
        const char* e = getenv("LDM_DDIMCHAIN");
        s->chain_ok = !(e && e[0] == '0') && !s->dense4 && unet->clp == 64 && tapconv_can_fuse_ddim(unet->out_conv, batch, H, W) &&
                      ddim_down1_shape_ok(unet->out_conv, unet->down1, batch, H, W);
        if (s->chain_ok) {
            LDM_TRY(prepare_ddim_down1());
            LDM_TRY(s->pool.alloc((void**)&s->z2, nz * 4));
            LDM_HIP(hipMemset(s->z2, 0, nz * 4));
        }
    }
    LDM_TRY(s->pool.alloc((void**)&s->step, 32));
    LDM_HIP(hipMemset(s->step, 0, 32));
    LDM_HIP(hipMemset(s->ub.tvec, 0, (size_t)num_time * tvt * 4));
    LDM_HIP(hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking));
    // one eager forward on zeros: validates the shapes and sets every kernel's launch attributes
    // before anything is captured into a graph
    TimeVec tv;
    tv.table = s->ub.tvec; tv.step_stride = tvt; tv.step_ptr = s->step;
    const void* xin = unet->dtype == LDM_BF16 ? s->zl : (const void*)s->z;
    if (s->dense4) LDM_TRY(run_unet_dense4(unet, s->ub, xin, s->ub.eps, s->Bp, tv, nullptr));
    else LDM_TRY(run_unet_body(unet, s->ub, xin, s->ub.eps, batch, H, W, tv, nullptr));
    LDM_HIP(hipDeviceSynchronize());
    *out = s.release();
    return LDM_OK;
}

int ldm_sampler_run(ldm_sampler* s, int mode, int ddim_steps, const float* z0, const float* noise_dev,
                    unsigned long long seed, unsigned int chain0, float* z_out, float* z_trace, float* eps_trace,
                    int use_graph, void* stream) {
    LDM_REQUIRE(s && z_out, "bad arguments");
    LDM_REQUIRE(mode == LDM_DDPM || mode == LDM_DDIM, "unknown mode");
    ldm_module* m = s->unet;
    hipStream_t st = (hipStream_t)stream;
    const int B = s->B, H = s->H, W = s->W, T = s->T;
    const int steps = mode == LDM_DDPM ? T : ddim_steps;
    LDM_REQUIRE(steps >= 1 && steps <= T, "the number of steps must be in [1, num_time]");
    // The timestep and coefficient tables depend on (mode, steps) only.  A pair seen for the first time gets device
    // tables of its own (a blocking copy into fresh memory that no earlier work reads: nothing has to be waited for);
    // every later call with that pair finds them, and its graphs, in place.
    auto key = std::make_pair(mode, steps);
    auto pit = s->plans.find(key);
    if (pit == s->plans.end()) {
        if (s->plans.size() >= 32) {             // a caller cycling through many step counts: start over
            LDM_HIP(hipDeviceSynchronize());
            s->drop_plans();
        }
        std::vector<float> c5, h_coef;
        std::vector<int> h_ts;
        if (mode == LDM_DDPM) {
            host_ddpm_coef(T, c5);
            h_ts.resize(T);
            h_coef.assign((size_t)T * 8, 0.f);
            for (int k = 0; k < T; ++k) {
                const int i = T - 1 - k;
                h_ts[k] = i;
                for (int q = 0; q < 5; ++q) h_coef[(size_t)k * 8 + q] = c5[(size_t)i * 5 + q];
                h_coef[(size_t)k * 8 + 5] = 0.f;
                h_coef[(size_t)k * 8 + 6] = i == 0 ? 1.f : 0.f;
            }
        } else {
            LDM_TRY(host_ddim_ts(T, steps, h_ts));
            LDM_TRY(host_ddim_coef(T, steps, c5));
            h_coef.assign((size_t)steps * 8, 0.f);
            for (int k = 0; k < steps; ++k) {
                for (int q = 0; q < 5; ++q) h_coef[(size_t)k * 8 + q] = c5[(size_t)k * 5 + q];
                h_coef[(size_t)k * 8 + 5] = 1.f;
            }
        }
        ldm_sampler::Plan pl;
        LDM_HIP(hipMalloc((void**)&pl.coef, (size_t)steps * 8 * 4));
        hipError_t e = hipMalloc((void**)&pl.d_ts, (size_t)steps * 4);
        if (e == hipSuccess) e = hipMemcpy(pl.d_ts, h_ts.data(), (size_t)steps * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(pl.coef, h_coef.data(), (size_t)steps * 8 * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(pl.coef);
            if (pl.d_ts) (void)hipFree(pl.d_ts);
            LDM_HIP(e);
        }
        pit = s->plans.emplace(key, pl).first;
    }
    ldm_sampler::Plan& plan = pit->second;
    s->coef = plan.coef;
    s->d_ts = plan.d_ts;
    const bool have_noise = noise_dev != nullptr && mode == LDM_DDPM;
    const size_t nz = (size_t)B * H * W * m->clp;
    if (have_noise) {
        const size_t need = (size_t)steps * nz * 4;
        if (need > s->noise_bytes) {
            LDM_HIP(hipDeviceSynchronize());
            if (s->noise) LDM_HIP(hipFree(s->noise));
            s->noise = nullptr; s->noise_bytes = 0;
            LDM_HIP(hipMalloc((void**)&s->noise, need));
            s->noise_bytes = need;
            s->drop_graphs();                     // they hold the old buffer's address
        }
    }
    // time path for every visited timestep at once: [steps, tv_total] table (all chains share t)
    LDM_TRY(launch_sin_embed(s->d_ts, steps, m->te, s->ub.sinemb, st));
    LDM_TRY(run_time_path(m, s->ub.sinemb, steps, s->ub, st, s->dense4));
    if (z0) {
        LDM_TRY(launch_nchw_to_nhwc(z0, s->z, LDM_F32, B, m->in_ch, m->clp, H, W, st));
        if (m->dtype == LDM_BF16) LDM_TRY(launch_cast_f32_to_bf16(s->z, s->zl, (long)nz, st));
    } else {
        LDM_TRY(launch_philox_init(s->z, m->dtype == LDM_BF16 ? s->zl : nullptr, (long)nz, (long)H * W, m->in_ch / 4,
                                   m->clp / 4, seed, chain0, 0xFFFFFFFFu, st));
    }
    if (have_noise) {
        const int per = 32768 / B > 0 ? 32768 / B : 1;   // steps per launch (grid.z limit)
        const size_t img = (size_t)B * m->in_ch * H * W;
        for (int k0 = 0; k0 < steps; k0 += per) {
            const int kn = steps - k0 < per ? steps - k0 : per;
            LDM_TRY(launch_nchw_to_nhwc(noise_dev + (size_t)k0 * img, s->noise + (size_t)k0 * nz, LDM_F32, kn * B,
                                        m->in_ch, m->clp, H, W, st));
        }
    }
    LDM_TRY(launch_set_ctl(s->step, seed, chain0, st));
    const bool tracing = z_trace || eps_trace;
    if (use_graph && !tracing) {
        hipGraphExec_t& exec = plan.exec[have_noise ? 1 : 0];
        if (!exec) {
            hipGraph_t graph = nullptr;
            LDM_HIP(hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeRelaxed));
            int rc = LDM_OK;
            for (int k = 0; k < steps && rc == LDM_OK; ++k)
                rc = sampler_step(s, mode, have_noise, seed, chain0, nullptr, nullptr, k, steps, s->cap_stream);
            hipError_t e = hipStreamEndCapture(s->cap_stream, &graph);
            if (rc != LDM_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            LDM_HIP(e);
            e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) exec = nullptr;
            LDM_HIP(e);
        }
        LDM_HIP(hipGraphLaunch(exec, st));
    } else {
        for (int k = 0; k < steps; ++k) LDM_TRY(sampler_step(s, mode, have_noise, seed, chain0, z_trace, eps_trace, k, steps, st));
    }
    // (chained steps leave the final state in the buffer of the last step's parity)
    const float* z_final = sampler_chains(s, mode, tracing) && ((steps - 1) & 1) ? s->z2 : s->z;
    LDM_TRY(launch_nhwc_to_nchw(z_final, LDM_F32, z_out, B, m->in_ch, m->clp, H, W, st));
    return LDM_OK;
}

void ldm_sampler_destroy(ldm_sampler* s) { delete s; }

}  // extern "C"
