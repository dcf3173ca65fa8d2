// Shared declarations of the HIP library behind include/ldm_hip.h (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/ldm_hip.h"

namespace ldm {

typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// Last error text of the calling thread (returned by ldm_last_error()).
void set_error(const std::string& msg);

#define LDM_FAIL(code, msg)                                   \
    do {                                                      \
        ::ldm::set_error(std::string(__func__) + ": " + (msg)); \
        return (code);                                        \
    } while (0)

#define LDM_REQUIRE(cond, msg) \
    do {                       \
        if (!(cond)) LDM_FAIL(LDM_ERR_INVALID, msg); \
    } while (0)

#define LDM_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) LDM_FAIL(LDM_ERR_HIP, std::string(#call) + " -> " + hipGetErrorString(e_)); \
    } while (0)

#define LDM_TRY(call)          \
    do {                       \
        int r_ = (call);       \
        if (r_ != LDM_OK) return r_; \
    } while (0)

// Dynamic LDS above the default limit is an attribute of the kernel on ONE device: it is set the first time a
// kernel is launched on each device of the process.  `done` is the caller's static flag array for that kernel.
constexpr int kMaxDevices = 64;
inline int allow_dynamic_lds(const void* kernel, int bytes, bool (&done)[kMaxDevices]) {
    int dev = 0;
    LDM_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) LDM_FAIL(LDM_ERR_INVALID, "device index out of range");
    if (!done[dev]) {
        LDM_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done[dev] = true;
    }
    return LDM_OK;
}

// Compute units of the current device (cached per device: a process may drive different parts).
inline int device_cu_count(int* out) {
    static int cus[kMaxDevices] = {};        // 0 = not asked yet; racing first calls store the same value
    int dev = 0;
    LDM_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) LDM_FAIL(LDM_ERR_INVALID, "device index out of range");
    int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (n == 0) {
        LDM_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
    }
    *out = n;
    return LDM_OK;
}

// f32 -> bf16 round-to-nearest-even on the host (NaN kept quiet).
inline bf16_t host_f2bf(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
inline float host_bf2f(bf16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// ---------------------------------------------------------------------------------------------
// Tap convolution: one kernel family for conv3x3 (stride 1 and 2), the four sub-pixel phases of
// convT4x4/stride 2, and conv1x1 over up to two channel-concatenated sources.
// ---------------------------------------------------------------------------------------------
enum ConvKind { CONV3x3_S1 = 0, CONV3x3_S2 = 1, CONVT4x4_S2 = 2, CONV1x1 = 3 };
enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_RELU = 2, ACT_TANH = 3 };

// A convolution with its BatchNorm folded in, packed for the kernel (device memory).
struct PackedConv {
    int kind = 0;
    int dtype = 0;          // LDM_F32 / LDM_BF16: element type of weights and activations
    int cin = 0;            // padded total input channels (multiple of chunk)
    int cout = 0;           // padded output channels (multiple of 64)
    int ntaps = 0, nphase = 1;
    int ntaps1 = 0;         // taps of the second source's chunks: = ntaps, or 1 for a transposed conv whose second
                            // source is sampled at the output pixel (fused convT + concat + 1x1)
    void* w = nullptr;      // [phase][chunk][tap][pair plane 0..3][cout][2 pieces of 16 bytes]
    float* bias = nullptr;  // [cout]
    size_t w_bytes = 0;
};

struct ConvCall {
    const void* in0 = nullptr;   // NHWC [B,Hin,Win,cin0]
    const void* in1 = nullptr;   // optional second source (channels appended after in0's)
    int cin0 = 0, cin1 = 0;
    int in0_stride = 0;          // channels between consecutive pixels of in0 when it is a slice of a wider tensor (0: cin0)
    const void* res = nullptr;   // NHWC [B,Hout,Wout,cout], added before the activation
    const float* tvec = nullptr; // added after the activation: tvec[img*t_img_stride + pixel*t_pix_stride + step*t_step_stride + c]
    const int* step_ptr = nullptr;
    int t_img_stride = 0, t_step_stride = 0;
    int t_pix_stride = 0;        // further offset per pixel of the image (a batch handed over as one image of B pixels)
    void* out = nullptr;         // NHWC [B,Hout,Wout,cout]
    int out_f32 = 0;             // write f32 although dtype is bf16
    int act = ACT_NONE;
    int B = 0, Hin = 0, Win = 0; // Hout/Wout follow from the kind
    // Fused deterministic DDIM update (only where tapconv_can_fuse_ddim says so): the conv result is
    // the predicted noise; ddim_z [B,H,W,64] f32 is updated in place with row *step_ptr of ddim_coef
    // ([steps][8], see SchedArgs) and `out` receives the bf16 copy of the new state.
    // With ddim_ticket (a device word that is 0 between launches) the launch also advances *step_ptr
    // by one when its last workgroup finishes.
    float* ddim_z = nullptr;
    const float* ddim_coef = nullptr;
    int* ddim_ticket = nullptr;
};
bool tapconv_can_fuse_ddim(const PackedConv& pc, int B, int H, int W);

// Fused up-sampling block (convT 4x4/s2 128 -> 64, concat with a 64-channel skip tensor, 1x1 conv to 64):
// x [B,Hin,Win,128], skip and out [B,2Hin,2Win,64], bf16 NHWC.  `pc` holds the composed weights.
bool upfuse_shape_ok(int dtype, int cin, int cskip, int cout, int B, int Hin, int Win);
int launch_upfuse(const PackedConv& pc, const void* x, const void* skip, void* out, int B, int Hin, int Win,
                  hipStream_t stream);

// Fused ResidualBlock (conv3x3 + SiLU + time vector, conv3x3 + residual + SiLU; src/main.rs:372-387) of the 64-channel
// level: x and out [B,H,64,64] bf16 NHWC, c1 / c2 the block's packed convolutions (BatchNorm folded in),
// tvec[img * t_img_stride + step * t_step_stride + c] the block's time vector.
bool resfuse_shape_ok(int dtype, int C, int B, int H, int W);
int launch_resfuse(const PackedConv& c1, const PackedConv& c2, const void* x, void* out, int B, int H, int W,
                   const float* tvec, int t_img_stride, int t_step_stride, const int* step_ptr, hipStream_t stream);

// The end of one bf16 DDIM step and the start of the next as one kernel: out_conv on h, the DDIM update of the f32 state
// (z_in -> z_out, two different buffers) and down1 of the next step on the bf16 rounding of z_out (a1_out).
// coef [steps][8], ctl = the sampler's control words (step counter at 0, ticket at 4).
bool ddim_down1_shape_ok(const PackedConv& out_conv, const PackedConv& down1, int B, int H, int W);
int prepare_ddim_down1();      // sets the kernel's launch attributes on the current device (call outside stream capture)
int launch_ddim_down1(const PackedConv& out_conv, const PackedConv& down1, const void* h, const float* z_in, float* z_out,
                      void* a1_out, int B, int H, int W, const float* coef, int* ctl, hipStream_t stream);

int conv_out_hw(int kind, int Hin, int Win, int* Hout, int* Wout);
int launch_tapconv(const PackedConv& pc, const ConvCall& cc, hipStream_t stream);
// Name of the kernel the calling thread's last launch_tapconv / launch_upfuse started, spelled as a profiler lists it
// ("void ldm::tapconv_kernel<unsigned short, 16, 64, 4, 2, 1, 3, 9, 2, 3, 9>(ldm::TapConvArgs)"); "" before the first.
const char* tapconv_last_kernel();

// ---------------------------------------------------------------------------------------------
// Attention core: softmax(q k^T / sqrt(C)) v over the tokens of one image, single head.
// qkv NHWC-token layout [B, N, 3C] (q | k | v), out [B, N, C]; element type by dtype.
// ---------------------------------------------------------------------------------------------
int launch_attention(int dtype, const void* qkv, void* out, int B, int N, int C, hipStream_t stream);
// Attention with the projections folded at pack time (see attention_stream_kernel, AT_FOLD): qv = [B, N, 2C] bf16
// (q' | v'), x = the block's input [B, N, C] (keys and residual), bias = bv Wo + bo (f32 [C]).
bool attention_can_fold(int dtype, int B, int N, int C);
int launch_attention_folded(const void* qv, const void* x, const float* bias, void* out, int B, int N, int C, hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// Bandwidth-bound kernels.
// ---------------------------------------------------------------------------------------------
// NCHW f32 [B,C,H,W] -> NHWC [B,H,W,Cpad] (dtype elements, channels >= C zero filled)
int launch_nchw_to_nhwc(const float* in, void* out, int dtype, int B, int C, int Cpad, int H, int W, hipStream_t s);
// NHWC [B,H,W,Cpad] (f32 or bf16 by in_dtype) -> NCHW f32 [B,C,H,W]
int launch_nhwc_to_nchw(const void* in, int in_dtype, float* out, int B, int C, int Cpad, int H, int W, hipStream_t s);
// out[r, :] = act(in[r, :] @ W + b), W [K,N] f32 row-major; plain f32 FMA (time path, tiny)
int launch_linear_f32(const float* in, const float* W, const float* b, float* out, int rows, int K, int N, int act,
                      hipStream_t s);
// out[r, n] = act(sum_k in[r,k] W[k,n] + b[n]) for a wide Linear whose weights dominate the bytes (the VAE
// decoder's transfer layer, src/main.rs:111,151-152): in f32 [rows,K], out bf16 [rows,N], wp = the weights
// as bf16 MFMA fragments [N/16][K/32][64 lanes][8] (lane (n & 15, g): W[ks*32 + g*8 + e][nt*16 + (n & 15)]).
// K a multiple of 128 and <= 1024, N a multiple of 16.
int launch_linear_bf16(const float* in, const void* wp, const float* b, void* out, int rows, int K, int N, int act,
                       hipStream_t s);
// NCHW f32 [B,C,H,W] in [-1,1] -> HWC u8 [B,H,W,C]: ((v + 1) * 127.5) clamped to [0,255], truncated
// (save_as_image, src/main.rs:1297-1308)
int launch_to_rgb8(const float* in, unsigned char* out, int B, int C, int H, int W, hipStream_t s);
// conv3x3 / stride 1 / pad 1 from 32 real bf16 input channels (NHWC, Cpad channels per pixel) to CO <= 4 output
// channels with an activation, written NCHW f32 [B,CO,H,W]; wfrag [9][64][8] bf16: lane (n = lane & 15,
// g = lane >> 4) of tap t = dy*3 + dx holds W[t][g*8 + e][n], e = 0..7, zero for n >= CO
int launch_conv3x3_fewout(const void* in, const void* wfrag, const float* bias, float* out, int B, int H, int W, int Cpad,
                          int CO, int act, hipStream_t s);
// forward process (src/main.rs:584-630): out[b] = x[b] * coef[t_b][0] + noise[b] * coef[t_b][1], coef [num_time][2]
int launch_add_noise(const float* x, const float* noise, const int* t, const float* coef, int num_time, float* out, int B,
                     long per_sample, hipStream_t s);
// out[0] = mean((a - b)^2) over n elements; partial_ws: 1024 floats of device scratch
int launch_mse(const float* a, const float* b, long n, float* partial_ws, float* out, hipStream_t s);
// the same with the partial sums in an array of the code object (no workspace argument, no allocation)
int launch_mse_noalloc(const float* a, const float* b, long n, float* out, hipStream_t s);
// the per-device address of that array, resolved ahead of the first launch_mse_noalloc on the current device
int mse_noalloc_prepare();
// *flag = 1 if all n elements of z are finite, else 0 (two launches: a fill and the probe)
int launch_finite_check(const float* z, long n, int* flag, hipStream_t s);
// sinusoidal embedding of integer timesteps (src/main.rs:309-335)
int launch_sin_embed(const int* t, int rows, int dim, float* out, hipStream_t s);

// Training-mode BatchNorm on NHWC tensors (batch statistics; src/main.rs:340-341, 356, 366 on the autodiff backend).
struct BnPlan { int G = 0, rows = 0, R = 0, nblk = 0; long pix_per_blk = 0; };
int bn_plan(int dtype, long P, int C, BnPlan* pl);
// floats of workspace for a tensor of P pixels x C channels: the slices' partial statistics, then 4 floats per channel
long bn_ws_floats(int dtype, long P, int C);
// batch mean and biased variance per channel of h [B][HW][C] (two launches, fixed summation order) -> ws;
// mean_out / var_out [C] optional
int launch_bn_stats(int dtype, const void* h, int B, long HW, int C, const float* gamma, const float* beta, float eps,
                    float* ws, float* mean_out, float* var_out, hipStream_t s);
// out = dropout(act(((h - mean) / sqrt(var + eps)) * gamma + beta + res) + tvec[b]) with the statistics launch_bn_stats
// left in ws; out may be h
int launch_bn_apply(int dtype, const void* h, const void* res, void* out, int B, long HW, int C, const float* ws, int act,
                    const float* tvec, int t_img_stride, float keep_prob, unsigned long long seed, hipStream_t s);

struct SchedArgs {
    float* z;              // [n] f32 state, updated in place
    const void* eps;       // [n] f32 or bf16 (eps_dtype)
    int eps_dtype;
    const float* noise;    // [steps][n] f32 or null (then Philox when sigma != 0)
    void* z_lowp;          // optional bf16 copy of the new state (next UNet input) or null
    const float* coef;     // [steps][8] rows: sa, s1m, c1, c2, sigma, mode, last, 0; null: use cval
    float cval[8];         // the row by value, for a single call
    const int* step_ptr;   // device step counter (row of coef / noise); null: step 0
    long n;                // elements per call (all chains, padded channels included)
    long pix_per_chain;    // pixels per chain (H*W); the state is NHWC with cpad4*4 channels per pixel
    int c4, cpad4;         // real and padded channel count / 4; noise group = pixel * c4 + channel / 4
    unsigned long long seed;
    unsigned int chain0;   // global index of the first chain held by this process
    const unsigned int* rng;   // optional device words {seed low, seed high, chain0}: override the two above
};
// ctl[0] = 0 (step counter), ctl[1..3] = {seed low, seed high, chain0}, ctl[4] = 0 (workgroup ticket of the fused update)
int launch_set_ctl(int* ctl, unsigned long long seed, unsigned int chain0, hipStream_t s);
int launch_sched_step(const SchedArgs& a, hipStream_t s);
int launch_step_advance(int* step_ptr, hipStream_t s);
int launch_set_int(int* p, int v, hipStream_t s);
// z[n] <- counter-based normals of `step` (0xFFFFFFFF = initial latent), NHWC chains
int launch_philox_init(float* z, void* z_lowp, long n, long pix_per_chain, int c4, int cpad4, unsigned long long seed,
                       unsigned int chain0, unsigned int step, hipStream_t s);
// the same field written NCHW f32 [B,C,H,W]
int launch_philox_nchw(float* out, int B, int C, int H, int W, unsigned long long seed, unsigned int chain0,
                       unsigned int step, hipStream_t s);
int launch_cast_f32_to_bf16(const float* in, void* out, long n, hipStream_t s);

}  // namespace ldm
