// Bandwidth-bound kernels of the sampling loop (gfx950): the fused DDPM/DDIM scheduler update
// (src/main.rs:712-756,765), layout changes at the loop boundary, the sinusoidal timestep embedding
// (src/main.rs:309-335), the tiny f32 Linears of the time path (src/main.rs:302-306,377-378) and a
// counter-based normal generator.  All of them stream 16 bytes per lane.
#include <mutex>

#include "common.h"

namespace ldm {

__device__ __forceinline__ bf16_t ew_f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float ew_bf2f(bf16_t h) { return __builtin_bit_cast(float, (unsigned int)h << 16); }
// bf16 vectors for the MFMA kernels of this file; lin_pack2 rounds like one v_cvt_pk_bf16_f32
typedef __bf16 lin_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 lin_bf16x2 __attribute__((ext_vector_type(2)));
typedef float lin_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int lin_pack2(float a, float b) {
    const lin_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f, lin_bf16x2));
}

// ---------------------------------------------------------------------------------------------
// Philox-4x32-10 and Box-Muller: four standard normals per counter.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                              unsigned int k0, unsigned int k1, unsigned int out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)c0 * 0xD2511F53ull;
        const unsigned long long p1 = (unsigned long long)c2 * 0xCD9E8D57ull;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        c1 = (unsigned int)p1;
        c3 = (unsigned int)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ f32x4 philox_normal4(unsigned int group, unsigned int step, unsigned int chain,
                                                unsigned long long seed) {
    unsigned int r[4];
    philox4x32_10(group, step, chain, 0x4C444D31u, (unsigned int)seed, (unsigned int)(seed >> 32), r);
    const float s = 5.9604644775390625e-08f;   // 2^-24
    const float u0 = ((float)(r[0] >> 8) + 0.5f) * s, u1 = ((float)(r[1] >> 8) + 0.5f) * s;
    const float u2 = ((float)(r[2] >> 8) + 0.5f) * s, u3 = ((float)(r[3] >> 8) + 0.5f) * s;
    const float two_pi = 6.283185307179586f;
    const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
    f32x4 n;
    n[0] = ra * cosf(two_pi * u1);
    n[1] = ra * sinf(two_pi * u1);
    n[2] = rb * cosf(two_pi * u3);
    n[3] = rb * sinf(two_pi * u3);
    return n;
}

// ---------------------------------------------------------------------------------------------
// Scheduler step.  Per element, f32, in the reference's operation order:
//   x0 = (z - eps * s1m) / sa
//   mode 0 (DDPM, src/main.rs:752-756): z' = (x0 * c1 + z * c2) + sigma * n     (last step: z' = x0)
//   mode 1 (DDIM, eta = 0):             z' = x0 * c1 + eps * c2
// coef row = {sa, s1m, c1, c2, sigma, mode, last, 0}.  Contraction to FMA is disabled so that the
// result is the same sequence of f32 roundings as the CPU statement.
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void sched_step_kernel(const SchedArgs a) {
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const float* cf = a.coef ? a.coef + (size_t)step * 8 : a.cval;
    const float sa = cf[0], s1m = cf[1], c1 = cf[2], c2 = cf[3], sigma = cf[4];
    const int mode = (int)cf[5], last = (int)cf[6];
    const long n4 = a.n >> 2;
    const long cg = a.pix_per_chain * a.cpad4;   // padded groups of 4 per chain
    const float* noise = a.noise ? a.noise + (size_t)step * a.n : nullptr;
    unsigned long long seed = a.seed;
    unsigned int chain0 = a.chain0;
    if (a.rng) {
        seed = (unsigned long long)a.rng[0] | ((unsigned long long)a.rng[1] << 32);
        chain0 = a.rng[2];
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 z = *(const f32x4*)(a.z + i * 4);
        f32x4 e;
        if (a.eps_dtype == LDM_BF16) {
            const uint2 r = *(const uint2*)((const char*)a.eps + i * 8);
            e[0] = ew_bf2f((bf16_t)(r.x & 0xffffu)); e[1] = ew_bf2f((bf16_t)(r.x >> 16));
            e[2] = ew_bf2f((bf16_t)(r.y & 0xffffu)); e[3] = ew_bf2f((bf16_t)(r.y >> 16));
        } else {
            e = *(const f32x4*)((const float*)a.eps + i * 4);
        }
        f32x4 nz = {0.f, 0.f, 0.f, 0.f};
        if (mode == 0 && !last) {
            if (noise) {
                nz = *(const f32x4*)(noise + i * 4);
            } else {
                const long chain = i / cg;
                const long rem = i - chain * cg;
                const long pix = rem / a.cpad4;
                const int cq = (int)(rem - pix * a.cpad4);
                if (cq < a.c4)
                    nz = philox_normal4((unsigned int)(pix * a.c4 + cq), (unsigned int)step,
                                        chain0 + (unsigned int)chain, seed);
            }
        }
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = (z[q] - e[q] * s1m) / sa;
            float r;
            if (last) r = x0;
            else if (mode == 0) r = (x0 * c1 + z[q] * c2) + sigma * nz[q];
            else r = x0 * c1 + e[q] * c2;
            o[q] = r;
        }
        *(f32x4*)(a.z + i * 4) = o;
        if (a.z_lowp) {
            uint2 w;
            w.x = (unsigned int)ew_f2bf(o[0]) | ((unsigned int)ew_f2bf(o[1]) << 16);
            w.y = (unsigned int)ew_f2bf(o[2]) | ((unsigned int)ew_f2bf(o[3]) << 16);
            *(uint2*)((char*)a.z_lowp + i * 8) = w;
        }
    }
}

// Forward process q(x_t | x_0), DiffusionModel::add_noise (src/main.rs:584-630):
//   noisy = x * sqrt_alpha[t_b] + noise * sqrt_one_minus_alpha[t_b]
// as two products and a sum (three tensor operations in the reference: no FMA).  coef = [num_time][2];
// a timestep outside [0, num_time) takes the last row (`(t as usize).min(len - 1)`, :606).
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                        const int* __restrict__ t, const float* __restrict__ coef, int num_time,
                                                        float* __restrict__ out, long per_sample4) {
    const int b = blockIdx.y;
    int ti = t[b];
    if (ti < 0 || ti > num_time - 1) ti = num_time - 1;
    const float sa = coef[2 * ti], s1m = coef[2 * ti + 1];
    const f32x4* xs = (const f32x4*)x + (size_t)b * per_sample4;
    const f32x4* ns = (const f32x4*)noise + (size_t)b * per_sample4;
    f32x4* os = (f32x4*)out + (size_t)b * per_sample4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 xv = xs[i], nv = ns[i];
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float p0 = xv[q] * sa;
            const float p1 = nv[q] * s1m;
            o[q] = p0 + p1;
        }
        os[i] = o;
    }
}
#pragma clang fp contract(on)

// Mean squared difference (the diffusion loss, src/main.rs:665-671): every workgroup sums a fixed, strided
// part of the elements in a fixed order (lane partials, then a tree over the workgroup), a second launch adds
// the workgroup sums in order and divides: the result does not depend on scheduling.
constexpr int MSE_BLOCKS = 1024;
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                          float* __restrict__ partial) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)MSE_BLOCKS * 256) {
        const float d = a[i] - b[i];
        acc += d * d;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void mse_final_kernel(const float* __restrict__ partial, long n, float* __restrict__ out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < MSE_BLOCKS; i += 256) acc += partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)n;
}

// Finiteness probe of a latent (the checks of src/main.rs:780-784, 788-792 on the mean: a mean is NaN or infinite
// as soon as one element is).  *flag is preset to 1 by a fill kernel; any wave that sees an element with an all-ones
// exponent stores 0 (every writer stores the same value).
// `head` elements in front of the first 16-byte boundary (a view that starts inside a larger tensor) and the tail
// behind the last whole vector are read one by one.
__global__ __launch_bounds__(256) void finite_check_kernel(const float* __restrict__ z, long n, int head,
                                                           int* __restrict__ flag) {
    bool bad = false;
    const long n4 = (n - head) / 4;
    const f32x4* z4 = (const f32x4*)(z + head);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = z4[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) bad |= (__builtin_bit_cast(unsigned int, (float)v[q]) & 0x7f800000u) == 0x7f800000u;
    }
    for (long i = head + n4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        bad |= (__builtin_bit_cast(unsigned int, z[i]) & 0x7f800000u) == 0x7f800000u;
    if (blockIdx.x == 0 && (int)threadIdx.x < head)
        bad |= (__builtin_bit_cast(unsigned int, z[threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
    if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 0;
}

static inline int ew_grid(long work_items) {
    long g = (work_items + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

int launch_sched_step(const SchedArgs& a, hipStream_t s) {
    LDM_REQUIRE(a.z && a.eps, "null pointer");
    LDM_REQUIRE(a.n > 0 && a.pix_per_chain > 0 && a.c4 > 0 && a.cpad4 >= a.c4 &&
                    a.n % (a.pix_per_chain * a.cpad4 * 4) == 0,
                "element count must be a whole number of chains");
    hipLaunchKernelGGL(sched_step_kernel, dim3(ew_grid(a.n / 4)), dim3(256), 0, s, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

__global__ void step_advance_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = *p + 1; }
__global__ void set_int_kernel(int* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }

int launch_step_advance(int* p, hipStream_t s) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, s, p);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}
__global__ void set_ctl_kernel(int* ctl, unsigned int s0, unsigned int s1, unsigned int chain0) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ctl[0] = 0;
        ctl[1] = (int)s0;
        ctl[2] = (int)s1;
        ctl[3] = (int)chain0;
        ctl[4] = 0;
    }
}
int launch_set_ctl(int* ctl, unsigned long long seed, unsigned int chain0, hipStream_t s) {
    hipLaunchKernelGGL(set_ctl_kernel, dim3(1), dim3(64), 0, s, ctl, (unsigned int)seed, (unsigned int)(seed >> 32), chain0);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}
int launch_set_int(int* p, int v, hipStream_t s) {
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(64), 0, s, p, v);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

__global__ __launch_bounds__(256) void philox_init_kernel(float* z, void* z_lowp, long n4, long pix_per_chain, int c4,
                                                          int cpad4, unsigned long long seed, unsigned int chain0,
                                                          unsigned int step) {
    const long cg = pix_per_chain * cpad4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const long chain = i / cg;
        const long rem = i - chain * cg;
        const long pix = rem / cpad4;
        const int cq = (int)(rem - pix * cpad4);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (cq < c4) o = philox_normal4((unsigned int)(pix * c4 + cq), step, chain0 + (unsigned int)chain, seed);
        *(f32x4*)(z + i * 4) = o;
        if (z_lowp) {
            uint2 w;
            w.x = (unsigned int)ew_f2bf(o[0]) | ((unsigned int)ew_f2bf(o[1]) << 16);
            w.y = (unsigned int)ew_f2bf(o[2]) | ((unsigned int)ew_f2bf(o[3]) << 16);
            *(uint2*)((char*)z_lowp + i * 8) = w;
        }
    }
}

int launch_philox_init(float* z, void* z_lowp, long n, long pix_per_chain, int c4, int cpad4, unsigned long long seed,
                       unsigned int chain0, unsigned int step, hipStream_t s) {
    LDM_REQUIRE(z && n > 0 && pix_per_chain > 0 && c4 > 0 && cpad4 >= c4 && n % (pix_per_chain * cpad4 * 4) == 0,
                "bad arguments");
    hipLaunchKernelGGL(philox_init_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, s, z, z_lowp, n / 4, pix_per_chain, c4,
                       cpad4, seed, chain0, step);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// One thread per group of 4 channels of one pixel; writes the four NCHW elements.
__global__ __launch_bounds__(256) void philox_nchw_kernel(float* out, long ngroups, int c4, long HW,
                                                          unsigned long long seed, unsigned int chain0,
                                                          unsigned int step) {
    const long per_chain = HW * c4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ngroups; i += (long)gridDim.x * blockDim.x) {
        const long chain = i / per_chain;
        const long grp = i - chain * per_chain;
        const long pix = grp / c4;
        const int cq = (int)(grp - pix * c4);
        const f32x4 o = philox_normal4((unsigned int)grp, step, chain0 + (unsigned int)chain, seed);
        float* dst = out + ((size_t)chain * c4 * 4 + (size_t)cq * 4) * HW + pix;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[(size_t)q * HW] = o[q];
    }
}

int launch_philox_nchw(float* out, int B, int C, int H, int W, unsigned long long seed, unsigned int chain0,
                       unsigned int step, hipStream_t s) {
    LDM_REQUIRE(out && B > 0 && C > 0 && C % 4 == 0 && H > 0 && W > 0, "bad arguments (C must be a multiple of 4)");
    const long ng = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(philox_nchw_kernel, dim3(ew_grid(ng)), dim3(256), 0, s, out, ng, C / 4, (long)H * W, seed, chain0,
                       step);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* in, char* out, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(in + i * 4);
        uint2 w;
        w.x = (unsigned int)ew_f2bf(v[0]) | ((unsigned int)ew_f2bf(v[1]) << 16);
        w.y = (unsigned int)ew_f2bf(v[2]) | ((unsigned int)ew_f2bf(v[3]) << 16);
        *(uint2*)(out + i * 8) = w;
    }
}

int launch_cast_f32_to_bf16(const float* in, void* out, long n, hipStream_t s) {
    LDM_REQUIRE(in && out && n > 0 && n % 4 == 0, "bad arguments");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, s, in, (char*)out, n / 4);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// ---------------------------------------------------------------------------------------------
// NCHW f32 <-> NHWC (f32 / bf16) through a 32(channel) x 64(pixel) LDS tile: reads and writes are
// both contiguous along the fast axis of their tensor.
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, TO* __restrict__ out, int C,
                                                           int Cpad, long HW) {
    __shared__ float tile[32][65];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const float* src = in + (size_t)b * C * HW;
    TO* dst = out + (size_t)b * HW * Cpad;
    for (int k = threadIdx.x; k < 32 * 64; k += 256) {
        const int c = k >> 6, p = k & 63;
        float v = 0.f;
        if (c0 + c < C && p0 + p < HW) v = src[(size_t)(c0 + c) * HW + p0 + p];
        tile[c][p] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 32 * 64; k += 256) {
        const int p = k >> 5, c = k & 31;
        if (c0 + c < Cpad && p0 + p < HW) {
            const float v = tile[c][p];
            if constexpr (sizeof(TO) == 2) dst[(size_t)(p0 + p) * Cpad + c0 + c] = ew_f2bf(v);
            else dst[(size_t)(p0 + p) * Cpad + c0 + c] = v;
        }
    }
}

template <typename TI>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const TI* __restrict__ in, float* __restrict__ out, int C,
                                                           int Cpad, long HW) {
    __shared__ float tile[32][65];
    const int b = blockIdx.z;
    const long p0 = (long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 32;
    const TI* src = in + (size_t)b * HW * Cpad;
    float* dst = out + (size_t)b * C * HW;
    for (int k = threadIdx.x; k < 32 * 64; k += 256) {
        const int p = k >> 5, c = k & 31;
        float v = 0.f;
        if (c0 + c < Cpad && p0 + p < HW) {
            if constexpr (sizeof(TI) == 2) v = ew_bf2f(src[(size_t)(p0 + p) * Cpad + c0 + c]);
            else v = src[(size_t)(p0 + p) * Cpad + c0 + c];
        }
        tile[c][p] = v;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 32 * 64; k += 256) {
        const int c = k >> 6, p = k & 63;
        if (c0 + c < C && p0 + p < HW) dst[(size_t)(c0 + c) * HW + p0 + p] = tile[c][p];
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 convolution (stride 1, zero padding 1) from 32 bf16 channels to at most 4 output channels with an
// activation, written NCHW f32: the last layer of the VAE decoder (32 -> 3 channels; src/main.rs:135-142,
// 167-168 as a flipped-kernel conv).  On the tap-conv kernel this layer computes 64 padded output channels
// for 3 real ones, writes them as f32 and needs a layout conversion afterwards.  Here D^T[16 x 16 pixels] +=
// W[16 x 32] X[32 x 16] is one v_mfma_f32_16x16x32_bf16 per tap: the weights (rows >= CO zero) are nine
// A fragments in registers, a workgroup stages the 18 x 18 window of a 16 x 16 tile (64 bytes per pixel) in
// LDS, a wave owns four rows, and the lanes that hold accumulator rows 0..3 store the real channels.
// in: NHWC bf16 [B,H,W,Cpad], wfrag: [9][64 lanes][8] bf16, out [B,CO,H,W] f32.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv3x3_fewout_kernel(const char* __restrict__ in, const char* __restrict__ wfrag,
                                                             const float* __restrict__ bias, float* __restrict__ out, int H,
                                                             int W, int Cpad, int CO, int act) {
    __shared__ __attribute__((aligned(16))) char win[18 * 18 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, l15 = lane & 15;
    const int tiles_x = (W + 15) / 16;
    const int ty0 = (blockIdx.x / tiles_x) * 16, tx0 = (blockIdx.x % tiles_x) * 16;
    const int b = blockIdx.y;
    const char* src = in + (size_t)b * H * W * Cpad * 2;
    u32x4 wreg[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wreg[t] = *(const u32x4*)(wfrag + (t * 64 + lane) * 16);
    for (int k = tid; k < 18 * 18 * 4; k += 256) {
        const int p = k >> 2, piece = k & 3;
        const int y = ty0 - 1 + p / 18, x = tx0 - 1 + p % 18;
        u32x4 v = {0u, 0u, 0u, 0u};
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
            v = *(const u32x4*)(src + ((size_t)y * W + x) * Cpad * 2 + piece * 16);
        *(u32x4*)(win + p * 64 + piece * 16) = v;
    }
    __syncthreads();
    f32x4 binit = {0.f, 0.f, 0.f, 0.f};
    if (g == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) binit[r] = r < CO ? bias[r] : 0.f;
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int row = wave * 4 + rr;
        f32x4 acc = binit;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const u32x4 xf = *(const u32x4*)(win + ((row + t / 3) * 18 + l15 + t % 3) * 64 + g * 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lin_bf16x8, wreg[t]), __builtin_bit_cast(lin_bf16x8, xf),
                                                          acc, 0, 0, 0);
        }
        // lane (l15, g) holds output channels g*4 .. g*4+3 of pixel (row, l15): the real ones sit in g == 0
        const int y = ty0 + row, x = tx0 + l15;
        if (g == 0 && y < H && x < W) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r < CO) {
                    float v = acc[r];
                    if (act == ACT_TANH) v = tanhf(v);
                    else if (act == ACT_RELU) v = fmaxf(v, 0.f);
                    out[(((size_t)b * CO + r) * H + y) * W + x] = v;
                }
            }
        }
    }
}

int launch_conv3x3_fewout(const void* in, const void* wfrag, const float* bias, float* out, int B, int H, int W, int Cpad,
                          int CO, int act, hipStream_t s) {
    LDM_REQUIRE(in && wfrag && bias && out && B > 0 && B <= 65535 && H > 0 && W > 0, "bad arguments");
    LDM_REQUIRE(CO >= 1 && CO <= 4 && Cpad >= 32 && Cpad % 8 == 0, "conv3x3_fewout serves 32 input channels and 1 to 4 output channels");
    dim3 grid((unsigned)(((W + 15) / 16) * ((H + 15) / 16)), (unsigned)B);
    hipLaunchKernelGGL(conv3x3_fewout_kernel, grid, dim3(256), 0, s, (const char*)in, (const char*)wfrag, bias, out, H, W, Cpad,
                       CO, act);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

int launch_nchw_to_nhwc(const float* in, void* out, int dtype, int B, int C, int Cpad, int H, int W, hipStream_t s) {
    LDM_REQUIRE(in && out && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, "bad arguments");
    LDM_REQUIRE(B <= 65535, "batch too large for one launch");
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((Cpad + 31) / 32), (unsigned)B);
    if (dtype == LDM_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, s, in, (bf16_t*)out, C, Cpad, HW);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, s, in, (float*)out, C, Cpad, HW);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

int launch_nhwc_to_nchw(const void* in, int in_dtype, float* out, int B, int C, int Cpad, int H, int W, hipStream_t s) {
    LDM_REQUIRE(in && out && B > 0 && C > 0 && Cpad >= C && H > 0 && W > 0, "bad arguments");
    LDM_REQUIRE(B <= 65535, "batch too large for one launch");
    const long HW = (long)H * W;
    dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 31) / 32), (unsigned)B);
    if (in_dtype == LDM_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, out, C, Cpad, HW);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, (const float*)in, out, C, Cpad, HW);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// ---------------------------------------------------------------------------------------------
// Time path (tiny): out[r, n] = act(sum_k in[r,k] * W[k,n] + b[n]).  One thread per output column,
// 8 rows per block kept in LDS; W rows are read coalesced.  f32 FMA chain in k order.
// ---------------------------------------------------------------------------------------------
// A workgroup owns 4 rows x 16 output columns and splits K over 16 thread slices (slice s takes
// k = s, s+16, ...), so that even the [50, 2048] x [2048, 256] Linear of the time path fills >100
// workgroups with short dependent chains; the slices are summed through LDS in a fixed order.
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                                         const float* __restrict__ b, float* __restrict__ out, int rows,
                                                         int K, int N, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* xs = (float*)smem_raw;                 // [4][K]
    float* red = xs + 4 * K;                      // [16 slices][4 rows][16 columns]
    const int r0 = blockIdx.y * 4;
    const int nr = min(4, rows - r0);
    for (int k = threadIdx.x; k < 4 * K; k += 256) xs[k] = k < nr * K ? in[(size_t)r0 * K + k] : 0.f;
    __syncthreads();
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int n = blockIdx.x * 16 + c;
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = 0.f;
    if (n < N) {
#pragma unroll 4
        for (int k = sl; k < K; k += 16) {
            const float w = W[(size_t)k * N + n];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(xs[r * K + k], w, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(sl * 4 + r) * 16 + c] = acc[r];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int r = threadIdx.x >> 4;
        float v = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) v += red[(s2 * 4 + r) * 16 + c];
        if (n < N && r < nr) {
            v += b ? b[n] : 0.f;
            if (act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == ACT_SILU) v = v / (1.0f + expf(-v));
            out[(size_t)(r0 + r) * N + n] = v;
        }
    }
}

int launch_linear_f32(const float* in, const float* W, const float* b, float* out, int rows, int K, int N, int act,
                      hipStream_t s) {
    LDM_REQUIRE(in && W && out && rows > 0 && K > 0 && N > 0, "bad arguments");
    const size_t lds = (size_t)4 * K * 4 + 16 * 4 * 16 * 4;
    LDM_REQUIRE(lds <= 64 * 1024, "K too large for the time-path Linear");
    dim3 grid((unsigned)((N + 15) / 16), (unsigned)((rows + 3) / 4));
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), lds, s, in, W, b, out, rows, K, N, act);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// ---------------------------------------------------------------------------------------------
// Wide Linear, weight-bandwidth bound (VAE decoder transfer layer: 1024 -> 32768, 64 MiB of bf16 weights
// for at most a few dozen rows).  D^T = W^T X^T on v_mfma_f32_16x16x32_bf16: the weights are the A operand,
// stored in fragment order so that a wave streams them from HBM in 1-KiB loads and every byte is read
// once; up to 64 input rows sit in LDS as bf16 (rows XOR-swizzled per 16-byte piece) and are the B operand.
// A wave owns 16 output columns at a time and all of K; more than 64 rows take another pass over the weights.
// ---------------------------------------------------------------------------------------------

template <int NB>     // weight fragments in flight per wave (K / 32 must be a multiple)
__global__ __launch_bounds__(256, 1) void linear_bf16_kernel(const float* __restrict__ in, const char* __restrict__ wp,
                                                             const float* __restrict__ bias, char* __restrict__ out,
                                                             int rows, int K, int N, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int KS = K / 32, PPR = K / 8;           // k-steps; 16-byte pieces per row
    const int RB = K * 2;
    // The weight stream is cut into batches of NB fragments, batch q = (column tile q / BPN of this wave, k-steps
    // (q % BPN) * NB ...); batch q+1 is loaded while batch q is multiplied, and batch 0 before the input rows
    // are staged, so the stream from HBM starts with the kernel.
    const int BPN = KS / NB;
    const int nt0 = blockIdx.x * 4 + wave, nt_stride = gridDim.x * 4;
    const int ntiles = nt0 < N / 16 ? (N / 16 - nt0 + nt_stride - 1) / nt_stride : 0;
    const int nq = ntiles * BPN;
    auto load_batch = [&](int q, u32x4 (&dst)[NB]) {
        const int nt = nt0 + (q / BPN) * nt_stride, ks0 = (q % BPN) * NB;
        const char* wt = wp + ((size_t)nt * KS + ks0) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < NB; ++j) dst[j] = *(const u32x4*)(wt + (size_t)j * 1024);
    };
    for (int m0 = 0; m0 < rows; m0 += 64) {
        u32x4 wa[NB], wb[NB];
        if (nq > 0) load_batch(0, wa);
        if (m0) __syncthreads();
        for (int idx = tid; idx < 64 * PPR; idx += 256) {
            const int m = idx / PPR, p = idx - m * PPR;
            f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
            if (m0 + m < rows) {
                const float* src = in + (size_t)(m0 + m) * K + p * 8;
                lo = *(const f32x4*)src;
                hi = *(const f32x4*)(src + 4);
            }
            u32x4 v;
            v[0] = lin_pack2(lo[0], lo[1]); v[1] = lin_pack2(lo[2], lo[3]);
            v[2] = lin_pack2(hi[0], hi[1]); v[3] = lin_pack2(hi[2], hi[3]);
            *(u32x4*)(smem_raw + m * RB + ((p ^ (m & 15)) << 4)) = v;
        }
        __syncthreads();
        f32x4 acc[4];
        auto compute = [&](int q, const u32x4 (&a)[NB]) {
            const int nt = nt0 + (q / BPN) * nt_stride, ks0 = (q % BPN) * NB;
            if (ks0 == 0) {
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    const int m = mf * 16 + l15;
                    const u32x4 b = *(const u32x4*)(smem_raw + m * RB + ((((ks0 + j) * 4 + g) ^ l15) << 4));
                    acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lin_bf16x8, a[j]),
                                                                      __builtin_bit_cast(lin_bf16x8, b), acc[mf], 0, 0, 0);
                }
            if (ks0 + NB == KS) {
                // lane holds columns nt*16 + g*4 + (0..3) of row m0 + mf*16 + l15
                const int n = nt * 16 + g * 4;
                const f32x4 bv = bias ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    const int m = m0 + mf * 16 + l15;
                    f32x4 v = acc[mf] + bv;
                    if (act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (m < rows) {
                        uint2 o;
                        o.x = lin_pack2(v[0], v[1]);
                        o.y = lin_pack2(v[2], v[3]);
                        *(uint2*)(out + ((size_t)m * N + n) * 2) = o;
                    }
                }
            }
        };
        for (int q = 0; q < nq; q += 2) {
            if (q + 1 < nq) load_batch(q + 1, wb);
            compute(q, wa);
            if (q + 1 < nq) {
                if (q + 2 < nq) load_batch(q + 2, wa);
                compute(q + 1, wb);
            }
        }
    }
}

int launch_linear_bf16(const float* in, const void* wp, const float* b, void* out, int rows, int K, int N, int act,
                       hipStream_t s) {
    LDM_REQUIRE(in && wp && out && rows > 0, "bad arguments");
    LDM_REQUIRE(K > 0 && K % 128 == 0 && K <= 1024 && N > 0 && N % 16 == 0, "Linear shape not served (K multiple of 128 up to 1024, N multiple of 16)");
    LDM_REQUIRE(act == ACT_NONE || act == ACT_RELU, "activation not served by the wide Linear");
    int ncu = 0;
    LDM_TRY(device_cu_count(&ncu));
    static bool lds_set16[kMaxDevices] = {}, lds_set4[kMaxDevices] = {};   // per kernel and device
    LDM_TRY(allow_dynamic_lds((const void*)linear_bf16_kernel<16>, 160 * 1024, lds_set16));
    LDM_TRY(allow_dynamic_lds((const void*)linear_bf16_kernel<4>, 160 * 1024, lds_set4));
    const size_t lds = (size_t)64 * K * 2;
    int grid = N / 64;
    if (grid > ncu) grid = ncu;
    if (grid < 1) grid = 1;
    // 16 KiB of weights in flight per wave when K allows (the stream from HBM is the whole cost)
    auto kern = (K / 32) % 16 == 0 ? linear_bf16_kernel<16> : linear_bf16_kernel<4>;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, in, (const char*)wp, b, (char*)out, rows, K, N, act);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// ((v + 1) * 127.5) clamped to [0, 255] and truncated, f32 like the reference (an add, then a multiply:
// nothing to contract).  One thread per pixel.
__global__ void to_rgb8_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, int C, long HW) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* src = in + (size_t)blockIdx.y * C * HW + p;
    unsigned char* dst = out + ((size_t)blockIdx.y * HW + p) * C;
    for (int c = 0; c < C; ++c) {
        float v = (src[(size_t)c * HW] + 1.0f) * 127.5f;
        v = fminf(fmaxf(v, 0.0f), 255.0f);       // NaN -> 0, as Rust's clamp-then-cast gives for `as u8`
        dst[c] = (unsigned char)(int)v;
    }
}

int launch_to_rgb8(const float* in, unsigned char* out, int B, int C, int H, int W, hipStream_t s) {
    LDM_REQUIRE(in && out && B > 0 && B <= 65535 && C > 0 && H > 0 && W > 0, "bad arguments");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(to_rgb8_kernel, dim3((unsigned)((HW + 255) / 256), (unsigned)B), dim3(256), 0, s, in, out, C, HW);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

int launch_add_noise(const float* x, const float* noise, const int* t, const float* coef, int num_time, float* out, int B,
                     long per_sample, hipStream_t s) {
    LDM_REQUIRE(x && noise && t && coef && out, "null tensor");
    LDM_REQUIRE(B > 0 && B <= 65535 && num_time > 0 && per_sample > 0 && per_sample % 4 == 0,
                "batch up to 65535 and a sample size that is a multiple of 4");
    const long p4 = per_sample / 4;
    long gx = (p4 + 255) / 256;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, s, x, noise, t, coef, num_time, out, p4);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

int launch_mse(const float* a, const float* b, long n, float* partial_ws, float* out, hipStream_t s) {
    LDM_REQUIRE(a && b && partial_ws && out && n > 0, "bad arguments");
    hipLaunchKernelGGL(mse_partial_kernel, dim3(MSE_BLOCKS), dim3(256), 0, s, a, b, n, partial_ws);
    hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, s, (const float*)partial_ws, n, out);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// The partial sums of the allocation-free entry point live in this array of the code object instead of a caller's
// workspace.  Every GPU the library runs on holds its own copy at its own address; the name below is a host-side
// stand-in, so host code only ever uses the per-device address that hipGetSymbolAddress returned.
static __device__ __align__(256) float scratch[MSE_BLOCKS];
static float* g_scratch_dev[kMaxDevices] = {};
static std::mutex g_scratch_mu;

static int mse_scratch(float** out) {
    int dev = 0;
    LDM_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) LDM_FAIL(LDM_ERR_INVALID, "device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_dev[dev]) {
        float* p = nullptr;
        LDM_HIP(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (!p) LDM_FAIL(LDM_ERR_HIP, "hipGetSymbolAddress returned a null address");
        g_scratch_dev[dev] = p;
    }
    *out = g_scratch_dev[dev];
    return LDM_OK;
}

// Resolves the array's address on the current device (what the first allocation-free call on a device would do
// itself): for callers that must not take the lock or load the code object inside the hot call, e.g. under capture.
int mse_noalloc_prepare() {
    float* ws = nullptr;
    return mse_scratch(&ws);
}

int launch_mse_noalloc(const float* a, const float* b, long n, float* out, hipStream_t s) {
    LDM_REQUIRE(a && b && out && n > 0, "bad arguments");
    LDM_REQUIRE(n <= LDM_MSE_NOALLOC_MAX_N, "more elements than the allocation-free entry point takes (LDM_MSE_NOALLOC_MAX_N)");
    float* ws = nullptr;
    LDM_TRY(mse_scratch(&ws));
    // every one of the MSE_BLOCKS words is written by the first kernel before the second reads it: nothing to clear
    return launch_mse(a, b, n, ws, out, s);
}

int launch_finite_check(const float* z, long n, int* flag, hipStream_t s) {
    LDM_REQUIRE(z && flag && n > 0, "bad arguments");
    LDM_REQUIRE(((uintptr_t)z & 3) == 0, "the tensor must be 4-byte aligned");
    long head = (long)((16 - ((uintptr_t)z & 15)) & 15) / 4;     // elements before the first 16-byte boundary
    if (head > n) head = n;
    LDM_TRY(launch_set_int(flag, 1, s));
    hipLaunchKernelGGL(finite_check_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, s, z, n, (int)head, flag);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// emb[r, i] = sin(t_r * f_i), emb[r, half + i] = cos(t_r * f_i), f_i = exp(-(i * 4 ln10 / half))
__global__ void sin_embed_kernel(const int* __restrict__ t, int rows, int dim, float* __restrict__ out) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * half) return;
    const int r = idx / half, i = idx - r * half;
    const float expo = (float)i * 4.0f * 2.302585092994046f / (float)half;
    const float f = expf(-expo);
    const float arg = (float)t[r] * f;
    out[(size_t)r * dim + i] = sinf(arg);
    out[(size_t)r * dim + half + i] = cosf(arg);
}

int launch_sin_embed(const int* t, int rows, int dim, float* out, hipStream_t s) {
    LDM_REQUIRE(t && out && rows > 0 && dim > 0 && dim % 2 == 0, "bad arguments");
    const int n = rows * (dim / 2);
    hipLaunchKernelGGL(sin_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, rows, dim, out);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

}  // namespace ldm
