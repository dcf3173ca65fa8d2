// Training-mode BatchNorm of the UNet's ResidualBlocks (gfx950): the statistics of the batch instead of the running
// ones.  DiffusionModel::forward (src/main.rs:635-679) only ever runs on the autodiff backend (:1318), where burn's
// BatchNorm (bn1 / bn2, :340-341, 356, 366, 374, 385) takes its training branch:
//     mean_c = mean over (B, H, W) of x[:, c];   var_c = mean over (B, H, W) of (x[:, c] - mean_c)^2   (biased)
//     y = ((x - mean_c) / sqrt(var_c + eps)) * gamma_c + beta_c
// and Dropout (mid_drop, :403, 458, 513) multiplies by a Bernoulli(1 - p) mask and by 1 / (1 - p).
//
// Three bandwidth-bound kernels on NHWC tensors (bf16 or f32, 16 bytes per lane):
//   bn_partial_kernel   every workgroup reads a slice of pixels ONCE into registers and writes, per channel, the
//                       slice's mean and its sum of squared deviations from THAT mean (no E[x^2] - mean^2);
//   bn_finalize_kernel  merges the slices in a fixed order in f64 (Chan et al.'s pairwise update) into the batch
//                       mean and variance: the result does not depend on scheduling;
//   bn_apply_kernel     normalise + affine, then the ResidualBlock's tail in the same pass: SiLU and the block's time
//                       vector (:375-382), or the residual add and SiLU (:386), optionally the dropout mask.
#include "common.h"

namespace ldm {

__device__ __forceinline__ bf16_t bn_f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ float bn_bf2f(bf16_t h) { return __builtin_bit_cast(float, (unsigned int)h << 16); }

template <typename T> struct BnVec;
template <> struct BnVec<float> {
    static constexpr int VE = 4;
    static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
        const f32x4 r = *(const f32x4*)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = r[e];
    }
    static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = v[e];
        *(f32x4*)p = r;
    }
};
template <> struct BnVec<bf16_t> {
    static constexpr int VE = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const u32x4 r = *(const u32x4*)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = bn_bf2f((bf16_t)(r[e] & 0xffffu));
            v[2 * e + 1] = bn_bf2f((bf16_t)(r[e] >> 16));
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (unsigned int)bn_f2bf(v[2 * e]) | ((unsigned int)bn_f2bf(v[2 * e + 1]) << 16);
        *(u32x4*)p = r;
    }
};

constexpr int BN_RMAX = 8;      // pixels a thread holds in registers
constexpr int BN_SEGS = 16;     // segments of the slice list in the second stage

// h [P][C]; G = C / VE lanes per pixel, `rows` = pixels per pass (rows * G <= 256 threads are active), R <= BN_RMAX
// passes: a workgroup owns rows * R consecutive pixels.  part [nblk][2][C] = {slice mean, slice M2}.
template <typename T>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ h, long P, int C, int G, int rows, int R,
                                                         float* __restrict__ part) {
    constexpr int VE = BnVec<T>::VE;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;                  // [rows][C]
    const int tid = threadIdx.x;
    const bool active = tid < rows * G;
    const int r = tid / G, g = tid - r * G;
    const long base = (long)blockIdx.x * rows * R;
    long nblk = P - base;
    if (nblk > (long)rows * R) nblk = (long)rows * R;
    float v[BN_RMAX][VE];
    float s[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s[e] = 0.f;
#pragma unroll
    for (int j = 0; j < BN_RMAX; ++j) {
        const long p = base + (long)j * rows + r;
        const bool ok = active && j < R && p < P;
        if (ok) BnVec<T>::load(h + (size_t)p * C + g * VE, v[j]);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            if (!ok) v[j][e] = 0.f;
            s[e] += v[j][e];
        }
    }
    if (active) {
#pragma unroll
        for (int e = 0; e < VE; ++e) red[r * C + g * VE + e] = s[e];
    }
    __syncthreads();
    float mean[VE];
    const float inv_n = 1.0f / (float)nblk;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        float a = 0.f;
        if (active)
            for (int q = 0; q < rows; ++q) a += red[q * C + g * VE + e];
        mean[e] = a * inv_n;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VE; ++e) s[e] = 0.f;
#pragma unroll
    for (int j = 0; j < BN_RMAX; ++j) {
        const long p = base + (long)j * rows + r;
        const bool ok = active && j < R && p < P;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float d = ok ? v[j][e] - mean[e] : 0.f;
            s[e] += d * d;
        }
    }
    if (active) {
#pragma unroll
        for (int e = 0; e < VE; ++e) red[r * C + g * VE + e] = s[e];
    }
    __syncthreads();
    if (active && r == 0) {
        float* dst = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float a = 0.f;
            for (int q = 0; q < rows; ++q) a += red[q * C + g * VE + e];
            dst[g * VE + e] = mean[e];
            dst[C + g * VE + e] = a;
        }
    }
}

// One workgroup per 16 channels: thread (segment, channel) merges its share of the slices one after the other, then
// the 16 segment results are merged by the segment-0 thread, all in f64:
//   n = na + nb;  d = mb - ma;  m = ma + d nb / n;  M2 = M2a + M2b + d^2 na nb / n.
// params [C][4] = {mean, sqrt(var + eps), gamma, beta}; mean_out / var_out [C] (optional): the batch statistics, for
// the caller's running-average update (running = (1 - momentum) running + momentum batch, burn's momentum 0.1).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, long pix_per_blk, long P,
                                                          int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ params, float* __restrict__ mean_out,
                                                          float* __restrict__ var_out) {
    __shared__ double sm[BN_SEGS][16], sM2[BN_SEGS][16], sn[BN_SEGS][16];
    const int cl = threadIdx.x & 15, seg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int per = (nblk + BN_SEGS - 1) / BN_SEGS;
    double n = 0.0, m = 0.0, M2 = 0.0;
    if (c < C) {
        const int b1 = min(nblk, (seg + 1) * per);
        for (int b = seg * per; b < b1; ++b) {
            long cnt = P - (long)b * pix_per_blk;
            if (cnt > pix_per_blk) cnt = pix_per_blk;
            const double nb = (double)cnt;
            const double mb = (double)part[(size_t)b * 2 * C + c], M2b = (double)part[(size_t)b * 2 * C + C + c];
            const double nn = n + nb, d = mb - m;
            m += d * nb / nn;
            M2 += M2b + d * d * n * nb / nn;
            n = nn;
        }
    }
    sm[seg][cl] = m; sM2[seg][cl] = M2; sn[seg][cl] = n;
    __syncthreads();
    if (seg == 0 && c < C) {
        for (int s2 = 1; s2 < BN_SEGS; ++s2) {
            const double nb = sn[s2][cl];
            if (nb == 0.0) continue;
            const double nn = n + nb, d = sm[s2][cl] - m;
            m += d * nb / nn;
            M2 += sM2[s2][cl] + d * d * n * nb / nn;
            n = nn;
        }
        const float mean = (float)m, var = (float)(M2 / n);
        params[c * 4 + 0] = mean;
        params[c * 4 + 1] = sqrtf(var + eps);
        params[c * 4 + 2] = gamma[c];
        params[c * 4 + 3] = beta[c];
        if (mean_out) mean_out[c] = mean;
        if (var_out) var_out[c] = var;
    }
}

struct BnApplyArgs {
    const void* h;         // [B][HW][C] conv output
    const void* res;       // [B][HW][C] added before the activation, or null
    void* out;             // [B][HW][C]; may be h
    const float* params;   // [C][4] from bn_finalize_kernel
    const float* tvec;     // added after the activation: tvec[b * t_img_stride + c], or null
    int t_img_stride;
    int C, G;
    long HW;
    int act;               // ACT_NONE / ACT_SILU / ACT_RELU
    // dropout after everything else (keep_prob >= 1: none): element (b, pixel, c) is kept when the Philox-4x32-10
    // word of counter {(pixel * C + c) / 4, 0xFFFFFFFE, b, 0x4C444D32} under key `seed`, lane c % 4, gives
    // ((word >> 8) + 0.5) 2^-24 < keep_prob; kept elements are multiplied by 1 / keep_prob.
    float keep_prob;
    unsigned long long seed;
};

__device__ __forceinline__ void bn_philox(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0,
                                          unsigned int k1, unsigned int out[4]) {
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

// grid (x over an image's vectors, y = image).  The operation order is burn's: subtract, divide, multiply, add.
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a) {
    constexpr int VE = BnVec<T>::VE;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* prm = (f32x4*)smem_raw;                  // [C]
    for (int c = threadIdx.x; c < a.C; c += 256) prm[c] = *(const f32x4*)(a.params + c * 4);
    __syncthreads();
    const int b = blockIdx.y;
    const int nvec = (int)(a.HW * a.G);
    const size_t img = (size_t)b * a.HW * a.C;
    const T* h = (const T*)a.h + img;
    const T* res = a.res ? (const T*)a.res + img : nullptr;
    T* out = (T*)a.out + img;
    const float* tv = a.tvec ? a.tvec + (size_t)b * a.t_img_stride : nullptr;
    const bool drop = a.keep_prob < 1.0f;
    const float inv_keep = drop ? 1.0f / a.keep_prob : 1.0f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
        const int pix = i / a.G;
        const int c0 = (i - pix * a.G) * VE;
        float v[VE], rv[VE];
        BnVec<T>::load(h + (size_t)i * VE, v);
        if (res) BnVec<T>::load(res + (size_t)i * VE, rv);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const f32x4 p = prm[c0 + e];
            // f32 tensors divide exactly; bf16 tensors multiply by the hardware reciprocal (1 ulp of f32, far below
            // the rounding of the stored result)
            float y;
            if constexpr (sizeof(T) == 4) y = (v[e] - p[0]) / p[1];
            else y = (v[e] - p[0]) * __builtin_amdgcn_rcpf(p[1]);
            y = y * p[2] + p[3];
            if (res) y += rv[e];
            if (a.act == ACT_SILU) {
                const float d = 1.0f + __expf(-y);
                if constexpr (sizeof(T) == 4) y = y / d;
                else y = y * __builtin_amdgcn_rcpf(d);
            } else if (a.act == ACT_RELU) {
                y = fmaxf(y, 0.f);
            }
            if (tv) y += tv[c0 + e];
            v[e] = y;
        }
        if (drop) {
#pragma unroll
            for (int q = 0; q < VE / 4; ++q) {
                unsigned int w[4];
                bn_philox((unsigned int)(((long)pix * a.C + c0) / 4 + q), 0xFFFFFFFEu, (unsigned int)b, 0x4C444D32u,
                          (unsigned int)a.seed, (unsigned int)(a.seed >> 32), w);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = ((float)(w[e] >> 8) + 0.5f) * 5.9604644775390625e-08f;
                    v[q * 4 + e] = u < a.keep_prob ? v[q * 4 + e] * inv_keep : 0.f;
                }
            }
        }
        BnVec<T>::store(out + (size_t)i * VE, v);
    }
}

// Launch geometry of the statistics stage for P pixels of C channels: lanes per pixel, pixels per pass, passes,
// slices.  About 2048 slices when the tensor is large enough, never more than BN_RMAX pixels per thread.
int bn_plan(int dtype, long P, int C, BnPlan* pl) {
    const int ve = dtype == LDM_BF16 ? 8 : 4;
    LDM_REQUIRE(P > 0 && C > 0 && C % ve == 0 && C / ve <= 256, "channel count not served by the batch-statistics kernels");
    pl->G = C / ve;
    pl->rows = 256 / pl->G;
    long R = (P + (long)pl->rows * 2048 - 1) / ((long)pl->rows * 2048);
    if (R < 1) R = 1;
    if (R > BN_RMAX) R = BN_RMAX;
    pl->R = (int)R;
    pl->pix_per_blk = (long)pl->rows * pl->R;
    const long nblk = (P + pl->pix_per_blk - 1) / pl->pix_per_blk;
    LDM_REQUIRE(nblk <= 0x7fffffffL, "tensor too large for the batch-statistics kernels");
    pl->nblk = (int)nblk;
    return LDM_OK;
}

long bn_ws_floats(int dtype, long P, int C) {
    BnPlan pl;
    if (bn_plan(dtype, P, C, &pl) != LDM_OK) return -1;
    return (long)pl.nblk * 2 * C + 4L * C;
}

// h [B][HW][C] NHWC of `dtype` elements.  ws: bn_ws_floats(dtype, B * HW, C) floats (slices, then params).
int launch_bn_stats(int dtype, const void* h, int B, long HW, int C, const float* gamma, const float* beta, float eps,
                    float* ws, float* mean_out, float* var_out, hipStream_t s) {
    LDM_REQUIRE(h && gamma && beta && ws && B > 0 && HW > 0, "bad arguments");
    LDM_REQUIRE(((uintptr_t)h & 15) == 0, "the tensor must be 16-byte aligned");
    const long P = (long)B * HW;
    BnPlan pl;
    LDM_TRY(bn_plan(dtype, P, C, &pl));
    float* part = ws;
    float* params = ws + (size_t)pl.nblk * 2 * C;
    const size_t lds = (size_t)pl.rows * C * 4;
    if (dtype == LDM_BF16)
        hipLaunchKernelGGL(bn_partial_kernel<bf16_t>, dim3((unsigned)pl.nblk), dim3(256), lds, s, (const bf16_t*)h, P, C, pl.G,
                           pl.rows, pl.R, part);
    else
        hipLaunchKernelGGL(bn_partial_kernel<float>, dim3((unsigned)pl.nblk), dim3(256), lds, s, (const float*)h, P, C, pl.G,
                           pl.rows, pl.R, part);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, (const float*)part, pl.nblk,
                       pl.pix_per_blk, P, C, gamma, beta, eps, params, mean_out, var_out);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

int launch_bn_apply(int dtype, const void* h, const void* res, void* out, int B, long HW, int C, const float* ws, int act,
                    const float* tvec, int t_img_stride, float keep_prob, unsigned long long seed, hipStream_t s) {
    LDM_REQUIRE(h && out && ws && B > 0 && B <= 65535 && HW > 0, "bad arguments (batch up to 65535)");
    LDM_REQUIRE((((uintptr_t)h | (uintptr_t)out | (uintptr_t)res) & 15) == 0, "tensors must be 16-byte aligned");
    LDM_REQUIRE(keep_prob > 0.f, "keep probability must be positive");
    LDM_REQUIRE(keep_prob >= 1.f || C % 4 == 0, "dropout needs a channel count that is a multiple of 4");
    BnPlan pl;
    LDM_TRY(bn_plan(dtype, (long)B * HW, C, &pl));
    BnApplyArgs a{};
    a.h = h; a.res = res; a.out = out;
    a.params = ws + (size_t)pl.nblk * 2 * C;
    a.tvec = tvec; a.t_img_stride = t_img_stride;
    a.C = C; a.G = pl.G; a.HW = HW; a.act = act;
    a.keep_prob = keep_prob; a.seed = seed;
    const long nvec = HW * pl.G;
    LDM_REQUIRE(nvec < 0x7fffffffL, "image too large for the normalisation kernel");
    // about 2048 workgroups over the whole batch, each thread with a few vectors in flight
    long gx = (nvec + 255) / 256;
    const long want = (2048 + B - 1) / B;
    if (gx > want) gx = want;
    if (gx < 1) gx = 1;
    const size_t lds = (size_t)C * 16;
    LDM_REQUIRE(lds <= 64 * 1024, "too many channels for the normalisation kernel");
    if (dtype == LDM_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3((unsigned)gx, (unsigned)B), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(bn_apply_kernel<float>, dim3((unsigned)gx, (unsigned)B), dim3(256), lds, s, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

}  // namespace ldm
