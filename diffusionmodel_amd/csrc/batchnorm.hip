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
//   bn_finalize_kernel  merges the slices in f64 and in a fixed order into the batch mean and variance (the exact
//                       decomposition M2 = sum of the slices' M2 + n (slice mean - mean)^2): the result does not
//                       depend on scheduling;
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
    static __device__ __forceinline__ void unpack(const u32x4 r, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned int u = r[e];       // a copy: bit_cast applied to the element expression itself read element 0
            v[e] = __builtin_bit_cast(float, u);
        }
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
    static __device__ __forceinline__ void unpack(const u32x4 r, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, r[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, r[e] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        u32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (unsigned int)bn_f2bf(v[2 * e]) | ((unsigned int)bn_f2bf(v[2 * e + 1]) << 16);
        *(u32x4*)p = r;
    }
};

constexpr int BN_RMAX = 16;     // pixels a thread holds in registers (as the 16 bytes it loaded)
constexpr int BN_SEGS = 64;     // segments of the slice list in the second stage
constexpr int BN_FCH = 4;       // channels per workgroup of the second stage

// Sum of s[e] over the threads of a workgroup that share the channel lanes g (thread = r * G + g), left in every one
// of them.  G a power of two up to 64: the rows inside a wave meet by butterfly exchanges (the same order in every
// lane), the waves through LDS; other G (rows <= 256 / G): a loop over the rows in LDS.  red: 256 * VE floats.
template <int VE>
__device__ __forceinline__ void bn_rows_sum(float (&s)[VE], float* red, int tid, int G, int rows, int C, bool active,
                                            bool pow2) {
    const int r = tid / G, g = tid - r * G;
    if (pow2) {
        for (int o = G; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < VE; ++e) s[e] += __shfl_xor(s[e], o, 64);
        }
        const int wave = tid >> 6, lane = tid & 63;
        const int nw = (rows * G + 63) >> 6;
        if (nw > 1) {
            if (active && lane < G) {
#pragma unroll
                for (int e = 0; e < VE; ++e) red[wave * C + g * VE + e] = s[e];
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                float a = 0.f;
                if (active)
                    for (int w = 0; w < nw; ++w) a += red[w * C + g * VE + e];
                s[e] = a;
            }
            __syncthreads();
        }
    } else {
        if (active) {
#pragma unroll
            for (int e = 0; e < VE; ++e) red[r * C + g * VE + e] = s[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float a = 0.f;
            if (active)
                for (int q = 0; q < rows; ++q) a += red[q * C + g * VE + e];
            s[e] = a;
        }
        __syncthreads();
    }
}

// h [P][C]; G = C / VE lanes per pixel, `rows` = pixels per pass (rows * G <= 256 threads are active), R <= BN_RMAX
// passes: a workgroup owns rows * R consecutive pixels.  part [nblk][2][C] = {slice mean, slice M2}.
// Every load is unconditional (a slot beyond the tensor reads the last pixel and is masked afterwards): a load under
// a branch is followed by a wait before the next one is issued, which costs a memory latency per slot.
template <typename T, int R>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ h, long P, int C, int G, int rows,
                                                         float* __restrict__ part) {
    constexpr int VE = BnVec<T>::VE;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = (float*)smem_raw;                  // [rows or waves][C]
    const int tid = threadIdx.x;
    const bool active = tid < rows * G;
    const bool pow2 = G <= 64 && (G & (G - 1)) == 0;
    const int r = tid / G, g = tid - r * G;
    const long base = (long)blockIdx.x * rows * R;
    long nblk = P - base;
    if (nblk > (long)rows * R) nblk = (long)rows * R;
    u32x4 raw[R];
    float s[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s[e] = 0.f;
    const int gl = active ? g : 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        long p = base + (long)j * rows + r;
        if (p > P - 1) p = P - 1;
        raw[j] = *(const u32x4*)(h + (size_t)p * C + gl * VE);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool ok = active && base + (long)j * rows + r < P;
        float v[VE];
        BnVec<T>::unpack(raw[j], v);
#pragma unroll
        for (int e = 0; e < VE; ++e) s[e] += ok ? v[e] : 0.f;
    }
    bn_rows_sum<VE>(s, red, tid, G, rows, C, active, pow2);
    float mean[VE];
    const float inv_n = 1.0f / (float)nblk;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        mean[e] = s[e] * inv_n;
        s[e] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const bool ok = active && base + (long)j * rows + r < P;
        float v[VE];
        BnVec<T>::unpack(raw[j], v);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float d = ok ? v[e] - mean[e] : 0.f;
            s[e] += d * d;
        }
    }
    bn_rows_sum<VE>(s, red, tid, G, rows, C, active, pow2);
    if (active && r == 0) {
        float* dst = part + (size_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            dst[g * VE + e] = mean[e];
            dst[C + g * VE + e] = s[e];
        }
    }
}

// Sum over the BN_SEGS segment values of one channel in a fixed order (a tree over LDS); every thread gets it.
__device__ __forceinline__ double bn_seg_sum(double v, double (*sh)[BN_FCH], int seg, int cl) {
    sh[seg][cl] = v;
    __syncthreads();
    for (int st = BN_SEGS / 2; st > 0; st >>= 1) {
        if (seg < st) sh[seg][cl] += sh[seg + st][cl];
        __syncthreads();
    }
    const double r = sh[0][cl];
    __syncthreads();
    return r;
}

// One workgroup per BN_FCH channels, thread (segment, channel): the slices' statistics merge exactly as
//   mean = sum_b n_b mean_b / N,   M2 = sum_b [M2_b + n_b (mean_b - mean)^2],   var = M2 / N      (biased)
// in f64, each sum over the thread's strided share of the slices and then over the segments in a fixed tree.
// params [C][4] = {mean, sqrt(var + eps), gamma, beta}; mean_out / var_out [C] (optional): the batch statistics, for
// the caller's running-average update (running = (1 - momentum) running + momentum batch, burn's momentum 0.1).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nblk, long pix_per_blk, long P,
                                                          int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float* __restrict__ params, float* __restrict__ mean_out,
                                                          float* __restrict__ var_out) {
    __shared__ double sh[BN_SEGS][BN_FCH];
    const int cl = threadIdx.x % BN_FCH, seg = threadIdx.x / BN_FCH;
    const int c = blockIdx.x * BN_FCH + cl;
    const bool ok = c < C;
    const int cc = ok ? c : C - 1;
    const double N = (double)P;
    const long last_cnt = P - (long)(nblk - 1) * pix_per_blk;
    // the slices of a thread are visited eight at a time with all their loads issued first: a loop that waits for
    // each load costs a memory latency per slice
    double acc = 0.0;
    for (int b0 = seg; b0 < nblk; b0 += 8 * BN_SEGS) {
        float mb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * BN_SEGS;
            mb[k] = part[(size_t)min(b, nblk - 1) * 2 * C + cc];     // unconditional: weighted by 0 below when unused
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * BN_SEGS;
            const double nb = b >= nblk ? 0.0 : (double)(b == nblk - 1 ? last_cnt : pix_per_blk);
            acc += nb * (double)mb[k];
        }
    }
    const double mean = bn_seg_sum(acc, sh, seg, cl) / N;
    acc = 0.0;
    for (int b0 = seg; b0 < nblk; b0 += 8 * BN_SEGS) {
        float mb[8], qb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * BN_SEGS;
            mb[k] = part[(size_t)min(b, nblk - 1) * 2 * C + cc];
            qb[k] = part[(size_t)min(b, nblk - 1) * 2 * C + C + cc];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + k * BN_SEGS;
            const double nb = b >= nblk ? 0.0 : (double)(b == nblk - 1 ? last_cnt : pix_per_blk);
            const double d = (double)mb[k] - mean;
            acc += b >= nblk ? 0.0 : (double)qb[k] + nb * d * d;
        }
    }
    const double M2 = bn_seg_sum(acc, sh, seg, cl);
    if (ok && seg == 0) {
        const float mf = (float)mean, var = (float)(M2 / N);
        params[c * 4 + 0] = mf;
        params[c * 4 + 1] = sqrtf(var + eps);
        params[c * 4 + 2] = gamma[c];
        params[c * 4 + 3] = beta[c];
        if (mean_out) mean_out[c] = mf;
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
    float* tvl = (float*)(prm + a.C);               // [C]: the image's time vector
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        prm[c] = *(const f32x4*)(a.params + c * 4);
        tvl[c] = a.tvec ? a.tvec[(size_t)b * a.t_img_stride + c] : 0.f;
    }
    __syncthreads();
    const int nvec = (int)(a.HW * a.G);
    const size_t img = (size_t)b * a.HW * a.C;
    const T* h = (const T*)a.h + img;
    const T* res = a.res ? (const T*)a.res + img : nullptr;
    T* out = (T*)a.out + img;
    const bool tv = a.tvec != nullptr;
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
            if (tv) y += tvl[c0 + e];
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
// slices.  Never more than BN_RMAX pixels per thread.
int bn_plan(int dtype, long P, int C, BnPlan* pl) {
    const int ve = dtype == LDM_BF16 ? 8 : 4;
    LDM_REQUIRE(P > 0 && C > 0 && C % ve == 0 && C / ve <= 256, "channel count not served by the batch-statistics kernels");
    pl->G = C / ve;
    pl->rows = 256 / pl->G;
    // about one slice per compute unit: with 1024 and more slices both stages took 2-3 times as long (a slice's
    // life is a chain of memory latencies whatever its size, and the second stage reads every slice's record)
    long R = (P + (long)pl->rows * 256 - 1) / ((long)pl->rows * 256);
    int Rp = 1;                       // the kernel is instantiated for 1, 2, 4, 8 and 16 pixels per thread
    while (Rp < R && Rp < BN_RMAX) Rp *= 2;
    pl->R = Rp;
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
    const size_t lds = (size_t)(pl.rows > 4 ? pl.rows : 4) * C * 4;
    auto go = [&](auto kern_bf, auto kern_f) {
        if (dtype == LDM_BF16)
            hipLaunchKernelGGL(kern_bf, dim3((unsigned)pl.nblk), dim3(256), lds, s, (const bf16_t*)h, P, C, pl.G, pl.rows, part);
        else
            hipLaunchKernelGGL(kern_f, dim3((unsigned)pl.nblk), dim3(256), lds, s, (const float*)h, P, C, pl.G, pl.rows, part);
    };
    switch (pl.R) {
        case 1: go(bn_partial_kernel<bf16_t, 1>, bn_partial_kernel<float, 1>); break;
        case 2: go(bn_partial_kernel<bf16_t, 2>, bn_partial_kernel<float, 2>); break;
        case 4: go(bn_partial_kernel<bf16_t, 4>, bn_partial_kernel<float, 4>); break;
        case 8: go(bn_partial_kernel<bf16_t, 8>, bn_partial_kernel<float, 8>); break;
        default: go(bn_partial_kernel<bf16_t, 16>, bn_partial_kernel<float, 16>); break;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + BN_FCH - 1) / BN_FCH)), dim3(256), 0, s, (const float*)part, pl.nblk,
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
    const size_t lds = (size_t)C * 20;
    LDM_REQUIRE(lds <= 64 * 1024, "too many channels for the normalisation kernel");
    if (dtype == LDM_BF16) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3((unsigned)gx, (unsigned)B), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(bn_apply_kernel<float>, dim3((unsigned)gx, (unsigned)B), dim3(256), lds, s, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

}  // namespace ldm
