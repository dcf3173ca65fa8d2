// Single-head self-attention core on MFMA for gfx950 (flash style, scores never materialised):
//   out[b, i, :] = sum_j softmax_j(q_i . k_j / sqrt(C)) v_j            (src/main.rs:265-270)
// over the N = H*W tokens of one image.  Input is the fused projection [B, N, 3C] (q | k | v) written
// by the conv1x1 kernel, output [B, N, C]; with NHWC activations the reference's NCHW<->token
// transposes (src/main.rs:258-259,273) disappear.
//
// A workgroup = 4 waves; each wave owns QF*16 queries.  K and V tiles of KT keys are staged by
// LDS-DMA into a two-deep ring.  Both products are computed transposed so that the query sits on the
// MFMA column (lane & 15) throughout:
//   S^T = K Q^T   (A = K rows from LDS by ds_read_b128, B = Q held in registers for the whole kernel)
//   O^T += V^T P^T (A = V^T; bf16: ds_read_b64_tr_b16 transposing reads of the row-major V tile,
//                   f32: ds_read_b32; B = P^T taken straight from the S^T accumulators: the MFMA k
//                   index is a permutation of the keys that both operands share)
// so soft-max statistics are per-lane scalars plus one 4-lane reduction, and nothing crosses LDS.
// LDS rows are XOR-swizzled per 16-byte piece (K: piece ^ (row & 15); V: piece ^ ((row & 7) << 1)) on the
// DMA source address and on the reads alike.
#include "common.h"

namespace ldm {

static __device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((ext_vector_type(4))) short short4v;

__device__ __forceinline__ void at_glds16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// by-value helper: bit_cast applied directly to a vector-element expression reads element 0
__device__ __forceinline__ float at_u2f(unsigned int u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned int at_pack2(float a, float b) {
    return (unsigned int)__builtin_bit_cast(unsigned short, (__bf16)a) |
           ((unsigned int)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}

template <typename T, int D, int QF, int KF, int MINB>
__global__ __launch_bounds__(256, MINB) void attention_kernel(const char* __restrict__ qkv, char* __restrict__ out, int N,
                                                           float scale_log2e) {
    constexpr int ES = sizeof(T);
    constexpr int RB = D * ES;              // bytes of one q / k / v row
    constexpr int KT = KF * 16;             // keys per tile
    constexpr int KS = RB / 64;             // k-steps over the head dimension (64 bytes per step)
    constexpr int DF = D / 16;
    constexpr int TILE = KT * RB;
    constexpr int NPIECE = TILE / 1024;
    constexpr int QB = 4 * QF * 16;
    static_assert(ES == 4 || KF == 2, "bf16 PV uses K = 32 keys per MFMA");
    static_assert(RB >= 256, "rows shorter than 16 pieces are not swizzled correctly");
    constexpr float RESCALE_LOG2 = 8.0f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * QB + wave * QF * 16;
    const char* base = qkv + (size_t)b * N * 3 * RB;
    const char* zero = (const char*)g_zero16;
    const int ntiles = (N + KT - 1) / KT;

    auto issue_tile = [&](int t) {
        char* kd = smem + (t & 1) * 2 * TILE;
        char* vd = kd + TILE;
        for (int i = wave; i < NPIECE; i += 4) {
            const int o = i * 1024 + lane * 16;
            const int r = o / RB;
            const int s = (o - r * RB) >> 4;
            const int key = t * KT + r;
            const char* row = base + (size_t)key * 3 * RB;
            const bool ok = key < N;
            at_glds16(ok ? row + RB + ((s ^ (r & 15)) << 4) : zero, kd + i * 1024);
            at_glds16(ok ? row + 2 * RB + ((s ^ ((r & 7) << 1)) << 4) : zero, vd + i * 1024);
        }
    };

    issue_tile(0);

    // Q fragments: B operand of S^T = K Q^T, column = query
    u32x4 qreg[QF][KS];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        const int q = q0 + qf * 16 + l15;
        const char* row = base + (size_t)(q < N ? q : 0) * 3 * RB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = *(const u32x4*)(row + (ks * 4 + g) * 16);
            if (q >= N) v = u32x4{0u, 0u, 0u, 0u};
            qreg[qf][ks] = v;
        }
    }

    f32x4 oacc[DF][QF];
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) oacc[df][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[QF], lrun[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) { mrun[qf] = -1e30f; lrun[qf] = 0.f; }

    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) issue_tile(t + 1);
        const char* kt = smem + (t & 1) * 2 * TILE;
        const char* vt = kt + TILE;

        // ---- S^T = K Q^T ----------------------------------------------------------------
        f32x4 sacc[KF][QF];
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) sacc[kf][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                const u32x4 kfrag = *(const u32x4*)(kt + (kf * 16 + l15) * RB + (((ks * 4 + g) ^ l15) << 4));
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) {
                    if constexpr (ES == 2) {
                        sacc[kf][qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, kfrag), __builtin_bit_cast(bf16x8, qreg[qf][ks]), sacc[kf][qf], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            sacc[kf][qf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                at_u2f(kfrag[e]), at_u2f(qreg[qf][ks][e]), sacc[kf][qf], 0, 0, 0);
                    }
                }
            }
        }

        // ---- online soft-max: lane holds keys kf*16 + g*4 + r of query column l15 --------
        float alpha[QF];
        bool moved = false;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float mx = -1e30f;
#pragma unroll
            for (int kf = 0; kf < KF; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * KT + kf * 16 + g * 4 + r;
                    float s = sacc[kf][qf][r] * scale_log2e;
                    if (key >= N) s = -1e30f;
                    sacc[kf][qf][r] = s;
                    mx = fmaxf(mx, s);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            // Lazy reference: the running reference m only moves when a tile's maximum exceeds it by more
            // than 2^RESCALE_LOG2; until then p = 2^(s - m) may be as large as that factor, which f32
            // sums and bf16 P values hold without loss, and O needs no rescaling (the common case
            // after the first tiles).  The first tile always moves it (m starts at -1e30, O is 0).
            const bool move = mx > mrun[qf] + RESCALE_LOG2;
            const float mnew = move ? mx : mrun[qf];
            alpha[qf] = move ? exp2f(mrun[qf] - mnew) : 1.0f;
            moved |= move;
            mrun[qf] = mnew;
            float rs = 0.f;
#pragma unroll
            for (int kf = 0; kf < KF; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = exp2f(sacc[kf][qf][r] - mnew);
                    sacc[kf][qf][r] = p;
                    rs += p;
                }
            lrun[qf] = lrun[qf] * alpha[qf] + rs;   // per-lane partial; the 4 key groups are summed at the end
        }
        if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {      // wave-uniform
#pragma unroll
            for (int df = 0; df < DF; ++df)
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) oacc[df][qf] *= alpha[qf];
        }

        // ---- O^T += V^T P^T -----------------------------------------------------------------
        if constexpr (ES == 2) {
            u32x4 pb[QF];
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) {
                pb[qf][0] = at_pack2(sacc[0][qf][0], sacc[0][qf][1]);
                pb[qf][1] = at_pack2(sacc[0][qf][2], sacc[0][qf][3]);
                pb[qf][2] = at_pack2(sacc[1][qf][0], sacc[1][qf][1]);
                pb[qf][3] = at_pack2(sacc[1][qf][2], sacc[1][qf][3]);
            }
            // transposing read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a
            // 4-key x 16-column block and receives column (lane & 15) of the four keys
            const int tq = l15 >> 2, tp = l15 & 3;
            const int row0 = g * 4 + tq, row1 = 16 + g * 4 + tq;
#pragma unroll
            for (int df = 0; df < DF; ++df) {
                const int piece = df * 2 + (tp >> 1);
                const char* a0 = vt + row0 * RB + ((piece ^ ((row0 & 7) << 1)) << 4) + (tp & 1) * 8;
                const char* a1 = vt + row1 * RB + ((piece ^ ((row1 & 7) << 1)) << 4) + (tp & 1) * 8;
                const short4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)a0);
                const short4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)a1);
                u32x4 vf;
                vf[0] = __builtin_bit_cast(uint2, v0).x; vf[1] = __builtin_bit_cast(uint2, v0).y;
                vf[2] = __builtin_bit_cast(uint2, v1).x; vf[3] = __builtin_bit_cast(uint2, v1).y;
#pragma unroll
                for (int qf = 0; qf < QF; ++qf)
                    oacc[df][qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pb[qf]), oacc[df][qf], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int df = 0; df < DF; ++df) {
#pragma unroll
                for (int kf = 0; kf < KF; ++kf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = kf * 16 + g * 4 + r;
                        const int piece = (df * 16 + l15) >> 2;
                        const float vv = *(const float*)(vt + row * RB + ((piece ^ ((row & 7) << 1)) << 4) + (l15 & 3) * 4);
#pragma unroll
                        for (int qf = 0; qf < QF; ++qf)
                            oacc[df][qf] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, sacc[kf][qf][r], oacc[df][qf], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }

    // ---- normalise and store: lane holds d = df*16 + g*4 + (0..3) of query q0 + qf*16 + l15 ----
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        float l = lrun[qf];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int q = q0 + qf * 16 + l15;
        if (q >= N) continue;
        char* orow = out + ((size_t)b * N + q) * RB;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const f32x4 v = oacc[df][qf] * inv;
            const int d = df * 16 + g * 4;
            if constexpr (ES == 2) {
                uint2 w;
                w.x = at_pack2(v[0], v[1]);
                w.y = at_pack2(v[2], v[3]);
                *(uint2*)(orow + d * 2) = w;
            } else {
                *(f32x4*)(orow + d * 4) = v;
            }
        }
    }
}

template <typename T, int D, int QF, int KF, int MINB = 1>
static int launch_attn_cfg(const void* qkv, void* out, int B, int N, hipStream_t stream) {
    constexpr int TILE = KF * 16 * D * (int)sizeof(T);
    constexpr int QB = 4 * QF * 16;
    const size_t lds = (size_t)4 * TILE;
    static_assert(4 * TILE <= 160 * 1024, "LDS budget");
    auto kern = attention_kernel<T, D, QF, KF, MINB>;
    static bool attr_set = false;
    if (!attr_set) {
        LDM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
    dim3 grid((unsigned)((N + QB - 1) / QB), (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const char*)qkv, (char*)out, N, scale_log2e);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

int launch_attention(int dtype, const void* qkv, void* out, int B, int N, int C, hipStream_t stream) {
    LDM_REQUIRE(qkv && out && B > 0 && N > 0, "bad arguments");
    LDM_REQUIRE(B <= 65535, "batch too large for one launch");
    if (dtype == LDM_BF16) {
        switch (C) {
            case 128: return launch_attn_cfg<bf16_t, 128, 2, 2>(qkv, out, B, N, stream);
            case 256:
                // 16 queries per wave while that still leaves every CU a workgroup; O then stays in VGPRs
                if ((long)B * ((N + 127) / 128) < 256) return launch_attn_cfg<bf16_t, 256, 1, 2, 2>(qkv, out, B, N, stream);
                return launch_attn_cfg<bf16_t, 256, 2, 2>(qkv, out, B, N, stream);
            case 512: return launch_attn_cfg<bf16_t, 512, 1, 2>(qkv, out, B, N, stream);
        }
    } else {
        switch (C) {
            case 64: return launch_attn_cfg<float, 64, 2, 2>(qkv, out, B, N, stream);
            case 128: return launch_attn_cfg<float, 128, 2, 2>(qkv, out, B, N, stream);
            case 256: return launch_attn_cfg<float, 256, 2, 1>(qkv, out, B, N, stream);
            case 512: return launch_attn_cfg<float, 512, 1, 1>(qkv, out, B, N, stream);
        }
    }
    LDM_FAIL(LDM_ERR_INVALID, "attention supports 128, 256 or 512 channels (and 64 in f32)");
}

}  // namespace ldm
