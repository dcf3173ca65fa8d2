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
#include <utility>

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
// the same rounding as one v_cvt_pk_bf16_f32
typedef __attribute__((ext_vector_type(2))) float at_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 at_bf16x2;
__device__ __forceinline__ unsigned int at_pack2v(float a, float b) {
    const at_f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(f, at_bf16x2));
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

// write-through 16-byte output store (the convolution kernels' store_out): nothing dirty is left in this XCD's L2 for
// the end of the launch to write back
__device__ __forceinline__ void at_store_out(char* p, const u32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int N_>
__device__ __forceinline__ void at_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// LDS reads that the compiler neither reorders among themselves nor waits for: the caller pairs each
// with at_wait_lgkm<N>(value) (N = LDS instructions issued after the read that may still be in flight;
// LDS instructions return in order) before the first use of the value.  Two reasons to bypass the
// compiler here: short of registers it waits for every read right before the two MFMAs it feeds, and
// before a transposing read it drains every LDS-DMA in flight.
template <int OFF>
__device__ __forceinline__ u32x4 at_lds_read128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ uint2 at_lds_read_tr16(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// A global load that the compiler does not track: it waits for a tracked load with vmcnt(0) at the first use, which
// would also drain every tile DMA issued behind it (that is what stood in front of the first interval: Q, and with it
// the three tiles of the prologue).  The caller pairs it with at_wait_vmcnt<N>() and at_landed(value) before the first use.
template <int OFF>
__device__ __forceinline__ u32x4 at_gload128(const char* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF));
    return v;
}
__device__ __forceinline__ void at_landed(u32x4& v) { asm volatile("" : "+v"(v)); }
template <int N_>
__device__ __forceinline__ void at_wait_lgkm(u32x4& v) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N_));
}
template <int... I, class F>
__device__ __forceinline__ void at_static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N_, class F>
__device__ __forceinline__ void at_static_for(F&& f) {
    at_static_for_impl(std::make_integer_sequence<int, N_>{}, f);
}

// ---------------------------------------------------------------------------------------------
// Long-sequence variant (bf16, N a multiple of 32, thousands of keys per query): same arithmetic as
// attention_kernel, organised for throughput.  8 waves (two per SIMD) x QF*16 queries share every
// 32-key K/V tile, so a tile crosses L2 -> LDS once per 128*QF queries.  Tiles go through a ring of four
// stages and are issued two intervals ahead; waits on the DMA are counted (only the newest pieces may be in
// flight), LDS fragment reads are ordered inline-asm reads with counted waits (a register ring two deep; three
// measured the same).
// (Running the second wave of every SIMD half an interval out of phase, so that one wave's exponentials
// overlap the other's MFMAs, measured no faster.)
// ---------------------------------------------------------------------------------------------
#ifndef ATTN_STREAM_XD
#define ATTN_STREAM_XD 4         // depth of the fragment register ring at 16 queries per wave (at 32 the registers only allow 2)
#endif
#ifndef ATTN_STREAM_SMALL
#define ATTN_STREAM_SMALL 1
#endif
// Diagnostic builds (-DATTN_STAMP): 100-MHz real-time stamps of every workgroup of the streaming kernel, eight slots
// each (start, first tiles landed, end of the key/value tiles, end), into the buffer whose address LDM_DBG_RT_PTR names.
#ifdef ATTN_STAMP
static __device__ unsigned long long* g_attn_rt = nullptr;
#define AT_STAMP(k_) do { if (g_attn_rt && threadIdx.x == 0) g_attn_rt[8 * blockIdx.x + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AT_STAMP(k_) do { } while (0)
#endif
struct AttnFoldArgs {      // AT_FOLD: the constant vector (f32 [D]) and the block's input (keys, residual)
    const float* bias;
    const char* xres;
};
// MODE of the streaming kernel:
//   AT_PLAIN  qkv = [B, N, 3D] (q | k | v), out = O
//   AT_FOLD   the projections folded at pack time (pack_attention): the scores only depend on x_i (Wq Wk^T) x_j^T +
//             (Wk bq) . x_j (terms that are constant along the keys cancel in the soft-max), and P (x Wv + bv) Wo + bo =
//             P (x Wv Wo) + (bv Wo + bo) because the rows of P sum to one.  So qkv = [B, N, 2D] (q' | v') with
//             q' = x (Wq Wk^T) + Wk bq and v' = x (Wv Wo), the KEYS ARE THE BLOCK'S INPUT x ITSELF (fa.xres, [B, N, D]),
//             and out = O + (bv Wo + bo) + x: no key projection, no output projection.
enum { AT_PLAIN = 0, AT_FOLD = 1 };
template <int D, int QF, int NW, int MODE = AT_PLAIN>
__global__ __launch_bounds__(NW * 64, 1) void attention_stream_kernel(const char* __restrict__ qkv, char* __restrict__ out,
                                                                  int N, float scale_log2e, int B, AttnFoldArgs fa) {
    constexpr bool FOLD = MODE == AT_FOLD;
    constexpr int RB = D * 2, KT = 32, KF = 2, KS = RB / 64, DF = D / 16;
    // row strides of the query, key and value matrices
    constexpr int QSTR = FOLD ? 2 * RB : 3 * RB, KSTR = FOLD ? RB : 3 * RB, VSTR = FOLD ? 2 * RB : 3 * RB;
    // ring depth: four stages where they fit (C = 256), else two (C = 512: a stage is 64 KiB)
    constexpr int TILE = KT * RB, NPIECE = TILE / 1024, NST = 4 * 2 * TILE <= 160 * 1024 ? 4 : 2;
    constexpr int QB = NW * QF * 16, PPT = NPIECE / NW, PPW = 2 * PPT;
    static_assert(NPIECE % NW == 0, "every wave issues the same number of pieces");
    static_assert(NST * 2 * TILE <= 160 * 1024, "LDS budget");
    constexpr float RESCALE_LOG2 = 8.0f;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    // Blocks i and i + 8 share an XCD and its L2: all query blocks of one image get ids that are equal modulo 8, so
    // the image's keys and values cross from memory into ONE L2 and the other query blocks read them there (in
    // dispatch order the four query blocks of a 256-token image would sit on four XCDs and fetch K and V four times).
    const int nqb = N / QB;
    const int b = ((int)(blockIdx.x >> 3) / nqb) * 8 + (int)(blockIdx.x & 7);
    if (b >= B) return;                      // (the grid is padded to a multiple of eight images)
    AT_STAMP(0);
    const int q0 = ((int)(blockIdx.x >> 3) % nqb) * QB + wave * QF * 16;
    const char* base = qkv + (size_t)b * N * QSTR;
    const char* kbase = FOLD ? fa.xres + (size_t)b * N * RB : base + RB;
    const char* vbase = base + (FOLD ? RB : 2 * RB);
    const int ntiles = N / KT;

    // DMA pieces of this wave: piece i = wave + NW j covers the RPP rows from row i * RPP of the tile; piece j
    // lies j * NW * RPP rows below piece 0, which changes the row swizzle by an XOR on bits that piece 0's row
    // leaves clear.  One 32-bit lane offset per matrix.
    constexpr int RPP = 1024 / RB, RPJ = NW * RPP;
    static_assert(RPJ <= 16 && (RPJ & (RPJ - 1)) == 0, "row step between the pieces of one wave");
    unsigned koff, voff;
    {
        const int o = wave * 1024 + lane * 16;
        const int r = o / RB, sp = (o - r * RB) >> 4;
        koff = r * KSTR + ((sp ^ (r & 15)) << 4);
        voff = r * VSTR + ((sp ^ ((r & 7) << 1)) << 4);
    }
    auto issue_tile = [&](int t) {
        char* kd = smem + (t & (NST - 1)) * 2 * TILE + wave * 1024;
        const char* ksrc = kbase + (size_t)t * KT * KSTR;
        const char* vsrc = vbase + (size_t)t * KT * VSTR;
        unsigned ko = koff, vo = voff;
        asm volatile("" : "+v"(ko), "+v"(vo));   // opaque: no per-piece 64-bit addresses kept across iterations
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            at_glds16(ksrc + (size_t)j * RPJ * KSTR + (ko ^ ((unsigned)((j * RPJ) & 15) << 4)), kd + j * NW * 1024);
            at_glds16(vsrc + (size_t)j * RPJ * VSTR + (vo ^ ((unsigned)((j * RPJ) & 7) << 5)), kd + TILE + j * NW * 1024);
        }
    };
    constexpr unsigned BIAS_OFF = NST * 2 * TILE;      // FOLD: the constant vector bv Wo + bo behind the ring (D floats)
    if constexpr (FOLD) {
        static_assert(D * 4 <= 2048, "the bias is one DMA piece of one or two waves");
        if (wave < (D * 4 + 1023) / 1024 && wave * 64 + lane < D / 4)
            at_glds16((const char*)(fa.bias + (wave * 64 + lane) * 4), smem + BIAS_OFF + wave * 1024);
    }

    // Issue order of the prologue: the bias, Q, the first tiles, then the residual.  vmcnt retires in issue order, so the
    // wait in front of the first interval leaves the second and third tile (and the residual) in flight.  Q and the
    // residual are loads the compiler does not track (at_gload128): its own wait for a tracked load is vmcnt(0).
    u32x4 qreg[QF][KS];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        const char* row = base + (size_t)(q0 + qf * 16 + l15) * QSTR + g * 16;
        at_static_for<KS>([&](auto kc) { qreg[qf][decltype(kc)::value] = at_gload128<decltype(kc)::value * 64>(row); });
    }
    // FOLD: the epilogue swaps the odd 16-lane rows of accumulator 2p with the even rows of accumulator 2p + 1
    // (v_permlane16_swap, as the convolution kernels do), after which this lane owns the 8 consecutive channels
    // 16 (2p + (g & 1)) + 8 (g >> 1) .. + 7 of query l15: residual loads and output stores are 16 bytes.  At 16 queries
    // per wave and 256 channels the residual is loaded here, BEHIND the first tiles (it is needed last: in front of them
    // every workgroup's first tile queued behind 16 KiB of residual); the other instances have no registers for it
    // and load it in the epilogue.
    constexpr bool XR_LATE = FOLD && QF == 1 && D <= 256;
    constexpr int NXR = DF / 2;                             // 16-byte pieces of a query's residual
    constexpr int NXRV = XR_LATE ? NXR : 0;                 // vector-memory operations behind the prologue tiles
    u32x4 xq[XR_LATE ? NXR : 1];
    const unsigned fold_co = (g & 1) * 16 + (g >> 1) * 8;   // FOLD: first channel of this lane inside a fragment pair
    asm volatile("" ::: "memory");
    constexpr int NPRO = NST >= 3 ? NST - 1 : 1;       // tiles issued before the first interval
#pragma unroll
    for (int t = 0; t < NPRO; ++t)
        if (t < ntiles) issue_tile(t);
    asm volatile("" ::: "memory");
    if constexpr (XR_LATE) {
        const char* xrow = fa.xres + ((size_t)b * N + q0 + l15) * RB + fold_co * 2;
        at_static_for<NXR>([&](auto ic) { xq[decltype(ic)::value] = at_gload128<decltype(ic)::value * 64>(xrow); });
        asm volatile("" ::: "memory");
    }
    f32x4 oacc[DF][QF];
#pragma unroll
    for (int df = 0; df < DF; ++df)
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) oacc[df][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
    float mrun[QF], lrun[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) { mrun[qf] = -1e30f; lrun[qf] = 0.f; }

    f32x4 sacc[KF][QF];
    u32x4 pb[QF];
    // LDS read addresses: everything that depends on the lane in one register per matrix, the fragment
    // index enters by XOR on bits the lane part may also set (computed per read: a table of
    // addresses per fragment would cost the registers the accumulators need) plus an immediate.
    const unsigned klane = l15 * RB + ((((l15 >> 2) << 2) | (g ^ (l15 & 3))) << 4);
    const int tq = l15 >> 2, tp = l15 & 3;
    const int row0 = g * 4 + tq;
    const unsigned vlane = TILE + row0 * RB + ((((row0 & 7) << 1) | (tp >> 1)) << 4) + (tp & 1) * 8;

    // Fragment f = (ks, kf) of the K tile / df of the V tile goes through a register ring XD deep: the read
    // of fragment f + XD - 1 is issued before the wait for fragment f.
    constexpr int XD = QF == 1 ? ATTN_STREAM_XD : 2, NFK = KS * KF;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    auto s_phase = [&](int t, const u32x4 (&breg)[QF][KS]) {
        const unsigned kb = lds0 + (unsigned)(t & (NST - 1)) * 2 * TILE + klane;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
            for (int qf = 0; qf < QF; ++qf) sacc[kf][qf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // piece (ks*4 + g) ^ l15 of row kf*16 + l15
        auto kread = [&](auto fc) {
            constexpr int f = decltype(fc)::value, ks = f >> 1, kf = f & 1;
            return at_lds_read128<(ks >> 2) * 256 + kf * 16 * RB>(kb ^ ((ks & 3) << 6));
        };
        u32x4 kr[XD];
        at_static_for<XD - 1>([&](auto fc) { kr[decltype(fc)::value] = kread(fc); });
        at_static_for<NFK>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f + XD - 1 < NFK) kr[(f + XD - 1) % XD] = kread(std::integral_constant<int, f + XD - 1>{});
            at_wait_lgkm<(NFK - 1 - f < XD - 1 ? NFK - 1 - f : XD - 1)>(kr[f % XD]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf)
                sacc[f & 1][qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, kr[f % XD]), __builtin_bit_cast(bf16x8, breg[qf][f >> 1]), sacc[f & 1][qf], 0, 0, 0);
        });
    };
    // lane holds keys kf*16 + g*4 + r of query column l15; statistics in the log2 domain
    auto softmax = [&]() {
        float alpha[QF];
        bool moved = false;
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float mx = -1e30f;
#pragma unroll
            for (int kf = 0; kf < KF; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kf][qf][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            mx *= scale_log2e;                       // the scale is positive: max commutes with it
            const bool move = mx > mrun[qf] + RESCALE_LOG2;
            const float mnew = move ? mx : mrun[qf];
            alpha[qf] = move ? __builtin_amdgcn_exp2f(mrun[qf] - mnew) : 1.0f;
            moved |= move;
            mrun[qf] = mnew;
            float p[KF][4];
            float rs = 0.f;
#pragma unroll
            for (int kf = 0; kf < KF; ++kf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[kf][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kf][qf][r], scale_log2e, -mnew));
                    rs += p[kf][r];
                }
            lrun[qf] = lrun[qf] * alpha[qf] + rs;    // per-lane partial; the 4 key groups are summed at the end
            pb[qf][0] = at_pack2v(p[0][0], p[0][1]);
            pb[qf][1] = at_pack2v(p[0][2], p[0][3]);
            pb[qf][2] = at_pack2v(p[1][0], p[1][1]);
            pb[qf][3] = at_pack2v(p[1][2], p[1][3]);
        }
        if (__builtin_amdgcn_ballot_w64(moved) != 0ull) {      // wave-uniform, rare after the first tiles
#pragma unroll
            for (int df = 0; df < DF; ++df)
#pragma unroll
                for (int qf = 0; qf < QF; ++qf) oacc[df][qf] *= alpha[qf];
        }
    };
    auto pv_phase = [&](int t) {
        const unsigned vb = lds0 + (unsigned)(t & (NST - 1)) * 2 * TILE + vlane;
        // piece (df*2 + (tp>>1)) ^ ((row & 7) << 1) of rows row0 and row0 + 16: two transposing reads
        auto vread = [&](auto dc) {
            constexpr int df = decltype(dc)::value;
            const unsigned a0 = vb ^ ((df & 7) << 5);
            const uint2 v0 = at_lds_read_tr16<(df >> 3) * 256>(a0);
            const uint2 v1 = at_lds_read_tr16<(df >> 3) * 256 + 16 * RB>(a0);
            return u32x4{v0.x, v0.y, v1.x, v1.y};
        };
        u32x4 vr[XD];
        at_static_for<XD - 1>([&](auto dc) { vr[decltype(dc)::value] = vread(dc); });
        at_static_for<DF>([&](auto dc) {
            constexpr int df = decltype(dc)::value;
            if constexpr (df + XD - 1 < DF) vr[(df + XD - 1) % XD] = vread(std::integral_constant<int, df + XD - 1>{});
            at_wait_lgkm<2 * (DF - 1 - df < XD - 1 ? DF - 1 - df : XD - 1)>(vr[df % XD]);
#pragma unroll
            for (int qf = 0; qf < QF; ++qf)
                oacc[df][qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, vr[df % XD]), __builtin_bit_cast(bf16x8, pb[qf]), oacc[df][qf], 0, 0, 0);
        });
    };
    // End of interval t: every wave is past barrier t-1, so stage (t-1) % 4 is free for tile t+3.  The barrier that
    // follows publishes tile t+1, the next one to be read: this wave waits for its pieces of THAT tile and leaves the
    // pieces of tiles t+2 and t+3 in flight (a key/value tile is PPW pieces per wave), so a tile has
    // two intervals to land.  (Waiting for tile t+2 here as well made every interval one memory latency long: 1.3 us
    // per 32 keys at 256 tokens, where an interval's arithmetic is a quarter of that.)
    // Only in the four-wave instance (one wave per SIMD, short sequences): the eight-wave instances are bound by their
    // arithmetic, and with two tiles in flight they measured 0.8 % slower (1 600 against 1 588 us at 16 384 tokens).
    constexpr bool LEAVE2 = NW == 4;
    auto tail = [&](int t) {
        const int t2 = t + NST - 2, t3 = t + NST - 1;            // the two youngest tiles that may exist
        if constexpr (!LEAVE2) {
            if (t3 < ntiles) {
                issue_tile(t3);
                at_wait_vmcnt<PPW>();
            } else {
                at_wait_vmcnt<0>();
            }
        } else if (t3 < ntiles) {
            issue_tile(t3);
            // (the late residual loads sit between the prologue tiles and tile NPRO: while tile t+1 is a prologue tile
            // they are younger than it and may stay in flight too)
            if (NXRV > 0 && t + 1 < NPRO) at_wait_vmcnt<2 * PPW + NXRV>();
            else at_wait_vmcnt<2 * PPW>();
        } else if (t2 < ntiles) {
            at_wait_vmcnt<PPW>();
        } else {
            at_wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
    };

    // the first tile (and everything older); the younger tiles of the prologue stay in flight
    if (NST >= 3 && ntiles >= NPRO) at_wait_vmcnt<(NPRO - 1) * PPW + NXRV>();
    else at_wait_vmcnt<0>();
#pragma unroll
    for (int qf = 0; qf < QF; ++qf)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) at_landed(qreg[qf][ks]);     // (Q is older than every tile)
    __builtin_amdgcn_s_barrier();
    AT_STAMP(1);

    if constexpr (NST >= 3) {
        for (int t = 0; t < ntiles; ++t) {
            s_phase(t, qreg);
            softmax();
            pv_phase(t);
            tail(t);
            if (t < 3) AT_STAMP(2 + t);
        }
        AT_STAMP(5);
    } else {
        // two stages: tile t+1 goes into the stage that barrier t-1 freed, at the START of interval t, and has the
        // whole interval to land
        for (int t = 0; t < ntiles; ++t) {
            if (t + 1 < ntiles) issue_tile(t + 1);
            s_phase(t, qreg);
            softmax();
            pv_phase(t);
            at_wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- normalise and store: lane holds d = df*16 + g*4 + (0..3) of query q0 + qf*16 + l15 ----
    if constexpr (FOLD) {
        // out = O / l + (bv Wo + bo) + x, in f32, rounded once; 16-byte write-through stores
        if constexpr (XR_LATE) {
            at_wait_vmcnt<0>();
#pragma unroll
            for (int i = 0; i < NXR; ++i) at_landed(xq[i]);
        }
#pragma unroll
        for (int qf = 0; qf < QF; ++qf) {
            float l = lrun[qf];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float inv = 1.0f / l;
            int qi = q0 + qf * 16 + l15;
            asm volatile("" : "+v"(qi));     // the addresses are formed here, not ahead of the tile loop (where they spill)
            const size_t ro = ((size_t)b * N + qi) * RB + fold_co * 2;
            // (residual loaded here: four pairs at a time, the accumulators leave no room for all of them)
            constexpr int GRP = XR_LATE ? NXR : 4;
#pragma unroll
            for (int p0 = 0; p0 < DF / 2; p0 += GRP) {
                u32x4 xl[GRP];
                if constexpr (!XR_LATE) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) xl[i] = *(const u32x4*)(fa.xres + ro + (p0 + i) * 64);
                }
#pragma unroll
                for (int i = 0; i < GRP; ++i) {
                    const int p = p0 + i;
                    f32x4 v0, v1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned int, oacc[2 * p][qf][r] * inv),
                                                                         __builtin_bit_cast(unsigned int, oacc[2 * p + 1][qf][r] * inv),
                                                                         false, false);
                        v0[r] = at_u2f(sw[0]);
                        v1[r] = at_u2f(sw[1]);
                    }
                    const char* bp = smem + BIAS_OFF + (p * 32 + fold_co) * 4;
                    v0 += *(const f32x4*)bp;
                    v1 += *(const f32x4*)(bp + 16);
                    const u32x4 r = XR_LATE ? xq[XR_LATE ? p : 0] : xl[i];
                    v0[0] += at_u2f(r[0] << 16); v0[1] += at_u2f(r[0] & 0xffff0000u);
                    v0[2] += at_u2f(r[1] << 16); v0[3] += at_u2f(r[1] & 0xffff0000u);
                    v1[0] += at_u2f(r[2] << 16); v1[1] += at_u2f(r[2] & 0xffff0000u);
                    v1[2] += at_u2f(r[3] << 16); v1[3] += at_u2f(r[3] & 0xffff0000u);
                    const u32x4 w = {at_pack2(v0[0], v0[1]), at_pack2(v0[2], v0[3]), at_pack2(v1[0], v1[1]), at_pack2(v1[2], v1[3])};
                    at_store_out(out + ro + p * 64, w);
                }
            }
        }
#ifdef ATTN_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        AT_STAMP(6);
#endif
        return;
    }
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
        float l = lrun[qf];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        char* orow = out + ((size_t)b * N + q0 + qf * 16 + l15) * RB;
#pragma unroll
        for (int df = 0; df < DF; ++df) {
            const f32x4 v = oacc[df][qf] * inv;
            uint2 w;
            w.x = at_pack2(v[0], v[1]);
            w.y = at_pack2(v[2], v[3]);
            *(uint2*)(orow + (df * 16 + g * 4) * 2) = w;
        }
    }
}

template <int D, int QF, int NW, int MODE = AT_PLAIN>
static int launch_attn_stream(const void* qkv, void* out, int B, int N, hipStream_t stream, AttnFoldArgs fa = AttnFoldArgs{}) {
    constexpr int QB = NW * QF * 16;
    constexpr size_t stage = (size_t)2 * 32 * D * 2;
    constexpr size_t lds = (4 * stage <= 160 * 1024 ? 4 : 2) * stage + (MODE != AT_PLAIN ? 2048 : 0);
    auto kern = attention_stream_kernel<D, QF, NW, MODE>;
    static bool lds_set[kMaxDevices] = {};   // per kernel (instantiation) and device
    LDM_TRY(allow_dynamic_lds((const void*)kern, 160 * 1024, lds_set));
    LDM_REQUIRE(N % QB == 0, "token count must be a multiple of the query block");
#ifdef ATTN_STAMP
    if (const char* e = getenv("LDM_DBG_RT_PTR")) {
        unsigned long long* p = (unsigned long long*)strtoull(e, nullptr, 16);
        LDM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_rt), &p, sizeof(p)));
    }
#endif
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
    dim3 grid((unsigned)((N / QB) * ((B + 7) / 8 * 8)));
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, (const char*)qkv, (char*)out, N, scale_log2e, B, fa);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

template <typename T, int D, int QF, int KF, int MINB = 1>
static int launch_attn_cfg(const void* qkv, void* out, int B, int N, hipStream_t stream) {
    constexpr int TILE = KF * 16 * D * (int)sizeof(T);
    constexpr int QB = 4 * QF * 16;
    const size_t lds = (size_t)4 * TILE;
    static_assert(4 * TILE <= 160 * 1024, "LDS budget");
    auto kern = attention_kernel<T, D, QF, KF, MINB>;
    static bool lds_set[kMaxDevices] = {};   // per kernel (instantiation) and device
    LDM_TRY(allow_dynamic_lds((const void*)kern, 160 * 1024, lds_set));
    const float scale_log2e = 1.4426950408889634f / sqrtf((float)D);
    dim3 grid((unsigned)((N + QB - 1) / QB), (unsigned)B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, (const char*)qkv, (char*)out, N, scale_log2e);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// The shapes the folded form serves: bf16 and the token counts that take an instance of the streaming kernel.
// LDM_ATTN_FOLD=0 (developer switch, read at every call like LDM_DENSE4) keeps the unfolded projections, which gives the
// parity tests a second GPU implementation of the block to compare with.
static int fold_instance(int B, int N, int C) {
    if (C == 256) {
        if (N % 256 == 0 && N >= 1024 && (long)B * (N / 256) >= 256) return 1;
        if (N % 64 == 0 && (long)B * ((N + 127) / 128) < 256) return 2;
    } else if (C == 512) {
        if (N % 128 == 0 && N >= 512 && (long)B * (N / 128) >= 256) return 3;
    }
    return 0;
}
bool attention_can_fold(int dtype, int B, int N, int C) {
    const char* e = getenv("LDM_ATTN_FOLD");
    return !(e && e[0] == '0') && dtype == LDM_BF16 && B > 0 && B <= 65535 && N > 0 && fold_instance(B, N, C) != 0;
}
int launch_attention_folded(const void* qv, const void* x, const float* bias, void* out, int B, int N, int C, hipStream_t stream) {
    LDM_REQUIRE(qv && x && bias && out && B > 0 && N > 0, "bad arguments");
    LDM_REQUIRE(attention_can_fold(LDM_BF16, B, N, C), "shape not served by the folded attention kernel");
    AttnFoldArgs fa{bias, (const char*)x};
    switch (fold_instance(B, N, C)) {
        case 1: return launch_attn_stream<256, 2, 8, AT_FOLD>(qv, out, B, N, stream, fa);
        case 2: return launch_attn_stream<256, 1, 4, AT_FOLD>(qv, out, B, N, stream, fa);
        default: return launch_attn_stream<512, 1, 8, AT_FOLD>(qv, out, B, N, stream, fa);
    }
}

int launch_attention(int dtype, const void* qkv, void* out, int B, int N, int C, hipStream_t stream) {
    LDM_REQUIRE(qkv && out && B > 0 && N > 0, "bad arguments");
    LDM_REQUIRE(B <= 65535, "batch too large for one launch");
    if (dtype == LDM_BF16) {
        switch (C) {
            case 128: return launch_attn_cfg<bf16_t, 128, 2, 2>(qkv, out, B, N, stream);
            case 256:
                // 16 queries per wave while that still leaves every CU a workgroup; O then stays in VGPRs
                if (N % 256 == 0 && N >= 1024 && (long)B * (N / 256) >= 256) return launch_attn_stream<256, 2, 8>(qkv, out, B, N, stream);
#if ATTN_STREAM_SMALL
                if (N % 64 == 0 && (long)B * ((N + 127) / 128) < 256) return launch_attn_stream<256, 1, 4>(qkv, out, B, N, stream);
#endif
                if ((long)B * ((N + 127) / 128) < 256) return launch_attn_cfg<bf16_t, 256, 1, 2, 2>(qkv, out, B, N, stream);
                return launch_attn_cfg<bf16_t, 256, 2, 2>(qkv, out, B, N, stream);
            case 512:
                if (N % 128 == 0 && N >= 512 && (long)B * (N / 128) >= 256) return launch_attn_stream<512, 1, 8>(qkv, out, B, N, stream);
                return launch_attn_cfg<bf16_t, 512, 1, 2>(qkv, out, B, N, stream);
        }
    } else {
        switch (C) {
            case 64: return launch_attn_cfg<float, 64, 2, 2>(qkv, out, B, N, stream);
            case 128: return launch_attn_cfg<float, 128, 2, 2>(qkv, out, B, N, stream);
            case 256:
                // few images: 64 queries per workgroup instead of 128 doubles the workgroups (8 -> 16 at 4 images of 256 tokens)
                if ((long)B * ((N + 127) / 128) < 256) return launch_attn_cfg<float, 256, 1, 1>(qkv, out, B, N, stream);
                return launch_attn_cfg<float, 256, 2, 1>(qkv, out, B, N, stream);
            case 512: return launch_attn_cfg<float, 512, 1, 1>(qkv, out, B, N, stream);
        }
    }
    LDM_FAIL(LDM_ERR_INVALID, "attention supports 128, 256 or 512 channels (and 64 in f32)");
}

}  // namespace ldm
