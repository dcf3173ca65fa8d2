// Tap convolution on MFMA for gfx950: implicit GEMM over NHWC activations.
//
// One kernel family serves every dense contraction of the UNet that has a spatial structure:
//   conv3x3 stride 1 / stride 2 (ResidualBlock convs, down1..3, out_conv; src/main.rs:348-366,423-452,486-492),
//   the four sub-pixel phases of convT 4x4 / stride 2 (up1, up2; src/main.rs:460-480),
//   conv1x1 over two channel-concatenated sources (up_conv1/2 on cat[up, skip]; src/main.rs:468-483,516-522)
//   and the token Linears of Attention (question/key/val/out; src/main.rs:249-252) as conv1x1.
//
// Work decomposition.  A workgroup (4 waves) owns TH x 16 output pixels of one image and BN output
// channels.  The input window of the tile ("halo", (TH-1)*S+KEXT rows x (15*S+KEXT) pixels) is staged
// into LDS once per 128-byte channel chunk (64 bf16 / 32 f32 channels) by LDS-DMA (global_load_lds,
// 16 B per lane, zero page for out-of-image pixels) and then re-used by all taps; the weight slab of
// one (chunk, tap) -- BN rows of 128 bytes -- is streamed through a two-deep LDS ring.  Every LDS row
// is 128 bytes; the 16-byte piece p of row r lives at piece p ^ (r & 7) so that ds_read_b128 of
// 16 consecutive rows is bank-conflict free (weights are stored pre-swizzled, halo loads swizzle the
// source address, because the DMA writes LDS linearly).
//
// MFMA orientation: D^T = W * X^T, i.e. the A operand is the weight slab (row = output channel) and
// the B operand the pixels (column = pixel).  A lane then owns 4 consecutive output channels of one
// pixel per 16x16 accumulator, which makes the NHWC epilogue store 8 (bf16) or 16 (f32) bytes wide.
// bf16 uses v_mfma_f32_16x16x32_bf16; f32 uses v_mfma_f32_16x16x4_f32 (exact f32 FMA chain) four times
// per 16-byte fragment, with the K permutation applied identically to both operands.
#include "common.h"

namespace ldm {

static __device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

struct TapConvArgs {
    const char* in0;
    const char* in1;
    const char* w;
    const float* bias;
    const char* res;
    const float* tvec;
    const int* step_ptr;
    char* out;
    int nch0, nch1;              // 128-byte channel chunks of each source
    int cin0_bytes, cin1_bytes;  // bytes per pixel of each source
    int B, Hin, Win;
    int Hg, Wg;                  // tile grid extent in output pixels per phase (Hout/osc, Wout/osc)
    int Hout, Wout, cout;
    int osc;                     // output scale: 2 for convT phases, else 1
    int ntaps;
    unsigned long long tapcode[4];  // per phase: 4 bits per tap = dy | dx << 2 (window coordinates)
    int tiles_x, tiles_y, ntiles, nblk_n;
    int act, out_f32;
    int t_img_stride, t_step_stride;
    long w_phase_stride;         // bytes
};

template <typename T> struct Elem;
template <> struct Elem<float> { static constexpr int CK = 32; };
template <> struct Elem<bf16_t> { static constexpr int CK = 64; };

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

__device__ __forceinline__ bf16_t f2bf(float f) {
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
    return __builtin_bit_cast(float, (unsigned int)h << 16);
}

// by-value helper: bit_cast applied directly to a vector-element expression reads element 0
__device__ __forceinline__ float u2f(unsigned int u) { return __builtin_bit_cast(float, u); }

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// TH: tile rows; BN: output channels per workgroup; WM x WN = 4 waves over (pixel rows, channels);
// S: stride; KEXT: window extent (3 for 3x3 and the convT phases, 1 for 1x1); MINW: waves per SIMD.
template <typename T, int TH, int BN, int WM, int WN, int S, int KEXT, int MINW>
__global__ __launch_bounds__(256, MINW) void tapconv_kernel(const TapConvArgs a) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int TW = 16;
    constexpr int HH = (TH - 1) * S + KEXT;
    constexpr int HW = (TW - 1) * S + KEXT;
    constexpr int HP = HH * HW;
    constexpr int NI = (HP + 7) / 8;            // 1-KiB DMA pieces per halo chunk
    constexpr int HALO_BYTES = NI * 1024;
    constexpr int WSLAB = BN * 128;
    constexpr int MF = TH / WM;                  // pixel fragments (tile rows) per wave
    constexpr int NF = BN / 16 / WN;             // channel fragments per wave
    constexpr int PAD = KEXT / 2;
    static_assert(TH % WM == 0 && (BN / 16) % WN == 0, "wave tiling");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, l15 = lane & 15;

    // XCD-aware block -> (tile, channel block): blocks b and b+8 share an XCD (and its L2), so the
    // channel blocks of one tile, which read the same input window, get ids 8 apart.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int tile = (j / a.nblk_n) * 8 + xcd;
    const int nb = j % a.nblk_n;
    if (tile >= a.ntiles) return;
    const int phase = blockIdx.z;
    const int img = tile / (a.tiles_x * a.tiles_y);
    const int trem = tile - img * (a.tiles_x * a.tiles_y);
    const int y0 = (trem / a.tiles_x) * TH, x0 = (trem % a.tiles_x) * TW;
    const int n0 = nb * BN;
    const int iy0 = y0 * S - PAD, ix0 = x0 * S - PAD;
    const int nchunks = a.nch0 + a.nch1;
    const int ntaps = a.ntaps;
    const int nsteps = nchunks * ntaps;

    char* hbuf0 = smem;
    char* wbuf0 = smem + (nchunks > 1 ? 2 : 1) * HALO_BYTES;

    const char* in0_img = a.in0 + (size_t)img * a.Hin * a.Win * a.cin0_bytes;
    const char* in1_img = a.in1 ? a.in1 + (size_t)img * a.Hin * a.Win * a.cin1_bytes : nullptr;
    const char* wbase = a.w + (size_t)phase * a.w_phase_stride + (size_t)n0 * 128;
    const size_t wslab_stride = (size_t)a.cout * 128;   // bytes between consecutive (chunk, tap) slabs
    const char* zero = (const char*)g_zero16;

    auto issue_halo = [&](int c, int first, int stride) {
        const char* base = c < a.nch0 ? in0_img + c * 128 : in1_img + (c - a.nch0) * 128;
        const int pix_bytes = c < a.nch0 ? a.cin0_bytes : a.cin1_bytes;
        char* dst = hbuf0 + (c & 1) * HALO_BYTES;
        for (int i = first; i < NI; i += stride) {
            const int hp = i * 8 + (lane >> 3);
            const int slot = lane & 7;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = hp < HP && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const char* src = ok ? base + ((size_t)(iy * a.Win + ix) * pix_bytes + ((slot ^ (hp & 7)) << 4)) : zero;
            glds16(src, dst + i * 1024);
        }
    };
    auto issue_w = [&](int s) {
        const char* src = wbase + (size_t)s * wslab_stride + lane * 16;
        char* dst = wbuf0 + (s & 1) * WSLAB;
#pragma unroll
        for (int i = 0; i < BN / 8 / 4; ++i) glds16(src + (wave + i * 4) * 1024, dst + (wave + i * 4) * 1024);
    };

    // per-lane LDS read offsets that do not change over the K loop
    int wrow_off[NF];   // weight rows
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) wrow_off[nf] = ((wn * NF + nf) * 16 + l15) * 128;
    int hp_base[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) hp_base[mf] = ((wm * MF + mf) * S) * HW + l15 * S;

    f32x4 acc[NF][MF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_halo(0, wave, 4);
    issue_w(0);
    __syncthreads();

    const unsigned long long tapcode = a.tapcode[phase];
    int c = 0, t = 0;
    for (int s = 0; s < nsteps; ++s) {
        if (s + 1 < nsteps) issue_w(s + 1);
        if (c + 1 < nchunks) issue_halo(c + 1, t + ntaps * wave, ntaps * 4);

        const char* hb = hbuf0 + (c & 1) * HALO_BYTES;
        const char* wb = wbuf0 + (s & 1) * WSLAB;
        const unsigned int code = (unsigned int)(tapcode >> (4 * t)) & 15u;
        const int tap_off = (code & 3) * HW + (code >> 2);

#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int piece = kk * 4 + g;
            u32x4 wf[NF], xf[MF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                wf[nf] = *(const u32x4*)(wb + wrow_off[nf] + ((piece ^ (l15 & 7)) << 4));
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int hp = hp_base[mf] + tap_off;
                xf[mf] = *(const u32x4*)(hb + hp * 128 + ((piece ^ (hp & 7)) << 4));
            }
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    if constexpr (sizeof(T) == 2) {
                        acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, wf[nf]), __builtin_bit_cast(bf16x8, xf[mf]), acc[nf][mf], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                u2f(wf[nf][q]), u2f(xf[mf][q]), acc[nf][mf], 0, 0, 0);
                    }
                }
        }
        __syncthreads();
        if (++t == ntaps) { t = 0; ++c; }
    }

    // ---- epilogue: + bias, (+ residual), activation, (+ time vector), store -------------------
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const float* tv = a.tvec ? a.tvec + (size_t)img * a.t_img_stride + (size_t)step * a.t_step_stride : nullptr;
    const int py = phase >> 1, px = phase & 1;
    const int gx = x0 + l15;
    const int ox = gx * a.osc + px;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int gy = y0 + wm * MF + mf;
        if (gy >= a.Hg || gx >= a.Wg) continue;
        const int oy = gy * a.osc + py;
        const size_t pix = ((size_t)img * a.Hout + oy) * a.Wout + ox;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int co = n0 + (wn * NF + nf) * 16 + g * 4;
            const f32x4 b4 = *(const f32x4*)(a.bias + co);
            f32x4 v = acc[nf][mf] + b4;
            const size_t eo = pix * a.cout + co;
            if (a.res) {
                if constexpr (sizeof(T) == 2) {
                    const uint2 r = *(const uint2*)(a.res + eo * 2);
                    v[0] += bf2f((bf16_t)(r.x & 0xffffu));
                    v[1] += bf2f((bf16_t)(r.x >> 16));
                    v[2] += bf2f((bf16_t)(r.y & 0xffffu));
                    v[3] += bf2f((bf16_t)(r.y >> 16));
                } else {
                    v += *(const f32x4*)(a.res + eo * 4);
                }
            }
            if (a.act == ACT_SILU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = silu_f(v[q]);
            } else if (a.act == ACT_RELU) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if (tv) v += *(const f32x4*)(tv + co);
            if (sizeof(T) == 4 || a.out_f32) {
                *(f32x4*)(a.out + eo * 4) = v;
            } else {
                uint2 o;
                o.x = (unsigned int)f2bf(v[0]) | ((unsigned int)f2bf(v[1]) << 16);
                o.y = (unsigned int)f2bf(v[2]) | ((unsigned int)f2bf(v[3]) << 16);
                *(uint2*)(a.out + eo * 2) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
int conv_out_hw(int kind, int Hin, int Win, int* Hout, int* Wout) {
    switch (kind) {
        case CONV3x3_S1:
        case CONV1x1: *Hout = Hin; *Wout = Win; return LDM_OK;
        case CONV3x3_S2: *Hout = (Hin - 1) / 2 + 1; *Wout = (Win - 1) / 2 + 1; return LDM_OK;
        case CONVT4x4_S2: *Hout = Hin * 2; *Wout = Win * 2; return LDM_OK;
    }
    LDM_FAIL(LDM_ERR_INVALID, "unknown conv kind");
}

template <typename T, int TH, int BN, int WM, int WN, int S, int KEXT, int MINW>
static int launch_cfg(TapConvArgs& a, int nphase, hipStream_t stream) {
    constexpr int HH = (TH - 1) * S + KEXT, HW = 15 * S + KEXT;
    constexpr int NI = (HH * HW + 7) / 8;
    const int nchunks = a.nch0 + a.nch1;
    const size_t lds = (size_t)(nchunks > 1 ? 2 : 1) * NI * 1024 + 2 * (size_t)BN * 128;
    a.tiles_x = (a.Wg + 15) / 16;
    a.tiles_y = (a.Hg + TH - 1) / TH;
    a.ntiles = a.tiles_x * a.tiles_y * a.B;
    a.nblk_n = a.cout / BN;
    auto kern = tapconv_kernel<T, TH, BN, WM, WN, S, KEXT, MINW>;
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        LDM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    dim3 grid((unsigned)(((a.ntiles + 7) / 8) * 8 * a.nblk_n), 1, (unsigned)nphase);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// Picks the tile height: the tallest tile that still yields at least `want` workgroups.
static int pick_th(long tiles16, long tiles8, int nblk_n, int nphase, long want, bool allow16) {
    if (allow16 && tiles16 * nblk_n * nphase >= want) return 16;
    if (tiles8 * nblk_n * nphase >= want) return 8;
    return 4;
}

template <typename T>
static int launch_typed(const PackedConv& pc, const ConvCall& cc, TapConvArgs& a, hipStream_t stream) {
    const int bn = (a.cout % 128 == 0) ? 128 : 64;
    const int nblk_n = a.cout / bn;
    const long tx = (a.Wg + 15) / 16;
    const long t16 = tx * ((a.Hg + 15) / 16) * a.B, t8 = tx * ((a.Hg + 7) / 8) * a.B;
    const int nph = pc.nphase;
    if (pc.kind == CONV3x3_S2) {
        if (bn == 128) return launch_cfg<T, 4, 128, 2, 2, 2, 3, 1>(a, nph, stream);
        return launch_cfg<T, 4, 64, 2, 2, 2, 3, 1>(a, nph, stream);
    }
    if (pc.kind == CONV1x1) {
        const int th = pick_th(t16, t8, nblk_n, nph, 256, true);
        if (bn == 128) {
            if (th == 16) return launch_cfg<T, 16, 128, 2, 2, 1, 1, 1>(a, nph, stream);
            if (th == 8) return launch_cfg<T, 8, 128, 2, 2, 1, 1, 1>(a, nph, stream);
            return launch_cfg<T, 4, 128, 2, 2, 1, 1, 1>(a, nph, stream);
        }
        if (th == 16) return launch_cfg<T, 16, 64, 4, 1, 1, 1, 2>(a, nph, stream);
        if (th == 8) return launch_cfg<T, 8, 64, 2, 2, 1, 1, 2>(a, nph, stream);
        return launch_cfg<T, 4, 64, 2, 2, 1, 1, 2>(a, nph, stream);
    }
    // 3x3-window kinds: conv3x3 stride 1 and the convT phases
    const int th = pick_th(t16, t8, nblk_n, nph, bn == 64 ? 512 : 256, true);
    if (bn == 128) {
        if (th == 16) return launch_cfg<T, 16, 128, 2, 2, 1, 3, 1>(a, nph, stream);
        if (th == 8) return launch_cfg<T, 8, 128, 2, 2, 1, 3, 1>(a, nph, stream);
        return launch_cfg<T, 4, 128, 2, 2, 1, 3, 1>(a, nph, stream);
    }
    if (th == 16) return launch_cfg<T, 16, 64, 4, 1, 1, 3, 2>(a, nph, stream);
    if (th == 8) return launch_cfg<T, 8, 64, 2, 2, 1, 3, 2>(a, nph, stream);
    return launch_cfg<T, 4, 64, 2, 2, 1, 3, 2>(a, nph, stream);
}

int launch_tapconv(const PackedConv& pc, const ConvCall& cc, hipStream_t stream) {
    const int es = pc.dtype == LDM_BF16 ? 2 : 4;
    const int ck = 128 / es;
    LDM_REQUIRE(cc.in0 && cc.out && pc.w && pc.bias, "null tensor");
    LDM_REQUIRE(cc.cin0 % ck == 0 && cc.cin1 % ck == 0 && cc.cin0 + cc.cin1 == pc.cin, "input channels do not match the packed conv");
    LDM_REQUIRE(pc.cout % 64 == 0, "output channels must be padded to a multiple of 64");
    LDM_REQUIRE(cc.B > 0 && cc.Hin > 0 && cc.Win > 0, "empty input");
    LDM_REQUIRE((cc.cin1 == 0) == (cc.in1 == nullptr), "second source and its channel count disagree");
    LDM_REQUIRE((size_t)cc.Hin * cc.Win * (size_t)(cc.cin0 > cc.cin1 ? cc.cin0 : cc.cin1) * es < (1ull << 31), "image too large");
    TapConvArgs a{};
    a.in0 = (const char*)cc.in0; a.in1 = (const char*)cc.in1;
    a.w = (const char*)pc.w; a.bias = pc.bias;
    a.res = (const char*)cc.res; a.tvec = cc.tvec; a.step_ptr = cc.step_ptr;
    a.out = (char*)cc.out;
    a.nch0 = cc.cin0 / ck; a.nch1 = cc.cin1 / ck;
    a.cin0_bytes = cc.cin0 * es; a.cin1_bytes = cc.cin1 * es;
    a.B = cc.B; a.Hin = cc.Hin; a.Win = cc.Win;
    LDM_TRY(conv_out_hw(pc.kind, cc.Hin, cc.Win, &a.Hout, &a.Wout));
    a.osc = pc.kind == CONVT4x4_S2 ? 2 : 1;
    a.Hg = a.Hout / a.osc; a.Wg = a.Wout / a.osc;
    a.cout = pc.cout;
    a.ntaps = pc.ntaps;
    a.act = cc.act; a.out_f32 = (pc.dtype == LDM_F32) ? 1 : cc.out_f32;
    a.t_img_stride = cc.t_img_stride; a.t_step_stride = cc.t_step_stride;
    a.w_phase_stride = (long)(pc.cin / ck) * pc.ntaps * pc.cout * 128;
    for (int p = 0; p < 4; ++p) a.tapcode[p] = 0;
    if (pc.kind == CONV3x3_S1 || pc.kind == CONV3x3_S2) {
        unsigned long long code = 0;
        for (int t = 0; t < 9; ++t) code |= (unsigned long long)((t / 3) | ((t % 3) << 2)) << (4 * t);
        a.tapcode[0] = code;
    } else if (pc.kind == CONVT4x4_S2) {
        // phase (py,px), tap (ta,tb): window row = py + ta, window col = px + tb
        // (input row iy = i - 1 + py + ta pairs with kernel row ky = 3 - py - 2*ta; see pack_convT)
        for (int p = 0; p < 4; ++p) {
            const int py = p >> 1, px = p & 1;
            unsigned long long code = 0;
            for (int t = 0; t < 4; ++t) code |= (unsigned long long)((py + (t >> 1)) | ((px + (t & 1)) << 2)) << (4 * t);
            a.tapcode[p] = code;
        }
    }
    if (pc.dtype == LDM_BF16) return launch_typed<bf16_t>(pc, cc, a, stream);
    return launch_typed<float>(pc, cc, a, stream);
}

}  // namespace ldm
