// Tap convolution on MFMA for gfx950: implicit GEMM over NHWC activations.
//
// One kernel family serves every dense contraction of the UNet that has a spatial structure:
//   conv3x3 stride 1 / stride 2 (ResidualBlock convs, down1..3, out_conv; src/main.rs:348-366,423-452,486-492),
//   the four sub-pixel phases of convT 4x4 / stride 2 (up1, up2; src/main.rs:460-480),
//   conv1x1 over two channel-concatenated sources (up_conv1/2 on cat[up, skip]; src/main.rs:468-483,516-522)
//   and the token Linears of Attention (question/key/val/out; src/main.rs:249-252) as conv1x1.
//
// Work decomposition.  A workgroup (4 or 8 waves) owns TH x 16 output pixels of one image and BN
// output channels.  The input window of the tile ("halo", (TH-1)*S+KEXT rows x (15*S+KEXT) pixels) is
// staged into LDS once per 128-byte channel chunk (64 bf16 / 32 f32 channels) by LDS-DMA
// (global_load_lds, 16 B per lane, zero page for out-of-image pixels) and then re-used by all taps;
// the weight slab of one (chunk, tap) -- BN rows of 128 bytes -- is streamed through a three-deep
// LDS ring, two steps ahead of its use.
//
// LDS layout ("pair planes").  The 128 bytes of a row (pixel or output channel) are eight 16-byte
// pieces; piece p lives in plane p >> 1 at row * 32 + (p & 1) * 16.  An MFMA fragment read
// (ds_read_b128: 16 rows x the four pieces kk*4 + lane/16) then touches 16 distinct 16-byte slots of
// every 256-byte bank row, i.e. it is conflict free with NO address swizzle, and every address is
// one per-lane base plus compile-time immediates: row offsets of taps and fragments are multiples of
// 32 bytes, the second half step is two planes further.  A DMA piece (1 KiB) fills 32 rows of one
// plane; its 64 lanes read 32 bytes from each of 32 consecutive pixels (or channels; the packed
// weights are stored plane-major so that this is a linear copy).
//
// Synchronisation.  One raw s_barrier per (chunk, tap) step.  Before it every wave waits with a
// COUNTED s_waitcnt vmcnt(N) only for the DMA pieces the next step reads (its weight slab, and the
// next chunk's halo when that chunk starts), so the slab two steps ahead and a freshly issued halo
// stay in flight across the barrier.  Every wave issues the same number of pieces per slab (NW) and per
// halo (NHW, padded with zero-page pieces into a dump area), which makes the counts constants.
// Inside a step the LDS reads of the next half step are interleaved with the MFMAs of this one
// (fragments double-buffered in registers).
//
// MFMA orientation: D^T = W * X^T, i.e. the A operand is the weight slab (row = output channel) and
// the B operand the pixels (column = pixel).  A lane then owns 4 consecutive output channels of one
// pixel per 16x16 accumulator, which makes the NHWC epilogue store 8 (bf16) or 16 (f32) bytes wide.
// bf16 uses v_mfma_f32_16x16x32_bf16; f32 uses v_mfma_f32_16x16x4_f32 (exact f32 FMA chain) four times
// per 16-byte fragment, with the K permutation applied identically to both operands.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <utility>

// Development builds only (-DLDM_DEV_BUILD, tools/devbuild.sh): timing ablations that compile parts of a kernel out
// (wrong results) and cycle stamps.  The shipped library contains none of it.
//   TAPCONV_ABLATE (generic kernel): 1 no MFMA, 2 no weight DMA in the loop, 4 no LDS reads and no MFMA, 8 no epilogue
//     stores, 16 no halo DMA, 32 one weight fragment read per half step, 64 every second window fragment read
//   WS64_ABLATE (64-channel weight-stationary kernel): 1 no window DMA after the first tile, 2 no output stores, 4 no
//     staging of the epilogue's inputs
//   TAPCONV_STAMP: real-time stamps of workgroups / steps (tools/wglife.py, wslife.py, rflife.py, stamps.py)
#ifndef LDM_DEV_BUILD
#undef TAPCONV_ABLATE
#undef WS64_ABLATE
#undef TAPCONV_STAMP
#endif
#ifndef TAPCONV_ABLATE
#define TAPCONV_ABLATE 0
#endif
#ifndef WS64_ABLATE
#define WS64_ABLATE 0
#endif

namespace ldm {

// Launch thresholds (measured cross-over points).
constexpr long kSkinny3MaxWgF32 = 1024;   // f32 3x3 convolutions with at most this many skinny workgroups take the skinny kernel
constexpr long kSkinny1x1MaxPix = 512;    // 1x1 convolutions over at most this many pixels take the skinny kernel
// Persistent kernels with two workgroups per CU: eighths of a CU's tiles that its first-placed workgroup takes (the
// second one's weights and first window land later; 5 : 3 measured best, 6 : 2 slower, results do not depend on it).
constexpr int kEarlyEighths = 5;

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
    int tiles_x, tiles_y, ntiles, nblk_n, nphase;
    int act, out_f32;
    int t_img_stride, t_step_stride, t_pix_stride;
    long w_phase_stride;         // bytes
    float* sched_z;              // fused DDIM update: f32 sampler state (in place), else null
    float* sched_z_out;          // DDIM update + next step's first convolution as one kernel: the new state (sched_z is only read)
    const float* sched_coef;     // its coefficient table [steps][8]
    int* sched_ticket;           // optional: workgroup counter; the last workgroup advances *step_ptr
    const char* w2;              // fused ResidualBlock: packed weights and bias of the second convolution
    const float* bias2;
    int early_num;               // persistent kernels with two workgroups per CU: eighths of a pair's tiles for its first workgroup (0: even split)
    unsigned magic_tpi, magic_tx; // stride-2 persistent kernel: ceil(2^32 / tiles per image), ceil(2^32 / tiles_x) (see fast_div)
#ifdef TAPCONV_STAMP
    unsigned long long* dbg;     // diagnostic builds: cycle stamps of workgroup 0
    unsigned long long* dbg_rt;  // diagnostic builds: {first, last} 100-MHz real-time stamp of every workgroup
#endif
};

// Diagnostic cycle stamps (builds with -DTAPCONV_STAMP only).  s_memtime is issued without a wait and
// collected after the step's own s_waitcnt lgkmcnt(0); the values go to LDS (no vmcnt side effect) and are
// copied to global memory at the end of the kernel.
#ifdef TAPCONV_STAMP
#define STAMP_ISSUE(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define STAMP_STORE(idx, var)                                                                        \
    do {                                                                                             \
        unsigned int lo_, hi_;                                                                       \
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(lo_), "=v"(hi_)                    \
                     : "s"((unsigned int)(var)), "s"((unsigned int)((var) >> 32)));                  \
        if (lane == 0 && stamp_on && (idx) < 250) {                                                   \
            *(unsigned int*)(smem + stamp_base + (idx) * 8) = lo_;                                    \
            *(unsigned int*)(smem + stamp_base + (idx) * 8 + 4) = hi_;                                \
        }                                                                                            \
    } while (0)
// The persistent kernels (diagnostic builds): 100-MHz real-time stamps of every workgroup, eight slots each: start, first
// window and weights landed, end of each of the first four tiles, end (tools/wslife.py prints them).
#define WS_STAMP(k_) do { if (a.dbg_rt && threadIdx.x == 0) a.dbg_rt[8 * blockIdx.x + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define WS_STAMP_END() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WS_STAMP(6); } while (0)
#else
#define STAMP_ISSUE(var) do { } while (0)
#define STAMP_STORE(idx, var) do { } while (0)
#define WS_STAMP(k_) do { } while (0)
#define WS_STAMP_END() do { } while (0)
#endif


#ifdef TAPCONV_STAMP
// diagnostic builds: the stamp buffer LDM_DBG_RT_PTR names, for the kernel family LDM_DBG_RT_ONLY names (all when unset)
static unsigned long long* dbg_rt_for(const char* who) {
    const char* e = getenv("LDM_DBG_RT_PTR");
    const char* only = getenv("LDM_DBG_RT_ONLY");
    if (!e) return nullptr;
    if (only) {
        int i = 0;
        while (only[i] && only[i] == who[i]) ++i;
        if (only[i] != who[i]) return nullptr;
    }
    return (unsigned long long*)strtoull(e, nullptr, 16);
}
#endif

// SiLU.  f32 mode divides exactly; bf16 mode multiplies by the hardware reciprocal (1 ulp of f32, far
// below the bf16 rounding of the result): the IEEE division costs about as many issue cycles per
// output element as all the MFMAs that produced it at the 64-channel level.
template <bool EXACT>
__device__ __forceinline__ float silu_f(float v) {
    const float d = 1.0f + __expf(-v);
    if constexpr (EXACT) return v / d;
    else return v * __builtin_amdgcn_rcpf(d);
}

__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// two f32 -> packed bf16 pair (one v_cvt_pk_bf16_f32, round to nearest even)
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __builtin_bit_cast(float, (unsigned int)h << 16); }
// by-value helper: bit_cast applied directly to a vector-element expression reads element 0
__device__ __forceinline__ float u2f(unsigned int u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned int f2u(float f) { return __builtin_bit_cast(unsigned int, f); }

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
// the same with the non-temporal cache policy (nt): data that is read once and not again soon
__device__ __forceinline__ void glds16_nt(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2);
}

// Output stores of the convolution kernels are write-through (sc1): the bytes go to memory as the kernel runs and the
// line is dropped from this XCD's L2, so a launch ends with nothing dirty to write back (a launch that leaves B dirty
// bytes delays the next one by about B / 6 TB/s; plain stores measured 1-7 % slower per kernel in the sampler's graph,
// nontemporal ones slower than plain).  The instruction is written as asm so that the cache bits are exactly these;
// like any store it counts once in vmcnt, in issue order, which the hand-counted waits of the weight-stationary
// kernels rely on.  `s_nop 1` keeps the data registers intact until the store has read them.  8-byte write-through
// stores are one fabric write each (2.7 times the time per byte of a 16-byte one): those stay plain.
template <typename V>
__device__ __forceinline__ void store_out(char* p, const V& v) {
    static_assert(sizeof(V) == 16 || sizeof(V) == 8, "8- or 16-byte stores");
    if constexpr (sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *(V*)p = v;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): a loop whose index is a constant expression in the body
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// t / d for the tile arithmetic of a persistent kernel: d is a launch constant, magic = ceil(2^32 / d) comes from the
// host, exact while t * d < 2^32 (the launcher checks it).  A run-time division is about 25 scalar instructions, in
// front of every tile's barrier.
__device__ __forceinline__ int fast_div(int t, int d, unsigned magic) {
    return d == 1 ? t : (int)__umulhi((unsigned)t, magic);
}
static inline unsigned div_magic(int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Window coordinates of tap t (compile time): 3x3 -> (t/3, t%3); convT phase -> (t>>1, t&1) on top of
// the phase's own (py, px) origin, which is added at run time; 1x1 -> (0, 0).
template <int NTAPS>
__device__ __forceinline__ constexpr int tap_dy(int t) { return NTAPS == 9 ? t / 3 : (NTAPS == 4 ? (t >> 1) : 0); }
template <int NTAPS>
__device__ __forceinline__ constexpr int tap_dx(int t) { return NTAPS == 9 ? t % 3 : (NTAPS == 4 ? (t & 1) : 0); }

// TH: tile rows; BN: output channels per workgroup; WM x WN waves (4 or 8) over (pixel rows, channels);
// S: stride; KEXT: window extent (3 for 3x3 and the convT phases, 1 for 1x1); NTAPS: taps per chunk
// (9, 4 or 1; the tap loop is unrolled); MINW: waves per SIMD the register budget must allow.
// NT1: taps of the second source's chunks.  NT1 = NTAPS: the second source continues the channels of the
// first (1x1 conv over a concatenation).  NT1 = 1 with NTAPS = 4: fused transposed conv + concat + 1x1; the
// second source lives at OUTPUT resolution and contributes the pixel (2y+py, 2x+px) itself.
template <typename T, int TH, int BN, int WM, int WN, int S, int KEXT, int NTAPS, int MINW, int RING, int NT1 = NTAPS>
__global__ __launch_bounds__(64 * WM * WN, MINW) void tapconv_kernel(const TapConvArgs a) {
    constexpr bool SKIPUP = NT1 != NTAPS;
    static_assert(!SKIPUP || (NTAPS == 4 && NT1 == 1 && S == 1), "single-tap second source: transposed conv only");
    constexpr int NWV = WM * WN;
    static_assert(NWV == 4 || NWV == 8 || NWV == 16, "four, eight or sixteen waves");
    constexpr int TW = 16;
    constexpr int HH = (TH - 1) * S + KEXT;
    constexpr int HW = (TW - 1) * S + KEXT;
    constexpr int HWE = (HW + 1) / 2;            // even columns of a window row (stride-2 layout)
    constexpr int HP = HH * HW;
    constexpr int NG = (HP + 31) / 32;           // groups of 32 halo pixels
    constexpr int HPLANE = NG * 1024;            // one pair plane of the halo
    constexpr int HALO_BYTES = 4 * HPLANE;
    constexpr int NHP = NG * 4;                  // 1-KiB DMA pieces per halo chunk
    constexpr int NHW = (NHP + NWV - 1) / NWV;   // pieces per wave (the excess goes to the dump area)
    constexpr int WPLANE = BN * 32;
    constexpr int WSLAB = 4 * WPLANE;
    constexpr int NW = BN / 8 / NWV;             // weight pieces per wave and slab
    constexpr int D = RING - 1;                  // RING weight slabs in LDS; slab s+D is issued during step s
    constexpr int MF = TH / WM;                  // pixel fragments (tile rows) per wave
    constexpr int NF = BN / 16 / WN;             // channel fragments per wave
    constexpr int PAD = KEXT / 2;
    static_assert(TH % WM == 0 && (BN / 16) % WN == 0 && (BN / 8) % NWV == 0, "wave tiling");
    static_assert((D - 1) * NW + NHW < 64 && D >= 1 && D <= 5, "vmcnt range");
    static_assert(HALO_BYTES + RING * WSLAB + 1024 <= 160 * 1024, "LDS budget");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, l15 = lane & 15;

    // XCD-aware block -> (tile, channel block, phase): blocks b and b+8 share an XCD (and its L2), so
    // the channel blocks and the convT phases of one tile, which all read the same input window, get
    // ids 8 apart: they run close together in time on one XCD and the window comes from HBM once.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int nsub = a.nblk_n * a.nphase;
    const int tile = (j / nsub) * 8 + xcd;
    const int sub = j % nsub;
    const int nb = sub % a.nblk_n;
    const int phase = sub / a.nblk_n;
    if (tile >= a.ntiles) return;
    const int img = tile / (a.tiles_x * a.tiles_y);
    const int trem = tile - img * (a.tiles_x * a.tiles_y);
    const int y0 = (trem / a.tiles_x) * TH, x0 = (trem % a.tiles_x) * TW;
    const int n0 = nb * BN;
    const int iy0 = y0 * S - PAD, ix0 = x0 * S - PAD;
    const int nchunks = a.nch0 + a.nch1;
    const int nsteps = a.nch0 * NTAPS + a.nch1 * NT1;
    const int nhalo = nchunks > 1 ? 2 : 1;

    // LDS: [halo 0][halo 1 if more than one chunk][weight ring][1 KiB dump]
    const unsigned hbuf0 = 0;
    const unsigned wbuf0 = nhalo * HALO_BYTES;
    const unsigned dump = wbuf0 + RING * WSLAB;
#ifdef TAPCONV_STAMP
    if (a.dbg_rt && tid == 0) a.dbg_rt[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    const unsigned stamp_base = dump + 1024 + (wave == 0 ? 0 : 2048);
    const bool stamp_on = a.dbg && blockIdx.x == 8 && (wave == 0 || wave == NWV - 1);
    unsigned long long tsA = 0, tsB = 0, tsC = 0;
    int sidx = 1;
    if (stamp_on) for (int i = lane; i < 500; i += 64) *(unsigned int*)(smem + stamp_base + i * 4) = 0u;
    STAMP_ISSUE(tsA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP_STORE(0, tsA);
#endif

    const char* in0_img = a.in0 + (size_t)img * a.Hin * a.Win * a.cin0_bytes;
    const char* in1_img = a.in1 ? a.in1 + (size_t)img * (SKIPUP ? a.Hout * a.Wout : a.Hin * a.Win) * a.cin1_bytes : nullptr;
    const char* wbase = a.w + (size_t)phase * a.w_phase_stride + (size_t)n0 * 32;
    const size_t wslab_stride = (size_t)a.cout * 128;   // bytes between consecutive (chunk, tap) slabs
    const size_t wplane_stride = (size_t)a.cout * 32;   // bytes between the pair planes of one slab
    const char* zero = (const char*)g_zero16;

    // Halo DMA.  Wave w issues pieces e = w + NWV*q; piece e = (pixel group e >> 2, plane e & 3).  A lane
    // covers pixel group*32 + lane/2, bytes plane*32 + (lane&1)*16 of the chunk.  The pixel's offset
    // inside the image is the same for every chunk: computed once (-1 = zero page).
    int hpix[NHW];
#pragma unroll
    for (int q = 0; q < NHW; ++q) {
        const int grp = (wave + NWV * q) >> 2;
        const int hp = grp * 32 + (lane >> 1);
        // stride 2: a window row is stored de-interleaved (even columns, then odd columns), so that the 16
        // pixels of a fragment (every second column) are consecutive slots and its read is conflict free
        const int hy = hp / HW, hr = hp - hy * HW;
        const int hx = S == 2 ? (hr < HWE ? 2 * hr : 2 * (hr - HWE) + 1) : hr;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = grp < NG && hp < HP && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        hpix[q] = ok ? (iy * a.Win + ix) : -1;
    }
    // single-tap second source (fused transposed conv): window position (hy, hx) holds the pixel that the
    // phase's centre tap reads for output (hy - py, hx - px), i.e. pixel (2*gy + py, 2*gx + px) of the
    // output-resolution tensor; every other window position is zero
    int hpix1[SKIPUP ? NHW : 1];
    if constexpr (SKIPUP) {
        const int py1 = phase >> 1, px1 = phase & 1;
#pragma unroll
        for (int q = 0; q < NHW; ++q) {
            const int grp = (wave + NWV * q) >> 2;
            const int hp = grp * 32 + (lane >> 1);
            const int hy = hp / HW, hx = hp - hy * HW;
            const int r = hy - py1, cc = hx - px1;
            const int gy = y0 + r, gx = x0 + cc;
            const bool ok = grp < NG && hp < HP && r >= 0 && r < TH && cc >= 0 && cc < TW && gy < a.Hg && gx < a.Wg;
            hpix1[q] = ok ? ((2 * gy + py1) * a.Wout + 2 * gx + px1) : -1;
        }
    }
    auto issue_halo = [&](int c) {
        const bool src1 = c >= a.nch0;
        const char* base = (!src1 ? in0_img + c * 128 : in1_img + (c - a.nch0) * 128) + (lane & 1) * 16 +
                           (wave & 3) * 32;
        const int pix_bytes = !src1 ? a.cin0_bytes : a.cin1_bytes;
        const unsigned hb = hbuf0 + (c & 1) * HALO_BYTES + (wave & 3) * HPLANE;
#pragma unroll
        for (int q = 0; q < NHW; ++q) {
            const int grp = (wave + NWV * q) >> 2;       // wave-uniform
            int pix = hpix[q];
            if constexpr (SKIPUP) pix = src1 ? hpix1[q] : pix;
            const char* src = pix >= 0 ? base + (size_t)pix * pix_bytes : zero;
            const unsigned dst = grp < NG ? hb + grp * 1024 : dump;
            if constexpr (!(TAPCONV_ABLATE & 16)) glds16(src, smem + dst);
        }
    };
    // Weight DMA: slab piece e = (plane e / (BN/32), 32 rows); wave w issues e = w + NWV*i.
    auto issue_w = [&](int s, int slot) {
        const char* sl = wbase + (size_t)s * wslab_stride + lane * 16;
        const unsigned dst = wbuf0 + slot * WSLAB;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int e = wave + NWV * i;
            const int pp = e / (BN / 32), r0 = (e % (BN / 32)) * 32;
            glds16(sl + pp * wplane_stride + r0 * 32, smem + dst + e * 1024);
        }
    };

    // Per-lane LDS read bases; everything else is an immediate.
    //   weights: plane (kk*2 + g/2), row (wn*NF + nf)*16 + l15, half g&1
    //   halo:    plane (kk*2 + g/2), pixel ((wm*MF + mf)*S + dy)*HW + l15*S + dx, half g&1
    const unsigned wlane = (g >> 1) * WPLANE + ((wn * NF) * 16 + l15) * 32 + (g & 1) * 16;
    const int py = phase >> 1, px = phase & 1;
    const int origin = NTAPS == 4 ? py * HW + px : 0;   // convT: the phase's window origin
    const unsigned hlane = (g >> 1) * HPLANE + ((wm * MF * S) * HW + l15 * (S == 2 ? 1 : S) + origin) * 32 + (g & 1) * 16;

    f32x4 acc[NF][MF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) acc[nf][mf] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Fragment loads and the MFMA batch of one half step (32 bf16 / 16 f32 channels of one tap).
    auto load_frags = [&](int kk, unsigned hb, unsigned wb, int t, u32x4 (&wf)[NF], u32x4 (&xf)[MF]) {
        const char* wp = smem + wb + wlane;
        const char* hp = smem + hb + hlane;
        const int dx = tap_dx<NTAPS>(t);
        const int toff = (tap_dy<NTAPS>(t) * HW + (S == 2 ? (dx & 1) * HWE + (dx >> 1) : dx)) * 32;
        // (ablation bits 32 / 64, timing only: one weight fragment read instead of NF / every second window fragment read)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            wf[nf] = ((TAPCONV_ABLATE & 32) && nf > 0) ? wf[0] : *(const u32x4*)(wp + kk * 2 * WPLANE + nf * 512);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
            xf[mf] = ((TAPCONV_ABLATE & 64) && (mf & 1)) ? xf[mf - 1] : *(const u32x4*)(hp + kk * 2 * HPLANE + mf * (S * HW * 32) + toff);
    };
    auto mfma_batch = [&](const u32x4 (&wf)[NF], const u32x4 (&xf)[MF]) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                if constexpr (TAPCONV_ABLATE & 1) {
                    asm volatile("" ::"v"(wf[nf]), "v"(xf[mf]));
                } else if constexpr (sizeof(T) == 2) {
                    acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wf[nf]), __builtin_bit_cast(bf16x8, xf[mf]), acc[nf][mf], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[nf][mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(u2f(wf[nf][q]), u2f(xf[mf][q]), acc[nf][mf],
                                                                            0, 0, 0);
                }
            }
    };
    // Scheduling directive for one half step: spread the next half's LDS reads between this half's
    // MFMAs (a burst of reads in front of the MFMAs stalls the wave at LDS issue while the matrix
    // pipe idles).
    auto interleave_reads_with_mfma = [&]() {
        constexpr int NR = MF + NF;
        constexpr int NM = MF * NF * (sizeof(T) == 2 ? 1 : 4);
        constexpr int PER = NM / NR;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NM - PER * NR, 0);
    };

    // Counted wait: leave the `young` youngest weight slabs (and a halo, if `halo`) in flight.
    auto wait_leaving = [&](int young, bool halo) {
        if (halo) {
            switch (young) {
                case 0: wait_vmcnt<NHW>(); break;
                case 1: wait_vmcnt<(D >= 2 ? 1 : 0) * NW + NHW>(); break;
                case 2: wait_vmcnt<(D >= 3 ? 2 : 0) * NW + NHW>(); break;
                case 3: wait_vmcnt<(D >= 4 ? 3 : 0) * NW + NHW>(); break;
                default: wait_vmcnt<(D >= 5 ? 4 : 0) * NW + NHW>(); break;
            }
        } else {
            switch (young) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<(D >= 2 ? 1 : 0) * NW>(); break;
                case 2: wait_vmcnt<(D >= 3 ? 2 : 0) * NW>(); break;
                case 3: wait_vmcnt<(D >= 4 ? 3 : 0) * NW>(); break;
                default: wait_vmcnt<(D >= 5 ? 4 : 0) * NW>(); break;
            }
        }
    };

    // The barrier at the end of step s publishes slab s+1, so the first reads of step s+1 are issued after it.
    // (Publishing slab s+2 instead, with the reads of step s+1 issued before the barrier, measured no faster.)
    constexpr int PUB = 1;                       // the barrier of step s publishes slab s + PUB

    // prologue: halo 0, then slabs 0 .. D-1; wait for halo 0 and the slabs below PUB
    issue_halo(0);
    const int npro = nsteps < D ? nsteps : D;
    for (int d = 0; d < npro; ++d) issue_w(d, d);
    {
        const int need = PUB - 1 < nsteps - 1 ? PUB - 1 : nsteps - 1;
        wait_leaving(npro - 1 - need, false);
    }
    __builtin_amdgcn_s_barrier();

    // Software pipeline over half steps: while the MFMAs of one half run, the LDS reads of the next
    // half are in flight (fragments double-buffered in registers: A = first half, B = second half).
    u32x4 wfA[NF], xfA[MF], wfB[NF], xfB[MF];
    if constexpr (!(TAPCONV_ABLATE & 4)) load_frags(0, hbuf0, wbuf0, 0, wfA, xfA);

    int slot = 0, s = 0;   // slot = s % RING
    // Every count a step waits for depends only on the tap and on whether the chunk is the last one.  With the tap
    // loop unrolled over constant indices and the last chunk as a second copy of the body, each step's wait is one
    // s_waitcnt with an immediate; computed at run time, the same counts were a tree of about ten scalar branches
    // between the step's last MFMA and its barrier.
    constexpr bool STATIC_WAITS = !SKIPUP && NTAPS > D && !(TAPCONV_ABLATE & 31);
    if constexpr (STATIC_WAITS) {
        auto run_chunk = [&](int c, auto lastc_tag) {
            constexpr bool LASTC = decltype(lastc_tag)::value;
            const unsigned hb = hbuf0 + (c & 1) * HALO_BYTES;
            static_for<NTAPS>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int r = NTAPS - 1 - t;             // steps of this chunk after this one
                STAMP_ISSUE(tsA);
                if constexpr (t == 0 && !LASTC) issue_halo(c + 1);
                // past the last slab the last one is issued again, into a ring slot that no later step reads
                if constexpr (!LASTC || r >= D) issue_w(s + D, slot == 0 ? RING - 1 : slot - 1);
                else issue_w(nsteps - 1, slot == 0 ? RING - 1 : slot - 1);
                const unsigned wb = wbuf0 + slot * WSLAB;
                const int slot_next = slot + 1 == RING ? 0 : slot + 1;
                constexpr bool wrap = t == NTAPS - 1;
                const unsigned hbn = wrap ? hbuf0 + (LASTC ? 0 : ((c + 1) & 1)) * HALO_BYTES : hb;
                load_frags(1, hb, wb, t, wfB, xfB);
                mfma_batch(wfA, xfA);
                interleave_reads_with_mfma();
                // what may stay in flight across the barrier (see the run-time form below for the reasoning)
                constexpr int need_k = LASTC ? (PUB < r ? PUB : r) : PUB;
                constexpr int last_k = LASTC ? (D < r ? D : r) : D;
                constexpr int nrep = LASTC && D > r ? D - r : 0;
                constexpr int young = (LASTC && r == 0) ? 0 : (last_k - need_k) + nrep;
                constexpr bool h_young = !LASTC && t <= D - 1 - PUB && t <= NTAPS - 1 - PUB;
                STAMP_ISSUE(tsB);
                wait_vmcnt<(young <= D - 1 ? young : 0) * NW + (h_young ? NHW : 0)>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef TAPCONV_STAMP
                STAMP_ISSUE(tsC);
                STAMP_STORE(sidx, tsA); STAMP_STORE(sidx + 1, tsB); STAMP_STORE(sidx + 2, tsC);
                sidx += 3;
#endif
                __builtin_amdgcn_s_barrier();
                slot = slot_next;
                load_frags(0, hbn, wbuf0 + slot * WSLAB, wrap ? 0 : t + 1, wfA, xfA);
                mfma_batch(wfB, xfB);
                interleave_reads_with_mfma();
                ++s;
            });
        };
        for (int c = 0; c + 1 < nchunks; ++c) run_chunk(c, std::bool_constant<false>{});
        run_chunk(nchunks - 1, std::bool_constant<true>{});
    } else {
    for (int c = 0; c < nchunks; ++c) {
        const unsigned hb = hbuf0 + (c & 1) * HALO_BYTES;
        const bool more_chunks = c + 1 < nchunks;
        const int ntc = (SKIPUP && c >= a.nch0) ? NT1 : NTAPS;      // taps of this chunk
#pragma unroll
        for (int t = 0; t < NTAPS; ++t, ++s) {
            if (SKIPUP && t >= ntc) break;
            // slab s+D goes to the ring slot that step s-1 read; every wave finished those reads
            // before it passed that step's barrier
            STAMP_ISSUE(tsA);
            const bool halo_now = t == 0 && more_chunks;
            if (halo_now) issue_halo(c + 1);      // older than slab s+D: a later counted wait covers it
            // Past the last slab the last one is issued again, into a ring slot that no later step reads: the
            // issue stays unconditional, so the second half of this step and the first half of the next are one
            // basic block (a branch here made the compiler wait for every LDS read in flight at the join).
            if (!(TAPCONV_ABLATE & 2)) issue_w(s + D < nsteps ? s + D : nsteps - 1, slot == 0 ? RING - 1 : slot - 1);
            const unsigned wb = wbuf0 + slot * WSLAB;
            const int slot_next = slot + 1 == RING ? 0 : slot + 1;
            // first half of the next step (after the very last step this reads LDS that nobody uses:
            // harmless, and reads and MFMAs stay in one basic block for the interleave)
            const bool wrap = t == ntc - 1;
            const unsigned hbn = wrap ? hbuf0 + (more_chunks ? ((c + 1) & 1) : 0) * HALO_BYTES : hb;
            if constexpr (!(TAPCONV_ABLATE & 4)) {
                load_frags(1, hb, wb, t, wfB, xfB);
                mfma_batch(wfA, xfA);
                interleave_reads_with_mfma();
            }
            // What must have landed before the barrier: slab s+PUB and, if step s+PUB opens a chunk,
            // that chunk's halo.  Younger pieces may stay in flight: the later slabs that exist, and the
            // next chunk's halo while it is younger than slab s+PUB (the halo is issued at tap 0, slab
            // s+PUB at step s+PUB-D) and step s+PUB still belongs to this chunk.
            const int need = s + PUB < nsteps - 1 ? s + PUB : nsteps - 1;
            const int last = s + D < nsteps - 1 ? s + D : nsteps - 1;
            // repeated issues of the last slab are younger than everything needed; the last step drains the queue
            const int nrep = s + D + 1 - nsteps > 0 ? s + D + 1 - nsteps : 0;
            const int young = (TAPCONV_ABLATE & 2) || s == nsteps - 1 ? 0 : (last - need > 0 ? last - need : 0) + nrep;
            const bool h_young = more_chunks && t <= D - 1 - PUB && t <= ntc - 1 - PUB && !(TAPCONV_ABLATE & 16);
#ifdef TAPCONV_STAMP
            STAMP_ISSUE(tsB);        // (waits for the LDS reads as a side effect)
#endif
            wait_leaving(young, h_young);
            // this step's LDS reads are complete before any wave may overwrite what they read
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef TAPCONV_STAMP
            STAMP_ISSUE(tsC);
            STAMP_STORE(sidx, tsA); STAMP_STORE(sidx + 1, tsB); STAMP_STORE(sidx + 2, tsC);
            sidx += 3;
#endif
            __builtin_amdgcn_s_barrier();
            slot = slot_next;
            if constexpr (!(TAPCONV_ABLATE & 4)) {
                load_frags(0, hbn, wbuf0 + slot * WSLAB, wrap ? 0 : t + 1, wfA, xfA);
                mfma_batch(wfB, xfB);
                interleave_reads_with_mfma();
            }
        }
    }
    }

    // ---- epilogue ---------------------------------------------------------------------------
    // The accumulators (4 channels of one pixel per lane) go through LDS so that global memory is
    // touched in whole rows: every thread then owns 8 consecutive channels of a pixel, i.e. 16 bytes of
    // bf16 (32 of f32), consecutive threads consecutive addresses; the residual is read the same way.
    //   v = acc + bias (+ residual) -> activation -> (+ time vector) -> store
    // All operand buffers in LDS are dead here (every wave is past the last barrier; the DMA queue
    // was drained by the last counted wait).
#ifdef TAPCONV_STAMP
    // copy the loop's stamps out before the staging area overwrites nothing of them (they sit past the dump)
    STAMP_ISSUE(tsA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP_STORE(sidx, tsA);
    ++sidx;
#endif
    constexpr int BM = TH * 16;
    constexpr int RS = BN * 4 + 16;              // staging row stride (bytes): +16 keeps ds_write_b128 conflict free
    constexpr int NTHR = 64 * NWV;
    constexpr int CG = BN / 8;                   // 8-channel groups per pixel
    constexpr int NIT = BM * CG / NTHR;
    static_assert(NTHR % CG == 0 && (BM * CG) % NTHR == 0, "epilogue tiling");
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int p = (wm * MF + mf) * 16 + l15;
            const int ch = (wn * NF + nf) * 16 + g * 4;
            *(f32x4*)(smem + p * RS + ch * 4) = acc[nf][mf];
        }
    const int cg = tid % CG;                     // the same for every iteration of this thread
    const int co = n0 + cg * 8;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const float* tv = a.tvec ? a.tvec + (size_t)img * a.t_img_stride + (size_t)step * a.t_step_stride + co : nullptr;
    const f32x4 b0 = *(const f32x4*)(a.bias + co), b1 = *(const f32x4*)(a.bias + co + 4);
    f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
    if (tv) { t0 = *(const f32x4*)tv; t1 = *(const f32x4*)(tv + 4); }
    size_t eo[NIT];
    bool ok[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int p = (tid + i * NTHR) / CG;
        const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
        ok[i] = gy < a.Hg && gx < a.Wg;
        const int oy = (gy < a.Hg ? gy : a.Hg - 1) * a.osc + py, ox = (gx < a.Wg ? gx : a.Wg - 1) * a.osc + px;
        eo[i] = (((size_t)img * a.Hout + oy) * a.Wout + ox) * a.cout + co;
    }
    f32x4 r0[NIT], r1[NIT];
    if (a.res) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if constexpr (sizeof(T) == 2) {
                const u32x4 r = *(const u32x4*)(a.res + eo[i] * 2);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned int u = r[q];
                    (q < 2 ? r0[i] : r1[i])[(q & 1) * 2] = bf2f((bf16_t)(u & 0xffffu));
                    (q < 2 ? r0[i] : r1[i])[(q & 1) * 2 + 1] = bf2f((bf16_t)(u >> 16));
                }
            } else {
                r0[i] = *(const f32x4*)(a.res + eo[i] * 4);
                r1[i] = *(const f32x4*)(a.res + eo[i] * 4 + 16);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int p = (tid + i * NTHR) / CG;
        f32x4 v0 = *(const f32x4*)(smem + p * RS + cg * 32) + b0;
        f32x4 v1 = *(const f32x4*)(smem + p * RS + cg * 32 + 16) + b1;
        if (a.res) { v0 += r0[i]; v1 += r1[i]; }
        if constexpr (NTAPS == 1) {     // 1x1 convolutions only (the 3x3 instantiations stay as they were)
            if (a.t_pix_stride) {       // a time vector per pixel: a batch handed over as one image
                const int gy = y0 + (p >> 4), gx = x0 + (p & 15);
                const float* tp = tv + (size_t)((gy < a.Hg ? gy : a.Hg - 1) * a.Wg + (gx < a.Wg ? gx : a.Wg - 1)) * a.t_pix_stride;
                t0 = *(const f32x4*)tp;
                t1 = *(const f32x4*)(tp + 4);
            }
        }
        if (a.act == ACT_SILU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { v0[q] = silu_f<sizeof(T) == 4>(v0[q]); v1[q] = silu_f<sizeof(T) == 4>(v1[q]); }
        } else if (a.act == ACT_RELU) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { v0[q] = fmaxf(v0[q], 0.f); v1[q] = fmaxf(v1[q], 0.f); }
        } else if (a.act == ACT_TANH) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { v0[q] = tanhf(v0[q]); v1[q] = tanhf(v1[q]); }
        }
        v0 += t0;
        v1 += t1;
        const bool st = ok[i] && (!(TAPCONV_ABLATE & 8) || v0[0] == 12345.678f);
        if (sizeof(T) == 4 || a.out_f32) {
            if (st) {
                store_out(a.out + eo[i] * 4, v0);
                store_out(a.out + eo[i] * 4 + 16, v1);
            }
        } else {
            u32x4 o;
            o[0] = pack2(v0[0], v0[1]);
            o[1] = pack2(v0[2], v0[3]);
            o[2] = pack2(v1[0], v1[1]);
            o[3] = pack2(v1[2], v1[3]);
            if (st) {
                store_out(a.out + eo[i] * 2, o);
            }
        }
    }
#ifdef TAPCONV_STAMP
    STAMP_ISSUE(tsA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP_STORE(sidx, tsA);
    __syncthreads();
    if (stamp_on)
        for (int i = lane; i < 250; i += 64)
            a.dbg[(wave == 0 ? 0 : 256) + i] = *(const unsigned long long*)(smem + stamp_base + i * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's output stores have left
    __syncthreads();
    if (a.dbg_rt && tid == 0) a.dbg_rt[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------
// Weight-stationary persistent variant: conv3x3 stride 1, 64 -> 64 channels, bf16 (the 64-channel
// level of the UNet: down1, down_res1, up_res2, out_conv).  At that level a launch moves 67-100 MB
// through HBM for 19 GFLOP; per tile the generic kernel spends more time on its DMA steps, barriers and
// epilogue than on MFMAs.  Here a workgroup of 4 waves walks over 8x16-pixel tiles; every wave keeps
// the weight fragments of its 32 output channels (18 half steps x 2 fragments = 144 registers) in
// registers for the whole kernel, so per tile only the halo crosses LDS (double-buffered) and there is
// ONE barrier per tile.
//
// A wave computes its 4 rows of a tile as two halves of 2 rows (72 MFMAs each).  Everything else is
// spread over the MFMAs of a half, one small piece per half step, so that no wave ever has a phase
// without MFMAs and the workgroups of the chip (which start together and do equal work) do not all
// store at the same moment:
//   * the DMA pieces of the NEXT tile's halo (3 per half),
//   * the loads the epilogue of THIS half needs (residual rows, or the rows of the sampler state),
//   * the whole epilogue of the PREVIOUS half (its 16 accumulator registers are kept): lane swap,
//     bias, residual, SiLU, time vector, packing and the stores.
// The epilogue needs no LDS: v_permlane16_swap exchanges the odd 16-lane rows of one accumulator with
// the even rows of the other, after which a lane owns 8 consecutive channels of a pixel.
//
// Modes (the epilogue shapes the UNet uses at this level):
//   WS_PLAIN     out = conv + bias
//   WS_SILU_TV   out = silu(conv + bias) + time vector
//   WS_SILU_RES  out = silu(conv + bias + residual)
//   WS_DDIM      conv + bias is the predicted noise e of the sampler; the deterministic DDIM update
//                z' = ((z - e*s1m)/sa)*c1 + e*c2 is applied to the f32 state in place and its bf16
//                copy (the next UNet input) is written; e itself never goes to memory.
// ---------------------------------------------------------------------------------------------
template <bool B> struct BoolC { static constexpr bool value = B; };
enum { WS_PLAIN = 0, WS_SILU_TV = 1, WS_SILU_RES = 2, WS_DDIM = 3 };

template <int MODE>
__global__ __launch_bounds__(256, 2) void tapconv_ws64_kernel(const TapConvArgs a) {
    constexpr bool RES = MODE == WS_SILU_RES, TV = MODE == WS_SILU_TV, SILU = RES || TV, DDIM = MODE == WS_DDIM;
    constexpr int TH = 8, HW = 18, HP = 10 * 18, NG = (HP + 31) / 32;
    constexpr int HPLANE = NG * 1024, HALO_BYTES = 4 * HPLANE;       // 4 waves: piece q of wave w = (group q, plane w)
    constexpr int WN = 2, NF = 2, KS = 18, HR = 2;                   // HR rows per half
    // epilogue inputs of a half (residual rows, rows of the sampler state, or the time vector): 1-KiB
    // LDS-DMA pieces per wave, 16 bytes per lane, read back by the same lane
    constexpr int NLDG = RES ? HR : (DDIM ? 2 * HR : (TV ? 2 : 0));
    constexpr int NST = DDIM ? 3 : 1;                                // stores per epilogue row
    constexpr int XD = 4;                                            // depth of the X fragment register ring
    static_assert(NG == 6, "three halo pieces per half");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int g = lane >> 4, l15 = lane & 15;
    const char* zero = (const char*)g_zero16;
    const int tiles_per_img = a.tiles_x * a.tiles_y;

    // Tiles of this workgroup: tile0, tile0 + tstride, ... (nmine of them; the launcher never starts more workgroups than
    // tiles).  Even split: workgroup b takes every gridDim.x-th tile from b.  Uneven split (a.early_num > 0, grid = two
    // workgroups per CU): the workgroups b and b + grid/2 share the tiles p, p + grid/2, p + grid, ... of their pair p;
    // b < grid/2 takes the first early_num/8 of them.  Measured: the dispatcher fills every CU once before it places a
    // second workgroup, the second one's weights and first window land 1.9 us after the first one's, and the SIMD's
    // arbiter serves the older wave first (2.9 against 4.4 us per tile while both run): with equal shares the launch
    // ends with the second workgroups, 4.5 us after the first ones are done.
    int tile, tstride, nmine;
    if (a.early_num > 0) {
        const int half = (int)gridDim.x >> 1, pr = (int)blockIdx.x >= half ? (int)blockIdx.x - half : (int)blockIdx.x;
        const int npair = (a.ntiles - pr + half - 1) / half;
        int ne = (npair * a.early_num + 4) >> 3;
        if (ne > npair - 1) ne = npair - 1;          // the second workgroup keeps at least one tile
        tstride = half;
        tile = (int)blockIdx.x >= half ? pr + ne * half : pr;
        nmine = (int)blockIdx.x >= half ? npair - ne : ne;
    } else {
        tile = blockIdx.x;
        tstride = gridDim.x;
        nmine = (a.ntiles - tile + tstride - 1) / tstride;
    }
    int kt = 0;                     // index of the current tile in this workgroup's list
    WS_STAMP(0);

    // this wave's 32 output channels of every (tap, half), straight from the packed weights (L2)
    u32x4 wreg[KS][NF];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            wreg[ks][nf] = *(const u32x4*)(a.w + (ks >> 1) * 8192 + ((ks & 1) * 2 + (g >> 1)) * 2048 +
                                           ((wn * NF + nf) * 16 + l15) * 32 + (g & 1) * 16);

    // halo DMA: window coordinates of this lane's pixel in each piece do not depend on the tile
    int hyx[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int hp = q * 32 + (lane >> 1);
        const int hy = hp / HW, hx = hp - hy * HW;
        hyx[q] = hp < HP ? (hy << 8 | hx) : -1;
    }
    // origin of a tile's window and the lane's source base; a tile past the end reads the zero page
    const char* hbase;
    int hiy0, hix0;
    bool hlive;
    auto halo_origin = [&](int t) {
        hlive = t < a.ntiles;
        const int tt = hlive ? t : 0;
        const int img = tt / tiles_per_img;
        const int trem = tt - img * tiles_per_img;
        hiy0 = (trem / a.tiles_x) * TH - 1;
        hix0 = (trem % a.tiles_x) * 16 - 1;
        hbase = a.in0 + (size_t)img * a.Hin * a.Win * 128 + (lane & 1) * 16 + wave * 32;
    };
    bool in_loop = false;
    auto halo_piece = [&](int q, int buf) {
        if ((WS64_ABLATE & 1) && in_loop) return;
        const int iy = hiy0 + (hyx[q] >> 8), ix = hix0 + (hyx[q] & 255);
        const bool ok = hlive && hyx[q] >= 0 && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        const char* src = ok ? hbase + (size_t)(iy * a.Win + ix) * 128 : zero;
        glds16(src, smem + buf * HALO_BYTES + wave * HPLANE + q * 1024);
    };
    halo_origin(tile);
#pragma unroll
    for (int q = 0; q < NG; ++q) halo_piece(q, 0);
    in_loop = true;

    // epilogue constants: after the lane swap this lane owns channels co .. co+7 of pixel column l15
    const unsigned hlane = (g >> 1) * HPLANE + ((wm * 2 * HR) * HW + l15) * 32 + (g & 1) * 16;
    const int co = wn * 32 + (g & 1) * 16 + (g >> 1) * 8;
    const int step = a.step_ptr ? *a.step_ptr : 0;
    const size_t erow = (size_t)a.Wout * 64;
    // the bias in the accumulators' own layout (fragment nf, rows 4g .. 4g+3): a half's accumulators start from it, so
    // its first MFMAs add onto the bias and the epilogue has 32 additions less per tile (+0.4 % on the 50-step chain)
    f32x4 bpre[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bpre[nf] = *(const f32x4*)(a.bias + (wn * NF + nf) * 16 + 4 * g);
    float k_s1m = 0.f, k_rsa = 0.f, k_c1 = 0.f, k_c2 = 0.f;
    if (DDIM) {
        const float* cf = a.sched_coef + (size_t)step * 8;
        const bool last = cf[6] != 0.f;
        k_s1m = cf[1];
        k_rsa = 1.0f / cf[0];
        k_c1 = last ? 1.0f : cf[2];
        k_c2 = last ? 0.0f : cf[3];
    }

    // state handed from one half to the next: the finished accumulators, the loads their epilogue
    // needs, the time vector and the element offset of their first row
    f32x4 accP[NF][HR];
    u32x4 ldP[NLDG > 1 ? NLDG : 2];
    size_t eP = 0;
    char* stage = smem + 2 * HALO_BYTES + wave * (NLDG * 1024);      // wave-uniform: DMA destination base
    f32x4 v0, v1;          // the row that is passing through the epilogue slices

    // Slice s (0..17) of the previous half's epilogue: row s/9; step 0 swaps lanes and adds bias and
    // residual, steps 1..7 apply the activation to the 8 values, step 8 finishes and stores.
    auto epi_slice = [&](int s18) {
        const int j = s18 / 9, s = s18 % 9;
        if (s == 0) {
            // lanes of odd 16-lane rows take the partner's fragment 1 channels, even rows keep
            // fragment 0 and receive the partner's: v0 | v1 = 8 consecutive channels
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto sw = __builtin_amdgcn_permlane16_swap(f2u(accP[0][j][q]), f2u(accP[1][j][q]), false, false);
                v0[q] = u2f(sw[0]);
                v1[q] = u2f(sw[1]);
            }
            if (RES) {
                const unsigned int u0 = ldP[j][0], u1 = ldP[j][1], u2 = ldP[j][2], u3 = ldP[j][3];
                v0[0] += bf2f((bf16_t)(u0 & 0xffffu)); v0[1] += bf2f((bf16_t)(u0 >> 16));
                v0[2] += bf2f((bf16_t)(u1 & 0xffffu)); v0[3] += bf2f((bf16_t)(u1 >> 16));
                v1[0] += bf2f((bf16_t)(u2 & 0xffffu)); v1[1] += bf2f((bf16_t)(u2 >> 16));
                v1[2] += bf2f((bf16_t)(u3 & 0xffffu)); v1[3] += bf2f((bf16_t)(u3 >> 16));
            }
        } else if (s < 8) {
            if (SILU) {
                if (s == 1) { v0[0] = silu_f<false>(v0[0]); v0[1] = silu_f<false>(v0[1]); }
                if (s == 2) v0[2] = silu_f<false>(v0[2]);
                if (s == 3) v0[3] = silu_f<false>(v0[3]);
                if (s == 4) v1[0] = silu_f<false>(v1[0]);
                if (s == 5) v1[1] = silu_f<false>(v1[1]);
                if (s == 6) v1[2] = silu_f<false>(v1[2]);
                if (s == 7) v1[3] = silu_f<false>(v1[3]);
            }
        } else {
            const size_t eo = eP + j * erow;
            if (TV) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { v0[q] += u2f(ldP[0][q]); v1[q] += u2f(ldP[1][q]); }
            }
            if (DDIM) {
                f32x4 z0, z1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    z0[q] = u2f(ldP[2 * j][q]);
                    z1[q] = u2f(ldP[2 * j + 1][q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    z0[q] = ((z0[q] - v0[q] * k_s1m) * k_rsa) * k_c1 + v0[q] * k_c2;
                    z1[q] = ((z1[q] - v1[q] * k_s1m) * k_rsa) * k_c1 + v1[q] * k_c2;
                }
                // plain stores for the sampler's state and its bf16 copy: written through, this kernel took 56.7
                // instead of 46.7 us (state) and the 50-step chain 0.35 % longer (copy)
                *(f32x4*)(a.sched_z + eo) = z0;
                *(f32x4*)(a.sched_z + eo + 4) = z1;
                v0 = z0;
                v1 = z1;
            }
            u32x4 o;
            o[0] = pack2(v0[0], v0[1]);
            o[1] = pack2(v0[2], v0[3]);
            o[2] = pack2(v1[0], v1[1]);
            o[3] = pack2(v1[2], v1[3]);
            if ((WS64_ABLATE & 2) && o[0] != 0x12345678u) return;
            if constexpr (DDIM) *(u32x4*)(a.out + eo * 2) = o;
            else store_out(a.out + eo * 2, o);
        }
    };

    int buf = 0;
    int img = 0;
    size_t e0 = 0;

    // Order of a wave's vector-memory operations in one half (compiler barriers keep it): 3 halo
    // pieces, NLDG pieces of this half's epilogue inputs, then the 2 x NST stores of the previous half's
    // epilogue rows.  vmcnt retires in issue order, so "at most 2*NST outstanding" at the start of the
    // next half means the halo pieces and the staged inputs have landed while the stores may still be
    // in flight.  There are no plain global loads in the loop: the compiler would drain every LDS-DMA
    // (vmcnt 0) at the first use of one.
    auto tile_top = [&](auto first_tag) {
        if (decltype(first_tag)::value) wait_vmcnt<0>();
        else wait_vmcnt<2 * NST>();
        // every wave is done reading the other halo buffer, which the next tile's pieces overwrite
        __builtin_amdgcn_s_barrier();
        halo_origin(kt + 1 < nmine ? tile + tstride : a.ntiles);
        img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int y0 = (trem / a.tiles_x) * TH, x0 = (trem % a.tiles_x) * 16;
        e0 = (((size_t)img * a.Hout + y0 + wm * 2 * HR) * a.Wout + x0 + l15) * 64 + co;
    };
    // (Loading these rows straight into registers with hand-written loads instead of staging them through LDS
    // measured the same in the SiLU modes and needs 16 more registers in the DDIM mode, which then spills.)
    auto read_staged = [&]() {
#pragma unroll
        for (int i = 0; i < NLDG; ++i) ldP[i] = *(const u32x4*)(stage + i * 1024 + lane * 16);
    };

    // half_tag: second half of the tile; epi_tag: a previous half exists (its epilogue runs here);
    // wait_tag: 0 = the tile top already waited, 1 = wait leaving the previous half's stores in flight,
    // 2 = wait for everything (the previous half issued no stores)
    auto do_half = [&](auto half_tag, auto epi_tag, auto wait_tag) {
        constexpr int HALF = decltype(half_tag)::value ? 1 : 0;
        constexpr bool EPI = decltype(epi_tag)::value;
        constexpr int WAIT = decltype(wait_tag)::value;
        if (EPI && NLDG > 0) {
            if (WAIT == 1) wait_vmcnt<2 * NST>();
            if (WAIT == 2) wait_vmcnt<0>();
            read_staged();
        }
        const size_t eC = e0 + HALF * HR * erow;
        f32x4 acc[NF][HR];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < HR; ++r) acc[nf][r] = bpre[nf];
        const char* hp0 = smem + buf * HALO_BYTES + hlane + HALF * HR * (HW * 32);
        // The X fragment of (window row h, column shift dx, channel half kk) serves every (output row
        // r, tap row dy) with r + dy = h, so it is read from LDS once: 24 fragments per half instead of
        // 36 (an LDS read per two MFMAs would saturate the LDS at two waves per SIMD).  Fragments go
        // through a register ring XD deep; fragment f = (h, dx, kk) with kk fastest.
        constexpr int NFR = (HR + 2) * 6;
        auto foff = [&](int f) { return (f & 1) * 2 * HPLANE + ((f / 6) * HW + ((f % 6) >> 1)) * 32; };
        u32x4 xf[XD];
#pragma unroll
        for (int f = 0; f < XD - 1; ++f) xf[f] = *(const u32x4*)(hp0 + foff(f));
        int slice = 0;
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
            if (f + XD - 1 < NFR) xf[(f + XD - 1) % XD] = *(const u32x4*)(hp0 + foff(f + XD - 1));
            // memory work, one operation per fragment step
            if (f < 3) halo_piece(HALF * 3 + f, buf ^ 1);
            if (f == 3 || f == 3 + NLDG) asm volatile("" ::: "memory");
            if (f >= 3 && f < 3 + NLDG) {
                const int i = f - 3;
                const char* src;
                if (RES) src = a.res + (eC + i * erow) * 2;
                else if (DDIM) src = (const char*)(a.sched_z + eC + (i >> 1) * erow + (i & 1) * 4);
                else src = (const char*)(a.tvec + (size_t)img * a.t_img_stride + (size_t)step * a.t_step_stride + co + i * 4);
                if (!(WS64_ABLATE & 4)) glds16(src, stage + i * 1024);
            }
            // 18 epilogue slices over the 24 fragment steps (the stores fall on steps 10 and 22)
            if (EPI && f % 4 != 3) epi_slice(slice);
            if (f % 4 != 3) ++slice;
            const int h = f / 6, dx = (f % 6) >> 1, kk = f & 1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = h - dy;
                if (r < 0 || r >= HR) continue;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    acc[nf][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wreg[(dy * 3 + dx) * 2 + kk][nf]), __builtin_bit_cast(bf16x8, xf[f % XD]),
                        acc[nf][r], 0, 0, 0);
            }
            static_assert(HR == 2, "MFMAs per fragment step: one (r, dy) pair for h = 0 and 3, two for h = 1 and 2");
            if (f + XD - 1 < NFR) {
                if (h == 0 || h == HR + 1) __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                else __builtin_amdgcn_sched_group_barrier(0x008, 2 * NF, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < HR; ++r) accP[nf][r] = acc[nf][r];
        eP = eC;
    };
    using W0 = std::integral_constant<int, 0>;
    using W1 = std::integral_constant<int, 1>;
    using W2 = std::integral_constant<int, 2>;

    tile_top(BoolC<true>{});
    WS_STAMP(1);
    do_half(BoolC<false>{}, BoolC<false>{}, W0{});
    do_half(BoolC<true>{}, BoolC<true>{}, W2{});
    WS_STAMP(2);
    tile += tstride;
    buf ^= 1;
    for (kt = 1; kt < nmine; ++kt, tile += tstride, buf ^= 1) {
        tile_top(BoolC<false>{});
        do_half(BoolC<false>{}, BoolC<true>{}, W0{});
        do_half(BoolC<true>{}, BoolC<true>{}, W1{});
        if (kt < 4) WS_STAMP(2 + kt);
    }
    // the last half's epilogue
    wait_vmcnt<0>();
    read_staged();
#pragma unroll
    for (int s = 0; s < KS; ++s) epi_slice(s);
    WS_STAMP_END();

    if (DDIM && a.sched_ticket) {
        // Advance the sampler's step counter once per launch.  Every wave read the counter before its
        // first tile, and a workgroup takes its ticket only after all its waves arrived here, so the
        // workgroup that draws the last ticket knows that nobody will read the old value any more.
        __syncthreads();
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(a.sched_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (int)gridDim.x - 1) {
                __hip_atomic_store(a.sched_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(const_cast<int*>(a.step_ptr), step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused ResidualBlock of the 64-channel level on 64-pixel-wide images (bf16, running statistics folded in):
//   y = silu(conv1(x) + b1) + t ;  out = silu(conv2(y) + b2 + x)                        (src/main.rs:372-387)
// as ONE kernel: y never goes to memory and x is read once (plus once, from L2, as the residual).
//
// A workgroup owns a band of RF_R = 16 output rows of one image over the whole width and walks down it two rows per
// step.  Full-width rows make every 16-pixel MFMA fragment a piece of ONE image row, so conv1 is recomputed only on
// the two rim rows above and below the band (18 rows of y for 16 rows of output).  Eight waves: waves 0-3 compute
// conv1 ("producers"), waves 4-7 conv2 ("consumers"); a wave owns 32 pixel columns x 32 output channels and keeps its
// convolution's 144 weight registers for the whole kernel, as in the kernel above; every SIMD hosts one wave of each.
//   x rows enter an 8-row LDS ring by LDS-DMA, a pair of rows per step, issued two steps ahead;
//   producers at step k write the y rows 2k-1 and 2k (band relative) as bf16 into a second 8-row LDS ring;
//   consumers run three steps behind and read the four y rows their two output rows need from that ring.
// Ring rows hold the 64 pixels of an image row in the pair-plane layout (no border columns): a lane whose pixel lies
// left or right of the image reads a zero pixel kept behind the ring instead (one select per affected read).  Rows
// above or below the image are zeros: the x ring gets them from the zero page, the y ring from the producers.
// One barrier per step; per half (16 columns) a wave runs the 72 MFMAs of two output rows with the previous half's
// epilogue and its share of the memory operations spread between them, exactly as the kernel above does.
// ---------------------------------------------------------------------------------------------
constexpr int RF_R = 16;                                // output rows per workgroup
constexpr int RF_RING = 8;                              // rows of each ring
constexpr int RF_ROWB = 64 * 32;                        // bytes of a ring row in one pair plane
constexpr int RF_PLANE = (RF_RING * 64 + 1) * 32;       // a pair plane: the ring rows, then the zero pixel
constexpr int RF_RINGB = 4 * RF_PLANE;
constexpr int RF_NPROD = RF_R / 2 + 1;                  // producer steps (18 rows of y)
constexpr int RF_LAG = 3;                               // consumers run this many steps behind
constexpr int RF_NS = RF_R / 2 + RF_LAG;                // steps of a workgroup
// two rings, a 1-KiB dump area per wave, the staging rows of the epilogue inputs (RF_RES: 2 KiB of residual per consumer
// wave; RF_DDIM: 4 KiB of sampler state per producer wave), and 128 bytes per wave: the bias of its 32 output channels
constexpr int rf_bias_off(int rm) { return 2 * RF_RINGB + 8 * 1024 + (rm == 0 ? 8 : 16) * 1024; }
constexpr int rf_lds(int rm) { return rf_bias_off(rm) + 8 * 128; }

// RM: what the two chained convolutions are.
//   RF_RES   the ResidualBlock described above.
//   RF_DDIM  the end of one sampler step and the start of the next (bf16 DDIM sampling, src/main.rs:486-492, :499, :525):
//            producers run out_conv and the DDIM update of the f32 state (as tapconv_ws64_kernel<WS_DDIM>: the predicted
//            noise never goes to memory; state rows are staged through LDS and read from a.sched_z, the new state goes to
//            a.sched_z_out - two buffers, because the rim rows of a band are recomputed by its neighbours and must see
//            the old state), the bf16 copy of the new state goes into the y ring only, and consumers run down1 of the
//            NEXT step on it (plain epilogue).  The bf16 copy is never written to memory and one launch is saved.
enum { RF_RES = 0, RF_DDIM = 1 };
template <int RM>
__global__ __launch_bounds__(512, 2) void resfuse64_kernel(const TapConvArgs a) {
    constexpr int NF = 2, KS = 18, HR = 2, XD = 3;
    constexpr bool DDIM = RM == RF_DDIM;
    constexpr int RF_BIAS = rf_bias_off(RM);
    // staged epilogue inputs per half (1-KiB pieces) and stores per epilogue row, by role
    constexpr int STG_P = DDIM ? 2 * HR : 0, STG_C = DDIM ? 0 : HR;
    constexpr int NSTO_P = DDIM ? 2 : 0, NSTO_C = 1;
#ifdef TAPCONV_STAMP
    // diagnostic builds: 100-MHz real-time stamps of wave 0 (producer) and wave 4 (consumer), 16 slots each: start,
    // prologue done, end of every step, end (tools/rflife.py prints them)
#define RF_STAMP(k_) do { if (a.dbg_rt && (threadIdx.x & 255) == 0) a.dbg_rt[(blockIdx.x * 2 + (threadIdx.x >> 8)) * 16 + (k_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RF_STAMP(k_) do { } while (0)
#endif
    RF_STAMP(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave >= 4;
    const int wn = wave & 1, wf = (wave >> 1) & 1;
    const int g = lane >> 4, l15 = lane & 15;
    const char* zero = (const char*)g_zero16;
    const int bands = a.Hin / RF_R;
    const int img = blockIdx.x / bands, y0 = (blockIdx.x % bands) * RF_R;
    char* const xring = smem;
    char* const yring = smem + RF_RINGB;
    char* const dump = smem + 2 * RF_RINGB + wave * 1024;
    // the staged rows of the half whose epilogue is pending, 1 KiB per piece (16 bytes per lane): the residual rows of a
    // consumer (RF_RES), the state rows of a producer as two pieces each (RF_DDIM)
    char* const stage = smem + 2 * RF_RINGB + 8 * 1024 + (wave & 3) * (DDIM ? 4096 : 2048);

    // the zero pixel of both rings (read by lanes whose pixel lies beside the image)
    if (tid < 16) {
        const int ring = tid >> 3, plane = (tid >> 1) & 3, sub = tid & 1;
        *(u32x4*)(smem + ring * RF_RINGB + plane * RF_PLANE + RF_RING * RF_ROWB + sub * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    // x rows: pair q holds the band-relative rows 2q-2 and 2q-1 in the ring rows (2q & 7) and (2q & 7) + 1.  Its 16
    // pieces (4 planes x 2 rows x 2 halves of a row) are issued by the 8 waves, two each: piece e = wave * 2 + i.
    const size_t img_base = (size_t)img * a.Hin * 64;
    auto issue_piece = [&](int q, int i, bool live) {
        // (RF_DDIM consumers: an opaque copy of the lane index, so that the lane's part of the address is formed here and
        // not once in front of both roles, where it is then spilled across the producers' loop)
        int ln = lane;
        if constexpr (DDIM) {
            if (consumer) asm volatile("" : "+v"(ln));
        }
        const int e = wave * 2 + i;
        const int plane = e >> 2, rr = (e >> 1) & 1, hx = (e & 1) * 32 + (ln >> 1);
        const int iy = y0 + 2 * q - 2 + rr;
        const bool ok = live && (unsigned)iy < (unsigned)a.Hin;
        const char* src = ok ? a.in0 + (img_base + (size_t)iy * 64 + hx) * 128 + plane * 32 + (ln & 1) * 16 : zero;
        char* dst = live ? xring + plane * RF_PLANE + (((2 * q) & 7) + rr) * RF_ROWB + (e & 1) * 1024 : dump;
        glds16(src, dst);
    };
    constexpr int QMAX = RF_NPROD;                          // the last producer step needs the pairs QMAX-1 and QMAX
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        issue_piece(q, 0, true);
        issue_piece(q, 1, true);
    }

    // this wave's 32 output channels of every (tap, half) of ITS convolution
    // (Loading the consumers' weights after the first step's barrier instead - their first MFMA is RF_LAG steps away -
    // shortens the prologue from 5.1 to 3.8 us and lengthens the next two steps by as much: measured, not kept.)
    const char* wsrc = consumer ? a.w2 : a.w;
    u32x4 wreg[KS][NF];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            wreg[ks][nf] = *(const u32x4*)(wsrc + (ks >> 1) * 8192 + ((ks & 1) * 2 + (g >> 1)) * 2048 +
                                           ((wn * NF + nf) * 16 + l15) * 32 + (g & 1) * 16);

    // after the lane swap of the epilogue this lane owns channels co .. co+7 of pixel column l15
    const int co = wn * 32 + (g & 1) * 16 + (g >> 1) * 8;
    const float* bsrc = consumer ? a.bias2 : a.bias;
    // A half's accumulators start from the bias (in their own layout: fragment nf, rows 4g .. 4g+3), so its first MFMAs
    // add onto it.  The eight values per lane are read from LDS at the top of every half: as registers held for the
    // whole kernel they cost this kernel four spilled registers.
    char* const biasl = smem + RF_BIAS + wave * 128;
    if (lane < 32) ((float*)biasl)[lane] = bsrc[wn * 32 + lane];
    const int step = a.step_ptr ? *a.step_ptr : 0;
    // ld: the residual rows of the half whose epilogue is pending (consumers), or the block's time vector of this
    // lane's eight channels (producers)
    u32x4 ld[DDIM ? 2 * HR : HR];
#pragma unroll
    for (int i = 0; i < (DDIM ? 2 * HR : HR); ++i) ld[i] = u32x4{0u, 0u, 0u, 0u};
    float k_s1m = 0.f, k_rsa = 0.f, k_c1 = 0.f, k_c2 = 0.f;
    if constexpr (DDIM) {
        if (!consumer) {
            const float* cf = a.sched_coef + (size_t)step * 8;
            const bool last = cf[6] != 0.f;
            // (wave-uniform values: kept in scalar registers, this mode has no vector register to spare)
            auto uni = [](float f) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f))); };
            k_s1m = uni(cf[1]);
            k_rsa = uni(1.0f / cf[0]);
            k_c1 = uni(last ? 1.0f : cf[2]);
            k_c2 = uni(last ? 0.0f : cf[3]);
        }
    } else if (!consumer) {
        const float* tp = a.tvec + (size_t)img * a.t_img_stride + (size_t)step * a.t_step_stride + co;
        ld[0] = *(const u32x4*)tp;
        ld[1] = *(const u32x4*)(tp + 4);
    }
    // (ordinary loads, issued behind the first window's pieces so that the two latencies overlap; the wait for
    // everything in front of the first barrier covers them, and the counted waits of the loop only ever see pieces,
    // residual rows and stores)


    // per-lane pieces of the fragment addresses: plane and half of the lane, pixel offsets of the three column shifts
    const unsigned lane_po = (g >> 1) * RF_PLANE + (g & 1) * 16;
    const unsigned zpix = RF_RING * RF_ROWB + lane_po;     // the zero pixel, in this lane's plane
    // output addressing: element offset of (row 0 of the band, column l15, channel co)
    const size_t out_base = (img_base + (size_t)y0 * 64 + l15) * 64 + co;

    f32x4 accP[NF][HR];
    int rowP = 0, colP = 0;        // first row (band relative) and first column of the half whose epilogue is pending
    f32x4 v0, v1;

    // Slice s (0..17) of the pending half's epilogue: row s/9; step 0 swaps lanes and adds bias (and residual), steps
    // 1..7 apply SiLU, step 8 finishes and writes: producers to the y ring, consumers to memory.
    auto epi_slice = [&](auto role_tag, int s18) {
        constexpr bool CONS = decltype(role_tag)::value;
        const int j = s18 / 9, s = s18 % 9;
        if (s == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto sw = __builtin_amdgcn_permlane16_swap(f2u(accP[0][j][q]), f2u(accP[1][j][q]), false, false);
                v0[q] = u2f(sw[0]);
                v1[q] = u2f(sw[1]);
            }
            if (CONS && !DDIM) {
                const unsigned int u0 = ld[j][0], u1 = ld[j][1], u2 = ld[j][2], u3 = ld[j][3];
                v0[0] += bf2f((bf16_t)(u0 & 0xffffu)); v0[1] += bf2f((bf16_t)(u0 >> 16));
                v0[2] += bf2f((bf16_t)(u1 & 0xffffu)); v0[3] += bf2f((bf16_t)(u1 >> 16));
                v1[0] += bf2f((bf16_t)(u2 & 0xffffu)); v1[1] += bf2f((bf16_t)(u2 >> 16));
                v1[2] += bf2f((bf16_t)(u3 & 0xffffu)); v1[3] += bf2f((bf16_t)(u3 >> 16));
            }
        } else if (s < 8) {
            if constexpr (!DDIM) {
                if (s == 1) { v0[0] = silu_f<false>(v0[0]); v0[1] = silu_f<false>(v0[1]); }
                if (s == 2) v0[2] = silu_f<false>(v0[2]);
                if (s == 3) v0[3] = silu_f<false>(v0[3]);
                if (s == 4) v1[0] = silu_f<false>(v1[0]);
                if (s == 5) v1[1] = silu_f<false>(v1[1]);
                if (s == 6) v1[2] = silu_f<false>(v1[2]);
                if (s == 7) v1[3] = silu_f<false>(v1[3]);
            }
        } else {
            const int row = rowP + j;                      // band relative
            if (!CONS) {
                // y rows above / below the image are the zero padding of conv2
                const bool inside = (unsigned)(y0 + row) < (unsigned)a.Hin;
                if constexpr (DDIM) {
                    // v = predicted noise; the update of tapconv_ws64_kernel<WS_DDIM>, operation for operation
                    f32x4 z0, z1;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        z0[q] = u2f(ld[2 * j][q]);
                        z1[q] = u2f(ld[2 * j + 1][q]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        z0[q] = ((z0[q] - v0[q] * k_s1m) * k_rsa) * k_c1 + v0[q] * k_c2;
                        z1[q] = ((z1[q] - v1[q] * k_s1m) * k_rsa) * k_c1 + v1[q] * k_c2;
                    }
                    // the band's own rows go to the new state; the rim rows belong to the neighbouring bands
                    if ((unsigned)row < (unsigned)RF_R) {
                        float* zo = a.sched_z_out + out_base + ((size_t)row * 64 + colP) * 64;
                        // The state is a stream: 67 MB read and 67 MB written per step, each byte touched once, next
                        // to activations that the following kernels read again within microseconds.  Loaded and stored
                        // non-temporally it stops displacing them from the memory-side cache: this kernel is no
                        // faster (54.6 -> 56.1 us), every other kernel of the step is (+3.2 % on the 50-step benchmark).
                        __builtin_nontemporal_store(z0, (f32x4*)zo);
                        __builtin_nontemporal_store(z1, (f32x4*)(zo + 4));
                    }
                    v0 = z0;
                    v1 = z1;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned int t0 = ld[0][q], t1 = ld[1][q];
                        v0[q] += u2f(t0);
                        v1[q] += u2f(t1);
                    }
                }
                u32x4 o;
                o[0] = inside ? pack2(v0[0], v0[1]) : 0u;
                o[1] = inside ? pack2(v0[2], v0[3]) : 0u;
                o[2] = inside ? pack2(v1[0], v1[1]) : 0u;
                o[3] = inside ? pack2(v1[2], v1[3]) : 0u;
                // y row index i = row + 1; channels co .. co+7 are 16-byte piece co/8 of the pixel: plane piece >> 1
                const unsigned pc = (unsigned)co >> 3;
                *(u32x4*)(yring + (pc >> 1) * RF_PLANE + (((row + 1) & 7) * 64 + colP + l15) * 32 + (pc & 1) * 16) = o;
            } else {
                u32x4 o;
                o[0] = pack2(v0[0], v0[1]);
                o[1] = pack2(v0[2], v0[3]);
                o[2] = pack2(v1[0], v1[1]);
                o[3] = pack2(v1[2], v1[3]);
                store_out(a.out + (out_base + ((size_t)row * 64 + colP) * 64) * 2, o);
            }
        }
    };

    // One half: output rows r0, r0+1 (band relative) x the 16 columns from c0 of this wave's convolution.  ring: the
    // ring it reads; jb: ring row index of window row 0 (the row above r0).  piece_q >= 0: the pair whose piece this
    // half issues.  EPI: a previous half's epilogue is pending (it runs between the MFMAs).
    // drain: the previous half issued no stores (it was this wave's first), so its residual loads are the youngest
    // operations and the wait must be for everything.
    auto do_half = [&](auto role_tag, auto half_tag, auto epi_tag, const char* ring, int jb, int r0, int c0, int piece_q,
                       bool piece_live, bool drain) {
        constexpr bool CONS = decltype(role_tag)::value;
        constexpr int HALF = decltype(half_tag)::value ? 1 : 0;
        constexpr bool EPI = decltype(epi_tag)::value;
        // Vector-memory operations of a consumer half, in issue order: the DMA piece (f = 0), the residual row 0
        // (f = 1), the store of the pending epilogue's row 0 (f = 10), the residual row 1 (f = 13), the store of row 1
        // (f = 22).  Row 0 of the PREVIOUS half's residual is used at f = 0: three younger operations may still fly.
        // RF_DDIM, producer half: the DMA piece (f = 0), the four state pieces (f = 1 .. 4), the two stores of the pending
        // epilogue's row 0 (f = 10) and of its row 1 (f = 22).  The state pieces of the previous half are read here; what
        // that half issued behind them are the stores of the half before IT, and a counted wait must not allow more
        // operations in flight than were issued: the stores of a rim row are not issued, so the halves of step 0 (rows
        // -1 and 0) leave two stores behind the state pieces of the next half, not four - they are waited for at the
        // top of the halves of step 1 (r0 = 1).
        constexpr int STG = CONS ? STG_C : STG_P;
        if (EPI && STG > 0) {
            if (drain) wait_vmcnt<0>();
            else if (DDIM && r0 == 1) wait_vmcnt<NSTO_P>();
            else wait_vmcnt<DDIM ? 2 * NSTO_P : 1>();
#pragma unroll
            for (int i = 0; i < STG; ++i) ld[i] = *(const u32x4*)(stage + i * 1024 + lane * 16);
        }
        f32x4 acc[NF][HR];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < HR; ++r) acc[nf][r] = *(const f32x4*)(biasl + (nf * 4 + g) * 16);
        // addresses of the window rows and column shifts; a lane beside the image reads the zero pixel
        unsigned rowb[HR + 2], pxo[3];
        bool edge[3];
#pragma unroll
        for (int h = 0; h < HR + 2; ++h) rowb[h] = ((jb + h) & 7) * RF_ROWB + lane_po;
        // (RF_DDIM consumers: an opaque copy, so that what depends on the column is computed here and not once in front
        // of both roles, where it is then spilled across the producers' loop)
        int lx = l15;
        if constexpr (CONS && DDIM) asm volatile("" : "+v"(lx));
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int px = c0 + lx + dx - 1;
            edge[dx] = (unsigned)px > 63u;
            pxo[dx] = (unsigned)px * 32u;
        }
        constexpr int NFR = (HR + 2) * 6;
        auto fr = [&](int f) -> const u32x4* {
            const int h = f / 6, dx = (f % 6) >> 1, kk = f & 1;
            const unsigned off = edge[dx] ? zpix : rowb[h] + pxo[dx];
            return (const u32x4*)(ring + off + kk * 2 * RF_PLANE);
        };
        u32x4 xf[XD];
#pragma unroll
        for (int f = 0; f < XD - 1; ++f) xf[f] = *fr(f);
        int slice = 0;
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
            if (f + XD - 1 < NFR) xf[(f + XD - 1) % XD] = *fr(f + XD - 1);
            // memory work, one operation per fragment step: the DMA piece, then (consumers) the residual rows
            if (f == (CONS && !DDIM ? 2 : 0)) issue_piece(piece_q, HALF, piece_live);
            if (EPI && f % 4 != 3) epi_slice(role_tag, slice);
            if (f % 4 != 3) ++slice;
            if (CONS && !DDIM && f < 2) {
                // this half's residual row f into its staging row, which was read back at the top
                const int i = f;
                asm volatile("" ::: "memory");
                glds16(a.in0 + (out_base + ((size_t)(r0 + i) * 64 + c0) * 64) * 2, stage + i * 1024);
                asm volatile("" ::: "memory");
            }
            if (!CONS && DDIM && f >= 1 && f <= 4) {
                // the state of this half's row (f - 1) / 2, channels co + 4 ((f - 1) & 1) .. + 3, into its staging piece; a
                // row outside the image (its y row is zero padding) reads the zero page
                const int i = f - 1, row = r0 + (i >> 1);
                const bool ok = (unsigned)(y0 + row) < (unsigned)a.Hin;
                asm volatile("" ::: "memory");
                glds16_nt(ok ? (const char*)(a.sched_z + out_base + ((long)row * 64 + c0) * 64 + (i & 1) * 4) : zero, stage + i * 1024);
                asm volatile("" ::: "memory");
            }
            const int h = f / 6, dx = (f % 6) >> 1, kk = f & 1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int r = h - dy;
                if (r < 0 || r >= HR) continue;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    acc[nf][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wreg[(dy * 3 + dx) * 2 + kk][nf]), __builtin_bit_cast(bf16x8, xf[f % XD]),
                        acc[nf][r], 0, 0, 0);
            }
            if (f + XD - 1 < NFR) {
                if (h == 0 || h == HR + 1) __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                else __builtin_amdgcn_sched_group_barrier(0x008, 2 * NF, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < HR; ++r) accP[nf][r] = acc[nf][r];
        rowP = r0;
        colP = c0;
    };
    auto flush_epilogue = [&](auto role_tag) {      // the epilogue of the last half of a role
        constexpr int STG = decltype(role_tag)::value ? STG_C : STG_P;
        if (STG > 0) {
            wait_vmcnt<0>();
#pragma unroll
            for (int i = 0; i < STG; ++i) ld[i] = *(const u32x4*)(stage + i * 1024 + lane * 16);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) epi_slice(role_tag, s);
    };
    // end of a step: the pieces of the previous step have landed (this step's two may still fly, consumers also keep
    // their last two stores in flight: vmcnt retires in order, so "at most N outstanding" covers everything older),
    // the y rows written in this step are in LDS, and every wave is done with the rows the next step overwrites
    auto step_end = [&](int in_flight) {
        // in_flight: how many of this wave's youngest operations, all issued in this step, may still fly
        if (in_flight >= 10) wait_vmcnt<10>();
        else if (in_flight >= 8) wait_vmcnt<8>();
        else if (in_flight >= 6) wait_vmcnt<6>();
        else wait_vmcnt<2>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    RF_STAMP(1);
    const int c_first = wf * 32;
    // the two roles run the same loop, each compiled for itself (no role test around a hand-written load)
    auto run = [&](auto role_tag) {
        constexpr bool CONS = decltype(role_tag)::value;
        bool pending = false;
        for (int k = 0; k < RF_NS; ++k) {
            const int pq = k + 3;                               // the pair issued in this step
            const bool plive = pq <= QMAX;
            const bool active = CONS ? k >= RF_LAG : k < RF_NPROD;
            if (!active) {
                if (pending) {                                  // producers after their last step
                    flush_epilogue(role_tag);
                    pending = false;
                }
                issue_piece(pq, 0, plive);
                issue_piece(pq, 1, plive);
                step_end(2);
                RF_STAMP(2 + k);
                continue;
            }
            // producers: y rows 2k-1, 2k from the x rows 2k-2 .. 2k+1 = ring rows 2k .. 2k+3;
            // consumers: output rows o, o+1 (o = 2(k - LAG)) from the y rows o-1 .. o+2 = ring rows o .. o+3
            const int o = 2 * (k - RF_LAG);
            const char* ring = CONS ? yring : xring;
            const int jb = CONS ? o : 2 * k;
            const int r0 = CONS ? o : 2 * k - 1;
            if (!pending) do_half(role_tag, BoolC<false>{}, BoolC<false>{}, ring, jb, r0, c_first, pq, plive, false);
            else do_half(role_tag, BoolC<false>{}, BoolC<true>{}, ring, jb, r0, c_first, pq, plive, false);
            do_half(role_tag, BoolC<true>{}, BoolC<true>{}, ring, jb, r0, c_first + 16, pq, plive, !pending);
            pending = true;
            // RF_RES: a consumer step issues ten operations (per half: two residual rows, the piece, two stores); the
            // youngest eight start with the first half's piece.  RF_DDIM: a consumer step behind its first issues six
            // (per half the piece and two stores), a producer step at least ten (per half the piece and four state
            // pieces, then up to four stores).
            if constexpr (DDIM) step_end(CONS ? (k > RF_LAG ? 6 : 2) : 10);
            else step_end(CONS && k > RF_LAG ? 8 : 2);
            RF_STAMP(2 + k);
        }
        if (pending) flush_epilogue(role_tag);
    };
    if (consumer) run(BoolC<true>{});
    else run(BoolC<false>{});
    RF_STAMP(2 + RF_NS);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (DDIM && a.sched_ticket) {
        // advance the sampler's step counter once per launch (see tapconv_ws64_kernel): every wave read the counter at
        // its start, and a workgroup draws its ticket when all its waves are here
        __syncthreads();
        if (tid == 0) {
            const int t = __hip_atomic_fetch_add(a.sched_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (int)gridDim.x - 1) {
                __hip_atomic_store(a.sched_ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(const_cast<int*>(a.step_ptr), step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused up-sampling block of the 64-channel level (bf16): convT 4x4 / stride 2 (128 -> 64), channel
// concatenation with the skip tensor and the 1x1 conv over the concatenation (src/main.rs:476-483),
// as ONE weight-stationary kernel.  The 1x1 conv is linear, so its up-sampled half is composed with
// the transposed conv when the weights are packed:
//   out = (W1a . WT) (*) x  +  W1b . skip  +  (b1 + W1a . bT)
// Per sub-pixel phase that is 4 taps x 128 input channels plus one "tap" of 64 skip channels: 18 half
// steps of 32 channels, the same register budget as the 3x3 kernel above (144 weight registers).
// A workgroup has 8 waves: wave = (phase, half of the 64 output channels); all waves share ONE input
// window (the four phases read the same 6x18 pixels, so it crosses L2->LDS once instead of four
// times) and the 8x32 output-resolution tile of the skip tensor, stored per phase.  Tiles of 4x16 input
// pixels (8x32 output pixels), double-buffered; rows are processed in two halves with the previous
// half's epilogue, the stores and the next tile's DMA pieces spread over the MFMAs as above.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void upfuse_ws_kernel(const TapConvArgs a) {
    constexpr int TH = 4, HW = 18, HP = (TH + 2) * HW, NG = (HP + 31) / 32;    // 108 halo pixels, 4 groups
    constexpr int XPLANE = NG * 1024, XCHUNK = 4 * XPLANE;                      // 16 KiB per input chunk
    constexpr int SPLANE = 2 * 1024, SPHASE = 4 * SPLANE;                       // skip: 64 pixels per phase, 8 KiB
    constexpr int TILE_BYTES = 2 * XCHUNK + 4 * SPHASE;                         // 64 KiB
    constexpr int NF = 2, KS = 18, HR = 2, XD = 3, NPW = 8;                     // NPW DMA pieces per wave and tile
    static_assert(NG == 4, "piece numbering");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int phase = wave >> 1, wn = wave & 1;
    const int py = phase >> 1, px = phase & 1;
    const int g = lane >> 4, l15 = lane & 15;
    const char* zero = (const char*)g_zero16;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    int tile = blockIdx.x;          // the launcher never starts more workgroups than tiles

    // weights of this wave's phase and channel half: slabs 0..7 = (chunk, tap), slab 8 = skip
    u32x4 wreg[KS][NF];
    const char* wph = a.w + (size_t)phase * a.w_phase_stride;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            wreg[ks][nf] = *(const u32x4*)(wph + (ks >> 1) * 8192 + ((ks & 1) * 2 + (g >> 1)) * 2048 +
                                           ((wn * NF + nf) * 16 + l15) * 32 + (g & 1) * 16);

    // DMA pieces: waves 0..3 stage the input window (piece e = wave*8 + i: chunk e>>4, group (e>>2)&3,
    // plane e&3), waves 4..7 the skip tile (e -= 32: phase e>>3, group (e>>2)&1, plane e&3).  Per
    // piece a lane keeps its pixel's window coordinates (input) or its offset inside the tile (skip).
    const bool skip_wave = wave >= 4;
    int pcoord[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        if (!skip_wave) {
            const int e = wave * NPW + i;
            const int hp = ((e >> 2) & 3) * 32 + (lane >> 1);
            const int hy = hp / HW, hx = hp - hy * HW;
            pcoord[i] = hp < HP ? (hy << 8 | hx) : -1;
        } else {
            const int e = (wave - 4) * NPW + i;
            const int sp = ((e >> 2) & 1) * 32 + (lane >> 1);
            const int ph = e >> 3;
            // output pixel (2y + py', 2x + px') relative to the tile's output origin
            pcoord[i] = ((2 * (sp >> 4) + (ph >> 1)) * a.Wout + 2 * (sp & 15) + (ph & 1)) * 128;
        }
    }
    const char* pbase;
    int piy0, pix0;
    bool plive;
    auto piece_origin = [&](int t) {
        plive = t < a.ntiles;
        const int tt = plive ? t : 0;
        const int img = tt / tiles_per_img;
        const int trem = tt - img * tiles_per_img;
        const int ty = trem / a.tiles_x, tx = trem % a.tiles_x;
        piy0 = ty * TH - 1;
        pix0 = tx * 16 - 1;
        if (!skip_wave) pbase = a.in0 + (size_t)img * a.Hin * a.Win * 256 + (lane & 1) * 16;
        else pbase = a.in1 + (((size_t)img * a.Hout + ty * 2 * TH) * a.Wout + tx * 32) * 128 + (lane & 1) * 16;
    };
    auto issue_piece = [&](int i, int buf) {
        const int e = (wave & 3) * NPW + i;
        const char* src;
        unsigned dst;
        if (!skip_wave) {
            const int iy = piy0 + (pcoord[i] >> 8), ix = pix0 + (pcoord[i] & 255);
            const bool ok = plive && pcoord[i] >= 0 && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            src = ok ? pbase + (size_t)(iy * a.Win + ix) * 256 + (e >> 4) * 128 + (e & 3) * 32 : zero;
            dst = (e >> 4) * XCHUNK + (e & 3) * XPLANE + ((e >> 2) & 3) * 1024;
        } else {
            src = plive ? pbase + pcoord[i] + (e & 3) * 32 : zero;
            dst = 2 * XCHUNK + (e >> 3) * SPHASE + (e & 3) * SPLANE + ((e >> 2) & 1) * 1024;
        }
        glds16(src, smem + buf * TILE_BYTES + dst);
    };
    piece_origin(tile);
#pragma unroll
    for (int i = 0; i < NPW; ++i) issue_piece(i, 0);

    // per-lane read bases: input window at the phase's origin, skip tile of the phase
    const unsigned xlane = (g >> 1) * XPLANE + ((py * HW + px) + l15) * 32 + (g & 1) * 16;
    const unsigned slane = 2 * XCHUNK + phase * SPHASE + (g >> 1) * SPLANE + l15 * 32 + (g & 1) * 16;
    // byte offset of (half step ks, row r) inside a tile buffer
    auto xoff = [&](int ks, int r) -> unsigned {
        const int sl = ks >> 1, kk = ks & 1;
        if (sl < 8) {
            const int c = sl >> 2, ta = (sl >> 1) & 1, tb = sl & 1;
            return xlane + c * XCHUNK + kk * 2 * XPLANE + ((r + ta) * HW + tb) * 32;
        }
        return slane + kk * 2 * SPLANE + r * 16 * 32;
    };

    const int co = wn * 32 + (g & 1) * 16 + (g >> 1) * 8;
    const size_t erow = (size_t)2 * a.Wout * 64;       // one phase row = two output rows
    // the bias in the accumulators' own layout (fragment nf, rows 4g .. 4g+3): a half's accumulators start from it
    f32x4 bpre[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bpre[nf] = *(const f32x4*)(a.bias + (wn * NF + nf) * 16 + 4 * g);

    f32x4 accP[NF][HR];
    size_t eP = 0;
    auto epilogue_row = [&](int j) {
        f32x4 v0, v1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const auto sw = __builtin_amdgcn_permlane16_swap(f2u(accP[0][j][q]), f2u(accP[1][j][q]), false, false);
            v0[q] = u2f(sw[0]);
            v1[q] = u2f(sw[1]);
        }
        u32x4 o;
        o[0] = pack2(v0[0], v0[1]);
        o[1] = pack2(v0[2], v0[3]);
        o[2] = pack2(v1[0], v1[1]);
        o[3] = pack2(v1[2], v1[3]);
        store_out(a.out + (eP + j * erow) * 2, o);
    };

    int buf = 0;
    size_t e0 = 0;
    // Order of a wave's vector-memory operations in one half: 4 DMA pieces, then the 2 stores of the
    // previous half's rows; "at most 2 outstanding" at a tile top = all pieces have landed.
    auto tile_top = [&](auto first_tag) {
        if (decltype(first_tag)::value) wait_vmcnt<0>();
        else wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        piece_origin(tile + (int)gridDim.x);
        const int img = tile / tiles_per_img;
        const int trem = tile - img * tiles_per_img;
        const int oy0 = (trem / a.tiles_x) * 2 * TH + py, ox0 = (trem % a.tiles_x) * 32 + px;
        e0 = (((size_t)img * a.Hout + oy0) * a.Wout + ox0 + 2 * l15) * 64 + co;
    };
    auto do_half = [&](auto half_tag, auto epi_tag) {
        constexpr int HALF = decltype(half_tag)::value ? 1 : 0;
        constexpr bool EPI = decltype(epi_tag)::value;
        const size_t eC = e0 + HALF * HR * erow;
        f32x4 acc[NF][HR];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < HR; ++r) acc[nf][r] = bpre[nf];
        const char* tb = smem + buf * TILE_BYTES;
        u32x4 xf[XD][HR];
#pragma unroll
        for (int k = 0; k < XD - 1; ++k)
#pragma unroll
            for (int r = 0; r < HR; ++r) xf[k][r] = *(const u32x4*)(tb + xoff(k, HALF * HR + r));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + XD - 1 < KS) {
#pragma unroll
                for (int r = 0; r < HR; ++r)
                    xf[(ks + XD - 1) % XD][r] = *(const u32x4*)(tb + xoff(ks + XD - 1, HALF * HR + r));
            }
            if (ks < NPW / 2) issue_piece(HALF * (NPW / 2) + ks, buf ^ 1);
            if (ks == NPW / 2) asm volatile("" ::: "memory");
            if (EPI && ks == 8) epilogue_row(0);
            if (EPI && ks == 17) epilogue_row(1);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < HR; ++r)
                    acc[nf][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wreg[ks][nf]), __builtin_bit_cast(bf16x8, xf[ks % XD][r]),
                        acc[nf][r], 0, 0, 0);
            if (ks + XD - 1 < KS) {
#pragma unroll
                for (int r = 0; r < HR; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < HR; ++r) accP[nf][r] = acc[nf][r];
        eP = eC;
    };

    tile_top(BoolC<true>{});
    do_half(BoolC<false>{}, BoolC<false>{});
    do_half(BoolC<true>{}, BoolC<true>{});
    tile += gridDim.x;
    buf ^= 1;
    for (; tile < a.ntiles; tile += gridDim.x, buf ^= 1) {
        tile_top(BoolC<false>{});
        do_half(BoolC<false>{}, BoolC<true>{});
        do_half(BoolC<true>{}, BoolC<true>{});
    }
    epilogue_row(0);
    epilogue_row(1);
}

// ---------------------------------------------------------------------------------------------
// Weight-stationary stride-2 convolution: conv3x3 / stride 2, bf16, bias only (down2 and down3 of the UNet).
// The generic kernel streams all weights of its channel block per 64 output pixels here and runs at a sixth
// of the MFMA peak.  Same scheme as tapconv_ws64_kernel: 4 waves, wave w keeps the weights of NF*16 output
// channels in 144 registers (NCH = 1 input chunk of 64 channels: 32 output channels per wave, 64 -> 128;
// NCH = 2: 16 per wave, a workgroup owns a block of 64 output channels, 128 -> 256); tiles of 2 x 16 output
// pixels (5 x 33 input window per chunk, double-buffered, columns de-interleaved into even | odd so that a
// fragment's 16 pixels are consecutive LDS slots); a fragment (chunk, window row, column shift, channel
// half) is read once and serves both output rows where they share the window row; the previous tile's
// epilogue, its stores and the next tile's DMA pieces are spread over the 72 MFMAs of the current tile.
// Workgroup -> (block of 64*NF output channels, first tile); a.nblk_n = workgroups (= tile stride) per block.
// ---------------------------------------------------------------------------------------------
template <int NCH, int NF, int KSPL = 1>
__global__ __launch_bounds__(256 * KSPL, (NCH == 1 ? 2 : 1)) void s2ws_kernel(const TapConvArgs a) {
    // KSPL = 2: eight waves; waves 4-7 take the second input chunk of every tile (NCH = 1 chunk per wave), so a wave
    // holds 144 weight registers, two waves share a SIMD and the workgroup still owns 64 * NF output channels.  A wave
    // finishes ONE output row of the tile: row kg of its own sums plus the partner wave's sums of that row, which cross
    // LDS at the end of the tile (2 KiB per wave, double-buffered by tile parity).
    static_assert(KSPL == 1 || (KSPL == 2 && NCH == 1 && NF == 2), "the chunk split is written for one chunk per wave, two fragments");
    constexpr int NCT = NCH * KSPL;                                              // input chunks of the layer
    constexpr int TR = 2, HH = 2 * TR + 1, HW = 33, HWE = 17, HP = HH * HW;      // 165 window pixels
    constexpr int NG = (HP + 31) / 32;                                           // 6 groups of 32
    // A pair plane holds exactly the 165 window pixels (32 bytes each): the last of the six 32-pixel DMA pieces starts at
    // pixel 133 and overlaps the fifth (same bytes twice) instead of padding the plane to 192 slots.
    constexpr int HPLANE = HP * 32, CHUNK_BYTES = 4 * HPLANE, HALO_BYTES = NCT * CHUNK_BYTES;
    // Windows in LDS: that of tile i + NBUF - 1 is issued during tile i.  Three fit (with the compact planes, also next to
    // the partial sums of the chunk split) and were measured: no faster per tile, and the two windows that every
    // workgroup then requests at the start delay the first tile (13.3 against 12.8 us for 128 -> 256 channels).
    constexpr int NBUF = 2;
    static_assert(NBUF == 2 || NBUF == 3, "two or three windows");
    static_assert(NBUF * HALO_BYTES + (KSPL == 2 ? 2 * 8 * 2048 : 0) <= 160 * 1024, "LDS budget");
    constexpr int KS = NCH * 18, XD = 4, NFR = NCH * HH * 6;                     // 30 fragments per chunk and tile
    constexpr int NPC = NCH * NG;                                                // DMA pieces per wave and tile
    constexpr int CIN_BYTES = NCT * 128, CB = 64 * NF;                           // output channels per workgroup
    static_assert(NG == 6, "six halo pieces per wave, chunk and tile");
    static_assert(NF == 1 || NF == 2, "one or two output-channel fragments per wave");
    static_assert(KS * NF * 4 <= 288, "weights of a wave in registers (288 at one wave per SIMD: the two-chunk, two-fragment form)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_id & 3, kg = wave_id >> 2;        // wave: channel quarter and window plane; kg: chunk group
    const int g = lane >> 4, l15 = lane & 15;
    const char* zero = (const char*)g_zero16;
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int cb = blockIdx.x / a.nblk_n;
    // Tiles of this workgroup: tile, tile + tstride, ... (nmine of them; the launcher never starts more workgroups per
    // block than tiles).  a.early_num > 0 (one channel block, two workgroups per CU): uneven shares for the workgroups b
    // and b + grid/2, as in tapconv_ws64_kernel.
    int tile, tstride, nmine;
    if (a.early_num > 0) {
        const int half = a.nblk_n >> 1, bb = (int)blockIdx.x, pr = bb >= half ? bb - half : bb;
        const int npair = (a.ntiles - pr + half - 1) / half;
        int ne = (npair * a.early_num + 4) >> 3;
        if (ne > npair - 1) ne = npair - 1;
        tstride = half;
        tile = bb >= half ? pr + ne * half : pr;
        nmine = bb >= half ? npair - ne : ne;
    } else {
        tile = blockIdx.x % a.nblk_n;
        tstride = a.nblk_n;
        nmine = (a.ntiles - tile + tstride - 1) / tstride;
    }
    int kt = 0;
    WS_STAMP(0);

    // slab (chunk c, tap t) = 4 pair planes of [cout][32 bytes]
    u32x4 wreg[KS][NF];
    const size_t wplane = (size_t)a.cout * 32, wslab = 4 * wplane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            wreg[ks][nf] = *(const u32x4*)(a.w + (size_t)(kg * NCH * 9 + (ks >> 1)) * wslab + ((ks & 1) * 2 + (g >> 1)) * wplane +
                                           (size_t)(cb * CB + (wave * NF + nf) * 16 + l15) * 32 + (g & 1) * 16);

    // Window DMA through a buffer resource over the tile's image.  Piece (chunk c, group q) of wave w = plane w of the
    // chunk; a lane's byte offset inside the window does not depend on the tile (one register per group, columns
    // de-interleaved: slot -> window pixel), the tile adds one scalar, the chunk an immediate.  With stride 2 and padding 1
    // only the window's first row and first column can lie outside the image (tiles at the top / left border): those
    // lanes get an offset beyond the resource, for which the hardware delivers zeros.  A tile past the end reads tile 0's
    // window.
    int hoff[NG];
    unsigned hbits = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int hp = (q * 32 < HP - 32 ? q * 32 : HP - 32) + (lane >> 1);
        const int hy = hp / HW, hr = hp - hy * HW;
        const int hx = hr < HWE ? 2 * hr : 2 * (hr - HWE) + 1;
        hoff[q] = (hy * a.Win + hx) * CIN_BYTES + (lane & 1) * 16 + wave * 32 + kg * NCH * 128;
        hbits |= (unsigned)((hy == 0) | (hx == 0) << 1) << (2 * q);
    }
    __amdgpu_buffer_rsrc_t hrsrc;
    int horg;
    unsigned hinval;
    const int img_bytes = a.Hin * a.Win * CIN_BYTES;
    auto halo_origin = [&](int t) {
        const int tt = t < a.ntiles ? t : 0;
        const int img = fast_div(tt, tiles_per_img, a.magic_tpi);
        const int trem = tt - img * tiles_per_img;
        const int ty = fast_div(trem, a.tiles_x, a.magic_tx), tx = trem - ty * a.tiles_x;
        hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in0 + (size_t)img * img_bytes), 0, img_bytes, 0x00020000);
        horg = ((ty * TR * 2 - 1) * a.Win + tx * 32 - 1) * CIN_BYTES;
        hinval = hbits & ((unsigned)((ty == 0) | (tx == 0) << 1) * 0x555u);
    };
    auto halo_piece = [&](int p, int buf) {
        const int c = p / NG, q = p % NG;                // c: chunk of this wave (compile time), on top of its group's first
        const unsigned off = ((hinval >> (2 * q)) & 3u) ? 0x80000000u : (unsigned)(horg + hoff[q] + c * 128);   // beyond any image (< 2^31 bytes), and + 16 cannot wrap
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            hrsrc, (__attribute__((address_space(3))) void*)(smem + buf * HALO_BYTES + (kg * NCH + c) * CHUNK_BYTES + wave * HPLANE +
                                                             (q * 32 < HP - 32 ? q * 32 : HP - 32) * 32),
            16, (int)off, 0, 0, 0);
    };
    halo_origin(tile);
#pragma unroll
    for (int p = 0; p < NPC; ++p) halo_piece(p, 0);
    if constexpr (NBUF == 3) {
        halo_origin(1 < nmine ? tile + tstride : a.ntiles);
#pragma unroll
        for (int p = 0; p < NPC; ++p) halo_piece(p, 1);
    }

    const unsigned hlane = (g >> 1) * HPLANE + l15 * 32 + (g & 1) * 16;
    // channels of a lane after the epilogue: NF = 2: 8 consecutive (two fragments exchanged between lane
    // halves); NF = 1: the 4 accumulator rows g*4 .. g*4+3
    const int co = cb * CB + (NF == 2 ? wave * 32 + (g & 1) * 16 + (g >> 1) * 8 : wave * 16 + g * 4);
    const size_t erow = (size_t)a.Wout * a.cout;
    const f32x4 b0 = *(const f32x4*)(a.bias + co);
    const f32x4 b1 = NF == 2 ? *(const f32x4*)(a.bias + co + 4) : f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 accP[NF][TR];
    size_t eP = 0;
    // chunk split: the other group's sums of this wave's row, left in LDS by the partner wave (wave_id ^ 4)
    char* const xchg = smem + NBUF * HALO_BYTES;
    int xpar = 0;                                     // parity of the tile whose sums are in accP
    // (kg is uniform over the wave but not a constant: rows are picked by selects, not by indexing registers)
    auto put_partial = [&](const f32x4 (&acc)[NF][TR], int par) {
        char* dst = xchg + (par * 8 + wave_id) * 2048 + lane * 16;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) *(f32x4*)(dst + nf * 1024) = kg ? acc[nf][0] : acc[nf][TR - 1];
    };
    // own sums of row kg plus the partner's, into row 0 of accP (the row the epilogue then takes)
    auto add_partial = [&]() {
        const char* src = xchg + (xpar * 8 + (wave_id ^ 4)) * 2048 + lane * 16;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) accP[nf][0] = (kg ? accP[nf][TR - 1] : accP[nf][0]) + *(const f32x4*)(src + nf * 1024);
    };
    auto epilogue_row = [&](int j, int jout) {      // j: row of accP; jout: output row of the tile
        if constexpr (NF == 2) {
            f32x4 v0, v1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const auto sw = __builtin_amdgcn_permlane16_swap(f2u(accP[0][j][q]), f2u(accP[NF - 1][j][q]), false, false);
                v0[q] = u2f(sw[0]);
                v1[q] = u2f(sw[1]);
            }
            v0 += b0;
            v1 += b1;
            u32x4 o;
            o[0] = pack2(v0[0], v0[1]);
            o[1] = pack2(v0[2], v0[3]);
            o[2] = pack2(v1[0], v1[1]);
            o[3] = pack2(v1[2], v1[3]);
            store_out(a.out + (eP + jout * erow) * 2, o);
        } else {
            const f32x4 v0 = accP[0][j] + b0;
            uint2 o;
            o.x = pack2(v0[0], v0[1]);
            o.y = pack2(v0[2], v0[3]);
            store_out(a.out + (eP + jout * erow) * 2, o);
        }
    };

    int buf = 0;
    size_t e0 = 0;
    // Per tile and wave, in issue order: the NPC DMA pieces of the window NBUF - 1 tiles ahead, then the NSTW stores of the
    // previous tile's rows (none in the first tile).  Before tile i ITS window must have landed; younger than it are the
    // pieces of the NBUF - 2 windows issued since and, from the third tile on, the stores of the tile before (with three
    // windows, from the fourth tile on, also the stores of the tile before that: those are waited for as well, they were
    // issued more than a tile ago).
    constexpr int NSTW = KSPL == 2 ? 1 : 2;
    auto tile_top = [&](int i) {
        if (i >= 2) wait_vmcnt<(NBUF - 2) * NPC + NSTW>();
        else wait_vmcnt<(NBUF - 2) * NPC>();
        if constexpr (KSPL == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's partial sums are in LDS
        __builtin_amdgcn_s_barrier();
        halo_origin(kt + NBUF - 1 < nmine ? tile + (NBUF - 1) * tstride : a.ntiles);
        const int img = fast_div(tile, tiles_per_img, a.magic_tpi);
        const int trem = tile - img * tiles_per_img;
        const int ty = fast_div(trem, a.tiles_x, a.magic_tx);
        const int y0 = ty * TR, x0 = (trem - ty * a.tiles_x) * 16;
        e0 = (((size_t)img * a.Hout + y0) * a.Wout + x0 + l15) * a.cout + co;
    };
    auto do_tile = [&](auto epi_tag) {
        constexpr bool EPI = decltype(epi_tag)::value;
        f32x4 acc[NF][TR];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < TR; ++r) acc[nf][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* hp0 = smem + buf * HALO_BYTES + kg * NCH * CHUNK_BYTES + hlane;
        // fragment f = (chunk c, window row h, column shift dx, channel half kk), kk fastest
        auto foff = [&](int f) {
            const int c = f / 30, fr = f % 30;
            const int h = fr / 6, dx = (fr % 6) >> 1, kk = fr & 1;
            return c * CHUNK_BYTES + kk * 2 * HPLANE + (h * HW + (dx & 1) * HWE + (dx >> 1)) * 32;
        };
        u32x4 xf[XD];
#pragma unroll
        for (int f = 0; f < XD - 1; ++f) xf[f] = *(const u32x4*)(hp0 + foff(f));
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
            if (f + XD - 1 < NFR) xf[(f + XD - 1) % XD] = *(const u32x4*)(hp0 + foff(f + XD - 1));
            if (f < NPC) halo_piece(f, buf >= 1 ? buf - 1 : NBUF - 1);
            if (f == NPC) asm volatile("" ::: "memory");
            if constexpr (KSPL == 2) {
                if (EPI && f == NFR / 2 - 1) { add_partial(); epilogue_row(0, kg); }
            } else {
                if (EPI && f == NFR / 2 - 1) epilogue_row(0, 0);
                if (EPI && f == NFR - 4) epilogue_row(1, 1);
            }
            const int c = f / 30, fr = f % 30;
            const int h = fr / 6, dx = (fr % 6) >> 1, kk = fr & 1;
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const int dy = h - 2 * r;
                if (dy < 0 || dy > 2) continue;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    acc[nf][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, wreg[c * 18 + (dy * 3 + dx) * 2 + kk][nf]),
                        __builtin_bit_cast(bf16x8, xf[f % XD]), acc[nf][r], 0, 0, 0);
            }
            if (f + XD - 1 < NFR) {
                if (h == 2) __builtin_amdgcn_sched_group_barrier(0x008, 2 * NF, 0);
                else __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < TR; ++r) accP[nf][r] = acc[nf][r];
        eP = e0;
        if constexpr (KSPL == 2) {
            xpar ^= 1;
            put_partial(acc, xpar);
        }
    };

    auto next_buf = [&]() { buf = buf + 1 == NBUF ? 0 : buf + 1; };
    tile_top(0);
    WS_STAMP(1);
    do_tile(BoolC<false>{});
    WS_STAMP(2);
    tile += tstride;
    next_buf();
    for (kt = 1; kt < nmine; ++kt, tile += tstride) {
        tile_top(kt);
        do_tile(BoolC<true>{});
        if (kt < 4) WS_STAMP(2 + kt);
        next_buf();
    }
    if constexpr (KSPL == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        add_partial();
        epilogue_row(0, kg);
    } else {
        epilogue_row(0, 0);
        epilogue_row(1, 1);
    }
    WS_STAMP_END();
}

// ---------------------------------------------------------------------------------------------
// Weight-stationary 1x1 convolution (token Linear): 256 input channels, output channels in blocks of 256,
// bf16, bias (+ residual).  The fused q|k|v projection of Attention (256 -> 768) is 6.4 GFLOP at B=64; the
// generic kernel re-stages every 256-pixel window once per 128-channel block and spends its time in
// prologues and epilogues (four steps per workgroup).  Here a workgroup of 4 waves owns 256 output
// channels (wave w: 64 of them, 8 half steps x 4 fragments = 128 weight registers) and walks over tiles of
// 32 consecutive pixels of the flattened [B*H*W, 256] activation; tiles are double-buffered in LDS
// (4 DMA pieces per wave), and the previous tile's epilogue and stores are spread over the 64 MFMAs of
// the current one, as in the 3x3 kernels above.
// ---------------------------------------------------------------------------------------------
template <bool RES>
__global__ __launch_bounds__(256, 2) void ws1x1_kernel(const TapConvArgs a) {
    constexpr int TR = 2, NF = 4, KS = 8, XD = 4, NFR = TR * KS;        // 16 fragments, 64 MFMAs per tile
    constexpr int TILE_BYTES = TR * 16 * 512;                           // 32 pixels x 512 bytes

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const char* zero = (const char*)g_zero16;
    // workgroup -> (block of 256 output channels, first tile, tile stride); a.nblk_n = workgroups per block
    const int cb = blockIdx.x / a.nblk_n, first = blockIdx.x % a.nblk_n;
    int tile = first;

    u32x4 wreg[KS][NF];
    const size_t slab = (size_t)a.cout * 128;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            wreg[ks][nf] = *(const u32x4*)(a.w + (ks >> 1) * slab +
                                           ((size_t)((ks & 1) * 2 + (g >> 1)) * a.cout + cb * 256 + wave * 64 + nf * 16 + l15) * 32 +
                                           (g & 1) * 16);

    // tile DMA: wave w stages channel chunk w (128 bytes of every pixel), one piece per pair plane
    auto issue_tile = [&](int t, int buf) {
        const bool live = t < a.ntiles;
        const char* src0 = a.in0 + ((size_t)(live ? t : 0) * 32 + (lane >> 1)) * 512 + wave * 128 + (lane & 1) * 16;
#pragma unroll
        for (int p = 0; p < 4; ++p) glds16(live ? src0 + p * 32 : zero, smem + buf * TILE_BYTES + wave * 4096 + p * 1024);
    };
    issue_tile(tile, 0);

    const unsigned xlane = (g >> 1) * 1024 + l15 * 32 + (g & 1) * 16;
    // after the lane swap a lane owns 8 consecutive channels: pair j covers fragments 2j, 2j+1
    const int co0 = cb * 256 + wave * 64 + (g & 1) * 16 + (g >> 1) * 8;
    f32x4 bia[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        bia[j][0] = *(const f32x4*)(a.bias + co0 + j * 32);
        bia[j][1] = *(const f32x4*)(a.bias + co0 + j * 32 + 4);
    }

    f32x4 accP[NF][TR];
    size_t eP = 0;
    // epilogue unit u = (row u >> 1, channel pair u & 1)
    auto epilogue_unit = [&](int u) {
        const int r = u >> 1, j = u & 1;
        f32x4 v0, v1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const auto sw = __builtin_amdgcn_permlane16_swap(f2u(accP[2 * j][r][q]), f2u(accP[2 * j + 1][r][q]), false, false);
            v0[q] = u2f(sw[0]);
            v1[q] = u2f(sw[1]);
        }
        v0 += bia[j][0];
        v1 += bia[j][1];
        const size_t eo = eP + (size_t)r * 16 * a.cout + j * 32;
        if (RES) {
            const u32x4 rr = *(const u32x4*)(a.res + eo * 2);
            const unsigned int u0 = rr[0], u1 = rr[1], u2 = rr[2], u3 = rr[3];
            v0[0] += bf2f((bf16_t)(u0 & 0xffffu)); v0[1] += bf2f((bf16_t)(u0 >> 16));
            v0[2] += bf2f((bf16_t)(u1 & 0xffffu)); v0[3] += bf2f((bf16_t)(u1 >> 16));
            v1[0] += bf2f((bf16_t)(u2 & 0xffffu)); v1[1] += bf2f((bf16_t)(u2 >> 16));
            v1[2] += bf2f((bf16_t)(u3 & 0xffffu)); v1[3] += bf2f((bf16_t)(u3 >> 16));
        }
        u32x4 o;
        o[0] = pack2(v0[0], v0[1]);
        o[1] = pack2(v0[2], v0[3]);
        o[2] = pack2(v1[0], v1[1]);
        o[3] = pack2(v1[2], v1[3]);
        store_out(a.out + eo * 2, o);
    };

    int buf = 0;
    size_t e0 = 0;
    // per tile and wave: 4 DMA pieces, then the 4 stores of the previous tile.  With a residual the
    // epilogue's plain loads make the compiler drain the queue itself; the counted wait is then moot.
    // (the first tile has no previous tile to store: after it nothing is younger than its DMA pieces)
    auto tile_top = [&](bool stores_behind) {
        if (stores_behind && !RES) wait_vmcnt<4>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        e0 = ((size_t)tile * 32 + l15) * a.cout + co0;
    };
    auto do_tile = [&](auto epi_tag) {
        constexpr bool EPI = decltype(epi_tag)::value;
        f32x4 acc[NF][TR];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < TR; ++r) acc[nf][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* xp = smem + buf * TILE_BYTES + xlane;
        // fragment f = (row r, half step ks): chunk ks >> 1, planes (ks & 1) * 2 + g / 2
        auto foff = [&](int f) {
            const int r = f / KS, ks = f % KS;
            return (ks >> 1) * 4096 + (ks & 1) * 2048 + r * 512;
        };
        u32x4 xf[XD];
#pragma unroll
        for (int f = 0; f < XD - 1; ++f) xf[f] = *(const u32x4*)(xp + foff(f));
#pragma unroll
        for (int f = 0; f < NFR; ++f) {
            if (f + XD - 1 < NFR) xf[(f + XD - 1) % XD] = *(const u32x4*)(xp + foff(f + XD - 1));
            if (f == 0) issue_tile(tile + a.nblk_n, buf ^ 1);
            if (f == 1) asm volatile("" ::: "memory");
            if (EPI && (f & 3) == 3) epilogue_unit(f >> 2);
            const int r = f / KS, ks = f % KS;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                acc[nf][r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, wreg[ks][nf]), __builtin_bit_cast(bf16x8, xf[f % XD]), acc[nf][r], 0, 0, 0);
            if (f + XD - 1 < NFR) {
                __builtin_amdgcn_sched_group_barrier(0x008, NF, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < TR; ++r) accP[nf][r] = acc[nf][r];
        eP = e0;
    };

    if (tile >= a.ntiles) return;     // (not reached: the launcher starts at most one workgroup per tile and block)
    tile_top(false);
    do_tile(BoolC<false>{});
    tile += a.nblk_n;
    buf ^= 1;
    bool stores_behind = false;
    for (; tile < a.ntiles; tile += a.nblk_n, buf ^= 1) {
        tile_top(stores_behind);
        do_tile(BoolC<true>{});
        stores_behind = true;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) epilogue_unit(u);
}

// ---------------------------------------------------------------------------------------------
// 1x1 convolution over a few pixels (a Linear over a handful of rows): the sampler at the reference's own
// sizes, batch 1-4 of 4x4 latents, where every layer is such a product and the run time is the latency of
// streaming the weights.  The tiled kernel gives a problem like that cout/64 workgroups, each walking all
// input chunks with a barrier per chunk.  Here a workgroup owns 16 output channels of up to 64 pixels, its 8 or
// 16 waves split the input chunks, and every fragment comes straight from global memory into registers (weights in the packed
// layout: chunk slab, pair plane, row; activations: the 16-byte piece kk*4 + g of the pixel's chunk), so all
// loads of a wave are in flight at once; the partial sums meet in LDS.  Epilogue as in the tiled kernel.
// ---------------------------------------------------------------------------------------------
template <typename T, int MF, int NWV>
__global__ __launch_bounds__(64 * NWV) void skinny1x1_kernel(const TapConvArgs a) {
    __shared__ __attribute__((aligned(16))) float part[NWV * MF * 16 * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int n0 = blockIdx.x * 16;
    const int p0 = blockIdx.y * (MF * 16);           // first pixel of this workgroup
    const int M = a.B * a.Hin * a.Win;
    const int nchunks = a.nch0 + a.nch1;
    const size_t slab = (size_t)a.cout * 128, plane = (size_t)a.cout * 32;
    const char* zero = (const char*)g_zero16;

    // Epilogue operands of this thread (pixel p0 + tid/4, channels n0 + (tid&3)*4 .. +3), loaded before the
    // main loop: behind it they would add a dependent round trip to memory (step counter -> time vector) to a
    // kernel that lasts three.
    const int pl = tid >> 2, cg = tid & 3;
    const int p = p0 + pl;
    const bool mine = p < M && pl < MF * 16;
    const int co = n0 + cg * 4;
    const size_t eo = (size_t)(mine ? p : 0) * a.cout + co;
    const f32x4 bias = *(const f32x4*)(a.bias + co);
    f32x4 resv = {0.f, 0.f, 0.f, 0.f}, tv = {0.f, 0.f, 0.f, 0.f};
    if (a.res) {
        if constexpr (sizeof(T) == 2) {
            const uint2 r = *(const uint2*)(a.res + eo * 2);
            resv[0] = bf2f((bf16_t)(r.x & 0xffffu)); resv[1] = bf2f((bf16_t)(r.x >> 16));
            resv[2] = bf2f((bf16_t)(r.y & 0xffffu)); resv[3] = bf2f((bf16_t)(r.y >> 16));
        } else {
            resv = *(const f32x4*)(a.res + eo * 4);
        }
    }
    if (a.tvec) {
        const int step = a.step_ptr ? *a.step_ptr : 0;
        const int img = (mine ? p : 0) / (a.Hin * a.Win);
        const int pin = (mine ? p : 0) - img * (a.Hin * a.Win);
        tv = *(const f32x4*)(a.tvec + (size_t)img * a.t_img_stride + (size_t)pin * a.t_pix_stride +
                             (size_t)step * a.t_step_stride + co);
    }

    f32x4 acc[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* wl = a.w + (size_t)(g >> 1) * plane + (size_t)(n0 + l15) * 32 + (g & 1) * 16;
    for (int c = wave; c < nchunks; c += NWV) {
        const bool s1 = c >= a.nch0;
        const char* src = s1 ? a.in1 + (size_t)(c - a.nch0) * 128 : a.in0 + (size_t)c * 128;
        const int pix_bytes = s1 ? a.cin1_bytes : a.cin0_bytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u32x4 wf = *(const u32x4*)(wl + (size_t)c * slab + (size_t)kk * 2 * plane);
            u32x4 xf[MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int q = p0 + mf * 16 + l15;
                xf[mf] = *(const u32x4*)(q < M ? src + (size_t)q * pix_bytes + (kk * 4 + g) * 16 : zero);
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                if constexpr (sizeof(T) == 2) {
                    acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf),
                                                                      __builtin_bit_cast(bf16x8, xf[mf]), acc[mf], 0, 0, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(u2f(wf[q]), u2f(xf[mf][q]), acc[mf], 0, 0, 0);
                }
            }
        }
    }
    // acc[mf] of lane (l15, g): channels n0 + g*4 .. +3 of pixel p0 + mf*16 + l15
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) *(f32x4*)(part + ((wave * MF * 16 + mf * 16 + l15) * 16 + g * 4)) = acc[mf];
    __syncthreads();
    if (!mine) return;
    f32x4 v = *(const f32x4*)(part + (pl * 16 + cg * 4));
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += *(const f32x4*)(part + ((w * MF * 16 + pl) * 16 + cg * 4));
    v += bias;
    v += resv;
    if (a.act == ACT_SILU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = silu_f<sizeof(T) == 4>(v[q]);
    } else if (a.act == ACT_RELU) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
    } else if (a.act == ACT_TANH) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = tanhf(v[q]);
    }
    v += tv;
    if (sizeof(T) == 4 || a.out_f32) {
        *(f32x4*)(a.out + eo * 4) = v;
    } else {
        uint2 o;
        o.x = pack2(v[0], v[1]);
        o.y = pack2(v[2], v[3]);
        *(uint2*)(a.out + eo * 2) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// The same idea for 3x3 convolutions (stride 1 or 2, padding 1) over few pixels: single images or small
// batches, where the tiled kernel's launch is a handful of workgroups that each walk all (chunk, tap) steps
// behind a barrier (9 us per launch whatever the batch, from 1 to 16 images of 64x64).  A workgroup owns 16
// output channels of a 4x16-pixel tile; its waves split the (chunk, tap) pairs; a lane reads its pixel's
// neighbour for the tap straight from global memory (the zero page outside the image), the weights from
// the pair's slab of the packed layout.  Partial sums meet in LDS; epilogue as in the tiled kernel.
// ---------------------------------------------------------------------------------------------
template <typename T, int S, int NWV>
__global__ __launch_bounds__(64 * NWV) void skinny3x3_kernel(const TapConvArgs a) {
    constexpr int MF = 4;
    __shared__ __attribute__((aligned(16))) float part[NWV * MF * 16 * 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l15 = lane & 15;
    const int n0 = blockIdx.x * 16;
    const int tile = blockIdx.y;
    const int img = tile / (a.tiles_x * a.tiles_y);
    const int trem = tile - img * (a.tiles_x * a.tiles_y);
    const int y0 = (trem / a.tiles_x) * MF, x0 = (trem % a.tiles_x) * 16;
    const size_t slab = (size_t)a.cout * 128, plane = (size_t)a.cout * 32;
    const char* zero = (const char*)g_zero16;
    const char* in_img = a.in0 + (size_t)img * a.Hin * a.Win * a.cin0_bytes;

    // epilogue operands first (see skinny1x1_kernel): thread -> pixel (y0 + tid/64, x0 + (tid/4)%16), 4 channels
    const int pl = tid >> 2, cg = tid & 3;
    const int ey = y0 + (pl >> 4), ex = x0 + (pl & 15);
    const bool mine = pl < MF * 16 && ey < a.Hout && ex < a.Wout;
    const int co = n0 + cg * 4;
    const size_t eo = (((size_t)img * a.Hout + (mine ? ey : 0)) * a.Wout + (mine ? ex : 0)) * a.cout + co;
    const f32x4 bias = *(const f32x4*)(a.bias + co);
    f32x4 resv = {0.f, 0.f, 0.f, 0.f}, tv = {0.f, 0.f, 0.f, 0.f};
    if (a.res) {
        if constexpr (sizeof(T) == 2) {
            const uint2 r = *(const uint2*)(a.res + eo * 2);
            resv[0] = bf2f((bf16_t)(r.x & 0xffffu)); resv[1] = bf2f((bf16_t)(r.x >> 16));
            resv[2] = bf2f((bf16_t)(r.y & 0xffffu)); resv[3] = bf2f((bf16_t)(r.y >> 16));
        } else {
            resv = *(const f32x4*)(a.res + eo * 4);
        }
    }
    if (a.tvec) {
        const int step = a.step_ptr ? *a.step_ptr : 0;
        tv = *(const f32x4*)(a.tvec + (size_t)img * a.t_img_stride + (size_t)step * a.t_step_stride + co);
    }

    f32x4 acc[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) acc[mf] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* wl = a.w + (size_t)(g >> 1) * plane + (size_t)(n0 + l15) * 32 + (g & 1) * 16;
    const int npairs = a.nch0 * 9;
    const int ox = x0 + l15;
    // One (chunk, tap) pair per iteration: its ten fragment loads go out together and the wave waits for them
    // once.  (Loading the next pair ahead of the MFMAs was measured slower: the loads that run past a wave's last
    // pair cost more than the overlap gains at 1-5 pairs per wave.)
    for (int q = wave; q < npairs; q += NWV) {
        const int c = q / 9, t = q - c * 9;
        const int dy = t / 3, dx = t - dy * 3;
        const int ix = ox * S + dx - 1;
        const bool xok = ox < a.Wout && (unsigned)ix < (unsigned)a.Win;
        const char* src[MF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int oy = y0 + mf, iy = oy * S + dy - 1;
            const bool ok = xok && oy < a.Hout && (unsigned)iy < (unsigned)a.Hin;
            src[mf] = ok ? in_img + ((size_t)iy * a.Win + ix) * a.cin0_bytes + (size_t)c * 128 + g * 16 : zero;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u32x4 wf = *(const u32x4*)(wl + (size_t)q * slab + (size_t)kk * 2 * plane);
            u32x4 xf[MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) xf[mf] = *(const u32x4*)(src[mf] + (src[mf] == zero ? 0 : kk * 64));
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                if constexpr (sizeof(T) == 2) {
                    acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf),
                                                                      __builtin_bit_cast(bf16x8, xf[mf]), acc[mf], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[mf] = __builtin_amdgcn_mfma_f32_16x16x4f32(u2f(wf[e]), u2f(xf[mf][e]), acc[mf], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) *(f32x4*)(part + ((wave * MF * 16 + mf * 16 + l15) * 16 + g * 4)) = acc[mf];
    __syncthreads();
    if (!mine) return;
    f32x4 v = *(const f32x4*)(part + (pl * 16 + cg * 4));
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += *(const f32x4*)(part + ((w * MF * 16 + pl) * 16 + cg * 4));
    v += bias;
    v += resv;
    if (a.act == ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_f<sizeof(T) == 4>(v[e]);
    } else if (a.act == ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (a.act == ACT_TANH) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
    }
    v += tv;
    if (sizeof(T) == 4 || a.out_f32) {
        *(f32x4*)(a.out + eo * 4) = v;
    } else {
        uint2 o;
        o.x = pack2(v[0], v[1]);
        o.y = pack2(v[2], v[3]);
        *(uint2*)(a.out + eo * 2) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static thread_local const char* g_last_kernel = "";
const char* tapconv_last_kernel() { return g_last_kernel; }

template <typename T> static const char* type_name() { return sizeof(T) == 2 ? "unsigned short" : "float"; }

int conv_out_hw(int kind, int Hin, int Win, int* Hout, int* Wout) {
    switch (kind) {
        case CONV3x3_S1:
        case CONV1x1: *Hout = Hin; *Wout = Win; return LDM_OK;
        case CONV3x3_S2: *Hout = (Hin - 1) / 2 + 1; *Wout = (Win - 1) / 2 + 1; return LDM_OK;
        case CONVT4x4_S2: *Hout = Hin * 2; *Wout = Win * 2; return LDM_OK;
    }
    LDM_FAIL(LDM_ERR_INVALID, "unknown conv kind");
}

template <typename T, int TH, int BN, int WM, int WN, int S, int KEXT, int NTAPS, int MINW, int RING = 3, int NT1 = NTAPS>
static int launch_cfg(TapConvArgs& a, int nphase, hipStream_t stream) {
    constexpr int NWV = WM * WN;
    constexpr int HH = (TH - 1) * S + KEXT, HW = 15 * S + KEXT;
    constexpr int HALO_BYTES = (HH * HW + 31) / 32 * 4096;
    const int nchunks = a.nch0 + a.nch1;
    size_t lds = (size_t)(nchunks > 1 ? 2 : 1) * HALO_BYTES + RING * (size_t)BN * 128 + 1024;
    const size_t staging = (size_t)TH * 16 * (BN * 4 + 16);   // epilogue: f32 tile with padded rows
    if (lds < staging) lds = staging;
#ifdef TAPCONV_STAMP
    lds += 1024 + 4096;
    a.dbg = nullptr;
    if (const char* e = getenv("LDM_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(e, nullptr, 16);
    a.dbg_rt = nullptr;
    a.dbg_rt = dbg_rt_for("tiled");
#endif
    static_assert((size_t)TH * 16 * (BN * 4 + 16) <= 160 * 1024, "epilogue staging exceeds LDS");
    a.tiles_x = (a.Wg + 15) / 16;
    a.tiles_y = (a.Hg + TH - 1) / TH;
    a.ntiles = a.tiles_x * a.tiles_y * a.B;
    a.nblk_n = a.cout / BN;
    if (a.ntaps != NTAPS) LDM_FAIL(LDM_ERR_INVALID, "tap count does not match the kernel");
    auto kern = tapconv_kernel<T, TH, BN, WM, WN, S, KEXT, NTAPS, MINW, RING, NT1>;
    static bool lds_set[kMaxDevices] = {};   // per kernel (instantiation) and device
    LDM_TRY(allow_dynamic_lds((const void*)kern, 160 * 1024, lds_set));
    a.nphase = nphase;
    static const std::string name = std::string("void ldm::tapconv_kernel<") + type_name<T>() + ", " + std::to_string(TH) + ", " +
        std::to_string(BN) + ", " + std::to_string(WM) + ", " + std::to_string(WN) + ", " + std::to_string(S) + ", " +
        std::to_string(KEXT) + ", " + std::to_string(NTAPS) + ", " + std::to_string(MINW) + ", " + std::to_string(RING) + ", " +
        std::to_string(NT1) + ">(ldm::TapConvArgs)";
    g_last_kernel = name.c_str();
    dim3 grid((unsigned)(((a.ntiles + 7) / 8) * 8 * a.nblk_n * nphase), 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWV), lds, stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// Picks the tile height: the tallest tile that still yields at least `want` workgroups.
// rows: height of the tiled grid; a tile taller than the image only multiplies zeros
static int pick_th(long tiles16, long tiles8, int nblk_n, int nphase, long want, bool allow16, int rows) {
    if (allow16 && rows > 8 && tiles16 * nblk_n * nphase >= want) return 16;
    if (rows > 4 && tiles8 * nblk_n * nphase >= want) return 8;
    return 4;
}

// Uneven shares for the two workgroups of a CU (kEarlyEighths of a pair's tiles for the first-placed one) need two
// workgroups on every CU and at least two tiles in every pair: with a single tile both workgroups of the pair would
// run it (the in-place DDIM update twice), with none the tile index goes negative.
static int early_eighths(int grid, int ntiles, int ncu) {
    return (grid == 2 * ncu && ntiles >= 2 * grid) ? kEarlyEighths : 0;
}

template <int MODE>
static int launch_ws64(TapConvArgs& a, hipStream_t stream) {
#ifdef TAPCONV_STAMP
    a.dbg = nullptr;
    a.dbg_rt = nullptr;
    a.dbg_rt = dbg_rt_for("ws64");
#endif
    int ncu = 0;
    LDM_TRY(device_cu_count(&ncu));
    a.tiles_x = a.Wg / 16;
    a.tiles_y = a.Hg / 8;
    a.ntiles = a.tiles_x * a.tiles_y * a.B;
    a.nblk_n = 1;
    const size_t lds = 2 * 6 * 4096 + 4 * 4 * 1024;  // two halo buffers of 10x18 pixels + staged epilogue inputs
    const int grid = a.ntiles < 2 * ncu ? a.ntiles : 2 * ncu;
    a.early_num = early_eighths(grid, a.ntiles, ncu);
    LDM_REQUIRE(a.early_num == 0 || a.ntiles >= 2 * grid, "uneven tile shares need two tiles per pair of workgroups");
    static const std::string name = "void ldm::tapconv_ws64_kernel<" + std::to_string(MODE) + ">(ldm::TapConvArgs)";
    g_last_kernel = name.c_str();
    hipLaunchKernelGGL(tapconv_ws64_kernel<MODE>, dim3(grid), dim3(256), lds, stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// The shapes the weight-stationary kernel serves: 64 -> 64 channels, bf16, exact 8x16 tiling, at least
// one tile per workgroup slot of the chip.
static bool ws64_shape(const PackedConv& pc, const TapConvArgs& a) {
    return a.cin0_bytes == a.nch0 * 128 && pc.dtype == LDM_BF16 && pc.kind == CONV3x3_S1 && a.nch0 == 1 && a.nch1 == 0 && a.cout == 64 &&
           a.Hg % 8 == 0 && a.Wg % 16 == 0 && (long)(a.Hg / 8) * (a.Wg / 16) * a.B >= 512;
}

bool tapconv_can_fuse_ddim(const PackedConv& pc, int B, int H, int W) {
    return pc.dtype == LDM_BF16 && pc.kind == CONV3x3_S1 && pc.cin == 64 && pc.cout == 64 && H % 8 == 0 && W % 16 == 0 &&
           (long)(H / 8) * (W / 16) * B >= 512;
}

static bool s2ws_shape(const PackedConv& pc, const TapConvArgs& a) {
    const bool c64 = a.nch0 == 1 && a.cout == 128, c128 = a.nch0 == 2 && a.cout == 256;
    return a.cin0_bytes == a.nch0 * 128 && pc.dtype == LDM_BF16 && pc.kind == CONV3x3_S2 && (c64 || c128) && a.nch1 == 0 &&
           a.Hin % 4 == 0 && a.Win % 32 == 0 && (long)(a.Hout / 2) * (a.Wout / 16) * a.B >= 512 &&
           // limits of fast_div and of the 32-bit offsets inside one image
           (long)a.B * (a.Hout / 2) * (a.Wout / 16) * (a.Hout / 2) * (a.Wout / 16) < (1l << 32) && (long)a.Hin * a.Win * a.cin0_bytes < (1l << 31) &&
           true;
}
static int launch_s2ws(TapConvArgs& a, hipStream_t stream) {
#ifdef TAPCONV_STAMP
    a.dbg = nullptr;
    a.dbg_rt = nullptr;
    a.dbg_rt = dbg_rt_for("s2ws");
#endif
    int ncu = 0;
    LDM_TRY(device_cu_count(&ncu));
    a.tiles_x = a.Wout / 16;
    a.tiles_y = a.Hout / 2;
    a.ntiles = a.tiles_x * a.tiles_y * a.B;
    a.magic_tpi = div_magic(a.tiles_x * a.tiles_y);
    a.magic_tx = div_magic(a.tiles_x);
    if (a.nch0 == 1) {
        const size_t lds = 2 * 165 * 128;               // two windows of 5 x 33 pixels
        a.nblk_n = a.ntiles < 2 * ncu ? a.ntiles : 2 * ncu;
        a.early_num = early_eighths(a.nblk_n, a.ntiles, ncu);
        LDM_REQUIRE(a.early_num == 0 || a.ntiles >= 2 * a.nblk_n, "uneven tile shares need two tiles per pair of workgroups");
        g_last_kernel = "void ldm::s2ws_kernel<1, 2, 1>(ldm::TapConvArgs)";
        hipLaunchKernelGGL((s2ws_kernel<1, 2>), dim3(a.nblk_n), dim3(256), lds, stream, a);
    } else {
        // two chunks split over two groups of four waves: one workgroup of eight waves per CU, 128 output channels each
        // (one group of four waves over both chunks, 64 or 128 channels per workgroup, measured slower)
        static bool lds_set[kMaxDevices] = {};   // per kernel (instantiation) and device
        LDM_TRY(allow_dynamic_lds((const void*)s2ws_kernel<1, 2, 2>, 160 * 1024, lds_set));
        const size_t lds = 2 * 2 * 165 * 128 + 2 * 8 * 2048;   // two windows of two chunks + the partial sums of eight waves, twice
        const int ncb = a.cout / 128;
        int per = ncu / ncb;
        if (per < 1) per = 1;
        if (per > a.ntiles) per = a.ntiles;
        a.nblk_n = per;
        g_last_kernel = "void ldm::s2ws_kernel<1, 2, 2>(ldm::TapConvArgs)";
        hipLaunchKernelGGL((s2ws_kernel<1, 2, 2>), dim3(per * ncb), dim3(512), lds, stream, a);
    }
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

static bool ws1x1_shape(const PackedConv& pc, const TapConvArgs& a) {
    const long px = (long)a.B * a.Hin * a.Win;
    return a.cin0_bytes == a.nch0 * 128 && pc.dtype == LDM_BF16 && pc.kind == CONV1x1 && a.nch0 == 4 && a.nch1 == 0 && a.cout % 256 == 0 && px % 32 == 0 &&
           px / 32 >= 512;
}

template <bool RES>
static int launch_ws1x1(TapConvArgs& a, hipStream_t stream) {
    int ncu = 0;
    LDM_TRY(device_cu_count(&ncu));
    const int ncb = a.cout / 256;
    a.ntiles = (int)((long)a.B * a.Hin * a.Win / 32);
    int per = 2 * ncu / ncb;                       // workgroups per block of 256 output channels
    if (per > a.ntiles) per = a.ntiles;
    a.nblk_n = per;
    const size_t lds = 2 * 16 * 1024;              // two tiles of 32 pixels x 512 bytes
    g_last_kernel = RES ? "void ldm::ws1x1_kernel<true>(ldm::TapConvArgs)" : "void ldm::ws1x1_kernel<false>(ldm::TapConvArgs)";
    hipLaunchKernelGGL(ws1x1_kernel<RES>, dim3((unsigned)(per * ncb)), dim3(256), lds, stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

template <typename T>
static int launch_skinny1x1(TapConvArgs& a, hipStream_t stream) {
    const int M = a.B * a.Hin * a.Win;
    // waves that split the input chunks (measured at 16 pixels, 1024 -> 1024: bf16 3.8 / 3.5 / 3.7 us per launch with
    // 4 / 8 / 16 waves; f32, twice the chunks, 6.0 / 4.6 / 4.3 us)
    constexpr int NWV = sizeof(T) == 2 ? 8 : 16;
    const dim3 block(64 * NWV);
    g_last_kernel = "ldm::skinny1x1_kernel";
    if (M <= 16) hipLaunchKernelGGL((skinny1x1_kernel<T, 1, NWV>), dim3((unsigned)(a.cout / 16), 1), block, 0, stream, a);
    else if (M <= 32) hipLaunchKernelGGL((skinny1x1_kernel<T, 2, NWV>), dim3((unsigned)(a.cout / 16), 1), block, 0, stream, a);
    else hipLaunchKernelGGL((skinny1x1_kernel<T, 4, NWV>), dim3((unsigned)(a.cout / 16), (unsigned)((M + 63) / 64)), block, 0, stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

template <typename T>
static int launch_skinny3x3(const PackedConv& pc, TapConvArgs& a, hipStream_t stream) {
    constexpr int NWV = sizeof(T) == 2 ? 8 : 16;
    a.tiles_x = (a.Wout + 15) / 16;
    a.tiles_y = (a.Hout + 3) / 4;
    a.ntiles = a.tiles_x * a.tiles_y * a.B;
    const dim3 grid((unsigned)(a.cout / 16), (unsigned)a.ntiles), block(64 * NWV);
    g_last_kernel = "ldm::skinny3x3_kernel";
    if (pc.kind == CONV3x3_S1) hipLaunchKernelGGL((skinny3x3_kernel<T, 1, NWV>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((skinny3x3_kernel<T, 2, NWV>), grid, block, 0, stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

template <typename T>
static int launch_typed(const PackedConv& pc, const ConvCall& cc, TapConvArgs& a, hipStream_t stream) {
    int bn = (a.cout % 128 == 0) ? 128 : 64;
    {
        // Small problems (a few images of a few pixels): every workgroup streams all weights of its channel
        // block through one CU, and with a handful of workgroups that stream is the whole run time.  Narrower
        // channel blocks spread it over twice as many CUs.
        const long t4 = (long)((a.Wg + 15) / 16) * ((a.Hg + 3) / 4) * a.B;
        if (bn == 128 && t4 * (a.cout / 128) * pc.nphase < 64) bn = 64;
    }
    if constexpr (sizeof(T) == 2) {
        if (ws64_shape(pc, a)) {
            if (a.sched_z) {
                if (a.act == ACT_NONE && !a.res && !a.tvec) return launch_ws64<WS_DDIM>(a, stream);
            } else if (!a.out_f32) {
                if (a.act == ACT_NONE && !a.res && !a.tvec) return launch_ws64<WS_PLAIN>(a, stream);
                if (a.act == ACT_SILU && !a.res && a.tvec) return launch_ws64<WS_SILU_TV>(a, stream);
                if (a.act == ACT_SILU && a.res && !a.tvec) return launch_ws64<WS_SILU_RES>(a, stream);
            }
        }
    }
    if (a.sched_z) LDM_FAIL(LDM_ERR_INVALID, "the fused DDIM update needs the 64-channel bf16 kernel (see tapconv_can_fuse_ddim)");
    if ((pc.kind == CONV3x3_S1 || pc.kind == CONV3x3_S2) && a.nch1 == 0) {
        // f32, a few images: the tiled kernel walks 18-72 (chunk, tap) steps per workgroup behind barriers.  Measured
        // on 50-step DDPM, 4 latents of 64x64, f32: 95 -> 124 images/s.  In bf16 (half the steps) the tiled kernel
        // is as fast at one image and faster from four on, so bf16 takes this kernel only when LDM_SKINNY3_MAXWG says so.
        static const long maxwg = [] {       // read once per process (and element type)
            const char* e = getenv("LDM_SKINNY3_MAXWG");
            return e ? atol(e) : (long)(sizeof(T) == 4 ? kSkinny3MaxWgF32 : 0);
        }();
        const long t64 = (long)((a.Wout + 15) / 16) * ((a.Hout + 3) / 4) * a.B;
        if (t64 <= 65535 && t64 * (a.cout / 16) <= maxwg) return launch_skinny3x3<T>(pc, a, stream);
    }
    // measured at 1024 -> 1024 channels: 145 vs 206 us per sampler step at 256 pixels, a tie at 1024, half the speed at 4096
    if (pc.kind == CONV1x1 && (long)a.B * a.Hin * a.Win <= kSkinny1x1MaxPix) return launch_skinny1x1<T>(a, stream);
    if constexpr (sizeof(T) == 2) {
        if (s2ws_shape(pc, a) && a.act == ACT_NONE && !a.res && !a.tvec && !a.out_f32) return launch_s2ws(a, stream);
        if (ws1x1_shape(pc, a) && a.act == ACT_NONE && !a.tvec && !a.out_f32)
            return a.res ? launch_ws1x1<true>(a, stream) : launch_ws1x1<false>(a, stream);
    }
    const int nblk_n = a.cout / bn;
    const long tx = (a.Wg + 15) / 16;
    const long t16 = tx * ((a.Hg + 15) / 16) * a.B, t8 = tx * ((a.Hg + 7) / 8) * a.B;
    const int nph = pc.nphase;
    int ncu_dev = 0;
    LDM_TRY(device_cu_count(&ncu_dev));
    const bool one_round = t16 * nblk_n * nph <= ncu_dev;      // 16-row tiles: at most one workgroup per CU
    if (pc.kind == CONV3x3_S2) {
        // one input chunk: a two-deep weight ring leaves room for two workgroups per CU, which hides the
        // halo prologue of one behind the MFMAs of the other (32 -> 24 us at 64 -> 128 channels)
        if (bn == 128 && a.nch0 == 1) return launch_cfg<T, 4, 128, 2, 2, 2, 3, 9, 2, 2>(a, nph, stream);
        if (bn == 128) return launch_cfg<T, 4, 128, 2, 2, 2, 3, 9, 1>(a, nph, stream);
        return launch_cfg<T, 4, 64, 2, 2, 2, 3, 9, 1>(a, nph, stream);
    }
    if (pc.kind == CONV1x1) {
        int th = pick_th(t16, t8, nblk_n, nph, 256, true, a.Hg);
        if (bn == 64 && th > 8) th = 8;       // memory-bound: two workgroups per CU beat the taller tile
        if (bn == 128) {
            if (th == 16) return launch_cfg<T, 16, 128, 4, 2, 1, 1, 1, 2>(a, nph, stream);
            if (th == 8) return launch_cfg<T, 8, 128, 2, 2, 1, 1, 1, 1>(a, nph, stream);
            return launch_cfg<T, 4, 128, 2, 2, 1, 1, 1, 1>(a, nph, stream);
        }
        if (th == 16) return launch_cfg<T, 16, 64, 4, 1, 1, 1, 1, 2>(a, nph, stream);
        if (th == 8) return launch_cfg<T, 8, 64, 2, 2, 1, 1, 1, 2, 2>(a, nph, stream);
        return launch_cfg<T, 4, 64, 2, 2, 1, 1, 1, 2>(a, nph, stream);
    }
    // 3x3-window kinds: conv3x3 stride 1 (9 taps) and the convT phases (4 taps)
    int th = pick_th(t16, t8, nblk_n, nph, bn == 64 ? 512 : 256, true, a.Hg);
    if (pc.kind == CONVT4x4_S2 && bn == 64 && th > 8) th = 8;   // two workgroups per CU (LDS) beat the taller tile
    if (pc.kind == CONV3x3_S1) {
        if (bn == 128) {
            // too few 16-row tiles for 128-channel blocks: 16 rows x 64 channels moves a third fewer bytes
            // per CU through L2 -> LDS than 8 rows x 128 channels (weights halve, the window grows less)
            if (th == 8 && a.Hg > 8 && t16 * (a.cout / 64) >= 256) return launch_cfg<T, 16, 64, 4, 2, 1, 3, 9, 2, 4>(a, nph, stream);
            // a launch of a single round of workgroups runs the tile with sixteen waves (four per SIMD cover each other's
            // barrier and DMA stalls: 24.1 -> 23.1 us at 256 workgroups); over many rounds eight waves are 1 % faster
            if (th == 16 && sizeof(T) == 2 && one_round)
                return launch_cfg<T, 16, 128, 8, 2, 1, 3, 9, 4, 3>(a, nph, stream);
            if (th == 16) return launch_cfg<T, 16, 128, 4, 2, 1, 3, 9, 2, 3>(a, nph, stream);
            if (th == 8) return launch_cfg<T, 8, 128, 4, 2, 1, 3, 9, 2, 4>(a, nph, stream);
            return launch_cfg<T, 4, 128, 2, 2, 1, 3, 9, 1>(a, nph, stream);
        }
        if (th == 16) return launch_cfg<T, 16, 64, 4, 1, 1, 3, 9, 2, 3>(a, nph, stream);
        if (th == 8) return launch_cfg<T, 8, 64, 2, 2, 1, 3, 9, 2>(a, nph, stream);
        return launch_cfg<T, 4, 64, 2, 2, 1, 3, 9, 2>(a, nph, stream);
    }
    if (pc.ntaps1 != pc.ntaps) {
        // fused transposed conv + concat + 1x1: the second source has one tap and output resolution
        if constexpr (sizeof(T) == 2) {
            if (pc.ntaps1 != 1 || !cc.in1) LDM_FAIL(LDM_ERR_INVALID, "fused up-sampling conv needs its skip source");
            if (bn == 128) {
                if (th == 16 && one_round) return launch_cfg<T, 16, 128, 8, 2, 1, 3, 4, 4, 3, 1>(a, nph, stream);   // 28.9 -> 26.3 us
                if (th == 16) return launch_cfg<T, 16, 128, 4, 2, 1, 3, 4, 2, 3, 1>(a, nph, stream);
                if (th == 8) return launch_cfg<T, 8, 128, 2, 2, 1, 3, 4, 1, 3, 1>(a, nph, stream);
                return launch_cfg<T, 4, 128, 2, 2, 1, 3, 4, 1, 3, 1>(a, nph, stream);
            }
            if (th == 8 || th == 16) return launch_cfg<T, 8, 64, 2, 2, 1, 3, 4, 2, 3, 1>(a, nph, stream);
            return launch_cfg<T, 4, 64, 2, 2, 1, 3, 4, 2, 3, 1>(a, nph, stream);
        } else {
            LDM_FAIL(LDM_ERR_INVALID, "fused up-sampling conv is bf16 only");
        }
    }
    if (bn == 128) {
        if (th == 16) return launch_cfg<T, 16, 128, 4, 2, 1, 3, 4, 2>(a, nph, stream);
        if (th == 8) return launch_cfg<T, 8, 128, 2, 2, 1, 3, 4, 1>(a, nph, stream);
        return launch_cfg<T, 4, 128, 2, 2, 1, 3, 4, 1>(a, nph, stream);
    }
    if (th == 16) return launch_cfg<T, 16, 64, 4, 1, 1, 3, 4, 2>(a, nph, stream);
    if (th == 8) return launch_cfg<T, 8, 64, 2, 2, 1, 3, 4, 2>(a, nph, stream);
    return launch_cfg<T, 4, 64, 2, 2, 1, 3, 4, 2>(a, nph, stream);
}

bool upfuse_shape_ok(int dtype, int cin, int cskip, int cout, int B, int Hin, int Win) {
    return dtype == LDM_BF16 && cin == 128 && cskip == 64 && cout == 64 && Hin % 4 == 0 && Win % 16 == 0 &&
           (long)B * (Hin / 4) * (Win / 16) >= 256 && (size_t)Hin * Win * 256 * 4 < (1ull << 31);
}

// ResidualBlock as one kernel: the shapes resfuse64_kernel serves (64 channels, 64-pixel-wide images whose height
// is a multiple of the band, enough bands to fill the chip)
bool resfuse_shape_ok(int dtype, int C, int B, int H, int W) {
    return dtype == LDM_BF16 && C == 64 && W == 64 && H % RF_R == 0 && (long)B * (H / RF_R) >= 128 && B > 0 &&
           (long)B * H * W * 64 * 2 < (1l << 40);
}

int launch_resfuse(const PackedConv& c1, const PackedConv& c2, const void* x, void* out, int B, int H, int W,
                   const float* tvec, int t_img_stride, int t_step_stride, const int* step_ptr, hipStream_t stream) {
    LDM_REQUIRE(x && out && tvec && c1.w && c1.bias && c2.w && c2.bias && x != out, "null or aliased tensor");
    LDM_REQUIRE(c1.kind == CONV3x3_S1 && c2.kind == CONV3x3_S1 && c1.cin == 64 && c1.cout == 64 && c2.cin == 64 && c2.cout == 64 &&
                    c1.dtype == c2.dtype && resfuse_shape_ok(c1.dtype, 64, B, H, W),
                "shape not served by the fused ResidualBlock kernel");
    TapConvArgs a{};
    a.in0 = (const char*)x; a.out = (char*)out;
    a.w = (const char*)c1.w; a.bias = c1.bias;
    a.w2 = (const char*)c2.w; a.bias2 = c2.bias;
    a.tvec = tvec; a.t_img_stride = t_img_stride; a.t_step_stride = t_step_stride; a.step_ptr = step_ptr;
    a.B = B; a.Hin = H; a.Win = W; a.Hout = H; a.Wout = W; a.cout = 64;
#ifdef TAPCONV_STAMP
    a.dbg_rt = dbg_rt_for("resfuse");
#endif
    static bool lds_set[kMaxDevices] = {};
    LDM_TRY(allow_dynamic_lds((const void*)resfuse64_kernel<RF_RES>, 160 * 1024, lds_set));
    g_last_kernel = "void ldm::resfuse64_kernel<0>(ldm::TapConvArgs)";
    hipLaunchKernelGGL(resfuse64_kernel<RF_RES>, dim3((unsigned)(B * (H / RF_R))), dim3(512), rf_lds(RF_RES), stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// The end of one bf16 DDIM step and the start of the next as one kernel (resfuse64_kernel<RF_DDIM>): out_conv on h, the
// DDIM update z_out = update(z_in, predicted noise) in f32, and down1 of the next step on the bf16 rounding of z_out,
// which never goes to memory.  z_in and z_out must be different buffers.  The shapes are those of the fused
// ResidualBlock; ctl = {step counter, ..., ticket at ctl[4]} as for the fused DDIM convolution.
bool ddim_down1_shape_ok(const PackedConv& out_conv, const PackedConv& down1, int B, int H, int W) {
    return out_conv.kind == CONV3x3_S1 && down1.kind == CONV3x3_S1 && out_conv.dtype == LDM_BF16 && down1.dtype == LDM_BF16 &&
           out_conv.cin == 64 && out_conv.cout == 64 && down1.cin == 64 && down1.cout == 64 && resfuse_shape_ok(LDM_BF16, 64, B, H, W);
}
// The kernel's launch attribute (160 KiB of dynamic LDS), set on the current device outside any stream capture.
static bool g_ddim_down1_lds_set[kMaxDevices] = {};
int prepare_ddim_down1() { return allow_dynamic_lds((const void*)resfuse64_kernel<RF_DDIM>, 160 * 1024, g_ddim_down1_lds_set); }
int launch_ddim_down1(const PackedConv& out_conv, const PackedConv& down1, const void* h, const float* z_in, float* z_out,
                      void* a1_out, int B, int H, int W, const float* coef, int* ctl, hipStream_t stream) {
    LDM_REQUIRE(h && z_in && z_out && a1_out && coef && ctl && z_in != z_out && h != a1_out, "null or aliased tensor");
    LDM_REQUIRE(out_conv.w && out_conv.bias && down1.w && down1.bias && ddim_down1_shape_ok(out_conv, down1, B, H, W),
                "shape not served by the fused DDIM / down1 kernel");
    TapConvArgs a{};
    a.in0 = (const char*)h; a.out = (char*)a1_out;
    a.w = (const char*)out_conv.w; a.bias = out_conv.bias;
    a.w2 = (const char*)down1.w; a.bias2 = down1.bias;
    a.sched_z = const_cast<float*>(z_in); a.sched_z_out = z_out; a.sched_coef = coef;
    a.step_ptr = ctl; a.sched_ticket = ctl + 4;
    a.B = B; a.Hin = H; a.Win = W; a.Hout = H; a.Wout = W; a.cout = 64;
#ifdef TAPCONV_STAMP
    a.dbg_rt = dbg_rt_for("chain");
#endif
    LDM_TRY(prepare_ddim_down1());
    g_last_kernel = "void ldm::resfuse64_kernel<1>(ldm::TapConvArgs)";
    hipLaunchKernelGGL(resfuse64_kernel<RF_DDIM>, dim3((unsigned)(B * (H / RF_R))), dim3(512), rf_lds(RF_DDIM), stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

// pc: packed by pack_conv as CONVT4x4_S2 over the sources {128 composed input channels, 64 skip channels},
// the skip chunk with a single tap
int launch_upfuse(const PackedConv& pc, const void* x, const void* skip, void* out, int B, int Hin, int Win,
                  hipStream_t stream) {
    LDM_REQUIRE(x && skip && out && pc.w && pc.bias, "null tensor");
    LDM_REQUIRE(pc.kind == CONVT4x4_S2 && pc.cin == 192 && pc.ntaps1 == 1 && upfuse_shape_ok(pc.dtype, 128, 64, pc.cout, B, Hin, Win),
                "shape not served by the fused up-sampling kernel");
    int ncu = 0;
    LDM_TRY(device_cu_count(&ncu));
    TapConvArgs a{};
    a.in0 = (const char*)x; a.in1 = (const char*)skip; a.w = (const char*)pc.w; a.bias = pc.bias; a.out = (char*)out;
    a.B = B; a.Hin = Hin; a.Win = Win; a.Hout = 2 * Hin; a.Wout = 2 * Win; a.cout = 64;
    a.tiles_x = Win / 16; a.tiles_y = Hin / 4; a.ntiles = a.tiles_x * a.tiles_y * B;
    a.w_phase_stride = (long)9 * 64 * 128;           // 2 chunks x 4 taps + 1 skip slab of 8 KiB per phase
    const size_t lds = 2 * 64 * 1024;
    static bool lds_set[kMaxDevices] = {};   // per kernel (instantiation) and device
    LDM_TRY(allow_dynamic_lds((const void*)upfuse_ws_kernel, 160 * 1024, lds_set));
    const int grid = a.ntiles < ncu ? a.ntiles : ncu;
    g_last_kernel = "ldm::upfuse_ws_kernel(ldm::TapConvArgs)";
    hipLaunchKernelGGL(upfuse_ws_kernel, dim3(grid), dim3(512), lds, stream, a);
    LDM_HIP(hipGetLastError());
    return LDM_OK;
}

int launch_tapconv(const PackedConv& pc, const ConvCall& cc, hipStream_t stream) {
    const int es = pc.dtype == LDM_BF16 ? 2 : 4;
    const int ck = 128 / es;
    LDM_REQUIRE(cc.in0 && cc.out && pc.w && pc.bias, "null tensor");
    LDM_REQUIRE(cc.cin0 % ck == 0 && cc.cin1 % ck == 0 && cc.cin0 + cc.cin1 == pc.cin, "input channels do not match the packed conv");
    LDM_REQUIRE(pc.cout % 64 == 0, "output channels must be padded to a multiple of 64");
    LDM_REQUIRE(cc.B > 0 && cc.Hin > 0 && cc.Win > 0, "empty input");
    LDM_REQUIRE((cc.cin1 == 0) == (cc.in1 == nullptr), "second source and its channel count disagree");
    LDM_REQUIRE((size_t)cc.Hin * cc.Win * (size_t)std::max(std::max(cc.cin0, cc.in0_stride), cc.cin1) * es < (1ull << 31), "image too large");
    if (pc.ntaps1 != pc.ntaps)   // the second source has the output's resolution
        LDM_REQUIRE((size_t)4 * cc.Hin * cc.Win * (size_t)cc.cin1 * es < (1ull << 31), "image too large");
    TapConvArgs a{};
    a.in0 = (const char*)cc.in0; a.in1 = (const char*)cc.in1;
    a.w = (const char*)pc.w; a.bias = pc.bias;
    a.res = (const char*)cc.res; a.tvec = cc.tvec; a.step_ptr = cc.step_ptr;
    a.out = (char*)cc.out;
    a.nch0 = cc.cin0 / ck; a.nch1 = cc.cin1 / ck;
    LDM_REQUIRE(cc.in0_stride == 0 || (cc.in0_stride >= cc.cin0 && cc.in0_stride % (16 / es) == 0), "bad pixel stride of the first source");
    a.cin0_bytes = (cc.in0_stride ? cc.in0_stride : cc.cin0) * es; a.cin1_bytes = cc.cin1 * es;
    a.B = cc.B; a.Hin = cc.Hin; a.Win = cc.Win;
    LDM_TRY(conv_out_hw(pc.kind, cc.Hin, cc.Win, &a.Hout, &a.Wout));
    a.osc = pc.kind == CONVT4x4_S2 ? 2 : 1;
    a.Hg = a.Hout / a.osc; a.Wg = a.Wout / a.osc;
    a.cout = pc.cout;
    a.ntaps = pc.ntaps;
    a.act = cc.act; a.out_f32 = (pc.dtype == LDM_F32) ? 1 : cc.out_f32;
    a.t_img_stride = cc.t_img_stride; a.t_step_stride = cc.t_step_stride; a.t_pix_stride = cc.t_pix_stride;
    LDM_REQUIRE(cc.t_pix_stride == 0 || (pc.kind == CONV1x1 && cc.tvec), "a time vector per pixel is a 1x1-convolution feature");
    a.sched_z = cc.ddim_z; a.sched_coef = cc.ddim_coef; a.sched_ticket = cc.ddim_ticket;
    LDM_REQUIRE(!cc.ddim_z || (cc.ddim_coef && cc.step_ptr), "fused DDIM update needs the coefficient table and the step counter");
    a.w_phase_stride = ((long)a.nch0 * pc.ntaps + (long)a.nch1 * pc.ntaps1) * pc.cout * 128;
    if (pc.dtype == LDM_BF16) return launch_typed<bf16_t>(pc, cc, a, stream);
    return launch_typed<float>(pc, cc, a, stream);
}

}  // namespace ldm
