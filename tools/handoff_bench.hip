// What does a layer boundary cost inside ONE launch, against a kernel boundary, for the dependency shape of the
// 256-channel level of the UNet?  There a workgroup owns (image, one of four 64-channel blocks) of a layer's output
// (32 KiB bf16 at 16x16 pixels) and the next layer's workgroup (image, block) reads all four blocks of the image
// (128 KiB).  This program moves exactly those bytes for R rounds, 256 workgroups of 256 threads (one per CU):
//   mode "launch":     one kernel per round (R graph nodes); the kernel boundary orders the rounds
//   mode "persistent": one kernel; per image an arrival counter: write-through 16-byte stores, vmcnt(0), barrier, one
//                      agent-scope add; the consumers poll with a bounded spin, run the agent acquire and read
// Every word read is checked against its closed form (a stale word counts as an error).  The spin is bounded: a
// workgroup that gives up counts a timeout and goes on, so the grid always drains.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_bench tools/handoff_bench.hip && ./handoff_bench [rounds] [filler_kb]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NWG = 256, NT = 256, BLK_BYTES = 32 * 1024, VEC_PER_T = BLK_BYTES / 16 / NT;   // 8 vectors per thread and block

__device__ __forceinline__ unsigned word_of(int round, int image, int block, int vec) {
    return (unsigned)(round * 2654435761u) ^ (unsigned)(image * 40503u + block * 977u) ^ (unsigned)(vec * 2246822519u);
}
// placement: same_xcd puts the four blocks of an image on workgroup ids that are equal modulo 8
__device__ __forceinline__ void whoami(int same_xcd, int* image, int* block) {
    const int b = blockIdx.x;
    if (same_xcd) { const int slot = b >> 3; *image = (b & 7) * 8 + (slot >> 2); *block = slot & 3; }
    else { *image = b >> 2; *block = b & 3; }
}
__device__ __forceinline__ void write_block(char* buf, int round, int image, int block) {
    char* dst = buf + ((size_t)image * 4 + block) * BLK_BYTES;
#pragma unroll
    for (int i = 0; i < VEC_PER_T; ++i) {
        const int vec = i * NT + threadIdx.x;
        const unsigned w = word_of(round, image, block, vec);
        const u32x4 v = {w, w + 1, w + 2, w + 3};
        // (s_nop: the data registers of a 16-byte store must not be overwritten in the next cycle; the compiler does not look inside asm)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + (size_t)vec * 16), "v"(v) : "memory");
    }
}
__device__ __forceinline__ unsigned read_image(const char* buf, int round, int image) {
    unsigned bad = 0;
    const char* src = buf + (size_t)image * 4 * BLK_BYTES;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        u32x4 v[VEC_PER_T];
#pragma unroll
        for (int i = 0; i < VEC_PER_T; ++i) v[i] = *(const u32x4*)(src + (size_t)blk * BLK_BYTES + (size_t)(i * NT + threadIdx.x) * 16);
#pragma unroll
        for (int i = 0; i < VEC_PER_T; ++i) {
            const unsigned w = word_of(round, image, blk, i * NT + threadIdx.x);
            bad += (v[i][0] != w) + (v[i][1] != w + 1) + (v[i][2] != w + 2) + (v[i][3] != w + 3);
        }
    }
    return bad;
}
// some streaming next to the hand-off: filler_vec 16-byte loads per thread from a buffer that sits in L2 / MALL
__device__ __forceinline__ unsigned filler(const char* fill, int filler_vec, int round) {
    unsigned acc = 0;
    const char* p = fill + ((size_t)(blockIdx.x * 37 + round * 11) % 64) * (1 << 20);
    for (int i = 0; i < filler_vec; i += 8) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const u32x4*)(p + (size_t)((i + j) * NT + threadIdx.x) * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][3];
    }
    return acc;
}

__global__ __launch_bounds__(NT) void round_kernel(char* buf0, char* buf1, int round, int same_xcd, const char* fill, int filler_vec,
                                                   unsigned* errs) {
    int image, block;
    whoami(same_xcd, &image, &block);
    unsigned bad = 0;
    if (round > 0) bad = read_image((round & 1) ? buf0 : buf1, round - 1, image);
    const unsigned f = filler(fill, filler_vec, round);
    write_block((round & 1) ? buf1 : buf0, round, image, block);
    if (bad) atomicAdd(&errs[0], bad);
    if (f == 0x12345u) errs[2] = f;
}

__global__ __launch_bounds__(NT) void persistent_kernel(char* buf0, char* buf1, int rounds, int same_xcd, const char* fill,
                                                        int filler_vec, unsigned* counters, unsigned* errs, int spin_limit) {
    int image, block;
    whoami(same_xcd, &image, &block);
    unsigned bad = 0, timeouts = 0, f = 0;
    for (int r = 0; r < rounds; ++r) {
        if (r > 0) {
            if (threadIdx.x == 0) {
                const unsigned want = 4u * (unsigned)r;
                int n = 0;
                while (__hip_atomic_load(&counters[image * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (++n > spin_limit) { ++timeouts; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            bad += read_image((r & 1) ? buf0 : buf1, r - 1, image);
        }
        f += filler(fill, filler_vec, r);
        write_block((r & 1) ? buf1 : buf0, r, image, block);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&counters[image * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (bad) atomicAdd(&errs[0], bad);
    if (timeouts) atomicAdd(&errs[1], timeouts);
    if (f == 0x12345u) errs[2] = f;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 24;
    const int filler_kb = argc > 2 ? atoi(argv[2]) : 0;            // per workgroup and round
    const int filler_vec = filler_kb * 1024 / 16 / NT / 8 * 8;
    char *buf0, *buf1, *fill;
    unsigned *counters, *errs;
    const size_t bytes = (size_t)64 * 4 * BLK_BYTES;
    CK(hipMalloc(&buf0, bytes)); CK(hipMalloc(&buf1, bytes)); CK(hipMalloc(&fill, (size_t)65 << 20));
    CK(hipMalloc(&counters, 64 * 32 * 4)); CK(hipMalloc(&errs, 16));
    CK(hipMemset(fill, 1, (size_t)65 << 20));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int same_xcd = 0; same_xcd < 2; ++same_xcd)
        for (int mode = 0; mode < 2; ++mode) {
            hipGraph_t graph;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            if (mode == 0) {
                for (int r = 0; r < rounds; ++r)
                    hipLaunchKernelGGL(round_kernel, dim3(NWG), dim3(NT), 0, st, buf0, buf1, r, same_xcd, fill, filler_vec, errs);
            } else {
                hipLaunchKernelGGL(persistent_kernel, dim3(NWG), dim3(NT), 0, st, buf0, buf1, rounds, same_xcd, fill, filler_vec,
                                   counters, errs, 1 << 16);
            }
            CK(hipStreamEndCapture(st, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            std::vector<float> ms;
            for (int it = 0; it < 12; ++it) {
                CK(hipMemsetAsync(counters, 0, 64 * 32 * 4, st));
                CK(hipMemsetAsync(errs, 0, 16, st));
                CK(hipEventRecord(e0, st));
                CK(hipGraphLaunch(exec, st));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (it >= 2) ms.push_back(t);
            }
            unsigned h[4];
            CK(hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost));
            std::sort(ms.begin(), ms.end());
            printf("%-10s blocks of an image on %s  filler %4d KiB: %7.2f us per round (median of %zu, %d rounds)  wrong words %u  timeouts %u\n",
                   mode ? "persistent" : "launch", same_xcd ? "one XCD " : "four XCDs", filler_kb, ms[ms.size() / 2] * 1e3 / rounds,
                   ms.size(), rounds, h[0], h[1]);
            fflush(stdout);
            CK(hipGraphExecDestroy(exec));
            CK(hipGraphDestroy(graph));
        }
    return 0;
}
