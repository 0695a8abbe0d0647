// Experiment (round 3): does the PRIVATE SEGMENT (scratch) of a wave survive a foreign process running on the same GPU?
// Background: k_head_fwd -- the only FNO-path kernel that returned wrong results beside another process -- was also the only one
// whose hot loop re-read register spills (scratch_load ... Folded Reload) that it had parked in its prologue.
//   victim:    every lane parks a signature in scratch (the spill code's own instructions), works for a while (global loads +
//              MFMAs, a few us), re-reads the slot and counts mismatches
//   aggressor: a 256-VGPR / 48-KB-LDS MFMA kernel without scratch
// usage: scratch_cotenancy victim|aggr|novictim <launches> [blocks]     (start two processes at once)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/scratch_cotenancy.hip -o tools/exp/scratch_cotenancy_exp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(2);                                                                \
        }                                                                           \
    } while (0)

template <bool SCRATCH>
__global__ __launch_bounds__(256, 3) void k_victim(const float* __restrict__ src, float* __restrict__ dst, unsigned* __restrict__ bad,
                                                   int n, int spin) {
    const unsigned sig = 0x9e3779b9u * (blockIdx.x * 256u + threadIdx.x + 1u);
    unsigned kept = sig;
    if constexpr (SCRATCH) {
        volatile int reserve[2];  // allocates the private segment; slot 0 is accessed by the asm below
        reserve[1] = 0;
        asm volatile("scratch_store_dword off, %0, off" ::"v"(sig) : "memory");
    } else {
        asm volatile("" : "+v"(kept));
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.001f * (threadIdx.x + k)); b[k] = (__bf16)(0.002f * (k + 1)); }
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (int it = 0; it < spin; ++it) {
        s += src[idx % n];
        idx = idx * 1664525u + 1013904223u;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
    unsigned back;
    if constexpr (SCRATCH) {
        asm volatile("scratch_load_dword %0, off, off\n\ts_waitcnt vmcnt(0)" : "=v"(back)::"memory");
    } else {
        back = kept;
    }
    if (back != sig) atomicAdd(bad, 1u);
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = s + acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256, 2) void k_aggr(const float* __restrict__ src, float* __restrict__ dst, int n, int spin) {
    __shared__ float lds[12288];  // 48 KB
    f32x4 acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = src[(blockIdx.x * 12288 + i) % n];
    __syncthreads();
    bf16x8 a, b;
    for (int it = 0; it < spin; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] = (__bf16)lds[(threadIdx.x * 8 + k + it) % 12288];
            b[k] = (__bf16)lds[(threadIdx.x * 8 + k + 3 * it + 7) % 12288];
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "victim";
    const int launches = argc > 2 ? atoi(argv[2]) : 1000;
    const int blocks = argc > 3 ? atoi(argv[3]) : 1024;
    const int n = 1 << 24;
    float *src, *dst;
    unsigned* bad;
    CK(hipMalloc(&src, n * sizeof(float)));
    CK(hipMalloc(&dst, (size_t)blocks * 256 * sizeof(float)));
    CK(hipMalloc(&bad, sizeof(unsigned)));
    CK(hipMemset(src, 0, n * sizeof(float)));
    CK(hipMemset(bad, 0, sizeof(unsigned)));
    unsigned long total_bad = 0;
    int bad_launches = 0;
    for (int l = 0; l < launches; ++l) {
        if (!strcmp(mode, "victim")) hipLaunchKernelGGL(k_victim<true>, dim3(blocks), dim3(256), 0, 0, src, dst, bad, n, 64);
        else if (!strcmp(mode, "novictim")) hipLaunchKernelGGL(k_victim<false>, dim3(blocks), dim3(256), 0, 0, src, dst, bad, n, 64);
        else hipLaunchKernelGGL(k_aggr, dim3(512), dim3(256), 0, 0, src, dst, n, 200);
        CK(hipDeviceSynchronize());
        unsigned h = 0;
        CK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
        if (h) {
            total_bad += h;
            ++bad_launches;
            CK(hipMemset(bad, 0, sizeof(unsigned)));
        }
    }
    printf("%s: %d launches, %d with scratch mismatches (%lu lanes)\n", mode, launches, bad_launches, total_bad);
    return 0;
}
