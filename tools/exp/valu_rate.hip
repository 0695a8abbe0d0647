// Dev microbenchmark: issue rate of fp32 VALU flavours on gfx950 (scalar fma, packed fma, v_exp_f32, v_rcp_f32, MFMA,
// and MFMA + VALU interleaved in one wave).  hipcc --offload-arch=gfx950 -O3 tools/exp/valu_rate.hip -o tools/exp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float a[8];
    v2f p[8];
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = v2f{a[i], a[i] + 0.5f}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 0.999f, 0.001f);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], (v2f)(0.999f), (v2f)(0.001f));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]);
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_rcpf(a[i]);
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc[i & 3], 0, 0, 0);
        } else if (MODE == 5) {  // 8 MFMA + 64 scalar fma per trip, interleaved by the scheduler pragma
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, acc[i & 3], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 8; ++r) a[r] = fmaf(a[r], 0.999f, 0.001f);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
        } else if (MODE == 6) {  // same work, all MFMAs first then all VALU
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, seed, acc[i & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 8; ++r) a[r] = fmaf(a[r], 0.999f, 0.001f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, int wgs_per_cu, double ops_per_iter_lane) {
    float* out;
    const int blocks = 256 * wgs_per_cu;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal: waves per SIMD = wgs_per_cu (4 waves/WG over 4 SIMDs)
    const double instr_per_simd = (double)ITERS * ops_per_iter_lane * wgs_per_cu;
    printf("%-28s wg/cu=%d  %8.1f us   %.2f nominal-cycles per wave-instr\n", name, wgs_per_cu, ms * 1e3,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0>("v_fma_f32", 1, 8); run<1>("v_pk_fma_f32", 1, 8); run<2>("v_exp_f32", 1, 8); run<3>("v_rcp_f32", 1, 8);
                      run<4>("mfma16x16x4f32", 1, 8); run<5>("8 mfma + 64 fma interleaved", 1, 72); run<6>("8 mfma then 64 fma", 1, 72); }
        if (w == 2) { run<0>("v_fma_f32", 2, 8); run<1>("v_pk_fma_f32", 2, 8); run<2>("v_exp_f32", 2, 8); run<3>("v_rcp_f32", 2, 8);
                      run<4>("mfma16x16x4f32", 2, 8); run<5>("8 mfma + 64 fma interleaved", 2, 72); run<6>("8 mfma then 64 fma", 2, 72); }
        if (w == 4) { run<0>("v_fma_f32", 4, 8); run<1>("v_pk_fma_f32", 4, 8); run<2>("v_exp_f32", 4, 8); run<3>("v_rcp_f32", 4, 8);
                      run<4>("mfma16x16x4f32", 4, 8); run<5>("8 mfma + 64 fma interleaved", 4, 72); run<6>("8 mfma then 64 fma", 4, 72); }
    }
    return 0;
}
