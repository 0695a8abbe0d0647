// Dev experiment: f32 MFMA 16x16x4 issue rate / clock under load, with 1..3 waves per SIMD and VALU/LDS filler.
#include <cstdio>
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int FILL>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, long long* cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
    float f0 = a, f1 = b;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < FILL; ++k) { f0 = fmaf(f0, 1.0001f, f1); f1 = fmaf(f1, 0.9999f, f0); }
        }
    }
    long long t1 = clock64();
    float s = f0 + f1;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 4096 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
#define RUN(NACC, FILL, BPC)                                                                                   \
    {                                                                                                          \
        hipLaunchKernelGGL((k_mfma<NACC, FILL>), dim3(256 * BPC), dim3(256), 0, 0, out, 10, cyc);              \
        hipDeviceSynchronize();                                                                                \
        hipEventRecord(e0, 0);                                                                                 \
        hipLaunchKernelGGL((k_mfma<NACC, FILL>), dim3(256 * BPC), dim3(256), 0, 0, out, iters, cyc);           \
        hipEventRecord(e1, 0); hipEventSynchronize(e1);                                                        \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                            \
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                                             \
        double nm = (double)iters * NACC;                                                                      \
        double tf = nm * 2048.0 * 4 * 256 * BPC / (ms * 1e-3) / 1e12;                                          \
        printf("nacc %d fill %d wg/cu %d: %8.1f us  %6.1f TF  clock64/mfma %.1f  us/mfma-per-simd %.4f -> eff MHz(32cyc) %.0f\n", \
               NACC, FILL, BPC, ms * 1e3, tf, (double)c / nm, ms * 1e3 / (nm * BPC), 32.0 * nm * BPC / (ms * 1e3)); \
    }
    RUN(8, 0, 1) RUN(8, 0, 2) RUN(8, 0, 3) RUN(4, 0, 1) RUN(2, 0, 1) RUN(1, 0, 1)
    RUN(8, 2, 1) RUN(8, 4, 1) RUN(8, 8, 1) RUN(8, 4, 2) RUN(8, 8, 2) RUN(8, 8, 3)
    return 0;
}
