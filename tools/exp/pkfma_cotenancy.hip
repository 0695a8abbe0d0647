// Experiment (round 3): which forms of v_pk_fma_f32 lose their LOW-half result beside a foreign process' kernels?
// Finding that led here (profiles/r03_det_root_cause.md): k_head_fwd's fc2 accumulation  o01 += {w2[0][j], w2[1][j]} * g_j  was written
// with the operand-select form of the packed FMA (one register of a pair broadcast to both halves: op_sel:[0,1,0] /
// op_sel_hi:[1,0,1]).  Beside another PROCESS running k_head_bwd, the low half of the op_sel:[0,1,0] instruction did not
// accumulate in lanes 48-63 of a wave (the same chain with v_fma_f32 in inline asm on the same registers was right).
// This kernel runs each form next to its scalar restatement and counts bitwise mismatches per 16-lane row.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/pkfma_cotenancy.hip -o tools/exp/pkfma_cotenancy_exp
//   pkfma_cotenancy_exp <launches> [blocks] [iters]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(2);                                                \
        }                                                           \
    } while (0)

#define NFORM 12
// form f: packed instruction on (acc, a, b) and its scalar restatement (lo, hi)
//  0  plain                         lo = a.lo*b.lo + acc.lo   hi = a.hi*b.hi + acc.hi
//  1  op_sel_hi:[1,0,1]             lo = a.lo*b.lo            hi = a.hi*b.lo      (broadcast b.lo)
//  2  op_sel:[0,1,0]                lo = a.lo*b.hi            hi = a.hi*b.hi      (broadcast b.hi)   <- the k_head_fwd form that failed
//  3  op_sel_hi:[0,1,1]             lo = a.lo*b.lo            hi = a.lo*b.hi      (broadcast a.lo; cfd_cmla step 1)
//  4  op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]   lo = -a.hi*b.hi   hi = a.hi*b.lo   (cfd_cmla step 2)
//  5  op_sel:[1,0,0]                lo = a.hi*b.lo            hi = a.hi*b.hi      (broadcast a.hi)
//  6  op_sel:[0,1,0] op_sel_hi:[0,1,1]   lo = a.lo*b.hi       hi = a.lo*b.hi
//  7  op_sel:[0,0,1]                lo = a.lo*b.lo + acc.hi   (third operand's high half into the low result; acc chain kept per half)
//  8  v_pk_mul_f32 op_sel:[0,1]     lo = a.lo*b.hi            hi = a.hi*b.hi
//  9  v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]   lo = a.lo+b.hi   hi = a.hi+b.lo
// 10  v_pk_mul_f32 op_sel:[1,0]     lo = a.hi*b.lo            hi = a.hi*b.hi
// 11  op_sel:[0,1,0] with the broadcast operand FIRST in the instruction (b, a swapped: op_sel:[1,0,0] on (b, a))  = form 5 on swapped operands
template <int F>
__device__ __forceinline__ void step(f2& accp, float& lo, float& hi, f2 a, f2 b) {
    if constexpr (F == 0) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(accp) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.y));
    } else if constexpr (F == 1) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accp) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.x));
    } else if constexpr (F == 2) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(accp) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.y));
    } else if constexpr (F == 3) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(accp) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.x), "v"(b.y));
    } else if constexpr (F == 4) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(accp) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(lo) : "v"(a.y), "v"(b.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.x));
    } else if constexpr (F == 5) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(accp) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.y), "v"(b.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.y), "v"(b.y));
    } else if constexpr (F == 6) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "+v"(accp) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(a.x), "v"(b.y));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(a.x), "v"(b.y));
    } else if constexpr (F == 7) {
        f2 t;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(t) : "v"(a), "v"(b), "v"(accp));
        float tl, th;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(tl) : "v"(a.x), "v"(b.x), "v"(hi));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(th) : "v"(a.y), "v"(b.y), "v"(hi));
        accp = t; lo = tl; hi = th;
    } else if constexpr (F == 8) {
        f2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(t) : "v"(a), "v"(b));
        accp += t;
        float tl, th;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tl) : "v"(a.x), "v"(b.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(th) : "v"(a.y), "v"(b.y));
        lo += tl; hi += th;
    } else if constexpr (F == 9) {
        f2 t;
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(b));
        accp += t;
        float tl, th;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(tl) : "v"(a.x), "v"(b.y));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(th) : "v"(a.y), "v"(b.x));
        lo += tl; hi += th;
    } else if constexpr (F == 10) {
        f2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(t) : "v"(a), "v"(b));
        accp += t;
        float tl, th;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(tl) : "v"(a.y), "v"(b.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(th) : "v"(a.y), "v"(b.y));
        lo += tl; hi += th;
    } else {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(accp) : "v"(b), "v"(a));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(lo) : "v"(b.y), "v"(a.x));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hi) : "v"(b.y), "v"(a.y));
    }
}

template <int F>
__device__ __forceinline__ void run_form(const float* __restrict__ src, int n, int iters, unsigned* __restrict__ bad) {
    const int lane = threadIdx.x & 63;
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned nlo = 0, nhi = 0;
    for (int it = 0; it < iters; ++it) {
        // operands come from memory each round (as the weights / activations of the real kernel do)
        f2 a[8], b[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + ((idx + 64 * k) * 4) % n);
            a[k] = f2{v.x, v.y};
            b[k] = f2{v.z, v.w};
        }
        idx = idx * 1664525u + 1013904223u;
        f2 accp = {0.f, 0.f};
        float lo = 0.f, hi = 0.f;
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) step<F>(accp, lo, hi, a[k], b[k]);
        nlo += __float_as_uint(accp.x) != __float_as_uint(lo);
        nhi += __float_as_uint(accp.y) != __float_as_uint(hi);
    }
    if (nlo) atomicAdd(&bad[(F * 2 + 0) * 4 + (lane >> 4)], nlo);
    if (nhi) atomicAdd(&bad[(F * 2 + 1) * 4 + (lane >> 4)], nhi);
}

__global__ __launch_bounds__(256, 3) void k_forms(const float* __restrict__ src, int n, int iters, unsigned* __restrict__ bad) {
    run_form<0>(src, n, iters, bad);
    run_form<1>(src, n, iters, bad);
    run_form<2>(src, n, iters, bad);
    run_form<3>(src, n, iters, bad);
    run_form<4>(src, n, iters, bad);
    run_form<5>(src, n, iters, bad);
    run_form<6>(src, n, iters, bad);
    run_form<7>(src, n, iters, bad);
    run_form<8>(src, n, iters, bad);
    run_form<9>(src, n, iters, bad);
    run_form<10>(src, n, iters, bad);
    run_form<11>(src, n, iters, bad);
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 1000;
    const int blocks = argc > 2 ? atoi(argv[2]) : 1024;
    const int iters = argc > 3 ? atoi(argv[3]) : 16;
    const int n = 1 << 22;
    float* src;
    unsigned* bad;
    CK(hipMalloc(&src, (size_t)n * sizeof(float)));
    CK(hipMalloc(&bad, NFORM * 8 * sizeof(unsigned)));
    float* h = (float*)malloc((size_t)n * sizeof(float));
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = ((int)(s >> 8) - (1 << 23)) / (float)(1 << 23);
    }
    CK(hipMemcpy(src, h, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(bad, 0, NFORM * 8 * sizeof(unsigned)));
    for (int l = 0; l < launches; ++l) {
        hipLaunchKernelGGL(k_forms, dim3(blocks), dim3(256), 0, 0, src, n, iters, bad);
        if ((l & 63) == 63) CK(hipDeviceSynchronize());
    }
    CK(hipDeviceSynchronize());
    unsigned hb[NFORM * 8];
    CK(hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost));
    const char* names[NFORM] = {"plain", "op_sel_hi:[1,0,1]", "op_sel:[0,1,0]", "op_sel_hi:[0,1,1]", "op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]",
                                "op_sel:[1,0,0]", "op_sel:[0,1,0] op_sel_hi:[0,1,1]", "op_sel:[0,0,1]", "v_pk_mul_f32 op_sel:[0,1]",
                                "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 op_sel:[1,0]", "op_sel:[1,0,0] (operands swapped)"};
    printf("%d launches x %d blocks x 256 lanes x %d rounds of 32 packed FMAs per form; mismatching rounds per 16-lane row\n", launches, blocks, iters);
    for (int f = 0; f < NFORM; ++f)
        printf("  form %d %-50s lo: %u %u %u %u   hi: %u %u %u %u\n", f, names[f], hb[f * 8], hb[f * 8 + 1], hb[f * 8 + 2], hb[f * 8 + 3],
               hb[f * 8 + 4], hb[f * 8 + 5], hb[f * 8 + 6], hb[f * 8 + 7]);
    return 0;
}
