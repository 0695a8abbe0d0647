// Shared host/device helpers for the cfdbench_amd HIP sources.
#pragma once
#include <cfd_intrinsics.h>
#include <hip/hip_runtime.h>

#include "../../include/cfdbench_amd.h"

void cfd_set_error(const char* fmt, ...);

#define CFD_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            cfd_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

#define CFD_LAUNCH_CHECK(name)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            cfd_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e_));  \
            return CFD_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)

#define CFD_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ != CFD_OK) return rc_; \
    } while (0)

// Per-kernel event timing (prof.cpp); a no-op unless cfd_prof_begin() was called.
struct CfdProfScope {
    hipStream_t st;
    int idx;
    CfdProfScope(const char* name, hipStream_t s);
    ~CfdProfScope();
};
#define CFD_PROF_CAT2(a, b) a##b
#define CFD_PROF_CAT(a, b) CFD_PROF_CAT2(a, b)
#define CFD_PROF(name, st) CfdProfScope CFD_PROF_CAT(cfd_prof_scope_, __LINE__)(name, st)

// Operator tables of one (H,W,m1,m2); all d_* pointers are device memory, fragment-major ([step][lane]).
struct cfd_plan {
    int H, W, m1, m2;
    int NJ;  // columns per lane = ceil(W/16)
    int KX;  // forward stage-1 k-steps over folded rows: ceil((H/2+1)/4)
    int T;   // inverse x tiles: ceil(H/16)
    int SA;  // inverse stage-A k-steps: 4 (cos, kappa 0..15) + ceil(m1/4) (sin, kappa 1..m1)
    int SB;  // inverse stage-B k-steps: ceil(2*m2/4)
    int n_fwd, n_inv;
    float* d_fwd;   // t1c[KX][64] | t1s[KX][64] | t2c[4*NJ][64] | t2s[4*NJ][64]
    float* d_inv;   // ta[T][SA][64] | tb[SB][NJ][64]
    float* d_clhw;  // [m2]  c_l / (H*W)
    float* d_gx;    // [H]  np.linspace(0,1,H) as float32   (fno2d.py:251)
    float* d_gy;    // [W]
};

static inline size_t cfd_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define CFD_SQRT1_2 0.70710678118654752440f
#define CFD_INV_SQRT_2PI 0.39894228040143267794f

// nn.GELU() exact-erf form (fno2d.py:147) and its derivative Phi(x) + x phi(x).
// Phi(x) = (1 + erf(x/sqrt 2))/2 with erf from Abramowitz & Stegun 7.1.26 (|err| <= 1.5e-7, i.e. fp32 round-off
// class: nMSE vs the fp64 GELU on N(0,1) inputs 8e-15, ATen's own fp32 erff-based GELU 4e-15) -- 14 VALU
// instructions instead of the ~36 of libdevice erff, which matters because GELU is recomputed on load by every
// consumer of an activation.  The same exp(-x^2/2) serves erf and the density phi, so the derivative is 3 more ops.
__device__ __forceinline__ void cfd_gelu_terms(float x, float& Phi, float& e) {
    const float az = fabsf(x) * CFD_SQRT1_2;
    const float t = cfd_rcpf(fmaf(0.3275911f, az, 1.0f));
    e = cfd_expf(-az * az);  // exp(-x^2/2)
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = fmaf(-p * t, e, 1.0f);  // erf(|x|/sqrt 2)
    Phi = fmaf(0.5f, copysignf(y, x), 0.5f);
}
__device__ __forceinline__ float cfd_gelu(float x) {
    float Phi, e;
    cfd_gelu_terms(x, Phi, e);
    return x * Phi;
}
__device__ __forceinline__ float cfd_gelu_grad(float x) {
    float Phi, e;
    cfd_gelu_terms(x, Phi, e);
    return fmaf(x * CFD_INV_SQRT_2PI, e, Phi);
}
