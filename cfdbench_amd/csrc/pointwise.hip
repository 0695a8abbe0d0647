// Channel-mixing (1x1 conv) kernels, the Fno2d input stem, MseLoss reductions and the flat Adam step.
//   nn.Conv2d(k=1)            src/models/fno/fno2d.py:104,150      -> k_chanmix / k_chan_wgrad
//   feature assembly + fc0    src/models/fno/fno2d.py:189-217      -> k_stem_fwd / k_chan_wgrad<STEM>
//   MseLoss                   src/models/loss.py:22-37             -> k_loss_part / k_loss_final
//   torch.optim.Adam          src/train_auto.py:213,256            -> k_adam
#include "cfd_common.h"
#include "cfd_tail.h"

// ------------------------------------------------------------------------------------------------------
// stem: [u, v, mask, grid_x, grid_y, props...] -> fc0
// ------------------------------------------------------------------------------------------------------
template <int CP, typename TA>
__global__ __launch_bounds__(256) void k_stem_fwd(const float* __restrict__ inputs, const float* __restrict__ mask,
                                                  const float* __restrict__ cp, const float* __restrict__ gx,
                                                  const float* __restrict__ gy, const float* __restrict__ w,
                                                  const float* __restrict__ bias, TA* __restrict__ out, int B,
                                                  int in_chan, int P, int C, int H, int W) {
    __shared__ float s_w[32 * CP];  // [feature][out], zero padded
    __shared__ float s_b[CP];
    const int F = in_chan + 3 + P;
    {   // (loads from clamped addresses first, zeroing afterwards: `inside ? w[..] : 0` is a branch with a full wait per load)
        constexpr int NV = (32 * CP + 255) / 256;
        float wv[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = threadIdx.x + 256 * k, f = i / CP, o = i % CP;
            wv[k] = w[(o < C ? o : C - 1) * F + (f < F ? f : F - 1)];
        }
        float bv = bias[(int)threadIdx.x < C ? threadIdx.x : C - 1];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = threadIdx.x + 256 * k, f = i / CP, o = i % CP;
            wv[k] = cfd_opaque_f(wv[k]);
            if (i < 32 * CP) s_w[i] = (f < F && o < C) ? wv[k] : 0.f;
        }
        bv = cfd_opaque_f(bv);
        if (threadIdx.x < CP) s_b[threadIdx.x] = (int)threadIdx.x < C ? bv : 0.f;
    }
    __syncthreads();
    const int HW = H * W;
    const long total = (long)B * HW;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / HW), p = (int)(idx - (long)b * HW);
        const int row = p / W, col = p - row * W;
        const int z0 = cfd_opaque(0);  // keeps the (loop-invariant) LDS weight reads from being hoisted into VGPRs
        float acc[CP];
#pragma unroll
        for (int o = 0; o < CP; ++o) acc[o] = s_b[z0 + o];
        auto add = [&](int f, float v) {
#pragma unroll
            for (int o = 0; o < CP; ++o) acc[o] = fmaf(s_w[z0 + f * CP + o], v, acc[o]);
        };
        // every operand of the pixel is requested before the first one is used (the loops below waited for each load in turn: ten
        // exposed memory round trips per pixel, which is what a 64-case rollout step paid 16 us for)
        constexpr int MC = 4, MP = 8;
        if (in_chan <= MC && P <= MP) {
            float vin[MC], cpv[MP];
#pragma unroll
            for (int c = 0; c < MC; ++c) vin[c] = inputs[((size_t)b * in_chan + (c < in_chan ? c : in_chan - 1)) * HW + p];
            float vm = mask ? mask[(size_t)b * HW + p] : 1.f;
            float gxv = gx[row], gyv = gy[col];
#pragma unroll
            for (int k = 0; k < MP; ++k) cpv[k] = P > 0 ? cp[(size_t)b * P + (k < P ? k : P - 1)] : 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) vin[c] = cfd_opaque_f(vin[c]);
            vm = cfd_opaque_f(vm), gxv = cfd_opaque_f(gxv), gyv = cfd_opaque_f(gyv);
#pragma unroll
            for (int k = 0; k < MP; ++k) cpv[k] = cfd_opaque_f(cpv[k]);
#pragma unroll
            for (int c = 0; c < MC; ++c)
                if (c < in_chan) add(c, vin[c]);
            add(in_chan, vm);        // fno2d.py:189-195
            add(in_chan + 1, gxv);   // grid_x varies along rows, fno2d.py:251-252
            add(in_chan + 2, gyv);
#pragma unroll
            for (int k = 0; k < MP; ++k)
                if (k < P) add(in_chan + 3 + k, cpv[k]);
        } else {
            for (int c = 0; c < in_chan; ++c) add(c, inputs[((size_t)b * in_chan + c) * HW + p]);
            add(in_chan, mask ? mask[(size_t)b * HW + p] : 1.f);
            add(in_chan + 1, gx[row]);
            add(in_chan + 2, gy[col]);
            for (int k = 0; k < P; ++k) add(in_chan + 3 + k, cp[(size_t)b * P + k]);
        }
#pragma unroll
        for (int o = 0; o < CP; ++o)
            if (o < C) cfd_st(out + ((size_t)b * C + o) * HW + p, acc[o]);
    }
}

// Four pixels of one row per thread (W % 4 == 0, 16-byte aligned tensors): 16-byte loads / stores, and every weight read
// from LDS (one ds_read_b128 = four output channels) is used for four pixels -- the one-pixel form above issues 240
// 4-byte LDS reads per pixel and is bound by that, not by its 84 MB of output.
template <int CP>
__global__ __launch_bounds__(256) void k_stem_fwd4(const float* __restrict__ inputs, const float* __restrict__ mask,
                                                   const float* __restrict__ cp, const float* __restrict__ gx,
                                                   const float* __restrict__ gy, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out, int B,
                                                   int in_chan, int P, int C, int H, int W) {
    static_assert(CP % 4 == 0, "k_stem_fwd4: output channels are read four at a time");
    __shared__ __attribute__((aligned(16))) float s_w[32 * CP];  // [feature][out], zero padded
    __shared__ __attribute__((aligned(16))) float s_b[CP];
    const int F = in_chan + 3 + P;
    {   // (loads from clamped addresses first, zeroing afterwards: `inside ? w[..] : 0` is a branch with a full wait per load)
        constexpr int NV = (32 * CP + 255) / 256;
        float wv[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = threadIdx.x + 256 * k, f = i / CP, o = i % CP;
            wv[k] = w[(o < C ? o : C - 1) * F + (f < F ? f : F - 1)];
        }
        float bv = bias[(int)threadIdx.x < C ? threadIdx.x : C - 1];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int i = threadIdx.x + 256 * k, f = i / CP, o = i % CP;
            wv[k] = cfd_opaque_f(wv[k]);
            if (i < 32 * CP) s_w[i] = (f < F && o < C) ? wv[k] : 0.f;
        }
        bv = cfd_opaque_f(bv);
        if (threadIdx.x < CP) s_b[threadIdx.x] = (int)threadIdx.x < C ? bv : 0.f;
    }
    __syncthreads();
    const int HW = H * W, Q = HW / 4;
    const long total = (long)B * Q;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / Q), p = (int)(idx - (long)b * Q) * 4;
        const int row = p / W, col = p - row * W;
        const int z0 = cfd_opaque(0);  // keeps the (loop-invariant) LDS weight reads from being hoisted into VGPRs
        const float4* w4 = reinterpret_cast<const float4*>(s_w + z0);
        float4 acc[CP];
#pragma unroll
        for (int o4 = 0; o4 < CP / 4; ++o4) {
            const float4 bb = reinterpret_cast<const float4*>(s_b + z0)[o4];
            acc[4 * o4] = make_float4(bb.x, bb.x, bb.x, bb.x);
            acc[4 * o4 + 1] = make_float4(bb.y, bb.y, bb.y, bb.y);
            acc[4 * o4 + 2] = make_float4(bb.z, bb.z, bb.z, bb.z);
            acc[4 * o4 + 3] = make_float4(bb.w, bb.w, bb.w, bb.w);
        }
        auto add = [&](int f, const float4 v) {
#pragma unroll
            for (int o4 = 0; o4 < CP / 4; ++o4) {
                const float4 ww = w4[f * (CP / 4) + o4];
                const float wv[4] = {ww.x, ww.y, ww.z, ww.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float4& a = acc[4 * o4 + k];
                    a.x = fmaf(wv[k], v.x, a.x);
                    a.y = fmaf(wv[k], v.y, a.y);
                    a.z = fmaf(wv[k], v.z, a.z);
                    a.w = fmaf(wv[k], v.w, a.w);
                }
            }
        };
        constexpr int MC = 4, MP = 8;
        if (in_chan <= MC && P <= MP) {  // (all operands requested before the first use: see k_stem_fwd)
            f32x4 vin[MC], vm = {1.f, 1.f, 1.f, 1.f}, gyv;
            float cpv[MP];
#pragma unroll
            for (int c = 0; c < MC; ++c)
                vin[c] = *reinterpret_cast<const f32x4*>(inputs + ((size_t)b * in_chan + (c < in_chan ? c : in_chan - 1)) * HW + p);
            if (mask) vm = *reinterpret_cast<const f32x4*>(mask + (size_t)b * HW + p);
            float gxv = gx[row];
            gyv = *reinterpret_cast<const f32x4*>(gy + col);
#pragma unroll
            for (int k = 0; k < MP; ++k) cpv[k] = P > 0 ? cp[(size_t)b * P + (k < P ? k : P - 1)] : 0.f;
#pragma unroll
            for (int c = 0; c < MC; ++c) vin[c] = cfd_opaque_f4(vin[c]);
            vm = cfd_opaque_f4(vm), gyv = cfd_opaque_f4(gyv), gxv = cfd_opaque_f(gxv);
#pragma unroll
            for (int k = 0; k < MP; ++k) cpv[k] = cfd_opaque_f(cpv[k]);
            const auto f4 = [](const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); };
#pragma unroll
            for (int c = 0; c < MC; ++c)
                if (c < in_chan) add(c, f4(vin[c]));
            add(in_chan, f4(vm));
            add(in_chan + 1, make_float4(gxv, gxv, gxv, gxv));
            add(in_chan + 2, f4(gyv));
#pragma unroll
            for (int k = 0; k < MP; ++k)
                if (k < P) add(in_chan + 3 + k, make_float4(cpv[k], cpv[k], cpv[k], cpv[k]));
        } else {
            for (int c = 0; c < in_chan; ++c)
                add(c, *reinterpret_cast<const float4*>(inputs + ((size_t)b * in_chan + c) * HW + p));
            add(in_chan, mask ? *reinterpret_cast<const float4*>(mask + (size_t)b * HW + p) : make_float4(1.f, 1.f, 1.f, 1.f));
            const float gxv = gx[row];
            add(in_chan + 1, make_float4(gxv, gxv, gxv, gxv));
            add(in_chan + 2, *reinterpret_cast<const float4*>(gy + col));
            for (int k = 0; k < P; ++k) {
                const float c1 = cp[(size_t)b * P + k];
                add(in_chan + 3 + k, make_float4(c1, c1, c1, c1));
            }
        }
#pragma unroll
        for (int o = 0; o < CP; ++o)
            if (o < C) *reinterpret_cast<float4*>(out + ((size_t)b * C + o) * HW + p) = acc[o];
    }
}

extern "C" int cfd_fno_stem_fwd(const cfd_plan* p, const float* inputs, const float* mask, const float* case_params,
                                const float* w, const float* bias, float* out, int B, int in_chan, int P, int C,
                                void* stream) {
    return cfd_int_fno_stem_fwd(p, inputs, mask, case_params, w, bias, out, B, in_chan, P, C, CFD_DT_F32, stream);
}

int cfd_int_fno_stem_fwd(const cfd_plan* p, const float* inputs, const float* mask, const float* case_params, const float* w,
                         const float* bias, void* out_, int B, int in_chan, int P, int C, int dt, void* stream) {
    float* out = (float*)out_;
    CFD_REQUIRE(p && inputs && w && bias && out && (P == 0 || case_params), CFD_ERR_INVALID_ARG, "cfd_fno_stem_fwd: NULL pointer");
    CFD_REQUIRE(B >= 0 && in_chan >= 1 && P >= 0 && C >= 1, CFD_ERR_INVALID_ARG, "cfd_fno_stem_fwd: bad sizes");
    CFD_REQUIRE(in_chan + 3 + P <= 32 && C <= 32, CFD_ERR_UNSUPPORTED,
                "cfd_fno_stem_fwd: features=%d (max 32) / hidden=%d (max 32) unsupported", in_chan + 3 + P, C);
    if (B == 0) return CFD_OK;
    const long total = (long)B * p->H * p->W;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipStream_t st = (hipStream_t)stream;
    CFD_PROF_W("k_stem_fwd", st, B * p->H * p->W * (4.0 * (in_chan + 1) + (double)cfd_dt_size(dt) * C),
               2.0 * B * p->H * p->W * (double)C * (in_chan + 3 + P));
#define CFD_STEM(CPV)                                                                                              \
    do {                                                                                                           \
        if (dt == CFD_DT_BF16)                                                                                     \
            hipLaunchKernelGGL((k_stem_fwd<CPV, __bf16>), dim3(blocks), dim3(256), 0, st, inputs, mask, case_params, \
                               (const float*)p->d_gx, (const float*)p->d_gy, w, bias, (__bf16*)out_, B, in_chan, P, C, p->H, p->W); \
        else                                                                                                       \
            hipLaunchKernelGGL((k_stem_fwd<CPV, float>), dim3(blocks), dim3(256), 0, st, inputs, mask, case_params, \
                               (const float*)p->d_gx, (const float*)p->d_gy, w, bias, out, B, in_chan, P, C, p->H, p->W); \
    } while (0)
    const bool v4 = dt == CFD_DT_F32 && p->W % 4 == 0 && (((uintptr_t)inputs | (uintptr_t)mask | (uintptr_t)out | (uintptr_t)p->d_gy) % 16) == 0;
    if (v4) {
        const long quads = total / 4;
        int blocks4 = (int)((quads + 255) / 256);
        if (blocks4 > 2048) blocks4 = 2048;
#define CFD_STEM4(CPV)                                                                                             \
    hipLaunchKernelGGL((k_stem_fwd4<CPV>), dim3(blocks4), dim3(256), 0, st, inputs, mask, case_params,             \
                       (const float*)p->d_gx, (const float*)p->d_gy, w, bias, out, B, in_chan, P, C, p->H, p->W)
        if (C <= 8) CFD_STEM4(8);
        else if (C <= 16) CFD_STEM4(16);
        else if (C <= 24) CFD_STEM4(24);
        else CFD_STEM4(32);
#undef CFD_STEM4
    }
    else if (C <= 8) CFD_STEM(8);
    else if (C <= 16) CFD_STEM(16);
    else if (C <= 24) CFD_STEM(24);
    else CFD_STEM(32);
#undef CFD_STEM
    CFD_LAUNCH_CHECK("cfd_fno_stem_fwd");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// 1x1 conv forward (and its input gradient with transpose=1)
// ------------------------------------------------------------------------------------------------------
template <int CPO, int VEC, bool ACT, typename TA = float, typename TO = TA>
__global__ __launch_bounds__(256) void k_chanmix(const TA* __restrict__ in, const float* __restrict__ w,
                                                 const float* __restrict__ bias, TO* __restrict__ out, int B, int Ci,
                                                 int Co, int HW, int transpose) {
    __shared__ float s_w[32 * CPO];  // [in][out], zero padded
    __shared__ float s_b[CPO];
    for (int i = threadIdx.x; i < 32 * CPO; i += blockDim.x) {
        const int ci = i / CPO, o = i % CPO;
        float v = 0.f;
        if (ci < Ci && o < Co) v = transpose ? w[ci * Co + o] : w[o * Ci + ci];
        s_w[i] = v;
    }
    for (int i = threadIdx.x; i < CPO; i += blockDim.x) s_b[i] = (bias && i < Co) ? bias[i] : 0.f;
    __syncthreads();
    const int units = HW / VEC;  // VEC divides HW (checked by the launcher)
    const long total = (long)B * units;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / units), p = (int)(idx - (long)b * units) * VEC;
        const int z0 = cfd_opaque(0);  // see k_stem_fwd
        float acc[CPO][VEC];
#pragma unroll
        for (int o = 0; o < CPO; ++o)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[o][k] = s_b[z0 + o];
        for (int ci = 0; ci < Ci; ++ci) {
            float v[VEC];
            const TA* src = in + ((size_t)b * Ci + ci) * HW + p;
            if constexpr (sizeof(TA) == 2) {  // bf16 storage: VEC consecutive 2-byte values
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[k] = cfd_ld(src + k);
            } else if constexpr (VEC == 4) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else if constexpr (VEC == 2) {
                const float2 t = *reinterpret_cast<const float2*>(src);
                v[0] = t.x; v[1] = t.y;
            } else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[k] = cfd_ld(src + k);
            }
            if constexpr (ACT) {
#pragma unroll
                for (int k = 0; k + 1 < VEC; k += 2) { const cfd_f2 g2 = cfd_gelu2(cfd_f2{v[k], v[k + 1]}); v[k] = g2.x; v[k + 1] = g2.y; }
                if constexpr (VEC & 1) v[VEC - 1] = cfd_gelu(v[VEC - 1]);
            }
#pragma unroll
            for (int o = 0; o < CPO; ++o) {
                const float wv = s_w[z0 + ci * CPO + o];
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[o][k] = fmaf(wv, v[k], acc[o][k]);
            }
        }
#pragma unroll
        for (int o = 0; o < CPO; ++o) {
            if (o < Co) {
                TO* dst = out + ((size_t)b * Co + o) * HW + p;
                if constexpr (sizeof(TO) == 2) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) cfd_st(dst + k, acc[o][k]);
                } else if constexpr (VEC == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]);
                } else if constexpr (VEC == 2) {
                    *reinterpret_cast<float2*>(dst) = make_float2(acc[o][0], acc[o][1]);
                } else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) cfd_st(dst + k, acc[o][k]);
                }
            }
        }
    }
}

// ---- the same contraction on the matrix pipe (round 2) -------------------------------------------------------------
// out[d][px] = bias[d] + sum_s W[d][s] f(in[s][px]) is ONE K = 32 split-bf16 MFMA step per 16 pixels and 16 output
// channels (Ci, Co <= 32): 6 MFMAs where the VALU kernel above spends Ci * Co FMAs per pixel -- at Ci = Co = 32 that
// kernel is VALU-bound (114 us at B = 256, 66 x 65: 31 % of HBM), this one streams.  Lane (n, q) of a wave loads four
// consecutive pixels 4n .. 4n+3 of the eight channels 8q .. 8q+7 (the B operand of the four 16-pixel phases: pixel 4n + j
// in phase j), the weight fragments are loop-invariant registers, and the accumulator rows d = 16 mt + 4q + r of the four
// phases leave as one 16-byte store per row.  The next tile's loads are in flight during the arithmetic of this one.
template <typename TA, int VEC>
__device__ __forceinline__ void cfd_ld4px(const TA* __restrict__ p, int nvalid, float (&v)[4]) {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (nvalid <= 0) return;
    if constexpr (VEC == 4 && sizeof(TA) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (VEC == 2 && sizeof(TA) == 4) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
        if (nvalid > 2) { const float2 u = *reinterpret_cast<const float2*>(p + 2); v[2] = u.x; v[3] = u.y; }
    } else if constexpr (VEC == 4 && sizeof(TA) == 2) {  // four bf16 values in one 8-byte load
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        v[0] = __builtin_bit_cast(float, t.x << 16); v[1] = __builtin_bit_cast(float, t.x & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, t.y << 16); v[3] = __builtin_bit_cast(float, t.y & 0xffff0000u);
    } else if constexpr (VEC == 2 && sizeof(TA) == 2) {
        const unsigned t = *reinterpret_cast<const unsigned*>(p);
        v[0] = __builtin_bit_cast(float, t << 16); v[1] = __builtin_bit_cast(float, t & 0xffff0000u);
        if (nvalid > 2) {
            const unsigned u = *reinterpret_cast<const unsigned*>(p + 2);
            v[2] = __builtin_bit_cast(float, u << 16); v[3] = __builtin_bit_cast(float, u & 0xffff0000u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nvalid) v[j] = cfd_ld(p + j);
    }
}

template <int MT, int VEC, bool ACT, typename TA, int AP>
__global__ __launch_bounds__(256, 2) void k_chanmix_b3(const TA* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int B,
                                                       int Ci, int Co, int HW, int transpose) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    // A operand: W[d = 16 mt + n][s = 8q + v] in THREE bf16 pieces (24 significant bits = the fp32 weight exactly): the
    // weights are the same in every launch, so their rounding is a fixed perturbation of the operator that a 200-step rollout
    // through a near-identity network accumulates coherently (2-piece weights: nMSE 3.0e-7 against the reference at step
    // 200 of tests/golden/rollout200_c32_66x65, 3-piece: see DESIGN.md); the activations keep two pieces, whose rounding
    // differs from value to value.  Five MFMAs per tile-step instead of three; the kernel is memory-bound either way.
    const int tpb = (HW + 63) / 64;
    const int total = B * tpb;  // < 2^30 (checked by the launcher)
    const int stride = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;
    int b = -1, px = 0;
    auto locate = [&](int t) {
        b = -1;
        px = 0;
        if (t < total) {
            const unsigned ub = (unsigned)t / (unsigned)tpb;
            b = (int)ub;
            px = (int)((unsigned)t - ub * (unsigned)tpb) * 64 + 4 * n;
        }
    };
    float hn[8][4];
    auto fetch = [&]() {
        // (round 4: unaligned 16-byte loads were tried here for the 66 x 65 planes and measured SLOWER than the 8-byte pairs -- 29.5 vs
        // 25.9 us at C = 32, 64 cases -- unlike in k_chan_wgrad and the head kernels, where they win)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = 8 * q + c;
            const bool live = b >= 0 && i < Ci;
            cfd_ld4px<TA, VEC>(in + ((size_t)(live ? b : 0) * Ci + (live ? i : 0)) * HW + (live ? px : 0), live ? HW - px : 0, hn[c]);
        }
    };
    locate(tile);
    fetch();  // (round 6: requested BEFORE the weight prologue -- with one tile per wave, as at 64 rollout cases, the two round trips ran one after the other)
    CfdTab3 wf[MT];    // (round 4: the activations carry AP = 2 or 3 pieces -- five or six MFMAs per tile-step, cfd_common.h)
    float bz[MT][4];   // bias of accumulator row d = 16 mt + 4q + r
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        float x[8];
        const int d = 16 * mt + n;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const int sc = 8 * q + v;
            x[v] = (d < Co && sc < Ci) ? (transpose ? w[sc * Co + d] : w[d * Ci + sc]) : 0.f;
        }
        // round-to-nearest pieces as in rounds 2-3 (hi, lo, then the bf16 of what two pieces miss): the default route's results stay bit-identical
        const CfdSplit8 w2 = cfd_split8(x);
        float x3[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) x3[v] = (x[v] - (float)w2.hi[v]) - (float)w2.lo[v];  // exact
        wf[mt].p[0] = w2.hi; wf[mt].p[1] = w2.lo; wf[mt].p[2] = cfd_split8(x3).hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[mt][r] = (bias && 16 * mt + 4 * q + r < Co) ? bias[16 * mt + 4 * q + r] : 0.f;
    }
    for (; tile < total; tile += stride) {
        float h[8][4];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) h[c][j] = hn[c][j];
        const int bc = b, pxc = px;
        locate(tile + stride);
        fetch();
        cfd_sched_fence();  // the prefetch stays ahead of this tile's arithmetic
        if constexpr (ACT) {
#pragma unroll
            for (int c = 0; c < 8; ++c) cfd_gelu4(h[c][0], h[c][1], h[c][2], h[c][3]);  // gelu(0) = 0: dead slots stay zero
        }
        float o[MT][4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float xk[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) xk[c] = h[c][j];
            const CfdAct8<AP> bs = cfd_act_split8<AP>(xk);
            f32x4 z[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) z[mt] = f32x4{bz[mt][0], bz[mt][1], bz[mt][2], bz[mt][3]};
            if constexpr (AP == 2) {  // five terms in the round-3 order (w_lo2 x_hi, w_lo x_lo, w_lo x_hi, w_hi x_lo, w_hi x_hi)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) z[mt] = cfd_mfma16x16x32_bf16(wf[mt].p[2], bs.p[0], z[mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) z[mt] = cfd_mfma16x16x32_bf16(wf[mt].p[1], bs.p[1], z[mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) z[mt] = cfd_mfma16x16x32_bf16(wf[mt].p[1], bs.p[0], z[mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) z[mt] = cfd_mfma16x16x32_bf16(wf[mt].p[0], bs.p[1], z[mt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) z[mt] = cfd_mfma16x16x32_bf16(wf[mt].p[0], bs.p[0], z[mt]);
            } else {
#pragma unroll
                for (int k = 0; k < cfd_nterm(AP); ++k)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) z[mt] = cfd_term<false, AP>(bs, wf[mt], k, z[mt]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[mt][r][j] = z[mt][r];
        }
        const int nvalid = HW - pxc;
        if (nvalid > 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * mt + 4 * q + r;
                    if (d >= Co) continue;
                    float* dst = out + ((size_t)bc * Co + d) * HW + pxc;
                    if constexpr (VEC == 4) {
                        *reinterpret_cast<float4*>(dst) = make_float4(o[mt][r][0], o[mt][r][1], o[mt][r][2], o[mt][r][3]);
                    } else if constexpr (VEC == 2) {
                        *reinterpret_cast<float2*>(dst) = make_float2(o[mt][r][0], o[mt][r][1]);
                        if (nvalid > 2) *reinterpret_cast<float2*>(dst + 2) = make_float2(o[mt][r][2], o[mt][r][3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j < nvalid) dst[j] = o[mt][r][j];
                    }
                }
        }
    }
}

template <typename TA>
static int launch_chanmix_b3(const TA* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int HW, int act,
                             int transpose, hipStream_t st) {
    CFD_REQUIRE((long)B * ((HW + 63) / 64) < (1L << 30), CFD_ERR_UNSUPPORTED, "cfd_chanmix: B * ceil(HW / 64) = %ld tiles (max 2^30)",
                (long)B * ((HW + 63) / 64));
    const long tiles = (long)B * ((HW + 63) / 64);
    int blocks = (int)((tiles + 3) / 4);
    if (blocks > 1024) blocks = 1024;
    // widest access both tensors allow: the 4-pixel groups of a (b, channel) row start at multiples of 4 elements from a
    // row start that is a multiple of HW elements
    const size_t ea = sizeof(TA);
    const int vec = (HW % 4 == 0 && ((uintptr_t)in % (4 * ea)) == 0 && ((uintptr_t)out % 16) == 0)   ? 4
                    : (HW % 2 == 0 && ((uintptr_t)in % (2 * ea)) == 0 && ((uintptr_t)out % 8) == 0) ? 2
                                                                                                    : 1;
    CFD_PROF_W(transpose ? "k_chanmix_t" : (act ? "k_chanmix_act" : "k_chanmix"), st, B * HW * ((double)ea * Ci + 4.0 * Co),
               2.0 * B * HW * (double)Ci * Co);
    const bool ap3 = sizeof(TA) == 4 && cfd_act_pieces() == 3;  // bf16 storage keeps two pieces
#define CFD_CB3_P(M_, V_, A_, P_) \
    hipLaunchKernelGGL((k_chanmix_b3<M_, V_, A_, TA, P_>), dim3(blocks), dim3(256), 0, st, in, w, bias, out, B, Ci, Co, HW, transpose)
#define CFD_CB3(M_, V_, A_)                                                                   \
    do {                                                                                      \
        if constexpr (sizeof(TA) == 4) { if (ap3) CFD_CB3_P(M_, V_, A_, 3); else CFD_CB3_P(M_, V_, A_, 2); } \
        else CFD_CB3_P(M_, V_, A_, 2);                                                        \
    } while (0)
#define CFD_CB3_V(M_)                                                      \
    do {                                                                   \
        if (vec == 4) { if (act) CFD_CB3(M_, 4, true); else CFD_CB3(M_, 4, false); } \
        else if (vec == 2) { if (act) CFD_CB3(M_, 2, true); else CFD_CB3(M_, 2, false); } \
        else { if (act) CFD_CB3(M_, 1, true); else CFD_CB3(M_, 1, false); }  \
    } while (0)
    if (Co <= 16) CFD_CB3_V(1);
    else CFD_CB3_V(2);
#undef CFD_CB3_V
#undef CFD_CB3
#undef CFD_CB3_P
    CFD_LAUNCH_CHECK("cfd_chanmix");
    return CFD_OK;
}

template <int CPO, int VEC>
static int launch_chanmix(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int HW,
                          int act, int transpose, hipStream_t st) {
    const long total = (long)B * (HW / VEC);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    CFD_PROF_W(transpose ? "k_chanmix_t" : (act ? "k_chanmix_act" : "k_chanmix"), st, 4.0 * B * HW * ((double)Ci + Co),
               2.0 * B * HW * (double)Ci * Co);
    if (act)
        hipLaunchKernelGGL((k_chanmix<CPO, VEC, true>), dim3(blocks), dim3(256), 0, st, in, w, bias, out, B, Ci, Co, HW, transpose);
    else
        hipLaunchKernelGGL((k_chanmix<CPO, VEC, false>), dim3(blocks), dim3(256), 0, st, in, w, bias, out, B, Ci, Co, HW, transpose);
    CFD_LAUNCH_CHECK("cfd_chanmix");
    return CFD_OK;
}

int cfd_int_chanmix(const void* in, const float* w, const float* bias, void* out, int B, int Ci, int Co, int HW, int act_in,
                    int transpose, int dt, void* stream) {
    if (dt == CFD_DT_F32) return cfd_chanmix((const float*)in, w, bias, (float*)out, B, Ci, Co, HW, act_in, transpose, stream);
    CFD_REQUIRE(in && w && out && B >= 0 && Ci >= 1 && Co >= 1 && HW >= 1 && Ci <= 32 && Co <= 32, CFD_ERR_INVALID_ARG, "cfd_chanmix: bad arguments");
    if (B == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    if (cfd_tune_get(CFD_TUNE_EXACT_FP32) != 1)
        return launch_chanmix_b3<__bf16>((const __bf16*)in, w, bias, (float*)out, B, Ci, Co, HW, act_in, transpose, st);
    const int VEC = HW % 2 == 0 ? 2 : 1;
    const long total = (long)B * (HW / VEC);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    CFD_PROF_W(act_in ? "k_chanmix_act" : "k_chanmix", st, B * HW * (2.0 * Ci + 4.0 * Co), 2.0 * B * HW * (double)Ci * Co);
#define CFD_CMB(CPV, VW, A_)                                                                                                  \
    hipLaunchKernelGGL((k_chanmix<CPV, VW, A_, __bf16, float>), dim3(blocks), dim3(256), 0, st, (const __bf16*)in, w, bias, (float*)out, B, \
                       Ci, Co, HW, transpose)
#define CFD_CMB_VA(CPV)                                                    \
    do {                                                                   \
        if (VEC == 2) { if (act_in) CFD_CMB(CPV, 2, true); else CFD_CMB(CPV, 2, false); } \
        else { if (act_in) CFD_CMB(CPV, 1, true); else CFD_CMB(CPV, 1, false); }          \
    } while (0)
    if (Co <= 8) CFD_CMB_VA(8);
    else if (Co <= 16) CFD_CMB_VA(16);
    else if (Co <= 24) CFD_CMB_VA(24);
    else CFD_CMB_VA(32);
#undef CFD_CMB_VA
#undef CFD_CMB
    CFD_LAUNCH_CHECK("cfd_chanmix(bf16 storage)");
    return CFD_OK;
}

extern "C" int cfd_chanmix(const float* in, const float* w, const float* bias, float* out, int B, int Ci, int Co, int HW,
                           int act_in, int transpose, void* stream) {
    CFD_REQUIRE(in && w && out, CFD_ERR_INVALID_ARG, "cfd_chanmix: NULL pointer");
    CFD_REQUIRE(B >= 0 && Ci >= 1 && Co >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "cfd_chanmix: bad sizes");
    CFD_REQUIRE(Ci <= 32 && Co <= 32, CFD_ERR_UNSUPPORTED, "cfd_chanmix: Ci=%d Co=%d (max 32) unsupported", Ci, Co);
    if (B == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    if (cfd_tune_get(CFD_TUNE_EXACT_FP32) != 1)  // split-bf16 MFMA form; the VALU kernel below is the exact-fp32 route
        return launch_chanmix_b3<float>(in, w, bias, out, B, Ci, Co, HW, act_in, transpose, st);
    // pixels per thread: 4 (16-B accesses) while the accumulator tile stays small, 2 (8-B) for wide outputs
    const bool v4 = HW % 4 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
    const bool v2 = HW % 2 == 0 && ((uintptr_t)in % 8) == 0 && ((uintptr_t)out % 8) == 0;
#define CFD_CM(CPV, VW)                                                                               \
    return (VW == 4 && v4) ? launch_chanmix<CPV, VW>(in, w, bias, out, B, Ci, Co, HW, act_in, transpose, st) \
           : v2            ? launch_chanmix<CPV, 2>(in, w, bias, out, B, Ci, Co, HW, act_in, transpose, st)  \
                           : launch_chanmix<CPV, 1>(in, w, bias, out, B, Ci, Co, HW, act_in, transpose, st)
    if (Co <= 8) CFD_CM(8, 4);
    if (Co <= 16) CFD_CM(16, 4);
    if (Co <= 24) CFD_CM(24, 2);
    CFD_CM(32, 2);
#undef CFD_CM
}

// ------------------------------------------------------------------------------------------------------
// weight/bias gradient of a 1x1 conv on the matrix pipe:  gw[o][i] = sum_{b,p} g[b,o,p] f(in[b,i,p])
// M = out channel, N = in channel (+1 "ones" column that yields the bias gradient), K = pixels.
// ------------------------------------------------------------------------------------------------------
struct StemSrc {  // feature source for the fc0 weight gradient (features are never materialised)
    const float* mask;
    const float* cp;
    const float* gx;
    const float* gy;
    int in_chan, P, W;
};

// eight consecutive pixels of one row of a (b, channel) plane: two 16-byte, four 8-byte (HW even: 66 x 65 grids) or eight
// 4-byte loads; pixels at or past `hw` read as zero (whole vectors: px and hw are multiples of the vector width)
template <int VEC>
__device__ __forceinline__ void cfd_ld8px(const float* __restrict__ src, int px, int hw, bool ok, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (!ok) return;
    if constexpr (VEC == 4) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (px + 4 * k < hw) {
                const float4 t = *reinterpret_cast<const float4*>(src + 4 * k);
                v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
            }
    } else {  // VEC = 2 / 1 (planes not 16-byte aligned, 66 x 65): unaligned 16-byte loads, scalars across a plane's end
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (px + 4 * k + 3 < hw) {
                const float4 t = cfd_ld4u(src + 4 * k);
                v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (px + 4 * k + j < hw) v[4 * k + j] = src[4 * k + j];
            }
        }
    }
}

// the same for a plane stored as bf16 (bf16-storage training): VEC = 4 -> ONE 16-byte load of the eight values
template <int VEC>
__device__ __forceinline__ void cfd_ld8px(const __bf16* __restrict__ src, int px, int hw, bool ok, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (!ok) return;
    if constexpr (VEC == 4) {
        if (px + 4 < hw) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
        } else if (px < hw) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (float)src[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (px + j < hw) v[j] = (float)src[j];
    }
}

// TIN: storage type of `in` (float; __bf16 for the saved activations of bf16-storage training -- the gradient `g` is always fp32)
template <int MT, int NT, int VEC, bool ACT, bool STEM, typename TIN, int AP>
__global__ __launch_bounds__(256) void k_chan_wgrad(const float* __restrict__ g, const TIN* __restrict__ in,
                                                    StemSrc ss, float* __restrict__ part, int B, int Ci, int Co,
                                                    int HW) {
    __shared__ float s_red[4 * MT * NT * 4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[a][c] = zero;
    const int cpb = (HW + 31) / 32;  // 32-pixel chunks per batch entry
    const int total = B * cpb;       // < 2^30 (checked by the launcher): 32-bit index arithmetic
    const int CiIn = STEM ? ss.in_chan : Ci;  // channels physically present in `in`
    // One chunk = 32 pixels of every channel = ONE K = 32 step of the split-bf16 MFMA (cfd_common.h): lane (q, n) holds
    // pixels 8q .. 8q+7 of rows 16a + n (gradient) and 16c + n (input), straight from global memory in operand order.
    // load() only issues the global reads (raw values); GELU, the bf16 split and the MFMAs happen one iteration later,
    // so the next chunk's reads are in flight while this chunk is on the matrix pipe.
    auto load = [&](int ch, float (&av)[MT][8], float (&bv)[NT][8]) {
        const int b = (int)((unsigned)ch / (unsigned)cpb);
        const int px = (ch - b * cpb) * 32 + 8 * q;  // this lane's 8 pixels: px .. px+7
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int o = 16 * a + n;
            const float* src = g + ((size_t)b * Co + (o < Co ? o : 0)) * HW + px;
            cfd_ld8px<VEC>(src, px, HW, o < Co, av[a]);
        }
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const int i = 16 * c + n;
            if (i < CiIn) {
                const TIN* src = in + ((size_t)b * CiIn + i) * HW + px;
                cfd_ld8px<VEC>(src, px, HW, true, bv[c]);
            } else {
                // generated columns: ones (bias gradient) and, for the stem, mask / grid_x / grid_y / case parameters.
                // One row/column split per lane and chunk (px .. px+7 share a row whenever W % 8 == 0).
                const int f = STEM ? i - ss.in_chan : -1;
                const int row0 = STEM ? px / ss.W : 0, col0 = STEM ? px - row0 * ss.W : 0;
                const bool same_row = STEM && (ss.W % 8 == 0);
                if constexpr (STEM && VEC == 4) {
                    // Fast path (W % 8 == 0: the lane's 8 pixels share a row, col0 is a multiple of 8): every generated
                    // column is either 8 contiguous floats (mask, grid_y) -> two 16-byte loads like the field channels,
                    // or one value for all 8 pixels (grid_x, case parameter, ones) -> one load.  The per-pixel form
                    // below costs up to 8 scalar loads per lane and column, in five divergent branches.
                    if (same_row && (((uintptr_t)ss.mask | (uintptr_t)ss.gy) % 16) == 0) {
                        const float* p8 = nullptr;
                        const float* p1 = nullptr;
                        float s = 0.f;
                        if (i == Ci) s = 1.f;
                        else if (f == 0) { if (ss.mask) p8 = ss.mask + (size_t)b * HW + px; else s = 1.f; }
                        else if (f == 1) p1 = ss.gx + row0;
                        else if (f == 2) p8 = ss.gy + col0;
                        else if (f < 3 + ss.P) p1 = ss.cp + (size_t)b * ss.P + (f - 3);
                        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
                        if (p8) {
                            if (px < HW) t0 = *reinterpret_cast<const float4*>(p8);
                            if (px + 4 < HW) t1 = *reinterpret_cast<const float4*>(p8 + 4);
                        } else {
                            if (p1) s = *p1;
                            if (px < HW) t0 = make_float4(s, s, s, s);
                            if (px + 4 < HW) t1 = make_float4(s, s, s, s);
                        }
                        bv[c][0] = t0.x; bv[c][1] = t0.y; bv[c][2] = t0.z; bv[c][3] = t0.w;
                        bv[c][4] = t1.x; bv[c][5] = t1.y; bv[c][6] = t1.z; bv[c][7] = t1.w;
                        continue;
                    }
                }
                // any other grid (66 x 65: rows are not a multiple of 8 pixels).  Three lane classes instead of one branch
                // per feature and pixel: a constant (ones, case parameter, absent mask), a plane read like a field channel
                // (mask), or a table value per pixel (grid_x by row, grid_y by column, the row / column pair advanced
                // without a division).
                float cval = 0.f;
                const float* plane = nullptr;
                const float* tab = nullptr;
                bool by_row = false;
                if (i == Ci) cval = 1.f;
                else if constexpr (STEM) {
                    if (f == 0) { if (ss.mask) plane = ss.mask + (size_t)b * HW + px; else cval = 1.f; }
                    else if (f == 1) { tab = ss.gx; by_row = true; }
                    else if (f == 2) tab = ss.gy;
                    else if (f < 3 + ss.P) cval = ss.cp[(size_t)b * ss.P + (f - 3)];
                }
                if (plane) {
                    if (((uintptr_t)ss.mask % (4 * VEC)) == 0) cfd_ld8px<VEC>(plane, px, HW, true, bv[c]);
                    else cfd_ld8px<1>(plane, px, HW, true, bv[c]);
                } else if (tab) {
                    int row = row0, col = col0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        bv[c][j] = (px + j < HW) ? tab[by_row ? row : col] : 0.f;
                        if (++col >= ss.W) { col = 0; ++row; }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) bv[c][j] = (px + j < HW) ? cval : 0.f;
                }
            }
        }
    };
    const int stride = (int)gridDim.x * 4;
    int ch = (int)blockIdx.x * 4 + wave;
    float av[MT][8], bv[NT][8], avn[MT][8], bvn[NT][8];
    if (ch < total) load(ch, av, bv);
    while (ch < total) {
        const int nx = ch + stride;
        if (nx < total) load(nx, avn, bvn);
        cfd_sched_fence();
        if constexpr (ACT) {
#pragma unroll
            for (int c = 0; c < NT; ++c)
                if (16 * c + n < CiIn) {  // the ones / generated columns are not activations
#pragma unroll
                    for (int j = 0; j < 8; j += 2) { const cfd_f2 g2 = cfd_gelu2(cfd_f2{bv[c][j], bv[c][j + 1]}); bv[c][j] = g2.x; bv[c][j + 1] = g2.y; }
                }
        }
        {
            // both operands are activations, AP pieces each (3: the six products of cfd_mfma_bf16x6; 2: lo*hi, hi*lo, hi*hi),
            // term-major so that consecutive MFMAs hit different accumulators
            CfdAct8<AP> as[MT], bs[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) as[a] = cfd_act_split8<AP>(av[a]);
#pragma unroll
            for (int c = 0; c < NT; ++c) bs[c] = cfd_act_split8<AP>(bv[c]);
#pragma unroll
            for (int k = 0; k < cfd_nterm_aa(AP); ++k)
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int c = 0; c < NT; ++c)
                        acc[a][c] = cfd_mfma16x16x32_bf16(as[a].p[cfd_term_aa_a(AP, k)], bs[c].p[cfd_term_aa_b(AP, k)], acc[a][c]);
        }
        cfd_sched_fence();
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int j = 0; j < 8; ++j) av[a][j] = avn[a][j];
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[c][j] = bvn[c][j];
        ch = nx;
    }
    // block reduction of the 4 waves, then one partial tile per block
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * MT * NT + a * NT + c) * 4 + r) * 64 + lane] = acc[a][c][r];
    __syncthreads();
    // partials are stored BLOCK-major, part[block][Co (Ci + 1)] (round 5): a wave's stores are runs of consecutive elements.  (Rounds 1-4:
    // element-major, "the scattered 4-byte stores cost the producer nothing" -- they were 0.43 M write transactions per launch, 13 MB of
    // counted write traffic for 1.7 MB of partial sums; cfd_record_sum16 has the numbers.)
    const int NW = Ci + 1;
    float* dst = part + (size_t)blockIdx.x * Co * NW;
    const size_t nb = 1;
    for (int e = threadIdx.x; e < MT * NT * 4 * 64; e += blockDim.x) {
        const int ln = e & 63, r = (e >> 6) & 3, tile = e >> 8;
        const int a = tile / NT, c = tile % NT;
        const int o = 16 * a + 4 * (ln >> 4) + r, i = 16 * c + (ln & 15);
        if (o < Co && i < NW) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) s += s_red[((wv * MT * NT + tile) * 4 + r) * 64 + ln];
            dst[(size_t)(o * NW + i) * nb] = s;
        }
    }
}

// Workgroup = 16 consecutive output elements of the block-major partial records (cfd_record_sum16): fixed summation order, deterministic.
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int nblk, float* __restrict__ gw,
                                                      float* __restrict__ gb, int Co, int Ci) {
    __shared__ float s_scr[64];
    chan_wgrad_reduce_group(blockIdx.x, part, nblk, gw, gb, Co, Ci, s_scr);
}

static int wgrad_blocks(int B, int HW) {
    const long chunks = (long)B * ((HW + 31) / 32);
    long blocks = (chunks + 7) / 8;    // >= 2 chunks per wave (one in flight while one is on the matrix pipe)
    if (blocks > 1024) blocks = 1024;  // 4 workgroups per CU (measured best of 512 / 1024 / 2048 at B = 256)
    // small batches: one workgroup per CU before waves get a second chunk (B = 8: 32 workgroups took 21.6 us for 2.6 MB)
    const long fill = (chunks + 3) / 4 < 256 ? (chunks + 3) / 4 : 256;
    if (blocks < fill) blocks = fill;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

extern "C" size_t cfd_chan_wgrad_workspace_bytes(int B, int Ci, int Co, int HW) {
    if (B <= 0) return 0;
    return (size_t)wgrad_blocks(B, HW) * Co * (Ci + 1) * sizeof(float);
}

// `defer` (may be NULL): leave the partial-sum reduction to a later launch (cfd_tail.h) instead of launching it here
template <bool STEM, typename TIN = float>
static int launch_wgrad(const float* g, const TIN* in, StemSrc ss, float* gw, float* gb, void* ws, int B, int Ci,
                        int Co, int HW, int act, hipStream_t st, ChanWgradTail* defer = nullptr) {
    const int blocks = wgrad_blocks(B, HW);
    CFD_REQUIRE((long)B * ((HW + 31) / 32) < (1L << 30), CFD_ERR_UNSUPPORTED, "cfd_chan_wgrad: B * ceil(HW / 32) = %ld chunks (max 2^30)",
                (long)B * ((HW + 31) / 32));
    const int MT = (Co + 15) / 16, NT = (Ci + 1 + 15) / 16;
    // bf16 `in`: the 8-pixel groups are 16 bytes, so the vector path needs HW % 8 == 0 (64 x 64); other grids read element-wise
    const int vec = sizeof(TIN) == 2 ? ((HW % 8 == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)in % 16) == 0) ? 4 : 1)
                    : (HW % 4 == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)in % 16) == 0)  ? 4
                    : (HW % 2 == 0 && ((uintptr_t)g % 8) == 0 && ((uintptr_t)in % 8) == 0) ? 2
                                                                                            : 1;
    float* part = (float*)ws;
    const bool ap3 = sizeof(TIN) == 4 && cfd_act_pieces() == 3;  // bf16 storage keeps two pieces
#define CFD_WG_P(M_, N_, V_, A_, P_)                                                                             \
    hipLaunchKernelGGL((k_chan_wgrad<M_, N_, V_, A_, STEM, TIN, P_>), dim3(blocks), dim3(256), 0, st, g, in, ss, part, B, \
                       Ci, Co, HW)
#define CFD_WG(M_, N_, V_, A_)                                                                \
    do {                                                                                      \
        if constexpr (sizeof(TIN) == 4) { if (ap3) CFD_WG_P(M_, N_, V_, A_, 3); else CFD_WG_P(M_, N_, V_, A_, 2); } \
        else CFD_WG_P(M_, N_, V_, A_, 2);                                                     \
    } while (0)
#define CFD_WG_VA(M_, N_)                                                                     \
    do {                                                                                      \
        if (vec == 4) { if (act) CFD_WG(M_, N_, 4, true); else CFD_WG(M_, N_, 4, false); }    \
        else if (vec == 2) { if constexpr (sizeof(TIN) == 4) { if (act) CFD_WG(M_, N_, 2, true); else CFD_WG(M_, N_, 2, false); } } \
        else { if (act) CFD_WG(M_, N_, 1, true); else CFD_WG(M_, N_, 1, false); }             \
    } while (0)
    {
    // stem: the gradient tensor + the raw input channels (features are generated); otherwise gradient + activation
    CFD_PROF_W(STEM ? "k_chan_wgrad_stem" : "k_chan_wgrad", st, (double)B * HW * (4.0 * Co + sizeof(TIN) * (STEM ? 3 : Ci)),
               2.0 * B * HW * (double)Co * (Ci + 1));
    if (MT == 1 && NT == 1) CFD_WG_VA(1, 1);
    else if (MT == 1 && NT == 2) CFD_WG_VA(1, 2);
    else if (MT == 2 && NT == 1) CFD_WG_VA(2, 1);
    else if (MT == 2 && NT == 2) CFD_WG_VA(2, 2);
    else if (MT == 2 && NT == 3) CFD_WG_VA(2, 3);
    else if (MT == 1 && NT == 3) CFD_WG_VA(1, 3);
    else {
        cfd_set_error("cfd_chan_wgrad: Co=%d Ci=%d unsupported tile shape", Co, Ci);
        return CFD_ERR_UNSUPPORTED;
    }
    }
#undef CFD_WG_VA
#undef CFD_WG
#undef CFD_WG_P
    CFD_LAUNCH_CHECK("cfd_chan_wgrad");
    if (defer) {
        *defer = ChanWgradTail{(const float*)part, gw, gb, blocks, Co, Ci};
        return CFD_OK;
    }
    CFD_PROF_W("k_wgrad_reduce", st, 0.0, 0.0);  // partial sums are an implementation detail
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((Co * (Ci + 1) + 15) / 16), dim3(256), 0, st, (const float*)part, blocks,
                       gw, gb, Co, Ci);
    CFD_LAUNCH_CHECK("cfd_chan_wgrad(reduce)");
    return CFD_OK;
}

extern "C" int cfd_chan_wgrad(const float* g, const float* in, float* gw, float* gb, void* ws, int B, int Ci, int Co,
                              int HW, int act_in, void* stream) {
    CFD_REQUIRE(g && in && gw && ws, CFD_ERR_INVALID_ARG, "cfd_chan_wgrad: NULL pointer");
    CFD_REQUIRE(B >= 1 && Ci >= 1 && Co >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "cfd_chan_wgrad: bad sizes");
    CFD_REQUIRE(Ci <= 32 && Co <= 32, CFD_ERR_UNSUPPORTED, "cfd_chan_wgrad: Ci=%d Co=%d (max 32) unsupported", Ci, Co);
    StemSrc ss{};
    return launch_wgrad<false>(g, in, ss, gw, gb, ws, B, Ci, Co, HW, act_in, (hipStream_t)stream);
}

int cfd_int_chan_wgrad(const float* g, const float* in, float* gw, float* gb, void* ws, int B, int Ci, int Co, int HW,
                       int act_in, void* stream, ChanWgradTail* defer) {
    return cfd_int_chan_wgrad_dt(g, in, gw, gb, ws, B, Ci, Co, HW, act_in, CFD_DT_F32, stream, defer);
}

// `in` stored as dt (CFD_DT_BF16: the saved activations of bf16-storage training); the gradient and the results are fp32
int cfd_int_chan_wgrad_dt(const float* g, const void* in, float* gw, float* gb, void* ws, int B, int Ci, int Co, int HW,
                          int act_in, int dt, void* stream, ChanWgradTail* defer) {
    if (defer) defer->part = nullptr;
    CFD_REQUIRE(g && in && gw && ws, CFD_ERR_INVALID_ARG, "cfd_chan_wgrad: NULL pointer");
    CFD_REQUIRE(B >= 1 && Ci >= 1 && Co >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "cfd_chan_wgrad: bad sizes");
    CFD_REQUIRE(Ci <= 32 && Co <= 32, CFD_ERR_UNSUPPORTED, "cfd_chan_wgrad: Ci=%d Co=%d (max 32) unsupported", Ci, Co);
    StemSrc ss{};
    if (dt == CFD_DT_BF16)
        return launch_wgrad<false, __bf16>(g, (const __bf16*)in, ss, gw, gb, ws, B, Ci, Co, HW, act_in, (hipStream_t)stream, defer);
    return launch_wgrad<false>(g, (const float*)in, ss, gw, gb, ws, B, Ci, Co, HW, act_in, (hipStream_t)stream, defer);
}

extern "C" size_t cfd_fno_stem_bwd_workspace_bytes(const cfd_plan* p, int B, int in_chan, int P, int C) {
    if (!p) return 0;
    return cfd_chan_wgrad_workspace_bytes(B, in_chan + 3 + P, C, p->H * p->W);
}

extern "C" int cfd_fno_stem_bwd(const cfd_plan* p, const float* g, const float* inputs, const float* mask,
                                const float* case_params, float* gw, float* gb, void* ws, int B, int in_chan, int P,
                                int C, void* stream) {
    CFD_REQUIRE(p && g && inputs && gw && gb && ws && (P == 0 || case_params), CFD_ERR_INVALID_ARG, "cfd_fno_stem_bwd: NULL pointer");
    CFD_REQUIRE(in_chan + 3 + P <= 32 && C <= 32, CFD_ERR_UNSUPPORTED, "cfd_fno_stem_bwd: too many features/channels");
    StemSrc ss{mask, case_params, p->d_gx, p->d_gy, in_chan, P, p->W};
    return launch_wgrad<true>(g, inputs, ss, gw, gb, ws, B, in_chan + 3 + P, C, p->H * p->W, 0, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------
// MseLoss sums  (src/models/loss.py:22-37)
// ------------------------------------------------------------------------------------------------------
#define CFD_LOSS_BLOCKS 256

__global__ __launch_bounds__(256) void k_loss_part(const float* __restrict__ p, const float* __restrict__ l, size_t n,
                                                   float* __restrict__ part) {
    __shared__ float s_r[3 * 4];
    // four element pairs in flight per thread and trip (one pair per trip: a chain of exposed memory round trips, 13 us for the
    // 2.2 M predictions of the Auto-DeepONet); fixed order
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float a = 0.f, b = 0.f, c = 0.f;
    for (; i + 3 * stride < n; i += 4 * stride) {
        float pv[4], lv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { pv[k] = p[i + k * stride]; lv[k] = l[i + k * stride]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = pv[k] - lv[k];
            a = fmaf(d, d, a);
            b += fabsf(d);
            c = fmaf(lv[k], lv[k], c);
        }
    }
    for (; i < n; i += stride) {
        const float lv = l[i], d = p[i] - lv;
        a = fmaf(d, d, a);
        b += fabsf(d);
        c = fmaf(lv, lv, c);
    }
    a = cfd_wave_sum(a); b = cfd_wave_sum(b); c = cfd_wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_r[wave] = a; s_r[4 + wave] = b; s_r[8 + wave] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float* s = s_r + 4 * threadIdx.x;
        part[blockIdx.x * 3 + threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
    }
}

// Deterministic final sum of per-block partials: sums = {sum sq err, sum abs err, sum sq label, count}.
__global__ __launch_bounds__(64) void k_loss_final(const float* __restrict__ part, int nblk, float count,
                                                   float* __restrict__ sums) {
    const int lane = threadIdx.x;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = lane; k < nblk; k += 64) { a += part[k * 3]; b += part[k * 3 + 1]; c += part[k * 3 + 2]; }
    a = cfd_wave_sum(a); b = cfd_wave_sum(b); c = cfd_wave_sum(c);
    if (lane == 0) { sums[0] = a; sums[1] = b; sums[2] = c; sums[3] = count; }
}

extern "C" size_t cfd_loss_workspace_bytes(size_t n) { (void)n; return CFD_LOSS_BLOCKS * 3 * sizeof(float); }

extern "C" int cfd_masked_loss_sums(const float* preds, const float* labels, float* sums, void* ws, size_t n,
                                    void* stream) {
    CFD_REQUIRE(preds && labels && sums && ws, CFD_ERR_INVALID_ARG, "cfd_masked_loss_sums: NULL pointer");
    CFD_REQUIRE(n >= 1, CFD_ERR_INVALID_ARG, "cfd_masked_loss_sums: empty tensors (mean of nothing)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_part, dim3(CFD_LOSS_BLOCKS), dim3(256), 0, st, preds, labels, n, (float*)ws);
    CFD_LAUNCH_CHECK("cfd_masked_loss_sums(part)");
    hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(64), 0, st, (const float*)ws, CFD_LOSS_BLOCKS, (float)n, sums);
    CFD_LAUNCH_CHECK("cfd_masked_loss_sums(final)");
    return CFD_OK;
}

__global__ __launch_bounds__(256) void k_loss_sums_bwd(const float* __restrict__ p, const float* __restrict__ l,
                                                       const float* __restrict__ gs, float* __restrict__ gp,
                                                       float* __restrict__ gl, size_t n) {
    const float g0 = gs[0], g1 = gs[1], g2 = gs[2];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float lv = l[i], d = p[i] - lv;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float g = g0 * 2.f * d + g1 * sg;
        if (gp) gp[i] = g;
        if (gl) gl[i] = g2 * 2.f * lv - g;
    }
}

extern "C" int cfd_loss_sums_bwd(const float* preds, const float* labels, const float* gsums, float* gp, float* gl,
                                 size_t n, void* stream) {
    CFD_REQUIRE(preds && labels && gsums, CFD_ERR_INVALID_ARG, "cfd_loss_sums_bwd: NULL pointer");
    if (n == 0 || (!gp && !gl)) return CFD_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_loss_sums_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, preds, labels, gsums,
                       gp, gl, n);
    CFD_LAUNCH_CHECK("cfd_loss_sums_bwd");
    return CFD_OK;
}

__global__ void k_loss_scores(const float* __restrict__ sums, float* __restrict__ scores) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float n = sums[3];
        const float mse = sums[0] / n;              // loss.py:27
        scores[0] = mse;
        scores[1] = sqrtf(mse);                     // loss.py:31
        scores[2] = sums[1] / n;                    // loss.py:28
        scores[3] = mse / (sums[2] / n);            // loss.py:35
    }
}

extern "C" int cfd_loss_scores(const float* sums, float* scores, void* stream) {
    CFD_REQUIRE(sums && scores, CFD_ERR_INVALID_ARG, "cfd_loss_scores: NULL pointer");
    hipLaunchKernelGGL(k_loss_scores, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, scores);
    CFD_LAUNCH_CHECK("cfd_loss_scores");
    return CFD_OK;
}

// d(scores)/d(sums) applied to the upstream gradients of mse / rmse / mae / nmse (each a device scalar or NULL): what autograd
// computes for loss.py:27-35 written on the sums tensor -- the same fp32 operations in the same order (x / y: gx = g / y,
// gy = -g x / (y y); sqrt: g / (2 sqrt)) -- as ONE launch instead of ~15 scalar kernels (select_backward fills, copies, divisions).
// The count sums[3] is data: its gradient is zero.
__global__ void k_loss_scores_bwd(const float* __restrict__ sums, const float* __restrict__ g_mse, const float* __restrict__ g_rmse,
                                  const float* __restrict__ g_mae, const float* __restrict__ g_nmse, float* __restrict__ gsums) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float n = sums[3];
        const float mse = sums[0] / n, den = sums[2] / n;
        float gm = g_mse ? g_mse[0] : 0.f;                       // gradient reaching mse
        if (g_rmse) gm += g_rmse[0] / (2.f * sqrtf(mse));
        float gden = 0.f;
        if (g_nmse) {
            gm += g_nmse[0] / den;
            gden = -g_nmse[0] * mse / (den * den);
        }
        gsums[0] = gm / n;
        gsums[1] = g_mae ? g_mae[0] / n : 0.f;
        gsums[2] = gden / n;
        gsums[3] = 0.f;
    }
}

extern "C" int cfd_loss_scores_bwd(const float* sums, const float* g_mse, const float* g_rmse, const float* g_mae, const float* g_nmse,
                                   float* gsums, void* stream) {
    CFD_REQUIRE(sums && gsums, CFD_ERR_INVALID_ARG, "cfd_loss_scores_bwd: NULL pointer");
    hipLaunchKernelGGL(k_loss_scores_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, g_mse, g_rmse, g_mae, g_nmse, gsums);
    CFD_LAUNCH_CHECK("cfd_loss_scores_bwd");
    return CFD_OK;
}

// ---- MseLoss as ONE autograd node (round 5): sums + scores from two launches, the whole backward pass from one -----------------------
// cfd_mse_loss_fwd = cfd_masked_loss_sums whose final one-workgroup launch also writes the four scores (k_loss_final + k_loss_scores);
// cfd_mse_loss_bwd = cfd_loss_scores_bwd folded into cfd_loss_sums_bwd: every thread forms the three sum gradients from the sums and the
// upstream score gradients (a dozen scalar operations, the same fp32 order as k_loss_scores_bwd) and applies them to its elements.
// Five launches per training step of every autograd model become three (each ~4.7 us at its dispatch floor: 3 % of an Auto-DeepONet step).
__global__ __launch_bounds__(64) void k_loss_final_scores(const float* __restrict__ part, int nblk, float count, float* __restrict__ sums,
                                                          float* __restrict__ scores) {
    const int lane = threadIdx.x;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int k = lane; k < nblk; k += 64) { a += part[k * 3]; b += part[k * 3 + 1]; c += part[k * 3 + 2]; }
    a = cfd_wave_sum(a); b = cfd_wave_sum(b); c = cfd_wave_sum(c);
    if (lane == 0) {
        sums[0] = a; sums[1] = b; sums[2] = c; sums[3] = count;
        const float mse = a / count;           // loss.py:27
        scores[0] = mse;
        scores[1] = sqrtf(mse);                // loss.py:31
        scores[2] = b / count;                 // loss.py:28
        scores[3] = mse / (c / count);         // loss.py:35
    }
}

extern "C" int cfd_mse_loss_fwd(const float* preds, const float* labels, float* sums, float* scores, void* ws, size_t n, void* stream) {
    CFD_REQUIRE(preds && labels && sums && scores && ws, CFD_ERR_INVALID_ARG, "cfd_mse_loss_fwd: NULL pointer");
    CFD_REQUIRE(n >= 1, CFD_ERR_INVALID_ARG, "cfd_mse_loss_fwd: empty tensors (mean of nothing)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_part, dim3(CFD_LOSS_BLOCKS), dim3(256), 0, st, preds, labels, n, (float*)ws);
    CFD_LAUNCH_CHECK("cfd_mse_loss_fwd(part)");
    hipLaunchKernelGGL(k_loss_final_scores, dim3(1), dim3(64), 0, st, (const float*)ws, CFD_LOSS_BLOCKS, (float)n, sums, scores);
    CFD_LAUNCH_CHECK("cfd_mse_loss_fwd(final)");
    return CFD_OK;
}

__global__ __launch_bounds__(256) void k_mse_loss_bwd(const float* __restrict__ p, const float* __restrict__ l, const float* __restrict__ sums,
                                                      const float* __restrict__ g_mse, const float* __restrict__ g_rmse,
                                                      const float* __restrict__ g_mae, const float* __restrict__ g_nmse,
                                                      float* __restrict__ gp, float* __restrict__ gl, size_t n) {
    // d(scores)/d(sums): k_loss_scores_bwd's operations, once per thread
    const float cnt = sums[3];
    const float mse = sums[0] / cnt, den = sums[2] / cnt;
    float gm = g_mse ? g_mse[0] : 0.f;
    if (g_rmse) gm += g_rmse[0] / (2.f * sqrtf(mse));
    float gden = 0.f;
    if (g_nmse) {
        gm += g_nmse[0] / den;
        gden = -g_nmse[0] * mse / (den * den);
    }
    const float g0 = gm / cnt, g1 = g_mae ? g_mae[0] / cnt : 0.f, g2 = gden / cnt;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float lv = l[i], d = p[i] - lv;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const float g = g0 * 2.f * d + g1 * sg;
        if (gp) gp[i] = g;
        if (gl) gl[i] = g2 * 2.f * lv - g;
    }
}

extern "C" int cfd_mse_loss_bwd(const float* preds, const float* labels, const float* sums, const float* g_mse, const float* g_rmse,
                                const float* g_mae, const float* g_nmse, float* gp, float* gl, size_t n, void* stream) {
    CFD_REQUIRE(preds && labels && sums, CFD_ERR_INVALID_ARG, "cfd_mse_loss_bwd: NULL pointer");
    if (n == 0 || (!gp && !gl)) return CFD_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mse_loss_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, preds, labels, sums, g_mse, g_rmse, g_mae,
                       g_nmse, gp, gl, n);
    CFD_LAUNCH_CHECK("cfd_mse_loss_bwd");
    return CFD_OK;
}

// ---- the same with STRIDED label rows (round 6): labels (rows, cols) with row stride ldl >= cols -- a channel slice label[:, 0] of a
// (B, C, H, W) tensor viewed as (B, H W) -- so that the caller need not make it contiguous first (one copy launch and 2 n floats of traffic
// per Auto-DeepONet step).  preds, gp: contiguous (rows, cols).  Same operations per element in the same order as the contiguous kernels
// (the element -> (row, column) split is the only difference); n < 2^31.
__global__ __launch_bounds__(256) void k_loss_part_ld(const float* __restrict__ p, const float* __restrict__ l, unsigned n, CfdDiv dcols,
                                                      unsigned cols, unsigned ldl, float* __restrict__ part) {
    __shared__ float s_r[3 * 4];
    const unsigned stride = gridDim.x * blockDim.x;
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    const auto lab = [&](unsigned e) {
        const unsigned r = cfd_div(e, dcols);
        return l[(size_t)r * ldl + (e - r * cols)];
    };
    float a = 0.f, b = 0.f, c = 0.f;
    for (; (size_t)i + 3 * (size_t)stride < n; i += 4 * stride) {
        float pv[4], lv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { pv[k] = p[i + k * stride]; lv[k] = lab(i + k * stride); }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = pv[k] - lv[k];
            a = fmaf(d, d, a);
            b += fabsf(d);
            c = fmaf(lv[k], lv[k], c);
        }
    }
    for (; i < n; i += stride) {
        const float lv = lab(i), d = p[i] - lv;
        a = fmaf(d, d, a);
        b += fabsf(d);
        c = fmaf(lv, lv, c);
    }
    a = cfd_wave_sum(a); b = cfd_wave_sum(b); c = cfd_wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { s_r[wave] = a; s_r[4 + wave] = b; s_r[8 + wave] = c; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float* s = s_r + 4 * threadIdx.x;
        part[blockIdx.x * 3 + threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
    }
}

extern "C" int cfd_mse_loss_fwd_ld(const float* preds, const float* labels, float* sums, float* scores, void* ws, size_t rows, size_t cols,
                                   size_t ldl, void* stream) {
    CFD_REQUIRE(preds && labels && sums && scores && ws, CFD_ERR_INVALID_ARG, "cfd_mse_loss_fwd_ld: NULL pointer");
    CFD_REQUIRE(rows >= 1 && cols >= 1 && ldl >= cols, CFD_ERR_INVALID_ARG, "cfd_mse_loss_fwd_ld: bad shape (rows %zu, cols %zu, ldl %zu)", rows, cols, ldl);
    const size_t n = rows * cols;
    if (ldl == cols) return cfd_mse_loss_fwd(preds, labels, sums, scores, ws, n, stream);
    CFD_REQUIRE(n < (1ull << 31) && ldl < (1ull << 31), CFD_ERR_UNSUPPORTED, "cfd_mse_loss_fwd_ld: %zu elements (max 2^31 - 1)", n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_loss_part_ld, dim3(CFD_LOSS_BLOCKS), dim3(256), 0, st, preds, labels, (unsigned)n, cfd_div_make((unsigned)cols),
                       (unsigned)cols, (unsigned)ldl, (float*)ws);
    CFD_LAUNCH_CHECK("cfd_mse_loss_fwd_ld(part)");
    hipLaunchKernelGGL(k_loss_final_scores, dim3(1), dim3(64), 0, st, (const float*)ws, CFD_LOSS_BLOCKS, (float)n, sums, scores);
    CFD_LAUNCH_CHECK("cfd_mse_loss_fwd_ld(final)");
    return CFD_OK;
}

__global__ __launch_bounds__(256) void k_mse_loss_bwd_ld(const float* __restrict__ p, const float* __restrict__ l, const float* __restrict__ sums,
                                                         const float* __restrict__ g_mse, const float* __restrict__ g_rmse,
                                                         const float* __restrict__ g_mae, const float* __restrict__ g_nmse,
                                                         float* __restrict__ gp, unsigned n, CfdDiv dcols, unsigned cols, unsigned ldl) {
    const float cnt = sums[3];
    const float mse = sums[0] / cnt, den = sums[2] / cnt;
    float gm = g_mse ? g_mse[0] : 0.f;
    if (g_rmse) gm += g_rmse[0] / (2.f * sqrtf(mse));
    if (g_nmse) gm += g_nmse[0] / den;
    const float g0 = gm / cnt, g1 = g_mae ? g_mae[0] / cnt : 0.f;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned r = cfd_div(i, dcols);
        const float lv = l[(size_t)r * ldl + (i - r * cols)], d = p[i] - lv;
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        gp[i] = g0 * 2.f * d + g1 * sg;
    }
}

extern "C" int cfd_mse_loss_bwd_ld(const float* preds, const float* labels, const float* sums, const float* g_mse, const float* g_rmse,
                                   const float* g_mae, const float* g_nmse, float* gp, size_t rows, size_t cols, size_t ldl, void* stream) {
    CFD_REQUIRE(preds && labels && sums, CFD_ERR_INVALID_ARG, "cfd_mse_loss_bwd_ld: NULL pointer");
    CFD_REQUIRE(cols >= 1 && ldl >= cols, CFD_ERR_INVALID_ARG, "cfd_mse_loss_bwd_ld: bad shape");
    const size_t n = rows * cols;
    if (n == 0 || !gp) return CFD_OK;
    if (ldl == cols) return cfd_mse_loss_bwd(preds, labels, sums, g_mse, g_rmse, g_mae, g_nmse, gp, nullptr, n, stream);
    CFD_REQUIRE(n < (1ull << 31) && ldl < (1ull << 31), CFD_ERR_UNSUPPORTED, "cfd_mse_loss_bwd_ld: %zu elements (max 2^31 - 1)", n);
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_mse_loss_bwd_ld, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, preds, labels, sums, g_mse, g_rmse, g_mae,
                       g_nmse, gp, (unsigned)n, cfd_div_make((unsigned)cols), (unsigned)cols, (unsigned)ldl);
    CFD_LAUNCH_CHECK("cfd_mse_loss_bwd_ld");
    return CFD_OK;
}

// out (rows, ka + kb) = [a (rows, ka; row stride lda) | b (rows, kb; row stride ldb)]: the branch input [u.flatten(), case_params] of the
// Auto-DeepONet family (src/models/auto_deeponet.py:109-116) assembled by ONE launch from the field's channel slice and the case parameters
// (two strided ATen copies before: 6.1 + 4.3 us of a 282-us step)
__global__ __launch_bounds__(256) void k_rows_concat2(const float* __restrict__ a, unsigned lda, unsigned ka, const float* __restrict__ b,
                                                      unsigned ldb, unsigned kb, float* __restrict__ out, unsigned n, CfdDiv dk) {
    const unsigned kt = ka + kb;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const unsigned r = cfd_div(e, dk), c = e - r * kt;
        out[e] = c < ka ? a[(size_t)r * lda + c] : b[(size_t)r * ldb + (c - ka)];
    }
}

extern "C" int cfd_rows_concat2(const float* a, size_t lda, size_t ka, const float* b, size_t ldb, size_t kb, float* out, size_t rows,
                                void* stream) {
    CFD_REQUIRE(a && b && out, CFD_ERR_INVALID_ARG, "cfd_rows_concat2: NULL pointer");
    CFD_REQUIRE(ka >= 1 && kb >= 1 && lda >= ka && ldb >= kb, CFD_ERR_INVALID_ARG, "cfd_rows_concat2: bad shape");
    const size_t n = rows * (ka + kb);
    if (n == 0) return CFD_OK;
    CFD_REQUIRE(n < (1ull << 31) && lda < (1ull << 31) && ldb < (1ull << 31), CFD_ERR_UNSUPPORTED, "cfd_rows_concat2: %zu elements (max 2^31 - 1)", n);
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_rows_concat2, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, (unsigned)lda, (unsigned)ka, b, (unsigned)ldb,
                       (unsigned)kb, out, (unsigned)n, cfd_div_make((unsigned)(ka + kb)));
    CFD_LAUNCH_CHECK("cfd_rows_concat2");
    return CFD_OK;
}

__global__ void k_loss_coef(const float* __restrict__ sums, float* __restrict__ coef, int which, float upstream) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float c0 = 0.f, c1 = 0.f;
        if (which == 0) c0 = upstream / sums[3];        // d mse / d preds = 2 (p-l) / n
        else if (which == 1) c0 = upstream / sums[2];   // d nmse / d preds = 2 (p-l) / sum l^2
        else c1 = upstream / sums[3];                   // d mae / d preds = sign(p-l) / n
        coef[0] = c0;
        coef[1] = c1;
    }
}

// sum (label * mask)^2 and the loss-gradient coefficients BEFORE the prediction exists: d mse / d p = 2 (p - l) / n,
// d nmse / d p = 2 (p - l) / sum l^2, d mae / d p = sign(p - l) / n need nothing else from the forward pass (loss.py:27-35),
// which is what lets the training head form predictions, loss sums and all gradients in one pass (head.hip, FUSE).
// Two launches, fixed summation order: partial sums of 256 workgroups, then one workgroup finishes sums[2], sums[3], coef.
template <int VEC>
__global__ __launch_bounds__(256) void k_label_energy_part(const float* __restrict__ label, const float* __restrict__ mask,
                                                           float* __restrict__ part, unsigned total_units, unsigned HW,
                                                           CfdDiv dU, CfdDiv dCo) {
    // unit = VEC consecutive pixels of one (b, channel) plane; plane = unit / (HW / VEC), mask row = plane / Co
    __shared__ float s_red[4];
    const unsigned U = HW / VEC;
    float acc = 0.f;
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned u0 = blockIdx.x * blockDim.x + threadIdx.x; u0 < total_units; u0 += 4 * stride) {
        float lv[4][VEC], mv[4][VEC];
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // four independent loads per operand in flight
            const unsigned u = u0 + k * stride;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { lv[k][v] = 0.f; mv[k][v] = 1.f; }
            if (u < total_units) {
                const unsigned pl = cfd_div(u, dU), i = (u - pl * U) * VEC;
                const float* l = label + (size_t)pl * HW + i;
                const float* m = mask ? mask + (size_t)cfd_div(pl, dCo) * HW + i : nullptr;
                if constexpr (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(l);
                    lv[k][0] = t.x; lv[k][1] = t.y; lv[k][2] = t.z; lv[k][3] = t.w;
                    if (m) { const float4 w = *reinterpret_cast<const float4*>(m); mv[k][0] = w.x; mv[k][1] = w.y; mv[k][2] = w.z; mv[k][3] = w.w; }
                } else {
                    lv[k][0] = l[0];
                    if (m) mv[k][0] = m[0];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int v = 0; v < VEC; ++v) { const float x = lv[k][v] * mv[k][v]; acc = fmaf(x, x, acc); }
    }
    acc = cfd_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(64) void k_label_energy_coef(const float* __restrict__ part, int nblk, float count,
                                                          float* __restrict__ sums, float* __restrict__ coef, int which,
                                                          float upstream) {
    float a = 0.f;
    for (int k = threadIdx.x; k < nblk; k += 64) a += part[k];
    a = cfd_wave_sum(a);
    if (threadIdx.x == 0) {
        sums[2] = a;
        sums[3] = count;
        float c0 = 0.f, c1 = 0.f;
        if (which == 0) c0 = upstream / count;
        else if (which == 1) c0 = upstream / a;
        else c1 = upstream / count;
        coef[0] = c0;
        coef[1] = c1;
    }
}

extern "C" size_t cfd_label_energy_workspace_bytes(void) { return 256 * sizeof(float); }

extern "C" int cfd_label_energy_coef(const float* label, const float* mask, float* sums, float* coef, void* ws, int B, int Co,
                                     int HW, int which, float upstream, void* stream) {
    CFD_REQUIRE(label && sums && coef && ws, CFD_ERR_INVALID_ARG, "cfd_label_energy_coef: NULL pointer");
    CFD_REQUIRE(B >= 1 && Co >= 1 && HW >= 1, CFD_ERR_INVALID_ARG, "cfd_label_energy_coef: bad sizes");
    CFD_REQUIRE(which >= 0 && which <= 2, CFD_ERR_INVALID_ARG, "cfd_label_energy_coef: which must be 0 (mse), 1 (nmse), 2 (mae)");
    hipStream_t st = (hipStream_t)stream;
    CFD_REQUIRE_I31((long)B * Co * HW, "cfd_label_energy_coef");
    const bool v4 = HW % 4 == 0 && ((uintptr_t)label % 16) == 0 && (!mask || ((uintptr_t)mask % 16) == 0);
    const unsigned vec = v4 ? 4 : 1;
    const unsigned units = (unsigned)((long)B * Co * HW / vec);
    long want = ((long)units + 1023) / 1024;  // four units per thread
    const int blocks = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
    CFD_PROF_W("k_label_energy", st, 4.0 * B * HW * (Co + 1.0), 2.0 * B * Co * HW);
    if (v4)
        hipLaunchKernelGGL((k_label_energy_part<4>), dim3(blocks), dim3(256), 0, st, label, mask, (float*)ws, units, (unsigned)HW,
                           cfd_div_make((unsigned)HW / 4), cfd_div_make((unsigned)Co));
    else
        hipLaunchKernelGGL((k_label_energy_part<1>), dim3(blocks), dim3(256), 0, st, label, mask, (float*)ws, units, (unsigned)HW,
                           cfd_div_make((unsigned)HW), cfd_div_make((unsigned)Co));
    CFD_LAUNCH_CHECK("cfd_label_energy_coef(part)");
    hipLaunchKernelGGL(k_label_energy_coef, dim3(1), dim3(64), 0, st, (const float*)ws, blocks, (float)((double)B * Co * HW), sums,
                       coef, which, upstream);
    CFD_LAUNCH_CHECK("cfd_label_energy_coef");
    return CFD_OK;
}

extern "C" int cfd_loss_coef(const float* sums, float* coef, int which, float upstream, void* stream) {
    CFD_REQUIRE(sums && coef, CFD_ERR_INVALID_ARG, "cfd_loss_coef: NULL pointer");
    CFD_REQUIRE(which >= 0 && which <= 2, CFD_ERR_INVALID_ARG, "cfd_loss_coef: which must be 0 (mse), 1 (nmse), 2 (mae)");
    hipLaunchKernelGGL(k_loss_coef, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, coef, which, upstream);
    CFD_LAUNCH_CHECK("cfd_loss_coef");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// stand-alone GELU (only the stand-alone FnoBlock module needs it)
// ------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void k_gelu(const float* __restrict__ x, const float* __restrict__ gy,
                                              float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = BWD ? gy[i] * cfd_gelu_grad(x[i]) : cfd_gelu(x[i]);
}

// gelu(dropout(x)) and its gradient as ONE pass each (ResidualBlock.forward, src/models/resnet.py:70-77: conv1 -> dropout -> GELU):
// the keep mask is the hash of (seed, index) of k_dropout, regenerated in the backward pass; value for value the two stand-alone
// passes (same roundings), which cost the ResNet step two launches and two trips through HBM per block and direction.
// `step` (optional): the stream's step counter in DEVICE memory; the seed is then mix(seed + *step) & 2^48 - 1 with the splitmix64
// finaliser -- what the host computed per step before (models/resnet.py) -- so that a training step replayed from a captured graph
// draws a new mask every time (the counter is incremented inside the graph).
__device__ __forceinline__ unsigned long long cfd_mix64(unsigned long long v) {
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ULL;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBULL;
    return v ^ (v >> 31);
}
template <bool BWD>
__global__ __launch_bounds__(256) void k_dropout_gelu(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ out, size_t n4, float p, unsigned long long seed,
                                                      const unsigned long long* __restrict__ step) {
    if (step) seed = cfd_mix64(seed + *step) & 0xFFFFFFFFFFFFULL;
    const float scale = 1.0f / (1.0f - p);
    const unsigned thresh = (unsigned)((double)p * 4294967296.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 xv = reinterpret_cast<const float4*>(x)[i];
        float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (BWD) gv = reinterpret_cast<const float4*>(gy)[i];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool keep = cfd_hash32(seed * 0x9E3779B97F4A7C15ULL + (4 * i + j)) >= thresh;
            const float d = keep ? xs[j] * scale : 0.f;
            if constexpr (BWD) {
                const float t = gs[j] * cfd_gelu_grad(d);
                o[j] = keep ? t * scale : 0.f;
            } else {
                o[j] = cfd_gelu(d);
            }
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

static int launch_dropout_gelu(const float* x, const float* gy, float* out, size_t n, float p, unsigned long long seed,
                               const unsigned long long* step, bool bwd,
                               hipStream_t st, const char* what) {
    CFD_REQUIRE(x && out && (!bwd || gy), CFD_ERR_INVALID_ARG, "%s: NULL pointer", what);
    CFD_REQUIRE(p >= 0.f && p < 1.f, CFD_ERR_INVALID_ARG, "%s: p must be in [0, 1)", what);
    CFD_REQUIRE(n % 4 == 0 && !(((size_t)x | (size_t)out | (size_t)gy) & 15), CFD_ERR_UNSUPPORTED,
                "%s: needs a multiple of four elements and 16-byte aligned tensors (use cfd_dropout + cfd_gelu_* otherwise)", what);
    if (n == 0) return CFD_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (bwd) hipLaunchKernelGGL((k_dropout_gelu<true>), dim3((unsigned)blocks), dim3(256), 0, st, x, gy, out, n / 4, p, seed, step);
    else hipLaunchKernelGGL((k_dropout_gelu<false>), dim3((unsigned)blocks), dim3(256), 0, st, x, gy, out, n / 4, p, seed, step);
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}

extern "C" int cfd_dropout_gelu_fwd(const float* x, float* y, size_t n, float p, unsigned long long seed, void* stream) {
    return launch_dropout_gelu(x, nullptr, y, n, p, seed, nullptr, false, (hipStream_t)stream, "cfd_dropout_gelu_fwd");
}

extern "C" int cfd_dropout_gelu_fwd_step(const float* x, float* y, size_t n, float p, unsigned long long base,
                                         const unsigned long long* step, void* stream) {
    CFD_REQUIRE(step, CFD_ERR_INVALID_ARG, "cfd_dropout_gelu_fwd_step: NULL step counter");
    return launch_dropout_gelu(x, nullptr, y, n, p, base, step, false, (hipStream_t)stream, "cfd_dropout_gelu_fwd_step");
}

extern "C" int cfd_dropout_gelu_bwd_step(const float* x, const float* gy, float* gx, size_t n, float p, unsigned long long base,
                                         const unsigned long long* step, void* stream) {
    CFD_REQUIRE(step, CFD_ERR_INVALID_ARG, "cfd_dropout_gelu_bwd_step: NULL step counter");
    return launch_dropout_gelu(x, gy, gx, n, p, base, step, true, (hipStream_t)stream, "cfd_dropout_gelu_bwd_step");
}

extern "C" int cfd_dropout_gelu_bwd(const float* x, const float* gy, float* gx, size_t n, float p, unsigned long long seed,
                                    void* stream) {
    return launch_dropout_gelu(x, gy, gx, n, p, seed, nullptr, true, (hipStream_t)stream, "cfd_dropout_gelu_bwd");
}

extern "C" int cfd_gelu_fwd(const float* x, float* y, size_t n, void* stream) {
    CFD_REQUIRE(x && y, CFD_ERR_INVALID_ARG, "cfd_gelu_fwd: NULL pointer");
    if (n == 0) return CFD_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_gelu<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, y, n);
    CFD_LAUNCH_CHECK("cfd_gelu_fwd");
    return CFD_OK;
}

extern "C" int cfd_gelu_bwd(const float* x, const float* gy, float* gx, size_t n, void* stream) {
    CFD_REQUIRE(x && gy && gx, CFD_ERR_INVALID_ARG, "cfd_gelu_bwd: NULL pointer");
    if (n == 0) return CFD_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((k_gelu<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, gy, gx, n);
    CFD_LAUNCH_CHECK("cfd_gelu_bwd");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Adam  (torch.optim.Adam defaults: amsgrad=False, maximize=False; train_auto.py:213)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_update(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd,
                                            float bc1, float rsqrt_bc2, float gscale) {
    float gi = g * gscale;
    if (wd != 0.f) gi = fmaf(wd, p, gi);
    m = fmaf(1.f - b1, gi - m, m);                 // m.lerp_(g, 1-b1)
    v = fmaf(b2, v, (1.f - b2) * gi * gi);         // v.mul_(b2).addcmul_(g, g, 1-b2)
    const float denom = sqrtf(v) * rsqrt_bc2 + eps;  // sqrt(v)/sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom);
}
// VEC: the four buffers are 16-byte aligned -- a thread owns four consecutive elements per trip (one 16-byte load per buffer: the
// flat buffer of the FNO engine, 926 k elements, is ONE trip of 905 workgroups instead of two trips of 2048); the last n % 4 elements
// go to the first threads of the grid one by one.
template <bool VEC>
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                              float wd, float bc1, float rsqrt_bc2, float gscale) {
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if constexpr (VEC) {
        const size_t n4 = n >> 2;
        for (size_t i = t0; i < n4; i += nt) {
            f32x4 pi = reinterpret_cast<const f32x4*>(p)[i], mi = reinterpret_cast<const f32x4*>(m)[i], vi = reinterpret_cast<const f32x4*>(v)[i];
            const f32x4 gi = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pi[j], mj = mi[j], vj = vi[j];
                adam_update(pj, gi[j], mj, vj, lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
                pi[j] = pj, mi[j] = mj, vi[j] = vj;
            }
            reinterpret_cast<f32x4*>(m)[i] = mi;
            reinterpret_cast<f32x4*>(v)[i] = vi;
            reinterpret_cast<f32x4*>(p)[i] = pi;
        }
        for (size_t i = 4 * n4 + t0; i < n; i += nt) adam_update(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
    } else {
        for (size_t i = t0; i < n; i += nt) adam_update(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
    }
}

// Round 6, the fused training step's optimiser launch (cfd_fno_adam_step): the flat Adam above PLUS the work that used to be launches of
// its own in front of it -- (a) the nMSE normaliser: the head ran with the mse coefficient upstream / n, so every gradient still lacks the
// factor n / sum (label*mask)^2 = sums[3] / sums[2] (both left by the head's reduction; k_label_energy_part + _coef: two launches, 10 us);
// (b) the lifting layer's gradient: workgroup c < job.C finishes row c of fc0 from the sum records of k_block<.., STEMG>, stores it (in
// the same, still unscaled units as every other gradient) and applies Adam to it; the flat part skips those two ranges.
template <bool VEC>
__global__ __launch_bounds__(256) void k_adam_f(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float rsqrt_bc2,
                                                float gscale, const float* __restrict__ sums, const StemAdamJob job) {
    __shared__ float s_r[CFD_STEMG_NA][256];
    if (sums) gscale *= sums[3] / sums[2];
    const int njob = job.part ? job.C : 0;
    const int F = job.in_chan + 3 + job.P;
    if ((int)blockIdx.x < njob) {
        const int c = blockIdx.x;
        stem_combine_channel(job.part, job.nrec, job.spl, job.cp, job.P, job.C, c, s_r);
        if ((int)threadIdx.x <= F) {
            const bool bias = (int)threadIdx.x == F;
            const float gv = bias ? s_r[0][0] : s_r[stem_feature_slot(threadIdx.x, job.in_chan)][0];
            const size_t i = bias ? (size_t)job.b_off + c : (size_t)job.w_off + (size_t)c * F + threadIdx.x;
            g[i] = gv;
            adam_update(p[i], gv, m[i], v[i], lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
        }
        return;
    }
    const size_t jw0 = (size_t)job.w_off, jw1 = jw0 + (size_t)job.C * F, jb0 = (size_t)job.b_off, jb1 = jb0 + job.C;
    auto mine = [&](size_t e) { return !(njob && ((e >= jw0 && e < jw1) || (e >= jb0 && e < jb1))); };
    const size_t t0 = (size_t)(blockIdx.x - njob) * blockDim.x + threadIdx.x, nt = (size_t)(gridDim.x - njob) * blockDim.x;
    if constexpr (VEC) {
        const size_t n4 = n >> 2;
        const size_t jlo = jw0 < jb0 ? jw0 : jb0, jhi = jw1 > jb1 ? jw1 : jb1;
        for (size_t i = t0; i < n4; i += nt) {
            if (njob && 4 * i < jhi && 4 * i + 4 > jlo) {  // a unit that touches the lifting layer's ranges: element by element
                for (size_t e = 4 * i; e < 4 * i + 4; ++e)
                    if (mine(e)) adam_update(p[e], g[e], m[e], v[e], lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
                continue;
            }
            f32x4 pi = reinterpret_cast<const f32x4*>(p)[i], mi = reinterpret_cast<const f32x4*>(m)[i], vi = reinterpret_cast<const f32x4*>(v)[i];
            const f32x4 gi = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pi[j], mj = mi[j], vj = vi[j];
                adam_update(pj, gi[j], mj, vj, lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
                pi[j] = pj, mi[j] = mj, vi[j] = vj;
            }
            reinterpret_cast<f32x4*>(m)[i] = mi;
            reinterpret_cast<f32x4*>(v)[i] = vi;
            reinterpret_cast<f32x4*>(p)[i] = pi;
        }
        for (size_t i = 4 * n4 + t0; i < n; i += nt)
            if (mine(i)) adam_update(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
    } else {
        for (size_t i = t0; i < n; i += nt)
            if (mine(i)) adam_update(p[i], g[i], m[i], v[i], lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
    }
}

int cfd_int_adam_flat_f(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, float grad_scale, const float* sums, const StemAdamJob* job, void* stream) {
    CFD_REQUIRE(param && grad && exp_avg && exp_avg_sq, CFD_ERR_INVALID_ARG, "cfd_fno_adam_step: NULL pointer");
    CFD_REQUIRE(step >= 1, CFD_ERR_INVALID_ARG, "cfd_fno_adam_step: step must be >= 1");
    if (n == 0) return CFD_OK;
    const StemAdamJob jb = job ? *job : StemAdamJob{};
    CFD_REQUIRE(!jb.part || (jb.P <= 8 && jb.in_chan + 3 + jb.P < 256), CFD_ERR_UNSUPPORTED, "cfd_fno_adam_step: lifting layer with %d case parameters", jb.P);
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const bool vec = ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0) && n >= 4;
    size_t blocks = ((vec ? n / 4 : n) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    blocks += jb.part ? jb.C : 0;
    CFD_PROF_W("k_adam", (hipStream_t)stream, 28.0 * n, 12.0 * n);  // read p, g, m, v; write p, m, v
    if (vec)
        hipLaunchKernelGGL(k_adam_f<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                           beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, sums, jb);
    else
        hipLaunchKernelGGL(k_adam_f<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                           beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale, sums, jb);
    CFD_LAUNCH_CHECK("cfd_fno_adam_step");
    return CFD_OK;
}

// The same step for MANY parameter tensors in one launch (the autograd training paths of the U-Net / ResNet / DeepONet families keep
// torch's per-tensor parameters: 136 / 28 / 33 tensors).  torch's fused multi-tensor Adam took 3 x 31 us for the U-Net's 4.4 MB and
// 40 us for the Auto-DeepONet's 2.3 MB -- 7 % of that model's step.  The (p, g, m, v, n) table travels as the kernel argument;
// workgroups [blk0, blk0 + nblk) belong to item i.  Learning rate and step count come from device scalars when given (a captured
// HIP graph replays with the values of the moment), bias corrections are formed in fp32 like torch's capturable path.
#define CFD_ADAM_MAX 80
#define CFD_MULTI_MAX_BLOCKS 2048u  // workgroups per tensor of the multi-tensor launches (beyond: several trips per thread)
struct AdamItem {
    float* p;
    const float* g;
    float* m;
    float* v;
    unsigned n, blk0;
};
struct AdamBatch {
    int n;
    AdamItem it[CFD_ADAM_MAX];
};

__global__ __launch_bounds__(256) void k_adam_multi(const AdamBatch b, const float* __restrict__ lr_dev, float lr,
                                                    const float* __restrict__ step_dev, float step, float b1, float b2, float eps,
                                                    float wd, float gscale) {
    // this workgroup's item: the last one whose first workgroup is <= blockIdx.x.  Bisection -- seven dependent scalar loads of the
    // kernel-argument table at most; the linear walk of rounds 3-4 cost the last items' workgroups up to 79 of them
    int i = 0, hi = b.n;
    while (hi - i > 1) {
        const int mid = (i + hi) >> 1;
        if (blockIdx.x >= b.it[mid].blk0) i = mid;
        else hi = mid;
    }
    const AdamItem& e = b.it[i];
    const unsigned nblk = (i + 1 < b.n ? b.it[i + 1].blk0 : gridDim.x) - e.blk0;
    if (lr_dev) lr = lr_dev[0];
    if (step_dev) step = step_dev[0];
    // the bias corrections once per workgroup (wave 0), not once per wave: two powf are ~300 VALU instructions, and with one element per
    // thread they were most of the kernel's issue time
    __shared__ float s_bc[2];
    if (threadIdx.x < 64) {
        const float c1 = 1.f - powf(b1, step), c2 = 1.f / sqrtf(1.f - powf(b2, step));
        if (threadIdx.x == 0) s_bc[0] = c1, s_bc[1] = c2;
    }
    __syncthreads();
    const float bc1 = s_bc[0], rsqrt_bc2 = s_bc[1];
    // (bit 31 of n: the item's four buffers are 16-byte aligned -- a thread owns four consecutive elements per trip, the last n % 4
    // elements go one by one to the item's first threads)
    const unsigned n = e.n & 0x7fffffffu, t0 = (blockIdx.x - e.blk0) * blockDim.x + threadIdx.x, nt = nblk * blockDim.x;
    unsigned first = 0;
    if (e.n >> 31) {
        const unsigned n4 = n >> 2;
        for (unsigned k = t0; k < n4; k += nt) {
            f32x4 pi = reinterpret_cast<const f32x4*>(e.p)[k], mi = reinterpret_cast<const f32x4*>(e.m)[k], vi = reinterpret_cast<const f32x4*>(e.v)[k];
            const f32x4 gi = reinterpret_cast<const f32x4*>(e.g)[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pj = pi[j], mj = mi[j], vj = vi[j];
                adam_update(pj, gi[j], mj, vj, lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
                pi[j] = pj, mi[j] = mj, vi[j] = vj;
            }
            reinterpret_cast<f32x4*>(e.m)[k] = mi;
            reinterpret_cast<f32x4*>(e.v)[k] = vi;
            reinterpret_cast<f32x4*>(e.p)[k] = pi;
        }
        first = 4 * n4;
    }
    for (unsigned k = first + t0; k < n; k += nt) adam_update(e.p[k], e.g[k], e.m[k], e.v[k], lr, b1, b2, eps, wd, bc1, rsqrt_bc2, gscale);
}

extern "C" int cfd_adam_multi(int n, float* const* param, const float* const* grad, float* const* exp_avg, float* const* exp_avg_sq,
                              const size_t* numel, const float* lr_dev, float lr, const float* step_dev, float step, float beta1,
                              float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
    CFD_REQUIRE(n >= 0, CFD_ERR_INVALID_ARG, "cfd_adam_multi: negative count");
    if (n == 0) return CFD_OK;
    CFD_REQUIRE(param && grad && exp_avg && exp_avg_sq && numel, CFD_ERR_INVALID_ARG, "cfd_adam_multi: NULL table");
    CFD_REQUIRE(step_dev || step >= 1.f, CFD_ERR_INVALID_ARG, "cfd_adam_multi: step must be >= 1");
    for (int base = 0; base < n; base += CFD_ADAM_MAX) {
        AdamBatch b{};
        b.n = n - base < CFD_ADAM_MAX ? n - base : CFD_ADAM_MAX;
        unsigned blocks = 0;
        double total = 0.0;
        for (int i = 0; i < b.n; ++i) {
            const int k = base + i;
            CFD_REQUIRE(param[k] && grad[k] && exp_avg[k] && exp_avg_sq[k], CFD_ERR_INVALID_ARG, "cfd_adam_multi: NULL pointer in item %d", k);
            CFD_REQUIRE(numel[k] < (1ull << 31), CFD_ERR_UNSUPPORTED, "cfd_adam_multi: item %d has 2^31 or more elements", k);
            AdamItem& e = b.it[i];
            e.p = param[k], e.g = grad[k], e.m = exp_avg[k], e.v = exp_avg_sq[k], e.n = (unsigned)numel[k], e.blk0 = blocks;
            // ONE trip per thread up to 2048 workgroups per tensor, four elements in 16-byte units where the buffers allow: a thread that
            // walks several trips pays one memory latency per trip (the Auto-DeepONet's 430 k-element first-layer weight: 7 trips on 256
            // workgroups), and one element per thread takes more workgroups than the chip holds at once (2300 for that model: two
            // rounds of the whole latency chain -- table lookup, step / rate scalars, bias corrections, loads)
            const bool vec = e.n >= 4 && ((((uintptr_t)e.p | (uintptr_t)e.g | (uintptr_t)e.m | (uintptr_t)e.v) & 15) == 0);
            unsigned nb = ((vec ? e.n / 4 : e.n) + 255) / 256;
            if (vec) e.n |= 0x80000000u;
            if (nb < 1) nb = 1;
            if (nb > CFD_MULTI_MAX_BLOCKS) nb = CFD_MULTI_MAX_BLOCKS;
            blocks += nb;
            total += (double)(e.n & 0x7fffffffu);
        }
        CFD_PROF_W("k_adam", (hipStream_t)stream, 28.0 * total, 12.0 * total);
        hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b, lr_dev, lr, step_dev, step, beta1, beta2,
                           eps, weight_decay, grad_scale);
        CFD_LAUNCH_CHECK("cfd_adam_multi");
    }
    return CFD_OK;
}

// dst[k][i] = scale * src[k][i] for n tensors in one launch per 80 (the data-parallel path of the graph-replayed models packs every
// parameter gradient, pre-scaled by 1 / world, into ONE flat buffer behind the captured backward pass: graph.GraphedTrainStep;
// torch.cat + mul_ + ~136 copy_ launches per U-Net step before).  Same table walk as k_adam_multi.
struct CopyItem {
    const float* s;
    float* d;
    unsigned n, blk0;
};
struct CopyBatch {
    int n;
    CopyItem it[CFD_ADAM_MAX];
};
__global__ __launch_bounds__(256) void k_scale_copy_multi(const CopyBatch b, float scale) {
    int i = 0, hi = b.n;  // (bisection, as k_adam_multi)
    while (hi - i > 1) {
        const int mid = (i + hi) >> 1;
        if (blockIdx.x >= b.it[mid].blk0) i = mid;
        else hi = mid;
    }
    const CopyItem& e = b.it[i];
    const unsigned nblk = (i + 1 < b.n ? b.it[i + 1].blk0 : gridDim.x) - e.blk0;
    for (unsigned k = (blockIdx.x - e.blk0) * blockDim.x + threadIdx.x; k < e.n; k += nblk * blockDim.x) e.d[k] = e.s[k] * scale;
}

extern "C" int cfd_scale_copy_multi(int n, const float* const* src, float* const* dst, const size_t* numel, float scale, void* stream) {
    CFD_REQUIRE(n >= 0, CFD_ERR_INVALID_ARG, "cfd_scale_copy_multi: negative count");
    if (n == 0) return CFD_OK;
    CFD_REQUIRE(src && dst && numel, CFD_ERR_INVALID_ARG, "cfd_scale_copy_multi: NULL table");
    for (int base = 0; base < n; base += CFD_ADAM_MAX) {
        CopyBatch b{};
        b.n = n - base < CFD_ADAM_MAX ? n - base : CFD_ADAM_MAX;
        unsigned blocks = 0;
        double total = 0.0;
        for (int i = 0; i < b.n; ++i) {
            const int k = base + i;
            CFD_REQUIRE(src[k] && dst[k], CFD_ERR_INVALID_ARG, "cfd_scale_copy_multi: NULL pointer in item %d", k);
            CFD_REQUIRE(numel[k] < (1ull << 31), CFD_ERR_UNSUPPORTED, "cfd_scale_copy_multi: item %d has 2^31 or more elements", k);
            CopyItem& e = b.it[i];
            e.s = src[k], e.d = dst[k], e.n = (unsigned)numel[k], e.blk0 = blocks;
            unsigned nb = (e.n + 255) / 256;  // (as cfd_adam_multi: one element per thread up to CFD_MULTI_MAX_BLOCKS workgroups per tensor)
            if (nb < 1) nb = 1;
            if (nb > CFD_MULTI_MAX_BLOCKS) nb = CFD_MULTI_MAX_BLOCKS;
            blocks += nb;
            total += (double)e.n;
        }
        CFD_PROF_W("k_scale_copy_multi", (hipStream_t)stream, 8.0 * total, total);
        hipLaunchKernelGGL(k_scale_copy_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b, scale);
        CFD_LAUNCH_CHECK("cfd_scale_copy_multi");
    }
    return CFD_OK;
}

extern "C" int cfd_adam_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                             void* stream) {
    CFD_REQUIRE(param && grad && exp_avg && exp_avg_sq, CFD_ERR_INVALID_ARG, "cfd_adam_flat: NULL pointer");
    CFD_REQUIRE(step >= 1, CFD_ERR_INVALID_ARG, "cfd_adam_flat: step must be >= 1");
    if (n == 0) return CFD_OK;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    const bool vec = ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0) && n >= 4;
    size_t blocks = ((vec ? n / 4 : n) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    CFD_PROF_W("k_adam", (hipStream_t)stream, 28.0 * n, 12.0 * n);  // read p, g, m, v; write p, m, v
    if (vec)
        hipLaunchKernelGGL(k_adam<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                           beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
    else
        hipLaunchKernelGGL(k_adam<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr,
                           beta1, beta2, eps, weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
    CFD_LAUNCH_CHECK("cfd_adam_flat");
    return CFD_OK;
}
