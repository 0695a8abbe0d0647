// SpectralConv2d_fast (src/models/fno/fno2d.py:17-82) as pruned DFTs on the CDNA4 matrix pipe.
//
// Only rows K = [0,m1) U [H-m1,H) and columns [0,m2) of rfft2(x) are ever consumed (fno2d.py:73-78), i.e. 288 of
// 2112 bins at 64x64 / modes 12, so both transforms are dense truncated DFTs on the matrix cores: the 64-wide fast
// paths (k_dft_fwd64_b3, k_idft64, k_block) as 3-term split-bf16 products on v_mfma_f32_16x16x32_bf16 (fp32-class,
// cfd_common.h), the generic-width kernels (k_dft_fwd, k_idft, k_dft_fwd64) with v_mfma_f32_16x16x4_f32 (exact fp32).
// One 64-lane wave owns one (batch, channel) image end to end; the chained-MFMA operand maps below are arranged so that
// the accumulator of the first stage IS the operand of the second (no LDS round trip for data, only the constant
// operator tables live in LDS).
//
// Forward (k_dft_fwd), image X[x][y]:
//   fold rows:  E[xf] = X[xf] + X[H-xf], O[xf] = X[xf] - X[H-xf]            (xf = 0..H/2; cos even / sin odd in x)
//   stage 1:    Tc[kap][y] = sum_xf cos(2pi kap xf/H) E[xf][y],  Ts likewise with sin and O   (kap = 0..m1)
//               row +kap of the x-DFT is Tc - i Ts, row -kap is Tc + i Ts
//   stage 2:    Pc = sum_y cos(2pi l y/W) Tc, Ps = sum_y sin(.) Tc, Qc = sum_y cos(.) Ts, Qs = sum_y sin(.) Ts
//               X^[+kap][l] = (Pc - Qs) - i (Ps + Qc);   X^[-kap][l] = (Pc + Qs) + i (Qc - Ps)
// Inverse (k_idft), modes Z[row][l]:
//   S = Z[+kap] + Z[-kap], D = Z[+kap] - Z[-kap]
//   stage A:    U_re[x][l] = sum_kap S_re cos(th) - D_im sin(th);  U_im[x][l] = sum_kap S_im cos(th) + D_re sin(th)
//   stage B:    y[x][y] = sum_l (c_l/HW) (U_re cos(2pi l y/W) - U_im sin(2pi l y/W))
#include "cfd_common.h"
#include "cfd_tail.h"

#define CFD_WAVES 4  // waves per workgroup (256 threads)
#define CFD_DFT_OS (2 * 15 * 16)  // complex modes of one image, m1 <= 15, m2 <= 16

// ------------------------------------------------------------------------------------------------------
// forward pruned DFT
// ------------------------------------------------------------------------------------------------------
template <int NJ, bool VEC4>
__device__ __forceinline__ void load_cols(const float* __restrict__ row, int n, int W, bool ok, float (&v)[NJ]) {
    if constexpr (VEC4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok && 4 * n < W) t = *reinterpret_cast<const float4*>(row + 4 * n);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int y = NJ * n + j;
            v[j] = (ok && y < W) ? row[y] : 0.f;
        }
    }
}

template <int NJ, bool VEC4, bool ACT>
__global__ __launch_bounds__(64 * CFD_WAVES) void k_dft_fwd(const float* __restrict__ x, float2* __restrict__ xh,
                                                             const float* __restrict__ tabs, int ntab, int nimg,
                                                             int H, int W, int m1, int m2, int KX) {
    __shared__ float s_tab[(2 * 17 + 8 * NJ) * 64];
    __shared__ float2 s_out[CFD_WAVES * CFD_DFT_OS];
    cfd_stage_lds(s_tab, tabs, ntab);
    __syncthreads();
    const float* t1c = s_tab;
    const float* t1s = t1c + KX * 64;
    const float* t2c = t1s + KX * 64;
    const float* t2s = t2c + 4 * NJ * 64;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int M = 2 * m1 * m2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    for (int img = blockIdx.x * CFD_WAVES + wave; img < nimg; img += gridDim.x * CFD_WAVES) {
        const float* xi = x + (size_t)img * H * W;
        f32x4 a1c[NJ], a1s[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { a1c[j] = zero; a1s[j] = zero; }
        for (int s = 0; s < KX; ++s) {
            const int xf = 4 * s + q;
            const bool valid = xf <= H / 2;
            const bool paired = valid && xf != 0 && 2 * xf != H;
            float v[NJ], u[NJ];
            load_cols<NJ, VEC4>(xi + (size_t)xf * W, n, W, valid, v);
            load_cols<NJ, VEC4>(xi + (size_t)(H - xf) * W, n, W, paired, u);
            const float tc = t1c[s * 64 + lane], ts = t1s[s * 64 + lane];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float vv = v[j], uu = u[j];
                if constexpr (ACT) { const cfd_f2 g2 = cfd_gelu2(cfd_f2{vv, uu}); vv = g2.x; uu = g2.y; }  // gelu(0) = 0 keeps the zero fill
                a1c[j] = cfd_mfma16x16x4(vv + uu, tc, a1c[j]);
                a1s[j] = cfd_mfma16x16x4(vv - uu, ts, a1s[j]);
            }
        }
        // a1c[j][r] = Tc[kap = n][y = NJ*(4q+r)+j]  -> contraction over y with the accumulators as B operands
        f32x4 Pc = zero, Ps = zero, Qc = zero, Qs = zero;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ac = t2c[(j * 4 + r) * 64 + lane], as = t2s[(j * 4 + r) * 64 + lane];
                Pc = cfd_mfma16x16x4(ac, a1c[j][r], Pc);
                Ps = cfd_mfma16x16x4(as, a1c[j][r], Ps);
                Qc = cfd_mfma16x16x4(ac, a1s[j][r], Qc);
                Qs = cfd_mfma16x16x4(as, a1s[j][r], Qs);
            }
        }
        // P*[r]: l = 4q + r, kap = n
        // modes -> this wave's LDS slice (scattered 8-byte writes are cheap there), then out in whole 512-byte runs
        {
            float2* so = s_out + wave * CFD_DFT_OS;
            const int kap = n;
            if (kap <= m1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = 4 * q + r;
                    if (l < m2) {
                        if (kap < m1) so[kap * m2 + l] = make_float2(Pc[r] - Qs[r], -(Ps[r] + Qc[r]));
                        if (kap >= 1) so[(2 * m1 - kap) * m2 + l] = make_float2(Pc[r] + Qs[r], Qc[r] - Ps[r]);
                    }
                }
            }
            cfd_wave_lds_sync();
            float2* o = xh + (size_t)img * M;
            for (int i = lane; i < M; i += 64) o[i] = so[i];
            cfd_wave_lds_sync();
        }
    }
}

// Fast path for W == 64 (16-B aligned rows): persistent waves with a rolling prefetch.  A wave keeps a ring
// of D k-steps (2*D float4 per lane) in flight: the moment the rows of k-step s have been folded into MFMA operands
// the same registers are re-armed with k-step s+D (running on into the wave's NEXT image), so the HBM stream never
// stops while the 136 MFMAs of the current image issue (measured: loads alone 14.6 us, MFMAs alone 17 us, unpipelined 31 us).
// Every row load is unconditional (rows past H/2 are clamped to row 0 and meet all-zero table entries).
template <int KXT, int D, bool ACT>
__global__ __launch_bounds__(64 * CFD_WAVES, 3) void k_dft_fwd64(const float* __restrict__ x, float2* __restrict__ xh,
                                                                  const float* __restrict__ tabs, int ntab, int nimg,
                                                                  int H, int m1, int m2) {
    constexpr int W = 64, NJ = 4;
    __shared__ float s_tab[(2 * KXT + 8 * NJ) * 64];
    __shared__ float2 s_out[CFD_WAVES * CFD_DFT_OS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int stride = gridDim.x * CFD_WAVES;
    int img = blockIdx.x * CFD_WAVES + wave;
    static_assert(KXT % D == 0, "ring depth must divide the k-step count");
    float4 v[D], u[D];  // ring of D k-steps in flight: step s lives in slot s % D
    // H == 8*(KXT-1): k-step s holds rows xf = 4s+q (valid while xf <= H/2) and their mirror rows H-xf (only for
    // 0 < xf < H/2).  Row offsets are (compile-time multiple of 4W) + one lane register, so they fold into the
    // load instructions' immediate fields.
    const int lv = q * W + 4 * n;        // row q, this lane's 4 columns
    const int lu = (H - q) * W + 4 * n;  // row H-q
    auto off_v = [&](int s) { return (s < KXT - 1 || q == 0) ? 4 * s * W + lv : 4 * n; };
    auto off_u = [&](int s) { return (s == 0 ? q != 0 : s < KXT - 1) ? lu - 4 * s * W : 4 * n; };
    auto is_paired = [&](int s) { return s == 0 ? q != 0 : s < KXT - 1; };
    {
        const float* xi = x + (size_t)(img < nimg ? img : 0) * H * W;
#pragma unroll
        for (int s = 0; s < D; ++s) {
            v[s] = *reinterpret_cast<const float4*>(xi + off_v(s));
            u[s] = *reinterpret_cast<const float4*>(xi + off_u(s));
        }
    }
    cfd_stage_lds(s_tab, tabs, ntab);
    __syncthreads();
    const float* t1c = s_tab;
    const float* t1s = t1c + KXT * 64;
    const float* t2c = t1s + KXT * 64;
    const float* t2s = t2c + 4 * NJ * 64;
    const int M = 2 * m1 * m2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    while (img < nimg) {
        const int nxt = img + stride;
        // no next image: re-read this one (cache hits, values never used) so the loop body stays branch-free
        const float* xn = x + (size_t)(nxt < nimg ? nxt : img) * H * W;
        f32x4 a1c[NJ], a1s[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { a1c[j] = zero; a1s[j] = zero; }
#pragma unroll
        for (int s = 0; s < KXT; ++s) {
            float e[4], o[4];
            {
                float a[4] = {v[s % D].x, v[s % D].y, v[s % D].z, v[s % D].w};
                float b[4] = {u[s % D].x, u[s % D].y, u[s % D].z, u[s % D].w};
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float aa = a[j], bb = is_paired(s) ? b[j] : 0.f;
                    if constexpr (ACT) { const cfd_f2 g2 = cfd_gelu2(cfd_f2{aa, bb}); aa = g2.x; bb = g2.y; }  // gelu(0) = 0 keeps the zero fill
                    e[j] = aa + bb;
                    o[j] = aa - bb;
                }
            }
            {   // re-arm the slot with k-step s+D: of this image while one remains, else of the wave's next image
                const float* src = s + D < KXT ? x + (size_t)img * H * W : xn;
                const int sn = (s + D) % KXT;
                cfd_sched_fence();
                v[s % D] = *reinterpret_cast<const float4*>(src + off_v(sn));
                u[s % D] = *reinterpret_cast<const float4*>(src + off_u(sn));
                cfd_sched_fence();
            }
            const float tc = t1c[s * 64 + lane], ts = t1s[s * 64 + lane];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                a1c[j] = cfd_mfma16x16x4(e[j], tc, a1c[j]);
                a1s[j] = cfd_mfma16x16x4(o[j], ts, a1s[j]);
            }
        }
        f32x4 Pc = zero, Ps = zero, Qc = zero, Qs = zero;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ac = t2c[(j * 4 + r) * 64 + lane], as = t2s[(j * 4 + r) * 64 + lane];
                Pc = cfd_mfma16x16x4(ac, a1c[j][r], Pc);
                Ps = cfd_mfma16x16x4(as, a1c[j][r], Ps);
                Qc = cfd_mfma16x16x4(ac, a1s[j][r], Qc);
                Qs = cfd_mfma16x16x4(as, a1s[j][r], Qs);
            }
        }
        // modes -> this wave's LDS slice (scattered 8-byte writes are cheap there), then out in whole 512-byte runs
        {
            float2* so = s_out + wave * CFD_DFT_OS;
            const int kap = n;
            if (kap <= m1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = 4 * q + r;
                    if (l < m2) {
                        if (kap < m1) so[kap * m2 + l] = make_float2(Pc[r] - Qs[r], -(Ps[r] + Qc[r]));
                        if (kap >= 1) so[(2 * m1 - kap) * m2 + l] = make_float2(Pc[r] + Qs[r], Qc[r] - Ps[r]);
                    }
                }
            }
            cfd_wave_lds_sync();
            float2* o = xh + (size_t)img * M;
            for (int i = lane; i < M; i += 64) o[i] = so[i];
            cfd_wave_lds_sync();
        }
        img = nxt;
    }
}

// Split-bf16 (3-term, cfd_common.h) form of the 64x64 fast path: the forward transform is MFMA-issue-bound in fp32
// (136 MFMAs x 32 cycles per image against ~14 us of pure streaming for B*C = 5120 images), and both of its
// contractions fit the K = 32 bf16 MFMA: stage 1 over the 32 folded rows xf = 4v + q (the unpaired Nyquist row H/2 rides
// in a second, otherwise empty K block of the cosine sums), stage 2 over the 64 columns in two K = 32 halves.
// 36 + 24 bf16 MFMAs (~17 cycles) per image.  The rows stream through the same rolling ring as in k_dft_fwd64 (a
// slot is re-armed the moment its rows are folded, running on into the wave's next image); the folded values collect
// in registers until the image's eight k-steps are in and the MFMAs issue in one burst.
#define CFD_DFT3_TABV (7 * CFD_TW * 64)  // 16-byte vectors of the split forward tables (three pieces per table)
#ifndef CFD_DFT_RING
#define CFD_DFT_RING 3  // row-quads in flight per wave (1, 3 or 9 = the whole image)
#endif
#ifndef CFD_DFT_OCC
#define CFD_DFT_OCC 3  // workgroups per CU the forward kernel is compiled for
#endif
// STEM (round 6): the LIFTING LAYER fused into the first forward transform (Fno2d.forward, fno2d.py:189-217 + :62).  The wave of image
// (b, c) builds a_0[b, c] = fc0(features)[c] row by row from the entry's u / v / mask planes (shared by the C waves of an entry: L2 hits),
// the coordinate tables and the case parameters -- the same fmaf chain, feature by feature, as k_stem_fwd4, so a_0 is bit for bit the
// stand-alone kernel's -- folds it into the transform AND stores it (the FnoBlock kernel and the backward pass read a_0): one launch and
// one activation-sized read (84 MB at B = 256) less per forward pass.  in_chan == 2, P <= 8; ring depth 1 (three planes per row quad).
struct CfdStemIn {
    const float* inputs;  // (B, 2, 64, 64)
    const float* mask;    // (B, 64, 64) or NULL (= ones)
    const float* cp;      // (B, P)
    const float* w;       // (C, 5 + P)
    const float* bias;    // (C)
    const float* gx;      // [64]  (plan.d_gx)
    const float* gy;      // [64]
    float* a0;            // (B, C, 64, 64): written
    int P, C;
};
template <int D, bool ACT, int AP, bool STEM = false>
__global__ __launch_bounds__(64 * CFD_WAVES, STEM ? 2 : CFD_DFT_OCC) void k_dft_fwd64_b3(const float* __restrict__ x, float2* __restrict__ xh,
                                                                     const bf16x8* __restrict__ tabs3, int nimg, int m1,
                                                                     int m2, const CfdStemIn si) {
    constexpr int H = 64, W = 64, NJ = 4, KXT = 9;
    constexpr int NF = STEM ? 3 : 1;  // planes behind one row quad: the activation itself, or u / v / mask
    static_assert(!(STEM && ACT), "the lifting layer's output enters the first FnoBlock without GELU");
    __shared__ bf16x8 s_tab3[CFD_DFT3_TABV];
    __shared__ float2 s_out[CFD_WAVES * CFD_DFT_OS];
    __shared__ float s_gx[STEM ? H : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int stride = gridDim.x * CFD_WAVES;
    int img = blockIdx.x * CFD_WAVES + wave;
    static_assert(KXT % D == 0, "ring depth must divide the k-step count");
    // Rolling prefetch exactly as k_dft_fwd64: k-step s holds rows xf = 4s+q (valid while xf <= H/2) and their mirror
    // rows H-xf (only for 0 < xf < H/2); the ring slot of step s is re-armed with step s+D (running on into the wave's
    // next image) the moment its rows have been folded.
    float4 v[D][NF], u[D][NF];
    const int lv = q * W + 4 * n;
    const int lu = (H - q) * W + 4 * n;
    auto off_v = [&](int s) { return (s < KXT - 1 || q == 0) ? 4 * s * W + lv : 4 * n; };
    auto off_u = [&](int s) { return (s == 0 ? q != 0 : s < KXT - 1) ? lu - 4 * s * W : 4 * n; };
    auto is_paired = [&](int s) { return s == 0 ? q != 0 : s < KXT - 1; };
    // the NF planes behind element offset `off` of image `im`
    auto ld = [&](int im, int off, float4 (&dst)[NF]) {
        if constexpr (STEM) {
            const int b = cfd_uniform(im) / si.C;
            const float* pin = si.inputs + (size_t)b * 2 * H * W + off;
            dst[0] = *reinterpret_cast<const float4*>(pin);
            dst[1] = *reinterpret_cast<const float4*>(pin + H * W);
            dst[2] = si.mask ? *reinterpret_cast<const float4*>(si.mask + (size_t)b * H * W + off) : make_float4(1.f, 1.f, 1.f, 1.f);
        } else {
            dst[0] = *reinterpret_cast<const float4*>(x + (size_t)im * H * W + off);
        }
    };
    {
        const int im0 = img < nimg ? img : 0;
#pragma unroll
        for (int s = 0; s < D; ++s) {
            ld(im0, off_v(s), v[s]);
            ld(im0, off_u(s), u[s]);
        }
    }
    for (int i = threadIdx.x; i < CFD_DFT3_TABV; i += blockDim.x) s_tab3[i] = tabs3[i];
    float gyl[4] = {0.f, 0.f, 0.f, 0.f};  // STEM: grid_y of the lane's four columns
    if constexpr (STEM) {
        if (threadIdx.x < H) s_gx[threadIdx.x] = si.gx[threadIdx.x];
#pragma unroll
        for (int j = 0; j < 4; ++j) gyl[j] = si.gy[4 * n + j];
    }
    __syncthreads();
    const int M = 2 * m1 * m2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    while (img < nimg) {
        const int nxt = img + stride;
        const int imn = nxt < nimg ? nxt : img;
        // STEM: this image's row of the lifting layer -- fc0.weight[c][0 .. 4 + P], fc0.bias[c], the entry's case parameters (wave-uniform)
        float sw[5 + 8], sb = 0.f, scp[8];
        if constexpr (STEM) {
            const int iu = cfd_uniform(img), bb = iu / si.C, c = iu - bb * si.C, F = 5 + si.P;
            sb = si.bias[c];
#pragma unroll
            for (int f = 0; f < 5 + 8; ++f) sw[f] = si.w[c * F + (f < F ? f : F - 1)];
#pragma unroll
            for (int k = 0; k < 8; ++k) scp[k] = si.P > 0 ? si.cp[(size_t)bb * si.P + (k < si.P ? k : si.P - 1)] : 0.f;
        }
        // a_0 of four pixels of row `row`: bias, then one fmaf per feature in k_stem_fwd4's order (u, v, mask, grid_x, grid_y, case parameters)
        auto lift = [&](const float4 (&r)[NF], int row, float (&out)[4]) {
            if constexpr (STEM) {
                const float gxv = s_gx[row];
                const float uu[4] = {r[0].x, r[0].y, r[0].z, r[0].w}, vv[4] = {r[1].x, r[1].y, r[1].z, r[1].w},
                            mm[4] = {r[NF - 1].x, r[NF - 1].y, r[NF - 1].z, r[NF - 1].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float acc = sb;
                    acc = fmaf(sw[0], uu[j], acc);
                    acc = fmaf(sw[1], vv[j], acc);
                    acc = fmaf(sw[2], mm[j], acc);
                    acc = fmaf(sw[3], gxv, acc);
                    acc = fmaf(sw[4], gyl[j], acc);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < si.P) acc = fmaf(sw[5 + k], scp[k], acc);
                    out[j] = acc;
                }
            } else {
                out[0] = r[0].x; out[1] = r[0].y; out[2] = r[0].z; out[3] = r[0].w;
            }
        };
        // ---- fold the rows into the K = 32 operands: slot v of lane group q is k-step v, i.e. row xf = 4v + q ----
        float e8[NJ][8], o8[NJ][8], ny[NJ];
#pragma unroll
        for (int s = 0; s < KXT; ++s) {
            {
                float a[4], b[4];
                const int rowv = (s < KXT - 1 || q == 0) ? 4 * s + q : 0, rowu = is_paired(s) ? H - q - 4 * s : 0;
                lift(v[s % D], rowv, a);
                lift(u[s % D], rowu, b);
                if constexpr (STEM) {  // every row of the image passes here exactly once: a_0 goes out as whole float4 row quads
                    float* ao = si.a0 + (size_t)img * H * W;
                    if (s < KXT - 1 || q == 0) *reinterpret_cast<float4*>(ao + off_v(s)) = make_float4(a[0], a[1], a[2], a[3]);
                    if (is_paired(s)) *reinterpret_cast<float4*>(ao + off_u(s)) = make_float4(b[0], b[1], b[2], b[3]);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    float aa = a[j], bb = is_paired(s) ? b[j] : 0.f;
                    if constexpr (ACT) { const cfd_f2 g2 = cfd_gelu2(cfd_f2{aa, bb}); aa = cfd_opaque_f(g2.x); bb = cfd_opaque_f(g2.y); }  // gelu(0) = 0; opaque: see k_dft_fwd_g
                    if (s < KXT - 1) { e8[j][s < 8 ? s : 0] = aa + bb; o8[j][s < 8 ? s : 0] = aa - bb; }
                    else ny[j] = q == 0 ? aa : 0.f;  // k-step 8: only row H/2 (lanes q == 0), cosine sums only
                }
            }
            {   // re-arm the slot with k-step s+D: of this image while one remains, else of the wave's next image
                const int ims = s + D < KXT ? img : imn;
                const int sn = (s + D) % KXT;
                cfd_sched_fence();
                ld(ims, off_v(sn), v[s % D]);
                ld(ims, off_u(sn), u[s % D]);
                cfd_sched_fence();
            }
        }
        // ---- stage 1: a1c[j] / a1s[j] = cosine / sine sums over the rows, for the lane's column 4n + j ----
        f32x4 a1c[NJ], a1s[NJ];
        {
            const int lo = cfd_opaque(lane);
            constexpr int TS = CFD_TW * 64;  // vectors per table
            const CfdTab3 tc = cfd_tab3(s_tab3, lo), ts = cfd_tab3(s_tab3 + TS, lo);
            const bf16x8 tnh = s_tab3[2 * TS + lo];  // the Nyquist row's +-1 are exact in one piece
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const CfdAct8<AP> es = cfd_act_split8<AP>(e8[j]), os = cfd_act_split8<AP>(o8[j]), ns = cfd_act_split1<AP>(ny[j]);
                f32x4 c = zero, s = zero;
#pragma unroll
                for (int k = 0; k < cfd_nterm(AP); ++k) {  // term-major: consecutive MFMAs hit different accumulators
                    c = cfd_term<true, AP>(es, tc, k, c);
                    s = cfd_term<true, AP>(os, ts, k, s);
                    if (k == cfd_nterm(AP) - 2) {  // the Nyquist row: every activation piece against the exact +-1
#pragma unroll
                        for (int pc = AP - 1; pc >= 0; --pc) c = cfd_mfma16x16x32_bf16(ns.p[pc], tnh, c);
                    }
                }
                a1c[j] = c;
                a1s[j] = s;
            }
        }
        // ---- stage 2: sums over the columns; k-slot (q, v = 4jj + r) of half h is column 4(4q+r) + 2h + jj ----
        f32x4 Pc = zero, Ps = zero, Qc = zero, Qs = zero;
        {
            const int lo = cfd_opaque(lane);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float cv[8] = {a1c[2 * h][0], a1c[2 * h][1], a1c[2 * h][2], a1c[2 * h][3],
                                     a1c[2 * h + 1][0], a1c[2 * h + 1][1], a1c[2 * h + 1][2], a1c[2 * h + 1][3]};
                const float sv[8] = {a1s[2 * h][0], a1s[2 * h][1], a1s[2 * h][2], a1s[2 * h][3],
                                     a1s[2 * h + 1][0], a1s[2 * h + 1][1], a1s[2 * h + 1][2], a1s[2 * h + 1][3]};
                const CfdAct8<AP> cs = cfd_act_split8<AP>(cv), ss = cfd_act_split8<AP>(sv);
                const CfdTab3 ct = cfd_tab3(s_tab3 + (CFD_TW * (3 + h)) * 64, lo), st = cfd_tab3(s_tab3 + (CFD_TW * (5 + h)) * 64, lo);
#pragma unroll
                for (int k = 0; k < cfd_nterm(AP); ++k) {
                    Pc = cfd_term<false, AP>(cs, ct, k, Pc);
                    Ps = cfd_term<false, AP>(cs, st, k, Ps);
                    Qc = cfd_term<false, AP>(ss, ct, k, Qc);
                    Qs = cfd_term<false, AP>(ss, st, k, Qs);
                }
            }
        }
        // modes -> this wave's LDS slice (scattered 8-byte writes are cheap there), then out in whole 512-byte runs
        {
            float2* so = s_out + wave * CFD_DFT_OS;
            const int kap = n;
            if (kap <= m1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = 4 * q + r;
                    if (l < m2) {
                        if (kap < m1) so[kap * m2 + l] = make_float2(Pc[r] - Qs[r], -(Ps[r] + Qc[r]));
                        if (kap >= 1) so[(2 * m1 - kap) * m2 + l] = make_float2(Pc[r] + Qs[r], Qc[r] - Ps[r]);
                    }
                }
            }
            cfd_wave_lds_sync();
            float2* o = xh + (size_t)img * M;
            for (int i = lane; i < M; i += 64) o[i] = so[i];
            cfd_wave_lds_sync();
        }
        img = nxt;
    }
}

// General-width split-bf16 forward transform (any W <= 80, H <= 70: the 66 x 65 grids of the tube / dam problems).
// Columns are dealt to lanes as y = 16 j + n (tile j, lane n), so every access of a wave is four runs of 16 consecutive
// elements (one per lane group q = one row each) whatever W is -- the generic fp32 kernel above deals y = NJ n + j (a
// stride-NJ gather) and pays 170 fp32 MFMAs (32 cycles each) per image.  Same contraction scheme as k_dft_fwd64_b3:
//   stage 1  a1c/a1s[j] = sum over folded rows xf, K block 0: xf = 4v + q (0 .. 31), K block 1: xf = 32 + q (v = 0 only:
//            rows 32 .. 35 cover H <= 70); data = A operand (lane (q, n): column 16 j + n, rows of its group q), so the
//            accumulator of tile j holds Tc / Ts[kap = n][y = 16 j + 4q + r]
//   stage 2  K = the columns, two tiles per K block (k-slot (q, v = 4jj + r) = column 16 (2h + jj) + 4q + r), accumulators
//            feed the B operand as they are
// One image per wave, persistent waves; the 18 row values of a tile are re-requested for the wave's NEXT image the moment they
// have been folded (a rolling ring one image deep: ~17 KB in flight per wave).  TA = storage type of the activations
// (float, or __bf16 for the bf16-storage inference path).
template <int NJ, bool ACT, typename TA, int AP>
__global__ __launch_bounds__(64 * CFD_WAVES, 2) void k_dft_fwd_g(const TA* __restrict__ x, float2* __restrict__ xh,
                                                                  const bf16x8* __restrict__ tabs3, int ntabv, int nimg, int H,
                                                                  int W, int m1, int m2) {
    constexpr int NH = (NJ + 1) / 2;
    CFD_DYN_SHARED(bf16x8, s_dyn);
    bf16x8* s_tab3 = s_dyn;                                              // [4 + 2 NH tables][hi|lo|lo2][64]
    float2* s_out = reinterpret_cast<float2*>(s_dyn + (4 + 2 * NH) * CFD_TW * 64);  // [CFD_WAVES][CFD_DFT_OS]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int stride = gridDim.x * CFD_WAVES;
    int img = blockIdx.x * CFD_WAVES + wave;
    const int HW = H * W, H2 = H / 2;
    // row offsets of this lane group (elements): block 0 rows xf = 4v + q and their mirrors H - xf, block 1 row 32 + q
    int offv[9], offu[9];
    bool pair[9], valid[9];
#pragma unroll
    for (int v = 0; v < 9; ++v) {
        const int xf = v < 8 ? 4 * v + q : 32 + q;
        valid[v] = xf <= H2;
        pair[v] = valid[v] && xf != 0 && 2 * xf != H;
        offv[v] = valid[v] ? xf * W : 0;
        offu[v] = pair[v] ? (H - xf) * W : 0;
    }
    int colc[NJ];  // this lane's column of tile j, clamped into the row (tables are zero beyond W)
#pragma unroll
    for (int j = 0; j < NJ; ++j) colc[j] = 16 * j + n < W ? 16 * j + n : W - 1;
    float rv[NJ][9], ru[NJ][9];  // the rolling ring: tile j of the image being transformed, then of the next one
    auto arm = [&](const TA* xi, int j) {
#pragma unroll
        for (int v = 0; v < 9; ++v) {
            rv[j][v] = cfd_ld(xi + offv[v] + colc[j]);
            ru[j][v] = cfd_ld(xi + offu[v] + colc[j]);
        }
    };
    {
        const TA* xi = x + (size_t)(img < nimg ? img : 0) * HW;
#pragma unroll
        for (int j = 0; j < NJ; ++j) arm(xi, j);
    }
    cfd_stage_lds(s_tab3, tabs3, ntabv);
    __syncthreads();
    const int M = 2 * m1 * m2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    while (img < nimg) {
        const int nxt = img + stride;
        const TA* xn = x + (size_t)(nxt < nimg ? nxt : img) * HW;  // no next image: re-read this one (values never used)
        f32x4 a1c[NJ], a1s[NJ];
        const int lo = cfd_opaque(lane);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float e8[8], o8[8], e1[8], o1[8];
#pragma unroll
            for (int v = 0; v < 9; ++v) {
                float aa = valid[v] ? rv[j][v] : 0.f, bb = pair[v] ? ru[j][v] : 0.f;
                // (opaque: the fold below must not become one v_pk_add_f32 on the GELU's register pair with op_sel:[0,1] -- build.py's lint)
                if constexpr (ACT) { const cfd_f2 g2 = cfd_gelu2(cfd_f2{aa, bb}); aa = cfd_opaque_f(g2.x); bb = cfd_opaque_f(g2.y); }  // gelu(0) = 0
                if (v < 8) { e8[v] = aa + bb; o8[v] = aa - bb; }
                else { e1[0] = aa + bb; o1[0] = aa - bb; }
            }
#pragma unroll
            for (int v = 1; v < 8; ++v) { e1[v] = 0.f; o1[v] = 0.f; }
            cfd_sched_fence();
            arm(xn, j);  // re-arm the slot with the same tile of the wave's next image
            cfd_sched_fence();
            const CfdAct8<AP> es = cfd_act_split8<AP>(e8), os = cfd_act_split8<AP>(o8), es1 = cfd_act_split8<AP>(e1), os1 = cfd_act_split8<AP>(o1);
            constexpr int TS = CFD_TW * 64;  // vectors per table
            const CfdTab3 c0 = cfd_tab3(s_tab3, lo), s0 = cfd_tab3(s_tab3 + TS, lo), c1 = cfd_tab3(s_tab3 + 2 * TS, lo), s1 = cfd_tab3(s_tab3 + 3 * TS, lo);
            f32x4 c = zero, sn = zero;
            if constexpr (AP == 2) {  // the round-3 term order (row block 1 between the terms of block 0): bit-identical results
                c = cfd_mfma16x16x32_bf16(es.p[1], c0.p[0], c);
                sn = cfd_mfma16x16x32_bf16(os.p[1], s0.p[0], sn);
                c = cfd_mfma16x16x32_bf16(es.p[0], c0.p[2], c);
                sn = cfd_mfma16x16x32_bf16(os.p[0], s0.p[2], sn);
                c = cfd_mfma16x16x32_bf16(es1.p[0], c1.p[2], c);
                sn = cfd_mfma16x16x32_bf16(os1.p[0], s1.p[2], sn);
                c = cfd_mfma16x16x32_bf16(es.p[0], c0.p[1], c);
                sn = cfd_mfma16x16x32_bf16(os.p[0], s0.p[1], sn);
                c = cfd_mfma16x16x32_bf16(es1.p[1], c1.p[0], c);
                sn = cfd_mfma16x16x32_bf16(os1.p[1], s1.p[0], sn);
                c = cfd_mfma16x16x32_bf16(es1.p[0], c1.p[1], c);
                sn = cfd_mfma16x16x32_bf16(os1.p[0], s1.p[1], sn);
                c = cfd_mfma16x16x32_bf16(es1.p[0], c1.p[0], c);
                sn = cfd_mfma16x16x32_bf16(os1.p[0], s1.p[0], sn);
                c = cfd_mfma16x16x32_bf16(es.p[0], c0.p[0], c);
                sn = cfd_mfma16x16x32_bf16(os.p[0], s0.p[0], sn);
            } else {
#pragma unroll
                for (int k = 0; k < cfd_nterm(AP); ++k) {
                    c = cfd_term<true, AP>(es, c0, k, c);
                    sn = cfd_term<true, AP>(os, s0, k, sn);
                    c = cfd_term<true, AP>(es1, c1, k, c);
                    sn = cfd_term<true, AP>(os1, s1, k, sn);
                }
            }
            a1c[j] = c;
            a1s[j] = sn;
        }
        // ---- stage 2: sums over the columns, two tiles per K block ----
        f32x4 Pc = zero, Ps = zero, Qc = zero, Qs = zero;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float cv[8], sv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                cv[r] = a1c[2 * h][r];
                sv[r] = a1s[2 * h][r];
                cv[4 + r] = 2 * h + 1 < NJ ? a1c[2 * h + 1 < NJ ? 2 * h + 1 : 0][r] : 0.f;
                sv[4 + r] = 2 * h + 1 < NJ ? a1s[2 * h + 1 < NJ ? 2 * h + 1 : 0][r] : 0.f;
            }
            const CfdAct8<AP> cs = cfd_act_split8<AP>(cv), ss = cfd_act_split8<AP>(sv);
            const CfdTab3 ct = cfd_tab3(s_tab3 + (CFD_TW * (4 + h)) * 64, lo), st = cfd_tab3(s_tab3 + (CFD_TW * (4 + NH + h)) * 64, lo);
#pragma unroll
            for (int k = 0; k < cfd_nterm(AP); ++k) {
                Pc = cfd_term<false, AP>(cs, ct, k, Pc);
                Ps = cfd_term<false, AP>(cs, st, k, Ps);
                Qc = cfd_term<false, AP>(ss, ct, k, Qc);
                Qs = cfd_term<false, AP>(ss, st, k, Qs);
            }
        }
        // P*[r]: l = 4q + r, kap = n.  Modes -> this wave's LDS slice, then out in whole 512-byte runs
        {
            float2* so = s_out + wave * CFD_DFT_OS;
            const int kap = n;
            if (kap <= m1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int l = 4 * q + r;
                    if (l < m2) {
                        if (kap < m1) so[kap * m2 + l] = make_float2(Pc[r] - Qs[r], -(Ps[r] + Qc[r]));
                        if (kap >= 1) so[(2 * m1 - kap) * m2 + l] = make_float2(Pc[r] + Qs[r], Qc[r] - Ps[r]);
                    }
                }
            }
            cfd_wave_lds_sync();
            float2* o = xh + (size_t)img * M;
            for (int i = lane; i < M; i += 64) o[i] = so[i];
            cfd_wave_lds_sync();
        }
        img = nxt;
    }
}

template <typename TA>
static bool launch_dft_g(const cfd_plan* p, const TA* x, float* xh, int nimg, int act, hipStream_t st) {
    if (!p->d_fwd_g || p->NJG < 1 || p->NJG > 5) return false;
    int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
    if (blocks > 2 * 256) blocks = 2 * 256;  // two resident workgroups per CU; the waves stride over the images
    const size_t lds = (size_t)p->n_fwd_gv * sizeof(bf16x8) + (size_t)CFD_WAVES * CFD_DFT_OS * sizeof(float2);
    // fp32 storage honours the act_pieces knob; the bf16-storage route keeps two pieces (its values have 8 significant bits)
    const bool ap3 = sizeof(TA) == 4 && cfd_act_pieces() == 3;
#define CFD_DFTG_P(NJ_, A_, P_)                                                                                        \
    hipLaunchKernelGGL((k_dft_fwd_g<NJ_, A_, TA, P_>), dim3(blocks), dim3(64 * CFD_WAVES), lds, st, x, (float2*)xh,     \
                       (const bf16x8*)p->d_fwd_g, p->n_fwd_gv, nimg, p->H, p->W, p->m1, p->m2)
#define CFD_DFTG(NJ_, A_)                                                             \
    do {                                                                              \
        if constexpr (sizeof(TA) == 4) { if (ap3) CFD_DFTG_P(NJ_, A_, 3); else CFD_DFTG_P(NJ_, A_, 2); } \
        else CFD_DFTG_P(NJ_, A_, 2);                                                  \
    } while (0)
#define CFD_DFTG_A(NJ_) do { if (act) CFD_DFTG(NJ_, true); else CFD_DFTG(NJ_, false); } while (0)
    switch (p->NJG) {
        case 1: CFD_DFTG_A(1); break;
        case 2: CFD_DFTG_A(2); break;
        case 3: CFD_DFTG_A(3); break;
        case 4: CFD_DFTG_A(4); break;
        default: CFD_DFTG_A(5); break;
    }
#undef CFD_DFTG_A
#undef CFD_DFTG
#undef CFD_DFTG_P
    return true;
}

template <int NJ, bool VEC4>
static int launch_dft(const cfd_plan* p, const float* x, float* xh, int nimg, int act, hipStream_t st) {
    CFD_PROF_W(act ? "k_dft_fwd_act" : "k_dft_fwd", st, (double)nimg * (4.0 * p->H * p->W + 16.0 * p->m1 * p->m2),
               (double)nimg * (4.0 * (p->m1 + 1) * (p->H / 2 + 1) * p->W + 8.0 * (p->m1 + 1) * p->W * p->m2));
    const bool exact = cfd_tune_get(CFD_TUNE_EXACT_FP32) == 1;
    if constexpr (VEC4) {
        if (p->W == 64 && p->H == 64 && p->d_fwd_b3 && !exact) {
            int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
#ifndef CFD_DFT_CAP
#define CFD_DFT_CAP (3 * 256)
#endif
            if (blocks > CFD_DFT_CAP) blocks = CFD_DFT_CAP;  // 3 resident workgroups per CU; the waves stride over the images
#define CFD_DFT64(A_, P_)                                                                                                  \
    hipLaunchKernelGGL((k_dft_fwd64_b3<CFD_DFT_RING, A_, P_>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, x, (float2*)xh, \
                       (const bf16x8*)p->d_fwd_b3, nimg, p->m1, p->m2, CfdStemIn{})
            if (cfd_act_pieces() == 3) { if (act) CFD_DFT64(true, 3); else CFD_DFT64(false, 3); }
            else { if (act) CFD_DFT64(true, 2); else CFD_DFT64(false, 2); }
#undef CFD_DFT64
            CFD_LAUNCH_CHECK("cfd_spectral_dft");
            return CFD_OK;
        }
        if (p->W == 64 && p->KX == 9 && p->H == 64) {
            int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
            // 3 workgroups per CU stay resident and the waves stride over the images (measured best of 1..5; a ring of 3
            // k-steps beats a whole image in flight: 23 vs 26 us at B*C = 5120)
            if (blocks > 3 * 256) blocks = 3 * 256;
            if (act)
                hipLaunchKernelGGL((k_dft_fwd64<9, 3, true>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, x, (float2*)xh,
                                   (const float*)p->d_fwd, p->n_fwd, nimg, p->H, p->m1, p->m2);
            else
                hipLaunchKernelGGL((k_dft_fwd64<9, 3, false>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, x, (float2*)xh,
                                   (const float*)p->d_fwd, p->n_fwd, nimg, p->H, p->m1, p->m2);
            CFD_LAUNCH_CHECK("cfd_spectral_dft");
            return CFD_OK;
        }
    }
    if (!exact && cfd_tune_get(CFD_TUNE_GENERAL_B3) != 0 && launch_dft_g<float>(p, x, xh, nimg, act, st)) {
        CFD_LAUNCH_CHECK("cfd_spectral_dft(general)");
        return CFD_OK;
    }
    int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
    if (blocks > 2048) blocks = 2048;
    if (act)
        hipLaunchKernelGGL((k_dft_fwd<NJ, VEC4, true>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, x, (float2*)xh,
                           (const float*)p->d_fwd, p->n_fwd, nimg, p->H, p->W, p->m1, p->m2, p->KX);
    else
        hipLaunchKernelGGL((k_dft_fwd<NJ, VEC4, false>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, x, (float2*)xh,
                           (const float*)p->d_fwd, p->n_fwd, nimg, p->H, p->W, p->m1, p->m2, p->KX);
    CFD_LAUNCH_CHECK("cfd_spectral_dft");
    return CFD_OK;
}

// The lifting layer fused into the first forward transform (k_dft_fwd64_b3<.., STEM>): whether it applies, and the launch.  a0 = the
// lifting layer's output (B, C, 64, 64), written; xh = its kept modes.  "stem_dft" = 0 keeps the two launches (tests, A/B).
bool cfd_int_dft_stem_ok(const cfd_plan* p, int B, int in_chan, int P, int C, const void* inputs, const void* mask, const void* a0) {
    // Measured (round 6): the fused launch takes 44 us at B = 256 against 25.5 + 22.4 for the two it replaces (ring depth 1 at two
    // workgroups per CU: nine dependent L2 round trips per image) -- step 1.227 against 1.224 ms, no gain -- but a 64-case rollout step
    // goes from 196.4 to 191.6 us.  Default: below 128 entries; "stem_dft" = 1 everywhere, 0 never.
    const int knob = cfd_tune_get(CFD_TUNE_STEM_DFT);
    if (knob == 0 || B < 1 || (knob != 1 && B >= 128)) return false;
    if (!(p->W == 64 && p->H == 64 && p->NJ == 4 && p->d_fwd_b3) || cfd_tune_get(CFD_TUNE_EXACT_FP32) == 1) return false;
    if (in_chan != 2 || P < 0 || P > 8 || C < 1 || C > 32) return false;
    return (((uintptr_t)inputs | (uintptr_t)mask | (uintptr_t)a0 | (uintptr_t)p->d_gy) % 16) == 0;
}

int cfd_int_spectral_dft_stem(const cfd_plan* p, const float* inputs, const float* mask, const float* cp, const float* w, const float* bias,
                              float* a0, float* xh, int B, int P, int C, void* stream) {
    CFD_REQUIRE(p && inputs && w && bias && a0 && xh && (P == 0 || cp), CFD_ERR_INVALID_ARG, "cfd_fno_stem_dft: NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    const int nimg = B * C;
    CFD_PROF_W("k_dft_fwd_stem", st, (double)B * p->H * p->W * (4.0 * 3 + 4.0 * C) + (double)nimg * 16.0 * p->m1 * p->m2,
               (double)nimg * (4.0 * (p->m1 + 1) * (p->H / 2 + 1) * p->W + 8.0 * (p->m1 + 1) * p->W * p->m2) + 2.0 * B * p->H * p->W * (double)C * (5 + P));
    int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
    if (blocks > 2 * 256) blocks = 2 * 256;  // two resident workgroups per CU (the lifting layer's row weights cost ~40 registers)
    const CfdStemIn si{inputs, mask, cp, w, bias, (const float*)p->d_gx, (const float*)p->d_gy, a0, P, C};
    if (cfd_act_pieces() == 3)
        hipLaunchKernelGGL((k_dft_fwd64_b3<1, false, 3, true>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, (const float*)nullptr, (float2*)xh,
                           (const bf16x8*)p->d_fwd_b3, nimg, p->m1, p->m2, si);
    else
        hipLaunchKernelGGL((k_dft_fwd64_b3<1, false, 2, true>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, (const float*)nullptr, (float2*)xh,
                           (const bf16x8*)p->d_fwd_b3, nimg, p->m1, p->m2, si);
    CFD_LAUNCH_CHECK("cfd_fno_stem_dft");
    return CFD_OK;
}

int cfd_int_spectral_dft(const cfd_plan* p, const void* x, float* xh, int nimg, int act_in, int dt, void* stream) {
    if (dt == CFD_DT_F32) return cfd_spectral_dft(p, (const float*)x, xh, nimg, act_in, stream);
    CFD_REQUIRE(p && x && xh && nimg >= 0, CFD_ERR_INVALID_ARG, "cfd_spectral_dft: NULL pointer or negative count");
    if (nimg == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    CFD_PROF_W(act_in ? "k_dft_fwd_act" : "k_dft_fwd", st, (double)nimg * (2.0 * p->H * p->W + 16.0 * p->m1 * p->m2),
               (double)nimg * (4.0 * (p->m1 + 1) * (p->H / 2 + 1) * p->W + 8.0 * (p->m1 + 1) * p->W * p->m2));
    CFD_REQUIRE(launch_dft_g<__bf16>(p, (const __bf16*)x, xh, nimg, act_in, st), CFD_ERR_UNSUPPORTED,
                "cfd_spectral_dft: bf16 activation storage needs H <= 70 and W <= 80 (grid %dx%d)", p->H, p->W);
    CFD_LAUNCH_CHECK("cfd_spectral_dft(bf16 storage)");
    return CFD_OK;
}

extern "C" int cfd_spectral_dft(const cfd_plan* p, const float* x, float* xh, int nimg, int act_in, void* stream) {
    CFD_REQUIRE(p && x && xh && nimg >= 0, CFD_ERR_INVALID_ARG, "cfd_spectral_dft: NULL pointer or negative count");
    if (nimg == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    if (p->NJ == 4) {
        if (p->W % 4 == 0 && ((uintptr_t)x % 16) == 0) return launch_dft<4, true>(p, x, xh, nimg, act_in, st);
        return launch_dft<4, false>(p, x, xh, nimg, act_in, st);
    }
    if (p->NJ == 5) return launch_dft<5, false>(p, x, xh, nimg, act_in, st);
    cfd_set_error("cfd_spectral_dft: W=%d needs NJ=%d columns per lane (supported: 4,5)", p->W, p->NJ);
    return CFD_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------
// mode mixing  (compl_mul2d, fno2d.py:54-57), its input-adjoint, and the spectral weight gradient
// ------------------------------------------------------------------------------------------------------
// All three are 288 independent tiny complex GEMMs over (batch, channel).  Layout: a lane owns one kept mode (the
// fastest-varying index of every operand, so every global access of a wave is one contiguous 512-B run), a wave
// owns one channel of the NON-contracted side and keeps its CR complex weights (mix) or CR complex accumulators
// (weight gradient) in registers for a whole chunk of the batch.  The contracted-side modes of SB batch entries
// are staged once per workgroup in LDS (double buffered, next stage prefetched into registers during the math).
// Round 6: at 20 channels (at 32: the mix alone) and >= 128 batch entries the contractions run on the fp32 matrix pipe (modes.hip); the kernels below
// remain for the other widths, small batches and as the mode_mfma = 0 route of the tests.
bool cfd_int_modes_mfma_ok(const cfd_plan* p, int B, int Cin, int Cout, const void* a, const void* b, const void* c, bool wgrad);
size_t cfd_int_modes_mfma_chunks(const cfd_plan* p, int B);
int cfd_int_modes_mix(const cfd_plan* p, const float* xin, const float* w1, const float* w2, float* z, int B, int C, int conj_t, void* stream);
int cfd_int_modes_mixadj_wgrad(const cfd_plan* p, const float* xh, const float* gh, const float* w1, const float* w2, float* gz, float* part,
                               int B, int C, int* nchunk_out, void* stream);
#define CFD_WGRAD_SB 4       // batch entries per LDS stage of the weight-gradient kernel
#define CFD_WGRAD_BCHUNK 16  // batch entries per workgroup (weight gradient) == partial-sum chunk

// Moves `nsb` batch entries x CR channels x 64 modes between global memory, registers and LDS xs[sb][cr][lane].
template <int CR, int WPB, int SB, bool EXACT>
struct MixStage {
    static constexpr int RPW = (SB * CR + WPB - 1) / WPB;  // rows (sb, cr) each wave moves per stage
    float2 r[RPW];
    // Row (sb, cr) of a stage is row b0*Cr + sb*Cr + cr of the (B*Cr, M) matrix xin: consecutive rows, no division.
    // Loads are unconditional: rows past the end of the batch chunk are clamped to its last row (never committed or
    // never stored), `modec` is the lane's mode clamped to M-1 (lanes past M never store).
    __device__ __forceinline__ void fetch(const float2* __restrict__ xin, int bend, int Cr, int M, int b0, int modec,
                                          int wave) {
        const int nsb = bend - b0 < SB ? bend - b0 : SB;
        const int last = nsb * Cr - 1;
        const unsigned base = (unsigned)b0 * Cr;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int row = wave + k * WPB;
            r[k] = xin[(size_t)((base + (row < last ? row : last)) * (unsigned)M + modec)];
        }
    }
    __device__ __forceinline__ void commit(float2* xs, int bend, int Cr, int b0, int lane, int wave) const {
        const int nsb = bend - b0 < SB ? bend - b0 : SB;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const int row = wave + k * WPB;
            if (row < nsb * Cr) {
                if constexpr (EXACT) xs[row * 64 + lane] = r[k];
                else { const int sb = row / Cr; xs[(sb * CR + row - sb * Cr) * 64 + lane] = r[k]; }
            }
        }
    }
};

// One round of workgroups: each takes a mode group x a group of WPB output channels x a contiguous range of BR batch
// entries, keeps its weights in registers and streams the range through LDS in stages of SB entries (double
// buffered; the next stage's global reads are issued before the current stage's FMAs).
#define CFD_MIX_SB 4
template <int CR, int WPB, bool CONJT, bool EXACT>
__global__ __launch_bounds__(64 * WPB) void k_mix(const float2* __restrict__ xin, const float2* __restrict__ w1,
                                                  const float2* __restrict__ w2, float2* __restrict__ z, int B, int BR,
                                                  int Cr_, int Cz, int CoutW, int m1, int m2) {
    constexpr int SB = CFD_MIX_SB;
    const int Cr = EXACT ? CR : Cr_;  // EXACT: the contracted channel count equals CR (index math folds)
    __shared__ float2 s_x[2 * SB * CR * 64];
    const int M = 2 * m1 * m2, half = m1 * m2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mode = blockIdx.x * 64 + lane;
    const bool mvalid = mode < M;
    const int modec = mvalid ? mode : M - 1;
    const int cz = blockIdx.y * WPB + wave;
    const bool czvalid = cz < Cz;
    const int bbeg = blockIdx.z * BR;
    const int bend = bbeg + BR < B ? bbeg + BR : B;
    if (bbeg >= bend) return;
    MixStage<CR, WPB, SB, EXACT> stg;
    stg.fetch(xin, bend, Cr, M, bbeg, modec, wave);
    // this wave's weights: forward reduces over Cin_w (= cr), the adjoint over Cout_w (= cr) with conj(W).
    // Unconditional loads (clamped indices): padding channels meet zeros in LDS, dead lanes / waves never store.
    float2 wreg[CR];
    {
        const float2* w = modec < half ? w1 : w2;
        const unsigned wm = modec < half ? modec : modec - half;
        const unsigned czc = czvalid ? cz : Cz - 1;
#pragma unroll
        for (int cr = 0; cr < CR; ++cr) {
            const unsigned crc = cr < Cr ? cr : Cr - 1;
            float2 v = CONJT ? w[(size_t)((czc * CoutW + crc) * (unsigned)half + wm)]
                             : w[(size_t)((crc * CoutW + czc) * (unsigned)half + wm)];
            if (CONJT) v.y = -v.y;
            wreg[cr] = v;
        }
    }
    if (!EXACT) {  // padding channels must read as zeros
        for (int i = threadIdx.x; i < 2 * SB * CR * 64; i += blockDim.x) s_x[i] = make_float2(0.f, 0.f);
        __syncthreads();
    }
    stg.commit(s_x, bend, Cr, bbeg, lane, wave);
    __syncthreads();
    int buf = 0;
    for (int b0 = bbeg; b0 < bend; b0 += SB) {
        const bool more = b0 + SB < bend;
        if (more) stg.fetch(xin, bend, Cr, M, b0 + SB, modec, wave);
        const float2* xs = s_x + buf * (SB * CR * 64);
#pragma unroll 1
        for (int sb = 0; sb < SB; ++sb) {  // rolled: one batch entry's LDS reads live at a time
            const int base = sb * CR * 64 + lane;
            cfd_f2 acc = {0.f, 0.f};  // same product order as k_mix_lds: results do not depend on which kernel ran
#pragma unroll
            for (int cr = 0; cr < CR; ++cr) {
                const float2 xv = xs[base + cr * 64];
                acc = cfd_cmla(acc, cfd_f2{xv.x, xv.y}, cfd_f2{wreg[cr].x, wreg[cr].y});
            }
            const int b = b0 + sb;
            if (b < bend && czvalid && mvalid) z[((size_t)b * Cz + cz) * M + mode] = make_float2(acc.x, acc.y);
        }
        if (more) stg.commit(s_x + (buf ^ 1) * (SB * CR * 64), bend, Cr, b0 + SB, lane, wave);
        __syncthreads();
        buf ^= 1;
    }
}

template <int CR, int WPB, bool CONJT>
static void launch_mix(const float2* xin, const float2* w1, const float2* w2, float2* z, int B, int Cr, int Cz,
                       int CoutW, int m1, int m2, hipStream_t st) {
    const int M = 2 * m1 * m2;
    const int gx = (M + 63) / 64, gy = (Cz + WPB - 1) / WPB;
    int nz = 256 / (gx * gy);  // about one workgroup per CU
    if (nz < 1) nz = 1;
    int BR = (B + nz - 1) / nz;
    BR = (BR + CFD_MIX_SB - 1) / CFD_MIX_SB * CFD_MIX_SB;
    dim3 grid(gx, gy, (B + BR - 1) / BR);
    if (Cr == CR)
        hipLaunchKernelGGL((k_mix<CR, WPB, CONJT, true>), grid, dim3(64 * WPB), 0, st, xin, w1, w2, z, B, BR, Cr, Cz,
                           CoutW, m1, m2);
    else
        hipLaunchKernelGGL((k_mix<CR, WPB, CONJT, false>), grid, dim3(64 * WPB), 0, st, xin, w1, w2, z, B, BR, Cr, Cz,
                           CoutW, m1, m2);
}

// ---- batch-in-lanes mode mixing, weights in LDS -----------------------------------------------------------------
// lane = (b-lane j = lane >> 3, mode m = lane & 7): one wave-load covers 8 batch entries x 8 modes (eight contiguous
// 64-B runs).  A wave owns 8 batch entries x ALL channels: it loads its CR contracted-side values once (nothing is
// read twice anywhere in the grid), the workgroup parks the Cz x CR weights of its 8 modes in LDS (25.6 KB at C = 20
// instead of the 102 KB of register weights per workgroup of the lane = mode kernel; pairs of contracted channels per
// 16-B word so one ds_read_b128 feeds two complex FMAs; the 8 b-lanes of a mode read the same word: broadcast,
// conflict-free), and every wave makes exactly ONE pass: loads in flight during the weight fill, Cz outputs, stores.
// With 8 x NWV batch entries per workgroup the whole batch is resident at once, so the kernel costs one memory
// latency plus the LDS-fed FMAs instead of one latency per pipeline step.  Workgroup ids are mapped so that the two
// 8-mode groups sharing each 128-B line run on the same XCD (same L2) back to back.
// (Measured alternative, removed: the same lane map with the weights of 2 output channels per wave in registers and
// the contracted side streamed through a register ring -- 14.1-14.7 us against 12.3-12.7 us here and 13.0-14.5 us
// for the lane = mode kernel, rocprofv3 kernel durations at B = 256, C = 20.)
template <int CR, bool CONJT>
__device__ __forceinline__ void mix_lds_body(float4* s_w, const int L, const float2* __restrict__ xin,
                                             const float2* __restrict__ w1, const float2* __restrict__ w2,
                                             float2* __restrict__ z, int B, int BC, int Cz, int CoutW, int M, int half,
                                             int nmg, int npair, int nchunk) {
    static_assert(CR % 2 == 0, "k_mix_lds: contracted channels are stored in pairs");
    // s_w: [cz][cr / 2][m] -> (w[cr].re, w[cr].im, w[cr+1].re, w[cr+1].im)
    const int q = ((L >> 4) << 3) + (L & 7);
    const int chunk = q / npair;
    const int mg = 2 * (q - chunk * npair) + ((L >> 3) & 1);
    if (chunk >= nchunk || mg >= nmg) return;
    const int lane = threadIdx.x & 63, wave = cfd_uniform(threadIdx.x >> 6);
    const int m = lane & 7, j = lane >> 3;
    const int mode = mg * 8 + m;
    const int modec = mode < M ? mode : M - 1;
    const int b = chunk * BC + wave * 8 + j;
    const unsigned bc = (unsigned)(b < B ? b : B - 1);
    cfd_f2 x[CR];
    {
        const float2* xr = xin + (size_t)(bc * (unsigned)CR) * M + modec;
#pragma unroll
        for (int cr = 0; cr < CR; ++cr) {
            const float2 v = xr[(size_t)cr * M];
            x[cr] = cfd_f2{v.x, v.y};
        }
    }
    {   // weight fill: forward reduces over Cin_w (= cr), the adjoint over Cout_w (= cr) with conj(W).  16 loads per
        // thread are issued back to back before the first LDS write (a load-wait-write loop costs one L2 latency each).
        float2* s_w2 = reinterpret_cast<float2*>(s_w);
        const int total = Cz * CR * 8;
        constexpr int F = 16;
        for (int base = threadIdx.x; base < total; base += F * blockDim.x) {
            float2 v[F];
            int dst[F];
#pragma unroll
            for (int k = 0; k < F; ++k) {
                const int idx = base + k * blockDim.x;
                const int idc = idx < total ? idx : total - 1;
                const int mm = idc & 7, pr = idc >> 3;
                const int cz = pr / CR, cr = pr - cz * CR;
                const int md = mg * 8 + mm < M ? mg * 8 + mm : M - 1;
                const float2* w = md < half ? w1 : w2;
                const unsigned wm = md < half ? md : md - half;
                v[k] = CONJT ? w[(size_t)(((unsigned)cz * CoutW + cr) * (unsigned)half + wm)]
                             : w[(size_t)(((unsigned)cr * CoutW + cz) * (unsigned)half + wm)];
                dst[k] = idx < total ? ((cz * (CR / 2) + (cr >> 1)) * 8 + mm) * 2 + (cr & 1) : -1;
            }
            cfd_sched_fence();
#pragma unroll
            for (int k = 0; k < F; ++k)
                if (dst[k] >= 0) s_w2[dst[k]] = make_float2(v[k].x, CONJT ? -v[k].y : v[k].y);
        }
    }
    __syncthreads();
    const bool ok = b < B && b < chunk * BC + BC && mode < M;
    float2* zr = z + (size_t)(bc * (unsigned)Cz) * M + mode;
#pragma unroll 1
    for (int cz = 0; cz < Cz; ++cz) {
        const float4* wr = s_w + (cz * (CR / 2)) * 8 + m;
        cfd_f2 acc = {0.f, 0.f};
#pragma unroll
        for (int c2 = 0; c2 < CR / 2; ++c2) {
            const float4 w = wr[c2 * 8];
            acc = cfd_cmla(acc, x[2 * c2], cfd_f2{w.x, w.y});
            acc = cfd_cmla(acc, x[2 * c2 + 1], cfd_f2{w.z, w.w});
        }
        if (ok) zr[(size_t)cz * M] = make_float2(acc.x, acc.y);
    }
}

template <int CR, bool CONJT>
__global__ __launch_bounds__(512) void k_mix_lds(const float2* __restrict__ xin, const float2* __restrict__ w1,
                                                 const float2* __restrict__ w2, float2* __restrict__ z, int B, int BC,
                                                 int Cz, int CoutW, int M, int half, int nmg, int npair, int nchunk) {
    CFD_DYN_SHARED(float4, s_w);
    mix_lds_body<CR, CONJT>(s_w, blockIdx.x, xin, w1, w2, z, B, BC, Cz, CoutW, M, half, nmg, npair, nchunk);
}

// Waves (= groups of 8 batch entries) per k_mix_lds workgroup: as many as keep >= 128 workgroups in the grid (the
// weight fill is per workgroup); 0 = use the lane = mode kernel.  The mix_nwv knob (tune.cpp) overrides it.
static int cfd_mix_lds_waves(int B, int nmg) {
    {
        const int v = cfd_tune_get(CFD_TUNE_MIX_NWV);
        if (v >= 0 && v <= 8) return v;  // 0: use the lane = mode kernel
    }
    // The weight fill + barrier cost k_mix_lds a fixed ~5 us that the lane = mode kernel does not pay; its per-entry cost
    // is lower.  Measured cross-over (C = 20, profiles/r01i_mode_variants_small_batch.txt): B = 16: 5.6 vs 8.8 us,
    // 64: 6.1 vs 9.0, 128: 9.6 vs 9.6, 256: 13.9 vs 12.3, 512: 26.8 vs 16.8; C = 32, B = 256: 32.2 vs 21.4.
    if (B < 192) return 0;
    for (int nwv = 8; nwv > 1; nwv >>= 1)
        if (nmg * ((B + 8 * nwv - 1) / (8 * nwv)) >= 128) return nwv;
    return 1;
}

template <bool CONJT>
static bool launch_mix_lds(const float2* xin, const float2* w1, const float2* w2, float2* z, int B, int Cr, int Cz,
                           int CoutW, int m1, int m2, hipStream_t st) {
    if (Cr != 20 && Cr != 32) return false;  // the contracted channel count is a template parameter
    const int M = 2 * m1 * m2, nmg = (M + 7) / 8, npair = (nmg + 1) / 2;
    const int nwv = cfd_mix_lds_waves(B, nmg);
    if (nwv < 1) return false;
    const int BC = 8 * nwv, nchunk = (B + BC - 1) / BC;
    const unsigned grid = (unsigned)((npair * nchunk + 7) / 8) * 16u;
    const size_t lds = (size_t)Cz * Cr * 8 * sizeof(float2);
    if (Cr == 20)
        hipLaunchKernelGGL((k_mix_lds<20, CONJT>), dim3(grid), dim3(64 * nwv), lds, st, xin, w1, w2, z, B, BC, Cz, CoutW, M,
                           m1 * m2, nmg, npair, nchunk);
    else
        hipLaunchKernelGGL((k_mix_lds<32, CONJT>), dim3(grid), dim3(64 * nwv), lds, st, xin, w1, w2, z, B, BC, Cz, CoutW, M,
                           m1 * m2, nmg, npair, nchunk);
    return true;
}

// (CR, WPB) instantiations: CR >= contracted channels, WPB waves = non-contracted channels per workgroup.
#define CFD_MIX_DISPATCH(FN, Cr_, Cz_, ...)                                        \
    do {                                                                            \
        if ((Cr_) <= 8) FN<8, 8>(__VA_ARGS__);                                      \
        else if ((Cr_) <= 20 && (Cz_) % 10 == 0) FN<20, 10>(__VA_ARGS__);           \
        else if ((Cr_) <= 16) FN<16, 8>(__VA_ARGS__);                               \
        else if ((Cr_) <= 24) FN<24, 8>(__VA_ARGS__);                               \
        else FN<32, 8>(__VA_ARGS__);                                                \
    } while (0)

template <int CR, int WPB>
static void launch_mix_fwd(const float2* a, const float2* b, const float2* c, float2* d, int B, int Cr, int Cz, int CoutW,
                           int m1, int m2, hipStream_t st) {
    launch_mix<CR, WPB, false>(a, b, c, d, B, Cr, Cz, CoutW, m1, m2, st);
}
template <int CR, int WPB>
static void launch_mix_adj(const float2* a, const float2* b, const float2* c, float2* d, int B, int Cr, int Cz, int CoutW,
                           int m1, int m2, hipStream_t st) {
    launch_mix<CR, WPB, true>(a, b, c, d, B, Cr, Cz, CoutW, m1, m2, st);
}

extern "C" int cfd_spectral_mix(const cfd_plan* p, const float* xh, const float* w1, const float* w2, float* z, int B,
                                int Cin, int Cout, int conj_t, void* stream) {
    CFD_REQUIRE(p && xh && w1 && w2 && z, CFD_ERR_INVALID_ARG, "cfd_spectral_mix: NULL pointer");
    CFD_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1, CFD_ERR_INVALID_ARG, "cfd_spectral_mix: bad sizes");
    CFD_REQUIRE(Cin <= 32 && Cout <= 32, CFD_ERR_UNSUPPORTED, "cfd_spectral_mix: Cin=%d Cout=%d (max 32) unsupported", Cin, Cout);
    if (B == 0) return CFD_OK;
    const int Cr = conj_t ? Cout : Cin, Cz = conj_t ? Cin : Cout;
    hipStream_t st = (hipStream_t)stream;
    CFD_PROF_W(conj_t ? "k_mix_adj" : "k_mix", st, 16.0 * p->m1 * p->m2 * ((double)B * (Cin + Cout) + (double)Cin * Cout),
               16.0 * B * (double)Cin * Cout * p->m1 * p->m2);
    if (cfd_int_modes_mfma_ok(p, B, Cin, Cout, xh, z, w1, false) && ((uintptr_t)w2 % 16) == 0)
        return cfd_int_modes_mix(p, xh, w1, w2, z, B, Cin, conj_t, stream);
    if (conj_t ? launch_mix_lds<true>((const float2*)xh, (const float2*)w1, (const float2*)w2, (float2*)z, B, Cr, Cz, Cout,
                                      p->m1, p->m2, st)
               : launch_mix_lds<false>((const float2*)xh, (const float2*)w1, (const float2*)w2, (float2*)z, B, Cr, Cz, Cout,
                                       p->m1, p->m2, st)) {
        CFD_LAUNCH_CHECK("cfd_spectral_mix(lds)");
        return CFD_OK;
    }
    if (conj_t)
        CFD_MIX_DISPATCH(launch_mix_adj, Cr, Cz, (const float2*)xh, (const float2*)w1, (const float2*)w2, (float2*)z, B, Cr,
                         Cz, Cout, p->m1, p->m2, st);
    else
        CFD_MIX_DISPATCH(launch_mix_fwd, Cr, Cz, (const float2*)xh, (const float2*)w1, (const float2*)w2, (float2*)z, B, Cr,
                         Cz, Cout, p->m1, p->m2, st);
    CFD_LAUNCH_CHECK("cfd_spectral_mix");
    return CFD_OK;
}

// spectral weight gradient: gw[i,o,mode] = sum_b conj(xh[b,i,mode]) * (c_l/HW) * gh[b,o,mode]
// wave = output channel o, accumulators over i in registers; one partial sum per CFD_WGRAD_BCHUNK batch entries,
// reduced in a fixed order by k_spec_wgrad_reduce (deterministic).
template <int CR, int WPB, bool EXACT>
__global__ __launch_bounds__(64 * WPB) void k_spec_wgrad_part(const float2* __restrict__ xh, const float2* __restrict__ gh,
                                                              float2* __restrict__ part, int B, int Cin_, int Cout,
                                                              int M) {
    const int Cin = EXACT ? CR : Cin_;
    __shared__ float2 s_x[2 * CFD_WGRAD_SB * CR * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int mode = blockIdx.x * 64 + lane;
    const bool mvalid = mode < M;
    const int o = blockIdx.y * WPB + wave;
    const bool ovalid = o < Cout;
    const int bbeg = blockIdx.z * CFD_WGRAD_BCHUNK;
    const int bend = bbeg + CFD_WGRAD_BCHUNK < B ? bbeg + CFD_WGRAD_BCHUNK : B;
    for (int i = threadIdx.x; i < 2 * CFD_WGRAD_SB * CR * 64; i += blockDim.x) s_x[i] = make_float2(0.f, 0.f);
    float2 acc[CR];
#pragma unroll
    for (int i = 0; i < CR; ++i) acc[i] = make_float2(0.f, 0.f);
    auto gload = [&](int b0, float2 (&g)[CFD_WGRAD_SB]) {
#pragma unroll
        for (int sb = 0; sb < CFD_WGRAD_SB; ++sb)
            g[sb] = (b0 + sb < bend && ovalid && mvalid) ? gh[((size_t)(b0 + sb) * Cout + o) * M + mode]
                                                         : make_float2(0.f, 0.f);
    };
    MixStage<CR, WPB, CFD_WGRAD_SB, EXACT> stg;
    float2 gcur[CFD_WGRAD_SB], gnxt[CFD_WGRAD_SB];
    const int modec = mvalid ? mode : M - 1;
    stg.fetch(xh, bend, Cin, M, bbeg, modec, wave);
    gload(bbeg, gcur);
    __syncthreads();
    stg.commit(s_x, bend, Cin, bbeg, lane, wave);
    __syncthreads();
    int buf = 0;
    for (int b0 = bbeg; b0 < bend; b0 += CFD_WGRAD_SB) {
        const bool more = b0 + CFD_WGRAD_SB < bend;
        if (more) {
            stg.fetch(xh, bend, Cin, M, b0 + CFD_WGRAD_SB, modec, wave);
            gload(b0 + CFD_WGRAD_SB, gnxt);
        }
        const float2* xs = s_x + buf * (CFD_WGRAD_SB * CR * 64);
#pragma unroll 1
        for (int sb = 0; sb < CFD_WGRAD_SB; ++sb) {  // rolled: one batch entry's LDS reads live at a time
            const float2 gv = gcur[0];
            const int base = sb * CR * 64 + lane;
#pragma unroll
            for (int i = 0; i < CR; ++i) {
                const float2 xv = xs[base + i * 64];
                // conj(x) * g
                acc[i].x = fmaf(xv.x, gv.x, fmaf(xv.y, gv.y, acc[i].x));
                acc[i].y = fmaf(xv.x, gv.y, fmaf(-xv.y, gv.x, acc[i].y));
            }
#pragma unroll
            for (int k = 0; k + 1 < CFD_WGRAD_SB; ++k) gcur[k] = gcur[k + 1];
        }
        if (more) {
            stg.commit(s_x + (buf ^ 1) * (CFD_WGRAD_SB * CR * 64), bend, Cin, b0 + CFD_WGRAD_SB, lane, wave);
#pragma unroll
            for (int sb = 0; sb < CFD_WGRAD_SB; ++sb) gcur[sb] = gnxt[sb];
        }
        __syncthreads();
        buf ^= 1;
    }
    if (ovalid && mvalid) {
        float2* dst = part + (size_t)blockIdx.z * Cin * Cout * M;
#pragma unroll
        for (int i = 0; i < CR; ++i)
            if (i < Cin) dst[((size_t)i * Cout + o) * M + mode] = acc[i];
    }
}

template <int CR, int WPB>
static void launch_spec_wgrad(const float2* xh, const float2* gh, float2* part, int B, int Cin, int Cout, int M,
                              hipStream_t st) {
    dim3 grid((M + 63) / 64, (Cout + WPB - 1) / WPB, (B + CFD_WGRAD_BCHUNK - 1) / CFD_WGRAD_BCHUNK);
    if (Cin == CR) hipLaunchKernelGGL((k_spec_wgrad_part<CR, WPB, true>), grid, dim3(64 * WPB), 0, st, xh, gh, part, B, Cin, Cout, M);
    else hipLaunchKernelGGL((k_spec_wgrad_part<CR, WPB, false>), grid, dim3(64 * WPB), 0, st, xh, gh, part, B, Cin, Cout, M);
}

// ---- register-tiled weight gradient -----------------------------------------------------------------------
// The lane = mode kernel above needs one partial sum per 16 batch entries to fill the chip (its partial-sum round trip
// is larger than its inputs) and reads the x modes once per output-channel group.  Here the BATCH is spread over
// lanes as well: lane = (b-lane j = lane >> 3, mode m = lane & 7), so one wave-load covers 8 batch entries x 8
// modes (eight contiguous 64-B runs), a wave owns an IT x OT tile of (input, output) channels with its IT*OT complex
// accumulators in registers, and the (Cin/IT) x (Cout/OT) waves of a workgroup cover every channel pair of one
// 8-mode group for a contiguous chunk of the batch.  No LDS and no barrier: the waves of a workgroup share their
// operands through the CU's vector L1 and each wave software-pipelines its own loads one step ahead.  The 8 b-lanes
// are summed by recursive halving (xor 32, 16, 8: each step exchanges half of the remaining accumulators), after
// which every lane holds ~IT*OT/8 finished sums and the partial tile leaves in 64-B runs.  Workgroup ids are mapped
// so that the two 8-mode groups sharing each 128-B line run on the same XCD (same L2) back to back.
static int cfd_wgrad_want_wg() {  // workgroups aimed at; the wgrad_wg knob overrides it (the tests shrink it to reach the
                                  // multi-step loops at small batch sizes)
    const int v = cfd_tune_get(CFD_TUNE_WGRAD_WG);
    return v >= 1 ? v : 256;
}
static inline void cfd_wgrad_tile_geometry(int B, int M, int* nmg, int* npair, int* BC, int* nchunk) {
    *nmg = (M + 7) / 8;
    *npair = (*nmg + 1) / 2;
    int want = cfd_wgrad_want_wg() / *nmg;  // about one workgroup per CU
    if (want < 1) want = 1;
    int bc = (B + want - 1) / want;
    bc = (bc + 7) / 8 * 8;
    if (bc < 8) bc = 8;
    *BC = bc;
    *nchunk = (B + bc - 1) / bc;
}

// `tile` = index of this wave's (input, output) channel tile, row-major over (Cin / IT) x (Cout / OT)
template <int IT, int OT, int NS>
__device__ __forceinline__ void wgrad_tile_body(const int L, const int tile, const float2* __restrict__ xh,
                                                const float2* __restrict__ gh, float2* __restrict__ part, int B, int BC,
                                                int Cin, int Cout, int M, int nmg, int npair, int nchunk) {
    constexpr int NA = IT * OT, NP = (NA + 7) / 8 * 8, H1 = NP / 2, H2 = NP / 4, H3 = NP / 8;
    // XCD-aware id map: consecutive workgroup ids round-robin over the 8 XCDs; ids L and L + 8 (same XCD, adjacent
    // in dispatch order) take the two mode groups of one 128-B line pair for the same batch chunk.
    const int q = ((L >> 4) << 3) + (L & 7);
    const int chunk = q / npair;
    const int mg = 2 * (q - chunk * npair) + ((L >> 3) & 1);
    if (chunk >= nchunk || mg >= nmg) return;
    const int lane = threadIdx.x & 63;
    const int m = lane & 7, j = lane >> 3;
    const int nto = Cout / OT;
    const int ti = tile / nto, to = tile - ti * nto;
    const int i0 = ti * IT, o0 = to * OT;
    const int mode = mg * 8 + m;
    const int modec = mode < M ? mode : M - 1;
    const int bbeg = chunk * BC;
    const int bend = bbeg + BC < B ? bbeg + BC : B;
    const float2* xp = xh + ((size_t)i0 * M + modec);
    const float2* gp = gh + ((size_t)o0 * M + modec);
    cfd_f2 acc[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[a] = cfd_f2{0.f, 0.f};
    // one step = 8 batch entries (one per b-lane); entries past the chunk read a clamped row and are masked at use
    auto load = [&](int b0, float2 (&xv)[IT], float2 (&gv)[OT]) {
        const int b = b0 + j;
        const unsigned bc = (unsigned)(b < bend ? b : bend - 1);
        const float2* xr = xp + (size_t)(bc * (unsigned)Cin) * M;
        const float2* gr = gp + (size_t)(bc * (unsigned)Cout) * M;
#pragma unroll
        for (int ii = 0; ii < IT; ++ii) xv[ii] = xr[(size_t)ii * M];
#pragma unroll
        for (int oo = 0; oo < OT; ++oo) gv[oo] = gr[(size_t)oo * M];
    };
    auto fma_step = [&](int b0, const float2 (&xv)[IT], const float2 (&gv)[OT]) {
        const bool ok = b0 + j < bend;
#pragma unroll
        for (int oo = 0; oo < OT; ++oo) {
            const cfd_f2 g = {ok ? gv[oo].x : 0.f, ok ? gv[oo].y : 0.f};
#pragma unroll
            for (int ii = 0; ii < IT; ++ii)  // conj(x) * g
                acc[ii * OT + oo] = cfd_cmla_conj(acc[ii * OT + oo], cfd_f2{xv[ii].x, xv[ii].y}, g);
        }
    };
    // NS register stages, all armed up front (NS x 8 batch entries in flight per wave); a stage is re-armed as soon as
    // it has been consumed.  The re-arm loads are unconditional (rows past the chunk are clamped and masked at use) so
    // the loop body is branch-free and its waits stay partial; the last NS steps run outside it, nothing left to arm.
    float2 xs[NS][IT], gs[NS][OT];
#pragma unroll
    for (int t = 0; t < NS; ++t) load(bbeg + 8 * t, xs[t], gs[t]);
    int b0 = bbeg;
    for (; b0 + 8 * NS < bend; b0 += 8 * NS) {
#pragma unroll
        for (int t = 0; t < NS; ++t) {
            fma_step(b0 + 8 * t, xs[t], gs[t]);
            load(b0 + 8 * (NS + t), xs[t], gs[t]);
            cfd_sched_fence();
        }
    }
#pragma unroll
    for (int t = 0; t < NS; ++t)
        if (b0 + 8 * t < bend) fma_step(b0 + 8 * t, xs[t], gs[t]);
    // recursive halving over the b-lanes: after the three exchanges lane (j, m) holds the finished sums of the
    // accumulators e = (j>>2&1)*H1 + (j>>1&1)*H2 + (j&1)*H3 + k, k < H3.
    float2 v[NP];
#pragma unroll
    for (int a = 0; a < NP; ++a) v[a] = a < NA ? make_float2(acc[a].x, acc[a].y) : make_float2(0.f, 0.f);
    {
        const bool h = (lane & 32) != 0;
#pragma unroll
        for (int k = 0; k < H1; ++k) {
            const float2 keep = h ? v[k + H1] : v[k], send = h ? v[k] : v[k + H1];
            v[k] = make_float2(keep.x + cfd_shfl_xor(send.x, 32), keep.y + cfd_shfl_xor(send.y, 32));
        }
    }
    {
        const bool h = (lane & 16) != 0;
#pragma unroll
        for (int k = 0; k < H2; ++k) {
            const float2 keep = h ? v[k + H2] : v[k], send = h ? v[k] : v[k + H2];
            v[k] = make_float2(keep.x + cfd_shfl_xor(send.x, 16), keep.y + cfd_shfl_xor(send.y, 16));
        }
    }
    {
        const bool h = (lane & 8) != 0;
#pragma unroll
        for (int k = 0; k < H3; ++k) {
            const float2 keep = h ? v[k + H3] : v[k], send = h ? v[k] : v[k + H3];
            v[k] = make_float2(keep.x + cfd_shfl_xor(send.x, 8), keep.y + cfd_shfl_xor(send.y, 8));
        }
    }
    const int ebase = ((lane >> 5) & 1) * H1 + ((lane >> 4) & 1) * H2 + ((lane >> 3) & 1) * H3;
    float2* dst = part + (size_t)chunk * Cin * Cout * M + mode;
#pragma unroll
    for (int k = 0; k < H3; ++k) {
        const int e = ebase + k;
        const int ii = e / OT, oo = e - ii * OT;
        if (e < NA && mode < M) dst[((size_t)(i0 + ii) * Cout + (o0 + oo)) * M] = v[k];
    }
}

template <int IT, int OT, int NW, int NS>
__global__ __launch_bounds__(64 * NW) void k_spec_wgrad_tile(const float2* __restrict__ xh, const float2* __restrict__ gh,
                                                             float2* __restrict__ part, int B, int BC, int Cin, int Cout,
                                                             int M, int nmg, int npair, int nchunk) {
    wgrad_tile_body<IT, OT, NS>(blockIdx.x, cfd_uniform(threadIdx.x >> 6), xh, gh, part, B, BC, Cin, Cout, M, nmg, npair,
                                nchunk);
}

// ---- adjoint mix + weight gradient in ONE launch (horizontal fusion) ---------------------------------------------
// Both consume the gradient modes gh and neither depends on the other; on this part every kernel costs ~4.5 us of
// dispatch floor whatever it does, and both are latency-bound (one memory latency, little arithmetic).  The first
// n_mix workgroups run the k_mix_lds body (adjoint), the rest the k_spec_wgrad_tile body; NSPLIT workgroups of 8
// waves share the channel tiles of one (mode group, batch chunk).
template <int C, int IT, int OT, int NS, int NSPLIT>
__global__ __launch_bounds__(512) void k_mixadj_wgrad(const float2* __restrict__ xh, const float2* __restrict__ gh,
                                                      const float2* __restrict__ w1, const float2* __restrict__ w2,
                                                      float2* __restrict__ gz, float2* __restrict__ part, int B, int M,
                                                      int half, int nmg, int npair, int n_mix, int mixBC, int mix_nchunk,
                                                      int wgBC, int wg_nchunk) {
    CFD_DYN_SHARED(float4, s_w);
    static_assert((C / IT) * (C / OT) == 8 * NSPLIT, "k_mixadj_wgrad: 8 waves per workgroup");
    if ((int)blockIdx.x < n_mix) {
        mix_lds_body<C, true>(s_w, blockIdx.x, gh, w1, w2, gz, B, mixBC, C, C, M, half, nmg, npair, mix_nchunk);
    } else {
        const int Lw = (int)blockIdx.x - n_mix;
        const int split = Lw % NSPLIT;
        wgrad_tile_body<IT, OT, NS>(Lw / NSPLIT, split * 8 + cfd_uniform(threadIdx.x >> 6), xh, gh, part, B, wgBC, C, C, M,
                                    nmg, npair, wg_nchunk);
    }
}

// the 1x1-conv reduction as a kernel of its own (only where a deferred one cannot ride, see cfd_int_fno_block_bwd_input)
__global__ __launch_bounds__(256) void k_chan_reduce_standalone(const ChanWgradTail t) {
    __shared__ float s_scr[64];
    chan_wgrad_reduce_group(blockIdx.x, t.part, t.nrow, t.gw, t.gb, t.Co, t.Ci, s_scr);
}

__global__ __launch_bounds__(256) void k_head_reduce_standalone(const HeadTail t) {  // (only where a deferred one cannot ride)
    __shared__ float s_scr[64];
    head_reduce_group(blockIdx.x, t, s_scr);
}

__global__ __launch_bounds__(256) void k_spec_wgrad_reduce(const float2* __restrict__ part, float2* __restrict__ gw1,
                                                           float2* __restrict__ gw2, const float* __restrict__ clhw,
                                                           int nchunk, int CC, int m1, int m2) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)CC * 2 * m1 * m2) return;
    spec_wgrad_reduce_one(gid, part, gw1, gw2, clhw, nchunk, CC, m1, m2);
}

extern "C" size_t cfd_spectral_wgrad_workspace_bytes(const cfd_plan* p, int B, int Cin, int Cout) {
    if (!p || B <= 0) return 0;
    size_t nchunk = (B + CFD_WGRAD_BCHUNK - 1) / CFD_WGRAD_BCHUNK;
    int nmg, npair, BC, nct;
    cfd_wgrad_tile_geometry(B, 2 * p->m1 * p->m2, &nmg, &npair, &BC, &nct);
    if ((size_t)nct > nchunk) nchunk = (size_t)nct;  // either kernel may be dispatched
    const size_t nmm = cfd_int_modes_mfma_chunks(p, B);  // ... or the matrix-pipe kernel of modes.hip
    if (nmm > nchunk) nchunk = nmm;
    return nchunk * (size_t)Cin * Cout * 2 * p->m1 * p->m2 * sizeof(float2);
}

template <int IT, int OT, int NS>
static bool launch_spec_wgrad_tile(const float2* xh, const float2* gh, float2* part, int B, int Cin, int Cout, int M,
                                   int* nchunk_out, hipStream_t st) {
    if (Cin % IT || Cout % OT) return false;
    const int nw = (Cin / IT) * (Cout / OT);
    int nmg, npair, BC, nchunk;
    cfd_wgrad_tile_geometry(B, M, &nmg, &npair, &BC, &nchunk);
    const unsigned grid = (unsigned)((npair * nchunk + 7) / 8) * 16u;
    if (nw <= 8) {
        if (nw <= 4) hipLaunchKernelGGL((k_spec_wgrad_tile<IT, OT, 4, NS>), dim3(grid), dim3(64 * nw), 0, st, xh, gh, part, B, BC, Cin, Cout, M, nmg, npair, nchunk);
        else hipLaunchKernelGGL((k_spec_wgrad_tile<IT, OT, 8, NS>), dim3(grid), dim3(64 * nw), 0, st, xh, gh, part, B, BC, Cin, Cout, M, nmg, npair, nchunk);
    } else if constexpr (IT * OT <= 32) {  // 16 waves: 128 registers per lane
        if (nw > 16) return false;
        hipLaunchKernelGGL((k_spec_wgrad_tile<IT, OT, 16, NS>), dim3(grid), dim3(64 * nw), 0, st, xh, gh, part, B, BC, Cin, Cout, M, nmg, npair, nchunk);
    } else return false;
    *nchunk_out = nchunk;
    return true;
}

extern "C" int cfd_spectral_wgrad(const cfd_plan* p, const float* xh, const float* gh, float* gw1, float* gw2, void* ws,
                                  int B, int Cin, int Cout, void* stream) {
    CFD_REQUIRE(p && xh && gh && gw1 && gw2 && ws, CFD_ERR_INVALID_ARG, "cfd_spectral_wgrad: NULL pointer");
    CFD_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1, CFD_ERR_INVALID_ARG, "cfd_spectral_wgrad: bad sizes");
    CFD_REQUIRE(Cin <= 32 && Cout <= 32, CFD_ERR_UNSUPPORTED, "cfd_spectral_wgrad: Cin=%d Cout=%d (max 32) unsupported", Cin, Cout);
    const int M = 2 * p->m1 * p->m2;
    int nchunk = (B + CFD_WGRAD_BCHUNK - 1) / CFD_WGRAD_BCHUNK;
    const long total = (long)Cin * Cout * M;
    hipStream_t st = (hipStream_t)stream;
    {
        CFD_PROF_W("k_spec_wgrad_part", st, 16.0 * p->m1 * p->m2 * ((double)B * (Cin + Cout) + (double)Cin * Cout),
                   16.0 * B * (double)Cin * Cout * p->m1 * p->m2);
        // register-tiled kernel where the channel counts divide into its wave tiles (the FNO widths 10 / 20 / 40 ...),
        // lane = mode kernel otherwise
        const bool done = launch_spec_wgrad_tile<5, 10, 2>((const float2*)xh, (const float2*)gh, (float2*)ws, B, Cin, Cout, M,
                                                           &nchunk, st);
        if (!done)
            CFD_MIX_DISPATCH(launch_spec_wgrad, Cin, Cout, (const float2*)xh, (const float2*)gh, (float2*)ws, B, Cin, Cout, M, st);
    }
    CFD_LAUNCH_CHECK("cfd_spectral_wgrad(part)");
    CFD_PROF_W("k_spec_wgrad_reduce", st, 0.0, 0.0);
    hipLaunchKernelGGL(k_spec_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float2*)ws, (float2*)gw1, (float2*)gw2, (const float*)p->d_clhw, nchunk, Cin * Cout,
                       p->m1, p->m2);
    CFD_LAUNCH_CHECK("cfd_spectral_wgrad(reduce)");
    return CFD_OK;
}

// Adjoint mode mixing (gz = sum_o conj(W[i,o]) gh[b,o]) and the spectral weight gradient of the same gradient modes.
// One launch for both where the fused kernel applies (Cin == Cout == 20 or 32), the two stand-alone entry points otherwise.
static bool cfd_fused_disabled() {  // fused_variant knob = 0: two launches (tests compare both routes)
    return cfd_tune_get(CFD_TUNE_FUSED_VARIANT) == 0;
}

template <int C, int IT, int OT, int NS, int NSPLIT>
static void launch_mixadj_wgrad(const float2* xh, const float2* gh, const float2* w1, const float2* w2, float2* gz,
                                float2* part, int B, int m1, int m2, int* nchunk_out, hipStream_t st) {
    const int M = 2 * m1 * m2;
    int nmg, npair, wgBC, wg_nchunk;
    cfd_wgrad_tile_geometry(B, M, &nmg, &npair, &wgBC, &wg_nchunk);
    const int mixBC = 64, mix_nchunk = (B + mixBC - 1) / mixBC;  // 8 waves x 8 batch entries per mix workgroup
    const int n_mix = (npair * mix_nchunk + 7) / 8 * 16;
    const int n_wg = (npair * wg_nchunk + 7) / 8 * 16 * NSPLIT;
    hipLaunchKernelGGL((k_mixadj_wgrad<C, IT, OT, NS, NSPLIT>), dim3((unsigned)(n_mix + n_wg)), dim3(512),
                       (size_t)C * C * 8 * sizeof(float2), st, xh, gh, w1, w2, gz, part, B, M, m1 * m2, nmg, npair, n_mix,
                       mixBC, mix_nchunk, wgBC, wg_nchunk);
    *nchunk_out = wg_nchunk;
}

// `defer` (may be NULL): where the fused kernel ran, leave the partial-sum reduction to the caller's next launch
// (defer->part != NULL describes it, cfd_tail.h); otherwise the reduction kernel is launched here and defer->part == NULL.
int cfd_int_spectral_mix_adj_wgrad(const cfd_plan* p, const float* xh, const float* gh, const float* w1, const float* w2,
                              float* gz, float* gw1, float* gw2, void* ws, int B, int Cin, int Cout, void* stream,
                              SpecWgradTail* defer) {
    if (defer) defer->part = nullptr;
    CFD_REQUIRE(p && xh && gh && w1 && w2 && gz && gw1 && gw2 && ws, CFD_ERR_INVALID_ARG, "cfd_spectral_mix_adj_wgrad: NULL pointer");
    CFD_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1, CFD_ERR_INVALID_ARG, "cfd_spectral_mix_adj_wgrad: bad sizes");
    if (cfd_fused_disabled() || Cin != Cout || (Cin != 20 && Cin != 32)) {
        CFD_TRY(cfd_spectral_wgrad(p, xh, gh, gw1, gw2, ws, B, Cin, Cout, stream));
        return cfd_spectral_mix(p, gh, w1, w2, gz, B, Cin, Cout, 1, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    const int M = 2 * p->m1 * p->m2;
    int nchunk = 0;
    {
        // 5 x 5 accumulator tiles, 3-stage ring, two workgroups per (mode group, chunk): 122 registers, so two
        // workgroups per CU (measured: 19.1 us per launch against 20.9 / 22.8 us for 5 x 10 tiles with 2 / 4 stages)
        // width 32 (the reference's default --fno_hidden_dim): 4 x 8 tiles, four workgroups per (mode group, chunk)
        CFD_PROF_W("k_mixadj_wgrad", st, 16.0 * p->m1 * p->m2 * (3.0 * B * Cin + 2.0 * Cin * Cout), 32.0 * B * (double)Cin * Cout * p->m1 * p->m2);
        if (cfd_int_modes_mfma_ok(p, B, Cin, Cout, xh, gh, gz, true) && (((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)ws) % 16) == 0)
            CFD_TRY(cfd_int_modes_mixadj_wgrad(p, xh, gh, w1, w2, gz, (float*)ws, B, Cin, &nchunk, stream));
        else if (Cin == 20)
            launch_mixadj_wgrad<20, 5, 5, 3, 2>((const float2*)xh, (const float2*)gh, (const float2*)w1, (const float2*)w2,
                                                (float2*)gz, (float2*)ws, B, p->m1, p->m2, &nchunk, st);
        else
            launch_mixadj_wgrad<32, 4, 8, 3, 4>((const float2*)xh, (const float2*)gh, (const float2*)w1, (const float2*)w2,
                                                (float2*)gz, (float2*)ws, B, p->m1, p->m2, &nchunk, st);
    }
    CFD_LAUNCH_CHECK("cfd_spectral_mix_adj_wgrad(fused)");
    const long total = (long)Cin * Cout * M;
    if (defer) {
        *defer = SpecWgradTail{(const float2*)ws, (float2*)gw1, (float2*)gw2, (const float*)p->d_clhw, nchunk, Cin * Cout,
                               p->m1, p->m2};
        return CFD_OK;
    }
    CFD_PROF_W("k_spec_wgrad_reduce", st, 0.0, 0.0);
    hipLaunchKernelGGL(k_spec_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float2*)ws,
                       (float2*)gw1, (float2*)gw2, (const float*)p->d_clhw, nchunk, Cin * Cout, p->m1, p->m2);
    CFD_LAUNCH_CHECK("cfd_spectral_mix_adj_wgrad(reduce)");
    return CFD_OK;
}

extern "C" int cfd_spectral_mix_adj_wgrad(const cfd_plan* p, const float* xh, const float* gh, const float* w1,
                                          const float* w2, float* gz, float* gw1, float* gw2, void* ws, int B, int Cin,
                                          int Cout, void* stream) {
    return cfd_int_spectral_mix_adj_wgrad(p, xh, gh, w1, w2, gz, gw1, gw2, ws, B, Cin, Cout, stream, nullptr);
}

// ------------------------------------------------------------------------------------------------------
// inverse pruned DFT (+ fused epilogue)
// ------------------------------------------------------------------------------------------------------
// Kept modes of one image: global (2*M floats, interleaved re/im) -> this wave's LDS slice zs[0 .. 2*M], plus one
// zero word at zs[2*M] that every masked-out gather index points at (so the gather below is branch-free).
__device__ __forceinline__ void idft_stage_z(const float* __restrict__ zi, float* zs, int M2, int lane) {
    for (int i = lane; i < M2; i += 64) zs[i] = zi[i];
    if (lane == 0) zs[M2] = 0.f;
    cfd_wave_lds_sync();
}

// Stage-A data operands (lane = (q, rho)): row rho of M-tile mu is c = 4*(4*mu + (rho&3)) + (rho>>2) so that
// accumulator register r of tile mu is exactly the stage-B operand of k-step 4*mu + r.  k-steps s < 4 are the cos
// block (kappa = 4s+q, value S = Z+ + Z-); s >= 4 the sin block (kappa = 4s+q-15, value -D_im for a real-part row,
// +D_re for an imaginary-part row, D = Z+ - Z-).
__device__ __forceinline__ void idft_gather(const float* zs, int m1, int m2, int SA, int q, int n, float (&va)[2][8]) {
    const int zero_slot = 4 * m1 * m2;
#pragma unroll
    for (int mu = 0; mu < 2; ++mu) {
        const int c = 4 * (4 * mu + (n & 3)) + (n >> 2);
        const bool cvalid = c < 2 * m2;
        const int part = c >= m2 ? 1 : 0;
        const int l = c - part * m2;
        const float sg = part ? 1.f : -1.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int kap = s < 4 ? 4 * s + q : 4 * s + q - 15;
            const bool kvalid = (s < SA) && kap <= m1 && cvalid;
            const int comp = s < 4 ? part : 1 - part;
            const int offP = (kvalid && kap < m1) ? (kap * m2 + l) * 2 + comp : zero_slot;
            const int offM = (kvalid && kap >= 1) ? ((2 * m1 - kap) * m2 + l) * 2 + comp : zero_slot;
            const float vp = zs[offP], vm = zs[offM];
            va[mu][s] = s < 4 ? vp + vm : sg * (vp - vm);
        }
    }
}

// One 16-row x tile: stage A (U' = sum over kept rows) chained into stage B (sum over kept columns).
// accB[j][r]: x = 16t + 4q + r, y = NJ*n + j.
template <int NJ>
__device__ __forceinline__ void idft_tile(const float (&va)[2][8], const float* ta, const float* tb, int t, int SA, int SB,
                                          int lane, f32x4 (&accB)[NJ]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 accA[2] = {zero, zero};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        if (s < SA) {
            const float tv = ta[(t * SA + s) * 64 + lane];
            accA[0] = cfd_mfma16x16x4(va[0][s], tv, accA[0]);
            accA[1] = cfd_mfma16x16x4(va[1][s], tv, accA[1]);
        }
    }
    // accA[mu][r] (lane = (q,i)): U'[x = 16t+i][c = 4*(4mu+r)+q]  == stage-B A operand of k-step 4mu+r
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
        if (sp < SB) {
            const float av = accA[sp >> 2][sp & 3];
#pragma unroll
            for (int j = 0; j < NJ; ++j) accB[j] = cfd_mfma16x16x4(av, tb[(sp * NJ + j) * 64 + lane], accB[j]);
        }
    }
}

// Split-bf16 form of the tile (cfd_common.h "bf16x3"): both stages have K <= 32 (stage A: SA <= 8 k-steps of 4 kept
// rows, stage B: SB <= 8 k-steps of 4 kept columns), so each becomes ONE K = 32 MFMA per output tile and split term
// -- 6 + 12 bf16 MFMAs (~17 cycles) instead of 14 + 24 fp32 ones (32 cycles).  k-slot (q, v) of the K = 32 operand
// is (k-step v, k index q) of the fp32 form, so va[mu][0..7] and the stage-A accumulators feed the A operands as they
// are; ta3 / tb3 point at the hi vectors of this tile's / column group's table (lo = +64 vectors).
template <int AP>
struct IdftSplitA {
    CfdAct8<AP> a[2];
};
template <int AP>
__device__ __forceinline__ IdftSplitA<AP> idft_split(const float (&va)[2][8]) {
    IdftSplitA<AP> s;
    s.a[0] = cfd_act_split8<AP>(va[0]);
    s.a[1] = cfd_act_split8<AP>(va[1]);
    return s;
}
// `tw` (may be NULL): fp32 stage-B factors [4][32] of the tail columns 64 .. 64+E-1 (plan.d_tail, in LDS); *tailv then receives this
// lane's value of tail column e = lane >> 4 for row 16t + (lane & 15): the stage-A result U'[x][c] times tw[e][c], summed over c with
// exact fp32 FMAs (a lane holds the eight c = 4 (4 mu + r) + q of its group q; the four groups of a row meet in cfd_row_sum4).
template <int NJ, int AP>
__device__ __forceinline__ void idft_tile_b3(const IdftSplitA<AP>& sa, const bf16x8* ta3, const bf16x8* tb3, int lane,
                                             f32x4 (&accB)[NJ], const float* tw = nullptr, int E = 0, float* tailv = nullptr) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const CfdTab3 ta = cfd_tab3(ta3, lane);
    f32x4 accA[2] = {zero, zero};
#pragma unroll
    for (int k = 0; k < cfd_nterm(AP); ++k) {
        accA[0] = cfd_term<true, AP>(sa.a[0], ta, k, accA[0]);
        accA[1] = cfd_term<true, AP>(sa.a[1], ta, k, accA[1]);
    }
    if (tw) {
        const int qq = lane >> 4;
        float mine = 0.f;
        for (int e = 0; e < E; ++e) {
            float pe = 0.f;
#pragma unroll
            for (int mu = 0; mu < 2; ++mu)
#pragma unroll
                for (int r = 0; r < 4; ++r) pe = fmaf(accA[mu][r], tw[e * 32 + 4 * (4 * mu + r) + qq], pe);
            pe = cfd_row_sum4(pe);
            if (e == qq) mine = pe;
        }
        *tailv = mine;
    }
    const float u[8] = {accA[0][0], accA[0][1], accA[0][2], accA[0][3], accA[1][0], accA[1][1], accA[1][2], accA[1][3]};
    const CfdAct8<AP> us = cfd_act_split8<AP>(u);
    // (the table pieces are read where they are used: NJ x 3 resident vectors would cost the fused block kernel 48 VGPRs)
#pragma unroll
    for (int k = 0; k < cfd_nterm(AP); ++k)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            accB[j] = cfd_mfma16x16x32_bf16(us.p[cfd_term_a(AP, k)], tb3[(CFD_TW * j + cfd_term_t(AP, k)) * 64 + lane], accB[j]);
}
#define CFD_B3_TABV ((CFD_TW * CFD_KB_TMAX + CFD_TW * 4) * 64)  // 16-byte vectors of the split tables at T = CFD_KB_TMAX, NJ = 4 (three pieces each)

#define CFD_BLK_ZS 577  // floats of one staged mode vector: 2*M (m1 = m2 = 12) + the zero slot
#define CFD_KB_ZS 580   // the same in k_block, rounded up to whole float4s (16-byte LDS stores)

template <int V>
struct CfdParity { static constexpr int value = V; };

#define CFD_IDFT_ZMAX (4 * 15 * 16 + 1)  // 2*M floats (m1 <= 15, m2 <= 16) + the zero slot

template <int NJ, bool VEC4, int EPI>
__global__ __launch_bounds__(64 * CFD_WAVES) void k_idft(const float* __restrict__ z, const float* addend,
                                                          const float* __restrict__ aprev, float* out,
                                                          const float* __restrict__ tabs, int ntab, int nimg, int H,
                                                          int W, int m1, int m2, int T, int SA, int SB) {
    __shared__ float s_tab[(8 * 8 + 8 * NJ) * 64];  // T<=8, SA<=8, SB<=8
    __shared__ float s_z[CFD_WAVES * CFD_IDFT_ZMAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int img = blockIdx.x * CFD_WAVES + wave;
    const bool live = img < nimg;
    const int M2 = 4 * m1 * m2;
    float* zs = s_z + wave * CFD_IDFT_ZMAX;
    if (live) idft_stage_z(z + (size_t)img * M2, zs, M2, lane);
    cfd_stage_lds(s_tab, tabs, ntab);
    __syncthreads();
    if (!live) return;
    const float* ta = s_tab;
    const float* tb = ta + T * SA * 64;
    float va[2][8];
    idft_gather(zs, m1, m2, SA, q, n, va);
    const size_t ibase = (size_t)img * H * W;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    if constexpr (VEC4) {
        // W % 4 == 0 and 16-B aligned rows: whole-float4 epilogue; the tile's addend / aprev rows are requested
        // before the tile's MFMA work so their latency hides behind it.
        const bool cok = 4 * n < W;
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            float4 ad[4], ap[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xx = 16 * t + 4 * q + r;
                const size_t off = ibase + (size_t)(xx < H ? xx : 0) * W + (cok ? 4 * n : 0);
                if constexpr (EPI >= 1) ad[r] = *reinterpret_cast<const float4*>(addend + off);
                if constexpr (EPI == 2) ap[r] = *reinterpret_cast<const float4*>(aprev + off);
            }
            f32x4 accB[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) accB[j] = zero;
            idft_tile<NJ>(va, ta, tb, t, SA, SB, cfd_opaque(lane), accB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xx = 16 * t + 4 * q + r;
                float4 v = make_float4(accB[0][r], accB[1][r], accB[2][r], accB[3][r]);
                if constexpr (EPI >= 1) { v.x += ad[r].x; v.y += ad[r].y; v.z += ad[r].z; v.w += ad[r].w; }
                if constexpr (EPI == 2) {
                    float4 gg = ap[r];
                    cfd_gelu_grad4(gg.x, gg.y, gg.z, gg.w);
                    v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
                }
                if (xx < H && cok) *reinterpret_cast<float4*>(out + ibase + (size_t)xx * W + 4 * n) = v;
            }
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            f32x4 accB[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) accB[j] = zero;
            idft_tile<NJ>(va, ta, tb, t, SA, SB, cfd_opaque(lane), accB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xx = 16 * t + 4 * q + r;
                if (xx < H) {
                    const size_t rbase = ibase + (size_t)xx * W;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int y = NJ * n + j;
                        if (y < W) {
                            float v = accB[j][r];
                            if constexpr (EPI >= 1) v += addend[rbase + y];
                            if constexpr (EPI == 2) v *= cfd_gelu_grad(aprev[rbase + y]);
                            out[rbase + y] = v;
                        }
                    }
                }
            }
        }
    }
}

// Persistent variant for W == 64 (float4 rows) with the 12x12-mode table sizes: each wave strides over images, the next
// image's kept modes are requested before the current image's 152 MFMAs and parked in the other LDS slice, and
// the operator tables are loaded once per workgroup instead of once per four images.
// TAIL: the first tail.nblk workgroups of the launch do not transform anything -- they sum the partial spectral weight
// gradients of the preceding launch (cfd_reduce_tail, cfd_tail.h).  That reduction is a ~0.5-us job which as a kernel
// of its own costs the ~4.5-us dispatch floor plus a launch gap; here it rides in front of the resident transform waves.
template <int EPI, bool TAIL, int AP>
__global__ __launch_bounds__(64 * CFD_WAVES, 3) void k_idft64(const float* __restrict__ z, const float* addend,
                                                               const float* __restrict__ aprev, float* out,
                                                               const bf16x8* __restrict__ tabs3, int nimg,
                                                               int H, int m1, int m2, int T, int SA, int SB,
                                                               const CfdReduceTail tail) {
    constexpr int W = 64, NJ = 4;
    __shared__ bf16x8 s_tab3[CFD_B3_TABV];               // split-bf16 tables, T <= 4
    __shared__ float s_z[CFD_WAVES * 2 * CFD_BLK_ZS];    // two slices per wave
    const int nskip = TAIL ? tail.nblk : 0;
    if (TAIL && (int)blockIdx.x < nskip) {
        cfd_reduce_tail(tail, blockIdx.x, s_z);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int M2 = 4 * m1 * m2;
    const int stride = (gridDim.x - nskip) * CFD_WAVES;
    int img = (blockIdx.x - nskip) * CFD_WAVES + wave;
    float* zs0 = s_z + wave * 2 * CFD_BLK_ZS;
    constexpr int ZR = (CFD_BLK_ZS - 1 + 63) / 64;  // dwords per lane of one mode vector
    float zr[ZR];
    auto zfetch = [&](int im) {
        const float* zi = z + (size_t)im * M2;
#pragma unroll
        for (int k = 0; k < ZR; ++k) zr[k] = zi[lane + 64 * k < M2 ? lane + 64 * k : 0];
    };
    auto zcommit = [&](float* zs) {
#pragma unroll
        for (int k = 0; k < ZR; ++k)
            if (lane + 64 * k < M2) zs[lane + 64 * k] = zr[k];
        if (lane == 0) zs[M2] = 0.f;
    };
    zfetch(img < nimg ? img : 0);
    cfd_stage_lds(s_tab3, tabs3, (CFD_TW * T + CFD_TW * NJ) * 64);
    zcommit(zs0);
    __syncthreads();
    const bf16x8* tb3 = s_tab3 + CFD_TW * T * 64;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    while (img < nimg) {
        const int nxt = img + stride;
        zfetch(nxt < nimg ? nxt : img);
        cfd_sched_fence();
        float va[2][8];
        idft_gather(zs0 + cur * CFD_BLK_ZS, m1, m2, SA, q, n, va);
        const IdftSplitA<AP> sa = idft_split<AP>(va);  // once per image: the stage-A data does not depend on the row tile
        const size_t ibase = (size_t)img * H * W + 4 * n;
#pragma unroll 1
        for (int t = 0; t < T; ++t) {
            float4 ad[4], ap[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t off = ibase + (size_t)(16 * t + 4 * q + r) * W;
                if constexpr (EPI >= 1) ad[r] = *reinterpret_cast<const float4*>(addend + off);
                if constexpr (EPI == 2) ap[r] = *reinterpret_cast<const float4*>(aprev + off);
            }
            f32x4 accB[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) accB[j] = zero;
            idft_tile_b3<NJ, AP>(sa, s_tab3 + CFD_TW * t * 64, tb3, cfd_opaque(lane), accB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 v = make_float4(accB[0][r], accB[1][r], accB[2][r], accB[3][r]);
                if constexpr (EPI >= 1) { v.x += ad[r].x; v.y += ad[r].y; v.z += ad[r].z; v.w += ad[r].w; }
                if constexpr (EPI == 2) {
                    float4 gg = ap[r];
                    cfd_gelu_grad4(gg.x, gg.y, gg.z, gg.w);
                    v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
                }
                *reinterpret_cast<float4*>(out + ibase + (size_t)(16 * t + 4 * q + r) * W) = v;
            }
        }
        cur ^= 1;
        zcommit(zs0 + cur * CFD_BLK_ZS);  // the other slice was last read one image ago (same wave: program order)
        cfd_wave_lds_sync();
        img = nxt;
    }
}

// General-width split-bf16 inverse transform: k_idft64's scheme (stage A over the kept rows, stage B over the kept
// columns, 6 + 3 NJ bf16 MFMAs per 16-row tile) with the columns dealt as y = 16 j + n, so the epilogue loads and the stores
// of a wave are runs of 16 consecutive elements per output row whatever W is (66 x 65 grids: the fp32 kernel k_idft deals
// y = NJ n + j and issues 14 + 8 NJ fp32 MFMAs of 32 cycles per tile).  Persistent waves, next image's modes prefetched.
// TA / TADD / TPREV: storage types of `out`, `addend` and `aprev` (bf16 activation storage: the forward pass writes bf16 `out` from an
// fp32 `addend`; the backward pass of bf16-storage TRAINING writes an fp32 gradient from an fp32 addend and a bf16 `aprev`).
template <int NJ, int EPI, typename TA, typename TADD, typename TPREV, int AP>
__global__ __launch_bounds__(64 * CFD_WAVES, 2) void k_idft_g(const float* __restrict__ z, const TADD* addend,
                                                               const TPREV* __restrict__ aprev, TA* out,
                                                               const bf16x8* __restrict__ tabs3, int ntabv, int nimg, int H,
                                                               int W, int m1, int m2, int T, int SA) {
    CFD_DYN_SHARED(bf16x8, s_dyn);
    bf16x8* s_tab3 = s_dyn;                                   // ta3[T][hi|lo][64] | tb3[NJ][hi|lo][64]
    float* s_z = reinterpret_cast<float*>(s_dyn + ntabv);     // two mode slices per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int M2 = 4 * m1 * m2;
    const int stride = gridDim.x * CFD_WAVES;
    int img = blockIdx.x * CFD_WAVES + wave;
    float* zs0 = s_z + wave * 2 * CFD_IDFT_ZMAX;
    constexpr int ZR = (CFD_IDFT_ZMAX - 1 + 63) / 64;  // dwords per lane of one mode vector
    float zr[ZR];
    auto zfetch = [&](int im) {
        const float* zi = z + (size_t)im * M2;
#pragma unroll
        for (int k = 0; k < ZR; ++k) zr[k] = zi[lane + 64 * k < M2 ? lane + 64 * k : 0];
    };
    auto zcommit = [&](float* zs) {
#pragma unroll
        for (int k = 0; k < ZR; ++k)
            if (lane + 64 * k < M2) zs[lane + 64 * k] = zr[k];
        if (lane == 0) zs[M2] = 0.f;
    };
    zfetch(img < nimg ? img : 0);
    cfd_stage_lds(s_tab3, tabs3, ntabv);
    zcommit(zs0);
    __syncthreads();
    const bf16x8* tb3 = s_tab3 + CFD_TW * T * 64;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int colc[NJ];
    bool cok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { cok[j] = 16 * j + n < W; colc[j] = cok[j] ? 16 * j + n : W - 1; }
    // One (image, row tile) per trip.  The addend / gelu' source rows of the NEXT trip are requested before this trip's MFMAs and stores
    // (round 6): vmcnt counts loads and stores in order, so loads issued after a tile's stores were waited for together with the stores'
    // write acknowledges -- two exposed memory latencies per row tile, ~3.9 us of a tile's ~4 at 64 rollout cases (66 x 65, 32 channels:
    // 19.6 us per launch for 70 MB).
    int cur = 0, t = 0;
    float ad[4][NJ], ap[4][NJ], adn[4][NJ], apn[4][NJ];
    auto ldtile = [&](int im, int tt, float (&a_)[4][NJ], float (&p_)[4][NJ]) {
        const size_t ib = (size_t)im * H * W;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int xx = 16 * tt + 4 * q + r;
            const size_t rb = ib + (size_t)(xx < H ? xx : H - 1) * W;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (EPI >= 1) a_[r][j] = cfd_ld(addend + rb + colc[j]);
                if constexpr (EPI == 2) p_[r][j] = cfd_ld(aprev + rb + colc[j]);
            }
        }
    };
    if (img < nimg) ldtile(img, 0, ad, ap);
    IdftSplitA<AP> sa;
    while (img < nimg) {
        if (t == 0) {  // (uniform) a new image: its modes were staged one image ago; request the next image's
            const int nxt = img + stride;
            zfetch(nxt < nimg ? nxt : img);
            float va[2][8];
            idft_gather(zs0 + cur * CFD_IDFT_ZMAX, m1, m2, SA, q, n, va);
            sa = idft_split<AP>(va);
        }
        int nt = t + 1, nimg_ = img;
        if (nt == T) { nt = 0; nimg_ = img + stride; }
        if (nimg_ < nimg) ldtile(nimg_, nt, adn, apn);
        cfd_sched_fence();
        const size_t ibase = (size_t)img * H * W;
        {
            f32x4 accB[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) accB[j] = zero;
            idft_tile_b3<NJ, AP>(sa, s_tab3 + CFD_TW * t * 64, tb3, cfd_opaque(lane), accB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xx = 16 * t + 4 * q + r;
                const size_t rb = ibase + (size_t)xx * W;
#pragma unroll
                for (int j = 0; j < NJ; j += 2) {
                    cfd_f2 v = {accB[j][r], j + 1 < NJ ? accB[j + 1 < NJ ? j + 1 : j][r] : 0.f};
                    if constexpr (EPI >= 1) v = v + cfd_f2{ad[r][j], j + 1 < NJ ? ad[r][j + 1 < NJ ? j + 1 : j] : 0.f};
                    if constexpr (EPI == 2) v = v * cfd_gelu_grad2(cfd_f2{ap[r][j], j + 1 < NJ ? ap[r][j + 1 < NJ ? j + 1 : j] : 0.f});
                    if (xx < H && cok[j]) cfd_st(out + rb + 16 * j + n, v.x);
                    if (j + 1 < NJ && xx < H && cok[j + 1 < NJ ? j + 1 : j]) cfd_st(out + rb + 16 * (j + 1) + n, v.y);
                }
            }
        }
        cfd_sched_fence();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if constexpr (EPI >= 1) ad[r][j] = adn[r][j];
                if constexpr (EPI == 2) ap[r][j] = apn[r][j];
            }
        if (nt == 0) {
            cur ^= 1;
            zcommit(zs0 + cur * CFD_IDFT_ZMAX);  // the other slice was last read one image ago (same wave: program order)
            cfd_wave_lds_sync();
        }
        img = nimg_;
        t = nt;
    }
}

template <typename TA, typename TADD = TA, typename TPREV = TA>
static bool launch_idft_g(const cfd_plan* p, const float* z, const TADD* addend, const TPREV* aprev, TA* out, int nimg, int epi,
                          hipStream_t st) {
    if (!p->d_inv_g || p->NJG < 1 || p->NJG > 5) return false;
    int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
    if (blocks > 2 * 256) blocks = 2 * 256;
    const size_t lds = (size_t)p->n_inv_gv * sizeof(bf16x8) + (size_t)CFD_WAVES * 2 * CFD_IDFT_ZMAX * sizeof(float);
    constexpr bool all_f32 = sizeof(TA) == 4 && sizeof(TADD) == 4 && sizeof(TPREV) == 4;
    const bool ap3 = all_f32 && cfd_act_pieces() == 3;
#define CFD_IDG_P(NJ_, E_, P_)                                                                                       \
    hipLaunchKernelGGL((k_idft_g<NJ_, E_, TA, TADD, TPREV, P_>), dim3(blocks), dim3(64 * CFD_WAVES), lds, st, z, addend, aprev, out,  \
                       (const bf16x8*)p->d_inv_g, p->n_inv_gv, nimg, p->H, p->W, p->m1, p->m2, p->T, p->SA)
#define CFD_IDG(NJ_, E_)                                                              \
    do {                                                                              \
        if constexpr (all_f32) { if (ap3) CFD_IDG_P(NJ_, E_, 3); else CFD_IDG_P(NJ_, E_, 2); } \
        else CFD_IDG_P(NJ_, E_, 2);                                                   \
    } while (0)
#define CFD_IDG_E(NJ_) do { if (epi == 0) CFD_IDG(NJ_, 0); else if (epi == 1) CFD_IDG(NJ_, 1); else CFD_IDG(NJ_, 2); } while (0)
    switch (p->NJG) {
        case 1: CFD_IDG_E(1); break;
        case 2: CFD_IDG_E(2); break;
        case 3: CFD_IDG_E(3); break;
        case 4: CFD_IDG_E(4); break;
        default: CFD_IDG_E(5); break;
    }
#undef CFD_IDG_E
#undef CFD_IDG
#undef CFD_IDG_P
    return true;
}

static bool idft64_applies(const cfd_plan* p) {
    return p->W == 64 && p->H % 16 == 0 && p->d_inv_b3 && 4 * p->m1 * p->m2 + 1 <= CFD_BLK_ZS &&
           cfd_tune_get(CFD_TUNE_EXACT_FP32) != 1;
}

// `tail` (may be NULL): a partial-sum reduction to run as extra workgroups of this launch; only the 64-wide kernel with
// epi == 0 takes it (callers check idft64_applies and the alignment first).
template <int NJ, bool VEC4>
static int launch_idft(const cfd_plan* p, const float* z, const float* addend, const float* aprev, float* out,
                       int nimg, int epi, hipStream_t st, const CfdReduceTail* tail = nullptr) {
    int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
    if constexpr (VEC4) {
        if (idft64_applies(p)) {
            if (blocks > CFD_DFT_CAP) blocks = CFD_DFT_CAP;  // resident workgroups; waves stride over the images
            CfdReduceTail none{};
            const bool ap3 = cfd_act_pieces() == 3;
#define CFD_IDFT64_P(E, P_)                                                                                        \
    hipLaunchKernelGGL((k_idft64<E, false, P_>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, z, addend, aprev, out,  \
                       (const bf16x8*)p->d_inv_b3, nimg, p->H, p->m1, p->m2, p->T, p->SA, p->SB, none)
#define CFD_IDFT64(E) do { if (ap3) CFD_IDFT64_P(E, 3); else CFD_IDFT64_P(E, 2); } while (0)
            CFD_PROF_W(epi == 0 ? "k_idft" : (epi == 1 ? "k_idft_add" : "k_idft_add_dgelu"), st,
               (double)nimg * (4.0 * p->H * p->W * (1 + epi) + 16.0 * p->m1 * p->m2), (double)nimg * (8.0 * (p->m1 + 1) * p->W * p->m2 + 4.0 * p->H * p->W * p->m2));
            if (epi == 0 && tail && tail->nblk > 0) {
                if (ap3)
                    hipLaunchKernelGGL((k_idft64<0, true, 3>), dim3(blocks + tail->nblk), dim3(64 * CFD_WAVES), 0, st, z, addend,
                                       aprev, out, (const bf16x8*)p->d_inv_b3, nimg, p->H, p->m1, p->m2, p->T, p->SA, p->SB, *tail);
                else
                    hipLaunchKernelGGL((k_idft64<0, true, 2>), dim3(blocks + tail->nblk), dim3(64 * CFD_WAVES), 0, st, z, addend,
                                       aprev, out, (const bf16x8*)p->d_inv_b3, nimg, p->H, p->m1, p->m2, p->T, p->SA, p->SB, *tail);
            }
            else if (epi == 0) CFD_IDFT64(0);
            else if (epi == 1) CFD_IDFT64(1);
            else CFD_IDFT64(2);
#undef CFD_IDFT64
#undef CFD_IDFT64_P
            CFD_LAUNCH_CHECK("cfd_spectral_idft");
            return CFD_OK;
        }
    }
    if (cfd_tune_get(CFD_TUNE_EXACT_FP32) != 1 && cfd_tune_get(CFD_TUNE_GENERAL_B3) != 0 && p->d_inv_g && p->NJG <= 5) {
        CFD_PROF_W(epi == 0 ? "k_idft" : (epi == 1 ? "k_idft_add" : "k_idft_add_dgelu"), st,
                   (double)nimg * (4.0 * p->H * p->W * (1 + epi) + 16.0 * p->m1 * p->m2), (double)nimg * (8.0 * (p->m1 + 1) * p->W * p->m2 + 4.0 * p->H * p->W * p->m2));
        if (launch_idft_g<float>(p, z, addend, aprev, out, nimg, epi, st)) {
            CFD_LAUNCH_CHECK("cfd_spectral_idft(general)");
            return CFD_OK;
        }
    }
#define CFD_IDFT_LAUNCH(E)                                                                                         \
    hipLaunchKernelGGL((k_idft<NJ, VEC4, E>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, z, addend, aprev, out,     \
                       (const float*)p->d_inv, p->n_inv, nimg, p->H, p->W, p->m1, p->m2, p->T, p->SA, p->SB)
    CFD_PROF_W(epi == 0 ? "k_idft" : (epi == 1 ? "k_idft_add" : "k_idft_add_dgelu"), st,
               (double)nimg * (4.0 * p->H * p->W * (1 + epi) + 16.0 * p->m1 * p->m2), (double)nimg * (8.0 * (p->m1 + 1) * p->W * p->m2 + 4.0 * p->H * p->W * p->m2));
    if (epi == 0) CFD_IDFT_LAUNCH(0);
    else if (epi == 1) CFD_IDFT_LAUNCH(1);
    else CFD_IDFT_LAUNCH(2);
#undef CFD_IDFT_LAUNCH
    CFD_LAUNCH_CHECK("cfd_spectral_idft");
    return CFD_OK;
}

int cfd_int_spectral_idft(const cfd_plan* p, const float* z, const void* addend, const void* aprev, void* out, int nimg, int epi,
                          int dt, void* stream) {
    if (dt == CFD_DT_F32) return cfd_spectral_idft(p, z, (const float*)addend, (const float*)aprev, (float*)out, nimg, epi, stream);
    CFD_REQUIRE(p && z && out && nimg >= 0 && epi >= 0 && epi <= 2 && (epi < 1 || addend) && (epi < 2 || aprev), CFD_ERR_INVALID_ARG,
                "cfd_spectral_idft: bad arguments");
    if (nimg == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    CFD_PROF_W(epi == 0 ? "k_idft" : (epi == 1 ? "k_idft_add" : "k_idft_add_dgelu"), st,
               (double)nimg * (2.0 * p->H * p->W * (1 + 2 * (epi >= 1) + (epi == 2)) + 16.0 * p->m1 * p->m2), (double)nimg * (8.0 * (p->m1 + 1) * p->W * p->m2 + 4.0 * p->H * p->W * p->m2));
    CFD_REQUIRE((launch_idft_g<__bf16, float>(p, z, (const float*)addend, (const __bf16*)aprev, (__bf16*)out, nimg, epi, st)), CFD_ERR_UNSUPPORTED,
                "cfd_spectral_idft: bf16 activation storage needs H <= 70 and W <= 80 (grid %dx%d)", p->H, p->W);
    CFD_LAUNCH_CHECK("cfd_spectral_idft(bf16 storage)");
    return CFD_OK;
}

// Input-gradient form of bf16-storage training: out (fp32 gradient) = (idft(z) + addend (fp32)) [* gelu'(aprev), aprev stored as bf16].
int cfd_int_spectral_idft_grad(const cfd_plan* p, const float* z, const float* addend, const void* aprev, float* out, int nimg,
                               int dt, void* stream) {
    const int epi = aprev ? 2 : 1;
    if (dt == CFD_DT_F32) return cfd_spectral_idft(p, z, addend, (const float*)aprev, out, nimg, epi, stream);
    CFD_REQUIRE(p && z && out && addend && nimg >= 0, CFD_ERR_INVALID_ARG, "cfd_spectral_idft(grad): bad arguments");
    if (nimg == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    CFD_PROF_W(epi == 1 ? "k_idft_add" : "k_idft_add_dgelu", st,
               (double)nimg * (p->H * p->W * (8.0 + (epi == 2 ? 2.0 : 0.0)) + 16.0 * p->m1 * p->m2), (double)nimg * (8.0 * (p->m1 + 1) * p->W * p->m2 + 4.0 * p->H * p->W * p->m2));
    CFD_REQUIRE((launch_idft_g<float, float, __bf16>(p, z, addend, (const __bf16*)aprev, out, nimg, epi, st)), CFD_ERR_UNSUPPORTED,
                "cfd_spectral_idft(grad): bf16 activation storage needs H <= 70 and W <= 80 (grid %dx%d)", p->H, p->W);
    CFD_LAUNCH_CHECK("cfd_spectral_idft(grad, bf16 storage)");
    return CFD_OK;
}

extern "C" int cfd_spectral_idft(const cfd_plan* p, const float* z, const float* addend, const float* aprev,
                                 float* out, int nimg, int epi, void* stream) {
    CFD_REQUIRE(p && z && out && nimg >= 0, CFD_ERR_INVALID_ARG, "cfd_spectral_idft: NULL pointer or negative count");
    CFD_REQUIRE(epi >= 0 && epi <= 2, CFD_ERR_INVALID_ARG, "cfd_spectral_idft: epi must be 0,1,2");
    CFD_REQUIRE(epi < 1 || addend, CFD_ERR_INVALID_ARG, "cfd_spectral_idft: epi>=1 needs addend");
    CFD_REQUIRE(epi < 2 || aprev, CFD_ERR_INVALID_ARG, "cfd_spectral_idft: epi==2 needs aprev");
    CFD_REQUIRE(p->T <= 8 && p->SA <= 8 && p->SB <= 8, CFD_ERR_UNSUPPORTED, "cfd_spectral_idft: plan too large");
    if (nimg == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool al = ((uintptr_t)out % 16) == 0 && (!addend || ((uintptr_t)addend % 16) == 0) &&
                    (!aprev || ((uintptr_t)aprev % 16) == 0);
    if (p->NJ == 4) {
        if (p->W % 4 == 0 && al) return launch_idft<4, true>(p, z, addend, aprev, out, nimg, epi, st);
        return launch_idft<4, false>(p, z, addend, aprev, out, nimg, epi, st);
    }
    if (p->NJ == 5) return launch_idft<5, false>(p, z, addend, aprev, out, nimg, epi, st);
    cfd_set_error("cfd_spectral_idft: W=%d needs NJ=%d columns per lane (supported: 4,5)", p->W, p->NJ);
    return CFD_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------------
// SpectralConv2d_fast forward / backward as one call each
// ------------------------------------------------------------------------------------------------------
extern "C" int cfd_spectral_conv2d_fwd(const cfd_plan* p, const float* x, const float* w1, const float* w2, float* y,
                                       float* xh_out, float* z_ws, int B, int Cin, int Cout, void* stream) {
    CFD_REQUIRE(p && x && w1 && w2 && y && xh_out && z_ws, CFD_ERR_INVALID_ARG, "cfd_spectral_conv2d_fwd: NULL pointer");
    CFD_TRY(cfd_spectral_dft(p, x, xh_out, B * Cin, 0, stream));
    CFD_TRY(cfd_spectral_mix(p, xh_out, w1, w2, z_ws, B, Cin, Cout, 0, stream));
    CFD_TRY(cfd_spectral_idft(p, z_ws, nullptr, nullptr, y, B * Cout, 0, stream));
    return CFD_OK;
}

extern "C" size_t cfd_spectral_conv2d_bwd_workspace_bytes(const cfd_plan* p, int B, int Cin, int Cout) {
    if (!p || B <= 0) return 0;
    const size_t M = 2 * (size_t)p->m1 * p->m2;
    const size_t gh = cfd_align_up((size_t)B * Cout * M * 8, 256);
    const size_t gz = cfd_align_up((size_t)B * Cin * M * 8, 256);
    return gh + gz + cfd_spectral_wgrad_workspace_bytes(p, B, Cin, Cout);
}

extern "C" int cfd_spectral_conv2d_bwd(const cfd_plan* p, const float* gy, const float* xh, const float* w1,
                                       const float* w2, float* gx, float* gw1, float* gw2, void* ws, int B, int Cin,
                                       int Cout, void* stream) {
    CFD_REQUIRE(p && gy && w1 && w2 && ws, CFD_ERR_INVALID_ARG, "cfd_spectral_conv2d_bwd: NULL pointer");
    CFD_REQUIRE((gw1 == nullptr) == (gw2 == nullptr), CFD_ERR_INVALID_ARG, "cfd_spectral_conv2d_bwd: gw1/gw2 must both be given or both NULL");
    CFD_REQUIRE(!gw1 || xh, CFD_ERR_INVALID_ARG, "cfd_spectral_conv2d_bwd: weight gradients need the saved modes xh");
    const size_t M = 2 * (size_t)p->m1 * p->m2;
    char* base = (char*)ws;
    float* gh = (float*)base;
    float* gz = (float*)(base + cfd_align_up((size_t)B * Cout * M * 8, 256));
    void* wws = base + cfd_align_up((size_t)B * Cout * M * 8, 256) + cfd_align_up((size_t)B * Cin * M * 8, 256);
    CFD_TRY(cfd_spectral_dft(p, gy, gh, B * Cout, 0, stream));
    CfdReduceTail tail{};
    if (gw1 && gx) {
        // the partial-sum reduction of the weight gradient rides in the inverse transform's launch where that is the
        // 64-wide kernel (one launch less on the critical path)
        const bool ride = idft64_applies(p) && p->NJ == 4 && ((uintptr_t)gx % 16) == 0;
        CFD_TRY(cfd_int_spectral_mix_adj_wgrad(p, xh, gh, w1, w2, gz, gw1, gw2, wws, B, Cin, Cout, stream,
                                               ride ? &tail.spec : nullptr));
        if (tail.spec.part) {
            const long nb = ((long)Cin * Cout * M + 255) / 256;
            tail.nblk = (int)(nb < 256 ? nb : 256);
        }
    } else if (gw1) {
        CFD_TRY(cfd_spectral_wgrad(p, xh, gh, gw1, gw2, wws, B, Cin, Cout, stream));
    } else if (gx) {
        CFD_TRY(cfd_spectral_mix(p, gh, w1, w2, gz, B, Cin, Cout, 1, stream));
    }
    if (gx) {
        if (tail.nblk > 0) CFD_TRY((launch_idft<4, true>(p, gz, nullptr, nullptr, gx, B * Cin, 0, (hipStream_t)stream, &tail)));
        else CFD_TRY(cfd_spectral_idft(p, gz, nullptr, nullptr, gx, B * Cin, 0, stream));
    }
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Fused FnoBlock body (fno2d.py:106-112 and its input-gradient): 1x1 conv + inverse pruned DFT (+ epilogue)
// ------------------------------------------------------------------------------------------------------
//   dst[b,d,x,y] = epi( bias[d] + sum_s wsel(d,s) f(src[b,s,x,y]) + idft(z[b,d])[x,y] )
// forward : src = a_l, f = GELU for l > 0, wsel(d,s) = w0[d,s], dst = a_{l+1} (pre-activation)
// backward: src = dL/da_{l+1}, f = id, wsel(d,s) = w0[s,d], epi = * gelu'(a_l[b,d,x,y]), dst = dL/da_l
// so one activation-sized read and one write replace the chanmix -> idft_add round trip (5N -> 3N per layer forward,
// 8N -> 6N backward).  Workgroup = one batch entry, all channels, its 16-row tiles one after another.  Each source tile streams through
// LDS in NCH chunks of NW channels (wave w fetches channel w of each chunk, two chunks ahead in registers); wave w
// owns destination channels w, w+NW, ... (DPW of them): their inverse transforms run on the matrix pipe while the
// chunks are in flight, the channel mix accumulates on the VALU straight into the MFMA accumulator layout
// (x = 16t+4q+r, y = 4n+j), and a_out leaves in whole float4 rows.  W == 64, H % 16 == 0.

// NW waves; wave w owns destination channels w, w+NW, ... (DPW of them) and fetches channel w of each of the NCH
// source chunks.  NW = 4 (one wave per SIMD) wherever the channel count allows.
// TAIL (backward only): the first tail.nblk workgroups of the launch run the reductions of this FnoBlock's two weight
// gradients (cfd_tail.h) instead of a batch entry -- two kernel launches less per block and backward pass.
// GEN (round 4): any grid with 64 <= W <= 68 and H <= 80 -- the 66 x 65 grids of the tube / dam / cylinder problems.  Rows have the
// pitch W (so a lane's four columns are a 16-byte access from a 4-byte-aligned address: cfd_f4u, measured at 94 % of the aligned
// rate), the row tiles cover ceil(H / 16) * 16 rows with the rows past H clamped on load and masked on store, and the E = W - 64
// tail columns ride beside the 64 MFMA columns on the VALU: their source values sit in a small LDS plane per chunk (lane n < 4 of a
// group fetches the tail of row 4n + q), lane (q = e, i = row) accumulates the channel mix of tail column e for its row, and the
// inverse transform's contribution comes from the stage-A accumulators (idft_tile_b3's tail sums).
template <bool GEN>
__device__ __forceinline__ float4 cfd_ldrow4(const float* p) {
    if constexpr (GEN) return cfd_ld4u(p);
    else return *reinterpret_cast<const float4*>(p);
}
template <bool GEN>
__device__ __forceinline__ void cfd_strow4(float* p, float4 v) {
    if constexpr (GEN) cfd_st4u(p, v);
    else *reinterpret_cast<float4*>(p) = v;
}
template <int NW, int DPW, int NCH, bool ACT, bool TRANS, bool DGELU, bool TAIL, int AP, bool GEN, bool STEMG = false>
__global__ __launch_bounds__(64 * NW) void k_block(const float* __restrict__ src, const float* __restrict__ z,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   const float* __restrict__ aprev, float* __restrict__ dst,
                                                   const bf16x8* __restrict__ tabs3, int Cs, int Cd, int H, int m1,
                                                   int m2, int T, int SA, int SB, const CfdReduceTail tail, int SPL,
                                                   int pitch, const float* __restrict__ tailtab, int ND, int NENT, const CfdStemG sg) {
    static_assert(!STEMG || (TRANS && !DGELU && !ACT), "the lifting-layer sums ride in the plain input-gradient kernel");
    constexpr int W = 64, NJ = 4;
    constexpr int WS = DPW <= 4 ? 4 : 8;              // floats per weight-table entry
    // SBUF (round 6, the (8,4,4) shape = 25 .. 32 channels in ONE workgroup per entry): a single source-chunk buffer -- the
    // second one is what pushed this shape to 168 KB of LDS; the price is a second workgroup barrier per chunk (all waves done with
    // the chunk before the next one is committed), the prefetch two chunks ahead in registers stays
    constexpr bool SBUF = NW == 8 && DPW == 4 && NCH == 4;
    constexpr int NBUF = SBUF ? 1 : 2;
    __shared__ float4 s_src[NBUF * NW * 16 * 16];     // [buf][channel in chunk][row][float4 column]
    __shared__ bf16x8 s_tab3[CFD_B3_TABV];            // split-bf16 inverse tables: ta3 of every tile (T <= 4) | tb3
    __shared__ float s_z[NW * DPW * CFD_KB_ZS];      // kept modes of this wave's destination channels
    if (TAIL && (int)blockIdx.x < tail.nblk) {
        cfd_reduce_tail(tail, blockIdx.x, s_z);  // (s_z: scratch of the 1x1 partial-record sums)
        return;
    }
    __shared__ float4 s_w[NW * NW * NCH * (WS / 4)];  // [wave][source channel] -> weights of the wave's DPW channels
    __shared__ float s_tail[GEN ? NBUF * NW * 16 * 4 : 4];  // GEN: [buf][channel in chunk][row][tail column e] source values
    __shared__ float s_tw[GEN ? 4 * 32 : 4];             // GEN: stage-B factors of the tail columns (plan.d_tail)
    static_assert(NW * NCH <= 64, "one lane per source channel fills the weight table");
    static_assert(DPW <= 8, "weight-table entry holds at most 8 destination channels");
    const int lane = threadIdx.x & 63;
    const int wave = cfd_uniform(threadIdx.x >> 6);
    const int q = lane >> 4, n = lane & 15;
    // SPL workgroups share a batch entry by row tiles (SPL divides T): with fewer batch entries than CUs one workgroup per
    // entry leaves most of the chip idle for the same ~45 us (B = 64: 44.9 us against 24.8 for the two-pass route); every
    // tile is computed exactly as in the unsplit kernel, so a sample's result does not depend on the batch size
    const int bi0 = TAIL ? (int)blockIdx.x - tail.nblk : (int)blockIdx.x;
    // ND > 1 (round 5: 25 .. 32 channels, the reference's default width): the destination channels of an entry are dealt to ND
    // workgroups of NW * DPW channels each (all of a wide entry's modes do not fit one workgroup's LDS); every one streams ALL the
    // source channels, so the source is read ND times -- the ND workgroups of one (entry, row split) sit 8 block indices apart, i.e.
    // on the same XCD (workgroups are dealt to the XCDs round robin), and the second read is served by that XCD's L2.
    int bi = bi0, dg = 0;
    if (ND > 1) {
        const int xcd = bi0 & 7, l = bi0 >> 3;
        dg = l % ND;
        bi = (l / ND) * 8 + xcd;
    }
    const int dbase = dg * NW * DPW;  // first destination channel of this workgroup
    // (round 4: SPL need not divide T -- five row tiles at 66 x 65: the last workgroup of an entry takes what is left)
    const int b = bi / SPL, TPWmax = (T + SPL - 1) / SPL, t0 = (bi - b * SPL) * TPWmax;
    if (bi >= NENT) return;  // (NENT = batch entries x row splits; with ND > 1 the grid is padded to whole groups of 8 * ND)
    const int TPW = T - t0 < TPWmax ? T - t0 : TPWmax;  // >= 1: the launcher sizes SPL as ceil(T / TPWmax)
    const int P = GEN ? pitch : W;  // row pitch in floats
    const int E = GEN ? pitch - W : 0;  // tail columns
    const int HW = H * P;
    const int M2 = 4 * m1 * m2;
    const int G = TPW * NCH;  // chunks of this workgroup's tiles, streamed tile after tile; chunk g lives in buffer g & 1
    auto rowc = [&](int x) { return GEN ? (x < H ? x : H - 1) : x; };  // rows past H: clamped on load (their results are never stored)
    // ---- this wave's slice of a source chunk: channel ((g % NCH)*NW + wave) of tile g / NCH, rows 4k+q ----
    float4 R[2][4];
    float RT[2][GEN ? 4 : 1];  // GEN: tail values of row 4n + q (lanes n < 4), chunk parity as R
    auto fetch = [&](int g, float4 (&r)[4], float (&rt)[GEN ? 4 : 1]) {
        const int tl = g / NCH, ch = (g - tl * NCH) * NW + wave, t = t0 + tl;
        const float* pl = src + ((size_t)b * Cs + (ch < Cs ? ch : 0)) * HW;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = cfd_ldrow4<GEN>(pl + (size_t)rowc(16 * t + q + 4 * k) * P + 4 * n);
        if constexpr (GEN) {
            const float* pt = pl + (size_t)rowc(16 * t + q + 4 * (n & 3)) * P + W;
#pragma unroll
            for (int e = 0; e < 4; ++e) rt[e] = pt[e < E ? e : 0];  // (clamped index: always inside the row)
        }
    };
    auto commit = [&](int c, int buf, const float4 (&r)[4], const float (&rt)[GEN ? 4 : 1]) {
        const bool live = c * NW + wave < Cs;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4 v = r[k];
            if constexpr (ACT) cfd_gelu4(v.x, v.y, v.z, v.w);
            if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
            s_src[((buf * NW + wave) * 16 + 4 * k + q) * 16 + n] = v;
        }
        if constexpr (GEN) {
            if (n < 4) {
                float4 v = make_float4(rt[0], rt[1], rt[2], rt[3]);
                if constexpr (ACT) cfd_gelu4(v.x, v.y, v.z, v.w);
                if (!live) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (E < 4) v.w = 0.f;
                if (E < 3) v.z = 0.f;
                if (E < 2) v.y = 0.f;
                *reinterpret_cast<float4*>(&s_tail[((buf * NW + wave) * 16 + 4 * n + q) * 4]) = v;
            }
        }
    };
    fetch(0, R[0], RT[0]);
    fetch(1, R[1], RT[1]);
    // once per batch entry: operator tables, this wave's mixing weights and kept modes.  ALL the global loads are issued
    // before the first LDS store (the first version copied table, weights and each channel's modes one after the other:
    // three to eleven dependent memory latencies, 13-18 k cycles = 13 % of the workgroup's life -- tools: CFD_BDIAG)
    constexpr int NTV = (CFD_TW * CFD_KB_TMAX + CFD_TW * NJ) * 64;  // table vectors at T = CFD_KB_TMAX (checked by the launcher)
    constexpr int TPT = (NTV + 64 * NW - 1) / (64 * NW);       // per thread
    const int ntab = (CFD_TW * T + CFD_TW * NJ) * 64;
    bf16x8 tv[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = threadIdx.x + k * 64 * NW;
        tv[k] = tabs3[i < ntab ? i : 0];
    }
    // (every load below comes from a clamped, always valid address and is zeroed / skipped only after ALL of them are in flight:
    // `inside ? w[..] : 0` and a clamped load whose only use sits under an `if` both compile to a branch around the load with a full
    // wait inside -- five serialized memory round trips at the start of every workgroup in the first version)
    float wl[WS];
#pragma unroll
    for (int dd = 0; dd < WS; ++dd) {
        const int d = dbase + wave + dd * NW, dc = d < Cd ? d : Cd - 1, lc = lane < Cs ? lane : Cs - 1;
        wl[dd] = dd < DPW ? (TRANS ? w[(size_t)lc * Cd + dc] : w[(size_t)dc * Cs + lc]) : 0.f;
    }
    float bv[DPW];
    constexpr int ZV = (CFD_KB_ZS - 1 + 255) / 256;  // float4 rounds per channel (M2 <= CFD_KB_ZS - 1, a multiple of 4)
    float4 zv[DPW][ZV];
#pragma unroll
    for (int dd = 0; dd < DPW; ++dd) {
        const int d = dbase + wave + dd * NW;
        bv[dd] = bias ? bias[d < Cd ? d : Cd - 1] : 0.f;
        const float4* zi = reinterpret_cast<const float4*>(z + ((size_t)b * Cd + (d < Cd ? d : 0)) * M2);
#pragma unroll
        for (int k = 0; k < ZV; ++k) {
            const int i = lane + 64 * k;
            zv[dd][k] = zi[4 * i < M2 ? i : 0];
        }
    }
#pragma unroll
    for (int dd = 0; dd < WS; ++dd) {
        const int d = dbase + wave + dd * NW;
        wl[dd] = cfd_opaque_f(wl[dd]);
        wl[dd] = (lane < NW * NCH && dd < DPW && d < Cd && lane < Cs) ? wl[dd] : 0.f;
    }
#pragma unroll
    for (int dd = 0; dd < DPW; ++dd) {
        bv[dd] = cfd_opaque_f(bv[dd]);
        bv[dd] = dbase + wave + dd * NW < Cd ? bv[dd] : 0.f;
#pragma unroll
        for (int k = 0; k < ZV; ++k) {
            f32x4 t = {zv[dd][k].x, zv[dd][k].y, zv[dd][k].z, zv[dd][k].w};
            t = cfd_opaque_f4(t);
            zv[dd][k] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = threadIdx.x + k * 64 * NW;
        if (i < ntab) s_tab3[i] = tv[k];
    }
    if (lane < NW * NCH) {
#pragma unroll
        for (int h = 0; h < WS / 4; ++h)
            s_w[(wave * (NW * NCH) + lane) * (WS / 4) + h] = make_float4(wl[4 * h], wl[4 * h + 1], wl[4 * h + 2], wl[4 * h + 3]);
    }
    if constexpr (GEN) {
        if (threadIdx.x < 4 * 32) s_tw[threadIdx.x] = tailtab[threadIdx.x];
    }
#pragma unroll
    for (int dd = 0; dd < DPW; ++dd) {
        float4* zs = reinterpret_cast<float4*>(s_z + (wave * DPW + dd) * CFD_KB_ZS);
#pragma unroll
        for (int k = 0; k < ZV; ++k) {
            const int i = lane + 64 * k;
            if (4 * i < M2) zs[i] = zv[dd][k];
        }
        if (lane == 0) s_z[(wave * DPW + dd) * CFD_KB_ZS + M2] = 0.f;  // the zero word every masked-out gather index points at
    }
    cfd_wave_lds_sync();
    const bf16x8* tb3 = s_tab3 + CFD_TW * T * 64;
    float4 APV[2][4];  // gelu'(aprev) operands of two destination channels in flight; STEMG: the tile's u / v rows (in_chan planes)
    float4 SMK[STEMG ? 4 : 1];  // STEMG: the tile's mask rows
    float SG[STEMG ? DPW : 1][6];  // STEMG: this lane's share of sum g_0 * {1, u, v, mask, grid_x, grid_y} of its destination channels
    float gyv[4] = {0.f, 0.f, 0.f, 0.f}, gxr[4] = {0.f, 0.f, 0.f, 0.f};  // grid_y of the lane's columns; grid_x of the tile's rows 4q + r
    if constexpr (STEMG) {
#pragma unroll
        for (int dd = 0; dd < DPW; ++dd)
#pragma unroll
            for (int k = 0; k < 6; ++k) SG[dd][k] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) gyv[j] = sg.gy[4 * n + j];
    }
    auto fetch_feat = [&](int t, int which) {  // which: 0 = u, 1 = v, 2 = mask rows 16t + 4q + r, columns 4n .. 4n+3
        if constexpr (STEMG) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const size_t off = (size_t)rowc(16 * t + 4 * q + r4) * P + 4 * n;  // (GEN: rows past H clamped; their products are skipped)
                if (which < 2) {
                    const int ch = which < sg.in_chan ? which : 0;
                    APV[which][r4] = cfd_ldrow4<GEN>(sg.inputs + ((size_t)b * sg.in_chan + ch) * HW + off);
                } else {
                    SMK[r4] = sg.mask ? cfd_ldrow4<GEN>(sg.mask + (size_t)b * HW + off) : make_float4(1.f, 1.f, 1.f, 1.f);
                    gxr[r4] = sg.gx[rowc(16 * t + 4 * q + r4)];  // (clamped: a row past H must not read beyond the table -- 0 * NaN)
                }
            }
        }
    };
    auto fetch_ap = [&](int t, int dd, float4 (&r)[4]) {
        const int d = dbase + wave + dd * NW;
        const float* pl = aprev + ((size_t)b * Cd + (d < Cd ? d : 0)) * HW;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) r[r4] = cfd_ldrow4<GEN>(pl + (size_t)rowc(16 * t + 4 * q + r4) * P + 4 * n);
    };
    // One 16-row tile.  PAR = parity of the tile's first chunk index (compile time, so the prefetch registers and
    // LDS buffers are indexed statically even when NCH is odd).
    auto tile = [&](auto par_c, int tl) {
        constexpr int PAR = decltype(par_c)::value;
        const int t = t0 + tl;
        f32x4 acc[DPW][NJ];
        float tacc[GEN ? DPW : 1], ttail[GEN ? DPW : 1];  // GEN: tail column e = q of row 16t + n: channel mix (+ bias) | inverse transform
#pragma unroll
        for (int dd = 0; dd < DPW; ++dd) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[dd][j] = f32x4{bv[dd], bv[dd], bv[dd], bv[dd]};
            if constexpr (GEN) { tacc[dd] = bv[dd]; ttail[dd] = 0.f; }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int g = tl * NCH + c;
            const int par = (PAR + c) & 1;
            const int lb = SBUF ? 0 : par;  // LDS buffer of this chunk
            if constexpr (SBUF) __syncthreads();  // every wave is done with the previous chunk
            commit(c, lb, R[par], RT[par]);
            __syncthreads();  // chunk g visible (first pass: also tables, weights); the other buffer is free again
            if (g + 2 < G) fetch(g + 2, R[par], RT[par]);
            if constexpr (DGELU) {
                if (c == NCH - 2) fetch_ap(t, 0, APV[0]);
                if (c == NCH - 1) fetch_ap(t, 1, APV[1]);
            }
            if constexpr (STEMG) {  // the tile's feature rows, requested while the last chunks are mixed
                if (c == NCH - 2) { fetch_feat(t, 0); fetch_feat(t, 1); }
                if (c == NCH - 1) fetch_feat(t, 2);
            }
            // inverse transform of destination channel dd = c (spread over the chunks so MFMA and VALU work interleave)
            if (c < DPW && dbase + wave + c * NW < Cd) {
                float va[2][8];
                idft_gather(s_z + (wave * DPW + (c < DPW ? c : 0)) * CFD_KB_ZS, m1, m2, SA, q, n, va);
                idft_tile_b3<NJ, AP>(idft_split<AP>(va), s_tab3 + CFD_TW * t * 64, tb3, lane, acc[c < DPW ? c : 0], GEN ? s_tw : nullptr, E,
                                     GEN ? &ttail[c < DPW ? c : 0] : nullptr);
            }
            if constexpr (DPW > NCH) {  // more destination channels than chunks: the rest ride on the last chunk
                if (c == NCH - 1) {
#pragma unroll
                    for (int dd = NCH; dd < DPW; ++dd) {
                        if (dbase + wave + dd * NW < Cd) {
                            float va[2][8];
                            idft_gather(s_z + (wave * DPW + dd) * CFD_KB_ZS, m1, m2, SA, q, n, va);
                            idft_tile_b3<NJ, AP>(idft_split<AP>(va), s_tab3 + CFD_TW * t * 64, tb3, lane, acc[dd], GEN ? s_tw : nullptr, E,
                                                 GEN ? &ttail[dd] : nullptr);
                        }
                    }
                }
            }
            // channel mix of this chunk (dead channels hold zeros in LDS and get zero weights); the LDS reads of
            // source channel sl+1 are issued before the FMAs of channel sl
            {
                float4 v[2][4], wq[2][WS / 4];
                auto lds_fetch = [&](int sl, float4 (&vv)[4], float4 (&ww)[WS / 4]) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vv[r] = s_src[((lb * NW + sl) * 16 + 4 * q + r) * 16 + n];
#pragma unroll
                    for (int h = 0; h < WS / 4; ++h) ww[h] = s_w[(wave * (NW * NCH) + c * NW + sl) * (WS / 4) + h];
                };
                lds_fetch(0, v[0], wq[0]);
#pragma unroll 1
                for (int sl = 0; sl < NW; ++sl) {  // rolled: registers rotate, one channel of LDS reads in flight
                    const int sn = sl + 1 < NW ? sl + 1 : sl;
                    lds_fetch(sn, v[1], wq[1]);
                    float wd[WS];
#pragma unroll
                    for (int h = 0; h < WS / 4; ++h) {
                        wd[4 * h] = wq[0][h].x; wd[4 * h + 1] = wq[0][h].y; wd[4 * h + 2] = wq[0][h].z; wd[4 * h + 3] = wq[0][h].w;
                    }
#pragma unroll
                    for (int dd = 0; dd < DPW; ++dd) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            acc[dd][0][r] = fmaf(wd[dd], v[0][r].x, acc[dd][0][r]);
                            acc[dd][1][r] = fmaf(wd[dd], v[0][r].y, acc[dd][1][r]);
                            acc[dd][2][r] = fmaf(wd[dd], v[0][r].z, acc[dd][2][r]);
                            acc[dd][3][r] = fmaf(wd[dd], v[0][r].w, acc[dd][3][r]);
                        }
                    }
                    if constexpr (GEN) {  // tail column e = q of row 16t + n, source channel sl of this chunk
                        const float tv = s_tail[((lb * NW + sl) * 16 + n) * 4 + q];
#pragma unroll
                        for (int dd = 0; dd < DPW; ++dd) tacc[dd] = fmaf(wd[dd], tv, tacc[dd]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[0][r] = v[1][r];
#pragma unroll
                    for (int h = 0; h < WS / 4; ++h) wq[0][h] = wq[1][h];
                }
            }
        }
        // ---- tile epilogue: [* gelu'(aprev)], whole-float4 row stores ----
#pragma unroll
        for (int dd = 0; dd < DPW; ++dd) {
            const int d = dbase + wave + dd * NW;
            float4 ap[4];
            if constexpr (DGELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ap[r] = APV[dd & 1][r];
                if (dd + 2 < DPW) fetch_ap(t, dd + 2, APV[dd & 1]);
            }
            if constexpr (STEMG) {
                // g_0 is not stored: its products with the lifting layer's features are summed (dead channels hold zeros)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = !GEN || 16 * t + 4 * q + r < H;  // (GEN: the last row tile is ragged)
                    const float g0 = live ? acc[dd][0][r] : 0.f, g1 = live ? acc[dd][1][r] : 0.f, g2 = live ? acc[dd][2][r] : 0.f,
                                g3 = live ? acc[dd][3][r] : 0.f;
                    const float gs = (g0 + g1) + (g2 + g3);
                    const float4 uu = APV[0][r], vv = APV[1][r], mm = SMK[r];
                    SG[dd][0] += gs;
                    SG[dd][1] += fmaf(g0, uu.x, g1 * uu.y) + fmaf(g2, uu.z, g3 * uu.w);
                    SG[dd][2] += fmaf(g0, vv.x, g1 * vv.y) + fmaf(g2, vv.z, g3 * vv.w);
                    SG[dd][3] += fmaf(g0, mm.x, g1 * mm.y) + fmaf(g2, mm.z, g3 * mm.w);
                    SG[dd][4] = fmaf(gs, gxr[r], SG[dd][4]);
                    SG[dd][5] += fmaf(g0, gyv[0], g1 * gyv[1]) + fmaf(g2, gyv[2], g3 * gyv[3]);
                }
                if constexpr (GEN) {  // tail column e = q of row 16t + n (this lane's value of the 65th .. 68th column)
                    const int x = 16 * t + n;
                    if (q < E && x < H) {
                        const float gv = tacc[dd] + ttail[dd];
                        const size_t off = (size_t)x * P + W + q;
                        const float* pin = sg.inputs + (size_t)b * sg.in_chan * HW;
                        SG[dd][0] += gv;
                        SG[dd][1] = fmaf(gv, pin[off], SG[dd][1]);
                        SG[dd][2] = fmaf(gv, pin[(sg.in_chan > 1 ? (size_t)HW : 0) + off], SG[dd][2]);
                        SG[dd][3] = fmaf(gv, sg.mask ? sg.mask[(size_t)b * HW + off] : 1.f, SG[dd][3]);
                        SG[dd][4] = fmaf(gv, sg.gx[x], SG[dd][4]);
                        SG[dd][5] = fmaf(gv, sg.gy[W + q], SG[dd][5]);
                    }
                }
                continue;
            }
            if (d < Cd) {
                float* o = dst + ((size_t)b * Cd + d) * HW + (size_t)(16 * t + 4 * q) * P + 4 * n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float4 v = make_float4(acc[dd][0][r], acc[dd][1][r], acc[dd][2][r], acc[dd][3][r]);
                    if constexpr (DGELU) {
                        float4 gg = ap[r];
                        cfd_gelu_grad4(gg.x, gg.y, gg.z, gg.w);
                        v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
                    }
                    if (!GEN || 16 * t + 4 * q + r < H) cfd_strow4<GEN>(o + (size_t)r * P, v);
                }
                if constexpr (GEN) {  // tail column e = q of row 16t + n
                    const int x = 16 * t + n;
                    if (q < E && x < H) {
                        const size_t off = ((size_t)b * Cd + d) * HW + (size_t)x * P + W + q;
                        float v = tacc[dd] + ttail[dd];
                        if constexpr (DGELU) v *= cfd_gelu_grad(aprev[off]);
                        dst[off] = v;
                    }
                }
            }
        }
    };
#pragma unroll 1
    for (int t = 0; t < TPW; t += 2) {  // local tile index
        tile(CfdParity<0>{}, t);
        if (t + 1 < TPW) {
            if constexpr (NCH % 2 == 1) tile(CfdParity<1>{}, t + 1);
            else tile(CfdParity<0>{}, t + 1);
        }
    }
    if constexpr (STEMG) {  // one record per (entry [x row split], destination channel): the wave's six sums
#pragma unroll
        for (int dd = 0; dd < DPW; ++dd) {
            const int d = dbase + wave + dd * NW;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float tot = cfd_wave_sum(SG[dd][k]);
                if (lane == 0 && d < Cd) sg.part[((size_t)bi * Cd + d) * 6 + k] = tot;
            }
        }
    }
}

static bool block_is_gen(const cfd_plan* p) { return p->W != 64 || p->H % 16 != 0; }  // pitch != 64 or a ragged last row tile
// B = batch entries, bwd = the input-gradient direction (c != NULL: with gelu')
static bool block_fused_ok(const cfd_plan* p, int Cs, int Cd, const void* a, const void* b, const void* c, const void* z, int B, bool bwd) {
    const int cmax = Cs > Cd ? Cs : Cd;
    if (cfd_tune_get(CFD_TUNE_EXACT_FP32) == 1 || !p->d_inv_b3 || p->T > CFD_KB_TMAX) return false;  // (the fused kernel's inverse transform is split-bf16)
    if (4 * p->m1 * p->m2 + 1 > CFD_BLK_ZS || cmax > 32 || ((uintptr_t)z % 16) != 0) return false;
    // 25 .. 32 channels (the reference's default width 32).  Round 5: two (8,2,4) workgroups per entry re-read every source value from
    // LDS twice as often per output as one workgroup would and only beat the two passes for the gelu' input gradient on the general
    // grids.  Round 6: ONE (8,4,4) workgroup per entry with a SINGLE source-chunk buffer (138 KB instead of 168; k_block: SBUF) -- measured
    // against the two passes (kbench, us, width 32; fwd+gelu / input gradient / input gradient with gelu'):
    //   64 x 64   B = 256: 110 / 88 / 107 against 106 / 102 / 128     B = 64: 31 / 28 / 31 against 37 / 32 / 40
    //   66 x 65   B = 256: 139 / 109 / 162 against 146 / 143 / 174    B = 64: 50 / 43 / 58 against 44 / 41 / 49
    // and inside the train step (HBM-cold) the fused forward wins at B = 256 too: width-32 step at 64 x 64 2.20 ms on two passes, 2.07 with
    // only the backward fused, 1.98 with both.  So: 64-wide grids always, general grids from 128 entries (a 64-case rollout step at 66 x 65:
    // 357 us on two passes, 374 fused).  "block_wide": 0 = never, 1 = the round-5 pair of workgroups everywhere, 2 = the single one everywhere.
    if (cmax > 24) {
        const int wide = cfd_tune_get(CFD_TUNE_BLOCK_WIDE);
        if (wide == 0) return false;
        if (wide != 1 && wide != 2 && block_is_gen(p) && B < 128) return false;
        (void)bwd;
    }
    if (block_is_gen(p))  // round 4: 64 <= W <= 68 (d_inv_b3 exists), any H <= 80; 4-byte aligned planes (round 5: both piece counts)
        return cfd_tune_get(CFD_TUNE_BLOCK_GEN) != 0 && p->d_tail;
    return ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0 && (!c || ((uintptr_t)c % 16) == 0);
}

// one workgroup per batch entry (and destination group), tiles streamed inside; below ~3/4 of the CU count the entries are split by row
// tiles: workgroups per (entry, destination group)
static int block_row_splits(const cfd_plan* p, int B, int nd) {
    int spl = 1;
    while (spl * 2 <= p->T && p->T % (spl * 2) == 0 && (long)B * nd * spl * 2 <= 288) spl *= 2;
    if (p->T % 2 != 0) {  // an odd tile count (five at 66 x 65): the most workgroups per entry that still fit one round, no empty ones
        int want = 1;
        while (want < p->T && (long)B * nd * (want + 1) <= 288) ++want;
        const int tpw = (p->T + want - 1) / want;
        spl = (p->T + tpw - 1) / tpw;
    }
    return spl;
}

// Returns true when the lifting-layer-sum instantiation (STEMG) was the one launched: the caller REQUIRES that whenever it asked for
// the sums (the plain kernel would store g_0 and leave the sum records unwritten -- garbage fc0 gradients, no error).
template <int NW, int DPW, int NCH>
static bool launch_block_cfg(const cfd_plan* p, const float* src, const float* z, const float* w, const float* bias,
                             const float* aprev, float* dst, int B, int Cs, int Cd, int act, int trans, int dgelu,
                             hipStream_t st, const CfdReduceTail* tail, const CfdStemG* stemg = nullptr) {
    const bool ride = trans && tail && tail->nblk > 0;
    const CfdReduceTail tl = ride ? *tail : CfdReduceTail{};
    // destination groups per entry (k_block: ND): one unless the entry's destination channels exceed what a workgroup owns
    const int nd = (Cd + NW * DPW - 1) / (NW * DPW);
    const int spl = block_row_splits(p, B, nd);
    long nwg = (long)B * spl;
    if (nd > 1) nwg = (nwg + 7) / 8 * 8 * nd;  // whole groups of 8 * nd block indices (the kernel drops the padding)
    const dim3 grid((unsigned)(nwg + tl.nblk)), block(64 * NW);
    const bool ap3 = cfd_act_pieces() == 3, gen = block_is_gen(p);
    const CfdStemG sg0{};
#define CFD_BLK_P(A_, T_, D_, R_, P_, G_)                                                                                \
    hipLaunchKernelGGL((k_block<NW, DPW, NCH, A_, T_, D_, R_, P_, G_>), grid, block, 0, st, src, z, w, bias, aprev, dst, \
                       (const bf16x8*)p->d_inv_b3, Cs, Cd, p->H, p->m1, p->m2, p->T, p->SA, p->SB, tl, spl, p->W, (const float*)p->d_tail, nd, B * spl, sg0)
    // the lifting-layer sums (k_block<.., STEMG>): plain input gradient of a 64-wide grid, one destination group, at most 24 channels
    if (stemg && stemg->inputs && trans && !dgelu && !act && nd == 1) {
        if constexpr (NW * NCH <= 24 && NW != 10) {  // (not the wide (8,2,4) shape: two destination groups; not ten waves: registers)
#define CFD_BLK_S(R_, P_, G_)                                                                                                   \
    hipLaunchKernelGGL((k_block<NW, DPW, NCH, false, true, false, R_, P_, G_, true>), grid, block, 0, st, src, z, w, bias, aprev, dst, \
                       (const bf16x8*)p->d_inv_b3, Cs, Cd, p->H, p->m1, p->m2, p->T, p->SA, p->SB, tl, spl, p->W, (const float*)p->d_tail, nd, B * spl, *stemg)
#define CFD_BLK_SG(R_, P_) do { if (gen) CFD_BLK_S(R_, P_, true); else CFD_BLK_S(R_, P_, false); } while (0)
            if (ride) { if (ap3) CFD_BLK_SG(true, 3); else CFD_BLK_SG(true, 2); }
            else { if (ap3) CFD_BLK_SG(false, 3); else CFD_BLK_SG(false, 2); }
#undef CFD_BLK_SG
#undef CFD_BLK_S
            return true;
        }
    }
#define CFD_BLK(A_, T_, D_, R_)                                                      \
    do {                                                                             \
        if (gen) { if constexpr (NW != 10) { if (ap3) CFD_BLK_P(A_, T_, D_, R_, 3, true); else CFD_BLK_P(A_, T_, D_, R_, 2, true); } }  /* launch_block: never (10,2,2) */ \
        else if (ap3) CFD_BLK_P(A_, T_, D_, R_, 3, false);                           \
        else CFD_BLK_P(A_, T_, D_, R_, 2, false);                                    \
    } while (0)
    if (!trans) { if (act) CFD_BLK(true, false, false, false); else CFD_BLK(false, false, false, false); }
    else if (ride) { if (dgelu) CFD_BLK(false, true, true, true); else CFD_BLK(false, true, false, true); }
    else { if (dgelu) CFD_BLK(false, true, true, false); else CFD_BLK(false, true, false, false); }
#undef CFD_BLK
#undef CFD_BLK_P
    return false;
}

// (waves, destination channels per wave, source chunks): waves*DPW >= Cd and waves*NCH >= Cs
static bool launch_block(const cfd_plan* p, const float* src, const float* z, const float* w, const float* bias,
                         const float* aprev, float* dst, int B, int Cs, int Cd, int act, int trans, int dgelu,
                         hipStream_t st, const CfdReduceTail* tail = nullptr, const CfdStemG* stemg = nullptr) {
    const int cmax = Cs > Cd ? Cs : Cd;
    if (cmax <= 8) return launch_block_cfg<4, 2, 2>(p, src, z, w, bias, aprev, dst, B, Cs, Cd, act, trans, dgelu, st, tail, stemg);
    else if (cmax <= 16) return launch_block_cfg<4, 4, 4>(p, src, z, w, bias, aprev, dst, B, Cs, Cd, act, trans, dgelu, st, tail, stemg);
    else if (cmax <= 20) {  // measured at B=256, C=20 (us): (10,2,2) 61/73/60/97, (5,4,4) 69/79/67/82, (4,5,5) 77/84/74/83
        // round 2, same box: (8,3,3) 52.8/59.3/53.5/62.2 against 49.4/60.9/50.9/66.8 -- two waves per SIMD, 3 + 2 destination
        // channels per SIMD; (10,2,2) and (5,4,4) leave SIMDs with 3 vs 2 and 2 vs 1 waves
        // general grids (pitch != 64, five row tiles): (10,2,2) would need 161 KB of LDS with the fifth tile's tables and the tail planes
        // (the lifting-layer sums need ~60 more registers: ten waves -- three per SIMD, 168 registers -- would spill 69 of them)
        if (dgelu || block_is_gen(p) || (stemg && stemg->inputs)) return launch_block_cfg<8, 3, 3>(p, src, z, w, bias, aprev, dst, B, Cs, Cd, act, trans, dgelu, st, tail, stemg);
        else return launch_block_cfg<10, 2, 2>(p, src, z, w, bias, aprev, dst, B, Cs, Cd, act, trans, dgelu, st, tail, stemg);
    }
    // 21 .. 24 channels: (8,3,3).
    else if (cmax <= 24) return launch_block_cfg<8, 3, 3>(p, src, z, w, bias, aprev, dst, B, Cs, Cd, act, trans, dgelu, st, tail, stemg);
    // 25 .. 32 channels (round 5; the reference's default width 32, src/args.py:190): with the tables in three pieces (CFD_TW) an
    // (8,4,4) workgroup would need 168 KB of LDS (source chunks 64 + modes 74 + tables 24 + weights 4), so the entry's destination
    // channels are dealt to TWO (8,2,4) workgroups of 16 channels (modes 37 KB: 133 KB), each streaming all the source channels.
    // round 6: ONE (8,4,4) workgroup per entry with a single source-chunk buffer (138 KB); block_wide = 1: the round-5 pair
    else if (cfd_tune_get(CFD_TUNE_BLOCK_WIDE) == 1) return launch_block_cfg<8, 2, 4>(p, src, z, w, bias, aprev, dst, B, Cs, Cd, act, trans, dgelu, st, tail);
    else return launch_block_cfg<8, 4, 4>(p, src, z, w, bias, aprev, dst, B, Cs, Cd, act, trans, dgelu, st, tail);
}

// out[b,o] = bias[o] + sum_i w0[o,i] f(a[b,i]) + idft(z[b,o])                       (FnoBlock.forward minus its GELU)
extern "C" int cfd_fno_block_fwd(const cfd_plan* p, const float* a, const float* z, const float* w0, const float* b0,
                                 float* out, int B, int Cin, int Cout, int act_in, void* stream) {
    CFD_REQUIRE(p && a && z && w0 && out, CFD_ERR_INVALID_ARG, "cfd_fno_block_fwd: NULL pointer");
    CFD_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && Cin <= 32 && Cout <= 32, CFD_ERR_UNSUPPORTED,
                "cfd_fno_block_fwd: channels (%d -> %d) unsupported (1..32)", Cin, Cout);
    if (B == 0) return CFD_OK;
    hipStream_t st = (hipStream_t)stream;
    if (!block_fused_ok(p, Cin, Cout, a, out, nullptr, z, B, false)) {  // general grids: two passes
        CFD_TRY(cfd_chanmix(a, w0, b0, out, B, Cin, Cout, p->H * p->W, act_in, 0, stream));
        return cfd_spectral_idft(p, z, out, nullptr, out, B * Cout, 1, stream);
    }
    CFD_PROF_W(act_in ? "k_block_fwd_act" : "k_block_fwd", st, (double)B * (4.0 * p->H * p->W * (Cin + Cout) + 16.0 * p->m1 * p->m2 * Cout),
               2.0 * B * p->H * p->W * (double)Cin * Cout);
    launch_block(p, a, z, w0, b0, nullptr, out, B, Cin, Cout, act_in, 0, 0, st);
    CFD_LAUNCH_CHECK("cfd_fno_block_fwd");
    return CFD_OK;
}

// gin[b,i] = (sum_o w0[o,i] g[b,o] + idft(gz[b,i])) * (aprev ? gelu'(aprev[b,i]) : 1)      (input gradient of the block)
// `tail` (may be NULL): reductions to carry in this launch.  Where the fused kernel does not apply (general grids) or
// the batch is empty they are launched as the kernels they replace, so the caller's gradients are final on return
// either way.
static int launch_reduce_tail_standalone(const CfdReduceTail* tail, hipStream_t st) {
    if (!tail || tail->nblk <= 0) return CFD_OK;
    if (tail->spec.part) {
        const long total = (long)tail->spec.CC * 2 * tail->spec.m1 * tail->spec.m2;
        CFD_PROF_W("k_spec_wgrad_reduce", st, 0.0, 0.0);
        hipLaunchKernelGGL(k_spec_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, tail->spec.part,
                           tail->spec.gw1, tail->spec.gw2, tail->spec.clhw, tail->spec.nchunk, tail->spec.CC, tail->spec.m1,
                           tail->spec.m2);
        CFD_LAUNCH_CHECK("cfd_fno_block_bwd_input(spectral reduce)");
    }
    if (tail->chan.part) {
        CFD_PROF_W("k_wgrad_reduce", st, 0.0, 0.0);
        hipLaunchKernelGGL(k_chan_reduce_standalone, dim3((tail->chan.Co * (tail->chan.Ci + 1) + 15) / 16), dim3(256), 0, st,
                           tail->chan);
        CFD_LAUNCH_CHECK("cfd_fno_block_bwd_input(1x1 reduce)");
    }
    if (tail->head.part) {
        CFD_PROF_W("k_head_reduce", st, 0.0, 0.0);
        hipLaunchKernelGGL(k_head_reduce_standalone, dim3((tail->head.PS + 15) / 16), dim3(256), 0, st, tail->head);
        CFD_LAUNCH_CHECK("cfd_fno_block_bwd_input(head reduce)");
    }
    return CFD_OK;
}

// whether cfd_int_fno_block_bwd_input will run the fused kernel (the one that carries tail jobs) for this plan / width / buffers
bool cfd_int_block_bwd_fused(const cfd_plan* p, int B, int C, const void* g, const void* gin, const void* aprev, const void* gz) {
    return block_fused_ok(p, C, C, g, gin, aprev, gz, B, true);
}

// ---- lifting-layer gradient from the sums of k_block<.., STEMG> (cfd_tail.h: CfdStemG) ------------------------------------------
// gw[c][f], f = (in_chan field channels, mask, grid_x, grid_y, P case parameters), and gb[c] from part[rec][c][6], rec = batch entry x row
// split: one workgroup per channel, thread-strided over the records in a fixed order, LDS tree; the case-parameter columns are
// sum_b cp[b][k] * S0[b][c] (a case parameter is constant over an entry's pixels).
__global__ __launch_bounds__(256) void k_stem_grad_combine(const float* __restrict__ part, int nrec, int spl, const float* __restrict__ cp,
                                                           int P, int in_chan, int C, float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float s_r[CFD_STEMG_NA][256];
    const int c = blockIdx.x;
    stem_combine_channel(part, nrec, spl, cp, P, C, c, s_r);  // (cfd_tail.h: shared with cfd_fno_adam_step's carrying kernel)
    const int F = in_chan + 3 + P;
    if ((int)threadIdx.x < F) gw[(size_t)c * F + threadIdx.x] = s_r[stem_feature_slot(threadIdx.x, in_chan)][0];
    if (threadIdx.x == 0) gb[c] = s_r[0][0];
}

// whether the input gradient of FnoBlock 0 can emit the lifting layer's sums instead of storing g_0 (same predicate in both phases)
bool cfd_int_stemg_ok(const cfd_plan* p, int B, int C, int in_chan, int P, const void* inputs, const void* mask, const void* z) {
    if (cfd_tune_get(CFD_TUNE_STEM_FUSE) == 0) return false;
    if (!p || C > 24 || in_chan < 1 || in_chan > 2 || P > 8 || B < 1) return false;
    if (cfd_tune_get(CFD_TUNE_EXACT_FP32) == 1 || !p->d_inv_b3 || p->T > CFD_KB_TMAX || 4 * p->m1 * p->m2 + 1 > CFD_BLK_ZS) return false;
    if (((uintptr_t)z % 16) != 0) return false;
    if (block_is_gen(p))  // general grids (66 x 65): built and tested, but OFF unless stem_fuse = 3 -- with the ragged tile and the tail column
                          // the (8,3,3) general kernel spills and takes 122 us against 75 + 40 for the plain kernel + the weight-gradient
                          // pass (train step 1.637 against 1.617 ms at 66 x 65, B = 256)
        return cfd_tune_get(CFD_TUNE_STEM_FUSE) == 3 && cfd_tune_get(CFD_TUNE_BLOCK_GEN) != 0 && p->d_tail;
    return ((uintptr_t)inputs % 16) == 0 && (!mask || ((uintptr_t)mask % 16) == 0);
}
size_t cfd_int_stemg_part_bytes(const cfd_plan* p, int B, int C) { return (size_t)B * (p ? p->T : 1) * C * 6 * sizeof(float); }

int cfd_int_stemg_splits(const cfd_plan* p, int B) { return block_row_splits(p, B, 1); }  // records per batch entry

int cfd_int_stemg_combine(const cfd_plan* p, const float* part, const float* cp, float* gw, float* gb, int B, int C, int in_chan, int P,
                          void* stream) {
    const int spl = block_row_splits(p, B, 1);
    CFD_PROF_W("k_stem_grad_combine", (hipStream_t)stream, 0.0, 0.0);
    hipLaunchKernelGGL(k_stem_grad_combine, dim3(C), dim3(256), 0, (hipStream_t)stream, part, B * spl, spl, cp, P, in_chan, C, gw, gb);
    CFD_LAUNCH_CHECK("cfd_fno_stem_bwd(combine)");
    return CFD_OK;
}

int cfd_int_fno_block_bwd_input(const cfd_plan* p, const float* g, const float* gz, const float* w0, const float* aprev,
                                float* gin, int B, int Cin, int Cout, void* stream, const CfdReduceTail* tail, const CfdStemG* stemg) {
    CFD_REQUIRE(p && g && gz && w0 && gin, CFD_ERR_INVALID_ARG, "cfd_fno_block_bwd_input: NULL pointer");
    CFD_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && Cin <= 32 && Cout <= 32, CFD_ERR_UNSUPPORTED,
                "cfd_fno_block_bwd_input: channels (%d -> %d) unsupported (1..32)", Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return launch_reduce_tail_standalone(tail, st);
    if (!block_fused_ok(p, Cout, Cin, g, gin, aprev, gz, B, true)) {
        CFD_REQUIRE(!(stemg && stemg->inputs), CFD_ERR_UNSUPPORTED, "cfd_fno_block_bwd_input: the lifting-layer sums need the fused kernel (cfd_int_stemg_ok)");
        CFD_TRY(launch_reduce_tail_standalone(tail, st));
        CFD_TRY(cfd_chanmix(g, w0, nullptr, gin, B, Cout, Cin, p->H * p->W, 0, 1, stream));
        return cfd_spectral_idft(p, gz, gin, aprev, gin, B * Cin, aprev ? 2 : 1, stream);
    }
    CFD_REQUIRE(!(stemg && stemg->inputs && aprev), CFD_ERR_INVALID_ARG, "cfd_fno_block_bwd_input: the lifting-layer sums ride in the plain input gradient");
    const bool sums = stemg && stemg->inputs;  // (reads g and the entry's feature planes, writes no gradient tensor)
    CFD_PROF_W(sums ? "k_block_bwd_stem" : (aprev ? "k_block_bwd_dgelu" : "k_block_bwd"), st,
               (double)B * (4.0 * p->H * p->W * ((sums ? 3 : Cin) + Cout + (aprev ? Cin : 0)) + 16.0 * p->m1 * p->m2 * Cin), 2.0 * B * p->H * p->W * (double)Cin * Cout);
    const bool emitted = launch_block(p, g, gz, w0, nullptr, aprev, gin, B, Cout, Cin, 0, 1, aprev ? 1 : 0, st, tail, stemg);
    CFD_LAUNCH_CHECK("cfd_fno_block_bwd_input");
    CFD_REQUIRE(emitted == sums, CFD_ERR_UNSUPPORTED, "cfd_fno_block_bwd_input: the lifting-layer sums were requested but this shape (Cin %d, Cout %d) has no "
                "k_block<.., STEMG> instantiation -- cfd_int_stemg_ok and the launch table are out of step", Cin, Cout);
    return CFD_OK;
}

extern "C" int cfd_fno_block_bwd_input(const cfd_plan* p, const float* g, const float* gz, const float* w0,
                                       const float* aprev, float* gin, int B, int Cin, int Cout, void* stream) {
    return cfd_int_fno_block_bwd_input(p, g, gz, w0, aprev, gin, B, Cin, Cout, stream, nullptr, nullptr);
}
