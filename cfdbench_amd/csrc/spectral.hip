// SpectralConv2d_fast (src/models/fno/fno2d.py:17-82) as pruned DFTs on the CDNA4 matrix pipe.
//
// Only rows K = [0,m1) U [H-m1,H) and columns [0,m2) of rfft2(x) are ever consumed (fno2d.py:73-78), i.e. 288 of
// 2112 bins at 64x64 / modes 12, so both transforms are dense truncated DFTs evaluated with
// v_mfma_f32_16x16x4_f32 (exact fp32).  One 64-lane wave owns one (batch, channel) image end to end; the
// chained-MFMA operand maps below are arranged so that the accumulator of the first stage IS the operand of
// the second (no LDS round trip for data, only the constant operator tables live in LDS).
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

#define CFD_WAVES 4  // waves per workgroup (256 threads)

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
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) s_tab[i] = tabs[i];
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
                if constexpr (ACT) { vv = cfd_gelu(vv); uu = cfd_gelu(uu); }  // gelu(0) = 0 keeps the zero fill
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
        float2* o = xh + (size_t)img * M;
        const int kap = n;
        if (kap <= m1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int l = 4 * q + r;
                if (l < m2) {
                    if (kap < m1) o[kap * m2 + l] = make_float2(Pc[r] - Qs[r], -(Ps[r] + Qc[r]));
                    if (kap >= 1) o[(2 * m1 - kap) * m2 + l] = make_float2(Pc[r] + Qs[r], Qc[r] - Ps[r]);
                }
            }
        }
    }
}

template <int NJ, bool VEC4>
static int launch_dft(const cfd_plan* p, const float* x, float* xh, int nimg, int act, hipStream_t st) {
    int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
    if (blocks > 2048) blocks = 2048;
    CFD_PROF(act ? "k_dft_fwd_act" : "k_dft_fwd", st);
    if (act)
        hipLaunchKernelGGL((k_dft_fwd<NJ, VEC4, true>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, x, (float2*)xh,
                           (const float*)p->d_fwd, p->n_fwd, nimg, p->H, p->W, p->m1, p->m2, p->KX);
    else
        hipLaunchKernelGGL((k_dft_fwd<NJ, VEC4, false>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, x, (float2*)xh,
                           (const float*)p->d_fwd, p->n_fwd, nimg, p->H, p->W, p->m1, p->m2, p->KX);
    CFD_LAUNCH_CHECK("cfd_spectral_dft");
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
// mode mixing  (compl_mul2d, fno2d.py:54-57) and its input-adjoint
// ------------------------------------------------------------------------------------------------------
// One thread owns BB batch entries of one (output channel, mode): the weight is loaded once per BB complex MACs.
template <int BB, bool CONJT>
__global__ __launch_bounds__(256) void k_mix(const float2* __restrict__ xin, const float2* __restrict__ w1,
                                             const float2* __restrict__ w2, float2* __restrict__ z, int B, int Cr,
                                             int Cz, int CoutW, int m1, int m2) {
    const int M = 2 * m1 * m2, half = m1 * m2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nbg = (B + BB - 1) / BB;
    if (gid >= (long)nbg * Cz * M) return;
    const int mode = (int)(gid % M);
    const int cz = (int)((gid / M) % Cz);
    const int b0 = (int)(gid / ((long)M * Cz)) * BB;
    const float2* w = mode < half ? w1 : w2;
    const int wm = mode < half ? mode : mode - half;
    float2 acc[BB];
#pragma unroll
    for (int k = 0; k < BB; ++k) acc[k] = make_float2(0.f, 0.f);
    for (int cr = 0; cr < Cr; ++cr) {
        // weights are (Cin_w, Cout_w, m1, m2): forward reduces over Cin_w (= cr), adjoint over Cout_w (= cr)
        const float2 wv = CONJT ? w[((size_t)cz * CoutW + cr) * half + wm] : w[((size_t)cr * CoutW + cz) * half + wm];
        const float wr = wv.x, wi = CONJT ? -wv.y : wv.y;
#pragma unroll
        for (int k = 0; k < BB; ++k) {
            if (b0 + k < B) {
                const float2 xv = xin[((size_t)(b0 + k) * Cr + cr) * M + mode];
                acc[k].x = fmaf(xv.x, wr, fmaf(-xv.y, wi, acc[k].x));
                acc[k].y = fmaf(xv.x, wi, fmaf(xv.y, wr, acc[k].y));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < BB; ++k)
        if (b0 + k < B) z[((size_t)(b0 + k) * Cz + cz) * M + mode] = acc[k];
}

extern "C" int cfd_spectral_mix(const cfd_plan* p, const float* xh, const float* w1, const float* w2, float* z, int B,
                                int Cin, int Cout, int conj_t, void* stream) {
    CFD_REQUIRE(p && xh && w1 && w2 && z, CFD_ERR_INVALID_ARG, "cfd_spectral_mix: NULL pointer");
    CFD_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1, CFD_ERR_INVALID_ARG, "cfd_spectral_mix: bad sizes");
    if (B == 0) return CFD_OK;
    const int M = 2 * p->m1 * p->m2;
    constexpr int BB = 4;
    const int Cr = conj_t ? Cout : Cin, Cz = conj_t ? Cin : Cout;
    const long total = (long)((B + BB - 1) / BB) * Cz * M;
    const int blocks = (int)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    CFD_PROF(conj_t ? "k_mix_adj" : "k_mix", st);
    if (conj_t)
        hipLaunchKernelGGL((k_mix<BB, true>), dim3(blocks), dim3(256), 0, st, (const float2*)xh, (const float2*)w1,
                           (const float2*)w2, (float2*)z, B, Cr, Cz, Cout, p->m1, p->m2);
    else
        hipLaunchKernelGGL((k_mix<BB, false>), dim3(blocks), dim3(256), 0, st, (const float2*)xh, (const float2*)w1,
                           (const float2*)w2, (float2*)z, B, Cr, Cz, Cout, p->m1, p->m2);
    CFD_LAUNCH_CHECK("cfd_spectral_mix");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// spectral weight gradient: gw[i,o,mode] = sum_b conj(xh[b,i,mode]) * (c_l/HW) * gh[b,o,mode]
// ------------------------------------------------------------------------------------------------------
#define CFD_WGRAD_BCHUNK 32

__global__ __launch_bounds__(256) void k_spec_wgrad_part(const float2* __restrict__ xh, const float2* __restrict__ gh,
                                                         float2* __restrict__ part, int B, int Cin, int Cout, int M) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)Cin * Cout * M) return;
    const int mode = (int)(gid % M);
    const int o = (int)((gid / M) % Cout);
    const int i = (int)(gid / ((long)M * Cout));
    const int b0 = blockIdx.y * CFD_WGRAD_BCHUNK;
    const int b1 = (b0 + CFD_WGRAD_BCHUNK < B) ? b0 + CFD_WGRAD_BCHUNK : B;
    float ar = 0.f, ai = 0.f;
    for (int b = b0; b < b1; ++b) {
        const float2 xv = xh[((size_t)b * Cin + i) * M + mode];
        const float2 gv = gh[((size_t)b * Cout + o) * M + mode];
        // conj(x) * g
        ar = fmaf(xv.x, gv.x, fmaf(xv.y, gv.y, ar));
        ai = fmaf(xv.x, gv.y, fmaf(-xv.y, gv.x, ai));
    }
    part[(size_t)blockIdx.y * Cin * Cout * M + gid] = make_float2(ar, ai);
}

__global__ __launch_bounds__(256) void k_spec_wgrad_reduce(const float2* __restrict__ part, float2* __restrict__ gw1,
                                                           float2* __restrict__ gw2, const float* __restrict__ clhw,
                                                           int nchunk, int CC, int m1, int m2) {
    const int M = 2 * m1 * m2, half = m1 * m2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)CC * M) return;
    const int mode = (int)(gid % M);
    const long io = gid / M;
    float ar = 0.f, ai = 0.f;
    for (int c = 0; c < nchunk; ++c) {
        const float2 v = part[(size_t)c * CC * M + gid];
        ar += v.x;
        ai += v.y;
    }
    const float sc = clhw[mode % m2];
    const float2 r = make_float2(ar * sc, ai * sc);
    if (mode < half) gw1[io * half + mode] = r;
    else gw2[io * half + (mode - half)] = r;
}

extern "C" size_t cfd_spectral_wgrad_workspace_bytes(const cfd_plan* p, int B, int Cin, int Cout) {
    if (!p || B <= 0) return 0;
    const size_t nchunk = (B + CFD_WGRAD_BCHUNK - 1) / CFD_WGRAD_BCHUNK;
    return nchunk * (size_t)Cin * Cout * 2 * p->m1 * p->m2 * sizeof(float2);
}

extern "C" int cfd_spectral_wgrad(const cfd_plan* p, const float* xh, const float* gh, float* gw1, float* gw2, void* ws,
                                  int B, int Cin, int Cout, void* stream) {
    CFD_REQUIRE(p && xh && gh && gw1 && gw2 && ws, CFD_ERR_INVALID_ARG, "cfd_spectral_wgrad: NULL pointer");
    CFD_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1, CFD_ERR_INVALID_ARG, "cfd_spectral_wgrad: bad sizes");
    const int M = 2 * p->m1 * p->m2;
    const int nchunk = (B + CFD_WGRAD_BCHUNK - 1) / CFD_WGRAD_BCHUNK;
    const long total = (long)Cin * Cout * M;
    hipStream_t st = (hipStream_t)stream;
    {
        CFD_PROF("k_spec_wgrad_part", st);
        hipLaunchKernelGGL(k_spec_wgrad_part, dim3((unsigned)((total + 255) / 256), nchunk), dim3(256), 0, st,
                           (const float2*)xh, (const float2*)gh, (float2*)ws, B, Cin, Cout, M);
    }
    CFD_LAUNCH_CHECK("cfd_spectral_wgrad(part)");
    CFD_PROF("k_spec_wgrad_reduce", st);
    hipLaunchKernelGGL(k_spec_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const float2*)ws, (float2*)gw1, (float2*)gw2, (const float*)p->d_clhw, nchunk, Cin * Cout,
                       p->m1, p->m2);
    CFD_LAUNCH_CHECK("cfd_spectral_wgrad(reduce)");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// inverse pruned DFT (+ fused epilogue)
// ------------------------------------------------------------------------------------------------------
template <int NJ, bool VEC4, int EPI>
__global__ __launch_bounds__(64 * CFD_WAVES) void k_idft(const float* __restrict__ z, const float* addend,
                                                          const float* __restrict__ aprev, float* out,
                                                          const float* __restrict__ tabs, int ntab, int nimg, int H,
                                                          int W, int m1, int m2, int T, int SA, int SB) {
    __shared__ float s_tab[(8 * 8 + 8 * NJ) * 64];  // T<=8, SA<=8, SB<=8
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) s_tab[i] = tabs[i];
    __syncthreads();
    const float* ta = s_tab;
    const float* tb = ta + T * SA * 64;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int M = 2 * m1 * m2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};

    // Stage-A data operand gather map (lane = (q, rho)): row rho of M-tile mu is c = 4*(4*mu + (rho&3)) + (rho>>2)
    // so that accumulator register r of tile mu is exactly the stage-B operand of k-step 4*mu + r.
    // k-steps s < 4 are the cos block (kappa = 4s+q, value S = Z+ + Z-); s >= 4 the sin block (kappa = 4s+q-15,
    // value -D_im for a real-part row, +D_re for an imaginary-part row, D = Z+ - Z-).
    int offP[2][8], offM[2][8];
    float sg[2];
#pragma unroll
    for (int mu = 0; mu < 2; ++mu) {
        const int c = 4 * (4 * mu + (n & 3)) + (n >> 2);
        const bool cvalid = c < 2 * m2;
        const int part = c >= m2 ? 1 : 0;
        const int l = c - part * m2;
        sg[mu] = part ? 1.f : -1.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int kap = s < 4 ? 4 * s + q : 4 * s + q - 15;
            const bool kvalid = (s < SA) && kap <= m1 && cvalid;
            const int comp = s < 4 ? part : 1 - part;
            offP[mu][s] = (kvalid && kap < m1) ? (kap * m2 + l) * 2 + comp : -1;
            offM[mu][s] = (kvalid && kap >= 1) ? ((2 * m1 - kap) * m2 + l) * 2 + comp : -1;
        }
    }

    for (int img = blockIdx.x * CFD_WAVES + wave; img < nimg; img += gridDim.x * CFD_WAVES) {
        const float* zi = z + (size_t)img * M * 2;
        float va[2][8];
#pragma unroll
        for (int mu = 0; mu < 2; ++mu) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float vp = offP[mu][s] >= 0 ? zi[offP[mu][s]] : 0.f;
                const float vm = offM[mu][s] >= 0 ? zi[offM[mu][s]] : 0.f;
                va[mu][s] = s < 4 ? vp + vm : sg[mu] * (vp - vm);
            }
        }
        const size_t ibase = (size_t)img * H * W;
        for (int t = 0; t < T; ++t) {
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
            f32x4 accB[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) accB[j] = zero;
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                if (sp < SB) {
                    const float av = accA[sp >> 2][sp & 3];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) accB[j] = cfd_mfma16x16x4(av, tb[(sp * NJ + j) * 64 + lane], accB[j]);
                }
            }
            // accB[j][r]: x = 16t + 4q + r, y = NJ*n + j
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int xx = 16 * t + 4 * q + r;
                if (xx < H) {
                    const size_t rbase = ibase + (size_t)xx * W;
                    if constexpr (VEC4) {
                        if (4 * n < W) {
                            float4 v = make_float4(accB[0][r], accB[1][r], accB[2][r], accB[3][r]);
                            if constexpr (EPI >= 1) {
                                const float4 ad = *reinterpret_cast<const float4*>(addend + rbase + 4 * n);
                                v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
                            }
                            if constexpr (EPI == 2) {
                                const float4 ap = *reinterpret_cast<const float4*>(aprev + rbase + 4 * n);
                                v.x *= cfd_gelu_grad(ap.x); v.y *= cfd_gelu_grad(ap.y);
                                v.z *= cfd_gelu_grad(ap.z); v.w *= cfd_gelu_grad(ap.w);
                            }
                            *reinterpret_cast<float4*>(out + rbase + 4 * n) = v;
                        }
                    } else {
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
}

template <int NJ, bool VEC4>
static int launch_idft(const cfd_plan* p, const float* z, const float* addend, const float* aprev, float* out,
                       int nimg, int epi, hipStream_t st) {
    int blocks = (nimg + CFD_WAVES - 1) / CFD_WAVES;
    if (blocks > 2048) blocks = 2048;
#define CFD_IDFT_LAUNCH(E)                                                                                         \
    hipLaunchKernelGGL((k_idft<NJ, VEC4, E>), dim3(blocks), dim3(64 * CFD_WAVES), 0, st, z, addend, aprev, out,     \
                       (const float*)p->d_inv, p->n_inv, nimg, p->H, p->W, p->m1, p->m2, p->T, p->SA, p->SB)
    CFD_PROF(epi == 0 ? "k_idft" : (epi == 1 ? "k_idft_add" : "k_idft_add_dgelu"), st);
    if (epi == 0) CFD_IDFT_LAUNCH(0);
    else if (epi == 1) CFD_IDFT_LAUNCH(1);
    else CFD_IDFT_LAUNCH(2);
#undef CFD_IDFT_LAUNCH
    CFD_LAUNCH_CHECK("cfd_spectral_idft");
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
    if (gw1) CFD_TRY(cfd_spectral_wgrad(p, xh, gh, gw1, gw2, wws, B, Cin, Cout, stream));
    if (gx) {
        CFD_TRY(cfd_spectral_mix(p, gh, w1, w2, gz, B, Cin, Cout, 1, stream));
        CFD_TRY(cfd_spectral_idft(p, gz, nullptr, nullptr, gx, B * Cin, 0, stream));
    }
    return CFD_OK;
}
