// Mode-domain contractions of SpectralConv2d on the fp32 matrix pipe (round 6): compl_mul2d (src/models/fno/fno2d.py:54-57), its
// input adjoint and the spectral weight gradient (what autograd derives from fno2d.py:73-78) as small REAL GEMMs per kept mode.
//
// Per mode the complex (B x C) . (C x C) product is the real (B x 2C) . (2C x 2C) product with the weight block
// [[Wr, Wi], [-Wi, Wr]] (conjugate-transposed for the adjoint), K = 2C = 40, N = 40 -> three 16-column tiles; the weight gradient
// gw[i,o] = sum_b conj(x[b,i]) g[b,o] is the (2C x B) . (B x 2C) product D[(i,ri)][(o,ro)] = sum_b X[b,i,ri] G[b,o,ro] with
// gw.re = D[(i,0)][(o,0)] + D[(i,1)][(o,1)], gw.im = D[(i,0)][(o,1)] - D[(i,1)][(o,0)].  v_mfma_f32_16x16x4_f32 is an exact fp32
// fmaf chain (cfd_intrinsics.h), so the results are fp32-exact class like the VALU kernels of spectral.hip they replace.
//
// Workgroup = (tile of 8 consecutive modes, chunk of <= RMAX batch entries), 8 waves, wave w owns mode 8 tile + w.  The operands
// keep the reference layout (B, C, 2 m1, m2) complex64: the 8 modes of one (entry, channel) are one 64-byte piece, a workgroup pulls
// its pieces with 16-byte accesses into LDS rows [entry][channel: 18 floats = 16 + 2 pad] (row pitch 18 C + 4: the 16 entry lanes x
// 2 k-lanes of a ds_read_b64 hit 64 different banks), every wave reads the slots of ITS mode in MFMA operand order, and results go back
// through the same rows (in place: a wave touches only its own mode's slots) and out in 64-byte pieces.  The two mode tiles that share
// a 128-byte line run on the same XCD back to back (workgroup ids L and L + 8).  All global loads of a workgroup are issued up front;
// the entries are processed in stages of 16 (one MFMA row tile) as they land.  The weight-gradient partial sums of a chunk leave
// through LDS in the layout k_spec_wgrad_reduce / cfd_tail.h already read: part[chunk][i][o][mode] complex.
#include "cfd_common.h"
#include "cfd_tail.h"

#ifndef CFD_MM_EXP
#define CFD_MM_EXP 0  // timing experiments (tools/build_variant.sh -DCFD_MM_EXP=..): 1 no mix MFMAs, 2 no wgrad MFMAs, 4 no weight staging, 16 no z stores
#endif
#define CFD_MM_T 8    // modes per tile = waves per workgroup
#define CFD_MM_CP 18  // floats of one (entry, channel) piece in LDS: 16 data + 2 pad

template <int C>
static constexpr int cfd_mm_bp() { return C * CFD_MM_CP + 4; }  // row pitch in floats
// RMAX = LDS rows (batch entries) of one chunk: 32 for the mix alone at 20 channels (72 KB: two workgroups per CU), 40 with the weight
// gradient (142 KB), 37 for the mix alone at 32 channels (156 KB: 256 / 7 entries per chunk = 252 workgroups, one per CU)
template <int C>
static size_t cfd_mm_lds_bytes(bool wgrad, int rmax) {
    return ((size_t)(wgrad ? 2 : 1) * rmax * cfd_mm_bp<C>() + (size_t)C * C * CFD_MM_CP) * sizeof(float);
}

// xin: the contracted operand of the mix (forward: xh, adjoint: gh); xw (WGRAD): xh, the conjugated side of the weight gradient.
template <int C, bool CONJT, bool WGRAD, int NST, int RMAX>
__global__ __launch_bounds__(512) void k_modes_mfma(const float* __restrict__ xin, const float* __restrict__ xw,
                                                    const float* __restrict__ w1, const float* __restrict__ w2,
                                                    float* __restrict__ z, float* __restrict__ part, int B, int BC, int M, int half,
                                                    int ntile, int npair, int nchunk) {
    constexpr int BP = cfd_mm_bp<C>(), CP = CFD_MM_CP;
    constexpr int NT = (2 * C + 15) / 16;              // 16-wide tiles of the (channel, re/im) index
    constexpr int KS = C / 2;                          // K = 2C in steps of 4
    constexpr int SF4 = 16 * 4 * C;                    // 16-byte vectors of one stage of one operand
    constexpr int NLD = (SF4 + 511) / 512;             // ... per thread
    constexpr int WF4 = C * C * 4;                     // 16-byte vectors of the tile's weights
    constexpr int NWL = (WF4 + 511) / 512;
    static_assert(C % 4 == 0, "k_modes_mfma: channels in K-steps of four");
    static_assert(16 * NST >= RMAX, "k_modes_mfma: stages cover the LDS rows");
    static_assert(!WGRAD || RMAX % 4 == 0, "k_modes_mfma: the weight gradient walks the rows in K-steps of four");
    CFD_DYN_SHARED(f32x4, smem4);  // (declared in 16-byte units: the weight rows move as whole vectors)
    float* smem = reinterpret_cast<float*>(smem4);
    float* GS = smem;                                  // rows of the mix operand (results written back in place)
    float* XS = smem + RMAX * BP;                      // WGRAD: rows of xw
    float* WS = smem + (WGRAD ? 2 : 1) * RMAX * BP;    // [C*C][8 modes][re, im] at the same 18-float pitch: weights in, weight-gradient partials out
                                                       // (16-float pieces put the eight channels of an operand read on the same banks: +2.2 us on the adjoint)

    // XCD-aware id map (as k_mix_lds): ids L and L + 8 (same XCD, adjacent in dispatch order) take the two mode tiles of one
    // 128-byte line for the same batch chunk.
    const int L = blockIdx.x;
    const int q = ((L >> 4) << 3) + (L & 7);
    const int chunk = q / npair;
    const int tile = 2 * (q - chunk * npair) + ((L >> 3) & 1);
    if (chunk >= nchunk || tile >= ntile) return;
    const int bbeg = chunk * BC;
    const int nrows = B - bbeg < BC ? B - bbeg : BC;
    if (nrows <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = cfd_uniform(tid >> 6);
    const int col = lane & 15, kq = lane >> 4;

    // ---- every global load of the workgroup, issued up front: weights, then the stages in order -------------------------
    f32x4 wl[NWL], gl[NST][NLD], xl[WGRAD ? NST : 1][NLD];
    {
        const int m0 = tile * CFD_MM_T;
        const float* wsrc = m0 < half ? w1 + (size_t)m0 * 2 : w2 + (size_t)(m0 - half) * 2;
#pragma unroll
        for (int k = 0; k < NWL; ++k) {
            const int f = tid + 512 * k, fc = f < WF4 ? f : WF4 - 1;
            wl[k] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)(fc >> 2) * half * 2 + 4 * (fc & 3));
        }
    }
#pragma unroll
    for (int s = 0; s < NST; ++s) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + 512 * k, fc = f < SF4 ? f : SF4 - 1;
            const int rs = fc / (4 * C), rem = fc - rs * (4 * C);
            const int row = 16 * s + rs;
            const unsigned b = (unsigned)(bbeg + (row < nrows ? row : nrows - 1));  // rows past the chunk: clamped, zeroed below
            const size_t off = ((size_t)(b * (unsigned)C + (unsigned)(rem >> 2)) * M + tile * CFD_MM_T) * 2 + 4 * (rem & 3);
            gl[s][k] = *reinterpret_cast<const f32x4*>(xin + off);
            if constexpr (WGRAD) xl[s][k] = *reinterpret_cast<const f32x4*>(xw + off);
        }
    }
    cfd_sched_fence();

    // ---- weights -> LDS -> this wave's B operands: bw[t][2 j + rk] = Wblock[k = (channel 4 j + kq, rk)][n = 16 t + col] ------
#pragma unroll
    for (int k = 0; k < NWL; ++k) {
        const int f = tid + 512 * k;
        if (f < WF4) {
            float* d = WS + (f >> 2) * CP + 4 * (f & 3);
            *reinterpret_cast<float2*>(d) = make_float2(wl[k][0], wl[k][1]);
            *reinterpret_cast<float2*>(d + 2) = make_float2(wl[k][2], wl[k][3]);
        }
    }
    __syncthreads();
    float bw[NT][2 * (C / 4)];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 16 * t + col, czr = n >> 1, rz = n & 1;
        const int cz = czr < C ? czr : C - 1;  // columns past 2C: never stored
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
            const int cr = 4 * j + kq;
            const int io = CONJT ? cz * C + cr : cr * C + cz;  // W[i][o]: forward contracts i, the adjoint o
            const float2 wv = (CFD_MM_EXP & 4) ? make_float2(wl[0][0] + io, wl[NWL - 1][1]) : *reinterpret_cast<const float2*>(WS + io * CP + 2 * w);
            // forward block [[Wr, Wi], [-Wi, Wr]] (rows (i, re/im), columns (o, re/im)); adjoint = its conjugate transpose
            bw[t][2 * j] = rz == 0 ? wv.x : (CONJT ? -wv.y : wv.y);
            bw[t][2 * j + 1] = rz == 0 ? (CONJT ? wv.y : -wv.y) : wv.x;
        }
    }

    f32x4 accw[WGRAD ? NT : 1][WGRAD ? NT : 1];
    if constexpr (WGRAD) {
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) accw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll
    for (int s = 0; s < NST; ++s) {
        if (16 * s >= nrows) break;  // (uniform per workgroup)
        // ---- stage s: registers -> LDS rows (entries past the chunk as zeros: they must not reach the weight gradient) ----
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + 512 * k;
            const int rs = f / (4 * C), rem = f - rs * (4 * C);
            const int row = 16 * s + rs;
            if (f < SF4 && row < RMAX) {
                const bool live = row < nrows;
                const int o = row * BP + (rem >> 2) * CP + 4 * (rem & 3);
                const f32x4 g = gl[s][k];
                *reinterpret_cast<float2*>(GS + o) = live ? make_float2(g[0], g[1]) : make_float2(0.f, 0.f);
                *reinterpret_cast<float2*>(GS + o + 2) = live ? make_float2(g[2], g[3]) : make_float2(0.f, 0.f);
                if constexpr (WGRAD) {
                    const f32x4 x = xl[s][k];
                    *reinterpret_cast<float2*>(XS + o) = live ? make_float2(x[0], x[1]) : make_float2(0.f, 0.f);
                    *reinterpret_cast<float2*>(XS + o + 2) = live ? make_float2(x[2], x[3]) : make_float2(0.f, 0.f);
                }
            }
        }
        __syncthreads();
        // ---- mix: rows = the 16 entries of the stage, A[row][k = (channel 4 j + kq, re/im)] ----
        f32x4 acc[NT];
        {
            const int rowA = 16 * s + col < RMAX ? 16 * s + col : RMAX - 1;
            float av[2 * (C / 4)];
#pragma unroll
            for (int j = 0; j < C / 4; ++j) {
                const float2 a = *reinterpret_cast<const float2*>(GS + rowA * BP + (4 * j + kq) * CP + 2 * w);
                av[2 * j] = a.x;
                av[2 * j + 1] = a.y;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if constexpr (CFD_MM_EXP & 1) acc[t][0] += av[ks] * bw[t][ks];
                    else acc[t] = cfd_mfma16x16x4(av[ks], bw[t][ks], acc[t]);
                }
        }
        // ---- weight gradient: K = the stage's entries in steps of four, A = xw^T, B = the gradient modes ----
        if constexpr (WGRAD) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (16 * s + 4 * kk >= nrows) break;  // (uniform; rows up to the next multiple of four hold zeros)
                const int rowK = (16 * s + 4 * kk + kq) * BP + 2 * w;
                float xa[NT], gb[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int rho = 16 * t + col;
                    const int o = rowK + ((rho < 2 * C ? rho : 2 * C - 1) >> 1) * CP + (rho & 1);
                    const float xv = XS[o], gv = GS[o];
                    xa[t] = rho < 2 * C ? xv : 0.f;
                    gb[t] = rho < 2 * C ? gv : 0.f;
                }
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b) {
                        if constexpr (CFD_MM_EXP & 2) accw[a][b][0] += xa[a] * gb[b];
                        else accw[a][b] = cfd_mfma16x16x4(xa[a], gb[b], accw[a][b]);
                    }
            }
        }
        // ---- mix results back into the rows (this wave's mode slots only), then out in 64-byte pieces ----
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = 16 * t + col;
            if (n < 2 * C) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * s + 4 * kq + e;
                    if (row < RMAX) GS[row * BP + (n >> 1) * CP + 2 * w + (n & 1)] = acc[t][e];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int f = tid + 512 * k;
            const int rs = f / (4 * C), rem = f - rs * (4 * C);
            const int row = 16 * s + rs;
            if (f < SF4 && row < nrows) {
                const int o = row * BP + (rem >> 2) * CP + 4 * (rem & 3);
                const float2 lo = *reinterpret_cast<const float2*>(GS + o), hi = *reinterpret_cast<const float2*>(GS + o + 2);
                const size_t off = ((size_t)((unsigned)(bbeg + row) * (unsigned)C + (unsigned)(rem >> 2)) * M + tile * CFD_MM_T) * 2 + 4 * (rem & 3);
                if (!(CFD_MM_EXP & 16) || lo.x == 1.2345f) *reinterpret_cast<f32x4*>(z + off) = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
        }
    }

    // ---- weight-gradient partial sums of this chunk: combine the four real blocks, transpose through LDS, out in pieces ----
    if constexpr (WGRAD) {
        const float sg = (col & 1) ? -1.f : 1.f;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // rows (i, 0), (i, 1) of this lane's column (o, ro): re = D[(i,0)][(o,0)] + D[(i,1)][(o,1)] on the ro = 0 lane,
                    // im = D[(i,0)][(o,1)] - D[(i,1)][(o,0)] on its ro = 1 neighbour
                    const float e0 = accw[a][b][2 * h], e1 = accw[a][b][2 * h + 1];
                    const float out = fmaf(sg, cfd_shfl_xor(e1, 1), e0);
                    const int i = 8 * a + 2 * kq + h, n = 16 * b + col;
                    if (i < C && n < 2 * C) WS[(i * C + (n >> 1)) * CP + 2 * w + (n & 1)] = out;
                }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NWL; ++k) {
            const int f = tid + 512 * k;
            if (f < WF4) {
                const float* sp = WS + (f >> 2) * CP + 4 * (f & 3);
                const float2 lo = *reinterpret_cast<const float2*>(sp), hi = *reinterpret_cast<const float2*>(sp + 2);
                *reinterpret_cast<f32x4*>(part + (((size_t)chunk * C * C + (f >> 2)) * M + tile * CFD_MM_T) * 2 + 4 * (f & 3)) = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
        }
    }
}

// Chunk geometry.  rmax = LDS rows of the instantiation; spread = aim at about one workgroup per CU (the shapes whose LDS allows only
// one: with the weight gradient, or 32 channels) instead of filling the rows.  The mode_bc knob overrides the rows.
static void cfd_mm_geometry(int B, int ntile, int rmax, bool spread, int* BC, int* nchunk) {
    int bc;
    const int knob = cfd_tune_get(CFD_TUNE_MODE_BC);
    if (knob >= 1) bc = knob < rmax ? knob : rmax;
    else if (!spread) bc = rmax;
    else {
        int want = 256 / ntile;
        if (want < 1) want = 1;
        bc = (B + want - 1) / want;
        if (bc > rmax) bc = rmax;
        if (bc < 16) bc = 16;
    }
    *BC = bc;
    *nchunk = (B + bc - 1) / bc;
}

bool cfd_int_modes_mfma_ok(const cfd_plan* p, int B, int Cin, int Cout, const void* a, const void* b, const void* c, bool wgrad) {
    const int knob = cfd_tune_get(CFD_TUNE_MODE_MFMA);
    if (knob == 0) return false;
    const int M = 2 * p->m1 * p->m2, half = p->m1 * p->m2;
    // the channel count is a template parameter: 20 (mix, adjoint, weight gradient) and 32 (mix / adjoint alone: two 32-channel operand
    // chunks plus the weights do not fit the LDS beside each other, the weight gradient stays on the VALU kernel there)
    if (Cin != Cout || !(Cin == 20 || (Cin == 32 && !wgrad))) return false;
    if (half % CFD_MM_T != 0) return false;                  // a tile never straddles weights1 / weights2
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16) return false;
    (void)M;
    // below ~128 entries the lane = mode kernels of spectral.hip win at 20 channels (few workgroups here, each paying the weight fill:
    // 64-case rollout step 191.6 against 193.8 us); at 32 channels the matrix pipe already wins at 64 (66 x 65: 359.4 -> 348.4 us)
    return knob == 1 || B >= (Cin == 32 ? 64 : 128);
}

size_t cfd_int_modes_mfma_chunks(const cfd_plan* p, int B) {  // partial-sum chunks the weight gradient may write
    if ((p->m1 * p->m2) % CFD_MM_T != 0 || B < 1) return 0;    // (never dispatched for such plans: cfd_int_modes_mfma_ok)
    int BC, nchunk;
    cfd_mm_geometry(B, p->m1 * p->m2 * 2 / CFD_MM_T, 40, true, &BC, &nchunk);
    return (size_t)nchunk;
}

template <int C, bool CONJT, bool WGRAD, int NST, int RMAX>
static int mm_launch(const float* xin, const float* xw, const float* w1, const float* w2, float* z, float* part, int B, int BC,
                     int nchunk, int M, int half, hipStream_t st) {
    const int ntile = M / CFD_MM_T, npair = (ntile + 1) / 2;
    const unsigned grid = (unsigned)((npair * nchunk + 7) / 8) * 16u;
    const size_t lds = cfd_mm_lds_bytes<C>(WGRAD, RMAX);
    static bool attr_set = false;  // > 64 KB of dynamic LDS needs the attribute once per kernel
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_modes_mfma<C, CONJT, WGRAD, NST, RMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((k_modes_mfma<C, CONJT, WGRAD, NST, RMAX>), dim3(grid), dim3(512), lds, st, xin, xw, w1, w2, z, part, B, BC, M, half,
                       ntile, npair, nchunk);
    return CFD_OK;
}

// z = mix(xin) (conj_t = 0: forward, 1: adjoint).  Caller checked cfd_int_modes_mfma_ok.
int cfd_int_modes_mix(const cfd_plan* p, const float* xin, const float* w1, const float* w2, float* z, int B, int C, int conj_t,
                      void* stream) {
    const int M = 2 * p->m1 * p->m2, half = p->m1 * p->m2;
    int BC, nchunk;
    hipStream_t st = (hipStream_t)stream;
    if (C == 32) {
        cfd_mm_geometry(B, M / CFD_MM_T, 37, true, &BC, &nchunk);
        if (conj_t) mm_launch<32, true, false, 3, 37>(xin, nullptr, w1, w2, z, nullptr, B, BC, nchunk, M, half, st);
        else mm_launch<32, false, false, 3, 37>(xin, nullptr, w1, w2, z, nullptr, B, BC, nchunk, M, half, st);
    } else {
        cfd_mm_geometry(B, M / CFD_MM_T, 32, false, &BC, &nchunk);
        if (conj_t) mm_launch<20, true, false, 2, 32>(xin, nullptr, w1, w2, z, nullptr, B, BC, nchunk, M, half, st);
        else mm_launch<20, false, false, 2, 32>(xin, nullptr, w1, w2, z, nullptr, B, BC, nchunk, M, half, st);
    }
    CFD_LAUNCH_CHECK("cfd_spectral_mix(mfma)");
    return CFD_OK;
}

// gz = adjoint mix of gh; part[chunk] = partial weight gradients (reduced by k_spec_wgrad_reduce / the tail workgroups)
int cfd_int_modes_mixadj_wgrad(const cfd_plan* p, const float* xh, const float* gh, const float* w1, const float* w2, float* gz,
                               float* part, int B, int C, int* nchunk_out, void* stream) {
    const int M = 2 * p->m1 * p->m2, half = p->m1 * p->m2;
    int BC, nchunk;
    cfd_mm_geometry(B, M / CFD_MM_T, 40, true, &BC, &nchunk);
    (void)C;
    mm_launch<20, true, true, 3, 40>(gh, xh, w1, w2, gz, part, B, BC, nchunk, M, half, (hipStream_t)stream);
    CFD_LAUNCH_CHECK("cfd_spectral_mix_adj_wgrad(mfma)");
    *nchunk_out = nchunk;
    return CFD_OK;
}
