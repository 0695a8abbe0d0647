// "Tail" jobs: tiny partial-sum reductions that ride as extra workgroups in front of a later kernel's launch instead
// of being kernels of their own.  On MI355X every kernel launch costs ~4.5 us of dispatch floor plus a ~1.5-us gap
// to its dependent successor whatever it does (rocprofv3: k_loss_coef, one workgroup, 4.6 us); the reductions of
// the weight-gradient partial sums are ~0.5-us jobs, two per FnoBlock and backward pass.  The host fills a
// CfdReduceTail, the carrying kernel (k_idft64, k_block) gives its first `nblk` workgroups to cfd_reduce_tail().
#pragma once
#include "cfd_common.h"

// spectral weight gradient: gw{1,2}[io, mode] = (c_l / HW) * sum_c part[c][io][mode]      (k_spec_wgrad_reduce)
struct SpecWgradTail {
    const float2* part;  // NULL: no such job
    float2* gw1;
    float2* gw2;
    const float* clhw;
    int nchunk, CC, m1, m2;
};
// 1x1-conv weight / bias gradient: block-major partial records part[block][Co (Ci + 1)] (k_chan_wgrad), nrow = blocks
struct ChanWgradTail {
    const float* part;  // NULL: no such job
    float* gw;
    float* gb;
    int nrow, Co, Ci;
};
// Lifting-layer (fc0) gradient sums emitted by the input-gradient kernel of FnoBlock 0 (k_block<.., STEMG>, round 5): the kernel holds
// g_0 = dL/da_0 tile by tile in its accumulators; instead of storing it (84 MB at B = 256) for a weight-gradient pass that reads it back,
// it adds, per (batch entry [x row split], channel c), the six sums  S = sum_px g_0 * {1, u, v, mask, grid_x(row), grid_y(col)}
// -- all the fc0 gradient needs: the case parameters are constant per entry (sum g_0 * cp[b,k] = cp[b,k] * S[0]) -- into
// part[entry][c][6]; k_stem_grad_combine finishes gw / gb.  inputs: (B, in_chan <= 2, H, W); mask: (B, H, W) or NULL (= ones).
struct CfdStemG {
    const float* inputs;  // NULL: not used (the kernel stores g_0 as usual)
    const float* mask;
    const float* gx;
    const float* gy;
    float* part;
    int in_chan;
};
struct CfdReduceTail {
    SpecWgradTail spec;
    ChanWgradTail chan;
    int nblk;  // workgroups of the carrying launch that run the jobs (0: none)
};

// Sums of the nchunk partial weight gradients of U (i, o, mode) elements gid0, gid0 + gstride, ..., scaled by c_l / HW and
// split into gw1 / gw2.  All U x 8 loads of a round are issued before the first add (one memory latency per round of 8
// chunks whatever U is); fixed summation order per element.
template <int U>
__device__ __forceinline__ void spec_wgrad_reduce_multi(const long gid0, const long gstride, const long total,
                                                        const float2* __restrict__ part, float2* __restrict__ gw1,
                                                        float2* __restrict__ gw2, const float* __restrict__ clhw,
                                                        int nchunk, int CC, int m1, int m2) {
    const int M = 2 * m1 * m2, half = m1 * m2;
    long gid[U];
    float ar[U], ai[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long g = gid0 + u * gstride;
        gid[u] = g < total ? g : total - 1;  // clamped: loaded, never stored
        ar[u] = 0.f;
        ai[u] = 0.f;
    }
    for (int c = 0; c < nchunk; c += 8) {
        float2 v[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[u][k] = part[(size_t)(c + k < nchunk ? c + k : nchunk - 1) * CC * M + gid[u]];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (c + k < nchunk) { ar[u] += v[u][k].x; ai[u] += v[u][k].y; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (gid0 + u * gstride >= total) continue;
        const int mode = (int)(gid[u] % M);
        const long io = gid[u] / M;
        const float sc = clhw[mode % m2];
        const float2 r = make_float2(ar[u] * sc, ai[u] * sc);
        if (mode < half) gw1[io * half + mode] = r;
        else gw2[io * half + (mode - half)] = r;
    }
}
__device__ __forceinline__ void spec_wgrad_reduce_one(const long gid, const float2* __restrict__ part,
                                                      float2* __restrict__ gw1, float2* __restrict__ gw2,
                                                      const float* __restrict__ clhw, int nchunk, int CC, int m1, int m2) {
    spec_wgrad_reduce_multi<1>(gid, 0, gid + 1, part, gw1, gw2, clhw, nchunk, CC, m1, m2);
}

// 1x1-conv weight / bias gradient elements [16 grp, 16 grp + 16) from the block-major partial records of k_chan_wgrad (cfd_record_sum16;
// every thread of the workgroup calls it, `scratch` >= 16 floats per wave)
__device__ __forceinline__ void chan_wgrad_reduce_group(const int grp, const float* __restrict__ part, int nrow, float* __restrict__ gw,
                                                        float* __restrict__ gb, int Co, int Ci, float* scratch) {
    const int NW = Ci + 1, total = Co * NW;
    const float s = cfd_record_sum16(part, nrow, total, 16 * grp, total, scratch);
    const int e = 16 * grp + (int)threadIdx.x;
    if (threadIdx.x < 16 && e < total) {
        const int o = e / NW, i = e - o * NW;
        if (i < Ci) gw[o * Ci + i] = s;
        else if (gb) gb[o] = s;
    }
}

// Workgroup `blk` of `t.nblk` (any workgroup size that is a multiple of 64): thread-strided over the spectral elements,
// workgroup-strided over the groups of 16 1x1-conv elements.  `scratch`: LDS of the carrying kernel, >= 16 floats per wave.
__device__ __forceinline__ void cfd_reduce_tail(const CfdReduceTail& t, int blk, float* scratch) {
    if (t.spec.part) {
        const long total = (long)t.spec.CC * 2 * t.spec.m1 * t.spec.m2;
        // three elements per thread and round: at 128 workgroups of 320 threads the 115 200 elements of C = 20 are ONE
        // round, i.e. one memory latency (a plain thread-strided loop costs one latency per element)
        const long gstride = (long)t.nblk * blockDim.x;
        for (long gid = (long)blk * blockDim.x + threadIdx.x; gid < total; gid += 3 * gstride)
            spec_wgrad_reduce_multi<3>(gid, gstride, total, t.spec.part, t.spec.gw1, t.spec.gw2, t.spec.clhw, t.spec.nchunk,
                                       t.spec.CC, t.spec.m1, t.spec.m2);
    }
    if (t.chan.part) {
        const int ngroups = (t.chan.Co * (t.chan.Ci + 1) + 15) / 16;
        for (int grp = blk; grp < ngroups; grp += t.nblk)  // (uniform per workgroup)
            chan_wgrad_reduce_group(grp, t.chan.part, t.chan.nrow, t.chan.gw, t.chan.gb, t.chan.Co, t.chan.Ci, scratch);
    }
}

// Internal (not exported through include/cfdbench_amd.h) entry points of the FNO backward phase: the weight-gradient
// producers leave their reductions in *tail, the fused block kernel carries them.
int cfd_int_spectral_mix_adj_wgrad(const cfd_plan* p, const float* xh, const float* gh, const float* w1, const float* w2,
                                   float* gz, float* gw1, float* gw2, void* ws, int B, int Cin, int Cout, void* stream,
                                   SpecWgradTail* defer);
int cfd_int_chan_wgrad(const float* g, const float* a, float* gw, float* gb, void* ws, int B, int Ci, int Co, int HW,
                       int act_in, void* stream, ChanWgradTail* defer);
int cfd_int_chan_wgrad_dt(const float* g, const void* a, float* gw, float* gb, void* ws, int B, int Ci, int Co, int HW,
                          int act_in, int dt, void* stream, ChanWgradTail* defer);
// stemg (may be NULL): emit the lifting layer's sums instead of storing the input gradient (only valid where cfd_int_stemg_ok holds and
// the call is the plain input gradient, aprev == NULL)
int cfd_int_fno_block_bwd_input(const cfd_plan* p, const float* g, const float* gz, const float* w0, const float* aprev,
                                float* gin, int B, int Cin, int Cout, void* stream, const CfdReduceTail* tail, const CfdStemG* stemg);
bool cfd_int_stemg_ok(const cfd_plan* p, int B, int C, int in_chan, int P, const void* inputs, const void* mask, const void* z);
size_t cfd_int_stemg_part_bytes(const cfd_plan* p, int B, int C);
int cfd_int_stemg_combine(const cfd_plan* p, const float* part, const float* cp, float* gw, float* gb, int B, int C, int in_chan, int P,
                          void* stream);
