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
// Projection head (k_head_bwd): block-major partial records part[block][PSTR] -> fc1 / fc2 gradients and, from the one-pass training head,
// the loss sums.  Round 6: in the fused training step this reduction rides in backward phase 1's block kernel instead of being a launch of
// its own behind the head (~5 us of dispatch floor; CFD_TRAIN_DEFER_HEAD).
#define CFD_HEAD_HD 128  // fno2d.py:175 (head.hip: HEAD_HD)
struct HeadTail {
    const float* part;  // NULL: no such job
    float* gw1;
    float* gb1;
    float* gw2;
    float* gb2;
    float* sums;   // may be NULL (stand-alone backward: no loss rows)
    int nrec, PS, PSTR, C, Co;
    float count;   // > 0 (deferred nMSE normaliser, CFD_TRAIN_DEFER_SCALE): row 2 of the loss rows is sum (label*mask)^2 -> sums[2], sums[3] = count
};
struct CfdReduceTail {
    SpecWgradTail spec;
    ChanWgradTail chan;
    HeadTail head;
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

// Elements [16 grp, 16 grp + 16) of the head's partial records (k_head_reduce's body; every thread of the workgroup calls it)
__device__ __forceinline__ void head_reduce_group(const int grp, const HeadTail& t, float* scratch) {
    const float s = cfd_record_sum16(t.part, t.nrec, t.PSTR, 16 * grp, t.PS, scratch);
    const int e = 16 * grp + (int)threadIdx.x;
    if (threadIdx.x < 16 && e < t.PS) {
        const int o_gb1 = CFD_HEAD_HD * t.C, o_gw2 = o_gb1 + CFD_HEAD_HD, o_gb2 = o_gw2 + t.Co * CFD_HEAD_HD;
        if (e < o_gb1) t.gw1[e] = s;
        else if (e < o_gw2) t.gb1[e - o_gb1] = s;
        else if (e < o_gb2) t.gw2[e - o_gw2] = s;
        else if (e < o_gb2 + t.Co) t.gb2[e - o_gb2] = s;
        else if (t.sums) {  // one-pass head: sums[0] = sum d^2, sums[1] = sum |d| (, sums[2] = sum (label*mask)^2, sums[3] = n)
            const int k = e - (o_gb2 + t.Co);
            t.sums[k] = s;
            if (k == 2) t.sums[3] = t.count;
        }
    }
}

// Lifting-layer gradient of channel c from the sums of k_block<.., STEMG> (CfdStemG): gw[c][f], f = (in_chan field channels, mask,
// grid_x, grid_y, P case parameters), and gb[c] from part[rec][c][6], rec = batch entry x row split -- thread-strided over the records in
// a fixed order, LDS tree; the case-parameter columns are sum_b cp[b][k] * S0[b][c] (a case parameter is constant over an entry's
// pixels).  256 threads; afterwards (barrier inside) s_r[k][0] holds: k = 0 bias, 1 .. 5 the five field sums, 6 + k case parameter k.
#define CFD_STEMG_NA (6 + 8)
__device__ __forceinline__ void stem_combine_channel(const float* __restrict__ part, int nrec, int spl, const float* __restrict__ cp, int P,
                                                     int C, int c, float (&s_r)[CFD_STEMG_NA][256]) {
    constexpr int NA = CFD_STEMG_NA;
    float a[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) a[k] = 0.f;
    for (int r = threadIdx.x; r < nrec; r += 256) {
        const float* q = part + ((size_t)r * C + c) * 6;
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = q[k];
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] += v[k];
        const float* cpb = cp + (size_t)(r / spl) * P;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < P) a[6 + k] = fmaf(cpb[k], v[0], a[6 + k]);
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) s_r[k][threadIdx.x] = a[k];
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h) {
#pragma unroll
            for (int k = 0; k < NA; ++k) s_r[k][threadIdx.x] += s_r[k][threadIdx.x + h];
        }
        __syncthreads();
    }
}
// feature f of the fc0 weight row (fno2d.py:189-217 channel order: fields, mask, grid_x, grid_y, case parameters) -> index into s_r
__device__ __forceinline__ int stem_feature_slot(int f, int in_chan) {
    return f < in_chan ? 1 + f : (f < in_chan + 3 ? 3 + (f - in_chan) : 6 + (f - in_chan - 3));
}
// The job as cfd_fno_adam_step carries it (CFD_TRAIN_DEFER_STEM): workgroup c < C finishes channel c and applies Adam to its row itself
struct StemAdamJob {
    const float* part;  // NULL: no such job
    const float* cp;
    int nrec, spl, P, in_chan, C;
    long w_off, b_off;  // element offsets of fc0.weight / fc0.bias inside the flat parameter buffer
};

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
    if (t.head.part) {
        const int ngroups = (t.head.PS + 15) / 16;
        for (int grp = blk; grp < ngroups; grp += t.nblk) head_reduce_group(grp, t.head, scratch);
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
// one-pass training head with its reduction optionally left to the caller (*defer filled, no reduce launch); c0v / c1v: the loss-gradient
// coefficients by value when coef == NULL (deferred nMSE normaliser); count > 0: also reduce sum (label*mask)^2 into sums[2], sums[3] = count
int cfd_int_fno_head_train_f(const void* a, const float* mask, const float* label, const float* coef, float c0v, float c1v, float count,
                             const float* w1, const float* b1, const float* w2, const float* b2, float* preds, float* sums, float* ga,
                             float* gw1, float* gb1, float* gw2, float* gb2, void* ws, int B, int C, int Hd, int Co, int HW, int act_in,
                             int dt, void* stream, HeadTail* defer);
HeadTail cfd_int_head_tail(const void* ws, float* gw1, float* gb1, float* gw2, float* gb2, float* sums, int B, int C, int Co, int HW, float count);
bool cfd_int_block_bwd_fused(const cfd_plan* p, int B, int C, const void* g, const void* gin, const void* aprev, const void* gz);
bool cfd_int_stemg_ok(const cfd_plan* p, int B, int C, int in_chan, int P, const void* inputs, const void* mask, const void* z);
size_t cfd_int_stemg_part_bytes(const cfd_plan* p, int B, int C);
int cfd_int_stemg_splits(const cfd_plan* p, int B);
int cfd_int_stemg_combine(const cfd_plan* p, const float* part, const float* cp, float* gw, float* gb, int B, int C, int in_chan, int P,
                          void* stream);

// Adam on the flat parameter buffer with the deferred work of the fused training step (fno.cpp: cfd_fno_adam_step): `sums` != NULL:
// every gradient is multiplied by sums[3] / sums[2] on top of grad_scale (deferred nMSE normaliser); job.part != NULL: workgroup c < C
// finishes the lifting layer's gradient row c from the block kernel's sum records, stores it and applies Adam to it.
int cfd_int_adam_flat_f(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, int step, float grad_scale, const float* sums, const StemAdamJob* job, void* stream);
