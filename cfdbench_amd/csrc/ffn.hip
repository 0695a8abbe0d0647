// A whole Linear(+activation) stack of the DeepONet family as ONE kernel per direction (Ffn, src/models/ffn.py:12-35):
//   y_0 = x;   z_l = y_{l-1} W_l^T + b_l,   y_l = act(z_l)   (l = 1 .. L; no activation after the last layer unless act_last)
// for stacks whose widths are all <= 128 (the reference's default width is 100, src/args.py:117-121).
//
// Why: at BASELINE configs[3] (Auto-DeepONet, width 100, depth 8 + 8, 512 x 4290) the layer-by-layer path issues 49 GEMM and 17
// split-K-reduce launches per training step for 13 MB of HBM traffic -- 0.62 + 0.20 ms of kernels that are each at their launch /
// latency floor.  Here a workgroup owns a tile of 64 rows and walks through ALL layers: the activations of the tile stay in LDS,
// the next layer's 40-KB weight matrix is prefetched from L2 into registers while the current layer is on the matrix pipe, and
// only what the backward pass needs crosses HBM (y_l, and z_l for gelu / swish).  The backward kernel walks the layers in
// reverse with the running gradient tile in LDS: dZ = G * act', bias / weight gradient partials per workgroup (reduced by one
// small kernel for the whole stack), G <- dZ W.  Arithmetic: exact fp32 on v_mfma_f32_16x16x4_f32 (the ReLU networks are not
// safe on split-bf16 products: DESIGN.md section 4), k-ordered fma chains like the stand-alone GEMM (dense.hip).
#include "cfd_common.h"

#ifdef CFD_FSDIAG  // experiment builds only (tools/build_variant.sh fsdiag ffn.hip -DCFD_FSDIAG, tools/exp/fs_diag.py): s_memtime stamps of
                   // the LAST workgroup's wave 0 at the phase boundaries of every layer of k_ffn_stack_fwd
__device__ unsigned long long fs_ts[256];  // [0..7][8]: forward, [8..15][8]: backward chain
extern "C" int cfd_dbg_fs_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fs_ts), (size_t)n * 8); }
#define FS_TS(l, slot)                                                                                      \
    do {                                                                                                    \
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && (l) < 16) fs_ts[(l) * 8 + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define FS_TS(l, slot) do { } while (0)
#endif

#define FS_MAXL 16    // layers per stack
#define FS_MAXD 128   // widest layer
#define FS_ROWS 16    // rows per workgroup: one MFMA row tile; the four waves split the output column tiles
#define FS_KSPLIT 8   // row slices of the batched weight-gradient kernel

struct FfnStack {
    const float* w[FS_MAXL];   // W_l (dims[l+1] x dims[l]), row-major (nn.Linear.weight)
    const float* b[FS_MAXL];   // b_l (dims[l+1]) or NULL
    float* y[FS_MAXL];         // y_l (R x dims[l+1])  -- forward: written; backward: read
    float* z[FS_MAXL];         // z_l (R x dims[l+1]) for act >= 3 (gelu / swish), else NULL
    int dims[FS_MAXL + 1];
    int L, act, act_last;
};

// Several independent stacks in ONE launch per direction (the branch and trunk nets of the DeepONet variants: 32 + 269 row tiles
// at BASELINE configs[3], each too few to fill 256 CUs and each at its latency floor): workgroups [blk0[s], blk0[s+1]) walk stack s.
#define FS_MAXS 3
struct FfnStackSet {
    FfnStack st[FS_MAXS];
    const float* x[FS_MAXS];
    int R[FS_MAXS];
    int blk0[FS_MAXS + 1];   // first workgroup (forward / chain) of each stack
    int lay0[FS_MAXS + 1];   // first (stack, layer) index of each stack in the weight-gradient / reduction grids
    int ns;
};

__device__ __forceinline__ float fs_act(float z, int act) {
    switch (act) {
        case 1: return z > 0.f ? z : 0.f;
        case 2: return tanhf(z);
        case 3: return cfd_gelu(z);
        case 4: return z * cfd_rcpf(1.f + cfd_expf(-z));
        default: return z;
    }
}
__device__ __forceinline__ float fs_act_grad(float y, float z, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return 1.f - y * y;
        case 3: return cfd_gelu_grad(z);
        case 4: { const float s = cfd_rcpf(1.f + cfd_expf(-z)); return s * (1.f + z * (1.f - s)); }
        default: return 1.f;
    }
}

// A weight matrix (Dout x Din, row-major) staged by the whole workgroup: thread (tj = tid / 32, ti = tid % 32) owns the four
// columns 4 ti .. 4 ti + 3 of rows tj, tj + 8, ...; registers first (the loads fly during the previous layer's MFMAs), LDS later.
// sW[j][i] = W[j][i] in both directions (the backward chain, which multiplies by W and not W^T, reads it transposed: fs_tile_gemm_t).
// Rows / columns up to the next multiple of 16 are written as zeros: they are MFMA padding.
#define FS_WROWS (FS_MAXD / 8)
struct FsWRegs { float4 v[FS_WROWS]; };
// Raw buffer loads: rows / columns outside the matrix carry the out-of-range offset and come back as zeros.  (Written as `if (inside)
// load`, each of a thread's 16 loads sat in its own branch with a wait behind it: the phase stamps of k_ffn_stack_fwd showed 3.2-5.7 k
// cycles for ISSUING one layer's weight loads where the loads were meant to fly beside the MFMAs; 1.6-3.4 k now:
// tools/exp/fs_diag.py, profiles/r04i_ffn_stack_phases.txt.)
__device__ __forceinline__ void fs_wfetch(const float* __restrict__ w, int Dout, int Din, FsWRegs& r) {
    const int tj = threadIdx.x >> 5, c0 = 4 * (threadIdx.x & 31);
    const bool vec = (Din & 3) == 0;  // (uniform) a 16-byte unit never straddles a row end
    const CfdBuf bf = cfd_buf(w, 4u * (unsigned)(Dout * Din));
#pragma unroll
    for (int k = 0; k < FS_WROWS; ++k) {
        const int j = tj + 8 * k;
        const bool ok = j < Dout && c0 < Din;
        const unsigned off = 4u * (unsigned)(j * Din + c0);
        if (vec) {
            const f32x4 t = cfd_buf_ld4(bf, ok ? off : CFD_BUF_OOB, 0);
            r.v[k] = make_float4(t[0], t[1], t[2], t[3]);
        } else {
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = cfd_buf_ld(bf, ok && c0 + e < Din ? off + 4u * e : CFD_BUF_OOB, 0);
            r.v[k] = make_float4(t[0], t[1], t[2], t[3]);
        }
    }
}
__device__ __forceinline__ void fs_wcommit(float* __restrict__ sW, int FS_LD, int Dout, int Din, const FsWRegs& r) {
    const int tj = threadIdx.x >> 5, c0 = 4 * (threadIdx.x & 31);
    const int Dop = (Dout + 15) & ~15, Dip = (Din + 15) & ~15;
#pragma unroll
    for (int k = 0; k < FS_WROWS; ++k) {
        const int j = tj + 8 * k;
        if (j < Dop && c0 < Dip) *reinterpret_cast<float4*>(sW + j * FS_LD + c0) = r.v[k];
    }
}

// out[16 rows][16 t + n] = sum_k sIn[row][k] sOp[16 t + n][k] for this wave's column tiles t = wave, wave + 4 (< NT), K padded to
// NG groups of 16.  k-slot map of a group: MFMA step i of lane (q, n) carries k = 16 g + 4 q + i, so a lane's four steps are ONE
// 16-byte LDS read per operand (the map is arbitrary as long as both operands use it).
// fs_tile_gemm_t: the operand matrix is stored TRANSPOSED to what the product needs (the backward chain multiplies by W, whose LDS image is the
// forward's sW[j][i]): out[row][16 t + n] = sum_k sIn[row][k] sOp[k][16 t + n]; a lane's four steps are then four 4-byte reads
// (conflict-free: the 16 lanes of a group read 16 consecutive floats).  Round 3 stored W^T instead, with scalar LDS stores whose
// addresses a thread's rows put 4 FS_LD floats apart -- 16 mod 32 banks, a 16-way conflict on each of 64 stores per thread and layer:
// SQ_LDS_BANK_CONFLICT was 87 % of the kernel's LDS cycles (profiles/r04e_ffn_bwd_lds.txt).
__device__ __forceinline__ void fs_tile_gemm_t(const float* __restrict__ sIn, const float* __restrict__ sOp, int FS_LD, int NG, int NT,
                                                int wave, int q, int n, f32x4 (&acc)[2]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    acc[0] = acc[1] = zero;
    const bool t1 = wave + 4 < NT;
    if (wave >= NT) return;
    const float* ap = sIn + n * FS_LD + 4 * q;
    const float* b0 = sOp + (4 * q) * FS_LD + 16 * wave + n;
    const float* b1 = sOp + (4 * q) * FS_LD + 16 * (t1 ? wave + 4 : wave) + n;
    auto ldb = [&](const float* b, int g) { return make_float4(b[(16 * g) * FS_LD], b[(16 * g + 1) * FS_LD], b[(16 * g + 2) * FS_LD], b[(16 * g + 3) * FS_LD]); };
    float4 a = *reinterpret_cast<const float4*>(ap), x0 = ldb(b0, 0), x1 = ldb(b1, 0);
    for (int g = 0; g < NG; ++g) {
        const int gn = g + 1 < NG ? g + 1 : g;  // next group's operands are requested before this group's MFMAs
        const float4 an = *reinterpret_cast<const float4*>(ap + 16 * gn);
        const float4 y0 = ldb(b0, gn), y1 = ldb(b1, gn);
        acc[0] = cfd_mfma16x16x4(a.x, x0.x, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.x, x1.x, acc[1]);
        acc[0] = cfd_mfma16x16x4(a.y, x0.y, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.y, x1.y, acc[1]);
        acc[0] = cfd_mfma16x16x4(a.z, x0.z, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.z, x1.z, acc[1]);
        acc[0] = cfd_mfma16x16x4(a.w, x0.w, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.w, x1.w, acc[1]);
        a = an; x0 = y0; x1 = y1;
    }
}
__device__ __forceinline__ void fs_tile_gemm(const float* __restrict__ sIn, const float* __restrict__ sOp, int FS_LD, int NG, int NT,
                                              int wave, int q, int n, f32x4 (&acc)[2]) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    acc[0] = acc[1] = zero;
    const bool t1 = wave + 4 < NT;
    if (wave >= NT) return;
    const float* ap = sIn + n * FS_LD + 4 * q;
    const float* b0 = sOp + (16 * wave + n) * FS_LD + 4 * q;
    const float* b1 = sOp + (16 * (t1 ? wave + 4 : wave) + n) * FS_LD + 4 * q;
    float4 a = *reinterpret_cast<const float4*>(ap), x0 = *reinterpret_cast<const float4*>(b0), x1 = *reinterpret_cast<const float4*>(b1);
    for (int g = 0; g < NG; ++g) {
        const int gn = g + 1 < NG ? g + 1 : g;  // next group's operands are requested before this group's MFMAs
        const float4 an = *reinterpret_cast<const float4*>(ap + 16 * gn);
        const float4 y0 = *reinterpret_cast<const float4*>(b0 + 16 * gn), y1 = *reinterpret_cast<const float4*>(b1 + 16 * gn);
        acc[0] = cfd_mfma16x16x4(a.x, x0.x, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.x, x1.x, acc[1]);
        acc[0] = cfd_mfma16x16x4(a.y, x0.y, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.y, x1.y, acc[1]);
        acc[0] = cfd_mfma16x16x4(a.z, x0.z, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.z, x1.z, acc[1]);
        acc[0] = cfd_mfma16x16x4(a.w, x0.w, acc[0]);
        acc[1] = cfd_mfma16x16x4(a.w, x1.w, acc[1]);
        a = an; x0 = y0; x1 = y1;
    }
}

// 16 rows of a (R x D) row-major matrix -> sT[row][col], zeros beyond R / D up to the next multiple of 16 columns
__device__ __forceinline__ void fs_tile_load(const float* __restrict__ src, int R, int D, int row0, float* __restrict__ sT, int FS_LD) {
    const int Dp = (D + 15) & ~15;
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;  // 16 x 16 threads
    for (int cc = c; cc < Dp; cc += 16) sT[r * FS_LD + cc] = (row0 + r < R && cc < D) ? src[(size_t)(row0 + r) * D + cc] : 0.f;
}

// ------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------
// Dynamic LDS: row stride FS_LD = (widest layer rounded up to 16) + 4 floats (16-byte aligned rows, staggered banks);
// [2][16][FS_LD] activations (ping-pong) | [widest][FS_LD] weights -- 67 KB at width 100, so two workgroups share a CU.
__global__ __launch_bounds__(256, 2) void k_ffn_stack_fwd(const FfnStackSet set, int FS_LD) {
    CFD_DYN_SHARED(float4, s_dyn4);
    int si = 0;
    while (si + 1 < set.ns && (int)blockIdx.x >= set.blk0[si + 1]) ++si;
    const FfnStack& st = set.st[si];
    const float* __restrict__ x = set.x[si];
    const int R = set.R[si];
    float* sA0 = reinterpret_cast<float*>(s_dyn4);
    float* sA[2] = {sA0, sA0 + FS_ROWS * FS_LD};   // the tile's activations y_{l-1} / y_l (ping-pong)
    float* sW = sA0 + 2 * FS_ROWS * FS_LD;         // W_l
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int row0 = ((int)blockIdx.x - set.blk0[si]) * FS_ROWS;
    FsWRegs wr;
    fs_wfetch(st.w[0], st.dims[1], st.dims[0], wr);
    fs_tile_load(x, R, st.dims[0], row0, sA[0], FS_LD);
    int cur = 0;
    for (int l = 0; l < st.L; ++l, cur ^= 1) {
        const int Din = st.dims[l], Dout = st.dims[l + 1];
        const int NG = (Din + 15) >> 4, NT = (Dout + 15) >> 4;
        FS_TS(l, 0);
        __syncthreads();  // every wave is done with W_{l-1} and has written its columns of sA[cur]
        FS_TS(l, 1);
        fs_wcommit(sW, FS_LD, Dout, Din, wr);
        FS_TS(l, 2);
        __syncthreads();
        FS_TS(l, 3);
        if (l + 1 < st.L) fs_wfetch(st.w[l + 1], st.dims[l + 2], st.dims[l + 1], wr);  // in flight during this layer's MFMAs
        FS_TS(l, 4);
        // (the bias values of this lane's two column tiles are requested BEFORE the MFMAs: loaded in the epilogue they cost one exposed
        // L2 latency per layer of a kernel that is a chain of such latencies)
        const float* bias = st.b[l];
        float bvs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = 16 * (wave + 4 * h) + n;
            bvs[h] = (bias && col < Dout) ? bias[col] : 0.f;
        }
        f32x4 acc[2];
        fs_tile_gemm(sA[cur], sW, FS_LD, NG, NT, wave, q, n, acc);
        FS_TS(l, 5);
        // epilogue: bias, activation, store what backward needs, the tile's next input into the other sA buffer
        const bool do_act = st.act != 0 && (l + 1 < st.L || st.act_last);
        float* yl = st.y[l];
        float* zl = st.z[l];
        float* sN = sA[cur ^ 1];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = wave + 4 * h;
            if (t < NT) {
                const int col = 16 * t + n;
                const float bv = bvs[h];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = 4 * q + r, row = row0 + lr;
                    const float zv = acc[h][r] + bv;
                    const float yv = do_act ? fs_act(zv, st.act) : zv;
                    if (col < Dout && row < R) {
                        yl[(size_t)row * Dout + col] = yv;
                        if (zl && do_act) zl[(size_t)row * Dout + col] = zv;
                    }
                    sN[lr * FS_LD + col] = col < Dout ? yv : 0.f;  // k padding of the next layer reads zeros
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, part 1: the gradient chain  dZ_l = G_l * act'(y_l, z_l),  G_{l-1} = dZ_l W_l  (l = L .. 1), dZ_l stored for part 2
// ------------------------------------------------------------------------------------------------------
struct FfnStackBwd {
    float* dz[FS_MAXL];    // dZ_l (R x dims[l+1]), workspace
    int poff[FS_MAXL];     // offset (floats) of layer l's [gw (Dout x Din) | gb (Dout)] inside one partial slice
    int pstride;           // floats per partial slice
};

struct FfnStackBwdSet {
    FfnStackBwd bw[FS_MAXS];
    const float* gy[FS_MAXS];
    float* gx[FS_MAXS];
    float* part[FS_MAXS];   // FS_KSPLIT partial slices of the stack's gradients
};

__global__ __launch_bounds__(256, 2) void k_ffn_stack_bwd_chain(const FfnStackSet set, const FfnStackBwdSet bset, int FS_LD) {
    CFD_DYN_SHARED(float4, s_dyn4);
    int si = 0;
    while (si + 1 < set.ns && (int)blockIdx.x >= set.blk0[si + 1]) ++si;
    const FfnStack& st = set.st[si];
    const FfnStackBwd& bw = bset.bw[si];
    const float* __restrict__ gy = bset.gy[si];
    float* __restrict__ gx = bset.gx[si];
    const int R = set.R[si];
    float* sG0 = reinterpret_cast<float*>(s_dyn4);
    float* sG[2] = {sG0, sG0 + FS_ROWS * FS_LD};   // dZ_l of the tile / the next G (ping-pong)
    float* sW = sG0 + 2 * FS_ROWS * FS_LD;         // W_l: sW[j][i] (rows = this layer's outputs j = the chain's K)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int row0 = ((int)blockIdx.x - set.blk0[si]) * FS_ROWS;
    FsWRegs wr;
    fs_wfetch(st.w[st.L - 1], st.dims[st.L], st.dims[st.L - 1], wr);
    fs_tile_load(gy, R, st.dims[st.L], row0, sG[0], FS_LD);
    int cur = 0;
    // The y / z values a thread needs for dZ_l = G * act'(y_l, z_l) do not depend on the chain: they are requested one layer AHEAD (beside
    // the weight loads, in flight during the previous layer's MFMAs) -- loaded inside the dZ pass, the pass took 4.3-5.8 k of a layer's
    // ~15 k cycles (phase stamps, profiles/r04i_ffn_stack_phases.txt).  Thread (r, c): row r of the tile, columns c, c + 16, ...
    float yv[FS_MAXD / 16], zv[FS_MAXD / 16];
    const int dr = threadIdx.x >> 4, dc = threadIdx.x & 15;
    const bool rowok = row0 + dr < R;
    const auto yfetch = [&](int l) {
        const int Dout = st.dims[l + 1];
        const bool act_l = st.act != 0 && (l + 1 < st.L || st.act_last);
        const float* yl = st.y[l];
        const float* zl = st.z[l];
        const size_t ro = (size_t)(rowok ? row0 + dr : 0) * Dout;
#pragma unroll
        for (int u = 0; u < FS_MAXD / 16; ++u) {
            const int cc = dc + 16 * u, ccl = cc < Dout ? cc : 0;  // (clamped: unconditional loads)
            yv[u] = act_l ? yl[ro + ccl] : 0.f;
            zv[u] = (act_l && zl) ? zl[ro + ccl] : 0.f;
        }
    };
    yfetch(st.L - 1);
    for (int l = st.L - 1; l >= 0; --l, cur ^= 1) {
        const int Din = st.dims[l], Dout = st.dims[l + 1];
        const bool do_act = st.act != 0 && (l + 1 < st.L || st.act_last);
        const bool need_g = l > 0 || gx != nullptr;
        FS_TS(8 + (st.L - 1 - l), 0);  // (rows 8.. of the stamp table: the chain kernel, layers in its own order)
        __syncthreads();  // sG[cur] is complete; every wave is done with W_{l+1}
        FS_TS(8 + (st.L - 1 - l), 1);
        {   // dZ_l = G * act' in place (16 x 16 threads: one row, strided columns), stored for the weight-gradient kernel
            const int Dp = (Dout + 15) & ~15;
            float* g = sG[cur] + dr * FS_LD;
            float* dz = bw.dz[l];
            const size_t ro = (size_t)(rowok ? row0 + dr : 0) * Dout;
#pragma unroll
            for (int u = 0; u < FS_MAXD / 16; ++u) {
                const int cc = dc + 16 * u;
                if (cc < Dp) {
                    float v = g[cc];
                    if (rowok && cc < Dout) {
                        if (do_act) v *= fs_act_grad(yv[u], zv[u], st.act);
                        dz[ro + cc] = v;
                        g[cc] = v;
                    }
                }
            }
        }
        FS_TS(8 + (st.L - 1 - l), 2);
        if (need_g) fs_wcommit(sW, FS_LD, Dout, Din, wr);  // sW[j][i] = W[j][i], 16-byte stores (fs_tile_gemm_t reads it transposed)
        FS_TS(8 + (st.L - 1 - l), 3);
        __syncthreads();
        if (l > 0) {
            fs_wfetch(st.w[l - 1], st.dims[l], st.dims[l - 1], wr);
            yfetch(l - 1);  // (this layer's values were consumed by the dZ pass above)
        }
        FS_TS(8 + (st.L - 1 - l), 4);
        if (!need_g) break;
        // G_{l-1}[row][i] = sum_j dZ[row][j] W[j][i]: operand rows of sW are the output columns i, K = j
        const int NG = (Dout + 15) >> 4, NT = (Din + 15) >> 4;
        f32x4 acc[2];
        fs_tile_gemm_t(sG[cur], sW, FS_LD, NG, NT, wave, q, n, acc);
        FS_TS(8 + (st.L - 1 - l), 5);
        float* sN = sG[cur ^ 1];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int t = wave + 4 * h;
            if (t < NT) {
                const int col = 16 * t + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int lr = 4 * q + r;
                    if (l > 0) sN[lr * FS_LD + col] = col < Din ? acc[h][r] : 0.f;
                    else if (col < Din && row0 + lr < R) gx[(size_t)(row0 + lr) * Din + col] = acc[h][r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, part 2: every layer's weight / bias gradient in ONE launch:  gW_l[j][i] = sum_r dZ_l[r][j] y_{l-1}[r][i], the bias
// gradient as the extra input column i = Din whose value is 1.  Grid (tile, layer, row slice); one wave per 16 x 16 tile and
// slice; partial tiles are summed by k_ffn_stack_reduce in a fixed order.
// ------------------------------------------------------------------------------------------------------
// Block -> (tile, layer, row slice).  The NTi x NTj tiles of one (layer, slice) read the SAME rows of dZ and of the layer input -- each
// column block once per tile of the other operand, 7x at width 100 -- and workgroups are dealt to the 8 XCDs (one L2 each) round robin
// in launch order: as a plain (tile, layer, slice) grid the tiles of a group landed on all eight L2s and every one of them fetched the
// rows again (rocprofv3 FETCH_SIZE: 223 MB per launch for 27 MB of operands, the kernel ran at the fabric's rate:
// profiles/r05p_auto_deeponet_pmc_traffic.json).  Round 5: a 1-D grid whose block l runs work item (l % 8) * (items / 8) + l / 8, so
// the tiles of a group are consecutive blocks of ONE XCD and the re-reads are L2 hits.
__global__ __launch_bounds__(64) void k_ffn_stack_wgrad(const FfnStackSet set, const FfnStackBwdSet bset, int tmax, int layers, int per_xcd) {
    const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (item >= tmax * layers * FS_KSPLIT) return;
    const int bx = item % tmax, by = (item / tmax) % layers, bz = item / (tmax * layers);
    int si = 0;
    while (si + 1 < set.ns && by >= set.lay0[si + 1]) ++si;
    const FfnStack& st = set.st[si];
    const FfnStackBwd& bw = bset.bw[si];
    const float* __restrict__ x = set.x[si];
    float* __restrict__ part = bset.part[si];
    const int R = set.R[si];
    const int l = by - set.lay0[si], slice = bz;
    const int Din = st.dims[l], Dout = st.dims[l + 1];
    const int NTi = (Din + 1 + 15) >> 4, NTj = (Dout + 15) >> 4;
    if (bx >= NTi * NTj) return;
    const int tj = bx / NTi, ti = bx - tj * NTi;
    const int lane = threadIdx.x, q = lane >> 4, n = lane & 15;
    const float* dz = bw.dz[l];
    const float* yin = l > 0 ? st.y[l - 1] : x;
    const int per = ((R + FS_KSPLIT - 1) / FS_KSPLIT + 3) & ~3;
    const int r0 = slice * per, r1 = r0 + per < R ? r0 + per : R;
    const int j = 16 * tj + n, i = 16 * ti + n;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // Eight MFMA steps (32 rows) per trip, all sixteen loads of a trip issued before its first MFMA: with one step per trip every
    // step waited for its own two loads -- 134 exposed L2 latencies per wave at the trunk net's 536 rows per slice, 32 us per launch
    // for 0.7 GFLOP.  Unconditional loads from clamped addresses (a `cond ? load : 0` is a branch and a full wait per element), zeroed
    // afterwards; the products are added in row order as before: the sums do not depend on the trip structure.
    const int jc = j < Dout ? j : Dout - 1, ic = i < Din ? i : Din - 1;
    const float bconst = i == Din ? 1.f : 0.f;  // the bias gradient's column of ones
    int r = r0 + q;
    for (; r + 28 < r1 + q; r += 32) {  // (wave-uniform) all eight steps start inside [r0 + q, r1 + q)
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = r + 4 * u, rc = rr < r1 ? rr : r1 - 1;
            a[u] = dz[(size_t)rc * Dout + jc];
            b[u] = yin[(size_t)rc * Din + ic];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = r + 4 * u < r1;
            const float av = (ok && j < Dout) ? a[u] : 0.f;
            const float bv = !ok ? 0.f : (i < Din ? b[u] : bconst);
            acc = cfd_mfma16x16x4(av, bv, acc);
        }
    }
    for (; r < r1 + q; r += 4) {  // wave-uniform trip count; lanes past the slice contribute zeros
        const bool ok = r < r1;
        const int rc = ok ? r : r1 - 1;
        const float a = dz[(size_t)rc * Dout + jc], b = yin[(size_t)rc * Din + ic];
        acc = cfd_mfma16x16x4((ok && j < Dout) ? a : 0.f, !ok ? 0.f : (i < Din ? b : bconst), acc);
    }
    float* p = part + (size_t)slice * bw.pstride + bw.poff[l];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int jj = 16 * tj + 4 * q + r;
        if (jj < Dout) {
            if (i < Din) p[jj * Din + i] = acc[r];
            else if (i == Din) p[Dout * Din + jj] = acc[r];
        }
    }
}

// gw_l / gb_l = sum over the row slices, fixed order (deterministic); one thread per output element
struct FfnStackGrads {
    float* gw[FS_MAXL];
    float* gb[FS_MAXL];
};
struct FfnStackGradsSet { FfnStackGrads g[FS_MAXS]; };
__global__ __launch_bounds__(256) void k_ffn_stack_reduce(const FfnStackSet set, const FfnStackBwdSet bset, const FfnStackGradsSet gset) {
    int si = 0;
    while (si + 1 < set.ns && (int)blockIdx.y >= set.lay0[si + 1]) ++si;
    const FfnStack& st = set.st[si];
    const FfnStackBwd& bw = bset.bw[si];
    const FfnStackGrads& g = gset.g[si];
    const float* __restrict__ part = bset.part[si];
    const int l = (int)blockIdx.y - set.lay0[si];
    const int Din = st.dims[l], Dout = st.dims[l + 1];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Dout * Din + Dout) return;
    const float* p = part + bw.poff[l] + e;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < FS_KSPLIT; ++k) s += p[(size_t)k * bw.pstride];
    if (e < Dout * Din) g.gw[l][e] = s;
    else if (g.gb[l]) g.gb[l][e - Dout * Din] = s;
}

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
static int fs_check(const char* fn, int R, const int* dims, int L, int act) {
    CFD_REQUIRE(dims && R >= 0 && L >= 1 && L <= FS_MAXL, CFD_ERR_INVALID_ARG, "%s: bad arguments (1 <= layers <= %d)", fn, FS_MAXL);
    CFD_REQUIRE(act >= 0 && act <= 4, CFD_ERR_INVALID_ARG, "%s: activation %d (0 none, 1 relu, 2 tanh, 3 gelu, 4 swish)", fn, act);
    for (int l = 0; l <= L; ++l)
        CFD_REQUIRE(dims[l] >= 1 && dims[l] <= FS_MAXD, CFD_ERR_UNSUPPORTED, "%s: width %d of layer %d outside 1..%d", fn, dims[l], l, FS_MAXD);
    return CFD_OK;
}

static int fs_ld(const int* dims, int L) {  // LDS row stride: widest layer rounded up to 16, + 4
    int m = 16;
    for (int l = 0; l <= L; ++l) { const int d = (dims[l] + 15) & ~15; if (d > m) m = d; }
    return m + 4;
}
static size_t fs_lds_bytes(int ld) { return (size_t)(2 * FS_ROWS + (ld - 4)) * ld * sizeof(float); }

static void fs_fill(FfnStack& st, const float* const* w, const float* const* b, float* const* y, float* const* z, const int* dims,
                    int L, int act, int act_last) {
    st.L = L; st.act = act; st.act_last = act_last;
    for (int l = 0; l <= L; ++l) st.dims[l] = dims[l];
    for (int l = 0; l < L; ++l) { st.w[l] = w[l]; st.b[l] = b ? b[l] : nullptr; st.y[l] = y[l]; st.z[l] = z ? z[l] : nullptr; }
}

static int fs_check_fwd(const char* fn, const cfd_ffn_stack_args& a) {
    CFD_TRY(fs_check(fn, a.R, a.dims, a.L, a.act));
    CFD_REQUIRE(a.x && a.w && a.y, CFD_ERR_INVALID_ARG, "%s: NULL pointer", fn);
    for (int l = 0; l < a.L; ++l) {
        CFD_REQUIRE(a.w[l] && a.y[l], CFD_ERR_INVALID_ARG, "%s: NULL weight / output of layer %d", fn, l);
        CFD_REQUIRE(a.act < 3 || !(l + 1 < a.L || a.act_last) || (a.z && a.z[l]), CFD_ERR_INVALID_ARG,
                    "%s: gelu / swish need the pre-activation buffer of layer %d", fn, l);
    }
    return CFD_OK;
}

static int fs_fwd(const char* fn, int n, const cfd_ffn_stack_args* a, hipStream_t s) {
    CFD_REQUIRE(a && n >= 1 && n <= FS_MAXS, CFD_ERR_INVALID_ARG, "%s: 1 .. %d stacks per call", fn, FS_MAXS);
    FfnStackSet set{};
    int ld = 0, blocks = 0;
    double fl = 0.0, by = 0.0;
    for (int i = 0; i < n; ++i) {
        CFD_TRY(fs_check_fwd(fn, a[i]));
        if (a[i].R == 0) continue;  // an empty stack takes no workgroup
        const int k = set.ns++;
        fs_fill(set.st[k], a[i].w, a[i].b, a[i].y, a[i].z, a[i].dims, a[i].L, a[i].act, a[i].act_last);
        set.x[k] = a[i].x;
        set.R[k] = a[i].R;
        set.blk0[k] = blocks;
        blocks += (a[i].R + FS_ROWS - 1) / FS_ROWS;
        const int l_ = fs_ld(a[i].dims, a[i].L);
        if (l_ > ld) ld = l_;
        by += 4.0 * a[i].R * a[i].dims[0];
        for (int l = 0; l < a[i].L; ++l) {
            fl += 2.0 * a[i].R * a[i].dims[l] * a[i].dims[l + 1];
            by += 4.0 * a[i].R * a[i].dims[l + 1] + 4.0 * a[i].dims[l] * a[i].dims[l + 1];
        }
    }
    if (set.ns == 0) return CFD_OK;
    set.blk0[set.ns] = blocks;
    CFD_PROF_W("k_ffn_stack_fwd", s, by, fl);
    hipLaunchKernelGGL(k_ffn_stack_fwd, dim3(blocks), dim3(256), fs_lds_bytes(ld), s, set, ld);
    CFD_LAUNCH_CHECK(fn);
    return CFD_OK;
}

extern "C" int cfd_ffn_stacks_fwd(int n, const cfd_ffn_stack_args* stacks, void* stream) {
    return fs_fwd("cfd_ffn_stacks_fwd", n, stacks, (hipStream_t)stream);
}

extern "C" int cfd_ffn_stack_fwd(const float* x, const float* const* w, const float* const* b, float* const* y, float* const* z,
                                 int R, const int* dims, int L, int act, int act_last, void* stream) {
    cfd_ffn_stack_args a{};
    a.x = x; a.w = w; a.b = b; a.y = y; a.z = z; a.R = R; a.dims = dims; a.L = L; a.act = act; a.act_last = act_last;
    return fs_fwd("cfd_ffn_stack_fwd", 1, &a, (hipStream_t)stream);
}

// workspace: dZ_l for every layer (R x dims[l+1] floats each, 256-byte aligned), then FS_KSPLIT partial slices of all gradients
static size_t fs_ws_layout(int R, const int* dims, int L, size_t* dz_off, int* poff, int* pstride) {
    size_t off = 0;
    for (int l = 0; l < L; ++l) {
        if (dz_off) dz_off[l] = off;
        off += cfd_align_up((size_t)R * dims[l + 1] * sizeof(float), 256);
    }
    int po = 0;
    for (int l = 0; l < L; ++l) {
        if (poff) poff[l] = po;
        po += dims[l + 1] * dims[l] + dims[l + 1];
    }
    po = (po + 3) & ~3;
    if (pstride) *pstride = po;
    if (dz_off) dz_off[L] = off;
    return off + (size_t)FS_KSPLIT * po * sizeof(float);
}

extern "C" size_t cfd_ffn_stack_bwd_workspace_bytes(int R, const int* dims, int L) {
    if (!dims || R <= 0 || L < 1 || L > FS_MAXL) return 0;
    return fs_ws_layout(R, dims, L, nullptr, nullptr, nullptr);
}

static int fs_bwd(const char* fn, int n, const cfd_ffn_stack_args* a, hipStream_t s) {
    CFD_REQUIRE(a && n >= 1 && n <= FS_MAXS, CFD_ERR_INVALID_ARG, "%s: 1 .. %d stacks per call", fn, FS_MAXS);
    FfnStackSet set{};
    FfnStackBwdSet bset{};
    FfnStackGradsSet gset{};
    int ld = 0, blocks = 0, layers = 0, emax = 0, tmax = 0;
    double fl = 0.0, by_chain = 0.0, by_wg = 0.0;
    for (int i = 0; i < n; ++i) {
        const cfd_ffn_stack_args& t = a[i];
        CFD_TRY(fs_check(fn, t.R, t.dims, t.L, t.act));
        CFD_REQUIRE(t.x && t.gy && t.w && t.y && t.gw && t.ws, CFD_ERR_INVALID_ARG, "%s: NULL pointer (stack %d)", fn, i);
        CFD_REQUIRE(t.R >= 1, CFD_ERR_INVALID_ARG, "%s: empty batch (stack %d)", fn, i);
        for (int l = 0; l < t.L; ++l) {
            CFD_REQUIRE(t.w[l] && t.y[l] && t.gw[l], CFD_ERR_INVALID_ARG, "%s: NULL weight / output / gradient of layer %d", fn, l);
            CFD_REQUIRE(t.act < 3 || !(l + 1 < t.L || t.act_last) || (t.z && t.z[l]), CFD_ERR_INVALID_ARG,
                        "%s: gelu / swish need the pre-activation buffer of layer %d", fn, l);
        }
        const int k = set.ns++;
        fs_fill(set.st[k], t.w, nullptr, t.y, t.z, t.dims, t.L, t.act, t.act_last);
        set.x[k] = t.x;
        set.R[k] = t.R;
        set.blk0[k] = blocks;
        set.lay0[k] = layers;
        blocks += (t.R + FS_ROWS - 1) / FS_ROWS;
        layers += t.L;
        size_t dz_off[FS_MAXL + 1];
        fs_ws_layout(t.R, t.dims, t.L, dz_off, bset.bw[k].poff, &bset.bw[k].pstride);
        for (int l = 0; l < t.L; ++l) bset.bw[k].dz[l] = (float*)((char*)t.ws + dz_off[l]);
        bset.part[k] = (float*)((char*)t.ws + dz_off[t.L]);
        bset.gy[k] = t.gy;
        bset.gx[k] = t.gx;
        const int l_ = fs_ld(t.dims, t.L);
        if (l_ > ld) ld = l_;
        by_chain += 4.0 * t.R * (t.dims[0] + t.dims[t.L]);
        for (int l = 0; l < t.L; ++l) {
            gset.g[k].gw[l] = t.gw[l];
            gset.g[k].gb[l] = t.gb ? t.gb[l] : nullptr;
            const int e = t.dims[l + 1] * t.dims[l] + t.dims[l + 1];
            const int tt = ((t.dims[l] + 1 + 15) / 16) * ((t.dims[l + 1] + 15) / 16);
            if (e > emax) emax = e;
            if (tt > tmax) tmax = tt;
            fl += 2.0 * t.R * t.dims[l] * t.dims[l + 1];
            by_chain += 8.0 * t.R * t.dims[l + 1] + 4.0 * t.dims[l] * t.dims[l + 1];
            by_wg += 4.0 * t.R * (t.dims[l] + t.dims[l + 1]) + 4.0 * t.dims[l] * t.dims[l + 1];
        }
    }
    set.blk0[set.ns] = blocks;
    set.lay0[set.ns] = layers;
    {
        CFD_PROF_W("k_ffn_stack_bwd_chain", s, by_chain, fl);
        hipLaunchKernelGGL(k_ffn_stack_bwd_chain, dim3(blocks), dim3(256), fs_lds_bytes(ld), s, set, bset, ld);
    }
    CFD_LAUNCH_CHECK(fn);
    {
        CFD_PROF_W("k_ffn_stack_wgrad", s, by_wg, fl);
        const int items = tmax * layers * FS_KSPLIT, per_xcd = (items + 7) / 8;
        hipLaunchKernelGGL(k_ffn_stack_wgrad, dim3(8 * per_xcd), dim3(64), 0, s, set, bset, tmax, layers, per_xcd);
    }
    CFD_LAUNCH_CHECK(fn);
    CFD_PROF_W("k_ffn_stack_reduce", s, 0.0, 0.0);
    hipLaunchKernelGGL(k_ffn_stack_reduce, dim3((emax + 255) / 256, layers), dim3(256), 0, s, set, bset, gset);
    CFD_LAUNCH_CHECK(fn);
    return CFD_OK;
}

extern "C" int cfd_ffn_stacks_bwd(int n, const cfd_ffn_stack_args* stacks, void* stream) {
    return fs_bwd("cfd_ffn_stacks_bwd", n, stacks, (hipStream_t)stream);
}

extern "C" int cfd_ffn_stack_bwd(const float* x, const float* gy, const float* const* w, float* const* y, float* const* z,
                                 float* const* gw, float* const* gb, float* gx, void* ws, int R, const int* dims, int L, int act,
                                 int act_last, void* stream) {
    cfd_ffn_stack_args a{};
    a.x = x; a.w = w; a.y = y; a.z = z; a.R = R; a.dims = dims; a.L = L; a.act = act; a.act_last = act_last;
    a.gy = gy; a.gw = gw; a.gb = gb; a.gx = gx; a.ws = ws;
    return fs_bwd("cfd_ffn_stack_bwd", 1, &a, (hipStream_t)stream);
}
