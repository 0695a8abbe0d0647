// Dense layers of the DeepONet family on the CDNA4 matrix pipe (exact-fp32 v_mfma_f32_16x16x4_f32):
//   nn.Linear + activation stacks  (Ffn, src/models/ffn.py:12-35)             -> cfd_linear_fwd / cfd_linear_bwd
//   branch x trunk inner product   (src/models/auto_deeponet.py:127-135,
//                                   src/models/deeponet.py:204-205)           -> cfd_deeponet_inner_fwd / _bwd
// The reference materialises the (b, k, p) broadcast product (879 MB at B=512 on the 66x65 lattice) and sums it; here
// it is one GEMM whose epilogue adds the bias and gathers the residual u[b, query].  All of these are instances of
// one LDS-tiled kernel C = epi(op(A) op(B)): 64x64 block tile, 4 waves x (2x2) MFMA tiles, K in slabs of 16 that are
// prefetched into registers while the previous slab is on the matrix pipe.
#include "cfd_common.h"
#include <type_traits>

#define GT 64   // block tile edge (M and N) of the small-problem kernel: 4 waves x (2 x 2) MFMA tiles
#define GK 16   // K slab

struct GemmEpi {
    int mode;             // 0: store; 1: + bias[n], activation; 2: + bias[0] (+ resid[m*ldr + (qidx ? qidx[n] : n)]);
                          // 3: * act'(resid[m*ldr + n] as the layer output, gz_z[m*ldr + n] as its pre-activation): the input gradient of
                          //    a Linear layer leaves the GEMM as the PREVIOUS layer's dZ (no separate activation-gradient pass)
    int act;              // mode 1: 0 none, 1 relu, 2 tanh, 3 gelu (exact erf), 4 swish
    const float* bias;
    float* preact;        // mode 1: pre-activation copy (needed by the gelu / swish derivative), may be NULL
    const float* resid;
    const int* qidx;
    int ldr;
    const float* gz_z;    // mode 3: pre-activations (gelu / swish), else NULL
    int ones1;            // != 0: column ones1 - 1 of C is VIRTUAL -- opB has a row of ones there (the column sums of opA^T: a Linear layer's
    float* out2;          //       bias gradient rides in its weight-gradient product) -- and goes to out2[m] instead of C (ldc = N - 1)
};

__device__ __forceinline__ float cfd_act(float z, int act) {
    switch (act) {
        case 1: return z > 0.f ? z : 0.f;
        case 2: return tanhf(z);
        case 3: return cfd_gelu(z);
        case 4: return z * cfd_rcpf(1.f + cfd_expf(-z));
        default: return z;
    }
}

// d act / d z from the layer output y (relu, tanh) or its pre-activation z (gelu, swish)
__device__ __forceinline__ float cfd_act_grad(float y, float z, int act) {
    switch (act) {
        case 1: return y > 0.f ? 1.f : 0.f;
        case 2: return 1.f - y * y;
        case 3: return cfd_gelu_grad(z);
        case 4: { const float s = cfd_rcpf(1.f + cfd_expf(-z)); return s * (1.f + z * (1.f - s)); }
        default: return 1.f;
    }
}

// One (64 NR) x 16 slab of op(X) -- the rows [row0, row0 + 64 NR) of a block tile, the reduction indices [k0, k0 + 16) -- in
// registers and its place in LDS.  KCONT: X is stored [rows][K] (k-contiguous), LDS layout [row][24] (16 k + 8 floats of padding:
// the operand fetch is ONE conflict-free ds_read_b128 per MFMA tile and slab, the staging store one ds_write_b128); else X is
// stored [K][rows], LDS layout [k][64 NR + 4] (staging: ds_write_b128 along the rows; operand fetch: ds_read_b32, the four k rows
// of an MFMA step 16 banks apart).  MFMA step kk of a slab contracts the reduction indices 4 q + kk of its lanes (q = lane / 16) --
// any assignment works as long as both operands use the same one, and this one makes a k-contiguous operand's four values of a
// lane one 16-byte unit.  Global loads are raw buffer loads of 16 bytes (4 bytes when the row length is not a multiple of four):
// out-of-range rows / reduction indices carry CFD_BUF_OOB and come back as zeros, no branch per lane.
template <bool KCONT, int NR>
struct SlabRegs {
    static constexpr int ROWS = 64 * NR;  // one 16-byte unit per thread and round of 64 rows
    static constexpr int LD = KCONT ? 24 : ROWS + 4;
    static constexpr int FLOATS = KCONT ? ROWS * 24 : GK * (ROWS + 4);
    f32x4 v[NR];
    // ones: index of a virtual row of ones behind the matrix' rows (rows <= ones), or -1
    __device__ __forceinline__ void load(const float* __restrict__ X, int ld, int rows, int K, int row0, int k0, int tid, bool vec,
                                         int ones = -1) {
        // resource from the slab's first element to the last element of the matrix (clamped to 2^31 bytes: lane offsets stay below
        // ROWS * ld * 4, launch_gemm bounds ld)
        const float* base = KCONT ? X + (size_t)row0 * ld + k0 : X + (size_t)k0 * ld + row0;
        const long ext = KCONT ? (long)(rows - row0 - 1) * ld + (K - k0) : (long)(K - k0 - 1) * ld + (rows - row0);
        const CfdBuf bf = cfd_buf(base, ext <= 0 ? 0u : (ext >= (1L << 29) ? 0x80000000u : (unsigned)(4 * ext)));
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            // (r, c): row within the block tile and reduction index within the slab of this thread's unit; the unit runs along k
            // (KCONT) or along the rows
            const int r = KCONT ? 64 * h + (tid >> 2) : 64 * h + (tid & 15) * 4;
            const int c = KCONT ? (tid & 3) * 4 : tid >> 4;
            const bool ok = row0 + r < rows && k0 + c < K;
            const unsigned off = 4u * (unsigned)(KCONT ? r * ld + c : c * ld + r);
            if (vec) {
                v[h] = cfd_buf_ld4(bf, ok ? off : CFD_BUF_OOB, 0);
            } else {
                const int left = KCONT ? K - (k0 + c) : rows - (row0 + r);  // elements of the unit inside the matrix
#pragma unroll
                for (int j = 0; j < 4; ++j) v[h][j] = cfd_buf_ld(bf, ok && j < left ? off + 4u * j : CFD_BUF_OOB, 0);
            }
            if constexpr (!KCONT) {
                if (ones >= 0) {  // (uniform)
                    const float one = k0 + c < K ? 1.f : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (row0 + r + j == ones) v[h][j] = one;
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* s, int tid) const {
#pragma unroll
        for (int h = 0; h < NR; ++h) {
            if constexpr (KCONT) *reinterpret_cast<f32x4*>(s + (64 * h + (tid >> 2)) * 24 + (tid & 3) * 4) = v[h];
            else *reinterpret_cast<f32x4*>(s + (tid >> 4) * LD + 64 * h + (tid & 15) * 4) = v[h];
        }
    }
    // this lane's operand values of 16-row tile `tile` for the four MFMA steps of the slab
    __device__ static __forceinline__ f32x4 fetch(const float* s, int tile, int q, int n) {
        if constexpr (KCONT) {
            return *reinterpret_cast<const f32x4*>(s + (16 * tile + n) * 24 + 4 * q);
        } else {
            f32x4 f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) f[kk] = s[(4 * q + kk) * LD + 16 * tile + n];
            return f;
        }
    }
};

// C[M][N] = epi( sum_k opA(m,k) opB(k,n) );  AT: A stored [K][M] (else [M][K]);  BT: B stored [N][K] (else [K][N]).
// Four waves as a WGM x WGN grid, WTM x WTN MFMA tiles per wave: block tile (16 WTM WGM) x (16 WTN WGN).  Shapes built:
//   2 x 2 waves of 2 x 2 tiles:   64 x 64   THE DEFAULT: 85 registers, 21-25 KB of LDS, five to seven blocks per CU
//   2 x 2 waves of 4 x 4 tiles:  128 x 128  (gemm_tile = 128) half the operand reads per flop
//   4 x 1 waves of 2 x 8 / 2 x 16 tiles: 128 rows x ALL columns up to 128 / 256 (gemm_tile = 1): A read once, 13 column tiles for N = 200
//     where the square blocks cover 16
//   4 x 2 waves (EIGHT waves, the first four stage) of 2 x 2 tiles: 128 x 64 (gemm_tile = 8): 25 % fewer operand loads per flop at the
//     64 x 64 kernel's registers per wave: +1 .. +6 % slower
// The large shapes are SLOWER on every product of the benchmark (131 k x 200 x 200 of the Auto-FFN: 64 x 64 3.41 ms of k_gemm per
// step, 128 x 128 3.83-3.97, all-columns 3.62-5.10; profiles/r04g_gemm_counters.txt): the kernel's time is the SUM of its parts -- the
// MFMAs, the global loads, the epilogue, the LDS / barrier skeleton, measured by taking each out -- not their maximum, and only many
// small blocks per CU overlap them; they stay in the build behind the knob, with their tests.  Row / column tiles beyond M / N are
// skipped.
// K in slabs of 16 through a three-deep pipeline: while slab s is on the matrix pipe, slab s + 1 -- loaded during slab s - 1 -- goes
// from registers to the other LDS buffer (the stores are issued in front of the MFMAs and complete beside them) and slab s + 2 is
// on its way from memory; one barrier per slab.
template <bool AT, bool BT, int WGM, int WTM, int WTN, int NW = 4, bool ONES = false>
__global__ __launch_bounds__(64 * NW, 2) void k_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                                 float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                                 GemmEpi epi, int slabs_per_split, int nbx, int nby, int nbz) {
    constexpr int WGN = NW / WGM;  // (NW = 8: the first four waves stage the slabs, all eight feed the matrix pipe)
    constexpr int BM = 16 * WTM * WGM, BN = 16 * WTN * WGN;  // block tile
    using RA = SlabRegs<!AT, BM / 64>;  // A is k-contiguous unless transposed
    using RB = SlabRegs<BT, BN / 64>;   // B is k-contiguous only when stored transposed
    CFD_DYN_SHARED(f32x4, s_dyn);  // (the all-columns tile holds 72 KB: beyond the static limit)
    float* const s_a[2] = {(float*)s_dyn, (float*)s_dyn + RA::FLOATS};
    float* const s_b[2] = {(float*)s_dyn + 2 * RA::FLOATS, (float*)s_dyn + 2 * RA::FLOATS + RB::FLOATS};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int wm = wave / WGN, wn = wave % WGN;
    // Block -> tile map.  Workgroups are dealt to the 8 XCDs round robin in launch order and each XCD has its own L2, so the blocks
    // that read the same operand rows -- the column blocks of one row tile; with split-K all tiles of one K range -- are made
    // CONSECUTIVE ON ONE XCD instead of consecutive in launch order.  Groups of `gs` blocks; group g = 8 (l / gs) + XCD for the
    // XCD's l-th block; the grid is padded to a multiple of 8 groups.  (Measured: no change on MI355X -- the re-reads of a plain
    // (x, y, z) grid are served by the memory-side cache -- kept because it halves the HBM fetches.)
    const int gs = nbz > 1 ? nbx * nby : nbx, ngroups = nbz > 1 ? nbz : nby;
    const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
    const int grp = 8 * (l / gs) + xcd, t = l % gs;
    if (grp >= ngroups) return;
    const int bx = t % nbx, by = nbz > 1 ? t / nbx : grp, bz = nbz > 1 ? grp : 0;
    const int m0 = by * BM, n0 = bx * BN;
    // (uniform) this wave's row / column tiles that hold an element of C: the others are skipped -- with several blocks per SIMD the
    // matrix-pipe time a wave does not use goes to its neighbours (N = 200 on 64-wide blocks: 13 live column tiles of 16)
    const int ntm = (M - m0 - 16 * WTM * wm + 15) / 16, ntn = (N - n0 - 16 * WTN * wn + 15) / 16;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) acc[i][j] = zero;
    // split-K: block bz owns slabs [s0, s1) and writes its partial tile to C + bz*M*ldc (epilogue in k_splitk_reduce)
    const int nslab_all = (K + GK - 1) / GK;
    const int s0 = bz * slabs_per_split;
    const int nslab = (s0 + slabs_per_split < nslab_all ? s0 + slabs_per_split : nslab_all) - s0;
    C += (size_t)bz * M * ldc;
    // the virtual column of C / row of ones of opB: its own instantiation (ONES) -- the patch in the operand loader costs registers,
    // and this kernel lives on its five to seven blocks per CU
    const int ones = ONES ? epi.ones1 - 1 : -1;
    const int Nb = ones >= 0 ? N - 1 : N;      // opB's real extent
    const bool va = ((AT ? M : K) & 3) == 0, vb = ((BT ? K : Nb) & 3) == 0;  // (uniform) 16-byte units never straddle a row end
    RA ra0, ra1;
    RB rb0, rb1;
    const bool stager = NW == 4 || tid < 256;  // (wave-uniform)
    const auto load = [&](RA& ra, RB& rb, int slab) {
        if (!stager) return;
        ra.load(A, lda, M, K, m0, (s0 + slab) * GK, tid, va);
        rb.load(B, ldb, Nb, K, n0, (s0 + slab) * GK, tid, vb, ones);
    };
    const auto commit = [&](const RA& ra, const RB& rb, int buf) {
        if (!stager) return;
        ra.store(s_a[buf], tid);
        rb.store(s_b[buf], tid);
    };
    const auto mma = [&](const float* sa, const float* sb) {
        f32x4 fa[WTM];
#pragma unroll
        for (int i = 0; i < WTM; ++i) fa[i] = RA::fetch(sa, WTM * wm + i, q, n);
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            if (j >= ntn) break;
            const f32x4 fb = RB::fetch(sb, WTN * wn + j, q, n);
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                if (i >= ntm) break;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc[i][j] = cfd_mfma16x16x4(fa[i][kk], fb[kk], acc[i][j]);
            }
        }
    };
    load(ra0, rb0, 0);
    if (nslab > 1) load(ra1, rb1, 1);
    commit(ra0, rb0, 0);
    __syncthreads();
    for (int s = 0; s < nslab; s += 2) {
        // slab s from buffer 0; slab s + 1 waits in (ra1, rb1)
        if (s + 1 < nslab) commit(ra1, rb1, 1);  // (buffer 1 was last read before the previous barrier)
        if (s + 2 < nslab) load(ra0, rb0, s + 2);
        mma(s_a[0], s_b[0]);
        __syncthreads();
        if (s + 1 >= nslab) break;
        // slab s + 1 from buffer 1; slab s + 2 waits in (ra0, rb0)
        if (s + 2 < nslab) commit(ra0, rb0, 0);
        if (s + 3 < nslab) load(ra1, rb1, s + 3);
        mma(s_a[1], s_b[1]);
        __syncthreads();
    }
    // acc[i][j][r] = C[m0 + 16 (WTM wm + i) + 4q + r][n0 + 16 (WTN wn + j) + n]
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            const int col = n0 + 16 * (WTN * wn + j) + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * (WTM * wm + i) + 4 * q + r;
                if (row < M && col < N) {
                    float v = acc[i][j][r];
                    if (epi.mode == 1) {
                        if (epi.bias) v += epi.bias[col];
                        if (epi.preact) epi.preact[(size_t)row * ldc + col] = v;
                        v = cfd_act(v, epi.act);
                    } else if (epi.mode == 2) {
                        v += epi.bias[0];
                        if (epi.resid) v += epi.resid[(size_t)row * epi.ldr + (epi.qidx ? epi.qidx[col] : col)];
                    } else if (epi.mode == 3) {
                        const size_t o = (size_t)row * epi.ldr + col;
                        v *= cfd_act_grad(epi.resid[o], epi.gz_z ? epi.gz_z[o] : 0.f, epi.act);
                    }
                    if (ONES && epi.out2 && col == epi.ones1 - 1) epi.out2[row] = v;  // (split-K partials keep the column: out2 == NULL)
                    else C[(size_t)row * ldc + col] = v;
                }
            }
        }
}

// out[m][n] = epi( sum_z part[z][m][n] ), fixed summation order (deterministic)
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ part, float* __restrict__ C, int M, int N,
                                                       int ldc, int splits, GemmEpi epi) {
    const size_t total = (size_t)M * N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(e / N), col = (int)(e - (size_t)row * N);
        float v = 0.f;
        int z = 0;
        // eight partials per trip, loaded before the first add (one per trip = `splits` exposed latencies per thread: 18 us for the 128
        // partial tiles of a 100 x 100 weight gradient); added in split order as before
        for (; z + 8 <= splits; z += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(z + u) * total + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; z < splits; ++z) v += part[(size_t)z * total + e];
        if (epi.mode == 1) {
            if (epi.bias) v += epi.bias[col];
            if (epi.preact) epi.preact[(size_t)row * ldc + col] = v;
            v = cfd_act(v, epi.act);
        } else if (epi.mode == 2) {
            v += epi.bias[0];
            if (epi.resid) v += epi.resid[(size_t)row * epi.ldr + (epi.qidx ? epi.qidx[col] : col)];
        } else if (epi.mode == 3) {
            const size_t o = (size_t)row * epi.ldr + col;
            v *= cfd_act_grad(epi.resid[o], epi.gz_z ? epi.gz_z[o] : 0.f, epi.act);
        }
        if (epi.out2 && col == epi.ones1 - 1) epi.out2[row] = v;
        else C[(size_t)row * ldc + col] = v;
    }
}

// Skinny products (few 64x64 output tiles, long K -- e.g. the 100x100 weight gradients over 4290 lattice points) would
// leave most CUs idle: K is split over blockIdx.z into partial tiles that a second kernel sums and finishes.
static int gemm_splits(int M, int N, int K) {
    const long tiles = (long)((M + GT - 1) / GT) * ((N + GT - 1) / GT);
    const int nslab = (K + GK - 1) / GK;
    const int forced = cfd_tune_get(CFD_TUNE_GEMM_SPLITS);
    if (forced > 0) return forced < nslab ? forced : nslab;
    // (round 5) 128 .. 255 tiles still leave every CU with a single four-wave block: with K >= 512 the split pays there too -- the
    // first-layer weight gradient of the Auto-DeepONet branch net (100 x 4295 over K = 512: 136 tiles) 39.8 -> 22.3 us at 8 splits,
    // tools/exp/gemm_shapes.py
    if (tiles >= 128 && tiles < 256 && nslab >= 32) return 8 < nslab / 4 ? 8 : nslab / 4;
    if (tiles >= 128 || nslab < 8) return 1;
    // four workgroups per CU: with one (256 in all: round 1's target) every SIMD holds a single wave and nothing hides its LDS and
    // barrier latencies -- the 200 x 200 weight gradients over 131 k rows of the Auto-FFN ran at 30 % of the matrix pipe, half the
    // rate of the forward product of the same size (profiles/r04g_gemm_counters.txt)
    long s = (1024 + tiles - 1) / tiles;
    const long smax = nslab / 16 > 256 / tiles ? nslab / 16 : (256 + tiles - 1) / tiles;  // beyond one block per CU: >= 16 slabs (256 k) per split
    if (s > smax) s = smax;
    if (s > nslab / 4) s = nslab / 4;  // at least 4 slabs (64 k) per split
    if (s > 128) s = 128;
    return s < 1 ? 1 : (int)s;
}

static size_t gemm_ws_bytes(int M, int N, int K) {
    const int s = gemm_splits(M, N, K);
    return s > 1 ? cfd_align_up((size_t)s * M * N * sizeof(float), 256) : 0;
}

// Block tile of a product: 64 x 64 unless the gemm_tile knob asks for 128 x 128 or (1) 128 rows x all columns (see k_gemm).
static int gemm_tile_kind(int M, int N, int splits) {
    const int t = cfd_tune_get(CFD_TUNE_GEMM_TILE);
    if (t == 64) return 64;
    if (t == 128) return 128;
    if (t == 1) return N <= 128 ? 1 : 2;  // (the all-columns kernels on any shape; N > 256 takes several column blocks)
    if (t == 8) return 8;                 // 128 x 64 blocks of eight waves
    (void)M; (void)splits;
    return 64;
}

static int launch_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int at,
                       int bt, const GemmEpi& epi, void* ws, hipStream_t st, const char* what) {
    if (M == 0 || N == 0) return CFD_OK;
    CFD_REQUIRE(lda < (1 << 22) && ldb < (1 << 22), CFD_ERR_UNSUPPORTED, "%s: leading dimensions beyond 2^22 (lda %d, ldb %d)", what, lda, ldb);
    int splits = ws ? gemm_splits(M, N, K) : 1;
    const int nslab = (K + GK - 1) / GK;
    const int per = (nslab + splits - 1) / splits;
    splits = (nslab + per - 1) / per;
    const int kind = gemm_tile_kind(M, N, splits);
    const int bm = kind == 64 ? 64 : 128, bn = (kind == 64 || kind == 8) ? 64 : (kind == 1 ? 128 : (kind == 2 ? 256 : 128));
    const int nbx = (N + bn - 1) / bn, nby = (M + bm - 1) / bm;
    const long ngroups = splits > 1 ? splits : nby, gsz = splits > 1 ? (long)nbx * nby : nbx;
    const long nblocks = ((ngroups + 7) / 8) * 8 * gsz;  // (see k_gemm: block -> tile map)
    CFD_REQUIRE(nblocks < (1L << 31), CFD_ERR_UNSUPPORTED, "%s: %ld blocks", what, nblocks);
    const dim3 grid((unsigned)nblocks);
    GemmEpi e0{};
    e0.ones1 = epi.ones1;  // (the split kernels still load the row of ones; their partial tiles keep its column)
    const GemmEpi& ek = splits > 1 ? e0 : epi;
    float* Ck = splits > 1 ? (float*)ws : C;
    const int ldk = splits > 1 ? N : ldc;
    {
        CFD_PROF_W("k_gemm", st, 4.0 * ((double)M * K + (double)K * N + (double)M * N), 2.0 * M * (double)N * K);
#define GEMM_K(AT_, BT_, G_, TM_, TN_, NW_)                                                                                                \
    do {                                                                                                                                  \
        constexpr int bm_ = 16 * TM_ * G_, bn_ = 16 * TN_ * (NW_ / G_);                                                                   \
        constexpr size_t lds_ = 2 * sizeof(float) * (SlabRegs<!AT_, bm_ / 64>::FLOATS + SlabRegs<BT_, bn_ / 64>::FLOATS);                \
        static bool attr_set = false;                                                                                                     \
        if (!attr_set) {                                                                                                                  \
            (void)hipFuncSetAttribute((const void*)k_gemm<AT_, BT_, G_, TM_, TN_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
            attr_set = true;                                                                                                              \
        }                                                                                                                                 \
        hipLaunchKernelGGL((k_gemm<AT_, BT_, G_, TM_, TN_, NW_>), grid, dim3(64 * NW_), lds_, st, A, B, Ck, M, N, K, lda, ldb, ldk, ek, per, nbx,   \
                           nby, splits);                                                                                                  \
    } while (0)
#define GEMM_L(AT_, BT_)                                  \
    do {                                                  \
        if (kind == 64) GEMM_K(AT_, BT_, 2, 2, 2, 4);        \
        else if (kind == 128) GEMM_K(AT_, BT_, 2, 4, 4, 4);  \
        else if (kind == 8) GEMM_K(AT_, BT_, 4, 2, 2, 8);    \
        else if (kind == 1) GEMM_K(AT_, BT_, 4, 2, 8, 4);    \
        else GEMM_K(AT_, BT_, 4, 2, 16, 4);                  \
    } while (0)
        if (ek.ones1) {  // (the weight-gradient form with its bias column: 64 x 64 blocks only)
            CFD_REQUIRE(at && !bt && kind == 64, CFD_ERR_UNSUPPORTED, "%s: the ones column rides in the A^T B form on 64 x 64 blocks only", what);
            constexpr size_t lds1 = 2 * sizeof(float) * (SlabRegs<false, 1>::FLOATS + SlabRegs<false, 1>::FLOATS);
            hipLaunchKernelGGL((k_gemm<true, false, 2, 2, 2, 4, true>), grid, dim3(256), lds1, st, A, B, Ck, M, N, K, lda, ldb, ldk, ek, per, nbx,
                               nby, splits);
        }
        else if (!at && !bt) GEMM_L(false, false);
        else if (!at && bt) GEMM_L(false, true);
        else if (at && !bt) GEMM_L(true, false);
        else GEMM_L(true, true);
#undef GEMM_L
#undef GEMM_K
    }
    CFD_LAUNCH_CHECK(what);
    if (splits > 1) {
        CFD_PROF_W("k_splitk_reduce", st, 0.0, 0.0);  // split-K partials are an implementation detail
        size_t blocks = ((size_t)M * N + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, C, M, N, ldc, splits, epi);
        CFD_LAUNCH_CHECK(what);
    }
    return CFD_OK;
}


// ------------------------------------------------------------------------------------------------------
// Tall products C[M][N] = epi(A[M][K] Bw) with a SMALL second operand (a Linear layer's weights: N, K <= 1024) on the bf16 matrix pipe
// with THREE-PIECE operands (round 6; cfd_common.h: six v_mfma_f32_16x16x32_bf16 per product term, everything down to 2^-24 -- the
// fp32-exact class the convolution stack runs on, so the ReLU networks see no kink crossings).  The 131 k x 200 x 200 products of the
// Auto-FFN ran at 42 % of the fp32 matrix pipe (167 us; the pipe's own time is 67 us); six 16-cycle MFMAs replace eight 32-cycle ones and
// the product becomes memory-bound (210 MB).
//   * Bw is split ONCE per call into MFMA fragment order by k_rowgemm6_prep: frag[((ks * NTall + nt) * 3 + piece) * 64 + lane],
//     lane (q, n): column 16 nt + n, reduction indices 32 ks + 8 q .. + 7.  TRANS: Bw[k][col] stored [k][col] (the input gradient gz w),
//     else stored [col][k] (the forward x w^T).
//   * workgroup = 64 rows (four waves x one 16-row tile) x ALL columns in passes of R6_NT column tiles; per K slab of 32 the pass's
//     fragments go global -> LDS directly (cfd_glds16) and the A rows come straight from global memory in operand order (a lane's eight
//     reduction indices are 32 contiguous bytes), split in registers.  The next slab's A loads are in flight during this slab's MFMAs;
//     four workgroups per CU cover each other's barriers and fragment loads.
// ------------------------------------------------------------------------------------------------------
#define R6_NT 13
#ifndef CFD_R6_EXP
#define CFD_R6_EXP 0  // timing experiments (results WRONG): 1 no MFMAs, 2 no A loads, 4 no fragment loads, 8 no stores
#endif
typedef cfd_u32x4 r6_u4;

template <bool TRANS>
__global__ __launch_bounds__(256) void k_rowgemm6_prep(const float* __restrict__ W, int ldw, r6_u4* __restrict__ frag, int N, int K,
                                                       int NTall, int nks) {
    const unsigned total = (unsigned)nks * NTall * 64;
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int lane = idx & 63, q = lane >> 4, n = lane & 15;
        const unsigned rest = idx >> 6;
        const int nt = rest % NTall, ks = rest / NTall;
        const int col = 16 * nt + n, k0 = 32 * ks + 8 * q;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (col < N && k0 + j < K) ? (TRANS ? W[(size_t)(k0 + j) * ldw + col] : W[(size_t)col * ldw + k0 + j]) : 0.f;
        const CfdSplit8x3 sp = cfd_split8x3(v);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) frag[((size_t)rest * 3 + pc) * 64 + lane] = __builtin_bit_cast(r6_u4, sp.p[pc]);
    }
}

// <NT, RT>: column tiles per pass, 16-row tiles per wave.  <13, 2>: up to 208 columns per pass on 128-row workgroups, two per CU;
// <7, 1>: layers of at most 112 columns (the DeepONets' width 100) on 64-row workgroups of ~100 registers, four per CU -- such a product
// is memory-bound (204 MB for 0.5 GFLOP-class work per layer) and needs the occupancy more than the fragment reuse
// NBUF: LDS buffers of a pass's fragments.  2: the next slab's fragments land while this slab's MFMAs run (one barrier per slab); 1: they are
// requested behind a second barrier -- half the LDS (39 KB at 13 column tiles), so that four light workgroups share a CU
template <int NT, int RT, int NBUF = 2>
__global__ __launch_bounds__(256, RT == 1 ? 4 : 2) void k_rowgemm6(const float* __restrict__ A, int lda, const r6_u4* __restrict__ frag,
                                                     float* __restrict__ C, int ldc, int M, int N, int K, GemmEpi epi, int NTall,
                                                     int nks) {
    constexpr int FB = NT * 192;  // 16-byte units of one pass's fragments per slab
    CFD_DYN_SHARED(f32x4, s_dyn);                         // (one extern array type per translation unit) [2][FB] 16-byte units
    r6_u4* const s_frag = reinterpret_cast<r6_u4*>(s_dyn);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int row0 = (int)blockIdx.x * (64 * RT) + wave * (16 * RT);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    // A rows of this lane (clamped: rows past M feed accumulator rows that are never stored)
    const float* arow[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = row0 + 16 * rt + n;
        arow[rt] = A + (size_t)(r < M ? r : M - 1) * lda;
    }
    for (int nt0 = 0; nt0 < NTall; nt0 += NT) {
        const int ntp = NTall - nt0 < NT ? NTall - nt0 : NT;  // (uniform) live column tiles of this pass
        const int fcount = ntp * 192;
        f32x4 acc[RT][NT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[rt][nt] = zero;
        f32x4 ar[RT][2];
        // loads of slab ks: the A rows unconditional from clamped addresses (a 16-byte unit past K is zeroed after the load: K % 4 == 0);
        // the pass's fragments straight into LDS buffer ks & 1 (cfd_glds16: 1 KB per wave instruction, no registers -- through
        // registers the 40 VGPRs of the fragment units spilled the kernel), rows of 64 units dealt to the four waves
        const auto issue_b = [&](int ks) {
            const r6_u4* fb = frag + ((size_t)ks * NTall + nt0) * 192;
            r6_u4* sbn = s_frag + (NBUF == 2 ? (ks & 1) * FB : 0);
            if constexpr (!(CFD_R6_EXP & 4))
            for (int u = wave; u < 3 * ntp; u += 4) cfd_glds16(fb + 64 * u + lane, sbn + 64 * u);
        };
        const auto issue = [&](int ks) {
            const int k0 = 32 * ks + 8 * q;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = k0 + 4 * h;
                    if constexpr (CFD_R6_EXP & 2) ar[rt][h] = f32x4{(float)k, 1.f, 2.f, (float)ks};
                    else ar[rt][h] = *reinterpret_cast<const f32x4*>(arow[rt] + (k < K ? k : K - 4));  // (zeroed at its use: a select here waits for the load)
                }
            if constexpr (NBUF == 2) issue_b(ks);
        };
        issue(0);
        if constexpr (NBUF == 1) issue_b(0);
        for (int ks = 0; ks < nks; ++ks) {
            const r6_u4* sb = s_frag + (NBUF == 2 ? (ks & 1) * FB : 0);
            CfdSplit8x3 as[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bool in0 = 32 * ks + 8 * q < K, in1 = 32 * ks + 8 * q + 4 < K;  // 16-byte units past K (last slab) are zeros
                const float x[8] = {in0 ? ar[rt][0][0] : 0.f, in0 ? ar[rt][0][1] : 0.f, in0 ? ar[rt][0][2] : 0.f, in0 ? ar[rt][0][3] : 0.f,
                                    in1 ? ar[rt][1][0] : 0.f, in1 ? ar[rt][1][1] : 0.f, in1 ? ar[rt][1][2] : 0.f, in1 ? ar[rt][1][3] : 0.f};
                as[rt] = cfd_split8x3(x);
            }
            __syncthreads();  // this slab's fragments have landed (requested one slab ago); the other buffer's readers are done
            if (ks + 1 < nks) issue(ks + 1);
            cfd_sched_fence();
            // fully unrolled with uniform guards (a `break` kept the loop rolled and the accumulators in scratch memory); the fragments of
            // column tile nt + 1 are requested before tile nt's MFMAs (the LDS latency of a tile's three reads was exposed per tile)
            bf16x8 bq[2][3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bq[0][pc] = __builtin_bit_cast(bf16x8, sb[pc * 64 + lane]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt < ntp) {
                    {   // unconditional (the last live tile re-reads itself): a guarded read joins the MFMAs behind a full lgkmcnt(0)
                        const int nn = nt + 1 < ntp ? nt + 1 : nt;
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) bq[(nt + 1) & 1][pc] = __builtin_bit_cast(bf16x8, sb[(nn * 3 + pc) * 64 + lane]);
                    }
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // cfd_mfma_bf16x6's products, small terms first
#pragma unroll
                    for (int p = 0; p < 6; ++p)
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            if constexpr (CFD_R6_EXP & 1) acc[rt][nt][p & 3] += (float)as[rt].p[PA[p]][rt] * (float)bq[nt & 1][PB[p]][p];
                            else acc[rt][nt] = cfd_mfma16x16x32_bf16(as[rt].p[PA[p]], bq[nt & 1][PB[p]], acc[rt][nt]);
                        }
                }
            }
            if constexpr (NBUF == 1) {
                if (ks + 1 < nks) {
                    __syncthreads();  // every wave has read this slab's fragments
                    issue_b(ks + 1);
                }
            }
        }
        __syncthreads();  // the next pass refills buffer 0
        // acc[rt][nt][r] = C[row0 + 16 rt + 4 q + r][16 (nt0 + nt) + n].  Stores through a raw buffer resource (32-bit offsets, rows / columns
        // past the matrix dropped by the range check: no branch and no 64-bit address per element -- 104 elements per lane and pass), the
        // bias once per column tile.
        const CfdBuf bC = cfd_buf(C, 4u * (unsigned)((size_t)(M - 1) * ldc + N));
        const CfdBuf bP = cfd_buf(epi.preact ? epi.preact : C, 4u * (unsigned)((size_t)(M - 1) * ldc + N));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (nt >= ntp) continue;
            const int col = 16 * (nt0 + nt) + n;
            const bool cok = col < N;
            const float bv = (epi.mode == 1 && epi.bias && cok) ? epi.bias[col] : 0.f;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = row0 + 16 * rt + 4 * q + r;
                    const bool ok = cok && row < M;
                    const unsigned off = ok ? 4u * (unsigned)(row * ldc + col) : CFD_BUF_OOB;
                    float v = acc[rt][nt][r];
                    if (epi.mode == 1) {
                        v += bv;
                        if (epi.preact) cfd_buf_st(bP, off, 0, v);
                        v = cfd_act(v, epi.act);
                    } else if (epi.mode == 3) {
                        const size_t o = (size_t)(ok ? row : 0) * epi.ldr + (ok ? col : 0);
                        v *= cfd_act_grad(epi.resid[o], epi.gz_z ? epi.gz_z[o] : 0.f, epi.act);
                    }
                    if (!(CFD_R6_EXP & 8) || v == 1.2345f) cfd_buf_st(bC, off, 0, v);
                }
        }
    }
}

static size_t rowgemm6_frag_bytes(int N, int K) {
    const long NTall = (N + 15) / 16, nks = (K + 31) / 32;
    return cfd_align_up((size_t)(nks * NTall * 3 * 64) * sizeof(r6_u4), 256);
}
// (A: M x K row-major with row stride lda; trans: see k_rowgemm6_prep)
static bool rowgemm6_covers(int M, int N, int K, const float* A, int lda, const GemmEpi& epi) {
    const int knob = cfd_tune_get(CFD_TUNE_GEMM_B3);  // 0 = never, 2 = at any row count (tests)
    if (knob == 0 || cfd_tune_get(CFD_TUNE_EXACT_FP32) == 1) return false;
    // tall, small second operand; by default from 160 columns / reduction indices on the <13, 2> shape (measured inside the legs, HBM-cold:
    // 131 k x 200 x 200 130 against 160-167 us, 131 k x 512 x 512 465 against 690 us; 256 k x 100 x 100 109 against 87 / 102 us there: the
    // pass machinery is sized for 13 column tiles and K = 100 pads its last 32-slab by 28 %), width ~100 on the light <7, 1> shape
    const bool narrow = N >= 96 && N <= 112 && K >= 96;  // the <7, 1> shape (DeepONet width 100: 75.9 us against 87 / 102 on k_gemm)
    const int lo = knob == 2 ? 16 : (narrow ? 96 : 160);
    if (M < (knob == 2 ? 1 : 4096) || N < lo || N > 1024 || K < lo || K > 1024) return false;
    if ((K & 3) || (lda & 3) || ((uintptr_t)A & 15)) return false;           // 16-byte units along a row
    if (epi.mode == 2 || epi.ones1 || epi.out2) return false;
    if ((size_t)M * (N > K ? N : K) * 4 >= (1ull << 31)) return false;  // 32-bit byte offsets into C (ldc <= max(N, K) at both call sites)
    return true;
}
static int rowgemm6_run(const float* A, int lda, const float* W, int ldw, bool trans, float* C, int ldc, int M, int N, int K,
                        const GemmEpi& epi, void* frag_ws, hipStream_t st, const char* what) {
    const int NTall = (N + 15) / 16, nks = (K + 31) / 32;
    const unsigned total = (unsigned)nks * NTall * 64;
    unsigned pb = (total + 255) / 256;
    if (pb > 512) pb = 512;
    if (trans) hipLaunchKernelGGL((k_rowgemm6_prep<true>), dim3(pb), dim3(256), 0, st, W, ldw, (r6_u4*)frag_ws, N, K, NTall, nks);
    else hipLaunchKernelGGL((k_rowgemm6_prep<false>), dim3(pb), dim3(256), 0, st, W, ldw, (r6_u4*)frag_ws, N, K, NTall, nks);
    CFD_LAUNCH_CHECK(what);
    CFD_PROF_W("k_rowgemm6", st, 4.0 * ((double)M * K + (double)K * N + (double)M * N), 2.0 * M * (double)N * K);
#define R6_LAUNCH(NT_, RT_, NB_)                                                                                                         \
    do {                                                                                                                                 \
        constexpr size_t lds = NB_ * (size_t)NT_ * 192 * sizeof(r6_u4);                                                                  \
        static bool attr_set = false;                                                                                                    \
        if (!attr_set) {                                                                                                                 \
            (void)hipFuncSetAttribute((const void*)k_rowgemm6<NT_, RT_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);          \
            attr_set = true;                                                                                                             \
        }                                                                                                                                \
        hipLaunchKernelGGL((k_rowgemm6<NT_, RT_, NB_>), dim3((unsigned)((M + 64 * RT_ - 1) / (64 * RT_))), dim3(256), lds, st, A, lda,        \
                           (const r6_u4*)frag_ws, C, ldc, M, N, K, epi, NTall, nks);                                                     \
    } while (0)
    // light shapes (one row tile per wave, four workgroups per CU): the product is memory-bound and its phases do not overlap inside a
    // workgroup -- 131 k x 200 x 200: 86 us on <13, 1, 1> against 118 on the first shape (two row tiles per wave, 235 VGPRs, two per CU)
    if (NTall <= 7) R6_LAUNCH(7, 1, 2);
    else R6_LAUNCH(R6_NT, 1, 1);
#undef R6_LAUNCH
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}


// ------------------------------------------------------------------------------------------------------
// The weight gradient of a tall Linear layer on the same operands: gw[n][k] = sum_m gz[m][n] x[m][k] (M rows, N, K <= 1024), and the
// bias gradient gb[n] = sum_m gz[m][n] as a virtual column K of ones behind x's columns (Kx = K + 1) when K is no multiple of 128.
// Both MFMA operands are COLUMNS of row-major matrices -- eight consecutive rows m of one column per lane -- so the blocks are
// transposed on their way into LDS: a staging item is (8 rows, 4 columns): eight 16-byte loads, regrouped in registers into four
// column octets, split into three bf16 pieces and written as whole fragment units [tile][piece][lane].
// Workgroup (256 threads) = a chunk of rows x up to 13 n-tiles (208 columns of gz) x 8 k-tiles (128 columns of x); wave w owns k-tiles
// 2 w, 2 w + 1 and all n-tiles: 26 accumulator tiles; per step of 32 rows 45 fragment reads feed 156 MFMAs per wave.  The next step's
// loads wait in registers while this step's MFMAs run.  Partial sums per chunk: part[chunk][n][Kx], reduced by k_splitk_reduce.
// ------------------------------------------------------------------------------------------------------
#define W6_NT 13  // n-tiles per workgroup
#define W6_KT 8   // k-tiles per workgroup (two per wave)
__global__ __launch_bounds__(256, 2) void k_rowwgrad6(const float* __restrict__ gz, const float* __restrict__ x, float* __restrict__ part,
                                                      int M, int N, int K, int Kx, int rows_per_chunk) {
    CFD_DYN_SHARED(f32x4, s_dyn);  // [A fragments W6_NT x 3 x 64 | B fragments W6_KT x 3 x 64] 16-byte units
    r6_u4* const s_a = reinterpret_cast<r6_u4*>(s_dyn);
    r6_u4* const s_b = s_a + W6_NT * 192;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = (int)blockIdx.z * (16 * W6_NT), k0 = (int)blockIdx.y * (16 * W6_KT);
    const int ntl = (N - n0 + 15) / 16 < W6_NT ? (N - n0 + 15) / 16 : W6_NT;     // (uniform) live n-tiles
    const int ktl = (Kx - k0 + 15) / 16 < W6_KT ? (Kx - k0 + 15) / 16 : W6_KT;   // live k-tiles
    const int m_beg = (int)blockIdx.x * rows_per_chunk;
    const int m_end = m_beg + rows_per_chunk < M ? m_beg + rows_per_chunk : M;
    // staging items of this thread: slot 0 = item tid, slot 1 = item tid + 256.  Items [0, 208): gz, (row octet o = i / 52, column group
    // cg = i % 52); items [208, 336): x, (o = j / 32, cg = j % 32) with j = i - 208.  A column group is four consecutive columns.
    constexpr int NA = 4 * (4 * W6_NT), NB = 4 * (4 * W6_KT);
    int it_oct[2], it_col[2], it_unit[2];  // row octet, first column (matrix coordinates), first LDS unit of the item's four columns
    int it_kind[2];                        // 0 gz, 1 x, 2 the column of ones, -1 none
    const float* it_base[2];
    int it_ld[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int i = tid + 256 * sl;
        it_kind[sl] = -1; it_oct[sl] = 0; it_col[sl] = 0; it_unit[sl] = 0; it_base[sl] = gz; it_ld[sl] = N;
        if (i < NA) {
            const int o = i / (4 * W6_NT), cg = i - o * (4 * W6_NT), col = n0 + 4 * cg;
            if (col < N) { it_kind[sl] = 0; it_oct[sl] = o; it_col[sl] = col; it_unit[sl] = ((4 * cg) >> 4) * 192 + o * 16 + ((4 * cg) & 15); }
        } else if (i < NA + NB) {
            const int j = i - NA, o = j / (4 * W6_KT), cg = j - o * (4 * W6_KT), col = k0 + 4 * cg;
            if (col < Kx) {
                it_kind[sl] = col < K ? 1 : 2; it_oct[sl] = o; it_col[sl] = col; it_base[sl] = x; it_ld[sl] = K;
                it_unit[sl] = W6_NT * 192 + ((4 * cg) >> 4) * 192 + o * 16 + ((4 * cg) & 15);
            }
        }
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[W6_NT][2];
#pragma unroll
    for (int nt = 0; nt < W6_NT; ++nt) acc[nt][0] = acc[nt][1] = zero;
    f32x4 rg[2][8];  // raw rows of the two items: rg[slot][row of the octet] = four consecutive columns
    const auto issue = [&](int m0) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (it_kind[sl] == 0 || it_kind[sl] == 1) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = m0 + 8 * it_oct[sl] + r;
                    rg[sl][r] = *reinterpret_cast<const f32x4*>(it_base[sl] + (size_t)(row < m_end ? row : m_end - 1) * it_ld[sl] + it_col[sl]);
                }
            }
        }
    };
    if (m_beg < m_end) issue(m_beg);
    for (int m0 = m_beg; m0 < m_end; m0 += 32) {
        __syncthreads();  // the previous step's fragments are consumed
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            if (it_kind[sl] < 0) continue;
            // four columns x eight rows -> four column octets (rows past the chunk are zeros; the virtual column: ones)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const bool in = m0 + 8 * it_oct[sl] + r < m_end;
                    v[r] = !in ? 0.f : (it_kind[sl] == 2 ? (c == 0 ? 1.f : 0.f) : rg[sl][r][c]);
                }
                const CfdSplit8x3 sp = cfd_split8x3(v);
                r6_u4* dst = s_a + it_unit[sl] + c;  // (columns of one group stay inside one 16-column tile: 4 | 16)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) dst[pc * 64] = __builtin_bit_cast(r6_u4, sp.p[pc]);
            }
        }
        __syncthreads();
        if (m0 + 32 < m_end) issue(m0 + 32);
        cfd_sched_fence();
        bf16x8 bq[2][3];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kt = 2 * wave + kk < ktl ? 2 * wave + kk : 0;  // (a dead k-tile re-reads tile 0: its accumulators are never stored)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bq[kk][pc] = __builtin_bit_cast(bf16x8, s_b[(kt * 3 + pc) * 64 + lane]);
        }
        bf16x8 aq[2][3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) aq[0][pc] = __builtin_bit_cast(bf16x8, s_a[pc * 64 + lane]);
#pragma unroll
        for (int nt = 0; nt < W6_NT; ++nt) {
            if (nt < ntl) {
                {
                    const int nn = nt + 1 < ntl ? nt + 1 : nt;
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) aq[(nt + 1) & 1][pc] = __builtin_bit_cast(bf16x8, s_a[(nn * 3 + pc) * 64 + lane]);
                }
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) acc[nt][kk] = cfd_mfma16x16x32_bf16(aq[nt & 1][PA[p]], bq[kk][PB[p]], acc[nt][kk]);
            }
        }
    }
    // acc[nt][kk][r] = gw[n0 + 16 nt + 4 q + r][k0 + 16 (2 wave + kk) + n]
    const int q = lane >> 4, n = lane & 15;
    float* dst = part + (size_t)blockIdx.x * N * Kx;
#pragma unroll
    for (int nt = 0; nt < W6_NT; ++nt) {
        if (nt >= ntl) continue;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int col = k0 + 16 * (2 * wave + kk) + n;
            if (2 * wave + kk >= ktl || col >= Kx) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = n0 + 16 * nt + 4 * q + r;
                if (row < N) dst[(size_t)row * Kx + col] = acc[nt][kk][r];
            }
        }
    }
}

static bool gb_rides6(int K) { return (K % (16 * W6_KT)) != 0; }  // (else the column of ones would open a k-group of its own)
static int roww6_chunks(int M, int N, int Kx) {
    const long groups = (long)((Kx + 16 * W6_KT - 1) / (16 * W6_KT)) * ((N + 16 * W6_NT - 1) / (16 * W6_NT));
#ifndef CFD_W6_WGS
#define CFD_W6_WGS 512  // workgroups aimed at (two per CU); measured (tools/build_variant.sh -DCFD_W6_WGS=..): 256 / 384 / 512: DeepONet leg 3.30 / 3.23 / 3.13 ms, Auto-FFN 2.90 / 2.88 / 2.70
#endif
    long chunks = (CFD_W6_WGS + groups - 1) / groups;
    const long maxc = (M + 255) / 256;          // at least 256 rows (8 steps) per chunk
    if (chunks > maxc) chunks = maxc;
    return chunks < 1 ? 1 : (int)chunks;
}
static size_t roww6_ws_bytes(int M, int N, int K) {  // with or without the column of ones (the chunk count depends on it): the larger
    const size_t a = (size_t)roww6_chunks(M, N, K) * N * K, b = (size_t)roww6_chunks(M, N, K + 1) * N * (K + 1);
    return cfd_align_up((a > b ? a : b) * sizeof(float), 256);
}
static bool roww6_covers(int M, int N, int K, const float* gz, const float* x) {
    const int knob = cfd_tune_get(CFD_TUNE_GEMM_B3);
    if (knob == 0 || cfd_tune_get(CFD_TUNE_EXACT_FP32) == 1) return false;
    const int lo = knob == 2 ? 16 : 96;  // (the weight gradient wins from width ~100: 256 k x 100 x 100 69.5 us against 135.7; DeepONet leg 3.74 -> 3.39 ms)
    if (M < (knob == 2 ? 1 : 4096) || N < lo || N > 1024 || K < lo || K > 1024) return false;
    if ((N & 3) || (K & 3) || (((uintptr_t)gz | (uintptr_t)x) & 15)) return false;
    return true;
}
// gw (N, K) and, if gb != NULL and it can ride, gb (N); returns whether gb was produced
static int roww6_run(const float* gz, const float* x, float* gw, float* gb, void* ws, int M, int N, int K, bool* gb_done, hipStream_t st,
                     const char* what) {
    const bool ride = gb && gb_rides6(K);
    const int Kx = ride ? K + 1 : K;
    const int chunks = roww6_chunks(M, N, Kx);
    int rpc = (M + chunks - 1) / chunks;
    rpc = (rpc + 31) / 32 * 32;
    const int nchunk = (M + rpc - 1) / rpc;
    CFD_REQUIRE(nchunk <= chunks, CFD_ERR_WORKSPACE, "%s: %d row chunks planned, %d launched", what, chunks, nchunk);  // (the workspace is sized by `chunks`)
    const dim3 grid((unsigned)nchunk, (unsigned)((Kx + 16 * W6_KT - 1) / (16 * W6_KT)), (unsigned)((N + 16 * W6_NT - 1) / (16 * W6_NT)));
    constexpr size_t lds = (size_t)(W6_NT + W6_KT) * 192 * sizeof(r6_u4);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_rowwgrad6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    CFD_PROF_W("k_rowwgrad6", st, 4.0 * ((double)M * K + (double)M * N + (double)N * K), 2.0 * M * (double)N * K);
    hipLaunchKernelGGL(k_rowwgrad6, grid, dim3(256), lds, st, gz, x, (float*)ws, M, N, K, Kx, rpc);
    CFD_LAUNCH_CHECK(what);
    GemmEpi epi{};
    if (ride) { epi.ones1 = K + 1; epi.out2 = gb; }
    CFD_PROF_W("k_splitk_reduce", st, 0.0, 0.0);
    size_t blocks = ((size_t)N * Kx + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, gw, N, Kx, K, nchunk, epi);
    CFD_LAUNCH_CHECK(what);
    *gb_done = ride;
    return CFD_OK;
}

extern "C" size_t cfd_gemm_workspace_bytes(int M, int N, int K) { return gemm_ws_bytes(M, N, K); }

extern "C" int cfd_gemm(const float* a, const float* b, float* c, void* ws, int M, int N, int K, int lda, int ldb, int ldc,
                        int trans_a, int trans_b, void* stream) {
    CFD_REQUIRE(a && b && c, CFD_ERR_INVALID_ARG, "cfd_gemm: NULL pointer");
    CFD_REQUIRE(M >= 0 && N >= 0 && K >= 1 && lda >= 1 && ldb >= 1 && ldc >= N, CFD_ERR_INVALID_ARG, "cfd_gemm: bad sizes");
    GemmEpi epi{};
    return launch_gemm(a, b, c, M, N, K, lda, ldb, ldc, trans_a, trans_b, epi, ws, (hipStream_t)stream, "cfd_gemm");
}

// ------------------------------------------------------------------------------------------------------
// nn.Linear + activation  (ffn.py:23-31): y = act(x w^T + b), x (M,K), w (N,K), y (M,N)
// ------------------------------------------------------------------------------------------------------
extern "C" size_t cfd_linear_fwd_workspace_bytes(int M, int K, int N) {
    const size_t a = gemm_ws_bytes(M, N, K), b = rowgemm6_frag_bytes(N, K);  // split-K partials or the weights' fragments (k_rowgemm6)
    return a > b ? a : b;
}

extern "C" int cfd_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* preact, void* ws, int M,
                              int K, int N, int act, void* stream) {
    CFD_REQUIRE(x && w && y, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: NULL pointer");
    CFD_REQUIRE(M >= 0 && K >= 1 && N >= 1, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: bad sizes");
    CFD_REQUIRE(act >= 0 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: act must be 0 none, 1 relu, 2 tanh, 3 gelu, 4 swish");
    CFD_REQUIRE(act < 3 || preact, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: gelu / swish need the pre-activation buffer");
    GemmEpi epi{};
    epi.mode = 1; epi.act = act; epi.bias = bias; epi.preact = preact;
    if (ws && M > 0 && rowgemm6_covers(M, N, K, x, K, epi))
        return rowgemm6_run(x, K, w, K, false, y, N, M, N, K, epi, ws, (hipStream_t)stream, "cfd_linear_fwd");
    return launch_gemm(x, w, y, M, N, K, K, K, N, 0, 1, epi, ws, (hipStream_t)stream, "cfd_linear_fwd");
}

__global__ __launch_bounds__(256) void k_act_bwd(const float* __restrict__ gy, const float* __restrict__ y,
                                                 const float* __restrict__ z, float* __restrict__ gz, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        gz[i] = gy[i] * cfd_act_grad(y ? y[i] : 0.f, z ? z[i] : 0.f, act);
}

__global__ __launch_bounds__(256) void k_act_fwd(const float* __restrict__ x, float* __restrict__ y, size_t n, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = cfd_act(x[i], act);
}

// stand-alone activation (get_act_fn(name) applied to a tensor that is not a GEMM output, act_fn.py:8-18)
extern "C" int cfd_act_fwd(const float* x, float* y, size_t n, int act, void* stream) {
    CFD_REQUIRE(x && y, CFD_ERR_INVALID_ARG, "cfd_act_fwd: NULL pointer");
    CFD_REQUIRE(act >= 0 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_act_fwd: unknown activation %d", act);
    if (n == 0) return CFD_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_act_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n, act);
    CFD_LAUNCH_CHECK("cfd_act_fwd");
    return CFD_OK;
}

// gx = gy * act'(x)   (y = act(x) is used by relu / tanh, x by gelu / swish)
extern "C" int cfd_act_bwd(const float* gy, const float* y, const float* x, float* gx, size_t n, int act, void* stream) {
    CFD_REQUIRE(gy && gx && (y || x), CFD_ERR_INVALID_ARG, "cfd_act_bwd: NULL pointer");
    CFD_REQUIRE(act >= 0 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_act_bwd: unknown activation %d", act);
    if (n == 0) return CFD_OK;
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, gy, y, x, gx, n, act);
    CFD_LAUNCH_CHECK("cfd_act_bwd");
    return CFD_OK;
}

// out[n] = sum_m g[m][n] (bias gradient of a Linear layer), two deterministic stages.  Stage 1: workgroup c of `nchunk` owns rows
// [c * rpc, (c+1) * rpc); its 256 threads are 4 row groups x 64 column lanes, so a wave reads 64 consecutive columns of one row
// (round 3's kernel gave each LANE its own row: 64 cache lines per load instruction -- 544 us for the 256 000 x 100 trunk
// activations of the non-autoregressive DeepONet, half of its train step); a thread keeps ceil(N / 64) <= 8 running sums, the four
// row groups meet in LDS.  Stage 2: one thread per column adds the nchunk partial rows in order.
#define CFD_COLSUM_MAXCHUNK 512
static int colsum_chunks(int M) {
    long c = ((long)M + 63) / 64;  // >= 64 rows per workgroup
    return (int)(c < 1 ? 1 : (c > CFD_COLSUM_MAXCHUNK ? CFD_COLSUM_MAXCHUNK : c));
}
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ g, float* __restrict__ part, int M, int N, int rpc) {
    __shared__ float s_p[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rpc, r1 = r0 + rpc < M ? r0 + rpc : M;
    for (int c0 = 0; c0 < N; c0 += 512) {  // N <= 512 in every model of the benchmark: one pass
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int nk = (N - c0 + 63) / 64 < 8 ? (N - c0 + 63) / 64 : 8;  // live column groups of this pass (uniform)
        int m = r0 + ty;
        // eight rows per trip: all of a trip's loads are issued before the first add (one row per trip left ~2 loads in flight per
        // wave: 290 us for 102 MB); rows are added in row order, so the sum does not depend on the trip structure
        for (; m + 28 < r1; m += 32) {
            float v[8][8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < nk) { const int c = c0 + 64 * k + tx; v[j][k] = c < N ? g[(size_t)(m + 4 * j) * N + c] : 0.f; }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < nk) acc[k] += v[j][k];
        }
        for (; m < r1; m += 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int c = c0 + 64 * k + tx;
                if (k < nk && c < N) acc[k] += g[(size_t)m * N + c];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (c0 + 64 * k >= N) break;  // uniform
            __syncthreads();
            s_p[ty][tx] = acc[k];
            __syncthreads();
            const int c = c0 + 64 * k + tx;
            if (ty == 0 && c < N) part[(size_t)blockIdx.x * N + c] = (s_p[0][tx] + s_p[1][tx]) + (s_p[2][tx] + s_p[3][tx]);
        }
    }
}
// stage 2: workgroup = 64 columns x 16 chunk groups; a thread adds chunks ty, ty + 16, ... (eight loads in flight), the 16 groups meet
// in LDS in a fixed order.  (One thread per column walking 1024 partial rows one dependent load at a time took ~250 us.)
__global__ __launch_bounds__(1024) void k_colsum_final(const float* __restrict__ part, float* __restrict__ out, int nchunk, int N) {
    __shared__ float s_p[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < N) {
        int k = ty;
        for (; k + 7 * 16 < nchunk; k += 8 * 16) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(k + 16 * j) * N + c];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; k < nchunk; k += 16) s += part[(size_t)k * N + c];
    }
    s_p[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += s_p[j][tx];
        out[c] = t;
    }
}
// small matrices: one wave per column, lanes stride over rows, fixed shuffle tree (one launch, no workspace)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ g, float* __restrict__ out, int M, int N) {
    const int lane = threadIdx.x & 63, col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    float s = 0.f;
    for (int m = lane; m < M; m += 64) s += g[(size_t)m * N + col];
    s = cfd_wave_sum(s);
    if (lane == 0) out[col] = s;
}
static size_t colsum_ws_bytes(int M, int N) { return M >= 2048 ? cfd_align_up((size_t)colsum_chunks(M) * N * sizeof(float), 256) : 0; }

extern "C" size_t cfd_linear_bwd_workspace_bytes(int M, int K, int N) {
    if (M <= 0) return 0;
    const size_t gz = cfd_align_up((size_t)M * N * sizeof(float), 256);
    size_t a = gemm_ws_bytes(M, K, N);
    size_t b = gemm_ws_bytes(N, K + 1, M);  // input gradient, weight gradient (+ the bias gradient's column)
    if (rowgemm6_frag_bytes(K, N) > a) a = rowgemm6_frag_bytes(K, N);  // (the input gradient on k_rowgemm6: w's fragments)
    if (roww6_ws_bytes(M, N, K) > b) b = roww6_ws_bytes(M, N, K);      // (the weight gradient on k_rowwgrad6: partial sums per row chunk)
    return gz + (a > b ? a : b) + colsum_ws_bytes(M, N);  // (+ room for the stand-alone column sum's partial rows: rounds 3-4, kept in the size)
}

// gx (M,K) = gz w;  gw (N,K) = gz^T x;  gb (N) = column sums of gz;  gz = gy * act'.  gx / gb may be NULL.
// in_act != 0: x is the OUTPUT of a layer with that activation (in_preact: its pre-activations for gelu / swish) and gx leaves as that
// layer's dZ = (gz w) * in_act'(x) -- the caller then runs that layer's backward with act = 0.
static int linear_bwd(const float* gy, const float* x, const float* w, const float* y, const float* preact, float* gx, float* gw,
                      float* gb, void* ws, int M, int K, int N, int act, int in_act, const float* in_preact, void* stream);

extern "C" int cfd_linear_bwd(const float* gy, const float* x, const float* w, const float* y, const float* preact,
                              float* gx, float* gw, float* gb, void* ws, int M, int K, int N, int act, void* stream) {
    return linear_bwd(gy, x, w, y, preact, gx, gw, gb, ws, M, K, N, act, 0, nullptr, stream);
}

extern "C" int cfd_linear_bwd_ex(const float* gy, const float* x, const float* w, const float* y, const float* preact, float* gx,
                                 float* gw, float* gb, void* ws, int M, int K, int N, int act, int in_act, const float* in_preact,
                                 void* stream) {
    CFD_REQUIRE(in_act >= 0 && in_act <= 4, CFD_ERR_INVALID_ARG, "cfd_linear_bwd_ex: bad in_act");
    CFD_REQUIRE(in_act < 3 || in_preact, CFD_ERR_INVALID_ARG, "cfd_linear_bwd_ex: gelu / swish need the input layer's pre-activation");
    return linear_bwd(gy, x, w, y, preact, gx, gw, gb, ws, M, K, N, act, in_act, in_preact, stream);
}

static int linear_bwd(const float* gy, const float* x, const float* w, const float* y, const float* preact, float* gx, float* gw,
                      float* gb, void* ws, int M, int K, int N, int act, int in_act, const float* in_preact, void* stream) {
    CFD_REQUIRE(gy && x && w && gw, CFD_ERR_INVALID_ARG, "cfd_linear_bwd: NULL pointer");
    CFD_REQUIRE(M >= 1 && K >= 1 && N >= 1, CFD_ERR_INVALID_ARG, "cfd_linear_bwd: bad sizes");
    CFD_REQUIRE(act >= 0 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_linear_bwd: bad act");
    CFD_REQUIRE(ws, CFD_ERR_WORKSPACE, "cfd_linear_bwd: workspace needed");
    CFD_REQUIRE(!(act == 1 || act == 2) || y, CFD_ERR_INVALID_ARG, "cfd_linear_bwd: relu / tanh need the layer output");
    CFD_REQUIRE(act < 3 || preact, CFD_ERR_INVALID_ARG, "cfd_linear_bwd: gelu / swish need the pre-activation");
    hipStream_t st = (hipStream_t)stream;
    const float* gz = gy;
    if (act != 0) {
        const size_t nel = (size_t)M * N;
        size_t blocks = (nel + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)blocks), dim3(256), 0, st, gy, y, preact, (float*)ws, nel, act);
        CFD_LAUNCH_CHECK("cfd_linear_bwd(act)");
        gz = (const float*)ws;
    }
    GemmEpi epi{};
    void* skws = (char*)ws + cfd_align_up((size_t)M * N * sizeof(float), 256);
    if (gx) {
        GemmEpi ex{};
        if (in_act != 0) { ex.mode = 3; ex.act = in_act; ex.resid = x; ex.ldr = K; ex.gz_z = in_preact; }
        if (rowgemm6_covers(M, K, N, gz, N, ex)) CFD_TRY(rowgemm6_run(gz, N, w, K, true, gx, K, M, K, N, ex, skws, st, "cfd_linear_bwd(gx)"));
        else CFD_TRY(launch_gemm(gz, w, gx, M, K, N, N, K, K, 0, 0, ex, skws, st, "cfd_linear_bwd(gx)"));
    }
    // gw = gz^T x, and gb = gz^T 1 as one more column of the same product (a row of ones behind x's K columns): no separate pass
    // over gz for the bias gradient (k_colsum_part + k_colsum_final: 38 us per 256 k x 100 layer)
    GemmEpi ew{};
    bool gb_rides = gb && (K % 64) != 0 && cfd_tune_get(CFD_TUNE_GEMM_TILE) <= 0;  // (K a multiple of 64: the column would add a column block)
    if (roww6_covers(M, N, K, gz, x)) {  // tall layers: the three-piece bf16 kernel (round 6)
        CFD_TRY(roww6_run(gz, x, gw, gb, skws, M, N, K, &gb_rides, st, "cfd_linear_bwd(gw)"));
    } else {
        if (gb_rides) { ew.ones1 = K + 1; ew.out2 = gb; }
        CFD_TRY(launch_gemm(gz, x, gw, N, gb_rides ? K + 1 : K, M, N, K, K, 1, 0, ew, skws, st, "cfd_linear_bwd(gw)"));
    }
    if (gb && !gb_rides) {
        if (colsum_ws_bytes(M, N)) {
            size_t a = gemm_ws_bytes(M, K, N);
            size_t b = gemm_ws_bytes(N, K + 1, M);
            if (rowgemm6_frag_bytes(K, N) > a) a = rowgemm6_frag_bytes(K, N);
            if (roww6_ws_bytes(M, N, K) > b) b = roww6_ws_bytes(M, N, K);
            float* part = (float*)((char*)skws + (a > b ? a : b));
            const int nchunk = colsum_chunks(M), rpc = (M + nchunk - 1) / nchunk;
            CFD_PROF_W("k_colsum", st, 4.0 * M * (double)N, (double)M * N);
            hipLaunchKernelGGL(k_colsum_part, dim3(nchunk), dim3(256), 0, st, gz, part, M, N, rpc);
            hipLaunchKernelGGL(k_colsum_final, dim3((N + 63) / 64), dim3(1024), 0, st, (const float*)part, gb, (M + rpc - 1) / rpc, N);
        } else {
            hipLaunchKernelGGL(k_colsum, dim3((N + 3) / 4), dim3(256), 0, st, gz, gb, M, N);
        }
        CFD_LAUNCH_CHECK("cfd_linear_bwd(gb)");
    }
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// DeepONet inner product + bias + residual  (auto_deeponet.py:127-135)
// ------------------------------------------------------------------------------------------------------
// preds[b][k] = sum_p branch[b][p] trunk[k][p] + bias[0] + (u ? u[b*HW + (qidx ? qidx[k] : k)] : 0)
// ldu: row stride of u in floats (>= HW) -- the residual field may be the leading HW columns of the branch net's input matrix
// [u.flatten(), case_params] (auto_deeponet.py:111-116), which then needs no copy of its own.
extern "C" int cfd_deeponet_inner_fwd_ex(const float* branch, const float* trunk, const float* bias, const float* u, int ldu,
                                         const int* qidx, float* preds, int B, int P, int Kq, int HW, void* stream) {
    CFD_REQUIRE(branch && trunk && bias && preds, CFD_ERR_INVALID_ARG, "cfd_deeponet_inner_fwd: NULL pointer");
    CFD_REQUIRE(B >= 0 && P >= 1 && Kq >= 0, CFD_ERR_INVALID_ARG, "cfd_deeponet_inner_fwd: bad sizes");
    CFD_REQUIRE(u || !qidx, CFD_ERR_INVALID_ARG, "cfd_deeponet_inner_fwd: query indices without a residual field");
    CFD_REQUIRE(!u || qidx || Kq <= HW, CFD_ERR_INVALID_ARG, "cfd_deeponet_inner_fwd: more queries than lattice points");
    CFD_REQUIRE(!u || ldu >= HW, CFD_ERR_INVALID_ARG, "cfd_deeponet_inner_fwd: row stride of u smaller than a row");
    GemmEpi epi{};
    epi.mode = 2; epi.bias = bias; epi.resid = u; epi.qidx = qidx; epi.ldr = ldu;  // u == NULL: bias only (deeponet.py:205)
    return launch_gemm(branch, trunk, preds, B, Kq, P, P, P, Kq, 0, 1, epi, nullptr, (hipStream_t)stream, "cfd_deeponet_inner_fwd");
}

extern "C" int cfd_deeponet_inner_fwd(const float* branch, const float* trunk, const float* bias, const float* u,
                                      const int* qidx, float* preds, int B, int P, int Kq, int HW, void* stream) {
    return cfd_deeponet_inner_fwd_ex(branch, trunk, bias, u, HW, qidx, preds, B, P, Kq, HW, stream);
}

__global__ __launch_bounds__(256) void k_sum_all(const float* __restrict__ g, size_t n, float* __restrict__ part) {
    __shared__ float s_r[4];
    // eight loads in flight per thread, eight running sums combined in a fixed order (one load per trip was a chain of 67 exposed
    // memory round trips for the 2.2 M gradients of the Auto-DeepONet bias: 17 us)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (; i + 7 * stride < n; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = g[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    for (; i < n; i += stride) acc[0] += g[i];
    float a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    a = cfd_wave_sum(a);
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s_r[0] + s_r[1]) + (s_r[2] + s_r[3]);
}

__global__ __launch_bounds__(64) void k_sum_final(const float* __restrict__ part, int nblk, float* __restrict__ out) {
    float a = 0.f;
    for (int k = threadIdx.x; k < nblk; k += 64) a += part[k];
    a = cfd_wave_sum(a);
    if (threadIdx.x == 0) out[0] = a;
}

#define CFD_SUM_BLOCKS 128
extern "C" size_t cfd_deeponet_inner_bwd_workspace_bytes(int B, int P, int Kq) {
    const size_t a = gemm_ws_bytes(B, P, Kq), b = gemm_ws_bytes(Kq, P, B);
    return cfd_align_up(CFD_SUM_BLOCKS * sizeof(float), 256) + (a > b ? a : b);
}

// gbranch (B,P) = g trunk;  gtrunk (Kq,P) = g^T branch;  gbias = sum g   (the residual's gradient is g itself)
extern "C" int cfd_deeponet_inner_bwd(const float* gpreds, const float* branch, const float* trunk, float* gbranch,
                                      float* gtrunk, float* gbias, void* ws, int B, int P, int Kq, void* stream) {
    CFD_REQUIRE(gpreds && branch && trunk && ws, CFD_ERR_INVALID_ARG, "cfd_deeponet_inner_bwd: NULL pointer");
    CFD_REQUIRE(B >= 1 && P >= 1 && Kq >= 1, CFD_ERR_INVALID_ARG, "cfd_deeponet_inner_bwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    GemmEpi epi{};
    void* skws = (char*)ws + cfd_align_up(CFD_SUM_BLOCKS * sizeof(float), 256);
    if (gbranch) CFD_TRY(launch_gemm(gpreds, trunk, gbranch, B, P, Kq, Kq, P, P, 0, 0, epi, skws, st, "cfd_deeponet_inner_bwd(gbranch)"));
    if (gtrunk) CFD_TRY(launch_gemm(gpreds, branch, gtrunk, Kq, P, B, Kq, P, P, 1, 0, epi, skws, st, "cfd_deeponet_inner_bwd(gtrunk)"));
    if (gbias) {
        hipLaunchKernelGGL(k_sum_all, dim3(CFD_SUM_BLOCKS), dim3(256), 0, st, gpreds, (size_t)B * Kq, (float*)ws);
        CFD_LAUNCH_CHECK("cfd_deeponet_inner_bwd(sum)");
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64), 0, st, (const float*)ws, CFD_SUM_BLOCKS, gbias);
        CFD_LAUNCH_CHECK("cfd_deeponet_inner_bwd(final)");
    }
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// NormAct  (src/models/act_fn.py:21-47): per sample (everything but dim 0), x^ = (x - mean) / std (unbiased std, no eps),
// y = act(x^) * std + mean.  One workgroup per sample; statistics in two passes (mean, then centred sum of squares).
// ------------------------------------------------------------------------------------------------------
// (round 4) 1024 threads per sample, 16-byte loads with four in flight per thread, statistics in ONE pass: sums of (x - K) and
// (x - K)^2 around K = the mean of the sample's first <= 1024 values (within a few percent of a standard deviation of the true
// mean, so sum (x-K)^2 - (sum (x-K))^2 / L loses no digits).  Round 3's kernels walked a 400-KB sample three times with 256 threads
// and one 4-byte load per loop trip: 165 / 197 us per launch at the non-autoregressive DeepONet's 256 x (1000 x 100) trunk activations
// (1.2 TB/s), a quarter of its train step.
#define NA_T 1024
__device__ __forceinline__ float block_sum_na(float v, float* s_r) {
    v = cfd_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NA_T / 64; ++w) t += s_r[w];
    return t;
}
// f(x_i) for i in this thread's share of [0, L): four float4 loads in flight when the sample is 16-byte aligned and L % 4 == 0
template <typename F>
__device__ __forceinline__ void na_for_each(const float* __restrict__ xs, long L, bool vec, F&& f) {
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xs);
        const long n4 = L >> 2;
        long i = threadIdx.x;
        for (; i + 3 * NA_T < n4; i += 4 * NA_T) {
            const float4 a = x4[i], b = x4[i + NA_T], c = x4[i + 2 * NA_T], d = x4[i + 3 * NA_T];
            f(4 * i, a); f(4 * (i + NA_T), b); f(4 * (i + 2 * NA_T), c); f(4 * (i + 3 * NA_T), d);
        }
        for (; i < n4; i += NA_T) f(4 * i, x4[i]);
    } else {
        for (long i = threadIdx.x; i < L; i += NA_T) f(i, make_float4(xs[i], 0.f, 0.f, 0.f));
    }
}

__global__ __launch_bounds__(NA_T) void k_normact_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                      float* __restrict__ stats, long L, int act) {
    __shared__ float s_r[NA_T / 64];
    const float* xs = x + (size_t)blockIdx.x * L;
    float* ys = y + (size_t)blockIdx.x * L;
    const bool vec = (L & 3) == 0 && ((uintptr_t)xs & 15) == 0 && ((uintptr_t)ys & 15) == 0;
    const long nk = L < NA_T ? L : NA_T;
    const float K = block_sum_na((long)threadIdx.x < nk ? xs[threadIdx.x] : 0.f, s_r) / (float)nk;
    float a = 0.f, q = 0.f;
    if (vec) na_for_each(xs, L, true, [&](long, const float4 v) {
        const float d0 = v.x - K, d1 = v.y - K, d2 = v.z - K, d3 = v.w - K;
        a += (d0 + d1) + (d2 + d3);
        q = fmaf(d0, d0, fmaf(d1, d1, fmaf(d2, d2, fmaf(d3, d3, q))));
    });
    else na_for_each(xs, L, false, [&](long, const float4 v) { const float d = v.x - K; a += d; q = fmaf(d, d, q); });
    a = block_sum_na(a, s_r);
    q = block_sum_na(q, s_r);
    const float mean = K + a / (float)L;
    float m2 = q - a * (a / (float)L);
    m2 = m2 > 0.f ? m2 : 0.f;
    const float sd = sqrtf(m2 / (float)(L - 1));
    const float inv = 1.0f / sd;
    if (vec) na_for_each(xs, L, true, [&](long i, const float4 v) {
        float4 o;
        o.x = fmaf(cfd_act((v.x - mean) * inv, act), sd, mean);
        o.y = fmaf(cfd_act((v.y - mean) * inv, act), sd, mean);
        o.z = fmaf(cfd_act((v.z - mean) * inv, act), sd, mean);
        o.w = fmaf(cfd_act((v.w - mean) * inv, act), sd, mean);
        *reinterpret_cast<float4*>(ys + i) = o;
    });
    else na_for_each(xs, L, false, [&](long i, const float4 v) { ys[i] = fmaf(cfd_act((v.x - mean) * inv, act), sd, mean); });
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = sd; }
}

// gx_i = gu_i / sd + Gmu / L + Gsd * u_i / (L - 1),  u = x^, gu = g * sd * act'(u),
// Gmu = sum g - sum gu / sd,  Gsd = sum g * act(u) - sum gu * u / sd
__global__ __launch_bounds__(NA_T) void k_normact_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                      const float* __restrict__ stats, float* __restrict__ gx, long L, int act) {
    __shared__ float s_r[NA_T / 64];
    const float* xs = x + (size_t)blockIdx.x * L;
    const float* gs = g + (size_t)blockIdx.x * L;
    float* os = gx + (size_t)blockIdx.x * L;
    const bool vec = (L & 3) == 0 && ((uintptr_t)xs & 15) == 0 && ((uintptr_t)gs & 15) == 0 && ((uintptr_t)os & 15) == 0;
    const float mean = stats[2 * blockIdx.x], sd = stats[2 * blockIdx.x + 1], inv = 1.0f / sd;
    float s_g = 0.f, s_gu = 0.f, s_ga = 0.f, s_guu = 0.f;
    auto acc1 = [&](float xv, float gi) {
        const float u = (xv - mean) * inv;
        const float av = cfd_act(u, act);
        const float gu = gi * sd * cfd_act_grad(av, u, act);
        s_g += gi; s_gu += gu; s_ga = fmaf(gi, av, s_ga); s_guu = fmaf(gu, u, s_guu);
    };
    if (vec) na_for_each(xs, L, true, [&](long i, const float4 v) {
        const float4 gv = *reinterpret_cast<const float4*>(gs + i);
        acc1(v.x, gv.x); acc1(v.y, gv.y); acc1(v.z, gv.z); acc1(v.w, gv.w);
    });
    else na_for_each(xs, L, false, [&](long i, const float4 v) { acc1(v.x, gs[i]); });
    s_g = block_sum_na(s_g, s_r);
    s_gu = block_sum_na(s_gu, s_r);
    s_ga = block_sum_na(s_ga, s_r);
    s_guu = block_sum_na(s_guu, s_r);
    const float Gmu = s_g - s_gu * inv, Gsd = s_ga - s_guu * inv;
    const float c0 = Gmu / (float)L, c1 = Gsd / (float)(L - 1);
    auto out1 = [&](float xv, float gi) {
        const float u = (xv - mean) * inv;
        const float gu = gi * sd * cfd_act_grad(cfd_act(u, act), u, act);
        return fmaf(gu, inv, fmaf(c1, u, c0));
    };
    if (vec) na_for_each(xs, L, true, [&](long i, const float4 v) {
        const float4 gv = *reinterpret_cast<const float4*>(gs + i);
        *reinterpret_cast<float4*>(os + i) = make_float4(out1(v.x, gv.x), out1(v.y, gv.y), out1(v.z, gv.z), out1(v.w, gv.w));
    });
    else na_for_each(xs, L, false, [&](long i, const float4 v) { os[i] = out1(v.x, gs[i]); });
}

// y (S,L) = NormAct(x); stats (S,2) receives (mean, std) for the backward pass.  act as in cfd_linear_fwd (1..4).
extern "C" int cfd_normact_fwd(const float* x, float* y, float* stats, int S, long L, int act, void* stream) {
    CFD_REQUIRE(x && y && stats, CFD_ERR_INVALID_ARG, "cfd_normact_fwd: NULL pointer");
    CFD_REQUIRE(S >= 0 && L >= 2 && act >= 1 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_normact_fwd: bad sizes / activation");
    if (S == 0) return CFD_OK;
    CFD_PROF_W("k_normact_fwd", (hipStream_t)stream, 8.0 * S * (double)L, 10.0 * S * (double)L);
    hipLaunchKernelGGL(k_normact_fwd, dim3(S), dim3(NA_T), 0, (hipStream_t)stream, x, y, stats, L, act);
    CFD_LAUNCH_CHECK("cfd_normact_fwd");
    return CFD_OK;
}

extern "C" int cfd_normact_bwd(const float* x, const float* gy, const float* stats, float* gx, int S, long L, int act,
                               void* stream) {
    CFD_REQUIRE(x && gy && stats && gx, CFD_ERR_INVALID_ARG, "cfd_normact_bwd: NULL pointer");
    CFD_REQUIRE(S >= 0 && L >= 2 && act >= 1 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_normact_bwd: bad sizes / activation");
    if (S == 0) return CFD_OK;
    CFD_PROF_W("k_normact_bwd", (hipStream_t)stream, 12.0 * S * (double)L, 14.0 * S * (double)L);
    hipLaunchKernelGGL(k_normact_bwd, dim3(S), dim3(NA_T), 0, (hipStream_t)stream, x, gy, stats, gx, L, act);
    CFD_LAUNCH_CHECK("cfd_normact_bwd");
    return CFD_OK;
}

// ------------------------------------------------------------------------------------------------------
// non-autoregressive DeepONet pieces (src/models/deeponet.py:184-205)
//   trunk input   x[b,k,p] = ft[b,p] + fxy[k,p]                       (:188-192)
//   output        preds[b,k] = sum_p branch[b,p] * trunk[b,k,p] + bias (:204-205)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bcast_add(const float* __restrict__ ft, const float* __restrict__ fxy,
                                                   float* __restrict__ out, int B, int K, int P) {
    const long total = (long)B * K * P;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int p = (int)(e % P), k = (int)((e / P) % K), b = (int)(e / ((long)P * K));
        out[e] = ft[(size_t)b * P + p] + fxy[(size_t)k * P + p];
    }
}

// out[b,p] = sum_k w[b,k] g[b,k,p] (w == NULL: 1): one workgroup of 1024 threads per sample b -- 8 row groups x 128 column lanes,
// eight rows per trip with their loads issued before the first add, the row groups summed through LDS in a fixed order.  (Rounds
// 1-3: one THREAD per output walking all K rows with one dependent load per trip -- 100 workgroups, 230-290 us for the 102 MB of the
// non-autoregressive DeepONet's 256 x 1000 x 100 trunk activations, three such launches = 16 % of its train step.)
__global__ __launch_bounds__(1024) void k_rows_wsum(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ out,
                                                    int K, int P) {
    __shared__ float s_p[8][128];
    const int b = blockIdx.x, tx = threadIdx.x & 127, ty = threadIdx.x >> 7;
    const float* gb = g + (size_t)b * K * P;
    const float* wb = w ? w + (size_t)b * K : nullptr;
    for (int p0 = 0; p0 < P; p0 += 128) {
        const int p = p0 + tx, pc = p < P ? p : P - 1;
        float s = 0.f;
        int k = ty;
        for (; k + 56 < K; k += 64) {  // rows k, k + 8, ..., k + 56 of this row group
            float v[8], c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = gb[(size_t)(k + 8 * u) * P + pc]; c[u] = wb ? wb[k + 8 * u] : 1.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(c[u], v[u], s);
        }
        for (; k < K; k += 8) s = fmaf(wb ? wb[k] : 1.f, gb[(size_t)k * P + pc], s);
        s_p[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && p < P) {
            float t = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) t += s_p[j][tx];
            out[(size_t)b * P + p] = t;
        }
        __syncthreads();
    }
}

// gfxy[k,p] = sum_b g[b,k,p]: thread per output (coalesced along p), eight samples per trip
__global__ __launch_bounds__(256) void k_bcast_add_bwd_b(const float* __restrict__ g, float* __restrict__ out, int B, int K, int P) {
    const long total = (long)K * P;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    float s = 0.f;
    int b = 0;
    for (; b + 8 <= B; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = g[(size_t)(b + u) * total + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < B; ++b) s += g[(size_t)b * total + e];
    out[e] = s;
}

extern "C" int cfd_bcast_add_fwd(const float* ft, const float* fxy, float* out, int B, int K, int P, void* stream) {
    CFD_REQUIRE(ft && fxy && out && B >= 0 && K >= 0 && P >= 1, CFD_ERR_INVALID_ARG, "cfd_bcast_add_fwd: bad arguments");
    const long total = (long)B * K * P;
    if (total == 0) return CFD_OK;
    long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(k_bcast_add, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, ft, fxy,
                       out, B, K, P);
    CFD_LAUNCH_CHECK("cfd_bcast_add_fwd");
    return CFD_OK;
}

extern "C" int cfd_bcast_add_bwd(const float* g, float* gft, float* gfxy, int B, int K, int P, void* stream) {
    CFD_REQUIRE(g && B >= 1 && K >= 1 && P >= 1, CFD_ERR_INVALID_ARG, "cfd_bcast_add_bwd: bad arguments");
    if (gft) hipLaunchKernelGGL(k_rows_wsum, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, g, (const float*)nullptr, gft, K, P);
    if (gfxy) hipLaunchKernelGGL(k_bcast_add_bwd_b, dim3((unsigned)(((long)K * P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, gfxy, B, K, P);
    CFD_LAUNCH_CHECK("cfd_bcast_add_bwd");
    return CFD_OK;
}

// one wave per (b, k) row
__global__ __launch_bounds__(256) void k_rowdot(const float* __restrict__ branch, const float* __restrict__ trunk,
                                                const float* __restrict__ bias, float* __restrict__ preds, long rows, int K, int P) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* br = branch + (size_t)(row / K) * P;
    const float* tr = trunk + (size_t)row * P;
    float s = 0.f;
    for (int p = lane; p < P; p += 64) s = fmaf(br[p], tr[p], s);
    s = cfd_wave_sum(s);
    if (lane == 0) preds[row] = s + bias[0];
}

// gtrunk[b,k,p] = g[b,k] * branch[b,p]
__global__ __launch_bounds__(256) void k_rowdot_bwd_trunk(const float* __restrict__ g, const float* __restrict__ branch,
                                                          float* __restrict__ gtrunk, int B, int K, int P) {
    const long total = (long)B * K * P;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int p = (int)(e % P);
        const long row = e / P;
        gtrunk[e] = g[row] * branch[(size_t)(row / K) * P + p];
    }
}

// gbranch[b,p] = sum_k g[b,k] * trunk[b,k,p]: k_rows_wsum with the weights g[b,:]

// preds (B,K) = per-sample dot of branch (B,P) with trunk (B,K,P), + bias[0]
extern "C" int cfd_rowdot_fwd(const float* branch, const float* trunk, const float* bias, float* preds, int B, int K, int P,
                              void* stream) {
    CFD_REQUIRE(branch && trunk && bias && preds && B >= 0 && K >= 0 && P >= 1, CFD_ERR_INVALID_ARG, "cfd_rowdot_fwd: bad arguments");
    const long rows = (long)B * K;
    if (rows == 0) return CFD_OK;
    hipLaunchKernelGGL(k_rowdot, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, branch, trunk, bias, preds, rows, K, P);
    CFD_LAUNCH_CHECK("cfd_rowdot_fwd");
    return CFD_OK;
}

// gbranch (B,P), gtrunk (B,K,P), gbias (1) from g (B,K); ws: CFD_SUM_BLOCKS floats (cfd_rowdot_bwd_workspace_bytes)
extern "C" size_t cfd_rowdot_bwd_workspace_bytes(void) { return CFD_SUM_BLOCKS * sizeof(float); }

extern "C" int cfd_rowdot_bwd(const float* g, const float* branch, const float* trunk, float* gbranch, float* gtrunk,
                              float* gbias, void* ws, int B, int K, int P, void* stream) {
    CFD_REQUIRE(g && branch && trunk && B >= 1 && K >= 1 && P >= 1, CFD_ERR_INVALID_ARG, "cfd_rowdot_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (gbranch) hipLaunchKernelGGL(k_rows_wsum, dim3((unsigned)B), dim3(1024), 0, st, trunk, g, gbranch, K, P);
    if (gtrunk) {
        long blocks = ((long)B * K * P + 255) / 256;
        hipLaunchKernelGGL(k_rowdot_bwd_trunk, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, st, g, branch, gtrunk, B, K, P);
    }
    if (gbias) {
        CFD_REQUIRE(ws, CFD_ERR_WORKSPACE, "cfd_rowdot_bwd: workspace needed for the bias gradient");
        hipLaunchKernelGGL(k_sum_all, dim3(CFD_SUM_BLOCKS), dim3(256), 0, st, g, (size_t)B * K, (float*)ws);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(64), 0, st, (const float*)ws, CFD_SUM_BLOCKS, gbias);
    }
    CFD_LAUNCH_CHECK("cfd_rowdot_bwd");
    return CFD_OK;
}
