// Dense layers of the DeepONet family on the CDNA4 matrix pipe (exact-fp32 v_mfma_f32_16x16x4_f32):
//   nn.Linear + activation stacks  (Ffn, src/models/ffn.py:12-35)             -> cfd_linear_fwd / cfd_linear_bwd
//   branch x trunk inner product   (src/models/auto_deeponet.py:127-135,
//                                   src/models/deeponet.py:204-205)           -> cfd_deeponet_inner_fwd / _bwd
// The reference materialises the (b, k, p) broadcast product (879 MB at B=512 on the 66x65 lattice) and sums it; here
// it is one GEMM whose epilogue adds the bias and gathers the residual u[b, query].  All of these are instances of
// one LDS-tiled kernel C = epi(op(A) op(B)): 64x64 block tile, 4 waves x (2x2) MFMA tiles, K in slabs of 16 that are
// prefetched into registers while the previous slab is on the matrix pipe.
#include "cfd_common.h"

#define GT 64   // block tile edge (M and N)
#define GK 16   // K slab
#define GLD 68  // LDS row stride in floats (64 + 4: keeps 16-B alignment for ds_write_b128, staggers banks)

struct GemmEpi {
    int mode;             // 0: store; 1: + bias[n], activation; 2: + bias[0] (+ resid[m*ldr + (qidx ? qidx[n] : n)])
    int act;              // mode 1: 0 none, 1 relu, 2 tanh, 3 gelu (exact erf), 4 swish
    const float* bias;
    float* preact;        // mode 1: pre-activation copy (needed by the gelu / swish derivative), may be NULL
    const float* resid;
    const int* qidx;
    int ldr;
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

// One 64 x 16 slab of op(X) into registers.  KCONT: the source is k-contiguous (X stored [rows][K]); else it is
// row-contiguous (X stored [K][rows]).  Either way s[k][row] is what lands in LDS.
template <bool KCONT>
struct SlabRegs {
    float v[4];
    __device__ __forceinline__ void load(const float* __restrict__ X, int ld, int rows, int K, int row0, int k0, int tid) {
        if constexpr (KCONT) {
            const int row = row0 + (tid >> 2), k = k0 + (tid & 3) * 4;
            const float* p = X + (size_t)row * ld + k;
            if (row < rows && k + 3 < K && ((((uintptr_t)p) & 15) == 0)) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (row < rows && k + j < K) ? p[j] : 0.f;
            }
        } else {
            const int k = k0 + (tid >> 4), row = row0 + (tid & 15) * 4;
            const float* p = X + (size_t)k * ld + row;
            if (k < K && row + 3 < rows && ((((uintptr_t)p) & 15) == 0)) {
                const float4 t = *reinterpret_cast<const float4*>(p);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (k < K && row + j < rows) ? p[j] : 0.f;
            }
        }
    }
    __device__ __forceinline__ void store(float* s, int tid) const {
        if constexpr (KCONT) {
            const int row = tid >> 2, k = (tid & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) s[(k + j) * GLD + row] = v[j];
        } else {
            const int k = tid >> 4, row = (tid & 15) * 4;
            *reinterpret_cast<float4*>(s + k * GLD + row) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
};

// C[M][N] = epi( sum_k opA(m,k) opB(k,n) );  AT: A stored [K][M] (else [M][K]);  BT: B stored [N][K] (else [K][N]).
template <bool AT, bool BT>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, const float* __restrict__ B,
                                              float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc,
                                              GemmEpi epi, int slabs_per_split) {
    __shared__ __attribute__((aligned(16))) float s_a[2][GK * GLD];
    __shared__ __attribute__((aligned(16))) float s_b[2][GK * GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    SlabRegs<!AT> ra;  // A is k-contiguous unless transposed
    SlabRegs<BT> rb;   // B is k-contiguous only when stored transposed
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[2][2] = {{zero, zero}, {zero, zero}};
    // split-K: blockIdx.z owns slabs [s0, s1) and writes its partial tile to C + z*M*ldc (epilogue in k_splitk_reduce)
    const int nslab_all = (K + GK - 1) / GK;
    const int s0 = blockIdx.z * slabs_per_split;
    const int nslab = (s0 + slabs_per_split < nslab_all ? s0 + slabs_per_split : nslab_all) - s0;
    C += (size_t)blockIdx.z * M * ldc;
    ra.load(A, lda, M, K, m0, s0 * GK, tid);
    rb.load(B, ldb, N, K, n0, s0 * GK, tid);
    ra.store(s_a[0], tid);
    rb.store(s_b[0], tid);
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
        const int cur = s & 1;
        if (s + 1 < nslab) {  // next slab's global reads fly during this slab's MFMAs
            ra.load(A, lda, M, K, m0, (s0 + s + 1) * GK, tid);
            rb.load(B, ldb, N, K, n0, (s0 + s + 1) * GK, tid);
        }
        const float* sa = s_a[cur];
        const float* sb = s_b[cur];
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                av[i] = sa[(4 * kk + q) * GLD + 32 * wm + 16 * i + n];
                bv[i] = sb[(4 * kk + q) * GLD + 32 * wn + 16 * i + n];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = cfd_mfma16x16x4(av[i], bv[j], acc[i][j]);
        }
        if (s + 1 < nslab) {
            ra.store(s_a[cur ^ 1], tid);  // the other buffer was last read before the previous barrier
            rb.store(s_b[cur ^ 1], tid);
        }
        __syncthreads();
    }
    // acc[i][j][r] = C[m0 + 32wm + 16i + 4q + r][n0 + 32wn + 16j + n]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 32 * wn + 16 * j + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 32 * wm + 16 * i + 4 * q + r;
                if (row < M && col < N) {
                    float v = acc[i][j][r];
                    if (epi.mode == 1) {
                        if (epi.bias) v += epi.bias[col];
                        if (epi.preact) epi.preact[(size_t)row * ldc + col] = v;
                        v = cfd_act(v, epi.act);
                    } else if (epi.mode == 2) {
                        v += epi.bias[0];
                        if (epi.resid) v += epi.resid[(size_t)row * epi.ldr + (epi.qidx ? epi.qidx[col] : col)];
                    }
                    C[(size_t)row * ldc + col] = v;
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
        for (int z = 0; z < splits; ++z) v += part[(size_t)z * total + e];
        if (epi.mode == 1) {
            if (epi.bias) v += epi.bias[col];
            if (epi.preact) epi.preact[(size_t)row * ldc + col] = v;
            v = cfd_act(v, epi.act);
        } else if (epi.mode == 2) {
            v += epi.bias[0];
            if (epi.resid) v += epi.resid[(size_t)row * epi.ldr + (epi.qidx ? epi.qidx[col] : col)];
        }
        C[(size_t)row * ldc + col] = v;
    }
}

// Skinny products (few 64x64 output tiles, long K -- e.g. the 100x100 weight gradients over 4290 lattice points) would
// leave most CUs idle: K is split over blockIdx.z into partial tiles that a second kernel sums and finishes.
static int gemm_splits(int M, int N, int K) {
    const long tiles = (long)((M + GT - 1) / GT) * ((N + GT - 1) / GT);
    const int nslab = (K + GK - 1) / GK;
    if (tiles >= 128 || nslab < 8) return 1;
    long s = (256 + tiles - 1) / tiles;
    if (s > nslab / 4) s = nslab / 4;  // at least 4 slabs (64 k) per split
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

static size_t gemm_ws_bytes(int M, int N, int K) {
    const int s = gemm_splits(M, N, K);
    return s > 1 ? cfd_align_up((size_t)s * M * N * sizeof(float), 256) : 0;
}

static int launch_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int at,
                       int bt, const GemmEpi& epi, void* ws, hipStream_t st, const char* what) {
    if (M == 0 || N == 0) return CFD_OK;
    int splits = ws ? gemm_splits(M, N, K) : 1;
    const int nslab = (K + GK - 1) / GK;
    const int per = (nslab + splits - 1) / splits;
    splits = (nslab + per - 1) / per;
    const dim3 grid((N + GT - 1) / GT, (M + GT - 1) / GT, splits);
    GemmEpi e0{};
    const GemmEpi& ek = splits > 1 ? e0 : epi;
    float* Ck = splits > 1 ? (float*)ws : C;
    const int ldk = splits > 1 ? N : ldc;
    {
        CFD_PROF_W("k_gemm", st, 4.0 * ((double)M * K + (double)K * N + (double)M * N), 2.0 * M * (double)N * K);
        if (!at && !bt) hipLaunchKernelGGL((k_gemm<false, false>), grid, dim3(256), 0, st, A, B, Ck, M, N, K, lda, ldb, ldk, ek, per);
        else if (!at && bt) hipLaunchKernelGGL((k_gemm<false, true>), grid, dim3(256), 0, st, A, B, Ck, M, N, K, lda, ldb, ldk, ek, per);
        else if (at && !bt) hipLaunchKernelGGL((k_gemm<true, false>), grid, dim3(256), 0, st, A, B, Ck, M, N, K, lda, ldb, ldk, ek, per);
        else hipLaunchKernelGGL((k_gemm<true, true>), grid, dim3(256), 0, st, A, B, Ck, M, N, K, lda, ldb, ldk, ek, per);
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
extern "C" size_t cfd_linear_fwd_workspace_bytes(int M, int K, int N) { return gemm_ws_bytes(M, N, K); }

extern "C" int cfd_linear_fwd(const float* x, const float* w, const float* bias, float* y, float* preact, void* ws, int M,
                              int K, int N, int act, void* stream) {
    CFD_REQUIRE(x && w && y, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: NULL pointer");
    CFD_REQUIRE(M >= 0 && K >= 1 && N >= 1, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: bad sizes");
    CFD_REQUIRE(act >= 0 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: act must be 0 none, 1 relu, 2 tanh, 3 gelu, 4 swish");
    CFD_REQUIRE(act < 3 || preact, CFD_ERR_INVALID_ARG, "cfd_linear_fwd: gelu / swish need the pre-activation buffer");
    GemmEpi epi{};
    epi.mode = 1; epi.act = act; epi.bias = bias; epi.preact = preact;
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

// out[n] = sum_m g[m][n]: one wave per column, lanes stride over rows, fixed shuffle tree (deterministic)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ g, float* __restrict__ out, int M, int N) {
    const int lane = threadIdx.x & 63, col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= N) return;
    float s = 0.f;
    for (int m = lane; m < M; m += 64) s += g[(size_t)m * N + col];
    s = cfd_wave_sum(s);
    if (lane == 0) out[col] = s;
}

extern "C" size_t cfd_linear_bwd_workspace_bytes(int M, int K, int N) {
    if (M <= 0) return 0;
    const size_t gz = cfd_align_up((size_t)M * N * sizeof(float), 256);
    const size_t a = gemm_ws_bytes(M, K, N), b = gemm_ws_bytes(N, K, M);  // input gradient, weight gradient
    return gz + (a > b ? a : b);
}

// gx (M,K) = gz w;  gw (N,K) = gz^T x;  gb (N) = column sums of gz;  gz = gy * act'.  gx / gb may be NULL.
extern "C" int cfd_linear_bwd(const float* gy, const float* x, const float* w, const float* y, const float* preact,
                              float* gx, float* gw, float* gb, void* ws, int M, int K, int N, int act, void* stream) {
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
    if (gx) CFD_TRY(launch_gemm(gz, w, gx, M, K, N, N, K, K, 0, 0, epi, skws, st, "cfd_linear_bwd(gx)"));
    CFD_TRY(launch_gemm(gz, x, gw, N, K, M, N, K, K, 1, 0, epi, skws, st, "cfd_linear_bwd(gw)"));
    if (gb) {
        hipLaunchKernelGGL(k_colsum, dim3((N + 3) / 4), dim3(256), 0, st, gz, gb, M, N);
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
__device__ __forceinline__ float block_sum_256(float v, float* s_r) {
    v = cfd_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_r[0] + s_r[1]) + (s_r[2] + s_r[3]);
}

__global__ __launch_bounds__(256) void k_normact_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                     float* __restrict__ stats, long L, int act) {
    __shared__ float s_r[4];
    const float* xs = x + (size_t)blockIdx.x * L;
    float* ys = y + (size_t)blockIdx.x * L;
    float a = 0.f;
    for (long i = threadIdx.x; i < L; i += blockDim.x) a += xs[i];
    const float mean = block_sum_256(a, s_r) / (float)L;
    float q = 0.f;
    for (long i = threadIdx.x; i < L; i += blockDim.x) { const float d = xs[i] - mean; q = fmaf(d, d, q); }
    const float sd = sqrtf(block_sum_256(q, s_r) / (float)(L - 1));
    const float inv = 1.0f / sd;
    for (long i = threadIdx.x; i < L; i += blockDim.x) ys[i] = fmaf(cfd_act((xs[i] - mean) * inv, act), sd, mean);
    if (threadIdx.x == 0) { stats[2 * blockIdx.x] = mean; stats[2 * blockIdx.x + 1] = sd; }
}

// gx_i = gu_i / sd + Gmu / L + Gsd * u_i / (L - 1),  u = x^, gu = g * sd * act'(u),
// Gmu = sum g - sum gu / sd,  Gsd = sum g * act(u) - sum gu * u / sd
__global__ __launch_bounds__(256) void k_normact_bwd(const float* __restrict__ x, const float* __restrict__ g,
                                                     const float* __restrict__ stats, float* __restrict__ gx, long L, int act) {
    __shared__ float s_r[4];
    const float* xs = x + (size_t)blockIdx.x * L;
    const float* gs = g + (size_t)blockIdx.x * L;
    float* os = gx + (size_t)blockIdx.x * L;
    const float mean = stats[2 * blockIdx.x], sd = stats[2 * blockIdx.x + 1], inv = 1.0f / sd;
    float s_g = 0.f, s_gu = 0.f, s_ga = 0.f, s_guu = 0.f;
    for (long i = threadIdx.x; i < L; i += blockDim.x) {
        const float u = (xs[i] - mean) * inv, gi = gs[i];
        const float av = cfd_act(u, act);
        const float gu = gi * sd * cfd_act_grad(av, u, act);
        s_g += gi; s_gu += gu; s_ga = fmaf(gi, av, s_ga); s_guu = fmaf(gu, u, s_guu);
    }
    s_g = block_sum_256(s_g, s_r);
    s_gu = block_sum_256(s_gu, s_r);
    s_ga = block_sum_256(s_ga, s_r);
    s_guu = block_sum_256(s_guu, s_r);
    const float Gmu = s_g - s_gu * inv, Gsd = s_ga - s_guu * inv;
    const float c0 = Gmu / (float)L, c1 = Gsd / (float)(L - 1);
    for (long i = threadIdx.x; i < L; i += blockDim.x) {
        const float u = (xs[i] - mean) * inv;
        const float gu = gs[i] * sd * cfd_act_grad(cfd_act(u, act), u, act);
        os[i] = fmaf(gu, inv, fmaf(c1, u, c0));
    }
}

// y (S,L) = NormAct(x); stats (S,2) receives (mean, std) for the backward pass.  act as in cfd_linear_fwd (1..4).
extern "C" int cfd_normact_fwd(const float* x, float* y, float* stats, int S, long L, int act, void* stream) {
    CFD_REQUIRE(x && y && stats, CFD_ERR_INVALID_ARG, "cfd_normact_fwd: NULL pointer");
    CFD_REQUIRE(S >= 0 && L >= 2 && act >= 1 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_normact_fwd: bad sizes / activation");
    if (S == 0) return CFD_OK;
    CFD_PROF_W("k_normact_fwd", (hipStream_t)stream, 8.0 * S * (double)L, 10.0 * S * (double)L);
    hipLaunchKernelGGL(k_normact_fwd, dim3(S), dim3(256), 0, (hipStream_t)stream, x, y, stats, L, act);
    CFD_LAUNCH_CHECK("cfd_normact_fwd");
    return CFD_OK;
}

extern "C" int cfd_normact_bwd(const float* x, const float* gy, const float* stats, float* gx, int S, long L, int act,
                               void* stream) {
    CFD_REQUIRE(x && gy && stats && gx, CFD_ERR_INVALID_ARG, "cfd_normact_bwd: NULL pointer");
    CFD_REQUIRE(S >= 0 && L >= 2 && act >= 1 && act <= 4, CFD_ERR_INVALID_ARG, "cfd_normact_bwd: bad sizes / activation");
    if (S == 0) return CFD_OK;
    CFD_PROF_W("k_normact_bwd", (hipStream_t)stream, 12.0 * S * (double)L, 14.0 * S * (double)L);
    hipLaunchKernelGGL(k_normact_bwd, dim3(S), dim3(256), 0, (hipStream_t)stream, x, gy, stats, gx, L, act);
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

// gft[b,p] = sum_k g[b,k,p] (which = 0) or gfxy[k,p] = sum_b g[b,k,p] (which = 1): thread per output, coalesced along p
__global__ __launch_bounds__(256) void k_bcast_add_bwd(const float* __restrict__ g, float* __restrict__ out, int B, int K, int P,
                                                       int which) {
    const long total = which == 0 ? (long)B * P : (long)K * P;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int p = (int)(e % P), o = (int)(e / P);
    float s = 0.f;
    if (which == 0) for (int k = 0; k < K; ++k) s += g[((size_t)o * K + k) * P + p];
    else for (int b = 0; b < B; ++b) s += g[((size_t)b * K + o) * P + p];
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
    if (gft) hipLaunchKernelGGL(k_bcast_add_bwd, dim3((unsigned)(((long)B * P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, gft, B, K, P, 0);
    if (gfxy) hipLaunchKernelGGL(k_bcast_add_bwd, dim3((unsigned)(((long)K * P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, gfxy, B, K, P, 1);
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

// gbranch[b,p] = sum_k g[b,k] * trunk[b,k,p]: thread per (b,p), coalesced along p
__global__ __launch_bounds__(256) void k_rowdot_bwd_branch(const float* __restrict__ g, const float* __restrict__ trunk,
                                                           float* __restrict__ gbranch, int B, int K, int P) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)B * P) return;
    const int p = (int)(e % P), b = (int)(e / P);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(g[(size_t)b * K + k], trunk[((size_t)b * K + k) * P + p], s);
    gbranch[e] = s;
}

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
    if (gbranch) hipLaunchKernelGGL(k_rowdot_bwd_branch, dim3((unsigned)(((long)B * P + 255) / 256)), dim3(256), 0, st, g, trunk, gbranch, B, K, P);
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
