// CPU SIMT emulator of the small HIP subset the cfdbench_amd kernels use.
// TEST INFRASTRUCTURE ONLY: lets `tests/` run the *unmodified* kernel sources on the host (one OS thread
// per work-item, pthread barriers for __syncthreads and for wave-collective ops such as MFMA / shuffles)
// so index math is validated against the oracle before a kernel ever reaches the GPU.  It is never built
// into, linked with, or imported by the product library.
#pragma once
#include <pthread.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <tuple>
#include <vector>

#define CFD_EMUL 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emul"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return hipSuccess; }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
inline int2 make_int2(int x, int y) { return {x, y}; }

namespace cfd_emul {
struct Idx { unsigned x, y, z; };
inline thread_local Idx t_threadIdx{0, 0, 0};
inline thread_local Idx t_blockIdx{0, 0, 0};
inline thread_local int t_lin = 0;       // linear thread id in block
inline Idx g_blockDim{1, 1, 1}, g_gridDim{1, 1, 1};
inline int g_nthreads = 1;

struct WaveCtx {
    pthread_barrier_t bar;
    float fa[64], fb[64];
    float fa8[64][8], fb8[64][8];  // per-lane 8-element operands of the K=32 MFMA forms
    unsigned ua[64];
    int lanes;
};
inline std::vector<WaveCtx*> g_waves;
inline pthread_barrier_t g_block_bar;
inline std::vector<char> g_dyn_shared;  // dynamic LDS of the running launch (extern __shared__)

inline WaveCtx& wave() { return *g_waves[t_lin >> 6]; }
inline int lane() { return t_lin & 63; }
inline void wave_sync() { pthread_barrier_wait(&wave().bar); }
inline void block_sync() { pthread_barrier_wait(&g_block_bar); }

template <typename F, typename Tup, size_t... I>
void call(F f, Tup& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }

template <typename... KArgs, typename... Args>
void launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    std::tuple<KArgs...> targs(static_cast<KArgs>(args)...);
    g_dyn_shared.assign(shmem + 16, 0);
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    int nthreads = block.x * block.y * block.z;
    g_nthreads = nthreads;
    int nwaves = (nthreads + 63) / 64;
    for (auto* w : g_waves) { pthread_barrier_destroy(&w->bar); delete w; }
    g_waves.clear();
    for (int w = 0; w < nwaves; ++w) {
        auto* ctx = new WaveCtx();
        ctx->lanes = std::min(64, nthreads - 64 * w);
        pthread_barrier_init(&ctx->bar, nullptr, ctx->lanes);
        g_waves.push_back(ctx);
    }
    pthread_barrier_init(&g_block_bar, nullptr, nthreads);
    long nblocks = (long)grid.x * grid.y * grid.z;
    auto worker = [&](int lin) {
        t_lin = lin;
        t_threadIdx = {lin % block.x, (lin / block.x) % block.y, lin / (block.x * block.y)};
        for (long b = 0; b < nblocks; ++b) {
            t_blockIdx = {(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y))};
            call(kern, targs, std::index_sequence_for<KArgs...>{});
            block_sync();  // static __shared__ is reused by the next block
        }
    };
    std::vector<std::thread> th;
    th.reserve(nthreads);
    for (int i = 0; i < nthreads; ++i) th.emplace_back(worker, i);
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&g_block_bar);
}
}  // namespace cfd_emul

#define threadIdx (cfd_emul::t_threadIdx)
#define blockIdx (cfd_emul::t_blockIdx)
#define blockDim (cfd_emul::g_blockDim)
#define gridDim (cfd_emul::g_gridDim)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    cfd_emul::launch(kern, dim3(grid), dim3(block), shmem, stream, __VA_ARGS__)

inline void __syncthreads() { cfd_emul::block_sync(); }

inline float atomicAdd(float* addr, float v) {
    std::atomic_ref<float> r(*addr);
    float old = r.load();
    while (!r.compare_exchange_weak(old, old + v)) {}
    return old;
}
inline unsigned atomicAdd(unsigned* addr, unsigned v) { return __atomic_fetch_add(addr, v, __ATOMIC_SEQ_CST); }
inline int atomicAdd(int* addr, int v) { return __atomic_fetch_add(addr, v, __ATOMIC_SEQ_CST); }
