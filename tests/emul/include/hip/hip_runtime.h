// CPU SIMT emulator of the small HIP subset the cfdbench_amd kernels use.
// TEST INFRASTRUCTURE ONLY: lets `tests/` run the *unmodified* kernel sources on the host so index math is validated against
// the oracle before a kernel ever reaches the GPU.  It is never built into, linked with, or imported by the product library.
// One FIBER per work-item (own stack, cooperative switch in tests/emul/fiber_switch.cpp), all fibers of a workgroup on the
// calling OS thread, workgroups one after the other.  __syncthreads and the wave-collective ops (MFMA, shuffles) are barriers:
// a fiber that arrives yields to the scheduler, which resumes the next fiber that is not waiting on an unreleased barrier.
// (The first version ran one OS thread per work-item with pthread barriers: every MFMA of a 256-thread workgroup was a futex
// storm on the 8 host cores, and the CPU suite took 14 minutes.)
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
// side.cpp: the emulator runs every launch synchronously on the calling thread, so a "side stream" is just another handle
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static int tag; *s = &tag; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
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

extern "C" void cfd_emul_switch(void** save_sp, void* load_sp);  // tests/emul/fiber_switch.cpp

namespace cfd_emul {
struct Idx { unsigned x, y, z; };
inline Idx g_blockDim{1, 1, 1}, g_gridDim{1, 1, 1};
inline int g_nthreads = 1;

struct Barrier {
    int count = 0, n = 0;
    unsigned gen = 0;
};
struct WaveCtx {
    Barrier bar;
    float fa[64], fb[64];
    float fa8[64][8], fb8[64][8];  // per-lane 8-element operands of the K=32 MFMA forms
    unsigned ua[64];
    int lanes;
};
struct Fiber {
    void* sp = nullptr;
    Idx tid{0, 0, 0}, bid{0, 0, 0};
    int lin = 0;
    bool done = false;
    Barrier* wait_bar = nullptr;  // blocked until wait_bar->gen != wait_gen
    unsigned wait_gen = 0;
};
inline std::vector<Fiber> g_fibers;
inline Fiber* g_cur = nullptr;
inline void* g_sched_sp = nullptr;
inline void (*g_body)() = nullptr;  // what a fiber runs (set per launch)
inline std::vector<WaveCtx*> g_waves;
inline Barrier g_block_bar;
inline std::vector<char> g_dyn_shared;  // dynamic LDS of the running launch (extern __shared__)
inline char* g_stacks = nullptr;
inline size_t g_stack_cap = 0;
constexpr size_t kStack = 512 * 1024;  // per fiber; the kernels keep tens of KB of accumulators and staging arrays in locals

inline WaveCtx& wave() { return *g_waves[g_cur->lin >> 6]; }
inline int lane() { return g_cur->lin & 63; }
inline void barrier_wait(Barrier& b) {
    if (++b.count == b.n) {  // last arrival releases the others (they become runnable again) and runs on
        b.count = 0;
        ++b.gen;
        return;
    }
    g_cur->wait_bar = &b;
    g_cur->wait_gen = b.gen;
    cfd_emul_switch(&g_cur->sp, g_sched_sp);
}
inline void wave_sync() { barrier_wait(wave().bar); }
inline void block_sync() { barrier_wait(g_block_bar); }

inline void fiber_entry() {
    g_body();
    g_cur->done = true;
    cfd_emul_switch(&g_cur->sp, g_sched_sp);
    abort();  // a finished fiber is never resumed
}

template <typename F, typename Tup, size_t... I>
void call(F f, Tup& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }

template <typename... KArgs, typename... Args>
void launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    static std::tuple<KArgs...>* s_args;
    static void (*s_kern)(KArgs...);
    static long s_nblocks;
    static dim3 s_grid;
    std::tuple<KArgs...> targs(static_cast<KArgs>(args)...);
    s_args = &targs;
    s_kern = kern;
    s_grid = grid;
    g_dyn_shared.assign(shmem + 16, 0);
    g_blockDim = {block.x, block.y, block.z};
    g_gridDim = {grid.x, grid.y, grid.z};
    const int nthreads = block.x * block.y * block.z;
    g_nthreads = nthreads;
    const int nwaves = (nthreads + 63) / 64;
    for (auto* w : g_waves) delete w;
    g_waves.clear();
    for (int w = 0; w < nwaves; ++w) {
        auto* ctx = new WaveCtx();
        ctx->lanes = std::min(64, nthreads - 64 * w);
        ctx->bar.n = ctx->lanes;
        g_waves.push_back(ctx);
    }
    g_block_bar = Barrier{};
    g_block_bar.n = nthreads;
    s_nblocks = (long)grid.x * grid.y * grid.z;
    g_body = [] {
        for (long b = 0; b < s_nblocks; ++b) {
            g_cur->bid = {(unsigned)(b % s_grid.x), (unsigned)((b / s_grid.x) % s_grid.y), (unsigned)(b / ((long)s_grid.x * s_grid.y))};
            call(s_kern, *s_args, std::index_sequence_for<KArgs...>{});
            block_sync();  // static __shared__ is reused by the next block
        }
    };
    if (g_stack_cap < (size_t)nthreads * kStack) {
        free(g_stacks);
        g_stack_cap = (size_t)nthreads * kStack;
        g_stacks = (char*)aligned_alloc(4096, g_stack_cap);  // untouched pages cost nothing
    }
    g_fibers.assign(nthreads, Fiber{});
    for (int i = 0; i < nthreads; ++i) {
        Fiber& f = g_fibers[i];
        f.lin = i;
        f.tid = {i % block.x, (i / block.x) % block.y, i / (block.x * block.y)};
        // initial frame for cfd_emul_switch: six callee-saved registers, then the entry address it "returns" to; the entry
        // function sees the stack alignment of a called function (rsp = 8 mod 16)
        uintptr_t top = ((uintptr_t)(g_stacks + (size_t)(i + 1) * kStack)) & ~(uintptr_t)15;
        void** sp = (void**)(top - 16);
        sp[1] = nullptr;
        sp[0] = (void*)&fiber_entry;
        sp -= 6;
        for (int k = 0; k < 6; ++k) sp[k] = nullptr;
        f.sp = sp;
    }
    // scheduler: resume every fiber that is not waiting on an unreleased barrier, round robin, until all are done
    int live = nthreads;
    while (live > 0) {
        bool progress = false;
        for (int i = 0; i < nthreads; ++i) {
            Fiber& f = g_fibers[i];
            if (f.done) continue;
            if (f.wait_bar) {
                if (f.wait_bar->gen == f.wait_gen) continue;
                f.wait_bar = nullptr;
            }
            g_cur = &f;
            cfd_emul_switch(&g_sched_sp, f.sp);
            progress = true;
            if (f.done) --live;
        }
        if (!progress) {
            fprintf(stderr, "cfd_emul: deadlock (a barrier that not every work-item reaches)\n");
            abort();
        }
    }
    g_cur = nullptr;
}
}  // namespace cfd_emul

#define threadIdx (cfd_emul::g_cur->tid)
#define blockIdx (cfd_emul::g_cur->bid)
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
