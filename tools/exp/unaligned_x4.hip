// Do 16-byte global loads / stores work -- and at what rate -- from addresses that are only 4- or 8-byte aligned (gfx950, ROCm 7.2)?
// The 66 x 65 grids of the tube / dam / cylinder problems make every (b, channel) plane start on an 8-byte boundary only.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/unaligned_x4.hip -o /tmp/ua && /tmp/ua
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
__global__ void k_copy(const float* __restrict__ a, float* __restrict__ b, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f4u v = *reinterpret_cast<const f4u*>(a + 4 * i);
        *reinterpret_cast<f4u*>(b + 4 * i) = f4u{v.x + 1.f, v.y + 1.f, v.z + 1.f, v.w + 1.f};
    }
}
int main() {
    const size_t n = 64u << 20;  // floats
    float *a, *b;
    hipMalloc(&a, (n + 16) * 4); hipMalloc(&b, (n + 16) * 4);
    std::vector<float> h(n + 16);
    for (size_t i = 0; i < n + 16; ++i) h[i] = (float)(i % 1000);
    hipMemcpy(a, h.data(), (n + 16) * 4, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; ++off) {
        hipMemset(b, 0, (n + 16) * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k_copy<<<2048, 256>>>(a + off, b + off, n / 4);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) k_copy<<<2048, 256>>>(a + off, b + off, n / 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> o(n + 16);
        hipMemcpy(o.data(), b, (n + 16) * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += o[off + i] != h[off + i] + 1.f;
        printf("offset %d floats (%2d-byte aligned): %s, %.2f TB/s (copy, read + write)\n", off, off == 0 ? 16 : (off == 2 ? 8 : 4), bad ? "WRONG" : "ok", 2.0 * n * 4 * 10 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
