// Dev check without torch (starts in a second on a fresh box): conv forward + statistics records -> one-launch BatchNorm through
// the C ABI of a given library, against an fp64 CPU evaluation, on ordinary inputs and on inputs whose output mean is ~100 std from
// the bias; then the two launches timed at a U-Net layer shape.
//   hipcc -O2 -o gpurun_out/stats_check tools/exp/stats_check.cpp -ldl
//   gpurun_out/stats_check cfdbench_amd/_C/libcfdbench_amd.so 4      (4 = floats per record; 2 for a pre-change library)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef size_t (*ws_fn)(int, int, int, int, int, int);
typedef int (*slots_fn)(int, int, int, int, int, int);
typedef int (*conv_fn)(const float*, const float*, const float*, float*, void*, float*, int, int, int, int, int, int, void*);
typedef int (*bn_fn)(const float*, const float*, const float*, float*, float*, float*, float*, float*, const float*, int, const float*,
                     int, int, int, float, float, int, void*);

static unsigned long long rs = 88172645463325252ull;
static double urand() {
    rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
    return (double)(rs >> 11) / 9007199254740992.0;
}
static double nrand() { return std::sqrt(-2.0 * std::log(urand() + 1e-300)) * std::cos(6.283185307179586 * urand()); }

static double nmse(const std::vector<float>& a, const std::vector<double>& r) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { num += (a[i] - r[i]) * (a[i] - r[i]); den += r[i] * r[i]; }
    return num / (den > 0 ? den : 1);
}

template <class T> static T* dev(const std::vector<T>& v) {
    T* p;
    if (hipMalloc(&p, v.size() * sizeof(T)) != hipSuccess) { puts("hipMalloc failed"); exit(2); }
    hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
    const int rec = atoi(argv[2]);
    ws_fn wsb = (ws_fn)dlsym(h, "cfd_conv2d_fwd_workspace_bytes");
    slots_fn nsl = (slots_fn)dlsym(h, "cfd_conv2d_fwd_stats_slots");
    conv_fn conv = (conv_fn)dlsym(h, "cfd_conv2d_fwd_stats");
    bn_fn bn = (bn_fn)dlsym(h, "cfd_batchnorm_fwd_stats");
    if (!wsb || !nsl || !conv || !bn) { puts("missing symbol"); return 2; }
    const int only_timing = argc > 3 ? atoi(argv[3]) : 0;
    for (int pass = only_timing ? 2 : 0; pass < 5; ++pass) {
        const bool timing = pass >= 2;
        const int lvl = timing ? pass - 2 : 0;  // U-Net levels: 12 channels at 64 x 64, 24 at 32 x 32, 48 at 16 x 16
        const int B = timing ? 128 : 12, Ci = 12 << lvl, Co = timing ? 12 << lvl : 20, H = 64 >> lvl, W = 64 >> lvl, HW = H * W;
        const double off = pass == 1 ? 5.0 : 0.0, spread = pass == 1 ? 0.1 : 1.0;
        std::vector<float> x((size_t)B * Ci * HW), w((size_t)Co * Ci * 9), b(Co), ga(Co), be(Co), rm(Co), rv(Co);
        for (auto& v : x) v = (float)(off + spread * nrand());
        for (auto& v : w) { v = (float)(nrand() / std::sqrt(Ci * 9.0)); if (pass == 1) v = std::fabs(v); }
        for (int c = 0; c < Co; ++c) { b[c] = (float)(3 * nrand()); ga[c] = (float)(1 + 0.3 * nrand()); be[c] = (float)(0.2 * nrand()); rm[c] = (float)(0.5 * nrand()); rv[c] = (float)(1 + urand()); }
        const int slots = nsl(B, Ci, Co, H, W, 3);
        if (slots <= 0) { puts("no statistics for this layer"); return 3; }
        float *dx = dev(x), *dw = dev(w), *db = dev(b), *dga = dev(ga), *dbe = dev(be), *drm = dev(rm), *drv = dev(rv);
        std::vector<float> zo((size_t)B * Co * HW), zc(Co), zs((size_t)Co * slots * rec);
        float *dout = dev(zo), *dy = dev(zo), *dsm = dev(zc), *dsr = dev(zc), *dst = dev(zs);
        void* ws;
        hipMalloc(&ws, wsb(B, Ci, Co, H, W, 3));
        int e1 = conv(dx, dw, db, dout, ws, dst, B, Ci, Co, H, W, 3, nullptr);
        int e2 = bn(dout, dga, dbe, drm, drv, dy, dsm, dsr, dst, slots, db, B, Co, HW, 1e-5f, 0.1f, 1, nullptr);
        if (hipDeviceSynchronize() != hipSuccess || e1 || e2) { printf("launch failed %d %d\n", e1, e2); return 4; }
        if (timing) {
            hipEvent_t t0, t1, t2;
            hipEventCreate(&t0); hipEventCreate(&t1); hipEventCreate(&t2);
            float tc = 0, tb = 0;
            for (int it = 0; it < 210; ++it) {
                hipEventRecord(t0, nullptr);
                conv(dx, dw, db, dout, ws, dst, B, Ci, Co, H, W, 3, nullptr);
                hipEventRecord(t1, nullptr);
                bn(dout, dga, dbe, drm, drv, dy, dsm, dsr, dst, slots, db, B, Co, HW, 1e-5f, 0.1f, 1, nullptr);
                hipEventRecord(t2, nullptr);
                hipEventSynchronize(t2);
                float a, c;
                hipEventElapsedTime(&a, t0, t1); hipEventElapsedTime(&c, t1, t2);
                if (it >= 10) { tc += a; tb += c; }
            }
            printf("timing B=%d Ci=%d Co=%d %dx%d slots=%d: conv+stats %.1f us, batchnorm %.1f us\n", B, Ci, Co, H, W, slots, tc * 5, tb * 5);
            continue;
        }
        std::vector<float> out(zo.size()), y(zo.size()), nrm(Co), nrv(Co);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(nrm.data(), drm, Co * 4, hipMemcpyDeviceToHost);
        hipMemcpy(nrv.data(), drv, Co * 4, hipMemcpyDeviceToHost);
        std::vector<double> ro(zo.size()), ry(zo.size()), rrm(Co), rrv(Co);
        for (int bi = 0; bi < B; ++bi)
            for (int o = 0; o < Co; ++o)
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx) {
                        double s = b[o];
                        for (int c = 0; c < Ci; ++c)
                            for (int ky = 0; ky < 3; ++ky)
                                for (int kx = 0; kx < 3; ++kx) {
                                    int sy = yy + ky - 1, sx = xx + kx - 1;
                                    sy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
                                    sx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
                                    s += (double)w[((o * Ci + c) * 3 + ky) * 3 + kx] * x[((size_t)(bi * Ci + c) * H + sy) * W + sx];
                                }
                        ro[((size_t)(bi * Co + o) * H + yy) * W + xx] = s;
                    }
        const double N = (double)B * HW;
        for (int o = 0; o < Co; ++o) {
            double m = 0, v = 0;
            for (int bi = 0; bi < B; ++bi) for (int p = 0; p < HW; ++p) m += ro[(size_t)(bi * Co + o) * HW + p];
            m /= N;
            for (int bi = 0; bi < B; ++bi) for (int p = 0; p < HW; ++p) { double d = ro[(size_t)(bi * Co + o) * HW + p] - m; v += d * d; }
            rrm[o] = 0.9 * rm[o] + 0.1 * m;
            rrv[o] = 0.9 * rv[o] + 0.1 * v / (N - 1);
            const double rsd = 1.0 / std::sqrt(v / N + 1e-5);
            for (int bi = 0; bi < B; ++bi)
                for (int p = 0; p < HW; ++p) {
                    double t = (ro[(size_t)(bi * Co + o) * HW + p] - m) * rsd * ga[o] + be[o];
                    ry[(size_t)(bi * Co + o) * HW + p] = t > 0 ? t : 0;
                }
        }
        printf("%s: nMSE out %.2e  y %.2e  run_mean %.2e  run_var %.2e\n", pass ? "mean far from bias" : "ordinary inputs   ", nmse(out, ro),
               nmse(y, ry), nmse(nrm, rrm), nmse(nrv, rrv));
    }
    return 0;
}
