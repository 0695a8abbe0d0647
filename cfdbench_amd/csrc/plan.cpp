// cfd_plan: host-built (double precision) pruned-DFT operator tables, laid out in MFMA fragment order.
// Replaces the FFT plans behind torch.fft.rfft2/irfft2 (src/models/fno/fno2d.py:62,81) and the per-call
// host np.linspace grids of Fno2d.get_coords (fno2d.py:244-255).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cfd_common.h"

static thread_local char g_err[512] = "";

void cfd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cfd_last_error(void) { return g_err; }
extern "C" int cfd_version(void) { return CFD_ABI_VERSION; }

// float -> bf16 bit pattern, round to nearest even (the device-side split uses the same rounding)
static unsigned short bf16_rne(float x) {
    unsigned int u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_to_float(unsigned short h) {
    const unsigned int u = (unsigned int)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

static int upload(const std::vector<float>& h, float** d) {
    if (hipMalloc((void**)d, h.size() * sizeof(float)) != hipSuccess) return CFD_ERR_HIP;
    if (hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return CFD_ERR_HIP;
    return CFD_OK;
}

extern "C" int cfd_plan_create(int H, int W, int m1, int m2, cfd_plan** out) {
    CFD_REQUIRE(out != nullptr, CFD_ERR_INVALID_ARG, "cfd_plan_create: out is NULL");
    CFD_REQUIRE(H >= 2 && H <= 128 && W >= 2 && W <= 80, CFD_ERR_UNSUPPORTED,
                "cfd_plan_create: grid %dx%d unsupported (need 2<=H<=128, 2<=W<=80)", H, W);
    CFD_REQUIRE(m1 >= 1 && m1 <= 15 && 2 * m1 <= H, CFD_ERR_UNSUPPORTED,
                "cfd_plan_create: modes1=%d unsupported (need 1<=m1<=15 and 2*m1<=H=%d)", m1, H);
    CFD_REQUIRE(m2 >= 1 && m2 <= 16 && m2 <= W / 2 + 1, CFD_ERR_UNSUPPORTED,
                "cfd_plan_create: modes2=%d unsupported (need 1<=m2<=16 and m2<=W/2+1, W=%d)", m2, W);
    cfd_plan* p = new cfd_plan();
    p->H = H; p->W = W; p->m1 = m1; p->m2 = m2;
    p->NJ = (W + 15) / 16;
    if (p->NJ < 4) p->NJ = 4;
    p->KX = (H / 2 + 1 + 3) / 4;
    p->T = (H + 15) / 16;
    p->SA = 4 + (m1 + 3) / 4;
    p->SB = (2 * m2 + 3) / 4;
    const int NJ = p->NJ, KX = p->KX, T = p->T, SA = p->SA, SB = p->SB;
    const double PI2 = 6.283185307179586476925286766559;

    // ---- forward tables ----
    std::vector<float> fwd((size_t)(2 * KX + 8 * NJ) * 64, 0.f);
    float* t1c = fwd.data();
    float* t1s = t1c + KX * 64;
    float* t2c = t1s + KX * 64;
    float* t2s = t2c + 4 * NJ * 64;
    for (int s = 0; s < KX; ++s)
        for (int lane = 0; lane < 64; ++lane) {
            const int xf = 4 * s + (lane >> 4), kap = lane & 15;
            if (xf > H / 2 || kap > m1) continue;
            const bool paired = (xf != 0) && (2 * xf != H);
            const double th = PI2 * (double)((long)kap * xf % H) / H;
            t1c[s * 64 + lane] = (float)std::cos(th);
            t1s[s * 64 + lane] = paired ? (float)std::sin(th) : 0.f;
        }
    for (int j = 0; j < NJ; ++j)
        for (int r = 0; r < 4; ++r)
            for (int lane = 0; lane < 64; ++lane) {
                const int l = lane & 15, y = NJ * (4 * (lane >> 4) + r) + j;
                if (l >= m2 || y >= W) continue;
                const double ph = PI2 * (double)((long)l * y % W) / W;
                t2c[(j * 4 + r) * 64 + lane] = (float)std::cos(ph);
                t2s[(j * 4 + r) * 64 + lane] = (float)std::sin(ph);
            }
    // ---- inverse tables ----
    std::vector<float> inv((size_t)(T * SA + SB * NJ) * 64, 0.f);
    float* ta = inv.data();
    float* tb = ta + T * SA * 64;
    for (int t = 0; t < T; ++t)
        for (int s = 0; s < SA; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int e = 4 * s + (lane >> 4), x = 16 * t + (lane & 15);
                if (x >= H) continue;
                const bool cosblk = e < 16;
                const int kap = cosblk ? e : e - 15;
                if (kap > m1) continue;
                const double th = PI2 * (double)((long)kap * x % H) / H;
                ta[(t * SA + s) * 64 + lane] = (float)(cosblk ? std::cos(th) : std::sin(th));
            }
    std::vector<float> clhw(m2);
    for (int l = 0; l < m2; ++l) {
        double c = (l == 0 || (W % 2 == 0 && l == W / 2)) ? 1.0 : 2.0;
        clhw[l] = (float)(c / ((double)H * W));
    }
    for (int s = 0; s < SB; ++s)
        for (int j = 0; j < NJ; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int c = 4 * s + (lane >> 4), y = NJ * (lane & 15) + j;
                if (c >= 2 * m2 || y >= W) continue;
                const bool im = c >= m2;
                const int l = im ? c - m2 : c;
                const double cl = ((l == 0 || (W % 2 == 0 && l == W / 2)) ? 1.0 : 2.0) / ((double)H * W);
                const double ph = PI2 * (double)((long)l * y % W) / W;
                tb[(s * NJ + j) * 64 + lane] = (float)(im ? -cl * std::sin(ph) : cl * std::cos(ph));
            }
    // ---- coordinate grids: np.linspace(0, 1, n) in float64, cast to float32 (fno2d.py:251,253) ----
    std::vector<float> gx(H), gy(W);
    for (int i = 0; i < H; ++i) gx[i] = (i == H - 1) ? 1.0f : (float)((double)i * (1.0 / (double)(H - 1)));
    for (int i = 0; i < W; ++i) gy[i] = (i == W - 1) ? 1.0f : (float)((double)i * (1.0 / (double)(W - 1)));

    // a table value in CFD_TW = 3 bf16 pieces (hi, lo, lo2: exact), written at 16-byte vector `vec` (+64, +128 for the later pieces)
    auto put3 = [&](std::vector<unsigned short>& dst, size_t vec, int v, double x) {
        float rem = (float)x;
        for (int piece = 0; piece < CFD_TW; ++piece) {
            const unsigned short b = bf16_rne(rem);
            dst[(vec + 64 * (size_t)piece) * 8 + v] = b;
            rem -= bf16_to_float(b);
        }
    };
    // ---- split-bf16 forward tables (64x64 grids): k-slot (q, v) of the K = 32 operand ----
    std::vector<unsigned short> fwd3;
    if (H == 64 && W == 64) {
        fwd3.assign((size_t)7 * CFD_TW * 64 * 8, 0);
        auto putf = [&](int tbl, int lane, int v, double x) { put3(fwd3, (size_t)(CFD_TW * tbl) * 64 + lane, v, x); };
        for (int lane = 0; lane < 64; ++lane) {
            const int q = lane >> 4, i = lane & 15;
            for (int v = 0; v < 8; ++v) {
                // stage 1, B operand: lane (q, kap = i), folded row xf = 4v + q (0..31: one load instruction of a wave
                // covers four consecutive rows = 1 KiB); T1N: only slot (0,0) = row H/2
                if (i <= m1) {
                    const int xf = 4 * v + q;
                    const double th = PI2 * (double)((long)i * xf % H) / H;
                    putf(0, lane, v, std::cos(th));
                    putf(1, lane, v, xf != 0 ? std::sin(th) : 0.0);
                    if (q == 0 && v == 0) putf(2, lane, v, (i % 2 == 0) ? 1.0 : -1.0);
                }
                // stage 2, A operand: lane (q, l = i), column y = 4(4q + r) + 2h + jj with v = 4jj + r
                if (i < m2) {
                    for (int h = 0; h < 2; ++h) {
                        const int jj = v >> 2, r = v & 3, y = 4 * (4 * q + r) + 2 * h + jj;
                        const double ph = PI2 * (double)((long)i * y % W) / W;
                        putf(3 + h, lane, v, std::cos(ph));
                        putf(5 + h, lane, v, std::sin(ph));
                    }
                }
            }
        }
    }
    // ---- split-bf16 inverse tables (K = 32 MFMA operand order: lane vector element v = k-step v) ----
    // Built for every grid the 64-column kernels (k_idft64, k_block) can walk: W = 64 exactly, or -- round 4, the fused FnoBlock on
    // the 66 x 65 grids of the tube / dam / cylinder problems -- 64 < W <= 68 with the columns 64 .. W-1 handled as a "tail" on the
    // VALU (tail table below); up to 5 row tiles (H <= 80; rows past H meet zero table entries).  Column map y = 4 n + j (j < 4).
    std::vector<unsigned short> inv3;
    std::vector<float> tail;  // [e = y - 64][c = 0 .. 31]: stage-B factor of U'[.][c] for tail column y (c < m2: cl cos, c - m2 < m2: -cl sin)
    p->E = 0;
    if (T <= CFD_KB_TMAX && SA <= 8 && SB <= 8 && W >= 64 && W <= 68) {
        const int NJ4 = 4;
        p->E = W - 64;
        auto tbval = [&](int c, int y) -> double {
            if (c >= 2 * m2 || y >= W) return 0.0;
            const bool im = c >= m2;
            const int l = im ? c - m2 : c;
            const double cl = ((l == 0 || (W % 2 == 0 && l == W / 2)) ? 1.0 : 2.0) / ((double)H * W);
            const double ph = PI2 * (double)((long)l * y % W) / W;
            return im ? -cl * std::sin(ph) : cl * std::cos(ph);
        };
        inv3.assign((size_t)(CFD_TW * T + CFD_TW * NJ4) * 64 * 8, 0);
        for (int t = 0; t < T; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int v = 0; v < SA; ++v) put3(inv3, (size_t)(CFD_TW * t) * 64 + lane, v, ta[(t * SA + v) * 64 + lane]);
        for (int j = 0; j < NJ4; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int v = 0; v < SB; ++v)  // stage B, B operand: lane (q, n): c = 4 v + q, column y = 4 n + j  (== tb[] when W == 64)
                    put3(inv3, (size_t)(CFD_TW * T + CFD_TW * j) * 64 + lane, v, (double)(float)tbval(4 * v + (lane >> 4), NJ4 * (lane & 15) + j));
        tail.assign(4 * 32, 0.f);
        for (int e = 0; e < p->E; ++e)
            for (int c = 0; c < 32; ++c) tail[e * 32 + c] = (float)tbval(c, 64 + e);
    }

    // ---- general-width split-bf16 tables (any W <= 80, H <= 70): y = 16 j + n column map ----
    std::vector<unsigned short> fwdg, invg;
    p->NJG = (W + 15) / 16;
    p->NHG = (p->NJG + 1) / 2;
    p->n_fwd_gv = p->n_inv_gv = 0;
    if (H <= 70 && W <= 80 && SA <= 8 && SB <= 8 && T <= 8) {
        const int NJG = p->NJG, NHG = p->NHG;
        const int ntbl = 4 + 2 * NHG;
        fwdg.assign((size_t)ntbl * CFD_TW * 64 * 8, 0);
        auto putg = put3;
        for (int lane = 0; lane < 64; ++lane) {
            const int q = lane >> 4, i = lane & 15;
            for (int v = 0; v < 8; ++v) {
                for (int b = 0; b < 2; ++b) {  // stage 1, B operand: lane (q, kap = i), folded row xf = 32 b + 4 v + q
                    const int xf = 32 * b + 4 * v + q;
                    if (i > m1 || xf > H / 2) continue;
                    const bool paired = (xf != 0) && (2 * xf != H);
                    const double th = PI2 * (double)((long)i * xf % H) / H;
                    putg(fwdg, (size_t)(CFD_TW * (2 * b)) * 64 + lane, v, std::cos(th));
                    putg(fwdg, (size_t)(CFD_TW * (2 * b + 1)) * 64 + lane, v, paired ? std::sin(th) : 0.0);
                }
                for (int h = 0; h < NHG; ++h) {  // stage 2, A operand: lane (q, l = i), column y = 16 (2h + jj) + 4q + r, v = 4jj + r
                    const int tile = 2 * h + (v >> 2), y = 16 * tile + 4 * q + (v & 3);
                    if (i >= m2 || tile >= NJG || y >= W) continue;
                    const double ph = PI2 * (double)((long)i * y % W) / W;
                    putg(fwdg, (size_t)(CFD_TW * (4 + h)) * 64 + lane, v, std::cos(ph));
                    putg(fwdg, (size_t)(CFD_TW * (4 + NHG + h)) * 64 + lane, v, std::sin(ph));
                }
            }
        }
        invg.assign((size_t)(CFD_TW * T + CFD_TW * NJG) * 64 * 8, 0);
        for (int t = 0; t < T; ++t)
            for (int lane = 0; lane < 64; ++lane)
                for (int v = 0; v < SA; ++v) putg(invg, (size_t)(CFD_TW * t) * 64 + lane, v, ta[(t * SA + v) * 64 + lane]);
        for (int j = 0; j < NJG; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int v = 0; v < SB; ++v) {  // stage B, B operand: lane (q, n): c = 4 v + q, column y = 16 j + n
                    const int c = 4 * v + (lane >> 4), y = 16 * j + (lane & 15);
                    if (c >= 2 * m2 || y >= W) continue;
                    const bool im = c >= m2;
                    const int l = im ? c - m2 : c;
                    const double cl = ((l == 0 || (W % 2 == 0 && l == W / 2)) ? 1.0 : 2.0) / ((double)H * W);
                    const double ph = PI2 * (double)((long)l * y % W) / W;
                    putg(invg, (size_t)(CFD_TW * T + CFD_TW * j) * 64 + lane, v, im ? -cl * std::sin(ph) : cl * std::cos(ph));
                }
        p->n_fwd_gv = (int)(fwdg.size() / 8);
        p->n_inv_gv = (int)(invg.size() / 8);
    }

    p->n_fwd = (int)fwd.size();
    p->n_inv = (int)inv.size();
    int rc = upload(fwd, &p->d_fwd);
    if (rc == CFD_OK) rc = upload(inv, &p->d_inv);
    if (rc == CFD_OK) rc = upload(clhw, &p->d_clhw);
    if (rc == CFD_OK) rc = upload(gx, &p->d_gx);
    if (rc == CFD_OK) rc = upload(gy, &p->d_gy);
    p->d_inv_b3 = nullptr;
    p->d_fwd_b3 = nullptr;
    if (rc == CFD_OK && !fwd3.empty()) {
        if (hipMalloc(&p->d_fwd_b3, fwd3.size() * 2) != hipSuccess ||
            hipMemcpy(p->d_fwd_b3, fwd3.data(), fwd3.size() * 2, hipMemcpyHostToDevice) != hipSuccess) rc = CFD_ERR_HIP;
    }
    if (rc == CFD_OK && !inv3.empty()) {
        if (hipMalloc(&p->d_inv_b3, inv3.size() * 2) != hipSuccess ||
            hipMemcpy(p->d_inv_b3, inv3.data(), inv3.size() * 2, hipMemcpyHostToDevice) != hipSuccess) rc = CFD_ERR_HIP;
    }
    p->d_tail = nullptr;
    if (rc == CFD_OK && !tail.empty()) rc = upload(tail, &p->d_tail);
    p->d_fwd_g = nullptr;
    p->d_inv_g = nullptr;
    if (rc == CFD_OK && !fwdg.empty()) {
        if (hipMalloc(&p->d_fwd_g, fwdg.size() * 2) != hipSuccess ||
            hipMemcpy(p->d_fwd_g, fwdg.data(), fwdg.size() * 2, hipMemcpyHostToDevice) != hipSuccess) rc = CFD_ERR_HIP;
    }
    if (rc == CFD_OK && !invg.empty()) {
        if (hipMalloc(&p->d_inv_g, invg.size() * 2) != hipSuccess ||
            hipMemcpy(p->d_inv_g, invg.data(), invg.size() * 2, hipMemcpyHostToDevice) != hipSuccess) rc = CFD_ERR_HIP;
    }
    if (rc != CFD_OK) {
        cfd_set_error("cfd_plan_create: device allocation/copy of operator tables failed");
        delete p;
        return rc;
    }
    *out = p;
    return CFD_OK;
}

extern "C" void cfd_plan_destroy(cfd_plan* p) {
    if (!p) return;
    hipFree(p->d_fwd);
    hipFree(p->d_inv);
    if (p->d_inv_b3) hipFree(p->d_inv_b3);
    if (p->d_fwd_b3) hipFree(p->d_fwd_b3);
    if (p->d_fwd_g) hipFree(p->d_fwd_g);
    if (p->d_inv_g) hipFree(p->d_inv_g);
    if (p->d_tail) hipFree(p->d_tail);
    hipFree(p->d_clhw);
    hipFree(p->d_gx);
    hipFree(p->d_gy);
    delete p;
}
