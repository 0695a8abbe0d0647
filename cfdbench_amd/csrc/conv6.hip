// k = 3, 5 and 7 replicate-padded convolutions (src/models/unet.py:20-43, auto_deeponet_cnn.py:17-33, resnet.py:35-55) as implicit GEMMs on the
// bf16 matrix pipe with THREE-PIECE operands: every fp32 value is carried as hi + lo + lo2 (8 + 8 + 8 significant bits, exact),
// a product is six v_mfma_f32_16x16x32_bf16 (cfd_mfma_bf16x6: everything down to 2^-24 of the product, fp32 accumulation), so the
// results are fp32-exact-class like the v_mfma_f32_16x16x4_f32 kernels of conv.hip they replace -- the ReLU / max-pool networks
// need that (two-piece products move pre-activations across the ReLU kink) -- at 6 x 16 cycles per
// 16x16x32 tile-step instead of 8 x 32.  What keeps the operand gathers cheap (the MFMA phase of a tile then runs at the pipe's
// rate; the staging phases around it do not overlap it -- profiles/r04f_conv6_phases.txt):
//   * an operand is split ONCE, on its way into LDS (weights: once per call by k_conv6_wprep into MFMA fragment order), and the
//     k x k gather re-reads it from there as ONE ds_read_b128 per piece: the eight K slots of a lane are eight CHANNELS of one tap
//     (forward / input gradient: LDS layout [pixel][channel]) or eight IMAGES of one pixel (weight gradient: layout
//     [pixel][channel][image]), never eight neighbouring pixels -- a tap shift then moves whole 16-byte units and every read
//     stays aligned;
//   * forward / input gradient: workgroup = 256 destination pixels x 16 MT output channels, channel chunks of CC stream through
//     LDS; a wave owns 4 pixel tiles so a weight fragment feeds 4 MFMA chains;
//   * weight gradient: K = (8 images x 4 pixels) per step, M = output channels, N = (tap, channel) columns with the channel
//     innermost, so a column tile is ONE tap and its B operand one aligned read; the four waves split the K steps of a 16-pixel
//     tile and are summed through LDS at the end (fixed order).
#include "cfd_conv.h"

typedef cfd_u32x4 u4;

#ifdef CFD_C6DIAG  // experiment builds only (tools/build_variant.sh c6diag conv6.hip -DCFD_C6DIAG, tools/exp/c6_diag.py): s_memtime
                   // stamps of workgroup 0 / wave 0 at the phase boundaries of its first 16 iterations
__device__ unsigned long long c6_ts[1024];
extern "C" int cfd_dbg_c6_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c6_ts), (size_t)n * 8); }
extern "C" int cfd_dbg_c6_clear() {
    static unsigned long long z[1024];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(c6_ts), z, sizeof(z));
}
#define C6_TS(slot)                                                                                     \
    do {                                                                                                \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && it < 16)      \
            c6_ts[it * 8 + (slot)] = __builtin_readcyclecounter();                                      \
    } while (0)
#else
#define C6_TS(slot) do { } while (0)
#endif


#ifndef CFD_CONV6_GRID
#define CFD_CONV6_GRID 512  // persistent workgroups per launch: two per CU (the CPU emulator build sets 2 so that small test shapes
                            // walk several tiles per workgroup)
#endif
// Tiles of a persistent weight-gradient workgroup: a CONTIGUOUS range (row-major neighbours share halo columns and, a tile row later, halo rows), and
// the workgroups of one XCD (launch order p, p + 8, ...: round-robin over the 8 XCDs) own neighbouring ranges, so that a halo line
// fetched for one tile is still in that XCD's L2 for the next.  (With the strided assignment tile = p + k G the weight gradient of a
// 12-channel 64x64 layer fetched 254 MB from HBM for 50 MB of operands: 2.5x halo x 2x from 40-byte runs in 64-byte lines, every
// one of them a miss -- profiles/r03y_unet_pmc_traffic.json.)
__device__ __forceinline__ void conv6_tile_range(int ntiles, int& tbeg, int& tend) {
    const int G = gridDim.x, p = blockIdx.x;
    const int pp = (G % 8 == 0) ? (p % 8) * (G / 8) + p / 8 : p;
    tbeg = (int)((long)ntiles * pp / G);
    tend = (int)((long)ntiles * (pp + 1) / G);
}

static int conv6_grid() {
    const int t = cfd_tune_get(CFD_TUNE_CONV6_GRID);
    return t > 0 ? t : CFD_CONV6_GRID;
}
#ifndef CFD_CONV6_MAX_LDS
#define CFD_CONV6_MAX_LDS (80 * 1024)  // two workgroups per CU: one stages while the other feeds the matrix pipe
#endif

// ------------------------------------------------------------------------------------------------------
// weight fragments: wfrag[(((chunk * KSTEPS + s) * MTall + mtile) * 3 + piece) * 64 + lane], lane (q, n):
//   row m = 16 mtile + n, K slot j of step s = tap s * PPS + q / CH8, channel chunk * CC + 8 (q % CH8) + j
//   (CH8 = CC / 8 channel octets per tap, PPS = 4 / CH8 taps per step).  EXT: A(m = i, (o, ky, kx)) = w[o][i][ky][kx].
// ------------------------------------------------------------------------------------------------------
template <bool EXT>
__device__ __forceinline__ void conv6_wprep_body(const float* __restrict__ w, u4* __restrict__ wfrag, int Ci, int Co, int ks, int CC,
                                                 int KSTEPS, int MTall, unsigned total, unsigned first, unsigned stride) {
    const int KK = ks * ks;
    const int Cs = EXT ? Co : Ci, Cm = EXT ? Ci : Co;
    const int CH8 = CC / 8, PPS = 4 / CH8;
    for (unsigned idx = first; idx < total; idx += stride) {
        const int lane = idx & 63;
        unsigned rest = idx >> 6;
        const int mtile = rest % MTall;
        rest /= MTall;
        const int s = rest % KSTEPS, chunk = rest / KSTEPS;
        const int q = lane >> 4, n = lane & 15;
        const int m = 16 * mtile + n, pos = s * PPS + q / CH8, cb = chunk * CC + 8 * (q % CH8);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb + j;
            v[j] = (m < Cm && c < Cs && pos < KK) ? (EXT ? w[((size_t)c * Ci + m) * KK + pos] : w[((size_t)m * Ci + c) * KK + pos]) : 0.f;
        }
        const CfdSplit8x3 sp = cfd_split8x3(v);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wfrag[((size_t)(idx >> 6) * 3 + pc) * 64 + lane] = __builtin_bit_cast(u4, sp.p[pc]);
    }
}

template <bool EXT>
__global__ __launch_bounds__(256) void k_conv6_wprep(const float* __restrict__ w, u4* __restrict__ wfrag, ConvGeom g, int CC,
                                                     int KSTEPS, int MTall, unsigned total) {
    conv6_wprep_body<EXT>(w, wfrag, g.Ci, g.Co, g.ks, CC, KSTEPS, MTall, total, blockIdx.x * blockDim.x + threadIdx.x,
                          gridDim.x * blockDim.x);
}

// The fragments of MANY weight tensors in one launch (cfd_conv2d_wprep_batch: a model prepares every layer's weights once per
// forward pass instead of one ~4 us launch in front of each convolution and each input gradient -- 35 launches of a U-Net step).
// Workgroups [blk0, blk0 + nblk) belong to item i; the table travels as the kernel argument.
#define CFD_WPREP_MAX 48  // (the kernel argument block holds 4 KB: 48 items of 56 bytes)
struct WprepItem {
    const float* w;
    u4* frag;
    int Ci, Co, ks, CC, KSTEPS, MTall, ext;
    unsigned total, blk0, nblk;
};
struct WprepBatch {
    int n;
    WprepItem it[CFD_WPREP_MAX];
};

__global__ __launch_bounds__(256) void k_conv6_wprep_batch(const WprepBatch b) {
    int i = 0;
    while (i + 1 < b.n && blockIdx.x >= b.it[i + 1].blk0) ++i;
    const WprepItem& e = b.it[i];
    const unsigned first = (blockIdx.x - e.blk0) * blockDim.x + threadIdx.x, stride = e.nblk * blockDim.x;
    if (e.ext) conv6_wprep_body<true>(e.w, e.frag, e.Ci, e.Co, e.ks, e.CC, e.KSTEPS, e.MTall, e.total, first, stride);
    else conv6_wprep_body<false>(e.w, e.frag, e.Ci, e.Co, e.ks, e.CC, e.KSTEPS, e.MTall, e.total, first, stride);
}

// ------------------------------------------------------------------------------------------------------
// forward / transposed-valid pass:  dst[b][m][p] = bias[m] + sum_k A(m, k) * src[b][c(k)][pos(p, k)]   (conv.hip has the algebra)
// Persistent workgroups: workgroup x of gridDim.x walks the pixel tiles x, x + gridDim.x, ... and, per tile, its range of channel
// chunks.  The global loads of the NEXT (tile, chunk) -- NI halo items of 8 channel values and the chunk's weight fragments -- are
// issued into registers right before the MFMA loop of the current one and land while it runs; splitting into pieces and the LDS
// stores happen at the top of the next iteration.  (The first version staged global -> LDS in plain loops inside each
// iteration: one exposed L2 round trip per loop trip, ~13 per chunk, 3x the MFMA time of a 12-channel layer.)
// LDS (all dynamic, 16-byte carve offsets): [3 piece planes of NB * LH * LW halo pixels x CC bf16 | weight fragments of this
// chunk and output group KSTEPS * MT * 3 * 64 x 16 B | tap offsets KSTEPS * 4 ints]
// ------------------------------------------------------------------------------------------------------
template <int KS, int MT, int CC, bool EXT, int NI, bool ST = false>  // ST: the forward that also emits `stats` (k = 3, NI == 3)
__global__ __launch_bounds__(256, 2) void k_conv6(const float* __restrict__ src, const u4* __restrict__ wfrag,
                                                  const float* __restrict__ bias, float* __restrict__ dst, ConvGeom g, ConvTile t,
                                                  int MTall, int ptiles, float* __restrict__ gin_direct,
                                                  float* __restrict__ stats) {
    constexpr int KK = KS * KS, PAD = KS / 2;
    constexpr int CH8 = CC / 8, PPS = 4 / CH8, KSTEPS = (KK + PPS - 1) / PPS;
    constexpr int WTOT = KSTEPS * MT * 192, NWV = (WTOT + 255) / 256;
    CFD_DYN_SHARED(u4, s_dyn);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int Cs = EXT ? g.Co : g.Ci, Cm = EXT ? g.Ci : g.Co;
    const int Hd = EXT ? g.H + 2 * PAD : g.H, Wd = EXT ? g.W + 2 * PAD : g.W;
    const int HWd = Hd * Wd, HWs = g.H * g.W;
    const int halo = t.LH * t.LW, NPX = t.NB * halo;
    const int plane = NPX * CH8;                       // (pixel, channel octet) units of the halo tile (<= 256 NI)
    char* s_in = (char*)s_dyn;                         // [pixel][channel octet][piece] x 16 B
    u4* s_w = s_dyn + 3 * plane;                       // [kstep][mt][piece][lane]
    int* s_koff = (int*)(s_w + WTOT);                  // [kstep][q]: BYTES from a pixel's own slot to its operand
    const int mb = blockIdx.y * MT;                    // first output-channel tile of this workgroup
    const int tpi = t.tiles_x * t.tiles_y;
    for (int i = threadIdx.x; i < KSTEPS * 4; i += blockDim.x) {
        const int s = i >> 2, qq = i & 3, pos = s * PPS + qq / CH8, ky = pos / KS, kx = pos - KS * ky;
        s_koff[i] = 48 * (pos < KK ? ((EXT ? (2 * PAD - ky) * t.LW + (2 * PAD - kx) : ky * t.LW + kx) * CH8 + qq % CH8) : qq % CH8);
    }
    // this wave's 4 pixel tiles: tile pixel pi = 64*wave + 16*tt + n -> (image bi, row, column) of the workgroup's tile
    int poff[4], pbi[4], ppix[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        const int pi = 64 * wave + 16 * tt + n;
        const int bi = pi / (t.TH * t.TW), rem = pi - bi * (t.TH * t.TW);
        const int r = rem / t.TW, c = rem - r * t.TW;
        const bool on = bi < t.NB;  // NB*TH*TW may be < 256 (tile shapes that are not powers of two): idle lanes
        pbi[tt] = on ? bi : -1;
        ppix[tt] = r | (c << 16);
        poff[tt] = on ? 48 * ((bi * halo + r * t.LW + c) * CH8) : 0;
    }
    // split-K over blockIdx.z: a contiguous range of channel chunks, own partial output slice (summed by k_splitk_sum)
    const int nch = (Cs + CC - 1) / CC, per = (nch + gridDim.z - 1) / gridDim.z;
    const int chbeg = blockIdx.z * per, chend = (chbeg + per) < nch ? (chbeg + per) : nch;
    const int nchw = chend - chbeg;
    const bool split = gridDim.z > 1;
    if (split) dst += (size_t)blockIdx.z * g.B * Cm * HWd;
    // (forward / input gradient: tiles p, p + G, ... -- the contiguous ranges of the weight gradient cut this kernel's HBM fetches by
    // 40 % too, but made the input-gradient launches 10 % slower: with one far-apart range per workgroup the 512 concurrent
    // workgroups touch 512 distant regions of the tensors at a time)
    const int ntile = ((int)blockIdx.x < ptiles) ? (ptiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int nit = nchw > 0 ? ntile * nchw : 0;

    // this thread's NI halo items (pixel, channel octet): tile-invariant coordinates, packed lx | ly << 8 | bi << 16.  The items of
    // channel octet 1 start at a multiple of 64, so that an item's octet is uniform per wave and its channels are wave-uniform
    // buffer offsets (conv6_plan counts the slots the same way).
    const int NPXp = (NPX + 63) & ~63;
    unsigned ipk[NI], ilds[NI], ibase[NI];
    int ic8[NI];
#pragma unroll
    for (int k = 0; k < NI; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int c8 = CH8 == 1 ? 0 : cfd_uniform(i >= NPXp ? 1 : 0);
        const bool live = i - c8 * NPXp < NPX;
        const int pxl = live ? i - c8 * NPXp : 0;
        const int r1 = (int)cfd_div((unsigned)pxl, t.dUsed), lx = pxl - r1 * t.LW;
        const int bi = (int)cfd_div((unsigned)r1, t.dLH), ly = r1 - bi * t.LH;
        ipk[k] = live ? (unsigned)(lx | (ly << 8) | (bi << 16)) : 0x80000000u;  // idle: loads nothing, never committed
        ilds[k] = 48u * (unsigned)(pxl * CH8 + c8);
        ibase[k] = 4u * (unsigned)(bi * Cs * HWs);
        ic8[k] = c8;
    }
    float dr[NI][8];
    u4 wr[NWV];
    // Operand loads and result stores go through raw buffer resources (cfd_intrinsics.h): per-lane 32-bit offset + wave-uniform
    // offset, out-of-range lanes dropped by the hardware.  (Written as `uniform pointer + lane offset` / `if (ok) store`, hipcc made
    // a 64-bit VALU address for every 4-byte access and an exec-mask region with its own branch for every store: ~70 VALU
    // instructions per halo item and ~100 branches per tile.  Removing them did NOT shorten the staging phases -- they are bound by
    // the CU's vector-memory and LDS paths, not by instruction issue: profiles/r04f_conv6_phases.txt.)
    const CfdBuf bsrc = cfd_buf(src, 4u * (unsigned)(g.B * Cs * HWs));
    // global loads of iteration `it` (tile, chunk): 8 channel values per halo item, the chunk's weight fragments.  Every load is
    // unconditional: halo pixels outside the image (EXT) or the batch carry the out-of-range offset and come back as zeros,
    // channels beyond Cs meet zero weights, weight tiles beyond MTall feed rows that are never stored.  (A `cond ? load : 0` form
    // compiles to one branch per load.)
    const auto issue = [&](int it, bool with_w) {
        const int tk = it / nchw, ch = chbeg + it - tk * nchw;
        const int tile = blockIdx.x + tk * gridDim.x;
        const int bg = tile / tpi, tr = tile - bg * tpi;
        const int ty0 = (tr / t.tiles_x) * t.TH, tx0 = (tr % t.tiles_x) * t.TW;
        const int b0 = bg * t.NB;
        const int oy = EXT ? ty0 - 2 * PAD : ty0 - PAD, ox = EXT ? tx0 - 2 * PAD : tx0 - PAD;  // source coords of the halo origin
        const int c0 = ch * CC, nbv = g.B - b0;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const unsigned pk = ipk[k];
            const int lx = pk & 255, ly = (pk >> 8) & 255, bi = (pk >> 16) & 255;
            int y = oy + ly, x = ox + lx;
            bool ok = !(pk >> 31) && bi < nbv;
            if constexpr (EXT) {  // outside the image: zeros
                ok = ok && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
            } else {              // replicate padding (zero padding: the clamped position is never loaded)
                if (g.zpad) ok = ok && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
                y = y < 0 ? 0 : (y >= g.H ? g.H - 1 : y);
                x = x < 0 ? 0 : (x >= g.W ? g.W - 1 : x);
            }
            const unsigned vo = ok ? ibase[k] + 4u * (unsigned)(y * g.W + x) : CFD_BUF_OOB;
            const int cb = c0 + 8 * ic8[k];  // (uniform) channels beyond Cs meet zero weights: any finite value of this image will do
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cb + j < Cs ? cb + j : Cs - 1;
                dr[k][j] = cfd_buf_ld(bsrc, vo, 4u * (unsigned)((b0 * Cs + c) * HWs));
            }
        }
        if (with_w) {
#pragma unroll
            for (int k = 0; k < NWV; ++k) {
                int i = threadIdx.x + 256 * k;
                i = i < WTOT ? i : WTOT - 1;
                const int s = i / (MT * 192), r = i - s * (MT * 192), mt = r / 192;
                const int mtile = mb + mt < MTall ? mb + mt : MTall - 1;
                wr[k] = wfrag[((ch * KSTEPS + s) * MTall + mtile) * 192 + (r - mt * 192)];
            }
        }
    };
    float bias_r[MT][4];  // this lane's output rows m = 16 (mb + mt) + 4 q + r
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * (mb + mt) + 4 * q + r;
            bias_r[mt][r] = cfd_opaque_f((bias && !split && m < Cm) ? bias[m] : 0.f);  // waited for HERE, not in the loop
        }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) acc[mt][tt] = zero;
    // A finished tile is stored one iteration LATER, after the next prefetch has been issued: vmcnt is one in-order counter for
    // loads and stores and the stores are conditional, so the wait for the prefetched loads at the top of an iteration is a
    // vmcnt(0) -- with the stores issued right before it, every iteration paid a full write-acknowledge latency (~4 us of the
    // ~6 us per tile in the profile of the first version).
    f32x4 outv[MT][4];
    int out_tile = -1;
    // forward with `stats`: per channel, each lane's running mean and sum of squared deviations of (out - bias) over its pixels
    // (Welford's update: no fixed shift, so nothing cancels when |mean| >> std; the same two registers per channel as plain sums,
    // plus the lane's pixel count) -- the BatchNorm that follows the conv takes its batch statistics from the merged records
    // instead of re-reading `out`
    float st1[MT][4], st2[MT][4], stn = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) st1[mt][r] = st2[mt][r] = 0.f;
    // rows m = 16 (mb + mt) + 4 q + r of this lane exist while r < rlim[mt]
    int rlim[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rlim[mt] = Cm - 16 * (mb + mt) - 4 * q;
    const bool c4 = (Cm & 3) == 0;  // (uniform) a lane's four rows of a tile exist together
    const CfdBuf bdst = cfd_buf(dst, 4u * (unsigned)(g.B * Cm * HWd));
    const CfdBuf bgin = cfd_buf(gin_direct ? gin_direct : dst, 4u * (unsigned)(g.B * Cm * (gin_direct ? HWs : HWd)));
    // the MT x 4 values of pixel tile tt to rows of `hw` floats: `o` = the lane's offset of row 16 mb + 4 q, or out of range
    const auto put = [&](const CfdBuf& bf, unsigned o, int hw, int tt, bool with_bias) {
        if (c4) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned om = rlim[mt] > 0 ? o : CFD_BUF_OOB;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cfd_buf_st(bf, om, 4u * (unsigned)((16 * mt + r) * hw), with_bias ? outv[mt][tt][r] + bias_r[mt][r] : outv[mt][tt][r]);
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    cfd_buf_st(bf, r < rlim[mt] ? o : CFD_BUF_OOB, 4u * (unsigned)((16 * mt + r) * hw),
                               with_bias ? outv[mt][tt][r] + bias_r[mt][r] : outv[mt][tt][r]);
        }
    };
    const auto store_tile = [&]() {
        const int bg = out_tile / tpi, tr = out_tile - bg * tpi;
        const int ty0 = (tr / t.tiles_x) * t.TH, tx0 = (tr % t.tiles_x) * t.TW;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int b = bg * t.NB + pbi[tt], y = ty0 + (ppix[tt] & 0xffff), x = tx0 + (ppix[tt] >> 16);
            const bool on = pbi[tt] >= 0 && b < g.B && y < Hd && x < Wd;
            const int row0 = b * Cm + 16 * mb + 4 * q;
            if constexpr (EXT) {
                // input-gradient pass with `gin_direct`: extended positions that map one-to-one onto an interior pixel of the image
                // go straight to gin; only the pad ring and the border rows / columns (whose pixels collect several extended
                // positions) take the detour through the extended buffer and k_fold_border
                // (zero padding: EVERY image pixel is one extended position -- all direct, the pad ring is dropped, nothing to fold)
                const int yi = y - PAD, xi = x - PAD, rim = g.zpad ? 0 : 1;
                const bool direct = gin_direct != nullptr && yi >= rim && yi <= g.H - 1 - rim && xi >= rim && xi <= g.W - 1 - rim;
                if (gin_direct) put(bgin, on && direct ? 4u * (unsigned)(row0 * HWs + yi * g.W + xi) : CFD_BUF_OOB, HWs, tt, false);
                if (!g.zpad && cfd_wave_any(on && !direct))  // (interior tiles have no such lane)
                    put(bdst, on && !direct ? 4u * (unsigned)(row0 * HWd + y * Wd + x) : CFD_BUF_OOB, HWd, tt, true);
            } else {
                put(bdst, on ? 4u * (unsigned)(row0 * HWd + y * Wd + x) : CFD_BUF_OOB, HWd, tt, true);
                if constexpr (ST) {
                    if (on) {
                        stn += 1.f;
                        const float rn = cfd_rcpf(stn);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = outv[mt][tt][r], d = v - st1[mt][r];
                                st1[mt][r] = fmaf(d, rn, st1[mt][r]);
                                st2[mt][r] = fmaf(d, v - st1[mt][r], st2[mt][r]);
                            }
                    }
                }
            }
        }
        out_tile = -1;
    };
    if (nit > 0) issue(0, true);
    for (int it = 0; it < nit; ++it) {
        C6_TS(0);
        __syncthreads();  // previous iteration's operands fully consumed (first pass: s_koff written)
        C6_TS(1);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (!(ipk[k] >> 31)) {
                const CfdSplit8x3 sp = cfd_split8x3(dr[k]);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) *(u4*)(s_in + ilds[k] + 16 * pc) = __builtin_bit_cast(u4, sp.p[pc]);
            }
        }
        if (it == 0 || nchw > 1) {
#pragma unroll
            for (int k = 0; k < NWV; ++k) {
                const int i = threadIdx.x + 256 * k;
                if (i < WTOT) s_w[i] = wr[k];
            }
        }
        C6_TS(2);
        __syncthreads();
        C6_TS(3);
        if (it + 1 < nit) issue(it + 1, nchw > 1);
        C6_TS(4);
        if (out_tile >= 0) store_tile();
        cfd_sched_fence();  // keep the loads above the MFMA loop: hipcc otherwise sinks them to their first use
        C6_TS(5);
#pragma unroll 1
        for (int s = 0; s < KSTEPS; ++s) {
            const int ko = s_koff[4 * s + q];
            const u4* wp = s_w + s * (MT * 192) + lane;
            bf16x8 bv[4][3], av[MT][3];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) bv[tt][pc] = __builtin_bit_cast(bf16x8, *(const u4*)(s_in + (poff[tt] + ko) + 16 * pc));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) av[mt][pc] = __builtin_bit_cast(bf16x8, wp[(mt * 3 + pc) * 64]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) cfd_mfma_bf16x6_n<4>(av[mt], bv, acc[mt]);
        }
        C6_TS(6);
        const int tk = it / nchw;
        if (it - tk * nchw == nchw - 1) {  // last chunk of the tile: hand the sums to the deferred store, start the next tile from zero
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) { outv[mt][tt] = acc[mt][tt]; acc[mt][tt] = zero; }
            out_tile = blockIdx.x + tk * gridDim.x;
        }
    }
    if (out_tile >= 0) store_tile();
    if constexpr (ST) {
        // the 16 pixel lanes of a row group merge pairwise, then the four waves through LDS in a fixed order -> one record
        // (mean, m2, pixels) per (channel, workgroup)
        float cn = stn, w_hi[4], w_prod[4];  // per merge round: n_b / (n_a + n_b) and n_a n_b / (n_a + n_b), the same for every channel
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float nb = cfd_shfl_xor(cn, 1 << k), nn = cn + nb, rn = nn > 0.f ? cfd_rcpf(nn) : 0.f;
            w_hi[k] = nb * rn;
            w_prod[k] = cn * w_hi[k];
            cn = nn;
        }
        __syncthreads();
        float* s_st = (float*)s_dyn;  // [wave][mt][q][r][4]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (every lane merges "its own" with its partner's: the two lanes of a pair agree up to rounding only, and only what
                //  reaches lane n == 0 is written -- any of these roundings is a valid merge, and the same one on every run)
                float m = st1[mt][r], m2 = st2[mt][r];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float mo = cfd_shfl_xor(m, 1 << k), m2o = cfd_shfl_xor(m2, 1 << k);
                    const float dl = mo - m;
                    m = fmaf(dl, w_hi[k], m);
                    m2 = fmaf(dl * dl, w_prod[k], m2 + m2o);
                }
                if (n == 0) *(f32x4*)(s_st + (((wave * MT + mt) * 4 + q) * 4 + r) * 4) = f32x4{m, m2, cn, 0.f};
            }
        __syncthreads();
        for (int e = threadIdx.x; e < MT * 16; e += blockDim.x) {
            const int mt = e >> 4, qr = e & 15, c = 16 * (mb + mt) + qr;
            if (c < Cm) {
                float m = 0.f, m2 = 0.f, cnt = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) {
                    const f32x4 b = *(const f32x4*)(s_st + ((wv * MT + mt) * 16 + qr) * 4);
                    if (b[2] > 0.f) {
                        const float nn = cnt + b[2], wh = b[2] * cfd_rcpf(nn), dl = b[0] - m;
                        m = fmaf(dl, wh, m);
                        m2 = fmaf(dl * dl, cnt * wh, m2 + b[1]);
                        cnt = nn;
                    }
                }
                *(f32x4*)(stats + ((size_t)c * gridDim.x + blockIdx.x) * 4) = f32x4{m, m2, cnt, 0.f};
            }
        }
    }
}

struct Conv6Plan {
    ConvTile t;
    long ptiles;
    int CC, KSTEPS, MTall, mtw, mgroups, ksplit, nch, NI, gx;
    size_t lds, wfrag_bytes, split_bytes;
    bool ok;
};

// The fragment form of a weight tensor depends on (Ci, Co, ks, ext) only -- not on the batch or the grid -- so fragments made once
// serve every call of the layer.
struct Conv6Frag {
    int CC, KSTEPS, MTall, nch;
    size_t bytes;
    bool ok;
};

static Conv6Frag conv6_frag(int Ci, int Co, int ks, bool ext) {
    Conv6Frag F{};
    if ((ks != 3 && ks != 5 && ks != 7) || Ci < 1 || Co < 1) return F;
    const int Cm = ext ? Ci : Co, Cs = ext ? Co : Ci;
    F.CC = (ks >= 5 || Cs <= 8) ? 8 : 16;  // k = 5 / 7: 7 / 13 k-steps of 4 taps x 8 channels keep two workgroups per CU
    const int PPS = 32 / F.CC;
    F.KSTEPS = (ks * ks + PPS - 1) / PPS;
    F.MTall = (Cm + 15) / 16;
    F.nch = (Cs + F.CC - 1) / F.CC;
    F.bytes = cfd_align_up((size_t)F.nch * F.KSTEPS * F.MTall * 3 * 1024, 256);
    F.ok = (long)F.nch * F.KSTEPS * F.MTall * 64 < (1L << 31);
    return F;
}

static Conv6Plan conv6_plan(const ConvGeom& g, bool ext) {
    Conv6Plan P{};
    if (g.ks != 3 && g.ks != 5 && g.ks != 7) return P;
    const int KS = g.ks, PAD = KS / 2;
    const int Hd = ext ? g.H + 2 * PAD : g.H, Wd = ext ? g.W + 2 * PAD : g.W;
    const int Cm = ext ? g.Ci : g.Co, Cs = ext ? g.Co : g.Ci;
    const Conv6Frag F = conv6_frag(g.Ci, g.Co, g.ks, ext);
    P.CC = F.CC;
    P.KSTEPS = F.KSTEPS;
    P.MTall = F.MTall;
    P.nch = F.nch;
    ConvTile& t = P.t;
    cfd_conv_tile_shape(Hd, Wd, g.B, t.TW, t.TH, t.NB);
    // a thread prefetches at most 5 halo items (8 channel values each) of the next tile: small images, many per workgroup.
    // Item slots of a tile: its halo pixels once per channel octet, the second octet starting at a multiple of 64 (k_conv6).
    const auto slots_of = [&](int nb) {
        const int npx = nb * (t.TH + KS - 1) * (t.TW + KS - 1);
        return P.CC == 8 ? npx : ((npx + 63) & ~63) + npx;
    };
    while (t.NB > 1 && slots_of(t.NB) > 5 * 256) --t.NB;
    t.tiles_x = (Wd + t.TW - 1) / t.TW;
    t.tiles_y = (Hd + t.TH - 1) / t.TH;
    t.LH = t.TH + KS - 1;
    t.LW = t.TW + KS - 1;
    t.dUsed = cfd_div_make((unsigned)t.LW);
    t.dLH = cfd_div_make((unsigned)t.LH);
    P.ptiles = (long)((g.B + t.NB - 1) / t.NB) * t.tiles_x * t.tiles_y;
    const auto lds_of = [&](int mtw) {
        return (size_t)3 * t.NB * t.LH * t.LW * P.CC * 2 + (size_t)P.KSTEPS * mtw * 3 * 1024 + (size_t)P.KSTEPS * 16;
    };
    // two output-channel tiles per workgroup (a staged halo feeds twice the MFMAs) while two workgroups still fit a CU and the
    // grid fills the chip
    const int plane = slots_of(t.NB);
    P.NI = plane <= 3 * 256 ? 3 : 5;
    int mtw = P.MTall >= 2 ? 2 : 1;
#ifndef CFD_CONV6_MT2_MIN_WGS
#define CFD_CONV6_MT2_MIN_WGS 64  // (256 -> 64: -1.4 % on the U-Net step; split-K supplies the workgroups of the deep levels)
#endif
    if (mtw == 2 && (lds_of(2) > CFD_CONV6_MAX_LDS || P.ptiles * ((P.MTall + 1) / 2) < CFD_CONV6_MT2_MIN_WGS)) mtw = 1;
    P.mtw = mtw;
    P.mgroups = (P.MTall + mtw - 1) / mtw;
    const long wgs = P.ptiles * P.mgroups;
    P.ksplit = 1;
    const long slots = conv6_grid();  // resident workgroups of a launch (two per CU)
    long gx = slots / P.mgroups;
    if (wgs < 256 && P.nch > 1) {
        // deep, narrow levels: the channel chunks split over blockIdx.z.  Round 6: (splits, workgroups per slice) chosen together for the
        // fewest (tile, chunk) trips of the slowest workgroup while the whole grid stays resident -- the fixed rule (512 / wgs splits, then
        // gx = 512 / (groups x splits)) left e.g. the 4 x 4 U-Net level with 8 tiles on 7 workgroups x 6 splits: one workgroup walked two
        // tiles while the rest idled.  Ties go to fewer splits (less partial-sum traffic for k_splitk_sum).
        long best_trips = -1;
        const int ksmax = P.nch < 16 ? P.nch : 16;
        for (int ks = 1; ks <= ksmax; ++ks) {
            const int per = (P.nch + ks - 1) / ks, kse = (P.nch + per - 1) / per;  // every slice owns at least one chunk
            if (kse != ks) continue;
            long g1 = slots / ((long)P.mgroups * ks);
            if (g1 < 1) g1 = 1;
            if (g1 > P.ptiles) g1 = P.ptiles;
            const long trips = ((P.ptiles + g1 - 1) / g1) * per;
            if (best_trips < 0 || trips < best_trips) { best_trips = trips; P.ksplit = ks; gx = g1; }
        }
    } else if (P.ksplit == 1) {
        gx = slots / P.mgroups;
    }
    P.lds = lds_of(mtw);
    // persistent workgroups: two per CU over the whole launch
    if (gx < 1) gx = 1;
    P.gx = (int)(gx < P.ptiles ? gx : P.ptiles);
    P.wfrag_bytes = F.bytes;
    P.split_bytes = P.ksplit > 1 ? cfd_align_up((size_t)P.ksplit * g.B * Cm * Hd * Wd * sizeof(float), 256) : 0;
    const bool small = (long)g.B * Cs * g.H * g.W <= (1L << 29) && (long)g.B * Cm * Hd * Wd <= (1L << 29);  // buffer resources: <= 2^31 bytes
    P.ok = small && P.lds <= 150 * 1024 && plane <= 5 * 256 && F.ok && P.ptiles < (1L << 30);
    return P;
}

bool cfd_conv6_covers(const ConvGeom& g, bool ext) { return conv6_plan(g, ext).ok; }

// zero padding runs on these kernels only: forward, the input gradient with every pixel written directly (no split-K partials, an
// image of at least 3 x 3) and the weight gradient must all be covered
bool cfd_conv6_zeropad_covers(const ConvGeom& g) {
    const Conv6Plan Pe = conv6_plan(g, true);
    return conv6_plan(g, false).ok && Pe.ok && Pe.ksplit == 1 && g.H >= 3 && g.W >= 3 && cfd_conv6_wgrad_covers(g);
}

size_t cfd_conv6_ws_bytes(const ConvGeom& g, bool ext) {
    const Conv6Plan P = conv6_plan(g, ext);
    return P.ok ? P.wfrag_bytes + P.split_bytes : 0;
}

template <int KS, int CC, bool EXT>
static void conv6_launch(const Conv6Plan& P, const float* src, const u4* wfrag, const float* bias, float* dst, const ConvGeom& g,
                         float* gin_direct, float* stats, hipStream_t st) {
    const dim3 grid((unsigned)P.gx, P.mgroups, P.ksplit);
#define C6_L(M_, N_, S_)                                                                                                          \
    do {                                                                                                                          \
        static bool attr_set = false;                                                                                             \
        if (!attr_set) {                                                                                                          \
            (void)hipFuncSetAttribute((const void*)k_conv6<KS, M_, CC, EXT, N_, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((k_conv6<KS, M_, CC, EXT, N_, S_>), grid, dim3(256), P.lds, st, src, wfrag, bias, dst, g, P.t, P.MTall, \
                           (int)P.ptiles, gin_direct, stats);                                                                     \
    } while (0)
    if constexpr (!EXT && KS == 3) {
        if (stats) {  // (conv6_run: NI == 3, one pass)
            if (P.mtw == 1) C6_L(1, 3, true); else C6_L(2, 3, true);
            return;
        }
    }
    if (P.mtw == 1) { if (P.NI == 3) C6_L(1, 3, false); else C6_L(1, 5, false); }
    else { if (P.NI == 3) C6_L(2, 3, false); else C6_L(2, 5, false); }
#undef C6_L
}

template <bool EXT>
static int conv6_run(const float* src, const float* w, const float* bias, float* dst, void* ws, const ConvGeom& g, float* gin,
                     bool* direct, float* stats, const void* wfrag_ready, hipStream_t st, const char* what) {
    const Conv6Plan P = conv6_plan(g, EXT);
    if (!P.ok) return CFD_ERR_UNSUPPORTED;
    const u4* wfrag = wfrag_ready ? (const u4*)wfrag_ready : (const u4*)ws;
    float* split_ws = P.ksplit > 1 ? (float*)((char*)ws + P.wfrag_bytes) : nullptr;
    if (!wfrag_ready) {
        const unsigned total = (unsigned)((long)P.nch * P.KSTEPS * P.MTall * 64);
        unsigned blocks = (total + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL((k_conv6_wprep<EXT>), dim3(blocks), dim3(256), 0, st, w, (u4*)ws, g, P.CC, P.KSTEPS, P.MTall, total);
        CFD_LAUNCH_CHECK(what);
    }
    float* kdst = P.ksplit > 1 ? split_ws : dst;
    // interior pixels of the input gradient straight to gin (no split-K partials, an interior exists, 32-bit offsets hold)
    float* gd = (EXT && gin && P.ksplit == 1 && g.H >= 3 && g.W >= 3) ? gin : nullptr;
    if (direct) *direct = gd != nullptr;
    if (EXT && g.zpad && !gd) return CFD_ERR_UNSUPPORTED;  // (zero padding has no fold pass: callers ask cfd_conv6_zeropad_covers first)
    if (stats && (EXT || g.ks != 3 || P.ksplit > 1 || P.NI != 3)) return CFD_ERR_UNSUPPORTED;  // (callers ask cfd_conv6_stats_slots first)
    if (g.ks == 3) {
        if (P.CC == 8) conv6_launch<3, 8, EXT>(P, src, wfrag, bias, kdst, g, gd, stats, st);
        else conv6_launch<3, 16, EXT>(P, src, wfrag, bias, kdst, g, gd, stats, st);
    } else if (g.ks == 5) {  // (round 4: the 5 x 5 convolutions of the Auto-DeepONet-CNN branch, src/models/auto_deeponet_cnn.py:17-33)
        conv6_launch<5, 8, EXT>(P, src, wfrag, bias, kdst, g, gd, stats, st);
    } else {
        conv6_launch<7, 8, EXT>(P, src, wfrag, bias, kdst, g, gd, stats, st);
    }
    CFD_LAUNCH_CHECK(what);
    if (P.ksplit > 1) {
        const int PAD = g.ks / 2, Cm = EXT ? g.Ci : g.Co;
        const long HWd = EXT ? (long)(g.H + 2 * PAD) * (g.W + 2 * PAD) : (long)g.H * g.W;
        CFD_TRY(cfd_conv_splitk_sum(split_ws, bias, dst, (long)g.B * Cm * HWd, P.ksplit, Cm, HWd, st, what));
    }
    return CFD_OK;
}

int cfd_conv6_run(const float* src, const float* w, const float* bias, float* dst, void* ws, const ConvGeom& g, bool ext, float* gin,
                  bool* direct, hipStream_t st, const char* what, float* stats, const void* wfrag) {
    if (direct) *direct = false;
    return ext ? conv6_run<true>(src, w, bias, dst, ws, g, gin, direct, nullptr, wfrag, st, what)
               : conv6_run<false>(src, w, bias, dst, ws, g, nullptr, nullptr, stats, wfrag, st, what);
}

size_t cfd_conv6_wfrag_bytes(int Ci, int Co, int ks, bool ext) {
    const Conv6Frag F = conv6_frag(Ci, Co, ks, ext);
    return F.ok ? F.bytes : 0;
}

int cfd_conv6_wprep_batch(int n, const float* const* w, void* const* wfrag, const int* Ci, const int* Co, const int* ks,
                          const int* ext, hipStream_t st, const char* what) {
    for (int base = 0; base < n; base += CFD_WPREP_MAX) {
        WprepBatch b{};
        b.n = n - base < CFD_WPREP_MAX ? n - base : CFD_WPREP_MAX;
        unsigned blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const int k = base + i;
            const Conv6Frag F = conv6_frag(Ci[k], Co[k], ks[k], ext[k] != 0);
            if (!F.ok || !w[k] || !wfrag[k]) {
                cfd_set_error("%s: item %d (Ci %d, Co %d, ks %d) has no fragment form or a NULL pointer", what, k, Ci[k], Co[k], ks[k]);
                return CFD_ERR_INVALID_ARG;
            }
            WprepItem& e = b.it[i];
            e.w = w[k];
            e.frag = (u4*)wfrag[k];
            e.Ci = Ci[k], e.Co = Co[k], e.ks = ks[k], e.CC = F.CC, e.KSTEPS = F.KSTEPS, e.MTall = F.MTall, e.ext = ext[k] != 0;
            e.total = (unsigned)((long)F.nch * F.KSTEPS * F.MTall * 64);
            e.nblk = (e.total + 255) / 256;
            if (e.nblk > 256) e.nblk = 256;
            e.blk0 = blocks;
            blocks += e.nblk;
        }
        hipLaunchKernelGGL(k_conv6_wprep_batch, dim3(blocks), dim3(256), 0, st, b);
        CFD_LAUNCH_CHECK(what);
    }
    return CFD_OK;
}

// partial (sum, sum of squares) pairs per output channel the forward kernel can emit for the BatchNorm that follows (0: not on
// this layer -- not a conv6 layer, or its channel chunks are split over workgroups)
int cfd_conv6_stats_slots(const ConvGeom& g) {
    const Conv6Plan P = conv6_plan(g, false);
    return (P.ok && g.ks == 3 && P.ksplit == 1 && P.NI == 3) ? P.gx : 0;  // one record per workgroup of the launch
}

// ------------------------------------------------------------------------------------------------------
// weight gradient: gw[o][c][ky][kx] = sum_{b, y, x} g[b][o][y][x] * in[b][c][clamp(y + ky - pad)][clamp(x + kx - pad)]
// Workgroup = one chunk of 16 input channels x 16 MT output channels x NKY tap rows (all KS columns); it walks tiles of
// 8 images x 16 pixels (TH x TW) in a grid-stride loop.  LDS per tile: the replicate-clamped halo of the input
// [piece][halo pixel][16 channels][8 images] and the upstream gradient [piece][16 pixels][16 MT channels][8 images], 16 bytes per
// (pixel, channel) with the channel slot rotated by the pixel index ((c + pixel) & 15) so that the staging stores -- lanes =
// consecutive pixels of one channel -- and the operand reads -- lanes = 16 channels of one pixel -- both spread over the banks.
// MFMA view: K slot j of lane (q, n) = image j of pixel 4 wave + q; A row n = output channel; B column n = input channel of ONE
// tap per column tile.  Accumulators stay in registers across the workgroup's tiles; one partial slice per pixel group.
// ------------------------------------------------------------------------------------------------------
struct Wg6Tile {
    int TW, TH, tiles_x, tiles_y, LW, LH, HP;  // HP = LH * LW halo pixels
    int tw_shift;
    CfdDiv dHP, dLW;
};

template <int KS, int MT, int NKY>
__global__ __launch_bounds__(256, 2) void k_conv6_wgrad(const float* __restrict__ gout, const float* __restrict__ in,
                                                        float* __restrict__ part, ConvGeom g, Wg6Tile t, int ntiles, int want_gb) {
    constexpr int KK = KS * KS, PAD = KS / 2, NT = NKY * KS, NKG = (KS + NKY - 1) / NKY;
    constexpr int NIW = 3;  // halo items per thread: HP * 16 <= 768 (HP <= 40 for k = 3, 42 for k = 7 with NKY = 2)
    CFD_DYN_SHARED(u4, s_dyn);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, n = lane & 15;
    const int HW = g.H * g.W;
    char* s_in = (char*)s_dyn;                    // [halo pixel][16 slots][piece] x 16 B
    char* s_g = (char*)(s_dyn + 3 * t.HP * 16);   // [16 pixels][MT][16 slots][piece] x 16 B
    const int c0 = blockIdx.y * 16;
    const int mg = blockIdx.z / NKG, ky0 = (blockIdx.z - mg * NKG) * NKY;
    const int mbase = mg * 16 * MT;
    const int tpi = t.tiles_x * t.tiles_y;
    const int pi = 4 * wave + q;                                       // this lane's pixel of the tile (its K step)
    const int pbase = (pi >> t.tw_shift) * t.LW + (pi & (t.TW - 1));   // halo pixel of tap (0, 0)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = zero;
    // this thread's staging items: tile-invariant coordinates and LDS byte offsets
    unsigned ipk[NIW], ilds[NIW];  // input halo items (pixel, channel): lx | ly << 8, channel offset folded into ich
    unsigned ich[NIW];
#pragma unroll
    for (int k = 0; k < NIW; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int c = (int)cfd_div((unsigned)i, t.dHP), hp = i - c * t.HP;
        const int ly = (int)cfd_div((unsigned)hp, t.dLW), lx = hp - ly * t.LW;
        ipk[k] = c < 16 ? (unsigned)(lx | (ly << 8)) : 0x80000000u;  // idle: never committed
        ich[k] = 4u * (unsigned)((c0 + c < g.Ci ? c0 + c : g.Ci - 1) * HW);
        ilds[k] = 48u * (unsigned)(hp * 16 + ((c + hp) & 15));
    }
    unsigned gpk[MT], glds[MT], gch[MT];  // gradient items (pixel, output channel)
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int i = threadIdx.x + 256 * k;
        const int o = i >> 4, p2 = i & 15;
        gpk[k] = (unsigned)((p2 & (t.TW - 1)) | ((p2 >> t.tw_shift) << 8));
        gch[k] = 4u * (unsigned)((mbase + o < g.Co ? mbase + o : g.Co - 1) * HW);
        glds[k] = 48u * (unsigned)((p2 * MT + (o >> 4)) * 16 + (((o & 15) + p2) & 15));
    }
    // bias gradient gb[o] = sum over (b, p) of gout[b][o][p]: the workgroups of the first channel chunk and tap-row group add up the
    // gradient items they stage anyway (one item = 8 images of one pixel; the 16 pixels of a channel sit in 16 consecutive lanes)
    const bool do_gb = want_gb && blockIdx.y == 0 && ky0 == 0;
    float gbacc[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) gbacc[k] = 0.f;
    // the next tile's global loads wait in registers while this tile's MFMAs run (see k_conv6); all of them unconditional from
    // clamped addresses (uniform image pointer + 32-bit byte offset)
    float dri[NIW][8], drg[MT][8];
    bool iok[NIW];  // zero padding: the halo pixel lies inside the image (outside: the operand is zero, not the clamped pixel)
    int gok[MT];  // images of the gradient item that are real (0: pixel outside the image): the only operand that must be zeroed --
                  // input values beside a zero gradient, input channels beyond Ci and output rows beyond Co never reach gw
    const auto issue = [&](int tile) {
        const int bg = tile / tpi, tr = tile - bg * tpi;
        const int ty0 = (tr / t.tiles_x) * t.TH, tx0 = (tr % t.tiles_x) * t.TW;
        const int b0 = bg * 8;
        const int oy = ty0 + ky0 - PAD, ox = tx0 - PAD;
        const float* inb[8];
        const float* gb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int b = b0 + j < g.B ? b0 + j : g.B - 1;
            inb[j] = in + (size_t)b * g.Ci * HW;
            gb[j] = gout + (size_t)b * g.Co * HW;
        }
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            int y = oy + (int)((ipk[k] >> 8) & 255), x = ox + (int)(ipk[k] & 255);
            iok[k] = !g.zpad || ((unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W);
            y = y < 0 ? 0 : (y >= g.H ? g.H - 1 : y);
            x = x < 0 ? 0 : (x >= g.W ? g.W - 1 : x);
            const unsigned off = ich[k] + 4u * (unsigned)(y * g.W + x);
#pragma unroll
            for (int j = 0; j < 8; ++j) dri[k][j] = cfd_ldg_off(inb[j], off);
        }
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            int y = ty0 + (int)(gpk[k] >> 8), x = tx0 + (int)(gpk[k] & 255);
            const bool ok = y < g.H && x < g.W;
            y = y < g.H ? y : g.H - 1;
            x = x < g.W ? x : g.W - 1;
            const unsigned off = gch[k] + 4u * (unsigned)(y * g.W + x);
#pragma unroll
            for (int j = 0; j < 8; ++j) drg[k][j] = cfd_ldg_off(gb[j], off);
            gok[k] = ok ? g.B - b0 : 0;
        }
    };
    int tbeg, tend;
    conv6_tile_range(ntiles, tbeg, tend);
    if (tbeg < tend) issue(tbeg);
    for (int tile = tbeg; tile < tend; ++tile) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int k = 0; k < NIW; ++k) {
            if (!(ipk[k] >> 31)) {
                if (!iok[k]) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) dri[k][j] = 0.f;
                }
                const CfdSplit8x3 sp = cfd_split8x3(dri[k]);
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) *(u4*)(s_in + ilds[k] + 16 * pc) = __builtin_bit_cast(u4, sp.p[pc]);
            }
        }
#pragma unroll
        for (int k = 0; k < MT; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) drg[k][j] = j < gok[k] ? drg[k][j] : 0.f;
            if (do_gb) {  // (uniform per workgroup)
                float sg = ((drg[k][0] + drg[k][1]) + (drg[k][2] + drg[k][3])) + ((drg[k][4] + drg[k][5]) + (drg[k][6] + drg[k][7]));
                sg += cfd_shfl_xor(sg, 1);
                sg += cfd_shfl_xor(sg, 2);
                sg += cfd_shfl_xor(sg, 4);
                sg += cfd_shfl_xor(sg, 8);
                gbacc[k] += sg;
            }
            const CfdSplit8x3 sp = cfd_split8x3(drg[k]);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) *(u4*)(s_g + glds[k] + 16 * pc) = __builtin_bit_cast(u4, sp.p[pc]);
        }
        __syncthreads();
        if (tile + 1 < tend) issue(tile + 1);
        cfd_sched_fence();
        bf16x8 av[MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                av[mt][pc] = __builtin_bit_cast(bf16x8, *(const u4*)(s_g + 48 * ((pi * MT + mt) * 16 + ((n + pi) & 15)) + 16 * pc));
        // column tiles in groups of NG: their B operands are read together and the MFMAs run product-major over the group, so that
        // consecutive MFMAs are independent
        constexpr int NG = KS == 3 ? 3 : (MT == 1 ? KS : 2);  // (k = 5 / 7: NT = 10 / 14 column tiles)
        static_assert(NT % NG == 0, "column tiles per MFMA group");
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // the six products of cfd_mfma_bf16x6, small terms first
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += NG) {
            bf16x8 bv[NG][3];
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int nt = n0 + i, P = pbase + (nt / KS) * t.LW + nt % KS;
                const char* bp = s_in + 48 * (P * 16 + ((n + P) & 15));
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) bv[i][pc] = __builtin_bit_cast(bf16x8, *(const u4*)(bp + 16 * pc));
            }
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < NG; ++i)
                        acc[mt][n0 + i] = cfd_mfma16x16x32_bf16(av[mt][PA[p]], bv[i][PB[p]], acc[mt][n0 + i]);
        }
    }
    // sum of the four waves (each owns 4 of the tile's 16 pixels) in a fixed order, RG column tiles of one output-channel tile at a
    // time (the staging buffers are free by now)
    constexpr int RG = KS == 3 ? 9 : KS;
    static_assert(NT % RG == 0, "column tiles per reduction round");
    float* s_red = (float*)s_dyn;  // [wave][RG][r][lane]
    const size_t n1 = (size_t)g.Co * g.Ci * KK;
    float* dst = part + (size_t)blockIdx.x * (n1 + (want_gb ? g.Co : 0));
    if (do_gb) {
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            const int i = threadIdx.x + 256 * k, o = mbase + (i >> 4);
            if ((i & 15) == 0 && o < g.Co) dst[n1 + o] = gbacc[k];
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int n0 = 0; n0 < NT; n0 += RG) {
            __syncthreads();
#pragma unroll
            for (int nt = 0; nt < RG; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s_red[((wave * RG + nt) * 4 + r) * 64 + lane] = acc[mt][n0 + nt][r];
            __syncthreads();
            for (int e = threadIdx.x; e < RG * 256; e += blockDim.x) {
                const int ln = e & 63, r = (e >> 6) & 3, nt = e >> 8;
                const int o = mbase + 16 * mt + 4 * (ln >> 4) + r, c = c0 + (ln & 15), ky = ky0 + (n0 + nt) / KS;
                if (o < g.Co && c < g.Ci && ky < KS) {  // (the last tap-row group of k = 7 has a dummy row)
                    float sum = 0.f;
#pragma unroll
                    for (int wv = 0; wv < 4; ++wv) sum += s_red[((wv * RG + nt) * 4 + r) * 64 + ln];
                    dst[((size_t)o * g.Ci + c) * KK + ky * KS + (n0 + nt) % KS] = sum;
                }
            }
        }
    }
}

struct Wg6Plan {
    Wg6Tile t;
    int ntiles, groups, chunks, mtw, mgroups, nkg;
    size_t lds, part_bytes;
    bool ok;
};

static Wg6Plan wg6_plan(const ConvGeom& g) {
    Wg6Plan P{};
    if (g.ks != 3 && g.ks != 5 && g.ks != 7) return P;
    const int KS = g.ks, NKY = KS == 3 ? 3 : 2;  // k = 7: tap rows {0,1}, {2,3}, {4,5}, {6}: 14 column tiles per output-channel tile; k = 5: {0,1}, {2,3}, {4}: 10
    Wg6Tile& t = P.t;
    t.TW = (g.W >= 8 || KS >= 5) ? 8 : 4;  // (k = 7 on 4 x 4 tiles would stage a 5 x 10 halo: more than the three items per thread)
    t.TH = 16 / t.TW;
    t.tw_shift = t.TW == 8 ? 3 : 2;
    t.tiles_x = (g.W + t.TW - 1) / t.TW;
    t.tiles_y = (g.H + t.TH - 1) / t.TH;
    t.LW = t.TW + KS - 1;
    t.LH = t.TH + NKY - 1;
    t.HP = t.LH * t.LW;
    t.dHP = cfd_div_make((unsigned)t.HP);
    t.dLW = cfd_div_make((unsigned)t.LW);
    P.ntiles = ((g.B + 7) / 8) * t.tiles_x * t.tiles_y;
    P.chunks = (g.Ci + 15) / 16;
    const int MTall = (g.Co + 15) / 16;
    P.mtw = MTall >= 2 ? 2 : 1;  // (three tiles would need > 256 registers with the prefetch: one wave per SIMD)
    P.mgroups = (MTall + P.mtw - 1) / P.mtw;
    P.nkg = (KS + NKY - 1) / NKY;
    // pixel groups: two workgroups per CU over the launch (measured: 2048 -> 1024 -> 512 workgroups took the ResNet weight
    // gradients from 196 / 141 us to 147 / 115 to ~125 / 100: a workgroup's reduction epilogue and its partial slice are paid once
    // per workgroup, not per tile), at most one per tile, partials capped at ~32 MB
    const int mul = cfd_tune_get(CFD_TUNE_CONV6_WGRAD_MUL);
    long want = (long)(mul > 0 ? mul : 1) * conv6_grid() / ((long)P.chunks * P.mgroups * P.nkg);
    if (want < 1) want = 1;
    const long cap = (32L << 20) / ((long)g.Co * g.Ci * KS * KS * 4 + 1);
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    P.groups = (int)(want < P.ntiles ? want : P.ntiles);
    const size_t stage = (size_t)3 * t.HP * 256 + (size_t)3 * 16 * P.mtw * 256;
    const size_t red = (size_t)4 * (KS == 3 ? 9 : KS) * 256 * sizeof(float);
    P.lds = stage > red ? stage : red;
    P.part_bytes = cfd_align_up((size_t)P.groups * ((size_t)g.Co * g.Ci * KS * KS + g.Co) * sizeof(float), 256);  // (+ bias-gradient row)
    P.ok = P.lds <= 150 * 1024 && t.HP * 16 <= 3 * 256 && (long)g.B * g.Ci * g.H * g.W < (1L << 30) && (long)g.B * g.Co * g.H * g.W < (1L << 30);  // 32-bit byte offsets
    return P;
}

bool cfd_conv6_wgrad_covers(const ConvGeom& g) { return wg6_plan(g).ok; }

size_t cfd_conv6_wgrad_ws_bytes(const ConvGeom& g) {
    const Wg6Plan P = wg6_plan(g);
    return P.ok ? P.part_bytes : 0;
}

int cfd_conv6_wgrad(const float* gout, const float* in, float* gw, float* gb, void* ws, const ConvGeom& g, hipStream_t st,
                    const char* what, CfdPartReduceJob* defer) {
    const Wg6Plan P = wg6_plan(g);
    if (!P.ok) return CFD_ERR_UNSUPPORTED;
    const dim3 grid(P.groups, P.chunks, P.mgroups * P.nkg);
#define W6_L(K_, M_, Y_)                                                                                                          \
    do {                                                                                                                          \
        static bool attr_set = false;                                                                                             \
        if (!attr_set) {                                                                                                          \
            (void)hipFuncSetAttribute((const void*)k_conv6_wgrad<K_, M_, Y_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
            attr_set = true;                                                                                                      \
        }                                                                                                                         \
        hipLaunchKernelGGL((k_conv6_wgrad<K_, M_, Y_>), grid, dim3(256), P.lds, st, gout, in, (float*)ws, g, P.t, P.ntiles,        \
                           gb ? 1 : 0);                                                                                           \
    } while (0)
    if (g.ks == 3) {
        if (P.mtw == 1) W6_L(3, 1, 3);
        else W6_L(3, 2, 3);
    } else if (g.ks == 5) {
        if (P.mtw == 1) W6_L(5, 1, 2);
        else W6_L(5, 2, 2);
    } else {
        if (P.mtw == 1) W6_L(7, 1, 2);
        else W6_L(7, 2, 2);
    }
#undef W6_L
    CFD_LAUNCH_CHECK(what);
    const long n1 = (long)g.Co * g.Ci * g.ks * g.ks;
    if (defer) {
        *defer = cfd_conv_part_reduce_job((const float*)ws, gw, n1 + (gb ? g.Co : 0), P.groups, gb, n1);
        return CFD_OK;
    }
    cfd_conv_part_reduce((const float*)ws, gw, n1 + (gb ? g.Co : 0), P.groups, st, gb, n1);
    CFD_LAUNCH_CHECK(what);
    return CFD_OK;
}
