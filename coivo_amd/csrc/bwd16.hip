// bwd16.hip -- input gradient AND weight gradient of a narrow full-resolution 3x3 layer in ONE pass (round 4).
//
// Why: the 16-channel layers at full resolution (DepthNet iconv1: 16 -> 16 at 256x320) do 6 GFLOP on 42 MB tensors -- they are
// HBM-bound, each kernel alone already near the bandwidth it can get (input gradient 25 us for 126 MB, weight gradient 32 us for 84 MB
// at 16 frames) -- and the two backward kernels of the layer read the SAME two tensors: dY (with a halo for the input gradient) and the
// layer's input x (as the ReLU mask of dx and as the second operand of dW).  Fused, a workgroup stages one dY patch and one x patch per
// 128-pixel tile and runs both products on them: 3 tensor passes (dY, x in; dx out) instead of 5, no separate mask read.
//
//   dx[p][ci] = (x[p][ci] > 0) * sum_{m, co} dy[p + m - 1][co] * w_bwd[ci][m][co]        (m: flipped tap, patch offset (m/3, m%3))
//   dw[co][t][ci] += sum_p dy[p][co] * x[p + t - 1][ci];     db[co] += sum_p dy[p][co]
//
// Layout per workgroup (256 threads = 4 waves, a tile = 8 rows x 16 columns):
//   sG  dY patch  [10 x 18 pixels][16 co]  bf16, 32 B per pixel      sX  x patch  [10 x 18][16 ci]      sW  w_bwd [16 ci][10 taps][16 co]
//   (tap 9 = zeros: K = 9 x 16 = 144 is padded to 5 k-steps of 32).  32-byte pixel pitch: conflict-free for the ds_read_b128 of the
//   input-gradient fragments (conv_common.h: r = 2 slots) and for the transposed 8-byte reads of the weight gradient (8 consecutive
//   pixels = 8 different 32-byte octets), because a fragment row is 16 consecutive pixels of ONE tile row (tow = 16).
//   input gradient:  wave w owns tile rows 2w, 2w+1 (two 16-pixel fragments) x 16 ci: 5 k-steps x 2 = 10 MFMAs per tile, operands
//                    swapped as in k_conv3x3 (accumulator = 4 consecutive channels of one pixel -> 8-byte stores, 512 B per fragment row)
//   weight gradient: fragment f = tap f (16 columns = the 16 ci of one tap), wave w owns taps w, w+4, w+8: 4 k-steps x <= 3 MFMAs,
//                    accumulated in registers over ALL tiles the workgroup walks, one set of fp32 atomics at the end
//   bias gradient:   thread (co, phase) sums dY over the tile from LDS.
// A workgroup walks `tiles_per_wg` consecutive tiles; the next tile's patches are loaded into registers before the MFMAs of the
// current one (as k_wgrad3x3 does).  The grid is capped (tuning: bwd16_wgs) because every workgroup ends with 2304 + 16 atomics on the
// SAME addresses.
//
// The two products run as one pinned software pipeline (operands of step n + 1 requested before the MFMAs of step n); the HEAD form makes
// its gradient patch by MFMA from the split d(pre) patch (make_g below).
// Built WITH -mllvm -amdgpu-mfma-vgpr-form (coivo_amd/build.py): the input-gradient accumulators are read every tile.
#define COLVO_ACC_CONSTRAINT "+v"
#include "conv_common.h"
#include "tuning.h"

namespace colvo {
namespace {

constexpr int TOH = 8, TOW = 16, PH = TOH + 2, PW = TOW + 2, NPIX = PH * PW;   // 180 patch pixels
constexpr int PIXB = 32;                                                    // bytes per patch pixel (16 bf16 channels)
// bytes per ci row of sW: 10 taps x 16 co = 320, padded to 22 sixteen-byte slots -- the weight-fragment ds_read_b128 puts 8 ci rows
// in a half group, conflict-free iff the row pitch is 2 (mod 4) slots (conv_common.h); 20 slots would take every such read twice
constexpr int WROWB = 352;

struct Bwd16K {
    const char* dy;      // [B][H][W][16] bf16
    const char* x;       // [B][H][W][16] bf16: the layer's input (post-ReLU)
    const char* w_bwd;   // [16 ci][9 flipped taps][16 co] bf16
    char* dx;            // [B][H][W][16] bf16
    float* dw;           // [16 co][9][16 ci], added to
    float* db;           // [16] or null
    int B, H, W;
    int tiles_x, tiles_y, ntiles, tiles_per_wg;
    int relu_mask;       // 1: dx is masked by x > 0
    // HEAD form: `dy` is NOT the gradient but the layer's OUTPUT y (post-ReLU), and the gradient is made on the fly as the input
    // gradient of the 3x3 16 -> 1 depth head behind the layer: dy[p][c] = (y[p][c] > 0) * sum_t head_w[t][c] * dpre[p + 1 - t]
    const float* dpre;   // [B][H][W] fp32
    const float* head_w; // [9][16] fp32
    // HEAD form, optional: the head's own weight / bias gradient as well -- dWh[t][c] = sum_p dpre[p] y[p + t - 1][c], db_h = sum_p dpre[p]
    // -- as one partial row of 9 * 16 + 1 floats per WAVE (4 rows per workgroup) for the table reduction of csrc/misc.hip
    float* head_partials;
};

constexpr int DH = PH + 2, DW = PW + 2;                                      // d(pre) patch of the HEAD form: 12 x 20

// MODE 0: dy given.  1: HEAD form (dy made from the layer's output and the depth head's d(pre)).  2: HEAD form + the head's own weight
// gradient.  Compile-time, not a kernel argument: a run-time branch around an MFMA makes hipcc merge the carried accumulators
// through v_mov copies at the join -- reads of MFMA results in front of the guard (tools/isa_check_mfma.py).
// Workgroups per CU by the form's registers (round 5: 106 / 120 / 134 VGPRs): four, four, three -- bwd16_grid() sizes the grid in whole
// rounds of them.  (Round 4's HEAD form needed 168 registers = three per CU; four instead of three measured level in the step,
// 3.354 against 3.356 ms at 32 pairs, and 5 % faster alone at 64 frames.)
#ifndef COLVO_BWD16_HEAD_WGS
#define COLVO_BWD16_HEAD_WGS 4
#endif
template <int MODE>
__global__ __launch_bounds__(NT, MODE == 2 ? 3 : MODE == 1 ? COLVO_BWD16_HEAD_WGS : 4) void k_bwd16(const Bwd16K a) {
    constexpr bool HEAD = MODE >= 1, headw = MODE == 2;
    __shared__ __attribute__((aligned(16))) char sG[NPIX * PIXB];
    __shared__ __attribute__((aligned(16))) char sX[NPIX * PIXB];
    __shared__ __attribute__((aligned(16))) char sW[16 * WROWB];
    __shared__ float sdb[NT];
    __shared__ __attribute__((aligned(16))) uint32_t sDs[HEAD ? DH * DW : 4];   // d(pre) around the patch, split: bf16 hi | bf16 lo << 16
    __shared__ __attribute__((aligned(16))) float sD[MODE == 2 ? DH * DW : 4];  // ... and as fp32 for the head's own weight gradient
    __shared__ __attribute__((aligned(16))) char sY[MODE == 2 ? NPIX * PIXB : 16];   // the layer's output patch (head weight gradient)
    float hb = 0.0f;                                                            // (dWh itself: acc[5], rows c = 4 kg + r, column t = l15)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;

    // ---- weights: [ci][9][co] -> sW [ci][10][co], tap 9 zero ----
    for (int i = tid; i < 16 * 10 * 2; i += NT) {                 // 16-byte granules: (ci, tap, half)
        const int half = i & 1, tap = (i >> 1) % 10, ci = i / 20;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (tap < 9) v = ld16(a.w_bwd + ((ci * 9 + tap) * 16 + half * 8) * 2);
        st16(sW + ci * WROWB + tap * 32 + half * 16, v);
    }

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const long long img_bytes = (long long)a.H * a.W * 16 * 2;
    const long long tot_bytes = img_bytes * a.B;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)(tot_bytes < 0x7fffffffLL ? tot_bytes : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(tot_bytes < 0x7fffffffLL ? tot_bytes : 0x7fffffffLL), 0x00020000);

    // ---- staging: 2 x 360 granules (patch pixel, channel half) over 256 threads: 3 per thread ----
    constexpr int NGRAN = 2 * NPIX * 2;                            // 720
    constexpr int PPF = (NGRAN + NT - 1) / NT;                     // 3
    // per granule ONE register: py | px << 8 | LDS offset << 16 (< 5760); which tensor and which channel half follow from (it, tid) --
    // granule i = it * 256 + tid: half = tid & 1 for every it, and `which` is a comparison of tid with a constant
    int s_pk[PPF];
    const int s_half = tid & 1;
    auto s_which = [&](int it) -> int { const int i = it * NT + tid; return i < NPIX * 2 ? 0 : i < NGRAN ? 1 : 2; };   // 2: no granule
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        const int i = it * NT + tid;
        const int rem = i < NPIX * 2 ? i : i - NPIX * 2;
        const int pix = (rem >> 1) < NPIX ? (rem >> 1) : 0;
        const int py = pix / PW, px = pix - py * PW;
        s_pk[it] = py | (px << 8) | ((pix * PIXB + s_half * 16) << 16);
    }
    struct TileC { int b, ty, tx; };
    auto tile_next = [&](TileC& c) {
        if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.b; } }
    };
    TileC cur;
    {
        const int t = __builtin_amdgcn_readfirstlane(t_begin);
        cur.b = t / tiles_per_img;
        const int tr_ = t - cur.b * tiles_per_img;
        cur.ty = tr_ / a.tiles_x; cur.tx = tr_ - cur.ty * a.tiles_x;
    }
    u32x4 pv[PPF];
    float dpv = 0.0f;                                              // HEAD: this thread's element of the 12 x 20 d(pre) patch
    const __amdgpu_buffer_rsrc_t rdp = __builtin_amdgcn_make_buffer_rsrc((void*)(HEAD ? a.dpre : nullptr), 0,
                                                                          HEAD ? (int)((long long)a.B * a.H * a.W * 4) : 0, 0x00020000);
    const int d_py = tid / DW, d_px = tid - d_py * DW;             // (tid < 240)
    auto load_tile = [&](const TileC& c) {
        const int oy0 = c.ty * TOH - 1, ox0 = c.tx * TOW - 1;
        const int base = (int)((long long)c.b * img_bytes);       // (< 1 GiB per tensor: checked on the host)
        if constexpr (HEAD) {
            const int vy = oy0 - 1 + d_py, vx = ox0 - 1 + d_px;
            const bool inb = tid < DH * DW && ((unsigned)vy < (unsigned)a.H) && ((unsigned)vx < (unsigned)a.W);
            dpv = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rdp, inb ? (vy * a.W + vx) * 4 : OOB_OFF, c.b * a.H * a.W * 4, 0));
        }
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int vy = oy0 + (s_pk[it] & 0xff), vx = ox0 + ((s_pk[it] >> 8) & 0xff);
            const bool inb = ((unsigned)vy < (unsigned)a.H) && ((unsigned)vx < (unsigned)a.W);
            const int off = inb ? ((vy * a.W + vx) * 16 + s_half * 8) * 2 : OOB_OFF;
            if (s_which(it) == 0) pv[it] = bld16(rdy, off, base);
            else pv[it] = bld16(rx, s_which(it) == 1 ? off : OOB_OFF, base);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int lds = (int)((unsigned)s_pk[it] >> 16);
            if (s_which(it) == 0) {
                st16(sG + lds, pv[it]);                           // HEAD: the layer's output y -- make_g() turns it into the gradient in place
                if constexpr (headw) st16(sY + lds, pv[it]);
            }
            else if (s_which(it) == 1) st16(sX + lds, pv[it]);
        }
        if constexpr (HEAD) {
            if (tid < DH * DW) {
                const uint16_t hi = f2bf(dpv);
                sDs[tid] = (uint32_t)hi | ((uint32_t)f2bf(dpv - bf2f(hi)) << 16);
                if constexpr (headw) sD[tid] = dpv;
            }
        }
    };
    // HEAD: the gradient g[p][c] = (y[p][c] > 0) * sum_t head_w[t][c] * dpre[p + 1 - t] on the 180 patch pixels, rounded to bf16 as
    // csrc/misc.hip k_depth_head_dgrad16 stores it -- BY MFMA (round 5).  Round 4 made it on the VALU, a thread per staged granule: 9
    // d(pre) reads and 72 fp32 FMAs per granule from 72 registers of head weights -- 195 of the tile loop's 324 VALU instructions in a
    // kernel that is issue-bound, and the registers that kept the MFMA phases from being pipelined.  As a product it is tiny:
    // [16 c] x [K = 9 taps] per pixel; both factors are fp32, so each is split into bf16 hi + lo and K carries the three products that
    // matter per tap -- hi_w hi_d, hi_w lo_d, lo_w hi_d: the sum is the fp32 product to ~2^-16 relative (lo_w lo_d and the second-order
    // remainders are dropped), far inside the bf16 rounding the result gets.  ONE MFMA per fragment of patch pixels, 3 per wave and
    // tile; the weight operand lives in 4 registers; y is read from / g written to the SAME 8 bytes of sG by the same lane.
    //   K layout: lane group kg = 0, 1, 2 holds tap ROW kg (taps a, b, c = 3 kg + 0, 1, 2): slots hh_a hh_b hh_c hl_a hl_b hl_c lh_a lh_b;
    //   group 3 holds the three left-over lh_c and five zero slots.  A lane's eight d(pre) operands then come from THREE words of the
    //   split patch (one row segment; group 3: one word per row) and four v_perm_b32 with the same constant selectors in every lane --
    //   3 LDS reads per fragment where a slot-by-slot gather took 8 (counters of that form: LDS busy 45 % of the kernel, 16 M
    //   bank-conflict cycles more than the plain form).
    //   The 180 patch pixels are cut into 12 fragments of FIFTEEN (lane 15 of a fragment idles): fragment 3 w + j of wave w is the 5 x 3
    //   block (rows 5 (w >> 1) .., columns 9 (w & 1) + 3 j ..) of the 10 x 18 patch, lane l15 its pixel (l15 / 3, l15 % 3) -- so that a
    //   lane's LDS addresses are the same registers for j = 0, 1, 2 with the fragment as an immediate offset.
    u32x4 hA = u32x4{0u, 0u, 0u, 0u};
    int g_da[3] = {0, 0, 0};                                       // byte addresses in sDs of this lane's three words, fragment j = 0
    int g_ya = 0;                                                  // byte address in sG of this lane's 4 channels, fragment j = 0
    const bool g_on = l15 < 15;
    if constexpr (HEAD) {
        const int lp = l15 < 15 ? l15 : 14;                        // (the idle lane repeats its neighbour's reads)
        const int gpy = 5 * (wave >> 1) + lp / 3, gpx = 9 * (wave & 1) + lp % 3;
        g_ya = (gpy * PW + gpx) * PIXB + kg * 8;
        // d(pre) of tap (ky, kx) for patch pixel (py, px) sits at sDs[(py + 2 - ky) * DW + px + 2 - kx]
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int ky = kg < 3 ? kg : i, kx = kg < 3 ? i : 2;
            g_da[i] = ((gpy + 2 - ky) * DW + gpx + 2 - kx) * 4;
        }
        uint32_t hv[4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            uint32_t pair = 0u;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = 2 * i2 + e;                          // slot of this lane's group
                // group 0..2: slot i = (part i / 3, tap 3 kg + i % 3) for i < 6, (lo_w, tap 3 kg + i - 6) for i = 6, 7;  group 3: lo_w of tap 3 i + 2
                const bool used = kg < 3 || i < 3;
                const int t = kg < 3 ? 3 * kg + (i < 6 ? i % 3 : i - 6) : (i < 3 ? 3 * i + 2 : 8);
                const bool low = kg == 3 || i >= 6;                // the lo_w hi_d products
                const float w = a.head_w[t * 16 + l15];
                const uint16_t hi = f2bf(w);
                const uint16_t lo = f2bf(w - bf2f(hi));
                pair |= (used ? (uint32_t)(low ? lo : hi) : 0u) << (16 * e);
            }
            hv[i2] = pair;
        }
        hA = u32x4{hv[0], hv[1], hv[2], hv[3]};
    }
    auto make_g = [&]() {
        u32x4 bfr[3];
        u32x2 yv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            uint32_t w[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) w[i] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(sDs) + g_da[i] + 12 * j);
            // v_perm_b32(s0, s1, sel): bytes 0..3 = s1, 4..7 = s0; a word is (hi = bytes 0, 1 | lo = bytes 2, 3)
            bfr[j] = u32x4{__builtin_amdgcn_perm(w[1], w[0], 0x05040100u),      // hh_a hh_b   (group 3: lh_c of rows 0, 1)
                           __builtin_amdgcn_perm(w[0], w[2], 0x07060100u),      // hh_c hl_a   (group 3: lh_c of row 2, -)
                           __builtin_amdgcn_perm(w[2], w[1], 0x07060302u),      // hl_b hl_c
                           __builtin_amdgcn_perm(w[1], w[0], 0x05040100u)};     // lh_a lh_b: hi_d again
            yv[j] = *reinterpret_cast<const u32x2*>(sG + g_ya + 3 * PIXB * j);
        }
        f32x4 gacc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            gacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hA), __builtin_bit_cast(bf16x8, bfr[j]), gacc[j], 0, 0, 0);
        mfma_result_guard<bf16_t>(gacc);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // bf16 > 0  <=>  its bits, read as a signed 16-bit integer, are > 0
            const bool p0 = (int32_t)(yv[j][0] << 16) > 0, p1 = (int32_t)(yv[j][0] & 0xffff0000u) > 0;
            const bool p2 = (int32_t)(yv[j][1] << 16) > 0, p3 = (int32_t)(yv[j][1] & 0xffff0000u) > 0;
            u32x2 o;
            o[0] = pack2bf(p0 ? gacc[j][0] : 0.0f, p1 ? gacc[j][1] : 0.0f);
            o[1] = pack2bf(p2 ? gacc[j][2] : 0.0f, p3 ? gacc[j][3] : 0.0f);
            if (g_on) *reinterpret_cast<u32x2*>(sG + g_ya + 3 * PIXB * j) = o;
        }
    };

    // ---- per-lane constants of the two products ----
    // input gradient: fragment mf = tile row 2 * wave + mf, pixel column l15
    int g_base[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) g_base[mf] = ((2 * wave + mf) * PW + l15) * PIXB + (kg & 1) * 16;
    const int w_base = l15 * WROWB + (kg & 1) * 16;                // + tap * 32
    // weight gradient: K position (ks, kg, h, q) -> tile pixel p (k_wgrad3x3's order: a read group covers 8 consecutive pixels)
    const int q = l15 >> 2, pp = lane & 3;
    constexpr int FPW = 3;                                         // taps wave, wave + 4, wave + 8 (< 9)
    // acc[0..1]: the input gradient of the current tile (cleared every tile); acc[2..4]: the weight gradient, carried over all tiles.
    // ONE array so that one mfma_result_guard closes every chain at the end of a tile: hipcc rotates the carried accumulators
    // through v_mov copies at the loop edge -- reads of MFMA results that must not come early (tools/isa_check_mfma.py)
    constexpr int NACC = 2 + FPW + (headw ? 1 : 0);                // (+ 1: MODE 2's head weight gradient)
    f32x4 acc[NACC];
#pragma unroll
    for (int fi = 0; fi < NACC; ++fi) acc[fi] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.0f;
    const int db_co = tid & 15, db_ph = tid >> 4;                  // 16 phases x 8 pixels

    if (t_begin < t_end) load_tile(cur);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        store_tile();
        __syncthreads();
        const TileC here = cur;
        tile_next(cur);
        if (t + 1 < t_end) load_tile(cur);                         // in flight during everything below
        if constexpr (HEAD) {
            make_g();
            __syncthreads();
        }

        if (a.db) {
            float s0 = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = db_ph * 8 + j;                       // tile pixel
                s0 += bf2f(*reinterpret_cast<const uint16_t*>(sG + (((p >> 4) + 1) * PW + (p & 15) + 1) * PIXB + db_co * 2));
            }
            dbacc += s0;
        }

        if constexpr (headw) {
            {
                // head weight gradient: wave w takes the 32 tile pixels of k-step w.  A = y^T by transposed reads (as the weight
                // gradient below), B[q][t] = dpre[q + 1 - t] built from the fp32 d(pre) patch (bf16: like y), columns t >= 9 zero
                int yo[2], p0[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 32 * wave + 16 * (kg >> 1) + 8 * h + 4 * (kg & 1);        // + q: this lane's run of 4 K positions
                    p0[h] = p;
                    const int pq = p + q;
                    yo[h] = (((pq >> 4) + 1) * PW + (pq & 15) + 1) * PIXB + 4 * pp * 2;
                }
                const s16x4 ylo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sY + yo[0]));
                const s16x4 yhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sY + yo[1]));
                const s16x8 ay = s16x8{ylo[0], ylo[1], ylo[2], ylo[3], yhi[0], yhi[1], yhi[2], yhi[3]};
                const int t = l15 > 8 ? 8 : l15, ky = t / 3, kx = t - 3 * ky;
                unsigned bw[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // pixels p0[h] .. p0[h] + 3 lie in one tile row (runs of 4 never cross a multiple of 16)
                    const float* dr = sD + ((p0[h] >> 4) + 3 - ky) * DW + (p0[h] & 15) + 3 - kx;    // sD(py + 2 - ky, px + 2 - kx), patch = tile + 1
                    const float d0 = l15 < 9 ? dr[0] : 0.0f, d1 = l15 < 9 ? dr[1] : 0.0f, d2 = l15 < 9 ? dr[2] : 0.0f, d3 = l15 < 9 ? dr[3] : 0.0f;
                    bw[2 * h] = pack2bf(d0, d1);
                    bw[2 * h + 1] = pack2bf(d2, d3);
                }
                const u32x4 bd = u32x4{bw[0], bw[1], bw[2], bw[3]};
                acc[NACC - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ay), __builtin_bit_cast(bf16x8, bd), acc[NACC - 1], 0, 0, 0);
                if (lane < 32) {                                      // bias: this wave's 32 pixels
                    const int p = 32 * wave + lane;
                    hb += sD[((p >> 4) + 2) * DW + (p & 15) + 2];
                }
            }
        }
        // ---- input gradient (5 k-steps of two taps x 2 pixel fragments) and weight gradient (4 k-steps of 32 tile pixels x 3 taps) as ONE
        // pinned software pipeline: the operands of step n + 1 are requested before the MFMAs of step n.  (Round 4 left the order to
        // hipcc, which sank every LDS read to just in front of the MFMA that needs it -- 22 exposed LDS round trips per tile with three
        // waves per SIMD to hide them -- and turned the wave-uniform `tap < 9` of the third weight-gradient fragment into a branch per
        // MFMA.  Now every wave runs all three fragments; the third accumulator of waves 1..3 is scratch and never flushed.) ----
        acc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            u32x4 dwv[2], dgv[2][2];
            // (the weight gradient's third tap fragment is not double-buffered: it is requested at the head of its own step, behind the
            // next step's operands, and waits out two MFMAs -- four registers that decide between three and four workgroups per CU)
            s16x8 waf[2], wbf[2][FPW - 1], wb2;
            auto dreads = [&](int s, u32x4& wv, u32x4 (&gv)[2]) {
                const int tap = 2 * s + (kg >> 1);                 // 9: the zero tap
                const int tp = tap > 8 ? 8 : tap;                  // its patch address: any valid one
                const int ky = tp / 3, kx = tp - 3 * ky;
                wv = ld16(sW + w_base + tap * 32);
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) gv[mf] = ld16(sG + g_base[mf] + (ky * PW + kx) * PIXB);
            };
            auto wtap = [&](int ks, int fi) -> s16x8 {
                int xo[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 32 * ks + 16 * (kg >> 1) + 8 * h + 4 * (kg & 1) + q;
                    xo[h] = ((p >> 4) * PW + (p & 15)) * PIXB + 4 * pp * 2;
                }
                const int tap = min(wave + 4 * fi, 8);              // wave-uniform; waves 1..3 repeat tap 8 into their scratch accumulator
                const int to = ((tap / 3) * PW + (tap % 3)) * PIXB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + xo[0] + to));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + xo[1] + to));
                return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            auto wreads = [&](int ks, s16x8& af, s16x8 (&bf)[FPW - 1]) {
                int go[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int p = 32 * ks + 16 * (kg >> 1) + 8 * h + 4 * (kg & 1) + q;
                    go[h] = (((p >> 4) + 1) * PW + (p & 15) + 1) * PIXB + 4 * pp * 2;
                }
                const s16x4 glo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + go[0]));
                const s16x4 ghi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + go[1]));
                af = s16x8{glo[0], glo[1], glo[2], glo[3], ghi[0], ghi[1], ghi[2], ghi[3]};
#pragma unroll
                for (int fi = 0; fi < FPW - 1; ++fi) bf[fi] = wtap(ks, fi);
            };
            dreads(0, dwv[0], dgv[0]);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                if (s + 1 < 5) dreads(s + 1, dwv[(s + 1) & 1], dgv[(s + 1) & 1]);
                else wreads(0, waf[0], wbf[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
                    acc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, dwv[s & 1]), __builtin_bit_cast(bf16x8, dgv[s & 1][mf]),
                                                                      acc[mf], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wb2 = wtap(ks, FPW - 1);
                if (ks + 1 < 4) wreads(ks + 1, waf[(ks + 1) & 1], wbf[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fi = 0; fi < FPW - 1; ++fi)
                    acc[2 + fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, waf[ks & 1]), __builtin_bit_cast(bf16x8, wbf[ks & 1][fi]),
                                                                          acc[2 + fi], 0, 0, 0);
                acc[2 + FPW - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, waf[ks & 1]), __builtin_bit_cast(bf16x8, wb2),
                                                                           acc[2 + FPW - 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- dx of this tile: mask by x > 0 (from the staged patch), 4 channels = 8 bytes per lane ----
        mfma_result_guard<bf16_t>(acc);
        {
            const int oyt = here.ty * TOH, oxt = here.tx * TOW;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                const int oy = 2 * wave + mf, ox = l15;
                const int gy = oyt + oy, gx = oxt + ox;
                f32x4 v = acc[mf];
                if (a.relu_mask) {
                    const u32x2 xm = *reinterpret_cast<const u32x2*>(sX + ((oy + 1) * PW + ox + 1) * PIXB + kg * 8);
                    // bf16 > 0  <=>  sign bit clear and not zero
                    const uint32_t m0 = xm[0] & 0xffffu, m1 = xm[0] >> 16, m2 = xm[1] & 0xffffu, m3 = xm[1] >> 16;
                    v[0] = (m0 != 0 && m0 < 0x8000u) ? v[0] : 0.0f;
                    v[1] = (m1 != 0 && m1 < 0x8000u) ? v[1] : 0.0f;
                    v[2] = (m2 != 0 && m2 < 0x8000u) ? v[2] : 0.0f;
                    v[3] = (m3 != 0 && m3 < 0x8000u) ? v[3] : 0.0f;
                }
                if (gy < a.H && gx < a.W) {
                    u32x2 o;
                    o[0] = pack2bf(v[0], v[1]);
                    o[1] = pack2bf(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(a.dx + (((long long)here.b * a.H + gy) * a.W + gx) * 32 + kg * 8) = o;
                }
            }
        }
    }

    // ---- flush: dw[co][tap][ci] += accw (D rows = co 4 kg + r, cols = ci l15), db ----
    mfma_result_guard<bf16_t>(acc);                 // (a workgroup without tiles never entered the loop)
#pragma unroll
    for (int fi = 0; fi < FPW; ++fi) {
        const int tap = wave + 4 * fi;
        if (tap < 9) {
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(a.dw + ((4 * kg + r) * 9 + tap) * 16 + l15, acc[2 + fi][r]);
        }
    }
    if constexpr (headw) {
        {
            float* row = a.head_partials + ((size_t)blockIdx.x * 4 + wave) * (9 * 16 + 1);
            if (l15 < 9) {
#pragma unroll
                for (int r = 0; r < 4; ++r) row[l15 * 16 + 4 * kg + r] = acc[NACC - 1][r];
            }
            const float hbs = wave_sum(hb);
            if (lane == 0) row[9 * 16] = hbs;
        }
    }
    if (a.db) {
        __syncthreads();
        sdb[tid] = dbacc;
        __syncthreads();
        if (tid < 16) {
            float s = 0.0f;
#pragma unroll
            for (int ph = 0; ph < 16; ++ph) s += sdb[ph * 16 + tid];
            atomicAdd(a.db + tid, s);
        }
    }
}

// ---- the depth head's weight / bias gradient by MFMA (colvo_depth_head_wgrad_mfma) ----
// dWh[t][c] = sum_p dpre[p] y[p + t - 1][c] = sum_q y[q][c] dpre[q + 1 - t], db = sum_p dpre[p]: a [16 c] x [9 t] product over the pixels.
// The VALU kernel of csrc/misc.hip (thread = strided pixels, 48 accumulators, nine scattered d(pre) loads per pixel) runs at 26 + 5 us
// for a 47 MB read at 16 frames -- on the weight-gradient streams, which are what ends the backward pass at 8 pairs.  Here a workgroup
// walks 8 x 16 tiles: y tile (256 granules, one per thread) and the 10 x 18 d(pre) patch in LDS, wave w takes the 32 pixels of k-step
// w: A = y^T by transposed reads, B[q][t] = dpre[q + 1 - t] rounded to bf16 (as y is), ONE MFMA per wave and tile; one partial row per
// workgroup for the table reduction (k_head_wgrad_reduce).
struct HeadWgradK {
    const char* y;          // [B][H][W][16] bf16
    const float* dpre;      // [B][H][W]
    float* partials;        // [grid][145]
    int B, H, W;
    int tiles_x, tiles_y, ntiles, tiles_per_wg;
};

__global__ __launch_bounds__(NT, 4) void k_head_wgrad_mfma(const HeadWgradK a) {
    __shared__ __attribute__((aligned(16))) char sY[TOH * TOW * PIXB];          // tile pixels only
    __shared__ __attribute__((aligned(16))) float sD[PH * PW];                   // d(pre) on tile + 1
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4, q = l15 >> 2, pp = lane & 3;
    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const long long img_bytes = (long long)a.H * a.W * 16 * 2;
    const long long tot_bytes = img_bytes * a.B;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)(tot_bytes < 0x7fffffffLL ? tot_bytes : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.dpre, 0, (int)((long long)a.B * a.H * a.W * 4), 0x00020000);
    const int g_pix = tid >> 1, g_half = tid & 1;                 // this thread's y granule: tile pixel, channel half
    const int g_oy = g_pix >> 4, g_ox = g_pix & 15;
    const int d_py = tid / PW, d_px = tid - d_py * PW;             // (tid < 180) its d(pre) patch element
    struct TileC { int b, ty, tx; };
    auto tile_next = [&](TileC& c) {
        if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.b; } }
    };
    TileC cur;
    {
        const int t = __builtin_amdgcn_readfirstlane(t_begin);
        cur.b = t / tiles_per_img;
        const int tr_ = t - cur.b * tiles_per_img;
        cur.ty = tr_ / a.tiles_x; cur.tx = tr_ - cur.ty * a.tiles_x;
    }
    u32x4 yv = u32x4{0u, 0u, 0u, 0u};
    float dv = 0.0f;
    auto load_tile = [&](const TileC& c) {
        const int oy0 = c.ty * TOH, ox0 = c.tx * TOW;
        {
            const int vy = oy0 + g_oy, vx = ox0 + g_ox;
            const bool inb = vy < a.H && vx < a.W;
            yv = bld16(ry, inb ? ((vy * a.W + vx) * 16 + g_half * 8) * 2 : OOB_OFF, (int)((long long)c.b * img_bytes));
        }
        {
            const int vy = oy0 - 1 + d_py, vx = ox0 - 1 + d_px;
            const bool inb = tid < PH * PW && ((unsigned)vy < (unsigned)a.H) && ((unsigned)vx < (unsigned)a.W);
            dv = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rd, inb ? (vy * a.W + vx) * 4 : OOB_OFF, c.b * a.H * a.W * 4, 0));
        }
    };
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    float hb = 0.0f;
    // this lane's K positions of k-step `wave`: two runs of 4 consecutive tile pixels (k_wgrad3x3's order)
    int yo[2], p0[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        p0[h] = 32 * wave + 16 * (kg >> 1) + 8 * h + 4 * (kg & 1);
        yo[h] = (p0[h] + q) * PIXB + 4 * pp * 2;
    }
    const int t9 = l15 > 8 ? 8 : l15, ky = t9 / 3, kx = t9 - 3 * ky;
    if (t_begin < t_end) load_tile(cur);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
        st16(sY + g_pix * PIXB + g_half * 16, yv);
        if (tid < PH * PW) sD[tid] = dv;
        __syncthreads();
        tile_next(cur);
        if (t + 1 < t_end) load_tile(cur);
        const s16x4 ylo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sY + yo[0]));
        const s16x4 yhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sY + yo[1]));
        const s16x8 ay = s16x8{ylo[0], ylo[1], ylo[2], ylo[3], yhi[0], yhi[1], yhi[2], yhi[3]};
        unsigned bw[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // dpre[q + 1 - t] for the run's 4 pixels: patch coordinates (oy + 1 + 1 - ky, ox + 1 + 1 - kx)
            const float* dr = sD + ((p0[h] >> 4) + 2 - ky) * PW + (p0[h] & 15) + 2 - kx;
            const float d0 = l15 < 9 ? dr[0] : 0.0f, d1 = l15 < 9 ? dr[1] : 0.0f, d2 = l15 < 9 ? dr[2] : 0.0f, d3 = l15 < 9 ? dr[3] : 0.0f;
            bw[2 * h] = pack2bf(d0, d1);
            bw[2 * h + 1] = pack2bf(d2, d3);
        }
        const u32x4 bd = u32x4{bw[0], bw[1], bw[2], bw[3]};
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ay), __builtin_bit_cast(bf16x8, bd), acc[0], 0, 0, 0);
        if (lane < 32) {
            const int p = 32 * wave + lane;
            hb += sD[((p >> 4) + 1) * PW + (p & 15) + 1];
        }
        mfma_result_guard<bf16_t>(acc);                 // (carried across the loop: closed every tile, see k_bwd16)
    }
    // ONE partial row per workgroup: the four waves' sums meet in LDS in a fixed order (the reduction's second launch reads its table
    // column-wise, 580 bytes from row to row: a quarter of the rows is a quarter of its cache lines -- beside k_bwd16 in the step it ran
    // 27-35 us for 4096 rows, 7 us alone)
    __shared__ float sR[4][9 * 16 + 1];
    if (l15 < 9) {
#pragma unroll
        for (int r = 0; r < 4; ++r) sR[wave][l15 * 16 + 4 * kg + r] = acc[0][r];
    }
    const float hbs = wave_sum(hb);
    if (lane == 0) sR[wave][9 * 16] = hbs;
    __syncthreads();
    if (tid < 9 * 16 + 1)
        a.partials[(size_t)blockIdx.x * (9 * 16 + 1) + tid] = (sR[0][tid] + sR[1][tid]) + (sR[2][tid] + sR[3][tid]);
}

// ---- input gradient of a stride-2 3x3 layer w.r.t. TWO input channels, as fp32 planes, by MFMA (colvo_conv_dgrad_planes) ----
// PoseNet's first layer (8 -> 16, stride 2): only the two depth channels of its input gradient are wanted, as planes for DepthNet's
// backward pass; the kernel sits between the two networks' backward passes on the main chain.  The VALU form (csrc/misc.hip: thread =
// pixel pair, weights read from LDS once per FMA) took 13-24 us there for a 5 MB read and a 5 MB write.  Here a 2 x 2 block of input
// pixels (2Y + py, 2X + px) is ONE row of a small product: it sees the four gradient pixels g[Y .. Y+1][X .. X+1] (16 channels each:
// K = 64) and its 4 classes x 2 channels are 8 of the 16 output columns,
//     dx[2Y+py][2X+px][c] = sum_{gy, gx, co} W'[(gy, gx, co)][(py, px, c)] g[Y+gy][X+gx][co],
//     W' = w[co][ky][kx][c] with ky = 1 (py = 0, gy = 0), 2 (py = 1, gy = 0), 0 (py = 1, gy = 1), kx likewise -- zero elsewhere.
// A wave takes 16 consecutive blocks of one block row per step: the gradient granules go from memory straight into the B operand (lane
// = block, k-group = (gx, channel half); no LDS), the weights -- fp32 masters split into THREE bf16 terms (hi + mid + lo: the product is
// exact to fp32 round-off, the plane test's bar) -- live in registers as the A operand: 6 MFMAs per 64 input pixels.
struct PlanesK {
    const char* g;          // [B][Ho][Wo][16] bf16
    const float* w;         // [16][9][Cin] fp32
    float* dst;             // [2][B][1][Hi][Wi]
    int B, Hi, Wi, Ho, Wo, Cin, c_begin, accumulate;
    int groups_x, ngroups, groups_per_wave;
};

__global__ __launch_bounds__(NT) void k_dgrad_planes_s2_mfma(const PlanesK a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    // A operand: row n = l15 = (py, px, c), k = 32 q + 8 kg + j = (gy = q, gx = kg >> 1, co = 8 (kg & 1) + j)
    bf16x8 wa[2][3];
    {
        const int n = l15, c = n & 1, px = (n >> 1) & 1, py = (n >> 2) & 1, gx = kg >> 1;
        const int kx = px == 0 ? (gx == 0 ? 1 : -1) : (gx == 0 ? 2 : 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ky = py == 0 ? (q == 0 ? 1 : -1) : (q == 0 ? 2 : 0);
            const bool on = n < 8 && kx >= 0 && ky >= 0;
            uint16_t t0[8], t1[8], t2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int co = 8 * (kg & 1) + j;
                const float wv = on ? a.w[((size_t)co * 9 + ky * 3 + kx) * a.Cin + a.c_begin + c] : 0.0f;
                t0[j] = f2bf(wv);
                const float r1 = wv - bf2f(t0[j]);
                t1[j] = f2bf(r1);
                t2[j] = f2bf(r1 - bf2f(t1[j]));
            }
            auto pack = [](const uint16_t (&t)[8]) {
                const u32x4 v = u32x4{(unsigned)t[0] | ((unsigned)t[1] << 16), (unsigned)t[2] | ((unsigned)t[3] << 16),
                                      (unsigned)t[4] | ((unsigned)t[5] << 16), (unsigned)t[6] | ((unsigned)t[7] << 16)};
                return __builtin_bit_cast(bf16x8, v);
            };
            wa[q][0] = pack(t0); wa[q][1] = pack(t1); wa[q][2] = pack(t2);
        }
    }
    const long long img_g = (long long)a.Ho * a.Wo * 32;                     // bytes of one image's gradient
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)a.g, 0, (int)(img_g * a.B), 0x00020000);
    const size_t plane = (size_t)a.B * a.Hi * a.Wi;
    const int g0 = (blockIdx.x * 4 + wave) * a.groups_per_wave;
    for (int i = 0; i < a.groups_per_wave; ++i) {
        const int gid = g0 + i;                                              // wave-uniform
        if (gid >= a.ngroups) break;
        const int rowid = gid / a.groups_x, X = (gid - rowid * a.groups_x) * 16 + l15;
        const int b = rowid / a.Ho, Y = rowid - b * a.Ho;
        const int gxx = X + (kg >> 1);
        u32x4 gv[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool inb = (Y + q < a.Ho) && (gxx < a.Wo);
            gv[q] = bld16(rg, inb ? (((Y + q) * a.Wo + gxx) * 16 + 8 * (kg & 1)) * 2 : OOB_OFF, (int)(b * img_g));
        }
        f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[q][sp], __builtin_bit_cast(bf16x8, gv[q]), acc[0], 0, 0, 0);
        mfma_result_guard<bf16_t>(acc);
        // lane (block l15, kg < 2): rows n = 4 kg + r = (py = kg, px = r >> 1, c = r & 1) -> input row 2Y + kg, columns 2X, 2X + 1
        if (kg < 2 && X < a.Wo) {
            const size_t o = ((size_t)b * a.Hi + 2 * Y + kg) * a.Wi + 2 * X;
            float2* d0 = reinterpret_cast<float2*>(a.dst + o);
            float2* d1 = reinterpret_cast<float2*>(a.dst + plane + o);
            float2 v0 = make_float2(acc[0][0], acc[0][2]), v1 = make_float2(acc[0][1], acc[0][3]);
            if (a.accumulate) { const float2 p0 = *d0, p1 = *d1; v0.x += p0.x; v0.y += p0.y; v1.x += p1.x; v1.y += p1.y; }
            *d0 = v0; *d1 = v1;
        }
    }
}

}  // namespace
}  // namespace colvo

using namespace colvo;

// (called by colvo_conv_dgrad_planes, csrc/misc.hip, for the shapes this form covers)
int colvo::launch_dgrad_planes_s2_mfma(const void* g, const float* w, int Cin, int c_begin, int B, int Hi, int Wi, int Ho, int Wo,
                                       float* dst, int accumulate, hipStream_t stream) {
    PlanesK k{};
    k.g = (const char*)g; k.w = w; k.dst = dst; k.B = B; k.Hi = Hi; k.Wi = Wi; k.Ho = Ho; k.Wo = Wo; k.Cin = Cin; k.c_begin = c_begin;
    k.accumulate = accumulate;
    k.groups_x = (Wo + 15) / 16; k.ngroups = B * Ho * k.groups_x;
    // 2 groups per wave up to 8 pairs of 256x320 (1280 workgroups), more beyond
    k.groups_per_wave = std::max((int)TUNE(planes_groups), (k.ngroups + 4 * 2048 - 1) / (4 * 2048));
    const int wgs = (k.ngroups + 4 * k.groups_per_wave - 1) / (4 * k.groups_per_wave);
    colvo::launch(k_dgrad_planes_s2_mfma, dim3((unsigned)wgs), dim3(NT), 0, stream, k);
    return 0;
}

static int head_wgrad_mfma_grid(int B, int H, int W, int* tiles_per_wg) {
    const int ntiles = B * ((W + TOW - 1) / TOW) * ((H + TOH - 1) / TOH);
    int wgs = std::min(ntiles, (int)TUNE(head_wgrad_wgs));
    const int tpw = (ntiles + wgs - 1) / wgs;
    if (tiles_per_wg) *tiles_per_wg = tpw;
    return (ntiles + tpw - 1) / tpw;
}

extern "C" int colvo_depth_head_wgrad_mfma_rows(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1 || (long long)B * H * W * 32 >= 0x40000000LL) return 0;
    return head_wgrad_mfma_grid(B, H, W, nullptr);
}

extern "C" int colvo_depth_head_wgrad_mfma(const void* y, const float* dpre, int B, int H, int W, float* partials, colvo_stream_t stream) {
    COLVO_CHECK_ARG(y && dpre && partials && colvo_depth_head_wgrad_mfma_rows(B, H, W) > 0, "colvo_depth_head_wgrad_mfma: bad arguments");
    HeadWgradK k{};
    k.y = (const char*)y; k.dpre = dpre; k.partials = partials; k.B = B; k.H = H; k.W = W;
    k.tiles_x = (W + TOW - 1) / TOW; k.tiles_y = (H + TOH - 1) / TOH; k.ntiles = B * k.tiles_x * k.tiles_y;
    const int wgs = head_wgrad_mfma_grid(B, H, W, &k.tiles_per_wg);
    colvo::launch(k_head_wgrad_mfma, dim3((unsigned)wgs), dim3(NT), 0, (hipStream_t)stream, k);
    COLVO_CHECK_LAUNCH("k_head_wgrad_mfma");
    return 0;
}

extern "C" int colvo_conv_bwd_fused_ok(const ColvoConvDesc* d) {
    if (!d) return 0;
    const long long bytes = (long long)d->B * d->Hi * d->Wi * 16 * 2;
    return d->dtype == COLVO_BF16 && d->ksize == 3 && d->stride == 1 && d->C0 == 16 && d->C1 == 0 && d->Cout == 16 && !d->up0 &&
           d->Ho == d->Hi && d->Wo == d->Wi && bytes < 0x40000000LL && TUNE(bwd16) != 0;
}

// MODE as in k_bwd16 (0: dy given, 1: HEAD form, 2: HEAD form + the head's weight gradient)
static int bwd16_grid(const ColvoConvDesc* d, int mode, int* tiles_per_wg) {
    const int ntiles = d->B * ((d->Wi + TOW - 1) / TOW) * ((d->Hi + TOH - 1) / TOH);
    // grid: bwd16_wgs workgroups at 16 frames, more from 40 tiles per workgroup on, at most four times as many -- every workgroup ends
    // with 2320 atomics on the same addresses.  Beyond bwd16_wgs the grid is a WHOLE number of rounds of the workgroups the chip holds
    // (256 CUs x 4 / 3 / 2 by the form's registers): until round 5 the HEAD form ran 1024 workgroups at 64 frames on 768 slots -- a
    // second round on a third of the chip; counters: the same wave cycles as the plain form in 1.7 x the time -- 226 us against 125
    int wgs = (int)TUNE(bwd16_wgs);
    const int slots = 256 * (mode == 2 ? 3 : mode == 1 ? COLVO_BWD16_HEAD_WGS : 4);
    const int want = std::min(4 * wgs, ntiles / 40);
    if (want > wgs) wgs = std::max(1, (want + slots / 2) / slots) * slots;
    if (wgs > ntiles) wgs = ntiles;
    const int tpw = (ntiles + wgs - 1) / wgs;
    if (tiles_per_wg) *tiles_per_wg = tpw;
    return (ntiles + tpw - 1) / tpw;
}

extern "C" int colvo_conv_bwd_fused_head_rows(const ColvoConvDesc* d) {
    if (!colvo_conv_bwd_fused_ok(d)) return 0;
    return 4 * bwd16_grid(d, 2, nullptr);
}

extern "C" int colvo_conv_bwd_fused(const ColvoConvDesc* d, const void* dy, const void* w_bwd, const void* x, int relu_mask, void* dx,
                                    float* dw, float* db, const float* head_dpre, const float* head_w, float* head_partials,
                                    colvo_stream_t stream) {
    COLVO_CHECK_ARG(d && dy && w_bwd && x && dx && dw && ((head_dpre == nullptr) == (head_w == nullptr)) && (!head_partials || head_dpre),
                    "colvo_conv_bwd_fused: null pointer argument");
    COLVO_CHECK_ARG(colvo_conv_bwd_fused_ok(d), "colvo_conv_bwd_fused: only bf16 16 -> 16 stride-1 layers over one directly stored source "
                                                "below 1 GiB per tensor (colvo_conv_bwd_fused_ok)");
    Bwd16K k{};
    k.dy = (const char*)dy; k.x = (const char*)x; k.w_bwd = (const char*)w_bwd; k.dx = (char*)dx; k.dw = dw; k.db = db;
    k.B = d->B; k.H = d->Hi; k.W = d->Wi; k.relu_mask = relu_mask;
    k.tiles_x = (k.W + TOW - 1) / TOW; k.tiles_y = (k.H + TOH - 1) / TOH;
    k.ntiles = k.B * k.tiles_x * k.tiles_y;
    const int wgs = bwd16_grid(d, head_partials ? 2 : head_dpre ? 1 : 0, &k.tiles_per_wg);
    k.dpre = head_dpre; k.head_w = head_w; k.head_partials = head_partials;
    if (head_partials) colvo::launch(k_bwd16<2>, dim3((unsigned)wgs), dim3(NT), 0, (hipStream_t)stream, k);
    else if (head_dpre) colvo::launch(k_bwd16<1>, dim3((unsigned)wgs), dim3(NT), 0, (hipStream_t)stream, k);
    else colvo::launch(k_bwd16<0>, dim3((unsigned)wgs), dim3(NT), 0, (hipStream_t)stream, k);
    COLVO_CHECK_LAUNCH("k_bwd16");
    return 0;
}
