// fwd16.hip -- the narrow full-resolution layer (16 -> 16, DepthNet's iconv1) AND the 3x3 16 -> 1 depth head behind it in ONE pass
// (round 4; the forward twin of bwd16.hip).
//
// Why: both are HBM-bound at 256x320 -- iconv1 reads and writes a 42 MB tensor (21 us at 16 frames), the head reads those 42 MB again
// (18 us) to make 5 MB of depth -- and the forward pass is one serial chain: every microsecond counts once.  Fused, a workgroup
// computes the layer's output y on a tile PLUS a one-pixel ring (10 x 18 positions from a 12 x 20 input patch: 1.4 x the MFMAs, which
// the layer has to spare), writes the 8 x 16 centre to memory (the backward pass needs y), keeps all 180 positions in LDS and
// evaluates the head on them: y is never read back.
//
//   y[p][co]  = relu(bias[co] + sum_{t, ci} w[co][t][ci] x[p + t - 1][ci])               (zero outside the image: the head's padding)
//   depth[p]  = 1 / (lo + (hi - lo) sigmoid(hb + sum_{t, c} hw[t][c] y[p + t - 1][c]))
//
// Layout per workgroup (256 threads = 4 waves, a tile = 8 rows x 16 columns of outputs):
//   sU  input patch [12 x 20 pixels][16 ci] bf16, 32 B per pixel       sW  w [16 co][10 taps][16 ci] (tap 9 = zeros), row pitch 352 B
//   sY  y on [10 x 18 positions][16 co] bf16
//   layer:  position fragments of 16 consecutive patch positions (12 fragments cover 192 >= 180), wave w owns fragments 3w .. 3w + 2;
//           5 k-steps of two taps; operands swapped as in k_conv3x3 (accumulator = 4 consecutive channels of one position)
//   head:   an MFMA of the same shape (16 pixels x K = 9 taps x 16 channels) against a weight operand whose row 0 is the bf16 rounding
//           of the fp32 head weights and row 1 what that rounding left over: hi + lo sums = the fp32-weight product to 2^-17.  (The
//           first version evaluated the head on the VALU, two threads per pixel: 280 of the kernel's ~480 issue slots per wave and
//           tile -- the kernel was issue-bound at 33.6 us in the step for 89 MB of traffic.)
// A workgroup walks `tiles_per_wg` consecutive tiles with the next tile's patch in flight during the MFMAs.
#define COLVO_ACC_CONSTRAINT "+v"
#include "conv_common.h"

namespace colvo {
namespace {

constexpr int TOH = 8, TOW = 16;
constexpr int YH = TOH + 2, YW = TOW + 2, NY = YH * YW;       // 180 positions of y
constexpr int UH = TOH + 4, UW = TOW + 4, NU = UH * UW;       // 240 input-patch pixels
constexpr int PIXB = 32, WROWB = 352;

struct Fwd16K {
    const char* x;        // [B][H][W][16] bf16
    const char* w;        // [16 co][9][16 ci] bf16 (forward operand layout)
    const float* bias;    // [16]
    const float* head_w;  // [9][16] fp32
    const float* head_b;  // [1]
    char* y;              // [B][H][W][16] bf16
    float* depth;         // [B][1][H][W] fp32
    char* pose_in;        // optional: PoseNet's input [B / 2][H][W][8] bf16 -- depth of image b goes to channel 6 (b < B/2) or 7 of pair b mod B/2
    float lo, hi;         // 1 / max_depth, 1 / min_depth
    int B, H, W;
    int tiles_x, tiles_y, ntiles, tiles_per_wg;
};

__global__ __launch_bounds__(NT, 2) void k_fwd16_head(const Fwd16K a) {
    __shared__ __attribute__((aligned(16))) char sU[NU * PIXB];
    __shared__ __attribute__((aligned(16))) char sY[(NY + 12) * PIXB];        // (+ 12: the dummy positions 180 .. 191 of the last fragment)
    __shared__ __attribute__((aligned(16))) char sW[16 * WROWB];
    // head weights as an MFMA operand: [16 rows][10 taps][16 c] bf16 like sW; row 0 = the bf16 rounding of the fp32 weights, row 1 = the
    // bf16 rounding of what that left over (hi + lo carries 16 mantissa bits: the sum of the two rows' products is the fp32-weight
    // product to 2^-17), rows 2..15 zero
    __shared__ __attribute__((aligned(16))) char sWh[16 * WROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;

    for (int i = tid; i < 16 * 10 * 2; i += NT) {                 // weights -> sW [co][10][ci], tap 9 zero
        const int half = i & 1, tap = (i >> 1) % 10, co = i / 20;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (tap < 9) v = ld16(a.w + ((co * 9 + tap) * 16 + half * 8) * 2);
        st16(sW + co * WROWB + tap * 32 + half * 16, v);
    }
    for (int i = tid; i < 16 * WROWB / 4; i += NT) reinterpret_cast<uint32_t*>(sWh)[i] = 0u;
    __syncthreads();
    if (tid < 9 * 16) {
        const int tq = tid >> 4, c = tid & 15;
        const float wv = a.head_w[tid];
        const uint16_t hi = f2bf(wv);
        const uint16_t lo = f2bf(wv - bf2f(hi));
        *reinterpret_cast<uint16_t*>(sWh + 0 * WROWB + tq * 32 + c * 2) = hi;
        *reinterpret_cast<uint16_t*>(sWh + 1 * WROWB + tq * 32 + c * 2) = lo;
    }
    const float hb = a.head_b[0];
    // bias of this lane's 4 output channels (4 kg .. 4 kg + 3)
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 4 * kg);

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const long long img_bytes = (long long)a.H * a.W * 16 * 2;
    const long long tot_bytes = img_bytes * a.B;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(tot_bytes < 0x7fffffffLL ? tot_bytes : 0x7fffffffLL), 0x00020000);

    // staging: 480 granules (patch pixel, channel half) over 256 threads
    constexpr int NGRAN = NU * 2, PPF = (NGRAN + NT - 1) / NT;     // 2
    int s_py[PPF], s_px[PPF], s_lds[PPF], s_half[PPF];
    bool s_on[PPF];
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        const int i = it * NT + tid;
        const int pix = i >> 1, half = i & 1;
        s_on[it] = i < NGRAN;
        s_py[it] = pix / UW; s_px[it] = pix - s_py[it] * UW; s_half[it] = half;
        s_lds[it] = pix * PIXB + half * 16;
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
    auto load_tile = [&](const TileC& c) {
        const int oy0 = c.ty * TOH - 2, ox0 = c.tx * TOW - 2;
        const int base = (int)((long long)c.b * img_bytes);
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int vy = oy0 + s_py[it], vx = ox0 + s_px[it];
            const bool inb = s_on[it] && ((unsigned)vy < (unsigned)a.H) && ((unsigned)vx < (unsigned)a.W);
            pv[it] = bld16(rx, inb ? ((vy * a.W + vx) * 16 + s_half[it] * 8) * 2 : OOB_OFF, base);
        }
    };

    // per-lane constants: the three position fragments of this wave
    int u_base[3], y_lds[3], pos_y[3], pos_x[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int pp = 16 * (3 * wave + j) + l15;                       // position index in the 10 x 18 grid (>= 180: dummy)
        y_lds[j] = pp * PIXB + kg * 8;
        if (pp >= NY) pp = 0;
        const int py = pp / YW, px = pp - py * YW;
        pos_y[j] = (16 * (3 * wave + j) + l15 < NY) ? py : -100;  // dummy positions fail every bounds test below
        pos_x[j] = px;
        u_base[j] = (py * UW + px) * PIXB + (kg & 1) * 16;        // input patch pixel of tap (0, 0)
    }
    const int w_base = l15 * WROWB + (kg & 1) * 16;
    // head: an MFMA like the layer's -- fragment mf = tile row 2 wave + mf, 16 pixels; D rows 0 / 1 (lanes kg == 0) = hi / lo sums
    int hy_base[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) hy_base[mf] = ((2 * wave + mf) * YW + l15) * PIXB + (kg & 1) * 16;

    if (t_begin < t_end) load_tile(cur);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();                                           // the previous tile's head has finished reading sY, its MFMAs sU
#pragma unroll
        for (int it = 0; it < PPF; ++it)
            if (s_on[it]) st16(sU + s_lds[it], pv[it]);
        __syncthreads();
        const TileC here = cur;
        tile_next(cur);
        if (t + 1 < t_end) load_tile(cur);

        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = 2 * s + (kg >> 1);
            const int tp = tap > 8 ? 8 : tap;
            const int ky = tp / 3, kx = tp - 3 * ky;
            const u32x4 wv = ld16(sW + w_base + tap * 32);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const u32x4 uv = ld16(sU + u_base[j] + (ky * UW + kx) * PIXB);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, uv), acc[j], 0, 0, 0);
            }
        }
        mfma_result_guard<bf16_t>(acc);
        const int oyt = here.ty * TOH, oxt = here.tx * TOW;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int gy = oyt - 1 + pos_y[j], gx = oxt - 1 + pos_x[j];
            const bool inimg = ((unsigned)gy < (unsigned)a.H) && ((unsigned)gx < (unsigned)a.W);
            u32x2 o = u32x2{0u, 0u};
            if (inimg) {
                const float v0 = fmaxf(acc[j][0] + bv[0], 0.0f), v1 = fmaxf(acc[j][1] + bv[1], 0.0f);
                const float v2 = fmaxf(acc[j][2] + bv[2], 0.0f), v3 = fmaxf(acc[j][3] + bv[3], 0.0f);
                o[0] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                o[1] = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
            }
            *reinterpret_cast<u32x2*>(sY + y_lds[j]) = o;          // zero outside the image: the head's zero padding
            const bool centre = pos_y[j] >= 1 && pos_y[j] <= TOH && pos_x[j] >= 1 && pos_x[j] <= TOW;
            if (centre && inimg) *reinterpret_cast<u32x2*>(a.y + (((long long)here.b * a.H + gy) * a.W + gx) * 32 + kg * 8) = o;
        }
        __syncthreads();
        // ---- head on the 128 centre pixels: 5 k-steps of two taps, 2 MFMAs each ----
        {
            f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int tap = 2 * s + (kg >> 1);
                const int tp = tap > 8 ? 8 : tap;
                const int ky = tp / 3, kx = tp - 3 * ky;
                const u32x4 wv = ld16(sWh + w_base + tap * 32);
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const u32x4 yv = ld16(sY + hy_base[mf] + (ky * YW + kx) * PIXB);
                    hacc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, yv), hacc[mf], 0, 0, 0);
                }
            }
            mfma_result_guard<bf16_t>(hacc);
            if (kg == 0) {
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const float pre = hb + (hacc[mf][0] + hacc[mf][1]);
                    const int gy = oyt + 2 * wave + mf, gx = oxt + l15;
                    if (gy < a.H && gx < a.W) {
                        const float sig = 1.0f / (1.0f + expf(-pre));
                        const float dep = 1.0f / (a.lo + (a.hi - a.lo) * sig);
                        a.depth[((long long)here.b * a.H + gy) * a.W + gx] = dep;
                        if (a.pose_in) {
                            const int Bh = a.B >> 1, pair = here.b < Bh ? here.b : here.b - Bh;
                            *reinterpret_cast<uint16_t*>(a.pose_in + ((((long long)pair * a.H + gy) * a.W + gx) * 8 + 6 + (here.b >= Bh)) * 2) = f2bf(dep);
                        }
                    }
                }
            }
        }
    }
}

}  // namespace
}  // namespace colvo

using namespace colvo;

extern "C" int colvo_conv_head_fused_ok(const ColvoConvDesc* d) {
    if (!d) return 0;
    const long long bytes = (long long)d->B * d->Hi * d->Wi * 16 * 2;
    return d->dtype == COLVO_BF16 && d->ksize == 3 && d->stride == 1 && d->C0 == 16 && d->C1 == 0 && d->Cout == 16 && !d->up0 && d->relu &&
           d->Ho == d->Hi && d->Wo == d->Wi && bytes < 0x40000000LL && TUNE(fwd16) != 0;
}

extern "C" int colvo_conv_head_fused(const ColvoConvDesc* d, const void* x, const void* w_fwd, const float* bias, const float* head_w,
                                     const float* head_b, float min_depth, float max_depth, void* y, float* depth, void* pose_in,
                                     colvo_stream_t stream) {
    COLVO_CHECK_ARG(d && x && w_fwd && bias && head_w && head_b && y && depth, "colvo_conv_head_fused: null pointer argument");
    COLVO_CHECK_ARG(!pose_in || d->B % 2 == 0, "colvo_conv_head_fused: pose_in needs a pair batch (B = 2 * pairs)");
    COLVO_CHECK_ARG(colvo_conv_head_fused_ok(d), "colvo_conv_head_fused: only bf16 16 -> 16 stride-1 ReLU layers over one directly stored "
                                                 "source below 1 GiB per tensor (colvo_conv_head_fused_ok)");
    COLVO_CHECK_ARG(min_depth > 0 && max_depth > min_depth, "colvo_conv_head_fused: bad depth range");
    Fwd16K k{};
    k.x = (const char*)x; k.w = (const char*)w_fwd; k.bias = bias; k.head_w = head_w; k.head_b = head_b; k.y = (char*)y; k.depth = depth;
    k.pose_in = (char*)pose_in;
    k.lo = 1.0f / max_depth; k.hi = 1.0f / min_depth;
    k.B = d->B; k.H = d->Hi; k.W = d->Wi;
    k.tiles_x = (k.W + TOW - 1) / TOW; k.tiles_y = (k.H + TOH - 1) / TOH;
    k.ntiles = k.B * k.tiles_x * k.tiles_y;
    int wgs = (int)TUNE(fwd16_wgs);
    wgs = std::max(wgs, std::min(8 * wgs, k.ntiles / 8));
    if (wgs > k.ntiles) wgs = k.ntiles;
    k.tiles_per_wg = (k.ntiles + wgs - 1) / wgs;
    wgs = (k.ntiles + k.tiles_per_wg - 1) / k.tiles_per_wg;
    colvo::launch(k_fwd16_head, dim3((unsigned)wgs), dim3(NT), 0, (hipStream_t)stream, k);
    COLVO_CHECK_LAUNCH("k_fwd16_head");
    return 0;
}
