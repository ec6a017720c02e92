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
//   sU  input patch [12 x 20 pixels][16 ci] bf16, 32 B per pixel       sY  y on [10 x 18 positions][16 co] bf16
//   sWh the head's weight operand [16 rows][10 taps][16 c] (see below), read inside the head's pipeline
//   the layer's weights ([16 co][10 taps][16 ci], tap 9 = zeros): five MFMA operand fragments per lane, resident in registers (round 5)
//   layer:  position fragments of 16 consecutive patch positions (12 fragments cover 192 >= 180), wave w owns fragments 3w .. 3w + 2;
//           5 k-steps of two taps; operands swapped as in k_conv3x3 (accumulator = 4 consecutive channels of one position)
//   head:   an MFMA of the same shape (16 pixels x K = 9 taps x 16 channels) against a weight operand whose row 0 is the bf16 rounding
//           of the fp32 head weights and row 1 what that rounding left over: hi + lo sums = the fp32-weight product to 2^-17.  (The
//           first version evaluated the head on the VALU, two threads per pixel: 280 of the kernel's ~480 issue slots per wave and
//           tile -- the kernel was issue-bound at 33.6 us in the step for 89 MB of traffic.)
// A workgroup walks `tiles_per_wg` consecutive tiles with the next TWO tiles' patches in flight (two register sets); the operand
// fragments of k-step s + 1 are requested before the MFMAs of step s and the order is pinned (round 5: DESIGN.md section 3.2).
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

__global__ __launch_bounds__(NT, 4) void k_fwd16_head(const Fwd16K a) {
    __shared__ __attribute__((aligned(16))) char sU[NU * PIXB];
    __shared__ __attribute__((aligned(16))) char sY[(NY + 12) * PIXB];        // (+ 12: the dummy positions 180 .. 191 of the last fragment)
    // head weights as an MFMA operand, [16 rows][10 taps][16 c] bf16, row pitch 352 B (conflict-free ds_read_b128 of 16 rows): read per
    // tile inside the head's pipeline -- resident they cost the 20 registers the second tile of loads in flight needs
    __shared__ __attribute__((aligned(16))) char sWh[16 * WROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;

    // Weight operands: RESIDENT IN REGISTERS for the whole walk, read straight from memory -- k-step s covers the taps 2 s and 2 s + 1
    // (tap 9 = zeros), lane (l15, kg) holds row l15, tap 2 s + (kg >> 1), channels 8 (kg & 1) .. + 7.  (Round 4 staged both operands in
    // LDS and re-read them per tile: 10 of the tile's 35 fragment reads, each in front of the MFMA that waits for it.)
    //   layer: row = output channel.   head: row 0 = the bf16 rounding of the fp32 head weights, row 1 = the bf16 rounding of what that
    //   left over (hi + lo carries 16 mantissa bits: the two rows' products sum to the fp32-weight product to 2^-17), rows 2..15 zero.
    u32x4 wfr[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + (kg >> 1);
        const int tq = tap > 8 ? 8 : tap;
        const u32x4 wv = ld16(a.w + ((l15 * 9 + tq) * 16 + (kg & 1) * 8) * 2);
        wfr[s] = tap < 9 ? wv : u32x4{0u, 0u, 0u, 0u};
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
    const int wh_base = l15 * WROWB + (kg & 1) * 16 + (kg >> 1) * 32;       // + 64 s: this lane's slot of k-step s (tap 2 s + (kg >> 1); tap 9 = zeros)
    const float hb = a.head_b[0];
    // bias of this lane's 4 output channels (4 kg .. 4 kg + 3)
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 4 * kg);

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_wg);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int img_px = a.H * a.W;
    const long long img_bytes = (long long)img_px * 16 * 2;
    const long long tot_bytes = img_bytes * a.B;                              // < 1 GiB (colvo_conv_head_fused_ok)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)tot_bytes, 0x00020000);
    // outputs through buffer descriptors too: 32-bit offsets, and a position that must not be written gets OOB_OFF (the store is dropped)
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)tot_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)a.depth, 0, (int)(tot_bytes >> 3), 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(a.pose_in ? a.pose_in : a.y), 0,
                                                                        a.pose_in ? (int)(tot_bytes >> 2) : 0, 0x00020000);

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
    // TWO tiles of patch loads are in flight per workgroup (15 KB; four workgroups per CU: 62 KB): with one, the kernel moved 2.4-2.7 TB/s
    // however few instructions the tile took -- 31 KB in flight per CU against a loaded HBM latency of a few microseconds
    u32x4 pvA[PPF], pvB[PPF];
    auto load_tile = [&](const TileC& c, u32x4 (&pv)[PPF]) {
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
    // tap offsets of this lane's k-slots (patch / y grid)
    int u_tap[5], y_tap[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + (kg >> 1);
        const int tp = tap > 8 ? 8 : tap;
        const int ky = tp / 3, kx = tp - 3 * ky;
        u_tap[s] = (ky * UW + kx) * PIXB;
        y_tap[s] = (ky * YW + kx) * PIXB;
    }
    // head: an MFMA like the layer's -- fragment mf = tile row 2 wave + mf, 16 pixels; D rows 0 / 1 (lanes kg == 0) = hi / lo sums
    int hy_base[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) hy_base[mf] = ((2 * wave + mf) * YW + l15) * PIXB + (kg & 1) * 16;

    TileC ld = cur;                                                // the next tile to request
    int t_ld = t_begin;
    auto request = [&](u32x4 (&pv)[PPF]) {
        if (t_ld < t_end) { load_tile(ld, pv); tile_next(ld); ++t_ld; }
    };
    request(pvA);
    request(pvB);
    auto tile_body = [&](u32x4 (&pv)[PPF]) {
        __syncthreads();                                           // the previous tile's head has finished reading sY, its MFMAs sU
#pragma unroll
        for (int it = 0; it < PPF; ++it)
            if (s_on[it]) st16(sU + s_lds[it], pv[it]);
        __syncthreads();
        const TileC here = cur;
        tile_next(cur);
        request(pv);                                               // the tile after next, into the registers just stored

        // ---- the layer: 5 k-steps x 3 position fragments; the fragments of step s + 1 are requested BEFORE the MFMAs of step s and the
        // order is pinned (left alone hipcc sinks every read to just in front of its MFMA: read, wait out the LDS round trip, one MFMA) ----
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        {
            u32x4 ub[2][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) ub[0][j] = ld16(sU + u_base[j] + u_tap[0]);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                if (s + 1 < 5) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) ub[(s + 1) & 1][j] = ld16(sU + u_base[j] + u_tap[s + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wfr[s]), __builtin_bit_cast(bf16x8, ub[s & 1][j]),
                                                                     acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mfma_result_guard<bf16_t>(acc);
        const int oyt = here.ty * TOH, oxt = here.tx * TOW;
        const int ybase = (int)((long long)here.b * img_bytes);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int gy = oyt - 1 + pos_y[j], gx = oxt - 1 + pos_x[j];
            const bool inimg = ((unsigned)gy < (unsigned)a.H) && ((unsigned)gx < (unsigned)a.W);
            const float v0 = fmaxf(acc[j][0] + bv[0], 0.0f), v1 = fmaxf(acc[j][1] + bv[1], 0.0f);
            const float v2 = fmaxf(acc[j][2] + bv[2], 0.0f), v3 = fmaxf(acc[j][3] + bv[3], 0.0f);
            u32x2 o;
            o[0] = inimg ? (pack2bf(v0, v1)) : 0u;      // zero outside the image: the head's zero padding
            o[1] = inimg ? (pack2bf(v2, v3)) : 0u;
            *reinterpret_cast<u32x2*>(sY + y_lds[j]) = o;
            const bool centre = pos_y[j] >= 1 && pos_y[j] <= TOH && pos_x[j] >= 1 && pos_x[j] <= TOW;
            __builtin_amdgcn_raw_buffer_store_b64(o, ry, (centre && inimg) ? (gy * a.W + gx) * 32 + kg * 8 : OOB_OFF, ybase, 0);
        }
        __syncthreads();
        // ---- head on the 128 centre pixels: 5 k-steps of two taps, 2 MFMAs each, same pipeline ----
        {
            f32x4 hacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            u32x4 yb[2][2], hw[2];
            hw[0] = ld16(sWh + wh_base);
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) yb[0][mf] = ld16(sY + hy_base[mf] + y_tap[0]);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                if (s + 1 < 5) {
                    hw[(s + 1) & 1] = ld16(sWh + wh_base + 64 * (s + 1));
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) yb[(s + 1) & 1][mf] = ld16(sY + hy_base[mf] + y_tap[s + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
                    hacc[mf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hw[s & 1]), __builtin_bit_cast(bf16x8, yb[s & 1][mf]),
                                                                       hacc[mf], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            mfma_result_guard<bf16_t>(hacc);
            // The two fragments' sums sit in the lanes kg == 0 of two registers; one row swap (v_permlane16_swap: lanes 16..31 of the
            // first <-> lanes 0..15 of the second) puts fragment 1 next to fragment 0, and the sigmoid / depth arithmetic (two fp32
            // divisions, an exp: ~50 VALU instructions) runs ONCE with 32 live lanes instead of twice with 16
            const float s0 = hacc[0][0] + hacc[0][1], s1 = hacc[1][0] + hacc[1][1];
            const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, s0), __builtin_bit_cast(uint32_t, s1), false, false);
            const float pre = hb + __builtin_bit_cast(float, (uint32_t)sw[0]);                  // lanes kg = 0 / 1: fragment 0 / 1
            const int gy = oyt + 2 * wave + (kg & 1), gx = oxt + l15;
            const bool ok = kg < 2 && gy < a.H && gx < a.W;
            const float sig = 1.0f / (1.0f + expf(-pre));
            const float dep = 1.0f / (a.lo + (a.hi - a.lo) * sig);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, dep), rd, ok ? (gy * a.W + gx) * 4 : OOB_OFF, here.b * img_px * 4, 0);
            if (a.pose_in) {
                const int Bh = a.B >> 1, pair = here.b < Bh ? here.b : here.b - Bh;
                __builtin_amdgcn_raw_buffer_store_b16((short)f2bf(dep), rp, ok ? ((gy * a.W + gx) * 8 + 6 + (here.b >= Bh)) * 2 : OOB_OFF,
                                                      pair * img_px * 16, 0);
            }
        }
    };
    for (int t = t_begin; t < t_end; t += 2) {
        tile_body(pvA);
        if (t + 1 < t_end) tile_body(pvB);
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
    // grid: fwd16_wgs workgroups, about one per 8 tiles beyond that -- in WHOLE rounds of the 1024 workgroups the chip holds (4 per CU at
    // the kernel's 124 registers; round 4's 88 registers held 5 per CU and its 1280 workgroups at 16 frames were one round)
    int wgs = (int)TUNE(fwd16_wgs);
    const int slots = 1024, want = std::min(8 * wgs, k.ntiles / std::max(1, (int)TUNE(fwd16_tiles_per_wg)));
    if (want > wgs) wgs = std::max(1, (want + slots / 2) / slots) * slots;
    if (wgs > k.ntiles) wgs = k.ntiles;
    k.tiles_per_wg = (k.ntiles + wgs - 1) / wgs;
    wgs = (k.ntiles + k.tiles_per_wg - 1) / k.tiles_per_wg;
    colvo::launch(k_fwd16_head, dim3((unsigned)wgs), dim3(NT), 0, (hipStream_t)stream, k);
    COLVO_CHECK_LAUNCH("k_fwd16_head");
    return 0;
}
