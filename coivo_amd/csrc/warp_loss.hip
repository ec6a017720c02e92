// warp_loss.hip -- a3..a7 of SURVEY.md §8: ONE fused kernel for
//   project -> bilinear-sample -> LCC-recalibrate -> SSIM(3x3, reflect) + L1 -> masked sum
// and its hand-derived backward (recompute-in-backward: nothing but 4 floats is saved).
//
// Concept: /root/reference/README.md:1 ("Photometric Consistency"), :7 ("alignment of geometric
// projections between consecutive frames"), :5/:7 (LCC "recalibrating the luminosity values of
// adjacent frames").  Results are specified by oracle/colvo_spec.py photometric_loss().
//
// Roofline: HBM.  Algorithmic bytes (fp32, per pixel): fwd 28 (tgt 12 + ref 12 + depth 4),
// bwd 32 (same reads + d_depth 4).  DESIGN.md §kernels.
//
// Work decomposition.
//   forward : "marching wave" -- one wave per 62-column strip marches down the rows; horizontal window sums by
//             DPP wave shifts, vertical by a rolling 3-row register window; no LDS, no barriers; loads of the next
//             rows fly under the SSIM arithmetic of the current one.
//   backward: the same marching organisation with two halo lanes / rows per side: three rolling 3-row register
//             windows (evaluated sample, horizontal window sums, horizontally gathered derivative coefficients);
//             every pixel is evaluated once, rows / columns beyond the image border hold the REFLECTED pixel,
//             which realises the SSIM reflection pad.
// HBM reads are row-contiguous per plane (lanes = consecutive columns); the 4-tap gather of `ref` is served by
// L1/L2 for smooth flows; loads are buffer loads (32-bit lane offset + scalar plane offset).
#include <type_traits>

#include <algorithm>
#include <stdlib.h>

#include "common.h"
#include "tuning.h"

namespace colvo {
namespace {

constexpr int NT = 256;
constexpr float SSIM_C1 = 0.01f * 0.01f;
constexpr float SSIM_C2 = 0.03f * 0.03f;
constexpr float Z_EPS = 1e-3f;

struct Geo {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22;
    float tx, ty, tz;
    float fx, fy, cx, cy;
    float ifx, ify;
    float a, b;
};
constexpr int GEO_N = 20;

// Thread 0 evaluates the per-image rotation exactly in the oracle's operation order
// (pose_vec2mat: R = Rz Ry Rx, no FMA contraction), everybody picks it up as wave-uniform scalars.
// `level`: intrinsics of the 2x2-average-pooled image, `level` times over (spec: scale_intrinsics, same operation order)
__device__ __forceinline__ void geo_compute(const float* pose, const float* K, const float* la,
                                            const float* lb, int b, float* s, int level = 0) {
#pragma clang fp contract(off)
    const float* p = pose + 6 * b;
    const float* k = K + 9 * b;
    float sx, cx, sy, cy, sz, cz;
    sx = sinf(p[3]); cx = cosf(p[3]);
    sy = sinf(p[4]); cy = cosf(p[4]);
    sz = sinf(p[5]); cz = cosf(p[5]);
    s[0] = cz * cy;
    s[1] = cz * sy * sx - sz * cx;
    s[2] = cz * sy * cx + sz * sx;
    s[3] = sz * cy;
    s[4] = sz * sy * sx + cz * cx;
    s[5] = sz * sy * cx - cz * sx;
    s[6] = -sy;
    s[7] = cy * sx;
    s[8] = cy * cx;
    s[9] = p[0]; s[10] = p[1]; s[11] = p[2];
    float kfx = k[0], kfy = k[4], kcx = k[2], kcy = k[5];
    for (int l = 0; l < level; ++l) {
        kfx = kfx * 0.5f; kfy = kfy * 0.5f;
        kcx = (kcx - 0.5f) * 0.5f; kcy = (kcy - 0.5f) * 0.5f;
    }
    s[12] = kfx; s[13] = kfy; s[14] = kcx; s[15] = kcy;
    s[16] = 1.0f / kfx; s[17] = 1.0f / kfy;
    s[18] = la ? la[b] : 1.0f;
    s[19] = lb ? lb[b] : 0.0f;
}

__device__ __forceinline__ Geo geo_load(const float* s) {
    Geo g;
    g.r00 = uniform_f(s[0]); g.r01 = uniform_f(s[1]); g.r02 = uniform_f(s[2]);
    g.r10 = uniform_f(s[3]); g.r11 = uniform_f(s[4]); g.r12 = uniform_f(s[5]);
    g.r20 = uniform_f(s[6]); g.r21 = uniform_f(s[7]); g.r22 = uniform_f(s[8]);
    g.tx = uniform_f(s[9]); g.ty = uniform_f(s[10]); g.tz = uniform_f(s[11]);
    g.fx = uniform_f(s[12]); g.fy = uniform_f(s[13]); g.cx = uniform_f(s[14]); g.cy = uniform_f(s[15]);
    g.ifx = uniform_f(s[16]); g.ify = uniform_f(s[17]);
    g.a = uniform_f(s[18]); g.b = uniform_f(s[19]);
    return g;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// reciprocal refined by one Newton step (~0.5 ulp): for the cancelling gradient sums
__device__ __forceinline__ float nr_rcp(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// index of the pixel a (possibly padded / overhanging) coordinate refers to: 1-px reflection pad,
// anything further out is clamped onto the pad (never consumed).
__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = max(-1, min(i, n));
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// how many taps of the 3-window centred at q land (after reflection) on pixel p
__device__ __forceinline__ float window_mult(int q, int p, int n) {
    int c = (q == p);
    int lo = q - 1, hi = q + 1;
    if (lo < 0) lo = -lo;
    if (hi >= n) hi = 2 * n - 2 - hi;
    c += (lo == p) + (hi == p);
    return (float)c;
}

// One image's planes behind buffer descriptors: loads take a 32-bit per-lane byte offset plus a scalar
// plane offset (no 64-bit VALU address arithmetic), and are bounds-checked by the hardware.
struct Img {
    __amdgpu_buffer_rsrc_t ref, tgt, dep;
    __amdgpu_buffer_rsrc_t dr;      // the reference frame's own depth (geometric-consistency variant only)
    int plane4;     // bytes per plane
};
__device__ __forceinline__ Img img_make(const float* tgt, const float* ref, const float* depth, int b, int H, int W) {
    Img im;
    const size_t plane = (size_t)H * W;
    im.plane4 = (int)(plane * 4);
    im.ref = __builtin_amdgcn_make_buffer_rsrc((void*)(ref + (size_t)b * 3 * plane), 0, 3 * im.plane4, 0x00020000);
    im.tgt = __builtin_amdgcn_make_buffer_rsrc((void*)(tgt + (size_t)b * 3 * plane), 0, 3 * im.plane4, 0x00020000);
    im.dep = __builtin_amdgcn_make_buffer_rsrc((void*)(depth + (size_t)b * plane), 0, im.plane4, 0x00020000);
    im.dr = im.dep;
    return im;
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// Workgroups are dealt round-robin over the 8 XCDs (private L2s): give each XCD a CONTIGUOUS range of the
// logical work ids so that neighbouring tiles (shared halos, shared tap neighbourhoods) hit the same L2.
// Bijective for any n (cdna_hip_programming.md T1).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

struct Proj {
    float x, y;        // sample position in ref (pixels)
    float Xh, Yh;      // K^-1 [u v 1] (x, y components)
    float Px, Py, Pz;  // point in the reference camera
    bool valid;
};

// a3: back-project, rigid transform, project (oracle project(); reciprocal-multiply and FMA contraction
// instead of its divisions: <= 2 ulp on x, y, which the fp32 tolerances absorb)
__device__ __forceinline__ Proj project_px(const Geo& g, float d, int u, int v, int H, int W) {
    Proj o;
    o.Xh = ((float)u - g.cx) * g.ifx;
    o.Yh = ((float)v - g.cy) * g.ify;
    const float X = o.Xh * d, Y = o.Yh * d;
    o.Px = fmaf(g.r00, X, fmaf(g.r01, Y, fmaf(g.r02, d, g.tx)));
    o.Py = fmaf(g.r10, X, fmaf(g.r11, Y, fmaf(g.r12, d, g.ty)));
    o.Pz = fmaf(g.r20, X, fmaf(g.r21, Y, fmaf(g.r22, d, g.tz)));
    const bool front = o.Pz > Z_EPS;
    const float rz = front ? fast_rcp(o.Pz) : 1.0f;
    o.x = fmaf(g.fx * o.Px, rz, g.cx);
    o.y = fmaf(g.fy * o.Py, rz, g.cy);
    o.valid = front && (o.x >= 0.0f) && (o.x <= (float)(W - 1)) && (o.y >= 0.0f) && (o.y <= (float)(H - 1));
    return o;
}

struct Taps {
    int o00, o01, o10, o11;   // byte offsets inside a plane
    float wx, wy;
};

// only called for valid points: 0 <= x <= W-1, 0 <= y <= H-1
__device__ __forceinline__ Taps make_taps(const Proj& p, int H, int W) {
    Taps t;
    const float x0f = floorf(p.x), y0f = floorf(p.y);
    t.wx = p.x - x0f;
    t.wy = p.y - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int dx = (x0 + 1 < W) ? 4 : 0;
    const int dy = (y0 + 1 < H) ? 4 * W : 0;
    t.o00 = (y0 * W + x0) * 4; t.o01 = t.o00 + dx;
    t.o10 = t.o00 + dy; t.o11 = t.o10 + dx;
    return t;
}

// Branch-free taps: invalid points read tap (0,0) with zero weights (the result is masked).
__device__ __forceinline__ Taps make_taps_safe(const Proj& p, int H, int W) {
    Taps t;
    const float xs = p.valid ? p.x : 0.0f, ys = p.valid ? p.y : 0.0f;
    const float x0f = floorf(xs), y0f = floorf(ys);
    t.wx = xs - x0f;
    t.wy = ys - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int dx = (x0 + 1 < W) ? 4 : 0;
    const int dy = (y0 + 1 < H) ? 4 * W : 0;
    t.o00 = (y0 * W + x0) * 4; t.o01 = t.o00 + dx;
    t.o10 = t.o00 + dy; t.o11 = t.o10 + dx;
    return t;
}

// SSIM pieces from the five 3x3 window SUMS of one channel (x = target, y = J), everything scaled by
// 81 = 9^2 so the means never have to be formed:  S = (A1 A2) / (B1 B2) is scale-free.
struct SsimTerms {
    float sx, sy, A1, A2, B1, B2;
};
constexpr float C1_81 = 81.0f * SSIM_C1, C2_81 = 81.0f * SSIM_C2;
__device__ __forceinline__ SsimTerms ssim_terms(float sx, float sy, float sxx, float syy, float sxy) {
    SsimTerms s;
    s.sx = sx; s.sy = sy;
    const float pxy = sx * sy;
    const float q = fmaf(sx, sx, sy * sy);
    s.A1 = fmaf(2.0f, pxy, C1_81);                       // 81 (2 mx my + C1)
    s.A2 = fmaf(18.0f, sxy, C2_81) - 2.0f * pxy;         // 81 (2 cov + C2)
    s.B1 = q + C1_81;                                    // 81 (mx^2 + my^2 + C1)
    s.B2 = fmaf(9.0f, sxx + syy, C2_81) - q;             // 81 (vx + vy + C2)
    return s;
}

// --------------------------------------------------------------------------------------------- //
// forward: "marching wave"                                                                      //
// --------------------------------------------------------------------------------------------- //
// One WAVE owns a strip of 62 output columns x MROWS output rows and marches down it row by row; lane l
// holds column x0-1+l (lanes 0 and 63 are the halo columns, reflected at the image border).  Per row every
// lane evaluates its own pixel once; the 3-wide horizontal window sums come from the two neighbouring lanes
// by DPP wave shifts (one v_add_f32_dpp each), the 3-tall vertical sums from a rolling window of three rows
// of registers.  No LDS, no barriers, no redundant products; a 3-stage software pipeline keeps the
// depth/target loads two rows ahead and the 12 tap gathers one row ahead of the arithmetic.
constexpr int MCOLS = 62;
constexpr int MROWS_MAX = 64;

// Rows per segment: the grid should fill the chip's resident-wave slots (256 CUs x 16 waves at 4 waves/SIMD) in
// whole rounds -- a 1.4-round grid costs 2 rounds -- while keeping the 2 halo rows + pipeline prologue cheap.
inline int pick_march_rows(int B, int H, int W, int cols = MCOLS, int halo_rows = 2, int waves_per_cu = 16) {
    const long strips = (W + cols - 1) / cols;
    const long cap = 256L * waves_per_cu;
    int best = 32;
    double best_cost = 1e30;
    for (int r = 4; r <= MROWS_MAX; ++r) {
        const long waves = (long)B * ((H + r - 1) / r) * strips;
        const long rounds = (waves + cap - 1) / cap;
        const double cost = (double)rounds * (r + halo_rows + 2.0);      // steps per wave + ~2 steps of prologue / epilogue
        if (cost < best_cost - 1e-9) { best_cost = cost; best = r; }
    }
    return best;
}

__device__ __forceinline__ float dpp_from_left(float v) {    // lane i <- lane i-1 (0 into lane 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {   // lane i <- lane i+1 (0 into lane 63)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ float hsum3(float v) { return v + dpp_from_left(v) + dpp_from_right(v); }

// Horizontal 3-sums of N values in ONE asm statement: 2 N v_add_f32_dpp.  Written by hand because hipcc fused only a third
// of the `v + dpp(v) + dpp(v)` forms into DPP adds and emitted v_mov_b32_dpp + v_add_f32 pairs for the rest (34 of the
// ~600 instructions of a backward step).  The leading s_nop 1 covers "VALU write -> DPP read of that VGPR: 2 wait states"
// for all N sources at once (hipcc cannot see into the statement); the second add of each pair reads its DPP source again
// (old by then) and the partial sum through the plain operand.  r[i] = v[i](lane-1) + v[i] + v[i](lane+1), 0 beyond the wave.
#define HS_A_(r, v) "v_add_f32_dpp %" #r ", %" #v ", %" #v " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define HS_B_(r, v) "v_add_f32_dpp %" #r ", %" #v ", %" #r " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
__device__ __forceinline__ void hsum3x5(float v0, float v1, float v2, float v3, float v4, float& r0, float& r1, float& r2,
                                        float& r3, float& r4) {
    asm("s_nop 1\n\t" HS_A_(0, 5) HS_A_(1, 6) HS_A_(2, 7) HS_A_(3, 8) HS_A_(4, 9) HS_B_(0, 5) HS_B_(1, 6) HS_B_(2, 7) HS_B_(3, 8)
        HS_B_(4, 9)
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4)
        : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4));
}
__device__ __forceinline__ void hsum3x3(float v0, float v1, float v2, float& r0, float& r1, float& r2) {
    asm("s_nop 1\n\t" HS_A_(0, 3) HS_A_(1, 4) HS_A_(2, 5) HS_B_(0, 3) HS_B_(1, 4) HS_B_(2, 5)
        : "=&v"(r0), "=&v"(r1), "=&v"(r2)
        : "v"(v0), "v"(v1), "v"(v2));
}
#undef HS_A_
#undef HS_B_

struct MarchState {
    float dv[3], T[3][3];                 // stage A: depth + target of rows j, j+1, j+2 (slot = row % 3)
    float v[3][3][4], wx[3], wy[3], m[3]; // stage B: taps + weights + validity
    float h[3][15];                       // horizontal window sums of rows j-2, j-1, j
    float l1prev[3], mprev;               // |T - J| and ownership*validity of the previous row
    float acc, cnt;
};

template <int K>
__device__ __forceinline__ void march_step(MarchState& st, const Geo& g, const Img& im, int j, int nrows, int y_first,
                                           int px, bool own_col, int H, int W, float alpha) {
    constexpr int K1 = (K + 1) % 3, K2 = (K + 2) % 3;
    if (j >= nrows) return;                              // wave-uniform
    // (1) consume what the previous step left in flight: blend row j (taps) and project row j+1 (depth).
    //     Everything outstanding is needed here, so the compiler's vmcnt(0) at this point is exact.
    const float ux = 1.0f - st.wx[K], uy = 1.0f - st.wy[K];
    float Jc[3], Tc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float top = fmaf(st.v[K][c][1], st.wx[K], st.v[K][c][0] * ux);
        const float bot = fmaf(st.v[K][c][3], st.wx[K], st.v[K][c][2] * ux);
        Jc[c] = fmaf(g.a, st.m[K] * fmaf(bot, st.wy[K], top * uy), g.b);
        Tc[c] = st.T[K][c];
    }
    // (2) issue the loads of the next rows; they fly under the arithmetic of (3).  Rows past the end are clamped by
    //     reflect_idx (always legal), so no step-index branches
    {
        const int py = reflect_idx(y_first + j + 1, H);
        const Proj p = project_px(g, st.dv[K1], px, py, H, W);
        const Taps t = make_taps_safe(p, H, W);
        st.wx[K1] = t.wx; st.wy[K1] = t.wy; st.m[K1] = p.valid ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int so = c * im.plane4;
            st.v[K1][c][0] = bload(im.ref, t.o00, so); st.v[K1][c][1] = bload(im.ref, t.o01, so);
            st.v[K1][c][2] = bload(im.ref, t.o10, so); st.v[K1][c][3] = bload(im.ref, t.o11, so);
        }
    }
    {
        const int py = reflect_idx(y_first + j + 2, H);
        const int o4 = (py * W + px) * 4;
        st.dv[K2] = bload(im.dep, o4, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) st.T[K2][c] = bload(im.tgt, o4, c * im.plane4);
    }
    // (3) window sums of row j, SSIM of the row above
    float l1cur[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float J = Jc[c], T = Tc[c];
        hsum3x5(T, J, T * T, J * J, T * J, st.h[K][5 * c + 0], st.h[K][5 * c + 1], st.h[K][5 * c + 2], st.h[K][5 * c + 3],
                st.h[K][5 * c + 4]);
        l1cur[c] = fabsf(T - J);
    }
    {                                                    // output row = slot row j-1 (mprev = 0 until it exists)
        float mrow = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const SsimTerms s = ssim_terms(st.h[K1][5 * c + 0] + st.h[K2][5 * c + 0] + st.h[K][5 * c + 0],
                                           st.h[K1][5 * c + 1] + st.h[K2][5 * c + 1] + st.h[K][5 * c + 1],
                                           st.h[K1][5 * c + 2] + st.h[K2][5 * c + 2] + st.h[K][5 * c + 2],
                                           st.h[K1][5 * c + 3] + st.h[K2][5 * c + 3] + st.h[K][5 * c + 3],
                                           st.h[K1][5 * c + 4] + st.h[K2][5 * c + 4] + st.h[K][5 * c + 4]);
            const float S = (s.A1 * s.A2) * fast_rcp(s.B1 * s.B2);
            const float ss = fminf(fmaxf(0.5f * (1.0f - S), 0.0f), 1.0f);
            mrow += alpha * ss + (1.0f - alpha) * st.l1prev[c];
        }
        st.acc = fmaf(mrow, st.mprev, st.acc);
        st.cnt += st.mprev;
    }
    // this row becomes "the row above": it is an output row iff it is inside the image and inside the segment
    const int gy = y_first + j;
    const bool own_row = (j >= 1) && (j <= nrows - 2) && (gy < H);
    st.mprev = (own_row && own_col) ? st.m[K] : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) st.l1prev[c] = l1cur[c];
}

__global__ __launch_bounds__(NT) void k_warp_loss_fwd_march(
    const float* __restrict__ tgt, const float* __restrict__ ref, const float* __restrict__ depth,
    const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ lcc_a,
    const float* __restrict__ lcc_b, int B, int H, int W, int strips_x, int nseg, int seg_rows, float alpha,
    float* __restrict__ partials) {
    __shared__ float s_geo[4][GEO_N + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform BY CONSTRUCTION for the compiler (readfirstlane): everything derived from it -- image index,
    // buffer descriptors -- stays scalar, otherwise every buffer load is wrapped in a waterfall loop
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nitems = B * nseg * strips_x;
    const int item_raw = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // (image, segment, strip), strip fastest
    const bool live = item_raw < nitems;
    const int item = live ? item_raw : nitems - 1;
    const int b = item / (nseg * strips_x), rem = item - b * (nseg * strips_x);
    const int seg = rem / strips_x, strip = rem - seg * strips_x;
    if (lane == 0) geo_compute(pose, K, lcc_a, lcc_b, b, s_geo[wave]);
    __syncthreads();
    const Geo g = geo_load(s_geo[wave]);
    const Img im = img_make(tgt, ref, depth, b, H, W);

    const int x0 = strip * MCOLS, y0 = seg * seg_rows;
    const int gx = x0 - 1 + lane;
    const int px = reflect_idx(gx, W);
    const bool own_col = (lane >= 1) && (lane <= MCOLS) && (gx < W);
    const int rows_here = min(seg_rows, H - y0);
    const int nrows = rows_here + 2;
    const int y_first = y0 - 1;

    MarchState st;
    st.acc = 0.0f; st.cnt = 0.0f; st.mprev = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) st.l1prev[c] = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r)          // zero window sums give S = 1 (zero dissimilarity): the first two steps are
#pragma unroll                           // harmless without a branch on the step index
        for (int q = 0; q < 15; ++q) st.h[r][q] = 0.0f;
    // prologue: A(0), A(1), B(0)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int py = reflect_idx(y_first + r, H);
        const int o4 = (py * W + px) * 4;
        st.dv[r] = bload(im.dep, o4, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) st.T[r][c] = bload(im.tgt, o4, c * im.plane4);
    }
    {
        const int py = reflect_idx(y_first, H);
        const Proj p = project_px(g, st.dv[0], px, py, H, W);
        const Taps t = make_taps_safe(p, H, W);
        st.wx[0] = t.wx; st.wy[0] = t.wy; st.m[0] = p.valid ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int so = c * im.plane4;
            st.v[0][c][0] = bload(im.ref, t.o00, so); st.v[0][c][1] = bload(im.ref, t.o01, so);
            st.v[0][c][2] = bload(im.ref, t.o10, so); st.v[0][c][3] = bload(im.ref, t.o11, so);
        }
    }
#pragma unroll 1
    for (int j = 0; j < nrows; j += 3) {
        march_step<0>(st, g, im, j, nrows, y_first, px, own_col, H, W, alpha);
        march_step<1>(st, g, im, j + 1, nrows, y_first, px, own_col, H, W, alpha);
        march_step<2>(st, g, im, j + 2, nrows, y_first, px, own_col, H, W, alpha);
    }
    float acc = live ? st.acc : 0.0f, cnt = live ? st.cnt : 0.0f;
    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    if (lane == 0 && live) {
        partials[2 * (size_t)item] = acc;
        partials[2 * (size_t)item + 1] = cnt;
    }
}

// deterministic two-stage reduction: fixed strided order, then an LDS tree
__global__ __launch_bounds__(NT) void k_warp_loss_fwd_finalize(const float* __restrict__ partials, int nblk,
                                                               float* __restrict__ loss_state) {
    __shared__ float s0[NT], s1[NT];
    float a = 0.0f, c = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += NT) { a += partials[2 * i]; c += partials[2 * i + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = c;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float denom = fmaxf(3.0f * s1[0], 1.0f);
        loss_state[0] = s0[0] / denom;
        loss_state[1] = 1.0f / denom;
        loss_state[2] = s1[0];
        loss_state[3] = s0[0];          // the masked sum itself: what a data-parallel caller adds up across ranks (colvo_warp_loss_rescale)
    }
}

constexpr int NPART = 14;                   // dt[3], dR[9], da, db
constexpr int NPART_F = 16;                 // fused pass: + loss sum, valid-pixel count

// --------------------------------------------------------------------------------------------- //
// backward: "marching wave"                                                                      //
// --------------------------------------------------------------------------------------------- //
// Same organisation as the forward: one wave per strip (60 output columns, lanes 2..61; two halo lanes per side)
// marching down `seg_rows` output rows (+2 halo rows per side).  Three rolling 3-row register windows carry
//   E  : the evaluated sample of a row (warp, its x/y derivatives, the projected point, depth, validity),
//   H  : the horizontal 3-sums of {T, J, T^2, J^2, TJ} per channel (DPP wave shifts),
//   HK : the horizontally gathered derivative coefficients of the SSIM windows (reflection multiplicities),
// so that at step j:  row j is evaluated, window row j-1 gets its coefficients, pixel row j-2 gets dJ and is
// chained through LCC / bilinear taps / projection.  Every pixel is evaluated ONCE (the tile version needed
// 2.3 evaluations), every window's coefficients are computed once, there is no LDS and no barrier.
constexpr int BCOLS = 60;

struct BwdState {
    // in flight / just landed (slot = row % 3)
    float dv[3], T[3][3];
    float v[3][3][4], wx[3], wy[3];
    // evaluated rows (T and depth are copied here when a row is consumed: their load slots are refilled one step
    // later, but the pixel row needs them two steps later)
    float Wp[3][3], gx[3][3], gy[3][3], Tk[3][3], dk[3], Px[3], Py[3], Pz[3], m[3];
    float H[3][15];
    float HK[3][9];
    float part[NPART];
    float lacc, lcnt;                       // fused pass only: loss sum and valid count of the pixels this wave owns
    float gacc;                             // geometric-consistency variant only: sum of the term over the owned pixels
};

// FUSED: the same pass also accumulates the loss itself (alpha, 1 - alpha in al / l1w), for the one-kernel
// loss + unnormalised-gradient forward of the training path (colvo_warp_loss_fused).
// EDGE: the strip touches the left / right image border (or hangs over it), so the horizontal gather of the window
// coefficients needs the reflection multiplicities wxw[]; interior strips (9 of 11 at W = 640) take plain 3-sums.
// GEO (with FUSED): the geometric-consistency term |D_proj - D_samp| / (D_proj + D_samp) of the same projection rides along
// (spec: geometric_consistency_loss; /root/reference/README.md:1, :7): the reference frame's own depth is a fourth sampled
// plane on the taps the warp uses anyway; the term's value is summed beside the photometric one, its gradient w.r.t. the
// projected depth and the sample position runs through the same chain rule (weight `grho` relative to the photometric
// gradient, both still unnormalised: the two terms share their valid-pixel count), and its gradient w.r.t. the sampled
// taps is scattered into `dr_acc` as 64-bit FIXED-POINT atomics (2^-32 units): integer addition is associative, so the
// scatter is bit-reproducible whatever order the waves arrive in.
constexpr float GEO_FIX = 4294967296.0f;     // 2^32
// q * 2^32 as a two's-complement 64-bit integer (|q| < 2^31; truncated below 2^-32).  floor(q) is exact in fp32 and so is the
// fraction q - floor(q) -- except for a tiny negative q (-1e-10: floor -1, fraction ROUNDS to 1.0f), whose fraction times 2^32
// would be out of range for the float -> unsigned conversion (undefined in C++; v_cvt_u32_f32 happens to saturate): such a q is
// carried as the integer above it with fraction 0, i.e. rounded to zero, an error below 2^-24 of one unit of the integer part.
__device__ __forceinline__ unsigned long long to_fix32(float q) {
    float h = floorf(q);
    float f = q - h;
    if (f >= 1.0f) { h += 1.0f; f = 0.0f; }
    const unsigned lo = (unsigned)(f * GEO_FIX);            // f in [0, 1 - 2^-24]: f * 2^32 <= 2^32 - 256
    return ((unsigned long long)(unsigned)(int)h << 32) | lo;
}

template <int K, bool FUSED, bool EDGE, bool GEO = false>
__device__ __forceinline__ void bwd_step(BwdState& st, const Geo& g, const Img& im, int j, int nrows, int y_first,
                                         int gxcol, int px, bool own_col, float Xh, const float (&wxw)[3], int H,
                                         int W, float kss, float kl1, float* __restrict__ d_depth_img, float al, float l1w,
                                         float grho = 0.0f, unsigned long long* __restrict__ dr_acc = nullptr,
                                         float* __restrict__ geo_img = nullptr) {
    constexpr int K1 = (K + 1) % 3, K2 = (K + 2) % 3;
    if (j >= nrows) return;                              // wave-uniform
    // (1) consume the taps of row j: warp, its spatial derivatives, LCC
    float Jc[3], Tc[3];
    {
        const float ux = 1.0f - st.wx[K], uy = 1.0f - st.wy[K], m = st.m[K];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float i00 = st.v[K][c][0], i01 = st.v[K][c][1], i10 = st.v[K][c][2], i11 = st.v[K][c][3];
            const float top = fmaf(i01, st.wx[K], i00 * ux);
            const float bot = fmaf(i11, st.wx[K], i10 * ux);
            const float w = m * fmaf(bot, st.wy[K], top * uy);
            st.Wp[K][c] = w;
            st.gx[K][c] = m * fmaf(st.wy[K], i11 - i10, uy * (i01 - i00));
            st.gy[K][c] = m * fmaf(st.wx[K], i11 - i01, ux * (i10 - i00));
            Jc[c] = fmaf(g.a, w, g.b);
            Tc[c] = st.T[K][c];
            st.Tk[K][c] = Tc[c];
        }
        st.dk[K] = st.dv[K];
    }
    // rows j-2 and j+1 share a slot: snapshot what the chain rule of pixel row j-2 needs before (2) overwrites it
    const float mp = st.m[K1], Pxv = st.Px[K1], Pyv = st.Py[K1], Pzv = st.Pz[K1];
    // (2) issue the loads of the next rows (they fly under the arithmetic below).  Rows past the end of the strip are
    //     clamped by reflect_idx, so the loads are always legal and no step-index branch is needed.
    float vd[4];                                         // GEO: the reference depth on the taps of row j+1, consumed in (6)
    int to00 = 0, to01 = 0, to10 = 0;
    {
        const int py = reflect_idx(y_first + j + 1, H);
        const Proj p = project_px(g, st.dv[K1], px, py, H, W);
        const Taps t = make_taps_safe(p, H, W);
        st.wx[K1] = t.wx; st.wy[K1] = t.wy; st.m[K1] = p.valid ? 1.0f : 0.0f;
        st.Px[K1] = p.Px; st.Py[K1] = p.Py; st.Pz[K1] = p.valid ? p.Pz : 1.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int so = c * im.plane4;
            st.v[K1][c][0] = bload(im.ref, t.o00, so); st.v[K1][c][1] = bload(im.ref, t.o01, so);
            st.v[K1][c][2] = bload(im.ref, t.o10, so); st.v[K1][c][3] = bload(im.ref, t.o11, so);
        }
        if constexpr (GEO) {
            vd[0] = bload(im.dr, t.o00, 0); vd[1] = bload(im.dr, t.o01, 0);
            vd[2] = bload(im.dr, t.o10, 0); vd[3] = bload(im.dr, t.o11, 0);
            to00 = t.o00; to01 = t.o01 - t.o00; to10 = t.o10 - t.o00;
        }
    }
    {
        const int py = reflect_idx(y_first + j + 2, H);
        const int o4 = (py * W + px) * 4;
        st.dv[K2] = bload(im.dep, o4, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) st.T[K2][c] = bload(im.tgt, o4, c * im.plane4);
    }
    // (3) horizontal window sums of row j
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float J = Jc[c], T = Tc[c];
        hsum3x5(T, J, T * T, J * J, T * J, st.H[K][5 * c + 0], st.H[K][5 * c + 1], st.H[K][5 * c + 2], st.H[K][5 * c + 3],
                st.H[K][5 * c + 4]);
    }
    // (4) derivative coefficients of window row j-1 (centre = row j-1, this lane's column), gathered horizontally.
    //     No branch on j: before two rows have been seen the zero-initialised H gives S = 1, i.e. zero coefficients,
    //     and the window row does not exist (wmask = 0).
    {
        const int gyw = y_first + j - 1;
        const bool wexists = (gyw >= 0) && (gyw < H) && (gxcol >= 0) && (gxcol < W) && (j >= 2);
        const float wmask = wexists ? st.m[K2] : 0.0f;   // row j-1 = slot (j+2) % 3
        const float wk2 = wmask * (2.0f * kss);
        float lrow = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const SsimTerms s = ssim_terms(st.H[K1][5 * c + 0] + st.H[K2][5 * c + 0] + st.H[K][5 * c + 0],
                                           st.H[K1][5 * c + 1] + st.H[K2][5 * c + 1] + st.H[K][5 * c + 1],
                                           st.H[K1][5 * c + 2] + st.H[K2][5 * c + 2] + st.H[K][5 * c + 2],
                                           st.H[K1][5 * c + 3] + st.H[K2][5 * c + 3] + st.H[K][5 * c + 3],
                                           st.H[K1][5 * c + 4] + st.H[K2][5 * c + 4] + st.H[K][5 * c + 4]);
            const float inv = nr_rcp(s.B1 * s.B2);
            const float S = s.A1 * s.A2 * inv;
            const float ss = 0.5f * (1.0f - S);
            // d ss / d{sx.., syy, sxy} of the UNclamped branch, times the window's weight: with t2 = 2 kss wmask / (B1 B2),
            //   A = t2 (sx (A2 - A1) - S sy (B2 - B1)),  B = -9 t2 S B1,  C = 9 t2 A1     (13 instructions per channel)
            const float t2 = (ss > 0.0f && ss < 1.0f) ? wk2 * inv : 0.0f;
            const float A = t2 * fmaf(s.sx, s.A2 - s.A1, -(S * s.sy) * (s.B2 - s.B1));
            const float Bc = (t2 * S) * (-9.0f * s.B1);
            const float Cc = t2 * (9.0f * s.A1);
            if constexpr (EDGE) {
                st.HK[K2][3 * c + 0] = fmaf(wxw[0], dpp_from_left(A), fmaf(wxw[2], dpp_from_right(A), wxw[1] * A));
                st.HK[K2][3 * c + 1] = fmaf(wxw[0], dpp_from_left(Bc), fmaf(wxw[2], dpp_from_right(Bc), wxw[1] * Bc));
                st.HK[K2][3 * c + 2] = fmaf(wxw[0], dpp_from_left(Cc), fmaf(wxw[2], dpp_from_right(Cc), wxw[1] * Cc));
            } else {
                hsum3x3(A, Bc, Cc, st.HK[K2][3 * c + 0], st.HK[K2][3 * c + 1], st.HK[K2][3 * c + 2]);
            }
            if constexpr (FUSED)
                lrow += al * fminf(fmaxf(ss, 0.0f), 1.0f) + l1w * fabsf(st.Tk[K2][c] - fmaf(g.a, st.Wp[K2][c], g.b));
        }
        if constexpr (FUSED) {                           // row j-1 is an output row of this wave for row indices 2 .. nrows-3
            const bool own1 = own_col && (j - 1 >= 2) && (j - 1 <= nrows - 3) && (gyw < H);
            const float mo = own1 ? wmask : 0.0f;
            st.lacc = fmaf(lrow, mo, st.lacc);
            st.lcnt += mo;
        }
    }
    // (5) pixel row j-2 (slot K1): vertical gather of the coefficient rows j-3, j-2, j-1 -> dJ, then the chain rule.
    //     Branch-free: pixels this wave does not own contribute through a 0/1 factor, only the store is predicated.
    {
        const int gyp = y_first + j - 2;
        float wyw[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int q = gyp + d - 1;
            wyw[d] = (gyp >= 0 && q >= 0 && q < H) ? window_mult(q, gyp, H) : 0.0f;
        }
        const bool own = own_col && (j - 2 >= 2) && (j - 2 <= nrows - 3) && (gyp < H);
        const float ownf = own ? 1.0f : 0.0f;
        // slots: row j-3 -> K, row j-2 -> K1, row j-1 -> K2   (j-3 = j mod 3)
        const float dprev2 = st.dk[K1];
        float da = 0.f, db = 0.f, gxs = 0.f, gys = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float ha = wyw[0] * st.HK[K][3 * c + 0] + wyw[1] * st.HK[K1][3 * c + 0] + wyw[2] * st.HK[K2][3 * c + 0];
            const float hb = wyw[0] * st.HK[K][3 * c + 1] + wyw[1] * st.HK[K1][3 * c + 1] + wyw[2] * st.HK[K2][3 * c + 1];
            const float hc = wyw[0] * st.HK[K][3 * c + 2] + wyw[1] * st.HK[K1][3 * c + 2] + wyw[2] * st.HK[K2][3 * c + 2];
            const float wv = st.Wp[K1][c];
            const float Jp = fmaf(g.a, wv, g.b), Tp = st.Tk[K1][c];
            const float diff = Jp - Tp;
            const float sgn = (diff > 0.0f) ? 1.0f : ((diff < 0.0f) ? -1.0f : 0.0f);
            const float dJ = ownf * (fmaf(hb, Jp, fmaf(hc, Tp, ha)) + kl1 * mp * sgn);
            da = fmaf(dJ, wv, da);
            db += dJ;
            const float dW = g.a * dJ;
            gxs = fmaf(dW, st.gx[K1][c], gxs);       // gx, gy are already zero where the sample is invalid
            gys = fmaf(dW, st.gy[K1][c], gys);
        }
        st.part[12] += da;
        st.part[13] += db;
        const float iz = nr_rcp(Pzv);                // Pz was stored as 1 for invalid samples
        const float dPx = gxs * g.fx * iz;
        const float dPy = gys * g.fy * iz;
        const float dPz = -(dPx * Pxv + dPy * Pyv) * iz;
        const float Yh = ((float)gyp - g.cy) * g.ify;
        const float rx_ = g.r00 * Xh + g.r01 * Yh + g.r02;
        const float ry_ = g.r10 * Xh + g.r11 * Yh + g.r12;
        const float rz_ = g.r20 * Xh + g.r21 * Yh + g.r22;
        const float dd = dPx * rx_ + dPy * ry_ + dPz * rz_;
        const float cX = Xh * dprev2, cY = Yh * dprev2, cZ = dprev2;
        st.part[0] += dPx; st.part[1] += dPy; st.part[2] += dPz;
        st.part[3] += dPx * cX; st.part[4] += dPx * cY; st.part[5] += dPx * cZ;
        st.part[6] += dPy * cX; st.part[7] += dPy * cY; st.part[8] += dPy * cZ;
        st.part[9] += dPz * cX; st.part[10] += dPz * cY; st.part[11] += dPz * cZ;
        if (own) d_depth_img[(size_t)gyp * W + gxcol] = dd;
    }
    // (6) GEO: the geometric-consistency term of pixel row j+1 -- the row projected in (2); its reference-depth taps have
    //     been in flight under (3)-(5) and the next step waits for the image taps issued before them anyway.  Pixel-local, so
    //     value, chain rule and scatter happen here at once and nothing is carried between steps: the term's depth gradient
    //     goes to a plane of its own (geo_img; the caller adds the two planes under their common normaliser).
    if constexpr (GEO) {
        const int r = j + 1, gyr = y_first + r;
        const bool own = own_col && (r >= 2) && (r <= nrows - 3) && (gyr < H);
        const float m = st.m[K1], mo = own ? m : 0.0f;
        const float wx = st.wx[K1], wy = st.wy[K1], ux = 1.0f - wx, uy = 1.0f - wy;
        const float bs = fmaf(fmaf(vd[3], wx, vd[2] * ux), wy, fmaf(vd[1], wx, vd[0] * ux) * uy);     // D_samp
        const float a = st.Pz[K1];                                                                    // D_proj (1 where invalid)
        const float inv = nr_rcp((m > 0.0f) ? a + bs : 1.0f);
        const float df = a - bs;
        st.gacc = fmaf(mo * fabsf(df), inv, st.gacc);
        const float sg = (df > 0.0f) ? 1.0f : ((df < 0.0f) ? -1.0f : 0.0f);
        const float k2 = mo * grho * sg * 2.0f * inv * inv;
        const float gb = -k2 * a, ga = k2 * bs;          // d / d D_samp, d / d D_proj
        const float gxd = gb * fmaf(wy, vd[3] - vd[2], uy * (vd[1] - vd[0]));
        const float gyd = gb * fmaf(wx, vd[3] - vd[1], ux * (vd[2] - vd[0]));
        const float iz = nr_rcp(a);
        const float dPx = gxd * g.fx * iz;
        const float dPy = gyd * g.fy * iz;
        const float dPz = ga - (dPx * st.Px[K1] + dPy * st.Py[K1]) * iz;
        const float Yh = ((float)gyr - g.cy) * g.ify;
        const float rx_ = g.r00 * Xh + g.r01 * Yh + g.r02;
        const float ry_ = g.r10 * Xh + g.r11 * Yh + g.r12;
        const float rz_ = g.r20 * Xh + g.r21 * Yh + g.r22;
        const float dcur = st.dv[K1];
        const float cX = Xh * dcur, cY = Yh * dcur, cZ = dcur;
        st.part[0] += dPx; st.part[1] += dPy; st.part[2] += dPz;
        st.part[3] += dPx * cX; st.part[4] += dPx * cY; st.part[5] += dPx * cZ;
        st.part[6] += dPy * cX; st.part[7] += dPy * cY; st.part[8] += dPy * cZ;
        st.part[9] += dPz * cX; st.part[10] += dPz * cY; st.part[11] += dPz * cZ;
        if (own) {
            geo_img[(size_t)gyr * W + gxcol] = dPx * rx_ + dPy * ry_ + dPz * rz_;
            if (m > 0.0f) {
                char* q = reinterpret_cast<char*>(dr_acc) + 2 * (size_t)to00;
                atomicAdd(reinterpret_cast<unsigned long long*>(q), to_fix32(gb * ux * uy));
                atomicAdd(reinterpret_cast<unsigned long long*>(q + 2 * to01), to_fix32(gb * wx * uy));
                atomicAdd(reinterpret_cast<unsigned long long*>(q + 2 * to10), to_fix32(gb * ux * wy));
                atomicAdd(reinterpret_cast<unsigned long long*>(q + 2 * to10 + 2 * to01), to_fix32(gb * wx * wy));
            }
        }
    }
}

// level: pyramid level of the images this launch works on (the intrinsics are scaled in the kernel); GEO: see bwd_step
// (depth_r: the reference frame's depth, dr_acc: [B,H,W] 64-bit fixed-point accumulators zeroed by the caller, geo_partials:
// one float per strip segment, grho: weight of the term's gradient relative to the photometric one)
struct GeoArgs {
    const float* depth_r;
    unsigned long long* dr_acc;
    float* geo_partials;
    float* d_depth_geo;          // [B,H,W]: the term's (unnormalised, grho-weighted) gradient w.r.t. the target depth
    float grho;
};

// one wave's strip segment: everything after the per-image geometry has been put into LDS (sgeo) by the caller
template <bool FUSED, bool GEO>
__device__ __forceinline__ void march_body(const float* __restrict__ tgt, const float* __restrict__ ref,
                                           const float* __restrict__ depth, int H, int W, int seg_rows, float alpha,
                                           const float* __restrict__ loss_state, const float* __restrict__ grad_loss,
                                           float* __restrict__ d_depth, float* __restrict__ partials, const GeoArgs& ga,
                                           int item, bool live, int b, int seg, int strip, int lane, const float* sgeo) {
    static_assert(FUSED || !GEO, "the geometric-consistency term rides on the one-pass form only");
    const Geo g = geo_load(sgeo);
    Img im = img_make(tgt, ref, depth, b, H, W);
    unsigned long long* dr_acc = nullptr;
    float* geo_img = nullptr;
    if constexpr (GEO) {
        im.dr = __builtin_amdgcn_make_buffer_rsrc((void*)(ga.depth_r + (size_t)b * H * W), 0, im.plane4, 0x00020000);
        dr_acc = ga.dr_acc + (size_t)b * H * W;
        geo_img = ga.d_depth_geo + (size_t)b * H * W;
    }
    const float grho = GEO ? ga.grho : 0.0f;
    // fused pass: unnormalised gradients (scaled by dL/dloss / max(3 n_valid, 1) once that is known)
    const float gscale = FUSED ? 1.0f : grad_loss[0] * loss_state[1];   // dL/dloss / max(3 n_valid, 1)
    const float kss = gscale * alpha * (-0.5f);
    const float kl1 = gscale * (1.0f - alpha);

    const int x0 = strip * BCOLS, y0 = seg * seg_rows;
    const int gxcol = x0 - 2 + lane;
    const int px = reflect_idx(gxcol, W);
    const bool own_col = live && (lane >= 2) && (lane < 2 + BCOLS) && (gxcol < W);
    const int rows_here = min(seg_rows, H - y0);
    const int nrows = rows_here + 4;
    const int y_first = y0 - 2;
    const float Xh = ((float)gxcol - g.cx) * g.ifx;
    float wxw[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int q = gxcol + d - 1;
        wxw[d] = (gxcol >= 0 && gxcol < W && q >= 0 && q < W) ? window_mult(q, gxcol, W) : 0.0f;
    }

    BwdState st;
    st.lacc = 0.0f; st.lcnt = 0.0f; st.gacc = 0.0f;
#pragma unroll
    for (int k = 0; k < NPART; ++k) st.part[k] = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        st.m[r] = 0.0f; st.Px[r] = 0.f; st.Py[r] = 0.f; st.Pz[r] = 1.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) { st.Wp[r][c] = 0.f; st.gx[r][c] = 0.f; st.gy[r][c] = 0.f; st.Tk[r][c] = 0.f; }
        st.dk[r] = 1.0f;
#pragma unroll
        for (int q = 0; q < 9; ++q) st.HK[r][q] = 0.0f;
#pragma unroll
        for (int q = 0; q < 15; ++q) st.H[r][q] = 0.0f;
    }
    // prologue: depth/target of rows 0 and 1, projection + taps of row 0
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int py = reflect_idx(y_first + r, H);
        const int o4 = (py * W + px) * 4;
        st.dv[r] = bload(im.dep, o4, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) st.T[r][c] = bload(im.tgt, o4, c * im.plane4);
    }
    st.dv[2] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) st.T[2][c] = 0.0f;
    {
        const int py = reflect_idx(y_first, H);
        const Proj p = project_px(g, st.dv[0], px, py, H, W);
        const Taps t = make_taps_safe(p, H, W);
        st.wx[0] = t.wx; st.wy[0] = t.wy; st.m[0] = p.valid ? 1.0f : 0.0f;
        st.Px[0] = p.Px; st.Py[0] = p.Py; st.Pz[0] = p.Pz;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int so = c * im.plane4;
            st.v[0][c][0] = bload(im.ref, t.o00, so); st.v[0][c][1] = bload(im.ref, t.o01, so);
            st.v[0][c][2] = bload(im.ref, t.o10, so); st.v[0][c][3] = bload(im.ref, t.o11, so);
        }
    }
    float* ddimg = d_depth + (size_t)b * H * W;
    // interior strip: every lane's column and both its neighbours are inside the image (all multiplicities are 1)
    const bool edge = !(x0 >= 3 && x0 + 61 <= W - 2);          // wave-uniform
    if (edge) {
#pragma unroll 1
        for (int j = 0; j < nrows; j += 3) {
            bwd_step<0, FUSED, true, GEO>(st, g, im, j, nrows, y_first, gxcol, px, own_col, Xh, wxw, H, W, kss, kl1, ddimg, alpha, 1.0f - alpha, grho, dr_acc, geo_img);
            bwd_step<1, FUSED, true, GEO>(st, g, im, j + 1, nrows, y_first, gxcol, px, own_col, Xh, wxw, H, W, kss, kl1, ddimg, alpha, 1.0f - alpha, grho, dr_acc, geo_img);
            bwd_step<2, FUSED, true, GEO>(st, g, im, j + 2, nrows, y_first, gxcol, px, own_col, Xh, wxw, H, W, kss, kl1, ddimg, alpha, 1.0f - alpha, grho, dr_acc, geo_img);
        }
    } else {
#pragma unroll 1
        for (int j = 0; j < nrows; j += 3) {
            bwd_step<0, FUSED, false, GEO>(st, g, im, j, nrows, y_first, gxcol, px, own_col, Xh, wxw, H, W, kss, kl1, ddimg, alpha, 1.0f - alpha, grho, dr_acc, geo_img);
            bwd_step<1, FUSED, false, GEO>(st, g, im, j + 1, nrows, y_first, gxcol, px, own_col, Xh, wxw, H, W, kss, kl1, ddimg, alpha, 1.0f - alpha, grho, dr_acc, geo_img);
            bwd_step<2, FUSED, false, GEO>(st, g, im, j + 2, nrows, y_first, gxcol, px, own_col, Xh, wxw, H, W, kss, kl1, ddimg, alpha, 1.0f - alpha, grho, dr_acc, geo_img);
        }
    }
    constexpr int NP = FUSED ? NPART_F : NPART;
#pragma unroll
    for (int k = 0; k < NPART; ++k) {
        const float v = wave_sum(live ? st.part[k] : 0.0f);
        if (lane == 0 && live) partials[(size_t)item * NP + k] = v;
    }
    if constexpr (FUSED) {
        const float la = wave_sum(live ? st.lacc : 0.0f), lc = wave_sum(live ? st.lcnt : 0.0f);
        if (lane == 0 && live) { partials[(size_t)item * NP + 14] = la; partials[(size_t)item * NP + 15] = lc; }
    }
    if constexpr (GEO) {
        const float gs = wave_sum(live ? st.gacc : 0.0f);
        if (lane == 0 && live) ga.geo_partials[item] = gs;
    }
}

template <bool FUSED, bool GEO = false>
__global__ __launch_bounds__(NT, 2) void k_warp_loss_bwd_march(
    const float* __restrict__ tgt, const float* __restrict__ ref, const float* __restrict__ depth,
    const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ lcc_a,
    const float* __restrict__ lcc_b, int B, int H, int W, int strips_x, int nseg, int seg_rows, float alpha,
    const float* __restrict__ loss_state, const float* __restrict__ grad_loss, float* __restrict__ d_depth,
    float* __restrict__ partials, int level, GeoArgs ga) {
    __shared__ float s_geo[4][GEO_N + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: keeps the buffer descriptors uniform
    const int nitems = B * nseg * strips_x;
    const int item_raw = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // (image, segment, strip), strip fastest
    const bool live = item_raw < nitems;
    const int item = live ? item_raw : nitems - 1;
    const int b = item / (nseg * strips_x), rem = item - b * (nseg * strips_x);
    const int seg = rem / strips_x, strip = rem - seg * strips_x;
    if (lane == 0) geo_compute(pose, K, lcc_a, lcc_b, b, s_geo[wave], level);
    __syncthreads();
    march_body<FUSED, GEO>(tgt, ref, depth, H, W, seg_rows, alpha, loss_state, grad_loss, d_depth, partials, ga, item, live, b, seg,
                           strip, lane, s_geo[wave]);
}

// The one-pass kernel over SEVERAL pyramid levels in one launch (the widened objective): the strip segments of level 0 come
// first, the small levels fill the tail of the launch instead of paying a launch + drain each.  Every level's workgroup count
// is padded to a multiple of 8 so that each XCD (workgroups are dealt round-robin) gets a contiguous eighth of every level.
constexpr int MARCH_MAX_LEVELS = 4;
struct MarchLevel {
    const float *tgt, *ref, *depth;
    float *d_depth, *partials;
    int H, W, strips_x, nseg, seg_rows;
    int wg8;                     // workgroups of this level per XCD
};
struct MarchLevels {
    int S;
    MarchLevel lv[MARCH_MAX_LEVELS];
};

template <bool GEO>
__global__ __launch_bounds__(NT, 2) void k_warp_loss_march_levels(MarchLevels ml, const float* __restrict__ pose,
                                                                  const float* __restrict__ K, const float* __restrict__ lcc_a,
                                                                  const float* __restrict__ lcc_b, int B, float alpha, GeoArgs ga) {
    __shared__ float s_geo[4][GEO_N + 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7;
    int k = blockIdx.x >> 3, l = 0;
    // the level this workgroup belongs to and its fields, by scalar selects (no dynamically indexed copy of the argument)
    MarchLevel L = ml.lv[0];
#pragma unroll
    for (int i = 1; i < MARCH_MAX_LEVELS; ++i) {
        if (i < ml.S && l == i - 1 && k >= L.wg8) { k -= L.wg8; l = i; L = ml.lv[i]; }
    }
    const int nitems = B * L.nseg * L.strips_x;
    const int item_raw = (xcd * L.wg8 + k) * 4 + wave;          // (image, segment, strip), strip fastest
    const bool live = item_raw < nitems;
    const int item = live ? item_raw : nitems - 1;
    const int b = item / (L.nseg * L.strips_x), rem = item - b * (L.nseg * L.strips_x);
    const int seg = rem / L.strips_x, strip = rem - seg * L.strips_x;
    if (lane == 0) geo_compute(pose, K, lcc_a, lcc_b, b, s_geo[wave], l);
    __syncthreads();
    if (GEO && l == 0)
        march_body<true, GEO>(L.tgt, L.ref, L.depth, L.H, L.W, L.seg_rows, alpha, nullptr, nullptr, L.d_depth, L.partials, ga, item,
                              live, b, seg, strip, lane, s_geo[wave]);
    else
        march_body<true, false>(L.tgt, L.ref, L.depth, L.H, L.W, L.seg_rows, alpha, nullptr, nullptr, L.d_depth, L.partials, ga,
                                item, live, b, seg, strip, lane, s_geo[wave]);
}

// fused forward, second kernel: blocks 0..B-1 fold the 14 gradient sums of one image (still unnormalised) into
// gpart[b][14]; block B folds the loss sum and the valid count of ALL strips into loss_state.  Fixed orders: deterministic.
// dR (row-major 3x3 gradient w.r.t. the rotation matrix) -> gradient w.r.t. the Euler angles (R = Rz Ry Rx)
__device__ __forceinline__ void dR_to_euler(const float* dR, const float* p, float& drx, float& dry, float& drz) {
    const float sx = sinf(p[3]), cx = cosf(p[3]), sy = sinf(p[4]), cy = cosf(p[4]), sz = sinf(p[5]), cz = cosf(p[5]);
    drx = dR[1] * (cz * sy * cx + sz * sx) + dR[2] * (-cz * sy * sx + sz * cx)
        + dR[4] * (sz * sy * cx - cz * sx) + dR[5] * (-sz * sy * sx - cz * cx)
        + dR[7] * (cy * cx) + dR[8] * (-cy * sx);
    dry = dR[0] * (-cz * sy) + dR[1] * (cz * cy * sx) + dR[2] * (cz * cy * cx)
        + dR[3] * (-sz * sy) + dR[4] * (sz * cy * sx) + dR[5] * (sz * cy * cx)
        + dR[6] * (-cy) + dR[7] * (-sy * sx) + dR[8] * (-sy * cx);
    drz = dR[0] * (-sz * cy) + dR[1] * (-sz * sy * sx - cz * cx) + dR[2] * (-sz * sy * cx + cz * sx)
        + dR[3] * (cz * cy) + dR[4] * (cz * sy * sx - sz * cx) + dR[5] * (cz * sy * cx + sz * sx);
}

// The finalize kernels run 1024 threads per workgroup: they are chains of dependent strided loads (one partial per strip
// segment), so their duration is the number of sequential trips to L2 -- 7 -> 2 for the loss sums of BASELINE configs[1], 30 -> 8
// at configs[2] -- not bandwidth.
constexpr int FT = 1024;
constexpr int FROWS = FT / NPART;           // 73 partial rows per pass

// sums of NV per-thread values over the FT threads: DPP inside the wave, then thread 0 adds the 16 wave totals in order
template <int NV>
__device__ __forceinline__ void block_sums(float (&v)[NV], float* sh) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const float t = wave_sum(v[q]);
        if ((tid & 63) == 0) sh[(tid >> 6) * NV + q] = t;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            float t = 0.0f;
            for (int w = 0; w < FT / 64; ++w) t += sh[w * NV + q];
            v[q] = t;
        }
    }
}

// the 14 gradient sums of image b over its strip segments (partials of NPART_F floats each), fixed order; the result is
// valid in threads 0..13 (and left in sh[0..13] after a barrier when `publish`)
__device__ __forceinline__ float image_sums(const float* __restrict__ pp, int b, int bpi, float* sh) {
    const int tid = threadIdx.x, k = tid % NPART, r = tid / NPART;
    float acc = 0.0f;
    if (r < FROWS) {
        for (int i = r; i < bpi; i += FROWS) acc += pp[((size_t)b * bpi + i) * NPART_F + k];
        sh[k * FROWS + r] = acc;
    }
    __syncthreads();
    float t = 0.0f;
    if (tid < NPART)
        for (int i = 0; i < FROWS; ++i) t += sh[tid * FROWS + i];
    return t;
}

// gunit (may be null): the pose / LCC gradients of every image, still UNNORMALISED, already converted to Euler angles, in
// PoseNet's planar output layout [d_pose B x 6 | d_a B | d_b B] -- for consumers that apply dL/dloss / max(3 n, 1) themselves
__global__ __launch_bounds__(FT) void k_warp_loss_fused_finalize(const float* __restrict__ partials, int blocks_per_image,
                                                                 int B, const float* __restrict__ pose,
                                                                 float* __restrict__ gpart, float* __restrict__ gunit,
                                                                 float* __restrict__ loss_state) {
    __shared__ float sh[FT];
    __shared__ float s1[NPART + 2];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x == B) {
        const int n = B * blocks_per_image;
        float v[2] = {0.0f, 0.0f};
        for (int i = tid; i < n; i += FT) { v[0] += partials[(size_t)i * NPART_F + 14]; v[1] += partials[(size_t)i * NPART_F + 15]; }
        block_sums<2>(v, sh);
        if (tid == 0) {
            const float denom = fmaxf(3.0f * v[1], 1.0f);
            loss_state[0] = v[0] / denom;
            loss_state[1] = 1.0f / denom;
            loss_state[2] = v[1];
            loss_state[3] = v[0];       // the masked sum itself (colvo_warp_loss_rescale)
        }
        return;
    }
    const int b = blockIdx.x;
    const float t = image_sums(partials, b, blocks_per_image, sh);
    if (tid < NPART) {
        gpart[b * NPART + tid] = t;
        s1[tid] = t;
    }
    if (gunit == nullptr) return;
    __syncthreads();
    if (tid == 0) {
        float drx, dry, drz;
        dR_to_euler(s1 + 3, pose + 6 * b, drx, dry, drz);
        gunit[6 * b + 0] = s1[0]; gunit[6 * b + 1] = s1[1]; gunit[6 * b + 2] = s1[2];
        gunit[6 * b + 3] = drx; gunit[6 * b + 4] = dry; gunit[6 * b + 5] = drz;
        gunit[6 * B + b] = s1[12];
        gunit[7 * B + b] = s1[13];
    }
}

// fused backward: scale = dL/dloss / max(3 n_valid, 1).  Blocks 0..B-1: pose / LCC gradients of one image from gpart;
// all blocks: d_depth = scale * d_depth_raw (grid-stride).
__global__ __launch_bounds__(NT) void k_warp_loss_fused_bwd(const float* __restrict__ loss_state,
                                                            const float* __restrict__ grad_loss,
                                                            const float* __restrict__ d_depth_raw,
                                                            const float* __restrict__ gpart, const float* __restrict__ pose,
                                                            int B, size_t n, float* __restrict__ d_depth,
                                                            float* __restrict__ d_pose, float* __restrict__ d_a,
                                                            float* __restrict__ d_b) {
    const float scale = grad_loss[0] * loss_state[1];
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) d_depth[i] = scale * d_depth_raw[i];
    if ((int)blockIdx.x < B && threadIdx.x == 0) {
        const int b = blockIdx.x;
        float tot[NPART];
#pragma unroll
        for (int k = 0; k < NPART; ++k) tot[k] = scale * gpart[b * NPART + k];
        const float* p = pose + 6 * b;
        const float sx = sinf(p[3]), cx = cosf(p[3]), sy = sinf(p[4]), cy = cosf(p[4]), sz = sinf(p[5]), cz = cosf(p[5]);
        const float* dR = tot + 3;  // row-major 3x3
        const float drx = dR[1] * (cz * sy * cx + sz * sx) + dR[2] * (-cz * sy * sx + sz * cx)
                        + dR[4] * (sz * sy * cx - cz * sx) + dR[5] * (-sz * sy * sx - cz * cx)
                        + dR[7] * (cy * cx) + dR[8] * (-cy * sx);
        const float dry = dR[0] * (-cz * sy) + dR[1] * (cz * cy * sx) + dR[2] * (cz * cy * cx)
                        + dR[3] * (-sz * sy) + dR[4] * (sz * cy * sx) + dR[5] * (sz * cy * cx)
                        + dR[6] * (-cy) + dR[7] * (-sy * sx) + dR[8] * (-sy * cx);
        const float drz = dR[0] * (-sz * cy) + dR[1] * (-sz * sy * sx - cz * cx) + dR[2] * (-sz * sy * cx + cz * sx)
                        + dR[3] * (cz * cy) + dR[4] * (cz * sy * sx - sz * cx) + dR[5] * (cz * sy * cx + sz * sx);
        d_pose[6 * b + 0] = tot[0]; d_pose[6 * b + 1] = tot[1]; d_pose[6 * b + 2] = tot[2];
        d_pose[6 * b + 3] = drx; d_pose[6 * b + 4] = dry; d_pose[6 * b + 5] = drz;
        d_a[b] = tot[12];
        d_b[b] = tot[13];
    }
}

// pose / LCC gradients only (one thread per image): the depth gradient's normalisation is applied by its consumer
__global__ __launch_bounds__(64) void k_warp_loss_fused_bwd_params(const float* __restrict__ loss_state,
                                                                   const float* __restrict__ grad_loss,
                                                                   const float* __restrict__ gpart,
                                                                   const float* __restrict__ pose, int B,
                                                                   float* __restrict__ d_pose, float* __restrict__ d_a,
                                                                   float* __restrict__ d_b) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float scale = grad_loss[0] * loss_state[1];
    float tot[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) tot[k] = scale * gpart[b * NPART + k];
    const float* p = pose + 6 * b;
    const float sx = sinf(p[3]), cx = cosf(p[3]), sy = sinf(p[4]), cy = cosf(p[4]), sz = sinf(p[5]), cz = cosf(p[5]);
    const float* dR = tot + 3;  // row-major 3x3
    const float drx = dR[1] * (cz * sy * cx + sz * sx) + dR[2] * (-cz * sy * sx + sz * cx)
                    + dR[4] * (sz * sy * cx - cz * sx) + dR[5] * (-sz * sy * sx - cz * cx)
                    + dR[7] * (cy * cx) + dR[8] * (-cy * sx);
    const float dry = dR[0] * (-cz * sy) + dR[1] * (cz * cy * sx) + dR[2] * (cz * cy * cx)
                    + dR[3] * (-sz * sy) + dR[4] * (sz * cy * sx) + dR[5] * (sz * cy * cx)
                    + dR[6] * (-cy) + dR[7] * (-sy * sx) + dR[8] * (-sy * cx);
    const float drz = dR[0] * (-sz * cy) + dR[1] * (-sz * sy * sx - cz * cx) + dR[2] * (-sz * sy * cx + cz * sx)
                    + dR[3] * (cz * cy) + dR[4] * (cz * sy * sx - sz * cx) + dR[5] * (cz * sy * cx + sz * sx);
    d_pose[6 * b + 0] = tot[0]; d_pose[6 * b + 1] = tot[1]; d_pose[6 * b + 2] = tot[2];
    d_pose[6 * b + 3] = drx; d_pose[6 * b + 4] = dry; d_pose[6 * b + 5] = drz;
    d_a[b] = tot[12];
    d_b[b] = tot[13];
}

// one workgroup per image: fixed-order sum of that image's tile partials, then dR -> d(euler)
__global__ __launch_bounds__(NT) void k_warp_loss_bwd_finalize(const float* __restrict__ partials, int blocks_per_image,
                                                               const float* __restrict__ pose,
                                                               float* __restrict__ d_pose, float* __restrict__ d_a,
                                                               float* __restrict__ d_b) {
    __shared__ float s[NPART][NT / NPART + 1];
    __shared__ float tot[NPART + 2];
    const int b = blockIdx.x;
    constexpr int ROWS = NT / NPART;  // 18 partial rows per pass
    const int k = threadIdx.x % NPART, r = threadIdx.x / NPART;
    float acc = 0.0f;
    if (r < ROWS)
        for (int i = r; i < blocks_per_image; i += ROWS)
            acc += partials[((size_t)b * blocks_per_image + i) * NPART + k];
    if (r < ROWS) s[k][r] = acc;
    __syncthreads();
    if (threadIdx.x < NPART) {
        float t = 0.0f;
        for (int i = 0; i < ROWS; ++i) t += s[threadIdx.x][i];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float* p = pose + 6 * b;
        const float sx = sinf(p[3]), cx = cosf(p[3]), sy = sinf(p[4]), cy = cosf(p[4]), sz = sinf(p[5]), cz = cosf(p[5]);
        const float* dR = tot + 3;  // row-major 3x3
        // dR/d(rx, ry, rz), entry by entry (R = Rz Ry Rx)
        const float drx = dR[1] * (cz * sy * cx + sz * sx) + dR[2] * (-cz * sy * sx + sz * cx)
                        + dR[4] * (sz * sy * cx - cz * sx) + dR[5] * (-sz * sy * sx - cz * cx)
                        + dR[7] * (cy * cx) + dR[8] * (-cy * sx);
        const float dry = dR[0] * (-cz * sy) + dR[1] * (cz * cy * sx) + dR[2] * (cz * cy * cx)
                        + dR[3] * (-sz * sy) + dR[4] * (sz * cy * sx) + dR[5] * (sz * cy * cx)
                        + dR[6] * (-cy) + dR[7] * (-sy * sx) + dR[8] * (-sy * cx);
        const float drz = dR[0] * (-sz * cy) + dR[1] * (-sz * sy * sx - cz * cx) + dR[2] * (-sz * sy * cx + cz * sx)
                        + dR[3] * (cz * cy) + dR[4] * (cz * sy * sx - sz * cx) + dR[5] * (cz * sy * cx + sz * sx);
        d_pose[6 * b + 0] = tot[0]; d_pose[6 * b + 1] = tot[1]; d_pose[6 * b + 2] = tot[2];
        d_pose[6 * b + 3] = drx; d_pose[6 * b + 4] = dry; d_pose[6 * b + 5] = drz;
        d_a[b] = tot[12];
        d_b[b] = tot[13];
    }
}

// --------------------------------------------------------------------------------------------- //
// un-fused inverse warp (debug entry)                                                            //
// --------------------------------------------------------------------------------------------- //
__global__ __launch_bounds__(NT) void k_inverse_warp(const float* __restrict__ ref, const float* __restrict__ depth,
                                                     const float* __restrict__ pose, const float* __restrict__ K,
                                                     int C, int H, int W, float* __restrict__ warped,
                                                     float* __restrict__ valid) {
    __shared__ float s_geo[GEO_N + 2];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) geo_compute(pose, K, nullptr, nullptr, b, s_geo);  // a = 1, b = 0
    __syncthreads();
    const Geo g = geo_load(s_geo);
    const size_t plane = (size_t)H * W;
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    if (o >= plane) return;
    const int v = (int)(o / W), u = (int)(o - (size_t)v * W);
    const Proj p = project_px(g, depth[(size_t)b * plane + o], u, v, H, W);
    valid[(size_t)b * plane + o] = p.valid ? 1.0f : 0.0f;
    const float* refb = ref + (size_t)b * C * plane;
    float* wb = warped + (size_t)b * C * plane;
    if (p.valid) {
        const Taps t = make_taps(p, H, W);
        for (int c = 0; c < C; ++c) {
            const char* r = reinterpret_cast<const char*>(refb + c * plane);
            const float top = *reinterpret_cast<const float*>(r + t.o00) * (1.0f - t.wx) + *reinterpret_cast<const float*>(r + t.o01) * t.wx;
            const float bot = *reinterpret_cast<const float*>(r + t.o10) * (1.0f - t.wx) + *reinterpret_cast<const float*>(r + t.o11) * t.wx;
            wb[c * plane + o] = top * (1.0f - t.wy) + bot * t.wy;
        }
    } else {
        for (int c = 0; c < C; ++c) wb[c * plane + o] = 0.0f;
    }
}


// --------------------------------------------------------------------------------------------- //
// geometric consistency (spec: oracle/colvo_spec.py geometric_consistency_loss)  //
// --------------------------------------------------------------------------------------------- //
// Concept: /root/reference/README.md:1 ("Considering Geometric and Photometric Consistency"), :7 ("alignment of geometric
// projections between consecutive frames").  Pixel-local (no window), so one thread per target pixel: project with the
// fused kernel's own project_px / make_taps_safe, sample the reference frame's depth with the same four taps, compare
// with the projected depth.  It is NOT folded into the marching photometric kernel: that kernel sits at the 256-VGPR cap
// (2 waves / SIMD), and this term adds 4 taps, 3 live values per rolling row and a scatter.
// Bound: HBM, 4 (depth_t) + 4 (depth_r, gathered) bytes read per pixel forward; backward the same + 4 written (d_depth_t)
// + 16 of float atomics (d_depth_r: the four taps; order-dependent in the last bits).
constexpr int GEO_NP = 12;                  // dt[3], dR[9]

struct GeoPix {
    float diff, m;                          // |a - b| / (a + b) (0 where invalid), validity
    float ga, gb;                           // d diff / d a, d diff / d b   (a = D_proj, b = D_samp)
    Proj p;
    Taps t;
    float d00, d01, d10, d11;
};

__device__ __forceinline__ GeoPix geo_pixel(const Geo& g, const float* __restrict__ dt_img, const float* __restrict__ dr_img,
                                            int u, int v, int H, int W) {
    GeoPix o;
    const float d = dt_img[(size_t)v * W + u];
    o.p = project_px(g, d, u, v, H, W);
    o.t = make_taps_safe(o.p, H, W);
    const char* r = reinterpret_cast<const char*>(dr_img);
    o.d00 = *reinterpret_cast<const float*>(r + o.t.o00); o.d01 = *reinterpret_cast<const float*>(r + o.t.o01);
    o.d10 = *reinterpret_cast<const float*>(r + o.t.o10); o.d11 = *reinterpret_cast<const float*>(r + o.t.o11);
    const float ux = 1.0f - o.t.wx, uy = 1.0f - o.t.wy;
    const float top = fmaf(o.d01, o.t.wx, o.d00 * ux), bot = fmaf(o.d11, o.t.wx, o.d10 * ux);
    const float b = fmaf(bot, o.t.wy, top * uy);
    const float a = o.p.Pz;
    o.m = o.p.valid ? 1.0f : 0.0f;
    const float den = o.p.valid ? (a + b) : 1.0f;
    const float inv = 1.0f / den;
    const float df = a - b;
    o.diff = o.m * fabsf(df) * inv;
    const float sg = (df > 0.0f) ? 1.0f : ((df < 0.0f) ? -1.0f : 0.0f);
    // d/da |a-b|/(a+b) = sg 2b / (a+b)^2 ;  d/db = -sg 2a / (a+b)^2
    o.ga = o.m * sg * 2.0f * b * inv * inv;
    o.gb = -o.m * sg * 2.0f * a * inv * inv;
    return o;
}

// forward: per-block partial {sum diff, sum mask}
__global__ __launch_bounds__(NT) void k_geo_loss_fwd(const float* __restrict__ depth_t, const float* __restrict__ depth_r,
                                                     const float* __restrict__ pose, const float* __restrict__ K, int H, int W,
                                                     float* __restrict__ partials) {
    __shared__ float s_geo[GEO_N + 4];
    __shared__ float red[4][2];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) geo_compute(pose, K, nullptr, nullptr, b, s_geo);
    __syncthreads();
    const Geo g = geo_load(s_geo);
    const size_t plane = (size_t)H * W;
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    float sd = 0.0f, sm = 0.0f;
    if (o < plane) {
        const int v = (int)(o / W), u = (int)(o - (size_t)v * W);
        const GeoPix q = geo_pixel(g, depth_t + (size_t)b * plane, depth_r + (size_t)b * plane, u, v, H, W);
        sd = q.diff; sm = q.m;
    }
    sd = wave_sum(sd); sm = wave_sum(sm);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sd; red[threadIdx.x >> 6][1] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t blk = (size_t)b * gridDim.x + blockIdx.x;
        partials[2 * blk] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partials[2 * blk + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

__global__ __launch_bounds__(NT) void k_geo_loss_finalize(const float* __restrict__ partials, int nblk, float* __restrict__ loss_state) {
    __shared__ float s0[NT], s1[NT];
    float a = 0.0f, c = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += NT) { a += partials[2 * i]; c += partials[2 * i + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = c;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float denom = fmaxf(s1[0], 1.0f);
        loss_state[0] = s0[0] / denom;
        loss_state[1] = 1.0f / denom;
        loss_state[2] = s1[0];
        loss_state[3] = 0.0f;
    }
}

// backward (recomputes the forward): d_depth_t written, d_depth_r accumulated with float atomics (zeroed by the caller),
// per-block partials of dt[3], dR[9]
__global__ __launch_bounds__(NT) void k_geo_loss_bwd(const float* __restrict__ depth_t, const float* __restrict__ depth_r,
                                                     const float* __restrict__ pose, const float* __restrict__ K, int H, int W,
                                                     const float* __restrict__ loss_state, const float* __restrict__ grad_loss,
                                                     float* __restrict__ d_depth_t, float* __restrict__ d_depth_r,
                                                     float* __restrict__ partials) {
    __shared__ float s_geo[GEO_N + 4];
    __shared__ float red[4][GEO_NP];
    const int b = blockIdx.y;
    if (threadIdx.x == 0) geo_compute(pose, K, nullptr, nullptr, b, s_geo);
    __syncthreads();
    const Geo g = geo_load(s_geo);
    const float scale = grad_loss[0] * loss_state[1];
    const size_t plane = (size_t)H * W;
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    float part[GEO_NP];
#pragma unroll
    for (int k = 0; k < GEO_NP; ++k) part[k] = 0.0f;
    if (o < plane) {
        const int v = (int)(o / W), u = (int)(o - (size_t)v * W);
        const GeoPix q = geo_pixel(g, depth_t + (size_t)b * plane, depth_r + (size_t)b * plane, u, v, H, W);
        const float ga = scale * q.ga, gb = scale * q.gb;          // zero where invalid
        // sampled depth -> its four taps (scatter) and the sample position
        const float ux = 1.0f - q.t.wx, uy = 1.0f - q.t.wy;
        if (q.p.valid) {
            char* r = reinterpret_cast<char*>(d_depth_r + (size_t)b * plane);
            atomicAdd(reinterpret_cast<float*>(r + q.t.o00), gb * ux * uy);
            atomicAdd(reinterpret_cast<float*>(r + q.t.o01), gb * q.t.wx * uy);
            atomicAdd(reinterpret_cast<float*>(r + q.t.o10), gb * ux * q.t.wy);
            atomicAdd(reinterpret_cast<float*>(r + q.t.o11), gb * q.t.wx * q.t.wy);
        }
        const float gx = gb * fmaf(q.t.wy, q.d11 - q.d10, uy * (q.d01 - q.d00));
        const float gy = gb * fmaf(q.t.wx, q.d11 - q.d01, ux * (q.d10 - q.d00));
        const float iz = q.p.valid ? 1.0f / q.p.Pz : 1.0f;
        const float dPx = gx * g.fx * iz;
        const float dPy = gy * g.fy * iz;
        const float dPz = ga - (dPx * q.p.Px + dPy * q.p.Py) * iz;
        const float d = depth_t[(size_t)b * plane + o];
        const float rx_ = g.r00 * q.p.Xh + g.r01 * q.p.Yh + g.r02;
        const float ry_ = g.r10 * q.p.Xh + g.r11 * q.p.Yh + g.r12;
        const float rz_ = g.r20 * q.p.Xh + g.r21 * q.p.Yh + g.r22;
        d_depth_t[(size_t)b * plane + o] = dPx * rx_ + dPy * ry_ + dPz * rz_;
        const float cX = q.p.Xh * d, cY = q.p.Yh * d, cZ = d;
        part[0] = dPx; part[1] = dPy; part[2] = dPz;
        part[3] = dPx * cX; part[4] = dPx * cY; part[5] = dPx * cZ;
        part[6] = dPy * cX; part[7] = dPy * cY; part[8] = dPy * cZ;
        part[9] = dPz * cX; part[10] = dPz * cY; part[11] = dPz * cZ;
    }
#pragma unroll
    for (int k = 0; k < GEO_NP; ++k) {
        const float t = wave_sum(part[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < GEO_NP) {
        const size_t blk = (size_t)b * gridDim.x + blockIdx.x;
        partials[blk * GEO_NP + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// one workgroup per image: fixed-order sum of the block partials, then dR -> d(euler)
__global__ __launch_bounds__(NT) void k_geo_loss_bwd_finalize(const float* __restrict__ partials, int blocks_per_image,
                                                              const float* __restrict__ pose, float* __restrict__ d_pose) {
    __shared__ float s[GEO_NP][NT / GEO_NP + 1];
    __shared__ float tot[GEO_NP];
    const int b = blockIdx.x;
    constexpr int ROWS = NT / GEO_NP;   // 21
    const int k = threadIdx.x % GEO_NP, r = threadIdx.x / GEO_NP;
    float acc = 0.0f;
    if (r < ROWS)
        for (int i = r; i < blocks_per_image; i += ROWS) acc += partials[((size_t)b * blocks_per_image + i) * GEO_NP + k];
    if (r < ROWS) s[k][r] = acc;
    __syncthreads();
    if (threadIdx.x < GEO_NP) {
        float t = 0.0f;
        for (int i = 0; i < ROWS; ++i) t += s[threadIdx.x][i];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float drx, dry, drz;
        dR_to_euler(tot + 3, pose + 6 * b, drx, dry, drz);
        d_pose[6 * b + 0] = tot[0]; d_pose[6 * b + 1] = tot[1]; d_pose[6 * b + 2] = tot[2];
        d_pose[6 * b + 3] = drx; d_pose[6 * b + 4] = dry; d_pose[6 * b + 5] = drz;
    }
}

// --------------------------------------------------------------------------------------------- //
// edge-aware smoothness (spec: smoothness_loss) and 2x2 average pooling for the  //
// multi-scale photometric term (spec: downsample2 / multiscale_photometric_loss)                 //
// --------------------------------------------------------------------------------------------- //
// weight of the neighbour pair (p, q): exp(-mean_c |I[p] - I[q]|)
__device__ __forceinline__ float edge_w(const float* __restrict__ img, size_t plane, size_t p, size_t q) {
    const float s = fabsf(img[p] - img[q]) + fabsf(img[plane + p] - img[plane + q]) + fabsf(img[2 * plane + p] - img[2 * plane + q]);
    return expf(-s * (1.0f / 3.0f));
}
__device__ __forceinline__ float sgnf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// forward: every pixel owns its right and its lower neighbour pair; per-block partial {Sx, Sy}
__global__ __launch_bounds__(NT) void k_smooth_fwd(const float* __restrict__ depth, const float* __restrict__ img, int H, int W,
                                                   float* __restrict__ partials) {
    __shared__ float red[4][2];
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    float sx = 0.0f, sy = 0.0f;
    if (o < plane) {
        const int v = (int)(o / W), u = (int)(o - (size_t)v * W);
        const float* dp = depth + (size_t)b * plane;
        const float* ip = img + (size_t)b * 3 * plane;
        const float d0 = 1.0f / dp[o];
        if (u + 1 < W) sx = fabsf(1.0f / dp[o + 1] - d0) * edge_w(ip, plane, o, o + 1);
        if (v + 1 < H) sy = fabsf(1.0f / dp[o + W] - d0) * edge_w(ip, plane, o, o + W);
    }
    sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sx; red[threadIdx.x >> 6][1] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const size_t blk = (size_t)b * gridDim.x + blockIdx.x;
        partials[2 * blk] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partials[2 * blk + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

__global__ __launch_bounds__(NT) void k_smooth_finalize(const float* __restrict__ partials, int nblk, float inv_nx, float inv_ny,
                                                        float* __restrict__ loss) {
    __shared__ float s0[NT], s1[NT];
    float a = 0.0f, c = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += NT) { a += partials[2 * i]; c += partials[2 * i + 1]; }
    s0[threadIdx.x] = a; s1[threadIdx.x] = c;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = s0[0] * inv_nx + s1[0] * inv_ny;
}

// backward, gather form (deterministic): a pixel collects from its up to four incident pairs
__global__ __launch_bounds__(NT) void k_smooth_bwd(const float* __restrict__ depth, const float* __restrict__ img, int H, int W,
                                                   float inv_nx, float inv_ny, const float* __restrict__ grad_loss,
                                                   float* __restrict__ d_depth) {
    const int b = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    if (o >= plane) return;
    const int v = (int)(o / W), u = (int)(o - (size_t)v * W);
    const float* dp = depth + (size_t)b * plane;
    const float* ip = img + (size_t)b * 3 * plane;
    const float dep = dp[o];
    const float d0 = 1.0f / dep;
    float gd = 0.0f;                      // d loss / d disp[o]
    if (u + 1 < W) gd -= sgnf(1.0f / dp[o + 1] - d0) * edge_w(ip, plane, o, o + 1) * inv_nx;
    if (u >= 1) gd += sgnf(d0 - 1.0f / dp[o - 1]) * edge_w(ip, plane, o - 1, o) * inv_nx;
    if (v + 1 < H) gd -= sgnf(1.0f / dp[o + W] - d0) * edge_w(ip, plane, o, o + W) * inv_ny;
    if (v >= 1) gd += sgnf(d0 - 1.0f / dp[o - W]) * edge_w(ip, plane, o - W, o) * inv_ny;
    d_depth[(size_t)b * plane + o] = grad_loss[0] * gd * (-d0 * d0);
}

__global__ __launch_bounds__(NT) void k_avgpool2_fwd(const float* __restrict__ x, int Ho, int Wo, float* __restrict__ y) {
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    if (o >= (size_t)Ho * Wo) return;
    const int v = (int)(o / Wo), u = (int)(o - (size_t)v * Wo);
    const float* xp = x + (size_t)blockIdx.y * 4 * Ho * Wo + (size_t)(2 * v) * (2 * Wo) + 2 * u;
    const float2 r0 = *reinterpret_cast<const float2*>(xp), r1 = *reinterpret_cast<const float2*>(xp + 2 * Wo);
    y[(size_t)blockIdx.y * Ho * Wo + o] = 0.25f * ((r0.x + r0.y) + (r1.x + r1.y));
}

__global__ __launch_bounds__(NT) void k_avgpool2_bwd(const float* __restrict__ dy, int Ho, int Wo, float* __restrict__ dx) {
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    if (o >= (size_t)Ho * Wo) return;
    const int v = (int)(o / Wo), u = (int)(o - (size_t)v * Wo);
    const float g = 0.25f * dy[(size_t)blockIdx.y * Ho * Wo + o];
    float* xp = dx + (size_t)blockIdx.y * 4 * Ho * Wo + (size_t)(2 * v) * (2 * Wo) + 2 * u;
    *reinterpret_cast<float2*>(xp) = make_float2(g, g);
    *reinterpret_cast<float2*>(xp + 2 * Wo) = make_float2(g, g);
}

// --------------------------------------------------------------------------------------------- //
// The widened objective as ONE native call (spec: dcdp_full_loss):       //
//   multi-scale photometric (+ the geometric-consistency term inside the level-0 pass)           //
//   + edge-aware smoothness, value and gradients, 6 launches forward and 1 backward              //
// --------------------------------------------------------------------------------------------- //
constexpr int FULL_MAX_LEVELS = 4;
constexpr int FULL_STATE_N = 32;     // [0] total, [1] geometric term, [2] smoothness term, [4+4s ..] level s: loss, c_s, n_valid

// one level of the 2x2-average pyramid for all 7 planes of every pair (target 3, reference 3, target depth 1) in one launch
__global__ __launch_bounds__(NT) void k_pyramid_level(const float* __restrict__ st, const float* __restrict__ sr,
                                                      const float* __restrict__ sd, int B, int Ho, int Wo,
                                                      float* __restrict__ dt, float* __restrict__ dr, float* __restrict__ dd) {
    const size_t o = (size_t)blockIdx.x * NT + threadIdx.x;
    if (o >= (size_t)Ho * Wo) return;
    int p = blockIdx.y;
    const float* x; float* y;
    if (p < 3 * B) { x = st; y = dt; } else if (p < 6 * B) { x = sr; y = dr; p -= 3 * B; } else { x = sd; y = dd; p -= 6 * B; }
    const int v = (int)(o / Wo), u = (int)(o - (size_t)v * Wo);
    const float* xp = x + (size_t)p * 4 * Ho * Wo + (size_t)(2 * v) * (2 * Wo) + 2 * u;
    const float2 r0 = *reinterpret_cast<const float2*>(xp), r1 = *reinterpret_cast<const float2*>(xp + 2 * Wo);
    y[(size_t)p * Ho * Wo + o] = 0.25f * ((r0.x + r0.y) + (r1.x + r1.y));
}

// 2x2-average pyramid cell of one plane: NL levels below the source in one go (NL = 2: a 4x4 block -> four level-1 pixels and
// one level-2 pixel, the same arithmetic as pooling twice).  Hs, Ws: extent of the SOURCE plane (divisible by 2^NL).
template <int NL>
__device__ __forceinline__ void pyramid_cell(const float* __restrict__ src, int Hs, int Ws, int cy, int cx,
                                             float* __restrict__ d1, float* __restrict__ d2) {
    if constexpr (NL == 1) {
        const float* xp = src + (size_t)(2 * cy) * Ws + 2 * cx;
        const float2 r0 = *reinterpret_cast<const float2*>(xp), r1 = *reinterpret_cast<const float2*>(xp + Ws);
        d1[(size_t)cy * (Ws >> 1) + cx] = 0.25f * ((r0.x + r0.y) + (r1.x + r1.y));
    } else {
        const float* xp = src + (size_t)(4 * cy) * Ws + 4 * cx;
        const float4 r0 = *reinterpret_cast<const float4*>(xp), r1 = *reinterpret_cast<const float4*>(xp + Ws);
        const float4 r2 = *reinterpret_cast<const float4*>(xp + 2 * (size_t)Ws), r3 = *reinterpret_cast<const float4*>(xp + 3 * (size_t)Ws);
        const float a00 = 0.25f * ((r0.x + r0.y) + (r1.x + r1.y)), a01 = 0.25f * ((r0.z + r0.w) + (r1.z + r1.w));
        const float a10 = 0.25f * ((r2.x + r2.y) + (r3.x + r3.y)), a11 = 0.25f * ((r2.z + r2.w) + (r3.z + r3.w));
        const int W1 = Ws >> 1;
        *reinterpret_cast<float2*>(d1 + (size_t)(2 * cy) * W1 + 2 * cx) = make_float2(a00, a01);
        *reinterpret_cast<float2*>(d1 + (size_t)(2 * cy + 1) * W1 + 2 * cx) = make_float2(a10, a11);
        d2[(size_t)cy * (Ws >> 2) + cx] = 0.25f * ((a00 + a01) + (a10 + a11));
    }
}

struct PrepArgs {
    const float *tgt, *ref, *depth;          // level 0: [B,3,H,W], [B,3,H,W], [B,1,H,W]
    float *t1, *r1, *d1, *t2, *r2, *d2;      // levels 1 (and 2) of the three tensors
    float *sm_partials, *sd_raw;             // smoothness: block partials {Sx, Sy}, d term / d depth
    unsigned long long* zero_acc;            // the geometric term's scatter accumulators to clear (or null)
    float inv_nx, inv_ny;
    int B, H, W;
    int sm_blocks;                           // B * ceil(H W / 256) when the smoothness term is on, else 0
};

// Everything the one-pass kernels wait for, in ONE launch: blocks [0, sm_blocks) evaluate the smoothness term (value partials
// and gradient in one pass, gather form as k_smooth_bwd), the rest build NL levels of the pyramid (one thread per cell of every
// plane).
template <int NL>
__global__ __launch_bounds__(NT) void k_full_prepare(PrepArgs a) {
    const int H = a.H, W = a.W, B = a.B;
    if ((int)blockIdx.x >= a.sm_blocks) {
        if constexpr (NL > 0) {
            const int hc = H >> NL, wc = W >> NL;
            const size_t cells = (size_t)hc * wc;
            const size_t q = (size_t)(blockIdx.x - a.sm_blocks) * NT + threadIdx.x;
            if (q >= cells * 7 * B) return;
            int p = (int)(q / cells);
            const size_t c = q - (size_t)p * cells;
            const int cy = (int)(c / wc), cx = (int)(c - (size_t)cy * wc);
            const float* src; float *d1, *d2;
            if (p < 3 * B) { src = a.tgt; d1 = a.t1; d2 = a.t2; } else if (p < 6 * B) { src = a.ref; d1 = a.r1; d2 = a.r2; p -= 3 * B; }
            else { src = a.depth; d1 = a.d1; d2 = a.d2; p -= 6 * B; }
            const size_t plane = (size_t)H * W;
            pyramid_cell<NL>(src + p * plane, H, W, cy, cx, d1 + p * (plane >> 2), NL > 1 ? d2 + p * (plane >> 4) : nullptr);
        }
        return;
    }
    __shared__ float red[4][2];
    const size_t plane = (size_t)H * W;
    const int nb = a.sm_blocks / B;
    const int b = blockIdx.x / nb;
    const size_t o = (size_t)(blockIdx.x - b * nb) * NT + threadIdx.x;
    float sx = 0.0f, sy = 0.0f;
    if (o < plane) {
        const int v = (int)(o / W), u = (int)(o - (size_t)v * W);
        const float* dp = a.depth + (size_t)b * plane;
        const float* ip = a.tgt + (size_t)b * 3 * plane;
        const float d0 = 1.0f / dp[o];
        float gd = 0.0f;
        if (u + 1 < W) {
            const float dd = 1.0f / dp[o + 1] - d0, w = edge_w(ip, plane, o, o + 1);
            sx = fabsf(dd) * w;
            gd -= sgnf(dd) * w * a.inv_nx;
        }
        if (u >= 1) gd += sgnf(d0 - 1.0f / dp[o - 1]) * edge_w(ip, plane, o - 1, o) * a.inv_nx;
        if (v + 1 < H) {
            const float dd = 1.0f / dp[o + W] - d0, w = edge_w(ip, plane, o, o + W);
            sy = fabsf(dd) * w;
            gd -= sgnf(dd) * w * a.inv_ny;
        }
        if (v >= 1) gd += sgnf(d0 - 1.0f / dp[o - W]) * edge_w(ip, plane, o - W, o) * a.inv_ny;
        a.sd_raw[(size_t)b * plane + o] = gd * (-d0 * d0);
        if (a.zero_acc) a.zero_acc[(size_t)b * plane + o] = 0ull;
    }
    sx = wave_sum(sx); sy = wave_sum(sy);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sx; red[threadIdx.x >> 6][1] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.sm_partials[2 * (size_t)blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        a.sm_partials[2 * (size_t)blockIdx.x + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

struct FullLevels {
    int S, B;
    int items_per_image[FULL_MAX_LEVELS];
    const float* partials[FULL_MAX_LEVELS];      // the level's march partials (NPART_F per strip segment)
    const float* raw[FULL_MAX_LEVELS];           // the level's unnormalised depth gradient [B, H >> s, W >> s]
};

// Blocks 0 .. S*B-1: the 14 gradient sums of one image at one level (still unnormalised) -> gpart[s][b][14].
// Last block: every term's value and normaliser -> state, the weighted total -> state[0] and loss_out: each thread gathers
// its share of all (at most 11) sums at once.  Fixed orders: deterministic.
__global__ __launch_bounds__(FT) void k_full_finalize(FullLevels lv, const float* __restrict__ geo_partials,
                                                      const float* __restrict__ sm_partials, int sm_nblk, float inv_nx,
                                                      float inv_ny, float w_geo, float w_sm, float* __restrict__ gpart,
                                                      float* __restrict__ state, float* __restrict__ loss_out) {
    __shared__ float sh[FT];
    const int tid = threadIdx.x, B = lv.B;
    if ((int)blockIdx.x == B * lv.S) {
        constexpr int NV = 2 * FULL_MAX_LEVELS + 3;
        float v[NV];
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = 0.0f;
#pragma unroll
        for (int s = 0; s < FULL_MAX_LEVELS; ++s) {
            if (s < lv.S) {
                const int n = B * lv.items_per_image[s];
                const float* pp = lv.partials[s];
                for (int i = tid; i < n; i += FT) { v[2 * s] += pp[(size_t)i * NPART_F + 14]; v[2 * s + 1] += pp[(size_t)i * NPART_F + 15]; }
            }
        }
        if (geo_partials) {
            const int n = B * lv.items_per_image[0];
            for (int i = tid; i < n; i += FT) v[2 * FULL_MAX_LEVELS] += geo_partials[i];
        }
        if (sm_partials)
            for (int i = tid; i < sm_nblk; i += FT) {
                v[2 * FULL_MAX_LEVELS + 1] += sm_partials[2 * i];
                v[2 * FULL_MAX_LEVELS + 2] += sm_partials[2 * i + 1];
            }
        block_sums<NV>(v, sh);
        if (tid == 0) {
            float total = 0.0f;
            const float w = 1.0f / (float)lv.S;
            for (int s = 0; s < lv.S; ++s) {
                const float denom = fmaxf(3.0f * v[2 * s + 1], 1.0f), l = v[2 * s] / denom;
                total += w * l;
                state[4 + 4 * s] = l; state[4 + 4 * s + 1] = w / denom; state[4 + 4 * s + 2] = v[2 * s + 1]; state[4 + 4 * s + 3] = 0.0f;
            }
            float geo = 0.0f, sm = 0.0f;
            if (geo_partials) { geo = v[2 * FULL_MAX_LEVELS] / fmaxf(v[1], 1.0f); total += w_geo * geo; }
            if (sm_partials) { sm = v[2 * FULL_MAX_LEVELS + 1] * inv_nx + v[2 * FULL_MAX_LEVELS + 2] * inv_ny; total += w_sm * sm; }
            state[0] = total; state[1] = geo; state[2] = sm; state[3] = 0.0f;
            loss_out[0] = total;
        }
        return;
    }
    const int s = blockIdx.x / B, b = blockIdx.x - s * B;
    const float t = image_sums(lv.partials[s], b, lv.items_per_image[s], sh);
    if (tid < NPART) gpart[((size_t)s * B + b) * NPART + tid] = t;
}

// backward of the whole objective: every gradient in one launch.  Grid (ceil(H W / 256), B + 1):
//   y < B:  d_dt = g (sum_s c_s / 4^s raw_s[y >> s, x >> s] + c_0 geo_raw + w_sm sd_raw)   (the pooling chain is the index shift)
//           d_dr = g c_0 acc / 2^32                                     (fixed-point scatter of the geometric term)
//   y == B: blocks x < B: pose / LCC gradients of image x from gpart (dR -> Euler angles)
__global__ __launch_bounds__(NT) void k_full_combine(FullLevels lv, const float* __restrict__ state,
                                                     const float* __restrict__ grad_loss, const float* __restrict__ sd_raw,
                                                     const float* __restrict__ geo_raw, const unsigned long long* __restrict__ acc,
                                                     const float* __restrict__ gpart, const float* __restrict__ pose, int H, int W,
                                                     float w_sm, float* __restrict__ d_dt, float* __restrict__ d_dr,
                                                     float* __restrict__ d_pose, float* __restrict__ d_a, float* __restrict__ d_b) {
    const float g = grad_loss[0];
    const int B = lv.B;
    float c[FULL_MAX_LEVELS];
#pragma unroll
    for (int s = 0; s < FULL_MAX_LEVELS; ++s) c[s] = (s < lv.S) ? g * state[4 + 4 * s + 1] : 0.0f;
    if ((int)blockIdx.y < B) {
        const unsigned plane = (unsigned)H * W, o = blockIdx.x * NT + threadIdx.x;
        if (o >= plane) return;
        const unsigned b = blockIdx.y, y = o / W, x = o - y * W;
        const size_t i = (size_t)b * plane + o;
        float v = c[0] * (geo_raw ? lv.raw[0][i] + geo_raw[i] : lv.raw[0][i]);
        float f = 0.25f;
#pragma unroll
        for (int s = 1; s < FULL_MAX_LEVELS; ++s) {
            if (s < lv.S) {
                const unsigned ws = W >> s, hs = H >> s;
                v = fmaf(c[s] * f, lv.raw[s][((size_t)b * hs + (y >> s)) * ws + (x >> s)], v);
                f *= 0.25f;
            }
        }
        if (sd_raw) v = fmaf(g * w_sm, sd_raw[i], v);
        d_dt[i] = v;
        if (d_dr) d_dr[i] = c[0] * ((float)(long long)acc[i] * (1.0f / GEO_FIX));
        return;
    }
    if ((int)blockIdx.x >= B) return;
    __shared__ float tot[NPART + 2];
    const int b = blockIdx.x;
    if (threadIdx.x < NPART) {
        float t = 0.0f;
#pragma unroll
        for (int s = 0; s < FULL_MAX_LEVELS; ++s)
            if (s < lv.S) t = fmaf(c[s], gpart[((size_t)s * B + b) * NPART + threadIdx.x], t);
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float drx, dry, drz;
        dR_to_euler(tot + 3, pose + 6 * b, drx, dry, drz);
        d_pose[6 * b + 0] = tot[0]; d_pose[6 * b + 1] = tot[1]; d_pose[6 * b + 2] = tot[2];
        d_pose[6 * b + 3] = drx; d_pose[6 * b + 4] = dry; d_pose[6 * b + 5] = drz;
        d_a[b] = tot[12];
        d_b[b] = tot[13];
    }
}

// where everything lives in the caller's workspace (fp32 units; the 64-bit accumulators come first)
struct FullPlan {
    int S;
    int h[FULL_MAX_LEVELS], w[FULL_MAX_LEVELS], seg_rows[FULL_MAX_LEVELS], strips[FULL_MAX_LEVELS], nseg[FULL_MAX_LEVELS];
    size_t tgt[FULL_MAX_LEVELS], ref[FULL_MAX_LEVELS], dep[FULL_MAX_LEVELS], raw[FULL_MAX_LEVELS], part[FULL_MAX_LEVELS];
    size_t acc, geo_part, geo_raw, sd, sm_part, gpart, state, total;
    int sm_nblk;
};

inline FullPlan full_plan(int B, int H, int W, int S) {
    FullPlan p{};
    p.S = S;
    size_t o = 0;
    auto take = [&](size_t n) { const size_t at = o; o += (n + 3) & ~(size_t)3; return at; };      // 16-byte granules
    p.acc = take(2 * (size_t)B * H * W);
    for (int s = 0; s < S; ++s) {
        const int h = H >> s, w = W >> s;
        p.h[s] = h; p.w[s] = w;
        int rows = pick_march_rows(B, h, w, BCOLS, 4, 8);
        if (TUNE(march_rows_bwd) > 0) rows = std::max(4, std::min(MROWS_MAX, (int)TUNE(march_rows_bwd)));
        p.seg_rows[s] = rows;
        p.strips[s] = (w + BCOLS - 1) / BCOLS;
        p.nseg[s] = (h + rows - 1) / rows;
        const size_t px = (size_t)B * h * w;
        if (s > 0) { p.tgt[s] = take(3 * px); p.ref[s] = take(3 * px); p.dep[s] = take(px); }
        p.raw[s] = take(px);
        p.part[s] = take((size_t)B * p.nseg[s] * p.strips[s] * NPART_F);
    }
    p.geo_part = take((size_t)B * p.nseg[0] * p.strips[0]);
    p.geo_raw = take((size_t)B * H * W);
    p.sd = take((size_t)B * H * W);
    p.sm_nblk = B * (int)(((size_t)H * W + NT - 1) / NT);
    p.sm_part = take(2 * (size_t)p.sm_nblk);
    p.gpart = take((size_t)S * B * NPART);
    p.state = take(FULL_STATE_N);
    p.total = o;
    return p;
}

inline bool full_shape_ok(int B, int H, int W, int S) {
    return B >= 1 && B <= 9000 && S >= 1 && S <= FULL_MAX_LEVELS && (H >> (S - 1)) >= 2 && (W >> (S - 1)) >= 2
           && H % (1 << (S - 1)) == 0 && W % (1 << (S - 1)) == 0 && (size_t)H * W < (1u << 28);
}


}  // namespace
}  // namespace colvo

using namespace colvo;

extern "C" size_t colvo_warp_loss_workspace_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t bwd = (size_t)B * ((W + BCOLS - 1) / BCOLS) * ((H + 3) / 4) * NPART_F; // <= 16 per strip segment (>= 4 rows)
    const size_t fwd = (size_t)B * ((W + MCOLS - 1) / MCOLS) * ((H + 3) / 4) * 2;   // 2 per strip segment (>= 4 rows each)
    return bwd > fwd ? bwd : fwd;
}

extern "C" int colvo_warp_loss_fwd(const float* tgt, const float* ref, const float* depth, const float* pose,
                                   const float* K, const float* lcc_a, const float* lcc_b, int B, int H, int W,
                                   float ssim_weight, float* workspace, float* loss_state, colvo_stream_t stream) {
    COLVO_CHECK_ARG(tgt && ref && depth && pose && K && lcc_a && lcc_b && workspace && loss_state,
                    "colvo_warp_loss_fwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && H >= 2 && W >= 2 && B <= 65535, "colvo_warp_loss_fwd: bad shape B=%d H=%d W=%d", B, H, W);
    COLVO_CHECK_ARG((size_t)H * W < (1u << 30), "colvo_warp_loss_fwd: image too large");
    hipStream_t s = (hipStream_t)stream;
    // marching-wave forward: one wave per (image, 32-row segment, 62-column strip)
    int seg_rows = pick_march_rows(B, H, W);
    if (TUNE(march_rows_fwd) > 0) seg_rows = std::max(4, std::min(MROWS_MAX, (int)TUNE(march_rows_fwd)));   // tuning knob
    const int strips_x = (W + MCOLS - 1) / MCOLS, nseg = (H + seg_rows - 1) / seg_rows;
    const long long nitems = (long long)B * nseg * strips_x;
    COLVO_CHECK_ARG(nitems < (1ll << 30), "colvo_warp_loss_fwd: too many strips");
    const int nblk = (int)nitems;
    colvo::launch(k_warp_loss_fwd_march, dim3((unsigned)((nitems + 3) / 4)), dim3(NT), 0, s, tgt, ref, depth, pose, K,
                       lcc_a, lcc_b, B, H, W, strips_x, nseg, seg_rows, ssim_weight, workspace);
    COLVO_CHECK_LAUNCH("k_warp_loss_fwd_march");
    colvo::launch(k_warp_loss_fwd_finalize, dim3(1), dim3(NT), 0, s, workspace, nblk, loss_state);
    COLVO_CHECK_LAUNCH("k_warp_loss_fwd_finalize");
    return 0;
}

extern "C" int colvo_warp_loss_bwd(const float* tgt, const float* ref, const float* depth, const float* pose,
                                   const float* K, const float* lcc_a, const float* lcc_b, int B, int H, int W,
                                   float ssim_weight, const float* loss_state, const float* grad_loss,
                                   float* workspace, float* d_depth, float* d_pose, float* d_a, float* d_b,
                                   colvo_stream_t stream) {
    COLVO_CHECK_ARG(tgt && ref && depth && pose && K && lcc_a && lcc_b && workspace && loss_state && grad_loss
                        && d_depth && d_pose && d_a && d_b,
                    "colvo_warp_loss_bwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && H >= 2 && W >= 2 && B <= 65535, "colvo_warp_loss_bwd: bad shape B=%d H=%d W=%d", B, H, W);
    COLVO_CHECK_ARG((size_t)H * W < (1u << 30), "colvo_warp_loss_bwd: image too large");
    hipStream_t s = (hipStream_t)stream;
    // marching-wave backward: one wave per (image, row segment, 60-column strip)
    int seg_rows = pick_march_rows(B, H, W, BCOLS, 4, 8);
    if (TUNE(march_rows_bwd) > 0) seg_rows = std::max(4, std::min(MROWS_MAX, (int)TUNE(march_rows_bwd)));   // tuning knob
    const int strips_x = (W + BCOLS - 1) / BCOLS, nseg = (H + seg_rows - 1) / seg_rows;
    const long long nitems = (long long)B * nseg * strips_x;
    COLVO_CHECK_ARG(nitems < (1ll << 30), "colvo_warp_loss_bwd: too many strips");
    colvo::launch((k_warp_loss_bwd_march<false>), dim3((unsigned)((nitems + 3) / 4)), dim3(NT), 0, s, tgt, ref, depth, pose, K,
                       lcc_a, lcc_b, B, H, W, strips_x, nseg, seg_rows, ssim_weight, loss_state, grad_loss, d_depth, workspace, 0, GeoArgs{});
    COLVO_CHECK_LAUNCH("k_warp_loss_bwd_march");
    colvo::launch(k_warp_loss_bwd_finalize, dim3(B), dim3(NT), 0, s, workspace, nseg * strips_x, pose,
                       d_pose, d_a, d_b);
    COLVO_CHECK_LAUNCH("k_warp_loss_bwd_finalize");
    return 0;
}

// Training path: loss AND its (unnormalised) gradients in ONE pass over the images -- the backward kernel evaluates
// everything the forward does, so running both costs a second read of every input and ~1/3 more instructions.
extern "C" int colvo_warp_loss_fused(const float* tgt, const float* ref, const float* depth, const float* pose,
                                     const float* K, const float* lcc_a, const float* lcc_b, int B, int H, int W,
                                     float ssim_weight, float* workspace, float* loss_state, float* d_depth_raw,
                                     float* grad_partials, float* grad_unit, colvo_stream_t stream) {
    COLVO_CHECK_ARG(tgt && ref && depth && pose && K && lcc_a && lcc_b && workspace && loss_state && d_depth_raw && grad_partials,
                    "colvo_warp_loss_fused: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && H >= 2 && W >= 2 && B <= 65534, "colvo_warp_loss_fused: bad shape B=%d H=%d W=%d", B, H, W);
    COLVO_CHECK_ARG((size_t)H * W < (1u << 30), "colvo_warp_loss_fused: image too large");
    hipStream_t s = (hipStream_t)stream;
    int seg_rows = pick_march_rows(B, H, W, BCOLS, 4, 8);
    if (TUNE(march_rows_bwd) > 0) seg_rows = std::max(4, std::min(MROWS_MAX, (int)TUNE(march_rows_bwd)));   // tuning knob
    const int strips_x = (W + BCOLS - 1) / BCOLS, nseg = (H + seg_rows - 1) / seg_rows;
    const long long nitems = (long long)B * nseg * strips_x;
    COLVO_CHECK_ARG(nitems < (1ll << 30), "colvo_warp_loss_fused: too many strips");
    colvo::launch((k_warp_loss_bwd_march<true>), dim3((unsigned)((nitems + 3) / 4)), dim3(NT), 0, s, tgt, ref, depth, pose,
                       K, lcc_a, lcc_b, B, H, W, strips_x, nseg, seg_rows, ssim_weight, (const float*)nullptr,
                       (const float*)nullptr, d_depth_raw, workspace, 0, GeoArgs{});
    COLVO_CHECK_LAUNCH("k_warp_loss_bwd_march<fused>");
    colvo::launch(k_warp_loss_fused_finalize, dim3(B + 1), dim3(FT), 0, s, workspace, nseg * strips_x, B, pose,
                       grad_partials, grad_unit, loss_state);
    COLVO_CHECK_LAUNCH("k_warp_loss_fused_finalize");
    return 0;
}

extern "C" int colvo_warp_loss_fused_bwd(const float* loss_state, const float* grad_loss, const float* d_depth_raw,
                                         const float* grad_partials, const float* pose, int B, int H, int W,
                                         float* d_depth, float* d_pose, float* d_a, float* d_b, colvo_stream_t stream) {
    COLVO_CHECK_ARG(loss_state && grad_loss && d_depth_raw && grad_partials && pose && d_depth && d_pose && d_a && d_b,
                    "colvo_warp_loss_fused_bwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && H >= 2 && W >= 2 && B <= 65534, "colvo_warp_loss_fused_bwd: bad shape");
    const size_t n = (size_t)B * H * W;
    unsigned blocks = (unsigned)std::min<size_t>((n + NT * 4 - 1) / (NT * 4), 4096);
    if (blocks < (unsigned)B) blocks = (unsigned)B;
    colvo::launch(k_warp_loss_fused_bwd, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, loss_state, grad_loss, d_depth_raw,
                       grad_partials, pose, B, n, d_depth, d_pose, d_a, d_b);
    COLVO_CHECK_LAUNCH("k_warp_loss_fused_bwd");
    return 0;
}

extern "C" int colvo_warp_loss_fused_bwd_params(const float* loss_state, const float* grad_loss, const float* grad_partials,
                                                const float* pose, int B, float* d_pose, float* d_a, float* d_b,
                                                colvo_stream_t stream) {
    COLVO_CHECK_ARG(loss_state && grad_loss && grad_partials && pose && d_pose && d_a && d_b,
                    "colvo_warp_loss_fused_bwd_params: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && B <= 65534, "colvo_warp_loss_fused_bwd_params: bad batch %d", B);
    colvo::launch(k_warp_loss_fused_bwd_params, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, loss_state, grad_loss,
                       grad_partials, pose, B, d_pose, d_a, d_b);
    COLVO_CHECK_LAUNCH("k_warp_loss_fused_bwd_params");
    return 0;
}

// Data parallel: loss_state[2] (valid pixels) and loss_state[3] (masked sum) have been added up over `world` ranks -> the loss of
// the WHOLE batch and the normaliser every rank's raw gradients take so that (1 / world) x the all-reduced gradient is the gradient
// of that loss: world / max(3 n_global, 1).  The same arithmetic as the finalize kernels (world = 1: bit for bit what they wrote).
__global__ void k_warp_loss_rescale(float* __restrict__ loss_state, float world, float* __restrict__ scale_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float denom = fmaxf(3.0f * loss_state[2], 1.0f);
    loss_state[0] = loss_state[3] / denom;
    const float sc = world == 1.0f ? 1.0f / denom : world / denom;
    loss_state[1] = sc;
    if (scale_out) *scale_out = sc;
}

extern "C" int colvo_warp_loss_rescale(float* loss_state, int world, colvo_stream_t stream) {
    COLVO_CHECK_ARG(loss_state && world >= 1, "colvo_warp_loss_rescale: null state or world < 1");
    colvo::launch(k_warp_loss_rescale, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_state, (float)world, (float*)nullptr);
    COLVO_CHECK_LAUNCH("k_warp_loss_rescale");
    return 0;
}

extern "C" int colvo_warp_loss_rescale_to(float* loss_state, int world, float* scale_out, colvo_stream_t stream) {
    COLVO_CHECK_ARG(loss_state && scale_out && world >= 1, "colvo_warp_loss_rescale_to: null pointer or world < 1");
    colvo::launch(k_warp_loss_rescale, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_state, (float)world, scale_out);
    COLVO_CHECK_LAUNCH("k_warp_loss_rescale");
    return 0;
}

extern "C" int colvo_inverse_warp(const float* ref, const float* depth, const float* pose, const float* K, int B,
                                  int C, int H, int W, float* warped, float* valid, colvo_stream_t stream) {
    COLVO_CHECK_ARG(ref && depth && pose && K && warped && valid, "colvo_inverse_warp: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && C >= 1 && H >= 1 && W >= 1 && B <= 65535, "colvo_inverse_warp: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const size_t plane = (size_t)H * W;
    dim3 grid((unsigned)((plane + NT - 1) / NT), B);
    colvo::launch(k_inverse_warp, grid, dim3(NT), 0, s, ref, depth, pose, K, C, H, W, warped, valid);
    COLVO_CHECK_LAUNCH("k_inverse_warp");
    return 0;
}

// ---- SURVEY.md §8f-1 / §8f-2 entry points ------------------------------------------------------------------------- //
extern "C" size_t colvo_geo_loss_workspace_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * (((size_t)H * W + NT - 1) / NT) * GEO_NP;
}

extern "C" int colvo_geo_loss_fwd(const float* depth_t, const float* depth_r, const float* pose, const float* K, int B, int H,
                                  int W, float* workspace, float* loss_state, colvo_stream_t stream) {
    COLVO_CHECK_ARG(depth_t && depth_r && pose && K && workspace && loss_state, "colvo_geo_loss_fwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && B <= 65535 && H >= 2 && W >= 2 && (size_t)H * W < (1u << 30), "colvo_geo_loss_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = (unsigned)(((size_t)H * W + NT - 1) / NT);
    colvo::launch(k_geo_loss_fwd, dim3(nb, B), dim3(NT), 0, s, depth_t, depth_r, pose, K, H, W, workspace);
    COLVO_CHECK_LAUNCH("k_geo_loss_fwd");
    colvo::launch(k_geo_loss_finalize, dim3(1), dim3(NT), 0, s, workspace, (int)(nb * B), loss_state);
    COLVO_CHECK_LAUNCH("k_geo_loss_finalize");
    return 0;
}

extern "C" int colvo_geo_loss_bwd(const float* depth_t, const float* depth_r, const float* pose, const float* K, int B, int H,
                                  int W, const float* loss_state, const float* grad_loss, float* workspace, float* d_depth_t,
                                  float* d_depth_r, float* d_pose, colvo_stream_t stream) {
    COLVO_CHECK_ARG(depth_t && depth_r && pose && K && loss_state && grad_loss && workspace && d_depth_t && d_depth_r && d_pose,
                    "colvo_geo_loss_bwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && B <= 65535 && H >= 2 && W >= 2 && (size_t)H * W < (1u << 30), "colvo_geo_loss_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_depth_r, 0, (size_t)B * H * W * sizeof(float), s);
    if (e != hipSuccess) { set_error("colvo_geo_loss_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e)); return (int)e; }
    const unsigned nb = (unsigned)(((size_t)H * W + NT - 1) / NT);
    colvo::launch(k_geo_loss_bwd, dim3(nb, B), dim3(NT), 0, s, depth_t, depth_r, pose, K, H, W, loss_state, grad_loss,
                       d_depth_t, d_depth_r, workspace);
    COLVO_CHECK_LAUNCH("k_geo_loss_bwd");
    colvo::launch(k_geo_loss_bwd_finalize, dim3(B), dim3(NT), 0, s, workspace, (int)nb, pose, d_pose);
    COLVO_CHECK_LAUNCH("k_geo_loss_bwd_finalize");
    return 0;
}

extern "C" int colvo_smooth_loss_fwd(const float* depth, const float* img, int B, int H, int W, float* workspace, float* loss,
                                     colvo_stream_t stream) {
    COLVO_CHECK_ARG(depth && img && workspace && loss, "colvo_smooth_loss_fwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && B <= 65535 && H >= 2 && W >= 2 && (size_t)H * W < (1u << 30), "colvo_smooth_loss_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = (unsigned)(((size_t)H * W + NT - 1) / NT);
    colvo::launch(k_smooth_fwd, dim3(nb, B), dim3(NT), 0, s, depth, img, H, W, workspace);
    COLVO_CHECK_LAUNCH("k_smooth_fwd");
    colvo::launch(k_smooth_finalize, dim3(1), dim3(NT), 0, s, workspace, (int)(nb * B), 1.0f / ((float)B * H * (W - 1)),
                       1.0f / ((float)B * (H - 1) * W), loss);
    COLVO_CHECK_LAUNCH("k_smooth_finalize");
    return 0;
}

extern "C" int colvo_smooth_loss_bwd(const float* depth, const float* img, int B, int H, int W, const float* grad_loss,
                                     float* d_depth, colvo_stream_t stream) {
    COLVO_CHECK_ARG(depth && img && grad_loss && d_depth, "colvo_smooth_loss_bwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && B <= 65535 && H >= 2 && W >= 2 && (size_t)H * W < (1u << 30), "colvo_smooth_loss_bwd: bad shape");
    const unsigned nb = (unsigned)(((size_t)H * W + NT - 1) / NT);
    colvo::launch(k_smooth_bwd, dim3(nb, B), dim3(NT), 0, (hipStream_t)stream, depth, img, H, W,
                       1.0f / ((float)B * H * (W - 1)), 1.0f / ((float)B * (H - 1) * W), grad_loss, d_depth);
    COLVO_CHECK_LAUNCH("k_smooth_bwd");
    return 0;
}

extern "C" int colvo_avgpool2_fwd(const float* x, int planes, int H, int W, float* y, colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && y, "colvo_avgpool2_fwd: null pointer argument");
    COLVO_CHECK_ARG(planes >= 1 && planes <= 65535 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "colvo_avgpool2_fwd: H, W must be even");
    const int Ho = H / 2, Wo = W / 2;
    colvo::launch(k_avgpool2_fwd, dim3((unsigned)(((size_t)Ho * Wo + NT - 1) / NT), planes), dim3(NT), 0, (hipStream_t)stream,
                       x, Ho, Wo, y);
    COLVO_CHECK_LAUNCH("k_avgpool2_fwd");
    return 0;
}

extern "C" int colvo_avgpool2_bwd(const float* dy, int planes, int H, int W, float* dx, colvo_stream_t stream) {
    COLVO_CHECK_ARG(dy && dx, "colvo_avgpool2_bwd: null pointer argument");
    COLVO_CHECK_ARG(planes >= 1 && planes <= 65535 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "colvo_avgpool2_bwd: H, W must be even");
    const int Ho = H / 2, Wo = W / 2;
    colvo::launch(k_avgpool2_bwd, dim3((unsigned)(((size_t)Ho * Wo + NT - 1) / NT), planes), dim3(NT), 0, (hipStream_t)stream,
                       dy, Ho, Wo, dx);
    COLVO_CHECK_LAUNCH("k_avgpool2_bwd");
    return 0;
}

// ---- the widened objective in one call ---------------------------------------------------------------------------- //
extern "C" size_t colvo_full_objective_workspace_floats(int B, int H, int W, int num_scales) {
    if (!full_shape_ok(B, H, W, num_scales)) return 0;
    return full_plan(B, H, W, num_scales).total;
}

extern "C" int colvo_full_objective_fwd(const float* tgt, const float* ref, const float* depth_t, const float* depth_r,
                                        const float* pose, const float* K, const float* lcc_a, const float* lcc_b, int B, int H,
                                        int W, int num_scales, float ssim_weight, float geo_weight, float smooth_weight,
                                        float* workspace, float* loss, colvo_stream_t stream) {
    COLVO_CHECK_ARG(tgt && ref && depth_t && pose && K && lcc_a && lcc_b && workspace && loss,
                    "colvo_full_objective_fwd: null pointer argument");
    COLVO_CHECK_ARG(geo_weight == 0.0f || depth_r, "colvo_full_objective_fwd: the geometric term needs the reference depth");
    COLVO_CHECK_ARG(full_shape_ok(B, H, W, num_scales),
                    "colvo_full_objective_fwd: bad shape B=%d H=%d W=%d scales=%d (H, W divisible by 2^(scales-1), scales <= %d)",
                    B, H, W, num_scales, FULL_MAX_LEVELS);
    COLVO_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "colvo_full_objective_fwd: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const FullPlan p = full_plan(B, H, W, num_scales);
    const bool geo = geo_weight != 0.0f, smooth = smooth_weight != 0.0f;
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(workspace + p.acc);
    const size_t plane = (size_t)H * W;
    const float inv_nx = 1.0f / ((float)B * H * (W - 1)), inv_ny = 1.0f / ((float)B * (H - 1) * W);      // (H, W >= 2)
    // (1) one launch: smoothness (value partials + raw gradient; clears the scatter accumulators on the way) and the first two
    //     pyramid levels of both frames and the target depth; a fourth level (num_scales = 4) takes a launch of its own
    {
        const int NL = std::min(num_scales - 1, 2);
        PrepArgs a{};
        a.tgt = tgt; a.ref = ref; a.depth = depth_t;
        if (NL >= 1) { a.t1 = workspace + p.tgt[1]; a.r1 = workspace + p.ref[1]; a.d1 = workspace + p.dep[1]; }
        if (NL >= 2) { a.t2 = workspace + p.tgt[2]; a.r2 = workspace + p.ref[2]; a.d2 = workspace + p.dep[2]; }
        a.sm_partials = workspace + p.sm_part; a.sd_raw = workspace + p.sd;
        a.zero_acc = (geo && smooth) ? acc : nullptr;
        a.inv_nx = inv_nx; a.inv_ny = inv_ny;
        a.B = B; a.H = H; a.W = W;
        a.sm_blocks = smooth ? p.sm_nblk : 0;
        if (geo && !smooth) {
            hipError_t e = hipMemsetAsync(acc, 0, (size_t)B * plane * 8, s);
            if (e != hipSuccess) { set_error("colvo_full_objective_fwd: hipMemsetAsync failed: %s", hipGetErrorString(e)); return (int)e; }
        }
        const size_t cells = NL ? 7 * (size_t)B * (plane >> (2 * NL)) : 0;
        const unsigned blocks = (unsigned)(a.sm_blocks + (cells + NT - 1) / NT);
        if (blocks) {
            if (NL == 2) colvo::launch((k_full_prepare<2>), dim3(blocks), dim3(NT), 0, s, a);
            else if (NL == 1) colvo::launch((k_full_prepare<1>), dim3(blocks), dim3(NT), 0, s, a);
            else colvo::launch((k_full_prepare<0>), dim3(blocks), dim3(NT), 0, s, a);
            COLVO_CHECK_LAUNCH("k_full_prepare");
        }
        for (int l = 3; l < num_scales; ++l) {
            const size_t px = (size_t)p.h[l] * p.w[l];
            colvo::launch(k_pyramid_level, dim3((unsigned)((px + NT - 1) / NT), 7 * B), dim3(NT), 0, s, workspace + p.tgt[l - 1],
                               workspace + p.ref[l - 1], workspace + p.dep[l - 1], B, p.h[l], p.w[l], workspace + p.tgt[l],
                               workspace + p.ref[l], workspace + p.dep[l]);
            COLVO_CHECK_LAUNCH("k_pyramid_level");
        }
    }
    // (2) one launch: the one-pass photometric loss + gradients of every level; level 0 carries the geometric term
    FullLevels lv{};
    lv.S = num_scales; lv.B = B;
    MarchLevels ml{};
    ml.S = num_scales;
    unsigned grid = 0;
    for (int l = 0; l < num_scales; ++l) {
        const long long nitems = (long long)B * p.nseg[l] * p.strips[l];
        COLVO_CHECK_ARG(nitems < (1ll << 28), "colvo_full_objective_fwd: too many strips");
        lv.items_per_image[l] = p.nseg[l] * p.strips[l];
        lv.partials[l] = workspace + p.part[l];
        lv.raw[l] = workspace + p.raw[l];
        MarchLevel& L = ml.lv[l];
        L.tgt = l ? workspace + p.tgt[l] : tgt; L.ref = l ? workspace + p.ref[l] : ref; L.depth = l ? workspace + p.dep[l] : depth_t;
        L.d_depth = workspace + p.raw[l]; L.partials = workspace + p.part[l];
        L.H = p.h[l]; L.W = p.w[l]; L.strips_x = p.strips[l]; L.nseg = p.nseg[l]; L.seg_rows = p.seg_rows[l];
        L.wg8 = (int)(((nitems + 3) / 4 + 7) / 8);
        grid += 8u * (unsigned)L.wg8;
    }
    if (geo) {
        // both terms are masked means over the SAME valid pixels (photometric: 3 channels each, weight 1/S): relative to the
        // photometric gradient the term's weighs  geo_weight / n  over  (1/S) / (3 n)
        GeoArgs ga{depth_r, acc, workspace + p.geo_part, workspace + p.geo_raw, 3.0f * (float)num_scales * geo_weight};
        colvo::launch((k_warp_loss_march_levels<true>), dim3(grid), dim3(NT), 0, s, ml, pose, K, lcc_a, lcc_b, B, ssim_weight, ga);
    } else {
        colvo::launch((k_warp_loss_march_levels<false>), dim3(grid), dim3(NT), 0, s, ml, pose, K, lcc_a, lcc_b, B, ssim_weight,
                           GeoArgs{});
    }
    COLVO_CHECK_LAUNCH("k_warp_loss_march_levels");
    // (3) every term's value, the total, the per-image gradient sums
    colvo::launch(k_full_finalize, dim3(B * num_scales + 1), dim3(FT), 0, s, lv, geo ? workspace + p.geo_part : (const float*)nullptr,
                       smooth ? workspace + p.sm_part : (const float*)nullptr, p.sm_nblk, inv_nx, inv_ny, geo_weight, smooth_weight,
                       workspace + p.gpart, workspace + p.state, loss);
    COLVO_CHECK_LAUNCH("k_full_finalize");
    return 0;
}

extern "C" int colvo_full_objective_terms(const float* workspace, int B, int H, int W, int num_scales, const float** state) {
    COLVO_CHECK_ARG(workspace && state && full_shape_ok(B, H, W, num_scales), "colvo_full_objective_terms: bad argument");
    *state = workspace + full_plan(B, H, W, num_scales).state;
    return 0;
}

extern "C" int colvo_full_objective_bwd(const float* workspace, const float* grad_loss, const float* pose, int B, int H, int W,
                                        int num_scales, float geo_weight, float smooth_weight, float* d_depth_t,
                                        float* d_depth_r, float* d_pose, float* d_a, float* d_b, colvo_stream_t stream) {
    COLVO_CHECK_ARG(workspace && grad_loss && pose && d_depth_t && d_pose && d_a && d_b,
                    "colvo_full_objective_bwd: null pointer argument");
    COLVO_CHECK_ARG(geo_weight == 0.0f || d_depth_r, "colvo_full_objective_bwd: the geometric term needs d_depth_r");
    COLVO_CHECK_ARG(full_shape_ok(B, H, W, num_scales), "colvo_full_objective_bwd: bad shape");
    const FullPlan p = full_plan(B, H, W, num_scales);
    FullLevels lv{};
    lv.S = num_scales; lv.B = B;
    for (int l = 0; l < num_scales; ++l) {
        lv.items_per_image[l] = p.nseg[l] * p.strips[l];
        lv.partials[l] = workspace + p.part[l];
        lv.raw[l] = workspace + p.raw[l];
    }
    const bool geo = geo_weight != 0.0f, smooth = smooth_weight != 0.0f;
    unsigned nbx = (unsigned)(((size_t)H * W + NT - 1) / NT);
    if (nbx < (unsigned)B) nbx = (unsigned)B;
    colvo::launch(k_full_combine, dim3(nbx, B + 1), dim3(NT), 0, (hipStream_t)stream, lv, workspace + p.state, grad_loss,
                       smooth ? workspace + p.sd : (const float*)nullptr, geo ? workspace + p.geo_raw : (const float*)nullptr,
                       geo ? reinterpret_cast<const unsigned long long*>(workspace + p.acc) : (const unsigned long long*)nullptr,
                       workspace + p.gpart, pose, H, W, smooth_weight, d_depth_t, geo ? d_depth_r : (float*)nullptr, d_pose, d_a, d_b);
    COLVO_CHECK_LAUNCH("k_full_combine");
    return 0;
}
