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
// Work decomposition: one 256-thread workgroup per 64x16 output tile of one image.  The
// recalibrated warp J = a*warp(ref)+b and the target T are evaluated once per tile slot (tile +
// 1-px halo forward, 2-px halo backward; halo slots beyond the image border hold the REFLECTED
// pixel, which is how the reflection pad of the SSIM window is realised) and staged in LDS; the
// 3x3 SSIM statistics then come from LDS with a sliding window (forward) or per window centre
// (backward).  HBM reads are row-contiguous per plane (lanes = consecutive columns); the 4-tap
// gather of `ref` is served by L1/L2 for smooth flows.
#include <type_traits>

#include "common.h"

namespace colvo {
namespace {

constexpr int TW = 64;   // tile width  (= wave width: one lane per column)
constexpr int TH = 16;   // tile height (4 row-groups of 4 rows)
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
__device__ __forceinline__ void geo_compute(const float* pose, const float* K, const float* la,
                                            const float* lb, int b, float* s) {
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
    s[12] = k[0]; s[13] = k[4]; s[14] = k[2]; s[15] = k[5];
    s[16] = 1.0f / k[0]; s[17] = 1.0f / k[4];
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
    int plane4;     // bytes per plane
};
__device__ __forceinline__ Img img_make(const float* tgt, const float* ref, const float* depth, int b, int H, int W) {
    Img im;
    const size_t plane = (size_t)H * W;
    im.plane4 = (int)(plane * 4);
    im.ref = __builtin_amdgcn_make_buffer_rsrc((void*)(ref + (size_t)b * 3 * plane), 0, 3 * im.plane4, 0x00020000);
    im.tgt = __builtin_amdgcn_make_buffer_rsrc((void*)(tgt + (size_t)b * 3 * plane), 0, 3 * im.plane4, 0x00020000);
    im.dep = __builtin_amdgcn_make_buffer_rsrc((void*)(depth + (size_t)b * plane), 0, im.plane4, 0x00020000);
    return im;
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
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

// a4 + a5 for one pixel: J[c] = a * bilinear(ref_c) + b  (warp = 0 where invalid)
template <bool WITH_GRAD>
__device__ __forceinline__ void sample_px(const Geo& g, const Img& im, const Proj& p, int H, int W, float J[3],
                                          float Wp[3], float gx[3], float gy[3]) {
    if (p.valid) {
        const Taps t = make_taps(p, H, W);
        const float ux = 1.0f - t.wx, uy = 1.0f - t.wy;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int so = c * im.plane4;
            const float i00 = bload(im.ref, t.o00, so), i01 = bload(im.ref, t.o01, so);
            const float i10 = bload(im.ref, t.o10, so), i11 = bload(im.ref, t.o11, so);
            const float top = fmaf(i01, t.wx, i00 * ux);
            const float bot = fmaf(i11, t.wx, i10 * ux);
            const float w = fmaf(bot, t.wy, top * uy);
            Wp[c] = w;
            J[c] = fmaf(g.a, w, g.b);
            if (WITH_GRAD) {
                gx[c] = fmaf(t.wy, i11 - i10, uy * (i01 - i00));
                gy[c] = fmaf(t.wx, i11 - i01, ux * (i10 - i00));
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Wp[c] = 0.0f;
            J[c] = g.b;
            if (WITH_GRAD) { gx[c] = 0.0f; gy[c] = 0.0f; }
        }
    }
}

// Branch-free variant for batched evaluation: invalid points read tap (0,0) with zero weights.
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

// Evaluate N slots with every global load of the batch in flight together: depth + target first (they
// depend on nothing), then all 12 N reference taps, then the blends.  No branches, so the compiler keeps
// it one basic block and the two dependent memory round trips are paid once per batch, not once per slot.
template <int N, bool WITH_GRAD>
__device__ __forceinline__ void eval_batch(const Geo& g, const Img& im, const int (&px)[N], const int (&py)[N], int H,
                                           int W, Proj (&p)[N], float (&dv)[N], float (&J)[N][3], float (&T)[N][3],
                                           float (&Wp)[N][3], float (&gx)[N][3], float (&gy)[N][3]) {
    int o4[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        o4[n] = (py[n] * W + px[n]) * 4;
        dv[n] = bload(im.dep, o4[n], 0);
    }
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[n][c] = bload(im.tgt, o4[n], c * im.plane4);
    Taps t[N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        p[n] = project_px(g, dv[n], px[n], py[n], H, W);
        t[n] = make_taps_safe(p[n], H, W);
    }
    float v[N][3][4];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int so = c * im.plane4;
            v[n][c][0] = bload(im.ref, t[n].o00, so); v[n][c][1] = bload(im.ref, t[n].o01, so);
            v[n][c][2] = bload(im.ref, t[n].o10, so); v[n][c][3] = bload(im.ref, t[n].o11, so);
        }
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float m = p[n].valid ? 1.0f : 0.0f;
        const float ux = 1.0f - t[n].wx, uy = 1.0f - t[n].wy;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float i00 = v[n][c][0], i01 = v[n][c][1], i10 = v[n][c][2], i11 = v[n][c][3];
            const float top = fmaf(i01, t[n].wx, i00 * ux);
            const float bot = fmaf(i11, t[n].wx, i10 * ux);
            const float w = m * fmaf(bot, t[n].wy, top * uy);
            Wp[n][c] = w;
            J[n][c] = fmaf(g.a, w, g.b);
            if (WITH_GRAD) {
                gx[n][c] = m * fmaf(t[n].wy, i11 - i10, uy * (i01 - i00));
                gy[n][c] = m * fmaf(t[n].wx, i11 - i01, ux * (i10 - i00));
            }
        }
    }
}

// The same evaluation split into its three dependency stages, so that a persistent kernel can keep the
// loads of the NEXT tile in flight while it computes on the current one.
template <int N>
struct Batch {
    float dv[N];
    float T[N][3];
    float wx[N], wy[N], m[N];
    float v[N][3][4];
};
template <int N>   // stage A: depth + target loads (depend on nothing)
__device__ __forceinline__ void batch_issue_dt(const Img& im, const int (&px)[N], const int (&py)[N], int W, Batch<N>& B) {
#pragma unroll
    for (int n = 0; n < N; ++n) B.dv[n] = bload(im.dep, (py[n] * W + px[n]) * 4, 0);
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int c = 0; c < 3; ++c) B.T[n][c] = bload(im.tgt, (py[n] * W + px[n]) * 4, c * im.plane4);
}
template <int N>   // stage B: project (needs depth) and issue the 12 N reference taps
__device__ __forceinline__ void batch_issue_taps(const Geo& g, const Img& im, const int (&px)[N], const int (&py)[N],
                                                 int H, int W, Batch<N>& B) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const Proj p = project_px(g, B.dv[n], px[n], py[n], H, W);
        const Taps t = make_taps_safe(p, H, W);
        B.wx[n] = t.wx; B.wy[n] = t.wy; B.m[n] = p.valid ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int so = c * im.plane4;
            B.v[n][c][0] = bload(im.ref, t.o00, so); B.v[n][c][1] = bload(im.ref, t.o01, so);
            B.v[n][c][2] = bload(im.ref, t.o10, so); B.v[n][c][3] = bload(im.ref, t.o11, so);
        }
    }
}
template <int N>   // stage C: blend + LCC
__device__ __forceinline__ void batch_blend(const Geo& g, const Batch<N>& B, float (&J)[N][3]) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float ux = 1.0f - B.wx[n], uy = 1.0f - B.wy[n];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = fmaf(B.v[n][c][1], B.wx[n], B.v[n][c][0] * ux);
            const float bot = fmaf(B.v[n][c][3], B.wx[n], B.v[n][c][2] * ux);
            J[n][c] = fmaf(g.a, B.m[n] * fmaf(bot, B.wy[n], top * uy), g.b);
        }
    }
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
// forward                                                                                        //
// --------------------------------------------------------------------------------------------- //
constexpr int FSW = TW + 2, FSH = TH + 2;   // slots incl. 1-px halo

// Persistent over a strip of tiles of ONE image (grid = strips x 1 x B): the masked loss sum is carried in
// registers across the strip and reduced once per workgroup (one partial per strip, fixed order).
__global__ __launch_bounds__(NT, 4) void k_warp_loss_fwd(
    const float* __restrict__ tgt, const float* __restrict__ ref, const float* __restrict__ depth,
    const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ lcc_a,
    const float* __restrict__ lcc_b, int H, int W, int tiles_x, int tiles_y, float alpha,
    float* __restrict__ partials) {
    __shared__ float sJ[3][FSH][FSW];
    __shared__ float sT[3][FSH][FSW];
    __shared__ float s_geo[GEO_N + 2];
    __shared__ float s_red[8];

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int ntiles = tiles_x * tiles_y;
    const int per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int t0 = blockIdx.x * per, t1 = min(ntiles, t0 + per);
    if (tid == 0) geo_compute(pose, K, lcc_a, lcc_b, b, s_geo);
    __syncthreads();
    const Geo g = geo_load(s_geo);
    const Img im = img_make(tgt, ref, depth, b, H, W);

    const int col = tid & 63, rg = tid >> 6;
    // slots of this thread inside a tile: 4 owned + one of the 1-px ring (first 164 threads)
    int rsy = 4 * rg + 1, rsx = col + 1;      // dummy = first owned slot for threads >= 164
    if (tid < FSW) { rsy = 0; rsx = tid; }
    else if (tid < 2 * FSW) { rsy = FSH - 1; rsx = tid - FSW; }
    else if (tid < 2 * FSW + TH) { rsy = 1 + (tid - 2 * FSW); rsx = 0; }
    else if (tid < 2 * FSW + 2 * TH) { rsy = 1 + (tid - 2 * FSW - TH); rsx = FSW - 1; }

    float acc = 0.0f, cnt = 0.0f;
#pragma unroll 1
    for (int t = t0; t < t1; ++t) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int x0 = tx * TW, y0 = ty * TH;
        float maskv[4];
        // phase 1: evaluate the slots (loads batched); slots that leave the image hold the reflected pixel
        auto phase1 = [&](auto nslots) {
            constexpr int N = decltype(nslots)::value;
            int px[N], py[N], ssy[N], ssx[N];
#pragma unroll
            for (int n = 0; n < N; ++n) {
                ssy[n] = (n < 4) ? 4 * rg + n + 1 : rsy;
                ssx[n] = (n < 4) ? col + 1 : rsx;
                py[n] = reflect_idx(y0 + ssy[n] - 1, H);
                px[n] = reflect_idx(x0 + ssx[n] - 1, W);
            }
            Proj p[N];
            float dv[N], J[N][3], T[N][3], Wp[N][3], dumx[N][3], dumy[N][3];
            eval_batch<N, false>(g, im, px, py, H, W, p, dv, J, T, Wp, dumx, dumy);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (n < 4) maskv[n] = (p[n].valid && (y0 + ssy[n] - 1) < H && (x0 + ssx[n] - 1) < W) ? 1.0f : 0.0f;
                if (n < 4 || tid < 2 * FSW + 2 * TH) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        sJ[c][ssy[n]][ssx[n]] = J[n][c];
                        sT[c][ssy[n]][ssx[n]] = T[n][c];
                    }
                }
            }
        };
        if (tid < 192) phase1(std::integral_constant<int, 5>{});   // waves 0..2 carry the ring (wave-uniform)
        else phase1(std::integral_constant<int, 4>{});
        __syncthreads();

        // phase 2: per channel, the 6 slot rows x 3 columns this thread's 4 output rows need are read from LDS
        // up front (one latency per channel), then the sliding 3x3 sums run from registers.
        float m4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float jv[6][3], tv[6][3];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    jv[j][k] = sJ[c][4 * rg + j][col + k];
                    tv[j][k] = sT[c][4 * rg + j][col + k];
                }
            float hx[6], hy[6], hxx[6], hyy[6], hxy[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                hx[j] = tv[j][0] + tv[j][1] + tv[j][2];
                hy[j] = jv[j][0] + jv[j][1] + jv[j][2];
                hxx[j] = fmaf(tv[j][0], tv[j][0], fmaf(tv[j][1], tv[j][1], tv[j][2] * tv[j][2]));
                hyy[j] = fmaf(jv[j][0], jv[j][0], fmaf(jv[j][1], jv[j][1], jv[j][2] * jv[j][2]));
                hxy[j] = fmaf(tv[j][0], jv[j][0], fmaf(tv[j][1], jv[j][1], tv[j][2] * jv[j][2]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const SsimTerms s = ssim_terms(hx[i] + hx[i + 1] + hx[i + 2], hy[i] + hy[i + 1] + hy[i + 2],
                                               hxx[i] + hxx[i + 1] + hxx[i + 2], hyy[i] + hyy[i + 1] + hyy[i + 2],
                                               hxy[i] + hxy[i + 1] + hxy[i + 2]);
                const float S = (s.A1 * s.A2) * fast_rcp(s.B1 * s.B2);
                const float ss = fminf(fmaxf(0.5f * (1.0f - S), 0.0f), 1.0f);
                m4[i] += alpha * ss + (1.0f - alpha) * fabsf(tv[i + 1][1] - jv[i + 1][1]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc = fmaf(m4[i], maskv[i], acc); cnt += maskv[i]; }
        __syncthreads();     // LDS is rewritten by the next tile
    }

    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    if ((tid & 63) == 0) { s_red[2 * (tid >> 6)] = acc; s_red[2 * (tid >> 6) + 1] = cnt; }
    __syncthreads();
    if (tid == 0) {
        const size_t blk = (size_t)b * gridDim.x + blockIdx.x;
        partials[2 * blk] = (s_red[0] + s_red[2]) + (s_red[4] + s_red[6]);
        partials[2 * blk + 1] = (s_red[1] + s_red[3]) + (s_red[5] + s_red[7]);
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
        loss_state[3] = 0.0f;
    }
}

// --------------------------------------------------------------------------------------------- //
// backward                                                                                       //
// --------------------------------------------------------------------------------------------- //
// Per tile:  (1) J, T of tile + 2-px halo -> LDS;  per channel: (2) derivative coefficients of every
// SSIM window (tile + 1-px halo) from sliding sums -> LDS, (3) every pixel gathers, with the reflection
// multiplicities, from the 9 windows that contain it -> dJ in registers;  (4) the sample is RECOMPUTED
// (taps are L1/L2 hits) to chain dJ through LCC, the bilinear taps and the projection.  Only dJ[4][3]
// lives across the phases, which keeps the kernel at <= 128 VGPRs (2 workgroups per SIMD set).
constexpr int BSW = TW + 4, BSH = TH + 4;   // J/T slots incl. 2-px halo
constexpr int WSW = TW + 2, WSH = TH + 2;   // window centres incl. 1-px halo
constexpr int NPART = 14;                   // dt[3], dR[9], da, db

__global__ __launch_bounds__(NT, 3) void k_warp_loss_bwd(
    const float* __restrict__ tgt, const float* __restrict__ ref, const float* __restrict__ depth,
    const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ lcc_a,
    const float* __restrict__ lcc_b, int H, int W, float alpha, const float* __restrict__ loss_state,
    const float* __restrict__ grad_loss, float* __restrict__ d_depth, float* __restrict__ partials) {
    __shared__ float sJ[3][BSH][BSW];
    __shared__ float sT[3][BSH][BSW];
    __shared__ float sM[WSH][WSW];       // validity mask of each window centre (0 outside the image)
    __shared__ float sK[3][WSH][WSW];    // per-window derivative coefficients of the current channel
    __shared__ float s_geo[GEO_N + 2];

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (tid == 0) geo_compute(pose, K, lcc_a, lcc_b, b, s_geo);
    __syncthreads();
    const Geo g = geo_load(s_geo);
    const float gscale = grad_loss[0] * loss_state[1];   // dL/dloss / max(3 n_valid, 1)

    const Img im = img_make(tgt, ref, depth, b, H, W);
    const size_t plane = (size_t)H * W;

    // phase 1: every slot of tile + 2-px halo (reflected where it leaves the image): 1360 slots, each thread
    // takes 2 batches of 3 (loads of a batch in flight together); slot ids beyond the end are clamped and
    // simply rewrite the last slot with the same values.
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        int px[3], py[3], ssy[3], ssx[3], uy[3], ux[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int sl = min((it * 3 + n) * NT + tid, BSH * BSW - 1);
            ssy[n] = sl / BSW; ssx[n] = sl - ssy[n] * BSW;
            uy[n] = y0 + ssy[n] - 2; ux[n] = x0 + ssx[n] - 2;      // unreflected coordinate
            py[n] = reflect_idx(uy[n], H); px[n] = reflect_idx(ux[n], W);
        }
        Proj p[3];
        float dv[3], J[3][3], T[3][3], W3[3][3], dumx[3][3], dumy[3][3];
        eval_batch<3, false>(g, im, px, py, H, W, p, dv, J, T, W3, dumx, dumy);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sJ[c][ssy[n]][ssx[n]] = J[n][c];
                sT[c][ssy[n]][ssx[n]] = T[n][c];
            }
            const int wy = ssy[n] - 1, wx = ssx[n] - 1;
            if (wy >= 0 && wy < WSH && wx >= 0 && wx < WSW) {
                const bool exists = (uy[n] >= 0 && uy[n] < H && ux[n] >= 0 && ux[n] < W);
                sM[wy][wx] = (exists && p[n].valid) ? 1.0f : 0.0f;
            }
        }
    }
    __syncthreads();

    const int col = tid & 63, rg = tid >> 6;
    // reflection multiplicities of the owned pixels (shared by the 3 channels)
    float wyv[4][3], wxv[3];
    {
        const int px = x0 + col;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int q = px + d - 1;
            wxv[d] = (q >= 0 && q < W) ? window_mult(q, px, W) : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int py = y0 + 4 * rg + i;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int q = py + d - 1;
                wyv[i][d] = (q >= 0 && q < H) ? window_mult(q, py, H) : 0.0f;
            }
        }
    }

    const float kss = gscale * alpha * (-0.5f);
    const float kl1 = gscale * (1.0f - alpha);
    // phase-2 role: 3 groups of 66 threads, each thread a column of 6 windows
    const int wcol = tid % WSW, wgrp = tid / WSW;

    float dJ[4][3];
#pragma unroll 1   // rolled: keeps register pressure down; dJ is written through a uniform switch (static index)
    for (int c = 0; c < 3; ++c) {
        // phase 2: derivative coefficients of every window centre (tile + 1-px halo), sliding sums
        if (wgrp < 3) {
            float hx0 = 0.f, hx1 = 0.f, hy0 = 0.f, hy1 = 0.f, hxx0 = 0.f, hxx1 = 0.f, hyy0 = 0.f, hyy1 = 0.f,
                  hxy0 = 0.f, hxy1 = 0.f;
#pragma unroll 1
            for (int j = 0; j < 8; ++j) {
                const int sr = 6 * wgrp + j;
                const float j0 = sJ[c][sr][wcol], j1 = sJ[c][sr][wcol + 1], j2 = sJ[c][sr][wcol + 2];
                const float t0 = sT[c][sr][wcol], t1 = sT[c][sr][wcol + 1], t2 = sT[c][sr][wcol + 2];
                const float hx2 = t0 + t1 + t2;
                const float hy2 = j0 + j1 + j2;
                const float hxx2 = t0 * t0 + t1 * t1 + t2 * t2;
                const float hyy2 = j0 * j0 + j1 * j1 + j2 * j2;
                const float hxy2 = t0 * j0 + t1 * j1 + t2 * j2;
                if (j >= 2) {
                    const int wy = sr - 2;
                    float A = 0.f, Bc = 0.f, Cc = 0.f;
                    if (sM[wy][wcol] != 0.0f) {
                        const SsimTerms s = ssim_terms(hx0 + hx1 + hx2, hy0 + hy1 + hy2, hxx0 + hxx1 + hxx2,
                                                       hyy0 + hyy1 + hyy2, hxy0 + hxy1 + hxy2);
                        // refined reciprocal: d_a / d_b / d_pose are heavily cancelling sums of these terms
                        const float inv = nr_rcp(s.B1 * s.B2);
                        const float S = s.A1 * s.A2 * inv;
                        const float ss = 0.5f * (1.0f - S);
                        if (ss > 0.0f && ss < 1.0f) {
                            // d S / d(sum J), d(sum J^2), d(sum T J) of the window
                            const float dS_dsy = 2.0f * (s.sx * (s.A2 - s.A1) - S * s.sy * (s.B2 - s.B1)) * inv;
                            A = kss * dS_dsy;
                            Bc = kss * -18.0f * S * (inv * s.B1);
                            Cc = kss * 18.0f * s.A1 * inv;
                        }
                    }
                    sK[0][wy][wcol] = A; sK[1][wy][wcol] = Bc; sK[2][wy][wcol] = Cc;
                }
                hx0 = hx1; hx1 = hx2; hy0 = hy1; hy1 = hy2; hxx0 = hxx1; hxx1 = hxx2;
                hyy0 = hyy1; hyy1 = hyy2; hxy0 = hxy1; hxy1 = hxy2;
            }
        }
        __syncthreads();
        // phase 3: every owned pixel gathers from the 9 windows that contain it (rows 4rg..4rg+5 of
        // the window grid serve the thread's 4 pixels): horizontal weighted sums first, then vertical
        {
            float ha[6], hb[6], hc[6], v[4];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int wr = 4 * rg + j;
                ha[j] = wxv[0] * sK[0][wr][col] + wxv[1] * sK[0][wr][col + 1] + wxv[2] * sK[0][wr][col + 2];
                hb[j] = wxv[0] * sK[1][wr][col] + wxv[1] * sK[1][wr][col + 1] + wxv[2] * sK[1][wr][col + 2];
                hc[j] = wxv[0] * sK[2][wr][col] + wxv[1] * sK[2][wr][col + 1] + wxv[2] * sK[2][wr][col + 2];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * rg + i;
                const float sa = wyv[i][0] * ha[i] + wyv[i][1] * ha[i + 1] + wyv[i][2] * ha[i + 2];
                const float sb = wyv[i][0] * hb[i] + wyv[i][1] * hb[i + 1] + wyv[i][2] * hb[i + 2];
                const float sc = wyv[i][0] * hc[i] + wyv[i][1] * hc[i + 1] + wyv[i][2] * hc[i + 2];
                const float j = sJ[c][row + 2][col + 2], t = sT[c][row + 2][col + 2];
                const float diff = j - t;
                const float sgn = (diff > 0.0f) ? 1.0f : ((diff < 0.0f) ? -1.0f : 0.0f);
                v[i] = sa + sb * j + sc * t + kl1 * sM[row + 1][col + 1] * sgn;
            }
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dJ[i][0] = v[i];
            } else if (c == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dJ[i][1] = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) dJ[i][2] = v[i];
            }
        }
        __syncthreads();
    }

    // phase 4: recompute the sample and chain through LCC, the bilinear taps and the projection
    float part[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) part[k] = 0.0f;
    {
        int px[4], py[4];
        bool own[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gy_ = y0 + 4 * rg + i, gx_ = x0 + col;
            own[i] = (gy_ < H && gx_ < W);            // overhanging slots own no output
            py[i] = min(gy_, H - 1); px[i] = min(gx_, W - 1);
        }
        Proj p[4];
        float dval[4], J[4][3], T[4][3], Wp[4][3], gx[4][3], gy[4][3];
        eval_batch<4, true>(g, im, px, py, H, W, p, dval, J, T, Wp, gx, gy);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!own[i]) continue;
            float da = 0.f, db = 0.f, gxs = 0.f, gys = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                da += dJ[i][c] * Wp[i][c];
                db += dJ[i][c];
                const float dW = g.a * dJ[i][c];
                gxs += dW * gx[i][c];
                gys += dW * gy[i][c];
            }
            part[12] += da;
            part[13] += db;
            float dd = 0.0f;
            if (p[i].valid) {
                const float iz = nr_rcp(p[i].Pz);
                const float dPx = gxs * g.fx * iz;
                const float dPy = gys * g.fy * iz;
                const float dPz = -(dPx * p[i].Px + dPy * p[i].Py) * iz;
                // P = R (d * [Xh Yh 1]) + t
                const float rx_ = g.r00 * p[i].Xh + g.r01 * p[i].Yh + g.r02;
                const float ry_ = g.r10 * p[i].Xh + g.r11 * p[i].Yh + g.r12;
                const float rz_ = g.r20 * p[i].Xh + g.r21 * p[i].Yh + g.r22;
                dd = dPx * rx_ + dPy * ry_ + dPz * rz_;
                const float cX = p[i].Xh * dval[i], cY = p[i].Yh * dval[i], cZ = dval[i];
                part[0] += dPx; part[1] += dPy; part[2] += dPz;
                part[3] += dPx * cX; part[4] += dPx * cY; part[5] += dPx * cZ;
                part[6] += dPy * cX; part[7] += dPy * cY; part[8] += dPy * cZ;
                part[9] += dPz * cX; part[10] += dPz * cY; part[11] += dPz * cZ;
            }
            d_depth[(size_t)b * plane + (size_t)py[i] * W + px[i]] = dd;
        }
    }
    // block reduction of the 14 partial sums through LDS (the staging arrays are free now): two fixed-order
    // stages instead of 14 x 6 cross-lane shuffles per thread
    float* sRed = &sJ[0][0][0];          // [NPART][NT]
    float* sRed2 = &sT[0][0][0];         // [NPART][16]
#pragma unroll
    for (int k = 0; k < NPART; ++k) sRed[k * NT + tid] = part[k];
    __syncthreads();
    if (tid < NPART * 16) {
        const int k = tid >> 4, seg = tid & 15;
        const float* r = sRed + k * NT + seg * 16;
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += r[i];
        sRed2[tid] = a;
    }
    __syncthreads();
    if (tid < NPART) {
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += sRed2[tid * 16 + i];
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[blk * NPART + tid] = a;
    }
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

inline int tiles_x(int W) { return (W + TW - 1) / TW; }
inline int tiles_y(int H) { return (H + TH - 1) / TH; }

}  // namespace
}  // namespace colvo

using namespace colvo;

extern "C" size_t colvo_warp_loss_workspace_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * tiles_x(W) * tiles_y(H) * NPART;
}

extern "C" int colvo_warp_loss_fwd(const float* tgt, const float* ref, const float* depth, const float* pose,
                                   const float* K, const float* lcc_a, const float* lcc_b, int B, int H, int W,
                                   float ssim_weight, float* workspace, float* loss_state, colvo_stream_t stream) {
    COLVO_CHECK_ARG(tgt && ref && depth && pose && K && lcc_a && lcc_b && workspace && loss_state,
                    "colvo_warp_loss_fwd: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && H >= 2 && W >= 2 && B <= 65535, "colvo_warp_loss_fwd: bad shape B=%d H=%d W=%d", B, H, W);
    COLVO_CHECK_ARG((size_t)H * W < (1u << 30), "colvo_warp_loss_fwd: image too large");
    hipStream_t s = (hipStream_t)stream;
    const int tx = tiles_x(W), ty = tiles_y(H);
    // persistent strips: ~4 workgroups per CU over the whole batch, each a contiguous run of tiles of one image
    int strips = (4 * 256 + B - 1) / B;
    if (strips > tx * ty) strips = tx * ty;
    if (strips < 1) strips = 1;
    hipLaunchKernelGGL(k_warp_loss_fwd, dim3(strips, 1, B), dim3(NT), 0, s, tgt, ref, depth, pose, K, lcc_a, lcc_b, H, W,
                       tx, ty, ssim_weight, workspace);
    COLVO_CHECK_LAUNCH("k_warp_loss_fwd");
    const int nblk = strips * B;
    hipLaunchKernelGGL(k_warp_loss_fwd_finalize, dim3(1), dim3(NT), 0, s, workspace, nblk, loss_state);
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
    dim3 grid(tiles_x(W), tiles_y(H), B);
    COLVO_CHECK_ARG(grid.y <= 65535, "colvo_warp_loss_bwd: H too large");
    hipLaunchKernelGGL(k_warp_loss_bwd, grid, dim3(NT), 0, s, tgt, ref, depth, pose, K, lcc_a, lcc_b, H, W,
                       ssim_weight, loss_state, grad_loss, d_depth, workspace);
    COLVO_CHECK_LAUNCH("k_warp_loss_bwd");
    hipLaunchKernelGGL(k_warp_loss_bwd_finalize, dim3(B), dim3(NT), 0, s, workspace, (int)(grid.x * grid.y), pose,
                       d_pose, d_a, d_b);
    COLVO_CHECK_LAUNCH("k_warp_loss_bwd_finalize");
    return 0;
}

extern "C" int colvo_inverse_warp(const float* ref, const float* depth, const float* pose, const float* K, int B,
                                  int C, int H, int W, float* warped, float* valid, colvo_stream_t stream) {
    COLVO_CHECK_ARG(ref && depth && pose && K && warped && valid, "colvo_inverse_warp: null pointer argument");
    COLVO_CHECK_ARG(B >= 1 && C >= 1 && H >= 1 && W >= 1 && B <= 65535, "colvo_inverse_warp: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const size_t plane = (size_t)H * W;
    dim3 grid((unsigned)((plane + NT - 1) / NT), B);
    hipLaunchKernelGGL(k_inverse_warp, grid, dim3(NT), 0, s, ref, depth, pose, K, C, H, W, warped, valid);
    COLVO_CHECK_LAUNCH("k_inverse_warp");
    return 0;
}
