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
    float a, b;
};
constexpr int GEO_N = 18;

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
    s[16] = la ? la[b] : 1.0f;
    s[17] = lb ? lb[b] : 0.0f;
}

__device__ __forceinline__ Geo geo_load(const float* s) {
    Geo g;
    g.r00 = uniform_f(s[0]); g.r01 = uniform_f(s[1]); g.r02 = uniform_f(s[2]);
    g.r10 = uniform_f(s[3]); g.r11 = uniform_f(s[4]); g.r12 = uniform_f(s[5]);
    g.r20 = uniform_f(s[6]); g.r21 = uniform_f(s[7]); g.r22 = uniform_f(s[8]);
    g.tx = uniform_f(s[9]); g.ty = uniform_f(s[10]); g.tz = uniform_f(s[11]);
    g.fx = uniform_f(s[12]); g.fy = uniform_f(s[13]); g.cx = uniform_f(s[14]); g.cy = uniform_f(s[15]);
    g.a = uniform_f(s[16]); g.b = uniform_f(s[17]);
    return g;
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

struct Proj {
    float x, y;        // sample position in ref (pixels)
    float Xh, Yh;      // K^-1 [u v 1] (x, y components)
    float Px, Py, Pz;  // point in the reference camera
    bool valid;
};

// a3: back-project, rigid transform, project.  Same operation order as oracle project().
__device__ __forceinline__ Proj project_px(const Geo& g, float d, int u, int v, int H, int W) {
#pragma clang fp contract(off)
    Proj o;
    o.Xh = ((float)u - g.cx) / g.fx;
    o.Yh = ((float)v - g.cy) / g.fy;
    const float X = o.Xh * d, Y = o.Yh * d, Z = d;
    o.Px = g.r00 * X + g.r01 * Y + g.r02 * Z + g.tx;
    o.Py = g.r10 * X + g.r11 * Y + g.r12 * Z + g.ty;
    o.Pz = g.r20 * X + g.r21 * Y + g.r22 * Z + g.tz;
    const bool front = o.Pz > Z_EPS;
    const float pzs = front ? o.Pz : 1.0f;
    o.x = g.fx * o.Px / pzs + g.cx;
    o.y = g.fy * o.Py / pzs + g.cy;
    o.valid = front && (o.x >= 0.0f) && (o.x <= (float)(W - 1)) && (o.y >= 0.0f) && (o.y <= (float)(H - 1));
    return o;
}

struct Taps {
    int o00, o01, o10, o11;
    float wx, wy;
};

__device__ __forceinline__ Taps make_taps(const Proj& p, int H, int W) {
    Taps t;
    const float xs = fminf(fmaxf(p.x, 0.0f), (float)(W - 1));
    const float ys = fminf(fmaxf(p.y, 0.0f), (float)(H - 1));
    const float x0f = floorf(xs), y0f = floorf(ys);
    t.wx = xs - x0f;
    t.wy = ys - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    t.o00 = y0 * W + x0; t.o01 = y0 * W + x1;
    t.o10 = y1 * W + x0; t.o11 = y1 * W + x1;
    return t;
}

// a4 + a5 for one pixel: J[c] = a * bilinear(ref_c) + b  (warp = 0 where invalid)
template <bool WITH_GRAD>
__device__ __forceinline__ void sample_px(const Geo& g, const float* __restrict__ refb, size_t plane,
                                          const Proj& p, int H, int W, float J[3], float Wp[3],
                                          float gx[3], float gy[3]) {
    if (p.valid) {
        const Taps t = make_taps(p, H, W);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* r = refb + c * plane;
            const float i00 = r[t.o00], i01 = r[t.o01], i10 = r[t.o10], i11 = r[t.o11];
            const float top = i00 * (1.0f - t.wx) + i01 * t.wx;
            const float bot = i10 * (1.0f - t.wx) + i11 * t.wx;
            const float w = top * (1.0f - t.wy) + bot * t.wy;
            Wp[c] = w;
            J[c] = g.a * w + g.b;
            if (WITH_GRAD) {
                gx[c] = (1.0f - t.wy) * (i01 - i00) + t.wy * (i11 - i10);
                gy[c] = (1.0f - t.wx) * (i10 - i00) + t.wx * (i11 - i01);
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

// SSIM numerator/denominator pieces from the five window sums of one channel (x = target, y = J)
struct SsimTerms {
    float mx, my, A1, A2, B1, B2;
};
__device__ __forceinline__ SsimTerms ssim_terms(float sx, float sy, float sxx, float syy, float sxy) {
    SsimTerms s;
    const float inv9 = 1.0f / 9.0f;
    s.mx = sx * inv9;
    s.my = sy * inv9;
    const float vx = sxx * inv9 - s.mx * s.mx;
    const float vy = syy * inv9 - s.my * s.my;
    const float cxy = sxy * inv9 - s.mx * s.my;
    s.A1 = 2.0f * s.mx * s.my + SSIM_C1;
    s.A2 = 2.0f * cxy + SSIM_C2;
    s.B1 = s.mx * s.mx + s.my * s.my + SSIM_C1;
    s.B2 = vx + vy + SSIM_C2;
    return s;
}

// --------------------------------------------------------------------------------------------- //
// forward                                                                                        //
// --------------------------------------------------------------------------------------------- //
constexpr int FSW = TW + 2, FSH = TH + 2;   // slots incl. 1-px halo

__global__ __launch_bounds__(NT) void k_warp_loss_fwd(
    const float* __restrict__ tgt, const float* __restrict__ ref, const float* __restrict__ depth,
    const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ lcc_a,
    const float* __restrict__ lcc_b, int H, int W, float alpha, float* __restrict__ partials) {
    __shared__ float sJ[3][FSH][FSW];
    __shared__ float sT[3][FSH][FSW];
    __shared__ float s_geo[GEO_N + 2];
    __shared__ float s_red[8];

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (tid == 0) geo_compute(pose, K, lcc_a, lcc_b, b, s_geo);
    __syncthreads();
    const Geo g = geo_load(s_geo);

    const size_t plane = (size_t)H * W;
    const float* tgtb = tgt + (size_t)b * 3 * plane;
    const float* refb = ref + (size_t)b * 3 * plane;
    const float* depb = depth + (size_t)b * plane;

    const int col = tid & 63, rg = tid >> 6;
    float maskv[4];

    // phase 1a: the 4 pixels this thread owns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * rg + i;
        const int gy_ = y0 + row, gx_ = x0 + col;
        float J[3], T[3], Wp[3], dumx[3], dumy[3];
        {
            // a slot overhanging the image is the reflection pad of the last row / column
            const int py = reflect_idx(gy_, H), px = reflect_idx(gx_, W);
            const size_t o = (size_t)py * W + px;
            const Proj p = project_px(g, depb[o], px, py, H, W);
            sample_px<false>(g, refb, plane, p, H, W, J, Wp, dumx, dumy);
            T[0] = tgtb[o]; T[1] = tgtb[plane + o]; T[2] = tgtb[2 * plane + o];
            maskv[i] = (p.valid && gy_ < H && gx_ < W) ? 1.0f : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sJ[c][row + 1][col + 1] = J[c];
            sT[c][row + 1][col + 1] = T[c];
        }
    }
    // phase 1b: the 1-px ring (reflected where it leaves the image)
    if (tid < 2 * FSW + 2 * TH) {
        int sy, sx;
        if (tid < FSW) { sy = 0; sx = tid; }
        else if (tid < 2 * FSW) { sy = FSH - 1; sx = tid - FSW; }
        else if (tid < 2 * FSW + TH) { sy = 1 + (tid - 2 * FSW); sx = 0; }
        else { sy = 1 + (tid - 2 * FSW - TH); sx = FSW - 1; }
        const int py = reflect_idx(y0 + sy - 1, H), px = reflect_idx(x0 + sx - 1, W);
        const size_t o = (size_t)py * W + px;
        float J[3], Wp[3], dumx[3], dumy[3];
        const Proj p = project_px(g, depb[o], px, py, H, W);
        sample_px<false>(g, refb, plane, p, H, W, J, Wp, dumx, dumy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sJ[c][sy][sx] = J[c];
            sT[c][sy][sx] = tgtb[c * plane + o];
        }
    }
    __syncthreads();

    // phase 2: sliding 3x3 window down the thread's 4 rows (6 slot rows), horizontal sums kept in
    // registers for three consecutive slot rows.
    float acc = 0.0f, cnt = 0.0f;
    float hx[3][3], hy[3][3], hxx[3][3], hyy[3][3], hxy[3][3];  // [slot-row mod 3][channel]
    float midJ[2][3], midT[2][3];                                // centre-column values of the last two rows
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int sr = 4 * rg + j;
        const int k = j % 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float j0 = sJ[c][sr][col], j1 = sJ[c][sr][col + 1], j2 = sJ[c][sr][col + 2];
            const float t0 = sT[c][sr][col], t1 = sT[c][sr][col + 1], t2 = sT[c][sr][col + 2];
            hx[k][c] = t0 + t1 + t2;
            hy[k][c] = j0 + j1 + j2;
            hxx[k][c] = t0 * t0 + t1 * t1 + t2 * t2;
            hyy[k][c] = j0 * j0 + j1 * j1 + j2 * j2;
            hxy[k][c] = t0 * j0 + t1 * j1 + t2 * j2;
            midJ[j & 1][c] = j1;
            midT[j & 1][c] = t1;
        }
        if (j >= 2) {
            const int i = j - 2;  // output row 4*rg + i, its centre is slot row sr-1
            float m = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const SsimTerms s = ssim_terms(hx[0][c] + hx[1][c] + hx[2][c], hy[0][c] + hy[1][c] + hy[2][c],
                                               hxx[0][c] + hxx[1][c] + hxx[2][c], hyy[0][c] + hyy[1][c] + hyy[2][c],
                                               hxy[0][c] + hxy[1][c] + hxy[2][c]);
                const float S = (s.A1 * s.A2) / (s.B1 * s.B2);
                const float ss = fminf(fmaxf(0.5f * (1.0f - S), 0.0f), 1.0f);
                const float l1 = fabsf(midT[(j - 1) & 1][c] - midJ[(j - 1) & 1][c]);
                m += alpha * ss + (1.0f - alpha) * l1;
            }
            acc += m * maskv[i];
            cnt += maskv[i];
        }
    }

    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    if ((tid & 63) == 0) { s_red[2 * (tid >> 6)] = acc; s_red[2 * (tid >> 6) + 1] = cnt; }
    __syncthreads();
    if (tid == 0) {
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
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
constexpr int BSW = TW + 4, BSH = TH + 4;   // J/T slots incl. 2-px halo
constexpr int WSW = TW + 2, WSH = TH + 2;   // window centres incl. 1-px halo
constexpr int NPART = 14;                   // dt[3], dR[9], da, db

__global__ __launch_bounds__(NT) void k_warp_loss_bwd(
    const float* __restrict__ tgt, const float* __restrict__ ref, const float* __restrict__ depth,
    const float* __restrict__ pose, const float* __restrict__ K, const float* __restrict__ lcc_a,
    const float* __restrict__ lcc_b, int H, int W, float alpha, const float* __restrict__ loss_state,
    const float* __restrict__ grad_loss, float* __restrict__ d_depth, float* __restrict__ partials) {
    __shared__ float sJ[3][BSH][BSW];
    __shared__ float sT[3][BSH][BSW];
    __shared__ float sM[WSH][WSW];       // validity mask of each window centre (0 outside the image)
    __shared__ float sK[3][WSH][WSW];    // per-window derivative coefficients of the current channel
    __shared__ float s_geo[GEO_N + 2];
    __shared__ float s_red[4][NPART + 2];

    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    if (tid == 0) geo_compute(pose, K, lcc_a, lcc_b, b, s_geo);
    __syncthreads();
    const Geo g = geo_load(s_geo);
    const float gscale = grad_loss[0] * loss_state[1];   // dL/dloss / max(3 n_valid, 1)

    const size_t plane = (size_t)H * W;
    const float* tgtb = tgt + (size_t)b * 3 * plane;
    const float* refb = ref + (size_t)b * 3 * plane;
    const float* depb = depth + (size_t)b * plane;

    const int col = tid & 63, rg = tid >> 6;

    // per owned pixel, kept in registers across the phases
    float Wp[4][3], gx[4][3], gy[4][3], dJ[4][3];
    Proj pj[4];
    float dval[4];
    bool inimg[4];

    // phase 1a: owned pixels (slot = pixel + 2)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * rg + i;
        const int gy_ = y0 + row, gx_ = x0 + col;
        float J[3], T[3];
        inimg[i] = (gy_ < H && gx_ < W);
        float m = 0.0f;
        {
            // a slot overhanging the image is the reflection pad of the last row / column: it holds
            // the reflected pixel's J and T but owns no output (valid = false, dJ = 0)
            const int py = reflect_idx(gy_, H), px = reflect_idx(gx_, W);
            const size_t o = (size_t)py * W + px;
            dval[i] = depb[o];
            pj[i] = project_px(g, dval[i], px, py, H, W);
            sample_px<true>(g, refb, plane, pj[i], H, W, J, Wp[i], gx[i], gy[i]);
            T[0] = tgtb[o]; T[1] = tgtb[plane + o]; T[2] = tgtb[2 * plane + o];
            if (!inimg[i]) pj[i].valid = false;
            m = pj[i].valid ? 1.0f : 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) dJ[i][c] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sJ[c][row + 2][col + 2] = J[c];
            sT[c][row + 2][col + 2] = T[c];
        }
        sM[row + 1][col + 1] = m;
    }
    // phase 1b: the 2-px ring: 2*BSW top + 2*BSW bottom + 4*TH sides = 336 slots
    for (int r = tid; r < 4 * BSW + 4 * TH; r += NT) {
        int sy, sx;
        if (r < 2 * BSW) { sy = r / BSW; sx = r % BSW; }
        else if (r < 4 * BSW) { const int q = r - 2 * BSW; sy = TH + 2 + q / BSW; sx = q % BSW; }
        else { const int q = r - 4 * BSW; sy = 2 + (q >> 2); const int k = q & 3; sx = (k < 2) ? k : (TW + k); }
        const int uy = y0 + sy - 2, ux = x0 + sx - 2;   // unreflected coordinate
        const int py = reflect_idx(uy, H), px = reflect_idx(ux, W);
        const size_t o = (size_t)py * W + px;
        float J[3], W3[3], dumx[3], dumy[3];
        const Proj p = project_px(g, depb[o], px, py, H, W);
        sample_px<false>(g, refb, plane, p, H, W, J, W3, dumx, dumy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            sJ[c][sy][sx] = J[c];
            sT[c][sy][sx] = tgtb[c * plane + o];
        }
        // window-centre mask for ring slots that are window centres (inner ring of the 2-px halo)
        const int wy = sy - 1, wx = sx - 1;
        if (wy >= 0 && wy < WSH && wx >= 0 && wx < WSW) {
            const bool exists = (uy >= 0 && uy < H && ux >= 0 && ux < W);
            sM[wy][wx] = (exists && p.valid) ? 1.0f : 0.0f;
        }
    }
    __syncthreads();

    // window weights of the owned pixels (reflection multiplicities), shared by the 3 channels
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

    const float kss = gscale * alpha * (-0.5f) * (1.0f / 9.0f);
    const float kl1 = gscale * (1.0f - alpha);

#pragma unroll   // fully unrolled: dJ[i][c] must stay in registers (static index)
    for (int c = 0; c < 3; ++c) {
        // phase 2: derivative coefficients of every window centre (tile + 1-px halo)
        for (int w = tid; w < WSH * WSW; w += NT) {
            const int wy = w / WSW, wx = w - wy * WSW;
            float A = 0.f, Bc = 0.f, Cc = 0.f;
            const float m = sM[wy][wx];
            if (m != 0.0f) {
                float sx_ = 0.f, sy_ = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float t = sT[c][wy + dy][wx + dx], j = sJ[c][wy + dy][wx + dx];
                        sx_ += t; sy_ += j; sxx += t * t; syy += j * j; sxy += t * j;
                    }
                const SsimTerms s = ssim_terms(sx_, sy_, sxx, syy, sxy);
                const float inv = 1.0f / (s.B1 * s.B2);
                const float S = s.A1 * s.A2 * inv;
                const float ss = 0.5f * (1.0f - S);
                if (ss > 0.0f && ss < 1.0f) {
                    const float dS_dmy = (2.0f * s.mx * (s.A2 - s.A1) - S * 2.0f * s.my * (s.B2 - s.B1)) * inv;
                    const float dS_deyy = -S / s.B2;
                    const float dS_dexy = 2.0f * s.A1 * inv;
                    A = kss * dS_dmy;
                    Bc = kss * 2.0f * dS_deyy;
                    Cc = kss * dS_dexy;
                }
            }
            sK[0][wy][wx] = A; sK[1][wy][wx] = Bc; sK[2][wy][wx] = Cc;
        }
        __syncthreads();
        // phase 3: every owned pixel gathers from the 9 windows that contain it
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * rg + i;
            float sa = 0.f, sb = 0.f, sc = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float w = wyv[i][dy] * wxv[dx];
                    sa += w * sK[0][row + dy][col + dx];
                    sb += w * sK[1][row + dy][col + dx];
                    sc += w * sK[2][row + dy][col + dx];
                }
            const float j = sJ[c][row + 2][col + 2], t = sT[c][row + 2][col + 2];
            const float diff = j - t;
            const float sgn = (diff > 0.0f) ? 1.0f : ((diff < 0.0f) ? -1.0f : 0.0f);
            const float v = sa + sb * j + sc * t + kl1 * sM[row + 1][col + 1] * sgn;
            dJ[i][c] = inimg[i] ? v : 0.0f;
        }
        __syncthreads();
    }

    // phase 4: chain rule through LCC, the bilinear sample and the projection
    float part[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) part[k] = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gy_ = y0 + 4 * rg + i, gx_ = x0 + col;
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
        if (pj[i].valid) {
            const Proj& p = pj[i];
            const float iz = 1.0f / p.Pz;
            const float dPx = gxs * g.fx * iz;
            const float dPy = gys * g.fy * iz;
            const float dPz = -(gxs * g.fx * p.Px + gys * g.fy * p.Py) * iz * iz;
            // P = R (d * [Xh Yh 1]) + t
            const float rx_ = g.r00 * p.Xh + g.r01 * p.Yh + g.r02;
            const float ry_ = g.r10 * p.Xh + g.r11 * p.Yh + g.r12;
            const float rz_ = g.r20 * p.Xh + g.r21 * p.Yh + g.r22;
            dd = dPx * rx_ + dPy * ry_ + dPz * rz_;
            const float cX = p.Xh * dval[i], cY = p.Yh * dval[i], cZ = dval[i];
            part[0] += dPx; part[1] += dPy; part[2] += dPz;
            part[3] += dPx * cX; part[4] += dPx * cY; part[5] += dPx * cZ;
            part[6] += dPy * cX; part[7] += dPy * cY; part[8] += dPy * cZ;
            part[9] += dPz * cX; part[10] += dPz * cY; part[11] += dPz * cZ;
        }
        if (inimg[i]) d_depth[(size_t)b * plane + (size_t)gy_ * W + gx_] = dd;
    }
#pragma unroll
    for (int k = 0; k < NPART; ++k) {
        const float v = wave_sum(part[k]);
        if ((tid & 63) == 0) s_red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < NPART) {
        const size_t blk = ((size_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[blk * NPART + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
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
            const float* r = refb + c * plane;
            const float top = r[t.o00] * (1.0f - t.wx) + r[t.o01] * t.wx;
            const float bot = r[t.o10] * (1.0f - t.wx) + r[t.o11] * t.wx;
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
    dim3 grid(tiles_x(W), tiles_y(H), B);
    COLVO_CHECK_ARG(grid.y <= 65535, "colvo_warp_loss_fwd: H too large");
    hipLaunchKernelGGL(k_warp_loss_fwd, grid, dim3(NT), 0, s, tgt, ref, depth, pose, K, lcc_a, lcc_b, H, W,
                       ssim_weight, workspace);
    COLVO_CHECK_LAUNCH("k_warp_loss_fwd");
    const int nblk = (int)(grid.x * grid.y * grid.z);
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
