// misc.hip -- the memory-bound helpers around the conv stack:
//   layout packing (NCHW fp32 planes <-> NHWC feature maps, weight operand layouts),
//   the 1-channel DepthNet head (conv3x3 + sigmoid + disp->depth) and the PoseNet head
//   (1x1 conv + spatial mean + pose/LCC scaling), ReLU backward, and Adam over the flat arena (a8).
// Spec: oracle/colvo_spec.py (DepthNet.head / disp_to_depth, PoseNet.pred, ADAM_KW).
#include "common.h"
#include "tuning.h"

namespace colvo {
namespace {

constexpr int NT = 256;

template <int ES> struct Elem;
template <> struct Elem<4> {
    static __device__ __forceinline__ float ld(const void* p, size_t i) { return reinterpret_cast<const float*>(p)[i]; }
    static __device__ __forceinline__ void st(void* p, size_t i, float v) { reinterpret_cast<float*>(p)[i] = v; }
};
template <> struct Elem<2> {
    static __device__ __forceinline__ float ld(const void* p, size_t i) { return bf2f(reinterpret_cast<const uint16_t*>(p)[i]); }
    static __device__ __forceinline__ void st(void* p, size_t i, float v) { reinterpret_cast<uint16_t*>(p)[i] = f2bf(v); }
};

// ---------------------------------------------------------------- weights -------------------- //
template <int ES>
__global__ __launch_bounds__(NT) void k_pack_weights(const float* __restrict__ w, int Cout, int kk, int Cin,
                                                     void* __restrict__ w_fwd, void* __restrict__ w_bwd) {
    const size_t n = (size_t)Cout * kk * Cin;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % Cin);
    const int t = (int)((i / Cin) % kk);
    const int co = (int)(i / ((size_t)Cin * kk));
    const float v = w[i];
    if (w_fwd) Elem<ES>::st(w_fwd, i, v);
    if (w_bwd) Elem<ES>::st(w_bwd, ((size_t)c * kk + (kk - 1 - t)) * Cout + co, v);   // taps flipped
}

// One launch for all layers of a network: `tab` (device) holds, per layer, the element offsets of the master
// weights in the fp32 arena and of the two operand copies in their flat buffers, plus the first workgroup
// of the layer; every workgroup finds its layer by a scan of that small table.
struct PackEntry {
    long long w_off, fwd_off, bwd_off;
    int Cout, kk, Cin, blk_begin;
};
constexpr int PK_CO = 32, PK_C = 64;       // transpose tile: 32 output channels x 64 input channels of one tap
constexpr int ADAM_PLAIN_PER_WG = COLVO_ADAM_PLAIN_PER_WG;
template <int ES>
__global__ __launch_bounds__(NT) void k_pack_weights_multi(const float* __restrict__ master, const PackEntry* __restrict__ tab,
                                                           int nlayers, void* __restrict__ fwd, void* __restrict__ bwd) {
    // LDS-tiled transpose: reads (and the forward copy) run along Cin, the transposed copy is written along Cout --
    // the element-per-thread version scattered 2-byte stores at a stride of Cout and ran at ~1 TB/s
    __shared__ float tile[PK_CO][PK_C + 1];
    int l = 0;
    for (int i = 1; i < nlayers; ++i)
        if ((int)blockIdx.x >= tab[i].blk_begin) l = i;
    const PackEntry e = tab[l];
    const int nct = (e.Cin + PK_C - 1) / PK_C, ncot = (e.Cout + PK_CO - 1) / PK_CO;
    const int lb = blockIdx.x - e.blk_begin;
    const int t = lb / (ncot * nct), r = lb - t * (ncot * nct);
    const int cot = r / nct, ct = r - cot * nct;
    const int co0 = cot * PK_CO, c0 = ct * PK_C;
    const int tid = threadIdx.x;
    {
        const int cc = c0 + (tid & 63);
#pragma unroll
        for (int i = 0; i < PK_CO / 4; ++i) {
            const int row = (tid >> 6) + 4 * i, co = co0 + row;
            if (co < e.Cout && cc < e.Cin) {
                const size_t idx = ((size_t)co * e.kk + t) * e.Cin + cc;
                const float v = master[e.w_off + idx];
                tile[row][tid & 63] = v;
                if (fwd && e.fwd_off >= 0) Elem<ES>::st(fwd, e.fwd_off + idx, v);
            }
        }
    }
    __syncthreads();
    {
        const int co = co0 + (tid & 31);
#pragma unroll
        for (int i = 0; i < PK_C / 8; ++i) {
            const int col = (tid >> 5) + 8 * i, cc = c0 + col;
            if (co < e.Cout && cc < e.Cin)
                Elem<ES>::st(bwd, e.bwd_off + ((size_t)cc * e.kk + (e.kk - 1 - t)) * e.Cout + co, tile[tid & 31][col]);   // taps flipped
        }
    }
}

// ---------------------------------------------------------------- NCHW <-> NHWC -------------- //
struct Planes {
    const float* p[4];
    int c[4];
    int n;
};

template <int ES>
__global__ __launch_bounds__(NT) void k_pack_nchw(Planes src, int HW, int Cpad, void* __restrict__ dst) {
    const int b = blockIdx.y;
    const size_t pix = (size_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= (size_t)HW) return;
    int ch = 0;
    const size_t o = ((size_t)b * HW + pix) * Cpad;
    for (int s = 0; s < src.n; ++s)
        for (int c = 0; c < src.c[s]; ++c, ++ch)
            Elem<ES>::st(dst, o + ch, src.p[s][((size_t)b * src.c[s] + c) * HW + pix]);
    for (; ch < Cpad; ++ch) Elem<ES>::st(dst, o + ch, 0.0f);
}

// The stems (Cpad = 8): the pixel's 8 channels are gathered in registers and leave as ONE 16-byte (bf16) / two 16-byte (f32)
// stores; the element-wise version above issued eight 2-byte stores per pixel (15.6 us for DepthNet's 16 frames, 2.4 TB/s).
template <int ES>
__global__ __launch_bounds__(NT) void k_pack_nchw8(Planes src, int HW, void* __restrict__ dst) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const int b = blockIdx.y;
    const size_t pix = (size_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= (size_t)HW) return;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.0f;
    int ch = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (s >= src.n) break;
        const int cs = src.c[s];
        const float* base = src.p[s] + (size_t)b * cs * HW + pix;
#pragma unroll
        for (int c = 0; c < 4; ++c) {            // a stem source has at most 3 (image) or 1 (depth) channels
            if (c < cs) {
                const float x = base[(size_t)c * HW];
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k == ch + c) v[k] = x;
            }
        }
        ch += cs;
    }
    const size_t o = ((size_t)b * HW + pix) * 8;
    if constexpr (ES == 2) {
        u4 w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = pack2bf(v[2 * k], v[2 * k + 1]);
        *reinterpret_cast<u4*>(reinterpret_cast<uint16_t*>(dst) + o) = w;
    } else {
        u4 w0, w1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { w0[k] = __float_as_uint(v[k]); w1[k] = __float_as_uint(v[4 + k]); }
        u4* d = reinterpret_cast<u4*>(reinterpret_cast<float*>(dst) + o);
        d[0] = w0; d[1] = w1;
    }
}

// DepthNet's stem pack for the DCDP pair batch [target frames | reference frames] (bf16), which ALSO writes the six rgb channels of
// PoseNet's 8-channel input [tgt rgb | ref rgb | depth_t | depth_r]: image i fills channels 0..2 (i < Bh) or 3..5 of pair i mod Bh.
// The two depth channels are written by the depth head's kernel (csrc/fwd16.hip): PoseNet's own packing pass -- a read of the same
// frames plus both depth maps, 13 us at 8 pairs on the forward chain -- disappears.
// A thread takes one pixel of one PAIR (six plane reads in flight), so PoseNet's pixel is one full 16-byte store: its depth channels are
// zero until the head's kernel -- later in the same pass -- writes them.
__global__ __launch_bounds__(NT) void k_pack_stem_pose(const float* __restrict__ frames, int HW, int Bh, void* __restrict__ stem,
                                                        void* __restrict__ pose_in) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    const int p = blockIdx.y;
    const size_t pix = (size_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= (size_t)HW) return;
    const float* t = frames + (size_t)p * 3 * HW + pix;
    const float* r = frames + (size_t)(p + Bh) * 3 * HW + pix;
    const float t0 = t[0], t1 = t[(size_t)HW], t2 = t[2 * (size_t)HW], r0 = r[0], r1 = r[(size_t)HW], r2 = r[2 * (size_t)HW];
    const unsigned a0 = f2bf(t0), a1 = f2bf(t1), a2 = f2bf(t2), b0 = f2bf(r0), b1 = f2bf(r1), b2 = f2bf(r2);
    u4 w;
    w[0] = a0 | (a1 << 16); w[1] = a2; w[2] = 0u; w[3] = 0u;
    *reinterpret_cast<u4*>(reinterpret_cast<uint16_t*>(stem) + ((size_t)p * HW + pix) * 8) = w;
    w[0] = b0 | (b1 << 16); w[1] = b2;
    *reinterpret_cast<u4*>(reinterpret_cast<uint16_t*>(stem) + ((size_t)(p + Bh) * HW + pix) * 8) = w;
    w[0] = a0 | (a1 << 16); w[1] = a2 | (b0 << 16); w[2] = b1 | (b2 << 16); w[3] = 0u;
    *reinterpret_cast<u4*>(reinterpret_cast<uint16_t*>(pose_in) + ((size_t)p * HW + pix) * 8) = w;
}

// PX pixels per thread: the reads are 2 / 4 useful bytes per 16 / 32-byte pixel, so a thread needs several in flight
template <int ES, int PX = 4>
__global__ __launch_bounds__(NT) void k_unpack_nhwc(const void* __restrict__ src, int HW, int Cpad, int c_begin,
                                                    int c_count, float* __restrict__ dst, int flags) {
    const int b = blockIdx.y;
    const bool accumulate = flags & 1, by_channel = flags & 2;      // by_channel: dst is [c_count][B][H][W]
    const size_t pix0 = (size_t)blockIdx.x * NT * PX + threadIdx.x;
    for (int c = 0; c < c_count; ++c) {
        float v[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const size_t pix = pix0 + (size_t)j * NT;
            v[j] = (pix < (size_t)HW) ? Elem<ES>::ld(src, ((size_t)b * HW + pix) * Cpad + c_begin + c) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const size_t pix = pix0 + (size_t)j * NT;
            if (pix >= (size_t)HW) continue;
            float* d = dst + (by_channel ? (size_t)c * gridDim.y + b : (size_t)b * c_count + c) * HW + pix;
            *d = accumulate ? (*d + v[j]) : v[j];
        }
    }
}

template <int ES>
__global__ __launch_bounds__(NT) void k_relu_bwd(const void* __restrict__ y, void* __restrict__ dy, size_t n) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    if (!(Elem<ES>::ld(y, i) > 0.0f)) Elem<ES>::st(dy, i, 0.0f);
}

// ---------------------------------------------------------------- DepthNet head -------------- //
// pre = conv3x3(x; w[9][C]) + bias;  depth = 1 / (lo + (hi - lo) * sigmoid(pre))
template <int ES>
__global__ __launch_bounds__(NT) void k_depth_head_fwd(const void* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int H, int W, int C, float lo,
                                                       float hi, float* __restrict__ depth) {
    extern __shared__ float sw[];   // 9*C
    for (int i = threadIdx.x; i < 9 * C; i += NT) sw[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const size_t pix = (size_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= (size_t)H * W) return;
    const int yy = (int)(pix / W), xx = (int)(pix - (size_t)yy * W);
    float acc = bias[0];
    for (int ky = 0; ky < 3; ++ky) {
        const int y2 = yy + ky - 1;
        if (y2 < 0 || y2 >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int x2 = xx + kx - 1;
            if (x2 < 0 || x2 >= W) continue;
            const size_t o = (((size_t)b * H + y2) * W + x2) * C;
            const float* wt = sw + (ky * 3 + kx) * C;
            for (int c = 0; c < C; ++c) acc += Elem<ES>::ld(x, o + c) * wt[c];
        }
    }
    const float sig = 1.0f / (1.0f + expf(-acc));
    depth[(size_t)b * H * W + pix] = 1.0f / (lo + (hi - lo) * sig);
}

// C = 16 specialisation: one 16-channel pixel is 32 B (bf16) / 64 B (f32) -> 16-byte vector loads, weights in LDS
template <int ES>
__global__ __launch_bounds__(NT) void k_depth_head_fwd16(const void* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, int H, int W, float lo, float hi,
                                                         float* __restrict__ depth) {
    constexpr int C = 16;
    __shared__ float sw[9 * C];
    if (threadIdx.x < 9 * C) sw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.y;
    const size_t pix = (size_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= (size_t)H * W) return;
    const int yy = (int)(pix / W), xx = (int)(pix - (size_t)yy * W);
    float acc = bias[0];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int y2 = yy + ky - 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int x2 = xx + kx - 1;
            if (y2 < 0 || y2 >= H || x2 < 0 || x2 >= W) continue;
            const char* p = reinterpret_cast<const char*>(x) + ((((size_t)b * H + y2) * W + x2) * C) * ES;
            const float* wt = sw + (ky * 3 + kx) * C;
            if constexpr (ES == 2) {
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const uint4 q = *reinterpret_cast<const uint4*>(p + 16 * v);
                    const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        acc = fmaf(__uint_as_float(u[k] << 16), wt[8 * v + 2 * k], acc);
                        acc = fmaf(__uint_as_float(u[k] & 0xFFFF0000u), wt[8 * v + 2 * k + 1], acc);
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 q = *reinterpret_cast<const float4*>(p + 16 * v);
                    acc = fmaf(q.x, wt[4 * v], acc); acc = fmaf(q.y, wt[4 * v + 1], acc);
                    acc = fmaf(q.z, wt[4 * v + 2], acc); acc = fmaf(q.w, wt[4 * v + 3], acc);
                }
            }
        }
    }
    const float sig = 1.0f / (1.0f + expf(-acc));
    depth[(size_t)b * H * W + pix] = 1.0f / (lo + (hi - lo) * sig);
}

// bf16: LDS-tiled form.  The kernel above pulls every input pixel through the L1 nine times (18 x 16 B per output: ~10 us of
// texture-path time for 16 frames of 256x320 before any latency); here a workgroup stages the (4 + 2) x (64 + 2) input pixels
// of its 4 x 64 output tile ONCE (coalesced 16-byte loads, zero outside the image) and the nine taps read LDS.  Pixel pitch
// 48 B: 16 consecutive pixels of a ds_read_b128 group start on 16 different 16-byte slots.
__global__ __launch_bounds__(NT) void k_depth_head_fwd16_lds(const void* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int H, int W, float lo, float hi,
                                                             float* __restrict__ depth) {
    constexpr int C = 16, TH = 4, TW = 64, PH = TH + 2, PW = TW + 2, PITCH = 48;
    __shared__ float sw[9 * C];
    __shared__ __attribute__((aligned(16))) char tile[PH * PW * PITCH];
    const int tid = threadIdx.x;
    if (tid < 9 * C) sw[tid] = w[tid];
    const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
    const char* img = reinterpret_cast<const char*>(x) + (size_t)b * H * W * C * 2;
    for (int i = tid; i < PH * PW * 2; i += NT) {          // 2 granules of 16 B per pixel
        const int pix = i >> 1, h = i & 1;
        const int py = pix / PW, px = pix - py * PW;
        const int yy = y0 - 1 + py, xx = x0 - 1 + px;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = *reinterpret_cast<const uint4*>(img + ((size_t)yy * W + xx) * C * 2 + 16 * h);
        *reinterpret_cast<uint4*>(tile + pix * PITCH + 16 * h) = v;
    }
    __syncthreads();
    const int ty = tid >> 6, tx = tid & 63;
    const int yy = y0 + ty, xx = x0 + tx;
    if (yy >= H || xx >= W) return;
    float acc0 = bias[0], acc1 = 0.0f, acc2 = 0.0f;        // one chain per tap row
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float a = 0.0f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const char* p = tile + ((ty + ky) * PW + tx + kx) * PITCH;
            const float* wt = sw + (ky * 3 + kx) * C;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint4 q = *reinterpret_cast<const uint4*>(p + 16 * h);
                const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a = fmaf(__uint_as_float(u[k] << 16), wt[8 * h + 2 * k], a);
                    a = fmaf(__uint_as_float(u[k] & 0xFFFF0000u), wt[8 * h + 2 * k + 1], a);
                }
            }
        }
        if (ky == 0) acc0 += a; else if (ky == 1) acc1 = a; else acc2 = a;
    }
    const float pre = acc0 + (acc1 + acc2);
    const float sig = 1.0f / (1.0f + expf(-pre));
    depth[(size_t)b * H * W + (size_t)yy * W + xx] = 1.0f / (lo + (hi - lo) * sig);
}

// (A four-pixels-per-thread variant -- 36 instead of 72 loads per four outputs, four independent accumulators -- measured
// 35.9 us against this kernel's 24.7: a lane stride of 128 B costs more in the load path than the reuse saves.)
// d(pre) from the saved depth:  sig = (1/depth - lo)/(hi-lo);  d depth/d pre = -(hi-lo) depth^2 sig (1-sig)
__device__ __forceinline__ float head_dpre(float depth, float d_depth, float lo, float hi) {
    const float k = hi - lo;
    const float sig = (1.0f / depth - lo) / k;
    return -d_depth * k * depth * depth * sig * (1.0f - sig);
}

__global__ __launch_bounds__(NT) void k_depth_head_dpre(const float* __restrict__ depth, const float* __restrict__ d_depth,
                                                        size_t n, float lo, float hi, float* __restrict__ dpre) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) dpre[i] = head_dpre(depth[i], d_depth[i], lo, hi);
}

// the same with the incoming gradient in parts: first half of the images g0 + sa*sb*graw, second half g1 + sa*sb*graw1
// (any may be null)
__global__ __launch_bounds__(NT) void k_depth_head_dpre_parts(const float* __restrict__ depth, const float* __restrict__ g0,
                                                              const float* __restrict__ g1, const float* __restrict__ graw,
                                                              const float* __restrict__ graw1,
                                                              const float* __restrict__ sa, const float* __restrict__ sb,
                                                              size_t n_half, float lo, float hi, float* __restrict__ dpre) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= 2 * n_half) return;
    float g;
    if (i < n_half) {
        g = g0 ? g0[i] : 0.0f;
        if (graw) g = fmaf((sa ? sa[0] : 1.0f) * (sb ? sb[0] : 1.0f), graw[i], g);
    } else {
        g = g1 ? g1[i - n_half] : 0.0f;
        if (graw1) g = fmaf((sa ? sa[0] : 1.0f) * (sb ? sb[0] : 1.0f), graw1[i - n_half], g);
    }
    dpre[i] = head_dpre(depth[i], g, lo, hi);
}

// dx[y,x,c] = (x[y,x,c] > 0) * sum_taps dpre[y-ky+1, x-kx+1] * w[ky,kx,c]
template <int ES>
__global__ __launch_bounds__(NT) void k_depth_head_dgrad(const void* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dpre, int H, int W, int C,
                                                         void* __restrict__ dx) {
    extern __shared__ float sw[];
    for (int i = threadIdx.x; i < 9 * C; i += NT) sw[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const size_t pix = (size_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= (size_t)H * W) return;
    const int yy = (int)(pix / W), xx = (int)(pix - (size_t)yy * W);
    float g[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y2 = yy - ky + 1, x2 = xx - kx + 1;
            g[ky * 3 + kx] = (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) ? dpre[((size_t)b * H + y2) * W + x2] : 0.0f;
        }
    const size_t o = ((size_t)b * H * W + pix) * C;
    for (int c = 0; c < C; ++c) {
        float v = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) v += g[t] * sw[t * C + c];
        Elem<ES>::st(dx, o + c, (Elem<ES>::ld(x, o + c) > 0.0f) ? v : 0.0f);
    }
}

// The same for C = 16 (the DepthNet head) with whole-pixel accesses: a thread reads its pixel's 16 channels as 16-byte granules,
// and writes them back the same way -- the generic kernel above issues 16 two-byte loads and stores per pixel.
template <int ES>
__global__ __launch_bounds__(NT) void k_depth_head_dgrad16(const void* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ dpre, int H, int W, void* __restrict__ dx) {
    constexpr int C = 16, NGR = C * ES / 16;       // granules per pixel: 2 (bf16) / 4 (f32)
    __shared__ float sw[9 * C];
    if (threadIdx.x < 9 * C) sw[threadIdx.x] = w[threadIdx.x];
    __syncthreads();
    const int b = blockIdx.y;
    const size_t pix = (size_t)blockIdx.x * NT + threadIdx.x;
    if (pix >= (size_t)H * W) return;
    const int yy = (int)(pix / W), xx = (int)(pix - (size_t)yy * W);
    float g[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y2 = yy - ky + 1, x2 = xx - kx + 1;
            g[ky * 3 + kx] = (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) ? dpre[((size_t)b * H + y2) * W + x2] : 0.0f;
        }
    const size_t o = ((size_t)b * H * W + pix) * NGR;
    const uint4* xin = reinterpret_cast<const uint4*>(x) + o;
    uint4* out = reinterpret_cast<uint4*>(dx) + o;
    uint4 xv[NGR];
#pragma unroll
    for (int q = 0; q < NGR; ++q) xv[q] = xin[q];
    float v[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float a = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) a += g[t] * sw[t * C + c];
        v[c] = a;
    }
    if constexpr (ES == 2) {
#pragma unroll
        for (int q = 0; q < NGR; ++q) {
            const unsigned xw[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
            unsigned ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 8 * q + 2 * j;
                const float x0 = bf2f((uint16_t)(xw[j] & 0xffffu)), x1 = bf2f((uint16_t)(xw[j] >> 16));
                ow[j] = pack2bf(x0 > 0.0f ? v[c] : 0.0f, x1 > 0.0f ? v[c + 1] : 0.0f);
            }
            out[q] = uint4{ow[0], ow[1], ow[2], ow[3]};
        }
    } else {
#pragma unroll
        for (int q = 0; q < NGR; ++q) {
            const unsigned xw[4] = {xv[q].x, xv[q].y, xv[q].z, xv[q].w};
            unsigned ow[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ow[j] = __float_as_uint(__uint_as_float(xw[j]) > 0.0f ? v[4 * q + j] : 0.0f);
            out[q] = uint4{ow[0], ow[1], ow[2], ow[3]};
        }
    }
}

// dw[t][c] += sum_pix dpre[pix] * x[pix + tap][c];  db += sum dpre.
// Written from the input pixel's side: dw[t][c] = sum_q x[q][c] * dpre[q - tap].  Each thread walks a
// strided set of pixels q, keeps all 9*C products in registers, and the workgroup reduces ONCE at the
// end (wave shuffle + LDS) before one fp32 atomic per weight.
// ROWS = 3: a thread keeps all 9 x C products (145 accumulators: two waves per SIMD).  ROWS = 1: blockIdx.z selects the tap
// row ky and a thread keeps 3 x C products -- three times the threads, each re-reading its pixel's C channels from L2, at a
// third of the registers: more waves to hide the load latency this kernel is bound by.
template <int ES, int C, int ROWS>
__global__ __launch_bounds__(NT) void k_depth_head_wgrad(const void* __restrict__ x, const float* __restrict__ dpre,
                                                         int H, int W, int px_per_block, float* __restrict__ dw,
                                                         float* __restrict__ db, float* __restrict__ partials) {
    constexpr int NTAP = 3 * ROWS;
    const int ky0 = (ROWS == 3) ? 0 : (int)blockIdx.z;
    const bool with_bias = (ROWS == 3) || ky0 == 1;
    __shared__ float red[4][NTAP * C + 1];
    const int b = blockIdx.y;
    const int HW = H * W;
    const int p0 = blockIdx.x * px_per_block, p1 = min(HW, p0 + px_per_block);
    float acc[NTAP][C];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[t][c] = 0.0f;
    float sb = 0.0f;
    // The loads of pixel q + 256 are issued before the 9*C multiply-adds of pixel q (two waves per SIMD at ~180 VGPRs:
    // without it every iteration waited out a full memory round trip; 54 -> us, profiles/r2_bench_kernel_stats.csv).
    typedef __attribute__((ext_vector_type(4))) unsigned int u4;
    constexpr int NV = C * ES / 16;              // the pixel's C channels as 16-byte vectors (C * ES is a multiple of 16)
    struct Px { u4 raw[NV]; float d[NTAP]; float dc; };
    auto fetch = [&](int q, Px& o) {
        const int qy = q / W, qx = q - qy * W;
        const u4* px = reinterpret_cast<const u4*>(reinterpret_cast<const char*>(x) + ((size_t)b * HW + q) * C * ES);
#pragma unroll
        for (int v = 0; v < NV; ++v) o.raw[v] = px[v];
        o.dc = with_bias ? dpre[(size_t)b * HW + q] : 0.0f;
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ky = ky0 + r;
                const int oy = qy - ky + 1, ox = qx - kx + 1;      // output pixel whose tap (ky,kx) lands on q
                o.d[r * 3 + kx] = (oy >= 0 && oy < H && ox >= 0 && ox < W) ? dpre[((size_t)b * H + oy) * W + ox] : 0.0f;
            }
    };
    int q = p0 + (int)threadIdx.x;
    Px cur;
    if (q < p1) fetch(q, cur);
    while (q < p1) {
        const int qn = q + NT;
        Px nxt = cur;
        if (qn < p1) fetch(qn, nxt);
        float xv[C];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const u4 t = cur.raw[v];
            if constexpr (ES == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) xv[4 * v + k] = __uint_as_float(t[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xv[8 * v + 2 * k] = __uint_as_float(t[k] << 16);
                    xv[8 * v + 2 * k + 1] = __uint_as_float(t[k] & 0xFFFF0000u);
                }
            }
        }
        sb += cur.dc;
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[t][c] = fmaf(cur.d[t], xv[c], acc[t][c]);
        cur = nxt;
        q = qn;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float v = wave_sum(acc[t][c]);
            if (lane == 0) red[wave][t * C + c] = v;
        }
    sb = wave_sum(sb);
    if (lane == 0) red[wave][NTAP * C] = sb;
    __syncthreads();
    for (int k = threadIdx.x; k < NTAP * C + 1; k += NT) {
        const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        if (partials) {
            // deterministic form: row (image, pixel range) of a [rows][9 C + 1] table, every entry written by exactly one workgroup
            float* row = partials + ((size_t)b * gridDim.x + blockIdx.x) * (9 * C + 1);
            if (k < NTAP * C) row[ky0 * 3 * C + k] = v;
            else if (with_bias) row[9 * C] = v;
        } else if (k < NTAP * C) atomicAdd(dw + ky0 * 3 * C + k, v);
        else if (with_bias) atomicAdd(db, v);
    }
}

// table form, second launch: dw[k] += sum over the table's rows (k < 9 C), db += column 9 C.  ONE WORKGROUP PER COLUMN: thread t adds
// rows t, t + 256, ... in order, the 256 partial sums meet in a fixed tree -- a fixed order, so the result is reproducible.  (The
// first form, one thread per column walking all rows, took 205 us for 640 rows: a chain of dependent strided loads.)
__global__ __launch_bounds__(NT) void k_head_wgrad_reduce(const float* __restrict__ partials, int rows, int ncol,
                                                          float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float s[NT];
    const int k = blockIdx.x, tid = threadIdx.x;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    int r = tid;
    for (; r + 3 * NT < rows; r += 4 * NT) {            // four loads in flight
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += partials[(size_t)(r + q * NT) * ncol + k];
    }
    for (int q = 0; r < rows; r += NT, ++q) v[q] += partials[(size_t)r * ncol + k];
    s[tid] = (v[0] + v[1]) + (v[2] + v[3]);
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if (tid < o) s[tid] += s[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        if (k < ncol - 1) dw[k] += s[0];
        else db[0] += s[0];
    }
}

// generic-C fallback: one (tap, c) pair per thread over a strip of rows
template <int ES>
__global__ __launch_bounds__(NT) void k_depth_head_wgrad_generic(const void* __restrict__ x, const float* __restrict__ dpre,
                                                                 int H, int W, int C, int rows_per_block,
                                                                 float* __restrict__ dw, float* __restrict__ db) {
    const int b = blockIdx.y;
    const int y0 = blockIdx.x * rows_per_block, y1 = min(H, y0 + rows_per_block);
    const int tid = threadIdx.x;
    const int nk = 9 * C;
    for (int k = tid; k < nk + 1; k += NT) {
        float acc = 0.0f;
        if (k < nk) {
            const int t = k / C, c = k - t * C;
            const int ky = t / 3, kx = t - 3 * ky;
            for (int yy = y0; yy < y1; ++yy) {
                const int y2 = yy + ky - 1;
                if (y2 < 0 || y2 >= H) continue;
                for (int xx = 0; xx < W; ++xx) {
                    const int x2 = xx + kx - 1;
                    if (x2 < 0 || x2 >= W) continue;
                    acc += dpre[((size_t)b * H + yy) * W + xx] * Elem<ES>::ld(x, (((size_t)b * H + y2) * W + x2) * C + c);
                }
            }
            atomicAdd(dw + k, acc);
        } else {
            for (int yy = y0; yy < y1; ++yy)
                for (int xx = 0; xx < W; ++xx) acc += dpre[((size_t)b * H + yy) * W + xx];
            atomicAdd(db, acc);
        }
    }
}

// ---------------------------------------------------------------- PoseNet head --------------- //
// o_j = s_j * (bias_j + mean_p sum_c x[b][p][c] w[j][c]) (+1 for j = 6);  s = pose_scale (j<6) | lcc_scale.
// Output is PLANAR: out = [ pose B x 6 | lcc_a B | lcc_b B ], so the three results are contiguous views.
template <int ES>
__global__ __launch_bounds__(NT) void k_pose_head_fwd(const void* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int HW, int C, float pose_scale,
                                                      float lcc_scale, float* __restrict__ out) {
    __shared__ float red[4][8];
    const int b = blockIdx.x, tid = threadIdx.x;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    for (int c = tid; c < C; c += NT) {
        float sx = 0.0f;
        for (int p = 0; p < HW; ++p) sx += Elem<ES>::ld(x, ((size_t)b * HW + p) * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += sx * w[j * C + c];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = wave_sum(acc[j]);
        if ((tid & 63) == 0) red[tid >> 6][j] = v;
    }
    __syncthreads();
    if (tid < 8) {
        const float pre = bias[tid] + ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) / (float)HW;
        float o = (tid < 6 ? pose_scale : lcc_scale) * pre;
        if (tid == 6) o += 1.0f;
        const int B = gridDim.x;
        if (tid < 6) out[b * 6 + tid] = o;
        else out[6 * B + (tid - 6) * B + b] = o;
    }
}

template <int ES>
__global__ __launch_bounds__(NT) void k_pose_head_bwd(const void* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ d_pose, const float* __restrict__ d_a,
                                                      const float* __restrict__ d_b, const float* __restrict__ sa,
                                                      const float* __restrict__ sb, int HW, int C, float pose_scale,
                                                      float lcc_scale, void* __restrict__ dx, float* __restrict__ dw,
                                                      float* __restrict__ db) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const float gs = (sa ? sa[0] : 1.0f) * (sb ? sb[0] : 1.0f);
    pose_scale *= gs;
    lcc_scale *= gs;
    float go[8];
#pragma unroll
    for (int j = 0; j < 6; ++j) go[j] = d_pose ? d_pose[b * 6 + j] * pose_scale : 0.0f;
    go[6] = d_a ? d_a[b] * lcc_scale : 0.0f;
    go[7] = d_b ? d_b[b] * lcc_scale : 0.0f;
    const float inv = 1.0f / (float)HW;
    for (int c = tid; c < C; c += NT) {
        float g = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) g += go[j] * w[j * C + c];
        g *= inv;
        float sx = 0.0f;
        for (int p = 0; p < HW; ++p) {
            const size_t o = ((size_t)b * HW + p) * C + c;
            const float xv = Elem<ES>::ld(x, o);
            sx += xv;
            Elem<ES>::st(dx, o, xv > 0.0f ? g : 0.0f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(dw + j * C + c, go[j] * sx * inv);
    }
    if (tid < 8) atomicAdd(db + tid, go[tid]);
}

// Deterministic form of the PoseNet head backward: thread = channel, the images are walked IN ORDER by the one thread that owns
// dw[.][c] (no atomics); dx as above.  The head sees <= 20 pixels per image: the serial walk costs a few microseconds.
template <int ES>
__global__ __launch_bounds__(NT) void k_pose_head_bwd_det(const void* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ d_pose, const float* __restrict__ d_a,
                                                          const float* __restrict__ d_b, const float* __restrict__ sa,
                                                          const float* __restrict__ sb, int B, int HW, int C, float pose_scale,
                                                          float lcc_scale, void* __restrict__ dx, float* __restrict__ dw,
                                                          float* __restrict__ db) {
    const int c = blockIdx.x * NT + threadIdx.x;
    const float gs = (sa ? sa[0] : 1.0f) * (sb ? sb[0] : 1.0f);
    pose_scale *= gs;
    lcc_scale *= gs;
    const float inv = 1.0f / (float)HW;
    float dwacc[8], dbacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwacc[j] = 0.0f; dbacc[j] = 0.0f; }
    for (int b = 0; b < B; ++b) {
        float go[8];
#pragma unroll
        for (int j = 0; j < 6; ++j) go[j] = d_pose ? d_pose[b * 6 + j] * pose_scale : 0.0f;
        go[6] = d_a ? d_a[b] * lcc_scale : 0.0f;
        go[7] = d_b ? d_b[b] * lcc_scale : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) dbacc[j] += go[j];
        if (c >= C) continue;
        float g = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) g += go[j] * w[j * C + c];
        g *= inv;
        float sx = 0.0f;
        for (int p = 0; p < HW; ++p) {
            const size_t o = ((size_t)b * HW + p) * C + c;
            const float xv = Elem<ES>::ld(x, o);
            sx += xv;
            Elem<ES>::st(dx, o, xv > 0.0f ? g : 0.0f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) dwacc[j] += go[j] * sx * inv;
    }
    if (c < C) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dw[j * C + c] += dwacc[j];
    }
    if (c < 8) db[c] += dbacc[c];
}

// ---------------------------------------------------------------- Adam ----------------------- //
// One thread updates 4 consecutive parameters (16-byte loads / stores: 48.1 -> 46.6 us per step for the two arenas; the entry
// points check the alignment).  (Measured and dropped: writing a bf16 mirror of the arena here as the forward operand copy and
// producing only the transposed copy in k_pack_weights_multi, on the side stream beside the forward pass -- Adam +3 us, the
// repacking pass -1 us, and the side-stream launch slowed the step by 0.6 %: DESIGN.md section 3.3.)
struct AdamCoef { float step_size, rs_bc2; };
__device__ __forceinline__ AdamCoef adam_coef(float lr, float b1, float b2, const int32_t* step_count, int t_host) {
    const int t = step_count ? step_count[0] + 1 : t_host;
    const float bc1 = 1.0f - powf(b1, (float)t);
    const float bc2 = 1.0f - powf(b2, (float)t);
    return AdamCoef{lr / bc1, 1.0f / sqrtf(bc2)};
}
__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, float b1, float b2, float eps, float gscale,
                                          const AdamCoef c) {
    const float gi = g * gscale;
    m = b1 * m + (1.0f - b1) * gi;
    v = b2 * v + (1.0f - b2) * gi * gi;
    p -= c.step_size * (m / (sqrtf(v) * c.rs_bc2 + eps));
    return p;
}
__global__ __launch_bounds__(NT) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                             float gscale, const int32_t* __restrict__ step_count, int t_host) {
    const AdamCoef c = adam_coef(lr, b1, b2, step_count, t_host);
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * NT;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
        float4 pi = p4[i], mi = m4[i], vi = v4[i];
        const float4 gi = g4[i];
        adam_one(pi.x, gi.x, mi.x, vi.x, b1, b2, eps, gscale, c);
        adam_one(pi.y, gi.y, mi.y, vi.y, b1, b2, eps, gscale, c);
        adam_one(pi.z, gi.z, mi.z, vi.z, b1, b2, eps, gscale, c);
        adam_one(pi.w, gi.w, mi.w, vi.w, b1, b2, eps, gscale, c);
        m4[i] = mi;
        v4[i] = vi;
        p4[i] = pi;
    }
    // tail (arenas of this library are multiples of 64 floats; other callers may pass any n)
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, b1, b2, eps, gscale, c);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
}

__global__ void k_inc_step(int32_t* step_count) { step_count[0] += 1; }

// Several arenas in ONE launch (a dependent launch costs ~2.7 us before it does anything, tools/ubench/launch_floor.hip, and the
// small arena alone does not fill the memory system): workgroups [first[i], first[i + 1]) walk arena i.
struct AdamArenas {
    int count;
    unsigned first[COLVO_MAX_ARENAS + 1];
    ColvoAdamArena a[COLVO_MAX_ARENAS];
};
__global__ __launch_bounds__(NT) void k_adam_multi(AdamArenas as, float lr, float b1, float b2, float eps, float gscale, int t_host) {
    const AdamCoef c = adam_coef(lr, b1, b2, nullptr, t_host);
    int i = 0;
#pragma unroll
    for (int q = 1; q < COLVO_MAX_ARENAS; ++q)
        if (q < as.count && blockIdx.x >= as.first[q]) i = q;
    ColvoAdamArena A = as.a[0];
    unsigned b0 = as.first[0], b1x = as.first[1];
#pragma unroll
    for (int q = 1; q < COLVO_MAX_ARENAS; ++q)
        if (i == q) { A = as.a[q]; b0 = as.first[q]; b1x = as.first[q + 1]; }
    const size_t n4 = A.n / 4, stride = (size_t)(b1x - b0) * NT;
    float4* p4 = reinterpret_cast<float4*>(A.param);
    const float4* g4 = reinterpret_cast<const float4*>(A.grad);
    float4* m4 = reinterpret_cast<float4*>(A.exp_avg);
    float4* v4 = reinterpret_cast<float4*>(A.exp_avg_sq);
    const size_t start = (size_t)(blockIdx.x - b0) * NT + threadIdx.x;
    for (size_t k = start; k < n4; k += stride) {
        float4 pi = p4[k], mi = m4[k], vi = v4[k];
        const float4 gi = g4[k];
        adam_one(pi.x, gi.x, mi.x, vi.x, b1, b2, eps, gscale, c);
        adam_one(pi.y, gi.y, mi.y, vi.y, b1, b2, eps, gscale, c);
        adam_one(pi.z, gi.z, mi.z, vi.z, b1, b2, eps, gscale, c);
        adam_one(pi.w, gi.w, mi.w, vi.w, b1, b2, eps, gscale, c);
        m4[k] = mi;
        v4[k] = vi;
        p4[k] = pi;
    }
    for (size_t k = n4 * 4 + start; k < A.n; k += stride) {
        float pi = A.param[k], mi = A.exp_avg[k], vi = A.exp_avg_sq[k];
        adam_one(pi, A.grad[k], mi, vi, b1, b2, eps, gscale, c);
        A.exp_avg[k] = mi; A.exp_avg_sq[k] = vi; A.param[k] = pi;
    }
}

// Adam AND the operand copies of the updated weights in one pass (both networks, one launch): the update has every new weight in
// a register, so the bf16 / transposed copies the next forward and backward pass read cost 4 more bytes per parameter here
// instead of a 12-byte-per-parameter repacking pass of their own (k_pack_weights_multi) plus its launches.  Table-driven like
// that kernel: an entry of kind 0 is one 3x3 layer's weights, walked in 32 x 64 transpose tiles of one tap; an entry of kind 1
// a plain range of the arena (biases, heads, padding) that only takes the update.  Same arithmetic as k_adam, element for element.
template <int ES>
__global__ __launch_bounds__(NT) void k_adam_pack(const ColvoAdamPackEntry* __restrict__ tab, int nentries, float lr, float b1,
                                                  float b2, float eps, float gscale_host, const float* __restrict__ gscale_dev,
                                                  const int32_t* __restrict__ step_count, int t_host) {
    __shared__ float tile[PK_CO][PK_C + 1];
    // gscale_dev: a second factor that is only known on the device (data parallel: world / max(3 n_valid of the WHOLE batch, 1), the
    // normaliser of raw loss gradients whose all-reduce overlapped the backward pass -- colvo_warp_loss_rescale_to)
    const float gscale = gscale_dev ? gscale_host * *gscale_dev : gscale_host;
    const AdamCoef c = adam_coef(lr, b1, b2, step_count, t_host);
    int l = 0;
    for (int i = 1; i < nentries; ++i)
        if ((int)blockIdx.x >= tab[i].blk_begin) l = i;
    const ColvoAdamPackEntry e = tab[l];
    const int lb = blockIdx.x - e.blk_begin, tid = threadIdx.x;
    float* __restrict__ P = e.param + e.w_off;
    float* __restrict__ G = const_cast<float*>(e.grad) + e.w_off;      // (written only where the entry asks for zeroed gradients)
    float* __restrict__ M = e.exp_avg + e.w_off;
    float* __restrict__ V = e.exp_avg_sq + e.w_off;
    const bool zg = e.zero_grad != 0;
    if (e.kind != 0) {                     // plain range: ADAM_PLAIN_PER_WG elements per workgroup
        const long long k0 = (long long)lb * ADAM_PLAIN_PER_WG;
        for (long long k = k0 + tid; k < e.n && k < k0 + ADAM_PLAIN_PER_WG; k += NT) {
            float pi = P[k], mi = M[k], vi = V[k];
            adam_one(pi, G[k], mi, vi, b1, b2, eps, gscale, c);
            M[k] = mi; V[k] = vi; P[k] = pi;
            if (zg) G[k] = 0.0f;
        }
        return;
    }
    const int nct = (e.Cin + PK_C - 1) / PK_C, ncot = (e.Cout + PK_CO - 1) / PK_CO;
    const int t = lb / (ncot * nct), r = lb - t * (ncot * nct);
    const int cot = r / nct, ct = r - cot * nct;
    const int co0 = cot * PK_CO, c0 = ct * PK_C;
    {
        const int cc = c0 + (tid & 63);
        float pv[PK_CO / 4], gv[PK_CO / 4], mv[PK_CO / 4], vv[PK_CO / 4];
#pragma unroll
        for (int i = 0; i < PK_CO / 4; ++i) {           // all loads first: 32 in flight per thread
            const int co = co0 + (tid >> 6) + 4 * i;
            const bool ok = co < e.Cout && cc < e.Cin;
            const size_t idx = ok ? ((size_t)co * e.kk + t) * e.Cin + cc : 0;
            pv[i] = P[idx]; gv[i] = G[idx]; mv[i] = M[idx]; vv[i] = V[idx];
        }
#pragma unroll
        for (int i = 0; i < PK_CO / 4; ++i) {
            const int row = (tid >> 6) + 4 * i, co = co0 + row;
            if (co < e.Cout && cc < e.Cin) {
                const size_t idx = ((size_t)co * e.kk + t) * e.Cin + cc;
                adam_one(pv[i], gv[i], mv[i], vv[i], b1, b2, eps, gscale, c);
                M[idx] = mv[i]; V[idx] = vv[i]; P[idx] = pv[i];
                if (zg) G[idx] = 0.0f;
                tile[row][tid & 63] = pv[i];
                if (e.fwd && e.fwd_off >= 0) Elem<ES>::st(e.fwd, e.fwd_off + idx, pv[i]);
            }
        }
    }
    __syncthreads();
    {
        const int co = co0 + (tid & 31);
#pragma unroll
        for (int i = 0; i < PK_C / 8; ++i) {
            const int col = (tid >> 5) + 8 * i, cc = c0 + col;
            if (co < e.Cout && cc < e.Cin)
                Elem<ES>::st(e.bwd, e.bwd_off + ((size_t)cc * e.kk + (e.kk - 1 - t)) * e.Cout + co, tile[tid & 31][col]);   // taps flipped
        }
    }
}

// zero several buffers in one launch (16-byte stores; sizes are multiples of 16 bytes)
struct ZeroArenas {
    int count;
    unsigned first[COLVO_MAX_ARENAS + 1];
    void* p[COLVO_MAX_ARENAS];
    size_t n16[COLVO_MAX_ARENAS];
};
__global__ __launch_bounds__(NT) void k_zero_multi(ZeroArenas zs) {
    int i = 0;
#pragma unroll
    for (int q = 1; q < COLVO_MAX_ARENAS; ++q)
        if (q < zs.count && blockIdx.x >= zs.first[q]) i = q;
    void* p = zs.p[0]; size_t n16 = zs.n16[0];
    unsigned b0 = zs.first[0], b1 = zs.first[1];
#pragma unroll
    for (int q = 1; q < COLVO_MAX_ARENAS; ++q)
        if (i == q) { p = zs.p[q]; n16 = zs.n16[q]; b0 = zs.first[q]; b1 = zs.first[q + 1]; }
    float4* d = reinterpret_cast<float4*>(p);
    const size_t stride = (size_t)(b1 - b0) * NT;
    for (size_t k = (size_t)(blockIdx.x - b0) * NT + threadIdx.x; k < n16; k += stride) d[k] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- input gradient of a 3x3 conv w.r.t. a FEW input channels, as fp32 planes (colvo_conv_dgrad_planes) ----
// PoseNet's first layer takes [tgt rgb | ref rgb | depth_t | depth_r]; of its input gradient only the two depth channels are wanted,
// as fp32 NCHW planes for DepthNet's backward pass.  The general path computed all 8 channels of the 256x320 gradient with the MFMA
// kernel (24 us at 8 pairs: it is a write of 10.5 MB NHWC for 0.4 GFLOP) and then unpacked two of them (+5 us), on the critical path
// between the two networks' backward passes.  Here: dx[c] = sum over the output pixels that see the input pixel and over Cout of
// w[co][tap][c] dy[co], weights of the wanted channels in LDS as [tap][co][c], dy rows through L2, planes written coalesced.
// Stride 2: a thread takes a PAIR of horizontally adjacent input pixels (2m, 2m + 1) -- the even one sees the middle tap column of
// output column m, the odd one the right column of m and the left column of m + 1 -- so every lane of a wave runs the same taps (with a
// thread per pixel the lanes alternated between the two parity classes: every tap body ran under half an EXEC mask, 21 us in the step).
template <int ES>
__device__ __forceinline__ void load_row16(const char* row, int co, float (&v)[8]) {
    if constexpr (ES == 2) {
        const uint4 q = *reinterpret_cast<const uint4*>(row + co * 2);
        const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(u[j] << 16); v[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u); }
    } else {
        const float4 q0 = *reinterpret_cast<const float4*>(row + co * 4), q1 = *reinterpret_cast<const float4*>(row + co * 4 + 16);
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    }
}

template <int ES, int NC>
__global__ __launch_bounds__(NT) void k_conv_dgrad_planes(const void* __restrict__ dy, const float* __restrict__ w, int Cout, int Cin,
                                                          int c_begin, int B, int Hi, int Wi, int Ho, int Wo, int S,
                                                          float* __restrict__ dst, int accumulate) {
    extern __shared__ float sw[];                       // [9][Cout][NC]
    for (int i = threadIdx.x; i < 9 * Cout * NC; i += NT) {
        const int c = i % NC, co = (i / NC) % Cout, tap = i / (NC * Cout);
        sw[i] = w[((size_t)co * 9 + tap) * Cin + c_begin + c];
    }
    __syncthreads();
    const size_t HW = (size_t)Hi * Wi;
    if (S == 2) {
        const int Wp = (Wi + 1) >> 1;                   // pixel pairs per row
        const size_t p = (size_t)blockIdx.x * NT + threadIdx.x;
        if (p >= (size_t)B * Hi * Wp) return;
        const int b = (int)(p / ((size_t)Hi * Wp));
        const int r = (int)(p - (size_t)b * Hi * Wp);
        const int iy = r / Wp, m = r - iy * Wp;
        float a0[NC], a1[NC];                           // pixel 2m, pixel 2m + 1
#pragma unroll
        for (int c = 0; c < NC; ++c) { a0[c] = 0.0f; a1[c] = 0.0f; }
        const char* dyb = (const char*)dy + (size_t)b * Ho * Wo * Cout * ES;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;                 // 2 oy
            if (ty < 0 || (ty & 1)) continue;           // (wave-uniform: a wave holds pixels of one row or two)
            const int oy = ty >> 1;
            if (oy >= Ho) continue;
            const char* r0 = dyb + ((size_t)oy * Wo + m) * Cout * ES;            // output column m
            const bool has1 = m + 1 < Wo;                                        // output column m + 1 (left tap of the odd pixel)
            const float* w0 = sw + (ky * 3 + 0) * Cout * NC, *w1 = sw + (ky * 3 + 1) * Cout * NC, *w2 = sw + (ky * 3 + 2) * Cout * NC;
            for (int co = 0; co < Cout; co += 8) {
                float v[8], u[8];
                load_row16<ES>(r0, co, v);
                if (has1) load_row16<ES>(r0 + Cout * ES, co, u);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) u[j] = 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        a0[c] = fmaf(w1[(co + j) * NC + c], v[j], a0[c]);                       // 2m     + 1 - 1 = 2 m
                        a1[c] = fmaf(w2[(co + j) * NC + c], v[j], a1[c]);                       // 2m + 1 + 1 - 2 = 2 m
                        a1[c] = fmaf(w0[(co + j) * NC + c], u[j], a1[c]);                       // 2m + 1 + 1 - 0 = 2 (m + 1)
                    }
            }
        }
        const int ix = 2 * m;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float* d = dst + ((size_t)c * B + b) * HW + (size_t)iy * Wi + ix;   // [c][B][1][Hi][Wi]
            d[0] = accumulate ? d[0] + a0[c] : a0[c];
            if (ix + 1 < Wi) d[1] = accumulate ? d[1] + a1[c] : a1[c];
        }
        return;
    }
    const size_t p = (size_t)blockIdx.x * NT + threadIdx.x;
    if (p >= (size_t)B * HW) return;
    const int b = (int)(p / HW);
    const int r = (int)(p - (size_t)b * HW);
    const int iy = r / Wi, ix = r - iy * Wi;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.0f;
    const char* dyb = (const char*)dy + (size_t)b * Ho * Wo * Cout * ES;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int oy = iy + 1 - ky;
        if (oy < 0 || oy >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ox = ix + 1 - kx;
            if (ox < 0 || ox >= Wo) continue;
            const char* row = dyb + ((size_t)oy * Wo + ox) * Cout * ES;
            const float* wt = sw + (ky * 3 + kx) * Cout * NC;
            for (int co = 0; co < Cout; co += 8) {
                float v[8];
                load_row16<ES>(row, co, v);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[c] = fmaf(wt[(co + j) * NC + c], v[j], acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float* d = dst + ((size_t)c * B + b) * HW + r;   // [c][B][1][Hi][Wi]: every channel a contiguous [B,1,H,W] tensor of its own
        *d = accumulate ? *d + acc[c] : acc[c];
    }
}

inline unsigned nblk(size_t n) { return (unsigned)((n + NT - 1) / NT); }
inline bool head_dgrad_generic() { return TUNE(head_dgrad_generic) != 0; }   // A/B switch

}  // namespace
}  // namespace colvo

using namespace colvo;

#define DISPATCH_ES(dtype, CALL)                        \
    do {                                                \
        if ((dtype) == COLVO_F32) { constexpr int ES = 4; CALL; } \
        else { constexpr int ES = 2; CALL; }            \
    } while (0)

extern "C" int colvo_pack_weights(int dtype, const float* w_master, int Cout, int kk, int Cin, void* w_fwd,
                                  void* w_bwd, colvo_stream_t stream) {
    COLVO_CHECK_ARG(w_master && (w_fwd || w_bwd), "colvo_pack_weights: null pointer argument");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_pack_weights: bad dtype");
    COLVO_CHECK_ARG(Cout > 0 && kk > 0 && Cin > 0, "colvo_pack_weights: bad shape");
    const size_t n = (size_t)Cout * kk * Cin;
    DISPATCH_ES(dtype, colvo::launch((k_pack_weights<ES>), dim3(nblk(n)), dim3(NT), 0, (hipStream_t)stream,
                                          w_master, Cout, kk, Cin, w_fwd, w_bwd));
    COLVO_CHECK_LAUNCH("k_pack_weights");
    return 0;
}

extern "C" int colvo_pack_weights_multi(int dtype, const float* master, const void* table, int nlayers, int nblocks,
                                        void* fwd, void* bwd, colvo_stream_t stream) {
    COLVO_CHECK_ARG(master && table && bwd && nlayers >= 1 && nblocks >= 1, "colvo_pack_weights_multi: bad arguments");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_pack_weights_multi: bad dtype");
    DISPATCH_ES(dtype, colvo::launch((k_pack_weights_multi<ES>), dim3(nblocks), dim3(NT), 0, (hipStream_t)stream, master,
                                          (const PackEntry*)table, nlayers, fwd, bwd));
    COLVO_CHECK_LAUNCH("k_pack_weights_multi");
    return 0;
}

extern "C" int colvo_pack_nchw(int dtype, const float* const* src, const int32_t* src_channels, int nsrc, int B, int H,
                               int W, int Cpad, void* dst, colvo_stream_t stream) {
    COLVO_CHECK_ARG(src && src_channels && dst && nsrc >= 1 && nsrc <= 4, "colvo_pack_nchw: bad arguments");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_pack_nchw: bad dtype");
    Planes pl{};
    int tot = 0;
    for (int i = 0; i < nsrc; ++i) {
        COLVO_CHECK_ARG(src[i] && src_channels[i] > 0, "colvo_pack_nchw: null source %d", i);
        pl.p[i] = src[i]; pl.c[i] = src_channels[i]; tot += src_channels[i];
    }
    pl.n = nsrc;
    COLVO_CHECK_ARG(tot <= Cpad && Cpad % 8 == 0 && B >= 1 && B <= 65535, "colvo_pack_nchw: bad channel padding %d for %d", Cpad, tot);
    const size_t HW = (size_t)H * W;
    bool narrow = Cpad == 8;
    for (int i = 0; i < nsrc; ++i) narrow = narrow && src_channels[i] <= 4;
    if (narrow) {
        DISPATCH_ES(dtype, colvo::launch((k_pack_nchw8<ES>), dim3(nblk(HW), B), dim3(NT), 0, (hipStream_t)stream, pl,
                                              (int)HW, dst));
    } else {
        DISPATCH_ES(dtype, colvo::launch((k_pack_nchw<ES>), dim3(nblk(HW), B), dim3(NT), 0, (hipStream_t)stream, pl,
                                              (int)HW, Cpad, dst));
    }
    COLVO_CHECK_LAUNCH("k_pack_nchw");
    return 0;
}

extern "C" int colvo_pack_stem_pose(const float* frames, int B2, int H, int W, void* stem, void* pose_in, colvo_stream_t stream) {
    COLVO_CHECK_ARG(frames && stem && pose_in && B2 >= 2 && B2 % 2 == 0 && B2 <= 65534 && H >= 1 && W >= 1,
                    "colvo_pack_stem_pose: bad arguments (B2 = 2 * pairs images)");
    const size_t HW = (size_t)H * W;
    colvo::launch(k_pack_stem_pose, dim3(nblk(HW), B2 / 2), dim3(NT), 0, (hipStream_t)stream, frames, (int)HW, B2 / 2, stem, pose_in);
    COLVO_CHECK_LAUNCH("k_pack_stem_pose");
    return 0;
}

extern "C" int colvo_unpack_nhwc_grad(int dtype, const void* dsrc, int B, int H, int W, int Cpad, int c_begin,
                                      int c_count, float* dst_nchw, int accumulate, colvo_stream_t stream) {
    COLVO_CHECK_ARG(dsrc && dst_nchw && c_begin >= 0 && c_count >= 1 && c_begin + c_count <= Cpad && B >= 1 && B <= 65535,
                    "colvo_unpack_nhwc_grad: bad arguments");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_unpack_nhwc_grad: bad dtype");
    const size_t HW = (size_t)H * W;
    DISPATCH_ES(dtype, colvo::launch((k_unpack_nhwc<ES, 4>), dim3(nblk((HW + 3) / 4), B), dim3(NT), 0, (hipStream_t)stream,
                                          dsrc, (int)HW, Cpad, c_begin, c_count, dst_nchw, accumulate));
    COLVO_CHECK_LAUNCH("k_unpack_nhwc");
    return 0;
}

extern "C" int colvo_conv_dgrad_planes(const ColvoConvDesc* d, const void* dy, const float* w_master, int c_begin, int c_count,
                                       float* dst, int accumulate, colvo_stream_t stream) {
    COLVO_CHECK_ARG(d && dy && w_master && dst, "colvo_conv_dgrad_planes: null pointer argument");
    COLVO_CHECK_ARG(d->dtype == COLVO_F32 || d->dtype == COLVO_BF16, "colvo_conv_dgrad_planes: bad dtype %d", d->dtype);
    COLVO_CHECK_ARG(d->ksize == 3 && (d->stride == 1 || d->stride == 2) && d->C1 == 0 && !d->up0,
                    "colvo_conv_dgrad_planes: a 3x3 conv over one directly stored source, stride 1 or 2");
    COLVO_CHECK_ARG(d->Ho == (d->Hi - 1) / d->stride + 1 && d->Wo == (d->Wi - 1) / d->stride + 1 && d->B >= 1,
                    "colvo_conv_dgrad_planes: output %dx%d does not match input %dx%d / stride %d", d->Ho, d->Wo, d->Hi, d->Wi, d->stride);
    COLVO_CHECK_ARG(d->Cout >= 8 && d->Cout % 8 == 0 && d->Cout <= 128 && (c_count == 1 || c_count == 2 || c_count == 4) && c_begin >= 0 &&
                    c_begin + c_count <= d->C0,
                    "colvo_conv_dgrad_planes: Cout a multiple of 8 up to 128, 1 / 2 / 4 channels inside [0, C0) (Cout=%d, channels %d..%d of %d)",
                    d->Cout, c_begin, c_begin + c_count - 1, d->C0);
    if (d->dtype == COLVO_BF16 && d->stride == 2 && d->Cout == 16 && c_count == 2 && d->Hi % 2 == 0 && d->Wi % 2 == 0 &&
        (long long)d->B * d->Ho * d->Wo * 32 < 0x7fffffffLL && TUNE(planes_mfma) != 0) {
        // PoseNet's first layer in the training step: one small MFMA product per 2 x 2 block of input pixels (csrc/bwd16.hip)
        const int rc = colvo::launch_dgrad_planes_s2_mfma(dy, w_master, d->C0, c_begin, d->B, d->Hi, d->Wi, d->Ho, d->Wo, dst, accumulate,
                                                          (hipStream_t)stream);
        if (rc != 0) return rc;
        COLVO_CHECK_LAUNCH("k_dgrad_planes_s2_mfma");
        return 0;
    }
    const size_t npix = d->stride == 2 ? (size_t)d->B * d->Hi * ((d->Wi + 1) / 2) : (size_t)d->B * d->Hi * d->Wi;   // threads
    const size_t lds = (size_t)9 * d->Cout * c_count * 4;
    hipStream_t s = (hipStream_t)stream;
#define COLVO_DGP(ES_, NC_)                                                                                                        \
    colvo::launch((k_conv_dgrad_planes<ES_, NC_>), dim3(nblk(npix)), dim3(NT), lds, s, dy, w_master, d->Cout, d->C0, c_begin, d->B, \
                       d->Hi, d->Wi, d->Ho, d->Wo, d->stride, dst, accumulate)
    if (d->dtype == COLVO_F32) { if (c_count == 1) COLVO_DGP(4, 1); else if (c_count == 2) COLVO_DGP(4, 2); else COLVO_DGP(4, 4); }
    else { if (c_count == 1) COLVO_DGP(2, 1); else if (c_count == 2) COLVO_DGP(2, 2); else COLVO_DGP(2, 4); }
#undef COLVO_DGP
    COLVO_CHECK_LAUNCH("k_conv_dgrad_planes");
    return 0;
}

extern "C" int colvo_relu_bwd_inplace(int dtype, const void* y, void* dy, size_t n, colvo_stream_t stream) {
    COLVO_CHECK_ARG(y && dy, "colvo_relu_bwd_inplace: null pointer argument");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_relu_bwd_inplace: bad dtype");
    if (n == 0) return 0;
    DISPATCH_ES(dtype, colvo::launch((k_relu_bwd<ES>), dim3(nblk(n)), dim3(NT), 0, (hipStream_t)stream, y, dy, n));
    COLVO_CHECK_LAUNCH("k_relu_bwd");
    return 0;
}

extern "C" int colvo_depth_head_fwd(int dtype, const void* x, const float* w, const float* bias, int B, int H, int W,
                                    int C, float min_depth, float max_depth, float* depth, colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && w && bias && depth, "colvo_depth_head_fwd: null pointer argument");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_depth_head_fwd: bad dtype");
    COLVO_CHECK_ARG(B >= 1 && B <= 65535 && H >= 1 && W >= 1 && C >= 1 && C <= 1024 && min_depth > 0 && max_depth > min_depth,
                    "colvo_depth_head_fwd: bad shape / range");
    const size_t HW = (size_t)H * W;
    const int head_lds = (int)TUNE(head_fwd_lds);   // A/B switch
    if (C == 16 && dtype == COLVO_BF16 && head_lds && (H + 3) / 4 <= 65535) {
        colvo::launch(k_depth_head_fwd16_lds, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(NT), 0, (hipStream_t)stream, x, w,
                           bias, H, W, 1.0f / max_depth, 1.0f / min_depth, depth);
    } else if (C == 16) {
        DISPATCH_ES(dtype, colvo::launch((k_depth_head_fwd16<ES>), dim3(nblk(HW), B), dim3(NT), 0, (hipStream_t)stream, x,
                                              w, bias, H, W, 1.0f / max_depth, 1.0f / min_depth, depth));
    } else {
        DISPATCH_ES(dtype, colvo::launch((k_depth_head_fwd<ES>), dim3(nblk(HW), B), dim3(NT), 9 * C * sizeof(float),
                                              (hipStream_t)stream, x, w, bias, H, W, C, 1.0f / max_depth, 1.0f / min_depth, depth));
    }
    COLVO_CHECK_LAUNCH("k_depth_head_fwd");
    return 0;
}

// scratch: B*H*W floats (the d(pre) plane), caller-provided.
extern "C" int colvo_depth_head_bwd(int dtype, const void* x, const float* w, const float* depth, const float* d_depth,
                                    int B, int H, int W, int C, float min_depth, float max_depth, float* scratch,
                                    void* dx, float* dw, float* db, colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && w && depth && d_depth && scratch && dx && ((dw == nullptr) == (db == nullptr)),
                    "colvo_depth_head_bwd: null pointer argument");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_depth_head_bwd: bad dtype");
    COLVO_CHECK_ARG(B >= 1 && B <= 65535 && H >= 1 && W >= 1 && C >= 1 && C <= 1024 && min_depth > 0 && max_depth > min_depth,
                    "colvo_depth_head_bwd: bad shape / range");
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)H * W, n = (size_t)B * HW;
    const float lo = 1.0f / max_depth, hi = 1.0f / min_depth;
    colvo::launch(k_depth_head_dpre, dim3(nblk(n)), dim3(NT), 0, s, depth, d_depth, n, lo, hi, scratch);
    COLVO_CHECK_LAUNCH("k_depth_head_dpre");
    if (dw) {
        if (int e = colvo_depth_head_wgrad(dtype, x, scratch, B, H, W, C, dw, db, stream)) return e;
    }
    if (C == 16 && ((uintptr_t)x | (uintptr_t)dx) % 16 == 0 && !head_dgrad_generic()) {
        DISPATCH_ES(dtype, colvo::launch((k_depth_head_dgrad16<ES>), dim3(nblk(HW), B), dim3(NT), 0, s, x, w, scratch, H, W, dx));
    } else {
        DISPATCH_ES(dtype, colvo::launch((k_depth_head_dgrad<ES>), dim3(nblk(HW), B), dim3(NT), 9 * C * sizeof(float), s,
                                              x, w, scratch, H, W, C, dx));
    }
    COLVO_CHECK_LAUNCH("k_depth_head_dgrad");
    return 0;
}

extern "C" int colvo_depth_head_bwd_parts(int dtype, const void* x, const float* w, const float* depth, const float* g_first,
                                          const float* g_second, const float* g_raw, const float* g_raw_second,
                                          const float* scale_a, const float* scale_b, int B, int H, int W, int C,
                                          float min_depth, float max_depth, float* scratch, void* dx, float* dw, float* db,
                                          colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && w && depth && scratch && ((dw == nullptr) == (db == nullptr)),
                    "colvo_depth_head_bwd_parts: null pointer argument");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_depth_head_bwd_parts: bad dtype");
    COLVO_CHECK_ARG(B >= 2 && B % 2 == 0 && B <= 65534 && H >= 1 && W >= 1 && C >= 1 && C <= 1024 && min_depth > 0 &&
                        max_depth > min_depth,
                    "colvo_depth_head_bwd_parts: bad shape / range (B = 2*Bh images)");
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)H * W, n = (size_t)B * HW;
    const float lo = 1.0f / max_depth, hi = 1.0f / min_depth;
    colvo::launch(k_depth_head_dpre_parts, dim3(nblk(n)), dim3(NT), 0, s, depth, g_first, g_second, g_raw, g_raw_second,
                       scale_a, scale_b, n / 2, lo, hi, scratch);
    COLVO_CHECK_LAUNCH("k_depth_head_dpre_parts");
    if (dw) {
        if (int e = colvo_depth_head_wgrad(dtype, x, scratch, B, H, W, C, dw, db, stream)) return e;
    }
    if (!dx) return 0;          // d(pre) only: the input gradient is made by colvo_conv_bwd_fused's HEAD form from `scratch`
    if (C == 16 && ((uintptr_t)x | (uintptr_t)dx) % 16 == 0 && !head_dgrad_generic()) {
        DISPATCH_ES(dtype, colvo::launch((k_depth_head_dgrad16<ES>), dim3(nblk(HW), B), dim3(NT), 0, s, x, w, scratch, H, W, dx));
    } else {
        DISPATCH_ES(dtype, colvo::launch((k_depth_head_dgrad<ES>), dim3(nblk(HW), B), dim3(NT), 9 * C * sizeof(float), s,
                                              x, w, scratch, H, W, C, dx));
    }
    COLVO_CHECK_LAUNCH("k_depth_head_dgrad");
    return 0;
}

// The weight / bias gradient alone, from the d(pre) plane colvo_depth_head_bwd left in `scratch` (so that it can run on
// another stream than the input gradient).
static int depth_head_wgrad_impl(int dtype, const void* x, const float* dpre, int B, int H, int W, int C, float* dw, float* db,
                                 float* partials, size_t partial_bytes, colvo_stream_t stream);

extern "C" int colvo_depth_head_wgrad(int dtype, const void* x, const float* dpre, int B, int H, int W, int C, float* dw,
                                      float* db, colvo_stream_t stream) {
    return depth_head_wgrad_impl(dtype, x, dpre, B, H, W, C, dw, db, nullptr, 0, stream);
}

// pixels per workgroup of the C = 16 kernel: a multiple of 256, at least 8 per thread, and at most ~512 workgroups (two per CU
// resident)
static int head_wgrad_ppb(size_t HW, int B) {
    int ppb = 2048;
    while ((HW + ppb - 1) / ppb * B > 512 && ppb < 16384) ppb += 256;
    return ppb;
}

extern "C" size_t colvo_depth_head_wgrad_scratch_bytes(int B, int H, int W, int C) {
    if (C != 16 || B < 1 || H < 1 || W < 1) return 0;
    const size_t HW = (size_t)H * W;
    const int ppb = head_wgrad_ppb(HW, B);
    return (size_t)B * ((HW + ppb - 1) / ppb) * (9 * C + 1) * sizeof(float);
}

extern "C" int colvo_depth_head_wgrad_det(int dtype, const void* x, const float* dpre, int B, int H, int W, int C, float* dw,
                                          float* db, void* scratch, size_t scratch_bytes, colvo_stream_t stream) {
    COLVO_CHECK_ARG(scratch && C == 16, "colvo_depth_head_wgrad_det: needs scratch and the 16-channel head");
    return depth_head_wgrad_impl(dtype, x, dpre, B, H, W, C, dw, db, (float*)scratch, scratch_bytes, stream);
}

extern "C" int colvo_depth_head_wgrad_reduce(const float* partials, int rows, float* dw, float* db, colvo_stream_t stream) {
    COLVO_CHECK_ARG(partials && dw && db && rows >= 1, "colvo_depth_head_wgrad_reduce: bad arguments");
    colvo::launch(k_head_wgrad_reduce, dim3(9 * 16 + 1), dim3(NT), 0, (hipStream_t)stream, partials, rows, 9 * 16 + 1, dw, db);
    COLVO_CHECK_LAUNCH("k_head_wgrad_reduce");
    return 0;
}

static int depth_head_wgrad_impl(int dtype, const void* x, const float* dpre, int B, int H, int W, int C, float* dw, float* db,
                                 float* partials, size_t partial_bytes, colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && dpre && dw && db, "colvo_depth_head_wgrad: null pointer argument");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_depth_head_wgrad: bad dtype");
    COLVO_CHECK_ARG(B >= 1 && B <= 65535 && H >= 1 && W >= 1 && C >= 1 && C <= 1024, "colvo_depth_head_wgrad: bad shape");
    hipStream_t s = (hipStream_t)stream;
    const size_t HW = (size_t)H * W;
    if (C == 16) {
        const int ppb = head_wgrad_ppb(HW, B);
        const int rows_tab = (int)(B * ((HW + ppb - 1) / ppb));
        COLVO_CHECK_ARG(!partials || (size_t)rows_tab * (9 * 16 + 1) * sizeof(float) <= partial_bytes,
                        "colvo_depth_head_wgrad_det: scratch too small (colvo_depth_head_wgrad_scratch_bytes)");
        // tap rows per thread: 1 (three workgroups per pixel range) measured 57 -> 45 us inside the step, step -1 %
        const int rows = (int)TUNE(head_wgrad_rows);   // A/B switch
        if (rows == 3) {
            DISPATCH_ES(dtype, colvo::launch((k_depth_head_wgrad<ES, 16, 3>), dim3((unsigned)((HW + ppb - 1) / ppb), B),
                                                  dim3(NT), 0, s, x, dpre, H, W, ppb, dw, db, partials));
        } else {
            DISPATCH_ES(dtype, colvo::launch((k_depth_head_wgrad<ES, 16, 1>), dim3((unsigned)((HW + ppb - 1) / ppb), B, 3),
                                                  dim3(NT), 0, s, x, dpre, H, W, ppb, dw, db, partials));
        }
        if (partials)
            colvo::launch(k_head_wgrad_reduce, dim3(9 * 16 + 1), dim3(NT), 0, s, (const float*)partials, rows_tab, 9 * 16 + 1, dw, db);
    } else {
        const int rows = 4;
        DISPATCH_ES(dtype, colvo::launch((k_depth_head_wgrad_generic<ES>), dim3((H + rows - 1) / rows, B), dim3(NT),
                                              0, s, x, dpre, H, W, C, rows, dw, db));
    }
    COLVO_CHECK_LAUNCH("k_depth_head_wgrad");
    return 0;
}

extern "C" int colvo_pose_head_fwd(int dtype, const void* x, const float* w, const float* bias, int B, int HW, int C,
                                   float pose_scale, float lcc_scale, float* out, colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && w && bias && out && B >= 1 && HW >= 1 && C >= 1, "colvo_pose_head_fwd: bad arguments");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_pose_head_fwd: bad dtype");
    DISPATCH_ES(dtype, colvo::launch((k_pose_head_fwd<ES>), dim3(B), dim3(NT), 0, (hipStream_t)stream, x, w, bias, HW,
                                          C, pose_scale, lcc_scale, out));
    COLVO_CHECK_LAUNCH("k_pose_head_fwd");
    return 0;
}

extern "C" int colvo_pose_head_bwd(int dtype, const void* x, const float* w, const float* d_pose, const float* d_a,
                                   const float* d_b, const float* scale_a, const float* scale_b, int B, int HW, int C,
                                   float pose_scale, float lcc_scale, void* dx, float* dw, float* db, colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && w && dx && dw && db && B >= 1 && HW >= 1 && C >= 1, "colvo_pose_head_bwd: bad arguments");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_pose_head_bwd: bad dtype");
    DISPATCH_ES(dtype, colvo::launch((k_pose_head_bwd<ES>), dim3(B), dim3(NT), 0, (hipStream_t)stream, x, w, d_pose, d_a,
                                          d_b, scale_a, scale_b, HW, C, pose_scale, lcc_scale, dx, dw, db));
    COLVO_CHECK_LAUNCH("k_pose_head_bwd");
    return 0;
}

extern "C" int colvo_pose_head_bwd_det(int dtype, const void* x, const float* w, const float* d_pose, const float* d_a,
                                       const float* d_b, const float* scale_a, const float* scale_b, int B, int HW, int C,
                                       float pose_scale, float lcc_scale, void* dx, float* dw, float* db, colvo_stream_t stream) {
    COLVO_CHECK_ARG(x && w && dx && dw && db && B >= 1 && HW >= 1 && C >= 8, "colvo_pose_head_bwd_det: bad arguments");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_pose_head_bwd_det: bad dtype");
    DISPATCH_ES(dtype, colvo::launch((k_pose_head_bwd_det<ES>), dim3((C + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, x, w,
                                          d_pose, d_a, d_b, scale_a, scale_b, B, HW, C, pose_scale, lcc_scale, dx, dw, db));
    COLVO_CHECK_LAUNCH("k_pose_head_bwd_det");
    return 0;
}

extern "C" int colvo_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                               float beta1, float beta2, float eps, float grad_scale, int32_t* step_count,
                               colvo_stream_t stream) {
    COLVO_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_count, "colvo_adam_step: null pointer argument");
    COLVO_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                    "colvo_adam_step: arenas must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (n) {
        unsigned blocks = nblk((n + 3) / 4);
        if (blocks > 4096) blocks = 4096;
        colvo::launch(k_adam, dim3(blocks), dim3(NT), 0, s, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2,
                           eps, grad_scale, step_count, 0);
        COLVO_CHECK_LAUNCH("k_adam");
    }
    colvo::launch(k_inc_step, dim3(1), dim3(1), 0, s, step_count);
    COLVO_CHECK_LAUNCH("k_inc_step");
    return 0;
}

// The same with the step number t (1-based) supplied by the host: no device counter, no second launch.  For callers that count
// steps themselves; a step captured into a hipGraph needs the device counter of colvo_adam_step (a baked-in t would repeat).
extern "C" int colvo_adam_step_t(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                                 float beta1, float beta2, float eps, float grad_scale, int t, colvo_stream_t stream) {
    COLVO_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && t >= 1, "colvo_adam_step_t: bad arguments");
    COLVO_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
                    "colvo_adam_step_t: arenas must be 16-byte aligned");
    if (n == 0) return 0;
    unsigned blocks = nblk((n + 3) / 4);
    if (blocks > 4096) blocks = 4096;
    colvo::launch(k_adam, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, grad_scale, (const int32_t*)nullptr, t);
    COLVO_CHECK_LAUNCH("k_adam");
    return 0;
}

extern "C" int colvo_adam_step_multi(const ColvoAdamArena* arenas, int count, float lr, float beta1, float beta2, float eps,
                                     float grad_scale, int t, colvo_stream_t stream) {
    COLVO_CHECK_ARG(arenas && count >= 1 && count <= COLVO_MAX_ARENAS && t >= 1, "colvo_adam_step_multi: bad arguments");
    AdamArenas as{};
    as.count = count;
    unsigned total = 0;
    for (int i = 0; i < count; ++i) {
        const ColvoAdamArena& a = arenas[i];
        COLVO_CHECK_ARG(a.param && a.grad && a.exp_avg && a.exp_avg_sq, "colvo_adam_step_multi: null pointer in arena %d", i);
        COLVO_CHECK_ARG(((uintptr_t)a.param | (uintptr_t)a.grad | (uintptr_t)a.exp_avg | (uintptr_t)a.exp_avg_sq) % 16 == 0,
                        "colvo_adam_step_multi: arenas must be 16-byte aligned");
        unsigned blocks = nblk((a.n + 3) / 4);
        if (blocks > 4096) blocks = 4096;
        if (blocks == 0) blocks = 1;
        as.a[i] = a;
        as.first[i] = total;
        total += blocks;
    }
    for (int i = count; i <= COLVO_MAX_ARENAS; ++i) as.first[i] = total;
    colvo::launch(k_adam_multi, dim3(total), dim3(NT), 0, (hipStream_t)stream, as, lr, beta1, beta2, eps, grad_scale, t);
    COLVO_CHECK_LAUNCH("k_adam_multi");
    return 0;
}

extern "C" int colvo_adam_pack_step(int dtype, const void* table, int nentries, int nblocks, float lr, float beta1, float beta2,
                                    float eps, float grad_scale, int32_t* step_count, int t, colvo_stream_t stream) {
    return colvo_adam_pack_step_scaled(dtype, table, nentries, nblocks, lr, beta1, beta2, eps, grad_scale, nullptr, step_count, t, stream);
}

extern "C" int colvo_adam_pack_step_scaled(int dtype, const void* table, int nentries, int nblocks, float lr, float beta1, float beta2,
                                           float eps, float grad_scale, const float* grad_scale_dev, int32_t* step_count, int t,
                                           colvo_stream_t stream) {
    COLVO_CHECK_ARG(table && nentries >= 1 && nblocks >= 1 && (step_count || t >= 1), "colvo_adam_pack_step: bad arguments");
    COLVO_CHECK_ARG(dtype == COLVO_F32 || dtype == COLVO_BF16, "colvo_adam_pack_step: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_ES(dtype, colvo::launch((k_adam_pack<ES>), dim3(nblocks), dim3(NT), 0, s, (const ColvoAdamPackEntry*)table,
                                          nentries, lr, beta1, beta2, eps, grad_scale, grad_scale_dev, (const int32_t*)step_count, t));
    COLVO_CHECK_LAUNCH("k_adam_pack");
    if (step_count) {
        colvo::launch(k_inc_step, dim3(1), dim3(1), 0, s, step_count);
        COLVO_CHECK_LAUNCH("k_inc_step");
    }
    return 0;
}

extern "C" int colvo_zero_multi(void* const* ptrs, const size_t* bytes, int count, colvo_stream_t stream) {
    COLVO_CHECK_ARG(ptrs && bytes && count >= 1 && count <= COLVO_MAX_ARENAS, "colvo_zero_multi: bad arguments");
    ZeroArenas zs{};
    zs.count = count;
    unsigned total = 0;
    for (int i = 0; i < count; ++i) {
        COLVO_CHECK_ARG(ptrs[i] && (uintptr_t)ptrs[i] % 16 == 0 && bytes[i] % 16 == 0,
                        "colvo_zero_multi: buffer %d must be 16-byte aligned and a multiple of 16 bytes long", i);
        zs.p[i] = ptrs[i];
        zs.n16[i] = bytes[i] / 16;
        unsigned blocks = nblk((zs.n16[i] + 3) / 4);          // four 16-byte stores per thread
        if (blocks > 2048) blocks = 2048;
        if (blocks == 0) blocks = 1;
        zs.first[i] = total;
        total += blocks;
    }
    for (int i = count; i <= COLVO_MAX_ARENAS; ++i) zs.first[i] = total;
    colvo::launch(k_zero_multi, dim3(total), dim3(NT), 0, (hipStream_t)stream, zs);
    COLVO_CHECK_LAUNCH("k_zero_multi");
    return 0;
}

// ---- gradient transport in bf16 (ddp.GradBuckets(transport_dtype=bfloat16)): fp32 slice <-> bf16 staging slice ----
// (round 4 staged through Tensor.copy_, i.e. at::native kernels inside the data-parallel step; same RNE rounding here)
__global__ __launch_bounds__(NT) void k_cast_f32_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * NT * 4;
    for (size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4*>(src + i);
            uint2 o;
            o.x = pack2bf(v.x, v.y);
            o.y = pack2bf(v.z, v.w);
            *reinterpret_cast<uint2*>(dst + i) = o;
        } else {
            for (size_t j = i; j < n; ++j) dst[j] = f2bf(src[j]);
        }
    }
}
__global__ __launch_bounds__(NT) void k_cast_bf16_f32(const uint16_t* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * NT * 4;
    for (size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const uint2 v = *reinterpret_cast<const uint2*>(src + i);
            *reinterpret_cast<float4*>(dst + i) = make_float4(bf2f((uint16_t)(v.x & 0xFFFFu)), bf2f((uint16_t)(v.x >> 16)),
                                                              bf2f((uint16_t)(v.y & 0xFFFFu)), bf2f((uint16_t)(v.y >> 16)));
        } else {
            for (size_t j = i; j < n; ++j) dst[j] = bf2f(src[j]);
        }
    }
}

extern "C" int colvo_cast_f32_bf16(const float* src, void* dst, size_t n, int to_bf16, colvo_stream_t stream) {
    COLVO_CHECK_ARG(src && dst, "colvo_cast_f32_bf16: null pointer argument");
    // (to_bf16 = 0: `src` is the bf16 buffer and `dst` the fp32 one -- the argument order stays source, destination)
    COLVO_CHECK_ARG((uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0, "colvo_cast_f32_bf16: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    unsigned blocks = nblk((n + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    if (to_bf16) colvo::launch(k_cast_f32_bf16, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, src, (uint16_t*)dst, n);
    else colvo::launch(k_cast_bf16_f32, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, (const uint16_t*)src, (float*)dst, n);
    COLVO_CHECK_LAUNCH("k_cast_f32_bf16");
    return 0;
}

extern "C" int colvo_zero(void* ptr, size_t bytes, colvo_stream_t stream) {
    COLVO_CHECK_ARG(ptr || bytes == 0, "colvo_zero: null pointer argument");
    if (bytes == 0) return 0;
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("colvo_zero: hipMemsetAsync failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
