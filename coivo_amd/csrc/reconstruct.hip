// reconstruct.hip -- the inference side of the path: dense depth maps back-projected through the
// integrated camera trajectory into one world-frame point cloud (README.md:9 "complete 3D reconstruction of the
// intestine", :29 "stitching together the dense depth maps of each frame using the colonoscopic trajectory").
// Spec: oracle/colvo_spec.py backproject / stitch_point_cloud (oracle/SPEC.md §6c).  All HBM-bound: 4 B read and
// 12 B written per pixel; the stitched cloud keeps the oracle's order (frame-major, row-major) through a per-block count,
// one scan over the block counts and an ordered in-block compaction -- no atomics, so the output is deterministic.
#include "common.h"

namespace colvo {
namespace {

constexpr int NT = 256;

struct Cam {          // per frame: intrinsics and the camera-to-world transform
    float fx, fy, cx, cy;
    float r[9];
    float t[3];
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ K, const float* __restrict__ M, int b) {
    Cam c;
    const float* k = K + (size_t)b * 9;
    const float* m = M + (size_t)b * 16;
    c.fx = uniform_f(k[0]);
    c.fy = uniform_f(k[4]);
    c.cx = uniform_f(k[2]);
    c.cy = uniform_f(k[5]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) c.r[i * 3 + j] = uniform_f(m[i * 4 + j]);
        c.t[i] = uniform_f(m[i * 4 + 3]);
    }
    return c;
}

// Same operation order as the oracle: ((u - cx) / fx) * d, then R p + t as a dot product in x, y, z order.
__device__ __forceinline__ void world_point(const Cam& c, float u, float v, float d, float& X, float& Y, float& Z) {
    const float px = (u - c.cx) / c.fx * d;
    const float py = (v - c.cy) / c.fy * d;
    X = c.r[0] * px + c.r[1] * py + c.r[2] * d + c.t[0];
    Y = c.r[3] * px + c.r[4] * py + c.r[5] * d + c.t[1];
    Z = c.r[6] * px + c.r[7] * py + c.r[8] * d + c.t[2];
}

// grid (ceil(H*W / NT), B).  Lane i of a wave owns pixel i; the three coordinates of 64 consecutive points are 768
// contiguous bytes, written as three fully coalesced 256-B rows after a transpose through LDS.
__global__ __launch_bounds__(NT) void k_backproject(const float* __restrict__ depth, const float* __restrict__ K,
                                                    const float* __restrict__ M, int H, int W,
                                                    float* __restrict__ points) {
    __shared__ float sm[NT * 3];
    const int b = blockIdx.y;
    const int HW = H * W;
    const int base = blockIdx.x * NT;
    const int p = base + threadIdx.x;
    const Cam c = load_cam(K, M, b);
    if (p < HW) {
        const int v = p / W, u = p - v * W;
        float X, Y, Z;
        world_point(c, (float)u, (float)v, depth[(size_t)b * HW + p], X, Y, Z);
        sm[threadIdx.x * 3 + 0] = X;
        sm[threadIdx.x * 3 + 1] = Y;
        sm[threadIdx.x * 3 + 2] = Z;
    }
    __syncthreads();
    const int nvalid = min(NT, HW - base) * 3;
    float* out = points + ((size_t)b * HW + base) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = k * NT + threadIdx.x;
        if (i < nvalid) out[i] = sm[i];
    }
}

// ---- stitched cloud: strided pixels of N frames, depth < max_depth, compacted in order ---------------------------- //
struct StitchGeom {
    int H, W, stride, Hs, Ws, per_frame, blocks_per_frame;
};

__device__ __forceinline__ bool stitch_pixel(const StitchGeom& g, int idx, int& u, int& v) {
    if (idx >= g.per_frame) return false;
    const int r = idx / g.Ws;
    v = r * g.stride;
    u = (idx - r * g.Ws) * g.stride;
    return true;
}

// grid (blocks_per_frame, N): counts[frame * blocks_per_frame + block] = kept pixels of this block
__global__ __launch_bounds__(NT) void k_stitch_count(const float* __restrict__ depth, StitchGeom g, float max_depth,
                                                     int32_t* __restrict__ counts) {
    __shared__ int wsum[NT / 64];
    int u = 0, v = 0;
    const bool in = stitch_pixel(g, blockIdx.x * NT + threadIdx.x, u, v);
    const bool keep = in && depth[((size_t)blockIdx.y * g.H + v) * g.W + u] < max_depth;
    const int n = __popcll(__ballot(keep));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.y * g.blocks_per_frame + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup: exclusive scan of `n` block counts in place; total -> *total
__global__ __launch_bounds__(NT) void k_stitch_scan(int32_t* __restrict__ counts, int n, int32_t* __restrict__ total) {
    __shared__ int part[NT];
    const int per = (n + NT - 1) / NT;
    const int lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < NT; ++i) { const int t = part[i]; part[i] = run; run += t; }
        *total = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int t = counts[i]; counts[i] = run; run += t; }
}

__global__ __launch_bounds__(NT) void k_stitch_write(const float* __restrict__ depth, const float* __restrict__ K,
                                                     const float* __restrict__ M, StitchGeom g, float max_depth,
                                                     const int32_t* __restrict__ offsets, float* __restrict__ points) {
    __shared__ int wsum[NT / 64];
    const int b = blockIdx.y;
    int u = 0, v = 0;
    const bool in = stitch_pixel(g, blockIdx.x * NT + threadIdx.x, u, v);
    const float d = in ? depth[((size_t)b * g.H + v) * g.W + u] : 0.0f;
    const bool keep = in && d < max_depth;
    const unsigned long long mask = __ballot(keep);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wsum[wv] = __popcll(mask);
    __syncthreads();
    int before = offsets[b * g.blocks_per_frame + blockIdx.x];
    for (int i = 0; i < wv; ++i) before += wsum[i];
    if (!keep) return;
    const int slot = before + __popcll(mask & ((1ull << lane) - 1ull));
    const Cam c = load_cam(K, M, b);
    float X, Y, Z;
    world_point(c, (float)u, (float)v, d, X, Y, Z);
    float* o = points + (size_t)slot * 3;
    o[0] = X; o[1] = Y; o[2] = Z;
}

bool stitch_geom(int H, int W, int stride, StitchGeom& g) {
    if (H <= 0 || W <= 0 || stride <= 0) return false;
    g.H = H; g.W = W; g.stride = stride;
    g.Hs = (H + stride - 1) / stride;
    g.Ws = (W + stride - 1) / stride;
    g.per_frame = g.Hs * g.Ws;
    g.blocks_per_frame = (g.per_frame + NT - 1) / NT;
    return true;
}

}  // namespace
}  // namespace colvo

using namespace colvo;

extern "C" int colvo_backproject(const float* depth, const float* K, const float* cam2world, int B, int H, int W,
                                 float* points, colvo_stream_t stream) {
    COLVO_CHECK_ARG(depth && K && cam2world && points, "colvo_backproject: null pointer argument");
    COLVO_CHECK_ARG(B > 0 && H > 0 && W > 0 && B <= 65535 && (long long)H * W < (1ll << 30),
                    "colvo_backproject: bad shape B=%d H=%d W=%d", B, H, W);
    colvo::launch(k_backproject, dim3((H * W + NT - 1) / NT, B), dim3(NT), 0, (hipStream_t)stream, depth, K,
                       cam2world, H, W, points);
    COLVO_CHECK_LAUNCH("k_backproject");
    return 0;
}

extern "C" size_t colvo_stitch_workspace_ints(int N, int H, int W, int stride) {
    StitchGeom g;
    if (N <= 0 || !stitch_geom(H, W, stride, g)) return 0;
    return (size_t)N * g.blocks_per_frame;
}

extern "C" int colvo_stitch_point_cloud(const float* depths, const float* K, const float* cam2world, int N, int H, int W,
                                        int stride, float max_depth, int32_t* workspace, float* points,
                                        int32_t* n_points, colvo_stream_t stream) {
    COLVO_CHECK_ARG(depths && K && cam2world && workspace && points && n_points,
                    "colvo_stitch_point_cloud: null pointer argument");
    StitchGeom g;
    COLVO_CHECK_ARG(N > 0 && N <= 65535 && stitch_geom(H, W, stride, g) && (long long)H * W < (1ll << 30) &&
                        (long long)N * g.blocks_per_frame < (1ll << 30),
                    "colvo_stitch_point_cloud: bad shape N=%d H=%d W=%d stride=%d", N, H, W, stride);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(g.blocks_per_frame, N);
    colvo::launch(k_stitch_count, grid, dim3(NT), 0, s, depths, g, max_depth, workspace);
    COLVO_CHECK_LAUNCH("k_stitch_count");
    colvo::launch(k_stitch_scan, dim3(1), dim3(NT), 0, s, workspace, N * g.blocks_per_frame, n_points);
    COLVO_CHECK_LAUNCH("k_stitch_scan");
    colvo::launch(k_stitch_write, grid, dim3(NT), 0, s, depths, K, cam2world, g, max_depth, workspace, points);
    COLVO_CHECK_LAUNCH("k_stitch_write");
    return 0;
}
