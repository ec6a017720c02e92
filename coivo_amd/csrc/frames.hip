// frames.hip -- the data format in front of the path: decoded camera frames arrive as interleaved
// 8-bit RGB at the camera's resolution; the networks and the loss take planar fp32 in [0,1] at the training resolution.
// One kernel: bilinear resize (half-pixel centres, the torch `interpolate(align_corners=False)` convention the oracle
// uses) + HWC->CHW + /255.  Spec: oracle/colvo_spec.py resize_frames_u8 (oracle/SPEC.md §6d).  HBM-bound: the source
// frame is read once through L2 (4 taps per output pixel, neighbours share them), 12 B written per output pixel.
#include "common.h"

namespace colvo {
namespace {

constexpr int NT = 256;

// grid (ceil(W/64), ceil(H/4), B): a 64x4 output tile per workgroup, one output pixel (3 channels) per thread
__global__ __launch_bounds__(NT) void k_frames_u8_to_f32(const uint8_t* __restrict__ src, int h, int w, int H, int W,
                                                         float sy, float sx, float* __restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int b = blockIdx.z;
    // source index = scale * (dst + 0.5) - 0.5, clamped at 0 (upsample_bilinear2d, align_corners=False)
    const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.0f);
    const float fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.0f);
    const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const uint8_t* img = src + (size_t)b * h * w * 3;
    const uint8_t* p00 = img + ((size_t)y0 * w + x0) * 3;
    const uint8_t* p01 = img + ((size_t)y0 * w + x1) * 3;
    const uint8_t* p10 = img + ((size_t)y1 * w + x0) * 3;
    const uint8_t* p11 = img + ((size_t)y1 * w + x1) * 3;
    const size_t plane = (size_t)H * W;
    float* o = dst + (size_t)b * 3 * plane + (size_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = hy * (hx * (float)p00[c] + lx * (float)p01[c]) + ly * (hx * (float)p10[c] + lx * (float)p11[c]);
        o[c * plane] = v / 255.0f;
    }
}

}  // namespace
}  // namespace colvo

using namespace colvo;

extern "C" int colvo_frames_u8_to_f32(const uint8_t* frames, int B, int h, int w, int H, int W, float* out,
                                      colvo_stream_t stream) {
    COLVO_CHECK_ARG(frames && out, "colvo_frames_u8_to_f32: null pointer argument");
    COLVO_CHECK_ARG(B > 0 && B <= 65535 && h > 0 && w > 0 && H > 0 && W > 0 && (long long)h * w < (1ll << 28) &&
                        (long long)H * W < (1ll << 28) && (H + 3) / 4 <= 65535,
                    "colvo_frames_u8_to_f32: bad shape B=%d %dx%d -> %dx%d", B, h, w, H, W);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipLaunchKernelGGL(k_frames_u8_to_f32, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(NT), 0, (hipStream_t)stream, frames,
                       h, w, H, W, sy, sx, out);
    COLVO_CHECK_LAUNCH("k_frames_u8_to_f32");
    return 0;
}
