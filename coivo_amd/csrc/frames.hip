// frames.hip -- the data format in front of the path: decoded camera frames arrive as interleaved
// 8-bit RGB at the camera's resolution; the networks and the loss take planar fp32 in [0,1] at the training resolution.
// One kernel: bilinear resize (half-pixel centres, the torch `interpolate(align_corners=False)` convention the oracle
// uses) + HWC->CHW + /255.  Spec: oracle/colvo_spec.py resize_frames_u8 (oracle/SPEC.md §6d).  HBM-bound: the source
// frame is read once through L2 (4 taps per output pixel, neighbours share them), 12 B written per output pixel.
#include "common.h"

#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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
    colvo::launch(k_frames_u8_to_f32, dim3((W + 63) / 64, (H + 3) / 4, B), dim3(NT), 0, (hipStream_t)stream, frames,
                       h, w, H, W, sy, sx, out);
    COLVO_CHECK_LAUNCH("k_frames_u8_to_f32");
    return 0;
}

// ---- host side of the input pipeline: raw frames from disk into the (pinned) staging buffer -------------------------------- //
// A `.npy` file of an [h,w,3] uint8 C-ordered array is a short text header followed by the frame itself.  Reading 16 of them per
// batch through the interpreter (open / parse / read per file under its lock) capped the loader at ~4500 pairs/s, below the
// training step's rate; here the header is checked in C and the payloads are read with pread() by a few threads, straight into
// the caller's buffer (data.PairLoader hands over its pinned staging buffer), with the interpreter lock released for the call.
namespace colvo {
namespace {

// 0 and *data_off on success; otherwise a message in `err`
int npy_u8_frame_offset(int fd, const char* path, int h, int w, long long* data_off, std::string* err) {
    unsigned char head[12];
    if (pread(fd, head, 12, 0) != 12 || memcmp(head, "\x93NUMPY", 6) != 0) { *err = std::string(path) + ": not a .npy file"; return 1; }
    const int major = head[6];
    size_t hlen, hoff;
    if (major == 1) { hlen = head[8] | (head[9] << 8); hoff = 10; }
    else if (major == 2 || major == 3) { hlen = (size_t)head[8] | ((size_t)head[9] << 8) | ((size_t)head[10] << 16) | ((size_t)head[11] << 24); hoff = 12; }
    else { *err = std::string(path) + ": unsupported .npy version"; return 1; }
    if (hlen == 0 || hlen > 65536) { *err = std::string(path) + ": bad .npy header length"; return 1; }
    std::string hd(hlen, '\0');
    if (pread(fd, &hd[0], hlen, (off_t)hoff) != (ssize_t)hlen) { *err = std::string(path) + ": truncated .npy header"; return 1; }
    auto has = [&](const char* a) { return hd.find(a) != std::string::npos; };
    const bool u8 = has("'descr': '|u1'") || has("'descr': 'u1'") || has("'descr': '<u1'");
    const bool c_order = has("'fortran_order': False");
    long long dims[4] = {0, 0, 0, 0};
    int nd = 0;
    const size_t sp = hd.find("'shape': (");
    if (sp != std::string::npos) {
        const char* q = hd.c_str() + sp + 10;
        while (*q && *q != ')' && nd < 4) {
            while (*q == ' ' || *q == ',') ++q;
            if (*q < '0' || *q > '9') break;
            long long v = 0;
            while (*q >= '0' && *q <= '9') v = v * 10 + (*q++ - '0');
            dims[nd++] = v;
        }
    }
    if (!u8 || !c_order || nd != 3 || dims[0] != h || dims[1] != w || dims[2] != 3) {
        *err = std::string(path) + ": expected an 8-bit RGB frame [" + std::to_string(h) + "," + std::to_string(w) + ",3] in C order, header says " + hd;
        return 1;
    }
    *data_off = (long long)(hoff + hlen);
    return 0;
}

}  // namespace
}  // namespace colvo

extern "C" int colvo_read_npy_u8_frames(const char* const* paths, int n, int h, int w, uint8_t* dst, int nthreads) {
    COLVO_CHECK_ARG(paths && dst && n >= 1 && h > 0 && w > 0 && (long long)h * w < (1ll << 28),
                    "colvo_read_npy_u8_frames: bad arguments");
    const size_t frame = (size_t)h * w * 3;
    std::atomic<int> next{0}, failed{0};
    std::string first_error;
    std::mutex mu;
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            std::string err;
            const int fd = open(paths[i], O_RDONLY | O_CLOEXEC);
            if (fd < 0) err = std::string(paths[i]) + ": " + strerror(errno);
            long long off = 0;
            if (err.empty() && npy_u8_frame_offset(fd, paths[i], h, w, &off, &err) == 0) {
                size_t got = 0;
                uint8_t* out = dst + (size_t)i * frame;
                while (got < frame) {
                    const ssize_t r = pread(fd, out + got, frame - got, (off_t)(off + (long long)got));
                    if (r <= 0) { err = std::string(paths[i]) + (r == 0 ? ": truncated file" : std::string(": ") + strerror(errno)); break; }
                    got += (size_t)r;
                }
            }
            if (fd >= 0) close(fd);
            if (!err.empty()) {
                std::lock_guard<std::mutex> lock(mu);
                if (!failed.exchange(1)) first_error = err;
                return;
            }
        }
    };
    const int nt = std::max(1, std::min(nthreads, n));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    if (failed.load()) { set_error("colvo_read_npy_u8_frames: %s", first_error.c_str()); return 1; }
    return 0;
}

