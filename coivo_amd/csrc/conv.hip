// conv.hip -- a1/a2 of SURVEY.md §8: the 3x3 convolution blocks of DepthNet / PoseNet as
// patch-staged implicit GEMMs on the gfx950 matrix cores.
//
// Concept: /root/reference/README.md:5,7 (DCDP depth + pose estimation networks).  Results are
// specified by oracle/colvo_spec.py (F.conv2d(k=3, pad=1, stride 1|2) + bias + ReLU, nearest 2x
// up-sampling and channel concat in front of decoder convs).
//
// Layout: feature maps NHWC, channels a multiple of 8; T = float (exact-f32 MFMA 16x16x4, parity mode)
// or bf16 (MFMA 16x16x32, throughput mode), fp32 accumulation in both.
//
// k_conv3x3 (forward AND input-gradient):  D[n][pixel] = sum_{tap,c} W[n][tap][c] * A[pixel@tap][c]
//   - one 256-thread workgroup = a tile of <=128 output pixels (toh x tow region of one image) x BN
//     output channels; wave w owns pixels 32w..32w+31 (2 MFMA fragments) x all BN channels.
//   - K loop over channel chunks of CK = NG*16 bytes: the input PATCH of the tile (tile + halo, CK
//     channels) is staged in LDS ONCE and re-read by all 9 taps (9x fewer global reads than a per-tap
//     gather); the weight slab [BN][9][CK] is staged beside it.  Pixel pitch / weight-row pitch are
//     padded so the lane groups of a ds_read_b128 fall on distinct banks.
//   - the gather that fills the patch folds in: zero padding, nearest 2x up-sampling (decoder "up"
//     convs), channel concat of two sources (skip connections), and zero-insertion (the input gradient
//     of a stride-2 conv is a stride-1 conv over the zero-dilated output gradient, flipped weights).
//   - the MFMA operands are swapped (A = weights, B = pixels), so a lane's accumulator holds 4 consecutive
//     channels of one pixel: the epilogue (bias, ReLU, producer's ReLU mask, accumulate for skip fan-out,
//     convert) stores straight from the accumulators; only the 2x2 sum-pooling (input gradient of an
//     up-sampled source) goes through an fp32 tile in LDS.
//   - address set-up is kept to a few dozen instructions per thread (magic-number divisions, scalar strides,
//     out-of-range offsets instead of tests): before that it was most of a workgroup's VALU time.
//   - measured (tools/ablate_conv.sh, COLVO_TRACE): a workgroup's life is 0.5 us set-up + 1-1.8 us first stage
//     + 1.3-1.5 us per 32-channel chunk + 1 us epilogue; with 2-3 workgroups per CU the chunk is bound by LDS
//     bandwidth (a 32x32 per-wave tile reads 1 KB of LDS per MFMA).
//
// k_wgrad3x3:  dW[co][tap][c] += sum_pixels dY[pixel][co] * X[pixel@tap][c]   (K = pixels)
//   - workgroup = (co tile of 16*MT) x (channel chunk CK of one source) x (a range of pixel tiles);
//     the sums of the whole range stay in MFMA accumulators, then ONE fp32 atomic add per element.
//   - both operands have the reduction index (pixel) as the slow LDS dimension: bf16 fragments are
//     read with ds_read_b64_tr_b16 (hardware transpose), f32 fragments with plain ds_read_b32.
#define COLVO_ACC_CONSTRAINT "+v"     // built with -mllvm -amdgpu-mfma-vgpr-form (coivo_amd/build.py)
#include "conv_common.h"
#include "conv_stage.h"

namespace colvo {
namespace {

// Stage the input patch of one tile / one channel chunk: sP[pix][CK] (pitch PIXP bytes).
template <typename T, int NG>
__device__ __forceinline__ void stage_patch(const Gather& g, int s, int c0, int b, int iy0, int ix0, int PH, int PW,
                                            char* sP) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int PIXP = pitch_bytes(NG * 16);
    const int total = PH * PW * NG;
    const char* base = g.src[s];
    const int C = g.C[s], Hs = g.Hs[s], Ws = g.Ws[s], mode = g.mode[s];
    for (int i = threadIdx.x; i < total; i += NT) {
        const int pix = i / NG, cg = i - pix * NG;
        const int py = pix / PW, px = pix - py * PW;
        const int vy = iy0 + py, vx = ix0 + px;
        bool inb = (vy >= 0) && (vy < g.Hi) && (vx >= 0) && (vx < g.Wi);
        int sy = vy, sx = vx;
        if (mode != MODE_DIRECT) {
            if (mode == MODE_DILATE) inb = inb && !((vy | vx) & 1);
            sy = vy >> 1; sx = vx >> 1;
        }
        inb = inb && (sy < Hs) && (sx < Ws);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (inb) v = ld16(base + ((((size_t)b * Hs + sy) * Ws + sx) * C + c0 + cg * G) * ES);
        st16(sP + pix * PIXP + cg * 16, v);
    }
}

// Epilogue of one output tile from the fp32 staging tile in LDS: bias, ReLU, 2x2 sum-pool, producer's ReLU mask,
// accumulate, convert, 16-byte coalesced stores.
template <typename T, int BN, int NTH = 256>
__device__ __forceinline__ void conv_epilogue(const ConvK& a, const float* sOut, int b, int oy0, int ox0, int n0, int tid) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int OUTP = BN + 4;
    constexpr int GPR = BN / G;   // output granules per pixel row of the tile
    const int Hout = a.pool2 ? (a.Ho >> 1) : a.Ho, Wout = a.pool2 ? (a.Wo >> 1) : a.Wo;
    const int eh = a.pool2 ? (a.toh >> 1) : a.toh, ew = a.pool2 ? (a.tow >> 1) : a.tow;
    const int ey0 = a.pool2 ? (oy0 >> 1) : oy0, ex0 = a.pool2 ? (ox0 >> 1) : ox0;
    for (int i = tid; i < eh * ew * GPR; i += NTH) {
        const int q = i / GPR, gch = i - q * GPR;
        const int qy = q / ew, qx = q - qy * ew;
        const int gy = ey0 + qy, gx = ex0 + qx;
        const int n = n0 + gch * G;
        if (gy >= Hout || gx >= Wout || n >= a.N) continue;
        float v[G];
        if (a.pool2) {
            const float* r0 = sOut + ((2 * qy) * a.tow + 2 * qx) * OUTP + gch * G;
            const float* r1 = r0 + a.tow * OUTP;
#pragma unroll
            for (int k = 0; k < G; ++k) v[k] = (r0[k] + r0[OUTP + k]) + (r1[k] + r1[OUTP + k]);
        } else {
            const float* r0 = sOut + q * OUTP + gch * G;
#pragma unroll
            for (int k = 0; k < G; ++k) v[k] = r0[k];
        }
        if (a.bias) {
#pragma unroll
            for (int k = 0; k < G; ++k) v[k] += a.bias[n + k];
        }
        if (a.relu) {
#pragma unroll
            for (int k = 0; k < G; ++k) v[k] = fmaxf(v[k], 0.0f);
        }
        const size_t off = ((((size_t)b * Hout + gy) * Wout + gx) * a.N + n) * ES;
        if constexpr (ES == 4) {
            if (a.mask) {
                const u32x4 m = ld16(a.mask + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = (__uint_as_float(m[k]) > 0.0f) ? v[k] : 0.0f;
            }
            if (a.accumulate) {
                const u32x4 o = ld16(a.out + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += __uint_as_float(o[k]);
            }
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __float_as_uint(v[k]);
            st16(a.out + off, o);
        } else {
            if (a.mask) {
                const u32x4 m = ld16(a.mask + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // bf16 > 0  <=>  sign clear and magnitude non-zero
                    const uint32_t lo = m[k] & 0xFFFFu, hi = m[k] >> 16;
                    if (!(lo != 0 && lo < 0x8000u)) v[2 * k] = 0.0f;
                    if (!(hi != 0 && hi < 0x8000u)) v[2 * k + 1] = 0.0f;
                }
            }
            if (a.accumulate) {
                const u32x4 o = ld16(a.out + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[2 * k] += bf2f((uint16_t)(o[k] & 0xFFFFu));
                    v[2 * k + 1] += bf2f((uint16_t)(o[k] >> 16));
                }
            }
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = pack2bf(v[2 * k], v[2 * k + 1]);
            st16(a.out + off, o);
        }
    }
}

// --------------------------------------------------------------------------------------------- //
// forward / input-gradient kernel                                                                //
// --------------------------------------------------------------------------------------------- //
// The MFMAs are issued with the operands SWAPPED (A = weight rows, B = pixels): lane (l15, kg) of accumulator
// [mf][nf] then holds 4 CONSECUTIVE output channels n0 + nf*16 + kg*4 + {0..3} of ONE pixel (wave*32 + mf*16 + l15),
// i.e. 8 / 16 contiguous bytes of the NHWC output -- the epilogue stores straight from the accumulators (bias, ReLU,
// mask, accumulate per lane; buffer stores drop out-of-range lanes), with no LDS transposition and no index division.
// Only pool2 (input gradient of an up-sampled source) still goes through the fp32 tile in LDS.
// Direct epilogue in three steps so that a caller can issue the mask / accumulate loads early (persistent kernel:
// before the tile's MFMAs): (1) byte offsets of the lane's two pixels, (2) loads, (3) arithmetic + stores.
// rout / rmask: descriptors whose base + soff (wave-uniform) is pixel (0, 0) of the output image of this tile.
template <typename T, int NF>
struct Epi {
    typedef typename EV<T>::type V;
    static constexpr int ES = TT<T>::ES;
    int off[2][NF];
    V pm[2][NF], pa[2][NF];

    __device__ __forceinline__ void offsets(const ConvK& a, int oy0, int ox0, int n0, int wave, int l15, int kg) {
        const int npix = a.toh * a.tow;
        int obase[2];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const int p = wave * 32 + mf * 16 + l15;
            const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
            const int gy = oy0 + oy, gx = ox0 + ox;
            const bool ok = (p < npix) && (gy < a.Ho) && (gx < a.Wo);
            obase[mf] = ok ? (gy * a.Wo + gx) * a.N * ES : OOB_OFF;
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int n = n0 + nf * 16 + kg * 4;
            const int noff = (n < a.N) ? n * ES : OOB_OFF;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) off[mf][nf] = obase[mf] + noff;     // out of range -> loads 0 / store dropped
        }
    }
    // k_dgrad_s2: the lane's two pixels are positions (gy, gx) of the OUTPUT-gradient tile; parity class (py, px) of the input
    // gradient lives at (2 gy + py, 2 gx + px) of an image twice that size
    __device__ __forceinline__ void offsets_s2(const ConvK& a, int oy0, int ox0, int n0, int wave, int l15, int kg, int py, int px) {
        const int npix = a.toh * a.tow;
        int obase[2];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const int p = wave * 32 + mf * 16 + l15;
            const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
            const int gy = oy0 + oy, gx = ox0 + ox;
            const bool ok = (p < npix) && (gy < a.Ho) && (gx < a.Wo);
            obase[mf] = ok ? ((2 * gy + py) * (2 * a.Wo) + 2 * gx + px) * a.N * ES : OOB_OFF;
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int n = n0 + nf * 16 + kg * 4;
            const int noff = (n < a.N) ? n * ES : OOB_OFF;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) off[mf][nf] = obase[mf] + noff;
        }
    }
    __device__ __forceinline__ void prefetch(const ConvK& a, __amdgpu_buffer_rsrc_t rout, __amdgpu_buffer_rsrc_t rmask, int soff) {
        if (a.mask) {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) pm[mf][nf] = epi_load<T>(rmask, off[mf][nf], soff);
        }
        if (a.accumulate) {
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) pa[mf][nf] = epi_load<T>(rout, off[mf][nf], soff);
        }
    }
    __device__ __forceinline__ void finish(const ConvK& a, const f32x4 (&acc)[2][NF], const u32x4 (&biasv)[NF],
                                           __amdgpu_buffer_rsrc_t rout, int soff) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const u32x4 bs = biasv[nf];                   // zeros without a bias / beyond N
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[mf][nf][r] + __uint_as_float(bs[r]);
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                }
                if constexpr (ES == 4) {
                    if (a.mask) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (__uint_as_float(pm[mf][nf][r]) > 0.0f) ? v[r] : 0.0f;
                    }
                    if (a.accumulate) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] += __uint_as_float(pa[mf][nf][r]);
                    }
                    u32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = __float_as_uint(v[r]);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rout, off[mf][nf], soff, 0);
                } else {
                    if (a.mask) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            // bf16 > 0  <=>  sign clear and magnitude non-zero
                            const uint32_t lo = pm[mf][nf][k] & 0xFFFFu, hi = pm[mf][nf][k] >> 16;
                            if (!(lo != 0 && lo < 0x8000u)) v[2 * k] = 0.0f;
                            if (!(hi != 0 && hi < 0x8000u)) v[2 * k + 1] = 0.0f;
                        }
                    }
                    if (a.accumulate) {
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            v[2 * k] += bf2f((uint16_t)(pa[mf][nf][k] & 0xFFFFu));
                            v[2 * k + 1] += bf2f((uint16_t)(pa[mf][nf][k] >> 16));
                        }
                    }
                    u32x2 o;
#pragma unroll
                    for (int k = 0; k < 2; ++k) o[k] = pack2bf(v[2 * k], v[2 * k + 1]);
                    __builtin_amdgcn_raw_buffer_store_b64(o, rout, off[mf][nf], soff, 0);
                }
            }
        }
    }
};

// TAIL: the patch may exceed PPF*256 granules (stride-2 tiles); compiled out otherwise -- its (never executed) loads made
// hipcc wait for ALL outstanding loads, the just-issued prefetch included, in front of the MFMA phase.
// NCH > 0: the chunk count is a compile-time constant and the K loop is fully unrolled -- in straight-line code hipcc counts
// the outstanding loads exactly, so the register ring really keeps DEPTH chunks in flight (with a loop it drains to
// vmcnt(0) at every other store phase).
// NTH: threads per workgroup.  256 = 4 waves = 128 output pixels; 512 = 8 waves = 256 output pixels x BN channels (the
// "wide" form: per MFMA it stages half the bytes of the 128 x 32 tile -- the mid layers are bound by the bytes a CU can
// keep in flight, profiles/r2_conv_pmc.json -- and reads 0.75 instead of 1 KB of LDS).
template <typename T, int BN, int NG, int DEPTH, bool TAIL, int NCH = 0, int NTH = 256>
__global__ __launch_bounds__(NTH, NTH == 512 ? 4 : ((BN <= 32 && DEPTH == 1 && !TAIL) ? 4 : 2)) void k_conv3x3(const ConvK a) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int CK = NG * G;
    constexpr int NGR = 9 * NG;                    // real granules per weight row and chunk
    constexpr int STEPS = (NGR + 3) / 4;           // MFMA k-groups (4 granules each) per chunk
    constexpr int WROW = wrow_bytes(STEPS * 4);     // weight-row pitch in LDS (bytes)
    // patch-pixel pitch in LDS (bytes).  TAIL instantiations are the stride-2 layers (a stride-1 patch of <= 128 outputs never
    // exceeds 3 x 256 granules): consecutive fragment rows are TWO patch pixels apart there, conv_common.h pitch_bytes_s2
    constexpr int PIXP = TAIL ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16);
    constexpr int NF = BN / 16;
    constexpr int OUTP = BN + 4;                   // pool2 epilogue row pitch (floats)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + BN * WROW;

    TRACE(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    // 1-D grid, XCD-contiguous: logical id = (image, pixel tile, output-channel tile), channel tile fastest
    const int lid = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, a.xcd));
    const int tlin = lid / a.ntn;
    const int n0 = (lid - tlin * a.ntn) * BN;
    const int tpi = a.tiles_x * a.tiles_y;
    const int b = tlin / tpi, trem = tlin - b * tpi;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int oy0 = ty * a.toh, ox0 = tx * a.tow;
    const int S = a.g.stride;
    const int PH = (a.toh - 1) * S + 3, PW = (a.tow - 1) * S + 3;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int npix = a.toh * a.tow;

    // ---- software-pipelined K loop over channel chunks ----
    // Staging goes global -> registers -> LDS.  The registers of chunk k+1 are loaded (all loads of a thread
    // issued back to back) BEFORE the MFMAs of chunk k and written to LDS after them, so global latency hides
    // under the matrix work instead of being paid once per 16-byte granule.
    constexpr int WTOT = BN * NGR;                         // real weight granules per chunk
    constexpr int WIT = (WTOT + NTH - 1) / NTH;
    // patch granules per thread that are prefetched: 3 cover every stride-1 patch (<= 10 x 18 pixels x 4 granules); the
    // stride-2 instantiations take 9 (17 x 33 x 4 = 2244 granules) so that nothing is left to the synchronous tail loop
    constexpr int PPF = TAIL ? 9 : 3;
    const int ptotal = PH * PW * NG;
    const int nch0 = a.g.C[0] / CK, nch = NCH ? NCH : nch0 + a.g.C[1] / CK;
    constexpr int KUNROLL = NCH > 0 ? NCH / DEPTH : 1;     // full unroll of the K loop when the chunk count is a template constant
    u32x4 wv[DEPTH][WIT], pv[DEPTH][PPF];                  // DEPTH chunks in flight (register ring)

    // All address arithmetic is chunk-invariant except for the channel offset, so it is done ONCE per thread, and
    // cheaply -- the set-up used to be most of a workgroup's VALU time:
    //  * weights: granule i = it*256 + tid of the slab [BN][9][CK] sits at a.w + woff0 + it * (256/NG taps), i.e. ONE
    //    per-thread offset plus a scalar stride (256 granules = 256/NG whole taps); rows beyond N fall outside the
    //    descriptor and read as zero without a test.
    //  * patch: magic-number division by the patch width, branch-free mode handling (shift / parity mask).
    const int tapB = a.Ctot * ES;                          // bytes from tap to tap inside a weight row
    int woff0, woffL, wlds[WIT];
    {
        const int n = tid / NGR, gi = tid - n * NGR;
        const int tap = gi / NG, cg = gi - tap * NG;
        woff0 = ((n0 + n) * 9 + tap) * tapB + cg * 16;
        woffL = ((WIT - 1) * NTH + tid < WTOT) ? woff0 : OOB_OFF;
    }
    // two named descriptors / offset sets (arrays of descriptors end up in scratch and turn every load into a
    // waterfall loop)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * 9 * a.Ctot * ES, 0x00020000);
    auto make_img = [&](int sidx) {
        const int Hs = a.g.Hs[sidx], Ws = a.g.Ws[sidx], Cs = a.g.C[sidx];
        const char* base = a.g.src[sidx] + (size_t)b * Hs * Ws * Cs * ES;
        return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, Cs > 0 ? Hs * Ws * Cs * ES : 0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rimg0 = make_img(0), rimg1 = make_img(1);
    auto patch_off = [&](int sidx, int i) -> int {
        const int Ws = a.g.Ws[sidx], Cs = a.g.C[sidx], mode = a.g.mode[sidx];
        const int sh = (mode != MODE_DIRECT) ? 1 : 0;       // up-sampled / zero-dilated sources are stored at half size
        const int par = (mode == MODE_DILATE) ? 1 : 0;      // zero-dilated: only even virtual positions hold data
        const int pix = i / NG, cg = i - pix * NG;
        const int py = mdiv(pix, a.m_pw), px = pix - py * PW;
        const int vy = iy0 + py, vx = ix0 + px;
        // Hi = Hs << sh, Wi = Ws << sh, so the virtual-extent test covers the stored extent too
        const bool inb = (i < ptotal) && ((unsigned)vy < (unsigned)a.g.Hi) && ((unsigned)vx < (unsigned)a.g.Wi) &&
                         (((vy | vx) & par) == 0);
        return inb ? (((vy >> sh) * Ws + (vx >> sh)) * Cs + cg * G) * ES : OOB_OFF;
    };
    int poff0[PPF], poff1[PPF], plds[PPF];          // global offsets per source, LDS offset (rows at the padded pitch a.pwp)
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        poff0[it] = patch_off(0, it * NTH + tid);
        poff1[it] = OOB_OFF;
        const int i = it * NTH + tid, pix = i / NG, cg = i - pix * NG;
        const int py = mdiv(pix, a.m_pw), px = pix - py * PW;
        plds[it] = (py * a.pwp + px) * PIXP + cg * 16;
    }
    if (a.g.C[1] > 0) {
#pragma unroll
        for (int it = 0; it < PPF; ++it) poff1[it] = patch_off(1, it * NTH + tid);
    }

    auto chunk_src = [&](int k, int& sidx, int& c0) { sidx = (k < nch0) ? 0 : 1; c0 = (k - (sidx ? nch0 : 0)) * CK; };
    // `dead` (wave-uniform, 0 or OOB_OFF) is OR-ed into every offset: the prefetch past the last chunk is issued all the
    // same and reads zeros, so the K loop has no branch around its loads (a join there costs a vmcnt(0))
    auto load_w = [&](int k, int dead, u32x4 (&w)[WIT]) {
        const int so = dead ? 0 : k * CK * ES;
#pragma unroll
        for (int it = 0; it < WIT; ++it)                  // branch-free, zero-filled
            w[it] = bld16(rw, (((it == WIT - 1) ? woffL : woff0) + it * (NTH / NG) * tapB) | dead, so);
    };
#ifdef COLVO_ABLATE
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
        for (int it = 0; it < WIT; ++it) wv[d][it] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int it = 0; it < PPF; ++it) pv[d][it] = u32x4{0u, 0u, 0u, 0u};
    }
#endif
    auto store_w = [&](const u32x4 (&w)[WIT]) {
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            if (WTOT % NTH == 0 || it < WIT - 1 || it * NTH + tid < WTOT) st16(sW + wlds[it], w[it]);
    };
    auto patch_granule = [&](int sidx, int c0, int i) -> u32x4 {      // only for the tail of large (stride-2) patches
        return sidx == 0 ? bld16(rimg0, patch_off(0, i), c0 * ES) : bld16(rimg1, patch_off(1, i), c0 * ES);
    };
    auto load_p = [&](int k, int dead, u32x4 (&pvv)[PPF]) {
        int sidx, c0;
        chunk_src(dead ? 0 : k, sidx, c0);
        const int so = c0 * ES;
        if (sidx == 0) {
#pragma unroll
            for (int it = 0; it < PPF; ++it) pvv[it] = bld16(rimg0, poff0[it] | dead, so);
        } else {
#pragma unroll
            for (int it = 0; it < PPF; ++it) pvv[it] = bld16(rimg1, poff1[it] | dead, so);
        }
    };
    auto store_p = [&](int k, const u32x4 (&pvv)[PPF]) {
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int i = it * NTH + tid;
            if (i < ptotal) st16(sP + plds[it], pvv[it]);
        }
        // patches larger than PPF*256 granules (stride-2 tiles): the rest is staged in place, 3 loads in flight
        // (the host keeps a.pwp == PW for these: linear pixel index = LDS pixel index)
        if constexpr (TAIL) {
            int sidx, c0;
            chunk_src(k, sidx, c0);
            for (int base = PPF * NTH; base < ptotal; base += 3 * NTH) {
                u32x4 t[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) t[u] = patch_granule(sidx, c0, base + u * NTH + tid);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int i = base + u * NTH + tid;
                    if (i < ptotal) { const int pix = i / NG, cg = i - pix * NG; st16(sP + pix * PIXP + cg * 16, t[u]); }
                }
            }
        }
    };
    // the first DEPTH chunks are requested as soon as their offsets exist; everything the loads do not need (fragment
    // columns, LDS offsets, the padding granules, the bias) is computed while they are in flight
    if (!ABL(4)) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int dead = (d < nch) ? 0 : OOB_OFF;
            load_w(d, dead, wv[d]);
            load_p(d, dead, pv[d]);
        }
    }
    // the two fragment columns (pixels) of this lane
    int pbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        int p = wave * 32 + mf * 16 + l15;
        if (p >= npix) p = 0;
        const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
        pbase[mf] = ((oy * S) * a.pwp + ox * S) * PIXP;
    }

    f32x4 acc[2][NF];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int it = 0; it < WIT; ++it) {
        const int i = it * NTH + tid;
        const int n = i / NGR;
        wlds[it] = i * 16 + n * (WROW - NGR * 16);
    }
    if constexpr (STEPS * 4 != NGR) {                      // zero the padding granules of every weight row once
        for (int i = tid; i < BN * (STEPS * 4 - NGR); i += NTH) {
            const int n = i / (STEPS * 4 - NGR), q = i - n * (STEPS * 4 - NGR);
            st16(sW + n * WROW + (NGR + q) * 16, u32x4{0u, 0u, 0u, 0u});
        }
    }
    // bias of this lane's output channels, fetched now so that its latency is not paid in the epilogue
    u32x4 biasv[NF];
    {
        const __amdgpu_buffer_rsrc_t rbias =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) biasv[nf] = bld16(rbias, (n0 + nf * 16 + kg * 4) * 4, 0);   // zeros without a bias
    }

    TRACE(1);
#pragma unroll KUNROLL
    for (int k0 = 0; k0 < nch; k0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int k = k0 + d;
        if (DEPTH > 1 && k >= nch) break;
        if (!ABL(16)) __syncthreads();    // the MFMAs of chunk k-1 have finished reading LDS
        if (!ABL(8)) { store_w(wv[d]); store_p(k, pv[d]); }
        if (!ABL(16)) __syncthreads();
        if (k == 0) TRACE(2);
        if (k == 1) TRACE(3);
        if (!ABL(4)) {                    // in flight during the next DEPTH MFMA phases
            const int dead = (k + DEPTH < nch) ? 0 : OOB_OFF;
            load_w(k + DEPTH, dead, wv[d]);
            load_p(k + DEPTH, dead, pv[d]);
        }
        // MFMA phase: the fragments of k-group m+FD are read from LDS BEFORE the MFMAs of k-group m are issued (FD+1
        // register sets).  FD = 1 hides the LDS latency when other waves share the SIMD; the lone-workgroup variant
        // (DEPTH 2: at most ~1 workgroup per CU, one wave per SIMD) needs FD = 2 -- a 32-wide k-group is only 64
        // MFMA cycles, less than the LDS round trip.
        {
            constexpr int FD = (DEPTH > 1) ? 2 : 1, NS = FD + 1;   // (measured: FD 2 alone changes nothing; kept with the ring)
            u32x4 av[NS][2], bv[NS][NF];
            auto read_frags = [&](int m, u32x4 (&ar)[2], u32x4 (&br)[NF]) {
                const int gi = 4 * m + kg;
                int tap = gi / NG;
                const int cg = gi - tap * NG;
                tap = min(tap, 8);                      // padded k-groups multiply real pixels by zero weights
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int aoff = (ky * a.pwp + kx) * PIXP + cg * 16;
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) ar[mf] = ld16(sP + pbase[mf] + aoff);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) br[nf] = ld16(sW + (nf * 16 + l15) * WROW + gi * 16);
            };
#ifdef COLVO_ABLATE
            for (int q = 0; q < NS; ++q) {
                for (int mf = 0; mf < 2; ++mf) av[q][mf] = u32x4{0u, 0u, 0u, 0u};
                for (int nf = 0; nf < NF; ++nf) bv[q][nf] = u32x4{0u, 0u, 0u, 0u};
            }
#endif
            if (!ABL(2)) {
#pragma unroll
                for (int q = 0; q < FD; ++q)
                    if (q < STEPS) read_frags(q, av[q], bv[q]);
            }
#pragma unroll
            for (int m = 0; m < STEPS; ++m) {
                const int cur = m % NS;
                if (m + FD < STEPS && !ABL(2)) read_frags(m + FD, av[(m + FD) % NS], bv[(m + FD) % NS]);
#ifdef COLVO_ABLATE
                if (ABL(1)) {
                    for (int mf = 0; mf < 2; ++mf) asm volatile("" ::"v"(av[cur][mf]));
                    for (int nf = 0; nf < NF; ++nf) asm volatile("" ::"v"(bv[cur][nf]));
                    continue;
                }
#endif
                if constexpr (ES == 2) {
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, bv[cur][nf]), __builtin_bit_cast(bf16x8, av[cur][mf]),
                                acc[mf][nf], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NF; ++nf)
                                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    __uint_as_float(bv[cur][nf][j]), __uint_as_float(av[cur][mf][j]), acc[mf][nf], 0, 0, 0);
                }
            }
        }
      }
    }

    mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[2 * NF]>(acc));
#ifdef COLVO_ABLATE
    if (ABL(32)) {
        for (int mf = 0; mf < 2; ++mf)
            for (int nf = 0; nf < NF; ++nf) asm volatile("" ::"v"(acc[mf][nf]));
        return;
    }
#endif
    TRACE(4);
    if (!a.pool2) {
        // two-output form (input gradient of a concat layer): this workgroup's channel tile lies in exactly one of the sources
        ConvK e = a;
        int n0e = n0;
        if (a.nsplit > 0) {
            const bool second = n0 >= a.nsplit;           // wave-uniform
            e.out = second ? a.out2 : a.out;
            e.mask = second ? a.mask2 : a.mask;
            e.N = second ? a.N - a.nsplit : a.nsplit;
            n0e = second ? n0 - a.nsplit : n0;
        }
        const int img_bytes = e.Ho * e.Wo * e.N * ES;
        const __amdgpu_buffer_rsrc_t rout =
            __builtin_amdgcn_make_buffer_rsrc((void*)(e.out + (size_t)b * img_bytes), 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(e.mask ? e.mask + (size_t)b * img_bytes : e.out), 0, e.mask ? img_bytes : 0, 0x00020000);
        Epi<T, NF> ep;
        ep.offsets(e, oy0, ox0, n0e, wave, l15, kg);
        ep.prefetch(e, rout, rmask, 0);
        ep.finish(e, acc, biasv, rout, 0);
        TRACE(5);
        return;
    }
    // ---- pool2 epilogue through LDS ----
    __syncthreads();
    float* sOut = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            *reinterpret_cast<f32x4*>(&sOut[(wave * 32 + mf * 16 + l15) * OUTP + nf * 16 + 4 * kg]) = acc[mf][nf];
    __syncthreads();
    conv_epilogue<T, BN, NTH>(a, sOut, b, oy0, ox0, n0, tid);
}

// --------------------------------------------------------------------------------------------- //
// input gradient of a STRIDE-2 conv, parity-decomposed                                           //
// --------------------------------------------------------------------------------------------- //
// dx[iy][ix][ci] = sum_{ky,kx,co} w[co][ky][kx][ci] * dy[oy][ox][co] with 2 oy + ky - 1 = iy: an even input row sees only
// ky = 1 (oy = iy/2), an odd one ky = 0 (oy = (iy+1)/2) and ky = 2 (oy = (iy-1)/2); likewise the columns.  So the four
// parity classes (iy & 1, ix & 1) of dx are 1-, 2-, 2- and 4-tap convolutions over dy.  As a stride-1 conv over the
// zero-dilated dy (the general kernel's MODE_DILATE) three quarters of the MFMAs multiply inserted zeros and the patch of a
// 128-pixel tile is 10 x 18 dilated pixels of which 45 are real.  Here a workgroup takes a tile of 128 dy POSITIONS, stages
// their (toh+1) x (tow+1) patch and the usual [BN][9][CK] weight slab per chunk, and runs the nine taps ONCE: with 32-channel
// chunks a k-group IS a tap, so each k-group's MFMAs go to the accumulator set of the tap's parity class (4 sets), reading
// the patch at the tap's (+1 row if ky == 0, +1 column if kx == 0) offset.  Same MFMA count as one ordinary 3x3 tile, but
// 512 input-gradient pixels instead of 128: a quarter of the MFMAs, a quarter of the workgroups, no dead patch bytes.
// The weight operand is the flipped layout w_bwd [Cin][8 - tap][Cout] the dilated form uses (slot m holds tap 8 - m).
template <typename T, int BN, int DEPTH, int NCH = 0>
__global__ __launch_bounds__(NT, (DEPTH == 1 && TT<T>::ES == 2) ? 3 : 2) void k_dgrad_s2(const ConvK a) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int NG = 4, CK = NG * G, STEPS = 9;
    constexpr int WROW = wrow_bytes(STEPS * 4);     // rows of 36 granules: nothing to zero-pad
    constexpr int PIXP = pitch_bytes(NG * 16);
    constexpr int NF = BN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + BN * WROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const TileCoord tc = tile_coord<BN>(a);
    const int b = tc.b, ty = tc.ty, tx = tc.tx, n0 = tc.n0;
    const int oy0 = ty * a.toh, ox0 = tx * a.tow;       // tile origin in dy
    const int PH = a.toh + 1, PW = a.tow + 1;
    const int npix = a.toh * a.tow;

    typedef SlabStage<T, BN, NG> Slab;
    typedef PatchStage<T, NG, 3> Patch;
    constexpr int WIT = Slab::WIT, PPF = 3;
    const int nch = NCH ? NCH : a.g.C[0] / CK;
    constexpr int KUNROLL = NCH > 0 ? NCH / DEPTH : 1;     // full unroll of the K loop when the chunk count is a template constant
    u32x4 wv[DEPTH][WIT], pv[DEPTH][PPF];
    Slab slab;
    Patch patch;
    slab.init(a, n0, tid);
    patch.init(a, b, tid, PH, PW, oy0, ox0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {                      // the first chunks are requested before anything the LDS side needs
        const int dead = (d < nch) ? 0 : OOB_OFF;
        slab.load(d, dead, wv[d]);
        patch.load(d, dead, pv[d]);
    }
    slab.lds_offsets(tid);
    slab.zero_padding(sW, tid);
    int pbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        int p = wave * 32 + mf * 16 + l15;
        if (p >= npix) p = 0;
        const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
        pbase[mf] = (oy * a.pwp + ox) * PIXP;
    }
    f32x4 acc[4][2][NF];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[c][mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll KUNROLL
    for (int k0 = 0; k0 < nch; k0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int k = k0 + d;
        if (DEPTH > 1 && k >= nch) break;
        __syncthreads();
        slab.store(sW, tid, wv[d]);
        patch.store(sP, tid, pv[d]);
        __syncthreads();
        {
            const int dead = (k + DEPTH < nch) ? 0 : OOB_OFF;
            slab.load(k + DEPTH, dead, wv[d]);
            patch.load(k + DEPTH, dead, pv[d]);
        }
        u32x4 av[2][2], bv[2][NF];
        auto read_frags = [&](int m, u32x4 (&ar)[2], u32x4 (&br)[NF]) {
            const int t = 8 - m, ky = t / 3, kx = t - 3 * ky;       // slot m of the flipped slab holds tap 8 - m
            const int aoff = ((ky == 0 ? a.pwp : 0) + (kx == 0 ? 1 : 0)) * PIXP + kg * 16;
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) ar[mf] = ld16(sP + pbase[mf] + aoff);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) br[nf] = ld16(sW + (nf * 16 + l15) * WROW + (4 * m + kg) * 16);
        };
        read_frags(0, av[0], bv[0]);
#pragma unroll
        for (int m = 0; m < STEPS; ++m) {
            const int cur = m & 1;
            if (m + 1 < STEPS) read_frags(m + 1, av[cur ^ 1], bv[cur ^ 1]);
            const int t = 8 - m, ky = t / 3, kx = t - 3 * ky;
            const int cls = (ky != 1 ? 2 : 0) | (kx != 1 ? 1 : 0);   // parity class (iy & 1, ix & 1) this tap feeds
            if constexpr (ES == 2) {
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        acc[cls][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, bv[cur][nf]), __builtin_bit_cast(bf16x8, av[cur][mf]),
                            acc[cls][mf][nf], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[cls][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(bv[cur][nf][j]), __uint_as_float(av[cur][mf][j]), acc[cls][mf][nf], 0, 0, 0);
            }
        }
      }
    }
    mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[8 * NF]>(acc));

    const int img_bytes = 4 * a.Ho * a.Wo * a.N * ES;          // the input gradient: (2 Ho) x (2 Wo) x N
    const __amdgpu_buffer_rsrc_t rout =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img_bytes), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.mask ? a.mask + (size_t)b * img_bytes : a.out), 0, a.mask ? img_bytes : 0, 0x00020000);
    u32x4 nobias[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) nobias[nf] = u32x4{0u, 0u, 0u, 0u};
    Epi<T, NF> ep[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {                                // all four classes' mask / accumulate loads first
        ep[c].offsets_s2(a, oy0, ox0, n0, wave, l15, kg, c >> 1, c & 1);
        ep[c].prefetch(a, rout, rmask, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) ep[c].finish(a, acc[c], nobias, rout, 0);
}

// --------------------------------------------------------------------------------------------- //
// forward conv over a nearest-2x UP-SAMPLED source, four output pixels per source position       //
// --------------------------------------------------------------------------------------------- //
// y[2a+py][2b+px] = sum_taps w[ky][kx] * x[a + r(py,ky)][b + r(px,kx)],  r(0,.) = (-1, 0, 0),  r(1,.) = (0, 0, +1): the 2x2
// outputs above one source position read the same 3x3 source neighbourhood.  The general kernel treats them as four
// unrelated pixels of the up-sampled image: a 128-pixel tile stages a 10x18-pixel patch and the [BN][9][CK] weight slab for
// 144 MFMAs.  Here a workgroup takes a tile of 128 SOURCE positions, stages their (toh+2) x (tow+2) patch and the slab once
// per chunk and runs 4 x 144 MFMAs on them (four accumulator sets, one per output parity class): per MFMA a quarter of the
// staging, and 0.47 instead of 1.0 LDS fragment reads (the weight fragments of a tap serve all four classes, and a centre
// row / column tap reads ONE patch fragment for both parities).  The chunk loop is then bound by the MFMA pipe, not by
// its staging latency chain.
template <typename T, int BN, int DEPTH, int NCH = 0>
__global__ __launch_bounds__(NT, 2) void k_conv_up2(const ConvK a) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int NG = 4, CK = NG * G, STEPS = 9;
    constexpr int WROW = wrow_bytes(STEPS * 4);
    constexpr int PIXP = pitch_bytes(NG * 16);
    constexpr int NF = BN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + BN * WROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const TileCoord tc = tile_coord<BN>(a);
    const int b = tc.b, ty = tc.ty, tx = tc.tx, n0 = tc.n0;
    const int oy0 = ty * a.toh, ox0 = tx * a.tow;       // tile origin in the stored (half-size) source
    const int PH = a.toh + 2, PW = a.tow + 2;
    const int npix = a.toh * a.tow;

    typedef SlabStage<T, BN, NG> Slab;
    typedef PatchStage<T, NG, 3> Patch;
    constexpr int WIT = Slab::WIT, PPF = 3;
    const int nch = NCH ? NCH : a.g.C[0] / CK;
    constexpr int KUNROLL = NCH > 0 ? NCH / DEPTH : 1;     // full unroll of the K loop when the chunk count is a template constant
    u32x4 wv[DEPTH][WIT], pv[DEPTH][PPF];
    Slab slab;
    Patch patch;
    slab.init(a, n0, tid);
    patch.init(a, b, tid, PH, PW, oy0 - 1, ox0 - 1);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {                      // the first chunks are requested before anything the LDS side needs
        const int dead = (d < nch) ? 0 : OOB_OFF;
        slab.load(d, dead, wv[d]);
        patch.load(d, dead, pv[d]);
    }
    slab.lds_offsets(tid);
    slab.zero_padding(sW, tid);
    int pbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        int p = wave * 32 + mf * 16 + l15;
        if (p >= npix) p = 0;
        const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
        pbase[mf] = (oy * a.pwp + ox) * PIXP;                  // patch pixel (oy, ox) = source position (a - 1, b - 1)
    }
    u32x4 biasv[NF];
    {
        const __amdgpu_buffer_rsrc_t rbias =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) biasv[nf] = bld16(rbias, (n0 + nf * 16 + kg * 4) * 4, 0);
    }
    f32x4 acc[4][2][NF];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[c][mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll KUNROLL
    for (int k0 = 0; k0 < nch; k0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int k = k0 + d;
        if (DEPTH > 1 && k >= nch) break;
        __syncthreads();
        slab.store(sW, tid, wv[d]);
        patch.store(sP, tid, pv[d]);
        __syncthreads();
        {
            const int dead = (k + DEPTH < nch) ? 0 : OOB_OFF;
            slab.load(k + DEPTH, dead, wv[d]);
            patch.load(k + DEPTH, dead, pv[d]);
        }
#pragma unroll
        for (int m = 0; m < STEPS; ++m) {
            const int ky = m / 3, kx = m - 3 * ky;
            u32x4 bv[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bv[nf] = ld16(sW + (nf * 16 + l15) * WROW + (4 * m + kg) * 16);
            // patch row of parity py for tap row ky: (0,1,1) for py = 0, (1,1,2) for py = 1 -- the centre tap serves both
            const int nrv = (ky == 1) ? 1 : 2, ncv = (kx == 1) ? 1 : 2;
            u32x4 av[2][2][2];
#pragma unroll
            for (int rv = 0; rv < 2; ++rv)
#pragma unroll
                for (int cv = 0; cv < 2; ++cv) {
                    if (rv >= nrv || cv >= ncv) continue;
                    const int ro = (ky == 0) ? rv : (ky == 1) ? 1 : 1 + rv;
                    const int co = (kx == 0) ? cv : (kx == 1) ? 1 : 1 + cv;
                    const int aoff = (ro * a.pwp + co) * PIXP + kg * 16;
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) av[rv][cv][mf] = ld16(sP + pbase[mf] + aoff);
                }
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int rv = (ky == 1) ? 0 : py, cv = (kx == 1) ? 0 : px;
                    const int cls = 2 * py + px;
                    if constexpr (ES == 2) {
#pragma unroll
                        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NF; ++nf)
                                acc[cls][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, bv[nf]), __builtin_bit_cast(bf16x8, av[rv][cv][mf]),
                                    acc[cls][mf][nf], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                                for (int nf = 0; nf < NF; ++nf)
                                    acc[cls][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        __uint_as_float(bv[nf][j]), __uint_as_float(av[rv][cv][mf][j]), acc[cls][mf][nf], 0, 0, 0);
                    }
                }
        }
      }
    }
    mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[8 * NF]>(acc));

    const int img_bytes = 4 * a.Ho * a.Wo * a.N * ES;          // the output: (2 Ho) x (2 Wo) x N
    const __amdgpu_buffer_rsrc_t rout =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img_bytes), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        Epi<T, NF> ep;
        ep.offsets_s2(a, oy0, ox0, n0, wave, l15, kg, c >> 1, c & 1);
        ep.prefetch(a, rout, rmask, 0);                          // (forward: no mask, no accumulate -- nothing is loaded)
        ep.finish(a, acc[c], biasv, rout, 0);
    }
}

// --------------------------------------------------------------------------------------------- //
// input gradient w.r.t. a nearest-2x UP-SAMPLED source: the 2x2 sum-pool folded into the K loop  //
// --------------------------------------------------------------------------------------------- //
// dx[a][b] = sum_{q,p in {0,1}} dxup[2a+q][2b+p],  dxup = stride-1 conv of dy with the flipped slab.  The general kernel computes
// the four dxup pixels in four different lanes / workgroups and pools them through an fp32 tile in LDS.  Here a workgroup takes a
// tile of 128 SOURCE positions, stages the (2 toh + 2) x (2 tow + 2) patch of dy and the slab once per chunk, and runs the
// 4 x 9 (class, tap) k-groups into ONE accumulator set: the pool is the accumulation.  Per MFMA a quarter of the weight staging,
// 0.625 instead of 1.0 LDS fragment reads, no pooling epilogue.
// NG: 16-byte channel granules per chunk: 4 (a k-group of the MFMA phase is one tap) or 2 (dy with 16 bf16 channels -- the
// full-resolution layer -- : a k-group covers two taps, lanes kg 0,1 the first and kg 2,3 the second; 18 granules per weight
// row, padded to 20 with zeros).
template <typename T, int BN, int DEPTH, int NCH = 0, int NG = 4>
__global__ __launch_bounds__(NT, 2) void k_dgrad_up2(const ConvK a) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int CK = NG * G, NGR = 9 * NG, STEPS = (NGR + 3) / 4;
    constexpr int WROW = wrow_bytes(STEPS * 4);
    // (Round 4, measured and reverted: pitch_bytes_s2 here -- consecutive fragment rows are TWO dy pixels apart, 39-40 % of this
    // kernel's LDS cycles are bank conflicts at the 96-byte pitch -- with the row padding the conflict model then asks for: up3 17.7
    // -> 23.6 us at 16 frames, up4 / up3 52 -> 71 / 77 us at 64 frames.  The padded 80-byte patch crosses an LDS occupancy step at
    // 16 frames (84.7 KB: one workgroup per CU instead of two) and is slower at equal occupancy too; not understood, not kept.)
    constexpr int PIXP = pitch_bytes(NG * 16);
    constexpr int NF = BN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + BN * WROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const TileCoord tc = tile_coord<BN>(a);
    const int b = tc.b, ty = tc.ty, tx = tc.tx, n0 = tc.n0;
    const int oy0 = ty * a.toh, ox0 = tx * a.tow;       // tile origin in the half-size source = in dx
    const int PH = 2 * a.toh + 2, PW = 2 * a.tow + 2;   // dy patch
    const int npix = a.toh * a.tow;

    typedef SlabStage<T, BN, NG> Slab;
    typedef PatchStage<T, NG, (10 * NG + 3) / 4> Patch;
    constexpr int WIT = Slab::WIT, PPF = (10 * NG + 3) / 4;
    const int nch = NCH ? NCH : a.g.C[0] / CK;
    constexpr int KUNROLL = NCH > 0 ? NCH / DEPTH : 1;     // full unroll of the K loop when the chunk count is a template constant
    u32x4 wv[DEPTH][WIT], pv[DEPTH][PPF];
    Slab slab;
    Patch patch;
    slab.init(a, n0, tid);
    patch.init(a, b, tid, PH, PW, 2 * oy0 - 1, 2 * ox0 - 1);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {                      // the first chunks are requested before anything the LDS side needs
        const int dead = (d < nch) ? 0 : OOB_OFF;
        slab.load(d, dead, wv[d]);
        patch.load(d, dead, pv[d]);
    }
    slab.lds_offsets(tid);
    slab.zero_padding(sW, tid);
    int pbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        int p = wave * 32 + mf * 16 + l15;
        if (p >= npix) p = 0;
        const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
        pbase[mf] = ((2 * oy) * a.pwp + 2 * ox) * PIXP;     // patch pixel (0, 0) = dy position (2 oy0 - 1, 2 ox0 - 1)
    }
    f32x4 acc[2][NF];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll KUNROLL
    for (int k0 = 0; k0 < nch; k0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int k = k0 + d;
        if (DEPTH > 1 && k >= nch) break;
        __syncthreads();
        slab.store(sW, tid, wv[d]);
        patch.store(sP, tid, pv[d]);
        __syncthreads();
        {
            const int dead = (k + DEPTH < nch) ? 0 : OOB_OFF;
            slab.load(k + DEPTH, dead, wv[d]);
            patch.load(k + DEPTH, dead, pv[d]);
        }
#pragma unroll
        for (int m = 0; m < STEPS; ++m) {
            // the tap(s) of this k-group: NG 4 -> tap m for every lane; NG 2 -> tap 2m (kg 0,1) and 2m + 1 (kg 2,3; beyond
            // tap 8 the weights are the zero padding: any valid patch address will do)
            const int ta = (NG == 4) ? m : 2 * m, tb = (NG == 4) ? m : (2 * m + 1 > 8 ? 8 : 2 * m + 1);
            const int mya = ta / 3, mxa = ta - 3 * mya, myb = tb / 3, mxb = tb - 3 * myb;
            u32x4 bv[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bv[nf] = ld16(sW + (nf * 16 + l15) * WROW + (4 * m + kg) * 16);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    int aoff;
                    if constexpr (NG == 4) {
                        aoff = ((q + mya) * a.pwp + (pp + mxa)) * PIXP + kg * 16;
                    } else {
                        const int oa = ((q + mya) * a.pwp + (pp + mxa)) * PIXP, ob = ((q + myb) * a.pwp + (pp + mxb)) * PIXP;
                        aoff = ((kg & 2) ? ob : oa) + (kg & 1) * 16;
                    }
                    u32x4 av[2];
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf) av[mf] = ld16(sP + pbase[mf] + aoff);
                    if constexpr (ES == 2) {
#pragma unroll
                        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NF; ++nf)
                                acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8, bv[nf]), __builtin_bit_cast(bf16x8, av[mf]), acc[mf][nf], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                                for (int nf = 0; nf < NF; ++nf)
                                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        __uint_as_float(bv[nf][j]), __uint_as_float(av[mf][j]), acc[mf][nf], 0, 0, 0);
                    }
                }
        }
      }
    }
    mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[2 * NF]>(acc));

    const int img_bytes = a.Ho * a.Wo * a.N * ES;               // dx: the half-size source's extent
    const __amdgpu_buffer_rsrc_t rout =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img_bytes), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.mask ? a.mask + (size_t)b * img_bytes : a.out), 0, a.mask ? img_bytes : 0, 0x00020000);
    u32x4 nobias[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) nobias[nf] = u32x4{0u, 0u, 0u, 0u};
    Epi<T, NF> ep;
    ep.offsets(a, oy0, ox0, n0, wave, l15, kg);
    ep.prefetch(a, rout, rmask, 0);
    ep.finish(a, acc, nobias, rout, 0);
}

// --------------------------------------------------------------------------------------------- //
// stride-1 conv, QUAD tile: four 128-pixel sub-tiles (2 x 2) per workgroup share one weight slab //
// --------------------------------------------------------------------------------------------- //
// The one-tile kernel stages a [BN][9][CK] weight slab and a (toh+2) x (tow+2) patch for 144 MFMAs per chunk and is bound by
// that staging chain, not by the MFMA pipe; k_conv_up2, which runs 4 x 144 MFMAs on one staging, reaches
// 52-56 % of the MFMA peak on large grids.  This is the same form for ordinary stride-1 layers (forward of the enc*b / iconv*
// layers and their input gradients): a (2 toh) x (2 tow) output tile, ONE patch of (2 toh + 2) x (2 tow + 2) pixels (15 % fewer
// halo pixels than four separate ones), one slab, four accumulator sets.  One or two direct sources (skip concat), bias / ReLU /
// mask / accumulate epilogue as in the one-tile kernel.  ~210 VGPRs: one chunk of prefetch, two workgroups per CU.
template <typename T, int BN>
__global__ __launch_bounds__(NT, 2) void k_conv_q(const ConvK a) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int NG = 4, CK = NG * G, STEPS = 9;
    constexpr int WROW = wrow_bytes(STEPS * 4);
    constexpr int PIXP = pitch_bytes(NG * 16);
    constexpr int NF = BN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + BN * WROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const TileCoord tc = tile_coord<BN>(a);              // (tiles_x, tiles_y count quad tiles)
    const int b = tc.b, ty = tc.ty, tx = tc.tx, n0 = tc.n0;
    const int oy0 = ty * 2 * a.toh, ox0 = tx * 2 * a.tow;
    const int PH = 2 * a.toh + 2, PW = 2 * a.tow + 2;
    const int npix = a.toh * a.tow;

    typedef SlabStage<T, BN, NG> Slab;
    constexpr int WIT = Slab::WIT;
    constexpr int PPF = 10;                             // host: (2 toh + 2)(2 tow + 2) x 4 granules <= 2560
    const int ptotal = PH * PW * NG;
    const int nch0 = a.g.C[0] / CK, nch = nch0 + a.g.C[1] / CK;
    u32x4 wv[WIT], pv[PPF];

    Slab slab;
    slab.init(a, n0, tid);
    // both sources have the conv input's extent (direct mode): one pixel index per staged granule serves both
    const int Hs = a.g.Hi, Ws = a.g.Wi;
    const int C0 = a.g.C[0], C1 = a.g.C[1];
    const __amdgpu_buffer_rsrc_t rimg0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.g.src[0] + (size_t)b * Hs * Ws * C0 * ES), 0, Hs * Ws * C0 * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rimg1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(C1 > 0 ? a.g.src[1] + (size_t)b * Hs * Ws * C1 * ES : a.g.src[0]), 0, C1 > 0 ? Hs * Ws * C1 * ES : 0, 0x00020000);
    int pidx[PPF], plds[PPF];                           // pixel index in the image (-1: outside), LDS offset
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        const int i = it * NT + tid;
        const int pix = i / NG, cg = i - pix * NG;
        const int py = mdiv(pix, a.m_pw), px = pix - py * PW;
        const int vy = oy0 - 1 + py, vx = ox0 - 1 + px;
        const bool inb = (i < ptotal) && ((unsigned)vy < (unsigned)Hs) && ((unsigned)vx < (unsigned)Ws);
        pidx[it] = inb ? vy * Ws + vx : -1;
        plds[it] = (py * a.pwp + px) * PIXP + cg * 16;
    }
    const int cgoff = (tid & (NG - 1)) * 16;            // NT is a multiple of NG: a thread's granules all have cg = tid % NG
    auto load_p = [&](int k, int dead) {
        const bool second = !dead && k >= nch0;
        const int Cs = second ? C1 : C0;
        const int so = dead ? 0 : (second ? k - nch0 : k) * CK * ES;
        const int pixB = Cs * ES;
        if (!second) {
#pragma unroll
            for (int it = 0; it < PPF; ++it)
                pv[it] = bld16(rimg0, ((pidx[it] >= 0) ? pidx[it] * pixB + cgoff : OOB_OFF) | dead, so);
        } else {
#pragma unroll
            for (int it = 0; it < PPF; ++it) pv[it] = bld16(rimg1, (pidx[it] >= 0) ? pidx[it] * pixB + cgoff : OOB_OFF, so);
        }
    };
    slab.load(0, 0, wv);
    load_p(0, 0);
    slab.lds_offsets(tid);
    int pbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        int p = wave * 32 + mf * 16 + l15;
        if (p >= npix) p = 0;
        const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
        pbase[mf] = (oy * a.pwp + ox) * PIXP;               // inside sub-tile (0, 0); patch pixel (0, 0) = input (oy0 - 1, ox0 - 1)
    }
    u32x4 biasv[NF];
    {
        const __amdgpu_buffer_rsrc_t rbias =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) biasv[nf] = bld16(rbias, (n0 + nf * 16 + kg * 4) * 4, 0);
    }
    f32x4 acc[4][2][NF];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[c][mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sub_row = a.toh * a.pwp * PIXP, sub_col = a.tow * PIXP;       // LDS offsets of sub-tiles (1, 0) and (0, 1)

    for (int k = 0; k < nch; ++k) {
        __syncthreads();
        slab.store(sW, tid, wv);
#pragma unroll
        for (int it = 0; it < PPF; ++it)
            if (it * NT + tid < ptotal) st16(sP + plds[it], pv[it]);
        __syncthreads();
        {
            const int dead = (k + 1 < nch) ? 0 : OOB_OFF;
            slab.load(k + 1, dead, wv);
            load_p(k + 1, dead);
        }
#pragma unroll
        for (int m = 0; m < STEPS; ++m) {
            const int ky = m / 3, kx = m - 3 * ky;
            u32x4 bv[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bv[nf] = ld16(sW + (nf * 16 + l15) * WROW + (4 * m + kg) * 16);
            const int toff = (ky * a.pwp + kx) * PIXP + kg * 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int soff = toff + (c >> 1) * sub_row + (c & 1) * sub_col;
                u32x4 av[2];
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) av[mf] = ld16(sP + pbase[mf] + soff);
                if constexpr (ES == 2) {
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[c][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(bf16x8, bv[nf]), __builtin_bit_cast(bf16x8, av[mf]), acc[c][mf][nf], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NF; ++nf)
                                acc[c][mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                    __uint_as_float(bv[nf][j]), __uint_as_float(av[mf][j]), acc[c][mf][nf], 0, 0, 0);
                }
            }
        }
    }
    mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[8 * NF]>(acc));

    const int img_bytes = a.Ho * a.Wo * a.N * ES;
    const __amdgpu_buffer_rsrc_t rout =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * img_bytes), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.mask ? a.mask + (size_t)b * img_bytes : a.out), 0, a.mask ? img_bytes : 0, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        Epi<T, NF> ep;
        ep.offsets(a, oy0 + (c >> 1) * a.toh, ox0 + (c & 1) * a.tow, n0, wave, l15, kg);
        ep.prefetch(a, rout, rmask, 0);
        ep.finish(a, acc[c], biasv, rout, 0);
    }
}

// --------------------------------------------------------------------------------------------- //
// forward / input-gradient kernel, weights-resident persistent form                              //
// --------------------------------------------------------------------------------------------- //
// For single-chunk layers (C <= 32 bf16 / 16 f32: the high-resolution ends of the networks) a workgroup's work is a
// few KB of traffic and ~10 MFMAs per wave, so the one-tile-per-workgroup kernel is bound by its fixed costs (address
// setup, weight staging, two exposed memory round trips) times ~10^4 workgroups.  Here a workgroup stages its weight
// slab ONCE, then loops over output tiles (grid-stride), prefetching the next tile's patch into registers while the
// current tile's MFMAs and epilogue run.  The fp32 output staging tile aliases the patch region.
// S2 (round 5): the same for STRIDE-2 single-chunk layers (enc2a, PoseNet's conv2 / conv3 at large batch).  The one-tile kernel
// spends such a workgroup's life on its set-up -- nine staged granules per thread, each with its own address arithmetic, for ONE
// chunk: 42 % of the wave cycles are instruction issue at 9 % MFMA-pipe busy (profiles/r5_conv_pmc_b64.json) -- and MIOpen's
// implicit GEMM beats it there (enc2a at 64 frames 54.9 vs 32.4 us, profiles/r5_conv_layers_b64.csv).  Here the per-thread patch
// coordinates are tile-invariant, the 17 x 33-pixel patch of the NEXT tile is in flight (nine registers) under the current tile's
// MFMAs, the pixel pitch is the stride-2 one (conv_common.h pitch_bytes_s2) and a 64-wide channel tile reads every input pixel once.
template <typename T, int BN, int NG, bool S2 = false>
__global__ __launch_bounds__(NT) void k_conv3x3_res(const ConvK a, int ntiles, uint32_t m_tpi, uint32_t m_tx) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int NGR = 9 * NG;
    constexpr int STEPS = (NGR + 3) / 4;
    constexpr int WROW = wrow_bytes(STEPS * 4);
    constexpr int PIXP = S2 ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16);
    constexpr int SS = S2 ? 2 : 1;
    constexpr int NF = BN / 16;
    constexpr int OUTP = BN + 4;
    constexpr int WTOT = BN * STEPS * 4;
    constexpr int WIT = (WTOT + NT - 1) / NT;
    // staged patch granules per thread -- stride 1: <= 10 x 18 pixels x NG <= 768; stride 2: <= 17 x 33 x NG <= 2304 (host-checked)
    constexpr int PPF = S2 ? (17 * 33 * NG + NT - 1) / NT : 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + BN * WROW;
    float* sOut = reinterpret_cast<float*>(sP);    // pool2 only: aliases the patch (dead once the MFMAs have read it)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.y * BN;
    const int PH = (a.toh - 1) * SS + 3, PW = (a.tow - 1) * SS + 3;
    const int npix = a.toh * a.tow;
    const int ptotal = PH * PW * NG;
    const int tiles_per_img = a.tiles_x * a.tiles_y;

    // weights: staged once
    {
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * 9 * a.Ctot * ES, 0x00020000);
        u32x4 wv[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = it * NT + tid;
            const int n = i / (STEPS * 4), gi = i - n * (STEPS * 4);
            int off = OOB_OFF;
            if (i < WTOT && gi < NGR && n0 + n < a.N) {
                const int tap = gi / NG, cg = gi - tap * NG;
                off = (((n0 + n) * 9 + tap) * a.Ctot + cg * G) * ES;
            }
            wv[it] = bld16(rw, off, 0);
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = it * NT + tid;
            if (i < WTOT) { const int n = i / (STEPS * 4), gi = i - n * (STEPS * 4); st16(sW + n * WROW + gi * 16, wv[it]); }
        }
    }
    int pbase[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
        int p = wave * 32 + mf * 16 + l15;
        if (p >= npix) p = 0;
        const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
        pbase[mf] = ((oy * SS) * a.pwp + ox * SS) * PIXP;
    }
    // patch granules of this thread: (pixel, granule) inside the patch are tile-invariant
    int ppy[PPF], ppx[PPF], pcg[PPF];
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        const int i = it * NT + tid;
        const int pix = i / NG;
        pcg[it] = i - pix * NG;
        ppy[it] = (i < ptotal) ? mdiv(pix, a.m_pw) : 0x4000; ppx[it] = pix - mdiv(pix, a.m_pw) * PW;   // 0x4000: never in range
    }
    const int Hs = a.g.Hs[0], Ws = a.g.Ws[0], Cs = a.g.C[0], mode = a.g.mode[0];
    const int sh = (mode != MODE_DIRECT) ? 1 : 0, par = (mode == MODE_DILATE) ? 1 : 0;
    // descriptors over the WHOLE tensors (host guarantees < 1 GiB each): offsets >= the size (OOB_OFF) read as zero
    const int nimg = ntiles / tiles_per_img;
    const int img_in = Hs * Ws * Cs * ES, img_out = a.Ho * a.Wo * a.N * ES;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.g.src[0], 0, nimg * img_in, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, nimg * img_out, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.mask ? a.mask : a.out), 0, a.mask ? nimg * img_out : 0, 0x00020000);
    u32x4 biasv[NF];
    {
        const __amdgpu_buffer_rsrc_t rbias =
            __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, 0x00020000);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) biasv[nf] = bld16(rbias, (n0 + nf * 16 + kg * 4) * 4, 0);
    }
    struct TileO { int b, oy0, ox0; };
    auto tile_origin = [&](int tile) -> TileO {      // wave-uniform; magic divisions (host checks the ranges)
        const int t = __builtin_amdgcn_readfirstlane(tile);
        const int b = mdiv(t, m_tpi), tr_ = t - b * tiles_per_img;
        const int ty = mdiv(tr_, m_tx), tx = tr_ - ty * a.tiles_x;
        return TileO{b, ty * a.toh, tx * a.tow};
    };
    u32x4 pv[PPF];
    auto load_p = [&](const TileO& o) {
        const int iy0 = o.oy0 * SS - 1, ix0 = o.ox0 * SS - 1;
        const int soff = o.b * img_in;
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int vy = iy0 + ppy[it], vx = ix0 + ppx[it];
            const bool inb = ((unsigned)vy < (unsigned)a.g.Hi) && ((unsigned)vx < (unsigned)a.g.Wi) && (((vy | vx) & par) == 0);
            const int off = inb ? (((vy >> sh) * Ws + (vx >> sh)) * Cs + pcg[it] * G) * ES : OOB_OFF;
            pv[it] = bld16(rx, off, soff);
        }
    };

    int tile = blockIdx.x;
    TileO cur = tile_origin(tile < ntiles ? tile : 0);
    if (tile < ntiles) load_p(cur);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                      // the previous tile's MFMAs (pool2: epilogue) have finished reading sP
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int i = it * NT + tid;
            if (i < ptotal) st16(sP + (ppy[it] * a.pwp + ppx[it]) * PIXP + pcg[it] * 16, pv[it]);
        }
        __syncthreads();
        const int next = tile + gridDim.x;
        const TileO nxt = tile_origin(next < ntiles ? next : 0);
        if (next < ntiles) load_p(nxt);       // in flight during the MFMAs and the epilogue below
        Epi<T, NF> ep;                        // mask / accumulate operands of THIS tile: also in flight during its MFMAs
        if (!a.pool2) {
            ep.offsets(a, cur.oy0, cur.ox0, n0, wave, l15, kg);
            ep.prefetch(a, rout, rmask, cur.b * img_out);
        }

        f32x4 acc[2][NF];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < STEPS; ++m) {
            const int gi = 4 * m + kg;
            int tap = gi / NG;
            const int cg = gi - tap * NG;
            tap = min(tap, 8);
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int aoff = (ky * a.pwp + kx) * PIXP + cg * 16;
            u32x4 av[2], bv[NF];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) av[mf] = ld16(sP + pbase[mf] + aoff);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bv[nf] = ld16(sW + (nf * 16 + l15) * WROW + gi * 16);
            // operands swapped (A = weights, B = pixels): see k_conv3x3
            if constexpr (ES == 2) {
#pragma unroll
                for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, bv[nf]), __builtin_bit_cast(bf16x8, av[mf]), acc[mf][nf], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                __uint_as_float(bv[nf][j]), __uint_as_float(av[mf][j]), acc[mf][nf], 0, 0, 0);
            }
        }
        mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[2 * NF]>(acc));
        if (!a.pool2) {
            ep.finish(a, acc, biasv, rout, cur.b * img_out);
        } else {
            __syncthreads();                  // every wave is done reading the patch: it becomes the output tile
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    *reinterpret_cast<f32x4*>(&sOut[(wave * 32 + mf * 16 + l15) * OUTP + nf * 16 + 4 * kg]) = acc[mf][nf];
            __syncthreads();
            conv_epilogue<T, BN>(a, sOut, cur.b, cur.oy0, cur.ox0, n0, tid);
        }
        cur = nxt;
    }
}

// --------------------------------------------------------------------------------------------- //
// host side                                                                                      //
// --------------------------------------------------------------------------------------------- //
template <typename T, int BN, int NG, int DEPTH, bool TAIL, int NCH = 0, int NTH = 256>
int launch_conv_tail(const ConvK& k, int B, hipStream_t s) {
    constexpr int STEPS = (9 * NG + 3) / 4;
    constexpr int WROW = wrow_bytes(STEPS * 4), PIXP = TAIL ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16);
    const int S = k.g.stride;
    const int PH = (k.toh - 1) * S + 3;
    size_t lds = (size_t)BN * WROW + (size_t)PH * k.pwp * PIXP;
    const size_t eplds = (size_t)(NTH / 2) * (BN + 4) * 4;
    if (eplds > lds) lds = eplds;
    COLVO_CHECK_ARG(lds <= 160 * 1024, "conv: tile needs %zu bytes of LDS", lds);
    static size_t configured = 0;   // per instantiation
    if (lds > 48 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3<T, BN, NG, DEPTH, TAIL, NCH, NTH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = 160 * 1024;
    }
    const int xcd_on = (int)TUNE(xcd_remap);
    ConvK kk = k;
    kk.ntn = (k.N + BN - 1) / BN;
    kk.xcd = xcd_on;
    const long long nwg_ll = (long long)k.tiles_x * k.tiles_y * kk.ntn * B;
    COLVO_CHECK_ARG(nwg_ll < (1ll << 30), "conv: too many workgroups");
    dim3 grid((unsigned)nwg_ll, 1, 1);
#ifdef COLVO_ABLATE
    ConvK ka = kk;
    { const char* e = getenv("COLVO_ABL"); ka.abl = e ? atoi(e) : 0; }
    ka.trace = nullptr;
    static long long* tbuf = nullptr;
    static int tcount = 0;
    const size_t nwg = (size_t)grid.x * grid.y * grid.z;
    const bool tracing = getenv("COLVO_TRACE") && nwg <= (1u << 16);
    if (tracing) {
        if (!tbuf) hipMalloc(&tbuf, (size_t)(1u << 16) * 8 * sizeof(long long));
        hipMemsetAsync(tbuf, 0, nwg * 8 * sizeof(long long), s);
        ka.trace = tbuf;
    }
    colvo::launch((k_conv3x3<T, BN, NG, DEPTH, TAIL, NCH, NTH>), grid, dim3(NTH), lds, s, ka);
    COLVO_CHECK_LAUNCH("k_conv3x3");
    if (tracing && (++tcount % atoi(getenv("COLVO_TRACE"))) == 0) {     // every n-th launch: print the phase statistics
        hipStreamSynchronize(s);
        std::vector<long long> h(nwg * 8);
        hipMemcpy(h.data(), tbuf, nwg * 8 * sizeof(long long), hipMemcpyDeviceToHost);
        long long t0 = h[0], t1 = 0;
        for (size_t i = 0; i < nwg; ++i) { t0 = std::min(t0, h[i * 8]); t1 = std::max(t1, h[i * 8 + 5] ? h[i * 8 + 5] : h[i * 8 + 4]); }
        double ph[5] = {0, 0, 0, 0, 0}, start = 0, life = 0;
        for (size_t i = 0; i < nwg; ++i) {
            const long long* r = &h[i * 8];
            const long long end = r[5] ? r[5] : r[4];
            ph[0] += r[1] - r[0]; ph[1] += r[2] - r[1]; ph[2] += (r[3] ? r[3] - r[2] : 0); ph[3] += r[4] - (r[3] ? r[3] : r[2]);
            ph[4] += end - r[4];
            start += r[0] - t0; life += end - r[0];
        }
        const double c = 0.01 / nwg;   // 100 MHz ticks -> us, mean over workgroups
        fprintf(stderr, "[trace] BN=%d NG=%d grid=%ux%ux%u nch=%d | span %.2f us | mean wg: start +%.2f life %.2f = setup %.2f"
                " first-stage %.2f chunk0 %.2f rest %.2f epilogue %.2f\n", BN, NG, grid.x, grid.y, grid.z,
                (k.g.C[0] + k.g.C[1]) / (NG * TT<T>::G), (t1 - t0) * 0.01, start * c, life * c, ph[0] * c, ph[1] * c, ph[2] * c,
                ph[3] * c, ph[4] * c);
    }
    return 0;
#endif
    colvo::launch((k_conv3x3<T, BN, NG, DEPTH, TAIL, NCH, NTH>), grid, dim3(NTH), lds, s, kk);
    COLVO_CHECK_LAUNCH("k_conv3x3");
    return 0;
}

template <typename T, int BN, int NG, int DEPTH = 1>
int launch_conv(const ConvK& k, int B, hipStream_t s) {
    const int S = k.g.stride;
    const long ptotal = (long)((k.toh - 1) * S + 3) * ((k.tow - 1) * S + 3) * NG;
    // (A persistent multi-chunk form -- grid = resident slots, a workgroup walks several pixel tiles with the register prefetch
    // running through the tile boundary -- was built and measured this round: 5-25 % SLOWER on every DepthNet layer at B = 16
    // (up3 18.8 -> 22.7 us): it needs 162 VGPRs, i.e. 3 instead of 4 resident workgroups per CU, and an evened-out walk leaves
    // 2.5 workgroups per CU.  Removed)
    if (ptotal > 3 * NT) return launch_conv_tail<T, BN, NG, (DEPTH > 2 ? 2 : DEPTH), true>(k, B, s);   // depth 3 would spill
    if constexpr (DEPTH >= 2) {
        const int nch = (k.g.C[0] + k.g.C[1]) / (NG * TT<T>::G);
        if (nch == 8) return launch_conv_tail<T, BN, NG, DEPTH, false, 8>(k, B, s);
        if (nch == 16) return launch_conv_tail<T, BN, NG, DEPTH, false, 16>(k, B, s);
    }
    return launch_conv_tail<T, BN, NG, DEPTH, false>(k, B, s);
}

template <typename T, int BN, int NG, bool S2 = false>
int launch_conv_res(const ConvK& k, int B, hipStream_t s) {
    constexpr int STEPS = (9 * NG + 3) / 4;
    constexpr int WROW = wrow_bytes(STEPS * 4), PIXP = S2 ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16);
    const int PH = (k.toh - 1) * (S2 ? 2 : 1) + 3;
    const size_t p_or_out = std::max((size_t)PH * k.pwp * PIXP, (size_t)BM * (BN + 4) * 4);
    const size_t lds = (size_t)BN * WROW + p_or_out;
    COLVO_CHECK_ARG(lds <= 160 * 1024, "conv (weights-resident): tile needs %zu bytes of LDS", lds);
    static bool configured = false;             // per instantiation
    if (lds > 48 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv3x3_res<T, BN, NG, S2>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = true;
    }
    const int ntiles = k.tiles_x * k.tiles_y * B;
    // workgroups per CU, each walking ntiles / gx tiles (stride 2: as many as its LDS footprint lets a CU hold)
    const int per_cu = S2 ? std::max(1, std::min((int)TUNE(res_wg_per_cu), (int)(160 * 1024 / lds))) : (int)TUNE(res_wg_per_cu);
    int gx = 256 * per_cu;
    if (gx > ntiles) gx = ntiles;
    dim3 grid(gx, (k.N + BN - 1) / BN, 1);
    if (S2) form_hit(FORM_CONV_RES_S2);
    colvo::launch((k_conv3x3_res<T, BN, NG, S2>), grid, dim3(NT), lds, s, k, ntiles, mdiv_magic(k.tiles_x * k.tiles_y),
                       mdiv_magic(k.tiles_x));
    COLVO_CHECK_LAUNCH("k_conv3x3_res");
    return 0;
}

// stride-2 single-chunk layers on large grids: weights-resident persistent kernel with the next tile's patch in flight (see
// k_conv3x3_res).  -1: the layer does not qualify.
template <typename T>
int try_launch_conv_res_s2(const ConvK& k, int B, int ng, hipStream_t s) {
    if (!TUNE(res_s2) || k.g.stride != 2 || k.nsplit != 0 || k.pool2 || k.g.C[1] != 0 || k.g.mode[0] != MODE_DIRECT) return -1;
    if (k.g.C[0] != ng * TT<T>::G || k.accumulate || k.mask) return -1;       // one chunk; forward only (its input gradient is k_dgrad_s2)
    const long long src_bytes = (long long)B * k.g.Hs[0] * k.g.Ws[0] * k.g.C[0] * TT<T>::ES;
    const long long out_bytes = (long long)B * k.Ho * k.Wo * k.N * TT<T>::ES;
    const long long tpi = (long long)k.tiles_x * k.tiles_y;
    if (src_bytes >= 0x40000000LL || out_bytes >= 0x40000000LL || tpi < 2 || k.tiles_x < 2 || tpi * tpi * B >= 0x100000000LL) return -1;
    if (tpi * B < TUNE(res_s2_min_tiles)) return -1;
    const long ptotal = (long)((k.toh - 1) * 2 + 3) * ((k.tow - 1) * 2 + 3) * ng;
    if (ptotal > (long)((17 * 33 * ng + NT - 1) / NT) * NT) return -1;        // (what the kernel's register prefetch holds)
    if (k.N >= 64) {
        switch (ng) {
            case 4: return launch_conv_res<T, 64, 4, true>(k, B, s);
            case 2: return launch_conv_res<T, 64, 2, true>(k, B, s);
            default: return -1;
        }
    }
    if (k.N >= 32) {
        switch (ng) {
            case 4: return launch_conv_res<T, 32, 4, true>(k, B, s);
            case 2: return launch_conv_res<T, 32, 2, true>(k, B, s);
            default: return -1;
        }
    }
    return -1;
}

template <typename T, int BN>
int launch_conv_ng(const ConvK& k, int B, int ng, hipStream_t s) {
    // single-chunk layers with narrow outputs: weights-resident persistent kernel
    if constexpr (BN <= 32) {
        const int ck = ng * TT<T>::G;
        const long long src_bytes = (long long)B * k.g.Hs[0] * k.g.Ws[0] * k.g.C[0] * TT<T>::ES;
        const long long out_bytes = (long long)B * k.Ho * k.Wo * k.N * TT<T>::ES / (k.pool2 ? 4 : 1);
        const long long tpi = (long long)k.tiles_x * k.tiles_y;
        const long res_min_tiles = TUNE(res_min_tiles);   // tuning knob
        if (k.nsplit == 0 && k.g.C[1] == 0 && k.g.C[0] == ck && k.g.stride == 1 && src_bytes < 0x40000000LL && out_bytes < 0x40000000LL &&
            tpi >= 2 && k.tiles_x >= 2 && tpi * tpi * B < 0x100000000LL &&        // magic-division ranges
            tpi * B >= res_min_tiles) {
            switch (ng) {
                case 4: return launch_conv_res<T, BN, 4>(k, B, s);
                case 2: return launch_conv_res<T, BN, 2>(k, B, s);
                default: return launch_conv_res<T, BN, 1>(k, B, s);
            }
        }
    }
    switch (ng) {
        case 4:
            // many-chunk layers at the lowest resolutions: two chunks in flight, because one MFMA phase (~0.5 us) is
            // shorter than the global-load latency it is supposed to hide
            if constexpr (BN == 32) {
                // measured: pays only when the grid is about one workgroup per CU (it costs occupancy: ~190 VGPRs)
                const int depth2_min = (int)TUNE(depth2_min_chunks);
                const long lone_max = TUNE(lone_max_wgs);
                const long wgs = (long)k.tiles_x * k.tiles_y * B * ((k.N + BN - 1) / BN);
                if (wgs <= lone_max && (k.g.C[0] + k.g.C[1]) / (4 * TT<T>::G) >= depth2_min) {
                    return launch_conv<T, BN, 4, 2>(k, B, s);     // (a three-chunk ring measured no better)
                }
            }
            return launch_conv<T, BN, 4>(k, B, s);
        case 2: return launch_conv<T, BN, 2>(k, B, s);
        default: return launch_conv<T, BN, 1>(k, B, s);
    }
}

inline void set_tile(ConvK& k, const Tile& t) {
    k.toh = t.toh; k.tow = t.tow;
    k.tiles_x = (k.Wo + t.tow - 1) / t.tow; k.tiles_y = (k.Ho + t.toh - 1) / t.toh;
    k.m_tow = mdiv_magic(t.tow); k.m_pw = mdiv_magic((t.tow - 1) * k.g.stride + 3);
    const int pw = (t.tow - 1) * k.g.stride + 3;
    k.pwp = t.pwp > pw ? t.pwp : pw;
    // patches beyond 3 x 256 granules are staged partly by the linear tail loop: no row padding there
    if ((long)((t.toh - 1) * k.g.stride + 3) * pw * 4 > 3 * NT) k.pwp = pw;
}

template <typename T, int BN>
int launch_conv_wide(const ConvK& k, int B, int ng, hipStream_t s) {
    switch (ng) {
        case 4: return launch_conv_tail<T, BN, 4, 1, false, 0, 512>(k, B, s);
        case 2: return launch_conv_tail<T, BN, 2, 1, false, 0, 512>(k, B, s);
        default: return launch_conv_tail<T, BN, 1, 1, false, 0, 512>(k, B, s);
    }
}

// `even`: tile extents must be even (2x2 sum-pool epilogue)
template <typename T>
int launch_conv_t(ConvK k, int B, bool even, hipStream_t s) {
    constexpr int G = TT<T>::G;
    int ng = 4;
    for (int i = 0; i < 2; ++i)
        if (k.g.C[i] > 0) while (ng > 1 && (k.g.C[i] % (ng * G)) != 0) ng >>= 1;
    for (int i = 0; i < 2; ++i)
        COLVO_CHECK_ARG(k.g.C[i] % (ng * G) == 0, "conv: channel count %d is not a multiple of %d", k.g.C[i], G);
    // Wide form (512 threads, 256 pixels x BN channels): multi-chunk stride-1 layers whose 256-pixel grid still covers the
    // chip.  Per MFMA it stages 100 (BN 64) / 136 (BN 32) bytes instead of 208, and the CUs' bytes in flight are what bounds
    // these layers.
    {
        // off by default: at B = 16 the 256-pixel grids are 1-1.25 workgroups per CU and measured 10-20 % slower (up3 18.9 ->
        // 21.7 us; gpurun_out/r2_bench_conv_w*.log); the form pays once the grid covers the chip several times (configs[2])
        const int wide_on = (int)TUNE(wide);                 // tuning knob
        const long wide_min_wgs = TUNE(wide_min_wgs);   // tuning knob
        const int wide_min_chunks = (int)TUNE(wide_min_chunks);
        const int nch = (k.g.C[0] + k.g.C[1]) / (ng * G);
        if (wide_on && k.nsplit == 0 && k.g.stride == 1 && k.N >= 32 && nch >= wide_min_chunks) {
            const Tile tw = pick_tile(k.Ho, k.Wo, 1, even, 256);
            const long patch = (long)(tw.toh + 2) * (tw.tow + 2) * ng;
            const int bn = k.N >= 64 ? 64 : 32;
            const long wgs = (long)((k.Ho + tw.toh - 1) / tw.toh) * ((k.Wo + tw.tow - 1) / tw.tow) * B * ((k.N + bn - 1) / bn);
            if (patch <= 3 * 512 && tw.toh * tw.tow > 128 && wgs >= wide_min_wgs) {
                set_tile(k, tw);
                return bn == 64 ? launch_conv_wide<T, 64>(k, B, ng, s) : launch_conv_wide<T, 32>(k, B, ng, s);
            }
        }
    }
    set_tile(k, pick_tile(k.Ho, k.Wo, k.g.stride, even, 128, true));
    if (k.g.stride == 2) {
        const int r = try_launch_conv_res_s2<T>(k, B, ng, s);
        if (r >= 0) return r;
    }
    // Output-channel tile: 64 wide by default; when that grid would leave CUs idle (deep, low-resolution layers at small
    // batch) use 32 -- twice the workgroups, each staging half the weight slab per chunk (the chunk is LDS-bound).
    const long tiles = (long)k.tiles_x * k.tiles_y * B;
    const long bn64_min_wgs = TUNE(bn64_min_wgs);   // tuning knob
    const long bn32_min_wgs = TUNE(bn32_min_wgs);      // tuning knob (16-wide tiles: measured ~neutral)
    // (two-output form: a channel tile must not straddle the two sources -- nsplit is a multiple of 32)
    if (k.nsplit == 0 && k.N >= 64 && tiles * ((k.N + 63) / 64) >= bn64_min_wgs) return launch_conv_ng<T, 64>(k, B, ng, s);
    if (k.N >= 32 && tiles * ((k.N + 31) / 32) >= bn32_min_wgs) return launch_conv_ng<T, 32>(k, B, ng, s);
    return launch_conv_ng<T, 16>(k, B, ng, s);
}

// quad-tile stride-1 kernel (k_conv_q): returns -1 when the layer does not qualify (the caller then takes the one-tile path)
template <typename T>
int try_launch_conv_q(const ConvK& k0, int B, hipStream_t s) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES, CK = 4 * G;
    const int q_on = (int)TUNE(conv_quad);
    // Measured (bf16; us, one-tile -> quad).  16 frames of 256x320: slower everywhere (enc2b 14.8 -> 23.5, iconv3 19.0 -> 34.2,
    // iconv2 22.0 -> 24.2, enc5b 17.1 -> 42.1).  64 frames of 512x640: the 2-4-chunk layers at 1/2-1/4 resolution win (enc2b 199 ->
    // 153, iconv3 265 -> 217, iconv2 370 -> 275), the 8-16-chunk layers lose to the one-tile kernel's two-chunk ring (enc4b 122 ->
    // 164, iconv5 215 -> 297).  Hence: at least 2048 quad-tile workgroups and at most 4 chunks.
    const long min_wgs = TUNE(quad_min_wgs);          // (the tests lower / raise these two through colvo_tune_set)
    const int max_chunks = (int)TUNE(quad_max_chunks);
    const Gather& g = k0.g;
    if (!q_on || g.stride != 1 || k0.pool2 || g.mode[0] != MODE_DIRECT || (g.C[1] > 0 && g.mode[1] != MODE_DIRECT)) return -1;
    if (g.C[0] % CK || g.C[1] % CK || (g.C[0] + g.C[1]) / CK < 2 || (g.C[0] + g.C[1]) / CK > max_chunks) return -1;
    if ((long long)g.Hi * g.Wi * std::max(g.C[0], g.C[1]) * ES >= 0x40000000LL || (long long)k0.Ho * k0.Wo * k0.N * ES >= 0x40000000LL)
        return -1;
    ConvK k = k0;
    const Tile t = pick_tile((k.Ho + 1) / 2, (k.Wo + 1) / 2, 1, false, 128, true, 3);   // the sub-tile: at most half the image each way
    const int PH = 2 * t.toh + 2, PW = 2 * t.tow + 2;
    if (PH * PW * 4 > 10 * NT) return -1;
    k.toh = t.toh; k.tow = t.tow;
    k.tiles_x = (k.Wo + 2 * t.tow - 1) / (2 * t.tow); k.tiles_y = (k.Ho + 2 * t.toh - 1) / (2 * t.toh);
    k.m_tow = mdiv_magic(t.tow); k.m_pw = mdiv_magic(PW);
    k.pwp = PW;
    // sub-tile reads are conflict-free when the sub-tile's own rows are (pick_tile) AND the patch pitch keeps the row phase:
    // keep the padding pick_tile chose relative to ITS patch width (tow + 2)
    if (t.pwp > t.tow + 2) k.pwp = PW + (t.pwp - (t.tow + 2));
    const long tiles = (long)k.tiles_x * k.tiles_y * B;
    const int bn = (k.N > 16 && tiles * ((k.N + 31) / 32) >= min_wgs) ? 32 : 16;
    k.ntn = (k.N + bn - 1) / bn;
    if (tiles * k.ntn < min_wgs) return -1;
    const int xcd_on = (int)TUNE(xcd_remap);
    k.xcd = xcd_on;
    constexpr int WROW = wrow_bytes(36), PIXP = pitch_bytes(64);
    const size_t lds = (size_t)bn * WROW + (size_t)PH * k.pwp * PIXP;
    if (lds > 160 * 1024) return -1;
    static bool configured[2] = {false, false};
    if (lds > 48 * 1024 && !configured[bn == 32]) {
        const void* f = bn == 32 ? reinterpret_cast<const void*>(&k_conv_q<T, 32>) : reinterpret_cast<const void*>(&k_conv_q<T, 16>);
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("conv: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured[bn == 32] = true;
    }
    const dim3 grid((unsigned)(tiles * k.ntn));
    form_hit(FORM_CONV_Q);
    if (bn == 32) colvo::launch((k_conv_q<T, 32>), grid, dim3(NT), lds, s, k);
    else colvo::launch((k_conv_q<T, 16>), grid, dim3(NT), lds, s, k);
    COLVO_CHECK_LAUNCH("k_conv_q");
    return 0;
}

// forward over an up-sampled source (k_conv_up2): tiles over the stored half-size source
template <typename T, int BN, int DEPTH, int NCH>
int launch_conv_up2_inst(ConvK k, int B, hipStream_t s) {
    constexpr int WROW = wrow_bytes(36), PIXP = pitch_bytes(64);
    const size_t lds = (size_t)BN * WROW + (size_t)(k.toh + 2) * k.pwp * PIXP;
    const int xcd_on = (int)TUNE(xcd_remap);
    k.ntn = (k.N + BN - 1) / BN;
    k.xcd = xcd_on;
    const long long nwg = (long long)k.tiles_x * k.tiles_y * k.ntn * B;
    COLVO_CHECK_ARG(nwg < (1ll << 30) && lds <= 48 * 1024, "conv (up-sampled source): bad launch geometry");
    colvo::launch((k_conv_up2<T, BN, DEPTH, NCH>), dim3((unsigned)nwg), dim3(NT), lds, s, k);
    COLVO_CHECK_LAUNCH("k_conv_up2");
    return 0;
}

template <typename T, int BN>
int launch_conv_up2_bn(const ConvK& k, int B, hipStream_t s) {
    const int nch = k.g.C[0] / (4 * TT<T>::G);
    if (nch == 8) return launch_conv_up2_inst<T, BN, 2, 8>(k, B, s);
    if (nch == 16) return launch_conv_up2_inst<T, BN, 2, 16>(k, B, s);
    return launch_conv_up2_inst<T, BN, 1, 0>(k, B, s);
}

// input gradient w.r.t. an up-sampled source (k_dgrad_up2): tiles over the half-size source
template <typename T, int BN, int DEPTH, int NCH, int NG = 4>
int launch_dgrad_up2_inst(ConvK k, int B, hipStream_t s) {
    constexpr int WROW = wrow_bytes((9 * NG + 3) / 4 * 4), PIXP = pitch_bytes(NG * 16);
    const size_t lds = (size_t)BN * WROW + (size_t)(2 * k.toh + 2) * k.pwp * PIXP;
    COLVO_CHECK_ARG(lds <= 160 * 1024, "dgrad (up-sampled source): tile needs %zu bytes of LDS", lds);
    static size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dgrad_up2<T, BN, DEPTH, NCH, NG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("dgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = 160 * 1024;
    }
    const int xcd_on = (int)TUNE(xcd_remap);
    k.ntn = (k.N + BN - 1) / BN;
    k.xcd = xcd_on;
    const long long nwg = (long long)k.tiles_x * k.tiles_y * k.ntn * B;
    COLVO_CHECK_ARG(nwg < (1ll << 30), "dgrad (up-sampled source): too many workgroups");
    colvo::launch((k_dgrad_up2<T, BN, DEPTH, NCH, NG>), dim3((unsigned)nwg), dim3(NT), lds, s, k);
    COLVO_CHECK_LAUNCH("k_dgrad_up2");
    return 0;
}

template <typename T, int BN>
int launch_dgrad_up2_bn(const ConvK& k, int B, hipStream_t s) {
    if (k.g.C[0] % (4 * TT<T>::G)) return launch_dgrad_up2_inst<T, BN, 1, 0, 2>(k, B, s);       // 2-granule chunks
    const int nch = k.g.C[0] / (4 * TT<T>::G);
    if (nch == 8) return launch_dgrad_up2_inst<T, BN, 2, 8>(k, B, s);
    if (nch == 16) return launch_dgrad_up2_inst<T, BN, 2, 16>(k, B, s);
    return launch_dgrad_up2_inst<T, BN, 1, 0>(k, B, s);
}

// parity-decomposed stride-2 input gradient (k_dgrad_s2): tiles over dy, four output pixels per tile position
template <typename T, int BN, int DEPTH, int NCH>
int launch_dgrad_s2_inst(ConvK k, int B, hipStream_t s) {
    constexpr int WROW = wrow_bytes(36), PIXP = pitch_bytes(64);
    const size_t lds = (size_t)BN * WROW + (size_t)(k.toh + 1) * k.pwp * PIXP;
    const int xcd_on = (int)TUNE(xcd_remap);
    k.ntn = (k.N + BN - 1) / BN;
    k.xcd = xcd_on;
    const long long nwg = (long long)k.tiles_x * k.tiles_y * k.ntn * B;
    COLVO_CHECK_ARG(nwg < (1ll << 30) && lds <= 48 * 1024, "dgrad (stride 2): bad launch geometry");
    colvo::launch((k_dgrad_s2<T, BN, DEPTH, NCH>), dim3((unsigned)nwg), dim3(NT), lds, s, k);
    COLVO_CHECK_LAUNCH("k_dgrad_s2");
    return 0;
}

template <typename T, int BN>
int launch_dgrad_s2_bn(const ConvK& k, int B, hipStream_t s) {
    const int nch = k.g.C[0] / (4 * TT<T>::G);
    const long wgs = (long)k.tiles_x * k.tiles_y * B * ((k.N + BN - 1) / BN);
    const long lone_max = TUNE(lone_max_wgs);
    if (wgs <= lone_max) {          // about one workgroup per CU: two chunks in flight, K loop unrolled (see launch_conv_ng)
        if (nch == 8) return launch_dgrad_s2_inst<T, BN, 2, 8>(k, B, s);
        if (nch == 16) return launch_dgrad_s2_inst<T, BN, 2, 16>(k, B, s);
    }
    return launch_dgrad_s2_inst<T, BN, 1, 0>(k, B, s);
}

template <typename T>
int launch_dgrad_s2(ConvK k, int B, hipStream_t s) {
    const Tile t = pick_tile(k.Ho, k.Wo, 1, false, 128, true, 2);
    k.toh = t.toh; k.tow = t.tow; k.pwp = std::max(t.pwp, t.tow + 1);
    k.tiles_x = (k.Wo + t.tow - 1) / t.tow; k.tiles_y = (k.Ho + t.toh - 1) / t.toh;
    k.m_tow = mdiv_magic(t.tow); k.m_pw = mdiv_magic(t.tow + 1);
    return k.N > 16 ? launch_dgrad_s2_bn<T, 32>(k, B, s) : launch_dgrad_s2_bn<T, 16>(k, B, s);
}

}  // namespace
}  // namespace colvo

using namespace colvo;

extern "C" int colvo_conv_fwd(const ColvoConvDesc* d, const void* x0, const void* x1, const void* w_fwd,
                              const float* bias, void* y, colvo_stream_t stream) {
    if (int e = check_desc(d, "colvo_conv_fwd")) return e;
    COLVO_CHECK_ARG(x0 && w_fwd && y && (d->C1 == 0 || x1), "colvo_conv_fwd: null pointer argument");
    ConvK k{};
    fill_gather(d, x0, d->C1 ? x1 : nullptr, k.g);
    k.Ho = d->Ho; k.Wo = d->Wo;
    k.w = (const char*)w_fwd; k.Ctot = d->C0 + d->C1; k.N = d->Cout;
    k.bias = bias; k.relu = d->relu; k.out = (char*)y; k.mask = nullptr; k.accumulate = 0; k.pool2 = 0;
    {
        // single up-sampled source in whole 32-channel chunks: four output pixels per source position (k_conv_up2; the one-chunk
        // full-resolution layer up1 too: 23.2 -> 18.4 us against the weights-resident one-tile kernel)
        const int up2_on = (int)TUNE(conv_up2);
        const int es = d->dtype == COLVO_F32 ? 4 : 2, ck = d->dtype == COLVO_F32 ? 16 : 32;
        const long long in_bytes = (long long)(d->Hi / 2) * (d->Wi / 2) * d->C0 * es, out_bytes = (long long)d->Ho * d->Wo * d->Cout * es;
        const int up2_min_chunks = (int)TUNE(up2_min_chunks);   // tuning knob
        if (up2_on && d->up0 && d->C1 == 0 && d->stride == 1 && d->C0 % ck == 0 && d->C0 / ck >= up2_min_chunks &&
            in_bytes < 0x40000000LL && out_bytes < 0x40000000LL) {
            ConvK u = k;
            u.g.mode[0] = MODE_DIRECT;                    // read the stored half-size source as it is
            u.g.Hi = d->Hi / 2; u.g.Wi = d->Wi / 2;
            u.Ho = d->Hi / 2; u.Wo = d->Wi / 2;           // tiles run over source positions; the kernel writes (2 Ho) x (2 Wo)
            const Tile t = pick_tile(u.Ho, u.Wo, 1, false, 128, true, 3);
            if ((t.toh + 2) * (t.tow + 2) * 4 <= 3 * NT) {
                u.toh = t.toh; u.tow = t.tow; u.pwp = std::max(t.pwp, t.tow + 2);
                u.tiles_x = (u.Wo + t.tow - 1) / t.tow; u.tiles_y = (u.Ho + t.toh - 1) / t.toh;
                u.m_tow = mdiv_magic(t.tow); u.m_pw = mdiv_magic(t.tow + 2);
                hipStream_t s = (hipStream_t)stream;
                // 16-wide channel tiles while 32-wide ones would leave CUs without a workgroup
                const long bn16_max = TUNE(up2_bn16_max_wgs);   // tuning knob
                const long wgs32 = (long)u.tiles_x * u.tiles_y * d->B * ((u.N + 31) / 32);
                const bool wide = u.N > 16 && wgs32 > bn16_max;
                if (d->dtype == COLVO_F32)
                    return wide ? launch_conv_up2_bn<float, 32>(u, d->B, s) : launch_conv_up2_bn<float, 16>(u, d->B, s);
                return wide ? launch_conv_up2_bn<bf16_t, 32>(u, d->B, s) : launch_conv_up2_bn<bf16_t, 16>(u, d->B, s);
            }
        }
    }
    {
        // large grids of ordinary stride-1 layers: the register-tiled kernel (conv_rt.hip), then the quad-tile kernel
        int r = try_launch_conv_rt(k, d->B, d->dtype, (hipStream_t)stream);
        if (r >= 0) return r;
        r = d->dtype == COLVO_F32 ? try_launch_conv_q<float>(k, d->B, (hipStream_t)stream)
                                  : try_launch_conv_q<bf16_t>(k, d->B, (hipStream_t)stream);
        if (r >= 0) return r;
    }
    return d->dtype == COLVO_F32 ? launch_conv_t<float>(k, d->B, false, (hipStream_t)stream)
                                 : launch_conv_t<bf16_t>(k, d->B, false, (hipStream_t)stream);
}

extern "C" int colvo_conv_dgrad(const ColvoConvDesc* d, int src, const void* dy, const void* w_bwd,
                                const void* relu_mask, void* dx, int accumulate, colvo_stream_t stream) {
    if (int e = check_desc(d, "colvo_conv_dgrad")) return e;
    COLVO_CHECK_ARG(dy && w_bwd && dx, "colvo_conv_dgrad: null pointer argument");
    COLVO_CHECK_ARG(src == 0 || (src == 1 && d->C1 > 0), "colvo_conv_dgrad: bad source index %d", src);
    const int es = d->dtype == COLVO_F32 ? 4 : 2;
    const int Csrc = src == 0 ? d->C0 : d->C1;
    const int coff = src == 0 ? 0 : d->C0;
    const int up = src == 0 ? d->up0 : d->up1;
    ConvK k{};
    {
        // stride 2, even input extent, 32-channel chunks of dy: the parity-decomposed kernel (a quarter of the MFMAs)
        const int s2_on = (int)TUNE(dgrad_s2);
        const int ck = d->dtype == COLVO_F32 ? 16 : 32;
        const long long out_bytes = (long long)d->Hi * d->Wi * Csrc * es, in_bytes = (long long)d->Ho * d->Wo * d->Cout * es;
        if (s2_on && d->stride == 2 && !up && d->Hi == 2 * d->Ho && d->Wi == 2 * d->Wo && d->Cout % ck == 0 &&
            out_bytes < 0x40000000LL && in_bytes < 0x40000000LL) {
            k.g.src[0] = (const char*)dy; k.g.src[1] = nullptr;
            k.g.C[0] = d->Cout; k.g.C[1] = 0;
            k.g.Hs[0] = d->Ho; k.g.Ws[0] = d->Wo; k.g.Hs[1] = k.g.Ws[1] = 0;
            k.g.mode[0] = k.g.mode[1] = MODE_DIRECT;
            k.g.Hi = d->Ho; k.g.Wi = d->Wo; k.g.stride = 1;
            k.Ho = d->Ho; k.Wo = d->Wo;                // tiles run over dy; the kernel writes a (2 Ho) x (2 Wo) image
            k.w = (const char*)w_bwd + (size_t)coff * 9 * d->Cout * es; k.Ctot = d->Cout; k.N = Csrc;
            k.bias = nullptr; k.relu = 0; k.out = (char*)dx; k.mask = (const char*)relu_mask;
            k.accumulate = accumulate; k.pool2 = 0;
            return d->dtype == COLVO_F32 ? launch_dgrad_s2<float>(k, d->B, (hipStream_t)stream)
                                         : launch_dgrad_s2<bf16_t>(k, d->B, (hipStream_t)stream);
        }
    }
    {
        // up-sampled source, stride 1, 32-channel chunks of dy: the 2x2 sum-pool folded into the K loop (k_dgrad_up2)
        const int up2_on = (int)TUNE(dgrad_up2);
        const int ck = d->dtype == COLVO_F32 ? 16 : 32;
        const long long out_bytes = (long long)(d->Hi / 2) * (d->Wi / 2) * Csrc * es, in_bytes = (long long)d->Ho * d->Wo * d->Cout * es;
        if (up2_on && up && d->stride == 1 && d->Cout % (ck / 2) == 0 && out_bytes < 0x40000000LL && in_bytes < 0x40000000LL) {
            ConvK u{};
            u.g.src[0] = (const char*)dy; u.g.src[1] = nullptr;
            u.g.C[0] = d->Cout; u.g.C[1] = 0;
            u.g.Hs[0] = d->Ho; u.g.Ws[0] = d->Wo; u.g.Hs[1] = u.g.Ws[1] = 0;
            u.g.mode[0] = u.g.mode[1] = MODE_DIRECT;
            u.g.Hi = d->Ho; u.g.Wi = d->Wo; u.g.stride = 1;
            u.Ho = d->Hi / 2; u.Wo = d->Wi / 2;          // tiles and output: the stored half-size source
            u.w = (const char*)w_bwd + (size_t)coff * 9 * d->Cout * es; u.Ctot = d->Cout; u.N = Csrc;
            u.bias = nullptr; u.relu = 0; u.out = (char*)dx; u.mask = (const char*)relu_mask;
            u.accumulate = accumulate; u.pool2 = 0;
            const Tile t = pick_tile(u.Ho, u.Wo, 2, false, 128, true, 4);   // patch rows of 2 tow + 2 pixels, pixel stride 2
            if ((2 * t.toh + 2) * (2 * t.tow + 2) * 4 <= 10 * NT) {                 // (x NG / 4 granules <= PPF x 256 for either NG)
                u.toh = t.toh; u.tow = t.tow; u.pwp = std::max(t.pwp, 2 * t.tow + 2);
                u.tiles_x = (u.Wo + t.tow - 1) / t.tow; u.tiles_y = (u.Ho + t.toh - 1) / t.toh;
                u.m_tow = mdiv_magic(t.tow); u.m_pw = mdiv_magic(2 * t.tow + 2);
                hipStream_t s = (hipStream_t)stream;
                // only where the grid still covers the chip: at batch 16 the 1/8- and 1/16-resolution layers measured 1-2 us
                // SLOWER in this form (up5 19.4 -> 21.0, up4 18.2 -> 19.4; up3 20.7 -> 19.4, up2 26.8 -> 19.1)
                // (read at every call, not once: the tests switch it to reach this kernel with small shapes)
                const long min_wgs = TUNE(dgrad_up2_min_wgs);
                const long wgs32 = (long)u.tiles_x * u.tiles_y * d->B * ((u.N + 31) / 32);
                if (wgs32 >= min_wgs) {
                    if (d->dtype == COLVO_F32)
                        return u.N > 16 ? launch_dgrad_up2_bn<float, 32>(u, d->B, s) : launch_dgrad_up2_bn<float, 16>(u, d->B, s);
                    return u.N > 16 ? launch_dgrad_up2_bn<bf16_t, 32>(u, d->B, s) : launch_dgrad_up2_bn<bf16_t, 16>(u, d->B, s);
                }
            }
        }
    }
    // the conv input is dy (Cout channels), dilated by zero insertion when the forward stride was 2
    k.g.src[0] = (const char*)dy; k.g.src[1] = nullptr;
    k.g.C[0] = d->Cout; k.g.C[1] = 0;
    k.g.Hs[0] = d->Ho; k.g.Ws[0] = d->Wo; k.g.Hs[1] = k.g.Ws[1] = 0;
    k.g.mode[0] = d->stride == 2 ? MODE_DILATE : MODE_DIRECT; k.g.mode[1] = MODE_DIRECT;
    k.g.Hi = d->stride == 2 ? 2 * d->Ho : d->Ho;
    k.g.Wi = d->stride == 2 ? 2 * d->Wo : d->Wo;
    k.g.stride = 1;
    k.Ho = d->Hi; k.Wo = d->Wi;                    // gradient w.r.t. the (virtual) forward input
    k.w = (const char*)w_bwd + (size_t)coff * 9 * d->Cout * es; k.Ctot = d->Cout; k.N = Csrc;
    k.bias = nullptr; k.relu = 0; k.out = (char*)dx; k.mask = (const char*)relu_mask;
    k.accumulate = accumulate; k.pool2 = up;
    if (d->stride == 1 && !up) {
        int r = try_launch_conv_rt(k, d->B, d->dtype, (hipStream_t)stream);
        if (r >= 0) return r;
        r = d->dtype == COLVO_F32 ? try_launch_conv_q<float>(k, d->B, (hipStream_t)stream)
                                  : try_launch_conv_q<bf16_t>(k, d->B, (hipStream_t)stream);
        if (r >= 0) return r;
    }
    return d->dtype == COLVO_F32 ? launch_conv_t<float>(k, d->B, up != 0, (hipStream_t)stream)
                                 : launch_conv_t<bf16_t>(k, d->B, up != 0, (hipStream_t)stream);
}

// Input gradient w.r.t. BOTH sources of a stride-1 concat layer in ONE launch: the two calls of colvo_conv_dgrad stage the
// same dy patches and differ only in the rows of w_bwd they read and the tensor they write; merged, the grid has the channel
// tiles of both sources (better fill at small batch) and half the launches.  Needs C0 to be a multiple of the channel tile
// (32) and direct (not up-sampled) sources; otherwise -- and for anything colvo_conv_dgrad would route to a special kernel --
// it falls back to two calls.
extern "C" int colvo_conv_dgrad_both(const ColvoConvDesc* d, const void* dy, const void* w_bwd, const void* relu_mask0,
                                     const void* relu_mask1, void* dx0, void* dx1, colvo_stream_t stream) {
    if (int e = check_desc(d, "colvo_conv_dgrad_both")) return e;
    COLVO_CHECK_ARG(dy && w_bwd && dx0 && dx1 && d->C1 > 0, "colvo_conv_dgrad_both: needs a two-source layer and both outputs");
    const int on = (int)TUNE(dgrad_both);
    const int es = d->dtype == COLVO_F32 ? 4 : 2;
    const long long cmax = std::max(d->C0, d->C1);
    const bool fits = (long long)d->Hi * d->Wi * cmax * es < 0x40000000LL && (long long)d->Ho * d->Wo * d->Cout * es < 0x40000000LL;
    if (!on || d->stride != 1 || d->up0 || d->up1 || d->C0 % 32 != 0 || !fits) {
        if (int e = colvo_conv_dgrad(d, 0, dy, w_bwd, relu_mask0, dx0, 0, stream)) return e;
        return colvo_conv_dgrad(d, 1, dy, w_bwd, relu_mask1, dx1, 0, stream);
    }
    ConvK k{};
    k.g.src[0] = (const char*)dy; k.g.src[1] = nullptr;
    k.g.C[0] = d->Cout; k.g.C[1] = 0;
    k.g.Hs[0] = d->Ho; k.g.Ws[0] = d->Wo; k.g.Hs[1] = k.g.Ws[1] = 0;
    k.g.mode[0] = k.g.mode[1] = MODE_DIRECT;
    k.g.Hi = d->Ho; k.g.Wi = d->Wo; k.g.stride = 1;
    k.Ho = d->Hi; k.Wo = d->Wi;
    k.w = (const char*)w_bwd; k.Ctot = d->Cout; k.N = d->C0 + d->C1;
    k.bias = nullptr; k.relu = 0; k.accumulate = 0; k.pool2 = 0;
    k.out = (char*)dx0; k.mask = (const char*)relu_mask0;
    k.out2 = (char*)dx1; k.mask2 = (const char*)relu_mask1; k.nsplit = d->C0;
    {
        const int r = try_launch_conv_rt(k, d->B, d->dtype, (hipStream_t)stream);
        if (r >= 0) return r;
    }
    return d->dtype == COLVO_F32 ? launch_conv_t<float>(k, d->B, false, (hipStream_t)stream)
                                 : launch_conv_t<bf16_t>(k, d->B, false, (hipStream_t)stream);
}

