// conv_stage.h -- staging helpers shared by the forward / input-gradient kernels of conv.hip and conv_rt.hip: epilogue load types,
// the logical workgroup id -> tile map, the weight-slab and patch stagers (global -> registers -> LDS).  Include after conv_common.h.
#pragma once
#include "conv_common.h"

namespace colvo {
namespace {

template <typename T> struct EV;
template <> struct EV<float> { typedef u32x4 type; };
template <> struct EV<bf16_t> { typedef u32x2 type; };
template <typename T>
__device__ __forceinline__ typename EV<T>::type epi_load(__amdgpu_buffer_rsrc_t r, int off, int soff) {
    if constexpr (TT<T>::ES == 4) return bld16(r, off, soff);
    else return __builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0);
}

// --------------------------------------------------------------------------------------------- //
// staging helpers shared by the fat-workgroup kernels (k_dgrad_s2, k_conv_up2, k_dgrad_up2, k_conv_q, k_conv_rt) //
// --------------------------------------------------------------------------------------------- //
// logical workgroup id (1-D grid, XCD-contiguous) -> image, tile row / column, first output channel; channel tile fastest
struct TileCoord { int b, ty, tx, n0; };
template <int BN>
__device__ __forceinline__ TileCoord tile_coord(const ConvK& a) {
    const int lid = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, a.xcd));
    const int tlin = lid / a.ntn;
    const int tpi = a.tiles_x * a.tiles_y;
    TileCoord c;
    c.n0 = (lid - tlin * a.ntn) * BN;
    c.b = tlin / tpi;
    const int trem = tlin - c.b * tpi;
    c.ty = trem / a.tiles_x;
    c.tx = trem - c.ty * a.tiles_x;
    return c;
}

// The [BN][9][CK] weight slab of one channel chunk: global -> registers -> LDS, exactly as in k_conv3x3 (one per-thread offset
// plus a scalar stride per staged granule; rows beyond N fall outside the descriptor and read as zero).
template <typename T, int BN, int NG>
struct SlabStage {
    static constexpr int G = TT<T>::G, ES = TT<T>::ES;
    static constexpr int CK = NG * G, NGR = 9 * NG, STEPS = (NGR + 3) / 4;
    static constexpr int WROW = wrow_bytes(STEPS * 4);
    static constexpr int WTOT = BN * NGR, WIT = (WTOT + NT - 1) / NT;
    int woff0, woffL, tapB;
    int wlds[WIT];
    __amdgpu_buffer_rsrc_t rw;

    __device__ __forceinline__ void init(const ConvK& a, int n0, int tid) {
        tapB = a.Ctot * ES;
        const int n = tid / NGR, gi = tid - n * NGR;
        const int tap = gi / NG, cg = gi - tap * NG;
        woff0 = ((n0 + n) * 9 + tap) * tapB + cg * 16;
        woffL = ((WIT - 1) * NT + tid < WTOT) ? woff0 : OOB_OFF;
        rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * 9 * a.Ctot * ES, 0x00020000);
    }
    // (kept apart from init(): the kernels issue their first loads before they compute what only the LDS side needs)
    __device__ __forceinline__ void lds_offsets(int tid) {
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = it * NT + tid;
            const int n = i / NGR;
            wlds[it] = i * 16 + n * (WROW - NGR * 16);
        }
    }
    __device__ __forceinline__ void load(int k, int dead, u32x4 (&w)[WIT]) const {
        const int so = dead ? 0 : k * CK * ES;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            w[it] = bld16(rw, (((it == WIT - 1) ? woffL : woff0) + it * (NT / NG) * tapB) | dead, so);
    }
    __device__ __forceinline__ void store(char* sW, int tid, const u32x4 (&w)[WIT]) const {
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            if (WTOT % NT == 0 || it < WIT - 1 || it * NT + tid < WTOT) st16(sW + wlds[it], w[it]);
    }
    __device__ __forceinline__ void zero_padding(char* sW, int tid) const {      // weight rows of 9 * NG granules, padded to 4 * STEPS
        if constexpr (STEPS * 4 != NGR) {
            for (int i = tid; i < BN * (STEPS * 4 - NGR); i += NT) {
                const int n = i / (STEPS * 4 - NGR), q = i - n * (STEPS * 4 - NGR);
                st16(sW + n * WROW + (NGR + q) * 16, u32x4{0u, 0u, 0u, 0u});
            }
        }
    }
};

// The input patch of one chunk from ONE directly stored source: patch pixel (py, px) = source pixel (y_org + py * 1, x_org + px),
// zero outside the source; LDS rows at the padded pitch a.pwp.
template <typename T, int NG, int PPF, bool S2 = false>      // S2: fragment rows two patch pixels apart (conv_common.h pitch_bytes_s2)
struct PatchStage {
    static constexpr int G = TT<T>::G, ES = TT<T>::ES, CK = NG * G;
    static constexpr int PIXP = S2 ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16);
    int poff[PPF], plds[PPF];
    int ptotal;
    __amdgpu_buffer_rsrc_t rimg;

    __device__ __forceinline__ void init(const ConvK& a, int b, int tid, int PH, int PW, int y_org, int x_org) {
        const int Hs = a.g.Hs[0], Ws = a.g.Ws[0], Cs = a.g.C[0];
        rimg = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g.src[0] + (size_t)b * Hs * Ws * Cs * ES), 0, Hs * Ws * Cs * ES,
                                                 0x00020000);
        ptotal = PH * PW * NG;
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int i = it * NT + tid;
            const int pix = i / NG, cg = i - pix * NG;
            const int py = mdiv(pix, a.m_pw), px = pix - py * PW;
            const int vy = y_org + py, vx = x_org + px;
            const bool inb = (i < ptotal) && ((unsigned)vy < (unsigned)Hs) && ((unsigned)vx < (unsigned)Ws);
            poff[it] = inb ? ((vy * Ws + vx) * Cs + cg * G) * ES : OOB_OFF;
            plds[it] = (py * a.pwp + px) * PIXP + cg * 16;
        }
    }
    __device__ __forceinline__ void load(int k, int dead, u32x4 (&pv)[PPF]) const {
        const int so = dead ? 0 : k * CK * ES;
#pragma unroll
        for (int it = 0; it < PPF; ++it) pv[it] = bld16(rimg, poff[it] | dead, so);
    }
    __device__ __forceinline__ void store(char* sP, int tid, const u32x4 (&pv)[PPF]) const {
#pragma unroll
        for (int it = 0; it < PPF; ++it)
            if (it * NT + tid < ptotal) st16(sP + plds[it], pv[it]);
    }
};

}  // namespace
}  // namespace colvo
