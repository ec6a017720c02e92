// wgrad_rt.hip -- a1/a2 of SURVEY.md §8: the register-tiled weight / bias gradient of the 3x3 conv blocks (round 6).
//
// Concept: /root/reference/README.md:5,7 (the DCDP depth and pose networks); results specified by oracle/colvo_spec.py (autograd of
// F.conv2d k=3 pad=1).  Same operands, same output layout ([Cout][9][Ctot] fp32, added to dw / db) and same ways out (fp32 atomics,
// per-split slabs, sole-writer read-modify-write, plain stores into a clean arena) as k_wgrad3x3 (wgrad.hip); bf16 only.
//
// Why a second decomposition (profiles/r5_conv_pmc_b64_final.json): k_wgrad3x3 gives the four waves of a workgroup
// the 9 x 32 (tap, ci) columns of one 32-channel chunk -- 2 x 5 accumulators of 16 x 16 per wave, K = 128-pixel tiles, 14 transposed
// LDS reads per 10 short MFMAs, every wave re-reading the dY fragments, staging through ~40 registers and two barriers per tile
// at ONE wave per SIMD: MFMA pipe 7-14 % busy.  And every workgroup ends with one fp32 atomic per element of its slab, which the
// memory side takes at ~1.3 TB/s chip-wide whatever the schedule (MI355X_MICROARCH.md, global float atomics): with 256-1024
// workgroups per layer that is 7-15 us of a 50 us launch.  Here:
//   * a WAVE owns a whole 32 (co) x 32 (ci) x 9 (taps) block of dw: nine v_mfma_f32_32x32x16_bf16 accumulators (144 registers), A = dY
//     (rows = co), B = x (columns = ci), K = 16 pixels of the tile.  Per K-step the wave reads ONE dY fragment and the nine shifted
//     x fragments (20 ds_read_b64_tr_b16) for nine 32-cycle MFMAs -- 2.2 reads per MFMA gap, inside the "<= 3 per gap" the LDS
//     feeds for free (MI355X_MICROARCH.md, LDS) -- and no fragment is read by two waves;
//   * the EIGHT waves of a workgroup (two per SIMD, 512 threads, <= 256 registers) split K: wave w takes the K-steps w, w + 8, ... of
//     every tile (rotated from tile to tile so that ragged step counts balance).  They meet in LDS at the very end (two passes of
//     18 accumulator chunks x 8 waves x 1 KiB; every wave then sums and flushes 4-5 chunks), so a whole CU ends with ONE 36 KB slab
//     of atomics: 256 workgroups = 9.4 MB per layer whatever the batch size;
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds, tools/ubench/lds_dma.hip holds the addressing rules): no staging registers,
//     no ds_write, two buffers and ONE barrier per tile -- the next tile lands while this one is computed.  The LDS images are
//     plain [pixel][32 channels] rows of 64 bytes: one DMA wave instruction = 16 pixels, and four consecutive pixels of a
//     transposed read cover all 64 banks once (the patch row pitch is congruent to the tile width modulo 4, so that holds
//     across tile rows too).  Waves 4-7 issue their share of the DMA after their first K-step, waves 0-3 in front of it: the two
//     waves of a SIMD do not stall the matrix pipe together;
//   * a K-step is ANY 16 consecutive tile pixels (row-major inside the tile, image after image for maps smaller than a tile),
//     so 8x10 ... 256x320 maps all run at full K efficiency.
// Up-sampled sources are read through (y >> 1, x >> 1) by the DMA's address arithmetic (the four-class kernel k_wgrad_up2 keeps the
// single-source up-sampled layers unless tuning says otherwise).  Stride 1 only.
// Built with -mllvm -amdgpu-mfma-vgpr-form (coivo_amd/build.py): at two waves per SIMD hipcc splits the 256 registers of a wave evenly
// between VGPRs and AGPRs as soon as a kernel uses AGPRs, and 144 accumulator registers do not fit into 128 -- as VGPRs they do.
#define COLVO_ACC_CONSTRAINT "+v"
#include "conv_common.h"

namespace colvo {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int WR_NT = 512, WR_WAVES = 8;
constexpr int WR_CONS = 4, WR_LOAD = 4;                       // consumer waves (0..3: accumulators, MFMAs) / loader waves (4..7: LDS-DMA)
constexpr int WR_MAXI = 14;                                   // DMA wave instructions per loader and tile (compile-time bound)
constexpr int WR_AHEAD = 4;                                   // x fragments requested this many MFMAs ahead of their use
constexpr int WR_EX_CHUNKS = 36;                              // accumulator chunks (4 registers x 64 lanes = 1 KiB) of a consumer
constexpr int WR_EX_BYTES = WR_EX_CHUNKS * WR_CONS * 1024;    // 147456
constexpr int WR_DB_BYTES = WR_NT * 8;                        // bias-gradient fold
constexpr int WR_LDS_MAX = 160 * 1024;
constexpr int WR_TAB_BYTES = 4096;                            // K-step address table: 4 bytes per tile pixel (<= 1024)

struct WgradRtK {
    const char* dy;           // [B][Ho][Wo][Cout] bf16
    const char* src[2];       // stored sources, bf16 NHWC
    float* dw;                // [Cout][9][Ctot]
    float* db;
    float* slabs;             // deterministic form: [nsplit][Cout * 9 * Ctot]; null: atomics / det
    float* db_slabs;
    int B, Ho, Wo, Cout, Ctot;
    int C[2], Hs[2], Ws[2], sh[2];     // per source: channels, stored extent, 1 = stored at half size (nearest-2x up-sampled)
    int nci0, nci, nco;                // 32-channel ci tiles of source 0 / of both sources, co tiles
    int ni, toh, tow;                  // tile = ni whole images (ni > 1: toh x tow is the image) or toh x tow pixels of one image
    int npix1, npix, ksteps;           // toh * tow, ni * npix1, ceil(npix / 16)
    int pwl, pimg, xinstr;             // LDS patch row pitch / image pitch in pixels, DMA instructions of the patch
    int tiles_x, tiles_y, ntiles, tiles_per_split, nsplit;
    uint32_t m_tow, m_npix1, m_pwl, m_pimg;
    int det;                           // as WgradK.det
    int xcd;
    int buf_bytes;                     // one staging buffer: (ksteps + xinstr) KiB
    int nbuf;                          // staging buffers in the ring (2..4): nbuf - 1 tiles are in flight while one is computed
    int tab_off;                       // LDS offset of the K-step address table (behind the ring)
#ifdef COLVO_WTRACE
    long long* trace;                  // developer build (tools/wtrace_wgrad.sh): [workgroup][wave][16] shader-clock sums per phase
#endif
};

// one LDS-DMA wave instruction: 64 lanes x 16 bytes from (descriptor, per-lane offset, scalar offset) to lds .. lds + 1023.  Lanes beyond
// the descriptor's range write zeros.  Inline asm on purpose: hipcc treats the builtin as an LDS store that every later ds_read may
// alias and drains vmcnt in front of the reads -- the pipeline below orders buffers by barriers instead.
#ifndef COLVO_RT_VARIANT
#define COLVO_RT_VARIANT 0          // developer ablations (tools/wtrace_wgrad.sh): 1 = no M0 save / restore, 2 = no DMA in the tile loop,
#endif                              // 4 = no MFMA phase, 8 = no bias sums, 16 = cache-hot DMA source
__device__ __forceinline__ void dma16(i32x4 rs, unsigned lds, int voff, int soff) {
#if COLVO_RT_VARIANT & 16          // every DMA reads the same cache-hot KiB: what the instruction itself costs
    voff = (int)(threadIdx.x & 63) * 16; soff = 0;
#endif
#if COLVO_RT_VARIANT & 1
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" : : "v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* p, long long bytes) {
    const unsigned long long a = (unsigned long long)p;
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    rs.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffff));
    rs.z = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffffLL ? bytes : 0x7fffffffLL));
    rs.w = __builtin_amdgcn_readfirstlane(0x00020000);
    return rs;
}
// a * b + c for 0 <= a, b < 2^24 (v_mad_u32_u24: full rate, where the 32-bit multiply takes four passes)
// (asm: given __umul24 of a value it knows to be small and a kernel argument it knows nothing about, hipcc falls back to v_mul_lo_u32)
__device__ __forceinline__ int mad24(int a, int b, int c) {
    int d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// i / d for 0 <= i < 1024, 2 <= d < 1024 with m = ceil(2^20 / d): one full-rate 24-bit multiply and a shift (every index divided in this
// kernel counts pixels of ONE tile or patch; the 32-bit mul_hi of conv_common.h mdiv takes four passes)
__device__ __forceinline__ int mdiv20(int i, uint32_t m) { return (int)(__umul24((unsigned)i, m) >> 20); }
inline uint32_t mdiv20_magic(int d) { return (uint32_t)(((1u << 20) + (uint32_t)d - 1) / (uint32_t)d); }
__device__ __forceinline__ s16x4 trd(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): all but the wave's n youngest vector-memory
// operations -- here: LDS-DMA instructions, the only ones in the tile loop -- are done
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
#define WVM_(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        WVM_(1) WVM_(2) WVM_(3) WVM_(4) WVM_(5) WVM_(6) WVM_(7) WVM_(8) WVM_(9) WVM_(10) WVM_(11) WVM_(12) WVM_(13) WVM_(14) WVM_(15)
        WVM_(16) WVM_(17) WVM_(18) WVM_(19) WVM_(20) WVM_(21) WVM_(22) WVM_(23) WVM_(24) WVM_(25) WVM_(26) WVM_(27) WVM_(28) WVM_(29) WVM_(30)
#undef WVM_
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// hand-over of the nine 32x32 accumulators to the VALU (see conv_common.h mfma_result_guard): one in-place terminator each
__device__ __forceinline__ void guard9(f32x16 (&acc)[9]) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %9, %9, %0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %1, %9, %9, %1\n\t"
                 "v_mfma_f32_32x32x16_bf16 %2, %9, %9, %2\n\t"
                 "v_mfma_f32_32x32x16_bf16 %3, %9, %9, %3\n\t"
                 "v_mfma_f32_32x32x16_bf16 %4, %9, %9, %4\n\t"
                 "v_mfma_f32_32x32x16_bf16 %5, %9, %9, %5\n\t"
                 "v_mfma_f32_32x32x16_bf16 %6, %9, %9, %6\n\t"
                 "v_mfma_f32_32x32x16_bf16 %7, %9, %9, %7\n\t"
                 "v_mfma_f32_32x32x16_bf16 %8, %9, %9, %8\n\t"
                 "s_nop 15\n\ts_nop 7"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]),
                   "+v"(acc[8])
                 : "v"(z));
}

#ifdef COLVO_WTRACE
#define WT(x) x = (long long)clock64()
#else
#define WT(x) do {} while (0)
#endif

__global__ __launch_bounds__(WR_NT) void k_wgrad_rt(const WgradRtK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef COLVO_WTRACE
    long long wt0, wt1, wt2, wt3, wt4, wt_start, wt_loop0, wt_loop1, wt_ex, wt_end;
    long long ws_vm = 0, ws_bar = 0, ws_issue = 0, ws_comp = 0;
    WT(wt_start);
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Roles: waves 0-3 (one per SIMD) are the CONSUMERS -- they own the accumulators and split K four ways --, waves 4-7 their SIMD
    // partners the LOADERS: all they do in the tile loop is issue LDS-DMA.  (First build, every wave both: a DMA instruction with its
    // address arithmetic cost the issuing wave ~350 cycles, six of them per wave and tile more than the wave's MFMA work, and with
    // all eight waves in that code at the same time the matrix pipes idled: 5600 cycles per tile for 1440 of MFMA, gpurun_out/r6c.)
    const bool loader = wave >= WR_CONS;
    const int lw = wave - WR_CONS;                         // loader index 0..3
    const int h = lane >> 5, c16 = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    // 1-D grid, XCD-contiguous: logical id = (pixel-range split, ci tile, co tile), co tile fastest
    const int lid = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, a.xcd));
    const int per_split = a.nco * a.nci;
    const int bsplit = lid / per_split, brem = lid - bsplit * per_split;
    const int bci = brem / a.nco, bco = brem - bci * a.nco;
    const int co0 = bco * 32;
    const int s = bci < a.nci0 ? 0 : 1;
    const int c0 = (s ? bci - a.nci0 : bci) * 32;
    const int wc0 = (s ? a.C[0] : 0) + c0;
    const int Cs = s ? a.C[1] : a.C[0], Hs = s ? a.Hs[1] : a.Hs[0], Ws = s ? a.Ws[1] : a.Ws[0], sh = s ? a.sh[1] : a.sh[0];
    const int Hi = a.Ho, Wi = a.Wo;                        // stride 1: the virtual input has the output's extent

    const i32x4 rdy = make_rsrc(a.dy, (long long)a.B * a.Ho * a.Wo * a.Cout * 2);
    const i32x4 rx = make_rsrc(s ? a.src[1] : a.src[0], (long long)a.B * Hs * Ws * Cs * 2);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int ninstr = a.ksteps + a.xinstr;
    const int nb = a.nbuf;
    const int t_begin = bsplit * a.tiles_per_split;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
    f32x16 acc[9];
    float dbacc = 0.0f;                                    // consumer lane (co = lane & 31, K half h): sum of its dY fragment elements
    // bias gradient: by the consumers of the workgroups of ci tile 0, from the dY fragments they hold anyway (v_dot2c_f32_bf16 with
    // a pair of ones: four instructions per K-step)
    const bool db_on = a.db != nullptr && bci == 0 && !(COLVO_RT_VARIANT & 8);

    // ---- ring of a.nbuf staging buffers: tile t is computed from buffer (t - t_begin) mod nbuf while the DMA of the next nbuf - 1 tiles
    // is in flight.  Every loader issues the same number of DMA instructions per tile, in tile order, so "my share of tile t has
    // landed" is a counted vmcnt; the workgroup barrier behind it makes that everyone's share, and says that tile t - 1's buffer is
    // free.  The two roles are two code paths with the same barrier sequence (their registers overlap: the loaders' instruction
    // table against the consumers' accumulators and fragments).
    {
        uint32_t* tabw = reinterpret_cast<uint32_t*>(smem + a.tab_off);
        for (int k = tid; k < a.ksteps * 16; k += WR_NT) {
            int kk = k < a.npix ? k : 0;                   // beyond the tile: dY is zero there, any finite x will do
            int img = 0, r = kk;
            if (a.ni > 1) { img = mdiv20(kk, a.m_npix1); r = kk - (int)__umul24(img, a.npix1); }
            const int oy = mdiv20(r, a.m_tow), ox = r - (int)__umul24(oy, a.tow);
            tabw[k] = (uint32_t)(img * a.pimg + oy * a.pwl + ox) * 64u;     // LDS patch pixel of tile pixel k at tap (0, 0), in bytes
        }
    }
    WT(wt_loop0);
    if (loader) {
        // ---- a loader's DMA instructions: ids lw, lw + 4, ...; id < ksteps: the dY pixels 16 id .. 16 id + 15 (K-step id), else the
        // patch pixels 16 j .. 16 j + 15, j = id - ksteps.  Lane = (pixel id >> 2, granule id & 3); destination = buffer + id KiB.
        const int gl = lane & 3, pl = lane >> 2;
        int pkc[WR_MAXI];                                  // (image << 24 | y << 12 | x) inside the tile / the patch; -1: never valid
#pragma unroll
        for (int n = 0; n < WR_MAXI; ++n) {
            const int id = lw + WR_LOAD * n;
            int pk = -1;
            if (id < a.ksteps) {
                const int k = 16 * id + pl;
                int img = 0, r = k;
                if (a.ni > 1) { img = mdiv20(k, a.m_npix1); r = k - (int)__umul24(img, a.npix1); }
                const int oy = mdiv20(r, a.m_tow), ox = r - (int)__umul24(oy, a.tow);
                if (k < a.npix && co0 + 8 * gl < a.Cout) pk = (img << 24) | (oy << 12) | ox;
            } else if (id < ninstr) {
                const int p = 16 * (id - a.ksteps) + pl;
                const int img = mdiv20(p, a.m_pimg), r = p - (int)__umul24(img, a.pimg);
                const int py = mdiv20(r, a.m_pwl), px = r - (int)__umul24(py, a.pwl);
                if (img < a.ni && py < a.toh + 2 && px < a.tow + 2 && c0 + 8 * gl < Cs) pk = (img << 24) | (py << 12) | px;
            }
            pkc[n] = pk;
        }
        struct TileC { int bg, ty, tx; };                  // image group, tile row, tile column (wave-uniform)
        auto tile_next = [&](TileC& c) {
            if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.bg; } }
        };
        // One tile's DMA.  Per instruction: unpack (image, y, x), test them against what is left of the image / batch, and three
        // 24-bit multiply-adds (full rate; every product stays below 2^30) for the byte offset; the tile's origin rides in the
        // scalar offset.
        const int gch_dy = (co0 + 8 * gl) * 2, gch_x = (c0 + 8 * gl) * 2;
        auto issue = [&](const TileC& c, int buf) {
            const int b0 = c.bg * a.ni, oy0 = c.ty * a.toh, ox0 = c.tx * a.tow;
            const int soff_dy = ((b0 * a.Ho + oy0) * a.Wo + ox0) * a.Cout * 2;
            const int soff_x = b0 * Hs * Ws * Cs * 2;
            const int ry = a.Ho - oy0, rx_ = a.Wo - ox0, rb = a.B - b0;  // rows / columns / images left from the tile's origin on
            const unsigned dst0 = lds0 + (unsigned)buf * (unsigned)a.buf_bytes;
#pragma unroll
            for (int n = 0; n < WR_MAXI; ++n) {
                const int id = lw + WR_LOAD * n;
                if (id >= ninstr) break;
                const int pk = pkc[n];
                const int i2 = pk >> 24, y12 = (pk >> 12) & 0xfff, x12 = pk & 0xfff;
                if (id < a.ksteps) {
                    const bool ok = pk >= 0 && y12 < ry && x12 < rx_ && i2 < rb;
                    const int pix = mad24(mad24(i2, a.Ho, y12), a.Wo, x12);          // relative to the origin
                    const int voff = ok ? mad24(pix, a.Cout * 2, gch_dy) : OOB_OFF;
                    dma16(rdy, dst0 + (unsigned)id * 1024u, voff, soff_dy);
                } else {
                    const int vy = oy0 - 1 + y12, vx = ox0 - 1 + x12;
                    const bool ok = pk >= 0 && (unsigned)vy < (unsigned)Hi && (unsigned)vx < (unsigned)Wi && i2 < rb;
                    const int pix = mad24(mad24(i2, Hs, vy >> sh), Ws, vx >> sh);
                    const int voff = ok ? mad24(pix, Cs * 2, gch_x) : OOB_OFF;
                    dma16(rx, dst0 + (unsigned)id * 1024u, voff, soff_x);
                }
            }
        };
        TileC nxt;                                         // the next tile to request
        {
            const int tpi = a.tiles_x * a.tiles_y;
            nxt.bg = t_begin / tpi;
            const int rem = t_begin - nxt.bg * tpi;
            nxt.ty = rem / a.tiles_x; nxt.tx = rem - nxt.ty * a.tiles_x;
        }
        const int cnt = lw < ninstr ? (ninstr - lw + WR_LOAD - 1) / WR_LOAD : 0;
        int t_issue = t_begin, ibuf = 0;
        for (int i = 0; i < nb - 1 && t_issue < t_end; ++i) { issue(nxt, ibuf); tile_next(nxt); ++t_issue; ++ibuf; }
        ibuf = nb - 1;                                     // the next request goes to tile t - 1's buffer
        for (int t = t_begin; t < t_end; ++t) {
            WT(wt0);
            wait_vm(cnt * (t_issue - 1 - t));              // this loader's share of tile t has landed
            WT(wt1);
            __syncthreads();
            WT(wt2);
            if (t_issue < t_end && !(COLVO_RT_VARIANT & 2)) {
                issue(nxt, ibuf);
                tile_next(nxt); ++t_issue; ibuf = ibuf + 1 == nb ? 0 : ibuf + 1;
            }
#ifdef COLVO_WTRACE
            WT(wt3);
            ws_vm += wt1 - wt0; ws_bar += wt2 - wt1; ws_issue += wt3 - wt2;
#endif
        }
    } else {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        // transposed-read lane constants: lane (h, c16, q, pp) of read j supplies the address of pixel 16 ks + 8 h + 4 j + q, channels
        // 16 c16 + 4 pp .. + 3, and receives channel (lane & 31) of the four pixels q = 0..3 (cdna_hip_programming.md T10)
        const int lane_dy = (8 * h + q) * 64 + 32 * c16 + 8 * pp;
        const int lane_x = 32 * c16 + 8 * pp;
        const int kl0 = 8 * h + q;
        const int rowb = a.pwl * 64;
        // Table of xidx * 64 for every tile pixel, behind the ring (loaders and consumers fill it; the tile loop's first barrier
        // publishes it): a K-step then costs two 4-byte LDS reads instead of ~25 VALU instructions of index arithmetic, which one
        // wave per SIMD cannot hide (second listing: 790 cycles per K-step for 288 of MFMA).
        const char* tab = smem + a.tab_off;
        const char* tabl = tab + kl0 * 4;                  // this lane's column of the table: entries 16 ks + kl0 and + 4
        int buf = 0, rot = 0;                              // buffer of tile t, (tiles walked * ksteps) mod 4
        for (int t = t_begin; t < t_end; ++t) {
            WT(wt1);
            __syncthreads();
            WT(wt2);
            const char* bufc = smem + buf * a.buf_bytes;
            const char* bufx = bufc + a.ksteps * 1024 + lane_x;
            // This consumer's K-steps ks, ks + 4, ... as ONE software pipeline over (step, tap): nine x-fragment registers, one per
            // tap; the fragment of the tap WR_AHEAD MFMAs ahead -- of this step or of the next one -- is requested before each MFMA,
            // the next step's dY fragment with its first tap, the next step's table entries at this step's first tap.  No branch
            // inside a step: behind the last step the "next" one is the step itself (reads nobody uses).  Pinned with sched_barrier:
            // left alone hipcc sinks every read to just in front of its MFMA (the first listing of this kernel).
            int ks = (wave - rot) & (WR_CONS - 1);
            bool have = ks < a.ksteps && !(COLVO_RT_VARIANT & 4);
            const char *pdy = bufc, *px0 = bufx, *px1 = bufx;
            s16x4 a_lo, a_hi, n_lo, n_hi, b_lo[9], b_hi[9];
            auto read_b = [&](const char* x0_, const char* x1_, int tap) {
                const int off = (tap / 3) * rowb + (tap % 3) * 64;
                b_lo[tap] = trd(x0_ + off); b_hi[tap] = trd(x1_ + off);
            };
            if (have) {
                const uint32_t t0 = *reinterpret_cast<const uint32_t*>(tabl + ks * 64), t1 = *reinterpret_cast<const uint32_t*>(tabl + ks * 64 + 16);
                pdy = bufc + ks * 1024 + lane_dy;
                px0 = bufx + t0; px1 = bufx + t1;
                a_lo = trd(pdy); a_hi = trd(pdy + 256);
#pragma unroll
                for (int tap = 0; tap < WR_AHEAD; ++tap) read_b(px0, px1, tap);
            }
            while (have) {
                const int ksn = ks + WR_CONS;
                const bool has_next = ksn < a.ksteps;
                const int kse = has_next ? ksn : ks;
                const char *ndy = pdy, *nx0 = px0, *nx1 = px1;
                uint32_t t0 = 0, t1 = 0;
                const s16x8 af = {a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
                if (db_on) {
                    // (the pairs are picked out of a bf16 vector: a bit_cast of a u32 ELEMENT to a bf16 pair makes hipcc 7.2 feed the
                    //  first dword to both instructions -- seen in this kernel's first listing, reproduced in isolation)
                    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_;
                    const bf16x4_ l4 = __builtin_bit_cast(bf16x4_, a_lo), h4 = __builtin_bit_cast(bf16x4_, a_hi);
                    const bf16x2_ ones = {(__bf16)1.0f, (__bf16)1.0f};
                    dbacc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_{l4[0], l4[1]}, ones, dbacc, false);
                    dbacc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_{l4[2], l4[3]}, ones, dbacc, false);
                    dbacc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_{h4[0], h4[1]}, ones, dbacc, false);
                    dbacc = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_{h4[2], h4[3]}, ones, dbacc, false);
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int tt = tap + WR_AHEAD;
                    if (tap == 0) {
                        t0 = *reinterpret_cast<const uint32_t*>(tabl + kse * 64); t1 = *reinterpret_cast<const uint32_t*>(tabl + kse * 64 + 16);
                    }
                    if (tt == 8) { ndy = bufc + kse * 1024 + lane_dy; nx0 = bufx + t0; nx1 = bufx + t1; }
                    if (tt < 9) read_b(px0, px1, tt);
                    else {
                        if (tt == 9) { n_lo = trd(ndy); n_hi = trd(ndy + 256); }
                        read_b(nx0, nx1, tt - 9);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const s16x8 bf = {b_lo[tap][0], b_lo[tap][1], b_lo[tap][2], b_lo[tap][3], b_hi[tap][0], b_hi[tap][1], b_hi[tap][2], b_hi[tap][3]};
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[tap], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                a_lo = n_lo; a_hi = n_hi;
                pdy = ndy; px0 = nx0; px1 = nx1;
                ks = ksn; have = has_next;
            }
            rot = (rot + a.ksteps) & (WR_CONS - 1);
            buf = buf + 1 == nb ? 0 : buf + 1;
#ifdef COLVO_WTRACE
            WT(wt4);
            ws_bar += wt2 - wt1; ws_comp += wt4 - wt2;
#endif
        }
        guard9(acc);
    }
    WT(wt_loop1);

    // ---- the four K slices meet: every consumer stores its 36 accumulator chunks (chunk c = tap c >> 2, registers 4 (c & 3) .. + 3:
    // rows co = (c & 3) * 8 + 4 h + e, column ci = lane & 31), lane-major 16-byte stores; then wave w -- loaders too -- sums the four
    // copies of the chunks w, w + 8, ... in a fixed order and flushes them: one register = two 128-byte row segments of dw
    f32x4* sEx = reinterpret_cast<f32x4*>(smem);
    const size_t wsize = (size_t)a.Cout * 9 * a.Ctot;
    const int ci = lane & 31;
    const bool civ = c0 + ci < Cs;
    __syncthreads();                                       // the staging buffers are free
    if (!loader) {
#pragma unroll
        for (int c = 0; c < WR_EX_CHUNKS; ++c) {
            const int tap = c >> 2, qd = c & 3;
            sEx[(c * WR_CONS + wave) * 64 + lane] = f32x4{acc[tap][4 * qd], acc[tap][4 * qd + 1], acc[tap][4 * qd + 2], acc[tap][4 * qd + 3]};
        }
    }
    __syncthreads();
    for (int c = wave; c < WR_EX_CHUNKS; c += WR_WAVES) {
        f32x4 v[WR_CONS];
#pragma unroll
        for (int w = 0; w < WR_CONS; ++w) v[w] = sEx[(c * WR_CONS + w) * 64 + lane];
        f32x4 t = v[0];
#pragma unroll
        for (int w = 1; w < WR_CONS; ++w) t += v[w];
        const int tap = c >> 2, qd = c & 3;
        const int corow = co0 + 8 * qd + 4 * h;
        const size_t e0 = ((size_t)corow * 9 + tap) * a.Ctot + wc0 + ci;
        const size_t erow = (size_t)9 * a.Ctot;
        const int nrow = civ ? min(4, a.Cout - corow) : 0;                 // rows of this lane inside the tensor (<= 0: none)
        if (a.slabs) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nrow) a.slabs[(size_t)bsplit * wsize + e0 + e * erow] = t[e];
        } else if (a.det == 2) {                           // one split, arena known to be zero: plain stores
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nrow) a.dw[e0 + e * erow] = t[e];
        } else if (a.det) {                                // one split = sole writer: plain read-modify-write, loads first
            float old[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) old[e] = e < nrow ? a.dw[e0 + e * erow] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nrow) a.dw[e0 + e * erow] = old[e] + t[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nrow) atomicAdd(a.dw + e0 + e * erow, t[e]);
        }
    }
    WT(wt_ex);
    if (a.db != nullptr && bci == 0) {                     // (wave-uniform: bci is the workgroup's)
        __syncthreads();
        float* sdb = reinterpret_cast<float*>(smem);
        if (!loader) sdb[tid] = dbacc;                     // [consumer][K half][co]
        __syncthreads();
        if (tid < 32 && co0 + tid < a.Cout) {
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < 2 * WR_CONS; ++j) t += sdb[j * 32 + tid];
            if (a.db_slabs) a.db_slabs[(size_t)bsplit * a.Cout + co0 + tid] = t;
            else if (a.det == 2) a.db[co0 + tid] = t;
            else if (a.det) a.db[co0 + tid] += t;
            else atomicAdd(a.db + co0 + tid, t);
        }
    }
#ifdef COLVO_WTRACE
    if (a.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the flush has left the wave
        WT(wt_end);
        if (lane == 0) {
            long long* r = a.trace + ((size_t)blockIdx.x * WR_WAVES + wave) * 16;
            r[0] = wt_loop0 - wt_start; r[1] = ws_vm; r[2] = ws_bar; r[3] = ws_issue; r[4] = ws_comp; r[5] = 0;
            r[6] = wt_ex - wt_loop1; r[7] = wt_end - wt_ex; r[8] = wt_end - wt_start; r[9] = t_end - t_begin; r[10] = wt_start;
        }
    }
#endif
}
#undef WT

struct RtTile { int ni, toh, tow, pwl, pimg, ksteps, xinstr; double cost; };

// LDS pitches of the patch: row pitch = tow (mod 4) and image pitch = toh * tow (mod 4), so that the LDS pixel index of tile pixel k is
// congruent to k modulo 4 -- the four consecutive pixels of a transposed read then sit on four different 64-byte bank quarters
inline void rt_pitches(int ni, int toh, int tow, int& pwl, int& pimg) {
    pwl = tow + 2;
    while ((pwl - tow) % 4) ++pwl;
    pimg = (toh + 2) * pwl;
    if (ni > 1) while ((pimg - toh * tow) % 4) ++pimg;
}

inline bool rt_tile_fits(const RtTile& t, int min_bufs) {
    const int instr = t.ksteps + t.xinstr;
    return instr <= WR_LOAD * WR_MAXI && min_bufs * instr * 1024 <= WR_LDS_MAX - WR_TAB_BYTES && t.toh + 2 < 4096 && t.tow + 2 < 4096 && t.ni < 128 &&
           t.ksteps * 16 <= 1024 && t.xinstr * 16 <= 1024 && t.tow + 2 < 1000 && t.pimg < 1024 && t.toh * t.tow < 1024;
}

// tile = the region that minimises  tiles x (MFMA cycles of its K-steps + DMA issue + per-tile fixed cost)
inline RtTile rt_pick_tile(int B, int Ho, int Wo) {
    const int max_px = std::max(16, (int)TUNE(wgrad_rt_max_px));
    const int min_bufs = std::min(4, std::max(2, (int)TUNE(wgrad_rt_min_bufs)));
    struct Key { int B, Ho, Wo, max_px, min_bufs; };
    static std::vector<std::pair<Key, RtTile>> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& e : cache)
        if (e.first.B == B && e.first.Ho == Ho && e.first.Wo == Wo && e.first.max_px == max_px && e.first.min_bufs == min_bufs) return e.second;
    RtTile best{}; best.cost = 1e300;
    auto consider = [&](int ni, int toh, int tow) {
        RtTile t{}; t.ni = ni; t.toh = toh; t.tow = tow;
        rt_pitches(ni, toh, tow, t.pwl, t.pimg);
        t.ksteps = (ni * toh * tow + 15) / 16;
        t.xinstr = (ni * t.pimg + 15) / 16;
        if (!rt_tile_fits(t, min_bufs)) return;
        const double tiles = ni > 1 ? (double)((B + ni - 1) / ni) : (double)B * ((Ho + toh - 1) / toh) * ((Wo + tow - 1) / tow);
        // per tile: nine 32-cycle MFMAs per K-step on four SIMDs, the DMA issue, a barrier and the pipeline's start -- or, if that is
        // longer, the tile's share of the load latency (~2 us under load) that the ring's depth leaves exposed
        const int nbuf = std::min(4, (WR_LDS_MAX - WR_TAB_BYTES) / ((t.ksteps + t.xinstr) * 1024));
        t.cost = tiles * std::max(72.0 * t.ksteps + 16.0 * (t.ksteps + t.xinstr) + 400.0, 5000.0 / (nbuf - 1));
        if (t.cost < best.cost) best = t;
    };
    for (int tow = 2; tow <= std::min(Wo, 160); ++tow) {
        if (tow != Wo && (tow & 3)) continue;              // (multiples of 4, or the whole row)
        for (int toh = 1; toh <= Ho; ++toh) {
            if (toh * tow > max_px) break;
            consider(1, toh, tow);
        }
    }
    if (Ho * Wo <= max_px / 2)
        for (int ni = 2; ni <= std::min(B, 64); ++ni) {
            if (ni * Ho * Wo > max_px) break;
            consider(ni, Ho, Wo);
        }
    cache.push_back({Key{B, Ho, Wo, max_px, min_bufs}, best});
    return best;
}

}  // namespace

bool wgrad_rt_plan(const ColvoConvDesc* d, WgradRtPlan& p) {
    if (!TUNE(wgrad_rt) || d->dtype != COLVO_BF16 || d->stride != 1 || d->Wo < 2 || d->Ho < 1) return false;
    const int minc = (int)TUNE(wgrad_rt_min_c);
    if (d->Cout < minc || d->C0 < minc || (d->C1 > 0 && d->C1 < minc)) return false;
    if ((long long)d->B * d->Ho * d->Wo < TUNE(wgrad_rt_min_px)) return false;
    const RtTile t = rt_pick_tile(d->B, d->Ho, d->Wo);
    if (t.cost >= 1e299) return false;
    p.ni = t.ni; p.toh = t.toh; p.tow = t.tow; p.pwl = t.pwl; p.pimg = t.pimg; p.ksteps = t.ksteps; p.xinstr = t.xinstr;
    p.tiles_x = t.ni > 1 ? 1 : (d->Wo + t.tow - 1) / t.tow;
    p.tiles_y = t.ni > 1 ? 1 : (d->Ho + t.toh - 1) / t.toh;
    p.ntiles = (t.ni > 1 ? (d->B + t.ni - 1) / t.ni : d->B) * p.tiles_x * p.tiles_y;
    p.nci0 = (d->C0 + 31) / 32;
    p.nci = p.nci0 + (d->C1 + 31) / 32;
    p.nco = (d->Cout + 31) / 32;
    const int per_split = p.nci * p.nco;
    int nsplit = std::max(1, (int)TUNE(wgrad_rt_wgs) / per_split);
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    p.tiles_per_split = (p.ntiles + nsplit - 1) / nsplit;
    p.nsplit = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
    return true;
}

int wgrad_rt_launch(const WgradRtPlan& p, const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw, float* db,
                    float* slabs, float* db_slabs, int det, hipStream_t s) {
    WgradRtK k{};
    k.dy = (const char*)dy; k.src[0] = (const char*)x0; k.src[1] = d->C1 ? (const char*)x1 : nullptr;
    k.dw = dw; k.db = db; k.slabs = slabs; k.db_slabs = db_slabs;
    k.B = d->B; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout; k.Ctot = d->C0 + d->C1;
    k.C[0] = d->C0; k.C[1] = d->C1;
    k.sh[0] = d->up0 ? 1 : 0; k.sh[1] = d->up1 ? 1 : 0;
    k.Hs[0] = d->up0 ? d->Hi / 2 : d->Hi; k.Ws[0] = d->up0 ? d->Wi / 2 : d->Wi;
    k.Hs[1] = d->up1 ? d->Hi / 2 : d->Hi; k.Ws[1] = d->up1 ? d->Wi / 2 : d->Wi;
    k.nci0 = p.nci0; k.nci = p.nci; k.nco = p.nco;
    k.ni = p.ni; k.toh = p.toh; k.tow = p.tow; k.npix1 = p.toh * p.tow; k.npix = p.ni * k.npix1; k.ksteps = p.ksteps;
    k.pwl = p.pwl; k.pimg = p.pimg; k.xinstr = p.xinstr;
    k.tiles_x = p.tiles_x; k.tiles_y = p.tiles_y; k.ntiles = p.ntiles; k.tiles_per_split = p.tiles_per_split; k.nsplit = p.nsplit;
    k.m_tow = mdiv20_magic(p.tow); k.m_npix1 = mdiv20_magic(std::max(2, k.npix1)); k.m_pwl = mdiv20_magic(p.pwl); k.m_pimg = mdiv20_magic(p.pimg);
    k.det = det; k.xcd = (int)TUNE(xcd_remap);
    k.buf_bytes = (p.ksteps + p.xinstr) * 1024;
    k.nbuf = std::min(4, (WR_LDS_MAX - WR_TAB_BYTES) / k.buf_bytes);
    k.tab_off = k.nbuf * k.buf_bytes;
    COLVO_CHECK_ARG(p.ksteps + p.xinstr <= WR_LOAD * WR_MAXI && p.tow >= 2, "wgrad_rt: bad tile plan");
    // always the whole exchange area: one workgroup per CU (eight waves at <= 256 registers fill its SIMDs two deep)
    COLVO_CHECK_ARG(k.nbuf >= 2, "wgrad_rt: a tile of %d KiB does not fit twice", k.buf_bytes / 1024);
    const size_t lds = std::max<size_t>((size_t)k.nbuf * k.buf_bytes + WR_TAB_BYTES, (size_t)WR_EX_BYTES + WR_DB_BYTES);
    COLVO_CHECK_ARG(lds <= (size_t)WR_LDS_MAX, "wgrad_rt: %zu bytes of LDS", lds);
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_rt), hipFuncAttributeMaxDynamicSharedMemorySize, WR_LDS_MAX);
        if (e != hipSuccess) { set_error("wgrad_rt: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = true;
    }
    const unsigned nwg = (unsigned)(p.nsplit * p.nci * p.nco);
#ifdef COLVO_WTRACE
    static long long* g_tr = nullptr;
    static int g_calls = 0;
    k.trace = nullptr;
    if (getenv("COLVO_WTRACE") && nwg <= 4096) {
        if (!g_tr) (void)hipMalloc(&g_tr, (size_t)4096 * WR_WAVES * 16 * sizeof(long long));
        (void)hipMemsetAsync(g_tr, 0, (size_t)nwg * WR_WAVES * 16 * sizeof(long long), s);
        k.trace = g_tr;
    }
#endif
    colvo::launch(k_wgrad_rt, dim3(nwg), dim3(WR_NT), (unsigned)lds, s, k);
    COLVO_CHECK_LAUNCH("k_wgrad_rt");
#ifdef COLVO_WTRACE
    if (k.trace && (++g_calls % atoi(getenv("COLVO_WTRACE"))) == 0) {
        (void)hipStreamSynchronize(s);
        std::vector<long long> h((size_t)nwg * WR_WAVES * 16);
        (void)hipMemcpy(h.data(), g_tr, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double m[2][16] = {{0}};                 // waves 0-3 | waves 4-7
        long long t0 = h[10], t1 = 0;
        for (unsigned i = 0; i < nwg * WR_WAVES; ++i) {
            const int half = (i % WR_WAVES) >= 4;
            for (int j = 0; j < 10; ++j) m[half][j] += (double)h[(size_t)i * 16 + j] / (nwg * 4);
            t0 = std::min(t0, h[(size_t)i * 16 + 10]); t1 = std::max(t1, h[(size_t)i * 16 + 10] + h[(size_t)i * 16 + 8]);
        }
        for (int half = 0; half < 2; ++half) {
            const double nt = std::max(1.0, m[half][9]);
            fprintf(stderr, "[wtrace rt] Cout=%d C=%d+%d %dx%d B=%d tile %dx%dx%d ks=%d xi=%d nbuf=%d grid=%u nsplit=%d tiles/wg %.1f | waves %s (cycles): life %.0f = "
                    "setup %.0f + loop %.0f + exchange %.0f + flush %.0f | per tile: vmwait %.0f barrier %.0f issue %.0f compute %.0f db %.0f | span %lld\n",
                    k.Cout, k.C[0], k.C[1], k.Ho, k.Wo, k.B, k.ni, k.toh, k.tow, k.ksteps, k.xinstr, k.nbuf, nwg, k.nsplit, nt, half ? "4-7" : "0-3",
                    m[half][8], m[half][0], m[half][1] + m[half][2] + m[half][3] + m[half][4] + m[half][5], m[half][6], m[half][7],
                    m[half][1] / nt, m[half][2] / nt, m[half][3] / nt, m[half][4] / nt, m[half][5] / nt, t1 - t0);
        }
    }
#endif
    return 0;
}

}  // namespace colvo
