// wgrad.hip -- weight / bias gradient of the 3x3 conv blocks (a1/a2 of SURVEY.md §8), see conv.hip for the layout.
// Its own translation unit: built WITHOUT -mllvm -amdgpu-mfma-vgpr-form -- with the accumulators in VGPRs the MFMA-dense
// pixel loop ran 16 % slower (k_wgrad3x3<bf16,2,4> 29.5 -> 34.3 us, gpurun_out/r2_bench_conv_*.log) -- so hipcc keeps
// them in AGPRs and mfma_result_guard ties them with "+a" (tests/test_isa_cpu.py checks the emitted ISA of both units).
#define COLVO_ACC_CONSTRAINT "+a"
#include "conv_common.h"

namespace colvo {
namespace {

// --------------------------------------------------------------------------------------------- //
// weight-gradient kernel                                                                         //
// --------------------------------------------------------------------------------------------- //
// LDS pitch of a dY row.  bf16: the transposed reads (ds_read_b64_tr_b16: two 32-lane groups, 8 bytes per lane, 64 banks) of
// one instruction touch 8 pixels x 32 bytes; they are conflict-free iff the 8 pixel rows start on 8 different 32-byte bank
// octets, i.e. the pitch is an ODD multiple of 32 bytes AND the 8 pixels are consecutive (see the pixel order in the kernel).
// With the former pitch of 80 bytes and pixel order every such read took two passes (profiles/r2_conv_pmc.json: 43-48 % of the
// weight-gradient kernels' LDS cycles were bank conflicts); removing them was worth 2 % of the kernels' time.
// (Also measured this round and dropped, all parity-green: requesting ALL fragments of k-step ks+1 before the MFMAs of step ks
// (two register sets; hipcc alone keeps one fragment pair of look-ahead): 534 -> 537 us over the DepthNet stack, no change;
// LDS-DMA staging (`buffer_load_dwordx4 ... offen lds` into 2 / 3 rotating stage buffers, one raw barrier per tile, counted
// vmcnt -- tools/ubench/lds_dma.hip holds the hardware check of the addressing rules): 528 -> 558 / 563 us, every stride-1 layer
// ~10 % slower (the padded LDS image costs 8 DMA instructions per thread and tile instead of 5 loads, and the third buffer
// bought nothing: the tile loop is not waiting for memory).  A compile-time phase ablation says the same: without the global
// loads -17 %, without fragment reads + MFMAs -25 %, without the LDS stores (and the then dead loads) -26 %, everything off
// still 45 % (launch, set-up, the atomics of the flush).  Per tile the loop moves ~134 KB through LDS (114 KB of transposed
// fragment reads) = 0.45 us at 128 B/clk against 0.27 us of MFMA time and ~0.95 us measured.
// Also: a two-tile register ring for the staging loads -- 12 % SLOWER, 208 registers;
// 256-pixel tiles for the multi-chunk layers -- 2 % faster alone (534 -> 524 us), 0.7 % SLOWER inside the training step,
// where the weight-gradient kernels share the CUs with the input-gradient chain.)
template <typename T, int MT>
constexpr int dy_pitch() {
#ifdef COLVO_WGRAD_OLD_LAYOUT                     // developer A/B build only
    return 16 * MT * TT<T>::ES + 16;
#else
    if (TT<T>::ES == 4) return 16 * MT * 4 + 16;
    return MT == 1 ? 32 : MT == 2 ? 96 : 160;
#endif
}

// TAIL: see k_conv3x3 (the stride-2 patch tail; its dead loads made hipcc drain the prefetch in front of the MFMA phase)
// KS: pixel-tile teams per workgroup.  The number of workgroups is capped by the fp32 atomics every one of them ends with
// (one per weight element of its (co tile, channel chunk) slab), and at that cap most layers ran ONE 4-wave workgroup per
// CU: one wave per SIMD, nothing to hide the staging latency or the LDS round trips behind (~1.4 us per 128-pixel tile,
// MFMA pipe 7-14 % busy).  A workgroup of KS teams of 4 waves walks KS pixel tiles at a time -- each team its own staging
// buffers, same barriers -- and the teams' accumulators meet in LDS (ds_add_f32) before ONE set of global atomics leaves
// the workgroup, coalesced along the weight rows: KS times the waves per CU at the same atomic traffic.
template <typename T, int MT, int NG, bool TAIL, int KS = 1>
__global__ __launch_bounds__(NT * KS) void k_wgrad3x3(const WgradK a) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int CK = NG * G;
    constexpr int NCOL = 9 * CK;                   // (tap, c) columns of this chunk
    constexpr int NFR = (NCOL + 15) / 16;          // column fragments
    constexpr int FPW = (NFR + 3) / 4;             // fragments per wave
    // patch-pixel pitch: TAIL instantiations are the stride-2 layers (a stride-1 patch of <= 128 outputs never exceeds 3 x 256
    // granules), where consecutive K positions are TWO patch pixels apart: conv_common.h pitch_bytes_s2
    constexpr int PIXP = TAIL ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16);
    constexpr int DYP = dy_pitch<T, MT>();         // dY row pitch (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef COLVO_WTRACE
    // developer build: shader-clock stamps taken right behind barriers only (a stamp is a scalar memory read: anywhere else its
    // lgkmcnt wait would drain the LDS reads it is meant to observe)
    long long wt_start = (long long)clock64(), wt_first = 0, wt_store = 0, wt_compute = 0, wt_b1 = 0, wt_b2 = 0, wt_loop_end = 0, wt_flush = 0;
    long long wt_issue = 0, wt_mfma = 0, wt_m0 = 0, wt_m1 = 0;        // inside `compute`: load issue (+ bias sums) | MFMA phase | the rest = barrier wait
    long long wt_wall0 = (long long)wall_clock64();
#define WT_STAMP(x) x = (long long)clock64()
#else
#define WT_STAMP(x) do {} while (0)
#endif
    const int tid = threadIdx.x & (NT - 1), lane = tid & 63, wave = tid >> 6;       // inside the team
    const int team = (KS > 1) ? __builtin_amdgcn_readfirstlane(threadIdx.x / NT) : 0;
    const int l15 = lane & 15, kg = lane >> 4;
    // 1-D grid, XCD-contiguous: logical id = (pixel-range split, channel chunk, co tile), co tile fastest -- the
    // workgroups of one pixel range (same dY tiles, same patches) sit on one L2
    const int lid = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, a.xcd));
    const int per_split = gridDim.x / a.nsplit;
    const int bsplit = lid / per_split, brem = lid - bsplit * per_split;
    const int bchunk = brem / a.cot, bco = brem - bchunk * a.cot;
    const int co0 = bco * 16 * MT;
    // chunk -> (source, channel offset)
    const int chunks0 = a.g.C[0] / CK;
    const int s = (bchunk < chunks0) ? 0 : 1;
    const int c0 = (s == 0 ? bchunk : bchunk - chunks0) * CK;
    const int wc0 = (s == 0 ? 0 : a.g.C[0]) + c0;
    const int S = a.g.stride;
    const int PH = (a.toh - 1) * S + 3, PW = (a.tow - 1) * S + 3;
    const int npix = a.toh * a.tow;
    const int PWL = a.pwl;                          // LDS pitch of a patch row in pixels (>= PW: conv_common.h wgrad_row_pitch)
    const int stage = (BM * DYP + PH * PWL * PIXP + 15) & ~15;    // one team's staging buffers
    char* sDY = smem + team * stage;               // [BM][16*MT]
    char* sX = sDY + BM * DYP;                     // patch

    // per-lane column decode of the owned fragments
    int boff[FPW];      // LDS byte offset of this lane's (tap, c) inside a patch pixel row-set
    int ocol[FPW];      // element offset tap*Ctot + c of this lane's OUTPUT column (-1: padding)
#pragma unroll
    for (int fi = 0; fi < FPW; ++fi) {
        const int f = wave + 4 * fi;
        // address column (tr read: lane supplies the address of columns 4p..4p+3; plain: own column)
        const int ncol_addr = 16 * f + ((ES == 2) ? 4 * (lane & 3) : l15);
        int tap = min(ncol_addr / CK, 8);
        const int c = ncol_addr - (ncol_addr / CK) * CK;
        boff[fi] = ((tap / 3) * PWL + (tap % 3)) * PIXP + c * ES;
        const int ncol = 16 * f + l15;
        ocol[fi] = (f < NFR && ncol < NCOL) ? ((ncol / CK) * a.Ctot + (ncol % CK)) : -1;
    }

    f32x4 acc[MT][FPW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int fi = 0; fi < FPW; ++fi) acc[mi][fi] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.0f;

    const int t_begin = bsplit * a.tiles_per_split;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
    const int tiles_per_img = a.tiles_x * a.tiles_y;

    // software pipeline over pixel tiles: the dY tile and the input patch of tile t+1 are loaded into registers
    // before the MFMAs of tile t and written to LDS after them
    constexpr int DGR = 16 * MT / G;                       // dY granules per pixel row
    constexpr int DIT = (BM * DGR + NT - 1) / NT;
    constexpr int PPF = TAIL ? 9 : 3;              // stride-2 patches: everything in the prefetch
    const int ptotal = PH * PW * NG;
    u32x4 dyv[DIT], pv[PPF];

    // Tile coordinates (image, tile row, tile column) are wave-uniform and advance incrementally; everything that
    // depends on the thread only (its pixel / granule inside the tile or patch) is computed ONCE, with magic-number
    // divisions.  Per tile a thread then needs a handful of adds and compares per staged granule -- this loop was
    // VALU-bound on ~20 runtime integer divisions per tile.
    struct TileC { int b, ty, tx; };
    auto tile_next = [&](TileC& c) {
        if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.b; } }
    };
    TileC cur;
    {
        const int t = __builtin_amdgcn_readfirstlane(t_begin + team);
        cur.b = t / tiles_per_img;
        const int tr_ = t - cur.b * tiles_per_img;
        cur.ty = tr_ / a.tiles_x; cur.tx = tr_ - cur.ty * a.tiles_x;
    }
    // branch-free staging loads: buffer descriptors over the whole dY tensor / the whole source tensor, invalid
    // granules get an out-of-range offset and come back as zeros (see k_conv3x3)
    const long long dy_bytes = (long long)a.B * a.Ho * a.Wo * a.Cout * ES;
    const long long x_bytes = (long long)a.B * a.g.Hs[s] * a.g.Ws[s] * a.g.C[s] * ES;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)(dy_bytes < 0x7fffffffLL ? dy_bytes : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.g.src[s], 0, (int)(x_bytes < 0x7fffffffLL ? x_bytes : 0x7fffffffLL), 0x00020000);
    const int Hs = a.g.Hs[s], Ws = a.g.Ws[s], Cs = a.g.C[s];
    const int sh = (a.g.mode[s] != MODE_DIRECT) ? 1 : 0;   // up-sampled source: stored at half size

    int dy_off[DIT], dy_yx[DIT], dy_lds[DIT];             // offset inside the tile's image region, (oy << 16 | ox), LDS
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
        const int i = it * NT + tid;
        const int pp_ = i / DGR, gch = i - pp_ * DGR;
        const int oy = mdiv(pp_, a.m_tow), ox = pp_ - oy * a.tow;
        const bool ok = (BM * DGR % NT == 0 || i < BM * DGR) && (pp_ < npix) && (co0 + gch * G < a.Cout);
        dy_off[it] = ok ? ((oy * a.Wo + ox) * a.Cout + co0 + gch * G) * ES : OOB_OFF;
        dy_yx[it] = (oy << 16) | ox;
        dy_lds[it] = pp_ * DYP + gch * 16;
    }
    int p_yx[PPF], p_cg[PPF], p_lds[PPF];                  // (py << 16 | px) inside the patch (py = 0x7fff: none), granule, LDS offset
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        const int i = it * NT + tid;
        const int pix = i / NG;
        p_cg[it] = i - pix * NG;
        const int py = mdiv(pix, a.m_pw), px = pix - py * PW;
        p_yx[it] = (i < ptotal) ? ((py << 16) | px) : (0x7fff << 16);
        p_lds[it] = (py * PWL + px) * PIXP + p_cg[it] * 16;
    }
    // `live`: false for a team whose tile index ran past the workgroup's range (it still takes part in the barriers; its
    // loads are out of range and stage zeros)
    auto load_dy = [&](const TileC& c, bool live) {
        const int oy0 = c.ty * a.toh, ox0 = c.tx * a.tow;
        const int base = live ? ((c.b * a.Ho + oy0) * a.Wo + ox0) * a.Cout * ES : 0;      // wave-uniform: the scalar offset
        const int remy = live ? a.Ho - oy0 : 0, remx = a.Wo - ox0;
        if (remy >= a.toh && remx >= a.tow) {              // interior tile: no per-granule test
#pragma unroll
            for (int it = 0; it < DIT; ++it) dyv[it] = bld16(rdy, dy_off[it], base);
        } else {
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const bool ok = ((dy_yx[it] >> 16) < remy) && ((dy_yx[it] & 0xffff) < remx);
                dyv[it] = bld16(rdy, ok ? dy_off[it] : OOB_OFF, base);
            }
        }
    };
    auto store_dy = [&]() {
#pragma unroll
        for (int it = 0; it < DIT; ++it)
            if (BM * DGR % NT == 0 || it * NT + tid < BM * DGR) st16(sDY + dy_lds[it], dyv[it]);
    };
    auto patch_voff = [&](const TileC& c, int yx, int cg) -> int {
        const int vy = c.ty * a.toh * S - 1 + (yx >> 16), vx = c.tx * a.tow * S - 1 + (yx & 0xffff);
        const bool inb = ((unsigned)vy < (unsigned)a.g.Hi) && ((unsigned)vx < (unsigned)a.g.Wi);
        return inb ? (((vy >> sh) * Ws + (vx >> sh)) * Cs + c0 + cg * G) * ES : OOB_OFF;
    };
    auto load_p = [&](const TileC& c, bool live) {
        const int base = live ? c.b * Hs * Ws * Cs * ES : 0;
        const int dead = live ? 0 : OOB_OFF;          // offsets are < OOB_OFF: OR-ing it in puts them out of range
#pragma unroll
        for (int it = 0; it < PPF; ++it) pv[it] = bld16(rx, patch_voff(c, p_yx[it], p_cg[it]) | dead, base);
    };
    auto store_p = [&](const TileC& c, bool live) {
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int i = it * NT + tid;
            if (i < ptotal) st16(sX + p_lds[it], pv[it]);
        }
        // patches larger than PPF*256 granules (stride-2 tiles): the rest is staged in place, 3 loads in flight
        const int base = live ? c.b * Hs * Ws * Cs * ES : 0;
        const int dead = live ? 0 : OOB_OFF;
        if constexpr (TAIL)
        for (int g0 = PPF * NT; g0 < ptotal; g0 += 3 * NT) {
            u32x4 tt[3];
            int tl[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = g0 + u * NT + tid;
                const int pix = i / NG, cg = i - pix * NG;
                const int py = mdiv(pix, a.m_pw), px = pix - py * PW;
                tt[u] = bld16(rx, ((i < ptotal) ? patch_voff(c, (py << 16) | px, cg) : OOB_OFF) | dead, base);
                tl[u] = (py * PWL + px) * PIXP + cg * 16;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (g0 + u * NT + tid < ptotal) st16(sX + tl[u], tt[u]);
        }
    };
    // bias gradient: all 256 threads, thread = (channel, pixel phase); NPH partial sums per channel meet in the atomics
    constexpr int NPH = NT / (16 * MT);
    const int db_co = tid % (16 * MT), db_ph = tid / (16 * MT);

    // team g walks the tiles t_begin + g, t_begin + g + KS, ...; every team runs the same number of iterations
    if (t_begin < t_end) { const bool live = t_begin + team < t_end; load_dy(cur, live); load_p(cur, live); }
    for (int t = t_begin; t < t_end; t += KS) {
        __syncthreads();
#ifdef COLVO_WTRACE
        WT_STAMP(wt_b1);
        if (t == t_begin) wt_first = wt_b1; else wt_compute += wt_b1 - wt_b2;
#endif
        store_dy();
        store_p(cur, t + team < t_end);
        __syncthreads();
#ifdef COLVO_WTRACE
        WT_STAMP(wt_b2);
        wt_store += wt_b2 - wt_b1;
#endif
#pragma unroll
        for (int q = 0; q < KS; ++q) tile_next(cur);
        if (t + KS < t_end) { const bool live = t + KS + team < t_end; load_dy(cur, live); load_p(cur, live); }   // in flight during the MFMAs below

        if (bchunk == 0 && a.db) {
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 4
            for (int p = db_ph; p < BM; p += 2 * NPH) {   // rows beyond the tile hold zeros
                if constexpr (ES == 2) {
                    s0 += bf2f(*reinterpret_cast<const uint16_t*>(sDY + p * DYP + db_co * 2));
                    s1 += bf2f(*reinterpret_cast<const uint16_t*>(sDY + (p + NPH) * DYP + db_co * 2));
                } else {
                    s0 += *reinterpret_cast<const float*>(sDY + p * DYP + db_co * 4);
                    s1 += *reinterpret_cast<const float*>(sDY + (p + NPH) * DYP + db_co * 4);
                }
            }
            dbacc += s0 + s1;
        }

#ifdef COLVO_WTRACE
        WT_STAMP(wt_m0);
        wt_issue += wt_m0 - wt_b2;
#endif
        if constexpr (ES == 2) {
            // K = 128 pixels in 4 steps of 32; fragments by hardware-transposed LDS reads.
            // (Round 4, measured and not kept: ALL fragments of step ks+1 requested into a second register set before the MFMAs of
            // step ks, pinned with scheduling barriers -- hipcc alone keeps one fragment pair of look-ahead.  The listing shows the
            // pipeline (28 reads, wait, 10 MFMAs, 14 reads, ...), the in-kernel stamps show no gain: 0.83-0.96 us per 128-pixel tile
            // before, 0.95-1.07 with the form held to 168 registers (3 spills), the stack's total 513 -> 508 us without that bound at
            // 176 registers = two instead of three waves per SIMD.  The tile's 1900 cycles are not LDS latency: 640 are MFMA issue,
            // ~500 the VALU address arithmetic between them (4 cycles an instruction), the rest barrier skew and the first k-step's
            // round trip.  The four-class kernel below, with 16 MFMAs per step, does gain from it and keeps it.)
            const int q = l15 >> 2, pp = lane & 3;
            auto read_step = [&](int ks, s16x8 (&af)[MT], s16x8 (&bfv)[FPW]) {
                int xo[2], yo[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // K position (ks, kg, h, q) -> pixel: any bijection works as long as dY and X use the same one; this one
                    // gives the 32 lanes of a read group 8 CONSECUTIVE pixels (kg & 1, q), see dy_pitch()
#ifdef COLVO_WGRAD_OLD_LAYOUT
                    int p = 32 * ks + 8 * kg + q + 4 * h;
#else
                    int p = 32 * ks + 16 * (kg >> 1) + 8 * h + 4 * (kg & 1) + q;
#endif
                    yo[h] = p * DYP;
                    if (p >= npix) p = 0;     // its dY row is zero
                    const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
                    xo[h] = ((oy * S) * PWL + ox * S) * PIXP;
                }
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(sDY + yo[0] + (16 * mi + 4 * pp) * 2));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(sDY + yo[1] + (16 * mi + 4 * pp) * 2));
                    af[mi] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int fi = 0; fi < FPW; ++fi) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(sX + xo[0] + boff[fi]));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(sX + xo[1] + boff[fi]));
                    bfv[fi] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
            };
            s16x8 af[MT], bfv[FPW];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                read_step(ks, af, bfv);
#pragma unroll
                for (int fi = 0; fi < FPW; ++fi)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        acc[mi][fi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, af[mi]), __builtin_bit_cast(bf16x8, bfv[fi]), acc[mi][fi], 0, 0, 0);
            }
        } else {
            // K = 128 pixels in 32 steps of 4 (exact f32 MFMA 16x16x4); lane k-slot = kg
            int p = kg;
            int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
            for (int ks = 0; ks < BM / 4; ++ks) {
                const bool live = p < npix;
                const int xo = live ? ((oy * S) * PWL + ox * S) * PIXP : 0;
                float av[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    av[mi] = *reinterpret_cast<const float*>(sDY + p * DYP + (16 * mi + l15) * 4);
#pragma unroll
                for (int fi = 0; fi < FPW; ++fi) {
                    const float bvv = *reinterpret_cast<const float*>(sX + xo + boff[fi]);
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        acc[mi][fi] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bvv, acc[mi][fi], 0, 0, 0);
                }
                p += 4; ox += 4;
                while (ox >= a.tow) { ox -= a.tow; ++oy; }
            }
        }
#ifdef COLVO_WTRACE
        asm volatile("" ::: "memory");
        WT_STAMP(wt_m1);
        wt_mfma += wt_m1 - wt_m0;
#endif
    }

    mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[MT * FPW]>(acc));
#ifdef COLVO_WTRACE
    __syncthreads();
    WT_STAMP(wt_loop_end);
    wt_compute += wt_loop_end - wt_b2;
#endif
    float* sdb = reinterpret_cast<float*>(smem);
    if constexpr (KS > 1) {
        // The teams' accumulators meet in team 0 through LDS, one team per round with plain 16-byte stores (ds_add_f32
        // measured ~170 cycles per wave instruction: 46 us for this exchange).  Then ONE set of global atomics per workgroup.
        f32x4* sRed = reinterpret_cast<f32x4*>(smem);
        sdb = reinterpret_cast<float*>(smem + MT * FPW * NT * 16);
        for (int g = 1; g < KS; ++g) {
            __syncthreads();                               // staging buffers / the previous round's slab are free
            if (team == g) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int fi = 0; fi < FPW; ++fi) sRed[(mi * FPW + fi) * NT + tid] = acc[mi][fi];
            }
            __syncthreads();
            if (team == 0) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int fi = 0; fi < FPW; ++fi) acc[mi][fi] += sRed[(mi * FPW + fi) * NT + tid];
            }
        }
    }
    if (team == 0) {
        // D rows = co (4*kg + r), cols = (tap, c).  Three ways out: one fp32 atomic per element (a.det == 0); deterministic form with
        // several splits: this split's own slab; deterministic form with ONE split = this workgroup is the only writer of its
        // elements: plain read-modify-write, all loads in flight before the first store (40 dependent load -> store round trips
        // otherwise) -- no slab, no second launch.  (Against the atomics the read-modify-write measured 1 us SLOWER, enc5b 25.6 ->
        // 26.7 us: the loads miss the XCD's L2, the atomics are fire-and-forget; so it serves the deterministic form only.)
        auto elem = [&](int mi, int fi, int r, size_t& e) -> bool {
            const int co = co0 + 16 * mi + 4 * kg + r;
            e = (size_t)co * 9 * a.Ctot + wc0 + ocol[fi];
            return ocol[fi] >= 0 && co < a.Cout;
        };
        if (!a.slabs && a.det == 2) {
            // ONE split and a gradient arena known to be zero (colvo_conv_wgrad_clean): this workgroup is the first and only writer
            // of its elements in this step -- plain stores, neither atomics (6-7.7 us of a workgroup's life, profiles/r4_wgrad_phases.md)
            // nor the loads of the read-modify-write
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int fi = 0; fi < FPW; ++fi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { size_t e; if (elem(mi, fi, r, e)) a.dw[e] = acc[mi][fi][r]; }
        } else if (!a.slabs && a.det) {
            float old[MT][FPW][4];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int fi = 0; fi < FPW; ++fi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { size_t e; old[mi][fi][r] = elem(mi, fi, r, e) ? a.dw[e] : 0.0f; }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int fi = 0; fi < FPW; ++fi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { size_t e; if (elem(mi, fi, r, e)) a.dw[e] = old[mi][fi][r] + acc[mi][fi][r]; }
        } else {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int fi = 0; fi < FPW; ++fi)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        size_t e;
                        if (!elem(mi, fi, r, e)) continue;
                        if (a.slabs) a.slabs[(size_t)bsplit * a.Cout * 9 * a.Ctot + e] = acc[mi][fi][r];   // this split's own slab
                        else atomicAdd(a.dw + e, acc[mi][fi][r]);
                    }
        }
    }
    if (bchunk == 0 && a.db) {                           // fold the NPH pixel phases (of every team) in LDS: one atomic per channel
        __syncthreads();
        sdb[threadIdx.x] = dbacc;
        __syncthreads();
        if (threadIdx.x < 16 * MT && co0 + tid < a.Cout) {
            float t = 0.0f;
#pragma unroll
            for (int ph = 0; ph < NPH * KS; ++ph) t += sdb[ph * 16 * MT + tid];
            if (a.db_slabs) a.db_slabs[(size_t)bsplit * a.Cout + co0 + tid] = t;
            else if (a.det == 2) a.db[co0 + tid] = t;
            else if (a.det) a.db[co0 + tid] += t;
            else atomicAdd(a.db + co0 + tid, t);
        }
    }
#ifdef COLVO_WTRACE
    if (a.trace) {
        __builtin_amdgcn_s_waitcnt(0);          // the flush's atomics / stores have left the wave (vmcnt 0)
        __syncthreads();
        WT_STAMP(wt_flush);
        if (threadIdx.x == 0) {
            long long* r = a.trace + (size_t)blockIdx.x * 16;
            r[0] = wt_start; r[1] = wt_first - wt_start; r[2] = wt_store; r[3] = wt_compute; r[4] = wt_flush - wt_loop_end;
            r[5] = wt_flush - wt_start; r[6] = (long long)wall_clock64() - wt_wall0; r[7] = (t_end - t_begin + KS - 1) / KS;
            r[8] = wt_issue; r[9] = wt_mfma;
        }
    }
#endif
}
#undef WT_STAMP

// deterministic form, second launch: dst[i] += slabs[0][i] + slabs[1][i] + ... in split order; the bias slabs ride in the same launch
// (blocks beyond `wblocks`).  Eight slab loads are in flight before the adds consume them in order (one load per add was a chain of
// up to 32 dependent trips to L2: 20 us a launch on average).
__device__ __forceinline__ void reduce_slabs(const float* __restrict__ slabs, int nsplit, size_t n, float* __restrict__ dst,
                                             size_t first, size_t stride) {
    for (size_t i = first; i < n; i += stride) {
        float t = 0.0f;
        int q = 0;
        for (; q + 8 <= nsplit; q += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(q + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; q < nsplit; ++q) t += slabs[(size_t)q * n + i];
        dst[i] += t;
    }
}
__global__ __launch_bounds__(NT) void k_wgrad_reduce(const float* __restrict__ slabs, int nsplit, size_t n, float* __restrict__ dst,
                                                     unsigned wblocks, const float* __restrict__ bslabs, size_t nb,
                                                     float* __restrict__ bdst) {
    if (blockIdx.x < wblocks) reduce_slabs(slabs, nsplit, n, dst, (size_t)blockIdx.x * NT + threadIdx.x, (size_t)wblocks * NT);
    else reduce_slabs(bslabs, nsplit, nb, bdst, (size_t)(blockIdx.x - wblocks) * NT + threadIdx.x, (size_t)(gridDim.x - wblocks) * NT);
}

// ---- grouped second launch (colvo_wgrad_reduce_group): dst += slabs[0] + slabs[1] + ... for up to 2 x COLVO_WGRAD_GROUP_MAX tensors ----
// One entry per tensor (a layer's weights, a layer's bias).  An entry's work is cut into float4 columns x SP split partitions:
// thread (s, q) of a block loads the float4 column q of the splits s, s + SP, s + 2 SP, ... (at most 8: all in flight), the SP
// partial sums of a column meet in LDS and thread (0, q) adds them IN ORDER to dst -- a fixed tree, so the result is bitwise
// repeatable.  SP is a power of two <= 64 chosen so that a thread has at most 8 loads (256 splits -> 32 partitions).
struct ReduceEntry {
    const float* slabs;
    float* dst;
    long long n4;             // float4 columns of the tensor
    int nsplit, sp;
    unsigned blk0;            // first block of this entry
};
struct ReduceGroup {
    ReduceEntry e[2 * COLVO_WGRAD_GROUP_MAX];
    int n;
};
__global__ __launch_bounds__(NT) void k_wgrad_reduce_group(const ReduceGroup g) {
    __shared__ f32x4 part[NT];
    int ei = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i)
        if (blockIdx.x >= g.e[i].blk0) ei = i;
    const ReduceEntry e = g.e[ei];
    const int qpb = NT / e.sp;                                    // columns per block
    const int s = threadIdx.x / qpb, ql = threadIdx.x - s * qpb;
    const long long q = (long long)(blockIdx.x - e.blk0) * qpb + ql;
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
    if (q < e.n4) {
        const f32x4* base = reinterpret_cast<const f32x4*>(e.slabs) + q;
        f32x4 v[8];
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sp_i = s + u * e.sp;
            if (sp_i < e.nsplit) { v[u] = base[(long long)sp_i * e.n4]; cnt = u + 1; } else v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
        (void)cnt;
        for (int sp_i = s + 8 * e.sp; sp_i < e.nsplit; sp_i += e.sp) t += base[(long long)sp_i * e.n4];     // (more than 8 x 64 splits)
    }
    if (e.sp > 1) {
        part[threadIdx.x] = t;
        __syncthreads();
        if (s == 0) {
            for (int j = 1; j < e.sp; ++j) t += part[j * qpb + ql];
        }
    }
    if (s == 0 && q < e.n4) {
        f32x4* d = reinterpret_cast<f32x4*>(e.dst) + q;
        *d = *d + t;
    }
}

#ifdef COLVO_WTRACE
// developer build (tools/wtrace_wgrad.sh): per-workgroup phase stamps of k_wgrad3x3, printed as means over the workgroups
static long long* g_wtrace = nullptr;
static int g_wtrace_calls = 0;
inline void wtrace_begin(WgradK& k, unsigned nwg, hipStream_t s) {
    k.trace = nullptr;
    if (!getenv("COLVO_WTRACE") || nwg > (1u << 14)) return;
    if (!g_wtrace) (void)hipMalloc(&g_wtrace, (size_t)(1u << 14) * 16 * sizeof(long long));
    (void)hipMemsetAsync(g_wtrace, 0, (size_t)nwg * 16 * sizeof(long long), s);
    k.trace = g_wtrace;
}
inline void wtrace_end(const WgradK& k, unsigned nwg, int MT, int NG, bool tail, int ks, hipStream_t s) {
    if (!k.trace || (++g_wtrace_calls % atoi(getenv("COLVO_WTRACE"))) != 0) return;
    (void)hipStreamSynchronize(s);
    std::vector<long long> h((size_t)nwg * 16);
    (void)hipMemcpy(h.data(), g_wtrace, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double m[16] = {0};
    long long t0 = h[0], t1 = 0;
    for (unsigned i = 0; i < nwg; ++i) {
        for (int j = 1; j < 16; ++j) m[j] += (double)h[(size_t)i * 16 + j] / nwg;
        t0 = std::min(t0, h[(size_t)i * 16]); t1 = std::max(t1, h[(size_t)i * 16] + h[(size_t)i * 16 + 5]);
    }
    const double cyc_per_us = m[6] > 0 ? m[5] / (m[6] * 0.01) : 0.0;       // shader clocks per microsecond (wall clock: 100 MHz)
    const double nt = std::max(1.0, m[7]);
    fprintf(stderr, "[wtrace] MT=%d NG=%d tail=%d KS=%d S=%d Cout=%d Ctot=%d %dx%d tile %dx%d grid=%u nsplit=%d tiles/wg %.1f | clock %.0f MHz | "
            "mean wg (us): life %.2f = setup+first-load %.2f + store %.2f + compute %.2f + flush %.2f | per tile: store %.3f compute %.3f = "
            "load issue %.3f + MFMA phase %.3f + barrier wait %.3f\n",
            MT, NG, (int)tail, ks, k.g.stride, k.Cout, k.Ctot, k.Ho, k.Wo, k.toh, k.tow, nwg, k.nsplit, m[7], cyc_per_us,
            m[5] / cyc_per_us, m[1] / cyc_per_us, m[2] / cyc_per_us, m[3] / cyc_per_us, m[4] / cyc_per_us, m[2] / cyc_per_us / nt,
            m[3] / cyc_per_us / nt, m[8] / cyc_per_us / nt, m[9] / cyc_per_us / nt, (m[3] - m[8] - m[9]) / cyc_per_us / nt);
}
#endif

// plan-only calls report the split count; deterministic calls point the kernel at the caller's scratch.  Returns 1 when the
// caller has nothing more to do (plan written), 0 to go on, < 0 never; errors are reported through *err.
inline bool wgrad_prepare(WgradK& k, int nsplit, int* err) {
    *err = 0;
    if (k.plan_out) { *k.plan_out = nsplit; return true; }
    k.slabs = nullptr; k.db_slabs = nullptr;
    // deterministic form with one split: every weight element has exactly one writer, which adds to dw / db with a plain
    // read-modify-write -- reproducible without slabs or a second launch
    k.det = (k.scratch && nsplit == 1 && !k.slabs_only) ? 1 : 0;
    // ... and when the caller vouches that dw / db are still zero (colvo_conv_wgrad_clean) the sole writer just stores
    if (k.clean && nsplit == 1 && !k.slabs_only && TUNE(wgrad_store_clean)) k.det = 2;
    if (k.scratch && (nsplit > 1 || k.slabs_only)) {
        const long long wsize = (long long)k.Cout * 9 * k.Ctot;
        const long long need = (long long)nsplit * (wsize + k.Cout) * 4;
        if (need > k.scratch_bytes) {
            set_error("colvo_conv_wgrad_det: scratch of %lld bytes, %lld needed (colvo_conv_wgrad_scratch_bytes)", k.scratch_bytes, need);
            *err = (int)hipErrorInvalidValue;
            return true;
        }
        k.slabs = (float*)const_cast<char*>(k.scratch);
        k.db_slabs = k.slabs + (size_t)nsplit * wsize;
    }
    return false;
}

inline int wgrad_finish(const WgradK& k, int nsplit, hipStream_t s) {
    if (!k.slabs || k.slabs_only) return 0;          // (slabs_only: colvo_wgrad_reduce_group adds them later)
    const size_t wsize = (size_t)k.Cout * 9 * k.Ctot;
    const unsigned blocks = (unsigned)std::min<size_t>((wsize + NT - 1) / NT, 2048);
    const unsigned bblocks = k.db ? (unsigned)((k.Cout + NT - 1) / NT) : 0;
    colvo::launch(k_wgrad_reduce, dim3(blocks + bblocks), dim3(NT), 0, s, (const float*)k.slabs, nsplit, wsize, k.dw, blocks,
                       (const float*)k.db_slabs, (size_t)k.Cout, k.db);
    COLVO_CHECK_LAUNCH("k_wgrad_reduce");
    return 0;
}

// The floor of a weight-gradient grid: `base` workgroups (tuning: wgrad_wg_lo / wgrad_team_wgs), but HALF of that while a workgroup of
// the full grid would walk fewer than wgrad_short_walk pixel tiles.  A workgroup's fixed costs -- 2 us of set-up and first loads, 6-7 us
// of fp32 atomics at the end (profiles/r4_wgrad_phases.md) -- are then most of its life; alone on the GPU the kernel is still fastest
// with one workgroup per CU (what round 2-3 tuned for), but in the step three kernels share the CUs and the sum of workgroup lives is
// what counts: 128 instead of 256 workgroups at 16 frames measured -2.8 % per step (1.3097 -> 1.2726 ms), 96 -> +1 %, 64 -> +9 %;
// at 64 / 128 frames, where the walks are 4-8 times longer, halving every layer cost 0.4-0.5 %.
static thread_local bool g_grid_halved = false;     // what the last wgrad_grid_floor() decided (colvo_form_counts)
static inline int wgrad_grid_floor(int base, int ntiles, int per_split) {
    const int short_walk = (int)TUNE(wgrad_short_walk);
    const int nsplit_at_base = std::max(1, (base + per_split - 1) / per_split);
    g_grid_halved = short_walk > 0 && ntiles / nsplit_at_base < short_walk;
    return g_grid_halved ? std::max(1, base / 2) : base;
}

template <typename T, int MT, int NG, bool TAIL, int KS>
int launch_wgrad_teams(WgradK k, hipStream_t s) {
    constexpr int G = TT<T>::G;
    constexpr int CK = NG * G, PIXP = TAIL ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16), DYP = dy_pitch<T, MT>();
    constexpr int NFR = (9 * CK + 15) / 16, FPW = (NFR + 3) / 4;
    const int S = k.g.stride;
    const int PH = (k.toh - 1) * S + 3;
    const size_t stage = ((size_t)BM * DYP + (size_t)PH * k.pwl * PIXP + 15) & ~(size_t)15;
    const size_t lds = std::max(stage * KS, (size_t)MT * FPW * NT * 16 + (size_t)NT * KS * 4) + 64;   // staging | exchange slab + db
    COLVO_CHECK_ARG(lds <= 160 * 1024, "wgrad: %d teams need %zu bytes of LDS", KS, lds);
    static size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3x3<T, MT, NG, TAIL, KS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = 160 * 1024;
    }
    const int chunks = (k.g.C[0] + k.g.C[1]) / CK;
    const int cot = (k.Cout + 16 * MT - 1) / (16 * MT);
    // one workgroup per CU (KS * 4 waves fill its SIMDs): as many pixel-range splits as keep the grid within 256
    const int per_split = chunks * cot;
    const int wg_target = wgrad_grid_floor((int)TUNE(wgrad_team_wgs), k.ntiles, per_split);
    int nsplit = std::max(1, wg_target / per_split);
    if (nsplit > k.ntiles) nsplit = k.ntiles;
    k.tiles_per_split = (k.ntiles + nsplit - 1) / nsplit;
    nsplit = (k.ntiles + k.tiles_per_split - 1) / k.tiles_per_split;
    const int xcd_on = (int)TUNE(xcd_remap);
    k.nsplit = nsplit; k.cot = cot; k.xcd = xcd_on;
    { int err; if (wgrad_prepare(k, nsplit, &err)) return err; }
    dim3 grid((unsigned)(nsplit * cot * chunks), 1, 1);
#ifdef COLVO_WTRACE
    wtrace_begin(k, grid.x, s);
#endif
    form_hit(g_grid_halved ? FORM_WGRAD_HALVED_GRID : FORM_WGRAD_FULL_GRID);
    if (k.det == 2) form_hit(FORM_WGRAD_STORE_CLEAN);
    colvo::launch((k_wgrad3x3<T, MT, NG, TAIL, KS>), grid, dim3(NT * KS), lds, s, k);
    COLVO_CHECK_LAUNCH("k_wgrad3x3 (teams)");
#ifdef COLVO_WTRACE
    wtrace_end(k, grid.x, MT, NG, TAIL, KS, s);
#endif
    return wgrad_finish(k, nsplit, s);
}

template <typename T, int MT, int NG, bool TAIL>
int launch_wgrad_tail(WgradK k, hipStream_t s) {
    constexpr int G = TT<T>::G;
    constexpr int CK = NG * G, PIXP = TAIL ? pitch_bytes_s2(NG * 16) : pitch_bytes(NG * 16), DYP = dy_pitch<T, MT>();
    const int S = k.g.stride;
    const int PH = (k.toh - 1) * S + 3, PW = k.pwl;
    {
        // Teams pay only on the two full-resolution layers (16 output channels: a one-fragment co tile, so the exchange is
        // 20 KB per team, and one or two slabs in the (co tile, channel chunk) grid, so 256 workgroups walk 40-80 tiles each).
        // Measured at batch 16, us: up1 47.8 -> 37.5, iconv1 38.5 -> 28.9; with 32-wide co tiles it loses (enc1b 30.7 -> 40.2,
        // up2 / iconv2 32.8 -> 37.5) and everywhere else the walk is short and the per-workgroup costs decide: -1...+8
        // (gpurun_out/r2_bench_conv_slabs*.log).  4 teams = 1024 threads, 128 registers per lane.
        const int teams = (int)TUNE(wgrad_teams);          // tuning knob; 1 = off
        const int max_slabs = (int)TUNE(wgrad_team_max_slabs);   // tuning knob
        const int slabs = ((k.g.C[0] + k.g.C[1]) / CK) * ((k.Cout + 16 * MT - 1) / (16 * MT));
        const size_t stage = ((size_t)BM * DYP + (size_t)PH * PW * PIXP + 15) & ~(size_t)15;
        if constexpr (!TAIL && MT == 1)
            if (teams >= 4 && slabs <= max_slabs && stage * 4 + 64 <= 160 * 1024) return launch_wgrad_teams<T, MT, NG, TAIL, 4>(k, s);
    }
    const size_t lds = (size_t)BM * DYP + (size_t)PH * PW * PIXP + 64;
    COLVO_CHECK_ARG(lds <= 160 * 1024, "wgrad: tile needs %zu bytes of LDS", lds);
    static size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad3x3<T, MT, NG, TAIL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = 160 * 1024;
    }
    const int chunks = (k.g.C[0] + k.g.C[1]) / CK;
    const int cot = (k.Cout + 16 * MT - 1) / (16 * MT);
    // Pixel-range splits: every split adds one fp32 atomic per weight (chip-wide ~1.3 TB/s of atomics), so cap
    // the atomic traffic at ~3 MB per launch (re-tuned for the 32-wide co tile: 12 MB 592 us, 6 MB 575, 3 MB 566),
    // but keep at least ~256 workgroups in flight and at most ~1024.
    const double wbytes = (double)k.Cout * 9.0 * k.Ctot * 4.0;
    const int per_split = chunks * cot;
    const double atomic_budget = TUNE_F(wgrad_atomic_mb) * 1e6;
    const int wg_lo = wgrad_grid_floor((int)TUNE(wgrad_wg_lo), k.ntiles, per_split);
    const int wg_hi = (int)TUNE(wgrad_wg_hi);
    int nsplit = (int)(atomic_budget / wbytes);
    const int lo = (wg_lo + per_split - 1) / per_split, hi = (wg_hi + per_split - 1) / per_split;
    if (nsplit > hi) nsplit = hi;
    if (nsplit < lo) nsplit = lo;
    if (nsplit > k.ntiles) nsplit = k.ntiles;
    if (nsplit < 1) nsplit = 1;
    k.tiles_per_split = (k.ntiles + nsplit - 1) / nsplit;
    nsplit = (k.ntiles + k.tiles_per_split - 1) / k.tiles_per_split;
    const int xcd_on = (int)TUNE(xcd_remap);
    k.nsplit = nsplit; k.cot = cot; k.xcd = xcd_on;
    { int err; if (wgrad_prepare(k, nsplit, &err)) return err; }
    dim3 grid((unsigned)(nsplit * cot * chunks), 1, 1);
#ifdef COLVO_WTRACE
    wtrace_begin(k, grid.x, s);
#endif
    form_hit(g_grid_halved ? FORM_WGRAD_HALVED_GRID : FORM_WGRAD_FULL_GRID);
    if (k.det == 2) form_hit(FORM_WGRAD_STORE_CLEAN);
    colvo::launch((k_wgrad3x3<T, MT, NG, TAIL>), grid, dim3(NT), lds, s, k);
    COLVO_CHECK_LAUNCH("k_wgrad3x3");
#ifdef COLVO_WTRACE
    wtrace_end(k, grid.x, MT, NG, TAIL, 1, s);
#endif
    return wgrad_finish(k, nsplit, s);
}

// --------------------------------------------------------------------------------------------- //
// weight gradient of a conv over a nearest-2x UP-SAMPLED source: four output classes, 16 products   //
// --------------------------------------------------------------------------------------------- //
// y[2a+py][2b+px] = sum_{ky,kx} w[ky][kx] x[a + r(py,ky)][b + r(px,kx)] with r(0,.) = (-1, 0, 0), r(1,.) = (0, 0, +1): the 3x3 taps of
// an output pixel of parity class (py, px) fall on only 2x2 source pixels.  So
//     dw[ky][kx] = sum_{py,px} dW'[py][px][rho(py,ky)][rho(px,kx)],   rho(0,.) = (0, 1, 1),  rho(1,.) = (0, 0, 1),
//     dW'[py][px][r][c] = sum_{a,b} dy[2a+py][2b+px] (x) x[a+py+r-1][b+px+c-1]
// -- 4 classes x 4 source taps = 16 products over the SOURCE positions instead of 9 taps over four times as many output pixels:
// 2.25 x fewer MFMAs.  The general kernel stages, per 512 output pixels, four dY tiles and four gathered 10x18-pixel patches of the
// virtual up-sampled image; here a workgroup takes a tile of <= 128 source positions, stages their four dY class planes and ONE
// (toh+2) x (tow+2) patch of real source pixels (-44 % staged bytes), and wave w owns class w: per 32-position k-step it reads MT dY
// fragments and 4 taps x CK/16 patch fragments for 4 * CK/16 * MT MFMAs (0.4-0.6 LDS fragment reads per MFMA instead of 0.7).
// At the end the four classes meet in LDS -- each dW' entry is added to the 1, 2 or 4 taps it stands for, one wave at a time -- and
// the workgroup adds its [9][16 MT][CK] block to dw exactly like k_wgrad3x3 (fp32 atomics, or this split's slab in the deterministic
// form).  Grid, pixel-range splits and XCD order as in k_wgrad3x3 (tiles run over source positions).
template <typename T, int MT>
__global__ __launch_bounds__(NT) void k_wgrad_up2(const WgradK a) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int NG = 4, CK = NG * G;               // 32 bf16 / 16 f32 channels per chunk
    constexpr int NCF = CK / 16;                     // column fragments per source tap
    constexpr int NACC = 4 * NCF;                    // accumulators per dY fragment: 4 source taps x NCF
    constexpr int PIXP = pitch_bytes(NG * 16);
    constexpr int DYP = dy_pitch<T, MT>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int cls = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave = output parity class
    const int cpy = cls >> 1, cpx = cls & 1;
    const int l15 = lane & 15, kg = lane >> 4;
    const int lid = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, a.xcd));
    const int per_split = gridDim.x / a.nsplit;
    const int bsplit = lid / per_split, brem = lid - bsplit * per_split;
    const int bchunk = brem / a.cot, bco = brem - bchunk * a.cot;
    const int co0 = bco * 16 * MT;
    const int c0 = bchunk * CK;                      // single source: channel offset = weight column offset
    const int PH = a.toh + 2, PW = a.tow + 2;        // patch of SOURCE pixels: tile + 1 each side
    const int PWL = a.pwl;                           // its LDS row pitch in pixels (conv_common.h wgrad_row_pitch)
    const int npix = a.toh * a.tow;
    char* sDY = smem;                                // [4 classes][BM][16 MT]
    char* sX = smem + 4 * BM * DYP;                  // [PH][PWL][CK]

    f32x4 acc[MT][NACC];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[mi][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.0f;

    const int t_begin = bsplit * a.tiles_per_split;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_split);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    constexpr int DGR = 16 * MT / G;                 // dY granules per (class, position)
    constexpr int DIT = 4 * BM * DGR / NT;
    constexpr int PPF = 3;                           // patch <= 10 x 18 positions x 4 granules
    const int ptotal = PH * PW * NG;
    u32x4 dyv[DIT], pv[PPF];
    struct TileC { int b, ty, tx; };
    auto tile_next = [&](TileC& c) {
        if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.b; } }
    };
    TileC cur;
    {
        const int t = __builtin_amdgcn_readfirstlane(t_begin);
        cur.b = t / tiles_per_img;
        const int tr_ = t - cur.b * tiles_per_img;
        cur.ty = tr_ / a.tiles_x; cur.tx = tr_ - cur.ty * a.tiles_x;
    }
    const int Hs = a.g.Hs[0], Ws = a.g.Ws[0], Cs = a.g.C[0];
    const long long dy_bytes = (long long)a.B * a.Ho * a.Wo * a.Cout * ES;
    const long long x_bytes = (long long)a.B * Hs * Ws * Cs * ES;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)(dy_bytes < 0x7fffffffLL ? dy_bytes : 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.g.src[0], 0, (int)(x_bytes < 0x7fffffffLL ? x_bytes : 0x7fffffffLL), 0x00020000);

    int dy_off[DIT], dy_yx[DIT], dy_lds[DIT];
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
        const int i = it * NT + tid;
        const int c4 = i / (BM * DGR), rem = i - c4 * (BM * DGR);
        const int pos = rem / DGR, gch = rem - pos * DGR;
        const int oy = mdiv(pos, a.m_tow), ox = pos - oy * a.tow;
        const bool ok = (pos < npix) && (co0 + gch * G < a.Cout);
        dy_off[it] = ok ? (((2 * oy + (c4 >> 1)) * a.Wo + 2 * ox + (c4 & 1)) * a.Cout + co0 + gch * G) * ES : OOB_OFF;
        dy_yx[it] = (oy << 16) | ox;
        dy_lds[it] = (c4 * BM + pos) * DYP + gch * 16;
    }
    int p_yx[PPF], p_cg[PPF], p_lds[PPF];
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        const int i = it * NT + tid;
        const int pix = i / NG;
        p_cg[it] = i - pix * NG;
        const int py = mdiv(pix, a.m_pw), px = pix - py * PW;
        p_yx[it] = (i < ptotal) ? ((py << 16) | px) : (0x7fff << 16);
        p_lds[it] = (py * PWL + px) * PIXP + p_cg[it] * 16;
    }
    auto load_tile = [&](const TileC& c) {
        const int oy0 = c.ty * a.toh, ox0 = c.tx * a.tow;
        {
            const int base = ((c.b * a.Ho + 2 * oy0) * a.Wo + 2 * ox0) * a.Cout * ES;      // wave-uniform: the scalar offset
            const int remy = Hs - oy0, remx = Ws - ox0;
            if (remy >= a.toh && remx >= a.tow) {
#pragma unroll
                for (int it = 0; it < DIT; ++it) dyv[it] = bld16(rdy, dy_off[it], base);
            } else {
#pragma unroll
                for (int it = 0; it < DIT; ++it) {
                    const bool ok = ((dy_yx[it] >> 16) < remy) && ((dy_yx[it] & 0xffff) < remx);
                    dyv[it] = bld16(rdy, ok ? dy_off[it] : OOB_OFF, base);
                }
            }
        }
        const int base = c.b * Hs * Ws * Cs * ES;
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int vy = oy0 - 1 + (p_yx[it] >> 16), vx = ox0 - 1 + (p_yx[it] & 0xffff);
            const bool inb = ((unsigned)vy < (unsigned)Hs) && ((unsigned)vx < (unsigned)Ws);
            pv[it] = bld16(rx, inb ? ((vy * Ws + vx) * Cs + c0 + p_cg[it] * G) * ES : OOB_OFF, base);
        }
    };
    constexpr int NPH = NT / (16 * MT);
    const int db_co = tid % (16 * MT), db_ph = tid / (16 * MT);

    if (t_begin < t_end) load_tile(cur);
    for (int t = t_begin; t < t_end; ++t) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < DIT; ++it) st16(sDY + dy_lds[it], dyv[it]);
#pragma unroll
        for (int it = 0; it < PPF; ++it)
            if (it * NT + tid < ptotal) st16(sX + p_lds[it], pv[it]);
        __syncthreads();
        tile_next(cur);
        if (t + 1 < t_end) load_tile(cur);             // in flight during the MFMAs below

        if (bchunk == 0 && a.db) {                     // bias gradient: all four class planes (rows beyond the tile hold zeros)
            float s0 = 0.0f;
#pragma unroll 4
            for (int p = db_ph; p < 4 * BM; p += NPH) {
                if constexpr (ES == 2) s0 += bf2f(*reinterpret_cast<const uint16_t*>(sDY + p * DYP + db_co * 2));
                else s0 += *reinterpret_cast<const float*>(sDY + p * DYP + db_co * 4);
            }
            dbacc += s0;
        }

        const char* sDYc = sDY + cls * BM * DYP;
        if constexpr (ES == 2) {
            // software pipeline over the k-steps as in k_wgrad3x3: the fragments of step ks+1 are in flight during the MFMAs of step ks
            const int q = l15 >> 2, pp = lane & 3;
            auto read_step = [&](int ks, s16x8 (&af)[MT], s16x8 (&bfv)[NACC]) {
                int xo[2], yo[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int p = 32 * ks + 16 * (kg >> 1) + 8 * h + 4 * (kg & 1) + q;       // K position -> tile position (see k_wgrad3x3)
                    yo[h] = p * DYP;
                    if (p >= npix) p = 0;                                              // its dY rows are zero
                    const int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
                    xo[h] = ((oy + cpy) * PWL + ox + cpx) * PIXP;                      // source tap (0, 0) of this class
                }
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(sDYc + yo[0] + (16 * mi + 4 * pp) * 2));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(sDYc + yo[1] + (16 * mi + 4 * pp) * 2));
                    af[mi] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                    for (int f = 0; f < NCF; ++f) {
                        const int bo = ((tp >> 1) * PWL + (tp & 1)) * PIXP + (16 * f + 4 * pp) * 2;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + xo[0] + bo));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sX + xo[1] + bo));
                        bfv[tp * NCF + f] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
            };
            s16x8 af[2][MT], bfv[2][NACC];
            read_step(0, af[0], bfv[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks + 1 < 4) read_step(ks + 1, af[(ks + 1) & 1], bfv[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qq = 0; qq < NACC; ++qq)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
                        acc[mi][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, af[ks & 1][mi]), __builtin_bit_cast(bf16x8, bfv[ks & 1][qq]), acc[mi][qq], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            int p = kg;
            int oy = mdiv(p, a.m_tow), ox = p - oy * a.tow;
            for (int ks = 0; ks < BM / 4; ++ks) {
                const bool live = p < npix;
                const int xo = live ? ((oy + cpy) * PWL + ox + cpx) * PIXP : 0;
                float av[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) av[mi] = *reinterpret_cast<const float*>(sDYc + p * DYP + (16 * mi + l15) * 4);
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
#pragma unroll
                    for (int f = 0; f < NCF; ++f) {
                        const float bvv = *reinterpret_cast<const float*>(sX + xo + ((tp >> 1) * PWL + (tp & 1)) * PIXP + (16 * f + l15) * 4);
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
                            acc[mi][tp * NCF + f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi], bvv, acc[mi][tp * NCF + f], 0, 0, 0);
                    }
                p += 4; ox += 4;
                while (ox >= a.tow) { ox -= a.tow; ++oy; }
            }
        }
    }

    mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[MT * NACC]>(acc));
    // ---- the four classes meet ----
    // Every wave stores its 16 dW' blocks [class][source tap r, c][co][ci] to LDS with plain conflict-free 16-byte stores (its
    // accumulators as they are: lane-major), all four at once; then all 256 threads gather: dw[ky][kx][co][ci] = sum over the four
    // classes of dW'[class][rho(py, ky)][rho(px, kx)][co][ci] -- four reads per output element, reading lane-major again.  (Round 3
    // added each block to the taps it stands for with LDS read-modify-writes, one wave at a time behind a zero-fill pass: 144
    // two-way-conflicting RMWs per lane, four times in a row -- a third of the kernel's LDS bank conflicts and ~3 us of a 30 us launch.)
    __syncthreads();                                   // staging buffers are free
    f32x4* sEx = reinterpret_cast<f32x4*>(smem);       // [class][tp][mi][f][lane]
    {
        f32x4* mine = sEx + (size_t)cls * 4 * MT * NCF * 64;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int f = 0; f < NCF; ++f) mine[((tp * MT + mi) * NCF + f) * 64 + lane] = acc[mi][tp * NCF + f];
    }
    __syncthreads();
    // output block (tap, mi, f): lane (l15, kg) holds rows co = 16 mi + 4 kg + r, column ci = 16 f + l15, exactly the accumulator
    // layout; wave w takes the blocks w, w + 4, ...
    // rho(0, k) = (0, 1, 1)[k], rho(1, k) = (0, 0, 1)[k]: the source-tap row / column that tap index k falls on for parity 0 / 1
    constexpr int NBLK = 9 * MT * NCF;
    for (int blk = cls; blk < NBLK; blk += 4) {
        const int tap = blk / (MT * NCF), rem = blk - tap * (MT * NCF);
        const int mi = rem / NCF, f = rem - mi * NCF;
        const int ky = tap / 3, kx = tap - 3 * ky;
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int py = c4 >> 1, px = c4 & 1;
            const int r = py == 0 ? (ky >= 1) : (ky >= 2), c = px == 0 ? (kx >= 1) : (kx >= 2);
            sum += sEx[(((size_t)c4 * 4 + (r * 2 + c)) * MT + mi) * NCF * 64 + f * 64 + lane];
        }
        const size_t e0 = (size_t)(co0 + 16 * mi + 4 * kg) * 9 * a.Ctot + (size_t)tap * a.Ctot + c0 + 16 * f + l15;
        const size_t erow = (size_t)9 * a.Ctot;
        const int nrow = min(4, a.Cout - (co0 + 16 * mi + 4 * kg));       // rows of this lane inside the tensor (<= 0: none)
        if (a.slabs) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (rr < nrow) a.slabs[(size_t)bsplit * a.Cout * 9 * a.Ctot + e0 + rr * erow] = sum[rr];
        } else if (a.det == 2) {            // one split, arena known to be zero: plain stores (see k_wgrad3x3)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (rr < nrow) a.dw[e0 + rr * erow] = sum[rr];
        } else if (a.det) {                 // deterministic form, one split = sole writer: plain read-modify-write, loads first
            float old[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) old[rr] = rr < nrow ? a.dw[e0 + rr * erow] : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (rr < nrow) a.dw[e0 + rr * erow] = old[rr] + sum[rr];
        } else {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (rr < nrow) atomicAdd(a.dw + e0 + rr * erow, sum[rr]);
        }
    }
    if (bchunk == 0 && a.db) {
        __syncthreads();
        float* sdb = reinterpret_cast<float*>(smem);
        sdb[tid] = dbacc;
        __syncthreads();
        if (tid < 16 * MT && co0 + tid < a.Cout) {
            float t = 0.0f;
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) t += sdb[ph * 16 * MT + tid];
            if (a.db_slabs) a.db_slabs[(size_t)bsplit * a.Cout + co0 + tid] = t;
            else if (a.det == 2) a.db[co0 + tid] = t;
            else if (a.det) a.db[co0 + tid] += t;
            else atomicAdd(a.db + co0 + tid, t);
        }
    }
}

template <typename T, int MT>
int launch_wgrad_up2(WgradK k, hipStream_t s) {
    constexpr int G = TT<T>::G, CK = 4 * G, PIXP = pitch_bytes(64), DYP = dy_pitch<T, MT>();
    // staging buffers | the class exchange: 4 classes x 4 source taps x MT x CK/16 accumulator blocks of 64 lanes x 16 bytes
    const size_t lds = std::max((size_t)4 * BM * DYP + (size_t)(k.toh + 2) * k.pwl * PIXP, (size_t)16 * MT * (CK / 16) * 64 * 16) + 64;
    COLVO_CHECK_ARG(lds <= 160 * 1024, "wgrad (up-sampled source): tile needs %zu bytes of LDS", lds);
    static size_t configured = 0;
    if (lds > 48 * 1024 && lds > configured && !k.plan_out) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wgrad_up2<T, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = 160 * 1024;
    }
    const int chunks = k.g.C[0] / CK;
    const int cot = (k.Cout + 16 * MT - 1) / (16 * MT);
    const double wbytes = (double)k.Cout * 9.0 * k.Ctot * 4.0;
    const int per_split = chunks * cot;
    int nsplit = (int)(TUNE_F(wgrad_atomic_mb) * 1e6 / wbytes);
    const int lo = (wgrad_grid_floor((int)TUNE(wgrad_wg_lo), k.ntiles, per_split) + per_split - 1) / per_split;
    const int hi = ((int)TUNE(wgrad_wg_hi) + per_split - 1) / per_split;
    if (nsplit > hi) nsplit = hi;
    if (nsplit < lo) nsplit = lo;
    if (nsplit > k.ntiles) nsplit = k.ntiles;
    if (nsplit < 1) nsplit = 1;
    k.tiles_per_split = (k.ntiles + nsplit - 1) / nsplit;
    nsplit = (k.ntiles + k.tiles_per_split - 1) / k.tiles_per_split;
    k.nsplit = nsplit; k.cot = cot; k.xcd = (int)TUNE(xcd_remap);
    { int err; if (wgrad_prepare(k, nsplit, &err)) return err; }
    form_hit(FORM_WGRAD_UP2);
    form_hit(g_grid_halved ? FORM_WGRAD_HALVED_GRID : FORM_WGRAD_FULL_GRID);
    if (k.det == 2) form_hit(FORM_WGRAD_STORE_CLEAN);
    colvo::launch((k_wgrad_up2<T, MT>), dim3((unsigned)(nsplit * cot * chunks)), dim3(NT), lds, s, k);
    COLVO_CHECK_LAUNCH("k_wgrad_up2");
    return wgrad_finish(k, nsplit, s);
}

template <typename T, int MT, int NG>
int launch_wgrad(const WgradK& k, hipStream_t s) {
    const int S = k.g.stride;
    const long ptotal = (long)((k.toh - 1) * S + 3) * ((k.tow - 1) * S + 3) * NG;
    if (ptotal > 3 * NT) return launch_wgrad_tail<T, MT, NG, true>(k, s);
    return launch_wgrad_tail<T, MT, NG, false>(k, s);
}

template <typename T, int MT>
int launch_wgrad_ng(const WgradK& k, int ng, hipStream_t s) {
    switch (ng) {
        case 4: return launch_wgrad<T, MT, 4>(k, s);
        case 2: return launch_wgrad<T, MT, 2>(k, s);
        default: return launch_wgrad<T, MT, 1>(k, s);
    }
}

template <typename T>
int launch_wgrad_t(const WgradK& k, hipStream_t s) {
    constexpr int G = TT<T>::G;
    const int ng_max = (int)TUNE(wgrad_ng_max);   // tuning knob
    int ng = ng_max;
    for (int i = 0; i < 2; ++i)
        if (k.g.C[i] > 0) while (ng > 1 && (k.g.C[i] % (ng * G)) != 0) ng >>= 1;
    for (int i = 0; i < 2; ++i)
        COLVO_CHECK_ARG(k.g.C[i] % (ng * G) == 0, "wgrad: channel count %d is not a multiple of %d", k.g.C[i], G);
    // co tile: 32 wide (MT = 2) measured better than 64 on every layer -- twice the (co, chunk) combinations, so half the
    // pixel-range splits and half the fp32 atomic traffic for the same number of workgroups (wgrad 638 -> 589 us)
    const int mt_max = (int)TUNE(wgrad_mt_max);   // tuning knob
    // ... and 16 wide where the input is a single channel chunk (the high-resolution encoder layers: the (co tile, chunk) grid
    // is then 1-2 slabs; measured enc1a 27.0 -> 19.5, enc1b 30.9 -> 24.4, enc2a 26.3 -> 23.5 us; two-chunk layers lose)
    const int one_chunk_rule = (int)TUNE(wgrad_one_chunk_rule);   // A/B switch
    const bool one_chunk = one_chunk_rule && (k.g.C[0] + k.g.C[1]) == ng * G;
    // ... except where the pixel-tile walk is long: with >= mt4_min tiles per workgroup at a 256-workgroup grid the 64-wide tile's
    // LDS reuse (0.45 instead of 0.7 fragment reads per MFMA) outweighs its doubled atomic traffic -- measured over the stack:
    // 16 images 533 -> 596 us (32-wide stays), 64 images 1651 -> 1423, 128 images 3162 -> 2662, 64 images of 512x640 5854 -> 4775
    // (profiles/r3_tuning_check.md)
    const long walk = (long)k.ntiles * ((k.g.C[0] + k.g.C[1]) / (ng * G)) * ((k.Cout + 63) / 64) / 256;
    if (k.Cout >= 64 && (mt_max >= 4 || walk >= TUNE(wgrad_mt4_min_walk))) return launch_wgrad_ng<T, 4>(k, ng, s);
    if (k.Cout >= 32 && mt_max >= 2 && !one_chunk) return launch_wgrad_ng<T, 2>(k, ng, s);
    return launch_wgrad_ng<T, 1>(k, ng, s);
}

}  // namespace
}  // namespace colvo

using namespace colvo;

static int wgrad_impl(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw, float* db,
                      void* scratch, size_t scratch_bytes, int* plan_out, colvo_stream_t stream, int slabs_only = 0, int clean = 0);

extern "C" int colvo_conv_wgrad(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw,
                                float* db, colvo_stream_t stream) {
    return wgrad_impl(d, x0, x1, dy, dw, db, nullptr, 0, nullptr, stream);
}

extern "C" int colvo_conv_wgrad_clean(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw,
                                      float* db, int arena_is_zero, colvo_stream_t stream) {
    return wgrad_impl(d, x0, x1, dy, dw, db, nullptr, 0, nullptr, stream, 0, arena_is_zero ? 1 : 0);
}

extern "C" int colvo_conv_wgrad_det(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw,
                                    float* db, void* scratch, size_t scratch_bytes, colvo_stream_t stream) {
    COLVO_CHECK_ARG(scratch, "colvo_conv_wgrad_det: null scratch");
    return wgrad_impl(d, x0, x1, dy, dw, db, scratch, scratch_bytes, nullptr, stream);
}

extern "C" int colvo_conv_wgrad_slabs(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, void* scratch,
                                      size_t scratch_bytes, colvo_stream_t stream) {
    COLVO_CHECK_ARG(scratch, "colvo_conv_wgrad_slabs: null scratch");
    static const float dummy = 0.0f;            // dw / db are not touched in this form (the kernel only needs db != NULL for the bias sums)
    return wgrad_impl(d, x0, x1, dy, const_cast<float*>(&dummy), const_cast<float*>(&dummy), scratch, scratch_bytes, nullptr, stream, 1);
}

extern "C" int colvo_conv_wgrad_splits(const ColvoConvDesc* d) {
    if (!d || check_desc(d, "colvo_conv_wgrad_splits")) return 0;
    int nsplit = 0;
    static const char dummy = 0;
    if (wgrad_impl(d, &dummy, d->C1 ? &dummy : nullptr, &dummy, (float*)&dummy, nullptr, nullptr, 0, &nsplit, nullptr)) return 0;
    return nsplit;
}

extern "C" int colvo_wgrad_reduce_group(const ColvoWgradSlabs* sets, int n, colvo_stream_t stream) {
    COLVO_CHECK_ARG(sets && n >= 1 && n <= COLVO_WGRAD_GROUP_MAX, "colvo_wgrad_reduce_group: 1..%d sets", COLVO_WGRAD_GROUP_MAX);
    ReduceGroup g{};
    unsigned blk = 0;
    auto add = [&](const float* slabs, float* dst, long long n, int nsplit) {
        ReduceEntry& e = g.e[g.n++];
        e.slabs = slabs; e.dst = dst; e.n4 = n / 4; e.nsplit = nsplit;
        int sp = 1;
        while (sp < 64 && sp * 8 < nsplit) sp *= 2;
        if (e.n4 < NT / sp) { while (sp < 64 && (long long)(NT / (sp * 2)) >= e.n4 && sp * 2 <= nsplit) sp *= 2; }   // tiny tensors: more partitions, fewer idle lanes
        e.sp = sp;
        e.blk0 = blk;
        blk += (unsigned)((e.n4 + NT / sp - 1) / (NT / sp));
    };
    for (int i = 0; i < n; ++i) {
        const ColvoWgradSlabs& w = sets[i];
        COLVO_CHECK_ARG(w.scratch && w.dw && w.nsplit >= 1 && w.Cout >= 8 && w.Cout % 4 == 0 && w.Ctot >= 8 && w.Ctot % 4 == 0,
                        "colvo_wgrad_reduce_group: bad set %d", i);
        const long long wsize = (long long)w.Cout * 9 * w.Ctot;
        const float* slabs = (const float*)w.scratch;
        COLVO_CHECK_ARG(((uintptr_t)slabs % 16) == 0 && ((uintptr_t)w.dw % 16) == 0 && (!w.db || ((uintptr_t)w.db % 16) == 0),
                        "colvo_wgrad_reduce_group: set %d is not 16-byte aligned", i);
        add(slabs, w.dw, wsize, w.nsplit);
        if (w.db) add(slabs + (size_t)w.nsplit * wsize, w.db, w.Cout, w.nsplit);
    }
    colvo::launch(k_wgrad_reduce_group, dim3(blk), dim3(NT), 0, (hipStream_t)stream, g);
    COLVO_CHECK_LAUNCH("k_wgrad_reduce_group");
    return 0;
}

extern "C" size_t colvo_conv_wgrad_scratch_bytes(const ColvoConvDesc* d) {
    if (!d || check_desc(d, "colvo_conv_wgrad_scratch_bytes")) return 0;
    // the launch planner itself, in plan-only mode (no pointer is dereferenced); a batch the call would slice (tensors >= 1 GiB)
    // is planned per slice, every slice re-using the same scratch
    int nsplit = 0;
    static const char dummy = 0;
    if (wgrad_impl(d, &dummy, d->C1 ? &dummy : nullptr, &dummy, (float*)&dummy, nullptr, nullptr, 0, &nsplit, nullptr)) return 0;
    return (size_t)nsplit * ((size_t)d->Cout * 9 * (d->C0 + d->C1) + d->Cout) * 4;
}

static int wgrad_impl(const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw, float* db,
                      void* scratch, size_t scratch_bytes, int* plan_out, colvo_stream_t stream, int slabs_only, int clean) {
    if (int e = check_desc(d, "colvo_conv_wgrad")) return e;
    COLVO_CHECK_ARG(x0 && dy && dw && (d->C1 == 0 || x1), "colvo_conv_wgrad: null pointer argument");
    // The kernel addresses dY and the sources with 32-bit buffer offsets (< 1 GiB per tensor): larger batches are
    // processed in image slices -- the gradient accumulates into dw / db anyway.
    {
        const long long es = d->dtype == COLVO_F32 ? 4 : 2;
        const long long cmax = d->C0 > d->C1 ? d->C0 : d->C1;
        const long long per_img = std::max((long long)d->Ho * d->Wo * d->Cout, (long long)d->Hi * d->Wi * cmax) * es;
        COLVO_CHECK_ARG(per_img < 0x40000000LL, "colvo_conv_wgrad: a single image of %lld bytes is not supported", per_img);
        const int bmax = (int)std::max(1LL, (0x40000000LL - 1) / per_img);
        if (d->B > bmax) {
            COLVO_CHECK_ARG(!slabs_only, "colvo_conv_wgrad_slabs: batch %d would be sliced (tensors >= 1 GiB); use colvo_conv_wgrad_det", d->B);
            const long long e0 = (long long)(d->up0 ? (d->Hi / 2) * (d->Wi / 2) : d->Hi * d->Wi) * d->C0 * es;
            const long long e1 = (long long)(d->up1 ? (d->Hi / 2) * (d->Wi / 2) : d->Hi * d->Wi) * d->C1 * es;
            const long long ey = (long long)d->Ho * d->Wo * d->Cout * es;
            for (int b0 = 0; b0 < d->B; b0 += bmax) {
                ColvoConvDesc sub = *d;
                sub.B = std::min(bmax, d->B - b0);
                int plan = 0;
                if (int e = wgrad_impl(&sub, (const char*)x0 + b0 * e0, x1 ? (const char*)x1 + b0 * e1 : nullptr,
                                       (const char*)dy + b0 * ey, dw, db, scratch, scratch_bytes, plan_out ? &plan : nullptr, stream, 0,
                                       b0 == 0 ? clean : 0))        // (only the first slice finds the arena clean)
                    return e;
                if (plan_out) *plan_out = std::max(*plan_out, plan);
            }
            return 0;
        }
    }
    WgradK k{};
    fill_gather(d, x0, d->C1 ? x1 : nullptr, k.g);
    k.Ho = d->Ho; k.Wo = d->Wo; k.B = d->B;
    k.dy = (const char*)dy; k.Cout = d->Cout; k.dw = dw; k.Ctot = d->C0 + d->C1; k.db = db;
    k.scratch = (const char*)scratch; k.scratch_bytes = scratch ? (long long)scratch_bytes : 0; k.plan_out = plan_out;
    k.slabs_only = slabs_only;
    k.clean = clean;
    // single up-sampled source in whole 32-channel (bf16) / 16-channel (f32) chunks: the four-class form over source positions
    const int ck = d->dtype == COLVO_F32 ? 16 : 32;
    const bool up2_form = TUNE(wgrad_up2) && d->up0 && d->C1 == 0 && d->stride == 1 && d->C0 % ck == 0 && d->Cout >= 16;
    {
        // bf16 stride-1 layers: the register-tiled form (csrc/wgrad_rt.hip); the ways out are set up here exactly as for the kernels below
        WgradRtPlan rp;
        if ((!up2_form || TUNE(wgrad_rt_over_up2)) && wgrad_rt_plan(d, rp)) {
            { int err; if (wgrad_prepare(k, rp.nsplit, &err)) return err; }
            form_hit(FORM_WGRAD_RT);
            if (int e = wgrad_rt_launch(rp, d, x0, d->C1 ? x1 : nullptr, dy, dw, db, k.slabs, k.db_slabs, k.det, (hipStream_t)stream)) return e;
            return wgrad_finish(k, rp.nsplit, (hipStream_t)stream);
        }
    }
    {
        if (up2_form) {
            const int Hs = d->Hi / 2, Ws = d->Wi / 2;
            const Tile t = pick_tile(Hs, Ws, 1, false);
            if ((t.toh + 2) * (t.tow + 2) * 4 <= 3 * NT) {
                k.toh = t.toh; k.tow = t.tow;
                k.tiles_x = (Ws + t.tow - 1) / t.tow; k.tiles_y = (Hs + t.toh - 1) / t.toh;
                k.ntiles = d->B * k.tiles_x * k.tiles_y;
                k.m_tow = mdiv_magic(t.tow); k.m_pw = mdiv_magic(t.tow + 2);
                k.pwl = wgrad_row_pitch(t.tow + 2, t.tow);
                hipStream_t s = (hipStream_t)stream;
                const bool wide = d->Cout >= 32;
                if (d->dtype == COLVO_F32) return wide ? launch_wgrad_up2<float, 2>(k, s) : launch_wgrad_up2<float, 1>(k, s);
                return wide ? launch_wgrad_up2<bf16_t, 2>(k, s) : launch_wgrad_up2<bf16_t, 1>(k, s);
            }
        }
    }
    const Tile t = pick_tile(d->Ho, d->Wo, d->stride, false);
    k.toh = t.toh; k.tow = t.tow;
    k.tiles_x = (d->Wo + t.tow - 1) / t.tow; k.tiles_y = (d->Ho + t.toh - 1) / t.toh;
    k.ntiles = d->B * k.tiles_x * k.tiles_y;
    k.m_tow = mdiv_magic(t.tow); k.m_pw = mdiv_magic((t.tow - 1) * d->stride + 3);
    k.pwl = wgrad_row_pitch((t.tow - 1) * d->stride + 3, t.tow);
    return d->dtype == COLVO_F32 ? launch_wgrad_t<float>(k, (hipStream_t)stream)
                                 : launch_wgrad_t<bf16_t>(k, (hipStream_t)stream);
}
