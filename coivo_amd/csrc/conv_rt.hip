// conv_rt.hip -- a1/a2 of SURVEY.md §8: the register-tiled form of the stride-1 3x3 convolution (forward AND input gradient)
// for grids that cover the chip several times (the data-parallel shapes of BASELINE configs[3] / [4] and configs[2]).
//
// Concept: /root/reference/README.md:5,7 (DCDP depth + pose networks); results specified by oracle/colvo_spec.py (F.conv2d k=3 pad=1
// + bias + ReLU, skip concat in front of the decoder convs).  Same arithmetic, operand layouts and epilogue as k_conv3x3 (conv.hip).
//
// Why (profiles/r5_conv_pmc_b64.json, 64 frames of 256x320): the one-tile kernel k_conv3x3<bf16, 32, 4> gives a wave 2 pixel
// fragments x 2 channel fragments -- one ds_read_b128 per MFMA, two workgroup barriers per 36 MFMAs of a wave -- and sits at
// 24-32 % MFMA-pipe busy with the LDS index unit active 37-48 % of the time and the waves waiting (barrier / waitcnt) 40 % of
// their cycles, while k_conv_up2 (0.47 fragment reads per MFMA, 144 MFMAs per wave between barriers) reaches 50-55 % on the same
// grids.  This kernel gives ordinary stride-1 layers that structure:
//   * a workgroup = 16 x 16 output pixels x BN = 64 (32) output channels; wave w owns tile rows 4w .. 4w+3, i.e. FOUR pixel
//     fragments (one image row of 16 pixels each) x NF = 4 (2) channel fragments: 16 (8) accumulators;
//   * per 32-channel chunk the 18 x 18 patch and the [BN][9][32] weight slab are staged once (global -> registers -> LDS, the next
//     chunk's loads in flight under the MFMAs) for 144 (72) MFMAs per wave between barriers;
//   * a pixel fragment is one ROW of the patch, so the fragment of output row r for tap row ky is patch row r + ky: for one tap
//     column kx the wave reads the SIX patch rows it touches once and uses each for every (r, ky) with r + ky = row -- 18 patch
//     reads per chunk instead of 36 -- and streams the weight fragments (one read per 4 MFMAs): 54 ds_read_b128 per 144 MFMAs
//     = 0.375 per MFMA (BN 32: 36 per 72 = 0.5) against 1.0 in k_conv3x3<32> and 0.75 in k_conv3x3<64>;
//   * 16 consecutive pixels at the 96-byte pixel pitch and 16 weight rows at the 608-byte row pitch are both conflict-free
//     ds_read_b128 groups (conv_common.h pitch_slots);
//   * epilogue straight from the accumulators (operands swapped: a lane holds 4 consecutive channels of one pixel): bias, ReLU,
//     producer's ReLU mask, accumulate, two-output form of colvo_conv_dgrad_both.
// Selected by try_launch_conv_rt for direct (not up-sampled / dilated) sources in whole 32-channel chunks when the 16 x 16 tiling
// wastes little of the image and the grid has at least rt_min_wgs workgroups (tuning.h).
#define COLVO_ACC_CONSTRAINT "+v"     // built with -mllvm -amdgpu-mfma-vgpr-form (coivo_amd/build.py)
#include "conv_common.h"
#include "conv_stage.h"

namespace colvo {
namespace {

constexpr int RT_TH = 16, RT_TW = 16;          // output tile
constexpr int RT_PH = RT_TH + 2, RT_PW = RT_TW + 2;
constexpr int RT_ROWS = 4;                      // tile rows (= pixel fragments) per wave
constexpr int RT_PPF = (RT_PH * RT_PW * 4 + NT - 1) / NT;      // staged patch granules per thread: 1296 / 256 -> 6

// One MFMA phase of a chunk: 3 tap columns x (3 tap rows x NF channel fragments) steps of RT_ROWS MFMAs.  Everything a step needs is
// requested ahead: the weight fragment of step t + BD before the MFMAs of step t (BD + 1 registers), the six patch rows of tap
// column kx + 1 under the last steps of column kx (two sets) -- ds_read latency (~100 cycles) is longer than the 64 MFMA cycles of
// a step, so a read issued and awaited inside its own step leaves the matrix pipe idle every step.
template <typename T, int NF>
__device__ __forceinline__ void rt_mfma_phase(const char* pA, const char* pB, f32x4 (&acc)[RT_ROWS][NF]) {
    constexpr int ES = TT<T>::ES, NG = 4;
    constexpr int WROW = wrow_bytes(36), PIXP = pitch_bytes(NG * 16);
    constexpr int SPC = 3 * NF;                        // steps per tap column
    // weight fragments in flight ahead of the step that consumes them (NF 2: one, which keeps the kernel within three waves per SIMD)
    constexpr int BD = NF >= 4 ? 2 : 1;
    u32x4 av[2][RT_ROWS + 2], bv[BD + 1];
    auto read_a = [&](int kx, u32x4 (&dst)[RT_ROWS + 2], int j) { dst[j] = ld16(pA + (j * RT_PW + kx) * PIXP); };
    auto read_b = [&](int t) -> u32x4 {                // step t = (kx, ky, nf), nf fastest
        const int kx = t / SPC, ky = (t - kx * SPC) / NF, nf = t - kx * SPC - ky * NF;
        return ld16(pB + nf * 16 * WROW + (ky * 3 + kx) * NG * 16);
    };
#pragma unroll
    for (int j = 0; j < RT_ROWS + 2; ++j) read_a(0, av[0], j);
#pragma unroll
    for (int t = 0; t < BD; ++t) bv[t] = read_b(t);
#pragma unroll
    for (int t = 0; t < 3 * SPC; ++t) {
        const int kx = t / SPC, u = t - kx * SPC, ky = u / NF, nf = u - ky * NF;
        if (t + BD < 3 * SPC) bv[(t + BD) % (BD + 1)] = read_b(t + BD);
        // the next column's patch rows, spread over this column's last RT_ROWS + 2 steps
        if (kx < 2 && u >= SPC - (RT_ROWS + 2)) read_a(kx + 1, av[(kx + 1) & 1], u - (SPC - (RT_ROWS + 2)));
        // (pinned: left alone, hipcc's scheduler sinks every read to just in front of its first use -- "read, wait, 4 MFMAs")
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 bw = bv[t % (BD + 1)];
        // operands swapped (A = weights, B = pixels): a lane's accumulator is 4 consecutive channels of one pixel
        if constexpr (ES == 2) {
#pragma unroll
            for (int r = 0; r < RT_ROWS; ++r)
                acc[r][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(bf16x8, bw), __builtin_bit_cast(bf16x8, av[kx & 1][r + ky]), acc[r][nf], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < RT_ROWS; ++r)
                    acc[r][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        __uint_as_float(bw[j]), __uint_as_float(av[kx & 1][r + ky][j]), acc[r][nf], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Workgroup w walks the logical tiles [w * tiles_per_wg, ...) -- (image, tile row, tile column, channel tile), channel tile fastest --
// as ONE sequence of (tile, chunk) items; the loads of the next item are in flight under the MFMAs of the current one ACROSS tile
// boundaries too, and a tile's epilogue runs under the next tile's first loads.  tiles_per_wg is 1 in production: the persistent
// form (a workgroup per resident slot, several tiles each) was built on the idea that a workgroup's exposed first-chunk loads --
// half of all staging loads on the two-chunk layers -- make the kernel alternate between an HBM phase and an MFMA phase, and measured
// 6-23 % SLOWER on every layer it applies to (tuning.h rt_wgs_per_cu): the hardware dispatcher already overlaps one workgroup's load
// phase with its neighbour's MFMA phase, and it balances 768-1280 tiles over 512 slots better than a static split does.  The walk
// stays (tests force it through rt_tiles_per_wg) -- it costs the one-tile case nothing (enc2b 35.7 us in both forms).
template <typename T, int NF>
__global__ __launch_bounds__(NT, (NF == 2 && TT<T>::ES == 2) ? 3 : 2) void k_conv_rt(const ConvK a, int tiles_total, int tiles_per_wg) {
    constexpr int G = TT<T>::G, ES = TT<T>::ES;
    constexpr int NG = 4, CK = NG * G;
    constexpr int BN = 16 * NF;
    constexpr int WROW = wrow_bytes(36);            // 9 taps x 4 granules, padded to a conflict-free row pitch
    constexpr int PIXP = pitch_bytes(NG * 16);
    constexpr int PPF = RT_PPF;
    constexpr int PTOTAL = RT_PH * RT_PW * NG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sP = smem + BN * WROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int lw = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x, a.xcd));
    int lt = lw * tiles_per_wg;
    const int lt_end = min(tiles_total, lt + tiles_per_wg);
    if (lt >= lt_end) return;                       // (uniform; the host sizes the grid so that every workgroup has a tile)

    const int C0 = a.g.C[0], C1 = a.g.C[1];
    const int nch0 = C0 / CK, nch = nch0 + C1 / CK;
    const int Hs = a.g.Hi, Ws = a.g.Wi;              // both sources are stored at the conv input's extent (direct mode)
    const int tapB = a.Ctot * ES;                    // bytes from tap to tap inside a weight row

    // wave-uniform tile coordinates, advanced incrementally (channel tile fastest)
    struct TileC { int b, ty, tx, nt; };
    TileC cur;
    {
        const int tlin = lt / a.ntn;
        cur.nt = lt - tlin * a.ntn;
        const int tpi = a.tiles_x * a.tiles_y;
        cur.b = tlin / tpi;
        const int trem = tlin - cur.b * tpi;
        cur.ty = trem / a.tiles_x;
        cur.tx = trem - cur.ty * a.tiles_x;
    }
    auto tile_next = [&](TileC c) -> TileC {
        if (++c.nt == a.ntn) { c.nt = 0; if (++c.tx == a.tiles_x) { c.tx = 0; if (++c.ty == a.tiles_y) { c.ty = 0; ++c.b; } } }
        return c;
    };

    // descriptors over the WHOLE tensors (the host guarantees < 1 GiB each): offsets >= the size (OOB_OFF) read as zero
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, a.N * 9 * a.Ctot * ES, 0x00020000);
    const int nimg = tiles_total / (a.ntn * a.tiles_x * a.tiles_y);
    const __amdgpu_buffer_rsrc_t rimg0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.g.src[0], 0, nimg * Hs * Ws * C0 * ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rimg1 = __builtin_amdgcn_make_buffer_rsrc((void*)(C1 > 0 ? a.g.src[1] : a.g.src[0]), 0,
                                                                           C1 > 0 ? nimg * Hs * Ws * C1 * ES : 0, 0x00020000);
    // tile-invariant per-thread staging state: (patch row, patch column) and LDS offset of each staged patch granule; weight granule
    // i = it * 256 + tid of the slab [BN][9][CK] = one per-thread offset plus a scalar stride per staged granule (see k_conv3x3)
    typedef SlabStage<T, BN, NG> Slab;
    constexpr int WIT = Slab::WIT, WTOT = Slab::WTOT, NGR = Slab::NGR;
    int p_yx[PPF];
#pragma unroll
    for (int it = 0; it < PPF; ++it) {
        const int i = it * NT + tid;
        const int pix = i / NG;
        const int py = pix / RT_PW, px = pix - py * RT_PW;
        p_yx[it] = (i < PTOTAL) ? ((py << 16) | px) : (0x4000 << 16);      // 0x4000: a row that is never inside an image
    }
    // (NT is a multiple of NG: a thread's granules all have cg = tid % NG, and granule it * 256 + tid sits 64 * it patch pixels on)
    const int cgoff = (tid & (NG - 1)) * 16;
    const int plds0 = (tid / NG) * PIXP + cgoff;
    int wthr;                                           // this thread's weight granule relative to channel tile 0
    bool wlast;
    {
        const int n = tid / NGR, gi = tid - n * NGR;
        const int tap = gi / NG, cg = gi - tap * NG;
        wthr = (n * 9 + tap) * tapB + cg * 16;
        wlast = (WIT - 1) * NT + tid < WTOT;
    }
    u32x4 wv[WIT], pv[PPF];
    // all loads of item (tile c, chunk k); dead (0 or OOB_OFF): past the last item -- issued all the same, reads zeros, no branch
    auto load_item = [&](const TileC& c, int k, int dead) {
        const int woff = wthr + c.nt * BN * 9 * tapB;     // rows beyond N fall outside the descriptor and read as zero
        const int wso = k * CK * ES;
#pragma unroll
        for (int it = 0; it < WIT; ++it)
            wv[it] = bld16(rw, (((it == WIT - 1 && !wlast) ? OOB_OFF : woff) + it * (NT / NG) * tapB) | dead, wso);
        const bool second = k >= nch0;
        const int pixB = (second ? C1 : C0) * ES;
        const int pso = (second ? k - nch0 : k) * CK * ES;
        const int iy0 = c.ty * RT_TH - 1, ix0 = c.tx * RT_TW - 1, ibase = c.b * Hs;
#pragma unroll
        for (int it = 0; it < PPF; ++it) {
            const int vy = iy0 + (p_yx[it] >> 16), vx = ix0 + (p_yx[it] & 0xffff);
            const bool inb = ((unsigned)vy < (unsigned)Hs) && ((unsigned)vx < (unsigned)Ws);
            const int off = (inb ? ((ibase + vy) * Ws + vx) * pixB + cgoff : OOB_OFF) | dead;
            pv[it] = second ? bld16(rimg1, off, pso) : bld16(rimg0, off, pso);
        }
    };
    const __amdgpu_buffer_rsrc_t rbias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, 0x00020000);
    u32x4 biasv[NF];
    auto load_bias = [&](const TileC& c) {
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) biasv[nf] = bld16(rbias, (c.nt * BN + nf * 16 + kg * 4) * 4, 0);   // zeros without a bias
    };

    load_item(cur, 0, 0);
    load_bias(cur);
    f32x4 acc[RT_ROWS][NF];
#pragma unroll
    for (int r = 0; r < RT_ROWS; ++r)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) acc[r][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment bases of this lane: patch pixel (4 wave + j, kx + l15), channel granule kg; weight row l15 of fragment nf, k-group = tap
    const char* pA = sP + ((RT_ROWS * wave) * RT_PW + l15) * PIXP + kg * 16;
    const char* pB = sW + l15 * WROW + kg * 16;
    typedef typename EV<T>::type V;

    int k = 0;
    for (;;) {
        __syncthreads();                    // the MFMAs of the previous item have finished reading LDS
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = it * NT + tid;                 // slab granule -> LDS row at the padded pitch (recomputed: registers are scarce)
            if (WTOT % NT == 0 || it < WIT - 1 || wlast) st16(sW + i * 16 + (i / NGR) * (WROW - NGR * 16), wv[it]);
        }
#pragma unroll
        for (int it = 0; it < PPF; ++it)
            if (it * NT + tid < PTOTAL) st16(sP + plds0 + it * (NT / NG) * PIXP, pv[it]);
        __syncthreads();
        // the next item: the tile's next chunk, or the first chunk of the workgroup's next tile
        const bool last_chunk = (k + 1 == nch);
        const TileC nxt = last_chunk ? tile_next(cur) : cur;
        const int nk = last_chunk ? 0 : k + 1;
        const bool more = !last_chunk || lt + 1 < lt_end;
        load_item(nxt, nk, more ? 0 : OOB_OFF);          // in flight during the MFMA phase (and the epilogue) below
        rt_mfma_phase<T, NF>(pA, pB, acc);
        k = nk;
        if (!last_chunk) continue;

        // ---- the tile is complete: epilogue straight from the accumulators ----
        mfma_result_guard<T>(reinterpret_cast<f32x4 (&)[RT_ROWS * NF]>(acc));
        {
            // two-output form (input gradient of a concat layer): this workgroup's channel tile lies in exactly one of the sources
            const int n0 = cur.nt * BN;
            char* out = a.out;
            const char* mask = a.mask;
            int N = a.N, n0e = n0;
            if (a.nsplit > 0) {
                const bool second = n0 >= a.nsplit;           // wave-uniform
                out = second ? a.out2 : a.out;
                mask = second ? a.mask2 : a.mask;
                N = second ? a.N - a.nsplit : a.nsplit;
                n0e = second ? n0 - a.nsplit : n0;
            }
            const int tot_bytes = nimg * a.Ho * a.Wo * N * ES;
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, tot_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rmask =
                __builtin_amdgcn_make_buffer_rsrc((void*)(mask ? mask : out), 0, mask ? tot_bytes : 0, 0x00020000);
            int noff[NF];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int n = n0e + nf * 16 + kg * 4;
                noff[nf] = (n < N) ? n * ES : OOB_OFF;         // out of range -> loads 0 / store dropped
            }
            const int oy0 = cur.ty * RT_TH, ox0 = cur.tx * RT_TW;
#pragma unroll
            for (int r = 0; r < RT_ROWS; ++r) {
                const int gy = oy0 + RT_ROWS * wave + r, gx = ox0 + l15;
                const int obase = (gy < a.Ho && gx < a.Wo) ? ((cur.b * a.Ho + gy) * a.Wo + gx) * N * ES : OOB_OFF;
                V pm[NF], pa[NF];
                if (mask) {
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) pm[nf] = epi_load<T>(rmask, obase + noff[nf], 0);
                }
                if (a.accumulate) {
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) pa[nf] = epi_load<T>(rout, obase + noff[nf], 0);
                }
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[r][nf][q] + __uint_as_float(biasv[nf][q]);
                    if (a.relu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.0f);
                    }
                    if constexpr (ES == 4) {
                        if (mask) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = (__uint_as_float(pm[nf][q]) > 0.0f) ? v[q] : 0.0f;
                        }
                        if (a.accumulate) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] += __uint_as_float(pa[nf][q]);
                        }
                        u32x4 o;
#pragma unroll
                        for (int q = 0; q < 4; ++q) o[q] = __float_as_uint(v[q]);
                        __builtin_amdgcn_raw_buffer_store_b128(o, rout, obase + noff[nf], 0, 0);
                    } else {
                        if (mask) {
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                // bf16 > 0  <=>  sign clear and magnitude non-zero
                                const uint32_t lo = pm[nf][q] & 0xFFFFu, hi = pm[nf][q] >> 16;
                                if (!(lo != 0 && lo < 0x8000u)) v[2 * q] = 0.0f;
                                if (!(hi != 0 && hi < 0x8000u)) v[2 * q + 1] = 0.0f;
                            }
                        }
                        if (a.accumulate) {
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                v[2 * q] += bf2f((uint16_t)(pa[nf][q] & 0xFFFFu));
                                v[2 * q + 1] += bf2f((uint16_t)(pa[nf][q] >> 16));
                            }
                        }
                        u32x2 o;
#pragma unroll
                        for (int q = 0; q < 2; ++q) o[q] = pack2bf(v[2 * q], v[2 * q + 1]);
                        __builtin_amdgcn_raw_buffer_store_b64(o, rout, obase + noff[nf], 0, 0);
                    }
                }
            }
        }
        if (++lt >= lt_end) break;
        cur = nxt;
        load_bias(cur);
#pragma unroll
        for (int r = 0; r < RT_ROWS; ++r)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) acc[r][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <typename T, int NF>
int launch_conv_rt(ConvK k, long long ntiles, hipStream_t s) {
    constexpr int BN = 16 * NF;
    constexpr size_t lds = (size_t)BN * wrow_bytes(36) + (size_t)RT_PH * RT_PW * pitch_bytes(64);
    static_assert(lds <= 80 * 1024, "k_conv_rt: two workgroups per CU need <= 80 KB of LDS each");
    static bool configured = false;        // per instantiation
    if (lds > 48 * 1024 && !configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_rt<T, NF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) { set_error("conv (register-tiled): hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
        configured = true;
    }
    // one tile per workgroup by default; rt_wgs_per_cu > 0: a persistent grid (256 CUs x n, + 1 for the 32-channel form), every
    // workgroup walking the same number of consecutive tiles (tuning.h has the measurement that keeps it off)
    const long per_cu = TUNE(rt_wgs_per_cu) > 0 ? TUNE(rt_wgs_per_cu) + (NF == 2 ? 1 : 0) : 0;
    long long slots = per_cu > 0 ? 256 * per_cu : ntiles;
    if (slots > ntiles) slots = ntiles;
    long long per_wg = (ntiles + slots - 1) / slots;
    if (TUNE(rt_tiles_per_wg) > 0) per_wg = std::min<long long>(TUNE(rt_tiles_per_wg), ntiles);
    const long long nwg = (ntiles + per_wg - 1) / per_wg;
    colvo::launch((k_conv_rt<T, NF>), dim3((unsigned)nwg), dim3(NT), (unsigned)lds, s, k, (int)ntiles, (int)per_wg);
    COLVO_CHECK_LAUNCH("k_conv_rt");
    return 0;
}

}  // namespace

int try_launch_conv_rt(const ConvK& k0, int B, int dtype, hipStream_t s) {
    if (!TUNE(conv_rt)) return -1;
    const Gather& g = k0.g;
    const int es = dtype == COLVO_F32 ? 4 : 2, ck = dtype == COLVO_F32 ? 16 : 32;
    if (g.stride != 1 || k0.pool2 || g.mode[0] != MODE_DIRECT || (g.C[1] > 0 && g.mode[1] != MODE_DIRECT)) return -1;
    if (g.C[0] % ck || g.C[1] % ck || k0.N <= 16) return -1;
    // single-chunk layers (enc1b, the input gradients of iconv2) stay with the weights-resident persistent kernel: they are bound
    // by their HBM traffic and it prefetches across tiles (64 frames: enc1b 35.4 -> 41.6 us here, iconv2 dgrad 47.6 -> 59.2)
    if ((g.C[0] + g.C[1]) / ck < TUNE(rt_min_chunks)) return -1;
    // (the kernel addresses whole tensors -- all images -- with 32-bit offsets and uses 1 GiB as its out-of-range mark)
    if ((long long)B * g.Hi * g.Wi * std::max(g.C[0], g.C[1]) * es >= 0x40000000LL || (long long)B * k0.Ho * k0.Wo * k0.N * es >= 0x40000000LL)
        return -1;
    ConvK k = k0;
    k.toh = RT_TH; k.tow = RT_TW; k.pwp = RT_PW;
    k.tiles_x = (k.Wo + RT_TW - 1) / RT_TW; k.tiles_y = (k.Ho + RT_TH - 1) / RT_TH;
    k.m_tow = mdiv_magic(RT_TW); k.m_pw = mdiv_magic(RT_PW);
    // the 16 x 16 tiling must not waste much of the image (a 32 x 40 map computes 1.2 x its pixels, a 16 x 20 map 1.6 x)
    const long long covered = (long long)k.tiles_x * k.tiles_y * RT_TH * RT_TW;
    if ((long long)k.Ho * k.Wo * 100 < covered * TUNE(rt_min_fill_pct)) return -1;
    // 64-wide channel tiles where the layer has them (two-output form: a tile must not straddle the two sources)
    const bool bn64 = k.N >= 64 && (k.nsplit == 0 || k.nsplit % 64 == 0);
    if (k.nsplit % 32 != 0) return -1;
    const int bn = bn64 ? 64 : 32;
    k.ntn = (k.N + bn - 1) / bn;
    k.xcd = (int)TUNE(xcd_remap);
    const long long nwg = (long long)k.tiles_x * k.tiles_y * B * k.ntn;
    if (nwg < (bn64 ? TUNE(rt_min_wgs) : TUNE(rt_bn32_min_wgs)) || nwg >= (1ll << 30)) return -1;
    form_hit(FORM_CONV_RT);
    if (dtype == COLVO_F32) return bn64 ? launch_conv_rt<float, 4>(k, nwg, s) : launch_conv_rt<float, 2>(k, nwg, s);
    return bn64 ? launch_conv_rt<bf16_t, 4>(k, nwg, s) : launch_conv_rt<bf16_t, 2>(k, nwg, s);
}

}  // namespace colvo
