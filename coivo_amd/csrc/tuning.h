// tuning.h -- THE table of dispatch thresholds and A/B switches of libcolvo (gfx950).
//
// Every choice between kernel forms is a function of the layer's geometry and of the grid it would launch; the constants of those
// functions live here, once, with the measurement they came from.  Production processes use the defaults below and read NO
// environment variable.  Developers change an entry in one of two ways:
//   * COLVO_DEV=1 in the environment when the library is loaded: every entry may then be overridden by COLVO_<UPPER-CASE NAME>
//     (read once, at first use) -- how tools/*.sh run their A/B comparisons;
//   * colvo_tune_set("name", value) at run time (include/colvo.h) -- how the tests reach the large-grid kernel forms with small
//     shapes; colvo_tune_get reads an entry.
// Defaults were measured on MI355X inside the training step at BASELINE configs[1] (16 DepthNet images) and re-checked this round
// at 64 / 128 images of 256x320 (the per-GPU shapes of configs[3] / [4]) and at 64 images of 512x640 (configs[2]):
// profiles/r3_tuning_check.md.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace colvo {
namespace tune {

// name, default, what it decides
#define COLVO_TUNE_TABLE(X)                                                                                                        \
    /* ---- forward / input-gradient kernels (conv.hip): grids are counted in 256-thread workgroups ---- */                        \
    X(bn64_min_wgs, 4096, "64-wide output-channel tiles only when that grid still has this many workgroups, else 32-wide "           \
                          "(round 4, in the step: 1024 -> 4096 -0.8 % at 32 pairs, level at 8 and 64)")                              \
    X(bn32_min_wgs, 0, "32-wide tiles only above this many workgroups, else 16-wide (0: measured neutral)")                        \
    X(lone_max_wgs, 1024, "grids up to this size take the two-chunk register ring (~1 workgroup per CU, one wave per SIMD)")       \
    X(depth2_min_chunks, 8, "... if the layer has at least this many 32-channel chunks")                                           \
    X(res_wg_per_cu, 4, "weights-resident persistent kernel (single-chunk layers): workgroups per CU")                              \
    X(res_min_tiles, 2048, "... selected from this many 128-pixel tiles on")                                                        \
    X(res_s2, 1, "... its stride-2 form for single-chunk stride-2 layers (enc2a, PoseNet conv2 / conv3; round 5)")                   \
    X(res_s2_min_tiles, 512, "... selected from this many 128-pixel tiles on (enc2a: 16 frames = 640 tiles 15.4 -> 13.3 us, 64 frames "   \
                             "55.3 -> 35.7 us; MIOpen 13.3 / 32.4)")                                                                    \
    X(wide, 0, "512-thread / 256-pixel tiles: 10-20 % slower at 16 images, pays at several workgroups per CU; off")                 \
    X(wide_min_wgs, 192, "... minimum grid when on")                                                                                \
    X(wide_min_chunks, 2, "... minimum chunk count when on")                                                                        \
    X(conv_up2, 1, "forward over an up-sampled source: four output pixels per source position (k_conv_up2)")                        \
    X(up2_min_chunks, 1, "... from this many chunks on")                                                                            \
    X(up2_bn16_max_wgs, 640, "... with 16-wide channel tiles while the 32-wide grid would have at most this many workgroups")       \
    X(dgrad_s2, 1, "parity-decomposed input gradient of stride-2 layers (k_dgrad_s2)")                                              \
    X(dgrad_up2, 1, "input gradient of up-sampled layers with the 2x2 sum-pool in the K loop (k_dgrad_up2)")                        \
    X(dgrad_up2_min_wgs, 640, "... only where its grid has this many workgroups (1/8, 1/16 resolution lose 1-2 us at 16 images)")  \
    X(dgrad_both, 1, "both sources' input gradients of a concat layer in one launch")                                               \
    X(conv_quad, 1, "quad-tile kernel for ordinary stride-1 layers (k_conv_q)")                                                     \
    X(quad_min_wgs, 8192, "... from this many quad-tile workgroups on (iconv2 at configs[2] size 370 -> 275 us; at 2560-5120 "      \
                          "workgroups of 256x320 frames it loses 3-6 % of the pass)")                                               \
    X(quad_max_chunks, 4, "... and at most this many chunks (deeper layers keep the two-chunk ring)")                               \
    X(conv_rt, 1, "register-tiled stride-1 kernel (k_conv_rt, csrc/conv_rt.hip: 16x16 pixels x 64/32 channels, 4x4 fragments per wave)") \
    X(rt_min_wgs, 1024, "... from this many of its workgroups on (64 frames, us fwd / dgrad, one-tile -> this: enc2b 42.4 / 52.6 -> 35.6 / "  \
                       "41.9, iconv3 65.9 / 52.3 -> 52.8 / 42.3, iconv4 61.8 / 43.6 -> 51.9 / 35.4, enc3b 36.6 / 43.4 -> 32.3 / 35.5, "   \
                       "enc4b 37.2 / 42.1 -> 35.4 / 38.2; at 16 frames its 128-320-workgroup grids lose 0.5-1.8 us per layer; 512 -> 1024: "   \
                       "the 640 workgroups of iconv3's two input gradients at 16 frames = 1.25 rounds of the 512 slots, step -0.6...1 % "  \
                       "at 8 pairs, level at 32 / 64 where it moves iconv4's 768)")                                                    \
    X(rt_bn32_min_wgs, 2048, "... the 32-channel form (iconv2 forward) from this many on: 75.9 -> 59.5 us at 64 frames (5120 workgroups), "  \
                             "19.0 -> 18.8 at 16 (1280) with the step 0.7 % slower")                                                    \
    X(rt_min_fill_pct, 60, "... and only where the 16x16 tiling covers at most 100/this times the image (32x40 maps: 83 %, 16x20: 62 %)") \
    X(rt_min_chunks, 2, "... and the layer has at least this many 32-channel chunks")                                                 \
    X(rt_wgs_per_cu, 0, "... 0: one tile per workgroup (default).  n > 0: persistent grid of n workgroups per CU (+1 for the 32-channel "  \
                        "form), each walking consecutive tiles with the next tile's loads in flight under the current tile's MFMAs -- "  \
                        "measured SLOWER at 64 frames (us, one tile -> persistent: enc2b 35.7 -> 39.4, enc3b 32.6 -> 39.3, iconv4 52.0 "  \
                        "-> 64.0, iconv3 52.0 -> 57.6, iconv2 55.7 -> 59.1; 32 pairs 3.40 -> 3.45 ms per step): a static split of "      \
                        "768-1280 tiles over 512 slots leaves a quarter of them idle where the hardware's own dispatcher fills them")     \
    X(rt_tiles_per_wg, 0, "... tests: force this many consecutive tiles per workgroup (0: from rt_wgs_per_cu)")                       \
    X(xcd_remap, 1, "XCD-contiguous 1-D grids (step +4.3 % without)")                                                               \
    X(lds_aware_tiles, 1, "tile shapes / padded LDS row pitch chosen against ds_read_b128 bank conflicts")                          \
    X(lds_tile_max_pad, 8, "... at most this many padding pixels per patch row")                                                    \
    /* ---- weight gradient (wgrad.hip) ---- */                                                                                     \
    X(wgrad_up2, 1, "weight gradient of up-sampled layers in the four-class form (16 instead of 36 products, k_wgrad_up2)")          \
    X(wgrad_mt_max, 2, "output-channel tile = 16 x this (32-wide measured better than 64 on every layer at 16 images)")             \
    X(wgrad_mt4_min_walk, 16, "... but 64-wide from this many pixel tiles per workgroup (at a 256-workgroup grid) on "              \
                               "(24 -> 16 in the step: -0.4 % at 32 pairs, level elsewhere)")                                        \
    X(wgrad_ng_max, 4, "16-byte channel granules per chunk")                                                                        \
    X(wgrad_one_chunk_rule, 1, "16-wide tiles where the input is a single chunk (enc1a/enc1b/enc2a 27/31/26 -> 20/24/24 us)")       \
    X(wgrad_atomic_mb, 3, "cap on the fp32 atomic traffic of a launch, MB (12 MB 592 us, 6 MB 575, 3 MB 566 over the stack)")       \
    X(wgrad_wg_lo, 256, "... but at least this many workgroups")                                                                    \
    X(wgrad_wg_hi, 1024, "... and at most this many")                                                                               \
    X(wgrad_short_walk, 32, "... half the floor while a workgroup of the full grid would walk fewer pixel tiles than this (0: off)")                                                                               \
    X(wgrad_store_clean, 1, "single-split layers STORE their sums when the caller vouches for a zero arena (colvo_conv_wgrad_clean)")    \
    X(wgrad_rt, 0, "register-tiled weight gradient (k_wgrad_rt, csrc/wgrad_rt.hip: consumer waves own 32 co x 32 ci x 9 taps on 32x32x16 MFMAs "  \
                   "and split K four ways, loader waves stage, ONE 36 KB slab of atomics per workgroup) for bf16 stride-1 layers.  OFF: level "  \
                   "with k_wgrad3x3 alone (64 frames 1287 vs 1290 us over the stack, deep layers -10...17 %, mid layers +5...14 %), and in "   \
                   "the step +2.6 % / +0.5 % / +0.2 % at 8 / 32 / 64 pairs even on the layers it wins alone (its 150 KB of LDS keep the "      \
                   "input-gradient chain off its CUs): profiles/r6_wgrad_rt.md")                                                          \
    X(wgrad_rt_min_c, 32, "... whose sources and outputs have at least this many channels (a 32-wide tile of a 16-channel tensor is half empty)") \
    X(wgrad_rt_min_px, 0, "... and at least this many output pixels in the batch")                                                       \
    X(wgrad_rt_wgs, 256, "... grid: pixel-range splits up to this many workgroups (one 512-thread workgroup per CU)")                     \
    X(wgrad_rt_max_px, 640, "... tile: at most this many pixels (K-steps of 16; two staging buffers of tile + patch must fit 156 KB of LDS)")  \
    X(wgrad_rt_over_up2, 0, "... also for single up-sampled sources (instead of the four-class kernel k_wgrad_up2)")                     \
    X(wgrad_teams, 4, "pixel-tile teams per workgroup on the full-resolution layers (1 = off)")                                     \
    X(wgrad_team_max_slabs, 2, "... for layers with at most this many (co tile, chunk) slabs")                                      \
    X(wgrad_team_wgs, 256, "... grid size of the team form")                                                                        \
    X(bwd16, 1, "input + weight gradient of the 16 -> 16 full-resolution layer in one pass (k_bwd16, csrc/bwd16.hip)")                \
    X(bwd16_wgs, 768, "... its grid: every workgroup ends with 2320 atomics on the same addresses")                                  \
    X(head_wgrad_wgs, 1024, "grid of k_head_wgrad_mfma (one partial row per workgroup)") \
    X(planes_mfma, 1, "input gradient of PoseNet's first layer w.r.t. the two depth channels by MFMA (k_dgrad_planes_s2_mfma)") \
    X(planes_groups, 2, "... 16-block groups a wave walks (its weight set-up is amortised over them)") \
    X(fwd16, 1, "the 16 -> 16 full-resolution layer and the depth head behind it in one pass (k_fwd16_head, csrc/fwd16.hip)")          \
    X(fwd16_wgs, 1024, "... its grid (at least; one workgroup per 8 tiles beyond that)")                                              \
    X(fwd16_tiles_per_wg, 32, "... tiles per workgroup beyond fwd16_wgs (grid in whole rounds of 1024)")                                       \
    /* ---- heads, fused loss, streams ---- */                                                                                      \
    X(head_wgrad_rows, 1, "tap rows per thread of the depth-head weight gradient (1: three workgroups per pixel range)")            \
    X(head_fwd_lds, 1, "depth-head forward stages its tile in LDS (24.7 -> 18.5 us)")                                               \
    X(head_dgrad_generic, 0, "depth-head input gradient: generic form instead of the 16-channel granule form")                      \
    X(march_rows_fwd, 0, "rows per strip segment of the fused-loss forward march (0: chosen to fill the wave slots in whole rounds)") \
    X(march_rows_bwd, 0, "... of the one-pass loss + gradient march")                                                               \
    X(fork_stop_event, 1, "a FORK waits on the producing kernel's own completion (hipExtLaunchKernel stopEvent) instead of a recorded marker") \
    X(side_streams, 2, "weight-gradient streams colvo_run_commands alternates between (the caller's + library-owned ones)")

struct Table {
#define X(name, dflt, doc) double name = dflt;
    COLVO_TUNE_TABLE(X)
#undef X
};

inline bool dev_mode() {
    static const bool v = [] { const char* e = getenv("COLVO_DEV"); return e && atoi(e) != 0; }();
    return v;
}

inline Table& table() {
    static Table t = [] {
        Table x;
        if (dev_mode()) {
            char env[96];
#define X(name, dflt, doc)                                                        \
    {                                                                             \
        snprintf(env, sizeof env, "COLVO_%s", #name);                             \
        for (char* c = env; *c; ++c) if (*c >= 'a' && *c <= 'z') *c -= 32;       \
        if (const char* e = getenv(env)) x.name = atof(e);                        \
    }
            COLVO_TUNE_TABLE(X)
#undef X
        }
        return x;
    }();
    return t;
}

inline double* find(const char* name) {
    Table& t = table();
#define X(n, dflt, doc) if (!strcmp(name, #n)) return &t.n;
    COLVO_TUNE_TABLE(X)
#undef X
    return nullptr;
}

#define TUNE(name) ((long)::colvo::tune::table().name)
#define TUNE_F(name) (::colvo::tune::table().name)

}  // namespace tune
}  // namespace colvo
