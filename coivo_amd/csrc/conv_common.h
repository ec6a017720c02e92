// conv_common.h -- what the forward / input-gradient kernels (conv.hip) and the weight-gradient kernel (wgrad.hip) share:
// element types, LDS pitches, the gather description, buffer-load helpers, the accumulator hand-over (mfma_result_guard),
// tile selection and descriptor checks.  Each translation unit defines COLVO_ACC_CONSTRAINT before including this file:
// "+v" where the unit is compiled with -mllvm -amdgpu-mfma-vgpr-form (accumulators in VGPRs), "+a" where hipcc keeps
// them in AGPRs -- either way the guard's asm statement must not make hipcc copy (= read) the results in front of it.
#pragma once
#include <algorithm>
#include <mutex>
#include <stdlib.h>
#include <utility>
#include <vector>

#include "common.h"
#include "tuning.h"

namespace colvo {

// (named namespace: the descriptor crosses translation units -- conv.hip dispatches to conv_rt.hip)
enum { MODE_DIRECT = 0, MODE_UP2 = 1, MODE_DILATE = 2 };

struct Gather {               // how the (virtual) conv input is read from the stored sources
    const char* src[2];
    int C[2];
    int Hs[2], Ws[2];
    int mode[2];
    int Hi, Wi;               // virtual input extent (zero outside)
    int stride;
};

struct ConvK {
    Gather g;
    int Ho, Wo;               // conv output extent
    const char* w;            // [N][9][Ctot]
    int Ctot, N;
    const float* bias;
    int relu;
    char* out;                // [B][Ho(/2)][Wo(/2)][N]
    const char* mask;         // same shape as out or null
    // input gradient w.r.t. BOTH sources of a concat layer in one launch (colvo_conv_dgrad_both): output channels >= nsplit
    // (a multiple of the channel tile) belong to the second source and go to out2 / mask2 with N - nsplit channels per pixel
    char* out2;
    const char* mask2;
    int nsplit;               // 0: single output
    int accumulate, pool2;
    int toh, tow, tiles_x, tiles_y;
    int pwp;                  // LDS pitch of a patch row, in pixels (>= patch width: padded against bank conflicts, pick_tile)
    uint32_t m_pw, m_tow;     // ceil(2^32 / patch width), ceil(2^32 / tow): index / d == umulhi(index, m) for index < 2^16
    int ntn, xcd;             // one-tile kernel: output-channel tiles per pixel tile (1-D grid, n-tile fastest), XCD remap on/off
#ifdef COLVO_ABLATE
    int abl;                  // developer build only (tools/ablate_conv.sh): bit mask of kernel phases to skip
    long long* trace;         // developer build only: [workgroup][8] wall-clock stamps (100 MHz) of the kernel phases
#endif
};

// conv_rt.hip: register-tiled stride-1 kernel for large grids (forward and input gradient); -1: the layer does not qualify
int try_launch_conv_rt(const ConvK& k, int B, int dtype, hipStream_t s);

// wgrad_rt.hip: register-tiled weight gradient (bf16, stride 1).  wgrad.hip asks for a plan (false: the layer does not qualify), sets up
// the way out that the plan's split count calls for (atomics / slabs / sole writer), launches, and runs its own second launch if any.
struct WgradRtPlan {
    int ni, toh, tow, pwl, pimg, ksteps, xinstr;
    int tiles_x, tiles_y, ntiles, tiles_per_split, nsplit;
    int nci0, nci, nco;
};
bool wgrad_rt_plan(const ColvoConvDesc* d, WgradRtPlan& p);
int wgrad_rt_launch(const WgradRtPlan& p, const ColvoConvDesc* d, const void* x0, const void* x1, const void* dy, float* dw, float* db,
                    float* slabs, float* db_slabs, int det, hipStream_t s);

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct bf16_t { uint16_t v; };

template <typename T> struct TT;
template <> struct TT<float> { static constexpr int G = 4; static constexpr int ES = 4; };
template <> struct TT<bf16_t> { static constexpr int G = 8; static constexpr int ES = 2; };

constexpr int NT = 256;
constexpr int BM = 128;   // output pixels per workgroup (4 waves x 2 fragments x 16 rows)

// LDS pitches.  A ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (not in
// contiguous 16-lane groups): 8 lanes of one k-group plus 8 lanes of the next one (+16 B).  With a pitch of r
// 16-byte slots per fragment row the group is conflict-free iff {r*l mod 16} are distinct EVEN slots for the 8 rows
// of a half: r = 6 (96 B) for 64-byte payloads, r = 2 (32 B, no padding) for 32-byte payloads.
// In general any pitch of r slots with r = 2 (mod 4) works: {r*l mod 16} are then 8 distinct even slots for the 8 rows
// of a half group, and the other k-group sits on the odd slots.
constexpr int pitch_slots(int n) { return n + ((2 - n % 4) + 4) % 4; }
constexpr int pitch_bytes(int payload) { return pitch_slots(payload / 16) * 16; }
constexpr int wrow_bytes(int granules) { return pitch_slots(granules) * 16; }
// Pixel pitch where consecutive fragment rows are TWO patch pixels apart (stride-2 forward / weight-gradient patches, the dy patch
// of k_dgrad_up2): the 8 rows of a ds_read_b128 half group then sit at {2 r l mod 16}, distinct even slots iff r is ODD -- with the
// r = 6 of pitch_bytes(64) every such read took two passes (profiles/r3_conv_pmc.json: 35-44 % of the LDS-active cycles of exactly
// these kernels were bank conflicts).  The transposed 8-byte reads of the weight gradient (two 32-lane groups, 8 pixels x 32 bytes
// each) need 2 x pitch = an odd multiple of 32 bytes: the same condition.  64-byte payload -> 80, 32 -> 48, 16 -> 16.
constexpr int pitch_bytes_s2(int payload) { return ((payload / 16) | 1) * 16; }
// LDS row pitch (in pixels, >= pw) of a weight-gradient patch such that the LDS pixel index of tile pixel p = oy * tow + ox is
// congruent to p modulo 8: a read group's 8 CONSECUTIVE tile pixels then keep their conflict-free spacing across a tile-row
// boundary as well (tiles whose width is not a multiple of 8: the 16x20, 8x10, 4x5 ... maps).  Multiples of 8 need no padding.
inline int wgrad_row_pitch(int pw, int tow) { return (tow % 8 == 0) ? pw : pw + (((tow - pw) % 8) + 8) % 8; }

#ifdef COLVO_ABLATE
#define ABL(bit) ((a.abl & (bit)) != 0)
#define TRACE(slot)                                                                                       \
    do {                                                                                                  \
        if (a.trace && threadIdx.x == 0)                                                                  \
            a.trace[(size_t)blockIdx.x * 8 + (slot)] = \
                (long long)wall_clock64();                                                                \
    } while (0)
#else
#define ABL(bit) false
#define TRACE(slot) do {} while (0)
#endif

struct WgradK {
    Gather g;
    int Ho, Wo, B;
    const char* dy;           // [B][Ho][Wo][Cout]
    int Cout;
    float* dw;                // [Cout][9][Ctot]
    int Ctot;
    float* db;
    int toh, tow, tiles_x, tiles_y, ntiles, tiles_per_split;
    uint32_t m_pw, m_tow;     // see ConvK
    int nsplit, cot, xcd;     // 1-D grid: (co tile, chunk) fastest, pixel-range split slowest; XCD remap on/off
    int pwl;                  // LDS pitch of a patch row, in pixels (wgrad_row_pitch)
    int det;                  // ONE split: 1 = deterministic form, plain read-modify-write instead of atomics (no slab); 2 = the arena is
                              // known to be zero: plain stores
    int slabs_only;           // host side: colvo_conv_wgrad_slabs -- always slabs (a single split too), no second launch
    int clean;                // host side: the caller vouches that dw / db are zero (colvo_conv_wgrad_clean)
    // Deterministic form (colvo_conv_wgrad_det): every pixel-range split STORES its sums into a slab of its own instead of
    // adding them to dw / db with float atomics; k_wgrad_reduce then adds the slabs in split order.  null: atomics.
    float* slabs;             // [nsplit][Cout * 9 * Ctot]
    float* db_slabs;          // [nsplit][Cout]
    // host side only
    const char* scratch;
    long long scratch_bytes;
    int* plan_out;            // non-null: write the number of splits the launch WOULD use and do not launch
#ifdef COLVO_WTRACE
    long long* trace;         // developer build only (tools/wtrace_wgrad.sh): [workgroup][8] shader-clock stamps of the kernel phases
#endif
};

// Workgroups are dealt round-robin over the 8 XCDs (private L2s): give each XCD a CONTIGUOUS range of the logical work ids,
// so that the workgroups that share operands -- the output-channel tiles of one pixel tile (same input patch), the
// (co tile, channel chunk) workgroups of one pixel range in the weight gradient -- run on ONE L2 at about the same time.
// Bijective for any n (cdna_hip_programming.md T1).  Speed only, never correctness.  COLVO_NO_XCD_REMAP=1 turns it off.
__device__ __forceinline__ int xcd_remap(int id, int n, int on) {
    if (!on) return id;
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// i / d for 0 <= i < 2^16, 2 <= d < 2^16 with m = ceil(2^32 / d): one v_mul_hi_u32 instead of the ~35-instruction
// runtime division (the address set-up of a workgroup was most of its VALU time)
__device__ __forceinline__ int mdiv(int i, uint32_t m) { return (int)__umulhi((uint32_t)i, m); }
inline uint32_t mdiv_magic(int d) { return (uint32_t)((0x100000000ULL + (uint32_t)d - 1) / (uint32_t)d); }

// Accumulator hand-over from the MFMA chain to the epilogue.
//
// hipcc's register allocator sometimes ROTATES the accumulators of a chain, i.e. emits MFMAs whose vDst is not their SrcC
// (`v_mfma a[4:7], .., .., a[8:11]`).  Measured on MI355X (tools/ubench/mfma_hazard.hip, profiles/r2_mfma_hazard.md):
//   * in-place MFMAs (vDst == SrcC): v_mfma_f32_16x16x4_f32 results are hardware-interlocked against VALU reads (correct
//     with ZERO wait states); v_mfma_f32_16x16x32_bf16 results need the 7 wait states of LLVM's table -- hipcc is right;
//   * rotated MFMAs: NOT interlocked; the f32 form needs >= 10, the bf16 form >= 8 wait states in a 2-waves-per-SIMD
//     micro-benchmark -- hipcc inserts 10 / 7, i.e. no margin / one too few -- and in the real f32 persistent kernel a
//     read 13 states after a rotated MFMA still came back stale in a few lanes (the round-1 "stale accumulator" bug).
// A wait-state pad cannot be sized for that, so the chain is closed with one IN-PLACE MFMA per accumulator
// (acc = 0 * 0 + acc, vDst tied to SrcC by the asm constraint): the SrcC hand-over from a rotated producer is interlocked
// (ubench ROT 2 / ROT 3), and the terminator's own result is of the safe in-place kind.  All terminators and the 7 + 4
// wait states the bf16 form needs sit in ONE asm statement whose outputs are the accumulators, so no read can be
// scheduled in between.  The library is built with -mllvm -amdgpu-mfma-vgpr-form (coivo_amd/build.py): accumulators live
// in VGPRs, otherwise the "+v" operands would make hipcc copy AGPR -> VGPR (= read the results) in FRONT of this statement.
// tools/isa_check_mfma.py (tests/test_isa_cpu.py) verifies on the emitted ISA that no rotated MFMA result is read early.
#define MT_(i) MFMA_TERM_OP " %" #i ", %[z], %[z], %" #i "\n\t"
#define MG_(i) COLVO_ACC_CONSTRAINT(acc[i])
template <typename T, int N>
__device__ __forceinline__ void mfma_result_guard(f32x4 (&acc)[N]) {
    static_assert(N == 1 || N == 2 || N == 3 || N == 4 || N == 5 || N == 6 || N == 8 || N == 10 || N == 12 || N == 16 || N == 20,
                  "mfma_result_guard: add a case for this accumulator count");
    // "s_nop 1" in front: the zero operand may have been written by the VALU instruction just before (VALU write -> MFMA read)
#define MFMA_GUARD_BODY()                                                                                                  \
    if constexpr (N == 1) asm volatile("s_nop 1\n\t" MT_(0) MFMA_TERM_TAIL : MG_(0) : [z] "v"(z));                             \
    else if constexpr (N == 2) asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MFMA_TERM_TAIL : MG_(0), MG_(1) : [z] "v"(z));             \
    else if constexpr (N == 3) asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MFMA_TERM_TAIL : MG_(0), MG_(1), MG_(2) : [z] "v"(z)); \
    else if constexpr (N == 4)                                                                                           \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MFMA_TERM_TAIL : MG_(0), MG_(1), MG_(2), MG_(3) : [z] "v"(z));      \
    else if constexpr (N == 5)                                                                                           \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MT_(4) MFMA_TERM_TAIL                                            \
                     : MG_(0), MG_(1), MG_(2), MG_(3), MG_(4) : [z] "v"(z));                                              \
    else if constexpr (N == 6)                                                                                           \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MT_(4) MT_(5) MFMA_TERM_TAIL                                     \
                     : MG_(0), MG_(1), MG_(2), MG_(3), MG_(4), MG_(5) : [z] "v"(z));                                      \
    else if constexpr (N == 8)                                                                                           \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MT_(4) MT_(5) MT_(6) MT_(7) MFMA_TERM_TAIL                       \
                     : MG_(0), MG_(1), MG_(2), MG_(3), MG_(4), MG_(5), MG_(6), MG_(7) : [z] "v"(z));                      \
    else if constexpr (N == 10)                                                                                          \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MT_(4) MT_(5) MT_(6) MT_(7) MT_(8) MT_(9) MFMA_TERM_TAIL         \
                     : MG_(0), MG_(1), MG_(2), MG_(3), MG_(4), MG_(5), MG_(6), MG_(7), MG_(8), MG_(9) : [z] "v"(z));      \
    else if constexpr (N == 12)                                                                                          \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MT_(4) MT_(5) MT_(6) MT_(7) MT_(8) MT_(9) MT_(10) MT_(11)        \
                     MFMA_TERM_TAIL                                                                                      \
                     : MG_(0), MG_(1), MG_(2), MG_(3), MG_(4), MG_(5), MG_(6), MG_(7), MG_(8), MG_(9), MG_(10), MG_(11)   \
                     : [z] "v"(z));                                                                                      \
    else if constexpr (N == 16)                                                                                          \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MT_(4) MT_(5) MT_(6) MT_(7) MT_(8) MT_(9) MT_(10) MT_(11)        \
                     MT_(12) MT_(13) MT_(14) MT_(15) MFMA_TERM_TAIL                                                      \
                     : MG_(0), MG_(1), MG_(2), MG_(3), MG_(4), MG_(5), MG_(6), MG_(7), MG_(8), MG_(9), MG_(10), MG_(11),  \
                       MG_(12), MG_(13), MG_(14), MG_(15)                                                                \
                     : [z] "v"(z));                                                                                      \
    else                                                                                                                 \
        asm volatile("s_nop 1\n\t" MT_(0) MT_(1) MT_(2) MT_(3) MT_(4) MT_(5) MT_(6) MT_(7) MT_(8) MT_(9) MT_(10) MT_(11)        \
                     MT_(12) MT_(13) MT_(14) MT_(15) MT_(16) MT_(17) MT_(18) MT_(19) MFMA_TERM_TAIL                      \
                     : MG_(0), MG_(1), MG_(2), MG_(3), MG_(4), MG_(5), MG_(6), MG_(7), MG_(8), MG_(9), MG_(10), MG_(11),  \
                       MG_(12), MG_(13), MG_(14), MG_(15), MG_(16), MG_(17), MG_(18), MG_(19)                            \
                     : [z] "v"(z))
    if constexpr (TT<T>::ES == 4) {
#define MFMA_TERM_OP "v_mfma_f32_16x16x4_f32"
#define MFMA_TERM_TAIL "s_nop 3"       /* interlocked in hardware; a token pad for the (unmodelled) asm boundary */
        const float z = 0.0f;
        MFMA_GUARD_BODY();
#undef MFMA_TERM_OP
#undef MFMA_TERM_TAIL
    } else {
#define MFMA_TERM_OP "v_mfma_f32_16x16x32_bf16"
#define MFMA_TERM_TAIL "s_nop 11"      /* in-place XDL 16x16x32 result -> VALU read: 7 wait states (+4 margin) */
        const u32x4 z = {0u, 0u, 0u, 0u};
        MFMA_GUARD_BODY();
#undef MFMA_TERM_OP
#undef MFMA_TERM_TAIL
    }
#undef MFMA_GUARD_BODY
}
#undef MT_
#undef MG_

__device__ __forceinline__ u32x4 ld16(const char* p) { return *reinterpret_cast<const u32x4*>(p); }
// 16-byte buffer load: 32-bit per-lane byte offset + scalar byte offset; an offset beyond the descriptor's size
// returns zeros (hardware bounds check), so padding / out-of-image granules need no branch: they get OOB_OFF.
__device__ __forceinline__ u32x4 bld16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}
constexpr int OOB_OFF = 0x40000000;
__device__ __forceinline__ void st16(char* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

struct Tile { int toh, tow, pwp; };   // pwp: LDS pitch of a patch row in pixels (>= patch width; 0: not chosen, use the width)

// LDS cycles of the patch-fragment ds_read_b128 of the conv kernels, relative to conflict-free (1.0 ... 4.0): lane (l15, kg)
// of wave w, fragment mf reads 16 bytes at ((oy*S)*pwp + ox*S)*16r + kg*16 with (oy, ox) = divmod(w*32 + mf*16 + l15, tow)
// (r = 6: pitch_bytes(64); r = 5 where S = 2: pitch_bytes_s2(64)).
// The instruction is serviced in the four 16-lane groups below, 64 banks of 4 bytes (MI355X_MICROARCH.md, LDS).
inline double patch_read_conflicts(int toh, int tow, int pwp, int S, int r = 6) {
    static const int grp[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                   {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                   {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                   {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    const int npix = toh * tow;
    long total = 0;
    for (int w = 0; w < 4; ++w)
        for (int mf = 0; mf < 2; ++mf)
            for (int g = 0; g < 4; ++g) {
                int slot_of[16];                            // 16-byte slots: a lane covers 4 consecutive banks = one slot of 16
                for (int j = 0; j < 16; ++j) {
                    const int lane = grp[g][j], l15 = lane & 15, kg = lane >> 4;
                    int p = w * 32 + mf * 16 + l15;
                    if (p >= npix) p = 0;
                    const int oy = p / tow, ox = p - oy * tow;
                    slot_of[j] = ((oy * S) * pwp + ox * S) * r + kg;      // r: 16-byte slots per patch pixel
                }
                int worst = 1;                              // distinct addresses on one bank set serialise; equal ones broadcast
                for (int r = 0; r < 16; ++r) {
                    int distinct[16], n = 0;
                    for (int j = 0; j < 16; ++j) {
                        if ((slot_of[j] & 15) != r) continue;
                        bool seen = false;
                        for (int q = 0; q < n; ++q) if (distinct[q] == slot_of[j]) { seen = true; break; }
                        if (!seen) distinct[n++] = slot_of[j];
                    }
                    if (n > worst) worst = n;
                }
                total += worst;
            }
    return (double)total / 32.0;
}

// Choose the tile region (<= BM pixels) that wastes the fewest fragment rows; ties: least staged patch.  With `lds_aware`
// (the forward / input-gradient kernels) the cost also carries the bank conflicts of the patch-fragment reads -- half of a
// 128 x 32 tile's LDS reads -- and the patch rows may be padded by up to 8 pixels in LDS: a 16 x 8 tile costs every such
// read two passes, 8 x 16 none, and where 8-wide tiles are the better fit a row pitch of 16 pixels makes them conflict-free
// (profiles/r2_conv_pmc.json: 26-48 % of the LDS-active cycles were conflicts).  ext = patch width - tile width at stride 1.
// Results are cached: the search runs once per shape.
inline Tile pick_tile(int Ho, int Wo, int stride, bool even, int BM = 128, bool lds_aware = false, int ext = 3) {
    const bool aware_off = !TUNE(lds_aware_tiles);
    const int max_pad = (int)TUNE(lds_tile_max_pad);
    if (aware_off) lds_aware = false;
    struct Key { int Ho, Wo, stride, even, BM, aware, ext; };
    static std::vector<std::pair<Key, Tile>> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& e : cache) {
        const Key& q = e.first;
        if (q.Ho == Ho && q.Wo == Wo && q.stride == stride && q.even == (int)even && q.BM == BM && q.aware == (int)lds_aware &&
            q.ext == ext)
            return e.second;
    }
    Tile best{even ? 2 : 1, even ? 2 : 1, 0};
    double best_cost = 1e30;
    const int step = even ? 2 : 1;
    for (int tow = step; tow <= (Wo + step - 1) / step * step && tow <= BM; tow += step) {
        int toh = BM / tow;
        if (even) toh &= ~1;
        const int hcap = (Ho + step - 1) / step * step;
        if (toh > hcap) toh = hcap;
        if (toh < step) continue;
        const long tiles = (long)((Ho + toh - 1) / toh) * ((Wo + tow - 1) / tow);
        const int ph = (toh - 1) * stride + ext, pw = (tow - 1) * stride + ext;
        const double base = (double)tiles * (BM + 0.25 * ph * pw);
        if (!lds_aware || BM != 128) {
            if (base < best_cost) { best_cost = base; best = Tile{toh, tow, pw}; }
            continue;
        }
        for (int pad = 0; pad <= max_pad; ++pad) {
            // (stride-2 forward patches, ext 3, are staged at pitch_bytes_s2 = 5 slots per pixel; k_dgrad_up2's dy patch, ext 4, keeps 6)
            const double cf = patch_read_conflicts(toh, tow, pw + pad, stride, (stride == 2 && ext == 3) ? 5 : 6);
            const double cost = base * (1.0 + 0.2 * (cf - 1.0)) * (1.0 + 0.002 * pad);
            if (cost < best_cost) { best_cost = cost; best = Tile{toh, tow, pw + pad}; }
        }
    }
    cache.push_back({Key{Ho, Wo, stride, (int)even, BM, (int)lds_aware, ext}, best});
    return best;
}

int check_desc(const ColvoConvDesc* d, const char* who) {
    COLVO_CHECK_ARG(d, "%s: null descriptor", who);
    COLVO_CHECK_ARG(d->dtype == COLVO_F32 || d->dtype == COLVO_BF16, "%s: bad dtype %d", who, d->dtype);
    COLVO_CHECK_ARG(d->ksize == 3, "%s: only 3x3 convolutions are on this path (got k=%d)", who, d->ksize);
    COLVO_CHECK_ARG(d->stride == 1 || d->stride == 2, "%s: stride must be 1 or 2", who);
    COLVO_CHECK_ARG(d->B >= 1 && d->B <= 65535 && d->Hi >= 1 && d->Wi >= 1, "%s: bad shape", who);
    COLVO_CHECK_ARG(d->Ho == (d->Hi - 1) / d->stride + 1 && d->Wo == (d->Wi - 1) / d->stride + 1,
                    "%s: output %dx%d does not match input %dx%d / stride %d", who, d->Ho, d->Wo, d->Hi, d->Wi, d->stride);
    COLVO_CHECK_ARG(d->C0 >= 8 && d->C0 % 8 == 0 && d->C1 >= 0 && d->C1 % 8 == 0 && d->Cout >= 8 && d->Cout % 8 == 0,
                    "%s: channel counts must be multiples of 8 (C0=%d C1=%d Cout=%d)", who, d->C0, d->C1, d->Cout);
    COLVO_CHECK_ARG(!(d->up0 && ((d->Hi | d->Wi) & 1)) && !(d->up1 && ((d->Hi | d->Wi) & 1)),
                    "%s: up-sampled sources need an even input size", who);
    COLVO_CHECK_ARG(!((d->up0 || d->up1) && d->stride != 1), "%s: up-sampled sources need stride 1", who);
    return 0;
}

void fill_gather(const ColvoConvDesc* d, const void* x0, const void* x1, Gather& g) {
    g.src[0] = (const char*)x0; g.src[1] = (const char*)x1;
    g.C[0] = d->C0; g.C[1] = x1 ? d->C1 : 0;
    g.mode[0] = d->up0 ? MODE_UP2 : MODE_DIRECT;
    g.mode[1] = d->up1 ? MODE_UP2 : MODE_DIRECT;
    g.Hs[0] = d->up0 ? d->Hi / 2 : d->Hi; g.Ws[0] = d->up0 ? d->Wi / 2 : d->Wi;
    g.Hs[1] = d->up1 ? d->Hi / 2 : d->Hi; g.Ws[1] = d->up1 ? d->Wi / 2 : d->Wi;
    g.Hi = d->Hi; g.Wi = d->Wi; g.stride = d->stride;
}

}  // namespace
}  // namespace colvo
