// common.h -- shared host/device helpers for libcolvo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/colvo.h"

namespace colvo {

void set_error(const char* fmt, ...);

#define COLVO_CHECK_ARG(cond, ...)                   \
    do {                                             \
        if (!(cond)) {                               \
            ::colvo::set_error(__VA_ARGS__);         \
            return (int)hipErrorInvalidValue;        \
        }                                            \
    } while (0)

#define COLVO_CHECK_LAUNCH(name)                                                    \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            ::colvo::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return (int)e_;                                                         \
        }                                                                           \
    } while (0)

__device__ __forceinline__ float uniform_f(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Sum over the 64 lanes, result in every lane.  Row rotations by DPP (row_ror: VALU only, no LDS crossbar
// round trip as with ds_bpermute-based shuffles), then the four 16-lane row totals through scalar registers.
__device__ __forceinline__ float wave_sum(float v) {
#define COLVO_ROR(x, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x120 + (n), 0xf, 0xf, false))
    v += COLVO_ROR(v, 8);
    v += COLVO_ROR(v, 4);
    v += COLVO_ROR(v, 2);
    v += COLVO_ROR(v, 1);
#undef COLVO_ROR
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// bf16 <-> f32 on raw 16-bit storage
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    __hip_bfloat16 b = __float2bfloat16(f);   // RNE, NaN-preserving (v_cvt_pk_bf16_f32 at -O3)
    return *reinterpret_cast<uint16_t*>(&b);
}

}  // namespace colvo
