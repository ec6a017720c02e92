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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// bf16 <-> f32 on raw 16-bit storage
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    __hip_bfloat16 b = __float2bfloat16(f);   // RNE, NaN-preserving (v_cvt_pk_bf16_f32 at -O3)
    return *reinterpret_cast<uint16_t*>(&b);
}

}  // namespace colvo
