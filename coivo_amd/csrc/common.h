// common.h -- shared host/device helpers for libcolvo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_ext.h>
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

// Every kernel launch of the library goes through launch(): a launch can then CARRY an event.  hipEventRecord puts a marker packet
// behind the kernel on its queue, and the next kernel of that queue waits for it: 2.8 us per record on the producing chain, 5.3 us
// when another queue already waits on the event -- the backward pass forks a weight-gradient kernel off every layer of its
// input-gradient chain (tools/ubench/fork_cost.hip: chain of 40 kernels of 18 us: 18.0 us per kernel alone, 20.8 with a record after
// each, 23.3 with record + waiter; 19.4 with the event as hipExtLaunchKernel's stopEvent, which rides on the kernel's own completion
// signal: no packet on the producer's queue).  colvo_run_commands arms the tap around a command that a FORK follows.
struct LaunchTap {
    hipEvent_t stop = nullptr;     // event to attach to launches on `stream` (nullptr: plain launches)
    hipStream_t stream = nullptr;
    int used = 0;                  // launches that carried it
};
extern thread_local LaunchTap g_launch_tap;      // program.hip (commands run on the thread that armed it)

template <typename... P, typename... A>
inline void launch(void (*kernel)(P...), dim3 grid, dim3 block, unsigned lds, hipStream_t s, A&&... a) {
    LaunchTap& t = g_launch_tap;
    if (t.stop != nullptr && t.stream == s) {
        ++t.used;
        hipExtLaunchKernelGGL(kernel, grid, block, lds, s, nullptr, t.stop, 0u, static_cast<P>(a)...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, static_cast<P>(a)...);
    }
}

// Which kernel FORM a dispatcher chose, counted per process (colvo_form_counts: the tests of the grid-size-selected forms read it --
// a parity test at a small shape must be able to say that the large-grid kernel is what actually ran).
enum { FORM_CONV_RT = 0, FORM_WGRAD_FULL_GRID, FORM_WGRAD_HALVED_GRID, FORM_WGRAD_UP2, FORM_WGRAD_RT, FORM_WGRAD_STORE_CLEAN, FORM_CONV_RES_S2,
       FORM_CONV_Q, FORM_COUNT };
void form_hit(int id);

// csrc/bwd16.hip: the MFMA form of colvo_conv_dgrad_planes (bf16, stride 2, 16 output channels, two input channels, even extents)
int launch_dgrad_planes_s2_mfma(const void* g, const float* w, int Cin, int c_begin, int B, int Hi, int Wi, int Ho, int Wo, float* dst,
                                int accumulate, hipStream_t stream);

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

// two floats -> two bf16 in one register (element 0 in the low half): ONE v_cvt_pk_bf16_f32, where two f2bf() and the shift / or that
// joins them cost four instructions -- same instruction, same rounding
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{lo, hi}, bf16x2_));
}

}  // namespace colvo
