// Microbenchmark: plain v_fma_f32 vs packed v_pk_fma_f32 issue rate on gfx950 (informs the fused-loss kernel design).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NACC>
__global__ __launch_bounds__(256) void k_plain(float* out, int iters, float a, float b) {
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_packed(float* out, int iters, float a, float b) {
    f2 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f2{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i};
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_elementwise_fma(acc[i], av, bv);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 256 * 8 * 4 * sizeof(float));
    const int iters = 20000;
    for (int wpc = 4; wpc <= 32; wpc *= 2) {           // waves per CU = blocks per CU * 4
        const int blocks = 256 * wpc / 4;
        float t1 = timeit([&] { hipLaunchKernelGGL((k_plain<8>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f); });
        float t2 = timeit([&] { hipLaunchKernelGGL((k_packed<8>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 1e-6f); });
        const double wi = (double)blocks * 4 * iters * 8;   // wave-instructions
        printf("waves/CU=%2d  plain: %.3f ms  %.2f wave-instr/ns chip (%.2f cyc/instr/SIMD @2.4GHz)   packed: %.3f ms  %.2f wave-instr/ns (%.2f cyc/instr/SIMD)  pk/plain flops ratio %.2f\n",
               wpc, t1, wi / t1 / 1e6, 1024.0 * 2.4 / (wi / t1 / 1e6), t2, wi / t2 / 1e6, 1024.0 * 2.4 / (wi / t2 / 1e6), 2 * t1 / t2);
    }
    return 0;
}
