// Microbenchmark: what does one kernel in a stream of dependent kernels cost on MI355X when it does (almost) nothing?
// -> the floor under every ~15 us conv layer.   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_empty(float* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_lds(float* p) {
    extern __shared__ float s[];
    s[threadIdx.x] = 1.0f;
    __syncthreads();
    if (p && s[(threadIdx.x + 1) & 255] == 2.0f) p[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_touch(float* p, int n) {     // every workgroup writes 1 KB: dirty lines to flush
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (float)i;
}
template <typename F> float per_launch_us(F f, int n = 400) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < n; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
    float* p; hipMalloc(&p, 64 << 20);
    hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int wgs : {1, 256, 1280, 5120, 40960}) {
        float a = per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0, p); });
        float b = per_launch_us([&] { hipLaunchKernelGGL(k_lds, dim3(wgs), dim3(256), 36 * 1024, 0, p); });
        float c = per_launch_us([&] { hipLaunchKernelGGL(k_touch, dim3(wgs), dim3(256), 0, 0, p, wgs * 256); });
        printf("workgroups %6d: empty %.2f us   36KB-LDS+barrier %.2f us   1KB-store per workgroup %.2f us\n", wgs, a, b, c);
    }
    return 0;
}
