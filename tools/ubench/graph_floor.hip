// Microbenchmark: cost of one (almost) empty kernel in a chain of dependent kernels -- launched eagerly on a stream, and the same
// chain captured once into a hipGraph and replayed.  hipcc --offload-arch=gfx950 -O3 -o graph_floor graph_floor.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_empty(float* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p[0] = 1.0f;
}
__global__ __launch_bounds__(256) void k_touch(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = p[i] + 1.0f;
}
int main() {
    float* p; hipMalloc(&p, 64 << 20);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 400;
    for (int wgs : {1, 256, 1280}) {
        for (int kind = 0; kind < 2; ++kind) {
            auto launch = [&] {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, s, p);
                else hipLaunchKernelGGL(k_touch, dim3(wgs), dim3(256), 0, s, p, wgs * 256);
            };
            for (int i = 0; i < 20; ++i) launch();
            hipStreamSynchronize(s);
            hipEventRecord(e0, s); for (int i = 0; i < N; ++i) launch(); hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms_e; hipEventElapsedTime(&ms_e, e0, e1);
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int i = 0; i < N; ++i) launch();
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
            hipStreamSynchronize(s);
            hipEventRecord(e0, s); for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms_g; hipEventElapsedTime(&ms_g, e0, e1);
            printf("workgroups %5d %s: eager %.2f us per kernel, graph replay %.2f us per kernel\n", wgs, kind ? "touch" : "empty",
                   ms_e * 1e3f / N, ms_g * 1e3f / (5 * N));
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
