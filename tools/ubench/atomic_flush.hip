// Microbenchmark behind the weight-gradient flush (csrc/wgrad.hip): 256 workgroups x 4 waves end with 40 float adds per lane into
// [32 co][288 cols] slabs of dw (row pitch 9*Ctot floats).  The in-kernel stamps of tools/wtrace_wgrad.sh put that flush at 6.0-6.9 us
// in EVERY layer -- a third of the kernel -- whatever the number of pixel-range splits that share a slab.  What does it cost as
//   A  agent-scope fp32 atomics in the accumulator layout (4 rows x 16 floats per wave instruction)          -- the kernel as it is
//   B  the same atomics with 64 CONSECUTIVE floats per wave instruction (accumulators transposed through LDS first)
//   C  workgroup-scope atomics (performed in the XCD's L2, NOT coherent across XCDs: speed reference only)
//   D  plain 4-byte stores into a slab of the workgroup's own (deterministic form), accumulator layout
//   E  plain 16-byte stores into its own slab, 4 consecutive floats per lane
// for 1 / 16 / 256 slabs shared by 256 / 16 / 1 workgroups?      hipcc --offload-arch=gfx950 -O3 -o atomic_flush atomic_flush.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
constexpr int ROWS = 32, COLS = 288, NOPS = 40;      // per workgroup: 32 x 288 = 9216 floats = 256 lanes x ... 36 per lane; 40 incl. padding columns

template <int MODE>
__global__ __launch_bounds__(256) void k_flush(float* dw, float* slabs, int nslabs, int row_pitch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, kg = lane >> 4;
    const int slab = blockIdx.x % nslabs;
    float* base = dw + (size_t)slab * ROWS * row_pitch;                 // slabs side by side along the rows
    float* mine = slabs + (size_t)blockIdx.x * ROWS * COLS;
    const float v = 1.0f + lane * 1e-3f;
    if (MODE == 0 || MODE == 2 || MODE == 3) {
        // accumulator layout: fragment f = wave + 4 fi (columns 16 f + l15), co rows 16 mi + 4 kg + r
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int fi = 0; fi < 5; ++fi) {
                const int f = wave + 4 * fi;
                if (f >= 18) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * mi + 4 * kg + r, col = 16 * f + l15;
                    if (MODE == 0) atomicAdd(base + (size_t)row * row_pitch + col, v);
                    else if (MODE == 2) __hip_atomic_fetch_add(base + (size_t)row * row_pitch + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else mine[row * COLS + col] = v;
                }
            }
    } else if (MODE == 1) {
        // 64 consecutive floats per wave instruction: 9216 floats = 144 instructions of 64 lanes, 36 per wave
        for (int i = wave; i < ROWS * COLS / 64; i += 4) {
            const int e = i * 64 + lane, row = e / COLS, col = e - row * COLS;
            atomicAdd(base + (size_t)row * row_pitch + col, v);
        }
    } else {
        // 16-byte stores: 9216 floats = 2304 float4 = 9 per lane
        for (int i = threadIdx.x; i < ROWS * COLS / 4; i += 256) reinterpret_cast<float4*>(mine)[i] = make_float4(v, v, v, v);
    }
}

template <int MODE> float run(float* dw, float* slabs, int nslabs, int row_pitch, int n = 200) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_flush<MODE>, dim3(256), dim3(256), 0, 0, dw, slabs, nslabs, row_pitch);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_flush<MODE>, dim3(256), dim3(256), 0, 0, dw, slabs, nslabs, row_pitch);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
__global__ void k_empty() {}
int main() {
    float *dw, *slabs;
    const int row_pitch = 9 * 512;                   // enc5b: Ctot = 512
    hipMalloc(&dw, (size_t)256 * ROWS * row_pitch * 4);
    hipMalloc(&slabs, (size_t)256 * ROWS * COLS * 4);
    hipMemset(dw, 0, (size_t)256 * ROWS * row_pitch * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("empty kernel (launch floor): %.2f us\n", ms * 1e3f / 200);
    for (int nslabs : {1, 4, 16, 64, 256}) {
        printf("slabs %3d (workgroups per slab %3d): A atomics acc-layout %.2f us | B atomics 64-consecutive %.2f | C wg-scope atomics %.2f | "
               "D own-slab 4B stores %.2f | E own-slab 16B stores %.2f\n", nslabs, 256 / nslabs,
               run<0>(dw, slabs, nslabs, row_pitch), run<1>(dw, slabs, nslabs, row_pitch), run<2>(dw, slabs, nslabs, row_pitch),
               run<3>(dw, slabs, nslabs, row_pitch), run<4>(dw, slabs, nslabs, row_pitch));
    }
    return 0;
}
