// What does a fork cost the MAIN chain?  The backward pass hands every layer's dy to a side stream: hipEventRecord(main) +
// hipStreamWaitEvent(side) + a weight-gradient kernel on the side.  Under rocprofv3 the main chain shows ~7 us between its kernels
// wherever such a fork sits (tools/step_listing.py); this measures it without the tracer: a chain of N kernels of ~T us on the main
// stream, as
//   A  nothing in between
//   B  a fork after every kernel (event with hipEventDisableTiming, side kernel of ~T us, two side streams alternating)
//   C  the same with ONE side stream
//   D  event record only (no waiter)
//   E  fork after every kernel, but the side kernel is empty
//   F/G/H  as B/E/D with the event given to hipExtLaunchKernelGGL as the kernel's stopEvent instead of hipEventRecord
// hipcc --offload-arch=gfx950 -O3 -o fork_cost fork_cost.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

__global__ __launch_bounds__(256) void k_spin(long long cycles, int* sink) {       // (100 MHz constant clock: s_memrealtime)
    const long long t0 = __builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) {}
    if (sink && threadIdx.x == 1025) *sink = 1;
}
__global__ void k_empty() {}
// ordering check of the stopEvent fork: the producer spins, THEN writes its token; the consumer (side stream, behind the event) counts
// the tokens it does not see
__global__ __launch_bounds__(256) void k_produce(long long cycles, int* slot, int token) {
    const long long t0 = __builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) {}
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) __hip_atomic_store(slot, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_consume(const int* slot, int token, int* misses) {
    if (__hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != token) atomicAdd(misses, 1);
}

int main(int argc, char** argv) {
    const int N = 40, REP = 20;
    const long long cyc = argc > 1 ? atoll(argv[1]) : 40000;      // shader clock cycles per kernel (~2.4 GHz: 40000 = ~17 us)
    const int blocks = argc > 2 ? atoi(argv[2]) : 256;
    hipStream_t m, s[2];
    hipStreamCreateWithFlags(&m, hipStreamNonBlocking);
    for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[64], e0, e1, j[2];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto& e : j) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"A plain chain", "B fork, 2 side streams", "C fork, 1 side stream", "D record only", "E fork, empty side kernel",
                           "F stopEvent fork, 2 sides", "G stopEvent fork, empty side", "H stopEvent only"};
    for (int mode = 0; mode < 8; ++mode) {
        float best = 1e9f, sum = 0;
        double host = 0;
        for (int rep = 0; rep < REP + 2; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, m);
            const auto h0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode >= 5) {
                    // the event rides on the kernel's own completion signal: no marker packet on the main queue
                    hipExtLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, m, nullptr, ev[i], 0, cyc, (int*)nullptr);
                    if (mode == 7) continue;
                    hipStream_t side = s[mode == 5 ? (i & 1) : 0];
                    hipStreamWaitEvent(side, ev[i], 0);
                    if (mode == 6) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, side);
                    else hipLaunchKernelGGL(k_spin, dim3(blocks / 2), dim3(256), 0, side, cyc, (int*)nullptr);
                    continue;
                }
                hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, m, cyc, (int*)nullptr);
                if (mode == 0) continue;
                hipEventRecord(ev[i], m);
                if (mode == 3) continue;
                hipStream_t side = s[mode == 1 ? (i & 1) : 0];
                hipStreamWaitEvent(side, ev[i], 0);
                if (mode == 4) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, side);
                else hipLaunchKernelGGL(k_spin, dim3(blocks / 2), dim3(256), 0, side, cyc, (int*)nullptr);
            }
            hipEventRecord(e1, m);                      // the main chain's own end (the side work is not waited for)
            if (rep >= 2) host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
            hipEventSynchronize(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-28s main chain of %d: mean %.1f us, best %.1f us  (%.2f us / kernel); host issue %.1f us\n", names[mode], N,
               sum / REP * 1e3, best * 1e3, best * 1e3 / N, host / REP);
    }
    // ordering check: 200 producer / consumer pairs through stopEvent forks (and, as the control, with NO wait at all)
    int *slots, *misses;
    hipMalloc(&slots, 256 * sizeof(int)); hipMalloc(&misses, 2 * sizeof(int));
    hipMemset(slots, 0, 256 * sizeof(int)); hipMemset(misses, 0, 2 * sizeof(int));
    hipDeviceSynchronize();
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 200; ++i) {
            const int token = 1000 * (pass + 1) + i;
            hipExtLaunchKernelGGL(k_produce, dim3(blocks), dim3(256), 0, m, nullptr, ev[i % 64], 0, cyc / 4, slots + (i & 255), token);
            hipStream_t side = s[i & 1];
            if (pass == 0) hipStreamWaitEvent(side, ev[i % 64], 0);
            hipLaunchKernelGGL(k_consume, dim3(1), dim3(1), 0, side, (const int*)(slots + (i & 255)), token, misses + pass);
        }
        hipDeviceSynchronize();
    }
    int h[2];
    hipMemcpy(h, misses, sizeof(h), hipMemcpyDeviceToHost);
    printf("ordering: %d of 200 consumers behind a stopEvent fork missed their token (must be 0); without any wait: %d\n", h[0], h[1]);
    return h[0] != 0;
}
