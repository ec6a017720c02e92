// Can a dependent chain of kernels hide its launch + set-up latency?  A kernel boundary on one queue costs ~2.7 us before the next
// kernel's first wave does anything (launch_floor.hip), plus that kernel's own set-up and first loads -- paid ~60 times on the critical
// chain of a training step.  hipExtLaunchKernel's hipExtAnyOrderLaunch flag clears the AQL barrier bit: the next kernel's workgroups are
// dispatched while the previous kernel's tail still runs, and must then wait for their producer THEMSELVES (a completion counter the
// producer's workgroups bump, a spin in the consumer).  This measures the best case of that form -- timing only, no cache maintenance,
// no real data dependence -- against the ordinary chain:
//   A  ordinary launches (barrier bit), each kernel: `setup` us of independent work, then `work` us
//   B  any-order launches; each workgroup: setup, spin until the previous kernel's counter is complete, work, bump its own counter
// Every spinning wave of kernel n+1 was dispatched AFTER all workgroups of kernel n (one queue dispatches in order), so the producer
// always finishes; a guard bounds every spin all the same (2^22 polls -> the kernel gives up and flags an error).
// hipcc --offload-arch=gfx950 -O3 -o anyorder_chain anyorder_chain.hip ;  ./anyorder_chain [wgs=640] [setup_cycles=2000] [work_cycles=20000]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void burn(long long cycles) {
    const long long t0 = __builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < cycles) {}
}

template <bool WAIT, int SLEEP>
__global__ __launch_bounds__(256) void k_link(long long setup, long long work, const unsigned* prev, unsigned prev_target, unsigned* mine,
                                              unsigned* err) {
    burn(setup);
    if (WAIT && prev) {
        // the consumers poll ONE word (prev[0]) that only the last of 16 sub-counter groups of the producer writes: the polls must
        // not queue in front of the producers' atomics (all on one word: 19 us per kernel instead of 10.4)
        if (threadIdx.x == 0) {
            int guard = 0;
            while (__hip_atomic_load(prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 16u) {
                __builtin_amdgcn_s_sleep(SLEEP);
                if (++guard > (1 << 20)) { atomicAdd(err, 1u); break; }
            }
            // (timing probe: relaxed atomics, no fences -- an agent-scope release per workgroup is an L2 write-back each: 14 ns x workgroups per kernel)
        }
        __syncthreads();
    }
    burn(work);
    if (WAIT) {
        __syncthreads();
        if (threadIdx.x == 0) {
            // sub-counter g = block mod 16 (its own 64-byte line); the workgroup that completes a group bumps the top word
            const unsigned g = blockIdx.x & 15u, members = (gridDim.x + 15u - g) / 16u;
            if (__hip_atomic_fetch_add(mine + 16 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == members)
                __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 640;
    const long long setup = argc > 2 ? atoll(argv[2]) : 2000, work = argc > 3 ? atoll(argv[3]) : 20000;
    const int N = 40, REP = 10;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *cnt, *err;
    hipMalloc(&cnt, (size_t)(N + 1) * 17 * 16 * sizeof(unsigned)); hipMalloc(&err, sizeof(unsigned));
    hipMemset(err, 0, sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"A ordinary launches", "B any-order + counters, poll sleep 2", "C any-order + counters, poll sleep 16",
                           "D any-order + counters, poll sleep 64"};
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f, sum = 0;
        for (int rep = 0; rep < REP + 2; ++rep) {
            hipMemsetAsync(cnt, 0, (size_t)(N + 1) * 17 * 16 * sizeof(unsigned), s);
            hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            for (int i = 0; i < N; ++i) {
                unsigned* mine = cnt + (size_t)i * 17 * 16;
                const unsigned* prev = i ? cnt + (size_t)(i - 1) * 17 * 16 : nullptr;
                const unsigned fl = i == 0 ? 0u : (unsigned)hipExtAnyOrderLaunch;
                if (mode == 0)
                    hipLaunchKernelGGL((k_link<false, 2>), dim3(wgs), dim3(256), 0, s, setup, work, (const unsigned*)nullptr, 0u, mine, err);
                else if (mode == 1)
                    hipExtLaunchKernelGGL((k_link<true, 2>), dim3(wgs), dim3(256), 0, s, nullptr, nullptr, fl, setup, work, prev, 0u, mine, err);
                else if (mode == 2)
                    hipExtLaunchKernelGGL((k_link<true, 16>), dim3(wgs), dim3(256), 0, s, nullptr, nullptr, fl, setup, work, prev, 0u, mine, err);
                else
                    hipExtLaunchKernelGGL((k_link<true, 64>), dim3(wgs), dim3(256), 0, s, nullptr, nullptr, fl, setup, work, prev, 0u, mine, err);
            }
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("%-44s chain of %d x %d workgroups: mean %.1f us, best %.1f us  (%.2f us / kernel)\n", names[mode], N, wgs, sum / REP * 1e3,
               best * 1e3, best * 1e3 / N);
    }
    unsigned h = 0; hipMemcpy(&h, err, sizeof(h), hipMemcpyDeviceToHost);
    printf("spin guard tripped %u times (must be 0)\n", h);
    return h != 0;
}
