// Micro-test of the LDS-DMA load `buffer_load_dwordx4 ... offen lds` on gfx950 (what k_wgrad3x3_dma relies on):
//   1. destination = M0 base + lane * 16 (lane-linear, 1 KiB per wave instruction), for bases beyond 64 KiB
//   2. lanes whose offset is out of the descriptor's range WRITE ZEROS (they do not skip the LDS write)
//   3. the scalar offset operand adds to the source address only
// hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_dma.hip -o /tmp/lds_dma && /tmp/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(int4v rs, int voff, int soff, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(lds) : "memory");
}

// 256 threads; LDS filled with 0xAB; each wave DMAs 1 KiB to base + wave * 1024; lanes with (lane % 5 == 4) are out of range
__global__ void k(const char* src, int nbytes, int soff, unsigned base, int lds_total, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x * 4; i < lds_total; i += 256 * 4) *reinterpret_cast<unsigned*>(smem + i) = 0xABABABABu;
    __syncthreads();
    int4v rs;
    const unsigned long long a = (unsigned long long)src;
    rs.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    rs.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffff));
    rs.z = __builtin_amdgcn_readfirstlane(nbytes);
    rs.w = __builtin_amdgcn_readfirstlane(0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int voff = (lane % 5 == 4) ? 0x7fffff00 : (int)threadIdx.x * 16;
    dma16(rs, voff, __builtin_amdgcn_readfirstlane(soff), __builtin_amdgcn_readfirstlane(base + wave * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = threadIdx.x; i < lds_total / 4; i += 256) out[i] = *reinterpret_cast<unsigned*>(smem + 4 * i);
}

int main() {
    const int N = 8192, LDS = 160 * 1024;
    std::vector<unsigned> h(N / 4);
    for (int i = 0; i < N / 4; ++i) h[i] = 0x1000000u + i;
    char* d; unsigned* o;
    hipMalloc(&d, N); hipMalloc(&o, LDS);
    hipMemcpy(d, h.data(), N, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    int bad_total = 0;
    for (unsigned base : {0u, 4096u, 61440u, 66560u, 100352u, 159744u - 3072u}) {
        for (int soff : {0, 4096}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(256), LDS, 0, d, N, soff, base, LDS, o);
            std::vector<unsigned> r(LDS / 4);
            hipMemcpy(r.data(), o, LDS, hipMemcpyDeviceToHost);
            int bad = 0, zero_lanes = 0, skipped = 0;
            for (int w = 0; w < LDS / 4; ++w) {
                const long byte = 4L * w - base;
                unsigned want = 0xABABABABu;
                if (byte >= 0 && byte < 4096) {
                    const int t = (int)(byte / 16), lane = t & 63;
                    if (lane % 5 == 4) { want = 0u; if (r[w] == 0u) ++zero_lanes; else if (r[w] == 0xABABABABu) ++skipped; }
                    else want = 0x1000000u + (unsigned)((t * 16 + soff + byte % 16) / 4);
                }
                if (r[w] != want) ++bad;
            }
            printf("base %6u soff %4d: mismatches %d (out-of-range lanes: %d words zeroed, %d words left untouched)\n", base, soff, bad,
                   zero_lanes, skipped);
            bad_total += bad;
        }
    }
    printf(bad_total ? "FAIL\n" : "OK\n");
    return bad_total ? 1 : 0;
}
