// mfma_hazard.hip -- how many wait states does gfx950 need between an MFMA and the first VALU read of its result,
// and what does an `s_nop N` really cost?  (conv.hip mfma_result_guard; DESIGN.md "MFMA result hazard".)
//
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/mfma_hazard.hip -o /tmp/mfma_hazard && /tmp/mfma_hazard
//
// Part 1 (timing): one wave, s_memtime around 256 copies of an instruction -> shader cycles per instruction.
// Part 2 (hazard): every wave of a 512-thread workgroup (2 waves per SIMD, so the matrix pipe is contended) runs
//   acc = mfma(a, b, acc)   twice (two accumulators, like the conv kernels), then PAD, then reads the LAST written
//   accumulator with v_accvgpr_read_b32, all inside ONE asm statement (hipcc adds nothing).  The result is compared
//   with the same sequence padded by 48 wait states.  Output: wrong reads per variant.
//   PAD kinds: 0 = one `s_nop K-1`;  1 = K x `s_nop 0`;  2 = K x `v_nop`;  3 = `s_nop K-1-4` + 4 reads of the OTHER
//   accumulator first (the shape hipcc emitted in k_conv3x3_res<float,...>).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------ timing
template <int WHAT>
__global__ void k_time(unsigned long long* out) {
    unsigned long long t0, t1;
    float x = (float)threadIdx.x;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if constexpr (WHAT == 0) asm volatile(".rept 256\n\ts_nop 0\n\t.endr" ::: "memory");
    if constexpr (WHAT == 1) asm volatile(".rept 256\n\ts_nop 7\n\t.endr" ::: "memory");
    if constexpr (WHAT == 2) asm volatile(".rept 256\n\ts_nop 15\n\t.endr" ::: "memory");
    if constexpr (WHAT == 3) asm volatile(".rept 256\n\tv_nop\n\t.endr" ::: "memory");
    if constexpr (WHAT == 4) asm volatile(".rept 256\n\tv_mov_b32 %0, %0\n\t.endr" : "+v"(x) :: "memory");
    if constexpr (WHAT == 5) asm volatile(".rept 256\n\tv_accvgpr_read_b32 %0, a0\n\t.endr" : "+v"(x) :: "memory", "a0");
    if constexpr (WHAT == 6) asm volatile(".rept 256\n\ts_nop 3\n\t.endr" ::: "memory");
    if constexpr (WHAT == 7) asm volatile(".rept 256\n\ts_nop 1\n\t.endr" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (x == -1.0f) out[1] = 0;
}

// ------------------------------------------------------------------------------------------------ hazard
// DT 0: v_mfma_f32_16x16x4_f32 (f32 in, "SGEMM", 8 passes); DT 1: v_mfma_f32_16x16x32_bf16 (XDL, 4 passes)
#define MFMA_F32 "v_mfma_f32_16x16x4_f32"
#define MFMA_BF "v_mfma_f32_16x16x32_bf16"

#define PRELOAD                                                                                   \
    "v_accvgpr_write_b32 a0, %[c]\n\tv_accvgpr_write_b32 a1, %[c]\n\tv_accvgpr_write_b32 a2, %[c]\n\t" \
    "v_accvgpr_write_b32 a3, %[c]\n\tv_accvgpr_write_b32 a4, %[c]\n\tv_accvgpr_write_b32 a5, %[c]\n\t" \
    "v_accvgpr_write_b32 a6, %[c]\n\tv_accvgpr_write_b32 a7, %[c]\n\ts_nop 7\n\t"
#define READ_LAST "v_accvgpr_read_b32 %[r0], a4\n\tv_accvgpr_read_b32 %[r1], a5\n\tv_accvgpr_read_b32 %[r2], a6\n\tv_accvgpr_read_b32 %[r3], a7\n\t"
#define READ_OTHER "v_accvgpr_read_b32 %[q0], a3\n\tv_accvgpr_read_b32 %[q1], a2\n\tv_accvgpr_read_b32 %[q2], a1\n\tv_accvgpr_read_b32 %[q3], a0\n\t"
#define OUTS [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3)
#define CLOB "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "memory"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int DT, int KIND, int K>
__device__ __forceinline__ void one_test(float a, float b, u32x4 a8, u32x4 b8, float c, float (&r)[4]) {
    float r0, r1, r2, r3, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    // chain: I7 a[0:3] += a*b ; I8 a[4:7] += a*b (LAST) ; pad ; read a[4:7]
#define BODY(MF, A, B, PAD)                                                                           \
    asm volatile(PRELOAD MF " a[0:3], " A ", " B ", a[0:3]\n\t" MF " a[4:7], " A ", " B ", a[4:7]\n\t"  \
                 MF " a[0:3], " A ", " B ", a[0:3]\n\t" MF " a[4:7], " A ", " B ", a[4:7]\n\t" PAD      \
                 : OUTS : [a] "v"(a), [b] "v"(b), [a8] "v"(a8), [b8] "v"(b8), [c] "v"(c), [k] "i"(K), [k1] "i"(K > 0 ? K - 1 : 0), [k5] "i"(K > 4 ? K - 5 : 0) : CLOB)
#define PAD0 "s_nop %c[k1]\n\t" READ_LAST
#define PAD1 ".rept %c[k]\n\ts_nop 0\n\t.endr\n\t" READ_LAST
#define PAD2 ".rept %c[k]\n\tv_nop\n\t.endr\n\t" READ_LAST
#define PAD3 "s_nop %c[k5]\n\t" READ_OTHER READ_LAST
#define PADZ READ_LAST
    if constexpr (DT == 0) {
        if constexpr (K == 0) BODY(MFMA_F32, "%[a]", "%[b]", PADZ);
        else if constexpr (KIND == 0) BODY(MFMA_F32, "%[a]", "%[b]", PAD0);
        else if constexpr (KIND == 1) BODY(MFMA_F32, "%[a]", "%[b]", PAD1);
        else if constexpr (KIND == 2) BODY(MFMA_F32, "%[a]", "%[b]", PAD2);
        else BODY(MFMA_F32, "%[a]", "%[b]", PAD3);
    } else {
        if constexpr (K == 0) BODY(MFMA_BF, "%[a8]", "%[b8]", PADZ);
        else if constexpr (KIND == 0) BODY(MFMA_BF, "%[a8]", "%[b8]", PAD0);
        else if constexpr (KIND == 1) BODY(MFMA_BF, "%[a8]", "%[b8]", PAD1);
        else if constexpr (KIND == 2) BODY(MFMA_BF, "%[a8]", "%[b8]", PAD2);
        else BODY(MFMA_BF, "%[a8]", "%[b8]", PAD3);
    }
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
    asm volatile("" :: "v"(q0), "v"(q1), "v"(q2), "v"(q3));
}

template <int DT>
__device__ __forceinline__ void ref_test(float a, float b, u32x4 a8, u32x4 b8, float c, float (&r)[4]) {
    float r0, r1, r2, r3, q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    constexpr int K = 0;
#define PADR "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t" READ_LAST
    if constexpr (DT == 0) BODY(MFMA_F32, "%[a]", "%[b]", PADR);
    else BODY(MFMA_BF, "%[a8]", "%[b8]", PADR);
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3;
    asm volatile("" :: "v"(q0), "v"(q1), "v"(q2), "v"(q3));
}

// counts[0..3]: wrong reads of register 0..3 of the last-written accumulator; counts[4]: tests
template <int DT, int KIND, int K>
__global__ __launch_bounds__(512) void k_hazard(unsigned long long* counts, int iters) {
    const int lane = threadIdx.x & 63;
    const float a = 1.0f + (float)(lane & 3), b = 1.0f + (float)((lane >> 2) & 3);
    // bf16 operands: 8 x the same small integer per lane (0x3F80 = 1.0, 0x4000 = 2.0)
    const unsigned ha = (lane & 1) ? 0x40004000u : 0x3F803F80u, hb = (lane & 2) ? 0x40004000u : 0x3F803F80u;
    const u32x4 a8 = {ha, ha, ha, ha}, b8 = {hb, hb, hb, hb};
    unsigned long long bad[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const float c = (float)(it & 7);
        float ref[4], got[4];
        ref_test<DT>(a, b, a8, b8, c, ref);
        // desynchronise the waves sharing a SIMD: a wave-dependent number of idle slots
        const int skew = (threadIdx.x >> 6) * 3 + (it % 5);
        for (int s = 0; s < skew; ++s) asm volatile("s_nop 1");
        one_test<DT, KIND, K>(a, b, a8, b8, c, got);
#pragma unroll
        for (int q = 0; q < 4; ++q) bad[q] += (__float_as_uint(ref[q]) != __float_as_uint(got[q])) ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (bad[q]) atomicAdd(&counts[q], bad[q]);
    if (threadIdx.x == 0 && blockIdx.x == 0) counts[4] = (unsigned long long)gridDim.x * blockDim.x * iters;
}

template <int DT, int KIND, int K>
void run(unsigned long long* d, const char* what) {
    CK(hipMemset(d, 0, 8 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((k_hazard<DT, KIND, K>), dim3(512), dim3(512), 0, 0, d, 200);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-8s kind %d K=%2d : wrong reads of regs [a4 a5 a6 a7] = %llu %llu %llu %llu of %llu lane-tests each\n", what, KIND, K, h[0],
           h[1], h[2], h[3], h[4]);
}

template <int DT, int KIND, int K0, int K1>
void sweep(unsigned long long* d, const char* what) {
    if constexpr (K0 <= K1) {
        run<DT, KIND, K0>(d, what);
        sweep<DT, KIND, K0 + 1, K1>(d, what);
    }
}


// ------------------------------------------------------------------------------------------------ rotated chains
// hipcc's register allocator sometimes ROTATES the accumulators of a chain (vDst != SrcC):
//     I5  a[4:7]  = mfma(.., a[4:7])
//     I6  a[8:11] = mfma(.., a[0:3])
//     I7  a[0:3]  = mfma(.., a[4:7])
//     I8  a[4:7]  = mfma(.., a[8:11])        <- k_conv3x3_res<float,16,4> without mfma_result_guard
// LLVM treats "SrcC == an earlier vDst exactly" as the back-to-back accumulate case (0 wait states) whatever the
// consumer's own vDst is.  ROT 1: that chain, then PAD (K x s_nop 0), then reads of a7..a4 (I8) -- are THEY interlocked?
// ROT 2: the SrcC hand-over itself: a[4:7] = mfma(.., a[4:7]); a[0:3] = mfma(.., a[4:7]) with K x s_nop 0 between them.
// Reference: the same chain with 48 wait states between all instructions.
#define PRELOAD12                                                                                  \
    "v_accvgpr_write_b32 a0, %[c]\n\tv_accvgpr_write_b32 a1, %[c]\n\tv_accvgpr_write_b32 a2, %[c]\n\t" \
    "v_accvgpr_write_b32 a3, %[c]\n\tv_accvgpr_write_b32 a4, %[c2]\n\tv_accvgpr_write_b32 a5, %[c2]\n\t" \
    "v_accvgpr_write_b32 a6, %[c2]\n\tv_accvgpr_write_b32 a7, %[c2]\n\tv_accvgpr_write_b32 a8, %[c]\n\t" \
    "v_accvgpr_write_b32 a9, %[c]\n\tv_accvgpr_write_b32 a10, %[c]\n\tv_accvgpr_write_b32 a11, %[c]\n\ts_nop 7\n\t"
#define BIGPAD "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define READ8 "v_accvgpr_read_b32 %[r3], a7\n\tv_accvgpr_read_b32 %[r2], a6\n\tv_accvgpr_read_b32 %[r1], a5\n\tv_accvgpr_read_b32 %[r0], a4\n\t" \
              "v_accvgpr_read_b32 %[q3], a3\n\tv_accvgpr_read_b32 %[q2], a2\n\tv_accvgpr_read_b32 %[q1], a1\n\tv_accvgpr_read_b32 %[q0], a0\n\t"
#define CLOB12 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "memory"
#define KPAD ".rept %c[k]\n\ts_nop 0\n\t.endr\n\t"

template <int DT, int ROT, int K, bool REF>
__device__ __forceinline__ void rot_test(float a, float b, u32x4 a8, u32x4 b8, float c, float (&r)[8]) {
    float r0, r1, r2, r3, q0, q1, q2, q3;
#define ROTBODY(MF, A, B, P56, P67, P78, P8R)                                                                  \
    asm volatile(PRELOAD12 MF " a[4:7], " A ", " B ", a[4:7]\n\t" P56 MF " a[8:11], " A ", " B ", a[0:3]\n\t" P67 \
                 MF " a[0:3], " A ", " B ", a[4:7]\n\t" P78 MF " a[4:7], " A ", " B ", a[8:11]\n\t" P8R READ8     \
                 : OUTS : [a] "v"(a), [b] "v"(b), [a8] "v"(a8), [b8] "v"(b8), [c] "v"(c), [c2] "v"(c + 0.5f), [k] "i"(K) : CLOB12)
#define ROT2BODY(MF, A, B, P, PR)                                                                              \
    asm volatile(PRELOAD12 MF " a[4:7], " A ", " B ", a[4:7]\n\t" P MF " a[0:3], " A ", " B ", a[4:7]\n\t" PR READ8 \
                 : OUTS : [a] "v"(a), [b] "v"(b), [a8] "v"(a8), [b8] "v"(b8), [c] "v"(c), [c2] "v"(c + 0.5f), [k] "i"(K) : CLOB12)
    // ROT 3: rotated producer, then IMMEDIATELY the in-place terminator conv.hip's mfma_result_guard() uses
    // (acc = 0 * 0 + acc), K x s_nop 0, then the reads
#define ROT3BODY(MF, A, B, Z, P, PR)                                                                           \
    asm volatile(PRELOAD12 MF " a[4:7], " A ", " B ", a[8:11]\n\t" P MF " a[4:7], " Z ", " Z ", a[4:7]\n\t" PR READ8 \
                 : OUTS : [a] "v"(a), [b] "v"(b), [a8] "v"(a8), [b8] "v"(b8), [c] "v"(c), [c2] "v"(c + 0.5f), [k] "i"(K), [z] "v"(0.0f), [z8] "v"((u32x4){0u, 0u, 0u, 0u}) : CLOB12)
    if constexpr (ROT == 3) {
        if constexpr (DT == 0) { if constexpr (REF) ROT3BODY(MFMA_F32, "%[a]", "%[b]", "%[z]", BIGPAD, BIGPAD); else ROT3BODY(MFMA_F32, "%[a]", "%[b]", "%[z]", "", KPAD); }
        else { if constexpr (REF) ROT3BODY(MFMA_BF, "%[a8]", "%[b8]", "%[z8]", BIGPAD, BIGPAD); else ROT3BODY(MFMA_BF, "%[a8]", "%[b8]", "%[z8]", "", KPAD); }
    } else
    if constexpr (DT == 0) {
        if constexpr (ROT == 1) { if constexpr (REF) ROTBODY(MFMA_F32, "%[a]", "%[b]", BIGPAD, BIGPAD, BIGPAD, BIGPAD); else ROTBODY(MFMA_F32, "%[a]", "%[b]", "", "", "", KPAD); }
        else { if constexpr (REF) ROT2BODY(MFMA_F32, "%[a]", "%[b]", BIGPAD, BIGPAD); else ROT2BODY(MFMA_F32, "%[a]", "%[b]", KPAD, BIGPAD); }
    } else {
        if constexpr (ROT == 1) { if constexpr (REF) ROTBODY(MFMA_BF, "%[a8]", "%[b8]", BIGPAD, BIGPAD, BIGPAD, BIGPAD); else ROTBODY(MFMA_BF, "%[a8]", "%[b8]", "", "", "", KPAD); }
        else { if constexpr (REF) ROT2BODY(MFMA_BF, "%[a8]", "%[b8]", BIGPAD, BIGPAD); else ROT2BODY(MFMA_BF, "%[a8]", "%[b8]", KPAD, BIGPAD); }
    }
    r[0] = r0; r[1] = r1; r[2] = r2; r[3] = r3; r[4] = q0; r[5] = q1; r[6] = q2; r[7] = q3;
}

template <int DT, int ROT, int K>
__global__ __launch_bounds__(1024) void k_rot(unsigned long long* counts, int iters) {
    const int lane = threadIdx.x & 63;
    const float a = 1.0f + (float)(lane & 3), b = 1.0f + (float)((lane >> 2) & 3);
    const unsigned ha = (lane & 1) ? 0x40004000u : 0x3F803F80u, hb = (lane & 2) ? 0x40004000u : 0x3F803F80u;
    const u32x4 a8 = {ha, ha, ha, ha}, b8 = {hb, hb, hb, hb};
    unsigned long long bad47 = 0, bad03 = 0;
    for (int it = 0; it < iters; ++it) {
        const float c = (float)(it & 7);
        float ref[8], got[8];
        rot_test<DT, ROT, K, true>(a, b, a8, b8, c, ref);
        const int skew = (threadIdx.x >> 6) * 3 + (it % 5);
        for (int s = 0; s < skew; ++s) asm volatile("s_nop 1");
        rot_test<DT, ROT, K, false>(a, b, a8, b8, c, got);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bad47 += (__float_as_uint(ref[q]) != __float_as_uint(got[q])) ? 1 : 0;
            bad03 += (__float_as_uint(ref[4 + q]) != __float_as_uint(got[4 + q])) ? 1 : 0;
        }
    }
    if (bad47) atomicAdd(&counts[0], bad47);
    if (bad03) atomicAdd(&counts[1], bad03);
    if (threadIdx.x == 0 && blockIdx.x == 0) counts[4] = (unsigned long long)gridDim.x * blockDim.x * iters * 4;
}

int g_block = 512;     // 512 threads = 2 waves per SIMD, 1024 = 4 (two workgroups per CU: 8)
template <int DT, int ROT, int K>
void run_rot(unsigned long long* d, const char* what) {
    CK(hipMemset(d, 0, 8 * sizeof(unsigned long long)));
    hipLaunchKernelGGL((k_rot<DT, ROT, K>), dim3(512), dim3(g_block), 0, 0, d, 200);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-5s rotated chain %d, %4d-thread blocks, K=%2d x s_nop 0 : wrong reads of a[4:7] = %llu, of a[0:3] = %llu (of %llu each)\n",
           what, ROT, g_block, K, h[0], h[1], h[4]);
}
template <int DT, int ROT, int K0, int K1>
void sweep_rot(unsigned long long* d, const char* what) {
    if constexpr (K0 <= K1) {
        run_rot<DT, ROT, K0>(d, what);
        sweep_rot<DT, ROT, K0 + 1, K1>(d, what);
    }
}

template <int WHAT>
void time_one(unsigned long long* d, const char* what) {
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL((k_time<WHAT>), dim3(1), dim3(64), 0, 0, d);
        CK(hipDeviceSynchronize());
        unsigned long long h;
        CK(hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost));
        if (h < best) best = h;
    }
    printf("timing  %-22s : %6.2f shader cycles per instruction (one wave, 256 copies, incl. ~40 of stamp cost)\n", what, best / 256.0);
}

int main() {
    unsigned long long* d;
    CK(hipMalloc(&d, 8 * sizeof(unsigned long long)));
    time_one<0>(d, "s_nop 0");
    time_one<7>(d, "s_nop 1");
    time_one<6>(d, "s_nop 3");
    time_one<1>(d, "s_nop 7");
    time_one<2>(d, "s_nop 15");
    time_one<3>(d, "v_nop");
    time_one<4>(d, "v_mov_b32 (dependent)");
    time_one<5>(d, "v_accvgpr_read_b32");
    printf("--- f32 16x16x4 (SGEMM, 8 passes): LLVM table = 10 wait states to a VALU read\n");
    sweep<0, 1, 0, 16>(d, "f32");
    sweep<0, 0, 1, 16>(d, "f32");
    sweep<0, 2, 1, 16>(d, "f32");
    sweep<0, 3, 5, 16>(d, "f32");
    printf("--- bf16 16x16x32 (XDL, 4 passes on paper; 16 cycles issue): LLVM table = 7 (4-pass) / 11 (8-pass)\n");
    sweep<1, 1, 0, 16>(d, "bf16");
    sweep<1, 0, 1, 16>(d, "bf16");
    sweep<1, 3, 5, 16>(d, "bf16");
    printf("--- rotated accumulators (vDst != SrcC), ROT 1 = pad between the LAST MFMA and the reads, ROT 2 = pad between producer and SrcC consumer\n");
    sweep_rot<0, 1, 0, 24>(d, "f32");
    sweep_rot<0, 2, 0, 12>(d, "f32");
    sweep_rot<1, 1, 0, 16>(d, "bf16");
    sweep_rot<1, 2, 0, 12>(d, "bf16");
    g_block = 1024;
    printf("--- the same with 1024-thread workgroups (4-8 waves per SIMD): does the requirement grow with contention?\n");
    sweep_rot<0, 1, 6, 24>(d, "f32");
    sweep_rot<1, 1, 4, 16>(d, "bf16");
    g_block = 512;
    printf("--- ROT 3 = rotated producer, then immediately the in-place zero-operand terminator, K x s_nop 0, reads of a[4:7]\n");
    sweep_rot<0, 3, 0, 12>(d, "f32");
    sweep_rot<1, 3, 0, 12>(d, "bf16");
    return 0;
}
