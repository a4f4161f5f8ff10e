// Cycle-level microbenchmark (s_memtime) of MFMA / VALU / transcendental co-issue on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o coissue.bin coissue.hip && ./coissue.bin
// Every pattern is ONE inline-asm block (hipcc cannot re-order it) repeated ITERS times in a loop; per-wave cycles are
// read with s_memtime around the loop.  Patterns with role B run on waves >= 4 of an 8-wave workgroup (second wave of each SIMD).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

#define MF "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
#define FMA(n) "v_fma_f32 %" #n ", %" #n ", %11, %12\n\t"
#define EXP(n) "v_exp_f32 %" #n ", %" #n "\n\t"
// operands: 0 acc, 1 a, 2 b, 3..10 fillers f0..f7, 11 c0, 12 c1
#define OPS : "+v"(acc), "+v"(a), "+v"(b), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(c0), "v"(c1)

enum { P_MFMA = 0, P_MFMA_F2, P_MFMA_F4, P_MFMA_F6, P_MFMA_F8, P_MFMA_E1, P_MFMA_E2, P_MFMA_E4, P_MFMA_E1F3, P_MFMA_E2F2, P_VALU8, P_EXP8, P_IDLE,
       P_CHAIN12_THEN_VALU48, P_MIX48, P_TV_GROUPED, P_TV_ALT, P_TV_ALT2, P_COUNT };

template <int P>
__device__ __forceinline__ void body(f32x16& acc, f16x8& a, f16x8& b, float& f0, float& f1, float& f2, float& f3, float& f4, float& f5,
                                     float& f6, float& f7, float c0, float c1) {
    if constexpr (P == P_MFMA) asm volatile(MF MF MF MF OPS);
    if constexpr (P == P_MFMA_F2) asm volatile(MF FMA(3) FMA(4) MF FMA(5) FMA(6) MF FMA(7) FMA(8) MF FMA(9) FMA(10) OPS);
    if constexpr (P == P_MFMA_F4)
        asm volatile(MF FMA(3) FMA(4) FMA(5) FMA(6) MF FMA(7) FMA(8) FMA(9) FMA(10) MF FMA(3) FMA(4) FMA(5) FMA(6) MF FMA(7) FMA(8) FMA(9) FMA(10) OPS);
    if constexpr (P == P_MFMA_F6)
        asm volatile(MF FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) MF FMA(9) FMA(10) FMA(3) FMA(4) FMA(5) FMA(6) MF FMA(7) FMA(8) FMA(9) FMA(10) FMA(3) FMA(4)
                         MF FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) OPS);
    if constexpr (P == P_MFMA_F8)
        asm volatile(MF FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) MF FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10)
                         MF FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) MF FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) OPS);
    if constexpr (P == P_MFMA_E1) asm volatile(MF EXP(3) MF EXP(4) MF EXP(5) MF EXP(6) OPS);
    if constexpr (P == P_MFMA_E2) asm volatile(MF EXP(3) EXP(4) MF EXP(5) EXP(6) MF EXP(7) EXP(8) MF EXP(9) EXP(10) OPS);
    if constexpr (P == P_MFMA_E4) asm volatile(MF EXP(3) EXP(4) EXP(5) EXP(6) MF EXP(7) EXP(8) EXP(9) EXP(10) MF EXP(3) EXP(4) EXP(5) EXP(6) MF EXP(7) EXP(8) EXP(9) EXP(10) OPS);
    if constexpr (P == P_MFMA_E1F3) asm volatile(MF EXP(3) FMA(4) FMA(5) FMA(6) MF EXP(7) FMA(8) FMA(9) FMA(10) MF EXP(3) FMA(4) FMA(5) FMA(6) MF EXP(7) FMA(8) FMA(9) FMA(10) OPS);
    if constexpr (P == P_MFMA_E2F2) asm volatile(MF EXP(3) FMA(4) EXP(5) FMA(6) MF EXP(7) FMA(8) EXP(9) FMA(10) MF EXP(3) FMA(4) EXP(5) FMA(6) MF EXP(7) FMA(8) EXP(9) FMA(10) OPS);
    if constexpr (P == P_VALU8) asm volatile(FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) OPS);
    if constexpr (P == P_EXP8) asm volatile(EXP(3) EXP(4) EXP(5) EXP(6) EXP(7) EXP(8) EXP(9) EXP(10) EXP(3) EXP(4) EXP(5) EXP(6) EXP(7) EXP(8) EXP(9) EXP(10) OPS);
    if constexpr (P == P_IDLE) asm volatile("s_nop 0" OPS);
    // 8 transcendentals + 8 plain VALU, grouped / alternating / two plain per transcendental (16 exp + 32 fma would be the activation mix)
    if constexpr (P == P_TV_GROUPED) asm volatile(EXP(3) EXP(4) EXP(5) EXP(6) EXP(7) EXP(8) EXP(9) EXP(10) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) OPS);
    if constexpr (P == P_TV_ALT) asm volatile(EXP(3) FMA(7) EXP(4) FMA(8) EXP(5) FMA(9) EXP(6) FMA(10) EXP(7) FMA(3) EXP(8) FMA(4) EXP(9) FMA(5) EXP(10) FMA(6) OPS);
    if constexpr (P == P_TV_ALT2) asm volatile(EXP(3) FMA(7) FMA(8) EXP(4) FMA(9) FMA(10) EXP(5) FMA(7) FMA(8) EXP(6) FMA(9) FMA(10) EXP(3) FMA(7) FMA(8) EXP(4) FMA(9) FMA(10) EXP(5) FMA(7) FMA(8) EXP(6) FMA(9) FMA(10) OPS);
    // what the shipped kernel does: a dependent chain of 12 MFMAs, then 48 VALU (8 of them transcendental), no interleave
    if constexpr (P == P_CHAIN12_THEN_VALU48)
        asm volatile(MF MF MF MF MF MF MF MF MF MF MF MF
                     EXP(3) FMA(4) FMA(5) EXP(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(3) FMA(4) FMA(5) FMA(6) EXP(7) FMA(8) FMA(9) EXP(10)
                     FMA(3) FMA(4) FMA(5) FMA(6) EXP(7) FMA(8) FMA(9) EXP(10) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10)
                     EXP(3) FMA(4) FMA(5) EXP(6) FMA(7) FMA(8) FMA(9) FMA(10) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) FMA(8) FMA(9) FMA(10) OPS);
    // the same 12 MFMAs and 48 VALU, 4 VALU per MFMA gap
    if constexpr (P == P_MIX48)
        asm volatile(MF EXP(3) FMA(4) FMA(5) FMA(8) MF EXP(6) FMA(7) FMA(9) FMA(10) MF FMA(3) FMA(4) FMA(5) FMA(6) MF EXP(7) FMA(8) FMA(9) FMA(3)
                     MF EXP(10) FMA(4) FMA(5) FMA(6) MF EXP(7) FMA(8) FMA(9) FMA(3) MF EXP(10) FMA(4) FMA(5) FMA(6) MF FMA(7) FMA(8) FMA(9) FMA(10)
                     MF EXP(3) FMA(4) FMA(5) FMA(7) MF EXP(6) FMA(8) FMA(9) FMA(10) MF FMA(3) FMA(4) FMA(5) FMA(6) MF FMA(7) FMA(8) FMA(9) FMA(10) OPS);
}

template <int PA, int PB>
__global__ __launch_bounds__(1024) void k(unsigned long long* cyc, float* sink, int iters) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc = {0};
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.001f * (threadIdx.x & 63) + q); b[q] = (_Float16)(1.0f - 0.1f * q); }
    float f0 = 0.1f * threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float c0 = 0.5f, c1 = 0.25f;
    asm volatile("" : "+v"(c0), "+v"(c1));
    __syncthreads();
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    if ((wave & 4) == 0) {
        for (int i = 0; i < iters; ++i) body<PA>(acc, a, b, f0, f1, f2, f3, f4, f5, f6, f7, c0, c1);
    } else {
        for (int i = 0; i < iters; ++i) body<PB>(acc, a, b, f0, f1, f2, f3, f4, f5, f6, f7, c0, c1);
    }
    asm volatile("s_nop 15\n\ts_nop 7\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float r = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    for (int q = 0; q < 16; ++q) r += acc[q];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
        cyc[256 * 16 + blockIdx.x * (blockDim.x >> 6) + wave] = r1 - r0;   // 100 MHz ticks
    }
}

static unsigned long long* d_cyc;
static float* d_sink;

template <int PA, int PB>
void run(const char* name, int waves, int iters, double unitsA, double unitsB) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<PA, PB>), dim3(256), dim3(waves * 64), 0, 0, d_cyc, d_sink, iters);
        (void)hipDeviceSynchronize();
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PA, PB>), dim3(256), dim3(waves * 64), 0, 0, d_cyc, d_sink, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * waves), hr(256 * waves);
    (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hr.data(), d_cyc + 256 * 16, hr.size() * 8, hipMemcpyDeviceToHost);
    double clk = 0;
    for (size_t i = 0; i < h.size(); ++i) clk += (double)h[i] / (double)hr[i] * 0.1;
    clk /= h.size();
    std::vector<double> A, B;
    for (int g = 0; g < 256; ++g)
        for (int w = 0; w < waves; ++w) ((w & 4) ? B : A).push_back((double)h[g * waves + w] / iters);
    std::sort(A.begin(), A.end());
    printf("%-34s waves/WG %2d  A: %8.1f cyc/iter (%6.2f per unit)", name, waves, A[A.size() / 2], A[A.size() / 2] / unitsA);
    if (!B.empty()) { std::sort(B.begin(), B.end()); printf("  B: %8.1f cyc/iter (%6.2f per unit)", B[B.size() / 2], B[B.size() / 2] / unitsB); }
    printf("  wall %.3f ms  clock %.2f GHz\n", ms, clk);
}

int main() {
    (void)hipMalloc(&d_cyc, 2 * 256 * 16 * 8);
    (void)hipMalloc(&d_sink, 256 * 1024 * 4);
    const int it = 20000;
    printf("units: per MFMA for MFMA patterns, per VALU instruction for VALU patterns; s_memtime cycles\n");
    // one wave per SIMD
    run<P_MFMA, P_IDLE>("4 dependent mfma", 4, it, 4, 1);
    run<P_MFMA_F2, P_IDLE>("mfma + 2 fma", 4, it, 4, 1);
    run<P_MFMA_F4, P_IDLE>("mfma + 4 fma", 4, it, 4, 1);
    run<P_MFMA_F6, P_IDLE>("mfma + 6 fma", 4, it, 4, 1);
    run<P_MFMA_F8, P_IDLE>("mfma + 8 fma", 4, it, 4, 1);
    run<P_MFMA_E1, P_IDLE>("mfma + 1 exp", 4, it, 4, 1);
    run<P_MFMA_E2, P_IDLE>("mfma + 2 exp", 4, it, 4, 1);
    run<P_MFMA_E4, P_IDLE>("mfma + 4 exp", 4, it, 4, 1);
    run<P_MFMA_E1F3, P_IDLE>("mfma + 1 exp + 3 fma", 4, it, 4, 1);
    run<P_MFMA_E2F2, P_IDLE>("mfma + 2 exp + 2 fma", 4, it, 4, 1);
    run<P_VALU8, P_IDLE>("16 fma", 4, it, 16, 1);
    run<P_EXP8, P_IDLE>("16 exp", 4, it, 16, 1);
    run<P_TV_GROUPED, P_IDLE>("8 exp then 8 fma", 4, it, 16, 1);
    run<P_TV_ALT, P_IDLE>("exp fma alternating x8", 4, it, 16, 1);
    run<P_TV_ALT2, P_IDLE>("exp fma fma x8", 4, it, 24, 1);
    run<P_TV_GROUPED, P_TV_GROUPED>("8 exp then 8 fma, 2 waves/SIMD", 8, it, 16, 16);
    run<P_TV_ALT, P_TV_ALT>("exp fma alternating, 2 waves/SIMD", 8, it, 16, 16);
    run<P_TV_GROUPED, P_TV_GROUPED>("8 exp then 8 fma, 4 waves/SIMD", 16, it, 16, 16);
    run<P_TV_ALT, P_TV_ALT>("exp fma alternating, 4 waves/SIMD", 16, it, 16, 16);
    run<P_EXP8, P_EXP8>("16 exp, 4 waves/SIMD", 16, it, 16, 16);
    run<P_VALU8, P_VALU8>("16 fma, 4 waves/SIMD", 16, it, 16, 16);
    run<P_CHAIN12_THEN_VALU48, P_IDLE>("12 mfma then 48 valu (1 wave/SIMD)", 4, it, 12, 1);
    run<P_MIX48, P_IDLE>("12 mfma mixed 48 valu (1 wave/SIMD)", 4, it, 12, 1);
    // two waves per SIMD, different roles
    run<P_MFMA, P_VALU8>("A mfma | B fma", 8, it, 4, 16);
    run<P_MFMA, P_EXP8>("A mfma | B exp", 8, it, 4, 16);
    run<P_VALU8, P_MFMA>("A fma | B mfma", 8, it, 16, 4);
    run<P_MFMA, P_MFMA>("A mfma | B mfma", 8, it, 4, 4);
    run<P_VALU8, P_VALU8>("A fma | B fma", 8, it, 16, 16);
    run<P_EXP8, P_VALU8>("A exp | B fma", 8, it, 16, 16);
    run<P_CHAIN12_THEN_VALU48, P_CHAIN12_THEN_VALU48>("12 mfma then 48 valu, 2 waves/SIMD", 8, it, 12, 12);
    run<P_MIX48, P_MIX48>("12 mfma mixed 48 valu, 2 waves/SIMD", 8, it, 12, 12);
    // four waves per SIMD (16-wave workgroup: waves 0-3, 8-11 role A; 4-7, 12-15 role B)
    run<P_CHAIN12_THEN_VALU48, P_CHAIN12_THEN_VALU48>("12 mfma then 48 valu, 4 waves/SIMD", 16, it, 12, 12);
    run<P_MIX48, P_MIX48>("12 mfma mixed 48 valu, 4 waves/SIMD", 16, it, 12, 12);
    run<P_MFMA, P_VALU8>("A mfma | B fma, 4 waves/SIMD", 16, it, 4, 16);
    return 0;
}
