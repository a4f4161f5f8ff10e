// How the VALU issue port of a gfx950 SIMD is shared: rates of instruction kinds at 1 ... 8 waves per SIMD, transcendental and plain
// streams from DIFFERENT waves of one SIMD (do they overlap?), and the activation sequence of k_mfma (exp, add, rcp, cvt_pk, fma_mix) with
// and without f16 MFMAs in the stream.  Output: SIMD cycles (s_memtime) per wave-instruction, median over the waves.
// build: hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define OPS : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(c0), "v"(c1)
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n\t"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n\t"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define I_ADD(n) "v_add_f32 %" #n ", 1.0, %" #n "\n\t"
#define I_CVT(n) "v_cvt_pk_f16_f32 %" #n ", %" #n ", %16\n\t"
#define I_MIXLO(n) "v_fma_mixlo_f16 %" #n ", %" #n ", -1.0, %16 op_sel_hi:[1,0,0]\n\t"
#define I_AND(n) "v_and_b32 %" #n ", 0xffffe000, %" #n "\n\t"
#define I_SUB(n) "v_sub_f32 %" #n ", %16, %" #n "\n\t"
// alternating exp / fma inside one wave
#define I_EF(n) "v_exp_f32 %" #n ", %" #n "\n\tv_fma_f32 %" #n ", %" #n ", %16, %17\n\t"

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// P: 0 fma, 1 exp, 2 rcp, 3 cvt_pk, 4 mixlo, 5 exp+fma alternating in one wave (32 instructions per iteration),
//    6 roles by SIMD slot: slots 0, 2, .. exp only, slots 1, 3, .. fma only, 7 activation sequence of 16 values (16 exp, 16 add, 16 rcp,
//    8 cvt, 16 mix = 72), 8 the same + 6 f16 MFMAs (two K steps of three products), 9 truncation split variant (16 exp, 16 add, 16 rcp,
//    16 and, 16 sub, 16 cvt = 96), 10: 9 + 6 MFMAs, 11: bare MFMAs (6 per iteration)
template <int P>
__global__ void __launch_bounds__(1024) k(unsigned long long* cyc, int iters, float c0, float c1) {
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = c0 * (threadIdx.x + i);
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    const int slot = threadIdx.x >> 8;   // waves w, w + 4, .. share a SIMD: slot = position among them
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (P == 0) asm volatile(R16(I_FMA) OPS);
        if constexpr (P == 1) asm volatile(R16(I_EXP) OPS);
        if constexpr (P == 2) asm volatile(R16(I_RCP) OPS);
        if constexpr (P == 3) asm volatile(R16(I_CVT) OPS);
        if constexpr (P == 4) asm volatile(R16(I_MIXLO) OPS);
        if constexpr (P == 5) asm volatile(R16(I_EF) OPS);
        if constexpr (P == 6) {
            if (slot & 1) asm volatile(R16(I_FMA) OPS);
            else asm volatile(R16(I_EXP) OPS);
        }
        if constexpr (P == 7 || P == 8) {
            if constexpr (P == 8) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            }
            asm volatile(R16(I_EXP) R16(I_ADD) R16(I_RCP) OPS);
            if constexpr (P == 8) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            }
            asm volatile(I_CVT(0) I_CVT(2) I_CVT(4) I_CVT(6) I_CVT(8) I_CVT(10) I_CVT(12) I_CVT(14) R16(I_MIXLO) OPS);
        }
        if constexpr (P == 9 || P == 10) {
            if constexpr (P == 10) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            }
            asm volatile(R16(I_EXP) R16(I_ADD) R16(I_RCP) OPS);
            if constexpr (P == 10) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            }
            asm volatile(R16(I_AND) R16(I_SUB) R16(I_CVT) OPS);
        }
        if constexpr (P == 11) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i] + acc[i];
    if (s == 123.456f) cyc[0] = 1;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int P>
void run(const char* name, int per_iter) {
    unsigned long long* d;
    (void)hipMalloc(&d, 512 * 16 * 8);
    const int iters = 512;
    printf("%-44s", name);
    // waves per SIMD: 1, 2, 3, 4 with one workgroup per CU; 6 and 8 with two workgroups of 12 / 16 waves per CU (grid 512; needs <= 64 VGPRs... the
    // kernel holds ~45, so two 1024-thread workgroups are co-resident)
    struct Cfg { int per_simd, waves, grid; };
    for (Cfg c : {Cfg{1, 4, 256}, Cfg{2, 8, 256}, Cfg{4, 16, 256}, Cfg{6, 12, 512}, Cfg{8, 16, 512}}) {
        (void)hipMemset(d, 0, 512 * 16 * 8);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<P><<<c.grid, c.waves * 64>>>(d, iters, 1.0001f, 0.5f);   // warm
        (void)hipEventRecord(e0);
        k<P><<<c.grid, c.waves * 64>>>(d, iters, 1.0001f, 0.5f);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(512 * 16);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int g = 0; g < c.grid; ++g) for (int w = 0; w < c.waves; ++w) v.push_back((double)h[g * 16 + w] / iters);
        std::sort(v.begin(), v.end());
        // cycles the SIMD spends per wave-instruction: a wave's loop time / (instructions per iteration x waves sharing the SIMD)
        const double per_simd = v[v.size() / 2] / ((double)per_iter * c.per_simd);
        // the same from the wall clock at a nominal 2.4 GHz (the kernel is 1024 SIMDs busy for ms): shows the clock the chip really held
        const double wall = ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * c.per_simd);
        printf("  %d/SIMD %5.2f (%5.2f)", c.per_simd, per_simd, wall);
    }
    printf("\n");
    (void)hipFree(d);
}

int main() {
    printf("SIMD cycles per wave-instruction by s_memtime (in brackets: from the wall clock at a nominal 2.4 GHz)\n");
    run<0>("v_fma_f32", 16); run<1>("v_exp_f32", 16); run<2>("v_rcp_f32", 16); run<3>("v_cvt_pk_f16_f32", 16); run<4>("v_fma_mixlo_f16", 16);
    run<5>("exp, fma alternating in ONE wave", 32);
    run<6>("exp waves beside fma waves on one SIMD", 16);
    run<7>("activation x16: exp add rcp cvt mix (72)", 72);
    run<8>("  + 6 f16 MFMA 32x32x16 (78)", 78);
    run<9>("activation x16, truncation split (96)", 96);
    run<10>("  + 6 f16 MFMA 32x32x16 (102)", 102);
    run<11>("bare f16 MFMA 32x32x16 x6", 6);
    return 0;
}
