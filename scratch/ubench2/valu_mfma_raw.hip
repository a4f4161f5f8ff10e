// Fourth reproducer for DESIGN.md §9: VALU write of an MFMA source register followed K wait states later by the MFMA that reads it.
// hipcc (ROCm 7.2) pads this read-after-write with 2 wait states.  Test waves (waves 0-3 of a 16-wave workgroup, one per SIMD):
//   B := producer(val(i)); K x s_nop; MFMA(acc = A*B + 0); wait; chk += acc[0..15]          (val alternates 1.0 / 2.0 every iteration)
// Hammer waves (the other 12) run a loop of transcendental / packed / plain VALU or MFMA instructions on the same SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define ADD16 "v_add_f32 %0, %0, v48\n\tv_add_f32 %0, %0, v49\n\tv_add_f32 %0, %0, v50\n\tv_add_f32 %0, %0, v51\n\t" \
              "v_add_f32 %0, %0, v52\n\tv_add_f32 %0, %0, v53\n\tv_add_f32 %0, %0, v54\n\tv_add_f32 %0, %0, v55\n\t" \
              "v_add_f32 %0, %0, v56\n\tv_add_f32 %0, %0, v57\n\tv_add_f32 %0, %0, v58\n\tv_add_f32 %0, %0, v59\n\t" \
              "v_add_f32 %0, %0, v60\n\tv_add_f32 %0, %0, v61\n\tv_add_f32 %0, %0, v62\n\tv_add_f32 %0, %0, v63\n\t"
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", \
             "v59", "v60", "v61", "v62", "v63", "memory"

// PROD 0: v_mov_b32 (packed halves), 1: v_cvt_pk_f16_f32 (from two floats), 2: v_fma_mixlo_f16 + v_fma_mixhi_f16, 3: v_pk_mul_f16
// HAM 0: none (all 16 waves test), 1: v_exp_f32, 2: v_pk_fma_f32, 3: v_fma_f32, 4: mfma, 5: v_exp + mfma mix
template <int PROD, int K, int HAM>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float chk = 0.0f;
    unsigned one = 0x3C003C00u;
    float fone = 1.0f, fzero = 0.0f;
    asm volatile("" : "+v"(one), "+v"(fone), "+v"(fzero));
    if (HAM != 0 && wave >= 4) {
        f32x16 acc = {0};
        u32x4 a = {one, one, one, one}, b = a;
        float f0 = 0.1f * threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
        for (int i = 0; i < iters * 3; ++i) {
            if (HAM == 1 || HAM == 5)
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
            if (HAM == 2)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %0\n\tv_pk_fma_f32 %1, %1, %0, %1"
                             : "+v"(*(double*)&f0), "+v"(*(double*)&f2));
            if (HAM == 3)
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t"
                             "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
            if (HAM == 4 || HAM == 5)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15" : "+v"(acc));
        chk = (acc[0] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) * 0.0f;
    } else {
        for (int i = 0; i < iters; ++i) {
            const unsigned bval = (i & 1) ? 0x40004000u : 0x3C003C00u;   // 2.0 : 1.0 (packed halves)
            const float fval = (i & 1) ? 2.0f : 1.0f;
            asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %1\n\t"
                         "s_nop 7\n\t"
                         ".if %c4 == 0\n\tv_mov_b32 v44, %2\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %2\n\tv_mov_b32 v47, %2\n\t.endif\n\t"
                         ".if %c4 == 1\n\tv_cvt_pk_f16_f32 v44, %3, %3\n\tv_cvt_pk_f16_f32 v45, %3, %3\n\tv_cvt_pk_f16_f32 v46, %3, %3\n\tv_cvt_pk_f16_f32 v47, %3, %3\n\t.endif\n\t"
                         ".if %c4 == 2\n\tv_fma_mixlo_f16 v44, %3, %6, %7\n\tv_fma_mixlo_f16 v45, %3, %6, %7\n\tv_fma_mixlo_f16 v46, %3, %6, %7\n\tv_fma_mixlo_f16 v47, %3, %6, %7\n\t"
                         "v_fma_mixhi_f16 v44, %3, %6, %7\n\tv_fma_mixhi_f16 v45, %3, %6, %7\n\tv_fma_mixhi_f16 v46, %3, %6, %7\n\tv_fma_mixhi_f16 v47, %3, %6, %7\n\t.endif\n\t"
                         ".if %c4 == 3\n\tv_pk_mul_f16 v44, %2, %1\n\tv_pk_mul_f16 v45, %2, %1\n\tv_pk_mul_f16 v46, %2, %1\n\tv_pk_mul_f16 v47, %2, %1\n\t.endif\n\t"
                         ".rept %c5\n\ts_nop 0\n\t.endr\n\t"
                         "v_mfma_f32_32x32x16_f16 v[48:63], v[40:43], v[44:47], 0\n\t"
                         "s_nop 15\n\t" ADD16
                         : "+v"(chk) : "v"(one), "v"(bval), "v"(fval), "i"(PROD), "i"(K), "v"(fone), "v"(fzero)
                         : CLOB);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = chk;
}

template <int PROD, int K, int HAM>
void run(float* d, int iters) {
    const int waves = 16;
    const size_t n = (size_t)256 * waves * 64;
    hipLaunchKernelGGL((k<PROD, K, HAM>), dim3(256), dim3(waves * 64), 0, 0, d, iters);
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    const float expect = 16.0f * 16.0f * (iters / 2) * 3.0f;
    size_t bad = 0, q[4] = {0, 0, 0, 0}, tested = 0;
    for (size_t i = 0; i < n; ++i) {
        const int wave = (i % (waves * 64)) / 64;
        if (HAM != 0 && wave >= 4) continue;
        ++tested;
        if (h[i] != expect) { ++bad; ++q[(i % 64) / 16]; }
    }
    static const char* pn[] = {"v_mov_b32", "v_cvt_pk_f16_f32", "v_fma_mixlo/hi_f16", "v_pk_mul_f16"};
    static const char* hn[] = {"none (16 test waves)", "v_exp_f32", "v_pk_fma_f32", "v_fma_f32", "mfma", "v_exp_f32 + mfma"};
    printf("producer %-18s %d wait states, hammer %-20s: %7zu wrong lanes of %zu (lane quarters %zu %zu %zu %zu)\n", pn[PROD], K, hn[HAM], bad, tested, q[0],
           q[1], q[2], q[3]);
}

template <int PROD, int HAM>
void sweepK(float* d, int it) {
    run<PROD, 0, HAM>(d, it); run<PROD, 1, HAM>(d, it); run<PROD, 2, HAM>(d, it); run<PROD, 3, HAM>(d, it); run<PROD, 4, HAM>(d, it); run<PROD, 6, HAM>(d, it);
}
template <int HAM>
void suite(float* d, int it) {
    sweepK<0, HAM>(d, it); sweepK<1, HAM>(d, it); sweepK<2, HAM>(d, it); sweepK<3, HAM>(d, it);
}

int main() {
    float* d;
    (void)hipMalloc(&d, (size_t)256 * 1024 * 4);
    const int it = 2000;
    suite<0>(d, it); suite<1>(d, it); suite<2>(d, it); suite<3>(d, it); suite<4>(d, it); suite<5>(d, it);
    return 0;
}
