// Second reproducer for DESIGN.md §9: which instructions really provide "wait states" between a v_mfma_f32_32x32x16_f16 and the first
// VALU read of its result?  hipcc (ROCm 7.2) counts every instruction it schedules into that window -- LDS, SALU, VMEM, VALU alike -- as one
// wait state (12 are required after this 8-pass MFMA) and pads the rest with s_nop.
// Test waves run: MFMA(acc = A*B, C = 0); <filler of F instructions>; s_nop (12 - F - 1 - SHORT); 16 x v_add chk += acc[k].
// B alternates between 1.0 and 2.0 every iteration, so a stale read (previous iteration's accumulator) changes the checksum.
// Hammer waves (HAMMER = 1: waves 4..15 of the 16-wave workgroup) issue back-to-back MFMAs on the same SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define ADD16 "v_add_f32 %0, %0, v48\n\tv_add_f32 %0, %0, v49\n\tv_add_f32 %0, %0, v50\n\tv_add_f32 %0, %0, v51\n\t" \
              "v_add_f32 %0, %0, v52\n\tv_add_f32 %0, %0, v53\n\tv_add_f32 %0, %0, v54\n\tv_add_f32 %0, %0, v55\n\t" \
              "v_add_f32 %0, %0, v56\n\tv_add_f32 %0, %0, v57\n\tv_add_f32 %0, %0, v58\n\tv_add_f32 %0, %0, v59\n\t" \
              "v_add_f32 %0, %0, v60\n\tv_add_f32 %0, %0, v61\n\tv_add_f32 %0, %0, v62\n\tv_add_f32 %0, %0, v63\n\t"
#define ADD16R "v_add_f32 %0, %0, v63\n\tv_add_f32 %0, %0, v62\n\tv_add_f32 %0, %0, v61\n\tv_add_f32 %0, %0, v60\n\t" \
               "v_add_f32 %0, %0, v59\n\tv_add_f32 %0, %0, v58\n\tv_add_f32 %0, %0, v57\n\tv_add_f32 %0, %0, v56\n\t" \
               "v_add_f32 %0, %0, v55\n\tv_add_f32 %0, %0, v54\n\tv_add_f32 %0, %0, v53\n\tv_add_f32 %0, %0, v52\n\t" \
               "v_add_f32 %0, %0, v51\n\tv_add_f32 %0, %0, v50\n\tv_add_f32 %0, %0, v49\n\tv_add_f32 %0, %0, v48\n\t"
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", \
             "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", \
             "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "memory"
// filler kinds: 0 none (all s_nop), 1 = 6 ds_read_b128, 2 = 6 SALU, 3 = 6 VALU, 4 = 6 s_waitcnt (already satisfied), 5 = 6 global_load_dwordx4
#define FILL_LDS "ds_read_b128 v[64:67], %4\n\tds_read_b128 v[68:71], %4 offset:16\n\tds_read_b128 v[72:75], %4 offset:32\n\t" \
                 "ds_read_b128 v[76:79], %4 offset:48\n\tds_read_b128 v[80:83], %4 offset:64\n\tds_read_b128 v[84:87], %4 offset:80\n\t"
#define FILL_SALU "s_mul_i32 s20, s20, 3\n\ts_add_u32 s21, s21, s20\n\ts_lshl_b32 s22, s21, 1\n\ts_and_b32 s23, s22, s20\n\ts_mul_i32 s20, s23, 5\n\ts_add_u32 s21, s20, 1\n\t"
#define FILL_VALU "v_fma_f32 v64, v64, v65, v66\n\tv_fma_f32 v67, v67, v65, v66\n\tv_fma_f32 v68, v68, v65, v66\n\tv_fma_f32 v69, v69, v65, v66\n\t" \
                  "v_fma_f32 v70, v70, v65, v66\n\tv_fma_f32 v71, v71, v65, v66\n\t"
#define FILL_WAIT "s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)\n\t"
#define FILL_VMEM "global_load_dwordx4 v[64:67], %5, off\n\tglobal_load_dwordx4 v[68:71], %5, off offset:16\n\tglobal_load_dwordx4 v[72:75], %5, off offset:32\n\t" \
                  "global_load_dwordx4 v[76:79], %5, off offset:48\n\tglobal_load_dwordx4 v[80:83], %5, off offset:64\n\tglobal_load_dwordx4 v[84:87], %5, off offset:80\n\t"

template <int FILL, int SHORT, int REV, int HAMMER>
__global__ __launch_bounds__(1024) void k(float* out, const float* gsrc, int iters) {
    __shared__ float sh[1024 * 4 + 64];
    for (int i = threadIdx.x; i < 1024 * 4 + 64; i += blockDim.x) sh[i] = 1.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float chk = 0.0f;
    unsigned one = 0x3C003C00u, lds_addr = (threadIdx.x & 63) * 16;
    const float* gp = gsrc + (threadIdx.x & 63) * 4;
    asm volatile("" : "+v"(one));
    if (HAMMER && wave >= 4) {
        f32x16 acc = {0};
        u32x4 a = {one, one, one, one}, b = a;
        for (int i = 0; i < iters * 3; ++i)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("s_nop 15" : "+v"(acc));
        chk = acc[0] * 0.0f;
    } else {
        for (int i = 0; i < iters; ++i) {
            const unsigned bval = (i & 1) ? 0x40004000u : 0x3C003C00u;   // 2.0 : 1.0
            constexpr int NOPS = 12 - (FILL ? 6 : 0) - SHORT;   // wait states still to pad
            asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %1\n\t"
                         "v_mov_b32 v44, %2\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %2\n\tv_mov_b32 v47, %2\n\t"
                         "s_nop 7\n\t"
                         "v_mfma_f32_32x32x16_f16 v[48:63], v[40:43], v[44:47], 0\n\t"
                         ".if %c3 == 1\n\t" FILL_LDS ".endif\n\t"
                         ".if %c3 == 2\n\t" FILL_SALU ".endif\n\t"
                         ".if %c3 == 3\n\t" FILL_VALU ".endif\n\t"
                         ".if %c3 == 4\n\t" FILL_WAIT ".endif\n\t"
                         ".if %c3 == 5\n\t" FILL_VMEM ".endif\n\t"
                         ".rept %c6\n\ts_nop 0\n\t.endr\n\t"
                         ".if %c7 == 0\n\t" ADD16 ".else\n\t" ADD16R ".endif\n\t"
                         "s_waitcnt vmcnt(0) lgkmcnt(0)"
                         : "+v"(chk) : "v"(one), "v"(bval), "i"(FILL), "v"(lds_addr), "v"(gp), "i"(NOPS), "i"(REV)
                         : CLOB, "s20", "s21", "s22", "s23");
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = chk;
}

template <int FILL, int SHORT, int REV, int HAMMER>
void run(float* d, const float* g, int waves, int iters) {
    const size_t n = (size_t)256 * waves * 64;
    hipLaunchKernelGGL((k<FILL, SHORT, REV, HAMMER>), dim3(256), dim3(waves * 64), 0, 0, d, g, iters);
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    const float expect = 16.0f * 16.0f * (iters / 2) * 3.0f;   // per pair of iterations: 16 registers x (16 + 32)
    size_t bad = 0, q[4] = {0, 0, 0, 0}, tested = 0;
    for (size_t i = 0; i < n; ++i) {
        const int wave = (i % (waves * 64)) / 64;
        if (HAMMER && wave >= 4) continue;
        ++tested;
        if (h[i] != expect) { ++bad; ++q[(i % 64) / 16]; }
    }
    static const char* names[] = {"s_nop only", "6 ds_read_b128", "6 SALU", "6 VALU", "6 s_waitcnt", "6 global_load"};
    printf("filler %-15s wait states %2d (%+d) read order %s hammer %d waves/WG %2d: %7zu wrong lanes of %zu (lane quarters %zu %zu %zu %zu)\n", names[FILL],
           12 - SHORT, -SHORT, REV ? "15..0" : "0..15", HAMMER, waves, bad, tested, q[0], q[1], q[2], q[3]);
}

template <int HAMMER>
void suite(float* d, const float* g, int waves, int it) {
    run<0, 0, 0, HAMMER>(d, g, waves, it); run<0, 0, 1, HAMMER>(d, g, waves, it);
    run<0, 2, 0, HAMMER>(d, g, waves, it); run<0, 2, 1, HAMMER>(d, g, waves, it);
    run<0, 4, 0, HAMMER>(d, g, waves, it); run<0, 4, 1, HAMMER>(d, g, waves, it);
    run<1, 0, 0, HAMMER>(d, g, waves, it); run<1, 0, 1, HAMMER>(d, g, waves, it);
    run<2, 0, 0, HAMMER>(d, g, waves, it); run<2, 0, 1, HAMMER>(d, g, waves, it);
    run<3, 0, 0, HAMMER>(d, g, waves, it); run<3, 0, 1, HAMMER>(d, g, waves, it);
    run<4, 0, 0, HAMMER>(d, g, waves, it); run<4, 0, 1, HAMMER>(d, g, waves, it);
    run<5, 0, 0, HAMMER>(d, g, waves, it); run<5, 0, 1, HAMMER>(d, g, waves, it);
}

int main() {
    float *d, *g;
    (void)hipMalloc(&d, (size_t)256 * 1024 * 4);
    (void)hipMalloc(&g, 4096);
    (void)hipMemset(g, 0, 4096);
    const int it = 2000;
    suite<0>(d, g, 4, it);
    suite<0>(d, g, 16, it);
    suite<1>(d, g, 16, it);
    return 0;
}
