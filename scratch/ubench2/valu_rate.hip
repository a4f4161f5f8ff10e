// Issue rate of single VALU instruction kinds on gfx950 at 1 and 4 waves per SIMD (s_memtime around a 1024-iteration loop of 16
// independent instructions).  Output: SIMD cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define OPS : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(c0), "v"(c1)
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n\t"
#define I_ADD(n) "v_add_f32 %" #n ", 1.0, %" #n "\n\t"
#define I_MUL(n) "v_mul_f32 %" #n ", %16, %" #n "\n\t"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n\t"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define I_LOG(n) "v_log_f32 %" #n ", %" #n "\n\t"
#define I_CVT(n) "v_cvt_pk_f16_f32 %" #n ", %" #n ", %16\n\t"
#define I_MIXLO(n) "v_fma_mixlo_f16 %" #n ", %" #n ", -1.0, %16 op_sel_hi:[1,0,0]\n\t"
#define I_MIXHI(n) "v_fma_mixhi_f16 %" #n ", %" #n ", -1.0, %16 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#define I_PKFMA(n) "v_pk_fma_f16 %" #n ", %" #n ", %16, %17\n\t"
#define I_MOV(n) "v_mov_b32 %" #n ", %16\n\t"
#define I_SWAP(n) "v_permlane32_swap_b32 %" #n ", %" #n "\n\t"
#define I_MAX(n) "v_max_f32 %" #n ", %" #n ", %16\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %16\n\t"
#define I_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n\t"
#define I_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n\t"
#define I_LDEXP(n) "v_ldexp_f32 %" #n ", %" #n ", %16\n\t"
#define I_FMAK(n) "v_fmac_f32 %" #n ", %16, %17\n\t"
#define I_FMALIT(n) "v_mul_f32 %" #n ", 0x3fb8aa3b, %" #n "\n\t"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n\t"

template <int P>
__global__ void __launch_bounds__(1024) k(unsigned long long* cyc, int iters, float c0, float c1) {
    float r[16];
    for (int i = 0; i < 16; ++i) r[i] = c0 * (threadIdx.x + i);
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (P == 0) asm volatile(R16(I_FMA) OPS);
        if constexpr (P == 1) asm volatile(R16(I_ADD) OPS);
        if constexpr (P == 2) asm volatile(R16(I_MUL) OPS);
        if constexpr (P == 3) asm volatile(R16(I_EXP) OPS);
        if constexpr (P == 4) asm volatile(R16(I_RCP) OPS);
        if constexpr (P == 5) asm volatile(R16(I_LOG) OPS);
        if constexpr (P == 6) asm volatile(R16(I_CVT) OPS);
        if constexpr (P == 7) asm volatile(R16(I_MIXLO) OPS);
        if constexpr (P == 8) asm volatile(R16(I_MIXHI) OPS);
        if constexpr (P == 9) asm volatile(R16(I_PKFMA) OPS);
        if constexpr (P == 10) asm volatile(R16(I_MOV) OPS);
        if constexpr (P == 11) asm volatile(R16(I_SWAP) OPS);
        if constexpr (P == 12) asm volatile(R16(I_MAX) OPS);
        if constexpr (P == 13) asm volatile(R16(I_LSHLADD) OPS);
        if constexpr (P == 14) asm volatile(R16(I_CVTI) OPS);
        if constexpr (P == 15) asm volatile(R16(I_FLOOR) OPS);
        if constexpr (P == 16) asm volatile(R16(I_LDEXP) OPS);
        if constexpr (P == 17) asm volatile(R16(I_FMAK) OPS);
        if constexpr (P == 18) asm volatile(R16(I_FMALIT) OPS);
        if constexpr (P == 19) asm volatile(R16(I_CNDMASK) OPS : "vcc");
    }
    asm volatile("s_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    if (s == 123.456f) cyc[0] = 1;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int P>
void run(const char* name) {
    unsigned long long* d;
    (void)hipMalloc(&d, 256 * 16 * 8);
    const int iters = 1024;
    printf("%-22s", name);
    for (int waves : {4, 8, 16}) {
        (void)hipMemset(d, 0, 256 * 16 * 8);
        k<P><<<256, waves * 64>>>(d, iters, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16);
        (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> v;
        for (int g = 0; g < 256; ++g) for (int w = 0; w < waves; ++w) v.push_back((double)h[g * 16 + w] / iters);
        std::sort(v.begin(), v.end());
        const double per_simd = v[v.size() / 2] / (16.0 * (waves / 4));
        printf("  %d/SIMD: %6.2f", waves / 4, per_simd);
    }
    printf("   (SIMD cycles per wave-instruction)\n");
    (void)hipFree(d);
}

int main() {
    run<0>("v_fma_f32"); run<17>("v_fmac_f32"); run<18>("v_mul_f32 literal"); run<1>("v_add_f32"); run<2>("v_mul_f32"); run<12>("v_max_f32"); run<10>("v_mov_b32");
    run<19>("v_cndmask_b32");
    run<3>("v_exp_f32"); run<4>("v_rcp_f32"); run<5>("v_log_f32");
    run<6>("v_cvt_pk_f16_f32"); run<7>("v_fma_mixlo_f16"); run<8>("v_fma_mixhi_f16"); run<9>("v_pk_fma_f16");
    run<11>("v_permlane32_swap"); run<13>("v_lshl_add_u32"); run<14>("v_cvt_i32_f32"); run<15>("v_floor_f32"); run<16>("v_ldexp_f32");
    return 0;
}
