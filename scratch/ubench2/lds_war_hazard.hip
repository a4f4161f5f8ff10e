// Third reproducer for DESIGN.md §9: an LDS load (asynchronous VGPR write) into a SOURCE register of a v_mfma_f32_32x32x16_f16 that
// the same wave issued just before it.  hipcc (ROCm 7.2) treats this write-after-read as safe (in-order issue).  Question: is it
// still safe when the matrix pipe is busy with other waves' MFMAs, i.e. can the MFMA still be waiting for its operands when the LDS
// data lands?   Test waves: B := val(i); MFMA(acc = A*B); [K x s_nop]; ds_read_b128 B <- garbage; wait; chk += acc[0..15].
// Hammer waves (waves >= NTEST of a 16-wave workgroup) issue dependent or independent MFMAs back to back.
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

// WHICH 0: LDS load into B, 1: into A, 2: global load into B;  K wait states between the MFMA and the load;  NTEST test waves
template <int WHICH, int K, int NTEST, int CHAIN>
__global__ __launch_bounds__(1024) void k(float* out, const unsigned* gsrc, int iters) {
    __shared__ unsigned sh[64 * 4];
    for (int i = threadIdx.x; i < 64 * 4; i += blockDim.x) sh[i] = 0x70007000u;   // 8192.0 halves
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float chk = 0.0f;
    unsigned one = 0x3C003C00u, lds_addr = (threadIdx.x & 63) * 16;
    const unsigned* gp = gsrc + (threadIdx.x & 63) * 4;
    asm volatile("" : "+v"(one));
    if (wave >= NTEST) {
        f32x16 acc = {0}, acc2 = {0};
        u32x4 a = {one, one, one, one}, b = a;
        for (int i = 0; i < iters * 4; ++i) {
            if (CHAIN)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
                             "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1\n\t"
                             "v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\tv_mfma_f32_32x32x16_f16 %1, %2, %3, %1" : "+v"(acc), "+v"(acc2) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15" : "+v"(acc), "+v"(acc2));
        chk = (acc[0] + acc2[0]) * 0.0f;
    } else {
        for (int i = 0; i < iters; ++i) {
            const unsigned bval = (i & 1) ? 0x40004000u : 0x3C003C00u;   // 2.0 : 1.0
            asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %1\n\t"
                         "v_mov_b32 v44, %2\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %2\n\tv_mov_b32 v47, %2\n\t"
                         "s_nop 7\n\t"
                         "v_mfma_f32_32x32x16_f16 v[48:63], v[40:43], v[44:47], 0\n\t"
                         ".rept %c6\n\ts_nop 0\n\t.endr\n\t"
                         ".if %c3 == 0\n\tds_read_b128 v[44:47], %4\n\t.endif\n\t"
                         ".if %c3 == 1\n\tds_read_b128 v[40:43], %4\n\t.endif\n\t"
                         ".if %c3 == 2\n\tglobal_load_dwordx4 v[44:47], %5, off\n\t.endif\n\t"
                         "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\t" ADD16
                         : "+v"(chk) : "v"(one), "v"(bval), "i"(WHICH), "v"(lds_addr), "v"(gp), "i"(K)
                         : CLOB);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = chk;
}

template <int WHICH, int K, int NTEST, int CHAIN>
void run(float* d, const unsigned* g, int iters) {
    const int waves = 16;
    const size_t n = (size_t)256 * waves * 64;
    hipLaunchKernelGGL((k<WHICH, K, NTEST, CHAIN>), dim3(256), dim3(waves * 64), 0, 0, d, g, iters);
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    const float expect = 16.0f * 16.0f * (iters / 2) * 3.0f;
    size_t bad = 0, q[4] = {0, 0, 0, 0}, tested = 0;
    for (size_t i = 0; i < n; ++i) {
        const int wave = (i % (waves * 64)) / 64;
        if (wave >= NTEST) continue;
        ++tested;
        if (h[i] != expect) { ++bad; ++q[(i % 64) / 16]; }
    }
    static const char* names[] = {"ds_read_b128 -> B", "ds_read_b128 -> A", "global_load -> B"};
    printf("%-18s %2d wait states, %2d test waves + %2d hammer waves (%s): %7zu wrong lanes of %zu (lane quarters %zu %zu %zu %zu)\n", names[WHICH], K, NTEST,
           16 - NTEST, CHAIN ? "dependent chain" : "two accumulators", bad, tested, q[0], q[1], q[2], q[3]);
}

template <int NTEST, int CHAIN>
void suite(float* d, const unsigned* g, int it) {
    run<0, 0, NTEST, CHAIN>(d, g, it); run<0, 1, NTEST, CHAIN>(d, g, it); run<0, 2, NTEST, CHAIN>(d, g, it); run<0, 4, NTEST, CHAIN>(d, g, it);
    run<0, 8, NTEST, CHAIN>(d, g, it); run<0, 16, NTEST, CHAIN>(d, g, it);
    run<1, 0, NTEST, CHAIN>(d, g, it); run<1, 2, NTEST, CHAIN>(d, g, it); run<1, 8, NTEST, CHAIN>(d, g, it);
    run<2, 0, NTEST, CHAIN>(d, g, it);
}

int main() {
    float* d;
    unsigned* g;
    (void)hipMalloc(&d, (size_t)256 * 1024 * 4);
    (void)hipMalloc(&g, 4096);
    std::vector<unsigned> hg(1024, 0x70007000u);
    (void)hipMemcpy(g, hg.data(), 4096, hipMemcpyHostToDevice);
    const int it = 2000;
    suite<16, 1>(d, g, it);   // no hammer: every wave is a test wave
    suite<4, 1>(d, g, it);    // 1 test + 3 hammer waves per SIMD
    suite<4, 0>(d, g, it);
    suite<8, 1>(d, g, it);    // 2 test + 2 hammer waves per SIMD
    suite<12, 1>(d, g, it);
    return 0;
}
