// Fifth reproducer for DESIGN.md §9: result of v_mfma_f32_32x32x2_f32 (16 passes; hipcc pads 18 wait states before a VALU read)
// read by plain and by packed-FP32 VALU instructions, with other waves of the same SIMD competing for the pipes.
// Test waves (waves 0..NTEST-1): B := val(i) (1.0 / 2.0 alternating); NM dependent f32 MFMAs acc = A*B + acc starting from C = 0;
// W wait states; chk += sum of the 16 accumulator registers, read either by 16 v_add_f32 or by 8 v_pk_add_f32 + 1 v_add.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f32x2 = __attribute__((ext_vector_type(2))) float;

#define CLOB "v40", "v44", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "memory"
#define RD_ADD "v_add_f32 %0, %0, v48\n\tv_add_f32 %0, %0, v49\n\tv_add_f32 %0, %0, v50\n\tv_add_f32 %0, %0, v51\n\t" \
               "v_add_f32 %0, %0, v52\n\tv_add_f32 %0, %0, v53\n\tv_add_f32 %0, %0, v54\n\tv_add_f32 %0, %0, v55\n\t" \
               "v_add_f32 %0, %0, v56\n\tv_add_f32 %0, %0, v57\n\tv_add_f32 %0, %0, v58\n\tv_add_f32 %0, %0, v59\n\t" \
               "v_add_f32 %0, %0, v60\n\tv_add_f32 %0, %0, v61\n\tv_add_f32 %0, %0, v62\n\tv_add_f32 %0, %0, v63\n\t"
// packed: v[64:65] = sum of the 8 register pairs, then chk += v64 + v65
#define RD_PK "v_pk_add_f32 v[64:65], v[48:49], v[50:51]\n\tv_pk_add_f32 v[64:65], v[64:65], v[52:53]\n\tv_pk_add_f32 v[64:65], v[64:65], v[54:55]\n\t" \
              "v_pk_add_f32 v[64:65], v[64:65], v[56:57]\n\tv_pk_add_f32 v[64:65], v[64:65], v[58:59]\n\tv_pk_add_f32 v[64:65], v[64:65], v[60:61]\n\t" \
              "v_pk_add_f32 v[64:65], v[64:65], v[62:63]\n\tv_add_f32 %0, %0, v64\n\tv_add_f32 %0, %0, v65\n\t"
// packed, last pair first
#define RD_PKR "v_pk_add_f32 v[64:65], v[62:63], v[60:61]\n\tv_pk_add_f32 v[64:65], v[64:65], v[58:59]\n\tv_pk_add_f32 v[64:65], v[64:65], v[56:57]\n\t" \
               "v_pk_add_f32 v[64:65], v[64:65], v[54:55]\n\tv_pk_add_f32 v[64:65], v[64:65], v[52:53]\n\tv_pk_add_f32 v[64:65], v[64:65], v[50:51]\n\t" \
               "v_pk_add_f32 v[64:65], v[64:65], v[48:49]\n\tv_add_f32 %0, %0, v64\n\tv_add_f32 %0, %0, v65\n\t"

// RD 0: v_add order 0..15, 1: pk order 0..15, 2: pk order 15..0;  W wait states;  HAM 0 none, 1 f16 mfma, 2 f32 mfma, 3 v_pk_fma, 4 v_exp, 5 f16 mfma + v_exp
template <int RD, int W, int HAM, int NTEST>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float chk = 0.0f;
    unsigned one = 0x3C003C00u;
    float fone = 1.0f;
    asm volatile("" : "+v"(one), "+v"(fone));
    if (HAM != 0 && wave >= NTEST) {
        f32x16 acc = {0};
        u32x4 a = {one, one, one, one}, b = a;
        float f0 = 0.1f * threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
        for (int i = 0; i < iters * 4; ++i) {
            if (HAM == 1 || HAM == 5)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
            if (HAM == 2)
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fone), "v"(fone));
            if (HAM == 3)
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %0\n\tv_pk_fma_f32 %1, %1, %0, %1"
                             : "+v"(*(double*)&f0), "+v"(*(double*)&f2));
            if (HAM == 4 || HAM == 5)
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
        }
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
        chk = (acc[0] + f0 + f1 + f2 + f3) * 0.0f;
    } else {
        for (int i = 0; i < iters; ++i) {
            const float fval = (i & 1) ? 2.0f : 1.0f;
            asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v44, %2\n\ts_nop 7\n\t"
                         "v_mfma_f32_32x32x2_f32 v[48:63], v40, v44, 0\n\t"
                         "v_mfma_f32_32x32x2_f32 v[48:63], v40, v44, v[48:63]\n\t"
                         "v_mfma_f32_32x32x2_f32 v[48:63], v40, v44, v[48:63]\n\t"
                         ".rept %c3\n\ts_nop 0\n\t.endr\n\t"
                         ".if %c4 == 0\n\t" RD_ADD ".endif\n\t"
                         ".if %c4 == 1\n\t" RD_PK ".endif\n\t"
                         ".if %c4 == 2\n\t" RD_PKR ".endif\n\t"
                         : "+v"(chk) : "v"(fone), "v"(fval), "i"(W), "i"(RD) : CLOB);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = chk;
}

template <int RD, int W, int HAM, int NTEST>
void run(float* d, int iters) {
    const int waves = 16;
    const size_t n = (size_t)256 * waves * 64;
    hipLaunchKernelGGL((k<RD, W, HAM, NTEST>), dim3(256), dim3(waves * 64), 0, 0, d, iters);
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    // per iteration every accumulator register = 3 MFMAs x K=2 x (1 * val): 6 or 12; 16 registers
    const float expect = 16.0f * (6.0f + 12.0f) * (iters / 2);
    size_t bad = 0, q[4] = {0, 0, 0, 0}, tested = 0;
    for (size_t i = 0; i < n; ++i) {
        const int wave = (i % (waves * 64)) / 64;
        if (HAM != 0 && wave >= NTEST) continue;
        ++tested;
        if (h[i] != expect) { ++bad; ++q[(i % 64) / 16]; }
    }
    static const char* rn[] = {"16 v_add_f32 (0..15)", "v_pk_add_f32 (0..15)", "v_pk_add_f32 (15..0)"};
    static const char* hn[] = {"none", "f16 mfma", "f32 mfma", "v_pk_fma_f32", "v_exp_f32", "f16 mfma + v_exp"};
    printf("read %-21s after %2d wait states, %2d test waves, hammer %-17s: %7zu wrong lanes of %zu (lane quarters %zu %zu %zu %zu)\n", rn[RD], W, NTEST, hn[HAM],
           bad, tested, q[0], q[1], q[2], q[3]);
}

template <int HAM, int NTEST>
void suite(float* d, int it) {
    run<0, 18, HAM, NTEST>(d, it); run<1, 18, HAM, NTEST>(d, it); run<2, 18, HAM, NTEST>(d, it);
    run<0, 17, HAM, NTEST>(d, it); run<1, 17, HAM, NTEST>(d, it); run<2, 17, HAM, NTEST>(d, it);
    run<0, 16, HAM, NTEST>(d, it); run<1, 16, HAM, NTEST>(d, it); run<2, 16, HAM, NTEST>(d, it);
    run<2, 14, HAM, NTEST>(d, it);
}

int main() {
    float* d;
    (void)hipMalloc(&d, (size_t)256 * 1024 * 4);
    const int it = 2000;
    suite<0, 16>(d, it);
    suite<1, 4>(d, it); suite<2, 4>(d, it); suite<3, 4>(d, it); suite<4, 4>(d, it); suite<5, 4>(d, it);
    suite<1, 8>(d, it); suite<2, 12>(d, it);
    return 0;
}
