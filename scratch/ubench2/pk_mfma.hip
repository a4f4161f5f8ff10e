// Sixth reproducer for DESIGN.md §9: packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32, what hipcc's SLP vectorizer makes
// of adjacent scalar f32 operations) issued by one wave while v_mfma_f32_32x32x16_f16 chains are in flight on the same SIMD.
// Every wave alternates two phases and checks all its packed results exactly (small-integer arithmetic):
//   phase 1: a dependent chain of 12 MFMAs with PK1 packed instructions placed in every MFMA gap (independent registers)
//   phase 2: a dependent chain of 16 v_pk_fma_f32 (the shape of the spline dot products), no MFMA
// MODE 0: phase 1 only, 1: phase 2 only, 2: both (as in the kernel), 3: both, but scalar v_fma_f32 instead of packed instructions in phase 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#define MF "v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\t"
// x = x * 1 + 1 on both halves (p0..p3 = %3..%6, ones = %7)
#define PK(n) "v_pk_fma_f32 %" #n ", %" #n ", %7, %7\n\t"
#define PKA(n) "v_pk_add_f32 %" #n ", %" #n ", %7\n\t"
#define SC(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n\t"

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    f32x16 acc = {0};
    unsigned one = 0x3C003C00u;
    u32x4 a = {one, one, one, one}, b = a;
    f32x2 p0 = {0, 0}, p1 = {0, 0}, p2 = {0, 0}, p3 = {0, 0}, ones = {1.0f, 1.0f}, q = {0, 0};
    f32x2 dummy, t, twos = {2.0f, 2.0f}, dev = {0, 0};
    float s1 = 1.0f, sc0 = 0, sc1 = 0, sc2 = 0, sc3 = 0;
    asm volatile("" : "+v"(ones), "+v"(s1), "+v"(a), "+v"(b));
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2)
            asm volatile(MF PK(3) PKA(4) MF PK(5) PKA(6) MF PK(3) PKA(4) MF PK(5) PKA(6) MF PK(3) PKA(4) MF PK(5) PKA(6)
                         MF PK(3) PKA(4) MF PK(5) PKA(6) MF PK(3) PKA(4) MF PK(5) PKA(6) MF PK(3) PKA(4) MF PK(5) PKA(6)
                         : "+v"(acc), "+v"(a), "+v"(b), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ones), "v"(s1));
        if (MODE == 3)
            asm volatile(MF SC(3) SC(4) MF SC(5) SC(6) MF SC(3) SC(4) MF SC(5) SC(6) MF SC(3) SC(4) MF SC(5) SC(6)
                         MF SC(3) SC(4) MF SC(5) SC(6) MF SC(3) SC(4) MF SC(5) SC(6) MF SC(3) SC(4) MF SC(5) SC(6)
                         : "+v"(acc), "+v"(a), "+v"(b), "+v"(sc0), "+v"(sc1), "+v"(sc2), "+v"(sc3) : "v"(ones), "v"(s1));
        if (MODE == 1 || MODE == 2 || MODE == 3)
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\t"
                         "v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\t"
                         "v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\t"
                         "v_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\tv_pk_fma_f32 %0, %0, %1, %1\n\t"
                         "v_pk_add_f32 %2, %0, %0 op_sel:[0,1] op_sel_hi:[1,0]\n\t"   // t = (q.lo + q.hi, q.hi + q.lo)
                         "v_pk_fma_f32 %2, %0, %3, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // t -= 2 q  (exactly 0 when q.lo == q.hi)
                         "v_pk_add_f32 %4, %4, %2"
                         : "+v"(q), "=&v"(dummy), "=&v"(t), "+v"(twos), "+v"(dev) : "v"(ones));
        if ((i & 15) == 15) {   // keep the numbers exactly representable: fold and reset
            // q after n rounds ... compare against the host's model by exporting and resetting every 16 iterations
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 12;
    o[0] = p0[0]; o[1] = p0[1]; o[2] = p1[0]; o[3] = p1[1]; o[4] = p2[0]; o[5] = p2[1]; o[6] = p3[0]; o[7] = p3[1];
    o[8] = q[0]; o[9] = q[1]; o[10] = dev[0] + dev[1]; o[11] = sc0 + sc1 + sc2 + sc3 + acc[0] * 0.0f;
}

template <int MODE>
void run(float* d, int waves, int iters) {
    const size_t n = (size_t)256 * waves * 64;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(waves * 64), 0, 0, d, iters);
    std::vector<float> h(n * 12);
    (void)hipMemcpy(h.data(), d, n * 12 * 4, hipMemcpyDeviceToHost);
    // host model
    float p = 0, qlo = 0, qhi = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE == 2) p += 6.0f;   // 6 increments of 1 per phase-1 pass for each of p0..p3 (3 fma-type + ... see below)
        if (MODE != 0) {
            for (int r = 0; r < 16; ++r) { qlo = qlo * 1.0f + 1.0f; qhi = qhi * 1.0f + 1.0f; }
        }
    }
    size_t bad_p = 0, bad_q = 0, quarters[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        const float* o = h.data() + i * 12;
        bool bp = false;
        if (MODE == 0 || MODE == 2) for (int e = 0; e < 8; ++e) bp |= (o[e] != p);
        const bool bq = (MODE != 0) && (o[8] != qlo || o[9] != qhi || o[10] != 0.0f);
        bad_p += bp; bad_q += bq;
        if (bp || bq) ++quarters[(i % 64) / 16];
    }
    static const char* mn[] = {"phase 1 only (mfma + packed in the gaps)", "phase 2 only (packed chain)", "both phases", "both, scalar fma in the gaps"};
    printf("%-42s %2d waves/WG: wrong lanes: gap results %zu, chain results %zu of %zu (lane quarters %zu %zu %zu %zu)  [model p=%g q=%g]\n", mn[MODE], waves, bad_p,
           bad_q, n, quarters[0], quarters[1], quarters[2], quarters[3], p, qlo);
}

int main() {
    float* d;
    (void)hipMalloc(&d, (size_t)256 * 1024 * 12 * 4);
    const int it = 4000;   // q grows by 16 per iteration: exact
    for (int waves : {4, 8, 12, 16}) {
        run<0>(d, waves, it); run<1>(d, waves, it); run<2>(d, waves, it); run<3>(d, waves, it);
    }
    for (int rep = 0; rep < 3; ++rep) { run<2>(d, 16, it); run<2>(d, 12, it); }
    return 0;
}
