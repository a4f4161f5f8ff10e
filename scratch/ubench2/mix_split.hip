// Checks the inline-asm fp16 split used by wf_mfma_impl.h (r_split8): hi = rn16(r), lo = rn16(r - hi) via v_cvt_pk_f16_f32 +
// v_fma_mixlo/hi_f16 with an f16 source operand, against the plain-C split.   hipcc --offload-arch=gfx950 -O3 mix_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ void split8(const float (&r)[8], u32x4& hi, u32x4& lo) {
    asm volatile(
        "s_nop 0\n\t"
        "v_cvt_pk_f16_f32 %0, %8, %9\n\tv_cvt_pk_f16_f32 %1, %10, %11\n\tv_cvt_pk_f16_f32 %2, %12, %13\n\tv_cvt_pk_f16_f32 %3, %14, %15\n\t"
        "v_fma_mixhi_f16 %4, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %5, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %6, %2, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %7, %3, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %4, %0, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %5, %1, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %6, %2, -1.0, %12 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %7, %3, -1.0, %14 op_sel_hi:[1,0,0]\n\t"
        "s_nop 1"
        : "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3])
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
}
__global__ void k(const float* x, u32x4* hi, u32x4* lo, float* rr) {
    float r[8];
    for (int j = 0; j < 8; ++j) { r[j] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x[threadIdx.x * 8 + j]) + 1.0f); rr[threadIdx.x * 8 + j] = r[j]; }
    u32x4 H, L;
    split8(r, H, L);
    hi[threadIdx.x] = H; lo[threadIdx.x] = L;
}
int main() {
    const int n = 1024 * 8;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = -30.0f + 60.0f * (float)rand() / RAND_MAX;
    float *dx, *dr; u32x4 *dh, *dl;
    hipMalloc(&dx, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&dl, n * 2);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, dx, dh, dl, dr);
    std::vector<_Float16> h(n), l(n); std::vector<float> r(n);
    hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(l.data(), dl, n * 2, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; double worst = 0;
    for (int i = 0; i < n; ++i) {
        const _Float16 eh = (_Float16)r[i]; const _Float16 el = (_Float16)(r[i] - (float)eh);
        if (h[i] != eh || l[i] != el) { if (bad < 5) printf("mismatch %d: r %.9g hi %.9g (%.9g) lo %.9g (%.9g)\n", i, r[i], (float)h[i], (float)eh, (float)l[i], (float)el); ++bad; }
        const double e = std::fabs((double)(float)h[i] + (double)(float)l[i] - (double)r[i]) / std::fmax((double)r[i], 1e-30);
        if (r[i] > 1e-4 && e > worst) worst = e;
    }
    printf("mix split: %d mismatches of %d against the plain split; worst relative |hi + lo - r| / r (r > 1e-4): %.3g (2^-23 = %.3g)\n", bad, n, worst, std::ldexp(1.0, -23));
    return bad != 0;
}
