// Does VALU work overlap with MFMA work on one SIMD?  (a) f32-input MFMA, (b) f16 MFMA.
// Workgroup = 8 waves (2 per SIMD): waves 0-3 run role A, waves 4-7 run role B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int ROLE_A, int ROLE_B>  // 0 = idle, 1 = VALU fma chain, 2 = mfma f32 32x32x2, 3 = mfma f16 32x32x16, 4 = VALU transcendental
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? ROLE_A : ROLE_B;
    float r = 0.f;
    if (role == 1) {
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_fmaf(a0, 0.999f, 0.001f); a1 = __builtin_fmaf(a1, 0.999f, 0.001f);
                a2 = __builtin_fmaf(a2, 0.999f, 0.001f); a3 = __builtin_fmaf(a3, 0.999f, 0.001f);
                a4 = __builtin_fmaf(a4, 0.999f, 0.001f); a5 = __builtin_fmaf(a5, 0.999f, 0.001f);
                a6 = __builtin_fmaf(a6, 0.999f, 0.001f); a7 = __builtin_fmaf(a7, 0.999f, 0.001f);
            }
        }
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;  // 64 fma per iter
    } else if (role == 4) {
        float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_exp2f(a0) * 0.5f; a1 = __builtin_amdgcn_exp2f(a1) * 0.5f;
                a2 = __builtin_amdgcn_exp2f(a2) * 0.5f; a3 = __builtin_amdgcn_exp2f(a3) * 0.5f;
            }
        }
        r = a0 + a1 + a2 + a3;  // 32 exp + 32 mul per iter
    } else if (role == 2) {
        f32x16 c0 = {0}, c1 = {0};
        float a = threadIdx.x * 1e-3f, b = 1.0f - a;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) r += c0[q] + c1[q];  // 8 mfma per iter = 512 cycles
    } else if (role == 3) {
        f32x16 c0 = {0}, c1 = {0};
        f16x8 a, b;
        for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(threadIdx.x * 1e-3f + q); b[q] = (_Float16)(1.0f - q * 0.1f); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
            }
        }
        for (int q = 0; q < 16; ++q) r += c0[q] + c1[q];  // 16 mfma per iter = 512 cycles
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int A, int B>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 4000;
    printf("valu-fma alone        %.3f ms\n", run<1, 0>(d, it));
    printf("valu-trans alone      %.3f ms\n", run<4, 0>(d, it));
    printf("mfma-f32 alone        %.3f ms\n", run<2, 0>(d, it));
    printf("mfma-f16 alone        %.3f ms\n", run<3, 0>(d, it));
    printf("mfma-f32 + valu-fma   %.3f ms\n", run<2, 1>(d, it));
    printf("mfma-f16 + valu-fma   %.3f ms\n", run<3, 1>(d, it));
    printf("mfma-f32 + valu-trans %.3f ms\n", run<2, 4>(d, it));
    printf("mfma-f16 + valu-trans %.3f ms\n", run<3, 4>(d, it));
    printf("valu-fma + valu-fma   %.3f ms\n", run<1, 1>(d, it));
    printf("mfma-f32 + mfma-f32   %.3f ms\n", run<2, 2>(d, it));
    printf("mfma-f16 + mfma-f16   %.3f ms\n", run<3, 3>(d, it));
    printf("mfma-f32 + mfma-f16   %.3f ms\n", run<2, 3>(d, it));
    return 0;
}
