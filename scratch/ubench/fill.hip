// How many independent VALU instructions hide in the gap of one MFMA (same wave, one wave per SIMD)?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int KIND, int K, int WAVES>  // KIND 0: f32 32x32x2, 1: f16 32x32x16, 2: f32 16x16x4 ; K fillers per MFMA
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters) {
    f32x16 c0 = {0}, c1 = {0};
    f32x4 d0 = {0}, d1 = {0};
    float a = threadIdx.x * 1e-3f, b = 1.0f - a;
    f16x8 ha, hb;
    for (int q = 0; q < 8; ++q) { ha[q] = (_Float16)(a + q); hb[q] = (_Float16)(b - q * 0.1f); }
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = a + q;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) { if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c0, 0, 0, 0); }
            else if (KIND == 1) { if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0); else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, c0, 0, 0, 0); }
            else { if (u & 1) d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0); else d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, d0, 0, 0, 0); }
#pragma unroll
            for (int q = 0; q < K; ++q) v[q % 16] = __builtin_fmaf(v[q % 16], 0.999f, 0.001f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0;
    for (int q = 0; q < 16; ++q) r += c0[q] + c1[q] + v[q];
    r += d0[0] + d1[0] + d0[1] + d1[1] + d0[2] + d1[2] + d0[3] + d1[3];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = r;
}

template <int KIND, int K, int WAVES>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, K, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int KIND, int WAVES>
void sweep(float* d, const char* name) {
    const int it = 4000;
    printf("%s waves/WG=%d: K=0 %.3f  K=2 %.3f  K=4 %.3f  K=8 %.3f  K=12 %.3f  K=16 %.3f  K=24 %.3f  K=32 %.3f ms\n", name, WAVES,
           run<KIND, 0, WAVES>(d, it), run<KIND, 2, WAVES>(d, it), run<KIND, 4, WAVES>(d, it), run<KIND, 8, WAVES>(d, it),
           run<KIND, 12, WAVES>(d, it), run<KIND, 16, WAVES>(d, it), run<KIND, 24, WAVES>(d, it), run<KIND, 32, WAVES>(d, it));
}

int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    sweep<0, 4>(d, "f32 32x32x2 ");
    sweep<2, 4>(d, "f32 16x16x4 ");
    sweep<1, 4>(d, "f16 32x32x16");
    sweep<0, 8>(d, "f32 32x32x2 ");
    sweep<1, 8>(d, "f16 32x32x16");
    return 0;
}
