// Reproducer for the gfx950 hazard behind DESIGN.md §9: a VALU write to a source register (A or B, 128-bit operands) of
// v_mfma_f32_32x32x16_f16 issued right behind the MFMA by the same wave.  hipcc (ROCm 7.2) treats this write-after-read as free;
// with >= 3 waves per SIMD competing for the matrix pipe the MFMA can still be waiting for its operands when the VALU write lands.
//   hipcc --offload-arch=gfx950 -O3 -o war_hazard.bin war_hazard.hip && ./war_hazard.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// WHICH: 0 = overwrite B after the MFMA, 1 = overwrite A, 2 = f32-input MFMA (32x32x2), overwrite B; K = wait states in between
template <int WHICH, int K>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    f32x16 acc = {0};
    u32x4 good_a, good_b;
    for (int q = 0; q < 4; ++q) {
        // A[row][k], B[k][col]: small exact values so that the fp32 sum is exact: A = 1, B = 1 -> every output = 16 per MFMA
        good_a[q] = 0x3C003C00u;  // two halves of 1.0
        good_b[q] = 0x3C003C00u;
    }
    const unsigned bad = 0x70007000u;  // 8192.0, 8192.0
    float fa = 1.0f, fb = 1.0f, fbad = 8192.0f;
    asm volatile("" : "+v"(good_a), "+v"(good_b), "+v"(fa), "+v"(fb), "+v"(fbad));
    for (int i = 0; i < iters; ++i) {
        // A = v[40:43], B = v[44:47] (hard registers: the overwrite must hit exactly the registers the MFMA reads)
        if constexpr (WHICH == 0 || WHICH == 1) {
            asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %1\n\tv_mov_b32 v42, %1\n\tv_mov_b32 v43, %1\n\t"
                         "v_mov_b32 v44, %2\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %2\n\tv_mov_b32 v47, %2\n\t"
                         "s_nop 7\n\t"
                         "v_mfma_f32_32x32x16_f16 %0, v[40:43], v[44:47], %0\n\t"
                         ".rept %c4\n\ts_nop 0\n\t.endr\n\t"
                         ".if %c5 == 0\n\t"
                         "v_mov_b32 v44, %3\n\tv_mov_b32 v45, %3\n\tv_mov_b32 v46, %3\n\tv_mov_b32 v47, %3\n\t"
                         ".else\n\t"
                         "v_mov_b32 v40, %3\n\tv_mov_b32 v41, %3\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %3\n\t"
                         ".endif\n\t"
                         "s_nop 7"
                         : "+v"(acc) : "v"(good_a[0]), "v"(good_b[0]), "v"(bad), "i"(K), "i"(WHICH)
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        } else {
            asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v44, %2\n\ts_nop 7\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, v40, v44, %0\n\t"
                         ".rept %c4\n\ts_nop 0\n\t.endr\n\t"
                         "v_mov_b32 v44, %3\n\ts_nop 7"
                         : "+v"(acc) : "v"(fa), "v"(fb), "v"(fbad), "i"(K) : "v40", "v44");
        }
    }
    asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc));
    float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    for (int q = 0; q < 16; ++q) o[q] = acc[q];
}

template <int WHICH, int K>
void run(float* d, int waves, int iters) {
    const size_t n = (size_t)256 * waves * 64 * 16;
    hipLaunchKernelGGL((k<WHICH, K>), dim3(256), dim3(waves * 64), 0, 0, d, iters);
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    const float expect = (WHICH == 2 ? 2.0f : 16.0f) * iters;
    size_t bad = 0, bad_lanes[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i)
        if (h[i] != expect) { ++bad; ++bad_lanes[((i / 16) % 64) / 16]; }
    printf("%s, %d wait states, %2d waves/WG: %8zu wrong accumulators of %zu  (by lane quarter 0-15/16-31/32-47/48-63: %zu %zu %zu %zu)\n",
           WHICH == 0 ? "f16 32x32x16, VALU write to B" : WHICH == 1 ? "f16 32x32x16, VALU write to A" : "f32 32x32x2,  VALU write to B", K, waves, bad, n,
           bad_lanes[0], bad_lanes[1], bad_lanes[2], bad_lanes[3]);
}

int main() {
    float* d;
    (void)hipMalloc(&d, (size_t)256 * 1024 * 16 * 4);
    const int it = 1000;
    for (int waves : {4, 8, 12, 16}) {
        run<0, 0>(d, waves, it); run<0, 1>(d, waves, it); run<0, 2>(d, waves, it); run<0, 3>(d, waves, it); run<0, 4>(d, waves, it); run<0, 6>(d, waves, it); run<0, 8>(d, waves, it);
        run<1, 0>(d, waves, it); run<1, 1>(d, waves, it); run<1, 2>(d, waves, it); run<1, 3>(d, waves, it); run<1, 4>(d, waves, it); run<1, 6>(d, waves, it); run<1, 8>(d, waves, it);
        run<2, 0>(d, waves, it); run<2, 1>(d, waves, it); run<2, 2>(d, waves, it);
    }
    return 0;
}
