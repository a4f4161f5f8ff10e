// cond16.hip -- micro-benchmark for DESIGN.md 10: the three-channel conditioner of a two-particle net (two hidden layers of 64 units + one 32-row output
// block, Taylor triples (f, f', f'') through r(x) = 1 / (2^x + 1), split-fp16 products) on 16-WALKER tiles with v_mfma_f32_16x16x32_f16, against the
// library's 32-walker form (k_etile_cond<false, 1, 3>, 8 waves of 245 registers; timed separately through rocprofv3 with WF_ENERGY_FUSED=0).
// Same arithmetic per walker, same bytes in (4 B) and out (384 B).  Build: hipcc --offload-arch=gfx950 -O3 -DWAVES=16 -I waveflow_amd/csrc -I include ...
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#include "wf_mfma_impl.h"   // split8, f16x8, permlane helpers

#ifndef WAVES
#define WAVES 16
#endif
using namespace wf::mfma;
constexpr int NCH = 3;
constexpr int kWaves = WAVES;
// LDS image (floats): W0 [64], b0 [64], b1 [64], b2 [32], W1 hi [ob 4][s 2][lane 64][8 halves] = 2048, lo 2048, W2 hi [rb 2][s 2][64][8] = 1024, lo 1024
constexpr int oW0 = 0, oB0 = 64, oB1 = 128, oB2 = 192, oW1h = 224, oW1l = oW1h + 2048, oW2h = oW1l + 2048, oW2l = oW2h + 1024, kImg = oW2l + 1024;

__device__ __forceinline__ float max4groups(float v) {   // max over the four lanes (walker, g = 0 .. 3)
    const unsigned u = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return xhalf_max(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
}
__device__ __forceinline__ void act4(f32x4 (&x)[NCH]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x[0][i]) + 1.0f);
        const float r1 = -0.6931471805599453f * __builtin_fmaf(-rr, rr, rr);
        const float k = __builtin_fmaf(1.3862943611198906f, rr, -0.6931471805599453f);
        const float x1 = x[1][i], x2 = x[2][i];
        x[0][i] = rr;
        x[1][i] = r1 * x1;
        x[2][i] = r1 * __builtin_fmaf(k * x1, x1, x2);
    }
}
struct Frag16 {
    f16x8 hi, lo;
};
// four 16-unit blocks of triples -> the B operands of the two K steps of the next layer (K step s contracts blocks 2 s and 2 s + 1: slot (g, i) = unit 4 g + i of
// block 2 s for i < 4, of block 2 s + 1 for i >= 4 -- the registers as they stand), derivative channels scaled by one power of two per (walker, channel)
__device__ __forceinline__ void to_frags16(const f32x4 (&a)[4][NCH], Frag16 (&f)[NCH][2], int (&e)[NCH]) {
    e[0] = 0;
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
        float amax = 0.0f;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fabsf(a[b][c][i]));
        const float m = max4groups(amax);
        e[c] = m > 0.0f ? __builtin_amdgcn_frexp_expf(m) : 0;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float sc = __builtin_amdgcn_ldexpf(1.0f, -e[c]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float r8[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) { r8[i] = a[2 * s][c][i] * sc; r8[4 + i] = a[2 * s + 1][c][i] * sc; }
            split8(r8, f[c][s].hi, f[c][s].lo);
        }
    }
}
// one 16-unit output block of a K = 64 layer, three channels: acc[c] += A (hi + lo) x B (hi + lo) without lo x lo
__device__ __forceinline__ void dense16(const _Float16* Wh, const _Float16* Wl, const Frag16 (&f)[NCH][2], f32x4 (&acc)[NCH], int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(Wh + (s * 64 + lane) * 8);
        const f16x8 al = *reinterpret_cast<const f16x8*>(Wl + (s * 64 + lane) * 8);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, f[c][s].hi, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f[c][s].lo, acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, f[c][s].hi, acc[c], 0, 0, 0);
    }
}
__device__ __forceinline__ void unscale4(f32x4 (&acc)[NCH], const int (&e)[NCH]) {
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
        const float sc = __builtin_amdgcn_ldexpf(1.0f, e[c]);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[c][i] = acc[c][i] * sc;
    }
}

__global__ __launch_bounds__(kWaves * 64) void k_cond16(const float* __restrict__ image, const float* __restrict__ u0g, long B, float* __restrict__ oj) {
    __shared__ __attribute__((aligned(16))) float lds[kImg];
    __shared__ int next_tile;
    if (threadIdx.x == 0) next_tile = 0;
    for (int i = threadIdx.x; i < kImg / 4; i += kWaves * 64) reinterpret_cast<f32x4*>(lds)[i] = reinterpret_cast<const f32x4*>(image)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wl = lane & 15, g = lane >> 4;
    const long n_tiles = (B + 15) >> 4;
    const long my_tiles = n_tiles > (long)blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const _Float16* W1h = reinterpret_cast<const _Float16*>(lds + oW1h);
    const _Float16* W1l = reinterpret_cast<const _Float16*>(lds + oW1l);
    const _Float16* W2h = reinterpret_cast<const _Float16*>(lds + oW2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(lds + oW2l);
    for (;;) {
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(&next_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= my_tiles) break;
        const long tile = (long)blockIdx.x + (long)q * gridDim.x;
        const long w = tile * 16 + wl;
        const bool valid = w < B;
        const float u0 = u0g[valid ? w : B - 1];
        f32x4 a[4][NCH];
        // layer 1: one input (u_0; the second coordinate is masked), Taylor seed (u_0, 1, 0): an FMA per unit
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(lds + oW0 + 16 * b + 4 * g), bv = *reinterpret_cast<const f32x4*>(lds + oB0 + 16 * b + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[b][0][i] = __builtin_fmaf(wv[i], u0, bv[i]); a[b][1][i] = wv[i]; a[b][2][i] = 0.0f; }
            act4(a[b]);
        }
        Frag16 f[NCH][2];
        int e[NCH];
        to_frags16(a, f, e);
        // layer 2
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(lds + oB1 + 16 * ob + 4 * g);
            a[ob][0] = bv; a[ob][1] = f32x4{0, 0, 0, 0}; a[ob][2] = f32x4{0, 0, 0, 0};
            dense16(W1h + ob * 1024, W1l + ob * 1024, f, a[ob], lane);
            unscale4(a[ob], e);
            act4(a[ob]);
        }
        to_frags16(a, f, e);
        // output block: 32 rows = two 16-row blocks
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            f32x4 o[NCH];
            o[0] = *reinterpret_cast<const f32x4*>(lds + oB2 + 16 * rb + 4 * g); o[1] = f32x4{0, 0, 0, 0}; o[2] = f32x4{0, 0, 0, 0};
            dense16(W2h + rb * 1024, W2l + rb * 1024, f, o, lane);
            unscale4(o, e);
            if (valid) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < NCH; ++c) oj[((tile * 32 + 16 * rb + 4 * g + i) * NCH + c) * 16 + wl] = o[c][i];
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const long B = argc > 1 ? atol(argv[1]) : (1L << 17);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.0f, 1.0f);
    std::vector<float> W0(64), b0(64), W1(64 * 64), b1(64), W2(64 * 32), b2(32);
    for (auto& v : W0) v = 2.0f * nd(rng);
    for (auto& v : b0) v = nd(rng);
    for (auto& v : W1) v = 0.4f * nd(rng);     // W1[k][u]
    for (auto& v : b1) v = nd(rng);
    for (auto& v : W2) v = 0.4f * nd(rng);     // W2[k][row]
    for (auto& v : b2) v = nd(rng);
    std::vector<float> img(kImg, 0.0f);
    for (int u = 0; u < 64; ++u) { img[oW0 + u] = W0[u]; img[oB0 + u] = b0[u]; img[oB1 + u] = b1[u]; }
    for (int r = 0; r < 32; ++r) img[oB2 + r] = b2[r];
    auto pack = [&](int off_h, int off_l, int n_ob, const std::vector<float>& W, int ld) {   // A[m = out unit 16 ob + (lane & 15)][slot (g, i)] of K step s
        _Float16* H = reinterpret_cast<_Float16*>(img.data() + off_h);
        _Float16* L = reinterpret_cast<_Float16*>(img.data() + off_l);
        for (int ob = 0; ob < n_ob; ++ob)
            for (int s = 0; s < 2; ++s)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int g = lane >> 4, out = 16 * ob + (lane & 15);
                        const int k = i < 4 ? 16 * (2 * s) + 4 * g + i : 16 * (2 * s + 1) + 4 * g + (i - 4);
                        const float v = W[(size_t)k * ld + out];
                        const _Float16 h = (_Float16)v;
                        H[((ob * 2 + s) * 64 + lane) * 8 + i] = h;
                        L[((ob * 2 + s) * 64 + lane) * 8 + i] = (_Float16)(v - (float)h);
                    }
    };
    pack(oW1h, oW1l, 4, W1, 64);
    pack(oW2h, oW2l, 2, W2, 32);
    std::vector<float> u0(B);
    std::uniform_real_distribution<float> ud(0.0f, 1.0f);
    for (auto& v : u0) v = ud(rng);
    float *d_img, *d_u0, *d_oj;
    const long n_tiles = (B + 15) / 16;
    CK(hipMalloc(&d_img, kImg * 4)); CK(hipMalloc(&d_u0, B * 4)); CK(hipMalloc(&d_oj, n_tiles * 16 * 96 * 4));
    CK(hipMemcpy(d_img, img.data(), kImg * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_u0, u0.data(), B * 4, hipMemcpyHostToDevice));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_cond16, kWaves * 64, 0));
    const unsigned blocks = (unsigned)std::min<long>((n_tiles + kWaves - 1) / kWaves, 256L * std::max(occ, 1));
    hipLaunchKernelGGL(k_cond16, dim3(blocks), dim3(kWaves * 64), 0, 0, d_img, d_u0, B, d_oj);
    CK(hipDeviceSynchronize());
    // check a few walkers against double arithmetic
    std::vector<float> oj((size_t)n_tiles * 16 * 96);
    CK(hipMemcpy(oj.data(), d_oj, oj.size() * 4, hipMemcpyDeviceToHost));
    auto rfun = [](double x, double x1, double x2, double& v0, double& v1, double& v2) {
        const double r = 1.0 / (std::exp2(x) + 1.0), ln2 = 0.6931471805599453;
        const double r1 = -ln2 * r * (1 - r), k = -ln2 * (1 - 2 * r);
        v0 = r; v1 = r1 * x1; v2 = r1 * (k * x1 * x1 + x2);
    };
    double worst = 0.0;
    for (long w : {0L, 1L, 17L, 4099L, B - 1}) {
        double h1[64][3], h2[64][3];
        for (int u = 0; u < 64; ++u) rfun((double)W0[u] * u0[w] + b0[u], W0[u], 0.0, h1[u][0], h1[u][1], h1[u][2]);
        for (int u = 0; u < 64; ++u) {
            double z[3] = {b1[u], 0, 0};
            for (int k = 0; k < 64; ++k) for (int c = 0; c < 3; ++c) z[c] += (double)W1[k * 64 + u] * h1[k][c];
            rfun(z[0], z[1], z[2], h2[u][0], h2[u][1], h2[u][2]);
        }
        for (int r = 0; r < 32; ++r) {
            double o[3] = {b2[r], 0, 0};
            for (int k = 0; k < 64; ++k) for (int c = 0; c < 3; ++c) o[c] += (double)W2[k * 32 + r] * h2[k][c];
            for (int c = 0; c < 3; ++c) {
                const double got = oj[(((w >> 4) * 32 + r) * 3 + c) * 16 + (w & 15)];
                worst = std::max(worst, std::fabs(got - o[c]) / (1e-3 + std::fabs(o[c])));
            }
        }
    }
    printf("16-walker tiles, %d waves per workgroup, occupancy %d workgroup(s) per CU, %u blocks: worst relative deviation from double arithmetic %.2e\n", kWaves, occ, blocks, worst);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_cond16, dim3(blocks), dim3(kWaves * 64), 0, 0, d_img, d_u0, B, d_oj);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_cond16, dim3(blocks), dim3(kWaves * 64), 0, 0, d_img, d_u0, B, d_oj);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %ld walkers: %.2f us per launch (%.3f ns per walker)\n", B, ms / 40 * 1e3, ms / 40 * 1e6 / B);
    }
    return worst < 1e-4 ? 0 : 2;
}
