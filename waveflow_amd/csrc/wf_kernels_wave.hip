// wf_kernels_wave.hip -- wave-cooperative ring kernels: local energy and parameter gradients, gfx950.
//
// One WAVE evaluates one sample (a walker in R1 / RF, or a (walker, direction) pair in the second-order ring R3 -- wf_ring.h); its 64
// lanes are the 64 hidden units of the conditioner (model_factory.py:72), and in the output layer the 2 x 32 basis rows of
// two dimensions.  A dense layer is then 64 fused multiply-adds per lane and coefficient instead of 4096: the input vector
// is broadcast from LDS (ds_read_b128, same address in every lane), the weights come in lane-major float4 groups (one
// 1 KB contiguous load per instruction, NetWave in wf_internal.h).  Normalisations and spline sums are 32-lane butterfly
// reductions.  Compared with the one-lane-per-walker form this cuts the serial chain of a sample by ~60x (what bounds a
// 128..256-walker training step) and needs no register spills.
//
//   k_wave_fwd   forward ring evaluation; writes the tape (layer inputs, hidden activations) and the tail
//                (per-dimension prior factors, log det, latent point) of each sample
//   k_energy_out psi, laplacian, H psi per walker from the tails      physics.py:50-52, 60-76, 79-93
//                (second order: RF<D> = (value, gradient, Laplacian / 2) jets, one sample per walker; R3 is the A/B form)
//   k_wave_bwd   reverse sweep from the tape: pre-activation adjoints into the tape for k_wgrad (wf_kernels_grad.hip)
//
// Derivative semantics of the table lerp and the adjoint-in-reversed-order ring trick: see wf_kernels_grad.hip / wf_ring.h.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "wf_internal.h"
#include "wf_ring.h"
#include "wf_scalar_impl.h"   // Philox4x32-10 and the box reverse (shared with the one-lane-per-walker sampler)

namespace wf {

namespace {

using namespace ring;
constexpr int H = kHidden;
constexpr int kWaves = 4;
// waves per SIMD the register allocation of each kernel family aims at (measured: DESIGN.md §4.5, scratch/occ_try.sh)
#ifndef WF_OCC_FWD1
#define WF_OCC_FWD1 3   // k_wave_fwd<D, R1>
#endif
#ifndef WF_OCC_FWD2
#define WF_OCC_FWD2 3   // k_wave_fwd<D, R3 / RF>
#endif
#ifndef WF_OCC_BWD1
#define WF_OCC_BWD1 3   // k_wave_bwd<D, R1>
#endif
#ifndef WF_OCC_BWD2
#define WF_OCC_BWD2 2   // k_wave_bwd<D, R3 / RF>: 256 registers per lane (RF<2>: 1.92e7 -> 2.31e7 walkers/s, batch 128: 86 -> 60 us)
#endif
#ifndef WF_OCC_SAMPLE
#define WF_OCC_SAMPLE 3
#endif
template <class T> constexpr int kOccFwd = T::NC == 1 ? WF_OCC_FWD1 : WF_OCC_FWD2;
template <class T> constexpr int kOccBwd = T::NC == 1 ? WF_OCC_BWD1 : WF_OCC_BWD2;
constexpr int kWB = 64 * kWaves;

// ---- coefficient access
__device__ __forceinline__ float coef(R1 a, int) { return a.c0; }
__device__ __forceinline__ float coef(R3 a, int k) { return k == 0 ? a.c0 : (k == 1 ? a.c1 : a.c2); }
__device__ __forceinline__ R1 from_arr(R1*, const float* c) { return R1{c[0]}; }
__device__ __forceinline__ R3 from_arr(R3*, const float* c) { return R3{c[0], c[1], c[2]}; }
template <int D> __device__ __forceinline__ float coef(RF<D> a, int k) {
    float r = a.c0;
#pragma unroll
    for (int i = 0; i < D; ++i) r = k == i + 1 ? a.g[i] : r;
    return k == D + 1 ? a.h : r;
}
template <int D> __device__ __forceinline__ RF<D> from_arr(RF<D>*, const float* c) {
    RF<D> r;
    r.c0 = c[0];
#pragma unroll
    for (int i = 0; i < D; ++i) r.g[i] = c[i + 1];
    r.h = c[D + 1];
    return r;
}
// componentwise map of a ring value
template <class T, class F> __device__ __forceinline__ T map_coefs(T a, F f) {
    float c[T::NC];
#pragma unroll
    for (int k = 0; k < T::NC; ++k) c[k] = f(coef(a, k));
    return from_arr((T*)nullptr, c);
}
template <class T> __device__ __forceinline__ T sel(bool first, T a, T b) { return first ? a : b; }

// ---- cross-lane: DPP butterflies inside a row of 16 lanes, the gfx950 permlane swaps across rows / halves
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// rows 0,1 and rows 2,3 exchanged (v_permlane16_swap) / halves exchanged (v_permlane32_swap); the wait states around the
// swaps are spelled out
__device__ __forceinline__ float swap16_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap32_other(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    // a = [lo, lo], b = [hi, hi]: the value of the other half is whichever differs from mine
    return (threadIdx.x & 32) ? a : b;
}
__device__ __forceinline__ float hsum(float v) {   // sum over the 32 lanes of this lane's half, in every lane
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror
    return swap16_sum(v);
}
__device__ __forceinline__ R1 half_sum(R1 a) { return R1{hsum(a.c0)}; }
__device__ __forceinline__ R3 half_sum(R3 a) { return R3{hsum(a.c0), hsum(a.c1), hsum(a.c2)}; }
template <int D> __device__ __forceinline__ RF<D> half_sum(RF<D> a) { return map_coefs(a, [](float v) { return hsum(v); }); }
__device__ __forceinline__ R1 xhalf(R1 a) { return R1{swap32_other(a.c0)}; }
__device__ __forceinline__ R3 xhalf(R3 a) { return R3{swap32_other(a.c0), swap32_other(a.c1), swap32_other(a.c2)}; }
template <int D> __device__ __forceinline__ RF<D> xhalf(RF<D> a) { return map_coefs(a, [](float v) { return swap32_other(v); }); }
template <class T> __device__ __forceinline__ T wave_sum(T a) {
    const T h = half_sum(a);
    return h + xhalf(h);
}
// sum over the rows of one dimension: the 32 lanes of a half (NBK = 1: a pass holds two dimensions) or the whole wave (NBK = 2)
template <int NBK, class T> __device__ __forceinline__ T rsum(T a) {
    if constexpr (NBK == 1) return half_sum(a);
    else return wave_sum(a);
}
template <int NBK> __device__ __forceinline__ float rsumf(float v) {
    const float h = hsum(v);
    if constexpr (NBK == 1) return h;
    else return h + swap32_other(h);
}
__device__ __forceinline__ R1 from_lane(R1 a, int src) { return R1{__shfl(a.c0, src)}; }
__device__ __forceinline__ R3 from_lane(R3 a, int src) { return R3{__shfl(a.c0, src), __shfl(a.c1, src), __shfl(a.c2, src)}; }
template <int D> __device__ __forceinline__ RF<D> from_lane(RF<D> a, int src) { return map_coefs(a, [src](float v) { return __shfl(v, src); }); }

// ---- per-wave LDS vector [NC][64]
template <class T> __device__ __forceinline__ void put(float (*buf)[64], int lane, T v) {
#pragma unroll
    for (int k = 0; k < T::NC; ++k) buf[k][lane] = coef(v, k);
    __builtin_amdgcn_wave_barrier();
}
// out[lane] = sum_a in[a] * W[lane][a], W in lane-major float4 groups.  Even and odd a accumulate in the two halves of a
// packed pair (v_pk_fma_f32: two FMAs per lane and instruction) and are added at the end.
using float2_t = __attribute__((ext_vector_type(2))) float;
#ifndef WF_GEMV_UNROLL_BWD
#define WF_GEMV_UNROLL_BWD 4
#endif
template <class T, bool PRELOAD = false, int UNROLL = 4>
__device__ __forceinline__ T gemv(const float4_t* __restrict__ img, const float (*buf)[64], int lane) {
    float2_t acc[T::NC];
#pragma unroll
    for (int k = 0; k < T::NC; ++k) acc[k] = float2_t{0.0f, 0.0f};
    if constexpr (T::NC == 1 && PRELOAD) {
        // the sampler: all 16 weight loads of the lane are requested up front (one L2 round trip per product instead of
        // four -- it runs at low occupancy on small batches, where that latency is the chain: 70 -> 58 us for 128 walkers;
        // in the sweeps the extra 64 live registers cost more than they save)
        float4_t w[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) w[g] = img[g * 64 + lane];
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float4_t x = *reinterpret_cast<const float4_t*>(&buf[0][4 * g]);
            acc[0] = __builtin_elementwise_fma(float2_t{w[g].x, w[g].y}, float2_t{x.x, x.y}, acc[0]);
            acc[0] = __builtin_elementwise_fma(float2_t{w[g].z, w[g].w}, float2_t{x.z, x.w}, acc[0]);
        }
    } else {
        // groups of 4 (full unrolling made hipcc hold 48 LDS reads live in the second-order kernels: 256 VGPRs)
#pragma unroll UNROLL
        for (int g = 0; g < 16; ++g) {
            const float4_t w = img[g * 64 + lane];
            const float2_t wlo = {w.x, w.y}, whi = {w.z, w.w};
#pragma unroll
            for (int k = 0; k < T::NC; ++k) {
                const float4_t x = *reinterpret_cast<const float4_t*>(&buf[k][4 * g]);
                acc[k] = __builtin_elementwise_fma(wlo, float2_t{x.x, x.y}, acc[k]);
                acc[k] = __builtin_elementwise_fma(whi, float2_t{x.z, x.w}, acc[k]);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    float out[T::NC];
#pragma unroll
    for (int k = 0; k < T::NC; ++k) out[k] = acc[k].x + acc[k].y;
    return from_arr((T*)nullptr, out);
}
// out[lane (half h, j)] = sum_{a < 32} in[h][a] * M[a][j]   (M: [32][32] row-major; in = buf of this lane's half);
// NBK = 2: out[lane j] = sum_{a < 64} in[a] * M[a][j], M [64][64]
template <class T, int NBK = 1>
__device__ __forceinline__ T gemv32_cols(const float* __restrict__ M, const float (*buf)[64], int half, int j) {
    constexpr int W = 32 * NBK;
    const int base = NBK == 1 ? half * 32 : 0;
    float acc[T::NC];
#pragma unroll
    for (int k = 0; k < T::NC; ++k) acc[k] = 0.0f;
#pragma unroll 8
    for (int a = 0; a < W; ++a) {
        const float m = M[a * W + j];
#pragma unroll
        for (int k = 0; k < T::NC; ++k) acc[k] = __builtin_fmaf(buf[k][base + a], m, acc[k]);
    }
    __builtin_amdgcn_wave_barrier();
    return from_arr((T*)nullptr, acc);
}
// out[lane (half h, a)] = sum_{j < 32} in[h][j] * M[a][j]
template <class T, int NBK = 1>
__device__ __forceinline__ T gemv32_rows(const float* __restrict__ M, const float (*buf)[64], int half, int a) {
    constexpr int W = 32 * NBK;
    const int base = NBK == 1 ? half * 32 : 0;
    float acc[T::NC];
#pragma unroll
    for (int k = 0; k < T::NC; ++k) acc[k] = 0.0f;
#pragma unroll 8
    for (int j = 0; j < W; ++j) {
        const float m = M[a * W + j];
#pragma unroll
        for (int k = 0; k < T::NC; ++k) acc[k] = __builtin_fmaf(buf[k][base + j], m, acc[k]);
    }
    __builtin_amdgcn_wave_barrier();
    return from_arr((T*)nullptr, acc);
}

// ---- tape: ws[((sample * n_nets + net) * NC + coefficient) * ROWS + row], ring::Rows
struct Tape {
    float* base;      // of this sample
    int rows;
};
template <class T> __device__ __forceinline__ void tput(const Tape& t, int net, int row, T v) {
#pragma unroll
    for (int k = 0; k < T::NC; ++k) t.base[((int64_t)net * T::NC + k) * t.rows + row] = coef(v, k);
}
template <class T> __device__ __forceinline__ T tget(const Tape& t, int net, int row) {
    float c[T::NC];
#pragma unroll
    for (int k = 0; k < T::NC; ++k) c[k] = t.base[((int64_t)net * T::NC + k) * t.rows + row];
    return from_arr((T*)nullptr, c);
}
// uniform ring values in the tail: tail[sample][slot][coefficient]
template <class T> __device__ __forceinline__ void tail_put(float* tl, int slot, T v) {
#pragma unroll
    for (int k = 0; k < T::NC; ++k) tl[slot * T::NC + k] = coef(v, k);
}
template <class T> __device__ __forceinline__ T tail_get(const float* tl, int slot) {
    float c[T::NC];
#pragma unroll
    for (int k = 0; k < T::NC; ++k) c[k] = tl[slot * T::NC + k];
    return from_arr((T*)nullptr, c);
}
template <int D> struct Tail {   // slots
    static constexpr int V = 0, LD = D, U = D + 1, N = 2 * D + 1;
};

// two masked tanh layers, lane = hidden unit; leaves h2 in `vec` and returns it
template <int D, class T, bool PRELOAD = false>
__device__ __forceinline__ T hidden_fwd(const NetWave& net, const T (&x)[D], float (*vec)[64], int lane, const Tape& tape, int n, bool taped) {
    T z = cst<T>(net.b0[lane]);
#pragma unroll
    for (int a = 0; a < D; ++a) z = z + x[a] * net.W0[a * H + lane];
    const T h1 = rtanh(z);
    if (taped) tput(tape, n, Rows<D>::H1 + lane, h1);
    put(vec, lane, h1);
    const T h2 = rtanh(gemv<T, PRELOAD>(net.W1f, vec, lane) + net.b1[lane]);
    if (taped) tput(tape, n, Rows<D>::H2 + lane, h2);
    put(vec, lane, h2);
    return h2;
}

// state of one sigmoid head lane (IMADE layer or M-spline prior): c = g (p / S0 + reg) / Q, see wf_kernels_grad.hip
template <class T> struct SigHead {
    T p, rS0, rQ, c;
    T s;   // sigmoid(o) itself (== p unless the head is gated)
};
// gate (forward sweep only): p = gq * sigmoid(o) + z, the gated head of model_factory.py:64-67 (gq: the jet of prod_{i<d} x_i^3)
template <class T, int NBK = 1>
__device__ __forceinline__ SigHead<T> sigmoid_head(T o, bool valid, bool valid_d, float g, float reg, bool gate = false, T gq = T{}, float z = 0.0f) {
    SigHead<T> h;
    h.p = valid ? rsigmoid(o) : cst<T>(0.0f);
    h.s = h.p;
    if (gate && valid) h.p = gq * h.p + z;
    T S0 = rsum<NBK>(h.p);
    if (!valid_d) S0 = cst<T>(1.0f);
    h.rS0 = rrcp(S0);
    const T q = (h.p * h.rS0 + reg) * (valid ? g : 0.0f);
    T Q = rsum<NBK>(q);
    if (!valid_d) Q = cst<T>(1.0f);
    h.rQ = rrcp(Q);
    h.c = q * h.rQ;
    return h;
}
// reverse of the head: cbar (this lane) -> obar (this lane)
// Gated head (p = gq * sigmoid(o) + z): obar = (pbar * gq) * s (1 - s); *pbar_out (if given) receives pbar, the adjoint of p, from
// which the caller takes the adjoints of the gate (pbar * s, summed over the rows) and of z (pbar itself).
template <class T, int NBK = 1>
__device__ __forceinline__ T sigmoid_head_bwd(const SigHead<T>& h, T gc, bool valid, float g, bool gate = false, T gq = T{}, T* pbar_out = nullptr) {
    const T dotC = rsum<NBK>(gc * h.c);
    const T gw0 = ((gc - dotC) * h.rQ) * (valid ? g : 0.0f);
    const T dot0 = rsum<NBK>(gw0 * (h.p * h.rS0));
    const T pbar = valid ? (gw0 - dot0) * h.rS0 : cst<T>(0.0f);
    if (pbar_out) *pbar_out = pbar;
    if (gate) return valid ? (pbar * gq) * (h.s * (1.0f - h.s)) : cst<T>(0.0f);
    return valid ? pbar * (h.p * (1.0f - h.p)) : cst<T>(0.0f);
}

// psi head lane state (wavefunctions.py:54-71, bsplines_jax.py:127-137 with zero-only constraints).  The reference divides the
// raw outputs by their sum S (model_factory.py:69) before the boundary rows are zeroed and the vector is L2-normalised; the
// scale cancels in that normalisation, only sign(S) survives: a = sign(S) (o keep) / |o keep|.  Evaluated in that form: S is a
// signed sum that passes through zero (a walker of a 4096-batch He run had S = 2.6e-7), where the quotient form turns fp32
// rounding into Laplacians of 1e3..inf while the function itself is smooth.
template <class T> struct PsiHead {
    T o, rN1, rN2, a, e;
    T sabs;      // |S| = sgn * sum of the raw outputs (used with a constant term of the boundary map only)
    float sgn;
};
template <class T, int NBK = 1>
__device__ __forceinline__ PsiHead<T> psi_head(T o, bool valid, bool valid_d, float keep, const float* __restrict__ o2b, float (*ov)[64], int lane,
                                               bool gate = false, T gq = T{}, float z = 0.0f, const float* __restrict__ cb = nullptr) {
    PsiHead<T> h;
    h.o = valid ? o : cst<T>(0.0f);
    if (gate && valid) h.o = gq * o + z;
    h.sgn = rsumf<NBK>(h.o.c0) < 0.0f ? -1.0f : 1.0f;
    const T w = h.o * (valid ? keep * h.sgn : 0.0f);
    T N1 = rsum<NBK>(w * w);
    if (!valid_d) N1 = cst<T>(1.0f);
    h.rN1 = rrsqrt(N1);
    h.a = w * h.rN1;
    put(ov, lane, h.a);
    T c = gemv32_cols<T, NBK>(o2b, ov, lane >> 5, NBK == 1 ? (lane & 31) : lane);   // c_j = sum_a a_a ob_to_b[a][j]
    h.sabs = cst<T>(0.0f);
    if (cb) {
        // a boundary constraint with a non-zero value (bsplines_jax.py:173-199): the weights reach the constraints divided by their sum S
        // (model_factory.py:69), w' = (A o + S b) / S, so sign(S) w' M is proportional to a M + (|S| / |o keep|) (b M), b M = cb
        h.sabs = rsum<NBK>(h.o) * h.sgn;
        c = c + (h.sabs * h.rN1) * cb[NBK == 1 ? (lane & 31) : lane];
    }
    T N2 = rsum<NBK>(c * c);
    if (!valid_d) N2 = cst<T>(1.0f);
    h.rN2 = rrsqrt(N2);
    h.e = c * h.rN2;
    return h;
}

template <class T> __device__ __forceinline__ T clip01(T u, bool& inside) {
    inside = true;
    if (u.c0 < 0.0f) { inside = false; return cst<T>(0.0f); }
    if (u.c0 > 1.0f) { inside = false; return cst<T>(1.0f); }
    return u;
}

template <int D, class T>
__device__ __forceinline__ void box_forward(const ModelDev& md, T (&cur)[D], T& logdet) {
    const float L = md.box_L, tol = 1e-7f;
    T nxt[D];
    if (md.box_kind == WF_BOX_MEAN) {
        T sm = cst<T>(0.0f);
#pragma unroll
        for (int d = 0; d < D; ++d) sm = sm + cur[d];
        const T mean = sm * (1.0f / (float)D);
        const T l = mean - cur[0];
        const T wd = cur[D - 1] - cur[0];
        T space = cst<T>(2 * L);
#pragma unroll
        for (int i = 0; i < D - 1; ++i) {
            const T diff = cur[i + 1] - cur[i];
            nxt[i] = diff * rrcp(space + tol);
            logdet = logdet - rlog(space + tol);
            space = space - diff;
        }
        const T den = (2 * L - wd) + tol;
        nxt[D - 1] = ((mean + L) - l) * rrcp(den);
        logdet = logdet - rlog(den);
    } else if (md.box_kind == WF_BOX_FIRST) {
        nxt[0] = (cur[0] + L) * (1.0f / (2 * L));
        T ls = cst<T>(0.0f);
#pragma unroll
        for (int i = 1; i < D; ++i) nxt[i] = (cur[i] - cur[i - 1]) * rrcp((L - cur[i - 1]) + tol);
#pragma unroll
        for (int i = 0; i < D - 1; ++i) ls = ls + rlog((L - cur[i]) + tol);
        logdet = cst<T>(-logf(2 * L)) - ls;
    } else {
        return;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = nxt[d];
}

// ------------------------------------------------------------------------------------------------ forward
// NBK = 1: an output pass covers two dimensions x 32 basis rows (lane = (dl, j)); NBK = 2 (33..64 bases): one dimension x 64 rows
template <int D, class T, int NBK = 1>
__global__ __launch_bounds__(kWB) __attribute__((amdgpu_waves_per_eu(kOccFwd<T>, kOccFwd<T>))) void k_wave_fwd(const ModelDev* __restrict__ mdp, const float* __restrict__ tabI, const float* __restrict__ tabP,
                                                   const float* __restrict__ fk_nat, const float* __restrict__ xg, int64_t B,
                                                   float* __restrict__ ws, float* __restrict__ tails, int taped) {
    __shared__ float lds[kWaves][2][T::NC][64];
    const ModelDev& md = *mdp;
    constexpr int DIRS = kDirs<T, D>;
    constexpr int P = NBK == 1 ? (D + 1) / 2 : D, W = 32 * NBK;
    using RW = Rows<D, NBK>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float (*vec)[64] = lds[wv][0];
    float (*ov)[64] = lds[wv][1];
    const int dl = NBK == 1 ? lane >> 5 : 0, j = NBK == 1 ? (lane & 31) : lane;
    const bool imade = md.layer_kind == WF_LAYER_IMADE;
    const bool has_pnet = md.prior_kind == WF_PRIOR_WAVEFLOW || md.prior_kind == WF_PRIOR_MFLOW;
    const int n_nets = md.n_layers + (has_pnet ? 1 : 0);
    const int n_mesh = (imade && md.n_layers > 0) ? md.isp.n_mesh : md.psp.n_mesh;   // (a model may have no flow layer at all)
    const size_t plane = (size_t)n_mesh * W;
    const float* __restrict__ gI = fk_nat;
    const float* __restrict__ kP = fk_nat + 64;
    T* const tag = nullptr;
    const int64_t n_samples = B * DIRS;
    for (int64_t s = (int64_t)blockIdx.x * kWaves + wv; s < n_samples; s += (int64_t)gridDim.x * kWaves) {
        const int64_t b = s / DIRS;
        const int dir = (int)(s - b * DIRS);
        const Tape tape{ws + s * (int64_t)n_nets * T::NC * RW::N, RW::N};
        float* tl = tails + s * (int64_t)Tail<D>::N * T::NC;
        T cur[D], nxt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = make_var(tag, xg[b * D + d], d, dir);
        T logdet = cst<T>(0.0f);
        box_forward<D, T>(md, cur, logdet);
        for (int l = 0; l < md.n_layers; ++l) {
            const NetWave& net = md.wnets[l];
            if (taped) {
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (lane == d) tput(tape, l, RW::U + d, cur[d]);
            }
            hidden_fwd<D, T>(net, cur, vec, lane, tape, l, taped);
            const bool gate_i = md.i_gate != 0;
            T grun = cst<T>(1.0f);   // gated heads: the jet of prod_{i<d} (layer input)_i^3, advanced pass by pass
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int d = NBK == 1 ? 2 * p + dl : p;
                const bool valid_d = d < D;
                const T o = gemv<T>(net.W2f + p * 1024, vec, lane) + net.b2[p * 64 + lane];
                if (taped && valid_d) tput(tape, l, RW::O + d * W + j, o);   // the reverse sweep reads it back and stores its adjoint here
                const T u = NBK == 1 ? sel(dl == 0, cur[2 * p < D ? 2 * p : 0], cur[2 * p + 1 < D ? 2 * p + 1 : 0]) : cur[p];
                T gq = grun;
                if (gate_i) {
                    if constexpr (NBK == 1) {
                        const T c0 = cur[2 * p < D ? 2 * p : 0];
                        const T g_odd = grun * (c0 * c0 * c0);
                        gq = sel(dl == 0, grun, g_odd);
                        const T c1 = cur[2 * p + 1 < D ? 2 * p + 1 : 0];
                        grun = g_odd * (c1 * c1 * c1);
                    } else {
                        grun = grun * (cur[p] * cur[p] * cur[p]);
                    }
                }
                T y, ld;
                if (imade) {
                    const int nb = md.isp.nb;
                    const bool valid = valid_d && j < nb;
                    const SigHead<T> hd = sigmoid_head<T, NBK>(o, valid, valid_d, gI[j], md.i_reg, gate_i, gq, net.z[p * 64 + lane]);
                    const Lerp lp = make_lerp(u.c0, n_mesh);
                    float t[4];
                    lerp4<W>(tabI, plane, lp, j, t);
                    y = rsum<NBK>(hd.c * lift(t, 0, u));
                    const T dy = rsum<NBK>(hd.c * lift(t, 1, u));
                    ld = valid_d ? rlog(dy + 1e-7f) : cst<T>(0.0f);
                } else {
                    // MADE (made.py:21-27): rows 0 / 1 of the dimension are log_weight / bias
                    const T lw = from_lane(o, dl * 32), bias = from_lane(o, dl * 32 + 1);
                    y = (u - bias) * rexp(cst<T>(0.0f) - lw);
                    ld = valid_d ? cst<T>(0.0f) - lw : cst<T>(0.0f);
                }
                if constexpr (NBK == 1) {
                    const T y_o = xhalf(y), ld_o = xhalf(ld);
                    nxt[2 * p] = sel(dl == 0, y, y_o);
                    if (2 * p + 1 < D) nxt[2 * p + 1] = sel(dl == 0, y_o, y);
                    logdet = logdet + ld + ld_o;
                } else {
                    nxt[p] = y;
                    logdet = logdet + ld;
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[D - 1 - d];
        }
        // ---- prior
        T v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) v[d] = cst<T>(1.0f);
        if (has_pnet) {
            const int NP = md.n_layers;
            const NetWave& net = md.wnets[NP];
            const int nb = md.psp.nb;
            if (taped) {
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (lane == d) tput(tape, NP, RW::U + d, cur[d]);
            }
            hidden_fwd<D, T>(net, cur, vec, lane, tape, NP, taped);
            const bool gate_p = md.p_gate != 0;
            T grun = cst<T>(1.0f);   // (the gate sees the conditioner's input: the unclipped u)
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int d = NBK == 1 ? 2 * p + dl : p;
                const bool valid_d = d < D, valid = valid_d && j < nb;
                const T o = gemv<T>(net.W2f + p * 1024, vec, lane) + net.b2[p * 64 + lane];
                if (taped && valid_d) tput(tape, NP, RW::O + d * W + j, o);
                T gq = grun;
                if (gate_p) {
                    if constexpr (NBK == 1) {
                        const T c0 = cur[2 * p < D ? 2 * p : 0];
                        const T g_odd = grun * (c0 * c0 * c0);
                        gq = sel(dl == 0, grun, g_odd);
                        const T c1 = cur[2 * p + 1 < D ? 2 * p + 1 : 0];
                        grun = g_odd * (c1 * c1 * c1);
                    } else {
                        grun = grun * (cur[p] * cur[p] * cur[p]);
                    }
                }
                bool inside;
                const T uc = clip01(NBK == 1 ? sel(dl == 0, cur[2 * p < D ? 2 * p : 0], cur[2 * p + 1 < D ? 2 * p + 1 : 0]) : cur[p], inside);
                const Lerp lp = make_lerp(uc.c0, n_mesh);
                float t[4];
                lerp4<W>(tabP, plane, lp, j, t);
                T val;
                if (md.prior_kind == WF_PRIOR_WAVEFLOW) {
                    const PsiHead<T> hd = psi_head<T, NBK>(o, valid, valid_d, kP[j], md.ob_to_b_t, ov, lane, gate_p, gq, net.z[p * 64 + lane], md.p_cb);
                    val = rsum<NBK>(hd.e * lift(t, 0, uc));
                } else {
                    const SigHead<T> hd = sigmoid_head<T, NBK>(o, valid, valid_d, kP[j], 0.0f, gate_p, gq, net.z[p * 64 + lane]);
                    val = rsum<NBK>(hd.c * lift(t, 0, uc));
                }
                if constexpr (NBK == 1) {
                    const T val_o = xhalf(val);
                    v[2 * p] = sel(dl == 0, val, val_o);
                    if (2 * p + 1 < D) v[2 * p + 1] = sel(dl == 0, val_o, val);
                } else {
                    v[p] = val;
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                tail_put(tl, Tail<D>::V + d, v[d]);
                tail_put(tl, Tail<D>::U + d, cur[d]);
            }
            tail_put(tl, Tail<D>::LD, logdet);
        }
    }
}

// psi of one sample from its tail: prod_d (v_d scale_d) * exp(logdet / 2)
template <int D, class T>
__device__ __forceinline__ T psi_from_tail(const float* tl, unsigned constrained_mask, T (&v)[D], T& E) {
    T prod = cst<T>(1.0f);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        v[d] = tail_get<T>(tl, Tail<D>::V + d);
        prod = prod * (v[d] * (((constrained_mask >> d) & 1u) ? 0.70710678118654752f : 1.0f));
    }
    E = rexp(tail_get<T>(tl, Tail<D>::LD) * 0.5f);
    return prod * E;
}

// ---- H psi = -1/2 laplacian + V psi per walker from the R3 tails of its D directions, or from its one RF tail
template <int D, class T>
__global__ void k_energy_out(const float* __restrict__ tails, const float* __restrict__ xg, int64_t B, unsigned constrained_mask, const Protons pr,
                             float* __restrict__ hpsi, float* __restrict__ psi_out, float* __restrict__ lap_out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float lap = 0.0f, psv = 0.0f;
    constexpr int DIRS = kDirs<T, D>;
#pragma unroll
    for (int dir = 0; dir < DIRS; ++dir) {
        T v[D], E;
        const T ps = psi_from_tail<D, T>(tails + (b * DIRS + dir) * (int64_t)Tail<D>::N * T::NC, constrained_mask, v, E);
        lap += lap_of(ps);
        psv = ps.c0;
    }
    float V = 0.0f;   // physics.py:60-76
    for (int p = 0; p < pr.n; ++p)
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float r = pr.pos[p] - xg[b * D + d];
            V -= 1.0f / sqrtf(1.0f + r * r);
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const float r = xg[b * D + i] - xg[b * D + k];
            V += 1.0f / sqrtf(1.0f + r * r);
        }
    hpsi[b] = -0.5f * lap + V * psv;
    if (psi_out) psi_out[b] = psv;
    if (lap_out) lap_out[b] = lap;
}

// ---- log_pdf / psi / (u, log det) per walker from the R1 tails (wavefunctions.py:33-71, distributions.py:95-102, 139-163)
// mode 0: log_pdf, 1: psi, 2: log det only;  u_out (may be null): the latent point (clipped where the reference clips it)
template <int D>
__global__ void k_tail_out(const float* __restrict__ tails, int64_t B, int mode, int prior_kind, unsigned constrained_mask, float normal_offset,
                           float* __restrict__ out, float* __restrict__ u_out, float* __restrict__ w_out, float w_value) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (w_out) w_out[b] = w_value;   // constant per-walker weight of a mean objective (seed of the reverse sweep)
    const float* tl = tails + b * (int64_t)Tail<D>::N;
    const float ld = tl[Tail<D>::LD];
    float res = ld;
    if (mode == 0) {
        float lp = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float v = tl[Tail<D>::V + d];
            if (prior_kind == WF_PRIOR_WAVEFLOW) {
                float pr = v * v;
                if ((constrained_mask >> d) & 1u) pr = pr / 2;
                lp = lp + logf(pr + 1e-7f);
            } else if (prior_kind == WF_PRIOR_MFLOW) {
                lp = lp + logf(v + 1e-7f);
            } else if (prior_kind == WF_PRIOR_NORMAL) {
                const float z = tl[Tail<D>::U + d] + normal_offset;
                lp = lp + (1.8378770664093453f + z * z) / -2.0f;
            }
        }
        res = lp + ld;
    } else if (mode == 1) {
        float prod = 1.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float v = tl[Tail<D>::V + d];
            if ((constrained_mask >> d) & 1u) v = v / sqrtf(2.0f);
            prod = prod * v;
        }
        res = prod * expf(0.5f * ld);
    }
    out[b] = res;
    if (u_out) {
        const bool clip = mode != 2 && prior_kind != WF_PRIOR_NORMAL;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float u = tl[Tail<D>::U + d];
            u_out[b * D + d] = clip ? fminf(fmaxf(u, 0.0f), 1.0f) : u;
        }
    }
}

// local energy and the weights of loss_fn_efficient's tangent rule from (psi, laplacian) of one walker at x[0..D)
template <int D>
__device__ __forceinline__ void seed_values(float ps, float lap, const float* __restrict__ x, const Protons& pr, float running_avg, float inv_count,
                                            float& e_loc, float& w_psi, float& w_lap) {
    float V = 0.0f;   // physics.py:60-76
    for (int p = 0; p < pr.n; ++p)
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float r = pr.pos[p] - x[d];
            V -= 1.0f / sqrtf(1.0f + r * r);
        }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int k = 0; k < i; ++k) {
            const float r = x[i] - x[k];
            V += 1.0f / sqrtf(1.0f + r * r);
        }
    const float hp = -0.5f * lap + V * ps;
    const float el = hp / (ps + 1e-8f);
    // 2 (E_L - avg) / psi - H psi / psi^2 (vqmc.py:205-210), written without psi^2: the product of D small factors squared
    // underflows in fp32 for larger D (an 8-electron chain at its initial parameters), and inf - inf would poison the step
    const float a = (2.0f * (el - running_avg) - hp / ps) / ps;
    const float c = 1.0f / ps;
    e_loc = el;
    w_psi = (a + c * V) * inv_count;
    w_lap = -0.5f * c * inv_count;
}

// ---- the same, continued to the weights of loss_fn_efficient's tangent rule (k_vqmc_seeds, wf_kernels_grad.hip): the fused
// training step needs neither H psi nor psi in memory
template <int D, class T>
__global__ void k_energy_seeds(const float* __restrict__ tails, const float* __restrict__ xg, int64_t B, unsigned constrained_mask, const Protons pr,
                               float running_avg, const float* __restrict__ running_avg_dev, float inv_count, float* __restrict__ e_loc,
                               float* __restrict__ w_psi, float* __restrict__ w_lap) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (running_avg_dev) running_avg = *running_avg_dev;
    float lap = 0.0f, ps = 0.0f;
    constexpr int DIRS = kDirs<T, D>;
#pragma unroll
    for (int dir = 0; dir < DIRS; ++dir) {
        T v[D], E;
        const T p3 = psi_from_tail<D, T>(tails + (b * DIRS + dir) * (int64_t)Tail<D>::N * T::NC, constrained_mask, v, E);
        lap += lap_of(p3);
        ps = p3.c0;
    }
    seed_values<D>(ps, lap, xg + b * D, pr, running_avg, inv_count, e_loc[b], w_psi[b], w_lap[b]);
}

// ------------------------------------------------------------------------------------------------ reverse
// pre-activation adjoints of one conditioner from hbar2 (this lane's hidden unit); adds the W0 path to gU
template <int D, class T>
__device__ __forceinline__ void hidden_bwd(const NetWave& net, T hb2, float (*vec)[64], int lane, const Tape& tape, int n, T (&gU)[D]) {
    const T h2 = tget<T>(tape, n, Rows<D>::H2 + lane);
    const T A2 = hb2 * (1.0f - h2 * h2);
    tput(tape, n, Rows<D>::A2 + lane, A2);
    put(vec, lane, A2);
    const T hb1 = gemv<T, false, WF_GEMV_UNROLL_BWD>(net.W1b, vec, lane);
    const T h1 = tget<T>(tape, n, Rows<D>::H1 + lane);
    const T A1 = hb1 * (1.0f - h1 * h1);
    tput(tape, n, Rows<D>::A1 + lane, A1);
#pragma unroll
    for (int a = 0; a < D; ++a) gU[a] = gU[a] + wave_sum(A1 * net.W0[a * H + lane]);
}

// mode 0: sum_b w1[b] log_pdf_b;  mode 1: sum_b (w1[b] psi_b + w2[b] laplacian_b)
// Reverse of the gates G_d = prod_{i<d} x_i^3 of a gated head (G_0 = 1): gl[d][.] holds the adjoint of G_d collected from the heads
// (d = 1 .. D-1); adds the adjoints of the conditioner's inputs x to gX.  Uniform work, done by every lane.
template <int D, class T>
__device__ __forceinline__ void gate_chain_bwd(const T (&x)[D], const float (*gl)[T::NC], T (&gX)[D]) {
    T Gf[D];
    Gf[0] = cst<T>(1.0f);
#pragma unroll
    for (int d = 1; d < D; ++d) Gf[d] = Gf[d - 1] * (x[d - 1] * x[d - 1] * x[d - 1]);
    T acc = cst<T>(0.0f);
#pragma unroll
    for (int d = D - 1; d >= 1; --d) {
        float c[T::NC];
#pragma unroll
        for (int k = 0; k < T::NC; ++k) c[k] = gl[d][k];
        acc = acc + from_arr((T*)nullptr, c);                         // the whole adjoint of G_d
        const T x2 = x[d - 1] * x[d - 1];
        gX[d - 1] = gX[d - 1] + (acc * Gf[d - 1]) * (x2 * 3.0f);      // G_d = G_{d-1} x_{d-1}^3
        acc = acc * (x2 * x[d - 1]);
    }
}

template <int D, class T, int NBK = 1>
__global__ __launch_bounds__(kWB) __attribute__((amdgpu_waves_per_eu(kOccBwd<T>, kOccBwd<T>))) void k_wave_bwd(const ModelDev* __restrict__ mdp, int mode, const float* __restrict__ tabI, const float* __restrict__ tabP,
                                                   const float* __restrict__ fk_nat, int64_t B, const float* __restrict__ w1, const float* __restrict__ w2,
                                                   float* __restrict__ ws, const float* __restrict__ tails, float* __restrict__ zws) {
    // zws (gated models only, else null): [sample][net][pass][lane] adjoint of zero_params per head lane, summed over the samples afterwards
    __shared__ float lds[kWaves][2][T::NC][64];
    __shared__ float glds[kWaves][D][T::NC];   // gated heads: adjoints of the gates of one net, per dimension
    const ModelDev& md = *mdp;
    constexpr int DIRS = kDirs<T, D>;
    constexpr int P = NBK == 1 ? (D + 1) / 2 : D, W = 32 * NBK;
    using RW = Rows<D, NBK>;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float (*vec)[64] = lds[wv][0];
    float (*ov)[64] = lds[wv][1];
    const int dl = NBK == 1 ? lane >> 5 : 0, j = NBK == 1 ? (lane & 31) : lane;
    const bool imade = md.layer_kind == WF_LAYER_IMADE;
    const bool has_pnet = md.prior_kind == WF_PRIOR_WAVEFLOW || md.prior_kind == WF_PRIOR_MFLOW;
    const int n_nets = md.n_layers + (has_pnet ? 1 : 0);
    const int n_mesh = (imade && md.n_layers > 0) ? md.isp.n_mesh : md.psp.n_mesh;   // (a model may have no flow layer at all)
    const size_t plane = (size_t)n_mesh * W;
    const float* __restrict__ gI = fk_nat;
    const float* __restrict__ kP = fk_nat + 64;
    T* const tag = nullptr;
    const int64_t n_samples = B * DIRS;
    for (int64_t s = (int64_t)blockIdx.x * kWaves + wv; s < n_samples; s += (int64_t)gridDim.x * kWaves) {
        const int64_t b = s / DIRS;
        const int dir = (int)(s - b * DIRS);
        const Tape tape{ws + s * (int64_t)n_nets * T::NC * RW::N, RW::N};
        const float* tl = tails + s * (int64_t)Tail<D>::N * T::NC;
        T v[D], E;
        const T psi = psi_from_tail<D, T>(tl, md.constrained_mask, v, E);
        T gLD, gProd = cst<T>(0.0f), gOut = cst<T>(0.0f);
        if (mode == 1) {
            const T gPsi = adj_value(tag, dir == 0 ? w1[b] : 0.0f) + adj_second(tag, w2 ? w2[b] : 0.0f);
            gLD = (gPsi * psi) * 0.5f;
            gProd = gPsi * E;
        } else {
            gOut = adj_value(tag, w1[b]);
            gLD = gOut;
        }
        T gU[D];
#pragma unroll
        for (int d = 0; d < D; ++d) gU[d] = cst<T>(0.0f);
        if (has_pnet) {
            const int NP = md.n_layers;
            const NetWave& net = md.wnets[NP];
            const int nb = md.psp.nb;
            T cur[D];
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = tail_get<T>(tl, Tail<D>::U + d);
            T hb2 = cst<T>(0.0f);
            const bool gate_p = md.p_gate != 0;
            T grun = cst<T>(1.0f);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int d = NBK == 1 ? 2 * p + dl : p;
                const bool valid_d = d < D, valid = valid_d && j < nb;
                const T o = valid_d ? tget<T>(tape, NP, RW::O + d * W + j) : cst<T>(0.0f);   // head pre-activation, left by the forward sweep
                T gq = grun;
                if (gate_p) {   // the gate of this lane's dimension, as in the forward sweep
                    if constexpr (NBK == 1) {
                        const T c0 = cur[2 * p < D ? 2 * p : 0];
                        const T g_odd = grun * (c0 * c0 * c0);
                        gq = sel(dl == 0, grun, g_odd);
                        const T c1 = cur[2 * p + 1 < D ? 2 * p + 1 : 0];
                        grun = g_odd * (c1 * c1 * c1);
                    } else {
                        grun = grun * (cur[p] * cur[p] * cur[p]);
                    }
                }
                T zbar = cst<T>(0.0f), gqb = cst<T>(0.0f);
                bool inside;
                const T uc = clip01(NBK == 1 ? sel(dl == 0, cur[2 * p < D ? 2 * p : 0], cur[2 * p + 1 < D ? 2 * p + 1 : 0]) : cur[p], inside);
                const Lerp lp = make_lerp(uc.c0, n_mesh);
                float t[4];
                lerp4<W>(tabP, plane, lp, j, t);
                // adjoint of this half's prior factor v_d
                T gv_lo, gv_hi;
                {
                    T gvs[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int dd = NBK == 1 ? (2 * p + q < D ? 2 * p + q : 2 * p) : p;
                        const float sc = ((md.constrained_mask >> dd) & 1u) ? 0.70710678118654752f : 1.0f;
                        if (mode == 1) {
                            T others = cst<T>(1.0f);
#pragma unroll
                            for (int e = 0; e < D; ++e) {
                                const float se = ((md.constrained_mask >> e) & 1u) ? 0.70710678118654752f : 1.0f;
                                others = e == dd ? others * se : others * (v[e] * se);
                            }
                            gvs[q] = gProd * others;
                        } else if (md.prior_kind == WF_PRIOR_WAVEFLOW) {
                            gvs[q] = (gOut * rrcp((v[dd] * v[dd]) * (sc * sc) + 1e-7f)) * (v[dd] * (2.0f * sc * sc));
                        } else {
                            gvs[q] = gOut * rrcp(v[dd] + 1e-7f);
                        }
                    }
                    gv_lo = gvs[0];
                    gv_hi = gvs[1];
                }
                const T gv = sel(dl == 0, gv_lo, gv_hi);
                T go, d1;
                if (md.prior_kind == WF_PRIOR_WAVEFLOW) {
                    const PsiHead<T> hd = psi_head<T, NBK>(o, valid, valid_d, kP[j], md.ob_to_b_t, ov, lane, gate_p, gq, net.z[p * 64 + lane], md.p_cb);
                    const T ge = gv * lift(t, 0, uc);
                    const T dotE = rsum<NBK>(ge * hd.e);
                    d1 = rsum<NBK>(hd.e * lift(t, 1, uc));
                    const T gc = (ge - hd.e * dotE) * hd.rN2;
                    put(ov, lane, gc);
                    const T ga = gemv32_rows<T, NBK>(md.ob_to_b_t, ov, dl, j);          // abar_a = sum_j cbar_j ob_to_b[a][j]
                    const T dotA = rsum<NBK>(ga * hd.a);
                    T wbar = (ga - hd.a * dotA) * hd.rN1;
                    T sbar = cst<T>(0.0f);
                    if (md.p_cb) {   // c = a M + beta cb, beta = |S| rN1: betabar = <cbar, cb>; through rN1 = |w|^-1 into w, through |S| into every raw output
                        const T bbar = rsum<NBK>(gc * md.p_cb[NBK == 1 ? (lane & 31) : lane]);
                        wbar = wbar - hd.a * ((bbar * hd.sabs) * (hd.rN1 * hd.rN1));
                        sbar = (bbar * hd.rN1) * hd.sgn;
                    }
                    go = wbar * (valid ? kP[j] * hd.sgn : 0.0f);
                    if (md.p_cb && valid) go = go + sbar;
                    if (gate_p) {   // w = gq * o + z: `go` so far is wbar
                        zbar = go;
                        gqb = rsum<NBK>(go * o);
                        go = go * gq;
                    }
                } else {
                    const SigHead<T> hd = sigmoid_head<T, NBK>(o, valid, valid_d, kP[j], 0.0f, gate_p, gq, net.z[p * 64 + lane]);
                    d1 = rsum<NBK>(hd.c * lift(t, 1, uc));
                    go = sigmoid_head_bwd<T, NBK>(hd, gv * lift(t, 0, uc), valid, kP[j], gate_p, gq, &zbar);
                    if (gate_p) gqb = rsum<NBK>(zbar * hd.s);
                }
                if (gate_p) {
                    if (valid_d && j == 0) {
#pragma unroll
                        for (int k = 0; k < T::NC; ++k) glds[wv][d][k] = coef(gqb, k);
                    }
                    if (zws) zws[((s * n_nets + NP) * P + p) * 64 + lane] = coef(zbar, T::NC - 1);
                }
                const T gu = (valid_d && inside) ? gv * d1 : cst<T>(0.0f);
                if constexpr (NBK == 1) {
                    const T gu_o = xhalf(gu);
                    gU[2 * p] = gU[2 * p] + sel(dl == 0, gu, gu_o);
                    if (2 * p + 1 < D) gU[2 * p + 1] = gU[2 * p + 1] + sel(dl == 0, gu_o, gu);
                } else {
                    gU[p] = gU[p] + gu;
                }
                if (valid_d) tput(tape, NP, RW::O + d * W + j, go);
                put(ov, lane, go);
                hb2 = hb2 + gemv<T, false, WF_GEMV_UNROLL_BWD>(net.W2b + p * 1024, ov, lane);
            }
            hidden_bwd<D, T>(net, hb2, vec, lane, tape, NP, gU);
            if (gate_p) {   // the gates see the conditioner's input: the unclipped u
                __builtin_amdgcn_wave_barrier();
                gate_chain_bwd<D, T>(cur, glds[wv], gU);
                __builtin_amdgcn_wave_barrier();
            }
        } else if (md.prior_kind == WF_PRIOR_NORMAL) {
#pragma unroll
            for (int d = 0; d < D; ++d) gU[d] = gOut * ((tail_get<T>(tl, Tail<D>::U + d) + md.normal_offset) * -1.0f);
        }
        // ---- flow layers, last to first
        for (int l = md.n_layers - 1; l >= 0; --l) {
            const NetWave& net = md.wnets[l];
            T gY[D], U[D];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                gY[d] = gU[D - 1 - d];   // Reverse (bijections.py:337-340)
                U[d] = tget<T>(tape, l, RW::U + d);
            }
#pragma unroll
            for (int d = 0; d < D; ++d) gU[d] = cst<T>(0.0f);
            T hb2 = cst<T>(0.0f);
            const bool gate_i = md.i_gate != 0;
            T grun = cst<T>(1.0f);
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int d = NBK == 1 ? 2 * p + dl : p;
                const bool valid_d = d < D;
                const T o = valid_d ? tget<T>(tape, l, RW::O + d * W + j) : cst<T>(0.0f);
                const T u = NBK == 1 ? sel(dl == 0, U[2 * p < D ? 2 * p : 0], U[2 * p + 1 < D ? 2 * p + 1 : 0]) : U[p];
                T gq = grun;
                if (gate_i) {
                    if constexpr (NBK == 1) {
                        const T c0 = U[2 * p < D ? 2 * p : 0];
                        const T g_odd = grun * (c0 * c0 * c0);
                        gq = sel(dl == 0, grun, g_odd);
                        const T c1 = U[2 * p + 1 < D ? 2 * p + 1 : 0];
                        grun = g_odd * (c1 * c1 * c1);
                    } else {
                        grun = grun * (U[p] * U[p] * U[p]);
                    }
                }
                const T gy = NBK == 1 ? sel(dl == 0, gY[2 * p < D ? 2 * p : 0], gY[2 * p + 1 < D ? 2 * p + 1 : 0]) : gY[p];
                T go, gu;
                if (imade) {
                    const int nb = md.isp.nb;
                    const bool valid = valid_d && j < nb;
                    const SigHead<T> hd = sigmoid_head<T, NBK>(o, valid, valid_d, gI[j], md.i_reg, gate_i, gq, net.z[p * 64 + lane]);
                    const Lerp lp = make_lerp(u.c0, n_mesh);
                    float t[4];
                    lerp4<W>(tabI, plane, lp, j, t);
                    const T b0 = lift(t, 0, u), b1 = lift(t, 1, u);
                    const T dy = rsum<NBK>(hd.c * b1);
                    const T y2 = rsum<NBK>(hd.c * lift(t, 2, u));
                    const T gdy = gLD * rrcp(dy + 1e-7f);
                    gu = valid_d ? gy * dy + gdy * y2 : cst<T>(0.0f);
                    T zbar = cst<T>(0.0f);
                    go = sigmoid_head_bwd<T, NBK>(hd, gy * b0 + gdy * b1, valid, gI[j], gate_i, gq, &zbar);
                    if (gate_i) {
                        const T gqb = rsum<NBK>(zbar * hd.s);
                        if (valid_d && j == 0) {
#pragma unroll
                            for (int k = 0; k < T::NC; ++k) glds[wv][d][k] = coef(gqb, k);
                        }
                        if (zws) zws[((s * n_nets + l) * P + p) * 64 + lane] = coef(zbar, T::NC - 1);
                    }
                } else {
                    const T lw = from_lane(o, dl * 32), bias = from_lane(o, dl * 32 + 1);
                    const T e = rexp(cst<T>(0.0f) - lw);
                    const T y = (u - bias) * e;
                    gu = valid_d ? gy * e : cst<T>(0.0f);
                    go = cst<T>(0.0f);
                    if (valid_d && j == 0) go = cst<T>(0.0f) - (gy * y) - gLD;   // d y / d lw = -y,  d logdet / d lw = -1
                    if (valid_d && j == 1) go = cst<T>(0.0f) - (gy * e);          // d y / d bias = -e
                }
                if constexpr (NBK == 1) {
                    const T gu_o = xhalf(gu);
                    gU[2 * p] = gU[2 * p] + sel(dl == 0, gu, gu_o);
                    if (2 * p + 1 < D) gU[2 * p + 1] = gU[2 * p + 1] + sel(dl == 0, gu_o, gu);
                } else {
                    gU[p] = gU[p] + gu;
                }
                if (valid_d) tput(tape, l, RW::O + d * W + j, go);
                put(ov, lane, go);
                hb2 = hb2 + gemv<T, false, WF_GEMV_UNROLL_BWD>(net.W2b + p * 1024, ov, lane);
            }
            hidden_bwd<D, T>(net, hb2, vec, lane, tape, l, gU);
            if (gate_i) {
                __builtin_amdgcn_wave_barrier();
                gate_chain_bwd<D, T>(U, glds[wv], gU);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ inverse / sampler
// Serial.inverse_fun (bijections.py:462-463), IMADE.inverse_fun (made.py:85-100, helpers.binary_search), MADE.inverse_fun
// (made.py:29-37), the box reverse (made.py:139-154, 186-197) and the samplers of Waveflow (wavefunctions.py:74-107,
// bsplines_jax.py:144-171), MFlow (distributions.py:165-190, msplines_jax.py:129-154) and Flow (distributions.py:104-108), one
// wave per walker.  Every lane runs the same Philox stream (keyed by seed and walker, as in wf_kernels_scalar.hip), so the
// accept / bisection decisions are wave-uniform.
__device__ __forceinline__ float hmax(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
// value held by the lanes of half `h`, in every lane
__device__ __forceinline__ float of_half(float v, int h, int dl) {
    const float o = swap32_other(v);
    return dl == h ? v : o;
}
// order-0 lerp of basis row j (tab: [orders][n_mesh][W])
template <int W = NBP>
__device__ __forceinline__ float lerp0(const float* __restrict__ tab, const Lerp& L, int j) {
    const float yl = tab[(size_t)L.il * W + j], yr = tab[(size_t)L.ir * W + j];
    return yl + ((yr - yl) * L.n) * L.dx;
}

// x with sum_j c_j I_j(x) = y as helpers.binary_search returns it (utils/helpers.py:150-166: K halvings of [0, 1], the lower
// end of the last bracket), without the K dependent table reads of the halving loop: the spline is monotone and piecewise
// linear on the mesh, so (1) the mesh interval that contains the root is found by two 64-way searches (every lane evaluates
// the spline at one mesh point), (2) the root follows from the line through its two mesh values, (3) it is rounded down to the
// halving grid 2^-K.  (A halving loop in fp32 takes a wrong turn when |f(mid)| is below its rounding noise, so does this; both
// stay within one grid step of the exact-arithmetic answer.)  c: this lane's weight (lanes of half `hd` hold the dimension).
// LDS copy of every 32nd row of the order-0 I-spline table (the first round of the mesh search): 64 rows, padded so that the lanes' 16-byte reads of
// their own rows spread over the banks.  The sampler is bound by L2 bandwidth on these rows (PMC: 1 350 L2 requests per walker, 10.7 TB/s).
template <int NBK> constexpr int kCoarseStride = 32 * NBK + 4;
template <int NBK>
__device__ __forceinline__ void stage_coarse_rows(float* coarse, const float* __restrict__ tab0, int n_mesh) {
    constexpr int W = 32 * NBK;
    for (int i = threadIdx.x; i < 64 * W; i += blockDim.x) {
        const int row = i / W, col = i % W;
        coarse[row * kCoarseStride<NBK> + col] = tab0[(size_t)min(row * 32, n_mesh - 1) * W + col];
    }
    __syncthreads();
}
template <int NBK = 1>
__device__ __forceinline__ float ispline_inverse(const float* __restrict__ tab0 /* [n_mesh][32 NBK], order 0 */, int n_mesh, int nb, float c,
                                                 float y, float tol, int hd, float (*ov)[64], int lane, const float* coarse) {
    constexpr int W = 32 * NBK;
    put(ov, lane, R1{c});
    const float* __restrict__ cw = &ov[0][NBK == 1 ? hd * 32 : 0];
    auto spline_row = [&](const float4_t* __restrict__ row) {   // sum_j c_j row[j], j ascending
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < W / 4; ++q) {
            const float4_t t = row[q], w = *reinterpret_cast<const float4_t*>(cw + 4 * q);
            acc = __builtin_fmaf(w.x, t.x, acc);
            acc = __builtin_fmaf(w.y, t.y, acc);
            acc = __builtin_fmaf(w.z, t.z, acc);
            acc = __builtin_fmaf(w.w, t.w, acc);
        }
        return acc;
    };
    auto spline_at = [&](int m) { return spline_row(reinterpret_cast<const float4_t*>(tab0 + (size_t)m * W)); };
    // round 1: mesh points 0, 32, 64, ... (rows min(32 lane, last): the workgroup's LDS copy, kCoarseStride floats apart, when the caller staged one);
    // round 2: the 32 points inside the interval found (weights beyond nb are zero)
    const int last = n_mesh - 1;
    int m1 = min(lane * 32, last);
    float g1 = coarse ? spline_row(reinterpret_cast<const float4_t*>(coarse + lane * kCoarseStride<NBK>)) : spline_at(m1);
    // (first lane whose mesh value exceeds y: robust against a rounding-level non-monotonicity of the fp32 sums)
    int cnt = __ffsll((long long)~__ballot(g1 <= y && lane * 32 <= last)) - 1;
    if (cnt < 0) cnt = 64;
    const int base = max(cnt - 1, 0) * 32;
    const int m2 = min(base + (lane & 31), last);
    const float g2 = spline_at(m2);
    cnt = __ffsll((long long)~__ballot(g2 <= y && lane < 32 && base + lane <= last)) - 1;
    if (cnt < 0) cnt = 64;
    cnt = min(cnt, 32);
    const int k = max(cnt - 1, 0);                      // lane k holds the largest mesh point with spline <= y
    const int m = min(base + k, last);
    const float yl = __shfl(g2, k);
    const float yr = m < last ? (k < 31 ? __shfl(g2, k + 1) : spline_at(m + 1)) : yl;
    __builtin_amdgcn_wave_barrier();
    const float n = (float)last;
    float xs = (float)m / n;
    if (yr > yl) xs = xs + (y - yl) / ((yr - yl) * n);
    // the halving grid
    int K = 0;
    float w = 1.0f;
    while (K < 64 && w * 0.5f > tol * 0.5f) { w *= 0.5f; ++K; }
    const float scale = ldexpf(1.0f, K);
    float q = floorf(xs * scale);
    q = fminf(fmaxf(q, 0.0f), scale - 1.0f);
    // The halving loop ends at the largest grid point whose spline value, evaluated by the table lerp, does not exceed y.
    // The line through the mesh values locates it to within a grid step; the lerp itself decides between the neighbours:
    // lanes 0..31 evaluate it at q, lanes 32..63 at q + 1 (one more table read, both in flight together).
    {
        const int side = lane >> 5, jj = lane & 31;
        const float xq = fminf(q + (float)side, scale - 1.0f) / scale;
        const Lerp L = make_lerp(xq, n_mesh);
        float term = cw[jj] * lerp0<W>(tab0, L, jj);
        if constexpr (NBK == 2) term += cw[jj + 32] * lerp0<W>(tab0, L, jj + 32);
        const float f = hsum(term) - y;
        const float f_other = swap32_other(f);
        const float f_lo = side == 0 ? f : f_other, f_hi = side == 0 ? f_other : f;
        if (f_hi <= 0.0f && q + 1.0f <= scale - 1.0f) q = q + 1.0f;
        else if (f_lo > 0.0f && q >= 1.0f) q = q - 1.0f;
    }
    return q / scale;
}

template <int D, int NBK = 1>
__device__ __forceinline__ void wave_serial_inverse(const ModelDev& md, const float* __restrict__ tabI, const float* __restrict__ gI, float (&cur)[D],
                                                    float (*vec)[64], float (*ov)[64], int lane, int exact, const float* coarse) {
    const int dl = NBK == 1 ? lane >> 5 : 0, j = NBK == 1 ? (lane & 31) : lane;
    const Tape no_tape{nullptr, 0};
    float nxt[D];
    for (int l = md.n_layers - 1; l >= 0; --l) {
        const NetWave& net = md.wnets[l];
#pragma unroll
        for (int d = 0; d < D; ++d) nxt[d] = cur[D - 1 - d];   // Reverse.inverse_fun
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = 0.0f;
        if (md.layer_kind == WF_LAYER_IMADE) {
            const int nb = md.isp.nb, n_mesh = md.isp.n_mesh;
            const float tol = md.reverse_tol;
            const bool gate_i = md.i_gate != 0;
            float gq = 1.0f;   // gated heads: prod_{i<d} x_i^3 of the vector the conditioner sees (made.py:88: the values being inverted, or the prefix)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int p = NBK == 1 ? d >> 1 : d, hd = NBK == 1 ? (d & 1) : 0;
                if (gate_i && d > 0) {
                    const float xp = exact ? cur[d - 1] : nxt[d - 1];
                    gq = gq * (xp * xp * xp);
                }
                R1 o;
                if (d == 0) {
                    // output dimension 0 sees no input (MADE mask, model_factory.py:15-18): its head is the bias alone
                    o = R1{net.b2[lane]};
                } else {
                    if (d == 1 || exact) {   // reference mode: one conditioner pass on the values being inverted (made.py:88)
                        R1 xin[D];
#pragma unroll
                        for (int a = 0; a < D; ++a) xin[a] = R1{exact ? cur[a] : nxt[a]};
                        hidden_fwd<D, R1, true>(net, xin, vec, lane, no_tape, 0, false);
                    }
                    o = gemv<R1, true>(net.W2f + p * 1024, vec, lane) + net.b2[p * 64 + lane];
                }
                const bool valid_d = (NBK == 1 ? 2 * p + dl : p) < D, valid = valid_d && j < nb;
                const SigHead<R1> hdw = sigmoid_head<R1, NBK>(o, valid, valid_d, gI[j], md.i_reg, gate_i, R1{gq}, net.z[p * 64 + lane]);
                cur[d] = ispline_inverse<NBK>(tabI, n_mesh, nb, hdw.c.c0, nxt[d], tol, hd, ov, lane, coarse);
            }
        } else {
#pragma unroll
            for (int c = 0; c < D; ++c) {
                const int p = NBK == 1 ? c >> 1 : c, hd = NBK == 1 ? (c & 1) : 0;
                R1 o;
                if (c == 0) {
                    o = R1{net.b2[lane]};
                } else {
                    R1 xin[D];
#pragma unroll
                    for (int a = 0; a < D; ++a) xin[a] = R1{cur[a]};
                    hidden_fwd<D, R1, true>(net, xin, vec, lane, no_tape, 0, false);
                    o = gemv<R1, true>(net.W2f + p * 1024, vec, lane) + net.b2[p * 64 + lane];
                }
                const float lw = __shfl(o.c0, hd * 32), bias = __shfl(o.c0, hd * 32 + 1);
                cur[c] = nxt[c] * expf(lw) + bias;
            }
        }
    }
    if (md.box_kind != WF_BOX_NONE) {
        scalar::box_reverse<D>(md, cur, nxt);
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = nxt[d];
    }
}

// seed_mode 0: invert the latent points ug;  1: draw the latent points from the prior first (and report them)
template <int D, int NBK = 1>
__global__ __launch_bounds__(kWB) __attribute__((amdgpu_waves_per_eu(WF_OCC_SAMPLE, WF_OCC_SAMPLE))) void k_wave_sample(
    const ModelDev* __restrict__ mdp, const float* __restrict__ tabI, const float* __restrict__ tabP, const float* __restrict__ fk_nat, int draw,
    unsigned long long seed, const float* __restrict__ ug, int64_t B, float* __restrict__ xg, float* __restrict__ latent, int exact,
    const unsigned long long* __restrict__ seed_offset_dev) {
    __shared__ float lds[kWaves][2][1][64];
    __shared__ __attribute__((aligned(16))) float coarse_s[64 * kCoarseStride<NBK>];
    if (seed_offset_dev) seed += *seed_offset_dev * 0x9E3779B97F4A7C15ull;   // a device counter advances the stream (captured steps)
    const ModelDev& md = *mdp;
    const bool use_coarse = md.layer_kind == WF_LAYER_IMADE && md.n_layers > 0;
    if (use_coarse) stage_coarse_rows<NBK>(coarse_s, tabI, md.isp.n_mesh);
    const float* coarse = use_coarse ? coarse_s : nullptr;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float (*vec)[64] = lds[wv][0];
    float (*ov)[64] = lds[wv][1];
    constexpr int W = 32 * NBK;
    const int dl = NBK == 1 ? lane >> 5 : 0, j = NBK == 1 ? (lane & 31) : lane;
    const float* __restrict__ gI = fk_nat;
    const float* __restrict__ kP = fk_nat + 64;
    const Tape no_tape{nullptr, 0};
    for (int64_t b = (int64_t)blockIdx.x * kWaves + wv; b < B; b += (int64_t)gridDim.x * kWaves) {
        float cur[D];
        if (!draw) {
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = ug[b * D + d];
        } else {
            scalar::Philox rng(seed, (unsigned long long)b);
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = 0.0f;
            if (md.prior_kind == WF_PRIOR_UNIFORM) {
#pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = rng.uniform();
            } else if (md.prior_kind == WF_PRIOR_NORMAL) {
#pragma unroll
                for (int d = 0; d < D; ++d) {   // Box-Muller
                    const float u1 = fmaxf(rng.uniform(), 5.9604645e-8f), u2 = rng.uniform();
                    cur[d] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
                }
            } else {
                const NetWave& net = md.wnets[md.n_layers];
                const int nb = md.psp.nb, n_mesh = md.psp.n_mesh;
                const bool wavefn = md.prior_kind == WF_PRIOR_WAVEFLOW;
                float gcol = 1.0f;
#pragma unroll
                for (int col = 0; col < D; ++col) {
                    const int p = NBK == 1 ? col >> 1 : col, hd = NBK == 1 ? (col & 1) : 0;
                    if (md.p_gate != 0 && col > 0) gcol = gcol * (cur[col - 1] * cur[col - 1] * cur[col - 1]);   // gated head: prod of the drawn columns' cubes
                    R1 o;
                    if (col == 0) {
                        o = R1{net.b2[lane]};   // column 0 is conditioned on nothing: bias only
                    } else {
                        R1 xin[D];
#pragma unroll
                        for (int a = 0; a < D; ++a) xin[a] = R1{cur[a]};   // conditioner on the columns drawn so far, zeros elsewhere
                        hidden_fwd<D, R1, true>(net, xin, vec, lane, no_tape, 0, false);
                        o = gemv<R1, true>(net.W2f + p * 1024, vec, lane) + net.b2[p * 64 + lane];
                    }
                    const bool valid_d = (NBK == 1 ? 2 * p + dl : p) < D, valid = valid_d && j < nb;
                    float cj, ymax;
                    // maximum over the rows of the column, in every lane
                    auto col_max = [&](float v) {
                        const float h = hmax(v);
                        if constexpr (NBK == 1) return of_half(h, hd, dl);
                        else return fmaxf(h, swap32_other(h));
                    };
                    if (wavefn) {
                        // sample_fun (bsplines_jax.py:144-171): obw = normalised(w @ ob_to_b); ymax = max((obw @ b_to_ob)^2)
                        const PsiHead<R1> hdw = psi_head<R1, NBK>(o, valid, valid_d, kP[j], md.ob_to_b_t, ov, lane, md.p_gate != 0, R1{gcol}, net.z[p * 64 + lane], md.p_cb);
                        cj = hdw.e.c0;
                        put(ov, lane, hdw.e);
                        const float q = gemv32_cols<R1, NBK>(md.b_to_ob, ov, dl, j).c0;
                        ymax = col_max(valid ? q * q : 0.0f);
                    } else {
                        const SigHead<R1> hdw = sigmoid_head<R1, NBK>(o, valid, valid_d, kP[j], 0.0f, md.p_gate != 0, R1{gcol}, net.z[p * 64 + lane]);
                        cj = hdw.c.c0;
                        ymax = col_max(valid ? cj : 0.0f) * (float)(nb + md.psp.degree);   // msplines_jax.py:147-150
                    }
                    // Rejection sampling, 32 proposals per round (lanes 0..31): lane t holds proposal number 32 * round + t of the sequence, the
                    // first accepted one in sequence order is taken (same distribution as proposing one by one; the acceptance
                    // rate of the reference's bound is 4-6 %, i.e. ~20 sequential table reads per column otherwise).  Each lane
                    // evaluates the whole spline at its own point: the column's coefficients come from LDS, two table rows per lane.
                    // Bounded (3126 rounds ~ 1e5 proposals): a pathological density cannot hang the GPU; a walker that exhausts
                    // the bound comes out as NaN (its later columns, its x and every batch sum over it), not as a plausible 0.5.
                    put(ov, lane, R1{cj});
                    const float* __restrict__ cw = &ov[0][NBK == 1 ? hd * 32 : 0];
                    float xs = __builtin_nanf("");
                    for (int round = 0; round < 3126; ++round) {
                        // (32 proposals per round on lanes 0..31: the first accepted one in sequence order does not depend on the round size, the
                        // expected number of table rows read does -- 40 proposals instead of 67 at the reference bound's 5 % acceptance)
                        scalar::Philox prop(seed, (unsigned long long)b);
                        prop.c0 = (unsigned)(round * 32 + (lane & 31));
                        prop.c1 = (unsigned)(col + 1);          // the shared stream of this walker uses c1 == 0
                        const float xc = prop.uniform(), yc = prop.uniform() * ymax;
                        float v = 0.0f;
                        if (lane < 32) {
                            const Lerp L = make_lerp(xc, n_mesh);
                            const float4_t* __restrict__ rl = reinterpret_cast<const float4_t*>(tabP + (size_t)L.il * W);
                            const float4_t* __restrict__ rr = reinterpret_cast<const float4_t*>(tabP + (size_t)L.ir * W);
#pragma unroll
                            for (int q = 0; q < W / 4; ++q) {
                                const float4_t a4 = rl[q], b4 = rr[q], w4 = *reinterpret_cast<const float4_t*>(cw + 4 * q);
                                v = __builtin_fmaf(w4.x, a4.x + ((b4.x - a4.x) * L.n) * L.dx, v);
                                v = __builtin_fmaf(w4.y, a4.y + ((b4.y - a4.y) * L.n) * L.dx, v);
                                v = __builtin_fmaf(w4.z, a4.z + ((b4.z - a4.z) * L.n) * L.dx, v);
                                v = __builtin_fmaf(w4.w, a4.w + ((b4.w - a4.w) * L.n) * L.dx, v);
                            }
                            if (wavefn) v = v * v;
                        }
                        const unsigned long long hit = __ballot(lane < 32 && yc < v);
                        if (hit) {
                            xs = __shfl(xc, __ffsll((long long)hit) - 1);
                            break;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    cur[col] = xs;
                }
            }
            if (latent && lane < D) {
#pragma unroll
                for (int d = 0; d < D; ++d)
                    if (lane == d) latent[b * D + d] = cur[d];
            }
        }
        wave_serial_inverse<D, NBK>(md, tabI, gI, cur, vec, ov, lane, exact, coarse);
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (lane == d) xg[b * D + d] = cur[d];
    }
}

int finish() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

unsigned wave_grid(int64_t n_samples) {
    int64_t blocks = (n_samples + kWaves - 1) / kWaves;
    const int64_t cap = 256 * 24;   // 24 workgroups of 4 waves per CU: ~8 rounds of the resident set, so that the dispatcher evens out the
                                     // SIMDs' oldest-wave-first progress (H psi at 2^17 walkers: +4 % over a cap of 8 per CU)
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

template <int D, class T, int NBK = 1>
int run_fwd(const ModelDev* md_dev, const float* tabI4, const float* tabP4, const float* fk_nat, const float* x, int64_t B, float* ws, float* tails,
            int taped, hipStream_t s) {
    const int64_t n_samples = B * kDirs<T, D>;
    hipLaunchKernelGGL((k_wave_fwd<D, T, NBK>), dim3(wave_grid(n_samples)), dim3(kWB), 0, s, md_dev, tabI4, tabP4, fk_nat, x, B, ws, tails, taped);
    return finish();
}
template <int D, class T, int NBK = 1>
int run_bwd(const ModelDev* md_dev, int mode, const float* tabI4, const float* tabP4, const float* fk_nat, int64_t B, const float* w1, const float* w2,
            float* ws, const float* tails, float* zws, hipStream_t s) {
    const int64_t n_samples = B * kDirs<T, D>;
    hipLaunchKernelGGL((k_wave_bwd<D, T, NBK>), dim3(wave_grid(n_samples)), dim3(kWB), 0, s, md_dev, mode, tabI4, tabP4, fk_nat, B, w1, w2, ws, tails, zws);
    return finish();
}

}  // namespace

int64_t wave_tail_floats(int D, int ring_kind) { return (int64_t)(2 * D + 1) * ring_coefs(D, ring_kind) * ring_samples(D, ring_kind); }   // per walker

#define WF_WAVE_DISPATCH(CALL)                    \
    switch (md.D) {                               \
        case 2: return CALL(2);                   \
        case 3: return CALL(3);                   \
        case 4: return CALL(4);                   \
        case 5: return CALL(5);                   \
        case 6: return CALL(6);                   \
        case 7: return CALL(7);                   \
        case 8: return CALL(8);                   \
        default: return WF_ERR_UNSUPPORTED;       \
    }

// 33..64 bases per dimension (nbp == 64): the one-dimension-per-pass kernels, built for D <= 4 (as the 64-row MFMA kernel)
#define WF_WAVE_DISPATCH64(CALL)                  \
    switch (md.D) {                               \
        case 2: return CALL(2);                   \
        case 3: return CALL(3);                   \
        case 4: return CALL(4);                   \
        default: return WF_ERR_UNSUPPORTED;       \
    }

int launch_wave_fwd(const ModelDev& md, const ModelDev* md_dev, int ring_kind, const float* tabI4, const float* tabP4, const float* fk_nat,
                    const float* x, int64_t B, float* ws, float* tails, int taped, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (ring_kind == 3 && taped) return WF_ERR_INVALID;   // whole-walker samples for every D: the untaped energy sweep only
#define CALLK(DD, K) (ring_kind == 3 ? run_fwd<DD, RF<DD>, K>(md_dev, tabI4, tabP4, fk_nat, x, B, ws, tails, 0, s)               \
                      : ring_kind == 2 ? run_fwd<DD, RF<rf_block(DD)>, K>(md_dev, tabI4, tabP4, fk_nat, x, B, ws, tails, taped, s)  \
                      : ring_kind == 1 ? run_fwd<DD, R3, K>(md_dev, tabI4, tabP4, fk_nat, x, B, ws, tails, taped, s) \
                                       : run_fwd<DD, R1, K>(md_dev, tabI4, tabP4, fk_nat, x, B, ws, tails, taped, s))
#define CALL(DD) CALLK(DD, 1)
#define CALL64(DD) CALLK(DD, 2)
    if (md.nbp == 64) { WF_WAVE_DISPATCH64(CALL64) }
    WF_WAVE_DISPATCH(CALL)
#undef CALL
#undef CALL64
#undef CALLK
}

int launch_wave_bwd(const ModelDev& md, const ModelDev* md_dev, int mode, int ring_kind, const float* tabI4, const float* tabP4,
                    const float* fk_nat, int64_t B, const float* w1, const float* w2, float* ws, const float* tails, float* zws, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALLK(DD, K) (ring_kind == 2 ? run_bwd<DD, RF<rf_block(DD)>, K>(md_dev, mode, tabI4, tabP4, fk_nat, B, w1, w2, ws, tails, zws, s)  \
                      : ring_kind == 1 ? run_bwd<DD, R3, K>(md_dev, mode, tabI4, tabP4, fk_nat, B, w1, w2, ws, tails, zws, s) \
                                       : run_bwd<DD, R1, K>(md_dev, mode, tabI4, tabP4, fk_nat, B, w1, w2, ws, tails, zws, s))
#define CALL(DD) CALLK(DD, 1)
#define CALL64(DD) CALLK(DD, 2)
    if (md.nbp == 64) { WF_WAVE_DISPATCH64(CALL64) }
    WF_WAVE_DISPATCH(CALL)
#undef CALL
#undef CALL64
#undef CALLK
}

// draw == 0: x = inverse(u);  draw == 1: latent ~ prior, x = inverse(latent)
int launch_wave_sample(const ModelDev& md, const ModelDev* md_dev, const float* tabI4, const float* tabP4, const float* fk_nat, int draw,
                       unsigned long long seed, const float* u, int64_t B, float* x, float* latent, int exact,
                       const unsigned long long* seed_offset_dev, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALLK(DD, K)                                                                                                                      \
    hipLaunchKernelGGL((k_wave_sample<DD, K>), dim3(wave_grid(B)), dim3(kWB), 0, s, md_dev, tabI4, tabP4, fk_nat, draw, seed, u, B, x, latent, \
                       exact, seed_offset_dev);                                                                                          \
    break
    if (md.nbp == 64) {
        switch (md.D) {
            case 2: CALLK(2, 2);
            case 3: CALLK(3, 2);
            case 4: CALLK(4, 2);
            default: return WF_ERR_UNSUPPORTED;
        }
        return finish();
    }
    switch (md.D) {
        case 2: CALLK(2, 1);
        case 3: CALLK(3, 1);
        case 4: CALLK(4, 1);
        case 5: CALLK(5, 1);
        case 6: CALLK(6, 1);
        case 7: CALLK(7, 1);
        case 8: CALLK(8, 1);
        default: return WF_ERR_UNSUPPORTED;
    }
#undef CALLK
    return finish();
}

#define WF_RING2_DISPATCH(KERNEL, ...)                                                             \
    switch (D) {                                                                                   \
        case 2: WF_RING2_CASE(KERNEL, 2, __VA_ARGS__);                                             \
        case 3: WF_RING2_CASE(KERNEL, 3, __VA_ARGS__);                                             \
        case 4: WF_RING2_CASE(KERNEL, 4, __VA_ARGS__);                                             \
        case 5: WF_RING2_CASE(KERNEL, 5, __VA_ARGS__);                                             \
        case 6: WF_RING2_CASE(KERNEL, 6, __VA_ARGS__);                                             \
        case 7: WF_RING2_CASE(KERNEL, 7, __VA_ARGS__);                                             \
        case 8: WF_RING2_CASE(KERNEL, 8, __VA_ARGS__);                                             \
        default: return WF_ERR_UNSUPPORTED;                                                        \
    }
#define WF_RING2_CASE(KERNEL, DD, ...)                                                             \
    if (ring_kind == 3) hipLaunchKernelGGL((KERNEL<DD, RF<DD>>), grid, block, 0, s, __VA_ARGS__);                     \
    else if (ring_kind == 2) hipLaunchKernelGGL((KERNEL<DD, RF<rf_block(DD)>>), grid, block, 0, s, __VA_ARGS__);     \
    else hipLaunchKernelGGL((KERNEL<DD, R3>), grid, block, 0, s, __VA_ARGS__);                                       \
    break

int launch_energy_out(int D, int ring_kind, const float* tails, const float* x, int64_t B, unsigned constrained_mask, const Protons& pr, float* hpsi,
                      float* psi, float* lap, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    if (ring_kind < 1 || ring_kind > 3) return WF_ERR_INVALID;
    WF_RING2_DISPATCH(k_energy_out, tails, x, B, constrained_mask, pr, hpsi, psi, lap)
    return finish();
}

int launch_energy_seeds(int D, int ring_kind, const float* tails, const float* x, int64_t B, unsigned constrained_mask, const Protons& pr, float running_avg,
                        const float* running_avg_dev, float inv_count, float* e_loc, float* w_psi, float* w_lap, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    if (ring_kind < 1 || ring_kind > 2) return WF_ERR_INVALID;
    WF_RING2_DISPATCH(k_energy_seeds, tails, x, B, constrained_mask, pr, running_avg, running_avg_dev, inv_count, e_loc, w_psi, w_lap)
    return finish();
}
#undef WF_RING2_CASE
#undef WF_RING2_DISPATCH

int launch_tail_out(const ModelDev& md, int mode, const float* tails, int64_t B, float* out, float* u, void* stream, float* w_out, float w_value) {
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
#define CALL(DD) hipLaunchKernelGGL(k_tail_out<DD>, grid, block, 0, s, tails, B, mode, md.prior_kind, md.constrained_mask, md.normal_offset, out, u, \
                                    w_out, w_value); break
    switch (md.D) {
        case 2: CALL(2);
        case 3: CALL(3);
        case 4: CALL(4);
        case 5: CALL(5);
        case 6: CALL(6);
        case 7: CALL(7);
        case 8: CALL(8);
        default: return WF_ERR_UNSUPPORTED;
    }
#undef CALL
    return finish();
}

int launch_wave_eval(const ModelDev& md, const ModelDev* md_dev, const float* tabI4, const float* tabP4, const float* fk_nat, int mode,
                     const float* x, int64_t B, float* out, float* u, float* tail_ws, void* stream) {
    int rc = launch_wave_fwd(md, md_dev, 0, tabI4, tabP4, fk_nat, x, B, nullptr, tail_ws, 0, stream);
    if (rc) return rc;
    return launch_tail_out(md, mode, tail_ws, B, out, u, stream);
}

// H psi, psi, laplacian of B walkers: forward without a tape, then the per-walker combination.  The forward sweep carries
// (value, gradient, Laplacian / 2) per walker (RF: D + 2 channels, one pass) -- measured 1.6-1.8x faster than D passes in R3
// (3 channels each) for every D = 2..8 (scratch/energy_ab.py).
int launch_wave_energy(const ModelDev& md, const ModelDev* md_dev, const float* tabI4, const float* tabP4, const float* fk_nat, const float* x,
                       int64_t B, const Protons& pr, float* hpsi, float* psi, float* lap, float* tail_ws, void* stream) {
    const bool force_r3 = getenv("WF_ENERGY_R3") != nullptr;   // A/B switch, read per call (tests compare the two sweeps)
    const int kind = force_r3 ? 1 : 3;
    int rc = launch_wave_fwd(md, md_dev, kind, tabI4, tabP4, fk_nat, x, B, nullptr, tail_ws, 0, stream);
    if (rc) return rc;
    return launch_energy_out(md.D, kind, tail_ws, x, B, md.constrained_mask, pr, hpsi, psi, lap, stream);
}

}  // namespace wf
