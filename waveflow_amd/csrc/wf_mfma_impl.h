// wf_mfma_impl.h -- the MFMA throughput kernel as templates; instantiated per shape in wf_mfma_inst_*.hip so that the
// shapes compile in parallel.  See wf_kernels_mfma.hip for the design notes.
//
// Build rule for every translation unit that includes this file: -fno-slp-vectorize (waveflow_amd/build.py).  With hipcc's SLP
// vectorizer on, adjacent scalar f32 operations become packed-FP32 VALU instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32);
// with those in the kernel and VALU work scheduled into the f16 MFMA chains, a few 16-lane groups per launch came out wrong at
// >= 3 waves per SIMD (elimination study, scratch/ubench2/).  Without packed-FP32 code the kernel is bit-reproducible
// with no scheduling fences at all, so the compiler is free to fill the MFMA shadow.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "wf_internal.h"

// the loops over a net's output dimensions: unrolled (kDimUnroll<D> = D).  -DWF_D8_ROLLED (experiment, round 4) rolls them for the long chains (D > 4):
// no spilled registers at 8 waves instead of 10, 75 / 207 at 12 / 16 waves instead of 172 / 346 -- and 2 - 5 % SLOWER at 8 waves (D = 8, 2^18 walkers:
// 0.319 against 0.314 ms), 12 and 16 waves slower still (0.365 / 0.472 ms): profiles/r04_c4_rolled_loops_and_waves.txt.  Same bits either way.
template <int D>
#ifdef WF_D8_ROLLED
constexpr int kDimUnroll = D > 4 ? 1 : D;
#else
constexpr int kDimUnroll = D;
#endif

// WF_PIN(): keeps the hand-written order of [MFMA K step | activation of another block] units (hidden_layers, out_block_first): the
// scheduler is free inside a unit, not across units.
#ifdef WF_NO_PIN
#define WF_PIN()
#else
#define WF_PIN() __builtin_amdgcn_sched_barrier(0)
#endif

namespace wf {
namespace mfma {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using i32x2 = __attribute__((ext_vector_type(2))) int;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

// Experiment builds (scratch/r04_parity_variants.py; timing is irrelevant there): one class of hardware approximations at a time replaced by fp64
// arithmetic, to attribute the kernel's deviation from the reference -- WF_X_ACT: hidden activations, WF_X_SIG: sigmoid heads, WF_X_LOG: logarithms,
// WF_X_RCP: reciprocals / reciprocal square roots of the heads.
__device__ __forceinline__ float x_rinv(float xs) {   // 1 / (2^xs + 1)
#if defined(WF_X_ACT) || defined(WF_X_SIG)
    return (float)(1.0 / (exp2((double)xs) + 1.0));
#else
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.0f);
#endif
}
__device__ __forceinline__ float x_rcp(float v) {
#ifdef WF_X_RCP
    return (float)(1.0 / (double)v);
#else
    return __builtin_amdgcn_rcpf(v);
#endif
}
__device__ __forceinline__ float x_rsq(float v) {
#ifdef WF_X_RCP
    return (float)(1.0 / sqrt((double)v));
#else
    return __builtin_amdgcn_rsqf(v);
#endif
}
__device__ __forceinline__ float act_tanh(float xs) {  // xs = 2*log2(e)*x (scale folded into the weights)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.0f), 1.0f);
}
__device__ __forceinline__ float act_sigmoid(float xs) {  // xs = -log2(e)*x
#ifdef WF_X_SIG
    return x_rinv(xs);
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xs));
#endif
}
__device__ __forceinline__ float fast_log(float x) {
#ifdef WF_X_LOG
    return (float)log((double)x);
#else
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;
#endif
}

// sum of the two lane halves (lane l and l^32), result in every lane
__device__ __forceinline__ float xhalf_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

#ifdef WF_ABL_TAB   // ablation build (timing only): no table loads
__device__ __forceinline__ f32x16 load16g(const float* p) {
    const float c = (float)(size_t)p * 1e-20f;
    return f32x16{c, c, c, c, c, c, c, c, c, c, c, c, c, c, c, c};
}
#else
#define load16g load16
#endif
#ifdef WF_ABL_TAB
__device__ __forceinline__ f32x4 load4g(const float* p) {
    const float c = (float)(size_t)p * 1e-20f;
    return f32x4{c, c, c, c};
}
#else
__device__ __forceinline__ f32x4 load4g(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
#endif
// a wave-uniform float into a scalar register (the builtin is folded away when the compiler knows the value to be uniform, and the value stays in a vector register)
__device__ __forceinline__ float uniform_f(float v) {
    float r;
    asm("v_readfirstlane_b32 %0, %1" : "=s"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float xhalf_max(float v) {
    const unsigned u = __float_as_uint(v);
    const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}

__device__ __forceinline__ f32x16 load16(const float* p) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

struct Lerp {
    int il, ir, xl, xr;
    float t;  // (x - x_l/n) * n
};

__device__ __forceinline__ int wrap_clamp(int i, int n) {
    if (i < 0) i += n;
    return min(max(i, 0), n - 1);
}

// x_l / n of isplines_jax.py:50 without the division sequence: q = x_l * (1/n), one residual correction; the result equals the
// correctly rounded quotient for every x_l the index arithmetic can produce (mfma_div_ok in wf_kernels_mfma.hip checks all of them
// on the host when the model is created; otherwise the model keeps the IEEE division: mm.exact_div).
__device__ __forceinline__ float div_by_n(float xl, float n, float rn, int exact_div) {
    if (exact_div) return xl / n;
    const float q = xl * rn;
    const float r = __builtin_fmaf(-q, n, xl);
    return __builtin_fmaf(r, rn, q);
}

__device__ __forceinline__ Lerp make_lerp(float x, int n_mesh, float rn, int exact_div) {
    Lerp L;
    const int n_points = n_mesh - 1;
    const float xs = x * (float)n_points;
    const float fl = floorf(xs);
    L.xl = (int)fl;
    L.xr = (int)ceilf(xs);
    L.il = wrap_clamp(L.xl, n_mesh);
    L.ir = wrap_clamp(L.xr, n_mesh);
    const float dx = x - div_by_n(fl, (float)n_points, rn, exact_div);
    // y = y_l + (y_r - y_l) * t with both ends from one row is y_l whatever t: t = 0 then, and the right end need not be read (fetch_block)
    L.t = L.il == L.ir ? 0.0f : dx * (float)n_points;
    return L;
}

// Output dimension 0 of every net has an empty mask (model_factory.py:15-18): its spline weights do not depend on the
// walker, and a lerp is linear in the table values, so  sum_j c_j lerp(T_j, x) == lerp(sum_j c_j T_j, x).  The composite
// tables (value, derivative) of every net are built once per parameter upload by k_prepare_dim0; a dimension-0 block is
// then two 16-byte loads and two lerps.  comp[net][mesh] = {Y, DY, 0, 0} (flow layers: spline value and derivative,
// already divided by sum(q); B prior: psi_0 with its sign and norm; M prior: density; MADE: {log_weight, bias}).
// comp2[net][mesh] = {Y_m, DY_m, Y_{m+1}, DY_{m+1}} (k_pair_dim0): both lerp ends in one 16-byte record.  x_r = x_l needs no right end
// (Lerp::t is 0 then); a right end that is not the record's second half (the reference's wrapped negative indices, isplines_jax.py:48-49:
// walkers outside the box) comes from its own record -- a wave-uniform branch, as in rows_dot / ispline_eval.
__device__ __forceinline__ bool any_far(const Lerp& Lp);
__device__ __forceinline__ f32x2 comp_lerp(const f32x4* __restrict__ comp2, const Lerp& Lp) {
    const f32x4 c = comp2[Lp.il];
    float b0 = c[2], b1 = c[3];
    if (any_far(Lp)) {
        const f32x4 r = comp2[Lp.ir];
        const bool far = Lp.ir != Lp.il && Lp.ir != Lp.il + 1;
        b0 = far ? r[0] : b0;
        b1 = far ? r[1] : b1;
    }
    return f32x2{__builtin_fmaf(b0 - c[0], Lp.t, c[0]), __builtin_fmaf(b1 - c[1], Lp.t, c[1])};
}

// 32 activations of one block (accumulator layout) -> the two K=16 B fragments, split hi / lo
struct Frag {
    f16x8 hi[2], lo[2];
};
// Hidden activations never materialise tanh: with r = 1 / (2^xs + 1) (xs = 2*log2(e)*x, scale folded into the weights),
// tanh(x) = 1 - 2r, so the next layer's  W^T tanh + b  equals  (-2W)^T r + (b + sum_k W_k): the factor -2 sits in the packed weights
// (describe_mfma_image) and the column sums in the packed bias (k_fold_bias, at every parameter upload).  What the MFMA consumes
// is the fp16 pair r = hi + lo, hi = rn16(r), lo = rn16(r - hi) (|hi + lo - r| <= 2^-24 r, 2^-25 absolute for tiny r), produced by
// 1.5 instructions per value: v_cvt_pk_f16_f32 for two hi halves, v_fma_mixlo/hi_f16 with the f16 hi as a source for each lo
// (hipcc's own code for the same split is 4 instructions per value).  Inline asm: hipcc pads nothing inside and does not know the
// block is VALU code, so the block carries its own wait states -- one in front (transcendental result -> VALU read), a different
// register between each v_fma_mixhi (op_sel write of a high half) and the v_fma_mixlo that merges into the same register, two behind
// (VALU write -> MFMA source read).  scratch/ubench2/mix_split.hip checks it bit for bit against the plain split.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ void split8(const float (&r)[8], f16x8& hi, f16x8& lo) {
#ifdef WF_SPLIT_TRUNC
    // experiment: hi = the value truncated to 11 significant bits (exact in fp16 above 2^-14), lo = rn16(r - hi): v_and + v_sub + two
    // v_cvt_pk per pair = 3 instructions per value, none of them on the transcendental / mixed-precision rate
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float h0 = __uint_as_float(__float_as_uint(r[j]) & 0xFFFFE000u), h1 = __uint_as_float(__float_as_uint(r[j + 1]) & 0xFFFFE000u);
        const f16x2 Hh = __builtin_convertvector((f32x2){h0, h1}, f16x2);
        const f16x2 Ll = __builtin_convertvector((f32x2){r[j] - h0, r[j + 1] - h1}, f16x2);
        hi[j] = Hh[0]; hi[j + 1] = Hh[1]; lo[j] = Ll[0]; lo[j + 1] = Ll[1];
    }
    return;
#endif
    u32x4 H, L;
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
        : "=&v"(H[0]), "=&v"(H[1]), "=&v"(H[2]), "=&v"(H[3]), "=&v"(L[0]), "=&v"(L[1]), "=&v"(L[2]), "=&v"(L[3])
        : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]), "v"(r[4]), "v"(r[5]), "v"(r[6]), "v"(r[7]));
    hi = __builtin_bit_cast(f16x8, H);
    lo = __builtin_bit_cast(f16x8, L);
}
// One K = 16 step (kt, s) of a 32-unit output block of a K=64 layer for the wave's T tiles:
// acc[t] += Ahi*Bhi + Ahi*Blo + Alo*Bhi (fp32 accumulation).  Wh / Wl: LDS images [k-step][lane][8 halves] of this block; every A
// fragment is read once and used by all T tiles, whose accumulator chains are independent.
template <int T>
__device__ __forceinline__ void mfma_step(const _Float16* Wh, const _Float16* Wl, int kt, int s, const Frag (&in)[T][2], f32x16 (&acc)[T], int lane) {
    const f16x8 ah = *reinterpret_cast<const f16x8*>(Wh + ((kt * 2 + s) * 64 + lane) * 8);
    const f16x8 al = *reinterpret_cast<const f16x8*>(Wl + ((kt * 2 + s) * 64 + lane) * 8);
#ifdef WF_ABL_MFMA   // ablation build (timing only): operands stay live, no matrix instructions
#pragma unroll
    for (int t = 0; t < T; ++t) asm volatile("" : "+v"(acc[t]) : "v"(ah), "v"(al), "v"(in[t][kt].hi[s]), "v"(in[t][kt].lo[s]));
    return;
#endif
#ifdef WF_SETPRIO
    __builtin_amdgcn_s_setprio(WF_SETPRIO);
#endif
#ifdef WF_MFMA_4PROD   // experiment: the lo * lo term as well (2^-24 relative)
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in[t][kt].lo[s], acc[t], 0, 0, 0);
#endif
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in[t][kt].hi[s], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[t][kt].lo[s], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[t][kt].hi[s], acc[t], 0, 0, 0);
#ifdef WF_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}
template <int T>
__device__ __forceinline__ void dense64_block(const _Float16* Wh, const _Float16* Wl, const Frag (&in)[T][2], f32x16 (&acc)[T], int lane) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 2; ++s) mfma_step<T>(Wh, Wl, kt, s, in, acc, lane);
}
// the activation of registers 8s .. 8s+7 of a block (one K = 16 fragment) of every tile: the unit of work that is placed between
// the K steps of an MFMA chain that does not depend on it (hidden_layers, out_block_first)
// CENTER: r - 1/2 instead of r (= -tanh / 2: the layer behind takes the UNFOLDED bias NetOff::b1c, see hidden_layers)
template <int T, bool CENTER = false>
__device__ __forceinline__ void act8(const f32x16 (&x)[T], int s, Frag (&f)[T][2], int ob) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
#ifdef WF_ABL_ACT
        f32x4 a = {x[t][8 * s], x[t][8 * s + 1], x[t][8 * s + 2], x[t][8 * s + 3]}, b = {x[t][8 * s + 4], x[t][8 * s + 5], x[t][8 * s + 6], x[t][8 * s + 7]};
        f[t][ob].hi[s] = __builtin_bit_cast(f16x8, a);
        f[t][ob].lo[s] = __builtin_bit_cast(f16x8, b);
#else
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#ifdef WF_X_ACT
            r[j] = CENTER ? (float)(1.0 / (exp2((double)x[t][8 * s + j]) + 1.0) - 0.5) : x_rinv(x[t][8 * s + j]);
#else
            r[j] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x[t][8 * s + j]) + 1.0f);
            if (CENTER) r[j] = r[j] - 0.5f;
#endif
        }
        split8(r, f[t][ob].hi[s], f[t][ob].lo[s]);
#endif
    }
}

// WF_CENTER (experiment build, round 4): the flow nets' first hidden layer hands r - 1/2 to the second one (hidden_layers<..., CENTER>).  The CPU
// emulation of the kernel's matrix arithmetic (scratch/r04_parity_attribution.py) named that product as the one that moves walkers; on the GPU the
// centred form changed nothing that can be measured (direct agreement with the fp32 oracle on C3's well-conditioned subset 0.935 against 0.940,
// profiles/r04_parity_variants_gpu.txt) at + 0.6 % time: not adopted, the switch and the unfolded bias (NetOff::b1c) stay for the record.
#ifdef WF_CENTER
constexpr bool kCenter = true;
#else
constexpr bool kCenter = false;
#endif
// float offsets inside a net image (wf_model.cpp: build_mfma_image); NBK = 32-row blocks per dimension (1 or 2)
template <int D, int NBK>
struct NetOff {
    static constexpr int S0 = (D + 1) / 2;
    static constexpr int W0 = 0;
    static constexpr int b0 = W0 + 2 * S0 * 64;
    static constexpr int W1h = b0 + 64;
    static constexpr int W1l = W1h + 2048;
    static constexpr int b1 = W1l + 2048;
    static constexpr int W2h = b1 + 64;
    static constexpr int W2l = W2h + (D - 1) * NBK * 1024;
    static constexpr int b2 = W2l + (D - 1) * NBK * 1024;
    static constexpr int z = b2 + 32 * D * NBK;     // zero_params of a gated head [D][NBK][2][16] (zeros otherwise)
    static constexpr int b1c = z + 32 * D * NBK;    // the second hidden layer's bias WITHOUT the column sums of k_fold_bias [2][2][16] (centred activations)
    static constexpr int total = b1c + 64;
};

// Hidden layers of one conditioner net for the wave's T tiles of 32 walkers.  Written in the order the instructions should issue:
// the activation of a finished block sits between the K steps of the next MFMA chain that does not need it yet --
//   chain(layer 2, block 0), K steps of layer-1 block 0  ||  activation of layer-1 block 1
//   chain(layer 2, block 1)                              ||  activation of layer-2 block 0
// Result: h2[t][0] complete, pend[t] = pre-activations of layer-2 block 1 (their activation goes under the first K steps of the
// output chain: out_block_first).
// CENTER (experiment, -DWF_CENTER; see kCenter): the first hidden layer hands r - 1/2 = -tanh/2 to the second one, whose bias is then the plain
// c b1 (NetOff::b1c) instead of c (b1 + sum_k W1_k).  With r itself the products (-2 c W1_k) r_k are of the size of the weights while their sum,
// after the constant cancels, is of the size of sum_k W1_k tanh_k; in a CPU emulation of the matrix arithmetic alone that showed (1.8 % of the
// well-conditioned walkers moved by more than 1e-5 relative through this product, a fourth product lo * lo changing nothing, the centred form
// 0.25 - 0.7 %: profiles/r04_parity_attribution_cpu.txt); in the real kernel it is below the fp32 roundings of everything else.
template <int D, int NBK, int T, bool CENTER = false>
__device__ __forceinline__ void hidden_layers(const float* net, const float (&in)[T][D], int lane, Frag (&h2)[T][2], f32x16 (&pend)[T]) {
    using O = NetOff<D, NBK>;
    const int h = lane >> 5;
    Frag h1[T][2];
    f32x16 a[2][T];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        const f32x16 bias = load16(net + O::b0 + (ob * 2 + h) * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) a[ob][t] = bias;
#pragma unroll
        for (int s = 0; s < O::S0; ++s) {
            const float w = net[O::W0 + (ob * O::S0 + s) * 64 + lane];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float lo = in[t][2 * s];
                const float hi = (2 * s + 1 < D) ? in[t][(2 * s + 1 < D) ? 2 * s + 1 : D - 1] : 0.0f;
                a[ob][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w, h ? hi : lo, a[ob][t], 0, 0, 0);
            }
        }
    }
    act8<T, CENTER>(a[0], 0, h1, 0);
    act8<T, CENTER>(a[0], 1, h1, 0);
    const _Float16* W1h = reinterpret_cast<const _Float16*>(net + O::W1h);
    const _Float16* W1l = reinterpret_cast<const _Float16*>(net + O::W1l);
    constexpr int kB1 = CENTER ? O::b1c : O::b1;
    f32x16 c0[T];
    {
        const f32x16 bias = load16(net + kB1 + h * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) c0[t] = bias;
    }
    WF_PIN();
    mfma_step<T>(W1h, W1l, 0, 0, h1, c0, lane);
    act8<T, CENTER>(a[1], 0, h1, 1);
    WF_PIN();
    mfma_step<T>(W1h, W1l, 0, 1, h1, c0, lane);
    act8<T, CENTER>(a[1], 1, h1, 1);
    WF_PIN();
    mfma_step<T>(W1h, W1l, 1, 0, h1, c0, lane);
    mfma_step<T>(W1h, W1l, 1, 1, h1, c0, lane);
    {
        const f32x16 bias = load16(net + kB1 + (2 + h) * 16);
#pragma unroll
        for (int t = 0; t < T; ++t) pend[t] = bias;
    }
    WF_PIN();
    mfma_step<T>(W1h + 2048, W1l + 2048, 0, 0, h1, pend, lane);
    mfma_step<T>(W1h + 2048, W1l + 2048, 0, 1, h1, pend, lane);
    act8<T>(c0, 0, h2, 0);
    WF_PIN();
    mfma_step<T>(W1h + 2048, W1l + 2048, 1, 0, h1, pend, lane);
    mfma_step<T>(W1h + 2048, W1l + 2048, 1, 1, h1, pend, lane);
    act8<T>(c0, 1, h2, 0);
    WF_PIN();
}

// Output block (dimension d >= 1, row block kb): raw (scaled) outputs o[basis row][walker] in accumulator layout, T tiles.
template <int D, int NBK, int T>
__device__ __forceinline__ void out_block(const float* net, const Frag (&h2)[T][2], int d, int kb, int lane, f32x16 (&o)[T]) {
    using O = NetOff<D, NBK>;
    const int h = lane >> 5;
    const _Float16* W2h = reinterpret_cast<const _Float16*>(net + O::W2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(net + O::W2l);
    const int blk = (d - 1) * NBK + kb;
    const f32x16 bias = load16(net + O::b2 + ((d * NBK + kb) * 2 + h) * 16);
#pragma unroll
    for (int t = 0; t < T; ++t) o[t] = bias;
    dense64_block<T>(W2h + blk * 2048, W2l + blk * 2048, h2, o, lane);
}
// The first output block of a net (d = 1, kb = 0) also finishes the second hidden layer: the activation of its block 1 (pend) runs
// under the two K steps that only need block 0.
template <int D, int NBK, int T>
__device__ __forceinline__ void out_block_first(const float* net, Frag (&h2)[T][2], const f32x16 (&pend)[T], int lane, f32x16 (&o)[T]) {
    using O = NetOff<D, NBK>;
    const int h = lane >> 5;
    const _Float16* W2h = reinterpret_cast<const _Float16*>(net + O::W2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(net + O::W2l);
    const f32x16 bias = load16(net + O::b2 + (NBK * 2 + h) * 16);   // (d = 1, kb = 0)
#pragma unroll
    for (int t = 0; t < T; ++t) o[t] = bias;
    WF_PIN();
    mfma_step<T>(W2h, W2l, 0, 0, h2, o, lane);
    act8<T>(pend, 0, h2, 1);
    WF_PIN();
    mfma_step<T>(W2h, W2l, 0, 1, h2, o, lane);
    act8<T>(pend, 1, h2, 1);
    WF_PIN();
    mfma_step<T>(W2h, W2l, 1, 0, h2, o, lane);
    mfma_step<T>(W2h, W2l, 1, 1, h2, o, lane);
}

// tab: [mesh][8 NBK pieces][NO][side: m, m + 1][4 rows] (wf_model.cpp: pack_rows_pairs); rs: [mesh][NO] (read when RS); bnd (LDS):
// [NBK][half][16: 4 x (lo, hi), 8 unused] support bounds of the lane's pieces (piece_bounds).  One record = both lerp ends and all
// orders of four rows: a lane reads four records per block, each at the mesh index clamped to the piece's support -- outside it the
// table holds the bits of the clamped record, so the values are those at the walker's own index, and the walkers outside a piece's
// support (about 60 % at 29 bases) read two shared, L1-resident records instead of one of their own from L2.  The right end is the
// record's second side when x_r = x_l + 1; x_r = x_l needs no right end (Lerp::t is 0 then); anything else (the reference's negative
// indices wrap around, isplines_jax.py:48-49) fetches the right end from its own record (FAR instantiations).
__device__ __forceinline__ int med3i(int x, int lo, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
}
// the four records of 32-row block kb of one walker half
template <int NBK, int NO, bool FAR>
__device__ __forceinline__ void fetch_block(f32x16 (&A)[NO], f32x16 (&Bv)[NO], const float* __restrict__ tab, const Lerp& Lp, int h, const float* bnd, int kb) {
    const char* tb = reinterpret_cast<const char*>(tab);
    constexpr unsigned kRec = 32u * NO, kMesh = 8u * NBK * kRec;   // bytes per record / per mesh point; 32-bit offsets (tables are < 4 GB)
    const bool far = FAR && Lp.ir != Lp.il && Lp.ir != Lp.il + 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const i32x2 lh = *reinterpret_cast<const i32x2*>(bnd + (kb * 2 + h) * 16 + q * 2);
        const unsigned off = (unsigned)med3i(Lp.il, lh[0], lh[1]) * kMesh + (unsigned)(8 * kb + 2 * q + h) * kRec;
        unsigned offb = off + 16u;
        if (FAR) {
            const unsigned offr = (unsigned)med3i(Lp.ir, lh[0], lh[1]) * kMesh + (unsigned)(8 * kb + 2 * q + h) * kRec;
            offb = far ? offr : offb;
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const f32x4 a = load4g(reinterpret_cast<const float*>(tb + off + o * 32)), b = load4g(reinterpret_cast<const float*>(tb + offb + o * 32));
#pragma unroll
            for (int e = 0; e < 4; ++e) { A[o][q * 4 + e] = a[e]; Bv[o][q * 4 + e] = b[e]; }
        }
    }
}
// part[o] += lerp of (sum_r v_r a_r, sum_r v_r b_r) for one block's rows: two partial sums per end (even / odd registers), in register order
template <int NO>
__device__ __forceinline__ void block_dot(const f32x16& v, const f32x16 (&A)[NO], const f32x16 (&Bv)[NO], float t, float (&part)[NO]) {
    float sa[NO][2], sb[NO][2];
#pragma unroll
    for (int o = 0; o < NO; ++o) sa[o][0] = sa[o][1] = sb[o][0] = sb[o][1] = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 2)
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            sa[o][0] = __builtin_fmaf(v[r], A[o][r], sa[o][0]);
            sb[o][0] = __builtin_fmaf(v[r], Bv[o][r], sb[o][0]);
            sa[o][1] = __builtin_fmaf(v[r + 1], A[o][r + 1], sa[o][1]);
            sb[o][1] = __builtin_fmaf(v[r + 1], Bv[o][r + 1], sb[o][1]);
        }
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        const float Aa = sa[o][0] + sa[o][1], Bb = sb[o][0] + sb[o][1];
        part[o] += __builtin_fmaf(Bb - Aa, t, Aa);
    }
}
// sum over the walker's 32 * NBK rows of v_r * lerp(T'_r) for NO derivative orders (summed over the two lane halves); the products
// follow the records' arrival order.  rsum (may be null): the row sums [mesh][NO] -> rl, rr.  One block per dimension: all records are
// requested first, the row sums behind them.  Two blocks: the row sums first, then block by block -- the second block's records are
// requested when the first block's products are done (both blocks in flight are 128 registers: spills at 12 and 16 waves).
// (scratch/time_waves.py, r02 notes in DESIGN.md: each form is 3 - 6 % slower in the other case.)
template <int NBK, int NO, bool FAR>
__device__ __forceinline__ void rows_lerp_dot(const f32x16 (&v)[NBK], const float* __restrict__ tab, const Lerp& Lp, int h, const float* bnd, float (&out)[NO],
                                              const float* __restrict__ rsum = nullptr, float* rl = nullptr, float* rr = nullptr) {
    auto row_sums = [&]() {   // rsum: [mesh]{R0_m, R1_m, R0_{m+1}, R1_{m+1}} (NO == 2 only): one 16-byte record holds both lerp ends
        if (rsum) {
            static_assert(NO <= 2, "row sums: orders 0 and 1");
            const f32x4 c = load4g(rsum + (size_t)Lp.il * 4);
            f32x4 r = c;
            bool far = false;
            if (FAR) {
                far = Lp.ir != Lp.il && Lp.ir != Lp.il + 1;
                r = load4g(rsum + (size_t)Lp.ir * 4);
            }
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                rl[o] = c[o];
                rr[o] = far ? r[o] : c[2 + o];
            }
        }
    };
    float part[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) part[o] = 0.0f;
    if (NBK > 1) row_sums();
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
        f32x16 A[NO], Bv[NO];
        if (kb > 0) __builtin_amdgcn_sched_barrier(0);
        fetch_block<NBK, NO, FAR>(A, Bv, tab, Lp, h, bnd, kb);
        if (NBK == 1) row_sums();
        block_dot<NO>(v[kb], A, Bv, Lp.t, part);
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) out[o] = xhalf_sum(part[o]);
}
// any lane of the wave whose right end is not its left record's second side (negative indices: rare) -> the FAR instantiation, for the
// whole evaluation (a branch around the fetch alone would make every row live at its end)
__device__ __forceinline__ bool any_far(const Lerp& Lp) { return __builtin_amdgcn_ballot_w64(Lp.ir != Lp.il && Lp.ir != Lp.il + 1) != 0; }

// the prior's sum_r v_r * lerp(row_r): one-order table
template <int NBK>
__device__ __forceinline__ float rows_dot(const f32x16 (&v)[NBK], const float* __restrict__ tab, const Lerp& Lp, int h, const float* bnd) {
    float out[1];
    if (!any_far(Lp)) rows_lerp_dot<NBK, 1, false>(v, tab, Lp, h, bnd, out);
    else rows_lerp_dot<NBK, 1, true>(v, tab, Lp, h, bnd, out);
    return out[0];
}

// y and log(dy + 1e-7) of one I-spline block from its weights v (unnormalised), rS = 1/sum(q), rs = reg * S1; rsum: [mesh][2] row sums
template <int NBK>
__device__ __forceinline__ void ispline_eval(const f32x16 (&v)[NBK], const float* __restrict__ tab, const float* __restrict__ rsum, const Lerp& Lp, int h,
                                             const float* bnd, float rS, float rs, float& y, float& logdy) {
    float num[2], rl[2], rr[2];
    if (!any_far(Lp)) rows_lerp_dot<NBK, 2, false>(v, tab, Lp, h, bnd, num, rsum, rl, rr);
    else rows_lerp_dot<NBK, 2, true>(v, tab, Lp, h, bnd, num, rsum, rl, rr);
    const float ynum = __builtin_fmaf(rs, __builtin_fmaf(rr[0] - rl[0], Lp.t, rl[0]), num[0]);
    const float dnum = __builtin_fmaf(rs, __builtin_fmaf(rr[1] - rl[1], Lp.t, rl[1]), num[1]);
    y = ynum * rS;
    logdy = fast_log(__builtin_fmaf(dnum, rS, 1e-7f));
}

// sigmoid weights of one dimension (NBK blocks) and their two sums: S1 = sum v, Sf = sum v*fk.  Gated head (model_factory.py:64-67):
// v = g * sigmoid(o) + |z|, g = prod_{i<d} x_i^3 of the conditioner's input, z from the net's LDS image.
template <int NBK>
__device__ __forceinline__ void sigmoid_block(f32x16 (&o)[NBK], const float* fk_lds /* [NBK][2][16] */, int h, float& S1, float& Sf,
                                              bool gate = false, float g = 1.0f, const float* z_lds = nullptr) {
    float s1 = 0.0f, sf = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
        const f32x16 fk = load16(fk_lds + (kb * 2 + h) * 16);
        f32x16 zz = fk;
        if (gate) zz = load16(z_lds + (kb * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = act_sigmoid(o[kb][r]);
            if (gate) v = __builtin_fmaf(g, v, zz[r]);
            o[kb][r] = v;
            s1 += v;
            sf = __builtin_fmaf(v, fk[r], sf);
        }
    }
    S1 = xhalf_sum(s1);
    Sf = xhalf_sum(sf);
}

#if defined(WF_STAMP) && defined(WF_STAMP_TILE)
// per-tile variant: slot (iteration & 7) accumulates the wave's elapsed cycles of that iteration of the tile loop (slot 0 includes the prologue)
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if ((k) == 6) {                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            unsigned long long t_;                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            stamp_acc[stamp_iter & 7] += t_ - stamp_last;                                          \
            stamp_last = t_;                                                                       \
            ++stamp_iter;                                                                          \
        }                                                                                          \
    } while (0)
#elif defined(WF_STAMP)
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        stamp_acc[k] += t_ - stamp_last;                                                           \
        stamp_last = t_;                                                                           \
    } while (0)
#else
#define STAMP(k)
#endif

// cooperative copy of n_floats (multiple of 4) global -> LDS, 16 B per lane
template <int kThreads>
__device__ __forceinline__ void stage_floats(const float* __restrict__ src, float* dst, int n_floats) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    const int n4 = n_floats >> 2;
    for (int base = threadIdx.x; base < n4; base += kThreads * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads;
            if (i < n4) v[u] = s4[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads;
            if (i < n4) d4[i] = v[u];
        }
    }
}

// One wave = T tiles of 32 walkers (T = 2: two independent dependency chains per wave, so that one tile's activation / spline
// VALU work sits in the other tile's MFMA shadow inside the same instruction stream; the A operands are read from LDS once per pair).
// IDX: the bin-index output (a diagnostic) is compiled into its own instantiation: the per-lane pointer tests would otherwise
// cut the hot loop into a hundred basic blocks that the instruction scheduler cannot move work across.
// SPEC: the headline family (mean-type box, IMADE layers, Waveflow prior, every net resident in LDS, division-free x_l / n) with the
// model switches as compile-time constants: the tile loop becomes one straight-line block per net.
template <int D, int NBK, int kWaves, int T, bool IDX, bool SPEC>
__global__ __launch_bounds__(kWaves * 64) void k_mfma(const MfmaDev mm, int mode_in, const float* __restrict__ xg, int64_t B,
                                                      float* __restrict__ out, float* __restrict__ u_out, int32_t* __restrict__ idx_out) {
    // mode_in & 3: 0 log_pdf, 1 psi, 2 flow only.  mode_in & 4 (kModePresort): the walkers arrive in ANY order -- every row is sorted in
    // registers (odd-even transposition network: adjacent exchanges only, so the number of exchanges IS the inversion count, ties included)
    // and psi gets the sign (-1)^inversions: the antisymmetrised wavefunction psi(sort(x)) * (-1)^inv of helpers.py:55-58 /
    // coordinates.py:41-51 without a host loop or a second pass over the walkers.
    const int mode = mode_in & 3;
    const bool presort = (mode_in & 4) != 0;
    // mm is passed BY VALUE: it lives in the kernarg segment, so its fields are scalar loads and the table pointers
    // are known to be global (with a pointer-to-struct argument hipcc emitted flat_load for every table access).
    // LDS: [constants][net slot(s)].  Resident mode: every net has its own slot, staged once.  Staged mode (the nets do
    // not fit together, e.g. D >= 4 with 3 layers): ONE slot; every workgroup walks its chunk of kWaves * T tiles through the
    // nets, re-staging the slot between two barriers per net (the per-tile state is D + 1 registers).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kThreads = kWaves * 64;
    const int box_kind = SPEC ? (int)WF_BOX_MEAN : mm.box_kind, layer_kind = SPEC ? (int)WF_LAYER_IMADE : mm.layer_kind;
    // (SPEC at D = 2: every net resident, a compile-time fact of that family; SPEC at D > 2 -- the electron chains -- may run staged)
    const int prior_kind = SPEC ? (int)WF_PRIOR_WAVEFLOW : mm.prior_kind, staged = (SPEC && D == 2) ? 0 : mm.staged, exact_div = SPEC ? 0 : mm.exact_div;
    // Resident mode hands the workgroup's tiles to its waves through a counter in LDS instead of a fixed share per wave: the SIMD's issue
    // arbitration favours its oldest wave (measured with per-tile s_memtime stamps: at 16 waves the first tile of SIMD slot 0 takes 39 k
    // cycles, that of slot 3 151 k), so with equal shares the old waves leave early and the last tiles run at one or two waves per SIMD --
    // a quarter of the launch time was that tail.  Every wave leaves the loop when the counter passes the workgroup's share.
    __shared__ int next_slot;
    if (threadIdx.x == 0) next_slot = 0;
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    if (!staged) stage_floats<kThreads>(mm.image, lds + mm.const_floats, mm.net_floats * mm.n_nets);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float* fkI = lds;                        // [NBK][2][16] remove_bias * keep factors of the flow-layer I-spline
    const float* fkP = lds + 32 * NBK;             // [NBK][2][16] prior: keep (B) or remove_bias * keep (M)
    const float* ob2b = lds + 64 * NBK;            // [NBK out][NBK in]{hi, lo}[2 K steps][64][8 halves] ob_to_b in f16-MFMA A order
    const float* bndI = ob2b + NBK * NBK * 1024;   // int32 [NBK][2][16]: support bounds of the table pieces (fetch_block), at the lane stride of fkI / fkP
    const float* bndP = bndI + 32 * NBK;
    const float* cbP = bndP + 32 * NBK;            // [NBK][2][16] constant term of the B prior's boundary map times ob_to_b (mm.p_bias)
    float* slots = lds + mm.const_floats;
    const int64_t n_tiles = (B + 31) >> 5;
    constexpr int kTilesPerChunk = kWaves * T;
    const int64_t n_chunks = (n_tiles + kTilesPerChunk - 1) / kTilesPerChunk;
    const int idx_stride = (mm.n_layers + 1) * D * 2;
    const float L = mm.box_L, tol = 1e-7f;
    const float rn_mesh = 1.0f / (float)(mm.n_mesh - 1);
    // wave-uniform terms of the mean-type box transform's first step, held in scalar registers (the compiler hoists them into vector ones)
    const float box_space0 = uniform_f(2 * L + tol), box_nlog0 = uniform_f(0.0f - fast_log(2 * L + tol));
#ifdef WF_STAMP
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    [[maybe_unused]] int stamp_iter = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
    [[maybe_unused]] const unsigned long long stamp_t0 = stamp_last;
#endif

#ifdef WF_STAGGER
    // The waves of one SIMD (w, w + 4, w + 8, ...) run the same program on tiles of equal cost and would stay in lockstep: matrix
    // phases against matrix phases, table-load waits against table-load waits.  A one-off start delay of a quarter / half net period
    // per SIMD slot keeps them out of phase (there is no barrier after this point in resident mode).
    for (int q = 0; q < ((wave >> 2) & 3); ++q) __builtin_amdgcn_s_sleep(WF_STAGGER);
#endif
    int f16_bad = 0;   // wave-uniform: scalar loads
    if (mm.f16_ovf)
        for (int n = 0; n < mm.n_nets; ++n) f16_bad |= mm.f16_ovf[n];
    // this workgroup's chunks: blockIdx.x, blockIdx.x + gridDim.x, ...; a slot = (chunk, wave position) = T tiles
    const int64_t my_chunks = n_chunks > (int64_t)blockIdx.x ? (n_chunks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int my_slots = (int)(my_chunks * kWaves);
    // ---- the pieces of one tile pass (lambdas: inlined).  Resident mode strings them together per tile; staged mode runs them net by net
    // over a super-chunk of tiles whose state waits in LDS, so that a net is staged once per super-chunk.
    auto tile_ids = [&](int64_t chunk, int wpos, int64_t (&w)[T], bool (&valid)[T], int32_t* (&idx)[T]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t tile = (chunk * kWaves + wpos) * T + t;    // may be >= n_tiles in the last chunk: computed, never stored
            w[t] = tile * 32 + j;
            valid[t] = w[t] < B;
            idx[t] = (IDX && idx_out && valid[t] && h == 0) ? idx_out + w[t] * idx_stride : nullptr;
        }
    };
    auto load_box = [&](const int64_t (&w)[T], const bool (&valid)[T], float (&cur)[T][D], float (&logdet)[T]) __attribute__((always_inline)) {
        float nxt[T][D];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int64_t wl = valid[t] ? w[t] : B - 1;
#pragma unroll
            for (int d = 0; d < D; ++d) cur[t][d] = xg[wl * D + d];
            if (presort) {
#pragma unroll
                for (int r = 0; r < D; ++r)
#pragma unroll
                    for (int i = r & 1; i + 1 < D; i += 2) {
                        const float a = cur[t][i], b = cur[t][i + 1];
                        const bool sw = a > b;
                        cur[t][i] = sw ? b : a;
                        cur[t][i + 1] = sw ? a : b;
                    }
            }
        }
        // ---- BoxTransformLayer (made.py:118-137, 156-183); IEEE divisions: layer-0 bin indices must be exact
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float ld = 0.0f;
            if (box_kind == WF_BOX_MEAN) {
                float s = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) s = s + cur[t][d];
                const float mean = s / (float)D;
                const float l = mean - cur[t][0];
                const float wd = cur[t][D - 1] - cur[t][0];
                float space_left = 2 * L;
#pragma unroll
                for (int i = 0; i < D - 1; ++i) {
                    const float diff = cur[t][i + 1] - cur[t][i];
                    nxt[t][i] = diff / (i == 0 ? box_space0 : space_left + tol);
                    ld = i == 0 ? box_nlog0 : ld - fast_log(space_left + tol);   // (0 - log: the same bits)
                    space_left = space_left - diff;
                }
                nxt[t][D - 1] = (mean + L - l) / (2 * L - wd + tol);
                ld = ld - fast_log(2 * L - wd + tol);
#pragma unroll
                for (int d = 0; d < D; ++d) cur[t][d] = nxt[t][d];
            } else if (box_kind == WF_BOX_FIRST) {
                nxt[t][0] = (cur[t][0] + L) / (2 * L);
                float ls = 0.0f;
#pragma unroll
                for (int i = 1; i < D; ++i) nxt[t][i] = (cur[t][i] - cur[t][i - 1]) / (L - cur[t][i - 1] + tol);
#pragma unroll
                for (int i = 0; i < D - 1; ++i) ls = ls + fast_log(L - cur[t][i] + tol);
                ld = -fast_log(2 * L) - ls;
#pragma unroll
                for (int d = 0; d < D; ++d) cur[t][d] = nxt[t][d];
            }
            logdet[t] = ld;
        }

    };
    auto flow_layer = [&](int l, const float* net, float (&cur)[T][D], float (&logdet)[T], int32_t* (&idx)[T]) __attribute__((always_inline)) {
        float nxt[T][D];
        {
            Frag h2[T][2];
            f32x16 pend[T];
            STAMP(0);
            hidden_layers<D, NBK, T, kCenter>(net, cur, lane, h2, pend);
            STAMP(1);
            if (layer_kind == WF_LAYER_IMADE) {
                // dimension 0: walker-independent weights -> composite table (k_prepare_dim0)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const Lerp Lp = make_lerp(cur[t][0], mm.n_mesh, rn_mesh, exact_div);
                    if (IDX && idx[t]) { idx[t][(l * D) * 2] = Lp.xl; idx[t][(l * D) * 2 + 1] = Lp.xr; }
                    const f32x2 c0 = comp_lerp(mm.comp2 + (size_t)l * mm.n_mesh, Lp);
                    nxt[t][0] = c0[0];
                    logdet[t] = logdet[t] + fast_log(c0[1] + 1e-7f);
                }
                STAMP(2);
                const bool gate_i = !SPEC && mm.i_gate != 0;
                float gl[T];   // gate of dimension d: prod_{i<d} (layer input)_i^3
#pragma unroll
                for (int t = 0; t < T; ++t) gl[t] = 1.0f;
#pragma unroll (kDimUnroll<D>)
                for (int d = 1; d < D; ++d) {
                    if (D > 4) __builtin_amdgcn_sched_barrier(0);   // long chains: one dimension at a time (the scheduler otherwise keeps the records of several dimensions in flight: 29 -> 218 spilled registers at 12 waves)
                    if (gate_i) {
#pragma unroll
                        for (int t = 0; t < T; ++t) gl[t] = gl[t] * (cur[t][d - 1] * cur[t][d - 1] * cur[t][d - 1]);
                    }
                    f32x16 v[T][NBK];
#pragma unroll
                    for (int kb = 0; kb < NBK; ++kb) {
                        f32x16 o[T];
                        if (d == 1 && kb == 0) out_block_first<D, NBK, T>(net, h2, pend, lane, o);
                        else out_block<D, NBK, T>(net, h2, d, kb, lane, o);
#pragma unroll
                        for (int t = 0; t < T; ++t) v[t][kb] = o[t];
                    }
                    STAMP(3);
                    // (Requesting the table rows before the output MFMAs, or before the sigmoids -- they depend on the layer input only --
                    // costs 64 * NBK live registers per tile: spills at 12 and 16 waves, -2 % at 8; r02 notes in DESIGN.md.)
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        float S1, Sf;
                        sigmoid_block<NBK>(v[t], fkI, h, S1, Sf, gate_i, gl[t], net + NetOff<D, NBK>::z + d * NBK * 32);
                        const float rs = mm.i_reg * S1;
                        const float rS = x_rcp(__builtin_fmaf(rs, mm.F_I, Sf));
                        const Lerp Lp = make_lerp(cur[t][d], mm.n_mesh, rn_mesh, exact_div);
                        if (IDX && idx[t]) { idx[t][(l * D + d) * 2] = Lp.xl; idx[t][(l * D + d) * 2 + 1] = Lp.xr; }
                        float ld;
                        ispline_eval<NBK>(v[t], mm.tabI, mm.rsI, Lp, h, bndI, rS, rs, nxt[t][d], ld);
                        logdet[t] = logdet[t] + ld;
                    }
                    STAMP(5);
                }
            } else {
                // MADE (made.py:21-27): rows 0 / 1 of block d = log_weight / bias (lane half 0, registers 0 / 1)
                float ls[T];
#pragma unroll
                for (int t = 0; t < T; ++t) ls[t] = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    if (d == 0) {
                        const f32x4 c0 = mm.comp[(size_t)l * mm.n_mesh];   // {log_weight, bias}: constants
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            nxt[t][0] = (cur[t][0] - c0[1]) * __expf(-c0[0]);
                            ls[t] = ls[t] + c0[0];
                        }
                    } else {
                        f32x16 o[T];
                        if (d == 1) out_block_first<D, NBK, T>(net, h2, pend, lane, o);
                        else out_block<D, NBK, T>(net, h2, d, 0, lane, o);
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            const float lw = __shfl(o[t][0], j);
                            const float bias = __shfl(o[t][1], j);
                            nxt[t][d] = (cur[t][d] - bias) * __expf(-lw);
                            ls[t] = ls[t] + lw;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < T; ++t) logdet[t] = logdet[t] - ls[t];
            }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int d = 0; d < D; ++d) cur[t][d] = nxt[t][D - 1 - d];  // Reverse (bijections.py:337-340)
                }
    };
    auto head_store = [&](const float* net_prior, float (&cur)[T][D], float (&logdet)[T], const int64_t (&w)[T], const bool (&valid)[T],
                          int32_t* (&idx)[T]) __attribute__((always_inline)) {
        float nxt[T][D];
        const float* net = net_prior;
        // ---- density head
        float result[T];
#pragma unroll
        for (int t = 0; t < T; ++t) result[t] = logdet[t];
        if (mode != 2) {
            if (prior_kind == WF_PRIOR_WAVEFLOW || prior_kind == WF_PRIOR_MFLOW) {
                const bool wavefn = prior_kind == WF_PRIOR_WAVEFLOW;
                const f32x4* comp_p = mm.comp2 + (size_t)mm.n_layers * mm.n_mesh;
                Frag h2[T][2];
                f32x16 pend[T];
                hidden_layers<D, NBK, T>(net, cur, lane, h2, pend);   // the conditioner sees the unclipped u (wavefunctions.py:40)
                float lp[T], prod[T], gp[T];
                const bool gate_p = !SPEC && mm.p_gate != 0;   // gated head: prod_{i<d} u_i^3 of the conditioner's input, the unclipped u
#pragma unroll
                for (int t = 0; t < T; ++t) { lp[t] = 0.0f; prod[t] = 1.0f; gp[t] = 1.0f; }
#pragma unroll (kDimUnroll<D>)
                for (int d = 0; d < D; ++d) {
                    if (D > 4) __builtin_amdgcn_sched_barrier(0);
                    if (gate_p && d > 0) {
#pragma unroll
                        for (int t = 0; t < T; ++t) gp[t] = gp[t] * (cur[t][d - 1] * cur[t][d - 1] * cur[t][d - 1]);
                    }
                    float val[T];   // psi_d (B prior) or the density factor (M prior)
                    float uc[T];
                    Lerp Lp[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        uc[t] = fminf(fmaxf(cur[t][d], 0.0f), 1.0f);   // the spline sees the clipped one (:45)
                        Lp[t] = make_lerp(uc[t], mm.n_mesh, rn_mesh, exact_div);
                        if (IDX && idx[t]) { idx[t][(mm.n_layers * D + d) * 2] = Lp[t].xl; idx[t][(mm.n_layers * D + d) * 2 + 1] = Lp[t].xr; }
                    }
                    if (d == 0) {
#pragma unroll
                        for (int t = 0; t < T; ++t) val[t] = comp_lerp(comp_p, Lp[t])[0];
                    } else if (wavefn) {
                        f32x16 o[NBK][T];
#pragma unroll
                        for (int kb = 0; kb < NBK; ++kb) {
                            if (d == 1 && kb == 0) out_block_first<D, NBK, T>(net, h2, pend, lane, o[kb]);
                            else out_block<D, NBK, T>(net, h2, d, kb, lane, o[kb]);
                        }
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            float s1 = 0.0f, amax = 0.0f;
                            if (gate_p) {   // w = g * o + z (signed head: zero_params as they are)
#pragma unroll
                                for (int kb = 0; kb < NBK; ++kb) {
                                    const f32x16 zz = load16(net + NetOff<D, NBK>::z + (d * NBK + kb) * 32 + h * 16);
#pragma unroll
                                    for (int r = 0; r < 16; ++r) o[kb][t][r] = __builtin_fmaf(gp[t], o[kb][t][r], zz[r]);
                                }
                            }
#pragma unroll
                            for (int kb = 0; kb < NBK; ++kb) {
                                const f32x16 keep = load16(fkP + (kb * 2 + h) * 16);
#pragma unroll
                                for (int r = 0; r < 16; ++r) {
                                    s1 += o[kb][t][r];
                                    o[kb][t][r] = o[kb][t][r] * keep[r];
                                    amax = fmaxf(amax, fabsf(o[kb][t][r]));
                                }
                            }
                            s1 = xhalf_sum(s1);
                            // c = (o * keep) @ ob_to_b as split-fp16 MFMA products.  The head's outputs are unbounded, fp16 is not: the
                            // walker's 32 * NBK values are scaled by a power of two so that the largest lies in [0.5, 1) -- exact, and
                            // psi_d = sign * (c . lerp) / |c| does not see a common factor.  WF_PRIOR_QUOTIENT=1 (debug): divide by the signed
                            // sum first, as model_factory.py:69 does (same function in real arithmetic, a different fp32 rounding pattern).
                            amax = xhalf_max(amax);
                            const int ex = amax > 0.0f ? __builtin_amdgcn_frexp_expf(amax) : 0;
                            // one multiplier per walker: the power of two (exact, same as ldexp) or, in quotient mode, 1 / sum
                            const float scale = mm.prior_quotient ? 1.0f / s1 : __builtin_amdgcn_ldexpf(1.0f, -ex);
                            Frag of[NBK];
#pragma unroll
                            for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
                                for (int s = 0; s < 2; ++s) {
                                    float r8[8];
#pragma unroll
                                    for (int jj = 0; jj < 8; ++jj) r8[jj] = o[kb][t][8 * s + jj] * scale;
                                    split8(r8, of[kb].hi[s], of[kb].lo[s]);
                                }
                            f32x16 c[NBK];
                            float n2 = 0.0f;
                            const _Float16* obh = reinterpret_cast<const _Float16*>(ob2b);
#pragma unroll
                            for (int ko = 0; ko < NBK; ++ko) {
                                c[ko] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                                for (int ki = 0; ki < NBK; ++ki)
#pragma unroll
                                    for (int s = 0; s < 2; ++s) {
                                        const _Float16* blk = obh + (size_t)(ko * NBK + ki) * 2048;
                                        const f16x8 ah = *reinterpret_cast<const f16x8*>(blk + (s * 64 + lane) * 8);
                                        const f16x8 al = *reinterpret_cast<const f16x8*>(blk + 1024 + (s * 64 + lane) * 8);
#ifdef WF_MFMA_4PROD
                                        c[ko] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, of[ki].lo[s], c[ko], 0, 0, 0);
#endif
                                        c[ko] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, of[ki].hi[s], c[ko], 0, 0, 0);
                                        c[ko] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, of[ki].lo[s], c[ko], 0, 0, 0);
                                        c[ko] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, of[ki].hi[s], c[ko], 0, 0, 0);
                                    }
                                if (!SPEC && mm.p_bias) {   // a boundary constraint with a non-zero value: c += (sum o) * cb, in the scaled units of c
                                    const float bs = scale * s1;
                                    const f32x16 cb = load16(cbP + (ko * 2 + h) * 16);
#pragma unroll
                                    for (int r = 0; r < 16; ++r) c[ko][r] = __builtin_fmaf(bs, cb[r], c[ko][r]);
                                }
#pragma unroll
                                for (int r = 0; r < 16; ++r) n2 = __builtin_fmaf(c[ko][r], c[ko][r], n2);
                            }
                            const float rnorm = x_rsq(xhalf_sum(n2));
                            const float v0 = rows_dot<NBK>(c, mm.tabP, Lp[t], h, bndP) * rnorm;
                            val[t] = (s1 < 0.0f && !mm.prior_quotient) ? -v0 : v0;
                        }
                    } else {
                        // MFlow (distributions.py:139-163): M-spline table with the row factors folded in
                        f32x16 v[T][NBK];
#pragma unroll
                        for (int kb = 0; kb < NBK; ++kb) {
                            f32x16 o[T];
                            if (d == 1 && kb == 0) out_block_first<D, NBK, T>(net, h2, pend, lane, o);
                            else out_block<D, NBK, T>(net, h2, d, kb, lane, o);
#pragma unroll
                            for (int t = 0; t < T; ++t) v[t][kb] = o[t];
                        }
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            float S1, Sf;
                            sigmoid_block<NBK>(v[t], fkP, h, S1, Sf, gate_p, gp[t], net + NetOff<D, NBK>::z + d * NBK * 32);
                            val[t] = rows_dot<NBK>(v[t], mm.tabP, Lp[t], h, bndP) * __builtin_amdgcn_rcpf(Sf);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        if (wavefn) {
                            const bool constrained = (mm.constrained_mask >> d) & 1u;
                            if (mode == 0) {
                                float pr = val[t] * val[t];
                                if (constrained) pr = pr * 0.5f;
                                lp[t] = lp[t] + fast_log(pr + 1e-7f);
                            } else {
                                const float vv = constrained ? val[t] * 0.70710678118654752f : val[t];
                                prod[t] = prod[t] * vv;
                            }
                        } else {
                            lp[t] = lp[t] + fast_log(val[t] + 1e-7f);
                        }
                        nxt[t][d] = uc[t];
                    }
                }
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    result[t] = (wavefn && mode != 0) ? prod[t] * __expf(0.5f * logdet[t]) : lp[t] + logdet[t];
#pragma unroll
                    for (int d = 0; d < D; ++d) cur[t][d] = nxt[t][d];
                }
            } else if (prior_kind == WF_PRIOR_UNIFORM) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
#pragma unroll
                    for (int d = 0; d < D; ++d) cur[t][d] = fminf(fmaxf(cur[t][d], 0.0f), 1.0f);
                    result[t] = logdet[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    float lpn = 0.0f;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const float z = cur[t][d] + mm.normal_offset;
                        lpn = lpn + (1.8378770664093453f + z * z) * -0.5f;
                    }
                    result[t] = lpn + logdet[t];
                }
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (valid[t] && h == 0) {
                if (f16_bad) result[t] = __builtin_nanf("");   // a packed weight outside the fp16 range (k_fold_bias): NaN, not what inf operands made of it
                if (presort && mode == 1) {   // (-1)^inversions of the walker as it arrived (coordinates.py:41-51): the row is read again, nothing rides through the nets
                    int inv = 0;
                    float xr[D];
#pragma unroll
                    for (int d = 0; d < D; ++d) xr[d] = xg[w[t] * D + d];
#pragma unroll
                    for (int i = 0; i < D; ++i)
#pragma unroll
                        for (int k = i + 1; k < D; ++k) inv += xr[i] > xr[k] ? 1 : 0;
                    if (inv & 1) result[t] = -result[t];
                }
                out[w[t]] = result[t];
                if (u_out) {
#pragma unroll
                    for (int d = 0; d < D; ++d) u_out[w[t] * D + d] = cur[t][d];
                }
            }
    };

    if (!staged) {
        for (;;) {
            int q = 0;
            if (lane == 0) q = __hip_atomic_fetch_add(&next_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            q = __builtin_amdgcn_readfirstlane(q);
            if (q >= my_slots) break;
            const int64_t chunk = (int64_t)blockIdx.x + (int64_t)(q / kWaves) * gridDim.x;
            float cur[T][D], logdet[T];
            int64_t w[T];
            bool valid[T];
            int32_t* idx[T];
            tile_ids(chunk, q % kWaves, w, valid, idx);
            load_box(w, valid, cur, logdet);
            for (int l = 0; l < mm.n_layers; ++l) flow_layer(l, slots + (size_t)l * mm.net_floats, cur, logdet, idx);
            head_store(slots + (size_t)mm.n_layers * mm.net_floats, cur, logdet, w, valid, idx);
            STAMP(6);
        }
    } else {
        // Staged mode (the nets do not fit LDS together): ONE slot.  A super-chunk = kWaves * T * tps tiles; every wave walks its tps
        // tile groups through net after net, the (D + 1) floats of a walker's state between two nets wait in LDS ([wave][group][T][D + 1][32]),
        // and the slot is re-staged between two barriers once per net and super-chunk (round 2 re-staged it per chunk of kWaves * T tiles).
        float* state = slots + mm.net_floats + (size_t)wave * kStagedGroups * T * (D + 1) * 32;
        const int tps = mm.staged_groups;                       // tile groups per wave and super-chunk, 1 .. kStagedGroups (host: by the batch)
        const int64_t n_super = (n_chunks + tps - 1) / tps;
        const bool prior_net = mode != 2 && (prior_kind == WF_PRIOR_WAVEFLOW || prior_kind == WF_PRIOR_MFLOW);
        for (int64_t sc = blockIdx.x; sc < n_super; sc += gridDim.x) {
            for (int p = 0; p <= mm.n_layers; ++p) {
                if (p < mm.n_layers || prior_net) {
                    __syncthreads();   // every wave is done with the previous occupant of the slot
                    stage_floats<kThreads>(mm.image + (size_t)p * mm.net_floats, slots, mm.net_floats);
                    __syncthreads();
                }
                for (int g = 0; g < tps; ++g) {
                    const int64_t chunk = sc * tps + g;
                    if (chunk >= n_chunks) break;
                    float cur[T][D], logdet[T];
                    int64_t w[T];
                    bool valid[T];
                    int32_t* idx[T];
                    tile_ids(chunk, wave, w, valid, idx);
                    float* st = state + (size_t)g * T * (D + 1) * 32;
                    if (p == 0) load_box(w, valid, cur, logdet);
                    else {
#pragma unroll
                        for (int t = 0; t < T; ++t) {
#pragma unroll
                            for (int d = 0; d < D; ++d) cur[t][d] = st[(t * (D + 1) + d) * 32 + j];
                            logdet[t] = st[(t * (D + 1) + D) * 32 + j];
                        }
                    }
                    if (p < mm.n_layers) {
                        flow_layer(p, slots, cur, logdet, idx);
                        if (h == 0) {
#pragma unroll
                            for (int t = 0; t < T; ++t) {
#pragma unroll
                                for (int d = 0; d < D; ++d) st[(t * (D + 1) + d) * 32 + j] = cur[t][d];
                                st[(t * (D + 1) + D) * 32 + j] = logdet[t];
                            }
                        }
                    } else head_store(slots, cur, logdet, w, valid, idx);
                }
            }
        }
    }
#ifdef WF_STAMP
    if (mm.dbg && lane == 0) {
        unsigned long long* g = reinterpret_cast<unsigned long long*>(mm.dbg) + ((size_t)blockIdx.x * kWaves + wave) * 8;
#ifdef WF_STAMP_SPAN   // absolute start / finish of the wave's tile loop and its number of iterations; [3]: finish on the 100 MHz
        // constant clock (s_memrealtime), which unlike s_memtime is comparable between workgroups on different XCDs
        g[0] = stamp_t0; g[1] = stamp_last; g[2] = (unsigned long long)stamp_iter;
        { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory"); g[3] = rt_; }
#else
        for (int k = 0; k < 8; ++k) g[k] = stamp_acc[k];
#endif
    }
#endif
}

template <int D, int NBK, int kWaves, int T, bool IDX, bool SPEC>
int launch_dwi(const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, hipStream_t s) {
    static DynLdsSlots cfg{};   // (one table per instantiation)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_mfma<D, NBK, kWaves, T, IDX, SPEC>), lds_bytes, &cfg)) return rc;
    const int64_t n_tiles = (B + 31) / 32;
    const int64_t n_chunks = (n_tiles + kWaves * T - 1) / (kWaves * T);
    MfmaDev md = *mdev;
    int64_t grid = n_chunks;
    if (md.staged) {   // tile groups per wave and super-chunk: as many as keep 256 workgroups busy, at most kStagedGroups (the LDS state area)
        int64_t g = (n_chunks + 255) / 256;
        md.staged_groups = (int)(g < 1 ? 1 : (g > kStagedGroups ? kStagedGroups : g));
        grid = (n_chunks + md.staged_groups - 1) / md.staged_groups;
    }
    if (grid > 256) grid = 256;  // one persistent workgroup per CU
    hipLaunchKernelGGL((k_mfma<D, NBK, kWaves, T, IDX, SPEC>), dim3((unsigned)grid), dim3(kWaves * 64), lds_bytes, s, md, mode, x, B, out, u, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

// launch_dw<D, NBK, kWaves, T>: the shape's kernel; with a bin-index buffer the request goes to the IDX instantiation, which is built
// for T = 1 only (kWaves as given).
template <int D, int NBK, int kWaves, int T>
int launch_dw(const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, hipStream_t s) {
    if (idx) {
        if constexpr (T == 1) return launch_dwi<D, NBK, kWaves, 1, true, false>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        else return WF_ERR_UNSUPPORTED;
    }
    if constexpr (D == 2 || (D == 8 && NBK == 1)) {   // the specialised build exists for the two-particle shapes and for the 8-electron chain (config C4)
        if (mdev->box_kind == WF_BOX_MEAN && mdev->layer_kind == WF_LAYER_IMADE && mdev->prior_kind == WF_PRIOR_WAVEFLOW && (D > 2 || !mdev->staged) &&
            !mdev->exact_div && !mdev->i_gate && !mdev->p_gate && !mdev->p_bias)
            return launch_dwi<D, NBK, kWaves, T, false, true>(mdev, lds_bytes, mode, x, B, out, u, nullptr, s);
    }
    return launch_dwi<D, NBK, kWaves, T, false, false>(mdev, lds_bytes, mode, x, B, out, u, nullptr, s);
}

}  // namespace mfma
}  // namespace wf
