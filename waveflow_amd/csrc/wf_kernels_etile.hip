// wf_kernels_etile.hip -- local energy of large batches on the matrix cores (gfx950).
//
// H psi = -1/2 laplacian(psi) + V psi (physics.py:50-52, 79-93) needs psi with its gradient and Laplacian with respect to the walker's
// coordinates.  The wave kernels (wf_kernels_wave.hip) carry that (value, gradient, Laplacian / 2) jet through the model with one wave per
// walker; their conditioner products are LDS-fed GEMVs.  For large batches of the two-particle family this file splits the work by what
// the hardware is good at, one launch pair per conditioner net:
//
//   k_etile_cond   32 walkers per wave tile, the conditioner of ONE net on the matrix cores.  With two particles the MADE masks leave the
//                  conditioner ONE input (hidden degrees arange(64) % (D - 1) = 0: model_factory.py:15): every hidden and output unit is a
//                  function of the scalar u_0, so what travels is its Taylor triple (f, f', f'') in u_0 -- three channels = three extra
//                  column groups of the same split-fp16 MFMA products k_mfma issues (the weight operand is shared: mfma_step<3>); the
//                  activations r(x) = 1 / (2^x + 1) propagate the triple on the VALU.  Derivative channels are unbounded, fp16 is not: every
//                  (walker, channel) column is scaled by a power of two around each product (exact).  Output: the head's pre-activation
//                  triples, [row][channel][walker] in HBM (for the prior: already multiplied by ob_to_b); the head kernels turn them into
//                  jets in (x0, x1) with the chain rule through the jet of u_0.
//   k_etile_flow / k_etile_prior   one LANE per walker: sigmoid head, normalisations, table lerps of derivative orders 0..3 and the
//                  log-determinant as jet arithmetic in registers; row sums are sequential loops (no cross-lane traffic), the walker
//                  index is the fastest-moving one of every array (coalesced).
//
// State between launches (SoA, walker fastest): u_0, u_1, log det as jets; 12 floats per walker.  Head triples through HBM: 2 x 384 B per
// walker and net.  Same function as k_wave_fwd<2, RF<2>> + k_energy_out (same derivative rule of the table lerp: order nd -> table nd + 1),
// checked against it and against the torch oracle (tests/test_gpu_energy.py).  Coverage: D = 2, <= 32 bases, mean-type box, IMADE layers,
// Waveflow prior, ungated heads (every homogeneous boundary dictionary: the tables carry the map); everything else stays on the wave kernel.
#include "wf_mfma_impl.h"

// The jet / Taylor algebra of this file is checked against oracles by tolerance, not by operation order: multiply-add pairs may fuse (the build's
// default is -ffp-contract=off).  The pragma is lexical: the index arithmetic of the table lerp (make_lerp, div_by_n in wf_mfma_impl.h, included
// above) keeps the reference's separate roundings, so the bin indices stay bit-exact.
#pragma clang fp contract(fast)

namespace wf {

namespace {
using namespace mfma;
constexpr int NCH = 3;   // channels of the conditioner: (f, df/du0, d2f/du0^2)

struct J {   // value, d/dx0, d/dx1, laplacian / 2
    float v, a, b, h;
};
__device__ __forceinline__ J jc(float c) { return J{c, 0.0f, 0.0f, 0.0f}; }
__device__ __forceinline__ J operator+(J x, J y) { return J{x.v + y.v, x.a + y.a, x.b + y.b, x.h + y.h}; }
__device__ __forceinline__ J operator-(J x, J y) { return J{x.v - y.v, x.a - y.a, x.b - y.b, x.h - y.h}; }
__device__ __forceinline__ J operator+(J x, float c) { return J{x.v + c, x.a, x.b, x.h}; }
__device__ __forceinline__ J operator*(J x, float c) { return J{x.v * c, x.a * c, x.b * c, x.h * c}; }
__device__ __forceinline__ J operator*(J x, J y) {
    return J{x.v * y.v, x.v * y.a + y.v * x.a, x.v * y.b + y.v * x.b, x.v * y.h + y.v * x.h + (x.a * y.a + x.b * y.b)};
}
// f(x) from f, f', f'' at x.v
__device__ __forceinline__ J japply(J x, float f, float f1, float f2) {
    return J{f, f1 * x.a, f1 * x.b, f1 * x.h + 0.5f * f2 * (x.a * x.a + x.b * x.b)};
}
__device__ __forceinline__ J jrcp(J x) { const float f = 1.0f / x.v; return japply(x, f, -f * f, 2.0f * f * f * f); }
__device__ __forceinline__ J jlog(J x) { const float f1 = 1.0f / x.v; return japply(x, logf(x.v), f1, -f1 * f1); }
__device__ __forceinline__ J jrsqrt(J x) { const float f = rsqrtf(x.v), q = 1.0f / x.v; return japply(x, f, -0.5f * f * q, 0.75f * f * q * q); }
__device__ __forceinline__ J jexp_half(J x) { const float f = expf(0.5f * x.v); return japply(x, f, 0.5f * f, 0.25f * f); }
// r(x) = 1 / (2^x + 1): the activation of the MFMA images (tanh = 1 - 2 r with 2 log2(e) folded into the weights; sigmoid = r with -log2(e))
__device__ __forceinline__ J jr(J x) {
    const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x.v) + 1.0f);
    const float r1 = -0.6931471805599453f * r * (1.0f - r);
    return japply(x, r, r1, -0.6931471805599453f * r1 * (1.0f - 2.0f * r));
}
// table function of a jet argument: value t0, first / second derivative t1 / t2 (the lerps of the next two cached orders)
__device__ __forceinline__ J jlift(float t0, float t1, float t2, J u) { return japply(u, t0, t1, t2); }

// ---- Two-variable Taylor algebra for the heads.  Behind the conditioner everything a head sums over its rows is SEPARABLE in the layer's two
// inputs: the weights v_j are functions of s = u_0 alone (the conditioner's Taylor triple), the table rows T_j functions of t = u_1 alone.
// So the row loop accumulates plain scalars -- sum_j v_j^(a)(s) g_j T_j^(k)(t) for the few (a, k) the second-order expansion needs -- and
// the quotients, logarithms and the change to (x0, x1) jets are done ONCE per walker on the six partials {f, f_s, f_t, f_ss, f_st, f_tt}
// (true partial derivatives), instead of carrying a four-channel (x0, x1) jet through every row (3 x fewer vector instructions per row).
struct T2 {
    float f, s, t, ss, st, tt;
};
__device__ __forceinline__ T2 operator*(T2 a, T2 b) {
    return T2{a.f * b.f, a.f * b.s + a.s * b.f, a.f * b.t + a.t * b.f, a.f * b.ss + 2.0f * (a.s * b.s) + a.ss * b.f,
              a.f * b.st + a.s * b.t + a.t * b.s + a.st * b.f, a.f * b.tt + 2.0f * (a.t * b.t) + a.tt * b.f};
}
__device__ __forceinline__ T2 operator+(T2 a, float c) { return T2{a.f + c, a.s, a.t, a.ss, a.st, a.tt}; }
__device__ __forceinline__ T2 operator*(T2 a, float c) { return T2{a.f * c, a.s * c, a.t * c, a.ss * c, a.st * c, a.tt * c}; }
// g(a) from g, g', g'' at a.f
__device__ __forceinline__ T2 t2apply(T2 a, float g0, float g1, float g2) {
    return T2{g0, g1 * a.s, g1 * a.t, g1 * a.ss + g2 * (a.s * a.s), g1 * a.st + g2 * (a.s * a.t), g1 * a.tt + g2 * (a.t * a.t)};
}
__device__ __forceinline__ T2 t2rcp(T2 a) { const float g = 1.0f / a.f; return t2apply(a, g, -g * g, 2.0f * g * g * g); }
__device__ __forceinline__ T2 t2log(T2 a) { const float g1 = 1.0f / a.f; return t2apply(a, logf(a.f), g1, -g1 * g1); }
__device__ __forceinline__ T2 t2rsqrt(T2 a) { const float g = rsqrtf(a.f), q = 1.0f / a.f; return t2apply(a, g, -0.5f * g * q, 0.75f * g * q * q); }
// F(s(x), t(x)) as a jet in (x0, x1): chain rule through the jets of s and t (h = Laplacian / 2)
__device__ __forceinline__ J t2jet(T2 F, J s, J t) {
    return J{F.f, F.s * s.a + F.t * t.a, F.s * s.b + F.t * t.b,
             F.s * s.h + F.t * t.h + 0.5f * (F.ss * (s.a * s.a + s.b * s.b) + 2.0f * F.st * (s.a * t.a + s.b * t.b) + F.tt * (t.a * t.a + t.b * t.b))};
}
// r(x) = 1 / (2^x + 1) of a pre-activation triple (x, x', x'') in s -> (r, r', r'')
__device__ __forceinline__ void r_triple(float x0, float x1, float x2, float& v0, float& v1, float& v2) {
    const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x0) + 1.0f);
    const float r1 = -0.6931471805599453f * __builtin_fmaf(-r, r, r);
    const float k = __builtin_fmaf(1.3862943611198906f, r, -0.6931471805599453f);
    v0 = r;
    v1 = r1 * x1;
    v2 = r1 * __builtin_fmaf(k * x1, x1, x2);
}

// state arrays: st[(slot * 4 + c) * B + w]
__device__ __forceinline__ J st_load(const float* __restrict__ st, int slot, int64_t B, int64_t w) {
    const float* p = st + (int64_t)slot * 4 * B + w;
    return J{p[0], p[B], p[2 * B], p[3 * B]};
}
__device__ __forceinline__ void st_store(float* __restrict__ st, int slot, int64_t B, int64_t w, J x) {
    float* p = st + (int64_t)slot * 4 * B + w;
    p[0] = x.v; p[B] = x.a; p[2 * B] = x.b; p[3 * B] = x.h;
}

#ifndef WF_ETILE_WAVES
#define WF_ETILE_WAVES 4
#endif
#ifndef WF_ETILE_OCC
#define WF_ETILE_OCC 2   // workgroups per CU the register budget is sized for: 256 registers, two waves per SIMD (the per-lane store addresses spill: 23 reloads per tile)
#endif
constexpr int kCondWaves = WF_ETILE_WAVES;   // 4 waves per workgroup, WF_ETILE_OCC workgroups per CU (unbounded, the three channel chains take 324 registers: one wave per SIMD)
using O2 = NetOff<2, 1>;

// ---------------------------------------------------------------------------- conditioner of one net, jets on the matrix cores
// one power of two per (walker, channel) so that the largest of the column's 64 (or 32) entries lies in [0.5, 1)
__device__ __forceinline__ int col_exponent(float amax) {
    const float m = xhalf_max(amax);
    return m > 0.0f ? __builtin_amdgcn_frexp_expf(m) : 0;
}
// (x, x', x'') of one 32-unit block (3 channels x 16 registers) -> (r, r' x', r' x'' + r'' x'^2), in place
__device__ __forceinline__ void act_block(f32x16 (&x)[NCH]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x[0][r]) + 1.0f);
        const float r1 = -0.6931471805599453f * __builtin_fmaf(-rr, rr, rr);                      // r' = -ln2 r (1 - r)
        const float k = __builtin_fmaf(1.3862943611198906f, rr, -0.6931471805599453f);           // r'' / r' = -ln2 (1 - 2 r)
        const float x1 = x[1][r], x2 = x[2][r];
        x[0][r] = rr;
        x[1][r] = r1 * x1;
        x[2][r] = r1 * __builtin_fmaf(k * x1, x1, x2);                                            // r' x'' + r'' x'^2
    }
}
// two blocks of r jets -> B fragments of the next layer, derivative channels scaled by 2^-e[c] (e[0] = 0: r lies in (0, 1))
__device__ __forceinline__ void to_frags(const f32x16 (&blk0)[NCH], const f32x16 (&blk1)[NCH], Frag (&f)[NCH][2], int (&e)[NCH]) {
    e[0] = 0;
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
        float amax = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fmaxf(fabsf(blk0[c][r]), fabsf(blk1[c][r])));
        e[c] = col_exponent(amax);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float sc = __builtin_amdgcn_ldexpf(1.0f, -e[c]);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float r8[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) r8[jj] = (ob == 0 ? blk0[c][8 * s + jj] : blk1[c][8 * s + jj]) * sc;
                split8(r8, f[c][ob].hi[s], f[c][ob].lo[s]);
            }
    }
}
__device__ __forceinline__ void unscale(f32x16 (&acc)[NCH], const int (&e)[NCH]) {
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
        const float sc = __builtin_amdgcn_ldexpf(1.0f, e[c]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = acc[c][r] * sc;
    }
}
__device__ __forceinline__ void init_acc(f32x16 (&acc)[NCH], const float* bias16) {
    acc[0] = load16(bias16);
#pragma unroll
    for (int c = 1; c < NCH; ++c) acc[c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
}

// The conditioner of one net for one tile, in pieces (NBK = 32-row blocks per dimension: 1 for <= 32 bases, 2 for <= 64).
//   cond_hidden   the two hidden layers: B fragments (split fp16, derivative channels scaled by 2^-e) of the second hidden layer's activations
//   cond_out      one 32-row output block of dimension 1: Taylor triples (f, f', f'') in u_0 of the head's pre-activations, accumulator layout
//   prior_c       the prior head's c = (o * keep) @ ob_to_b as triples, one 32-row block of c at a time, + the sum of the raw outputs (sign)
template <int NBK>
__device__ __forceinline__ void cond_hidden(const float* net, float u0v, float u1v, int lane, Frag (&f)[NCH][2], int (&e)[NCH]) {
    using O = NetOff<2, NBK>;
    const int h = lane >> 5;
    // the conditioner's inputs: (u_0, u_1) values; the Taylor seed in u_0 is (u_0, 1, 0) (u_1 reaches no hidden unit: masked weights)
    const float in0[2] = {u0v, 1.0f}, in1[2] = {u1v, 0.0f};
    // ---- layer 1 (f32 MFMA, K = 2: the two coordinates), both 32-unit blocks; the second-derivative channel starts at zero
    f32x16 a0[NCH], a1[NCH];
    init_acc(a0, net + O::b0 + (0 * 2 + h) * 16);
    init_acc(a1, net + O::b0 + (1 * 2 + h) * 16);
    {
        const float w0 = net[O::W0 + 0 * 64 + lane], w1 = net[O::W0 + 1 * 64 + lane];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            a0[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, h ? in1[c] : in0[c], a0[c], 0, 0, 0);
            a1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, h ? in1[c] : in0[c], a1[c], 0, 0, 0);
        }
    }
    act_block(a0);
    act_block(a1);
    to_frags(a0, a1, f, e);
    // ---- layer 2
    const _Float16* W1h = reinterpret_cast<const _Float16*>(net + O::W1h);
    const _Float16* W1l = reinterpret_cast<const _Float16*>(net + O::W1l);
    init_acc(a0, net + O::b1 + (0 + h) * 16);
    init_acc(a1, net + O::b1 + (2 + h) * 16);
    dense64_block<NCH>(W1h, W1l, f, a0, lane);
    dense64_block<NCH>(W1h + 2048, W1l + 2048, f, a1, lane);
    unscale(a0, e);
    unscale(a1, e);
    act_block(a0);
    act_block(a1);
    to_frags(a0, a1, f, e);
}
// output block kb of dimension 1 (dimension 0 is table-driven: k_prepare_dim0)
template <int NBK>
__device__ __forceinline__ void cond_out(const float* net, const Frag (&f)[NCH][2], const int (&e)[NCH], int kb, int lane, f32x16 (&a0)[NCH]) {
    using O = NetOff<2, NBK>;
    const int h = lane >> 5;
    const _Float16* W2h = reinterpret_cast<const _Float16*>(net + O::W2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(net + O::W2l);
    init_acc(a0, net + O::b2 + ((1 * NBK + kb) * 2 + h) * 16);
    dense64_block<NCH>(W2h + kb * 2048, W2l + kb * 2048, f, a0, lane);
    unscale(a0, e);
}
// the prior head behind cond_out: of[ki][c] = fragments of w = o * keep (every (walker, channel) column of the 32 * NBK rows scaled by one power of
// two: the head is unbounded), eo[c] the exponents, s1 = sum of the raw outputs (model_factory.py:69: its sign, as in k_mfma)
template <int NBK>
__device__ __forceinline__ void prior_frags(f32x16 (&o)[NBK][NCH], const float* fkP, int lane, Frag (&of)[NBK][NCH], int (&eo)[NCH], float& s1) {
    const int h = lane >> 5;
    s1 = 0.0f;
    float amax[NCH] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
        const f32x16 keep = load16(fkP + (kb * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) s1 += o[kb][0][r];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[kb][c][r] = o[kb][c][r] * keep[r];
                amax[c] = fmaxf(amax[c], fabsf(o[kb][c][r]));
            }
    }
    s1 = xhalf_sum(s1);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        eo[c] = col_exponent(amax[c]);
        const float sc = __builtin_amdgcn_ldexpf(1.0f, -eo[c]);
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float r8[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) r8[jj] = o[kb][c][8 * s + jj] * sc;
                split8(r8, of[kb][c].hi[s], of[kb][c].lo[s]);
            }
    }
}
// block ko of c = w @ ob_to_b (obh: [ko][ki]{hi 1024, lo 1024} halves in f16-MFMA A order), three channels
template <int NBK>
__device__ __forceinline__ void prior_c_block(const _Float16* obh, const Frag (&of)[NBK][NCH], const int (&eo)[NCH], int ko, int lane, f32x16 (&cblk)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x16 acc = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ki = 0; ki < NBK; ++ki)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const _Float16* blk = obh + (size_t)(ko * NBK + ki) * 2048;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(blk + (s * 64 + lane) * 8);
                const f16x8 al = *reinterpret_cast<const f16x8*>(blk + 1024 + (s * 64 + lane) * 8);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, of[ki][c].hi[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, of[ki][c].lo[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, of[ki][c].hi[s], acc, 0, 0, 0);
            }
        const float sc = __builtin_amdgcn_ldexpf(1.0f, eo[c]);
#pragma unroll
        for (int r = 0; r < 16; ++r) cblk[c][r] = acc[r] * sc;
    }
}
// the whole conditioner for <= 32 bases (the launch-per-net path): head triples (PRIOR: of c) in a0, the sum of the raw outputs in s1
template <bool PRIOR>
__device__ __forceinline__ void cond_net(const float* net, const float* fkP, const _Float16* obh, float u0v, float u1v, int lane, f32x16 (&a0)[NCH], float& s1) {
    Frag f[NCH][2];
    int e[NCH];
    cond_hidden<1>(net, u0v, u1v, lane, f, e);
    if (!PRIOR) {
        cond_out<1>(net, f, e, 0, lane, a0);
    } else {
        f32x16 o[1][NCH];
        cond_out<1>(net, f, e, 0, lane, o[0]);
        Frag of[1][NCH];
        int eo[NCH];
        prior_frags<1>(o, fkP, lane, of, eo, s1);
        prior_c_block<1>(obh, of, eo, 0, lane, a0);
    }
}

template <bool PRIOR>
__global__ __launch_bounds__(kCondWaves * 64, WF_ETILE_OCC) void k_etile_cond(const MfmaDev mm, int net_index, const float* __restrict__ st, int64_t B,
                                                                float* __restrict__ oj, float* __restrict__ s1buf) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int next_tile;
    constexpr int kThreads = kCondWaves * 64;
    if (threadIdx.x == 0) next_tile = 0;
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    stage_floats<kThreads>(mm.image + (size_t)net_index * mm.net_floats, lds + mm.const_floats, mm.net_floats);
    __syncthreads();
    const float* net = lds + mm.const_floats;
    const float* fkP = lds + 32;
    const _Float16* obh = reinterpret_cast<const _Float16*>(lds + 64);
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (B + 31) >> 5;
    // this workgroup's tiles: blockIdx.x, blockIdx.x + gridDim.x, ... handed to its waves through a counter (oldest-wave-first arbitration)
    const int64_t my_tiles = n_tiles > (int64_t)blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    for (;;) {
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(&next_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= my_tiles) break;
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)q * gridDim.x;
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        const float u0v = st[wl], u1v = st[(int64_t)4 * B + wl];
        f32x16 a0[NCH];
        float s1 = 0.0f;
        cond_net<PRIOR>(net, fkP, obh, u0v, u1v, lane, a0, s1);
        if (PRIOR && valid && h == 0) s1buf[w] = s1;
        // ---- store: oj[tile][row][c][32 walkers] (one contiguous 12 KB block per tile), row = accumulator row of register r in lane half h
#ifdef WF_ABL_OJ   // ablation build (timing only): the head triples are computed, not stored
        if (valid && a0[0][0] == 12345.678f) {
#else
        if (valid) {
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                for (int c = 0; c < NCH; ++c) oj[(tile * (32 * NCH) + row * NCH + c) * 32 + j] = a0[c][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------- lane-per-walker stages
struct LerpN {
    int il, ir;
    float t;
};
__device__ __forceinline__ LerpN nlerp(float x, int n_mesh) {
    const Lerp L = make_lerp(x, n_mesh, 1.0f / (float)(n_mesh - 1), 1);
    return LerpN{L.il, L.ir, L.t};
}

// BoxTransformLayer, mean type, two particles (made.py:156-183) as jets of (x0, x1)
__global__ void k_etile_box(const float* __restrict__ xg, int64_t B, float L, float* __restrict__ st) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float tol = 1e-7f;
    const J x0 = J{xg[b * 2], 1.0f, 0.0f, 0.0f}, x1 = J{xg[b * 2 + 1], 0.0f, 1.0f, 0.0f};
    const J mean = (x0 + x1) * 0.5f;
    const J l = mean - x0, wd = x1 - x0;
    J ld = jc(0.0f);
    const J space = jc(2 * L);
    const J diff = x1 - x0;
    const J u0 = diff * jrcp(space + tol);
    ld = ld - jlog(space + tol);
    const J den = (jc(2 * L) - wd) + tol;
    const J u1 = ((mean + L) - l) * jrcp(den);
    ld = ld - jlog(den);
    st_store(st, 0, B, b, u0);
    st_store(st, 1, B, b, u1);
    st_store(st, 2, B, b, ld);
}

// y_1 = N_0 / Q and log(dy_1 + 1e-7) from the row sums of one flow head (see T2): N_k(s, t) = V_k(s, t) / S(s) + reg R_k(t), k = 0 (value) and 1
// (derivative in t), Q(s) = Qv(s) / S(s) + reg G
__device__ __forceinline__ void flow_head_finish(const float (&S)[3], const float (&Qv)[3], const float (&R)[4], float G, const float (&V0)[4],
                                                 const float (&V1)[3], const float (&V2)[2], float reg, J u0, J u1, J& y1, J& ld) {
    const T2 iS = t2rcp(T2{S[0], S[1], 0.0f, S[2], 0.0f, 0.0f});
    const T2 Q = T2{Qv[0], Qv[1], 0.0f, Qv[2], 0.0f, 0.0f} * iS + reg * G;
    const T2 rQ = t2rcp(Q);
    // numerator k: partials of V_k are V[a][k + b]
    T2 N0 = T2{V0[0], V1[0], V0[1], V2[0], V1[1], V0[2]} * iS;
    N0.f += reg * R[0]; N0.t += reg * R[1]; N0.tt += reg * R[2];
    T2 N1 = T2{V0[1], V1[1], V0[2], V2[1], V1[2], V0[3]} * iS;
    N1.f += reg * R[1]; N1.t += reg * R[2]; N1.tt += reg * R[3];
    y1 = t2jet(N0 * rQ, u0, u1);
    ld = ld + t2jet(t2log(N1 * rQ + 1e-7f), u0, u1);
}

// One IMADE layer behind its conditioner (made.py:66-81) + Reverse: dimension 0 from the composite table, dimension 1 from the head jets
__global__ __launch_bounds__(256) void k_etile_flow(const float4_t* __restrict__ comp /* this net: [n_mesh] {Y, Y', Y'', Y'''} */,
                                                    const float* __restrict__ tabI /* [n_mesh][8 row chunks][4 orders][4 rows] */, const float* __restrict__ gI, int nb,
                                                    int n_mesh, float reg, const float* __restrict__ oj, int64_t B, float* __restrict__ st) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const J u0 = st_load(st, 0, B, b), u1 = st_load(st, 1, B, b);
    J ld = st_load(st, 2, B, b);
    // ---- dimension 0
    J y0;
    {
        const LerpN L = nlerp(u0.v, n_mesh);
        const float4_t ca = comp[L.il], cb = comp[L.ir];
        const float t0 = __builtin_fmaf(cb.x - ca.x, L.t, ca.x), t1 = __builtin_fmaf(cb.y - ca.y, L.t, ca.y);
        const float t2 = __builtin_fmaf(cb.z - ca.z, L.t, ca.z), t3 = __builtin_fmaf(cb.w - ca.w, L.t, ca.w);
        y0 = jlift(t0, t1, t2, u0);
        ld = ld + jlog(jlift(t1, t2, t3, u0) + 1e-7f);
    }
    // ---- dimension 1: c_j = g_j (v_j / S0 + reg) / Q (calculate_bijection_params, + reg, remove_bias, boundary map: wf_model.cpp).
    // One pass over the rows: with V_k = sum_j v_j g_j B^(k)_j, R_k = sum_j g_j B^(k)_j, Qv = sum_j v_j g_j, G = sum_j g_j, S0 = sum_j v_j the
    // numerators are N_k = V_k / S0 + reg R_k and the normaliser Q = Qv / S0 + reg G.  Four rows per step: the lane's table rows come as
    // 16-byte loads (8 per step: 4 orders x the two mesh rows; rows >= nb are zero padding).
    const LerpN L = nlerp(u1.v, n_mesh);
    const int* bnd = reinterpret_cast<const int*>(tabI + (size_t)n_mesh * 128);   // [8 chunks][lo, hi] behind the table (wf_model.cpp: upload_chunked)
    // sums over the rows (see T2): S^(a) = sum v_j^(a), Qv^(a) = sum g_j v_j^(a), V[a][k] = sum v_j^(a) g_j T_j^(k) for a + k <= 3 (k <= 3 - a ... the
    // nine pairs the two numerators need), R[k] = sum g_j T_j^(k), G = sum g_j
    float S[3] = {0.0f, 0.0f, 0.0f}, Qv[3] = {0.0f, 0.0f, 0.0f}, R[4] = {0.0f, 0.0f, 0.0f, 0.0f}, G = 0.0f;
    float V0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, V1[3] = {0.0f, 0.0f, 0.0f}, V2[2] = {0.0f, 0.0f};   // V[a][k]: a = 0: k 0..3, a = 1: k 0..2, a = 2: k 0..1
    for (int j0 = 0; j0 < nb; j0 += 4) {
        // the chunk at the mesh index clamped to its support: the same bits, and the walkers outside the support read two shared lines
        const int lo = bnd[j0 >> 1], hi = bnd[(j0 >> 1) + 1];
        const float4_t* rl = reinterpret_cast<const float4_t*>(tabI + (size_t)min(max(L.il, lo), hi) * 128);   // [8 chunks][4 orders] float4
        const float4_t* rr = reinterpret_cast<const float4_t*>(tabI + (size_t)min(max(L.ir, lo), hi) * 128);
        float4_t ta[4], tb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ta[k] = rl[j0 + k];     // (chunk j0 / 4) * 4 + k
            tb[k] = rr[j0 + k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jr_ = j0 + q;
            if (jr_ >= nb) break;
            const float* p = oj + ((b >> 5) * (32 * NCH) + jr_ * NCH) * 32 + (b & 31);     // [tile][row][channel][32 walkers]
            float v0, v1, v2;
            r_triple(p[0], p[32], p[64], v0, v1, v2);       // the head's weight and its first two derivatives in u_0
            const float g = gI[jr_];
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = q == 0 ? ta[k].x : (q == 1 ? ta[k].y : (q == 2 ? ta[k].z : ta[k].w));
                const float bb = q == 0 ? tb[k].x : (q == 1 ? tb[k].y : (q == 2 ? tb[k].z : tb[k].w));
                t[k] = __builtin_fmaf(bb - a, L.t, a) * g;   // g_j folded into the row
            }
            S[0] += v0; S[1] += v1; S[2] += v2;
            Qv[0] = __builtin_fmaf(v0, g, Qv[0]); Qv[1] = __builtin_fmaf(v1, g, Qv[1]); Qv[2] = __builtin_fmaf(v2, g, Qv[2]);
#pragma unroll
            for (int k = 0; k < 4; ++k) { V0[k] = __builtin_fmaf(v0, t[k], V0[k]); R[k] += t[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) V1[k] = __builtin_fmaf(v1, t[k], V1[k]);
#pragma unroll
            for (int k = 0; k < 2; ++k) V2[k] = __builtin_fmaf(v2, t[k], V2[k]);
            G += g;
        }
    }
    J y1;
    flow_head_finish(S, Qv, R, G, V0, V1, V2, reg, u0, u1, y1, ld);
    st_store(st, 0, B, b, y1);   // Reverse (bijections.py:337-340)
    st_store(st, 1, B, b, y0);
    st_store(st, 2, B, b, ld);
}

// Waveflow prior (wavefunctions.py:54-71) + H psi (physics.py:60-93)
__global__ __launch_bounds__(256) void k_etile_prior(const float4_t* __restrict__ comp /* prior: {P, P', P''} with sign and norm */,
                                                     const float* __restrict__ tabP /* orthogonal B, [n_mesh][8][4][4] like tabI */, int nb, int n_mesh,
                                                     unsigned constrained_mask, const float* __restrict__ oj, const float* __restrict__ s1buf,
                                                     const float* __restrict__ st, const float* __restrict__ xg, int64_t B, const Protons pr,
                                                     float* __restrict__ hpsi, float* __restrict__ psi_out, float* __restrict__ lap_out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const J u0 = st_load(st, 0, B, b), u1 = st_load(st, 1, B, b), ld = st_load(st, 2, B, b);
    // the spline sees the clipped coordinate (:45): outside [0, 1] it is a constant
    const J uc0 = (u0.v < 0.0f) ? jc(0.0f) : (u0.v > 1.0f ? jc(1.0f) : u0);
    const J uc1 = (u1.v < 0.0f) ? jc(0.0f) : (u1.v > 1.0f ? jc(1.0f) : u1);
    J val0;
    {
        const LerpN L = nlerp(uc0.v, n_mesh);
        const float4_t ca = comp[L.il], cb = comp[L.ir];
        val0 = jlift(__builtin_fmaf(cb.x - ca.x, L.t, ca.x), __builtin_fmaf(cb.y - ca.y, L.t, ca.y), __builtin_fmaf(cb.z - ca.z, L.t, ca.z), uc0);
    }
    const LerpN L = nlerp(uc1.v, n_mesh);
    const int* bnd = reinterpret_cast<const int*>(tabP + (size_t)n_mesh * 128);
    // sums over the rows (see T2): D[a][k] = sum c_i^(a)(s) B_i^(k)(t), a + k <= 2; |c|^2 and its first two derivatives in s from cc, cc', c'c', cc''
    float D0[3] = {0.0f, 0.0f, 0.0f}, D1[2] = {0.0f, 0.0f}, D2 = 0.0f, cc = 0.0f, cc1 = 0.0f, c1c1 = 0.0f, cc2 = 0.0f;
    for (int i0 = 0; i0 < nb; i0 += 4) {
        const int lo = bnd[i0 >> 1], hi = bnd[(i0 >> 1) + 1];
        const float4_t* rl = reinterpret_cast<const float4_t*>(tabP + (size_t)min(max(L.il, lo), hi) * 128);
        const float4_t* rr = reinterpret_cast<const float4_t*>(tabP + (size_t)min(max(L.ir, lo), hi) * 128);
        float4_t ta[3], tb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ta[k] = rl[i0 + k];
            tb[k] = rr[i0 + k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            if (i >= nb) break;
            const float* p = oj + ((b >> 5) * (32 * NCH) + i * NCH) * 32 + (b & 31);
            const float c0 = p[0], c1 = p[32], c2 = p[64];        // c_i and its derivatives in the unclipped u_0 (wavefunctions.py:40)
            float t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float a = q == 0 ? ta[k].x : (q == 1 ? ta[k].y : (q == 2 ? ta[k].z : ta[k].w));
                const float bb = q == 0 ? tb[k].x : (q == 1 ? tb[k].y : (q == 2 ? tb[k].z : tb[k].w));
                t[k] = __builtin_fmaf(bb - a, L.t, a);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) D0[k] = __builtin_fmaf(c0, t[k], D0[k]);
            D1[0] = __builtin_fmaf(c1, t[0], D1[0]); D1[1] = __builtin_fmaf(c1, t[1], D1[1]);
            D2 = __builtin_fmaf(c2, t[0], D2);
            cc = __builtin_fmaf(c0, c0, cc); cc1 = __builtin_fmaf(c0, c1, cc1); c1c1 = __builtin_fmaf(c1, c1, c1c1); cc2 = __builtin_fmaf(c0, c2, cc2);
        }
    }
    const float sgn = s1buf[b] < 0.0f ? -1.0f : 1.0f;
    const T2 N2 = T2{cc, 2.0f * cc1, 0.0f, 2.0f * (c1c1 + cc2), 0.0f, 0.0f};
    const T2 dotp = T2{D0[0], D1[0], D0[1], D2, D1[1], D0[2]};
    const J val1 = t2jet(dotp * t2rsqrt(N2), u0, uc1) * sgn;
    const float sc0 = (constrained_mask & 1u) ? 0.70710678118654752f : 1.0f, sc1 = (constrained_mask & 2u) ? 0.70710678118654752f : 1.0f;
    const J psi = ((val0 * sc0) * (val1 * sc1)) * jexp_half(ld);
    const float lap = 2.0f * psi.h;
    float V = 0.0f;   // physics.py:60-76
    for (int p = 0; p < pr.n; ++p)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float r = pr.pos[p] - xg[b * 2 + d];
            V -= 1.0f / sqrtf(1.0f + r * r);
        }
    {
        const float r = xg[b * 2 + 1] - xg[b * 2];
        V += 1.0f / sqrtf(1.0f + r * r);
    }
    hpsi[b] = -0.5f * lap + V * psi.v;
    if (psi_out) psi_out[b] = psi.v;
    if (lap_out) lap_out[b] = lap;
}

// ---------------------------------------------------------------------------- H psi in ONE kernel (every net resident in LDS)
// The conditioner's head triples stay in the accumulator registers: lane (walker j, half h) holds registers 4q .. 4q + 3 = rows 8q + 4h .. + 3 =
// row CHUNK 2q + h of the regrouped tables ([mesh][8 chunks][4 orders][4 rows]: one 64-byte segment per chunk and lerp end).  The head runs on
// those 16 rows per lane with the separable row sums of T2 (scalars, no jet per row); the two lane halves are combined with one
// v_permlane32_swap per sum; quotients, logarithms and the change to (x0, x1) jets once per walker.  No exchange buffer, no state in HBM:
// 8 B per walker in, 4 .. 12 B out.  One persistent workgroup of 8 waves per CU (two per SIMD, 256 registers), tiles from an LDS counter.
constexpr int kFusedWaves = 8;

__device__ __forceinline__ void box_mean2(float x0v, float x1v, float L, J& u0, J& u1, J& ld) {   // (k_etile_box)
    const float tol = 1e-7f;
    const J x0 = J{x0v, 1.0f, 0.0f, 0.0f}, x1 = J{x1v, 0.0f, 1.0f, 0.0f};
    const J mean = (x0 + x1) * 0.5f;
    const J l = mean - x0, wd = x1 - x0;
    const J space = jc(2 * L);
    const J diff = x1 - x0;
    u0 = diff * jrcp(space + tol);
    ld = jc(0.0f) - jlog(space + tol);
    const J den = (jc(2 * L) - wd) + tol;
    u1 = ((mean + L) - l) * jrcp(den);
    ld = ld - jlog(den);
}
// chunk 2q + h of the lane at both lerp ends, every order; bnd: [8 chunks][lo, hi] support bounds (the chunk at the clamped index holds the same bits)
template <int NO>
__device__ __forceinline__ void chunk_rows(const float* __restrict__ tab, int mesh_stride, const int* bnd, const LerpN& L, int ch, float4_t (&ta)[NO], float4_t (&tb)[NO]) {
    const int lo = bnd[2 * ch], hi = bnd[2 * ch + 1];
    const float4_t* rl = reinterpret_cast<const float4_t*>(tab + (size_t)min(max(L.il, lo), hi) * mesh_stride) + ch * 4;
    const float4_t* rr = reinterpret_cast<const float4_t*>(tab + (size_t)min(max(L.ir, lo), hi) * mesh_stride) + ch * 4;
#pragma unroll
    for (int k = 0; k < NO; ++k) {
        ta[k] = rl[k];
        tb[k] = rr[k];
    }
}
// row sums of one flow head (see T2) over the lane's 16 rows of block kb
struct FlowSums {
    float S[3], Qv[3], R[4], V0[4], V1[3], V2[2];
};
__device__ __forceinline__ void flow_rows(FlowSums& a, const f32x16 (&o)[NCH], const f32x16& g16, const float* __restrict__ tabI, int mesh_stride, const int* bnd,
                                          const LerpN& L, int kb, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4_t ta[4], tb[4];
        chunk_rows<4>(tabI, mesh_stride, bnd, L, 8 * kb + 2 * q + h, ta, tb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            float v0, v1, v2;
            r_triple(o[0][r], o[1][r], o[2][r], v0, v1, v2);
            const float g = g16[r];
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_fmaf(tb[k][e] - ta[k][e], L.t, ta[k][e]) * g;
            a.S[0] += v0; a.S[1] += v1; a.S[2] += v2;
            a.Qv[0] = __builtin_fmaf(v0, g, a.Qv[0]); a.Qv[1] = __builtin_fmaf(v1, g, a.Qv[1]); a.Qv[2] = __builtin_fmaf(v2, g, a.Qv[2]);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.V0[k] = __builtin_fmaf(v0, t[k], a.V0[k]); a.R[k] += t[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) a.V1[k] = __builtin_fmaf(v1, t[k], a.V1[k]);
#pragma unroll
            for (int k = 0; k < 2; ++k) a.V2[k] = __builtin_fmaf(v2, t[k], a.V2[k]);
        }
    }
}
struct PriorSums {
    float D0[3], D1[2], D2, cc, cc1, c1c1, cc2;
};
__device__ __forceinline__ void prior_rows(PriorSums& a, const f32x16 (&c)[NCH], const float* __restrict__ tabP, int mesh_stride, const int* bnd, const LerpN& L,
                                           int kb, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4_t ta[3], tb[3];
        chunk_rows<3>(tabP, mesh_stride, bnd, L, 8 * kb + 2 * q + h, ta, tb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            const float c0 = c[0][r], c1 = c[1][r], c2 = c[2][r];
            float t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = __builtin_fmaf(tb[k][e] - ta[k][e], L.t, ta[k][e]);
#pragma unroll
            for (int k = 0; k < 3; ++k) a.D0[k] = __builtin_fmaf(c0, t[k], a.D0[k]);
            a.D1[0] = __builtin_fmaf(c1, t[0], a.D1[0]); a.D1[1] = __builtin_fmaf(c1, t[1], a.D1[1]);
            a.D2 = __builtin_fmaf(c2, t[0], a.D2);
            a.cc = __builtin_fmaf(c0, c0, a.cc); a.cc1 = __builtin_fmaf(c0, c1, a.cc1); a.c1c1 = __builtin_fmaf(c1, c1, a.c1c1); a.cc2 = __builtin_fmaf(c0, c2, a.cc2);
        }
    }
}

template <int NBK>
__global__ __launch_bounds__(kFusedWaves * 64) void k_efused(const MfmaDev mm, const float* __restrict__ tabI, const float* __restrict__ tabP,
                                                             const float* __restrict__ xg, int64_t B, const Protons pr, float* __restrict__ hpsi,
                                                             float* __restrict__ psi_out, float* __restrict__ lap_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int next_tile;
    __shared__ int bnd_s[32 * NBK];   // support bounds of the table chunks: [I: 8 NBK][lo, hi], [prior: 8 NBK][lo, hi]
    constexpr int kThreads = kFusedWaves * 64;
    constexpr int kMeshStride = 128 * NBK;   // floats per mesh point of the regrouped tables: [8 NBK chunks][4 orders][4 rows]
    if (threadIdx.x == 0) next_tile = 0;
    if (threadIdx.x < 16 * NBK) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabI + (size_t)mm.n_mesh * kMeshStride)[threadIdx.x];
    else if (threadIdx.x < 32 * NBK) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabP + (size_t)mm.n_mesh * kMeshStride)[threadIdx.x - 16 * NBK];
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    stage_floats<kThreads>(mm.image, lds + mm.const_floats, mm.net_floats * mm.n_nets);
    __syncthreads();
    const float* fkI = lds;
    const float* fkP = lds + 32 * NBK;
    const _Float16* obh = reinterpret_cast<const _Float16*>(lds + 64 * NBK);
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int n_mesh = mm.n_mesh;
    const int64_t n_tiles = (B + 31) >> 5;
    const int64_t my_tiles = n_tiles > (int64_t)blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    for (;;) {
        int q_ = 0;
        if (lane == 0) q_ = __hip_atomic_fetch_add(&next_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q_ = __builtin_amdgcn_readfirstlane(q_);
        if (q_ >= my_tiles) break;
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)q_ * gridDim.x;
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        const float x0v = xg[wl * 2], x1v = xg[wl * 2 + 1];
        J u0, u1, ld;
        box_mean2(x0v, x1v, mm.box_L, u0, u1, ld);
        // ---- flow layers (made.py:66-81 + Reverse)
        for (int l = 0; l < mm.n_layers; ++l) {
            const float* net = lds + mm.const_floats + (size_t)l * mm.net_floats;
            Frag f[NCH][2];
            int e[NCH];
            cond_hidden<NBK>(net, u0.v, u1.v, lane, f, e);
            // dimension 0: composite table of the net, all four orders
            J y0;
            {
                const LerpN L0 = nlerp(u0.v, n_mesh);
                const float4_t* comp = mm.comp + (size_t)l * n_mesh;
                const float4_t ca = comp[L0.il], cb = comp[L0.ir];
                const float t0 = __builtin_fmaf(cb.x - ca.x, L0.t, ca.x), t1 = __builtin_fmaf(cb.y - ca.y, L0.t, ca.y);
                const float t2 = __builtin_fmaf(cb.z - ca.z, L0.t, ca.z), t3 = __builtin_fmaf(cb.w - ca.w, L0.t, ca.w);
                y0 = jlift(t0, t1, t2, u0);
                ld = ld + jlog(jlift(t1, t2, t3, u0) + 1e-7f);
            }
            // dimension 1: the lane's 16 rows of every 32-row block
            const LerpN L = nlerp(u1.v, n_mesh);
            FlowSums a = {};
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb) {
                f32x16 o[NCH];
                cond_out<NBK>(net, f, e, kb, lane, o);
                flow_rows(a, o, load16(fkI + (kb * 2 + h) * 16), tabI, kMeshStride, bnd_s, L, kb, h);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { a.S[k] = xhalf_sum(a.S[k]); a.Qv[k] = xhalf_sum(a.Qv[k]); a.V1[k] = xhalf_sum(a.V1[k]); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.R[k] = xhalf_sum(a.R[k]); a.V0[k] = xhalf_sum(a.V0[k]); }
#pragma unroll
            for (int k = 0; k < 2; ++k) a.V2[k] = xhalf_sum(a.V2[k]);
            J y1;
            flow_head_finish(a.S, a.Qv, a.R, mm.F_I, a.V0, a.V1, a.V2, mm.i_reg, u0, u1, y1, ld);
            u0 = y1;   // Reverse (bijections.py:337-340)
            u1 = y0;
        }
        // ---- Waveflow prior (wavefunctions.py:54-71)
        J psi;
        {
            const float* net = lds + mm.const_floats + (size_t)mm.n_layers * mm.net_floats;
            float s1 = 0.0f;
            Frag of[NBK][NCH];
            int eo[NCH];
            {
                Frag f[NCH][2];
                int e[NCH];
                cond_hidden<NBK>(net, u0.v, u1.v, lane, f, e);   // (the conditioner sees the unclipped u_0, wavefunctions.py:40)
                f32x16 o[NBK][NCH];
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) cond_out<NBK>(net, f, e, kb, lane, o[kb]);
                prior_frags<NBK>(o, fkP, lane, of, eo, s1);
            }
            const J uc0 = (u0.v < 0.0f) ? jc(0.0f) : (u0.v > 1.0f ? jc(1.0f) : u0);   // the spline sees the clipped coordinate (:45)
            const J uc1 = (u1.v < 0.0f) ? jc(0.0f) : (u1.v > 1.0f ? jc(1.0f) : u1);
            J val0;
            {
                const LerpN L0 = nlerp(uc0.v, n_mesh);
                const float4_t* comp = mm.comp + (size_t)mm.n_layers * n_mesh;
                const float4_t ca = comp[L0.il], cb = comp[L0.ir];
                val0 = jlift(__builtin_fmaf(cb.x - ca.x, L0.t, ca.x), __builtin_fmaf(cb.y - ca.y, L0.t, ca.y), __builtin_fmaf(cb.z - ca.z, L0.t, ca.z), uc0);
            }
            const LerpN L = nlerp(uc1.v, n_mesh);
            PriorSums a = {};
#pragma unroll
            for (int ko = 0; ko < NBK; ++ko) {
                f32x16 cblk[NCH];
                prior_c_block<NBK>(obh, of, eo, ko, lane, cblk);
                prior_rows(a, cblk, tabP, kMeshStride, bnd_s + 16 * NBK, L, ko, h);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) a.D0[k] = xhalf_sum(a.D0[k]);
            a.D1[0] = xhalf_sum(a.D1[0]); a.D1[1] = xhalf_sum(a.D1[1]); a.D2 = xhalf_sum(a.D2);
            a.cc = xhalf_sum(a.cc); a.cc1 = xhalf_sum(a.cc1); a.c1c1 = xhalf_sum(a.c1c1); a.cc2 = xhalf_sum(a.cc2);
            const float sgn = s1 < 0.0f ? -1.0f : 1.0f;
            const T2 N2 = T2{a.cc, 2.0f * a.cc1, 0.0f, 2.0f * (a.c1c1 + a.cc2), 0.0f, 0.0f};
            const T2 dotp = T2{a.D0[0], a.D1[0], a.D0[1], a.D2, a.D1[1], a.D0[2]};
            const J val1 = t2jet(dotp * t2rsqrt(N2), u0, uc1) * sgn;
            const float sc0 = (mm.constrained_mask & 1u) ? 0.70710678118654752f : 1.0f, sc1 = (mm.constrained_mask & 2u) ? 0.70710678118654752f : 1.0f;
            psi = ((val0 * sc0) * (val1 * sc1)) * jexp_half(ld);
        }
        if (valid && h == 0) {
            const float lap = 2.0f * psi.h;
            float V = 0.0f;   // physics.py:60-76
            for (int p = 0; p < pr.n; ++p) {
                const float r0 = pr.pos[p] - x0v, r1 = pr.pos[p] - x1v;
                V -= 1.0f / sqrtf(1.0f + r0 * r0);
                V -= 1.0f / sqrtf(1.0f + r1 * r1);
            }
            {
                const float r = x1v - x0v;
                V += 1.0f / sqrtf(1.0f + r * r);
            }
            hpsi[w] = -0.5f * lap + V * psi.v;
            if (psi_out) psi_out[w] = psi.v;
            if (lap_out) lap_out[w] = lap;
        }
    }
}

int check() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

}  // namespace

bool energy_tile_fused(const MfmaDev* mdev) {
    const char* e = getenv("WF_ENERGY_FUSED");
    // (two row blocks per dimension: the static LDS of k_efused<2> -- 256 B of chunk bounds -- has to fit beside the resident nets too)
    const bool fits = (mdev->const_floats + mdev->net_floats * mdev->n_nets) * 4 + 512 <= 160 * 1024;
    return !mdev->staged && fits && !(e && atoi(e) == 0);
}

// workspace: state (12 floats), head triples (96 floats), the sign sum (1 float) per walker
int64_t energy_tile_floats(int64_t B) { return B * (12 + 1) + ((B + 31) / 32) * 32 * (32 * NCH); }   // state, s1, head triples of whole tiles

// mdev: the model's MFMA description (resident or not: one net is staged per launch); md: ModelDev on the host (spline sizes, masks)
int launch_energy_tile(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* fk_nat, const float* x, int64_t B,
                       const Protons& pr, float* hpsi, float* psi, float* lap, float* ws, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return WF_OK;
    // every net resident in LDS (the shipped shapes): the whole of H psi in one launch, nothing through HBM but the walkers and the results.
    // WF_ENERGY_FUSED=0 (read per call) keeps the launch-per-net path below (A/B tests; models whose nets do not fit together take it anyway).
    {
        if (energy_tile_fused(mdev)) {
            const int lds_all = (mdev->const_floats + mdev->net_floats * mdev->n_nets) * (int)sizeof(float);
            const int64_t n_tiles = (B + 31) / 32;
            const unsigned blocks = (unsigned)std::min<int64_t>((n_tiles + kFusedWaves - 1) / kFusedWaves, 256);
            if (mdev->nbk == 1) {
                static DynLdsSlots cfg1{};
                if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_efused<1>), lds_all, &cfg1)) return rc;
                hipLaunchKernelGGL(k_efused<1>, dim3(blocks), dim3(kFusedWaves * 64), lds_all, s, *mdev, tabI4, tabP4, x, B, pr, hpsi, psi, lap);
            } else {
                static DynLdsSlots cfg2{};
                if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_efused<2>), lds_all, &cfg2)) return rc;
                hipLaunchKernelGGL(k_efused<2>, dim3(blocks), dim3(kFusedWaves * 64), lds_all, s, *mdev, tabI4, tabP4, x, B, pr, hpsi, psi, lap);
            }
            return check();
        }
    }
    float* st = ws;
    float* s1 = st + 12 * B;
    float* oj = s1 + B;
    const unsigned lane_blocks = (unsigned)((B + 255) / 256);
    const int lds_bytes = (mdev->const_floats + mdev->net_floats) * (int)sizeof(float);
    static DynLdsSlots cfg_flow{}, cfg_prior{};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_etile_cond<false>), lds_bytes, &cfg_flow)) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_etile_cond<true>), lds_bytes, &cfg_prior)) return rc;
    const int64_t n_tiles = (B + 31) / 32;
    const unsigned cond_blocks = (unsigned)std::min<int64_t>((n_tiles + kCondWaves - 1) / kCondWaves, 256 * 4);
    hipLaunchKernelGGL(k_etile_box, dim3(lane_blocks), dim3(256), 0, s, x, B, md.box_L, st);
    for (int l = 0; l < md.n_layers; ++l) {
        hipLaunchKernelGGL(k_etile_cond<false>, dim3(cond_blocks), dim3(kCondWaves * 64), lds_bytes, s, *mdev, l, (const float*)st, B, oj, s1);
        hipLaunchKernelGGL(k_etile_flow, dim3(lane_blocks), dim3(256), 0, s, mdev->comp + (size_t)l * mdev->n_mesh, tabI4, fk_nat, md.isp.nb,
                           md.isp.n_mesh, md.i_reg, (const float*)oj, B, st);
    }
    hipLaunchKernelGGL(k_etile_cond<true>, dim3(cond_blocks), dim3(kCondWaves * 64), lds_bytes, s, *mdev, md.n_layers, (const float*)st, B, oj, s1);
    hipLaunchKernelGGL(k_etile_prior, dim3(lane_blocks), dim3(256), 0, s, mdev->comp + (size_t)md.n_layers * mdev->n_mesh, tabP4, md.psp.nb,
                       md.psp.n_mesh, md.constrained_mask, (const float*)oj, (const float*)s1, (const float*)st, x, B, pr, hpsi, psi, lap);
    return check();
}

}  // namespace wf
