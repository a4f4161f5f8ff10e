// wf_etile_common.h -- jet / Taylor algebra, conditioner pieces and head row sums shared by the matrix-core energy kernels
// (wf_kernels_etile.hip: two particles; wf_kernels_edir.hip: D particles, one coordinate direction at a time).  Everything sits in an
// anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "wf_mfma_impl.h"

// The jet / Taylor algebra of these files is checked against oracles by tolerance, not by operation order: multiply-add pairs may fuse (the build's
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


// ---------------------------------------------------------------------------- conditioner of one net, jets on the matrix cores
// one power of two per (walker, channel) so that the largest of the column's 64 (or 32) entries lies in [0.5, 1)
__device__ __forceinline__ int col_exponent(float amax) {
    const float m = xhalf_max(amax);
    return m > 0.0f ? __builtin_amdgcn_frexp_expf(m) : 0;
}
// (x, x', x'') of one 32-unit block (3 channels x 16 registers) -> (r, r' x', r' x'' + r'' x'^2), in place.  CH = 1: the value channel alone (the staged
// sampler's conditioner launches, which read nothing else)
template <int CH = NCH>
__device__ __forceinline__ void act_block(f32x16 (&x)[CH]) {
    static_assert(CH == 1 || CH == NCH, "value channel alone, or the Taylor triple");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x[0][r]) + 1.0f);
        x[0][r] = rr;
        if (CH == NCH) {
            const float r1 = -0.6931471805599453f * __builtin_fmaf(-rr, rr, rr);                      // r' = -ln2 r (1 - r)
            const float k = __builtin_fmaf(1.3862943611198906f, rr, -0.6931471805599453f);           // r'' / r' = -ln2 (1 - 2 r)
            const float x1 = x[CH - 2][r], x2 = x[CH - 1][r];
            x[CH - 2][r] = r1 * x1;
            x[CH - 1][r] = r1 * __builtin_fmaf(k * x1, x1, x2);                                        // r' x'' + r'' x'^2
        }
    }
}
// two blocks of r jets -> B fragments of the next layer, derivative channels scaled by 2^-e[c] (e[0] = 0: r lies in (0, 1))
template <int CH = NCH>
__device__ __forceinline__ void to_frags(const f32x16 (&blk0)[CH], const f32x16 (&blk1)[CH], Frag (&f)[CH][2], int (&e)[CH]) {
    e[0] = 0;
#pragma unroll
    for (int c = 1; c < CH; ++c) {
        float amax = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fmaxf(fabsf(blk0[c][r]), fabsf(blk1[c][r])));
        e[c] = col_exponent(amax);
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
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
template <int CH = NCH>
__device__ __forceinline__ void unscale(f32x16 (&acc)[CH], const int (&e)[CH]) {
#pragma unroll
    for (int c = 1; c < CH; ++c) {
        const float sc = __builtin_amdgcn_ldexpf(1.0f, e[c]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = acc[c][r] * sc;
    }
}
template <int CH = NCH>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[CH], const float* bias16) {
    acc[0] = load16(bias16);
#pragma unroll
    for (int c = 1; c < CH; ++c) acc[c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
}


struct LerpN {
    int il, ir;
    float t;
};
__device__ __forceinline__ LerpN nlerp(float x, int n_mesh) {
    const Lerp L = make_lerp(x, n_mesh, 1.0f / (float)(n_mesh - 1), 1);
    return LerpN{L.il, L.ir, L.t};
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


// the prior head behind cond_out: of[ki][c] = fragments of w = o * keep (every (walker, channel) column of the 32 * NBK rows scaled by one power of
// two: the head is unbounded), eo[c] the exponents, s1 = sum of the raw outputs (model_factory.py:69: its sign, as in k_mfma)
template <int NBK, int CH = NCH>
__device__ __forceinline__ void prior_frags(f32x16 (&o)[NBK][CH], const float* fkP, int lane, Frag (&of)[NBK][CH], int (&eo)[CH], float& s1,
                                            float* sder = nullptr /* [2]: the sums of the derivative channels (a boundary map with a constant term needs them) */) {
    const int h = lane >> 5;
    s1 = 0.0f;
    float amax[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) amax[c] = 0.0f;
    if (CH == NCH && sder) {
        float d1 = 0.0f, d2 = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { d1 += o[kb][CH - 2][r]; d2 += o[kb][CH - 1][r]; }
        sder[0] = xhalf_sum(d1);
        sder[1] = xhalf_sum(d2);
    }
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
        const f32x16 keep = load16(fkP + (kb * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) s1 += o[kb][0][r];
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[kb][c][r] = o[kb][c][r] * keep[r];
                amax[c] = fmaxf(amax[c], fabsf(o[kb][c][r]));
            }
    }
    s1 = xhalf_sum(s1);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
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
template <int NBK, int CH = NCH>
__device__ __forceinline__ void prior_c_block(const _Float16* obh, const Frag (&of)[NBK][CH], const int (&eo)[CH], int ko, int lane, f32x16 (&cblk)[CH]) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
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

}  // namespace
}  // namespace wf
