// wf_ring.h -- the coefficient rings of the gradient / local-energy kernels and the ring lift of the table lerp.
//
// R3 = IR[t]/t^3 holds the Taylor coefficients a0 + a1 t + a2 t^2 of a quantity along one coordinate direction
// x + t e_i (psi'' along e_i = 2 psi_2); R1 = IR is the first-order case.  Adjoints are kept in REVERSED coefficient order,
// which makes the adjoint of a ring product a ring product (see wf_kernels_wave.hip).
// RF<D> = (value, gradient, half the Laplacian) with respect to D coordinates of a walker (all of them, or a block): the commutative algebra
// IR[t_1..t_D] / (t_i t_j (i != j), t_i^2 - t_1^2, t^3) with basis 1, t_1..t_D, s = t_i^2 -- the second-order jet of a function
// projected to what the Laplacian needs.  The sum over the D directions of the R3 second-order coefficients is carried as one
// number, the value channel and the weight loads are shared.  Like R3 it is a Frobenius algebra (the pairing <a, b> = top
// coefficient of a b is non-degenerate: 1 <-> s, t_i <-> t_i), so the same reverse sweep works with adjoints stored under that
// pairing: value and top slot exchanged, gradient slots in place.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wf_internal.h"

namespace wf {
namespace ring {

constexpr int NBP = 32;

// Rows of one net in the tape of the reverse pass ([sample][net][coefficient][row]): layer input U (D <= 8 rows in a slot of 8),
// hidden activations H1, H2, pre-activation adjoints A1, A2, head-output adjoints O (row d * 32 + j).  Every group starts
// at a multiple of 4 rows so that k_wgrad can stage with 16-byte loads.
template <int D, int NBK = 1> struct Rows {   // NBK: 32-row blocks per dimension (1: <= 32 bases, 2: <= 64)
    static constexpr int U = 0, H1 = 8, H2 = 8 + 64, A1 = 8 + 128, A2 = 8 + 192, O = 8 + 256, N = 8 + 256 + D * NBP * NBK;
    static_assert(D <= 8, "the U slot holds 8 rows");
};

// ---- the rings: R3 = IR[t]/t^3 (Taylor coefficients) for psi and its Laplacian, R1 = IR for first-order objectives
struct R1 {
    float c0;
    static constexpr int NC = 1;
};
struct R3 {
    float c0, c1, c2;
    static constexpr int NC = 3;
};
template <int D> struct RF {
    float c0, g[D], h;   // h = laplacian / 2, the coefficient of s
    static constexpr int NC = D + 2;
};
__device__ __forceinline__ R1 make_cst(R1*, float c) { return R1{c}; }
__device__ __forceinline__ R3 make_cst(R3*, float c) { return R3{c, 0.0f, 0.0f}; }
template <int D> __device__ __forceinline__ RF<D> make_cst(RF<D>*, float c) {
    RF<D> r;
    r.c0 = c;
#pragma unroll
    for (int i = 0; i < D; ++i) r.g[i] = 0.0f;
    r.h = 0.0f;
    return r;
}
template <class T> __device__ __forceinline__ T cst(float c) { return make_cst((T*)nullptr, c); }
__device__ __forceinline__ R1 operator+(R1 a, R1 b) { return R1{a.c0 + b.c0}; }
__device__ __forceinline__ R1 operator-(R1 a, R1 b) { return R1{a.c0 - b.c0}; }
__device__ __forceinline__ R1 operator+(R1 a, float c) { return R1{a.c0 + c}; }
__device__ __forceinline__ R1 operator-(R1 a, float c) { return R1{a.c0 - c}; }
__device__ __forceinline__ R1 operator-(float c, R1 a) { return R1{c - a.c0}; }
__device__ __forceinline__ R1 operator*(R1 a, float c) { return R1{a.c0 * c}; }
__device__ __forceinline__ R1 operator*(R1 a, R1 b) { return R1{a.c0 * b.c0}; }
__device__ __forceinline__ R3 operator+(R3 a, R3 b) { return R3{a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
__device__ __forceinline__ R3 operator-(R3 a, R3 b) { return R3{a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
__device__ __forceinline__ R3 operator+(R3 a, float c) { return R3{a.c0 + c, a.c1, a.c2}; }
__device__ __forceinline__ R3 operator-(R3 a, float c) { return R3{a.c0 - c, a.c1, a.c2}; }
__device__ __forceinline__ R3 operator-(float c, R3 a) { return R3{c - a.c0, -a.c1, -a.c2}; }
__device__ __forceinline__ R3 operator*(R3 a, float c) { return R3{a.c0 * c, a.c1 * c, a.c2 * c}; }
__device__ __forceinline__ R3 operator*(R3 a, R3 b) {
    return R3{a.c0 * b.c0, a.c1 * b.c0 + a.c0 * b.c1, a.c2 * b.c0 + a.c1 * b.c1 + a.c0 * b.c2};
}
// RF: componentwise linear maps; products and compositions by the Leibniz / chain rules of gradient and Laplacian
template <int D, class F> __device__ __forceinline__ RF<D> rf_zip(RF<D> a, RF<D> b, F f) {
    RF<D> r;
    r.c0 = f(a.c0, b.c0);
#pragma unroll
    for (int i = 0; i < D; ++i) r.g[i] = f(a.g[i], b.g[i]);
    r.h = f(a.h, b.h);
    return r;
}
template <int D> __device__ __forceinline__ RF<D> operator+(RF<D> a, RF<D> b) { return rf_zip(a, b, [](float x, float y) { return x + y; }); }
template <int D> __device__ __forceinline__ RF<D> operator-(RF<D> a, RF<D> b) { return rf_zip(a, b, [](float x, float y) { return x - y; }); }
template <int D> __device__ __forceinline__ RF<D> operator*(RF<D> a, float c) { return rf_zip(a, a, [c](float x, float) { return x * c; }); }
template <int D> __device__ __forceinline__ RF<D> operator+(RF<D> a, float c) { a.c0 = a.c0 + c; return a; }
template <int D> __device__ __forceinline__ RF<D> operator-(RF<D> a, float c) { a.c0 = a.c0 - c; return a; }
template <int D> __device__ __forceinline__ RF<D> operator-(float c, RF<D> a) {
    RF<D> r = rf_zip(a, a, [](float x, float) { return -x; });
    r.c0 = c - a.c0;
    return r;
}
template <int D> __device__ __forceinline__ RF<D> operator*(RF<D> a, RF<D> b) {
    RF<D> r;
    r.c0 = a.c0 * b.c0;
    float dot = 0.0f;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        r.g[i] = a.g[i] * b.c0 + a.c0 * b.g[i];
        dot += a.g[i] * b.g[i];
    }
    r.h = a.h * b.c0 + a.c0 * b.h + dot;
    return r;
}
// f(a) from f, f', f'' at a.c0
template <int D> __device__ __forceinline__ RF<D> lift_fn(RF<D> a, float f, float f1, float f2) {
    RF<D> r;
    r.c0 = f;
    float q = 0.0f;   // f2 |g|^2, with f2 inside the squares (as in R3: a vanishing f2 meets a large g without overflow)
#pragma unroll
    for (int i = 0; i < D; ++i) {
        r.g[i] = f1 * a.g[i];
        q += (f2 * a.g[i]) * a.g[i];
    }
    r.h = f1 * a.h + 0.5f * q;
    return r;
}
__device__ __forceinline__ R1 lift_fn(R1, float f, float, float) { return R1{f}; }
__device__ __forceinline__ R3 lift_fn(R3 a, float f, float f1, float f2) { return R3{f, f1 * a.c1, f1 * a.c2 + 0.5f * f2 * a.c1 * a.c1}; }
// Elementary functions through the hardware transcendentals (v_rcp / v_rsq / v_exp / v_log, 1 ulp): the derivative sweeps
// are compared with an fp64 oracle at ~1e-3, and the IEEE-rounded library forms cost 10-30 instructions each.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
template <class T> __device__ __forceinline__ T rrcp(T a) {
    const float r = frcp(a.c0);
    return lift_fn(a, r, -r * r, 2.0f * r * r * r);
}
template <class T> __device__ __forceinline__ T rexp(T a) {
    const float e = fexp(a.c0);
    return lift_fn(a, e, e, e);
}
template <class T> __device__ __forceinline__ T rlog(T a) {
    const float r = frcp(a.c0);
    return lift_fn(a, flog(a.c0), r, -r * r);
}
template <class T> __device__ __forceinline__ T rrsqrt(T a) {   // a^(-1/2)
    const float s = __builtin_amdgcn_rsqf(a.c0), r = s * s;
    return lift_fn(a, s, -0.5f * s * r, 0.75f * s * r * r);
}
template <class T> __device__ __forceinline__ T rtanh(T a) {
    // tanh(x) = 1 - 2 / (e^(2x) + 1); exp2 saturates to +inf / 0 at the ends, which gives exactly +-1
    const float t = 1.0f - 2.0f * frcp(__builtin_amdgcn_exp2f(a.c0 * 2.8853900817779268f) + 1.0f), g = 1.0f - t * t;
    return lift_fn(a, t, g, -2.0f * t * g);
}
template <class T> __device__ __forceinline__ T rsigmoid(T a) {
    const float s = frcp(1.0f + __builtin_amdgcn_exp2f(a.c0 * -1.4426950408889634f)), g = s * (1.0f - s);
    return lift_fn(a, s, g, g * (1.0f - 2.0f * s));
}
// the coordinate x_d along direction `dir`; adjoint seed of the value coefficient (reversed order: last slot)
__device__ __forceinline__ R1 make_var(R1*, float x, int, int) { return R1{x}; }
__device__ __forceinline__ R3 make_var(R3*, float x, int d, int dir) { return R3{x, d == dir ? 1.0f : 0.0f, 0.0f}; }
// (sample `dir` of a walker carries the block of directions dir * K .. dir * K + K - 1; the other coordinates are constants in it)
template <int K> __device__ __forceinline__ RF<K> make_var(RF<K>*, float x, int d, int dir) {
    RF<K> r = make_cst((RF<K>*)nullptr, x);
#pragma unroll
    for (int i = 0; i < K; ++i) r.g[i] = dir * K + i == d ? 1.0f : 0.0f;
    return r;
}
// Laplacian carried by psi: the sum over the directions of 2 psi_2 (R3), or the last component (RF)
__device__ __forceinline__ float lap_of(R3 a) { return 2.0f * a.c2; }
template <int D> __device__ __forceinline__ float lap_of(RF<D> a) { return 2.0f * a.h; }
__device__ __forceinline__ R1 adj_value(R1*, float w) { return R1{w}; }
__device__ __forceinline__ R3 adj_value(R3*, float w) { return R3{0.0f, 0.0f, w}; }
// ... and of the second-derivative along the direction (psi'' = 2 psi_2)
__device__ __forceinline__ R1 adj_second(R1*, float) { return R1{0.0f}; }
__device__ __forceinline__ R3 adj_second(R3*, float w) { return R3{2.0f * w, 0.0f, 0.0f}; }
template <int D> __device__ __forceinline__ RF<D> adj_value(RF<D>*, float w) {
    RF<D> r = make_cst((RF<D>*)nullptr, 0.0f);
    r.h = w;
    return r;
}
template <int D> __device__ __forceinline__ RF<D> adj_second(RF<D>*, float w) { return make_cst((RF<D>*)nullptr, 2.0f * w); }   // laplacian = 2 h

// ---- table lerp (same index arithmetic as the evaluation kernels) and its ring lift
struct Lerp {
    int il, ir;
    float dx, n;
};
__device__ __forceinline__ int wrap_clamp(int i, int n) {
    if (i < 0) i += n;
    return min(max(i, 0), n - 1);
}
__device__ __forceinline__ Lerp make_lerp(float x0, int n_mesh) {
    const int n_points = n_mesh - 1;
    const float xs = x0 * (float)n_points;
    const int xl = (int)floorf(xs), xr = (int)ceilf(xs);
    return Lerp{wrap_clamp(xl, n_mesh), wrap_clamp(xr, n_mesh), x0 - (float)xl / (float)n_points, (float)n_points};
}
// t[o] = order-o lerp of basis j, o = 0..3; tab [4][n_mesh][W]
template <int W = NBP>
__device__ __forceinline__ void lerp4(const float* __restrict__ tab, size_t plane, const Lerp& L, int j, float (&t)[4]) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float yl = tab[o * plane + (size_t)L.il * W + j], yr = tab[o * plane + (size_t)L.ir * W + j];
        t[o] = yl + ((yr - yl) * L.n) * L.dx;
    }
}
// ring value of the order-nd basis at the ring point u (orders beyond 3 clamp to 3)
__device__ __forceinline__ R1 lift(const float (&t)[4], int nd, R1) { return R1{t[min(nd, 3)]}; }
__device__ __forceinline__ R3 lift(const float (&t)[4], int nd, R3 u) {
    const float t0 = t[min(nd, 3)], t1 = t[min(nd + 1, 3)], t2 = t[min(nd + 2, 3)];
    return R3{t0, t1 * u.c1, t1 * u.c2 + 0.5f * t2 * u.c1 * u.c1};
}

template <int D> __device__ __forceinline__ RF<D> lift(const float (&t)[4], int nd, RF<D> u) {
    return lift_fn(u, t[min(nd, 3)], t[min(nd + 1, 3)], t[min(nd + 2, 3)]);
}
// samples per walker of a ring: D directions in R3, ceil(D / K) blocks of directions in RF<K>, one in R1
template <class T> struct RingBlock { static constexpr int K = 0; };
template <> struct RingBlock<R3> { static constexpr int K = 1; };
template <int KK> struct RingBlock<RF<KK>> { static constexpr int K = KK; };
template <class T, int D> constexpr int kDirs = RingBlock<T>::K == 0 ? 1 : (D + RingBlock<T>::K - 1) / (RingBlock<T>::K == 0 ? 1 : RingBlock<T>::K);

}  // namespace ring
}  // namespace wf
