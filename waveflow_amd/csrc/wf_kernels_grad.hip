// wf_kernels_grad.hip -- parameter gradient of the Waveflow wavefunction and of its Laplacian, gfx950.
//
//   grad[p] = sum_b ( w_psi[b] * d psi_b / d theta_p  +  w_lap[b] * d laplacian(psi)_b / d theta_p )
//
// is the vector-Jacobian product behind vqmc.train_step_efficient (vqmc.py:193-221): value_and_grad of the mean local
// energy with the custom tangent rule 2 t_psi (E_L - avg)/psi + (t_Hpsi psi - Hpsi t_psi)/psi^2, where
// Hpsi = -1/2 laplacian + V psi (utils/physics.py:79-93).  k_vqmc_seeds turns (Hpsi, psi, running average) into the two
// weight vectors; the Adam update itself is host code (waveflow_amd/vqmc.py).
//
// Method.  Along one coordinate direction x + t e_i every intermediate quantity is a truncated Taylor polynomial
// a0 + a1 t + a2 t^2 (an element of the ring R = IR[t]/t^3); psi'' along e_i is 2 * psi_2.  The evaluation of psi is a
// composition of ring operations, and the adjoint of a ring product y = a * b with respect to a, written with the
// adjoint coefficients in REVERSED order (abar~ = (abar_2, abar_1, abar_0)), is again a ring product: abar~ = ybar~ * b.
// Hence the reverse sweep over the ring-valued evaluation is ordinary back-propagation with every scalar replaced by a
// ring element, f'(a) replaced by the ring lift of f', and the gradient of a real parameter theta in y = theta * a is the
// top coefficient (ybar~ * a)_2.  The table lerp keeps the reference's derivative rule (the derivative of the order-nd
// lerp is the order-(nd+1) lerp, isplines_jax.py:60-66, bsplines_jax.py:32-38); the reference reaches order 4 in this
// sweep and JAX clamps that traced index to the last cached table (order 3) -- so does lift().
//
// The same sweep over IR itself (ring R1) gives first-order objectives: sum_b w[b] d log_pdf_b / d theta for every model
// the library evaluates (IMADE / MADE layers; Waveflow, M-spline, Normal, Uniform priors) -- the maximum-likelihood
// gradient of benchmark_tests.train_model (benchmark_tests.py:84-101).
//
// One lane = one (walker, direction) sample.  Kernel 1 (k_vjp) runs the forward ring evaluation, keeps the layer
// inputs and hidden activations in an HBM workspace, runs the reverse sweep and leaves the pre-activation adjoints in the
// same workspace; kernel 2 (k_wgrad) contracts activations with adjoints over all samples (LDS-tiled, split over the
// sample axis, fp32 atomics) into a gradient image in the forward weight-image layout; kernel 3 scatters that image to the
// reference's flat leaf order.  Checker: oracle/energy_torch.py (torch reverse mode through the Hessian trace).
#include <hip/hip_runtime.h>

#include "wf_internal.h"

namespace wf {

namespace {

constexpr int H = kHidden;
constexpr int NBP = 32;
constexpr int kGBlock = 64;
constexpr int kRows = 64;

// ---- the rings: R3 = IR[t]/t^3 (Taylor coefficients) for psi and its Laplacian, R1 = IR for first-order objectives
struct R1 {
    float c0;
    static constexpr int NC = 1;
};
struct R3 {
    float c0, c1, c2;
    static constexpr int NC = 3;
};
template <class T> __device__ __forceinline__ T cst(float c);
template <> __device__ __forceinline__ R1 cst<R1>(float c) { return R1{c}; }
template <> __device__ __forceinline__ R3 cst<R3>(float c) { return R3{c, 0.0f, 0.0f}; }
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
// f(a) from f, f', f'' at a.c0
__device__ __forceinline__ R1 lift_fn(R1, float f, float, float) { return R1{f}; }
__device__ __forceinline__ R3 lift_fn(R3 a, float f, float f1, float f2) { return R3{f, f1 * a.c1, f1 * a.c2 + 0.5f * f2 * a.c1 * a.c1}; }
template <class T> __device__ __forceinline__ T rrcp(T a) {
    const float r = 1.0f / a.c0;
    return lift_fn(a, r, -r * r, 2.0f * r * r * r);
}
template <class T> __device__ __forceinline__ T rexp(T a) {
    const float e = expf(a.c0);
    return lift_fn(a, e, e, e);
}
template <class T> __device__ __forceinline__ T rlog(T a) {
    const float r = 1.0f / a.c0;
    return lift_fn(a, logf(a.c0), r, -r * r);
}
template <class T> __device__ __forceinline__ T rrsqrt(T a) {   // a^(-1/2)
    const float s = 1.0f / sqrtf(a.c0), r = 1.0f / a.c0;
    return lift_fn(a, s, -0.5f * s * r, 0.75f * s * r * r);
}
template <class T> __device__ __forceinline__ T rtanh(T a) {
    const float t = tanhf(a.c0), g = 1.0f - t * t;
    return lift_fn(a, t, g, -2.0f * t * g);
}
template <class T> __device__ __forceinline__ T rsigmoid(T a) {
    const float s = 1.0f / (1.0f + expf(-a.c0)), g = s * (1.0f - s);
    return lift_fn(a, s, g, g * (1.0f - 2.0f * s));
}
// the coordinate x_d along direction `dir`; adjoint seed of the value coefficient (reversed order: last slot)
__device__ __forceinline__ R1 make_var(R1*, float x, bool) { return R1{x}; }
__device__ __forceinline__ R3 make_var(R3*, float x, bool along) { return R3{x, along ? 1.0f : 0.0f, 0.0f}; }
__device__ __forceinline__ R1 adj_value(R1*, float w) { return R1{w}; }
__device__ __forceinline__ R3 adj_value(R3*, float w) { return R3{0.0f, 0.0f, w}; }
// ... and of the second-derivative along the direction (psi'' = 2 psi_2)
__device__ __forceinline__ R1 adj_second(R1*, float) { return R1{0.0f}; }
__device__ __forceinline__ R3 adj_second(R3*, float w) { return R3{2.0f * w, 0.0f, 0.0f}; }

// ---- thread-private LDS columns, one plane per coefficient
#define PL(k, j) scr[((k) * kRows + (j)) * kGBlock + threadIdx.x]
__device__ __forceinline__ void sget_(const float* scr, int j, R1& a) { a.c0 = PL(0, j); }
__device__ __forceinline__ void sget_(const float* scr, int j, R3& a) { a.c0 = PL(0, j); a.c1 = PL(1, j); a.c2 = PL(2, j); }
template <class T> __device__ __forceinline__ T sget(const float* scr, int j) { T a; sget_(scr, j, a); return a; }
__device__ __forceinline__ void sset(float* scr, int j, R1 a) { PL(0, j) = a.c0; }
__device__ __forceinline__ void sset(float* scr, int j, R3 a) { PL(0, j) = a.c0; PL(1, j) = a.c1; PL(2, j) = a.c2; }

// ---- HBM workspace: ws[((net * rows + row) * NC + coefficient) * S + sample]
struct Ws {
    float* base;
    int64_t S, s;
    int rows;
};
__device__ __forceinline__ void wget_(const Ws& w, int net, int row, R1& a) { a.c0 = w.base[(int64_t)(net * w.rows + row) * w.S + w.s]; }
__device__ __forceinline__ void wget_(const Ws& w, int net, int row, R3& a) {
    const float* p = w.base + ((int64_t)(net * w.rows + row) * 3) * w.S + w.s;
    a.c0 = p[0]; a.c1 = p[w.S]; a.c2 = p[2 * w.S];
}
template <class T> __device__ __forceinline__ T wget(const Ws& w, int net, int row) { T a; wget_(w, net, row, a); return a; }
__device__ __forceinline__ void wset(const Ws& w, int net, int row, R1 a) { w.base[(int64_t)(net * w.rows + row) * w.S + w.s] = a.c0; }
__device__ __forceinline__ void wset(const Ws& w, int net, int row, R3 a) {
    float* p = w.base + ((int64_t)(net * w.rows + row) * 3) * w.S + w.s;
    p[0] = a.c0; p[w.S] = a.c1; p[2 * w.S] = a.c2;
}
// rows of one net
template <int D> struct Rows {
    static constexpr int U = 0, H1 = D, H2 = D + H, A1 = D + 2 * H, A2 = D + 3 * H, O = D + 4 * H, N = D + 4 * H + D * NBP;
};

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
// t[o] = order-o lerp of basis j, o = 0..3; tab [4][n_mesh][NBP]
__device__ __forceinline__ void lerp4(const float* __restrict__ tab, size_t plane, const Lerp& L, int j, float (&t)[4]) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const float yl = tab[o * plane + (size_t)L.il * NBP + j], yr = tab[o * plane + (size_t)L.ir * NBP + j];
        t[o] = yl + ((yr - yl) * L.n) * L.dx;
    }
}
// ring value of the order-nd basis at the ring point u (orders beyond 3 clamp to 3)
__device__ __forceinline__ R1 lift(const float (&t)[4], int nd, R1) { return R1{t[min(nd, 3)]}; }
__device__ __forceinline__ R3 lift(const float (&t)[4], int nd, R3 u) {
    const float t0 = t[min(nd, 3)], t1 = t[min(nd + 1, 3)], t2 = t[min(nd + 2, 3)];
    return R3{t0, t1 * u.c1, t1 * u.c2 + 0.5f * t2 * u.c1 * u.c1};
}

// ---- conditioner pieces
template <int NIN>
__device__ __forceinline__ R1 dot_ring(const R1 (&v)[NIN], const float* __restrict__ w) {
    float a0 = 0.0f;
#pragma unroll
    for (int a = 0; a < NIN; ++a) a0 = __builtin_fmaf(v[a].c0, w[a], a0);
    return R1{a0};
}
template <int NIN>
__device__ __forceinline__ R3 dot_ring(const R3 (&v)[NIN], const float* __restrict__ w) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
    for (int a = 0; a < NIN; ++a) {
        a0 = __builtin_fmaf(v[a].c0, w[a], a0);
        a1 = __builtin_fmaf(v[a].c1, w[a], a1);
        a2 = __builtin_fmaf(v[a].c2, w[a], a2);
    }
    return R3{a0, a1, a2};
}

// two masked tanh layers; h1 and h2 go to the workspace, h2 stays in registers
template <int D, class T>
__device__ __forceinline__ void hidden_fwd(const NetPlain& net, const T (&x)[D], float* scr, T (&h)[H], const Ws& ws, int n) {
    const float* __restrict__ W0 = net.W0;
    const float* __restrict__ b0 = net.b0;
    for (int j = 0; j < H; ++j) {
        T acc = cst<T>(b0[j]);
#pragma unroll
        for (int a = 0; a < D; ++a) acc = acc + x[a] * W0[a * H + j];
        sset(scr, j, rtanh(acc));
    }
#pragma unroll
    for (int a = 0; a < H; ++a) {
        h[a] = sget<T>(scr, a);
        wset(ws, n, Rows<D>::H1 + a, h[a]);
    }
    const float* __restrict__ W1t = net.W1t;
    const float* __restrict__ b1 = net.b1;
    for (int j = 0; j < H; ++j) sset(scr, j, rtanh(dot_ring<H>(h, W1t + j * H) + b1[j]));
#pragma unroll
    for (int a = 0; a < H; ++a) {
        h[a] = sget<T>(scr, a);
        wset(ws, n, Rows<D>::H2 + a, h[a]);
    }
}

template <class T>
__device__ __forceinline__ T out_ring(const NetPlain& net, const T (&h)[H], int d, int j) {
    return dot_ring<H>(h, net.W2t + ((size_t)d * NBP + j) * H) + net.b2[d * NBP + j];
}

// Sigmoid head of dimension d (IMADE layers: made.py:66-73; M-spline prior: distributions.py:139-152): rows 0..nb-1 <-
// p_j = sigmoid(o_j).  The reference's chain (normalise, + reg, remove_bias, normalise, zero the constrained ends, normalise)
// collapses to c_j = g_j (p_j / S0 + reg) / Q with g = remove_bias factor * kept-by-the-constraints, S0 = sum p,
// Q = sum_j g_j (p_j / S0 + reg).
template <class T>
__device__ __forceinline__ void sigmoid_head(const NetPlain& net, const T (&h)[H], int d, int nb, const float* __restrict__ g, float reg,
                                             float* scr, T& rS0, T& rQ) {
    T S0 = cst<T>(0.0f);
    for (int j = 0; j < nb; ++j) {
        const T p = rsigmoid(out_ring(net, h, d, j));
        sset(scr, j, p);
        S0 = S0 + p;
    }
    rS0 = rrcp(S0);
    T Q = cst<T>(0.0f);
    for (int j = 0; j < nb; ++j) Q = Q + (sget<T>(scr, j) * rS0 + reg) * g[j];
    rQ = rrcp(Q);
}
// ... and its reverse: rows NBP+j hold cbar_j, dotC = sum_j cbar_j c_j; writes obar rows (d, j) of net n to the workspace
template <int D, class T>
__device__ __forceinline__ void sigmoid_head_bwd(int d, int nb, const float* __restrict__ g, T rS0, T rQ, T dotC, float* scr, const Ws& ws, int n) {
    // qbar_j = (cbar_j - dotC) / Q;  w0bar_j = g_j qbar_j;  dot0 = sum w0bar_j w0_j
    T dot0 = cst<T>(0.0f);
    for (int j = 0; j < nb; ++j) {
        const T gw0 = ((sget<T>(scr, NBP + j) - dotC) * rQ) * g[j];
        sset(scr, NBP + j, gw0);
        dot0 = dot0 + gw0 * (sget<T>(scr, j) * rS0);
    }
    for (int j = 0; j < NBP; ++j) {
        T go = cst<T>(0.0f);
        if (j < nb) {
            const T p = sget<T>(scr, j);
            go = ((sget<T>(scr, NBP + j) - dot0) * rS0) * (p * (1.0f - p));
        }
        wset(ws, n, Rows<D>::O + d * NBP + j, go);
    }
}

// psi head of dimension d (wavefunctions.py:54-71, bsplines_jax.py:127-137, 173-199 with zero-only constraints):
// rows 0..nb-1 <- o_j;  rows 32..32+nb-1 <- c_j = sum_a (k_a o_a) ob_to_b[a][j] * (rS rN1);  e_j = c_j * rN2
template <class T>
__device__ __forceinline__ void prior_head(const NetPlain& net, const T (&h)[H], int d, int nb, const float* __restrict__ keep,
                                           const float* __restrict__ o2b, float* scr, T& rS, T& rN1, T& rN2) {
    T S = cst<T>(0.0f);
    for (int j = 0; j < nb; ++j) {
        const T o = out_ring(net, h, d, j);
        sset(scr, j, o);
        S = S + o;
    }
    rS = rrcp(S);
    T N1 = cst<T>(0.0f);
    for (int j = 0; j < nb; ++j) {
        const T w = (sget<T>(scr, j) * rS) * keep[j];
        N1 = N1 + w * w;
    }
    rN1 = rrsqrt(N1);
    const T f = rS * rN1;
    T N2 = cst<T>(0.0f);
    for (int j = 0; j < nb; ++j) {
        T acc = cst<T>(0.0f);
        for (int a = 0; a < nb; ++a) acc = acc + sget<T>(scr, a) * (keep[a] * o2b[a * NBP + j]);
        const T c = acc * f;
        sset(scr, NBP + j, c);
        N2 = N2 + c * c;
    }
    rN2 = rrsqrt(N2);
}

// Reverse sweep through one conditioner.  In: adjoints of the head outputs in ws rows O (all D * NBP rows written),
// h2 in registers, h1 in the workspace.  Out: pre-activation adjoints A2, A1 in the workspace; gU += W0-path adjoint.
template <int D, class T>
__device__ __forceinline__ void hidden_bwd(const NetPlain& net, T (&h)[H], float* scr, const Ws& ws, int n, T (&gU)[D]) {
    // hbar2_a = sum_{d, j} obar_{d j} W2[a][d][j]
#pragma unroll 1
    for (int d = 0; d < D; ++d) {
        T o[NBP];
#pragma unroll
        for (int j = 0; j < NBP; ++j) o[j] = wget<T>(ws, n, Rows<D>::O + d * NBP + j);
        for (int a = 0; a < H; ++a) {
            const T acc = dot_ring<NBP>(o, net.W2n + ((size_t)a * D + d) * NBP);
            sset(scr, a, d == 0 ? acc : sget<T>(scr, a) + acc);
        }
    }
    // abar2 = hbar2 * tanh'(z2) = hbar2 * (1 - h2^2)
#pragma unroll
    for (int a = 0; a < H; ++a) {
        const T A = sget<T>(scr, a) * (1.0f - h[a] * h[a]);
        wset(ws, n, Rows<D>::A2 + a, A);
        h[a] = A;
    }
    for (int a = 0; a < H; ++a) sset(scr, a, dot_ring<H>(h, net.W1n + (size_t)a * H));
#pragma unroll
    for (int a = 0; a < H; ++a) {
        const T h1 = wget<T>(ws, n, Rows<D>::H1 + a);
        const T A = sget<T>(scr, a) * (1.0f - h1 * h1);
        wset(ws, n, Rows<D>::A1 + a, A);
        h[a] = A;
    }
#pragma unroll
    for (int a = 0; a < D; ++a) gU[a] = gU[a] + dot_ring<H>(h, net.W0 + (size_t)a * H);
}

// mode 0: sum_b w[b] log_pdf_b (any model);  mode 1: sum_b (w[b] psi_b + w2[b] laplacian_b) (Waveflow, T = R3; T = R1: psi only)
template <int D, class T>
__global__ __launch_bounds__(kGBlock) void k_vjp(const ModelDev* __restrict__ mdp, int mode, const float* __restrict__ tabI, const float* __restrict__ tabP,
                                                  const float* __restrict__ fk_nat, const float* __restrict__ xg, int64_t B,
                                                  const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ wsb, int64_t S) {
    __shared__ float scr[T::NC * kRows * kGBlock];
    const ModelDev& md = *mdp;
    constexpr int DIRS = T::NC == 3 ? D : 1;
    const int64_t s = (int64_t)blockIdx.x * kGBlock + threadIdx.x;
    if (s >= B * DIRS) return;
    const int64_t b = s / DIRS;
    const int dir = (int)(s - b * DIRS);
    const Ws ws{wsb, S, s, Rows<D>::N};
    const float L = md.box_L, tol = 1e-7f;
    const int n_mesh = md.layer_kind == WF_LAYER_IMADE ? md.isp.n_mesh : md.psp.n_mesh;
    const size_t plane = (size_t)n_mesh * NBP;
    const float* __restrict__ gI = fk_nat;        // remove_bias * kept, I-spline rows
    const float* __restrict__ kP = fk_nat + 64;   // prior rows: kept (orthogonal-B), remove_bias * kept (M)
    const int L_layers = md.n_layers;
    const bool imade = md.layer_kind == WF_LAYER_IMADE;
    const bool has_pnet = md.prior_kind == WF_PRIOR_WAVEFLOW || md.prior_kind == WF_PRIOR_MFLOW;
    T* const tag = nullptr;

    T cur[D], nxt[D];
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = make_var(tag, xg[b * D + d], d == dir);
    // ---- BoxTransformLayer (made.py:118-137, 156-183); it has no parameters, only its value is needed
    T logdet = cst<T>(0.0f);
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
#pragma unroll
        for (int d = 0; d < D; ++d) nxt[d] = cur[d];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) cur[d] = nxt[d];

    T h[H];
    // ================================================================ forward
    for (int l = 0; l < L_layers; ++l) {
        const NetPlain& net = md.nets[l];
#pragma unroll
        for (int d = 0; d < D; ++d) wset(ws, l, Rows<D>::U + d, cur[d]);
        hidden_fwd<D, T>(net, cur, scr, h, ws, l);
        if (imade) {
            const int nb = md.isp.nb;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                T rS0, rQ;
                sigmoid_head(net, h, d, nb, gI, md.i_reg, scr, rS0, rQ);
                const Lerp lp = make_lerp(cur[d].c0, n_mesh);
                T y = cst<T>(0.0f), dy = cst<T>(0.0f);
                for (int j = 0; j < nb; ++j) {
                    float t[4];
                    lerp4(tabI, plane, lp, j, t);
                    const T c = ((sget<T>(scr, j) * rS0 + md.i_reg) * gI[j]) * rQ;
                    y = y + c * lift(t, 0, cur[d]);
                    dy = dy + c * lift(t, 1, cur[d]);
                }
                nxt[d] = y;
                logdet = logdet + rlog(dy + 1e-7f);
            }
        } else {
            // MADE (made.py:21-27): y = (x - bias) * exp(-log_weight), log det = -sum log_weight
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T lw = out_ring(net, h, d, 0), bias = out_ring(net, h, d, 1);
                nxt[d] = (cur[d] - bias) * rexp(cst<T>(0.0f) - lw);
                logdet = logdet - lw;
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = nxt[D - 1 - d];
    }
    // ---- prior
    const int NP = L_layers;
    const NetPlain& pnet = md.nets[NP];
    const int nbp_ = md.psp.nb;
    T v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = cst<T>(1.0f);
    T lp_sum = cst<T>(0.0f);
    if (has_pnet) {
#pragma unroll
        for (int d = 0; d < D; ++d) wset(ws, NP, Rows<D>::U + d, cur[d]);
        hidden_fwd<D, T>(pnet, cur, scr, h, ws, NP);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            T uc = cur[d];   // np.clip(u, 0, 1)
            if (uc.c0 < 0.0f) uc = cst<T>(0.0f);
            else if (uc.c0 > 1.0f) uc = cst<T>(1.0f);
            const Lerp lp = make_lerp(uc.c0, n_mesh);
            T acc = cst<T>(0.0f);
            if (md.prior_kind == WF_PRIOR_WAVEFLOW) {
                T rS, rN1, rN2;
                prior_head(pnet, h, d, nbp_, kP, md.ob_to_b, scr, rS, rN1, rN2);
                for (int j = 0; j < nbp_; ++j) {
                    float t[4];
                    lerp4(tabP, plane, lp, j, t);
                    acc = acc + (sget<T>(scr, NBP + j) * rN2) * lift(t, 0, uc);
                }
                const float sc2 = ((md.constrained_mask >> d) & 1u) ? 0.5f : 1.0f;
                lp_sum = lp_sum + rlog((acc * acc) * sc2 + 1e-7f);
            } else {
                T rS0, rQ;
                sigmoid_head(pnet, h, d, nbp_, kP, 0.0f, scr, rS0, rQ);
                for (int j = 0; j < nbp_; ++j) {
                    float t[4];
                    lerp4(tabP, plane, lp, j, t);
                    acc = acc + ((sget<T>(scr, j) * rS0) * kP[j]) * rQ * lift(t, 0, uc);
                }
                lp_sum = lp_sum + rlog(acc + 1e-7f);
            }
            v[d] = acc;
        }
    }
    T prod = cst<T>(1.0f);
#pragma unroll
    for (int d = 0; d < D; ++d) prod = prod * (v[d] * (((md.constrained_mask >> d) & 1u) ? 0.70710678118654752f : 1.0f));
    const T E = rexp(logdet * 0.5f);
    const T psi = prod * E;

    // ================================================================ reverse (adjoints in reversed coefficient order)
    T gLD, gProd = cst<T>(0.0f), gOut = cst<T>(0.0f);
    if (mode == 1) {
        const T gPsi = adj_value(tag, dir == 0 ? w1[b] : 0.0f) + adj_second(tag, w2 ? w2[b] : 0.0f);
        gLD = (gPsi * psi) * 0.5f;
        gProd = gPsi * E;
    } else {
        gOut = adj_value(tag, w1[b]);   // log_pdf = log prior + log det
        gLD = gOut;
    }
    T gU[D];
#pragma unroll
    for (int d = 0; d < D; ++d) gU[d] = cst<T>(0.0f);
    if (has_pnet) {
        // h still holds the prior net's h2
#pragma unroll
        for (int d = 0; d < D; ++d) {
            T uc = cur[d];
            bool inside = true;
            if (uc.c0 < 0.0f) { uc = cst<T>(0.0f); inside = false; }
            else if (uc.c0 > 1.0f) { uc = cst<T>(1.0f); inside = false; }
            const Lerp lp = make_lerp(uc.c0, n_mesh);
            const float sc = ((md.constrained_mask >> d) & 1u) ? 0.70710678118654752f : 1.0f;
            T gv;
            if (mode == 1) {
                T others = cst<T>(1.0f);
#pragma unroll
                for (int e = 0; e < D; ++e) {
                    const float se = ((md.constrained_mask >> e) & 1u) ? 0.70710678118654752f : 1.0f;
                    others = e == d ? others * se : others * (v[e] * se);
                }
                gv = gProd * others;
            } else if (md.prior_kind == WF_PRIOR_WAVEFLOW) {
                gv = (gOut * rrcp((v[d] * v[d]) * (sc * sc) + 1e-7f)) * (v[d] * (2.0f * sc * sc));
            } else {
                gv = gOut * rrcp(v[d] + 1e-7f);
            }
            if (md.prior_kind == WF_PRIOR_WAVEFLOW) {
                T rS, rN1, rN2;
                prior_head(pnet, h, d, nbp_, kP, md.ob_to_b, scr, rS, rN1, rN2);
                T dotE = cst<T>(0.0f), d1 = cst<T>(0.0f);
                for (int j = 0; j < nbp_; ++j) {
                    float t[4];
                    lerp4(tabP, plane, lp, j, t);
                    const T e = sget<T>(scr, NBP + j) * rN2;
                    dotE = dotE + (gv * lift(t, 0, uc)) * e;
                    d1 = d1 + e * lift(t, 1, uc);
                }
                if (inside) gU[d] = gU[d] + gv * d1;
                T gc[NBP];
#pragma unroll
                for (int j = 0; j < NBP; ++j) {
                    gc[j] = cst<T>(0.0f);
                    if (j < nbp_) {
                        float t[4];
                        lerp4(tabP, plane, lp, j, t);
                        const T e = sget<T>(scr, NBP + j) * rN2;
                        gc[j] = ((gv * lift(t, 0, uc)) - e * dotE) * rN2;
                    }
                }
                const T rSN = rS * rN1;
                T dotA = cst<T>(0.0f);
                for (int a = 0; a < nbp_; ++a) {
                    const T ga = dot_ring<NBP>(gc, md.ob_to_b + (size_t)a * NBP);
                    sset(scr, NBP + a, ga);
                    dotA = dotA + ga * ((sget<T>(scr, a) * kP[a]) * rSN);
                }
                T dotW = cst<T>(0.0f);
                for (int a = 0; a < nbp_; ++a) {
                    const T aa = (sget<T>(scr, a) * kP[a]) * rSN;
                    const T gw = ((sget<T>(scr, NBP + a) - aa * dotA) * rN1) * kP[a];
                    sset(scr, NBP + a, gw);
                    dotW = dotW + gw * (sget<T>(scr, a) * rS);
                }
                for (int a = 0; a < NBP; ++a)
                    wset(ws, NP, Rows<D>::O + d * NBP + a, a < nbp_ ? (sget<T>(scr, NBP + a) - dotW) * rS : cst<T>(0.0f));
            } else {
                T rS0, rQ;
                sigmoid_head(pnet, h, d, nbp_, kP, 0.0f, scr, rS0, rQ);
                T dotC = cst<T>(0.0f), d1 = cst<T>(0.0f);
                for (int j = 0; j < nbp_; ++j) {
                    float t[4];
                    lerp4(tabP, plane, lp, j, t);
                    const T c = ((sget<T>(scr, j) * rS0) * kP[j]) * rQ;
                    const T gcj = gv * lift(t, 0, uc);
                    sset(scr, NBP + j, gcj);
                    dotC = dotC + gcj * c;
                    d1 = d1 + c * lift(t, 1, uc);
                }
                if (inside) gU[d] = gU[d] + gv * d1;
                sigmoid_head_bwd<D, T>(d, nbp_, kP, rS0, rQ, dotC, scr, ws, NP);
            }
        }
        hidden_bwd<D, T>(pnet, h, scr, ws, NP, gU);
    } else if (md.prior_kind == WF_PRIOR_NORMAL) {
        // Normal(offset) (distributions.py:25-37): log p = sum_d -(log 2 pi + z^2) / 2, z = u + offset
#pragma unroll
        for (int d = 0; d < D; ++d) gU[d] = gOut * ((cur[d] + md.normal_offset) * -1.0f);
    }

    // ---- flow layers, last to first
    for (int l = L_layers - 1; l >= 0; --l) {
        const NetPlain& net = md.nets[l];
        T gY[D], U[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            gY[d] = gU[D - 1 - d];   // Reverse (bijections.py:337-340)
            U[d] = wget<T>(ws, l, Rows<D>::U + d);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) gU[d] = cst<T>(0.0f);
#pragma unroll
        for (int a = 0; a < H; ++a) h[a] = wget<T>(ws, l, Rows<D>::H2 + a);
        if (imade) {
            const int nb = md.isp.nb;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                T rS0, rQ;
                sigmoid_head(net, h, d, nb, gI, md.i_reg, scr, rS0, rQ);
                const Lerp lp = make_lerp(U[d].c0, n_mesh);
                T dy = cst<T>(0.0f), y2 = cst<T>(0.0f);
                for (int j = 0; j < nb; ++j) {
                    float t[4];
                    lerp4(tabI, plane, lp, j, t);
                    const T c = ((sget<T>(scr, j) * rS0 + md.i_reg) * gI[j]) * rQ;
                    dy = dy + c * lift(t, 1, U[d]);
                    y2 = y2 + c * lift(t, 2, U[d]);
                }
                const T gdy = gLD * rrcp(dy + 1e-7f);
                gU[d] = gU[d] + gY[d] * dy + gdy * y2;
                T dotC = cst<T>(0.0f);
                for (int j = 0; j < nb; ++j) {
                    float t[4];
                    lerp4(tabI, plane, lp, j, t);
                    const T c = ((sget<T>(scr, j) * rS0 + md.i_reg) * gI[j]) * rQ;
                    const T gcj = gY[d] * lift(t, 0, U[d]) + gdy * lift(t, 1, U[d]);
                    sset(scr, NBP + j, gcj);
                    dotC = dotC + gcj * c;
                }
                sigmoid_head_bwd<D, T>(d, nb, gI, rS0, rQ, dotC, scr, ws, l);
            }
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const T lw = out_ring(net, h, d, 0), bias = out_ring(net, h, d, 1);
                const T e = rexp(cst<T>(0.0f) - lw);
                const T y = (U[d] - bias) * e;
                gU[d] = gU[d] + gY[d] * e;
                for (int j = 0; j < NBP; ++j) {
                    T go = cst<T>(0.0f);
                    if (j == 0) go = cst<T>(0.0f) - (gY[d] * y) - gLD;   // d y / d lw = -y,  d logdet / d lw = -1
                    else if (j == 1) go = cst<T>(0.0f) - (gY[d] * e);     // d y / d bias = -e
                    wset(ws, l, Rows<D>::O + d * NBP + j, go);
                }
            }
        }
        hidden_bwd<D, T>(net, h, scr, ws, l, gU);
    }
}

// ---- kernel 2: weight gradients.  C[m][n] = sum_s sum_k X[m][k][s] * Y[n][NC-1-k][s];  bias[n] = sum_s Y[n][NC-1][s]
struct WJob {
    int xrow, M, yrow, N;      // workspace rows (within a net) of the activations (X) and of the adjoints (Y)
    int out, sm, sn, bias;     // gradient-image offsets (within a net): C[m][n] -> out + m*sm + n*sn; bias[n] -> bias + n
};
struct WJobs {
    WJob j[3];
};
constexpr int kWT = 64;   // output tile
constexpr int kWK = 32;   // samples per staged slab
template <int NC>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ ws, int64_t S, int64_t n_samples, int rows, const WJobs jobs,
                                               int n_ntiles_max, float* __restrict__ gimg, int64_t net_img_floats) {
    __shared__ float Xs[NC][kWT][kWK + 1];
    __shared__ float Ys[NC][kWT][kWK + 1];
    const int net = blockIdx.z;
    const int job_i = blockIdx.y / n_ntiles_max, ntile = blockIdx.y % n_ntiles_max;
    const WJob jb = jobs.j[job_i];
    const int n0 = ntile * kWT;
    if (n0 >= jb.N) return;
    const int tid = threadIdx.x, tm = tid & 15, tn = tid >> 4;
    float acc[4][4] = {};
    float bacc[4] = {};
    const float* __restrict__ base = ws + (int64_t)net * rows * NC * S;
    const int64_t n_slabs = (n_samples + kWK - 1) / kWK;
    for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
        const int64_t s0 = slab * kWK;
        __syncthreads();
        for (int e = tid; e < NC * kWT * kWK; e += 256) {
            const int kk = e % kWK, r = (e / kWK) % kWT, k = e / (kWK * kWT);
            const int64_t s = s0 + kk;
            const bool ok = s < n_samples;
            Xs[k][r][kk] = (ok && r < jb.M) ? base[((int64_t)(jb.xrow + r) * NC + k) * S + s] : 0.0f;
            Ys[k][r][kk] = (ok && n0 + r < jb.N) ? base[((int64_t)(jb.yrow + n0 + r) * NC + k) * S + s] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < kWK; ++kk) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                float xv[4], yv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xv[i] = Xs[k][tm * 4 + i][kk];
                    yv[i] = Ys[NC - 1 - k][tn * 4 + i][kk];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(xv[i], yv[j], acc[i][j]);
            }
            if (tm == 0)
#pragma unroll
                for (int j = 0; j < 4; ++j) bacc[j] += Ys[NC - 1][tn * 4 + j][kk];
        }
    }
    float* __restrict__ g = gimg + (int64_t)net * net_img_floats;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = tm * 4 + i, n = n0 + tn * 4 + j;
            if (m < jb.M && n < jb.N) atomicAdd(&g[jb.out + (int64_t)m * jb.sm + (int64_t)n * jb.sn], acc[i][j]);
        }
    if (tm == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tn * 4 + j;
            if (n < jb.N) atomicAdd(&g[jb.bias + n], bacc[j]);
        }
}

__global__ void k_grad_scatter(const float* __restrict__ gimg, const int32_t* __restrict__ map, int64_t n_img, float* __restrict__ flat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img) return;
    const int32_t t = map[i];
    if (t >= 0) flat[t] = gimg[i];
}

// ---- loss_fn_efficient's tangent rule as per-walker weights (vqmc.py:198-212)
__global__ void k_vqmc_seeds(const float* __restrict__ xg, int64_t B, int D, const Protons pr, const float* __restrict__ hpsi,
                             const float* __restrict__ psi, float running_avg, float inv_count, float* __restrict__ e_loc,
                             float* __restrict__ w_psi, float* __restrict__ w_lap) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    // potential (physics.py:60-76)
    float V = 0.0f;
    for (int p = 0; p < pr.n; ++p)
        for (int d = 0; d < D; ++d) {
            const float r = pr.pos[p] - xg[b * D + d];
            V -= 1.0f / sqrtf(1.0f + r * r);
        }
    for (int i = 0; i < D; ++i)
        for (int j = 0; j < i; ++j) {
            const float r = xg[b * D + i] - xg[b * D + j];
            V += 1.0f / sqrtf(1.0f + r * r);
        }
    const float ps = psi[b], hp = hpsi[b];
    const float el = hp / (ps + 1e-8f);
    // d loss = [2 (E - avg)/psi - Hpsi/psi^2] dpsi + (1/psi) dHpsi,  dHpsi = -1/2 dlap + V dpsi
    const float a = 2.0f * (el - running_avg) / ps - hp / (ps * ps);
    const float c = 1.0f / ps;
    e_loc[b] = el;
    w_psi[b] = (a + c * V) * inv_count;
    w_lap[b] = -0.5f * c * inv_count;
}

int finish() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

template <int D, class T>
int run_vjp(const ModelDev& md, const ModelDev* md_dev, int mode, const float* tabI4, const float* tabP4, const float* fk_nat, const float* x,
            int64_t B, const float* w1, const float* w2, float* ws, int64_t S, float* grad_img, int64_t net_img_floats, hipStream_t s) {
    using R = Rows<D>;
    constexpr int NC = T::NC;
    const int64_t n_samples = B * (NC == 3 ? D : 1);
    const bool has_pnet = md.prior_kind == WF_PRIOR_WAVEFLOW || md.prior_kind == WF_PRIOR_MFLOW;
    const int n_nets = md.n_layers + (has_pnet ? 1 : 0);
    hipLaunchKernelGGL((k_vjp<D, T>), dim3((unsigned)((n_samples + kGBlock - 1) / kGBlock)), dim3(kGBlock), 0, s, md_dev, mode, tabI4, tabP4,
                       fk_nat, x, B, w1, w2, ws, S);
    if (n_nets == 0) return finish();
    // forward-image layout of one net: W0 [D][64], b0 [64], W1t [64 out][64 in], b1 [64], W2t [D*NBP][64], b2 [D*NBP]
    const int oW0 = 0, ob0 = D * H, oW1 = ob0 + H, ob1 = oW1 + H * H, oW2 = ob1 + H, ob2 = oW2 + D * NBP * H;
    WJobs jobs;
    jobs.j[0] = WJob{R::U, D, R::A1, H, oW0, H, 1, ob0};            // dW0[a][j]
    jobs.j[1] = WJob{R::H1, H, R::A2, H, oW1, 1, H, ob1};           // dW1t[j][a]
    jobs.j[2] = WJob{R::H2, H, R::O, D * NBP, oW2, 1, H, ob2};      // dW2t[(d, jb)][a]
    const int n_ntiles = (D * NBP + kWT - 1) / kWT;
    const int64_t n_slabs = (n_samples + kWK - 1) / kWK;
    int split = (int)(n_slabs < 256 ? n_slabs : 256);
    if (split < 1) split = 1;
    hipLaunchKernelGGL((k_wgrad<NC>), dim3((unsigned)split, (unsigned)(3 * n_ntiles), (unsigned)n_nets), dim3(256), 0, s, (const float*)ws, S,
                       n_samples, R::N, jobs, n_ntiles, grad_img, net_img_floats);
    return finish();
}

}  // namespace

int grad_ws_rows(int D) { return D + 4 * H + D * NBP; }

// second_order: the ring IR[t]/t^3 with one sample per (walker, direction); otherwise first order, one sample per walker
int launch_vjp(const ModelDev& md, const ModelDev* md_dev, int mode, int second_order, const float* tabI4, const float* tabP4, const float* fk_nat,
               const float* x, int64_t B, const float* w1, const float* w2, float* ws, int64_t S, float* grad_img, int64_t net_img_floats,
               void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALL(DD)                                                                                                                       \
    return second_order ? run_vjp<DD, R3>(md, md_dev, mode, tabI4, tabP4, fk_nat, x, B, w1, w2, ws, S, grad_img, net_img_floats, s)    \
                        : run_vjp<DD, R1>(md, md_dev, mode, tabI4, tabP4, fk_nat, x, B, w1, w2, ws, S, grad_img, net_img_floats, s)
    switch (md.D) {
        case 2: CALL(2);
        case 3: CALL(3);
        case 4: CALL(4);
        default: return WF_ERR_UNSUPPORTED;
    }
#undef CALL
}

int launch_grad_scatter(const float* grad_img, const int32_t* map, int64_t n_img, float* grad_flat, void* stream) {
    hipLaunchKernelGGL(k_grad_scatter, dim3((unsigned)((n_img + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_img, map, n_img, grad_flat);
    return finish();
}

int launch_vqmc_seeds(const float* x, int64_t B, int D, const Protons& pr, const float* hpsi, const float* psi, float running_avg,
                      float inv_count, float* e_loc, float* w_psi, float* w_lap, void* stream) {
    hipLaunchKernelGGL(k_vqmc_seeds, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, D, pr, hpsi, psi, running_avg,
                       inv_count, e_loc, w_psi, w_lap);
    return finish();
}

}  // namespace wf
