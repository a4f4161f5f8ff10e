// wf_kernels_energy.hip -- local energy of the Waveflow wavefunction, gfx950.
//
//   H psi = -1/2 * laplacian(psi) + V * psi          physics.construct_hamiltonian_function (utils/physics.py:79-93)
//   laplacian = trace of the per-walker Hessian       physics.laplacian (utils/physics.py:50-52, jax.hessian)
//   V = soft-Coulomb, one space dimension             physics.get_potential (utils/physics.py:60-76)
//
// The reference differentiates psi (wavefunctions.py:54-71) with JAX; the table lerp carries a custom JVP: the
// derivative of the order-nd lerp is the order-(nd+1) lerp of the cached derivative tables (isplines_jax.py:60-66,
// bsplines_jax.py:32-38) -- that is why four derivative orders are cached.  Here the same derivative semantics are
// propagated in forward (Taylor) mode: every quantity is a second-order jet (value, d/dt, d^2/dt^2) along one
// coordinate direction x + t*e_i, one pass per direction, laplacian = sum_i psi''_i.  One lane = one walker, weights
// are wave-uniform scalar loads (as in wf_kernels_scalar.hip), jets that need runtime indexing live in thread-private
// LDS columns.  Checker: oracle/energy_torch.py (reverse-mode autograd with the same custom rule).
#include <hip/hip_runtime.h>

#include "wf_internal.h"

namespace wf {

namespace {

constexpr int H = kHidden;
constexpr int NBP = 32;
constexpr int kEBlock = 64;
constexpr int kRows = 64;   // max(hidden, 2 * NBP)

struct J {
    float v, d, dd;
};
__device__ __forceinline__ J jc(float c) { return J{c, 0.0f, 0.0f}; }
__device__ __forceinline__ J operator+(J a, J b) { return J{a.v + b.v, a.d + b.d, a.dd + b.dd}; }
__device__ __forceinline__ J operator-(J a, J b) { return J{a.v - b.v, a.d - b.d, a.dd - b.dd}; }
__device__ __forceinline__ J operator+(J a, float c) { return J{a.v + c, a.d, a.dd}; }
__device__ __forceinline__ J operator-(J a, float c) { return J{a.v - c, a.d, a.dd}; }
__device__ __forceinline__ J operator-(float c, J a) { return J{c - a.v, -a.d, -a.dd}; }
__device__ __forceinline__ J operator*(J a, float c) { return J{a.v * c, a.d * c, a.dd * c}; }
__device__ __forceinline__ J operator*(J a, J b) { return J{a.v * b.v, a.d * b.v + a.v * b.d, a.dd * b.v + 2.0f * a.d * b.d + a.v * b.dd}; }
// f(a) given f, f', f'' at a.v
__device__ __forceinline__ J chain(J a, float f, float f1, float f2) { return J{f, f1 * a.d, f2 * a.d * a.d + f1 * a.dd}; }
__device__ __forceinline__ J jrcp(J a) {
    const float r = 1.0f / a.v;
    return chain(a, r, -r * r, 2.0f * r * r * r);
}
__device__ __forceinline__ J operator/(J a, J b) { return a * jrcp(b); }
__device__ __forceinline__ J jexp(J a) {
    const float e = expf(a.v);
    return chain(a, e, e, e);
}
__device__ __forceinline__ J jlog(J a) {
    const float r = 1.0f / a.v;
    return chain(a, logf(a.v), r, -r * r);
}
__device__ __forceinline__ J jsqrt(J a) {
    const float s = sqrtf(a.v);
    return chain(a, s, 0.5f / s, -0.25f / (s * a.v));
}
__device__ __forceinline__ J jtanh(J a) {
    const float t = tanhf(a.v), g = 1.0f - t * t;
    return chain(a, t, g, -2.0f * t * g);
}
__device__ __forceinline__ J jsigmoid(J a) {
    const float s = 1.0f / (1.0f + expf(-a.v)), g = s * (1.0f - s);
    return chain(a, s, g, g * (1.0f - 2.0f * s));
}

// thread-private LDS columns, three planes
#define SV(j) scr[(j) * kEBlock + threadIdx.x]
#define SD(j) scr[(kRows + (j)) * kEBlock + threadIdx.x]
#define SDD(j) scr[(2 * kRows + (j)) * kEBlock + threadIdx.x]
__device__ __forceinline__ J sget(const float* scr, int j) { return J{SV(j), SD(j), SDD(j)}; }
__device__ __forceinline__ void sset(float* scr, int j, J a) { SV(j) = a.v; SD(j) = a.d; SDD(j) = a.dd; }

__device__ __forceinline__ int wrap_clamp(int i, int n) {
    if (i < 0) i += n;
    return min(max(i, 0), n - 1);
}

// sum_j c_j * X_cached(x, j, nd) as a jet: basis_j = (T_nd, T_{nd+1} x', T_{nd+2} x'^2 + T_{nd+1} x'') at x.v
// tab: [orders][n_mesh][NBP]; c_j = scratch rows row0 + j
__device__ __forceinline__ J spline_jet(const float* __restrict__ tab, int n_mesh, int nd, J x, const float* scr, int row0, int nb) {
    const int n_points = n_mesh - 1;
    const float xs = x.v * (float)n_points;
    const int xl = (int)floorf(xs), xr = (int)ceilf(xs);
    const int il = wrap_clamp(xl, n_mesh), ir = wrap_clamp(xr, n_mesh);
    const float dx = x.v - (float)xl / (float)n_points, n = (float)n_points;
    const size_t plane = (size_t)n_mesh * NBP;
    J acc = jc(0.0f);
    for (int j = 0; j < nb; ++j) {
        float t[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float yl = tab[(nd + o) * plane + (size_t)il * NBP + j], yr = tab[(nd + o) * plane + (size_t)ir * NBP + j];
            t[o] = yl + ((yr - yl) * n) * dx;
        }
        const J basis{t[0], t[1] * x.d, t[2] * x.d * x.d + t[1] * x.dd};
        acc = acc + sget(scr, row0 + j) * basis;
    }
    return acc;
}

// enforce_boundary_conditions on jets (isplines_jax.py:158-194, bsplines_jax.py:173-199); rows row0..row0+nb
__device__ __forceinline__ void enforce_bc_jet(const SplineDev& s, int kind, float* scr, int row0) {
    const int nb = s.nb;
    for (int p = 0; p < s.n_left; ++p) {
        const int nd = s.left_nd[p];
        J sum = jc(0.0f);
        for (int j = 0; j < nd; ++j) sum = sum + sget(scr, row0 + j) * s.left_prev[p][j];
        sset(scr, row0 + nd, (s.left_val[p] - sum) * (1.0f / s.left_value[p]));
    }
    for (int p = 0; p < s.n_right; ++p) {
        const int nd = s.right_nd[p];
        if (kind == WF_SPLINE_I && nd == 0) {
            sset(scr, row0 + nb - 1, jc(0.0f));
            continue;
        }
        J sum = jc(0.0f);
        for (int j = 0; j < nd; ++j) sum = sum + sget(scr, row0 + nb - 1 - j) * s.right_prev[p][j];
        sset(scr, row0 + nb - nd - 1, (s.right_val[p] - sum) * (1.0f / s.right_value[p]));
    }
    J ss = jc(0.0f);
    if (kind == WF_SPLINE_B) {
        for (int j = 0; j < nb; ++j) { const J w = sget(scr, row0 + j); ss = ss + w * w; }
        ss = jsqrt(ss);
    } else {
        for (int j = 0; j < nb; ++j) ss = ss + sget(scr, row0 + j);
    }
    const J r = jrcp(ss);
    for (int j = 0; j < nb; ++j) sset(scr, row0 + j, sget(scr, row0 + j) * r);
}

// two masked tanh layers on jets; result in registers
template <int D>
__device__ __forceinline__ void hidden_jet(const NetPlain& net, const J (&x)[D], float* scr, J (&h)[H]) {
    const float* __restrict__ W0 = net.W0;
    const float* __restrict__ b0 = net.b0;
    for (int j = 0; j < H; ++j) {
        J acc = jc(0.0f);
#pragma unroll
        for (int a = 0; a < D; ++a) acc = acc + x[a] * W0[a * H + j];
        sset(scr, j, jtanh(acc + b0[j]));
    }
#pragma unroll
    for (int a = 0; a < H; ++a) h[a] = sget(scr, a);
    const float* __restrict__ W1t = net.W1t;
    const float* __restrict__ b1 = net.b1;
    for (int j = 0; j < H; ++j) {
        const float* __restrict__ w = W1t + j * H;
        float av = 0.0f, ad = 0.0f, add = 0.0f;
#pragma unroll
        for (int a = 0; a < H; ++a) {
            av = __builtin_fmaf(h[a].v, w[a], av);
            ad = __builtin_fmaf(h[a].d, w[a], ad);
            add = __builtin_fmaf(h[a].dd, w[a], add);
        }
        sset(scr, j, jtanh(J{av + b1[j], ad, add}));
    }
#pragma unroll
    for (int a = 0; a < H; ++a) h[a] = sget(scr, a);
}

__device__ __forceinline__ J out_jet(const NetPlain& net, const J (&h)[H], int d, int j) {
    const float* __restrict__ w = net.W2t + ((size_t)d * NBP + j) * H;
    float av = 0.0f, ad = 0.0f, add = 0.0f;
#pragma unroll
    for (int a = 0; a < H; ++a) {
        av = __builtin_fmaf(h[a].v, w[a], av);
        ad = __builtin_fmaf(h[a].d, w[a], ad);
        add = __builtin_fmaf(h[a].dd, w[a], add);
    }
    return J{av + net.b2[d * NBP + j], ad, add};
}

// calculate_bijection_params for dimension d into rows 0..nb
__device__ __forceinline__ void bijection_jet(const NetPlain& net, const J (&h)[H], int d, int nb, bool sigmoid, float* scr) {
    J ss = jc(0.0f);
    for (int j = 0; j < nb; ++j) {
        J v = out_jet(net, h, d, j);
        if (sigmoid) v = jsigmoid(v);
        sset(scr, j, v);
        ss = ss + v;
    }
    const J r = jrcp(ss);
    for (int j = 0; j < nb; ++j) sset(scr, j, sget(scr, j) * r);
}

template <int D>
__global__ __launch_bounds__(kEBlock) void k_energy(const ModelDev* __restrict__ mdp, const float* __restrict__ tabI4, const float* __restrict__ tabP3,
                                                    const float* __restrict__ xg, int64_t B, const Protons pr,
                                                    float* __restrict__ hpsi_out, float* __restrict__ psi_out, float* __restrict__ lap_out) {
    __shared__ float scr[3 * kRows * kEBlock];
    const ModelDev& md = *mdp;
    const float L = md.box_L, tol = 1e-7f;
    for (int64_t b = (int64_t)blockIdx.x * kEBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kEBlock) {
        float x[D];
#pragma unroll
        for (int d = 0; d < D; ++d) x[d] = xg[b * D + d];
        float lap = 0.0f, psi_v = 0.0f;
        for (int dir = 0; dir < D; ++dir) {
            J cur[D], nxt[D];
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = J{x[d], d == dir ? 1.0f : 0.0f, 0.0f};
            // ---- BoxTransformLayer (made.py:118-137, 156-183)
            J logdet = jc(0.0f);
            if (md.box_kind == WF_BOX_MEAN) {
                J s = jc(0.0f);
#pragma unroll
                for (int d = 0; d < D; ++d) s = s + cur[d];
                const J mean = s * (1.0f / (float)D);
                const J l = mean - cur[0];
                const J wd = cur[D - 1] - cur[0];
                J space = jc(2 * L);
#pragma unroll
                for (int i = 0; i < D - 1; ++i) {
                    const J diff = cur[i + 1] - cur[i];
                    nxt[i] = diff / (space + tol);
                    logdet = logdet - jlog(space + tol);
                    space = space - diff;
                }
                const J den = (2 * L - wd) + tol;
                nxt[D - 1] = ((mean + L) - l) / den;
                logdet = logdet - jlog(den);
            } else if (md.box_kind == WF_BOX_FIRST) {
                nxt[0] = (cur[0] + L) * (1.0f / (2 * L));
                J ls = jc(0.0f);
#pragma unroll
                for (int i = 1; i < D; ++i) nxt[i] = (cur[i] - cur[i - 1]) / ((L - cur[i - 1]) + tol);
#pragma unroll
                for (int i = 0; i < D - 1; ++i) ls = ls + jlog((L - cur[i]) + tol);
                logdet = jc(-logf(2 * L)) - ls;
            } else {
#pragma unroll
                for (int d = 0; d < D; ++d) nxt[d] = cur[d];
            }
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[d];
            // ---- IMADE layers (made.py:66-81)
            for (int l = 0; l < md.n_layers; ++l) {
                const NetPlain& net = md.nets[l];
                const SplineDev& sp = md.isp;
                const int nb = sp.nb;
                J h[H];
                hidden_jet<D>(net, cur, scr, h);
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    bijection_jet(net, h, d, nb, true, scr);
                    for (int j = 0; j < nb; ++j) sset(scr, j, sget(scr, j) + md.i_reg);
                    // remove_bias (isplines_jax.py:196-202)
                    for (int i = 0; i < sp.degree; ++i) {
                        const float f = (float)(i + 1) / (float)sp.degree;
                        sset(scr, i + 1, sget(scr, i + 1) * f);
                        sset(scr, nb - (i + 2), sget(scr, nb - (i + 2)) * f);
                    }
                    J ss = jc(0.0f);
                    for (int j = 0; j < nb; ++j) ss = ss + sget(scr, j);
                    const J r = jrcp(ss);
                    for (int j = 0; j < nb; ++j) sset(scr, j, sget(scr, j) * r);
                    enforce_bc_jet(sp, WF_SPLINE_I, scr, 0);
                    nxt[d] = spline_jet(tabI4, sp.n_mesh, 0, cur[d], scr, 0, nb);
                    const J dy = spline_jet(tabI4, sp.n_mesh, 1, cur[d], scr, 0, nb);
                    logdet = logdet + jlog(dy + 1e-7f);
                }
#pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = nxt[D - 1 - d];
            }
            // ---- psi head (wavefunctions.py:54-71)
            const NetPlain& net = md.nets[md.n_layers];
            const SplineDev& sp = md.psp;
            const int nb = sp.nb;
            J h[H];
            hidden_jet<D>(net, cur, scr, h);
            J prod = jc(1.0f);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                bijection_jet(net, h, d, nb, false, scr);
                enforce_bc_jet(sp, WF_SPLINE_B, scr, 0);
                // c = w @ ob_to_b; c /= |c|   (bsplines_jax.py:134-135)
                J ss = jc(0.0f);
                for (int j = 0; j < nb; ++j) {
                    J acc = jc(0.0f);
                    for (int a = 0; a < nb; ++a) acc = acc + sget(scr, a) * md.ob_to_b[a * NBP + j];
                    sset(scr, NBP + j, acc);
                    ss = ss + acc * acc;
                }
                const J r = jrcp(jsqrt(ss));
                for (int j = 0; j < nb; ++j) sset(scr, NBP + j, sget(scr, NBP + j) * r);
                J uc = cur[d];   // np.clip(u, 0, 1): derivative 1 inside, 0 where clipped
                if (uc.v < 0.0f) uc = jc(0.0f);
                else if (uc.v > 1.0f) uc = jc(1.0f);
                J v = spline_jet(tabP3, sp.n_mesh, 0, uc, scr, NBP, nb);
                if ((md.constrained_mask >> d) & 1u) v = v * 0.70710678118654752f;
                prod = prod * v;
            }
            const J psi = prod * jexp(logdet * 0.5f);
            lap += psi.dd;
            psi_v = psi.v;
        }
        // ---- potential (physics.py:60-76)
        float V = 0.0f;
        for (int p = 0; p < pr.n; ++p)
#pragma unroll
            for (int d = 0; d < D; ++d) V -= 1.0f / sqrtf(1.0f + (pr.pos[p] - x[d]) * (pr.pos[p] - x[d]));
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
            for (int j2 = 0; j2 < i; ++j2) V += 1.0f / sqrtf(1.0f + (x[i] - x[j2]) * (x[i] - x[j2]));
        hpsi_out[b] = -0.5f * lap + V * psi_v;
        if (psi_out) psi_out[b] = psi_v;
        if (lap_out) lap_out[b] = lap;
    }
}

}  // namespace

int launch_energy(const ModelDev& md, const ModelDev* md_dev, const float* tabI4, const float* tabP3, const float* x, int64_t B,
                  const Protons& pr, float* hpsi, float* psi, float* lap, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    int64_t blocks = (B + kEBlock - 1) / kEBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
#define CALL(DD) hipLaunchKernelGGL(k_energy<DD>, dim3((unsigned)blocks), dim3(kEBlock), 0, s, md_dev, tabI4, tabP3, x, B, pr, hpsi, psi, lap)
    switch (md.D) {
        case 2: CALL(2); break;
        case 3: CALL(3); break;
        case 4: CALL(4); break;
        default: return WF_ERR_UNSUPPORTED;
    }
#undef CALL
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

}  // namespace wf
