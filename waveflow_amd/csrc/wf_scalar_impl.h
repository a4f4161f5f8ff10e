// wf_scalar_impl.h -- the "one lane = one walker" kernels as templates; instantiated per shape in wf_scalar_inst_*.hip so
// that the shapes compile in parallel.  See wf_kernels_scalar.hip for the notes.
#pragma once
#include <hip/hip_runtime.h>

#include "wf_internal.h"

namespace wf {
namespace scalar {

constexpr int H = kHidden;

// NBP = padded bases per dimension (32 or 64); the workgroup is 256 / (NBP / 32) lanes so that the thread-private
// LDS columns (max(64 hidden units, 2 * NBP) rows) stay at 64 KB per workgroup.
template <int NBP>
struct Cfg {
    static constexpr int kBlock = NBP == 32 ? 256 : 128;
    static constexpr int kRows = 2 * NBP > H ? 2 * NBP : H;
};

#define SCR(j) scr[(j) * (int)blockDim.x + threadIdx.x]

struct Lerp {
    int il, ir;   // wrapped + clamped gather indices
    int xl, xr;   // as computed (reported as "bin index")
    float dx, n;
};

__device__ __forceinline__ int wrap_clamp(int i, int n) {
    if (i < 0) i += n;  // jnp indexing: negative indices wrap once ...
    return min(max(i, 0), n - 1);  // ... and out-of-bounds gathers clamp
}

__device__ __forceinline__ Lerp make_lerp(float x, int n_mesh) {
    Lerp L;
    const int n_points = n_mesh - 1;
    const float xs = x * (float)n_points;
    L.xl = (int)floorf(xs);
    L.xr = (int)ceilf(xs);
    L.il = wrap_clamp(L.xl, n_mesh);
    L.ir = wrap_clamp(L.xr, n_mesh);
    L.dx = x - (float)L.xl / (float)n_points;
    L.n = (float)n_points;
    return L;
}

// sum_j c_j * X_cached(x, j), j ascending; c_j = SCR(row0 + j); tab = one derivative order, [n_mesh][NBP]
template <int NBP>
__device__ __forceinline__ float spline_dot(const float* __restrict__ tab, const Lerp& L, const float* scr, int row0, int nb) {
    const float4* rl = reinterpret_cast<const float4*>(tab + (size_t)L.il * NBP);
    const float4* rr = reinterpret_cast<const float4*>(tab + (size_t)L.ir * NBP);
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < NBP / 4; ++q) {
        const float4 a = rl[q], b = rr[q];
        const float yl[4] = {a.x, a.y, a.z, a.w}, yr[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 4 * q + e;
            if (j < nb) {
                const float slope = (yr[e] - yl[e]) * L.n;
                const float y = yl[e] + slope * L.dx;
                acc = acc + SCR(row0 + j) * y;
            }
        }
    }
    return acc;
}

// kind: WF_SPLINE_I / _M / _B
__device__ __forceinline__ void enforce_bc(const SplineDev& s, int kind, float* scr, int row0) {
    const int nb = s.nb;
    for (int p = 0; p < s.n_left; ++p) {
        const int nd = s.left_nd[p];
        float sum = 0.0f;
        for (int j = 0; j < nd; ++j) sum = sum + s.left_prev[p][j] * SCR(row0 + j);
        SCR(row0 + nd) = (s.left_val[p] - sum) / s.left_value[p];
    }
    for (int p = 0; p < s.n_right; ++p) {
        const int nd = s.right_nd[p];
        if (kind == WF_SPLINE_I && nd == 0) {
            SCR(row0 + nb - 1) = 0.0f;
            continue;
        }
        float sum = 0.0f;
        for (int j = 0; j < nd; ++j) sum = sum + s.right_prev[p][j] * SCR(row0 + nb - 1 - j);
        SCR(row0 + nb - nd - 1) = (s.right_val[p] - sum) / s.right_value[p];
    }
    float ss = 0.0f;
    if (kind == WF_SPLINE_B) {
        for (int j = 0; j < nb; ++j) ss = ss + SCR(row0 + j) * SCR(row0 + j);
        ss = sqrtf(ss);
    } else {
        for (int j = 0; j < nb; ++j) ss = ss + SCR(row0 + j);
    }
    for (int j = 0; j < nb; ++j) SCR(row0 + j) = SCR(row0 + j) / ss;
}

__device__ __forceinline__ void remove_bias(int kind, int k, int nb, float* scr) {
    for (int i = 0; i < k; ++i) {
        const int a = kind == WF_SPLINE_I ? i + 1 : i;
        const int b = kind == WF_SPLINE_I ? nb - (i + 2) : nb - (i + 1);
        SCR(a) = SCR(a) * (float)(i + 1) / (float)k;
        SCR(b) = SCR(b) * (float)(i + 1) / (float)k;
    }
    float ss = 0.0f;
    for (int j = 0; j < nb; ++j) ss = ss + SCR(j);
    for (int j = 0; j < nb; ++j) SCR(j) = SCR(j) / ss;
}

// two masked tanh layers; result h[64] in registers
template <int D>
__device__ __forceinline__ void hidden_layers(const NetPlain& net, const float (&x)[D], float* scr, float (&h)[H]) {
    const float* __restrict__ W0 = net.W0;
    const float* __restrict__ b0 = net.b0;
    for (int j = 0; j < H; ++j) {
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < D; ++a) acc = __builtin_fmaf(x[a], W0[a * H + j], acc);
        SCR(j) = tanhf(acc + b0[j]);
    }
#pragma unroll
    for (int a = 0; a < H; ++a) h[a] = SCR(a);
    const float* __restrict__ W1t = net.W1t;
    const float* __restrict__ b1 = net.b1;
    for (int j = 0; j < H; ++j) {
        const float* __restrict__ w = W1t + j * H;
        float acc = 0.0f;
#pragma unroll
        for (int a = 0; a < H; ++a) acc = __builtin_fmaf(h[a], w[a], acc);
        SCR(j) = tanhf(acc + b1[j]);
    }
#pragma unroll
    for (int a = 0; a < H; ++a) h[a] = SCR(a);
}

template <int NBP>
__device__ __forceinline__ float out_unit(const NetPlain& net, const float (&h)[H], int d, int j) {
    const float* __restrict__ w = net.W2t + ((size_t)d * NBP + j) * H;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < H; ++a) acc = __builtin_fmaf(h[a], w[a], acc);
    return acc + net.b2[d * NBP + j];
}

// calculate_bijection_params for dimension d into SCR(0..nb).  gate (set_nn_output_grad_to_zero, model_factory.py:64-67):
// bij = g * head(o) + z with g = prod_{i<d} x_i^3 of the conditioner's input (the caller's running product) and z = net.zero[d][.]
template <int NBP>
__device__ __forceinline__ void bijection_params(const NetPlain& net, const float (&h)[H], int d, int nb, bool sigmoid, float* scr,
                                                 bool gate = false, float g = 1.0f) {
    float ss = 0.0f;
    for (int j = 0; j < nb; ++j) {
        float v = out_unit<NBP>(net, h, d, j);
        if (sigmoid) v = 1.0f / (1.0f + expf(-v));
        if (gate) v = g * v + net.zero[d * NBP + j];
        SCR(j) = v;
        ss = ss + v;
    }
    for (int j = 0; j < nb; ++j) SCR(j) = SCR(j) / ss;
}

template <int D, int NBP>
__device__ __forceinline__ float imade_direct(const ModelDev& md, const NetPlain& net, const float (&x)[D], float (&y)[D],
                                              float* scr, int32_t* idx) {
    float h[H];
    hidden_layers<D>(net, x, scr, h);
    const SplineDev& sp = md.isp;
    const int nb = sp.nb;
    const float* tab0 = sp.tab;
    const float* tab1 = sp.tab + (size_t)sp.n_mesh * NBP;
    float ld = 0.0f, g = 1.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (d > 0) g = g * (x[d - 1] * x[d - 1] * x[d - 1]);
        bijection_params<NBP>(net, h, d, nb, true, scr, md.i_gate != 0, g);
        for (int j = 0; j < nb; ++j) SCR(j) = SCR(j) + md.i_reg;
        remove_bias(WF_SPLINE_I, sp.degree, nb, scr);
        enforce_bc(sp, WF_SPLINE_I, scr, 0);
        const Lerp L = make_lerp(x[d], sp.n_mesh);
        if (idx) { idx[2 * d] = L.xl; idx[2 * d + 1] = L.xr; }
        y[d] = spline_dot<NBP>(tab0, L, scr, 0, nb);
        const float dy = spline_dot<NBP>(tab1, L, scr, 0, nb);
        ld = ld + logf(dy + 1e-7f);
    }
    return ld;
}

template <int D, int NBP>
__device__ __forceinline__ float made_direct(const NetPlain& net, const float (&x)[D], float (&y)[D], float* scr) {
    float h[H];
    hidden_layers<D>(net, x, scr, h);
    float ls = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float lw = out_unit<NBP>(net, h, d, 0), bias = out_unit<NBP>(net, h, d, 1);
        y[d] = (x[d] - bias) * expf(-lw);
        ls = ls + lw;
    }
    return -ls;
}

template <int D>
__device__ __forceinline__ float box_direct(const ModelDev& md, const float (&x)[D], float (&u)[D]) {
    const float L = md.box_L, tol = 1e-7f;
    if (md.box_kind == WF_BOX_MEAN) {
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < D; ++d) s = s + x[d];
        const float mean = s / (float)D;
        const float l = mean - x[0];
        const float w = x[D - 1] - x[0];
        float space_left = 2 * L, ld = 0.0f;
#pragma unroll
        for (int i = 0; i < D - 1; ++i) {
            const float diff = x[i + 1] - x[i];
            u[i] = diff / (space_left + tol);
            ld = ld - logf(space_left + tol);
            space_left = space_left - diff;
        }
        u[D - 1] = (mean + L - l) / (2 * L - w + tol);
        return ld - logf(2 * L - w + tol);
    }
    u[0] = (x[0] + L) / (2 * L);
    float ls = 0.0f;
#pragma unroll
    for (int i = 1; i < D; ++i) u[i] = (x[i] - x[i - 1]) / (L - x[i - 1] + tol);
#pragma unroll
    for (int i = 0; i < D - 1; ++i) ls = ls + logf(L - x[i] + tol);
    return -logf(2 * L) - ls;
}

__device__ __forceinline__ float clip01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

template <int D, int NBP>
__global__ __launch_bounds__(Cfg<NBP>::kBlock) void k_eval(const ModelDev* __restrict__ mdp, int mode, const float* __restrict__ xg, int64_t B,
                                                 float* __restrict__ out, float* __restrict__ u_out, int32_t* __restrict__ idx_out) {
    constexpr int kBlock = Cfg<NBP>::kBlock;
    __shared__ float scr[Cfg<NBP>::kRows * kBlock];
    const ModelDev& md = *mdp;
    const int idx_stride = (md.n_layers + 1) * D * 2;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        float cur[D], nxt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = xg[b * D + d];
        int32_t* idx = idx_out ? idx_out + b * idx_stride : nullptr;
        float logdet = 0.0f;
        if (md.box_kind != WF_BOX_NONE) {
            logdet = logdet + box_direct<D>(md, cur, nxt);
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[d];
        }
        for (int l = 0; l < md.n_layers; ++l) {
            float ld;
            if (md.layer_kind == WF_LAYER_IMADE) ld = imade_direct<D, NBP>(md, md.nets[l], cur, nxt, scr, idx ? idx + l * D * 2 : nullptr);
            else ld = made_direct<D, NBP>(md.nets[l], cur, nxt, scr);
            logdet = logdet + ld;
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[D - 1 - d];  // Reverse
        }
        float result = logdet;
        if (mode != 2) {
            if (md.prior_kind == WF_PRIOR_WAVEFLOW) {
                const NetPlain& net = md.nets[md.n_layers];
                const SplineDev& sp = md.psp;
                const int nb = sp.nb;
                float h[H];
                hidden_layers<D>(net, cur, scr, h);
                float lp = 0.0f, prod = 1.0f, g = 1.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    bijection_params<NBP>(net, h, d, nb, false, scr, md.p_gate != 0, g);
                    enforce_bc(sp, WF_SPLINE_B, scr, 0);
                    g = g * (cur[d] * cur[d] * cur[d]);   // the gate sees the conditioner's input: the unclipped u
                    cur[d] = clip01(cur[d]);
                    // BSpline_fun.apply_fun: c = w @ ob_to_b; c /= |c|  (bsplines_jax.py:134-135)
                    float ss = 0.0f;
                    for (int j = 0; j < nb; ++j) {
                        float acc = 0.0f;
                        for (int a = 0; a < nb; ++a) acc = acc + SCR(a) * md.ob_to_b[a * NBP + j];
                        SCR(NBP + j) = acc;
                        ss = ss + acc * acc;
                    }
                    const float nrm = sqrtf(ss);
                    for (int j = 0; j < nb; ++j) SCR(NBP + j) = SCR(NBP + j) / nrm;
                    const Lerp L = make_lerp(cur[d], sp.n_mesh);
                    if (idx) { idx[(md.n_layers * D + d) * 2] = L.xl; idx[(md.n_layers * D + d) * 2 + 1] = L.xr; }
                    float v = spline_dot<NBP>(sp.tab, L, scr, NBP, nb);
                    const bool constrained = (md.constrained_mask >> d) & 1u;
                    if (mode == 0) {
                        float pr = v * v;
                        if (constrained) pr = pr / 2;
                        lp = lp + logf(pr + 1e-7f);
                    } else {
                        if (constrained) v = v / sqrtf(2.0f);
                        prod = prod * v;
                    }
                }
                result = mode == 0 ? lp + logdet : prod * expf(0.5f * logdet);
            } else if (md.prior_kind == WF_PRIOR_MFLOW) {
                const NetPlain& net = md.nets[md.n_layers];
                const SplineDev& sp = md.psp;
                const int nb = sp.nb;
                float h[H];
                hidden_layers<D>(net, cur, scr, h);
                float lp = 0.0f, g = 1.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    bijection_params<NBP>(net, h, d, nb, true, scr, md.p_gate != 0, g);
                    remove_bias(WF_SPLINE_M, sp.degree, nb, scr);
                    enforce_bc(sp, WF_SPLINE_M, scr, 0);
                    g = g * (cur[d] * cur[d] * cur[d]);
                    cur[d] = clip01(cur[d]);
                    const Lerp L = make_lerp(cur[d], sp.n_mesh);
                    if (idx) { idx[(md.n_layers * D + d) * 2] = L.xl; idx[(md.n_layers * D + d) * 2 + 1] = L.xr; }
                    const float v = spline_dot<NBP>(sp.tab, L, scr, 0, nb);
                    lp = lp + logf(v + 1e-7f);
                }
                result = lp + logdet;
            } else if (md.prior_kind == WF_PRIOR_UNIFORM) {
#pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = clip01(cur[d]);
                result = 0.0f + logdet;
            } else {
                float lp = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float z = cur[d] + md.normal_offset;
                    lp = lp + (1.8378770664093453f + z * z) / -2.0f;
                }
                result = lp + logdet;
            }
        }
        out[b] = result;
        if (u_out) {
#pragma unroll
            for (int d = 0; d < D; ++d) u_out[b * D + d] = cur[d];
        }
    }
}

template <int D, int NBP>
__global__ __launch_bounds__(Cfg<NBP>::kBlock) void k_layer(const ModelDev* __restrict__ mdp, int layer, const float* __restrict__ ug, int64_t B,
                                                  float* __restrict__ yg, float* __restrict__ ldg, int32_t* __restrict__ idx_out) {
    constexpr int kBlock = Cfg<NBP>::kBlock;
    __shared__ float scr[Cfg<NBP>::kRows * kBlock];
    const ModelDev& md = *mdp;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        float cur[D], nxt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = ug[b * D + d];
        float ld;
        if (md.layer_kind == WF_LAYER_IMADE) ld = imade_direct<D, NBP>(md, md.nets[layer], cur, nxt, scr, idx_out ? idx_out + b * D * 2 : nullptr);
        else ld = made_direct<D, NBP>(md.nets[layer], cur, nxt, scr);
        ldg[b] = ld;
#pragma unroll
        for (int d = 0; d < D; ++d) yg[b * D + d] = nxt[d];
    }
}


// ------------------------------------------------------------------------------------------------------------
// Inverse direction: Serial.inverse_fun (bijections.py:462-463) and the samplers of
// Waveflow (wavefunctions.py:74-107), MFlow (distributions.py:165-190) and Flow (distributions.py:104-108).
// PRNG: the reference draws with JAX's threefry; here Philox4x32-10 keyed by (seed, walker) -- parity unpinned.

struct Philox {
    unsigned key0, key1, c0, c1, c2, c3;
    unsigned out[4];
    int have;
    __device__ Philox(unsigned long long seed, unsigned long long stream) : key0((unsigned)seed), key1((unsigned)(seed >> 32)), c0(0), c1(0), c2((unsigned)stream), c3((unsigned)(stream >> 32)), have(0) {}
    __device__ void round(unsigned& a0, unsigned& a1, unsigned& a2, unsigned& a3, unsigned k0, unsigned k1) {
        const unsigned long long p0 = 0xD2511F53ull * a0, p1 = 0xCD9E8D57ull * a2;
        const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
        a0 = h1 ^ a1 ^ k0; a1 = l1; a2 = h0 ^ a3 ^ k1; a3 = l0;
    }
    __device__ void refill() {
        unsigned a0 = c0, a1 = c1, a2 = c2, a3 = c3, k0 = key0, k1 = key1;
#pragma unroll
        for (int r = 0; r < 10; ++r) { round(a0, a1, a2, a3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        out[0] = a0; out[1] = a1; out[2] = a2; out[3] = a3;
        if (++c0 == 0) ++c1;
        have = 4;
    }
    __device__ float uniform() {   // [0, 1) with 24 random bits, like jax.random.uniform's fp32 mantissa fill
        if (!have) refill();
        return (float)(out[--have] >> 8) * (1.0f / 16777216.0f);
    }
};

// helpers.binary_search (utils/helpers.py:150-166) on spline(w, x) - y over [0, 1]; weights in SCR(0..nb)
template <int NBP>
__device__ __forceinline__ float ispline_reverse(const SplineDev& sp, const float* scr, float y, float tol) {
    float low = 0.0f, high = 1.0f;
    for (int it = 0; it < 64; ++it) {   // the loop ends after ~log2(1/tol) halvings; 64 bounds it for any tol
        const float mid = 0.5f * (low + high);
        if (!((low + tol / 2 < mid) && (mid < high - tol / 2))) break;
        const Lerp L = make_lerp(mid, sp.n_mesh);
        const float f = spline_dot<NBP>(sp.tab, L, scr, 0, sp.nb) - y;
        if (f > 0) high = mid; else low = mid;
    }
    return low;
}

// IMADE.inverse_fun (made.py:85-100).  exact == 0 reproduces the reference: the conditioner sees `in` (the values
// being inverted) for every column; exact != 0 conditions on the reconstructed prefix (true inverse of direct_fun).
template <int D, int NBP>
__device__ __forceinline__ void imade_inverse(const ModelDev& md, const NetPlain& net, const float (&in)[D], float (&out)[D], float* scr,
                                              int exact) {
    const SplineDev& sp = md.isp;
    const int nb = sp.nb;
    float h[H];
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = 0.0f;
    float g = 1.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (d == 0 || exact) {
            if (exact) hidden_layers<D>(net, out, scr, h); else hidden_layers<D>(net, in, scr, h);
        }
        if (d > 0) { const float xp = exact ? out[d - 1] : in[d - 1]; g = g * (xp * xp * xp); }
        bijection_params<NBP>(net, h, d, nb, true, scr, md.i_gate != 0, g);
        for (int j = 0; j < nb; ++j) SCR(j) = SCR(j) + md.i_reg;
        remove_bias(WF_SPLINE_I, sp.degree, nb, scr);
        enforce_bc(sp, WF_SPLINE_I, scr, 0);
        out[d] = ispline_reverse<NBP>(sp, scr, in[d], md.reverse_tol);
    }
}

// MADE.inverse_fun (made.py:29-37)
template <int D, int NBP>
__device__ __forceinline__ void made_inverse(const NetPlain& net, const float (&in)[D], float (&out)[D], float* scr) {
    float h[H];
#pragma unroll
    for (int d = 0; d < D; ++d) out[d] = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
        hidden_layers<D>(net, out, scr, h);
        const float lw = out_unit<NBP>(net, h, c, 0), bias = out_unit<NBP>(net, h, c, 1);
        out[c] = in[c] * expf(lw) + bias;
    }
}

// BoxTransformLayer.reverse_fun_mean (made.py:186-197) / reverse_fun_first (made.py:139-154)
template <int D>
__device__ __forceinline__ void box_reverse(const ModelDev& md, const float (&u)[D], float (&x)[D]) {
    const float L = md.box_L;
    if (md.box_kind == WF_BOX_MEAN && D > 2) {
        // The reference's reverse_fun_mean is written for two particles (made.py:186-197, "TODO" at :188).  For D > 2 this is the
        // inverse of direct_fun_mean (made.py:156-183) itself: differences from the shrinking remaining space, then the offset of
        // the first particle from the last coordinate.
        const float tol = 1e-7f;
        float o[D];
        float space = 2 * L, c = 0.0f;
        o[0] = 0.0f;
#pragma unroll
        for (int i = 0; i < D - 1; ++i) {
            const float diff = u[i] * (space + tol);
            space = space - diff;
            c = c + diff;
            o[i + 1] = c;
        }
        const float w = o[D - 1];                          // x[D-1] - x[0]
        const float x0 = u[D - 1] * (2 * L - w + tol) - L; // u[D-1] = (mean + L - l) / (2L - w + tol) with mean - l = x[0]
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = x0 + o[i];
    } else if (md.box_kind == WF_BOX_MEAN) {
        float o[D];
        float c = 0.0f, s = 0.0f;
        o[0] = 0.0f;
#pragma unroll
        for (int i = 0; i < D - 1; ++i) { c = c + u[i]; o[i + 1] = c; }
#pragma unroll
        for (int i = 0; i < D; ++i) s = s + o[i];
        const float mean = s / (float)D;
        const float w = o[D - 1];
        const float pm = u[D - 1] * (1 - w) - (0.5f - mean);
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = (o[i] - mean + pm) * 2 * L;
    } else {
        x[0] = (u[0] - 0.5f) * 2 * L;
#pragma unroll
        for (int i = 1; i < D; ++i) x[i] = u[i] * (L - x[i - 1]) + x[i - 1];
    }
}

template <int D, int NBP>
__device__ __forceinline__ void serial_inverse(const ModelDev& md, float (&cur)[D], float* scr, int exact) {
    float nxt[D];
    for (int l = md.n_layers - 1; l >= 0; --l) {
#pragma unroll
        for (int d = 0; d < D; ++d) nxt[d] = cur[D - 1 - d];   // Reverse.inverse_fun
        if (md.layer_kind == WF_LAYER_IMADE) imade_inverse<D, NBP>(md, md.nets[l], nxt, cur, scr, exact);
        else made_inverse<D, NBP>(md.nets[l], nxt, cur, scr);
    }
    if (md.box_kind != WF_BOX_NONE) {
        box_reverse<D>(md, cur, nxt);
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = nxt[d];
    }
}

template <int D, int NBP>
__global__ __launch_bounds__(Cfg<NBP>::kBlock) void k_inverse(const ModelDev* __restrict__ mdp, const float* __restrict__ ug, int64_t B,
                                                              float* __restrict__ xg, int exact) {
    constexpr int kBlock = Cfg<NBP>::kBlock;
    __shared__ float scr[Cfg<NBP>::kRows * kBlock];
    const ModelDev& md = *mdp;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        float cur[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = ug[b * D + d];
        serial_inverse<D, NBP>(md, cur, scr, exact);
#pragma unroll
        for (int d = 0; d < D; ++d) xg[b * D + d] = cur[d];
    }
}

template <int D, int NBP>
__global__ __launch_bounds__(Cfg<NBP>::kBlock) void k_sample(const ModelDev* __restrict__ mdp, unsigned long long seed, int64_t B,
                                                             float* __restrict__ xg, float* __restrict__ latent, int exact) {
    constexpr int kBlock = Cfg<NBP>::kBlock;
    __shared__ float scr[Cfg<NBP>::kRows * kBlock];
    const ModelDev& md = *mdp;
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < B; b += (int64_t)gridDim.x * kBlock) {
        Philox rng(seed, (unsigned long long)b);
        float cur[D];
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
            const NetPlain& net = md.nets[md.n_layers];
            const SplineDev& sp = md.psp;
            const int nb = sp.nb;
            const bool wavefn = md.prior_kind == WF_PRIOR_WAVEFLOW;
            float g = 1.0f;
#pragma unroll
            for (int col = 0; col < D; ++col) {
                float h[H];
                hidden_layers<D>(net, cur, scr, h);   // conditioner on the columns drawn so far, zeros elsewhere
                if (col > 0) g = g * (cur[col - 1] * cur[col - 1] * cur[col - 1]);
                bijection_params<NBP>(net, h, col, nb, !wavefn, scr, md.p_gate != 0, g);
                float ymax = 0.0f;
                int row0 = 0;
                if (wavefn) {
                    enforce_bc(sp, WF_SPLINE_B, scr, 0);
                    // sample_fun (bsplines_jax.py:144-171): obw = normalised(w @ ob_to_b); ymax = max((obw @ b_to_ob)^2)
                    float ss = 0.0f;
                    for (int j = 0; j < nb; ++j) {
                        float acc = 0.0f;
                        for (int a = 0; a < nb; ++a) acc = acc + SCR(a) * md.ob_to_b[a * NBP + j];
                        SCR(NBP + j) = acc;
                        ss = ss + acc * acc;
                    }
                    const float nrm = sqrtf(ss);
                    for (int j = 0; j < nb; ++j) SCR(NBP + j) = SCR(NBP + j) / nrm;
                    for (int j = 0; j < nb; ++j) {
                        float acc = 0.0f;
                        for (int a = 0; a < nb; ++a) acc = acc + SCR(NBP + a) * md.b_to_ob[a * NBP + j];
                        ymax = fmaxf(ymax, acc * acc);
                    }
                    row0 = NBP;
                } else {
                    remove_bias(WF_SPLINE_M, sp.degree, nb, scr);
                    enforce_bc(sp, WF_SPLINE_M, scr, 0);
                    float mx = SCR(0);
                    for (int j = 1; j < nb; ++j) mx = fmaxf(mx, SCR(j));
                    ymax = mx * (float)(nb + sp.degree);   // params.max() * n_knots (msplines_jax.py:147-150)
                }
                // rejection sampling (bounded: a pathological density cannot hang the GPU; a walker that exhausts the bound is
                // written as NaN -- visible in every later reduction -- rather than parked at a plausible 0.5)
                float xs = __builtin_nanf("");
                for (int it = 0; it < 100000; ++it) {
                    const float xc = rng.uniform(), yc = rng.uniform() * ymax;
                    const Lerp L = make_lerp(xc, sp.n_mesh);
                    float v = spline_dot<NBP>(sp.tab, L, scr, row0, nb);
                    if (wavefn) v = v * v;
                    if (yc < v) { xs = xc; break; }
                }
                cur[col] = xs;
            }
        }
        if (latent) {
#pragma unroll
            for (int d = 0; d < D; ++d) latent[b * D + d] = cur[d];
        }
        serial_inverse<D, NBP>(md, cur, scr, exact);
#pragma unroll
        for (int d = 0; d < D; ++d) xg[b * D + d] = cur[d];
    }
}


template <int NBP>
int grid_for(int64_t B) {
    constexpr int kBlock = Cfg<NBP>::kBlock;
    int64_t n = (B + kBlock - 1) / kBlock;
    const int64_t cap = 256 * 8;  // 256 CUs, grid-stride beyond that
    if (n > cap) n = cap;
    return (int)n;
}

template <int D, int NBP>
int run_eval(const ModelDev* mdp, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, hipStream_t s) {
    hipLaunchKernelGGL((k_eval<D, NBP>), dim3(grid_for<NBP>(B)), dim3(Cfg<NBP>::kBlock), 0, s, mdp, mode, x, B, out, u, idx);
    return 0;
}
template <int D, int NBP>
int run_layer(const ModelDev* mdp, int layer, const float* u_in, int64_t B, float* y, float* ld, int32_t* idx, hipStream_t s) {
    hipLaunchKernelGGL((k_layer<D, NBP>), dim3(grid_for<NBP>(B)), dim3(Cfg<NBP>::kBlock), 0, s, mdp, layer, u_in, B, y, ld, idx);
    return 0;
}
template <int D, int NBP>
int run_inverse(const ModelDev* mdp, const float* u, int64_t B, float* x, int exact, hipStream_t s) {
    hipLaunchKernelGGL((k_inverse<D, NBP>), dim3(grid_for<NBP>(B)), dim3(Cfg<NBP>::kBlock), 0, s, mdp, u, B, x, exact);
    return 0;
}
template <int D, int NBP>
int run_sample(const ModelDev* mdp, unsigned long long seed, int64_t B, float* x, float* latent, int exact, hipStream_t s) {
    hipLaunchKernelGGL((k_sample<D, NBP>), dim3(grid_for<NBP>(B)), dim3(Cfg<NBP>::kBlock), 0, s, mdp, seed, B, x, latent, exact);
    return 0;
}

#define WF_SCALAR_SHAPE(KW, DD, NN)                                                                                                    \
    KW template int run_eval<DD, NN>(const ModelDev*, int, const float*, int64_t, float*, float*, int32_t*, hipStream_t);              \
    KW template int run_layer<DD, NN>(const ModelDev*, int, const float*, int64_t, float*, float*, int32_t*, hipStream_t);             \
    KW template int run_inverse<DD, NN>(const ModelDev*, const float*, int64_t, float*, int, hipStream_t);                             \
    KW template int run_sample<DD, NN>(const ModelDev*, unsigned long long, int64_t, float*, float*, int, hipStream_t);

}  // namespace scalar
}  // namespace wf
