// wf_kernels_mfma.hip -- throughput kernel: one wave = one tile of 32 walkers, conditioner GEMMs on
// v_mfma_f32_32x32x2_f32 (exact fp32), everything else fused around them (gfx950 / CDNA4).
//
// Orientation.  Every dense layer is computed transposed, OUT[unit][walker] = W^T[unit][k] * IN[k][walker]:
//   * the MFMA's C/D layout puts the walker on the lane (column j = lane & 31) and the 32 output units of a
//     block in the 16 accumulator registers of the two lane halves (row = (r&3) + 8*(r>>2) + 4*(lane>>5));
//   * that is exactly the B-operand layout of the next layer's MFMA (B[k][j]: lane half h supplies k = h of
//     each K=2 step), so an accumulator register -- after tanh -- IS the next B operand: the whole
//     2 -> 64 -> 64 -> D*n_bases chain runs with no LDS transposes and no cross-lane traffic;
//   * the A operand (weights) is pre-permuted on the host into that k order and streamed from LDS with one
//     ds_read_b128 per four MFMAs (wf_model.cpp: build_mfma_image);
//   * the per-walker spline-weight post-processing (sigmoid, normalisations, bias removal, boundary
//     conditions) and the table lerp act on the 16 registers of a lane; the two halves of a walker are
//     combined with v_permlane32_swap.
// Algebra used (exact in real arithmetic, fewer roundings than the reference's sequence):
//   with q_j = (sigmoid(o_j) + reg * S1) * f_j * keep_j  (S1 = sum sigmoid, f = remove_bias factors, keep = 0 on
//   rows a {0: 0} / {0: 1} constraint zeroes), the reference's weights are c_j = q_j / sum(q); so
//   y = (sum_j q_j lerp_j(x)) / sum(q) and only two reductions are needed (made.py:66-79,
//   isplines_jax.py:158-202).  For the B-spline prior the two L2 normalisations and the division by the signed
//   sum collapse to psi_d = sign(sum o) * (c . lerp) / |c| with c = (o * keep) @ ob_to_b
//   (wavefunctions.py:40-46, bsplines_jax.py:127-137, 173-199).
// The table index arithmetic (floor/ceil of u * (n_mesh-1), isplines_jax.py:46-48) is kept verbatim.
//
// Tables stay in global memory (L2-resident, 256 B per mesh row); LDS holds every net's weights for the
// whole launch, so waves run free of barriers after the prologue.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "wf_internal.h"

namespace wf {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;


__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); v_exp_f32 is 2^x
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t + 1.0f), 1.0f);
}

__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// sum of the two lane halves (lane l and l^32), result in every lane
__device__ __forceinline__ float xhalf_sum(float v) {
    const unsigned u = __float_as_uint(v);
    const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

__device__ __forceinline__ f32x16 lds_load16(const float* p) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

__device__ __forceinline__ f32x16 glb_load16(const float* __restrict__ p) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

// row of accumulator register r in lane half h
__device__ __forceinline__ constexpr int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct Lerp {
    int il, ir, xl, xr;
    float t;  // (x - x_l/n) * n
};

__device__ __forceinline__ int wrap_clamp(int i, int n) {
    if (i < 0) i += n;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ Lerp make_lerp(float x, int n_mesh) {
    Lerp L;
    const int n_points = n_mesh - 1;
    const float xs = x * (float)n_points;
    L.xl = (int)floorf(xs);
    L.xr = (int)ceilf(xs);
    L.il = wrap_clamp(L.xl, n_mesh);
    L.ir = wrap_clamp(L.xr, n_mesh);
    const float dx = x - (float)L.xl / (float)n_points;
    L.t = dx * (float)n_points;
    return L;
}

// Hidden layers of one conditioner net for the wave's 32 walkers.  in[]: the D inputs of this lane's walker.
// net: LDS image (see build_mfma_image).  Result: h2[2] (64 units x 32 walkers) in accumulator layout.
template <int D>
__device__ __forceinline__ void hidden_layers(const float* net, const float (&in)[D], int lane, f32x16 (&h2)[2]) {
    constexpr int S0 = (D + 1) / 2;
    const int h = lane >> 5;
    const float* W0 = net;
    const float* b0 = W0 + 2 * S0 * 64;
    const float* W1 = b0 + 64;
    const float* b1 = W1 + 4096;
    f32x16 h1[2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        h1[ob] = lds_load16(b0 + (ob * 2 + h) * 16);
#pragma unroll
        for (int s = 0; s < S0; ++s) {
            const float a = W0[(ob * S0 + s) * 64 + lane];
            const float lo = in[2 * s];
            const float hi = (2 * s + 1 < D) ? in[(2 * s + 1 < D) ? 2 * s + 1 : D - 1] : 0.0f;
            const float b = h ? hi : lo;
            h1[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, h1[ob], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[ob][r] = fast_tanh(h1[ob][r]);
    }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        h2[ob] = lds_load16(b1 + (ob * 2 + h) * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(W1 + (((ob * 2 + t) * 4 + r4) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) h2[ob] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], h1[t][4 * r4 + e], h2[ob], 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) h2[ob][r] = fast_tanh(h2[ob][r]);
    }
}

// Output block of dimension d: o[basis row][walker] in accumulator layout.  Dimension 0 depends on no hidden
// unit (output degree -1, model_factory.py:15-18): bias only.
template <int D>
__device__ __forceinline__ f32x16 out_block(const float* net, const f32x16 (&h2)[2], int d, int lane) {
    constexpr int S0 = (D + 1) / 2;
    const int h = lane >> 5;
    const float* W2 = net + 2 * S0 * 64 + 64 + 4096 + 64;
    const float* b2 = W2 + (D - 1) * 2048;
    f32x16 o = lds_load16(b2 + (d * 2 + h) * 16);
    if (d > 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(W2 + ((((d - 1) * 2 + t) * 4 + r4) * 64 + lane) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], h2[t][4 * r4 + e], o, 0, 0, 0);
            }
    }
    return o;
}

// q_j = (sigmoid(o_j) + reg * S1) * fk_j on this lane's 16 rows; returns sum(q) over the walker's 32 rows.
__device__ __forceinline__ float spline_weights(f32x16& o, const float* fk_lds, float reg, int nb, int h) {
    const f32x16 fk = lds_load16(fk_lds + h * 16);
    float s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = row_of(r, 0) + 4 * h < nb ? fast_sigmoid(o[r]) : 0.0f;
        o[r] = v;
        s1 += v;
    }
    s1 = xhalf_sum(s1);
    const float rs = reg * s1;
    float sq = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o[r] = (o[r] + rs) * fk[r];
        sq += o[r];
    }
    return xhalf_sum(sq);
}

// sum_j q_j * lerp_j over the walker's rows for one table order; rows of this half at tl / tr
__device__ __forceinline__ float lerp_dot(const f32x16& q, const float* __restrict__ tl, const float* __restrict__ tr, float t) {
    const f32x16 a = glb_load16(tl), b = glb_load16(tr);
    float sa0 = 0.0f, sa1 = 0.0f, sb0 = 0.0f, sb1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        sa0 = __builtin_fmaf(q[r], a[r], sa0);
        sb0 = __builtin_fmaf(q[r], b[r], sb0);
        sa1 = __builtin_fmaf(q[r + 1], a[r + 1], sa1);
        sb1 = __builtin_fmaf(q[r + 1], b[r + 1], sb1);
    }
    const float A = sa0 + sa1, Bv = sb0 + sb1;
    return xhalf_sum(__builtin_fmaf(Bv - A, t, A));
}

template <int D, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void k_mfma(const MfmaDev* __restrict__ mp, int mode, const float* __restrict__ xg, int64_t B,
                                                      float* __restrict__ out, float* __restrict__ u_out, int32_t* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const MfmaDev& mm = *mp;
    // ---- prologue: stage every net's weight image into LDS (one pass, 16 B per lane)
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(mm.image);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        const int n4 = mm.image_floats >> 2;
        for (int i = threadIdx.x; i < n4; i += kWaves * 64) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (B + 31) >> 5;
    const int idx_stride = (mm.n_layers + 1) * D * 2;
    const float* consts = lds + mm.const_off;
    const float* fkI = consts;            // [2][16]
    const float* fkP = consts + 32;       // [2][16]
    const float* ob2b = consts + 64;      // [4][64][4]
    const float L = mm.box_L, tol = 1e-7f;

    for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kWaves) {
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        float cur[D], nxt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = xg[wl * D + d];
        int32_t* idx = (idx_out && valid && h == 0) ? idx_out + w * idx_stride : nullptr;

        // ---- BoxTransformLayer (made.py:118-137, 156-183)
        float logdet = 0.0f;
        if (mm.box_kind == WF_BOX_MEAN) {
            float s = 0.0f;
#pragma unroll
            for (int d = 0; d < D; ++d) s = s + cur[d];
            const float mean = s / (float)D;
            const float l = mean - cur[0];
            const float wd = cur[D - 1] - cur[0];
            float space_left = 2 * L;
#pragma unroll
            for (int i = 0; i < D - 1; ++i) {
                const float diff = cur[i + 1] - cur[i];
                nxt[i] = diff / (space_left + tol);
                logdet = logdet - fast_log(space_left + tol);
                space_left = space_left - diff;
            }
            nxt[D - 1] = (mean + L - l) / (2 * L - wd + tol);
            logdet = logdet - fast_log(2 * L - wd + tol);
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[d];
        } else if (mm.box_kind == WF_BOX_FIRST) {
            nxt[0] = (cur[0] + L) / (2 * L);
            float ls = 0.0f;
#pragma unroll
            for (int i = 1; i < D; ++i) nxt[i] = (cur[i] - cur[i - 1]) / (L - cur[i - 1] + tol);
#pragma unroll
            for (int i = 0; i < D - 1; ++i) ls = ls + fast_log(L - cur[i] + tol);
            logdet = -fast_log(2 * L) - ls;
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[d];
        }

        // ---- flow layers
        for (int l = 0; l < mm.n_layers; ++l) {
            const float* net = lds + mm.net_off[l];
            f32x16 h2[2];
            hidden_layers<D>(net, cur, lane, h2);
            if (mm.layer_kind == WF_LAYER_IMADE) {
                const int nb = mm.i_nb;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    f32x16 q = out_block<D>(net, h2, d, lane);
                    const float S = spline_weights(q, fkI, mm.i_reg, nb, h);
                    const Lerp Lp = make_lerp(cur[d], mm.n_mesh);
                    if (idx) { idx[(l * D + d) * 2] = Lp.xl; idx[(l * D + d) * 2 + 1] = Lp.xr; }
                    const float* tl = mm.tabI + ((size_t)Lp.il * 4 + h) * 16;   // [mesh][nd][h][16]
                    const float* tr = mm.tabI + ((size_t)Lp.ir * 4 + h) * 16;
                    const float ynum = lerp_dot(q, tl, tr, Lp.t);
                    const float dnum = lerp_dot(q, tl + 32, tr + 32, Lp.t);
                    const float rS = 1.0f / S;
                    nxt[d] = ynum * rS;
                    logdet = logdet + fast_log(dnum * rS + 1e-7f);
                }
            } else {
                // MADE (made.py:21-27): rows 0 / 1 of block d = log_weight / bias (lane half 0, registers 0 / 1)
                float ls = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const f32x16 o = out_block<D>(net, h2, d, lane);
                    const float lw = __shfl(o[0], j), bias = __shfl(o[1], j);
                    nxt[d] = (cur[d] - bias) * __expf(-lw);
                    ls = ls + lw;
                }
                logdet = logdet - ls;
            }
#pragma unroll
            for (int d = 0; d < D; ++d) cur[d] = nxt[D - 1 - d];  // Reverse (bijections.py:337-340)
        }

        // ---- density head
        float result = logdet;
        if (mode != 2) {
            if (mm.prior_kind == WF_PRIOR_WAVEFLOW) {
                const float* net = lds + mm.net_off[mm.n_layers];
                const int nb = mm.p_nb;
                f32x16 h2[2];
                hidden_layers<D>(net, cur, lane, h2);
                const f32x16 keep = lds_load16(fkP + h * 16);
                float lp = 0.0f, prod = 1.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    f32x16 o = out_block<D>(net, h2, d, lane);
                    float s1 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        o[r] = row_of(r, 0) + 4 * h < nb ? o[r] : 0.0f;
                        s1 += o[r];
                        o[r] = o[r] * keep[r];
                    }
                    s1 = xhalf_sum(s1);
                    // c = (o * keep) @ ob_to_b on the matrix pipe (K = 32)
                    f32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ob2b + (r4 * 64 + lane) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], o[4 * r4 + e], c, 0, 0, 0);
                    }
                    float n2 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) n2 = __builtin_fmaf(c[r], c[r], n2);
                    n2 = xhalf_sum(n2);
                    cur[d] = fminf(fmaxf(cur[d], 0.0f), 1.0f);
                    const Lerp Lp = make_lerp(cur[d], mm.n_mesh);
                    if (idx) { idx[(mm.n_layers * D + d) * 2] = Lp.xl; idx[(mm.n_layers * D + d) * 2 + 1] = Lp.xr; }
                    const float* tl = mm.tabP + ((size_t)Lp.il * 2 + h) * 16;   // [mesh][h][16]
                    const float* tr = mm.tabP + ((size_t)Lp.ir * 2 + h) * 16;
                    const float num = lerp_dot(c, tl, tr, Lp.t);
                    float v = num * __builtin_amdgcn_rsqf(n2);
                    v = s1 < 0.0f ? -v : v;
                    const bool constrained = (mm.constrained_mask >> d) & 1u;
                    if (mode == 0) {
                        float pr = v * v;
                        if (constrained) pr = pr * 0.5f;
                        lp = lp + fast_log(pr + 1e-7f);
                    } else {
                        if (constrained) v = v * 0.70710678118654752f;
                        prod = prod * v;
                    }
                }
                result = mode == 0 ? lp + logdet : prod * __expf(0.5f * logdet);
            } else if (mm.prior_kind == WF_PRIOR_MFLOW) {
                const float* net = lds + mm.net_off[mm.n_layers];
                const int nb = mm.p_nb;
                f32x16 h2[2];
                hidden_layers<D>(net, cur, lane, h2);
                float lp = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    f32x16 q = out_block<D>(net, h2, d, lane);
                    const float S = spline_weights(q, fkP, 0.0f, nb, h);
                    cur[d] = fminf(fmaxf(cur[d], 0.0f), 1.0f);
                    const Lerp Lp = make_lerp(cur[d], mm.n_mesh);
                    if (idx) { idx[(mm.n_layers * D + d) * 2] = Lp.xl; idx[(mm.n_layers * D + d) * 2 + 1] = Lp.xr; }
                    const float* tl = mm.tabP + ((size_t)Lp.il * 2 + h) * 16;
                    const float* tr = mm.tabP + ((size_t)Lp.ir * 2 + h) * 16;
                    const float v = lerp_dot(q, tl, tr, Lp.t) / S;
                    lp = lp + fast_log(v + 1e-7f);
                }
                result = lp + logdet;
            } else if (mm.prior_kind == WF_PRIOR_UNIFORM) {
#pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = fminf(fmaxf(cur[d], 0.0f), 1.0f);
                result = logdet;
            } else {
                float lp = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float z = cur[d] + mm.normal_offset;
                    lp = lp + (1.8378770664093453f + z * z) * -0.5f;
                }
                result = lp + logdet;
            }
        }
        if (valid && h == 0) {
            out[w] = result;
            if (u_out) {
#pragma unroll
                for (int d = 0; d < D; ++d) u_out[w * D + d] = cur[d];
            }
        }
    }
}

template <int D, int kWaves>
int launch_dw(const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, hipStream_t s) {
    static int configured_bytes = -1;
    if (lds_bytes > configured_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma<D, kWaves>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) {
            set_hip_error((int)e);
            return WF_ERR_HIP;
        }
        configured_bytes = lds_bytes;
    }
    const int64_t n_tiles = (B + 31) / 32;
    int64_t grid = (n_tiles + kWaves - 1) / kWaves;
    if (grid > 256) grid = 256;  // one persistent workgroup per CU
    hipLaunchKernelGGL((k_mfma<D, kWaves>), dim3((unsigned)grid), dim3(kWaves * 64), lds_bytes, s, mdev, mode, x, B, out, u, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

int waves_per_group() {
    static int w = [] {
        const char* e = getenv("WF_MFMA_WAVES");  // tuning knob: 8, 12 or 16 waves per workgroup
        const int v = e ? atoi(e) : 0;
        return (v == 8 || v == 12 || v == 16) ? v : 12;
    }();
    return w;
}

template <int D>
int launch_d(const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, hipStream_t s) {
    switch (waves_per_group()) {
        case 8: return launch_dw<D, 8>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        case 16: return launch_dw<D, 16>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        default: return launch_dw<D, 12>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
    }
}

}  // namespace

int launch_mfma(int D, const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx,
                void* stream) {
    hipStream_t s = (hipStream_t)stream;
    switch (D) {
        case 2: return launch_d<2>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        case 3: return launch_d<3>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        case 4: return launch_d<4>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        default: return WF_ERR_UNSUPPORTED;
    }
}

}  // namespace wf
