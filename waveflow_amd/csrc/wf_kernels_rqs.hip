// wf_kernels_rqs.hip -- rational-quadratic spline bijector (Durkan et al.), elementwise, HBM-bound (gfx950).
//
// Restates flows/bijections/neural_splines.py:11-184 (searchsorted :11-13, unconstrained_RQS :16-71, RQS :74-184).
// That module is dead code in the reference (it calls jax.ops.index_update, removed from JAX) and has no caller and
// no fixture: PARITY UNPINNED -- checked against oracle/wf_oracle.c (wfo_rqs) and by self-consistency only.
//
// Data movement: each element owns rows uw[K], uh[K], ud[K-1 | K+1].  A workgroup stages a [256][K] tile through LDS
// with fully coalesced 16-byte loads (rows are contiguous in HBM), then every lane walks its own row in LDS
// (row stride K+1 dwords: conflict-free).  Only the two derivatives of the selected bin are read (softplus on 2
// values instead of K+1).  Algorithmic traffic: (2K + 2 + 3) * 4 bytes per element.
#include <hip/hip_runtime.h>

#include "wf_internal.h"
#include "wf_scalar_impl.h"   // scalar::Philox

namespace wf {

namespace {

constexpr int kRqsBlock = 256;
constexpr float kMinBinWidth = 1e-3f, kMinBinHeight = 1e-3f, kMinDerivative = 1e-3f;

// cooperative, coalesced copy of rows [e0, e0+n) x K floats into LDS with row stride K+1
__device__ __forceinline__ void stage_tile(const float* __restrict__ g, int64_t e0, int n, int K, float* lds) {
    const int total = n * K;
    const float* src = g + e0 * K;
    if ((((uintptr_t)src) & 15) == 0) {
        const int n4 = total >> 2;
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (int i = threadIdx.x; i < n4; i += kRqsBlock) {
            const float4 v = s4[i];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int f = 4 * i + q;
                lds[(f / K) * (K + 1) + (f % K)] = vv[q];
            }
        }
        for (int f = (n4 << 2) + threadIdx.x; f < total; f += kRqsBlock) lds[(f / K) * (K + 1) + (f % K)] = src[f];
    } else {
        for (int f = threadIdx.x; f < total; f += kRqsBlock) lds[(f / K) * (K + 1) + (f % K)] = src[f];
    }
}

// softmax-normalised bin sizes of one row (in place: row[i] <- min + (1 - min*K) * softmax_i)
__device__ __forceinline__ void normalise_row(float* row, int K, float min_size) {
    float mx = row[0];
    for (int i = 1; i < K; ++i) mx = fmaxf(mx, row[i]);
    float s = 0.0f;
    for (int i = 0; i < K; ++i) {
        const float e = expf(row[i] - mx);
        row[i] = e;
        s = s + e;
    }
    for (int i = 0; i < K; ++i) row[i] = min_size + (1 - min_size * K) * (row[i] / s);
}

// knot i of the cumulative array: lo + (hi - lo) * cumsum_i, end knots forced (neural_splines.py:100-119)
struct Bin {
    int idx;
    float knot, size;
};

// search: bin = sum(x >= knots) - 1 with eps on the last knot (neural_splines.py:11-13), clamped to [0, K-1]
__device__ __forceinline__ Bin search_bin(const float* row, int K, float lo, float hi, float x) {
    int count = 0;
    float cum = 0.0f, prev = lo, knot_b = lo, size_b = 0.0f;
    // knot 0
    if (x >= lo) ++count;
    for (int i = 0; i < K; ++i) {
        cum = cum + row[i];
        float knot = (hi - lo) * cum + lo;
        if (i == K - 1) knot = hi;
        const float cmp = (i == K - 1) ? knot + 1e-6f : knot;
        const bool ge = x >= cmp;
        // bin i spans [prev, knot]
        if (count == i + 1 && !ge) { knot_b = prev; size_b = knot - prev; }
        if (ge) ++count;
        prev = knot;
    }
    Bin b;
    b.idx = min(max(count - 1, 0), K - 1);
    if (count - 1 != b.idx || count == 0 || count == K + 1) {
        // clamped: recompute the selected bin's knots
        cum = 0.0f; prev = lo;
        for (int i = 0; i < K; ++i) {
            cum = cum + row[i];
            float knot = (hi - lo) * cum + lo;
            if (i == K - 1) knot = hi;
            if (i == b.idx) { knot_b = prev; size_b = knot - prev; }
            prev = knot;
        }
    }
    b.knot = knot_b;
    b.size = size_b;
    return b;
}

// the given bin of a cumulative array
__device__ __forceinline__ Bin pick_bin(const float* row, int K, float lo, float hi, int idx) {
    float cum = 0.0f, prev = lo;
    Bin b;
    b.idx = idx; b.knot = lo; b.size = 0.0f;
    for (int i = 0; i < K; ++i) {
        cum = cum + row[i];
        float knot = (hi - lo) * cum + lo;
        if (i == K - 1) knot = hi;
        if (i == idx) { b.knot = prev; b.size = knot - prev; }
        prev = knot;
    }
    return b;
}

__device__ __forceinline__ float softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(kRqsBlock) void k_rqs(const float* __restrict__ xg, const float* __restrict__ uw, const float* __restrict__ uh,
                                                   const float* __restrict__ ud, int64_t N, int K, int n_deriv, int inverse, float left,
                                                   float right, float bottom, float top, float* __restrict__ yg,
                                                   float* __restrict__ ldg, int32_t* __restrict__ bing) {
    extern __shared__ float tile[];  // [256][K+1]
    const bool unconstrained = n_deriv == K - 1;
    // boundary derivative constant of unconstrained_RQS: softplus(c) + min_d == 1 (neural_splines.py:36)
    const float edge = logf(expf(1 - kMinDerivative) - 1);
    for (int64_t e0 = (int64_t)blockIdx.x * kRqsBlock; e0 < N; e0 += (int64_t)gridDim.x * kRqsBlock) {
        const int n = (int)min((int64_t)kRqsBlock, N - e0);
        const int64_t e = e0 + threadIdx.x;
        const bool active = threadIdx.x < n;
        const float x = active ? xg[e] : 0.0f;
        const bool inside = unconstrained ? (x >= left && x <= right) : true;
        float* row = tile + threadIdx.x * (K + 1);
        // pass 1: the array that is searched (widths for the forward map, heights for the inverse)
        __syncthreads();
        stage_tile(inverse ? uh : uw, e0, n, K, tile);
        __syncthreads();
        Bin s{0, 0.0f, 1.0f};
        if (active) {
            normalise_row(row, K, inverse ? kMinBinHeight : kMinBinWidth);
            s = search_bin(row, K, inverse ? bottom : left, inverse ? top : right, x);
        }
        // pass 2: the other array, same bin
        __syncthreads();
        stage_tile(inverse ? uw : uh, e0, n, K, tile);
        __syncthreads();
        if (!active) continue;
        normalise_row(row, K, inverse ? kMinBinWidth : kMinBinHeight);
        const Bin o = pick_bin(row, K, inverse ? left : bottom, inverse ? right : top, s.idx);
        const float in_cw = inverse ? o.knot : s.knot, in_w = inverse ? o.size : s.size;
        const float in_ch = inverse ? s.knot : o.knot, in_h = inverse ? s.size : o.size;
        const int b = s.idx;
        float u0, u1;
        if (unconstrained) {
            u0 = b == 0 ? edge : ud[e * n_deriv + (b - 1)];
            u1 = b == K - 1 ? edge : ud[e * n_deriv + b];
        } else {
            u0 = ud[e * n_deriv + b];
            u1 = ud[e * n_deriv + b + 1];
        }
        const float d0 = kMinDerivative + softplus(u0), d1 = kMinDerivative + softplus(u1);
        const float delta = in_h / in_w;
        float y, ld;
        if (inverse) {
            const float a = (x - in_ch) * (d0 + d1 - 2 * delta) + in_h * (delta - d0);
            const float bq = in_h * d0 - (x - in_ch) * (d0 + d1 - 2 * delta);
            const float cq = -delta * (x - in_ch);
            const float disc = bq * bq - 4 * a * cq;
            const float root = (2 * cq) / (-bq - sqrtf(disc));
            y = root * in_w + in_cw;
            const float t1 = root * (1 - root);
            const float den = delta + ((d0 + d1 - 2 * delta) * t1);
            const float num = delta * delta * (d1 * root * root + 2 * delta * t1 + d0 * (1 - root) * (1 - root));
            ld = -(logf(num) - 2 * logf(den));
        } else {
            const float th = (x - in_cw) / in_w;
            const float t1 = th * (1 - th);
            const float num = in_h * (delta * th * th + d0 * t1);
            const float den = delta + ((d0 + d1 - 2 * delta) * t1);
            y = in_ch + num / den;
            const float dnum = delta * delta * (d1 * th * th + 2 * delta * t1 + d0 * (1 - th) * (1 - th));
            ld = logf(dnum) - 2 * logf(den);
        }
        if (!inside) { y = x; ld = 0.0f; }   // identity tails (neural_splines.py:26-50)
        yg[e] = y;
        ldg[e] = ld;
        if (bing) bing[e] = inside ? b : -1;
    }
}


// ---- register path: K in {4, 8, 16, 32}; every lane loads its own rows with 16-byte loads (a row is 16..128
// contiguous bytes, so the wave's loads cover whole cache lines between them) and keeps them in registers.
template <int K>
__device__ __forceinline__ void load_row(const float* __restrict__ g, int64_t e, float (&r)[K]) {
    const float4* p = reinterpret_cast<const float4*>(g + e * K);
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        const float4 v = p[q];
        r[4 * q] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}

// r[i] <- bin size i = min + (1 - min*K) * softmax_i(r)
template <int K>
__device__ __forceinline__ void normalise_reg(float (&r)[K], float min_size) {
    float mx = r[0];
#pragma unroll
    for (int i = 1; i < K; ++i) mx = fmaxf(mx, r[i]);
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        r[i] = __builtin_amdgcn_exp2f((r[i] - mx) * 1.4426950408889634f);
        s = s + r[i];
    }
    const float c = (1 - min_size * K) / s;
#pragma unroll
    for (int i = 0; i < K; ++i) r[i] = __builtin_fmaf(c, r[i], min_size);
}

template <int K>
__global__ __launch_bounds__(kRqsBlock) void k_rqs_reg(const float* __restrict__ xg, const float* __restrict__ uw, const float* __restrict__ uh,
                                                       const float* __restrict__ ud, int64_t N, int n_deriv, int inverse, float left,
                                                       float right, float bottom, float top, float* __restrict__ yg,
                                                       float* __restrict__ ldg, int32_t* __restrict__ bing) {
    const bool unconstrained = n_deriv == K - 1;
    const float edge = logf(expf(1 - kMinDerivative) - 1);
    for (int64_t e = (int64_t)blockIdx.x * kRqsBlock + threadIdx.x; e < N; e += (int64_t)gridDim.x * kRqsBlock) {
        const float x = xg[e];
        float sr[K], orow[K];   // searched array (widths fwd / heights inv) and the other one
        load_row<K>(inverse ? uh : uw, e, sr);
        load_row<K>(inverse ? uw : uh, e, orow);
        const bool inside = unconstrained ? (x >= left && x <= right) : true;
        const float slo = inverse ? bottom : left, shi = inverse ? top : right;
        const float olo = inverse ? left : bottom, ohi = inverse ? right : top;
        normalise_reg<K>(sr, inverse ? kMinBinHeight : kMinBinWidth);
        normalise_reg<K>(orow, inverse ? kMinBinWidth : kMinBinHeight);
        // knots of both arrays in one sweep; bin = sum(x >= knot) - 1 with eps on the last knot, clamped
        int count = x >= slo ? 1 : 0;
        float cs = 0.0f, co = 0.0f, sprev = slo, oprev = olo;
        float s_knot = slo, s_size = sr[0], o_knot = olo, o_size = orow[0];
        bool first = true;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            cs = cs + sr[i];
            co = co + orow[i];
            float sk = (shi - slo) * cs + slo, ok = (ohi - olo) * co + olo;
            if (i == K - 1) { sk = shi; ok = ohi; }
            const bool ge = x >= ((i == K - 1) ? sk + 1e-6f : sk);
            // bin i is selected if x >= knot_i and not x >= knot_{i+1}; clamping: i == 0 also takes x < knot_0,
            // i == K-1 also takes x >= last knot
            const bool sel = (count == i + 1 && !ge) || (i == 0 && count == 0) || (i == K - 1 && ge && count == K);
            if (sel || (i == 0 && first)) {
                if (sel) { s_knot = sprev; s_size = sk - sprev; o_knot = oprev; o_size = ok - oprev; }
            }
            first = false;
            if (ge) ++count;
            sprev = sk;
            oprev = ok;
        }
        const int b = min(max(count - 1, 0), K - 1);
        const float in_cw = inverse ? o_knot : s_knot, in_w = inverse ? o_size : s_size;
        const float in_ch = inverse ? s_knot : o_knot, in_h = inverse ? s_size : o_size;
        float u0, u1;
        if (unconstrained) {
            u0 = b == 0 ? edge : ud[e * n_deriv + (b - 1)];
            u1 = b == K - 1 ? edge : ud[e * n_deriv + b];
        } else {
            u0 = ud[e * n_deriv + b];
            u1 = ud[e * n_deriv + b + 1];
        }
        const float d0 = kMinDerivative + softplus(u0), d1 = kMinDerivative + softplus(u1);
        const float delta = in_h / in_w;
        float y, ld;
        if (inverse) {
            const float a = (x - in_ch) * (d0 + d1 - 2 * delta) + in_h * (delta - d0);
            const float bq = in_h * d0 - (x - in_ch) * (d0 + d1 - 2 * delta);
            const float cq = -delta * (x - in_ch);
            const float disc = bq * bq - 4 * a * cq;
            const float root = (2 * cq) / (-bq - sqrtf(disc));
            y = root * in_w + in_cw;
            const float t1 = root * (1 - root);
            const float den = delta + ((d0 + d1 - 2 * delta) * t1);
            const float num = delta * delta * (d1 * root * root + 2 * delta * t1 + d0 * (1 - root) * (1 - root));
            ld = -(logf(num) - 2 * logf(den));
        } else {
            const float th = (x - in_cw) / in_w;
            const float t1 = th * (1 - th);
            const float num = in_h * (delta * th * th + d0 * t1);
            const float den = delta + ((d0 + d1 - 2 * delta) * t1);
            y = in_ch + num / den;
            const float dnum = delta * delta * (d1 * th * th + 2 * delta * t1 + d0 * (1 - th) * (1 - th));
            ld = logf(dnum) - 2 * logf(den);
        }
        if (!inside) { y = x; ld = 0.0f; }
        yg[e] = y;
        ldg[e] = ld;
        if (bing) bing[e] = inside ? b : -1;
    }
}

template <int K>
int launch_reg(const float* x, const float* uw, const float* uh, const float* ud, int64_t N, int n_deriv, int inverse, float left, float right,
               float bottom, float top, float* y, float* ld, int32_t* bin, hipStream_t s) {
    int64_t blocks = (N + kRqsBlock - 1) / kRqsBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_rqs_reg<K>, dim3((unsigned)blocks), dim3(kRqsBlock), 0, s, x, uw, uh, ud, N, n_deriv, inverse, left, right, bottom,
                       top, y, ld, bin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_hip_error((int)e); return WF_ERR_HIP; }
    return WF_OK;
}

}  // namespace

// ---- NeuralSplineCoupling (neural_splines.py:244-300): conditioner of one half-step.  FCNN (:187-188) = Dense(hidden), Tanh,
// Dense(hidden), Tanh, Dense((3K-1) * dh) on the conditioning half; its output, per transformed coordinate, is split into
// K widths, K heights, K-1 derivatives, the first two soft-maxed and scaled by 2B, the last soft-plussed (:255-259) -- and then
// handed to unconstrained_RQS as *unnormalised* parameters, which soft-maxes / soft-plusses them again (the reference's
// double application is kept).  One lane per walker; the network is tiny (hidden 8), its weights are scalar loads.
// params: W1 [din][h], b1 [h], W2 [h][h], b2 [h], W3 [h][(3K-1)*dh], b3 [(3K-1)*dh]  (stax.Dense leaf order)
constexpr int kNscMaxHidden = 64;
__global__ __launch_bounds__(256) void k_nsc_cond(const float* __restrict__ xg, int64_t B, int dim, int cond_off, int trans_off, int dh, int K,
                                                  float tail, int hidden, const float* __restrict__ params, const float* __restrict__ cond_src,
                                                  float* __restrict__ uw, float* __restrict__ uh, float* __restrict__ ud, float* __restrict__ xt) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int per = 3 * K - 1, dout = per * dh;
    const float* W1 = params;
    const float* b1 = W1 + dh * hidden;
    const float* W2 = b1 + hidden;
    const float* b2 = W2 + hidden * hidden;
    const float* W3 = b2 + hidden;
    const float* b3 = W3 + hidden * dout;
    float h1[kNscMaxHidden], h2[kNscMaxHidden];
    // the conditioning half: columns of x, or (second half-step) the half just transformed
    for (int j = 0; j < hidden; ++j) {
        float z = b1[j];
        for (int a = 0; a < dh; ++a) {
            const float v = cond_src ? cond_src[b * dh + a] : xg[b * dim + cond_off + a];
            z = __builtin_fmaf(v, W1[a * hidden + j], z);
        }
        h1[j] = tanhf(z);
    }
    for (int j = 0; j < hidden; ++j) {
        float z = b2[j];
        for (int a = 0; a < hidden; ++a) z = __builtin_fmaf(h1[a], W2[a * hidden + j], z);
        h2[j] = tanhf(z);
    }
    for (int d = 0; d < dh; ++d) {
        const int64_t e = b * dh + d;
        float o[96];   // per <= 3 * 32 - 1
        for (int c = 0; c < per; ++c) {
            float z = b3[d * per + c];
            for (int a = 0; a < hidden; ++a) z = __builtin_fmaf(h2[a], W3[a * dout + d * per + c], z);
            o[c] = z;
        }
        // onp.array_split(out, 3, axis=2): sizes K, K, K-1
        for (int part = 0; part < 2; ++part) {
            float mx = o[part * K];
            for (int c = 1; c < K; ++c) mx = fmaxf(mx, o[part * K + c]);
            float sum = 0.0f;
            for (int c = 0; c < K; ++c) sum += expf(o[part * K + c] - mx);
            float* dst = (part == 0 ? uw : uh) + e * K;
            for (int c = 0; c < K; ++c) dst[c] = 2.0f * tail * (expf(o[part * K + c] - mx) / sum);
        }
        for (int c = 0; c < K - 1; ++c) ud[e * (K - 1) + c] = softplus(o[2 * K + c]);
        xt[e] = xg[b * dim + trans_off + d];
    }
}

// y = [lower', upper'], logdet = sum of the per-coordinate log-dets of both half-steps
__global__ void k_nsc_finish(int64_t B, int dim, int dh, const float* __restrict__ lower, const float* __restrict__ upper,
                             const float* __restrict__ ld_a, const float* __restrict__ ld_b, float* __restrict__ y, float* __restrict__ logdet) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.0f;
    for (int d = 0; d < dh; ++d) {
        y[b * dim + d] = lower[b * dh + d];
        y[b * dim + dh + d] = upper[b * dh + d];
        s += ld_a[b * dh + d];
    }
    for (int d = 0; d < dh; ++d) s += ld_b[b * dh + d];
    logdet[b] = s;
}

// ---- the coupling stack in one kernel.  Flow(Serial(NeuralSplineCoupling [, Reverse]) x L, Normal | Uniform) -- or one bare layer --
// with one lane per walker and everything in registers: the walker's D coordinates, the conditioner's hidden units (HID <= 32), the
// 3K - 1 spline parameters of the coordinate being transformed (K <= KM).  The weights are the same for every lane: the compiler
// reads them with scalar loads (constant cache), so the kernel's HBM traffic is x in, log_pdf (and u) out: (D + 1) * 4 bytes per
// walker instead of the (6K + 8) * D * 2 bytes the launch-per-half-step path (launch_nsc) moves through its workspace.
// unconstrained_RQS on register rows (same arithmetic as k_rqs_reg: widths / heights min + (1 - min K) softmax, knots by running
// sum with the end knot forced, bin = sum(x >= knot) - 1 clamped, derivative edge constant, identity outside the tails).
// fast transcendentals of the one-kernel stack: hardware exp2 / log2 / rcp (1 ulp), tanh = 1 - 2 / (e^{2x} + 1) (2^-23 absolute)
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f); }

template <int KM>
__device__ __forceinline__ void rqs_rows(float x, int K, float (&sr)[KM], float (&orow)[KM], const float (&ud)[KM], float tail, bool inverse,
                                         float& y, float& ld) {
    const float edge = logf(expf(1 - kMinDerivative) - 1);
    const float lo = -tail, hi = tail;
    // (forward: sr = widths, orow = heights; inverse: sr = heights, orow = widths -- min sizes are equal)
    float mxs = sr[0], mxo = orow[0];
#pragma unroll
    for (int i = 1; i < KM; ++i)
        if (i < K) { mxs = fmaxf(mxs, sr[i]); mxo = fmaxf(mxo, orow[i]); }
    float ss = 0.0f, so = 0.0f;
#pragma unroll
    for (int i = 0; i < KM; ++i)
        if (i < K) {
            sr[i] = __builtin_amdgcn_exp2f((sr[i] - mxs) * 1.4426950408889634f);
            orow[i] = __builtin_amdgcn_exp2f((orow[i] - mxo) * 1.4426950408889634f);
            ss = ss + sr[i];
            so = so + orow[i];
        }
    const float cs_ = (1 - kMinBinWidth * K) * __builtin_amdgcn_rcpf(ss), co_ = (1 - kMinBinHeight * K) * __builtin_amdgcn_rcpf(so);
    int count = x >= lo ? 1 : 0;
    float cs = 0.0f, co = 0.0f, sprev = lo, oprev = lo;
    float s_knot = lo, s_size = 1.0f, o_knot = lo, o_size = 1.0f, u0 = edge, u1 = edge;
#pragma unroll
    for (int i = 0; i < KM; ++i)
        if (i < K) {
            const float wi = __builtin_fmaf(cs_, sr[i], kMinBinWidth), oi = __builtin_fmaf(co_, orow[i], kMinBinHeight);
            cs = cs + wi;
            co = co + oi;
            const bool last = i == K - 1;
            float sk = (hi - lo) * cs + lo, ok = (hi - lo) * co + lo;
            if (last) { sk = hi; ok = hi; }
            const bool ge = x >= (last ? sk + 1e-6f : sk);
            const bool sel = (count == i + 1 && !ge) || (i == 0 && count == 0) || (last && ge && count == K);
            if (sel) {
                s_knot = sprev; s_size = sk - sprev; o_knot = oprev; o_size = ok - oprev;
                u0 = i == 0 ? edge : ud[i > 0 ? i - 1 : 0];
                u1 = last ? edge : ud[i];
            }
            if (ge) ++count;
            sprev = sk;
            oprev = ok;
        }
    const float in_cw = inverse ? o_knot : s_knot, in_w = inverse ? o_size : s_size;
    const float in_ch = inverse ? s_knot : o_knot, in_h = inverse ? s_size : o_size;
    const float d0 = kMinDerivative + softplus(u0), d1 = kMinDerivative + softplus(u1);
    const float rw = __builtin_amdgcn_rcpf(in_w);
    const float delta = inverse ? in_h / in_w : in_h * rw;   // (the inverse keeps IEEE division / sqrt: its error is amplified by 1 / slope)
    if (inverse) {
        const float a = (x - in_ch) * (d0 + d1 - 2 * delta) + in_h * (delta - d0);
        const float bq = in_h * d0 - (x - in_ch) * (d0 + d1 - 2 * delta);
        const float cq = -delta * (x - in_ch);
        const float disc = bq * bq - 4 * a * cq;
        const float root = (2 * cq) / (-bq - sqrtf(disc));
        y = root * in_w + in_cw;
        const float t1 = root * (1 - root);
        const float den = delta + ((d0 + d1 - 2 * delta) * t1);
        const float num = delta * delta * (d1 * root * root + 2 * delta * t1 + d0 * (1 - root) * (1 - root));
        ld = -(logf(num) - 2 * logf(den));
    } else {
        const float th = (x - in_cw) * rw;
        const float t1 = th * (1 - th);
        const float num = in_h * (delta * th * th + d0 * t1);
        const float den = delta + ((d0 + d1 - 2 * delta) * t1);
        y = in_ch + num * __builtin_amdgcn_rcpf(den);
        const float dnum = delta * delta * (d1 * th * th + 2 * delta * t1 + d0 * (1 - th) * (1 - th));
        ld = flog(dnum) - 2 * flog(den);
    }
    if (!(x >= lo && x <= hi)) { y = x; ld = 0.0f; }
}

// one half-step (neural_splines.py:254-262): the dh coordinates t[] transformed given the dh coordinates c[]
template <int HID, int KM>
__device__ __forceinline__ float nsc_half(const float* __restrict__ net, const int dh, const int K, float tail, bool inverse, const float (&c)[4],
                                          float (&t)[4]) {
    const int per = 3 * K - 1, dout = per * dh;
    const float* __restrict__ W1 = net;
    const float* __restrict__ b1 = W1 + dh * HID;
    const float* __restrict__ W2 = b1 + HID;
    const float* __restrict__ b2 = W2 + HID * HID;
    const float* __restrict__ W3 = b2 + HID;
    const float* __restrict__ b3 = W3 + HID * dout;
    // Input unit outermost everywhere: the weights a unit feeds forward are contiguous in the stax.Dense layout ([in][out]), i.e. one wide
    // scalar load per unit instead of one dword per (unit, output), and few of them are live at a time (SGPR budget ~100).
    float h1[HID], h2[HID];
#pragma unroll
    for (int j = 0; j < HID; ++j) h1[j] = b1[j];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        if (a < dh) {
#pragma unroll
            for (int j = 0; j < HID; ++j) h1[j] = __builtin_fmaf(c[a], W1[a * HID + j], h1[j]);
        }
#pragma unroll
    for (int j = 0; j < HID; ++j) { h1[j] = ftanh(h1[j]); h2[j] = b2[j]; }
#pragma unroll
    for (int a = 0; a < HID; ++a) {
#pragma unroll
        for (int j = 0; j < HID; ++j) h2[j] = __builtin_fmaf(h1[a], W2[a * HID + j], h2[j]);
    }
#pragma unroll
    for (int j = 0; j < HID; ++j) h2[j] = ftanh(h2[j]);
    float logdet = 0.0f;
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d < dh) {
            // onp.array_split(out, 3, axis=2): K widths, K heights, K - 1 derivatives; the first two soft-maxed and scaled by 2B, the last
            // soft-plussed (:255-259) -- and unconstrained_RQS normalises them again (the reference's double application is kept)
            float o[3 * KM];   // the coordinate's 3K - 1 outputs, contiguous columns d * per ..
#pragma unroll
            for (int q = 0; q < 3 * KM; ++q) o[q] = q < per ? b3[d * per + q] : 0.0f;
#pragma unroll
            for (int a = 0; a < HID; ++a) {
#pragma unroll
                for (int q = 0; q < 3 * KM; ++q)
                    if (q < per) o[q] = __builtin_fmaf(h2[a], W3[a * dout + d * per + q], o[q]);
            }
            float uw[KM], uh[KM], ud[KM];
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                float e[KM];
                float mx = o[part * K];
#pragma unroll
                for (int q = 1; q < KM; ++q)
                    if (q < K) mx = fmaxf(mx, o[part * K + q]);
                float sum = 0.0f;
#pragma unroll
                for (int q = 0; q < KM; ++q) {
                    e[q] = 0.0f;
                    if (q < K) { e[q] = fexp(o[part * K + q] - mx); sum += e[q]; }
                }
                const float sc = 2.0f * tail * __builtin_amdgcn_rcpf(sum);
#pragma unroll
                for (int q = 0; q < KM; ++q) {
                    const float v = q < K ? sc * e[q] : 0.0f;
                    if (part == 0) uw[q] = v; else uh[q] = v;
                }
            }
#pragma unroll
            for (int q = 0; q < KM; ++q) ud[q] = q < K - 1 ? softplus(o[2 * K + q]) : 0.0f;
            float y, ld;
            if (inverse) rqs_rows<KM>(t[d], K, uh, uw, ud, tail, true, y, ld);
            else rqs_rows<KM>(t[d], K, uw, uh, ud, tail, false, y, ld);
            t[d] = y;
            logdet += ld;
        }
    return logdet;
}

// mode 0: log_pdf = prior(z) + logdet; 2: z and logdet (flow only); 3: inverse (x <- z, out = logdet of the inverse)
// DHT / KT != 0: the half width and the bin count as compile-time constants (the reference's shapes), every guard folds away
template <int HID, int KM, int DHT = 0, int KT = 0>
__global__ __launch_bounds__(256) void k_nsc_model(NscModelDev md, int mode, const float* __restrict__ xg, int64_t B, float* __restrict__ out,
                                                   float* __restrict__ ug) {
    const int dh = DHT ? DHT : md.D / 2, D = 2 * dh, nK = KT ? KT : md.K;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        float lo[4] = {0, 0, 0, 0}, up[4] = {0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < dh) { lo[a] = xg[b * D + a]; up[a] = xg[b * D + dh + a]; }
        float logdet = 0.0f;
        for (int li = 0; li < md.L; ++li) {
            const int l = mode == 3 ? md.L - 1 - li : li;
            const float* __restrict__ f1 = md.params + (int64_t)l * 2 * md.net_floats;
            const float* __restrict__ f2 = f1 + md.net_floats;
            if (mode != 3) {
                logdet += nsc_half<HID, KM>(f1, dh, nK, md.tail, false, lo, up);   // upper' = RQS(upper | f1(lower))
                logdet += nsc_half<HID, KM>(f2, dh, nK, md.tail, false, up, lo);   // lower' = RQS(lower | f2(upper'))
            }
            if (md.reverse) {   // flows.Reverse (bijections.py): x[:, ::-1] -- after the layer going forward, before it going back
                float nl[4], nu[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    nl[a] = a < dh ? up[a < dh ? dh - 1 - a : 0] : 0.0f;
                    nu[a] = a < dh ? lo[a < dh ? dh - 1 - a : 0] : 0.0f;
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) { lo[a] = nl[a]; up[a] = nu[a]; }
            }
            if (mode == 3) {
                logdet += nsc_half<HID, KM>(f2, dh, nK, md.tail, true, up, lo);    // lower' = RQS^-1(lower | f2(upper))
                logdet += nsc_half<HID, KM>(f1, dh, nK, md.tail, true, lo, up);    // upper' = RQS^-1(upper | f1(lower'))
            }
        }
        float res = logdet;
        if (mode == 0) {
            float lp = 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < dh) {
                    if (md.prior_kind == WF_PRIOR_NORMAL) {
                        const float z0 = lo[a] + md.normal_offset, z1 = up[a] + md.normal_offset;
                        lp = lp + (1.8378770664093453f + z0 * z0) * -0.5f + (1.8378770664093453f + z1 * z1) * -0.5f;
                    } else {   // Uniform with prior_support (0, 1): the sample is clipped, the density is 1
                        lo[a] = fminf(fmaxf(lo[a], 0.0f), 1.0f);
                        up[a] = fminf(fmaxf(up[a], 0.0f), 1.0f);
                    }
                }
            res = lp + logdet;
        }
        out[b] = res;
        if (ug) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < dh) { ug[b * D + a] = lo[a]; ug[b * D + dh + a] = up[a]; }
        }
    }
}

template <int HID, int KM, int DHT = 0, int KT = 0>
int launch_nsc_model_t(const NscModelDev& md, int mode, const float* x, int64_t B, float* out, float* u, hipStream_t s) {
    int64_t blocks = (B + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((k_nsc_model<HID, KM, DHT, KT>), dim3((unsigned)blocks), dim3(256), 0, s, md, mode, x, B, out, u);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_hip_error((int)e); return WF_ERR_HIP; }
    return WF_OK;
}

// z ~ Normal(0, 1)^D (distributions.py:18-19: the offset only enters log_pdf) or Uniform(0, 1)^D; stream (seed, walker) like wf_sample's
__global__ void k_nsc_latent(int prior_kind, int D, unsigned long long seed, int64_t B, float* __restrict__ z) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    scalar::Philox rng(seed, (unsigned long long)b);
    for (int d = 0; d < D; ++d) {
        if (prior_kind == WF_PRIOR_UNIFORM) {
            z[b * D + d] = rng.uniform();
        } else {
            const float u1 = fmaxf(rng.uniform(), 5.9604645e-8f), u2 = rng.uniform();
            z[b * D + d] = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
        }
    }
}
int launch_nsc_latent(int prior_kind, int D, unsigned long long seed, int64_t B, float* z, void* stream) {
    if (B == 0) return WF_OK;
    hipLaunchKernelGGL(k_nsc_latent, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, prior_kind, D, seed, B, z);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_hip_error((int)e); return WF_ERR_HIP; }
    return WF_OK;
}

bool nsc_model_built(int D, int K, int hidden) { return D >= 2 && D <= 8 && D % 2 == 0 && K >= 2 && K <= 16 && (hidden == 8 || hidden == 32); }

int launch_nsc_model(const NscModelDev& md, int mode, const float* x, int64_t B, float* out, float* u, void* stream) {
    if (B == 0) return WF_OK;
    if (!nsc_model_built(md.D, md.K, md.hidden)) return WF_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (md.hidden == 8 && md.K == 5 && md.D == 2) return launch_nsc_model_t<8, 8, 1, 5>(md, mode, x, B, out, u, s);   // NeuralSplineCoupling()'s defaults
    if (md.hidden == 8 && md.K == 5 && md.D == 4) return launch_nsc_model_t<8, 8, 2, 5>(md, mode, x, B, out, u, s);
    if (md.hidden == 8) return md.K <= 8 ? launch_nsc_model_t<8, 8>(md, mode, x, B, out, u, s) : launch_nsc_model_t<8, 16>(md, mode, x, B, out, u, s);
    return md.K <= 8 ? launch_nsc_model_t<32, 8>(md, mode, x, B, out, u, s) : launch_nsc_model_t<32, 16>(md, mode, x, B, out, u, s);
}

int launch_rqs(const float* x, const float* uw, const float* uh, const float* ud, int64_t N, int K, int n_deriv, int inverse, float left,
               float right, float bottom, float top, float* y, float* ld, int32_t* bin, void* stream) {
    if (N == 0) return WF_OK;
    const bool aligned = ((((uintptr_t)uw) | ((uintptr_t)uh)) & 15) == 0;
    if (aligned) {
        hipStream_t st = (hipStream_t)stream;
        switch (K) {
            case 4: return launch_reg<4>(x, uw, uh, ud, N, n_deriv, inverse, left, right, bottom, top, y, ld, bin, st);
            case 8: return launch_reg<8>(x, uw, uh, ud, N, n_deriv, inverse, left, right, bottom, top, y, ld, bin, st);
            case 16: return launch_reg<16>(x, uw, uh, ud, N, n_deriv, inverse, left, right, bottom, top, y, ld, bin, st);
            case 32: return launch_reg<32>(x, uw, uh, ud, N, n_deriv, inverse, left, right, bottom, top, y, ld, bin, st);
            default: break;
        }
    }
    const int lds_bytes = kRqsBlock * (K + 1) * (int)sizeof(float);
    int64_t blocks = (N + kRqsBlock - 1) / kRqsBlock;
    if (blocks > 256 * 8) blocks = 256 * 8;
    static DynLdsSlots cfg{};
    if (lds_bytes > 64 * 1024)
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_rqs), lds_bytes, &cfg)) return rc;
    hipLaunchKernelGGL(k_rqs, dim3((unsigned)blocks), dim3(kRqsBlock), lds_bytes, (hipStream_t)stream, x, uw, uh, ud, N, K, n_deriv, inverse,
                       left, right, bottom, top, y, ld, bin);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_hip_error((int)e); return WF_ERR_HIP; }
    return WF_OK;
}

int64_t nsc_workspace_floats(int64_t B, int dim, int K) {
    const int64_t E = B * (dim / 2);
    return E * (3 * K - 1) + 5 * E + 64;   // uw, uh, ud rows; the half being transformed; both results; two log-det vectors
}

// direct (inverse == 0): upper' = RQS(upper | f1(lower)); lower' = RQS(lower | f2(upper'))   (neural_splines.py:254-271)
// inverse:               lower' = RQS^-1(lower | f2(upper)); upper' = RQS^-1(upper | f1(lower'))   (:273-292)
int launch_nsc(const float* x, int64_t B, int dim, int K, float tail, int hidden, const float* params, int inverse, float* y, float* logdet,
               float* ws, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return WF_OK;
    const int dh = dim / 2;
    const int64_t E = B * dh;
    const int64_t net_floats = (int64_t)dh * hidden + hidden + (int64_t)hidden * hidden + hidden + (int64_t)hidden * (3 * K - 1) * dh + (3 * K - 1) * dh;
    const float *f1 = params, *f2 = params + net_floats;
    float* uw = ws;
    float* uh = uw + E * K;
    float* ud = uh + E * K;
    float* xt = ud + E * (K - 1);      // the half being transformed, contiguous
    float* ha = xt + E;                // result of the first half-step
    float* hb = ha + E;                // result of the second half-step
    float* lda = hb + E;
    float* ldb = lda + E;
    // (uw / uh must be 16-byte aligned for the register kernel: E * K floats apart -- launch_rqs falls back otherwise)
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    const int first_cond = inverse ? dh : 0, first_trans = inverse ? 0 : dh;
    hipLaunchKernelGGL(k_nsc_cond, grid, block, 0, s, x, B, dim, first_cond, first_trans, dh, K, tail, hidden, inverse ? f2 : f1,
                       (const float*)nullptr, uw, uh, ud, xt);
    int rc = launch_rqs(xt, uw, uh, ud, E, K, K - 1, inverse, -tail, tail, -tail, tail, ha, lda, nullptr, stream);
    if (rc) return rc;
    // second half-step: conditioned on the half just transformed, transforms the other original half
    hipLaunchKernelGGL(k_nsc_cond, grid, block, 0, s, x, B, dim, 0, inverse ? dh : 0, dh, K, tail, hidden, inverse ? f1 : f2, (const float*)ha, uw,
                       uh, ud, xt);
    rc = launch_rqs(xt, uw, uh, ud, E, K, K - 1, inverse, -tail, tail, -tail, tail, hb, ldb, nullptr, stream);
    if (rc) return rc;
    // direct: ha = upper', hb = lower';  inverse: ha = lower', hb = upper'
    hipLaunchKernelGGL(k_nsc_finish, grid, block, 0, s, B, dim, dh, inverse ? (const float*)ha : (const float*)hb,
                       inverse ? (const float*)hb : (const float*)ha, (const float*)lda, (const float*)ldb, y, logdet);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

}  // namespace wf
