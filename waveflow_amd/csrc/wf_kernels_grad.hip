// wf_kernels_grad.hip -- parameter gradients, part 2: weight-gradient contraction, scatter, loss seeds.
//
//   grad[p] = sum_b ( w_psi[b] * d psi_b / d theta_p  +  w_lap[b] * d laplacian(psi)_b / d theta_p )
//
// is the vector-Jacobian product behind vqmc.train_step_efficient (vqmc.py:193-221): value_and_grad of the mean local
// energy with the custom tangent rule 2 t_psi (E_L - avg)/psi + (t_Hpsi psi - Hpsi t_psi)/psi^2, where
// Hpsi = -1/2 laplacian + V psi (utils/physics.py:79-93).  k_vqmc_seeds turns (Hpsi, psi, running average) into the two
// weight vectors; the Adam update itself is host code (waveflow_amd/vqmc.py).
//
// Method.  Along one coordinate direction x + t e_i every intermediate quantity is a truncated Taylor polynomial
// a0 + a1 t + a2 t^2 (an element of the ring R3 = IR[t]/t^3); psi'' along e_i is 2 * psi_2.  The evaluation of psi is a
// composition of ring operations, and the adjoint of a ring product y = a * b with respect to a, written with the
// adjoint coefficients in REVERSED order (abar~ = (abar_2, abar_1, abar_0)), is again a ring product: abar~ = ybar~ * b.
// Hence the reverse sweep over the ring-valued evaluation is ordinary back-propagation with every scalar replaced by a
// ring element, f'(a) replaced by the ring lift of f', and the gradient of a real parameter theta in y = theta * a is the
// top coefficient (ybar~ * a)_2.  The table lerp keeps the reference's derivative rule (the derivative of the order-nd
// lerp is the order-(nd+1) lerp, isplines_jax.py:60-66, bsplines_jax.py:32-38); the reference reaches order 4 in this
// sweep and JAX clamps that traced index to the last cached table (order 3) -- so does ring::lift().
// The sweeps run by default over RF<D> (wf_ring.h): value, gradient and Laplacian / 2 of every intermediate with respect to the D
// coordinates -- the D directional jets with their common parts shared.  It is a Frobenius algebra like R3 (pairing = top coefficient
// of the product), so everything above holds with "reversed order" read as "value and top slot exchanged, gradient slots in place".
// The same sweep over IR itself (ring R1) gives first-order objectives: sum_b w[b] d log_pdf_b / d theta for every model
// the library evaluates (IMADE / MADE layers; Waveflow, M-spline, Normal, Uniform priors) -- the maximum-likelihood
// gradient of benchmark_tests.train_model (benchmark_tests.py:84-101).
//
// The forward and reverse sweeps are the wave-cooperative kernels of wf_kernels_wave.hip; they leave, per sample and
// net, the layer input U, the hidden activations H1, H2 and the pre-activation adjoints A1, A2, O in a tape in HBM
// ([sample][net][coefficient][row], row-contiguous).  Here: k_wgrad contracts activations with adjoints over all samples
// (LDS-tiled 64x64x32, split over the sample axis into per-split partial images that are summed in split order: no
// atomics, bitwise reproducible) into a gradient image in the forward weight-image
// layout; k_grad_scatter moves that image to the reference's flat leaf order.  Checker: oracle/energy_torch.py (torch
// reverse mode through the Hessian trace) and central differences of the fp64 C oracle.
#include <hip/hip_runtime.h>

#include "wf_internal.h"
#include "wf_ring.h"

namespace wf {

namespace {

using namespace ring;
constexpr int H = kHidden;
constexpr int kGBlock = 64;
constexpr int kRows = 64;


// ---- kernel 2: weight gradients.  C[m][n] = sum_s sum_k X[m][k][s] * Y[n][partner(k)][s];  bias[n] = sum_s Y[n][NC-1][s]
// partner(k): the slot the pairing of the ring couples with slot k -- value and top slot exchanged, the slots between paired with
// themselves (R1: 0; R3: 2, 1, 0; RF: D+1, 1..D, 0 -- wf_ring.h)
struct WJob {
    int xrow, M, yrow, N;      // workspace rows (within a net) of the activations (X) and of the adjoints (Y)
    int out, sm, sn, bias;     // gradient-image offsets (within a net): C[m][n] -> out + m*sm + n*sn; bias[n] -> bias + n
};
struct WJobs {
    WJob j[3];
};
constexpr int kWT = 64;   // output tile
constexpr int wgrad_slab(int NC) { return NC <= 3 ? 32 : (NC <= 6 ? 16 : 8); }   // samples per staged slab: 2 * NC * slab * 256 B of LDS
__host__ __device__ constexpr int ring_partner(int NC, int k) { return k == 0 ? NC - 1 : (k == NC - 1 ? 0 : k); }
constexpr int kWgradSplit = 64;   // workgroups along the sample axis per (net, matrix, tile): 64 x 3..6 x n_nets fills the chip
template <int NC>
__global__ __launch_bounds__(256) void k_wgrad(const float* __restrict__ ws, int n_nets, int64_t n_samples, int rows, const WJobs jobs,
                                               int n_ntiles_max, float* __restrict__ gimg, int64_t net_img_floats) {
    constexpr int kWK = wgrad_slab(NC);
    __shared__ float4_t Xs[NC][kWK][kWT / 4];   // [coefficient][sample of the slab][row], rows contiguous
    __shared__ float4_t Ys[NC][kWK][kWT / 4];
    const int net = blockIdx.z;
    const int job_i = blockIdx.y / n_ntiles_max, ntile = blockIdx.y % n_ntiles_max;
    const WJob jb = jobs.j[job_i];
    const int n0 = ntile * kWT;
    if (n0 >= jb.N) return;
    const int tid = threadIdx.x, tm = tid & 15, tn = tid >> 4;
    float acc[4][4] = {};
    float bacc[4] = {};
    const int64_t sample_stride = (int64_t)n_nets * NC * rows;
    const float* __restrict__ base = ws + (int64_t)net * NC * rows;
    const int64_t n_slabs = (n_samples + kWK - 1) / kWK;
    const float4_t zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
        const int64_t s0 = slab * kWK;
        __syncthreads();
        for (int e = tid; e < NC * kWK * (kWT / 4); e += 256) {
            const int r4 = e % (kWT / 4), kk = (e / (kWT / 4)) % kWK, k = e / (kWK * (kWT / 4));
            const int64_t s = s0 + kk;
            float4_t xv = zero4, yv = zero4;
            if (s < n_samples) {
                const float* __restrict__ q = base + s * sample_stride + (int64_t)k * rows;   // every group of rows is 16-byte aligned
                if (4 * r4 < jb.M) xv = *reinterpret_cast<const float4_t*>(q + jb.xrow + 4 * r4);
                if (n0 + 4 * r4 < jb.N) yv = *reinterpret_cast<const float4_t*>(q + jb.yrow + n0 + 4 * r4);
                // rows beyond M (the padding of the U slot) hold no data
                if (4 * r4 + 1 >= jb.M) xv.y = 0.0f;
                if (4 * r4 + 2 >= jb.M) xv.z = 0.0f;
                if (4 * r4 + 3 >= jb.M) xv.w = 0.0f;
            }
            Xs[k][kk][r4] = xv;
            Ys[k][kk][r4] = yv;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < kWK; ++kk) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const float4_t x4 = Xs[k][kk][tm], y4 = Ys[ring_partner(NC, k)][kk][tn];
                const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(xv[i], yv[j], acc[i][j]);
                if (k == 0 && tm == 0)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bacc[j] += yv[j];
            }
        }
    }
    // every image entry belongs to exactly one (job, tile, thread): plain stores into this split's partial image
    float* __restrict__ g = gimg + ((int64_t)blockIdx.x * n_nets + net) * net_img_floats;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = tm * 4 + i, n = n0 + tn * 4 + j;
            if (m < jb.M && n < jb.N) g[jb.out + (int64_t)m * jb.sm + (int64_t)n * jb.sn] = acc[i][j];
        }
    if (tm == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tn * 4 + j;
            if (n < jb.N) g[jb.bias + n] = bacc[j];
        }
}

// grad_img[i] (+)= sum over the splits, in split order: the gradient is bitwise reproducible
__global__ void k_wgrad_reduce(const float* __restrict__ partial, int split, int64_t n_img, int accumulate, float* __restrict__ gimg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_img) return;
    float s = accumulate ? gimg[i] : 0.0f;
    for (int p = 0; p < split; ++p) s += partial[(int64_t)p * n_img + i];
    gimg[i] = s;
}

// flat[p] = gradient-image entry of parameter p (inv[p]; -1: the parameter reaches no image entry -- masked weight or
// zero_params leaf -- and its gradient is 0).  A gather over the flat vector: every entry is written, no memset needed.
__global__ void k_grad_gather(const float* __restrict__ gimg, const int32_t* __restrict__ inv, int64_t n_params, float* __restrict__ flat) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_params) return;
    const int32_t t = inv[p];
    flat[p] = t >= 0 ? gimg[t] : 0.0f;
}
// the same reading the per-split partial images directly (single-chunk batches: k_wgrad_reduce + gather in one launch)
__global__ void k_grad_gather_partials(const float* __restrict__ partial, int split, int64_t n_img, const int32_t* __restrict__ inv,
                                       int64_t n_params, float* __restrict__ flat) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_params) return;
    const int32_t t = inv[p];
    float s = 0.0f;
    if (t >= 0)
        for (int q = 0; q < split; ++q) s += partial[(int64_t)q * n_img + t];
    flat[p] = s;
}

// ---- distributed training step: the packed fp64 buffer [gradient, sum E_L, sum E_L^2, n] that travels in the step's one all-reduce
// red[i] = gradient entry i from the per-split partial images (split > 0) or from the flat fp32 gradient (split == 0); the three sums follow
__global__ void k_pack_reduce_buffer(const float* __restrict__ partial, int split, int64_t n_img, const int32_t* __restrict__ inv,
                                     const float* __restrict__ grad, int64_t n_params, double* __restrict__ red) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_params) return;
    float s = 0.0f;
    if (split > 0) {
        const int32_t t = inv[p];
        if (t >= 0)
            for (int q = 0; q < split; ++q) s += partial[(int64_t)q * n_img + t];
    } else {
        s = grad[p];
    }
    red[p] = (double)s;
}
// Adam from the reduced buffer (gradient rounded back to fp32, as the single-GPU step holds it)
__global__ void k_adam_reduced(float* __restrict__ x, const double* __restrict__ red, float* __restrict__ m, float* __restrict__ v, int64_t n,
                               float step_size, float b1, float b2, float eps, const unsigned long long* __restrict__ step_dev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = (float)(*step_dev + 1);
    const float c1 = 1.0f - powf(b1, e), c2 = 1.0f - powf(b2, e);
    const float gi = (float)red[i];
    const float mi = (1.0f - b1) * gi + b1 * m[i];
    const float vi = (1.0f - b2) * gi * gi + b2 * v[i];
    m[i] = mi;
    v[i] = vi;
    x[i] = x[i] - step_size * (mi / c1) / (sqrtf(vi / c2) + eps);
}
// the reduced sums go to slot (counter mod ring_len) of the loss ring; the counter advances
__global__ void k_ring_push(const double* __restrict__ sums, double* __restrict__ ring, int ring_len, unsigned long long* __restrict__ counter) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long c = *counter;
        double* slot = ring + (c % (unsigned long long)ring_len) * 3;
        slot[0] = sums[0]; slot[1] = sums[1]; slot[2] = sums[2];
        *counter = c + 1;
    }
}

// ---- weight images from the flat parameter vector (PackRec, wf_internal.h)
struct PackBases {
    void* img[3];   // plain, wave, mfma image
};
__global__ void k_pack(const float* __restrict__ flat, const PackRec* __restrict__ recs, int64_t n, const PackBases bases) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PackRec r = recs[i];
    void* __restrict__ image = bases.img[r.kind >> 8];
    const bool take_abs = (r.kind & 0x10) != 0;
    r.kind &= 0x0F;
    const float v = r.src >= 0 ? (float)(r.scale * (double)(take_abs ? fabsf(flat[r.src]) : flat[r.src])) : (float)r.scale;
    if (r.kind == 0) {
        reinterpret_cast<float*>(image)[r.dst] = v;
    } else {
        const _Float16 hi = (_Float16)v;
        reinterpret_cast<_Float16*>(image)[r.dst] = hi;
        reinterpret_cast<_Float16*>(image)[r.dst_lo] = (_Float16)(v - (float)hi);
    }
}

// ---- zero_params gradient of gated heads: the reverse sweep leaves one adjoint per (sample, head lane) in zws; two-stage sum over the samples
// in a fixed order (bitwise reproducible like the weight gradients), then the entries go to their leaves of the flat gradient
constexpr int kZSplit = 64;
__global__ void k_zgrad_stage1(const float* __restrict__ zws, int64_t n_samples, int n_rows, float* __restrict__ zpart) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y;
    if (r >= n_rows) return;
    const int64_t lo = n_samples * q / kZSplit, hi = n_samples * (q + 1) / kZSplit;
    float s = 0.0f;
    for (int64_t i = lo; i < hi; ++i) s += zws[i * n_rows + r];
    zpart[(int64_t)q * n_rows + r] = s;
}
__global__ void k_zgrad_stage2(const float* __restrict__ zpart, int n_rows, int accumulate, float* __restrict__ zgrad) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    float s = accumulate ? zgrad[r] : 0.0f;
    for (int q = 0; q < kZSplit; ++q) s += zpart[(int64_t)q * n_rows + r];
    zgrad[r] = s;
}
// flat[zmap[r]] = zgrad[r] (* sign of the raw leaf where the head uses |zero_params|: zraw_off[r] >= 0 is its offset in the plain image)
__global__ void k_zgrad_scatter(const float* __restrict__ zgrad, int n_rows, const int32_t* __restrict__ zmap, const int32_t* __restrict__ zraw_off,
                                const float* __restrict__ plain, float* __restrict__ flat) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int32_t t = zmap[r];
    if (t < 0) return;
    float g = zgrad[r];
    if (zraw_off[r] >= 0) {   // d|z|/dz = sign(z), 0 at z = 0 (as jnp.abs differentiates)
        const float raw = plain[zraw_off[r]];
        g = raw < 0.0f ? -g : (raw > 0.0f ? g : 0.0f);
    }
    flat[t] = g;
}

// ---- Adam as in jax.example_libraries.optimizers.adam (vqmc.py:136), step index i as passed to opt_update
__global__ void k_adam(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n, float c1, float c2,
                       float step_size, float b1, float b2, float eps, const unsigned long long* __restrict__ step_dev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (step_dev) {   // step index from the device counter (captured training step): bias corrections 1 - b^(step + 1)
        const float e = (float)(*step_dev + 1);
        c1 = 1.0f - powf(b1, e);
        c2 = 1.0f - powf(b2, e);
    }
    const float gi = g[i];
    const float mi = (1.0f - b1) * gi + b1 * m[i];
    const float vi = (1.0f - b2) * gi * gi + b2 * v[i];
    m[i] = mi;
    v[i] = vi;
    x[i] = x[i] - step_size * (mi / c1) / (sqrtf(vi / c2) + eps);
}

// the same with g[i] gathered from the per-split partial gradient images (same summation order as k_grad_gather_partials)
__global__ void k_adam_partials(float* __restrict__ x, const float* __restrict__ partial, int split, int64_t n_img, const int32_t* __restrict__ inv,
                                float* __restrict__ m, float* __restrict__ v, int64_t n, float step_size, float b1, float b2, float eps,
                                const unsigned long long* __restrict__ step_dev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float e = (float)(*step_dev + 1);
    const float c1 = 1.0f - powf(b1, e), c2 = 1.0f - powf(b2, e);
    const int32_t t = inv[i];
    float gi = 0.0f;
    if (t >= 0)
        for (int q = 0; q < split; ++q) gi += partial[(int64_t)q * n_img + t];
    const float mi = (1.0f - b1) * gi + b1 * m[i];
    const float vi = (1.0f - b2) * gi * gi + b2 * v[i];
    m[i] = mi;
    v[i] = vi;
    x[i] = x[i] - step_size * (mi / c1) / (sqrtf(vi / c2) + eps);
}

// ---- loss_fn_efficient's tangent rule as per-walker weights (vqmc.py:198-212)
__global__ void k_vqmc_seeds(const float* __restrict__ xg, int64_t B, int D, const Protons pr, const float* __restrict__ hpsi,
                             const float* __restrict__ psi, float running_avg, float inv_count, float* __restrict__ e_loc,
                             float* __restrict__ w_psi, float* __restrict__ w_lap, const float* __restrict__ running_avg_dev) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (running_avg_dev) running_avg = *running_avg_dev;
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
    // 2 (E_L - avg) / psi - H psi / psi^2 (vqmc.py:205-210), written without psi^2: the product of D small factors squared
    // underflows in fp32 for larger D (an 8-electron chain at its initial parameters), and inf - inf would poison the step
    const float a = (2.0f * (el - running_avg) - hp / ps) / ps;
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

template <int D, int NC, int NBK = 1>
int run_wgrad(int n_nets, int64_t n_samples, const float* ws, float* partial, int accumulate, float* grad_img, int64_t net_img_floats,
              int* split_out, hipStream_t s) {
    using R = Rows<D, NBK>;
    constexpr int W = NBP * NBK;
    if (split_out) *split_out = 0;
    if (n_nets == 0 || n_samples == 0) return WF_OK;
    // forward-image layout of one net: W0 [D][64], b0 [64], W1t [64 out][64 in], b1 [64], W2t [D*NBP][64], b2 [D*NBP]
    const int oW0 = 0, ob0 = D * H, oW1 = ob0 + H, ob1 = oW1 + H * H, oW2 = ob1 + H, ob2 = oW2 + D * W * H;
    WJobs jobs;
    jobs.j[0] = WJob{R::U, D, R::A1, H, oW0, H, 1, ob0};            // dW0[a][j]
    jobs.j[1] = WJob{R::H1, H, R::A2, H, oW1, 1, H, ob1};           // dW1t[j][a]
    jobs.j[2] = WJob{R::H2, H, R::O, D * W, oW2, 1, H, ob2};      // dW2t[(d, jb)][a]
    const int n_ntiles = (D * W + kWT - 1) / kWT;
    constexpr int kWK = wgrad_slab(NC);
    const int64_t n_slabs = (n_samples + kWK - 1) / kWK;
    int split = (int)(n_slabs < kWgradSplit ? n_slabs : kWgradSplit);
    if (split < 1) split = 1;
    hipLaunchKernelGGL((k_wgrad<NC>), dim3((unsigned)split, (unsigned)(3 * n_ntiles), (unsigned)n_nets), dim3(256), 0, s, ws, n_nets, n_samples,
                       R::N, jobs, n_ntiles, partial, net_img_floats);
    if (split_out) {   // the caller sums the partial images itself (launch_grad_gather_partials)
        *split_out = split;
        return finish();
    }
    const int64_t n_img = (int64_t)n_nets * net_img_floats;
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((n_img + 255) / 256)), dim3(256), 0, s, (const float*)partial, split, n_img, accumulate,
                       grad_img);
    return finish();
}

}  // namespace

int grad_ws_rows(int D, int nbp) { return 8 + 4 * H + D * nbp; }

// tape -> gradient image: sum over samples of activation (x) adjoint, top ring coefficient
int wgrad_partial_floats(int n_nets, int64_t net_img_floats) { return kWgradSplit * n_nets * (int)net_img_floats; }

// partial: wgrad_partial_floats scratch; accumulate != 0 adds to grad_img (further chunks of a batch) instead of overwriting it
// split_out != NULL: leave the partial images unreduced and report how many there are
int launch_wgrad(int D, int nbp, int ring_kind, int n_nets, int64_t n_samples, const float* ws, float* partial, int accumulate, float* grad_img,
                 int64_t net_img_floats, int* split_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALLK(DD, K) return ring_kind == 2 ? run_wgrad<DD, rf_block(DD) + 2, K>(n_nets, n_samples, ws, partial, accumulate, grad_img, net_img_floats, split_out, s) \
                    : ring_kind == 1 ? run_wgrad<DD, 3, K>(n_nets, n_samples, ws, partial, accumulate, grad_img, net_img_floats, split_out, s)       \
                                     : run_wgrad<DD, 1, K>(n_nets, n_samples, ws, partial, accumulate, grad_img, net_img_floats, split_out, s)
    if (nbp == 64) {
        switch (D) {
            case 2: CALLK(2, 2);
            case 3: CALLK(3, 2);
            case 4: CALLK(4, 2);
            default: return WF_ERR_UNSUPPORTED;
        }
    }
    switch (D) {
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
}

int launch_zgrad_reduce(const float* zws, int64_t n_samples, int n_rows, int accumulate, float* zpart, float* zgrad, void* stream) {
    if (n_rows <= 0) return WF_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_zgrad_stage1, dim3((n_rows + 255) / 256, kZSplit), dim3(256), 0, s, zws, n_samples, n_rows, zpart);
    hipLaunchKernelGGL(k_zgrad_stage2, dim3((n_rows + 255) / 256), dim3(256), 0, s, (const float*)zpart, n_rows, accumulate, zgrad);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_hip_error((int)e); return WF_ERR_HIP; }
    return WF_OK;
}
int launch_zgrad_scatter(const float* zgrad, int n_rows, const int32_t* zmap, const int32_t* zraw_off, const float* plain, float* grad_flat, void* stream) {
    if (n_rows <= 0) return WF_OK;
    hipLaunchKernelGGL(k_zgrad_scatter, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, zgrad, n_rows, zmap, zraw_off, plain, grad_flat);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_hip_error((int)e); return WF_ERR_HIP; }
    return WF_OK;
}

int launch_pack(const float* flat_dev, const PackRec* recs, int64_t n, void* plain, void* wave, void* mfma, void* stream) {
    if (n <= 0) return WF_OK;
    PackBases b;
    b.img[0] = plain; b.img[1] = wave; b.img[2] = mfma;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat_dev, recs, n, b);
    return finish();
}

int launch_adam(float* params, const float* grad, float* m, float* v, int64_t n, int64_t step, float step_size, float b1, float b2, float eps,
                const unsigned long long* step_dev, void* stream) {
    // bias corrections 1 - b^(i+1) in fp32, as the reference's optimiser computes them
    const float c1 = 1.0f - powf(b1, (float)(step + 1)), c2 = 1.0f - powf(b2, (float)(step + 1));
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grad, m, v, n, c1, c2, step_size, b1,
                       b2, eps, step_dev);
    return finish();
}

int launch_adam_partials(float* params, const float* partial, int split, int64_t n_img, const int32_t* inv, float* m, float* v, int64_t n,
                         float step_size, float b1, float b2, float eps, const unsigned long long* step_dev, void* stream) {
    if (n <= 0) return WF_OK;
    hipLaunchKernelGGL(k_adam_partials, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, partial, split, n_img, inv,
                       m, v, n, step_size, b1, b2, eps, step_dev);
    return finish();
}

int launch_pack_reduce_buffer(const float* partial, int split, int64_t n_img, const int32_t* inv, const float* grad, int64_t n_params, double* red,
                              void* stream) {
    if (n_params <= 0) return WF_OK;
    hipLaunchKernelGGL(k_pack_reduce_buffer, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, split, n_img, inv,
                       grad, n_params, red);
    return finish();
}
int launch_adam_reduced(float* params, const double* red, float* m, float* v, int64_t n, float step_size, float b1, float b2, float eps,
                        const unsigned long long* step_dev, void* stream) {
    if (n > 0)
        hipLaunchKernelGGL(k_adam_reduced, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, red, m, v, n, step_size, b1,
                           b2, eps, step_dev);
    return finish();
}
int launch_ring_push(const double* sums, double* ring, int ring_len, unsigned long long* counter, void* stream) {
    hipLaunchKernelGGL(k_ring_push, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, ring, ring_len, counter);
    return finish();
}

int launch_grad_gather(const float* grad_img, const int32_t* inv, int64_t n_params, float* grad_flat, void* stream) {
    if (n_params <= 0) return WF_OK;
    hipLaunchKernelGGL(k_grad_gather, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_img, inv, n_params, grad_flat);
    return finish();
}
int launch_grad_gather_partials(const float* partial, int split, int64_t n_img, const int32_t* inv, int64_t n_params, float* grad_flat, void* stream) {
    if (n_params <= 0) return WF_OK;
    hipLaunchKernelGGL(k_grad_gather_partials, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partial, split, n_img,
                       inv, n_params, grad_flat);
    return finish();
}

int launch_vqmc_seeds(const float* x, int64_t B, int D, const Protons& pr, const float* hpsi, const float* psi, float running_avg,
                      float inv_count, float* e_loc, float* w_psi, float* w_lap, const float* running_avg_dev, void* stream) {
    hipLaunchKernelGGL(k_vqmc_seeds, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, D, pr, hpsi, psi, running_avg,
                       inv_count, e_loc, w_psi, w_lap, running_avg_dev);
    return finish();
}

}  // namespace wf
