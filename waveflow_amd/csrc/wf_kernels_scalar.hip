// wf_kernels_scalar.hip -- "one lane = one walker" evaluation kernel (gfx950).
//
// This is the correctness-first kernel: it keeps the reference's operation order (sequential
// fp32 sums, j-ascending spline sums, the exact floor/ceil index arithmetic) so that it can be
// compared with the CPU oracle almost bit for bit.  Weights are wave-uniform, so hipcc turns
// their loads into scalar (s_load) instructions and every FMA takes its weight from an SGPR;
// the per-walker vectors that need runtime indexing live in thread-private LDS columns.
// The throughput kernel is wf_kernels_mfma.hip; this one also serves configurations that one
// does not cover.
//
// Reference functions restated (paths relative to /root/reference/waveflow):
//   X_cached lerp ............ splines/isplines_jax.py:45-56, msplines_jax.py:30-41, bsplines_jax.py:19-30
//   ispline / bspline sum .... isplines_jax.py:69-79, bsplines_jax.py:42-45
//   remove_bias .............. isplines_jax.py:196-202, msplines_jax.py:186-192
//   enforce_boundary_cond. ... isplines_jax.py:158-194, bsplines_jax.py:173-199, msplines_jax.py:156-184
//   conditioner .............. model_factory.py:21-35, 56-70
//   IMADE / MADE direct ...... flows/bijections/made.py:66-81, :21-27
//   BoxTransformLayer ........ made.py:118-137, :156-183
//   Reverse / Serial ......... flows/bijections/bijections.py:337-340, :452-457
//   Waveflow log_pdf / psi ... wavefunctions.py:33-71;  MFlow / Flow log_pdf: flows/distributions.py:139-163, :95-102
#include "wf_scalar_impl.h"

namespace wf {

namespace scalar {
WF_SCALAR_SHAPE(extern, 2, 32) WF_SCALAR_SHAPE(extern, 3, 32) WF_SCALAR_SHAPE(extern, 4, 32) WF_SCALAR_SHAPE(extern, 5, 32)
WF_SCALAR_SHAPE(extern, 6, 32) WF_SCALAR_SHAPE(extern, 7, 32) WF_SCALAR_SHAPE(extern, 8, 32)
WF_SCALAR_SHAPE(extern, 2, 64) WF_SCALAR_SHAPE(extern, 3, 64) WF_SCALAR_SHAPE(extern, 4, 64)
}  // namespace scalar

namespace {
using namespace scalar;

// deterministic fp64 block sums: stage 1 one partial pair per block, stage 2 one block
constexpr int kSumBlock = 256;
constexpr int kSumMaxBlocks = 1024;

__device__ __forceinline__ void block_reduce2(double& a, double& b, double* sm) {
    // fixed-order tree in LDS: result independent of scheduling
    sm[threadIdx.x] = a;
    sm[kSumBlock + threadIdx.x] = b;
    __syncthreads();
    for (int s = kSumBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sm[threadIdx.x] += sm[threadIdx.x + s];
            sm[kSumBlock + threadIdx.x] += sm[kSumBlock + threadIdx.x + s];
        }
        __syncthreads();
    }
    a = sm[0];
    b = sm[kSumBlock];
}

// end of a captured training step, done by the thread that holds the sums: they go to slot (counter mod ring_len) of the loss ring and
// the counter advances (it seeds the next step's sampler and is Adam's step index)
__device__ __forceinline__ void ring_push(double s, double q, double n, double* __restrict__ ring, int ring_len, unsigned long long* __restrict__ counter) {
    const unsigned long long c = *counter;
    double* slot = ring + (c % (unsigned long long)ring_len) * 3;
    slot[0] = s; slot[1] = q; slot[2] = n;
    *counter = c + 1;
}

// final != NULL (single-block launch): the block's sums are the result, stage 2 is skipped
__global__ __launch_bounds__(kSumBlock) void k_sums_stage1(const float* __restrict__ v, int64_t B, double* __restrict__ partial,
                                                           double* __restrict__ final, double* __restrict__ ring, int ring_len,
                                                           unsigned long long* __restrict__ counter) {
    __shared__ double sm[2 * kSumBlock];
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kSumBlock + threadIdx.x; i < B; i += (int64_t)gridDim.x * kSumBlock) {
        const double x = (double)v[i];
        s += x;
        q += x * x;
    }
    block_reduce2(s, q, sm);
    if (threadIdx.x == 0) {
        if (final) {
            final[0] = s;
            final[1] = q;
            final[2] = (double)B;
            if (ring) ring_push(s, q, (double)B, ring, ring_len, counter);
        } else {
            partial[2 * blockIdx.x] = s;
            partial[2 * blockIdx.x + 1] = q;
        }
    }
}

__global__ __launch_bounds__(kSumBlock) void k_sums_stage2(const double* __restrict__ partial, int n_blocks, int64_t B, double* __restrict__ out,
                                                           double* __restrict__ ring, int ring_len, unsigned long long* __restrict__ counter) {
    __shared__ double sm[2 * kSumBlock];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += kSumBlock) {
        s += partial[2 * i];
        q += partial[2 * i + 1];
    }
    block_reduce2(s, q, sm);
    if (threadIdx.x == 0) {
        out[0] = s;
        out[1] = q;
        out[2] = (double)B;
        if (ring) ring_push(s, q, (double)B, ring, ring_len, counter);
    }
}

int sums_blocks(int64_t B) {
    int64_t n = (B + kSumBlock - 1) / kSumBlock;
    if (n < 1) n = 1;
    if (n > kSumMaxBlocks) n = kSumMaxBlocks;
    return (int)n;
}

int finish_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

}  // namespace

#define WF_DISPATCH_D(D_, NBP_, CALL)                                         \
    if ((NBP_) == 32) {                                                          \
        switch (D_) {                                                            \
            case 2: CALL(2, 32); break;                                          \
            case 3: CALL(3, 32); break;                                          \
            case 4: CALL(4, 32); break;                                          \
            case 5: CALL(5, 32); break;                                          \
            case 6: CALL(6, 32); break;                                          \
            case 7: CALL(7, 32); break;                                          \
            case 8: CALL(8, 32); break;                                          \
            default: return WF_ERR_UNSUPPORTED;                                  \
        }                                                                        \
    } else if ((NBP_) == 64) {                                                   \
        switch (D_) {                                                            \
            case 2: CALL(2, 64); break;                                          \
            case 3: CALL(3, 64); break;                                          \
            case 4: CALL(4, 64); break;                                          \
            default: return WF_ERR_UNSUPPORTED;                                  \
        }                                                                        \
    } else {                                                                     \
        return WF_ERR_UNSUPPORTED;                                               \
    }

int launch_scalar(const ModelDev& md, const ModelDev* md_dev, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx,
                  void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALL(DD, NN) run_eval<DD, NN>(md_dev, mode, x, B, out, u, idx, s)
    WF_DISPATCH_D(md.D, md.nbp, CALL)
#undef CALL
    return finish_launch();
}

int launch_scalar_layer(const ModelDev& md, const ModelDev* md_dev, int layer, const float* u_in, int64_t B, float* y, float* logdet,
                        int32_t* idx, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALL(DD, NN) run_layer<DD, NN>(md_dev, layer, u_in, B, y, logdet, idx, s)
    WF_DISPATCH_D(md.D, md.nbp, CALL)
#undef CALL
    return finish_launch();
}

int launch_scalar_inverse(const ModelDev& md, const ModelDev* md_dev, const float* u, int64_t B, float* x, int exact, void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALL(DD, NN) run_inverse<DD, NN>(md_dev, u, B, x, exact, s)
    WF_DISPATCH_D(md.D, md.nbp, CALL)
#undef CALL
    return finish_launch();
}

int launch_scalar_sample(const ModelDev& md, const ModelDev* md_dev, unsigned long long seed, int64_t B, float* x, float* latent, int exact,
                         void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define CALL(DD, NN) run_sample<DD, NN>(md_dev, seed, B, x, latent, exact, s)
    WF_DISPATCH_D(md.D, md.nbp, CALL)
#undef CALL
    return finish_launch();
}

// ---- walkers in any order (helpers.py:55-58, coordinates.py:41-51): per row the ascending sort and the inversion count.  The MFMA kernel does
// both in registers (mode bit kModePresort); the other kernels get sorted rows from here and psi its sign afterwards.
__global__ void k_sort_rows(const float* __restrict__ x, int64_t B, int D, float* __restrict__ xs, int32_t* __restrict__ inv) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float r[WF_MAX_DIM];
    for (int d = 0; d < D; ++d) r[d] = x[b * D + d];
    int n = 0;
    for (int pass = 0; pass < D; ++pass)          // odd-even transposition: adjacent exchanges only, #exchanges == #inversions (ties: none)
        for (int i = pass & 1; i + 1 < D; i += 2)
            if (r[i] > r[i + 1]) {
                const float t = r[i]; r[i] = r[i + 1]; r[i + 1] = t;
                ++n;
            }
    if (xs)
        for (int d = 0; d < D; ++d) xs[b * D + d] = r[d];
    if (inv) inv[b] = n;
}
__global__ void k_apply_sign(float* __restrict__ v, const int32_t* __restrict__ inv, int64_t B) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && (inv[b] & 1)) v[b] = -v[b];
}

int launch_sort_rows(const float* x, int64_t B, int D, float* xs, int32_t* inv, void* stream) {
    if (B <= 0) return WF_OK;
    hipLaunchKernelGGL(k_sort_rows, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, D, xs, inv);
    return finish_launch();
}
int launch_apply_sign(float* v, const int32_t* inv, int64_t B, void* stream) {
    if (B <= 0) return WF_OK;
    hipLaunchKernelGGL(k_apply_sign, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, inv, B);
    return finish_launch();
}

int64_t block_sums_ws_bytes(int64_t) { return (int64_t)kSumMaxBlocks * 2 * sizeof(double); }

// ring != NULL: the kernel that ends up with the sums also pushes them to the loss ring and advances the step counter
int launch_block_sums(const float* v, int64_t B, double* out, void* ws, int64_t, void* stream, double* ring, int ring_len,
                      unsigned long long* counter) {
    hipStream_t s = (hipStream_t)stream;
    const int nb = sums_blocks(B);
    double* partial = (double*)ws;
    if (nb == 1) {   // small batches (a training step's 128..256 local energies): one launch
        hipLaunchKernelGGL(k_sums_stage1, dim3(1), dim3(kSumBlock), 0, s, v, B, partial, out, ring, ring_len, counter);
        return finish_launch();
    }
    hipLaunchKernelGGL(k_sums_stage1, dim3(nb), dim3(kSumBlock), 0, s, v, B, partial, (double*)nullptr, (double*)nullptr, 0,
                       (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_sums_stage2, dim3(1), dim3(kSumBlock), 0, s, (const double*)partial, nb, B, out, ring, ring_len, counter);
    return finish_launch();
}

}  // namespace wf
