// wf_kernels_mfma.hip -- throughput kernel: one wave = one tile of 32 walkers, conditioner GEMMs on the
// matrix cores, everything else fused around them (gfx950 / CDNA4).
//
// Orientation.  Every dense layer is computed transposed, OUT[unit][walker] = W^T[unit][k] * IN[k][walker]:
//   * the MFMA C/D layout puts the walker on the lane (column j = lane & 31) and the 32 output units of a
//     block in the 16 accumulator registers of the two lane halves (row = (r&3) + 8*(r>>2) + 4*(lane>>5));
//   * that is also the B-operand layout of the next layer's MFMA (lane half h supplies the k's of its own
//     registers), so an accumulator tile -- after the activation -- IS the next B operand: the whole
//     D -> 64 -> 64 -> D*n_bases chain runs with no LDS transposes and no cross-lane traffic;
//   * the A operand (weights) is pre-permuted on the host into that k order and streamed from LDS
//     (wf_model.cpp: build_mfma_image).
// Precision.  On gfx950 the f32-input MFMA runs at the f32 VALU rate and does not overlap with VALU work
// (measured: profiles/r01_ubench_coexec.txt), so the two K=64 layers use v_mfma_f32_32x32x16_f16 with a
// two-way fp16 split of both operands: x = hi + lo (hi = rn16(x), lo = rn16(x - hi); fp16 subnormals are not
// flushed, so lo keeps 2^-25 absolute precision), three products hi*hi + hi*lo + lo*hi accumulated in fp32.  Its error is
// indistinguishable from an fp32 FMA chain (dominated by the fp32 accumulation; tests/test_gpu_parity.py).
// The K = D input layer and the 32x32 ob_to_b product of the B-prior stay on v_mfma_f32_32x32x2_f32.
// Constant folding done on the host: 2*log2(e) into (W0,b0,W1,b1) so tanh(x) = 1 - 2/(2^x' + 1) needs no
// scaling multiply, -log2(e) into (W2,b2) of sigmoid heads, +1e30 biases on padding rows (sigmoid -> 0), the
// remove_bias / boundary-condition row factors f_j*keep_j into the spline tables (plus their row sums).
// Algebra used (exact in real arithmetic, fewer roundings than the reference's sequence):
//   with v_j = sigmoid(o_j), S1 = sum v, the reference's weights are c_j = q_j / sum(q),
//   q_j = (v_j + reg*S1) * f_j * keep_j (made.py:66-79, isplines_jax.py:158-202), hence
//   y = [sum_j v_j T'_j(x) + reg*S1*R(x)] / [sum_j v_j fk_j + reg*S1*F] with T' = fk*T, R = sum_j T'_j, F = sum fk.
//   For the B-spline prior the two L2 normalisations and the division by the signed sum collapse to
//   psi_d = sign(sum o) * (c . lerp) / |c|, c = (o * keep) @ ob_to_b (wavefunctions.py:40-46,
//   bsplines_jax.py:127-137, 173-199).
//   Output dimension 0 of every net depends on no input (output degree -1, model_factory.py:15-18): its spline
//   weights are per-net constants and, the lerp being linear in the table, sum_j c_j lerp(T_j, x) = lerp(sum_j c_j T_j, x):
//   a composite table per net (k_prepare_dim0, rebuilt at every parameter upload) turns that block into two lerps.
// The table index arithmetic (floor/ceil of u * (n_mesh-1), isplines_jax.py:46-48) is kept verbatim.
#include "wf_mfma_impl.h"

#include <cmath>

namespace wf {

namespace mfma {
// shapes built in wf_mfma_inst_*.hip
#define WF_MFMA_EXTERN(DD, KK, WW, TT) \
    extern template int launch_dw<DD, KK, WW, TT>(const MfmaDev*, int, int, const float*, int64_t, float*, float*, int32_t*, hipStream_t)
WF_MFMA_EXTERN(2, 1, 8, 1); WF_MFMA_EXTERN(2, 1, 12, 1); WF_MFMA_EXTERN(2, 1, 16, 1);
WF_MFMA_EXTERN(2, 1, 8, 2); WF_MFMA_EXTERN(2, 1, 4, 2);
WF_MFMA_EXTERN(3, 1, 8, 1); WF_MFMA_EXTERN(4, 1, 8, 1); WF_MFMA_EXTERN(8, 1, 8, 1);
WF_MFMA_EXTERN(5, 1, 8, 1); WF_MFMA_EXTERN(6, 1, 8, 1); WF_MFMA_EXTERN(7, 1, 8, 1);
WF_MFMA_EXTERN(2, 2, 8, 1); WF_MFMA_EXTERN(3, 2, 8, 1); WF_MFMA_EXTERN(4, 2, 8, 1);
WF_MFMA_EXTERN(2, 2, 12, 1); WF_MFMA_EXTERN(2, 2, 16, 1);
#ifdef WF_D8_WAVES_ALL
WF_MFMA_EXTERN(8, 1, 12, 1); WF_MFMA_EXTERN(8, 1, 16, 1);
#endif
#undef WF_MFMA_EXTERN
}  // namespace mfma

namespace {
using mfma::f32x4;
using mfma::launch_dw;

// ---- composite tables of output dimension 0 (see comp_lerp), from the plain weight image, in two stages with the arithmetic
// order of the straightforward per-mesh-point evaluation (the tables are bit-identical to it):
//   k_dim0_coeffs   one workgroup per net: the walker-independent spline coefficients of dimension 0 (<= 64 values + 2 scalars)
//   k_prepare_dim0  one thread per (net, mesh point): their dot product with that mesh row
constexpr int kCoefStride = 66;   // [64 coefficients][scale][unused]
__global__ __launch_bounds__(64) void k_dim0_coeffs(const ModelDev* __restrict__ mdp, const float* __restrict__ fk_nat /* [2][64]: I, prior */,
                                                    float F_I, float F_P, float* __restrict__ coef) {
    __shared__ float sh[64];
    const ModelDev& md = *mdp;
    const int n = blockIdx.x, t = threadIdx.x;
    const NetPlain& net = md.nets[n];
    const bool is_prior = n == md.n_layers;
    float* out = coef + (size_t)n * kCoefStride;
    if (!is_prior && md.layer_kind == WF_LAYER_MADE) {
        if (t < 2) out[t] = net.b2[t];   // log_weight, bias of dimension 0 (rows j = 0 / 1 of block d = 0)
        return;
    }
    if (is_prior && md.prior_kind == WF_PRIOR_WAVEFLOW) {
        const SplineDev& sp = md.psp;
        const int nb = sp.nb, nbp = sp.nbp;
        const float* keep = fk_nat + 64;
        // dimension 0 of a gated head: g = 1, w = o + z (model_factory.py:64-67); net.zero holds zeros for an ungated net
        const bool gate = md.p_gate != 0;
        float c = 0.0f;
        if (t < nb) {
            for (int a = 0; a < nb; ++a) c = __builtin_fmaf((gate ? net.b2[a] + net.zero[a] : net.b2[a]) * keep[a], md.ob_to_b_t[a * nbp + t], c);
            if (md.p_cb) {   // constant term of the boundary map (a constraint with a non-zero value): + (sum of the raw outputs) * cb
                float sraw = 0.0f;
                for (int jj = 0; jj < nb; ++jj) sraw += gate ? net.b2[jj] + net.zero[jj] : net.b2[jj];
                c = __builtin_fmaf(sraw, md.p_cb[t], c);
            }
        }
        sh[t] = c;
        out[t] = c;
        __syncthreads();
        if (t == 0) {
            float s1 = 0.0f, n2 = 0.0f;
            for (int j = 0; j < nb; ++j) s1 += gate ? net.b2[j] + net.zero[j] : net.b2[j];
            for (int i = 0; i < nb; ++i) n2 = __builtin_fmaf(sh[i], sh[i], n2);
            out[64] = (s1 < 0.0f ? -1.0f : 1.0f);
            out[65] = n2;
        }
        return;
    }
    const SplineDev& sp = is_prior ? md.psp : md.isp;
    const int nb = sp.nb;
    const float* fk = fk_nat + (is_prior ? 64 : 0);
    const float reg = is_prior ? 0.0f : md.i_reg, F = is_prior ? F_P : F_I;
    float v = t < nb ? 1.0f / (1.0f + expf(-net.b2[t])) : 0.0f;
    if ((is_prior ? md.p_gate : md.i_gate) != 0 && t < nb) v = v + net.zero[t];   // gated head, dimension 0: g = 1
    sh[t] = v;
    __syncthreads();
    float s1 = 0.0f, sf = 0.0f;   // every thread repeats the two ordered sums (64 terms): no second barrier
    for (int j = 0; j < nb; ++j) {
        s1 += sh[j];
        sf = __builtin_fmaf(sh[j], fk[j], sf);
    }
    const float rs = reg * s1, rS = 1.0f / __builtin_fmaf(rs, F, sf);
    out[t] = t < nb ? (v + rs) * fk[t] : 0.0f;
    if (t == 0) out[64] = rS;
}

// sum_j c_j row[j], j ascending (the order of the per-mesh-point evaluation), the row fetched with 16-byte loads
__device__ __forceinline__ float row_dot(const float* __restrict__ row, const float* __restrict__ c, int nb) {
    float acc = 0.0f;
    for (int q = 0; 4 * q < nb; ++q) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(row + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * q + e < nb) acc = __builtin_fmaf(c[4 * q + e], t[e], acc);
    }
    return acc;
}

// tab_i / tab_p: the natural-order tables [orders][n_mesh][nbp] with the boundary map folded into their rows (wf_model.cpp: bc_map; the
// wave kernels' d_tabI4 / d_tabP3) -- identical to ModelDev's plain ones when the constraints only zero end coefficients
__global__ void k_prepare_dim0(const ModelDev* __restrict__ mdp, int nm, const float* __restrict__ coef, const float* __restrict__ tab_i,
                               const float* __restrict__ tab_p, f32x4* __restrict__ comp) {
    // one thread per (net, mesh point, derivative order) since round 4 (one per (net, mesh point) before: 32 workgroups, four dependent row sums per thread,
    // 20 us of every training step); each row sum in the same order as before: the tables keep their bits
    const ModelDev& md = *mdp;
    const int n_nets = md.n_layers + ((md.prior_kind == WF_PRIOR_WAVEFLOW || md.prior_kind == WF_PRIOR_MFLOW) ? 1 : 0);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = tid >> 2, o = tid & 3;
    if (nm <= 0 || gid >= n_nets * nm) return;
    const int n = gid / nm, m = gid % nm;
    const bool is_prior = n == md.n_layers;
    const float* __restrict__ c = coef + (size_t)n * kCoefStride;
    float out = 0.0f;
    if (!is_prior && md.layer_kind == WF_LAYER_MADE) {
        if (o < 2) out = c[o];
    } else if (is_prior && md.prior_kind == WF_PRIOR_WAVEFLOW) {
        const SplineDev& sp = md.psp;
        const int nb = sp.nb, nbp = sp.nbp;
        if (o == 0) out = (c[64] < 0.0f ? -row_dot(tab_p + (size_t)m * nbp, c, nb) : row_dot(tab_p + (size_t)m * nbp, c, nb)) * __builtin_amdgcn_rsqf(c[65]);
        else if (o < 3) {
            // derivative orders 1, 2 with the same sign and norm: the jet of psi_0 (local energy on the matrix cores, wf_kernels_etile.hip)
            const float sn = (c[64] < 0.0f ? -1.0f : 1.0f) * __builtin_amdgcn_rsqf(c[65]);
            out = row_dot(tab_p + ((size_t)o * sp.n_mesh + m) * nbp, c, nb) * sn;
        }
    } else {
        const SplineDev& sp = is_prior ? md.psp : md.isp;
        const float* __restrict__ tab = is_prior ? tab_p : tab_i;
        const int nb = sp.nb, nbp = sp.nbp;
        // (orders 2, 3 of a flow layer: jets of y_0 and of log dy_0; an M-spline prior has the value alone)
        if (o == 0 || !is_prior) out = row_dot(tab + ((size_t)o * sp.n_mesh + m) * nbp, c, nb) * c[64];
    }
    reinterpret_cast<float*>(comp)[(size_t)gid * 4 + o] = out;
}

// comp2[n][m] = {comp[n][m].xy, comp[n][m + 1].xy}: the two lerp ends k_mfma needs of a composite table in one 16-byte record
__global__ void k_pair_dim0(const f32x4* __restrict__ comp, int n_nets, int nm, f32x4* __restrict__ comp2) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_nets * nm) return;
    const int m = gid % nm;
    const f32x4 a = comp[gid], b = comp[m + 1 < nm ? gid + 1 : gid];
    comp2[gid] = f32x4{a[0], a[1], b[0], b[1]};
}

// tuning knobs of the headline shape (read at every launch): WF_MFMA_WAVES = 4, 8, 12 or 16 waves per workgroup,
// WF_MFMA_TILES = 1 or 2 tiles of 32 walkers per wave (built: 8 / 12 / 16 waves x 1 tile, 4 / 8 waves x 2 tiles)
// b' = b + sum_k W_k for the layers whose input is a tanh (see act_split_block): the packed weights already hold -2 c W as fp16
// pairs in MFMA A-operand order, so the column sum is -0.5 * sum over the 64 k's of (hi + lo).  One thread per bias entry, fixed
// summation order (deterministic); runs after k_pack at every parameter upload.  Padding rows have zero weights: unchanged.
// ovf[net] (may be null) = 1 when a packed weight of the net left the fp16 range (|scale * W| >= 65 520: its hi half is +-inf), else 0 -- rewritten at
// every upload.  The matrix-core kernels would turn such a weight into inf / NaN for every walker; wf_model.cpp routes around them while it is set
// (include/waveflow_hip.h: "fp16 range") and k_mfma / k_efused poison their outputs with NaN if they are launched anyway (a captured graph).
__global__ void k_fold_bias(float* __restrict__ image, int net_floats, int D, int nbk, int* __restrict__ ovf) {
    const int S0 = (D + 1) / 2;
    const int W1h = 128 * S0 + 64, b1 = W1h + 4096, W2h = b1 + 64;
    const int n_out_blocks = (D - 1) * nbk;
    const int b2 = W2h + n_out_blocks * 2048;
    float* net = image + (size_t)blockIdx.x * net_floats;
    const int n_entries = 64 + n_out_blocks * 32;
    int bad = 0;
    for (int e = threadIdx.x; e < n_entries; e += blockDim.x) {
        const _Float16* halves;
        int n_pairs, blk, bias_at;
        int h, r;
        if (e < 64) {   // hidden layer 2: bias entry (ob, h, r)
            const int ob = e >> 5;
            h = (e >> 4) & 1; r = e & 15;
            halves = reinterpret_cast<const _Float16*>(net + W1h);
            n_pairs = 4096; blk = ob; bias_at = b1 + e;
        } else {        // output layer, dimension d >= 1, row block kb
            const int q = e - 64, ob = q >> 5;
            h = (q >> 4) & 1; r = q & 15;
            const int d = 1 + ob / nbk, kb = ob % nbk;
            halves = reinterpret_cast<const _Float16*>(net + W2h);
            n_pairs = n_out_blocks * 2048; blk = ob; bias_at = b2 + ((d * nbk + kb) * 2 + h) * 16 + r;
        }
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;   // unit / basis row inside the 32-row block (accumulator order)
        float sum = 0.0f;
        for (int ts = 0; ts < 4; ++ts)
            for (int hh = 0; hh < 2; ++hh)
                for (int j = 0; j < 8; ++j) {
                    const int idx = ((blk * 4 + ts) * 64 + row + 32 * hh) * 8 + j;
                    const float hi = (float)halves[idx];
                    bad |= !(fabsf(hi) <= 65504.0f);   // +-inf (or NaN from a NaN parameter)
                    sum += hi + (float)halves[n_pairs + idx];
                }
        net[bias_at] += -0.5f * sum;
    }
    bad = __syncthreads_or(bad);
    if (ovf && threadIdx.x == 0) ovf[blockIdx.x] = bad ? 1 : 0;
}

int waves_per_group(int tiles, int nbk) {
    const char* e = getenv("WF_MFMA_WAVES");
    const int v = e ? atoi(e) : 0;
    if (tiles == 2) return (v == 4 || v == 8) ? v : 8;
    // (two row blocks per dimension: 0.347 ms at 12 waves, 0.354 at 16, 0.369 at 8 -- scratch/time_waves.py, 33-knot He, 2^20 walkers)
    return (v == 8 || v == 12 || v == 16) ? v : (nbk == 2 ? 12 : 16);
}
int tiles_per_wave() {
    const char* e = getenv("WF_MFMA_TILES");
    const int v = e ? atoi(e) : 0;
    return (v == 1 || v == 2) ? v : 1;
}

}  // namespace

int mfma_extra_lds_floats(int) { return 0; }

// div_by_n of wf_mfma_impl.h, restated on the host: is q = fma(fma(-x*rn, n, x), rn, x*rn) the correctly rounded x / n for every
// integer x the bin-index arithmetic can produce?
bool mfma_div_ok(int n_mesh) {
    const float n = (float)(n_mesh - 1), rn = 1.0f / n;
    for (int x = -1; x <= n_mesh; ++x) {
        const float xf = (float)x, q = xf * rn;
        const float r = fmaf(-q, n, xf);
        if (fmaf(r, rn, q) != xf / n) return false;
    }
    return true;
}
int dim0_coef_floats(int n_nets) { return n_nets * kCoefStride; }

int launch_fold_bias(float* image_dev, int n_nets, int net_floats, int D, int nbk, int* ovf_dev, void* stream) {
    if (n_nets <= 0 || !image_dev) return WF_OK;
    hipLaunchKernelGGL(k_fold_bias, dim3(n_nets), dim3(128), 0, (hipStream_t)stream, image_dev, net_floats, D, nbk, ovf_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

int launch_prepare_dim0(const ModelDev* md_dev, int n_nets, int n_mesh, const float* fk_nat_dev, float F_I, float F_P, const float* tab_i_dev,
                        const float* tab_p_dev, void* comp_dev, void* stream) {
    const int total = n_nets * n_mesh;
    if (total <= 0) return WF_OK;
    // the coefficient block sits behind the tables in the same allocation (wf_model.cpp reserves dim0_coef_floats)
    float* coef = reinterpret_cast<float*>(comp_dev) + (size_t)total * 4;
    hipLaunchKernelGGL(k_dim0_coeffs, dim3(n_nets), dim3(64), 0, (hipStream_t)stream, md_dev, fk_nat_dev, F_I, F_P, coef);
    hipLaunchKernelGGL(k_prepare_dim0, dim3((4 * total + 255) / 256), dim3(256), 0, (hipStream_t)stream, md_dev, n_mesh, (const float*)coef,
                       tab_i_dev, tab_p_dev, reinterpret_cast<f32x4*>(comp_dev));
    f32x4* comp2 = reinterpret_cast<f32x4*>(coef + ((dim0_coef_floats(n_nets) + 3) & ~3));
    hipLaunchKernelGGL(k_pair_dim0, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const f32x4*>(comp_dev), n_nets,
                       n_mesh, comp2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

int launch_mfma(int D, int nbk, const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx,
                void* stream) {
    hipStream_t s = (hipStream_t)stream;
#define GO(DD, KK, WW) return launch_dw<DD, KK, WW, 1>(mdev, lds_bytes, mode, x, B, out, u, idx, s)
    if (D == 2 && nbk == 1) {   // the headline shape: several workgroup shapes are built (tuning / reproducibility test)
        if (tiles_per_wave() == 2 && !idx) {   // (bin indices: one-tile kernels only)
            if (waves_per_group(2, 1) == 4) return launch_dw<2, 1, 4, 2>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
            return launch_dw<2, 1, 8, 2>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        }
        switch (waves_per_group(1, 1)) {
            case 8: GO(2, 1, 8);
            case 12: GO(2, 1, 12);
            default: GO(2, 1, 16);
        }
    }
    if (nbk == 1) {
        switch (D) {
            case 3: GO(3, 1, 8);
            case 4: GO(4, 1, 8);
            case 5: GO(5, 1, 8);
            case 6: GO(6, 1, 8);
            case 7: GO(7, 1, 8);
            case 8:
#ifdef WF_D8_WAVES_ALL
                if (waves_per_group(1, 1) == 12 && getenv("WF_MFMA_WAVES")) GO(8, 1, 12);
                if (waves_per_group(1, 1) == 16 && getenv("WF_MFMA_WAVES")) GO(8, 1, 16);
#endif
                GO(8, 1, 8);   // (16 waves: 0.62 ms against 0.45 ms at 2^18 walkers -- register spills)
            default: return WF_ERR_UNSUPPORTED;
        }
    }
    if (nbk == 2) {
        switch (D) {
            case 2:
                switch (waves_per_group(1, 2)) {
                    case 8: GO(2, 2, 8);
                    case 12: GO(2, 2, 12);
                    default: GO(2, 2, 16);
                }
            case 3: GO(3, 2, 8);
            case 4: GO(4, 2, 8);
            default: return WF_ERR_UNSUPPORTED;
        }
    }
#undef GO
    return WF_ERR_UNSUPPORTED;
}

bool mfma_shape_built(int D, int nbk) {
    return (nbk == 1 && D >= 2 && D <= 8) || (nbk == 2 && D >= 2 && D <= 4);
}

}  // namespace wf
