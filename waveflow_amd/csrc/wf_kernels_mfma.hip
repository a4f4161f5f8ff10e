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
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "wf_internal.h"

namespace wf {

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;


__device__ __forceinline__ float act_tanh(float xs) {  // xs = 2*log2(e)*x (scale folded into the weights)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.0f), 1.0f);
}
__device__ __forceinline__ float act_sigmoid(float xs) {  // xs = -log2(e)*x
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xs));
}
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// sum of the two lane halves (lane l and l^32), result in every lane.
// v_permlane32_swap is issued from inline asm with its own wait states: with the builtin, hipcc (ROCm 7.2) pads
// only the "VALU write -> permlane read" side, and this kernel then produced wrong sums on a few tiles per
// launch (non-deterministically; gone with ds_bpermute, gone with the padding below).  See DESIGN.md §9.
__device__ __forceinline__ float xhalf_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    return a + b;
}

__device__ __forceinline__ f32x16 load16(const float* p) {
    const f32x4* q = reinterpret_cast<const f32x4*>(p);
    const f32x4 a = q[0], b = q[1], c = q[2], d = q[3];
    return f32x16{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3], c[0], c[1], c[2], c[3], d[0], d[1], d[2], d[3]};
}

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

// Output dimension 0 of every net has an empty mask (model_factory.py:15-18): its spline weights do not depend on the
// walker, and a lerp is linear in the table values, so  sum_j c_j lerp(T_j, x) == lerp(sum_j c_j T_j, x).  The composite
// tables (value, derivative) of every net are built once per parameter upload by k_prepare_dim0; a dimension-0 block is
// then two 16-byte loads and two lerps.  comp[net][mesh] = {Y, DY, 0, 0} (flow layers: spline value and derivative,
// already divided by sum(q); B prior: psi_0 with its sign and norm; M prior: density; MADE: {log_weight, bias}).
__device__ __forceinline__ f32x4 comp_lerp(const f32x4* __restrict__ comp, const Lerp& Lp) {
    const f32x4 a = comp[Lp.il], b = comp[Lp.ir];
    f32x4 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = __builtin_fmaf(b[q] - a[q], Lp.t, a[q]);
    return r;
}

// 32 activations of one block (accumulator layout) -> the two K=16 B fragments, split hi / lo
struct Frag {
    f16x8 hi[2], lo[2];
};
__device__ __forceinline__ void split_block(const f32x16& x, Frag& f) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[8 * s + j];
            const _Float16 h = (_Float16)v;
            f.hi[s][j] = h;
            f.lo[s][j] = (_Float16)(v - (float)h);   // exact difference; fp16 subnormals keep 2^-25 absolute precision
        }
}

// one 32-unit output block of a K=64 layer: acc += Ahi*Bhi + Ahi*Blo + Alo*Bhi (fp32 accumulation)
// Wh / Wl: LDS images [t][s][lane][8 halves] of this block
__device__ __forceinline__ f32x16 dense64_block(const _Float16* Wh, const _Float16* Wl, const Frag (&in)[2], f32x16 bias, int lane) {
    f32x16 acc = bias;
    // Scheduling fences around the f16 MFMA chain: when hipcc (ROCm 7.2) interleaved unrelated VALU / memory
    // instructions of the neighbouring code into this chain, a few tiles per launch came out wrong,
    // non-deterministically.  With the chain fenced the kernel is bit-reproducible; cost < 1 %.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(Wh + ((t * 2 + s) * 64 + lane) * 8);
            const f16x8 al = *reinterpret_cast<const f16x8*>(Wl + ((t * 2 + s) * 64 + lane) * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, in[t].hi[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[t].lo[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, in[t].hi[s], acc, 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}

// float offsets inside a net image (wf_model.cpp: build_mfma_image)
template <int D>
struct NetOff {
    static constexpr int S0 = (D + 1) / 2;
    static constexpr int W0 = 0;
    static constexpr int b0 = W0 + 2 * S0 * 64;
    static constexpr int W1h = b0 + 64;
    static constexpr int W1l = W1h + 2048;
    static constexpr int b1 = W1l + 2048;
    static constexpr int W2h = b1 + 64;
    static constexpr int W2l = W2h + (D - 1) * 1024;
    static constexpr int b2 = W2l + (D - 1) * 1024;
    static constexpr int total = b2 + 32 * D;
};

// Hidden layers of one conditioner net for the wave's 32 walkers; result: second hidden layer as B fragments.
template <int D>
__device__ __forceinline__ void hidden_layers(const float* net, const float (&in)[D], int lane, Frag (&h2)[2]) {
    using O = NetOff<D>;
    const int h = lane >> 5;
    Frag h1[2];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        f32x16 a = load16(net + O::b0 + (ob * 2 + h) * 16);
#pragma unroll
        for (int s = 0; s < O::S0; ++s) {
            const float w = net[O::W0 + (ob * O::S0 + s) * 64 + lane];
            const float lo = in[2 * s];
            const float hi = (2 * s + 1 < D) ? in[(2 * s + 1 < D) ? 2 * s + 1 : D - 1] : 0.0f;
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(w, h ? hi : lo, a, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = act_tanh(a[r]);
        split_block(a, h1[ob]);
    }
    const _Float16* W1h = reinterpret_cast<const _Float16*>(net + O::W1h);
    const _Float16* W1l = reinterpret_cast<const _Float16*>(net + O::W1l);
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        f32x16 a = dense64_block(W1h + ob * 2048, W1l + ob * 2048, h1, load16(net + O::b1 + (ob * 2 + h) * 16), lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = act_tanh(a[r]);
        split_block(a, h2[ob]);
    }
}

// Output block of dimension d >= 1: raw (scaled) outputs o[basis row][walker] in accumulator layout.
template <int D>
__device__ __forceinline__ f32x16 out_block(const float* net, const Frag (&h2)[2], int d, int lane) {
    using O = NetOff<D>;
    const int h = lane >> 5;
    const _Float16* W2h = reinterpret_cast<const _Float16*>(net + O::W2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(net + O::W2l);
    return dense64_block(W2h + (d - 1) * 2048, W2l + (d - 1) * 2048, h2, load16(net + O::b2 + (d * 2 + h) * 16), lane);
}

// The 16 table values of this lane half at x_l (a) and x_r (b) for one derivative order.
// (The rows depend on the layer input only, but requesting them before the conditioner MFMAs costs 64 live VGPRs,
// i.e. a wave per SIMD, and measured no faster: the kernel is issue-bound, not latency-bound.)
struct Rows {
    f32x16 a, b;
};
__device__ __forceinline__ Rows load_rows(const float* __restrict__ tl, const float* __restrict__ tr) {
    Rows r;
    r.a = load16(tl);
    r.b = load16(tr);
    return r;
}

// sum_r v_r * lerp(T'_r) over the walker's 32 rows (both lane halves summed)
__device__ __forceinline__ float lerp_dot(const f32x16& v, const Rows& R, float t) {
    float sa0 = 0.0f, sa1 = 0.0f, sb0 = 0.0f, sb1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        sa0 = __builtin_fmaf(v[r], R.a[r], sa0);
        sb0 = __builtin_fmaf(v[r], R.b[r], sb0);
        sa1 = __builtin_fmaf(v[r + 1], R.a[r + 1], sa1);
        sb1 = __builtin_fmaf(v[r + 1], R.b[r + 1], sb1);
    }
    const float A = sa0 + sa1, Bv = sb0 + sb1;
    return xhalf_sum(__builtin_fmaf(Bv - A, t, A));
}

// y and log(dy + 1e-7) of one I-spline block from its weights v (unnormalised), rS = 1/sum(q), rs = reg * S1.
// Table rows [mesh][nd][h][16] (fk pre-multiplied) and their row sums [mesh][nd] are fetched here, one derivative
// order at a time (32 live registers instead of 64).
__device__ __forceinline__ void ispline_eval(const MfmaDev& mm, const f32x16& v, const Lerp& Lp, int h, float rS, float rs, float& y,
                                             float& logdy) {
    const float* tl = mm.tabI + ((size_t)Lp.il * 4 + h) * 16;
    const float* tr = mm.tabI + ((size_t)Lp.ir * 4 + h) * 16;
    const f32x2 rl = *reinterpret_cast<const f32x2*>(mm.rsI + (size_t)Lp.il * 2);
    const f32x2 rr = *reinterpret_cast<const f32x2*>(mm.rsI + (size_t)Lp.ir * 2);
    float ynum = lerp_dot(v, load_rows(tl, tr), Lp.t);
    float dnum = lerp_dot(v, load_rows(tl + 32, tr + 32), Lp.t);
    ynum = __builtin_fmaf(rs, __builtin_fmaf(rr[0] - rl[0], Lp.t, rl[0]), ynum);
    dnum = __builtin_fmaf(rs, __builtin_fmaf(rr[1] - rl[1], Lp.t, rl[1]), dnum);
    y = ynum * rS;
    logdy = fast_log(__builtin_fmaf(dnum, rS, 1e-7f));
}

// sigmoid weights of one block and their two sums: S1 = sum v, Sf = sum v*fk (over the walker's 32 rows)
__device__ __forceinline__ void sigmoid_block(f32x16& o, const float* fk_lds, int h, float& S1, float& Sf) {
    const f32x16 fk = load16(fk_lds + h * 16);
    float s1 = 0.0f, sf = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = act_sigmoid(o[r]);
        o[r] = v;
        s1 += v;
        sf = __builtin_fmaf(v, fk[r], sf);
    }
    S1 = xhalf_sum(s1);
    Sf = xhalf_sum(sf);
}

#ifdef WF_STAMP
#define STAMP(k)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long t_;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        stamp_acc[k] += t_ - stamp_last;                                                           \
        stamp_last = t_;                                                                           \
    } while (0)
#else
#define STAMP(k)
#endif

template <int D, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void k_mfma(const MfmaDev mm, int mode, const float* __restrict__ xg, int64_t B,
                                                      float* __restrict__ out, float* __restrict__ u_out, int32_t* __restrict__ idx_out) {
    // mm is passed BY VALUE: it lives in the kernarg segment, so its fields are scalar loads and the table pointers
    // are known to be global (with a pointer-to-struct argument hipcc emitted flat_load for every table access).
    using O = NetOff<D>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // ---- prologue 1: stage every net's weight image + constants into LDS (one pass, 16 B per lane)
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(mm.image);
        f32x4* dst = reinterpret_cast<f32x4*>(lds);
        const int n4 = mm.image_floats >> 2;
        for (int i = threadIdx.x; i < n4; i += kWaves * 64) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float* consts = lds + mm.const_off;
    const float* fkI = consts;        // [2][16] remove_bias * keep factors of the flow-layer I-spline
    const float* fkP = consts + 32;   // [2][16] prior: keep (B) or remove_bias * keep (M)
    const float* ob2b = consts + 64;  // [4][64][4] ob_to_b in f32-MFMA A order
    const int64_t n_tiles = (B + 31) >> 5;
    const int idx_stride = (mm.n_layers + 1) * D * 2;
    const float L = mm.box_L, tol = 1e-7f;
#ifdef WF_STAMP
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

    for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < n_tiles; tile += (int64_t)gridDim.x * kWaves) {
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        float cur[D], nxt[D];
#pragma unroll
        for (int d = 0; d < D; ++d) cur[d] = xg[wl * D + d];
        int32_t* idx = (idx_out && valid && h == 0) ? idx_out + w * idx_stride : nullptr;

        // ---- BoxTransformLayer (made.py:118-137, 156-183); IEEE divisions: layer-0 bin indices must be exact
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
            Frag h2[2];
            STAMP(0);
            Lerp Lp[D];
            if (mm.layer_kind == WF_LAYER_IMADE) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    Lp[d] = make_lerp(cur[d], mm.n_mesh);
                    if (idx) { idx[(l * D + d) * 2] = Lp[d].xl; idx[(l * D + d) * 2 + 1] = Lp[d].xr; }
                }
            }
            hidden_layers<D>(net, cur, lane, h2);
            STAMP(1);
            if (mm.layer_kind == WF_LAYER_IMADE) {
                // dimension 0: walker-independent weights -> composite table (k_prepare_dim0)
                {
                    const f32x4 c0 = comp_lerp(mm.comp + (size_t)l * mm.n_mesh, Lp[0]);
                    nxt[0] = c0[0];
                    logdet = logdet + fast_log(c0[1] + 1e-7f);
                }
                STAMP(2);
#pragma unroll
                for (int d = 1; d < D; ++d) {
                    f32x16 v = out_block<D>(net, h2, d, lane);
                    STAMP(3);
                    float S1, Sf;
                    sigmoid_block(v, fkI, h, S1, Sf);
                    STAMP(4);
                    const float rs = mm.i_reg * S1;
                    const float rS = __builtin_amdgcn_rcpf(__builtin_fmaf(rs, mm.F_I, Sf));
                    float ld;
                    ispline_eval(mm, v, Lp[d], h, rS, rs, nxt[d], ld);
                    logdet = logdet + ld;
                    STAMP(5);
                }
            } else {
                // MADE (made.py:21-27): rows 0 / 1 of block d = log_weight / bias (lane half 0, registers 0 / 1)
                float ls = 0.0f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    float lw, bias;
                    if (d == 0) {
                        const f32x4 c0 = mm.comp[(size_t)l * mm.n_mesh];   // {log_weight, bias}: constants
                        lw = c0[0];
                        bias = c0[1];
                    } else {
                        const f32x16 o = out_block<D>(net, h2, d, lane);
                        lw = __shfl(o[0], j);
                        bias = __shfl(o[1], j);
                    }
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
            if (mm.prior_kind == WF_PRIOR_WAVEFLOW || mm.prior_kind == WF_PRIOR_MFLOW) {
                const bool wavefn = mm.prior_kind == WF_PRIOR_WAVEFLOW;
                const float* net = lds + mm.net_off[mm.n_layers];
                const f32x4* comp_p = mm.comp + (size_t)mm.n_layers * mm.n_mesh;
                // the conditioner sees the unclipped u (wavefunctions.py:40), the spline the clipped one (:45)
                float uc[D];
                Lerp Lp[D];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    uc[d] = fminf(fmaxf(cur[d], 0.0f), 1.0f);
                    Lp[d] = make_lerp(uc[d], mm.n_mesh);
                    if (idx) { idx[(mm.n_layers * D + d) * 2] = Lp[d].xl; idx[(mm.n_layers * D + d) * 2 + 1] = Lp[d].xr; }
                }
                auto prior_rows = [&](int d) {   // [mesh][h][16], nd 0
                    return load_rows(mm.tabP + ((size_t)Lp[d].il * 2 + h) * 16, mm.tabP + ((size_t)Lp[d].ir * 2 + h) * 16);
                };
                Frag h2[2];
                hidden_layers<D>(net, cur, lane, h2);
                float lp = 0.0f, prod = 1.0f;
                if (wavefn) {
                    const f32x16 keep = load16(fkP + h * 16);
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        f32x16 c;
                        float rnorm = 1.0f, sgn = 1.0f, num;
                        if (d == 0) {
                            num = comp_lerp(comp_p, Lp[0])[0];   // psi_0 incl. sign and norm
                        } else {
                            f32x16 o = out_block<D>(net, h2, d, lane);
                            float s1 = 0.0f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) { s1 += o[r]; o[r] = o[r] * keep[r]; }
                            s1 = xhalf_sum(s1);
                            // c = (o * keep) @ ob_to_b on v_mfma_f32_32x32x2_f32 (K = 32, unnormalised operands)
                            c = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) {
                                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ob2b + (r4 * 64 + lane) * 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], o[4 * r4 + e], c, 0, 0, 0);
                            }
                            float n2 = 0.0f;
#pragma unroll
                            for (int r = 0; r < 16; ++r) n2 = __builtin_fmaf(c[r], c[r], n2);
                            rnorm = __builtin_amdgcn_rsqf(xhalf_sum(n2));
                            sgn = s1 < 0.0f ? -1.0f : 1.0f;
                            num = lerp_dot(c, prior_rows(d), Lp[d].t);
                        }
                        float v = num * rnorm * sgn;
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
                } else {
                    // MFlow (distributions.py:139-163): M-spline table with the row factors folded in
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        float num, rS = 1.0f;
                        if (d == 0) {
                            num = comp_lerp(comp_p, Lp[0])[0];
                        } else {
                            f32x16 v = out_block<D>(net, h2, d, lane);
                            float S1, Sf;
                            sigmoid_block(v, fkP, h, S1, Sf);
                            rS = __builtin_amdgcn_rcpf(Sf);
                            num = lerp_dot(v, prior_rows(d), Lp[d].t);
                        }
                        lp = lp + fast_log(__builtin_fmaf(num, rS, 1e-7f));
                    }
                    result = lp + logdet;
                }
#pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = uc[d];
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
        STAMP(6);
    }
#ifdef WF_STAMP
    if (mm.dbg && lane == 0) {
        unsigned long long* g = reinterpret_cast<unsigned long long*>(mm.dbg) + ((size_t)blockIdx.x * kWaves + wave) * 8;
        for (int k = 0; k < 8; ++k) g[k] = stamp_acc[k];
    }
#endif
}


// ---- composite tables of output dimension 0 (see comp_lerp); one thread per (net, mesh point); plain weight image
__global__ void k_prepare_dim0(const ModelDev* __restrict__ mdp, int nm, const float* __restrict__ fk_nat /* [2][32]: I, prior */,
                               float F_I, float F_P, f32x4* __restrict__ comp) {
    const ModelDev& md = *mdp;
    const int n_nets = md.n_layers + ((md.prior_kind == WF_PRIOR_WAVEFLOW || md.prior_kind == WF_PRIOR_MFLOW) ? 1 : 0);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (nm <= 0 || gid >= n_nets * nm) return;
    const int n = gid / nm, m = gid % nm;
    const NetPlain& net = md.nets[n];
    const bool is_prior = n == md.n_layers;
    f32x4 out = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!is_prior && md.layer_kind == WF_LAYER_MADE) {
        out[0] = net.b2[0];   // log_weight, dimension 0 (rows j = 0 / 1 of block d = 0)
        out[1] = net.b2[1];
    } else if (is_prior && md.prior_kind == WF_PRIOR_WAVEFLOW) {
        const SplineDev& sp = md.psp;
        const int nb = sp.nb, nbp = sp.nbp;
        const float* keep = fk_nat + 32;
        float s1 = 0.0f;
        for (int j = 0; j < nb; ++j) s1 += net.b2[j];
        float n2 = 0.0f, num = 0.0f;
        for (int i = 0; i < nb; ++i) {
            float c = 0.0f;
            for (int a = 0; a < nb; ++a) c = __builtin_fmaf(net.b2[a] * keep[a], md.ob_to_b[a * nbp + i], c);
            n2 = __builtin_fmaf(c, c, n2);
            num = __builtin_fmaf(c, sp.tab[(size_t)m * nbp + i], num);
        }
        out[0] = (s1 < 0.0f ? -num : num) * __builtin_amdgcn_rsqf(n2);
    } else {
        const SplineDev& sp = is_prior ? md.psp : md.isp;
        const int nb = sp.nb, nbp = sp.nbp;
        const float* fk = fk_nat + (is_prior ? 32 : 0);
        const float reg = is_prior ? 0.0f : md.i_reg, F = is_prior ? F_P : F_I;
        float s1 = 0.0f, sf = 0.0f;
        for (int j = 0; j < nb; ++j) {
            const float v = 1.0f / (1.0f + expf(-net.b2[j]));
            s1 += v;
            sf = __builtin_fmaf(v, fk[j], sf);
        }
        const float rs = reg * s1, rS = 1.0f / __builtin_fmaf(rs, F, sf);
        float y = 0.0f, dy = 0.0f;
        for (int j = 0; j < nb; ++j) {
            const float q = (1.0f / (1.0f + expf(-net.b2[j])) + rs) * fk[j];
            y = __builtin_fmaf(q, sp.tab[(size_t)m * nbp + j], y);
            if (!is_prior) dy = __builtin_fmaf(q, sp.tab[((size_t)sp.n_mesh + m) * nbp + j], dy);
        }
        out[0] = y * rS;
        out[1] = dy * rS;
    }
    comp[gid] = out;
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
    hipLaunchKernelGGL((k_mfma<D, kWaves>), dim3((unsigned)grid), dim3(kWaves * 64), lds_bytes, s, *mdev, mode, x, B, out, u, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

int waves_per_group() {
    const char* e = getenv("WF_MFMA_WAVES");  // tuning knob (read at every launch): 8, 12 or 16 waves per workgroup
    const int v = e ? atoi(e) : 0;
    return (v == 8 || v == 12 || v == 16) ? v : 16;
}

template <int D>
int launch_d(const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, hipStream_t s) {
    switch (waves_per_group()) {
        case 8: return launch_dw<D, 8>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        case 12: return launch_dw<D, 12>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
        default: return launch_dw<D, 16>(mdev, lds_bytes, mode, x, B, out, u, idx, s);
    }
}

}  // namespace

int mfma_extra_lds_floats(int) { return 0; }

int launch_prepare_dim0(const ModelDev* md_dev, int n_nets, int n_mesh, const float* fk_nat_dev, float F_I, float F_P, void* comp_dev,
                        void* stream) {
    const int total = n_nets * n_mesh;
    if (total <= 0) return WF_OK;
    hipLaunchKernelGGL(k_prepare_dim0, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, md_dev, n_mesh, fk_nat_dev, F_I, F_P,
                       reinterpret_cast<f32x4*>(comp_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

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
