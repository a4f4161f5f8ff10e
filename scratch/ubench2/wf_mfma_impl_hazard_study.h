// wf_mfma_impl.h -- the MFMA throughput kernel as templates; instantiated per shape in wf_mfma_inst_*.hip so that the
// shapes compile in parallel.  See wf_kernels_mfma.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "wf_internal.h"

#ifndef WF_XHALF
#define WF_XHALF 0
#endif
#ifdef WF_NO_FENCE
#define WF_FENCE()
#elif defined(WF_FENCE_MASK)   // experiment: which instruction classes may cross (see __builtin_amdgcn_sched_barrier)
#define WF_FENCE() __builtin_amdgcn_sched_barrier(WF_FENCE_MASK)
#else
#define WF_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

namespace wf {
namespace mfma {

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
#if WF_XHALF == 1   // experiment: the builtin (hipcc pads the VALU-write -> permlane-read side itself)
    const unsigned u = __float_as_uint(v);
    const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
#elif WF_XHALF == 2  // experiment: LDS crossbar
    return v + __uint_as_float(__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ 32) << 2), (int)__float_as_uint(v)));
#elif WF_XHALF == 3  // experiment: asm swap with the documented 2 wait states in front only
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
#else
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3" : "+v"(a), "+v"(b));
    return a + b;
#endif
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
template <int SITE = 0>
__device__ __forceinline__ f32x16 dense64_block(const _Float16* Wh, const _Float16* Wl, const Frag (&in)[2], f32x16 bias, int lane) {
    f32x16 acc = bias;
#ifdef WF_OPEN_SITES   // experiment: the sites in this bit mask get the weak fence WF_FENCE_MASK, every other site the full one
#undef WF_FENCE
#define WF_FENCE() do { if ((WF_OPEN_SITES >> SITE) & 1) __builtin_amdgcn_sched_barrier(0x402); else __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
    // Scheduling fences around the f16 MFMA chain: when hipcc (ROCm 7.2) interleaved unrelated VALU / memory
    // instructions of the neighbouring code into this chain, a few tiles per launch came out wrong,
    // non-deterministically.  With the chain fenced the kernel is bit-reproducible; cost < 1 %.
    WF_FENCE();
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
    WF_FENCE();
    return acc;
}

// float offsets inside a net image (wf_model.cpp: build_mfma_image); NBK = 32-row blocks per dimension (1 or 2)
template <int D, int NBK>
struct NetOff {
    static constexpr int S0 = (D + 1) / 2;
    static constexpr int W0 = 0;
    static constexpr int b0 = W0 + 2 * S0 * 64;
    static constexpr int W1h = b0 + 64;
    static constexpr int W1l = W1h + 2048;
    static constexpr int b1 = W1l + 2048;
    static constexpr int W2h = b1 + 64;
    static constexpr int W2l = W2h + (D - 1) * NBK * 1024;
    static constexpr int b2 = W2l + (D - 1) * NBK * 1024;
    static constexpr int total = b2 + 32 * D * NBK;
};

// Hidden layers of one conditioner net for the wave's 32 walkers; result: second hidden layer as B fragments.
template <int D, int NBK, int PRIOR = 0>
__device__ __forceinline__ void hidden_layers(const float* net, const float (&in)[D], int lane, Frag (&h2)[2]) {
    using O = NetOff<D, NBK>;
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
    {
        f32x16 a = dense64_block<PRIOR * 3 + 0>(W1h, W1l, h1, load16(net + O::b1 + h * 16), lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = act_tanh(a[r]);
        split_block(a, h2[0]);
    }
    {
        f32x16 a = dense64_block<PRIOR * 3 + 1>(W1h + 2048, W1l + 2048, h1, load16(net + O::b1 + (2 + h) * 16), lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = act_tanh(a[r]);
        split_block(a, h2[1]);
    }
}

// Output block (dimension d >= 1, row block kb): raw (scaled) outputs o[basis row][walker] in accumulator layout.
template <int D, int NBK, int PRIOR = 0>
__device__ __forceinline__ f32x16 out_block(const float* net, const Frag (&h2)[2], int d, int kb, int lane) {
    using O = NetOff<D, NBK>;
    const int h = lane >> 5;
    const _Float16* W2h = reinterpret_cast<const _Float16*>(net + O::W2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(net + O::W2l);
    const int blk = (d - 1) * NBK + kb;
    return dense64_block<PRIOR * 3 + 2>(W2h + blk * 2048, W2l + blk * 2048, h2, load16(net + O::b2 + ((d * NBK + kb) * 2 + h) * 16), lane);
}

// The 16 table values of this lane half at x_l (a) and x_r (b) for one derivative order.
// (The rows depend on the layer input only, but requesting them before the conditioner MFMAs costs 64 live VGPRs,
// i.e. a wave per SIMD, and measured slower every time it was tried: 8 / 12 / 16 waves per workgroup 0.385 / 0.404 / 0.509 ms
// against 0.34 ms without: the kernel is bound by issue and by register-limited occupancy, not by this latency.)
struct Rows {
    f32x16 a, b;
};
__device__ __forceinline__ Rows load_rows(const float* __restrict__ tl, const float* __restrict__ tr) {
    Rows r;
    r.a = load16(tl);
    r.b = load16(tr);
    return r;
}

// sum_r v_r * lerp(T'_r): this lane's 16 rows of one block, NOT yet summed over the lane halves
__device__ __forceinline__ float lerp_dot_part(const f32x16& v, const Rows& R, float t) {
    float sa0 = 0.0f, sa1 = 0.0f, sb0 = 0.0f, sb1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        sa0 = __builtin_fmaf(v[r], R.a[r], sa0);
        sb0 = __builtin_fmaf(v[r], R.b[r], sb0);
        sa1 = __builtin_fmaf(v[r + 1], R.a[r + 1], sa1);
        sb1 = __builtin_fmaf(v[r + 1], R.b[r + 1], sb1);
    }
    const float A = sa0 + sa1, Bv = sb0 + sb1;
    return __builtin_fmaf(Bv - A, t, A);
}

// sum over the walker's 32 * NBK rows; base_l / base_r: row (mesh point, order) of this lane half, blocks 32 floats apart
template <int NBK>
__device__ __forceinline__ float lerp_dot(const f32x16 (&v)[NBK], const float* __restrict__ base_l, const float* __restrict__ base_r, float t) {
    float part = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) part += lerp_dot_part(v[kb], load_rows(base_l + kb * 32, base_r + kb * 32), t);
    return xhalf_sum(part);
}

// y and log(dy + 1e-7) of one I-spline block from its weights v (unnormalised), rS = 1/sum(q), rs = reg * S1.
// Table rows [mesh][nd][kb][h][16] (fk pre-multiplied) and their row sums [mesh][nd] are fetched here, one derivative
// order at a time.
template <int NBK>
__device__ __forceinline__ void ispline_eval(const MfmaDev& mm, const f32x16 (&v)[NBK], const Lerp& Lp, int h, float rS, float rs, float& y,
                                             float& logdy) {
    const float* tl = mm.tabI + (size_t)Lp.il * (64 * NBK) + h * 16;
    const float* tr = mm.tabI + (size_t)Lp.ir * (64 * NBK) + h * 16;
    const f32x2 rl = *reinterpret_cast<const f32x2*>(mm.rsI + (size_t)Lp.il * 2);
    const f32x2 rr = *reinterpret_cast<const f32x2*>(mm.rsI + (size_t)Lp.ir * 2);
    float ynum = lerp_dot<NBK>(v, tl, tr, Lp.t);
    float dnum = lerp_dot<NBK>(v, tl + 32 * NBK, tr + 32 * NBK, Lp.t);
    ynum = __builtin_fmaf(rs, __builtin_fmaf(rr[0] - rl[0], Lp.t, rl[0]), ynum);
    dnum = __builtin_fmaf(rs, __builtin_fmaf(rr[1] - rl[1], Lp.t, rl[1]), dnum);
    y = ynum * rS;
    logdy = fast_log(__builtin_fmaf(dnum, rS, 1e-7f));
}

// sigmoid weights of one dimension (NBK blocks) and their two sums: S1 = sum v, Sf = sum v*fk
template <int NBK>
__device__ __forceinline__ void sigmoid_block(f32x16 (&o)[NBK], const float* fk_lds /* [NBK][2][16] */, int h, float& S1, float& Sf) {
    float s1 = 0.0f, sf = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
        const f32x16 fk = load16(fk_lds + (kb * 2 + h) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = act_sigmoid(o[kb][r]);
            o[kb][r] = v;
            s1 += v;
            sf = __builtin_fmaf(v, fk[r], sf);
        }
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

// cooperative copy of n_floats (multiple of 4) global -> LDS, 16 B per lane
template <int kThreads>
__device__ __forceinline__ void stage_floats(const float* __restrict__ src, float* dst, int n_floats) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
    const int n4 = n_floats >> 2;
    // batches of 8 independent loads per lane (measured: no effect on the kernel's ~38 us small-batch floor, which is one
    // tile's chain of dependent table loads, MFMA and transcendental latencies; small batches take the wave kernel instead)
    for (int base = threadIdx.x; base < n4; base += kThreads * 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads;
            if (i < n4) v[u] = s4[i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * kThreads;
            if (i < n4) d4[i] = v[u];
        }
    }
}

template <int D, int NBK, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void k_mfma(const MfmaDev mm, int mode, const float* __restrict__ xg, int64_t B,
                                                      float* __restrict__ out, float* __restrict__ u_out, int32_t* __restrict__ idx_out) {
    // mm is passed BY VALUE: it lives in the kernarg segment, so its fields are scalar loads and the table pointers
    // are known to be global (with a pointer-to-struct argument hipcc emitted flat_load for every table access).
    // LDS: [constants][net slot(s)].  Resident mode: every net has its own slot, staged once.  Staged mode (the nets do
    // not fit together, e.g. D >= 4 with 3 layers): ONE slot; every workgroup walks its chunk of kWaves tiles through the
    // nets, re-staging the slot between two barriers per net (the per-tile state is D + 1 registers).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kThreads = kWaves * 64;
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    if (!mm.staged) stage_floats<kThreads>(mm.image, lds + mm.const_floats, mm.net_floats * mm.n_nets);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float* fkI = lds;                        // [NBK][2][16] remove_bias * keep factors of the flow-layer I-spline
    const float* fkP = lds + 32 * NBK;             // [NBK][2][16] prior: keep (B) or remove_bias * keep (M)
    const float* ob2b = lds + 64 * NBK;            // [NBK out][NBK in][4][64][4] ob_to_b in f32-MFMA A order
    float* slots = lds + mm.const_floats;
    const int64_t n_tiles = (B + 31) >> 5;
    const int64_t n_chunks = (n_tiles + kWaves - 1) / kWaves;
    const int idx_stride = (mm.n_layers + 1) * D * 2;
    const float L = mm.box_L, tol = 1e-7f;
#ifdef WF_STAMP
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif

    for (int64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int64_t tile = chunk * kWaves + wave;    // may be >= n_tiles in the last chunk: computed, never stored
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
            const float* net = slots + (mm.staged ? 0 : l * mm.net_floats);
            if (mm.staged) {
                __syncthreads();   // every wave is done with the previous occupant of the slot
                stage_floats<kThreads>(mm.image + (size_t)l * mm.net_floats, slots, mm.net_floats);
                __syncthreads();
            }
            Frag h2[2];
            STAMP(0);
            hidden_layers<D, NBK>(net, cur, lane, h2);
            STAMP(1);
            if (mm.layer_kind == WF_LAYER_IMADE) {
                // dimension 0: walker-independent weights -> composite table (k_prepare_dim0)
                {
                    const Lerp Lp = make_lerp(cur[0], mm.n_mesh);
                    if (idx) { idx[(l * D) * 2] = Lp.xl; idx[(l * D) * 2 + 1] = Lp.xr; }
                    const f32x4 c0 = comp_lerp(mm.comp + (size_t)l * mm.n_mesh, Lp);
                    nxt[0] = c0[0];
                    logdet = logdet + fast_log(c0[1] + 1e-7f);
                }
                STAMP(2);
#pragma unroll
                for (int d = 1; d < D; ++d) {
                    f32x16 v[NBK];
#pragma unroll
                    for (int kb = 0; kb < NBK; ++kb) v[kb] = out_block<D, NBK>(net, h2, d, kb, lane);
                    STAMP(3);
                    float S1, Sf;
                    sigmoid_block<NBK>(v, fkI, h, S1, Sf);
                    STAMP(4);
                    const float rs = mm.i_reg * S1;
                    const float rS = __builtin_amdgcn_rcpf(__builtin_fmaf(rs, mm.F_I, Sf));
                    const Lerp Lp = make_lerp(cur[d], mm.n_mesh);
                    if (idx) { idx[(l * D + d) * 2] = Lp.xl; idx[(l * D + d) * 2 + 1] = Lp.xr; }
                    float ld;
                    ispline_eval<NBK>(mm, v, Lp, h, rS, rs, nxt[d], ld);
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
                        const f32x16 o = out_block<D, NBK>(net, h2, d, 0, lane);
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
                const float* net = slots + (mm.staged ? 0 : mm.n_layers * mm.net_floats);
                if (mm.staged) {
                    __syncthreads();
                    stage_floats<kThreads>(mm.image + (size_t)mm.n_layers * mm.net_floats, slots, mm.net_floats);
                    __syncthreads();
                }
                const f32x4* comp_p = mm.comp + (size_t)mm.n_layers * mm.n_mesh;
                Frag h2[2];
                hidden_layers<D, NBK, 1>(net, cur, lane, h2);   // the conditioner sees the unclipped u (wavefunctions.py:40)
                float lp = 0.0f, prod = 1.0f;
#ifdef WF_DBG_PRIOR
                float dbg_a = 0.0f, dbg_b = 0.0f;
#endif
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float uc = fminf(fmaxf(cur[d], 0.0f), 1.0f);   // the spline sees the clipped one (:45)
                    const Lerp Lp = make_lerp(uc, mm.n_mesh);
                    if (idx) { idx[(mm.n_layers * D + d) * 2] = Lp.xl; idx[(mm.n_layers * D + d) * 2 + 1] = Lp.xr; }
                    const float* tl = mm.tabP + (size_t)Lp.il * (32 * NBK) + h * 16;   // [mesh][kb][h][16], nd 0
                    const float* tr = mm.tabP + (size_t)Lp.ir * (32 * NBK) + h * 16;
                    float val;   // psi_d (B prior) or the density factor (M prior)
                    if (d == 0) {
                        val = comp_lerp(comp_p, Lp)[0];
                    } else if (wavefn) {
                        f32x16 o[NBK];
                        float s1 = 0.0f;
#pragma unroll
                        for (int kb = 0; kb < NBK; ++kb) {
                            o[kb] = out_block<D, NBK, 1>(net, h2, d, kb, lane);
                            const f32x16 keep = load16(fkP + (kb * 2 + h) * 16);
#pragma unroll
                            for (int r = 0; r < 16; ++r) { s1 += o[kb][r]; o[kb][r] = o[kb][r] * keep[r]; }
                        }
                        s1 = xhalf_sum(s1);
                        // c = (o * keep) @ ob_to_b on v_mfma_f32_32x32x2_f32 (unnormalised operands: no fp16 split)
                        f32x16 c[NBK];
                        float n2 = 0.0f;
#if defined(WF_FENCE_OB2B) && (WF_FENCE_OB2B & 1)
                        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                        for (int ko = 0; ko < NBK; ++ko) {
                            c[ko] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                            for (int ki = 0; ki < NBK; ++ki)
#pragma unroll
                                for (int r4 = 0; r4 < 4; ++r4) {
                                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(ob2b + (((ko * NBK + ki) * 4 + r4) * 64 + lane) * 4);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) c[ko] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], o[ki][4 * r4 + e], c[ko], 0, 0, 0);
                                }
#if defined(WF_FENCE_OB2B) && (WF_FENCE_OB2B & 2)
                            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
                            for (int r = 0; r < 16; ++r) n2 = __builtin_fmaf(c[ko][r], c[ko][r], n2);
                        }
                        const float n2s = xhalf_sum(n2);
                        const float rnorm = __builtin_amdgcn_rsqf(n2s);
#if defined(WF_DBG_PRIOR) && WF_DBG_PRIOR >= 5
                        // taps inside the numerator (NBK = 1 only)
                        const Rows Rw = load_rows(tl, tr);
                        float sa = 0.0f, sb = 0.0f;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { sa = __builtin_fmaf(c[0][r], Rw.a[r], sa); sb = __builtin_fmaf(c[0][r], Rw.b[r], sb); }
                        const float part = __builtin_fmaf(sb - sa, Lp.t, sa);
                        const float numer = xhalf_sum(part);
                        if (WF_DBG_PRIOR == 5) { dbg_a = part; dbg_b = Lp.t; }
                        if (WF_DBG_PRIOR == 6) { dbg_a = sa; dbg_b = sb; }
                        if (WF_DBG_PRIOR == 7) { dbg_a = (float)Lp.il; dbg_b = (float)Lp.ir; }
                        if (WF_DBG_PRIOR == 8) { dbg_a = numer - part; dbg_b = Rw.a[0] + Rw.b[15]; }
#else
                        const float numer = lerp_dot<NBK>(c, tl, tr, Lp.t);
#endif
                        val = numer * rnorm;
                        val = s1 < 0.0f ? -val : val;
#ifdef WF_DBG_PRIOR
                        if (WF_DBG_PRIOR == 1) { dbg_a = s1; dbg_b = n2s; }
                        if (WF_DBG_PRIOR == 2) { dbg_a = numer; dbg_b = c[0][0]; }
                        if (WF_DBG_PRIOR == 3) { dbg_a = o[0][0]; dbg_b = o[0][15]; }
                        if (WF_DBG_PRIOR == 4) { dbg_a = c[0][15]; dbg_b = c[0][7]; }
#endif
                    } else {
                        // MFlow (distributions.py:139-163): M-spline table with the row factors folded in
                        f32x16 v[NBK];
#pragma unroll
                        for (int kb = 0; kb < NBK; ++kb) v[kb] = out_block<D, NBK>(net, h2, d, kb, lane);
                        float S1, Sf;
                        sigmoid_block<NBK>(v, fkP, h, S1, Sf);
                        val = lerp_dot<NBK>(v, tl, tr, Lp.t) * __builtin_amdgcn_rcpf(Sf);
                    }
                    if (wavefn) {
                        const bool constrained = (mm.constrained_mask >> d) & 1u;
                        if (mode == 0) {
                            float pr = val * val;
                            if (constrained) pr = pr * 0.5f;
                            lp = lp + fast_log(pr + 1e-7f);
                        } else {
                            if (constrained) val = val * 0.70710678118654752f;
                            prod = prod * val;
                        }
                    } else {
                        lp = lp + fast_log(val + 1e-7f);
                    }
                    nxt[d] = uc;
                }
                result = (wavefn && mode != 0) ? prod * __expf(0.5f * logdet) : lp + logdet;
#pragma unroll
                for (int d = 0; d < D; ++d) cur[d] = nxt[d];
#ifdef WF_DBG_PRIOR
                cur[0] = dbg_a;
                cur[D - 1] = dbg_b;
#endif
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

template <int D, int NBK, int kWaves>
int launch_dw(const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, hipStream_t s) {
    static int configured_bytes = -1;
    if (lds_bytes > configured_bytes) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma<D, NBK, kWaves>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           lds_bytes);
        if (e != hipSuccess) {
            set_hip_error((int)e);
            return WF_ERR_HIP;
        }
        configured_bytes = lds_bytes;
    }
    const int64_t n_tiles = (B + 31) / 32;
    int64_t grid = (n_tiles + kWaves - 1) / kWaves;
    if (grid > 256) grid = 256;  // one persistent workgroup per CU
    hipLaunchKernelGGL((k_mfma<D, NBK, kWaves>), dim3((unsigned)grid), dim3(kWaves * 64), lds_bytes, s, *mdev, mode, x, B, out, u, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}


}  // namespace mfma
}  // namespace wf
