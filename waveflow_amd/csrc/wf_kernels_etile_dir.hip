// wf_kernels_etile_dir.hip -- H psi of large batches BEYOND two particles on the matrix cores (gfx950): one coordinate direction at a time.
//
// physics.laplacian (physics.py:50-52) is the trace of the Hessian of psi: sum_i d^2 psi / d x_i^2.  Along ONE coordinate direction every intermediate
// of the model is a function of a scalar parameter tau (x + tau e_i), and its truncated Taylor triple (f, f', f'') travels through the conditioner
// exactly like the u_0-triples of the two-particle kernels (wf_kernels_etile.hip): three channels of the same split-fp16 MFMA products, the
// activation's chain rule on the VALU (act_block), one power of two per (walker, channel) around every product.  With D particles the conditioner has
// D - 1 live inputs (model_factory.py:8-19), so the (1 + (D-1) + (D-1)D/2)-channel Taylor jets in its inputs that a one-pass form would need do not
// fit the register file; D passes of three channels do, at D times the work of a value evaluation -- and the heads keep the separable
// row sums of the two-particle kernel: behind the conditioner a head sums v_j(tau) * T_j(u_d(tau)) over its rows, a function F(s, t) of the direct
// parameter s = tau (through the weights) and of t = u_d(tau) (through the table rows), whose partials are the SAME sums sum_j v_j^(a) g_j T_j^(k) the
// T2 algebra accumulates; the total derivatives along tau follow from the jets of s = (tau: 0, 1, 0) and t = u_d once per walker (t2jet).  A direction's
// jet is stored as J{value, d/dtau, 0, (d^2/dtau^2) / 2}: the product rule of J (x.v y.h + y.v x.h + x.a y.a) is then that of half second derivatives.
//
// One launch per net (k_edir<D, PRIOR>; the net's image is staged into LDS once per workgroup: nets of D >= 4 do not fit LDS together), work items =
// (tile of 32 walkers, direction); the (u_0 .. u_{D-1}, log det) jets of every (direction, walker) wait in HBM between launches: 12 (D + 1) bytes per
// direction.  The prior's launch loops over the directions of its tile, sums the second derivatives into the Laplacian and writes H psi.
// Same function as k_wave_fwd<D, RF<D>> + k_energy_out (same derivative rule of the table lerp: order nd -> table nd + 1), checked against it and
// against the torch oracle (tests/test_gpu_energy.py).  Coverage: D = 3 .. 8, <= 32 bases, mean-type box, IMADE layers, Waveflow prior, ungated
// heads, homogeneous boundary dictionaries (the tables carry the map); everything else stays on the wave kernel.
#include "wf_etile_common.h"

#pragma clang fp contract(fast)

namespace wf {

namespace {
constexpr int kDirWaves = 8;   // one workgroup per CU, two waves per SIMD (256 registers)

__device__ __forceinline__ float chan(const J& u, int c) { return c == 0 ? u.v : (c == 1 ? u.a : 2.0f * u.h); }   // (f, f', f'') of a direction's jet

// state of one (direction, walker): slot 0 .. D-1 = u_d, slot D = log det; st[((dir * (D + 1) + slot) * 3 + c) * B + w], c = (v, a, h)
template <int D>
__device__ __forceinline__ J dst_load(const float* __restrict__ st, int dir, int slot, int64_t B, int64_t w) {
    const float* p = st + ((size_t)(dir * (D + 1) + slot) * 3) * B + w;
    return J{p[0], p[B], 0.0f, p[2 * B]};
}
template <int D>
__device__ __forceinline__ void dst_store(float* __restrict__ st, int dir, int slot, int64_t B, int64_t w, J x) {
    float* p = st + ((size_t)(dir * (D + 1) + slot) * 3) * B + w;
    p[0] = x.v; p[B] = x.a; p[2 * B] = x.h;
}

// BoxTransformLayer, mean type, D particles (made.py:156-183) as jets along direction dir: x_i -> (x_i, delta_{i, dir}, 0)
template <int D>
__global__ void k_edir_box(const float* __restrict__ xg, int64_t B, float L, float* __restrict__ st) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int dir = blockIdx.y;
    if (b >= B) return;
    const float tol = 1e-7f;
    J x[D];
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = J{xg[b * D + d], d == dir ? 1.0f : 0.0f, 0.0f, 0.0f};
    J sum = jc(0.0f);
#pragma unroll
    for (int d = 0; d < D; ++d) sum = sum + x[d];
    const J mean = sum * (1.0f / (float)D);
    const J l = mean - x[0], wd = x[D - 1] - x[0];
    J ld = jc(0.0f), space = jc(2 * L);
#pragma unroll
    for (int i = 0; i < D - 1; ++i) {
        const J diff = x[i + 1] - x[i];
        dst_store<D>(st, dir, i, B, b, diff * jrcp(space + tol));
        ld = ld - jlog(space + tol);
        space = space - diff;
    }
    const J den = (jc(2 * L) - wd) + tol;
    dst_store<D>(st, dir, D - 1, B, b, ((mean + L) - l) * jrcp(den));
    ld = ld - jlog(den);
    dst_store<D>(st, dir, D, B, b, ld);
}

// the two hidden layers of one conditioner for a tile whose inputs are the direction's jets u[0 .. D-1] -> fragments of the second hidden layer's
// activation triples (as cond_hidden of wf_kernels_etile.hip, with D inputs: (D + 1) / 2 K steps of the f32 input layer per channel)
template <int D>
__device__ __forceinline__ void dir_hidden(const float* net, const J (&u)[D], int lane, Frag (&f)[NCH][2], int (&e)[NCH]) {
    using O = NetOff<D, 1>;
    const int h = lane >> 5;
    f32x16 a0[NCH], a1[NCH];
    init_acc(a0, net + O::b0 + (0 * 2 + h) * 16);
    init_acc(a1, net + O::b0 + (1 * 2 + h) * 16);
#pragma unroll
    for (int s = 0; s < O::S0; ++s) {
        const float w0 = net[O::W0 + (0 * O::S0 + s) * 64 + lane], w1 = net[O::W0 + (1 * O::S0 + s) * 64 + lane];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float lo = chan(u[2 * s], c);
            const float hi = (2 * s + 1 < D) ? chan(u[(2 * s + 1 < D) ? 2 * s + 1 : D - 1], c) : 0.0f;
            a0[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, h ? hi : lo, a0[c], 0, 0, 0);
            a1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, h ? hi : lo, a1[c], 0, 0, 0);
        }
    }
    act_block(a0);
    act_block(a1);
    to_frags(a0, a1, f, e);
    const _Float16* W1h = reinterpret_cast<const _Float16*>(net + O::W1h);
    const _Float16* W1l = reinterpret_cast<const _Float16*>(net + O::W1l);
    init_acc(a0, net + O::b1 + (0 + h) * 16);
    init_acc(a1, net + O::b1 + (2 + h) * 16);
    dense64_block<NCH>(W1h, W1l, f, a0, lane);
    dense64_block<NCH>(W1h + 2048, W1l + 2048, f, a1, lane);
    unscale(a0, e);
    unscale(a1, e);
    act_block(a0);
    act_block(a1);
    to_frags(a0, a1, f, e);
}
// head pre-activation triples of output dimension d >= 1 (one 32-row block)
template <int D>
__device__ __forceinline__ void dir_out(const float* net, const Frag (&f)[NCH][2], const int (&e)[NCH], int d, int lane, f32x16 (&o)[NCH]) {
    using O = NetOff<D, 1>;
    const int h = lane >> 5;
    const _Float16* W2h = reinterpret_cast<const _Float16*>(net + O::W2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(net + O::W2l);
    init_acc(o, net + O::b2 + (d * 2 + h) * 16);
    dense64_block<NCH>(W2h + (d - 1) * 2048, W2l + (d - 1) * 2048, f, o, lane);
    unscale(o, e);
}
__device__ __forceinline__ J comp_jet(const float4_t* __restrict__ comp, const J& u, int n_mesh, J* dlog /* may be null: += log(d/du + 1e-7) */) {
    const LerpN L0 = nlerp(u.v, n_mesh);
    const float4_t ca = comp[L0.il], cb = comp[L0.ir];
    const float t0 = __builtin_fmaf(cb.x - ca.x, L0.t, ca.x), t1 = __builtin_fmaf(cb.y - ca.y, L0.t, ca.y);
    const float t2 = __builtin_fmaf(cb.z - ca.z, L0.t, ca.z), t3 = __builtin_fmaf(cb.w - ca.w, L0.t, ca.w);
    if (dlog) *dlog = *dlog + jlog(jlift(t1, t2, t3, u) + 1e-7f);
    return jlift(t0, t1, t2, u);
}

// One net of the model for every (tile, direction).  Flow nets (PRIOR = false): IMADE layer + Reverse (made.py:66-81, bijections.py:337-340), the
// jets go back to st.  Prior (PRIOR = true): Waveflow prior (wavefunctions.py:54-71) for every direction of the tile, Laplacian = sum of the second
// derivatives, H psi (physics.py:60-93).
template <int D, bool PRIOR>
__global__ __launch_bounds__(kDirWaves * 64) void k_edir(const MfmaDev mm, int net_index, const float* __restrict__ tabI, const float* __restrict__ tabP,
                                                         float* __restrict__ st, int64_t B, const float* __restrict__ xg, const Protons pr,
                                                         float* __restrict__ hpsi, float* __restrict__ psi_out, float* __restrict__ lap_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int next_item;
    __shared__ int bnd_s[32];   // support bounds of the table chunks: [I: 8][lo, hi], [prior: 8][lo, hi]
    constexpr int kThreads = kDirWaves * 64;
    if (threadIdx.x == 0) next_item = 0;
    if (threadIdx.x < 16) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabI + (size_t)mm.n_mesh * 128)[threadIdx.x];
    else if (threadIdx.x < 32) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabP + (size_t)mm.n_mesh * 128)[threadIdx.x - 16];
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    stage_floats<kThreads>(mm.image + (size_t)net_index * mm.net_floats, lds + mm.const_floats, mm.net_floats);
    __syncthreads();
    const float* net = lds + mm.const_floats;
    const float* fkI = lds;
    const float* fkP = lds + 32;
    const _Float16* obh = reinterpret_cast<const _Float16*>(lds + 64);
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int n_mesh = mm.n_mesh;
    const int64_t n_tiles = (B + 31) >> 5;
    const int64_t n_items = PRIOR ? n_tiles : n_tiles * D;
    const int64_t my_items = n_items > (int64_t)blockIdx.x ? (n_items - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const J sJ = J{0.0f, 1.0f, 0.0f, 0.0f};   // the direct parameter of the direction: tau itself
    for (;;) {
        int q_ = 0;
        if (lane == 0) q_ = __hip_atomic_fetch_add(&next_item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q_ = __builtin_amdgcn_readfirstlane(q_);
        if (q_ >= my_items) break;
        const int64_t item = (int64_t)blockIdx.x + (int64_t)q_ * gridDim.x;
        const int64_t tile = PRIOR ? item : item / D;
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        if (!PRIOR) {
            const int dir = (int)(item - tile * D);
            J u[D], ld = dst_load<D>(st, dir, D, B, wl);
#pragma unroll
            for (int d = 0; d < D; ++d) u[d] = dst_load<D>(st, dir, d, B, wl);
            Frag f[NCH][2];
            int e[NCH];
            dir_hidden<D>(net, u, lane, f, e);
            J y[D];
            y[0] = comp_jet(mm.comp + (size_t)net_index * n_mesh, u[0], n_mesh, &ld);   // dimension 0: composite table of the net, all four orders
            const f32x16 g16 = load16(fkI + h * 16);
            // (a loop, not unrolled: the fragments f -- 96 registers -- are live across every dimension; unrolled, the dimensions' table rows and sums
            // pile up on top of them: 304 spilled registers at D = 8.  The jets u, y are then indexed at run time and live in 12 D bytes of scratch.)
#pragma unroll 1
            for (int d = 1; d < D; ++d) {
                f32x16 o[NCH];
                dir_out<D>(net, f, e, d, lane, o);
                const LerpN L = nlerp(u[d].v, n_mesh);
                FlowSums a = {};
                flow_rows(a, o, g16, tabI, 128, bnd_s, L, 0, h);
#pragma unroll
                for (int k = 0; k < 3; ++k) { a.S[k] = xhalf_sum(a.S[k]); a.Qv[k] = xhalf_sum(a.Qv[k]); a.V1[k] = xhalf_sum(a.V1[k]); }
#pragma unroll
                for (int k = 0; k < 4; ++k) { a.R[k] = xhalf_sum(a.R[k]); a.V0[k] = xhalf_sum(a.V0[k]); }
#pragma unroll
                for (int k = 0; k < 2; ++k) a.V2[k] = xhalf_sum(a.V2[k]);
                flow_head_finish(a.S, a.Qv, a.R, mm.F_I, a.V0, a.V1, a.V2, mm.i_reg, sJ, u[d], y[d], ld);
            }
            if (valid && h == 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) dst_store<D>(st, dir, d, B, w, y[D - 1 - d]);   // Reverse
                dst_store<D>(st, dir, D, B, w, ld);
            }
        } else {
            float lap = 0.0f, psiv = 0.0f;
            for (int dir = 0; dir < D; ++dir) {
                J u[D];
                const J ld = dst_load<D>(st, dir, D, B, wl);
#pragma unroll
                for (int d = 0; d < D; ++d) u[d] = dst_load<D>(st, dir, d, B, wl);
                Frag f[NCH][2];
                int e[NCH];
                dir_hidden<D>(net, u, lane, f, e);      // (the conditioner sees the unclipped u, wavefunctions.py:40)
                J psi = jexp_half(ld);
#pragma unroll 1
                for (int d = 0; d < D; ++d) {
                    const J uc = (u[d].v < 0.0f) ? jc(0.0f) : (u[d].v > 1.0f ? jc(1.0f) : u[d]);   // the spline sees the clipped coordinate (:45)
                    J val;
                    if (d == 0) {
                        val = comp_jet(mm.comp + (size_t)net_index * n_mesh, uc, n_mesh, nullptr);
                    } else {
                        f32x16 o[1][NCH];
                        dir_out<D>(net, f, e, d, lane, o[0]);
                        float s1 = 0.0f;
                        Frag of[1][NCH];
                        int eo[NCH];
                        prior_frags<1>(o, fkP, lane, of, eo, s1, nullptr);
                        f32x16 cblk[NCH];
                        prior_c_block<1>(obh, of, eo, 0, lane, cblk);
                        const LerpN L = nlerp(uc.v, n_mesh);
                        PriorSums a = {};
                        prior_rows(a, cblk, tabP, 128, bnd_s + 16, L, 0, h);
#pragma unroll
                        for (int k = 0; k < 3; ++k) a.D0[k] = xhalf_sum(a.D0[k]);
                        a.D1[0] = xhalf_sum(a.D1[0]); a.D1[1] = xhalf_sum(a.D1[1]); a.D2 = xhalf_sum(a.D2);
                        a.cc = xhalf_sum(a.cc); a.cc1 = xhalf_sum(a.cc1); a.c1c1 = xhalf_sum(a.c1c1); a.cc2 = xhalf_sum(a.cc2);
                        const float sgn = s1 < 0.0f ? -1.0f : 1.0f;
                        const T2 N2 = T2{a.cc, 2.0f * a.cc1, 0.0f, 2.0f * (a.c1c1 + a.cc2), 0.0f, 0.0f};
                        const T2 dotp = T2{a.D0[0], a.D1[0], a.D0[1], a.D2, a.D1[1], a.D0[2]};
                        val = t2jet(dotp * t2rsqrt(N2), sJ, uc) * sgn;
                    }
                    const float sc = ((mm.constrained_mask >> d) & 1u) ? 0.70710678118654752f : 1.0f;
                    psi = psi * (val * sc);
                }
                lap += 2.0f * psi.h;
                psiv = psi.v;
            }
            if (valid && h == 0) {
                float V = 0.0f;   // physics.py:60-76: soft-Coulomb, one space dimension
                for (int p = 0; p < pr.n; ++p)
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const float r = pr.pos[p] - xg[w * D + d];
                        V -= 1.0f / sqrtf(1.0f + r * r);
                    }
#pragma unroll
                for (int i = 0; i < D; ++i)
#pragma unroll
                    for (int k = 0; k < i; ++k) {
                        const float r = xg[w * D + i] - xg[w * D + k];
                        V += 1.0f / sqrtf(1.0f + r * r);
                    }
                hpsi[w] = -0.5f * lap + V * psiv;
                if (psi_out) psi_out[w] = psiv;
                if (lap_out) lap_out[w] = lap;
            }
        }
    }
}

template <int D>
int launch_dir_t(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* x, int64_t B, const Protons& pr, float* hpsi,
                 float* psi, float* lap, float* st, hipStream_t s) {
    const int lds_bytes = (mdev->const_floats + mdev->net_floats) * (int)sizeof(float);
    static DynLdsSlots cfg_f{}, cfg_p{};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_edir<D, false>), lds_bytes, &cfg_f)) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_edir<D, true>), lds_bytes, &cfg_p)) return rc;
    const int64_t n_tiles = (B + 31) / 32;
    hipLaunchKernelGGL(k_edir_box<D>, dim3((unsigned)((B + 255) / 256), D), dim3(256), 0, s, x, B, md.box_L, st);
    const unsigned gf = (unsigned)std::min<int64_t>((n_tiles * D + kDirWaves - 1) / kDirWaves, 256);
    const unsigned gp = (unsigned)std::min<int64_t>((n_tiles + kDirWaves - 1) / kDirWaves, 256);
    for (int l = 0; l < md.n_layers; ++l)
        hipLaunchKernelGGL((k_edir<D, false>), dim3(gf), dim3(kDirWaves * 64), lds_bytes, s, *mdev, l, tabI4, tabP4, st, B, x, pr, hpsi, psi, lap);
    hipLaunchKernelGGL((k_edir<D, true>), dim3(gp), dim3(kDirWaves * 64), lds_bytes, s, *mdev, md.n_layers, tabI4, tabP4, st, B, x, pr, hpsi, psi, lap);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}
}  // namespace

// workspace: the jets of every (direction, walker): D directions x (D + 1) slots x 3 floats
int64_t energy_dir_floats(int64_t B, int D) { return (int64_t)D * (D + 1) * 3 * B; }
bool energy_dir_capable(const MfmaDev* mdev) {
    return mdev->D >= 3 && mdev->D <= 8 && mdev->nbk == 1 && mdev->n_layers >= 0 && mdev->n_layers < kMaxLayers && !mdev->i_gate && !mdev->p_gate && !mdev->p_bias &&
           mdev->comp != nullptr && (mdev->const_floats + mdev->net_floats) * 4 <= 160 * 1024 - 512;
}
int launch_energy_dir(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* x, int64_t B, const Protons& pr,
                      float* hpsi, float* psi, float* lap, float* ws, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return WF_OK;
    switch (md.D) {
        case 3: return launch_dir_t<3>(mdev, md, tabI4, tabP4, x, B, pr, hpsi, psi, lap, ws, s);
        case 4: return launch_dir_t<4>(mdev, md, tabI4, tabP4, x, B, pr, hpsi, psi, lap, ws, s);
        case 5: return launch_dir_t<5>(mdev, md, tabI4, tabP4, x, B, pr, hpsi, psi, lap, ws, s);
        case 6: return launch_dir_t<6>(mdev, md, tabI4, tabP4, x, B, pr, hpsi, psi, lap, ws, s);
        case 7: return launch_dir_t<7>(mdev, md, tabI4, tabP4, x, B, pr, hpsi, psi, lap, ws, s);
        case 8: return launch_dir_t<8>(mdev, md, tabI4, tabP4, x, B, pr, hpsi, psi, lap, ws, s);
    }
    return WF_ERR_UNSUPPORTED;
}

}  // namespace wf
