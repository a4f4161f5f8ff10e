// wf_kernels_etile.hip -- local energy of large batches on the matrix cores (gfx950).
//
// H psi = -1/2 laplacian(psi) + V psi (physics.py:50-52, 79-93) needs psi with its gradient and Laplacian with respect to the walker's
// coordinates.  The wave kernels (wf_kernels_wave.hip) carry that (value, gradient, Laplacian / 2) jet through the model with one wave per
// walker; their conditioner products are LDS-fed GEMVs.  For large batches of the two-particle family this file splits the work by what
// the hardware is good at, one launch pair per conditioner net:
//
//   k_etile_cond   32 walkers per wave tile, the conditioner of ONE net on the matrix cores.  With two particles the MADE masks leave the
//                  conditioner ONE input (hidden degrees arange(64) % (D - 1) = 0: model_factory.py:15): every hidden and output unit is a
//                  function of the scalar u_0, so what travels is its Taylor triple (f, f', f'') in u_0 -- three channels = three extra
//                  column groups of the same split-fp16 MFMA products k_mfma issues (the weight operand is shared: mfma_step<3>); the
//                  activations r(x) = 1 / (2^x + 1) propagate the triple on the VALU.  Derivative channels are unbounded, fp16 is not: every
//                  (walker, channel) column is scaled by a power of two around each product (exact).  Output: the head's pre-activation
//                  triples, [row][channel][walker] in HBM (for the prior: already multiplied by ob_to_b); the head kernels turn them into
//                  jets in (x0, x1) with the chain rule through the jet of u_0.
//   k_etile_flow / k_etile_prior   one LANE per walker: sigmoid head, normalisations, table lerps of derivative orders 0..3 and the
//                  log-determinant as jet arithmetic in registers; row sums are sequential loops (no cross-lane traffic), the walker
//                  index is the fastest-moving one of every array (coalesced).
//
// State between launches (SoA, walker fastest): u_0, u_1, log det as jets; 12 floats per walker.  Head triples through HBM: 2 x 384 B per
// walker and net.  Same function as k_wave_fwd<2, RF<2>> + k_energy_out (same derivative rule of the table lerp: order nd -> table nd + 1),
// checked against it and against the torch oracle (tests/test_gpu_energy.py).  Coverage: D = 2, <= 64 bases (launch-per-net form: <= 32), mean-type box, IMADE layers,
// Waveflow prior, ungated heads (every homogeneous boundary dictionary: the tables carry the map); everything else stays on the wave kernel.
// (the adjoint header first: its head algebra is compiled WITHOUT the contraction pragma of the common header -- with it the reverse kernel came out 10 %
// slower, 189 instead of 138 spilled registers: profiles/r04_grad33_times.txt, "adjoint header under fp contract")
#include <hip/hip_runtime.h>

#include "wf_etile_adjoint.h"
#include "wf_etile_common.h"

// The jet / Taylor algebra of this file is checked against oracles by tolerance, not by operation order: multiply-add pairs may fuse (the build's
// default is -ffp-contract=off).  The pragma is lexical: the index arithmetic of the table lerp (make_lerp, div_by_n in wf_mfma_impl.h, included
// above) keeps the reference's separate roundings, so the bin indices stay bit-exact.
#pragma clang fp contract(fast)

namespace wf {

namespace {
#ifndef WF_ETILE_WAVES
#define WF_ETILE_WAVES 4
#endif
#ifndef WF_ETILE_OCC
#define WF_ETILE_OCC 2   // workgroups per CU the register budget is sized for: 256 registers, two waves per SIMD (the per-lane store addresses spill: 23 reloads per tile)
#endif
constexpr int kCondWaves = WF_ETILE_WAVES;   // 4 waves per workgroup, WF_ETILE_OCC workgroups per CU (unbounded, the three channel chains take 324 registers: one wave per SIMD)
using O2 = NetOff<2, 1>;

// The conditioner of one net for one tile, in pieces (NBK = 32-row blocks per dimension: 1 for <= 32 bases, 2 for <= 64).
//   cond_hidden   the two hidden layers: B fragments (split fp16, derivative channels scaled by 2^-e) of the second hidden layer's activations
//   cond_out      one 32-row output block of dimension 1: Taylor triples (f, f', f'') in u_0 of the head's pre-activations, accumulator layout
//   prior_c       the prior head's c = (o * keep) @ ob_to_b as triples, one 32-row block of c at a time, + the sum of the raw outputs (sign)
template <int NBK, int CH = NCH>
__device__ __forceinline__ void cond_hidden(const float* net, float u0v, float u1v, int lane, Frag (&f)[CH][2], int (&e)[CH]) {
    using O = NetOff<2, NBK>;
    const int h = lane >> 5;
    // the conditioner's inputs: (u_0, u_1) values; the Taylor seed in u_0 is (u_0, 1, 0) (u_1 reaches no hidden unit: masked weights)
    const float in0[2] = {u0v, 1.0f}, in1[2] = {u1v, 0.0f};
    // ---- layer 1 (f32 MFMA, K = 2: the two coordinates), both 32-unit blocks; the second-derivative channel starts at zero
    f32x16 a0[CH], a1[CH];
    init_acc<CH>(a0, net + O::b0 + (0 * 2 + h) * 16);
    init_acc<CH>(a1, net + O::b0 + (1 * 2 + h) * 16);
    {
        const float w0 = net[O::W0 + 0 * 64 + lane], w1 = net[O::W0 + 1 * 64 + lane];
#pragma unroll
        for (int c = 0; c < (CH < 2 ? CH : 2); ++c) {
            a0[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, h ? in1[c] : in0[c], a0[c], 0, 0, 0);
            a1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, h ? in1[c] : in0[c], a1[c], 0, 0, 0);
        }
    }
    act_block<CH>(a0);
    act_block<CH>(a1);
    to_frags<CH>(a0, a1, f, e);
    // ---- layer 2
    const _Float16* W1h = reinterpret_cast<const _Float16*>(net + O::W1h);
    const _Float16* W1l = reinterpret_cast<const _Float16*>(net + O::W1l);
    init_acc<CH>(a0, net + O::b1 + (0 + h) * 16);
    init_acc<CH>(a1, net + O::b1 + (2 + h) * 16);
    dense64_block<CH>(W1h, W1l, f, a0, lane);
    dense64_block<CH>(W1h + 2048, W1l + 2048, f, a1, lane);
    unscale<CH>(a0, e);
    unscale<CH>(a1, e);
    act_block<CH>(a0);
    act_block<CH>(a1);
    to_frags<CH>(a0, a1, f, e);
}
// output block kb of dimension 1 (dimension 0 is table-driven: k_prepare_dim0)
template <int NBK, int CH = NCH>
__device__ __forceinline__ void cond_out(const float* net, const Frag (&f)[CH][2], const int (&e)[CH], int kb, int lane, f32x16 (&a0)[CH]) {
    using O = NetOff<2, NBK>;
    const int h = lane >> 5;
    const _Float16* W2h = reinterpret_cast<const _Float16*>(net + O::W2h);
    const _Float16* W2l = reinterpret_cast<const _Float16*>(net + O::W2l);
    init_acc<CH>(a0, net + O::b2 + ((1 * NBK + kb) * 2 + h) * 16);
    dense64_block<CH>(W2h + kb * 2048, W2l + kb * 2048, f, a0, lane);
    unscale<CH>(a0, e);
}
// the whole conditioner (the launch-per-net path): head triples (PRIOR: of c) in a0, the sum of the raw outputs in s1
// cbP: the constant term of the B prior's boundary map times ob_to_b ([NBK][2][16], accumulator layout) or null; it is added to the VALUE channel only --
// the staged sampler, which reads nothing else, is the one caller with such models (the launch-per-net energy path leaves them to k_efused)
template <bool PRIOR, int NBK = 1, int CH = NCH>
__device__ __forceinline__ void cond_net(const float* net, const float* fkP, const _Float16* obh, float u0v, float u1v, int lane, f32x16 (&a0)[NBK][CH], float& s1,
                                         const float* cbP = nullptr, f32x16* wkeep = nullptr /* PRIOR: [NBK] the value channel of o * keep, or null */) {
    Frag f[CH][2];
    int e[CH];
    cond_hidden<NBK, CH>(net, u0v, u1v, lane, f, e);
    if (!PRIOR) {
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb) cond_out<NBK, CH>(net, f, e, kb, lane, a0[kb]);
    } else {
        f32x16 o[NBK][CH];
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb) cond_out<NBK, CH>(net, f, e, kb, lane, o[kb]);
        Frag of[NBK][CH];
        int eo[CH];
        prior_frags<NBK, CH>(o, fkP, lane, of, eo, s1);
        if (wkeep) {
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb) wkeep[kb] = o[kb][0];
        }
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb) {
            prior_c_block<NBK, CH>(obh, of, eo, kb, lane, a0[kb]);
            if (cbP) {
                const f32x16 cb = load16(cbP + (kb * 2 + (lane >> 5)) * 16);
#pragma unroll
                for (int r = 0; r < 16; ++r) a0[kb][0][r] = __builtin_fmaf(s1, cb[r], a0[kb][0][r]);
            }
        }
    }
}

// NBK row blocks per dimension: the head outputs go out as oj[tile][row 0 .. 32 NBK)[channel][32 walkers] (NBK = 2: the staged sampler of 33 .. 64 bases).
// CH = 1 (the staged sampler: round 4): the value channel alone -- a third of the matrix products, no derivative algebra in the activations, 128 instead of
// 384 B per walker and row block out (oj[tile][row][32 walkers])
template <bool PRIOR, int NBK = 1, int CH = NCH>
__global__ __launch_bounds__(kCondWaves * 64, WF_ETILE_OCC) void k_etile_cond(const MfmaDev mm, int net_index, const float* __restrict__ st, int64_t B,
                                                                float* __restrict__ oj, float* __restrict__ s1buf, float* __restrict__ ow = nullptr) {
    // ow (PRIOR, CH = 1; may be null): the value channel of o * keep, [tile][row][32 walkers] -- where the boundary map only zeroes coefficients these ARE the
    // plain B-spline coefficients of c (c = (o keep) @ ob_to_b, and ob_to_b @ b_to_ob = 1): the staged sampler's envelope reads them instead of forming c @ b_to_ob
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int next_tile;
    constexpr int kThreads = kCondWaves * 64;
    if (threadIdx.x == 0) next_tile = 0;
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    stage_floats<kThreads>(mm.image + (size_t)net_index * mm.net_floats, lds + mm.const_floats, mm.net_floats);
    __syncthreads();
    const float* net = lds + mm.const_floats;
    const float* fkP = lds + 32 * NBK;
    const _Float16* obh = reinterpret_cast<const _Float16*>(lds + 64 * NBK);
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (B + 31) >> 5;
    // this workgroup's tiles: blockIdx.x, blockIdx.x + gridDim.x, ... handed to its waves through a counter (oldest-wave-first arbitration)
    const int64_t my_tiles = n_tiles > (int64_t)blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    for (;;) {
        int q = 0;
        if (lane == 0) q = __hip_atomic_fetch_add(&next_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= my_tiles) break;
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)q * gridDim.x;
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        const float u0v = st[wl], u1v = st[(int64_t)4 * B + wl];
        f32x16 a0[NBK][CH], wk[NBK];
        float s1 = 0.0f;
        cond_net<PRIOR, NBK, CH>(net, fkP, obh, u0v, u1v, lane, a0, s1, (PRIOR && mm.p_bias) ? lds + 64 * NBK + NBK * NBK * 1024 + 64 * NBK : nullptr,
                                 (PRIOR && CH == 1 && ow) ? wk : nullptr);
        if (PRIOR && valid && h == 0) s1buf[w] = s1;
        if (PRIOR && CH == 1 && ow && valid) {
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) ow[(tile * (32 * NBK) + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = wk[kb][r];
        }
        // ---- store: oj[tile][row][c][32 walkers] (one contiguous block per tile), row = accumulator row of register r in lane half h of block kb
#ifdef WF_ABL_OJ   // ablation build (timing only): the head triples are computed, not stored
        if (valid && a0[0][0][0] == 12345.678f) {
#else
        if (valid) {
#endif
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                    for (int c = 0; c < CH; ++c) oj[(tile * (32 * NBK * CH) + row * CH + c) * 32 + j] = a0[kb][c][r];
                }
        }
    }
}

// ---------------------------------------------------------------------------- lane-per-walker stages
// BoxTransformLayer, mean type, two particles (made.py:156-183) as jets of (x0, x1)
__global__ void k_etile_box(const float* __restrict__ xg, int64_t B, float L, float* __restrict__ st) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float tol = 1e-7f;
    const J x0 = J{xg[b * 2], 1.0f, 0.0f, 0.0f}, x1 = J{xg[b * 2 + 1], 0.0f, 1.0f, 0.0f};
    const J mean = (x0 + x1) * 0.5f;
    const J l = mean - x0, wd = x1 - x0;
    J ld = jc(0.0f);
    const J space = jc(2 * L);
    const J diff = x1 - x0;
    const J u0 = diff * jrcp(space + tol);
    ld = ld - jlog(space + tol);
    const J den = (jc(2 * L) - wd) + tol;
    const J u1 = ((mean + L) - l) * jrcp(den);
    ld = ld - jlog(den);
    st_store(st, 0, B, b, u0);
    st_store(st, 1, B, b, u1);
    st_store(st, 2, B, b, ld);
}

// One IMADE layer behind its conditioner (made.py:66-81) + Reverse: dimension 0 from the composite table, dimension 1 from the head jets
__global__ __launch_bounds__(256) void k_etile_flow(const float4_t* __restrict__ comp /* this net: [n_mesh] {Y, Y', Y'', Y'''} */,
                                                    const float* __restrict__ tabI /* [n_mesh][8 row chunks][4 orders][4 rows] */, const float* __restrict__ gI, int nb,
                                                    int n_mesh, float reg, const float* __restrict__ oj, int64_t B, float* __restrict__ st) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const J u0 = st_load(st, 0, B, b), u1 = st_load(st, 1, B, b);
    J ld = st_load(st, 2, B, b);
    // ---- dimension 0
    J y0;
    {
        const LerpN L = nlerp(u0.v, n_mesh);
        const float4_t ca = comp[L.il], cb = comp[L.ir];
        const float t0 = __builtin_fmaf(cb.x - ca.x, L.t, ca.x), t1 = __builtin_fmaf(cb.y - ca.y, L.t, ca.y);
        const float t2 = __builtin_fmaf(cb.z - ca.z, L.t, ca.z), t3 = __builtin_fmaf(cb.w - ca.w, L.t, ca.w);
        y0 = jlift(t0, t1, t2, u0);
        ld = ld + jlog(jlift(t1, t2, t3, u0) + 1e-7f);
    }
    // ---- dimension 1: c_j = g_j (v_j / S0 + reg) / Q (calculate_bijection_params, + reg, remove_bias, boundary map: wf_model.cpp).
    // One pass over the rows: with V_k = sum_j v_j g_j B^(k)_j, R_k = sum_j g_j B^(k)_j, Qv = sum_j v_j g_j, G = sum_j g_j, S0 = sum_j v_j the
    // numerators are N_k = V_k / S0 + reg R_k and the normaliser Q = Qv / S0 + reg G.  Four rows per step: the lane's table rows come as
    // 16-byte loads (8 per step: 4 orders x the two mesh rows; rows >= nb are zero padding).
    const LerpN L = nlerp(u1.v, n_mesh);
    const int* bnd = reinterpret_cast<const int*>(tabI + (size_t)n_mesh * 128);   // [8 chunks][lo, hi] behind the table (wf_model.cpp: upload_chunked)
    // sums over the rows (see T2): S^(a) = sum v_j^(a), Qv^(a) = sum g_j v_j^(a), V[a][k] = sum v_j^(a) g_j T_j^(k) for a + k <= 3 (k <= 3 - a ... the
    // nine pairs the two numerators need), R[k] = sum g_j T_j^(k), G = sum g_j
    float S[3] = {0.0f, 0.0f, 0.0f}, Qv[3] = {0.0f, 0.0f, 0.0f}, R[4] = {0.0f, 0.0f, 0.0f, 0.0f}, G = 0.0f;
    float V0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, V1[3] = {0.0f, 0.0f, 0.0f}, V2[2] = {0.0f, 0.0f};   // V[a][k]: a = 0: k 0..3, a = 1: k 0..2, a = 2: k 0..1
    for (int j0 = 0; j0 < nb; j0 += 4) {
        // the chunk at the mesh index clamped to its support: the same bits, and the walkers outside the support read two shared lines
        const int lo = bnd[j0 >> 1], hi = bnd[(j0 >> 1) + 1];
        const float4_t* rl = reinterpret_cast<const float4_t*>(tabI + (size_t)min(max(L.il, lo), hi) * 128);   // [8 chunks][4 orders] float4
        const float4_t* rr = reinterpret_cast<const float4_t*>(tabI + (size_t)min(max(L.ir, lo), hi) * 128);
        float4_t ta[4], tb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ta[k] = rl[j0 + k];     // (chunk j0 / 4) * 4 + k
            tb[k] = rr[j0 + k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jr_ = j0 + q;
            if (jr_ >= nb) break;
            const float* p = oj + ((b >> 5) * (32 * NCH) + jr_ * NCH) * 32 + (b & 31);     // [tile][row][channel][32 walkers]
            float v0, v1, v2;
            r_triple(p[0], p[32], p[64], v0, v1, v2);       // the head's weight and its first two derivatives in u_0
            const float g = gI[jr_];
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = q == 0 ? ta[k].x : (q == 1 ? ta[k].y : (q == 2 ? ta[k].z : ta[k].w));
                const float bb = q == 0 ? tb[k].x : (q == 1 ? tb[k].y : (q == 2 ? tb[k].z : tb[k].w));
                t[k] = __builtin_fmaf(bb - a, L.t, a) * g;   // g_j folded into the row
            }
            S[0] += v0; S[1] += v1; S[2] += v2;
            Qv[0] = __builtin_fmaf(v0, g, Qv[0]); Qv[1] = __builtin_fmaf(v1, g, Qv[1]); Qv[2] = __builtin_fmaf(v2, g, Qv[2]);
#pragma unroll
            for (int k = 0; k < 4; ++k) { V0[k] = __builtin_fmaf(v0, t[k], V0[k]); R[k] += t[k]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) V1[k] = __builtin_fmaf(v1, t[k], V1[k]);
#pragma unroll
            for (int k = 0; k < 2; ++k) V2[k] = __builtin_fmaf(v2, t[k], V2[k]);
            G += g;
        }
    }
    J y1;
    flow_head_finish(S, Qv, R, G, V0, V1, V2, reg, u0, u1, y1, ld);
    st_store(st, 0, B, b, y1);   // Reverse (bijections.py:337-340)
    st_store(st, 1, B, b, y0);
    st_store(st, 2, B, b, ld);
}

// Waveflow prior (wavefunctions.py:54-71) + H psi (physics.py:60-93)
__global__ __launch_bounds__(256) void k_etile_prior(const float4_t* __restrict__ comp /* prior: {P, P', P''} with sign and norm */,
                                                     const float* __restrict__ tabP /* orthogonal B, [n_mesh][8][4][4] like tabI */, int nb, int n_mesh,
                                                     unsigned constrained_mask, const float* __restrict__ oj, const float* __restrict__ s1buf,
                                                     const float* __restrict__ st, const float* __restrict__ xg, int64_t B, const Protons pr,
                                                     float* __restrict__ hpsi, float* __restrict__ psi_out, float* __restrict__ lap_out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const J u0 = st_load(st, 0, B, b), u1 = st_load(st, 1, B, b), ld = st_load(st, 2, B, b);
    // the spline sees the clipped coordinate (:45): outside [0, 1] it is a constant
    const J uc0 = (u0.v < 0.0f) ? jc(0.0f) : (u0.v > 1.0f ? jc(1.0f) : u0);
    const J uc1 = (u1.v < 0.0f) ? jc(0.0f) : (u1.v > 1.0f ? jc(1.0f) : u1);
    J val0;
    {
        const LerpN L = nlerp(uc0.v, n_mesh);
        const float4_t ca = comp[L.il], cb = comp[L.ir];
        val0 = jlift(__builtin_fmaf(cb.x - ca.x, L.t, ca.x), __builtin_fmaf(cb.y - ca.y, L.t, ca.y), __builtin_fmaf(cb.z - ca.z, L.t, ca.z), uc0);
    }
    const LerpN L = nlerp(uc1.v, n_mesh);
    const int* bnd = reinterpret_cast<const int*>(tabP + (size_t)n_mesh * 128);
    // sums over the rows (see T2): D[a][k] = sum c_i^(a)(s) B_i^(k)(t), a + k <= 2; |c|^2 and its first two derivatives in s from cc, cc', c'c', cc''
    float D0[3] = {0.0f, 0.0f, 0.0f}, D1[2] = {0.0f, 0.0f}, D2 = 0.0f, cc = 0.0f, cc1 = 0.0f, c1c1 = 0.0f, cc2 = 0.0f;
    for (int i0 = 0; i0 < nb; i0 += 4) {
        const int lo = bnd[i0 >> 1], hi = bnd[(i0 >> 1) + 1];
        const float4_t* rl = reinterpret_cast<const float4_t*>(tabP + (size_t)min(max(L.il, lo), hi) * 128);
        const float4_t* rr = reinterpret_cast<const float4_t*>(tabP + (size_t)min(max(L.ir, lo), hi) * 128);
        float4_t ta[3], tb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ta[k] = rl[i0 + k];
            tb[k] = rr[i0 + k];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + q;
            if (i >= nb) break;
            const float* p = oj + ((b >> 5) * (32 * NCH) + i * NCH) * 32 + (b & 31);
            const float c0 = p[0], c1 = p[32], c2 = p[64];        // c_i and its derivatives in the unclipped u_0 (wavefunctions.py:40)
            float t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float a = q == 0 ? ta[k].x : (q == 1 ? ta[k].y : (q == 2 ? ta[k].z : ta[k].w));
                const float bb = q == 0 ? tb[k].x : (q == 1 ? tb[k].y : (q == 2 ? tb[k].z : tb[k].w));
                t[k] = __builtin_fmaf(bb - a, L.t, a);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) D0[k] = __builtin_fmaf(c0, t[k], D0[k]);
            D1[0] = __builtin_fmaf(c1, t[0], D1[0]); D1[1] = __builtin_fmaf(c1, t[1], D1[1]);
            D2 = __builtin_fmaf(c2, t[0], D2);
            cc = __builtin_fmaf(c0, c0, cc); cc1 = __builtin_fmaf(c0, c1, cc1); c1c1 = __builtin_fmaf(c1, c1, c1c1); cc2 = __builtin_fmaf(c0, c2, cc2);
        }
    }
    const float sgn = s1buf[b] < 0.0f ? -1.0f : 1.0f;
    const T2 N2 = T2{cc, 2.0f * cc1, 0.0f, 2.0f * (c1c1 + cc2), 0.0f, 0.0f};
    const T2 dotp = T2{D0[0], D1[0], D0[1], D2, D1[1], D0[2]};
    const J val1 = t2jet(dotp * t2rsqrt(N2), u0, uc1) * sgn;
    const float sc0 = (constrained_mask & 1u) ? 0.70710678118654752f : 1.0f, sc1 = (constrained_mask & 2u) ? 0.70710678118654752f : 1.0f;
    const J psi = ((val0 * sc0) * (val1 * sc1)) * jexp_half(ld);
    const float lap = 2.0f * psi.h;
    float V = 0.0f;   // physics.py:60-76
    for (int p = 0; p < pr.n; ++p)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float r = pr.pos[p] - xg[b * 2 + d];
            V -= 1.0f / sqrtf(1.0f + r * r);
        }
    {
        const float r = xg[b * 2 + 1] - xg[b * 2];
        V += 1.0f / sqrtf(1.0f + r * r);
    }
    hpsi[b] = -0.5f * lap + V * psi.v;
    if (psi_out) psi_out[b] = psi.v;
    if (lap_out) lap_out[b] = lap;
}

// ---------------------------------------------------------------------------- H psi in ONE kernel (every net resident in LDS)
// The conditioner's head triples stay in the accumulator registers: lane (walker j, half h) holds registers 4q .. 4q + 3 = rows 8q + 4h .. + 3 =
// row CHUNK 2q + h of the regrouped tables ([mesh][8 chunks][4 orders][4 rows]: one 64-byte segment per chunk and lerp end).  The head runs on
// those 16 rows per lane with the separable row sums of T2 (scalars, no jet per row); the two lane halves are combined with one
// v_permlane32_swap per sum; quotients, logarithms and the change to (x0, x1) jets once per walker.  No exchange buffer, no state in HBM:
// 8 B per walker in, 4 .. 12 B out.  One persistent workgroup of 8 waves per CU (two per SIMD, 256 registers), tiles from an LDS counter.
#ifndef WF_FUSED_WAVES   // (experiment switch; round 4, 2^20 walkers, one / two row blocks: 8 waves 0.81 / 1.16 ms, 6 waves 0.88 / 1.36, 4 waves -- no spills -- 0.98 / 1.31, 12 waves -- 133 / 330 spilled -- 0.96 / 2.21, 16 waves 2.31 / 3.98)
#define WF_FUSED_WAVES 8
#endif
constexpr int kFusedWaves = WF_FUSED_WAVES;

__device__ __forceinline__ void box_mean2(float x0v, float x1v, float L, J& u0, J& u1, J& ld) {   // (k_etile_box)
    const float tol = 1e-7f;
    const J x0 = J{x0v, 1.0f, 0.0f, 0.0f}, x1 = J{x1v, 0.0f, 1.0f, 0.0f};
    const J mean = (x0 + x1) * 0.5f;
    const J l = mean - x0, wd = x1 - x0;
    const J space = jc(2 * L);
    const J diff = x1 - x0;
    u0 = diff * jrcp(space + tol);
    ld = jc(0.0f) - jlog(space + tol);
    const J den = (jc(2 * L) - wd) + tol;
    u1 = ((mean + L) - l) * jrcp(den);
    ld = ld - jlog(den);
}
// PBIAS: the B prior's boundary map has a constant term (a constraint with a non-zero value): its own instantiation, so that the derivative channels'
// sums do not lengthen live ranges in the common one
template <int NBK, bool PBIAS = false>
__global__ __launch_bounds__(kFusedWaves * 64) void k_efused(const MfmaDev mm, const float* __restrict__ tabI, const float* __restrict__ tabP,
                                                             const float* __restrict__ xg, int64_t B, const Protons pr, float* __restrict__ hpsi,
                                                             float* __restrict__ psi_out, float* __restrict__ lap_out, float* __restrict__ st_out) {
    // st_out (may be null): the (u_0, u_1, log det) jets at the input of every net, [net][slot][channel][B] -- what the gradient path's
    // per-net reverse kernels (k_ebwd) restart from
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int next_tile;
    __shared__ int bnd_s[32 * NBK];   // support bounds of the table chunks: [I: 8 NBK][lo, hi], [prior: 8 NBK][lo, hi]
    constexpr int kThreads = kFusedWaves * 64;
    constexpr int kMeshStride = 128 * NBK;   // floats per mesh point of the regrouped tables: [8 NBK chunks][4 orders][4 rows]
    if (threadIdx.x == 0) next_tile = 0;
    if (threadIdx.x < 16 * NBK) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabI + (size_t)mm.n_mesh * kMeshStride)[threadIdx.x];
    else if (threadIdx.x < 32 * NBK) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabP + (size_t)mm.n_mesh * kMeshStride)[threadIdx.x - 16 * NBK];
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    stage_floats<kThreads>(mm.image, lds + mm.const_floats, mm.net_floats * mm.n_nets);
    __syncthreads();
    const float* fkI = lds;
    const float* fkP = lds + 32 * NBK;
    const _Float16* obh = reinterpret_cast<const _Float16*>(lds + 64 * NBK);
    const float* cbP = lds + 64 * NBK + NBK * NBK * 1024 + 64 * NBK;   // [NBK][2][16] constant term of the B prior's boundary map times ob_to_b (mm.p_bias; wf_model.cpp)
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int n_mesh = mm.n_mesh;
    const int64_t n_tiles = (B + 31) >> 5;
    const int64_t my_tiles = n_tiles > (int64_t)blockIdx.x ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    int f16_bad = 0;
    if (mm.f16_ovf)
        for (int n = 0; n < mm.n_nets; ++n) f16_bad |= mm.f16_ovf[n];
    for (;;) {
        int q_ = 0;
        if (lane == 0) q_ = __hip_atomic_fetch_add(&next_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        q_ = __builtin_amdgcn_readfirstlane(q_);
        if (q_ >= my_tiles) break;
        const int64_t tile = (int64_t)blockIdx.x + (int64_t)q_ * gridDim.x;
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        const float x0v = xg[wl * 2], x1v = xg[wl * 2 + 1];
        J u0, u1, ld;
        box_mean2(x0v, x1v, mm.box_L, u0, u1, ld);
        // ---- flow layers (made.py:66-81 + Reverse)
        for (int l = 0; l < mm.n_layers; ++l) {
            const float* net = lds + mm.const_floats + (size_t)l * mm.net_floats;
            if (st_out && valid && h == 0) {
                st_store(st_out + (size_t)l * 12 * B, 0, B, w, u0);
                st_store(st_out + (size_t)l * 12 * B, 1, B, w, u1);
                st_store(st_out + (size_t)l * 12 * B, 2, B, w, ld);
            }
            Frag f[NCH][2];
            int e[NCH];
            cond_hidden<NBK>(net, u0.v, u1.v, lane, f, e);
            // dimension 0: composite table of the net, all four orders
            J y0;
            {
                const LerpN L0 = nlerp(u0.v, n_mesh);
                const float4_t* comp = mm.comp + (size_t)l * n_mesh;
                const float4_t ca = comp[L0.il], cb = comp[L0.ir];
                const float t0 = __builtin_fmaf(cb.x - ca.x, L0.t, ca.x), t1 = __builtin_fmaf(cb.y - ca.y, L0.t, ca.y);
                const float t2 = __builtin_fmaf(cb.z - ca.z, L0.t, ca.z), t3 = __builtin_fmaf(cb.w - ca.w, L0.t, ca.w);
                y0 = jlift(t0, t1, t2, u0);
                ld = ld + jlog(jlift(t1, t2, t3, u0) + 1e-7f);
            }
            // dimension 1: the lane's 16 rows of every 32-row block
            const LerpN L = nlerp(u1.v, n_mesh);
            FlowSums a = {};
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb) {
                f32x16 o[NCH];
                cond_out<NBK>(net, f, e, kb, lane, o);
                flow_rows(a, o, load16(fkI + (kb * 2 + h) * 16), tabI, kMeshStride, bnd_s, L, kb, h);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) { a.S[k] = xhalf_sum(a.S[k]); a.Qv[k] = xhalf_sum(a.Qv[k]); a.V1[k] = xhalf_sum(a.V1[k]); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.R[k] = xhalf_sum(a.R[k]); a.V0[k] = xhalf_sum(a.V0[k]); }
#pragma unroll
            for (int k = 0; k < 2; ++k) a.V2[k] = xhalf_sum(a.V2[k]);
            J y1;
            flow_head_finish(a.S, a.Qv, a.R, mm.F_I, a.V0, a.V1, a.V2, mm.i_reg, u0, u1, y1, ld);
            u0 = y1;   // Reverse (bijections.py:337-340)
            u1 = y0;
        }
        // ---- Waveflow prior (wavefunctions.py:54-71)
        J psi;
        {
            const float* net = lds + mm.const_floats + (size_t)mm.n_layers * mm.net_floats;
            if (st_out && valid && h == 0) {
                st_store(st_out + (size_t)mm.n_layers * 12 * B, 0, B, w, u0);
                st_store(st_out + (size_t)mm.n_layers * 12 * B, 1, B, w, u1);
                st_store(st_out + (size_t)mm.n_layers * 12 * B, 2, B, w, ld);
            }
            float s1 = 0.0f, sder[2] = {0.0f, 0.0f};
            Frag of[NBK][NCH];
            int eo[NCH];
            {
                Frag f[NCH][2];
                int e[NCH];
                cond_hidden<NBK>(net, u0.v, u1.v, lane, f, e);   // (the conditioner sees the unclipped u_0, wavefunctions.py:40)
                f32x16 o[NBK][NCH];
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) cond_out<NBK>(net, f, e, kb, lane, o[kb]);
                prior_frags<NBK>(o, fkP, lane, of, eo, s1, PBIAS ? sder : nullptr);
            }
            const J uc0 = (u0.v < 0.0f) ? jc(0.0f) : (u0.v > 1.0f ? jc(1.0f) : u0);   // the spline sees the clipped coordinate (:45)
            const J uc1 = (u1.v < 0.0f) ? jc(0.0f) : (u1.v > 1.0f ? jc(1.0f) : u1);
            J val0;
            {
                const LerpN L0 = nlerp(uc0.v, n_mesh);
                const float4_t* comp = mm.comp + (size_t)mm.n_layers * n_mesh;
                const float4_t ca = comp[L0.il], cb = comp[L0.ir];
                val0 = jlift(__builtin_fmaf(cb.x - ca.x, L0.t, ca.x), __builtin_fmaf(cb.y - ca.y, L0.t, ca.y), __builtin_fmaf(cb.z - ca.z, L0.t, ca.z), uc0);
            }
            const LerpN L = nlerp(uc1.v, n_mesh);
            PriorSums a = {};
#pragma unroll
            for (int ko = 0; ko < NBK; ++ko) {
                f32x16 cblk[NCH];
                prior_c_block<NBK>(obh, of, eo, ko, lane, cblk);
                if (PBIAS) {   // a boundary constraint with a non-zero value (bsplines_jax.py:173-199): c += (sum o) * (b @ ob_to_b), channel by channel
                    const f32x16 cb = load16(cbP + (ko * 2 + h) * 16);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        cblk[0][r] = __builtin_fmaf(s1, cb[r], cblk[0][r]);
                        cblk[1][r] = __builtin_fmaf(sder[0], cb[r], cblk[1][r]);
                        cblk[2][r] = __builtin_fmaf(sder[1], cb[r], cblk[2][r]);
                    }
                }
                prior_rows(a, cblk, tabP, kMeshStride, bnd_s + 16 * NBK, L, ko, h);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) a.D0[k] = xhalf_sum(a.D0[k]);
            a.D1[0] = xhalf_sum(a.D1[0]); a.D1[1] = xhalf_sum(a.D1[1]); a.D2 = xhalf_sum(a.D2);
            a.cc = xhalf_sum(a.cc); a.cc1 = xhalf_sum(a.cc1); a.c1c1 = xhalf_sum(a.c1c1); a.cc2 = xhalf_sum(a.cc2);
            const float sgn = s1 < 0.0f ? -1.0f : 1.0f;
            const T2 N2 = T2{a.cc, 2.0f * a.cc1, 0.0f, 2.0f * (a.c1c1 + a.cc2), 0.0f, 0.0f};
            const T2 dotp = T2{a.D0[0], a.D1[0], a.D0[1], a.D2, a.D1[1], a.D0[2]};
            const J val1 = t2jet(dotp * t2rsqrt(N2), u0, uc1) * sgn;
            const float sc0 = (mm.constrained_mask & 1u) ? 0.70710678118654752f : 1.0f, sc1 = (mm.constrained_mask & 2u) ? 0.70710678118654752f : 1.0f;
            psi = ((val0 * sc0) * (val1 * sc1)) * jexp_half(ld);
        }
        if (valid && h == 0) {
            const float lap = 2.0f * psi.h;
            float V = 0.0f;   // physics.py:60-76
            for (int p = 0; p < pr.n; ++p) {
                const float r0 = pr.pos[p] - x0v, r1 = pr.pos[p] - x1v;
                V -= 1.0f / sqrtf(1.0f + r0 * r0);
                V -= 1.0f / sqrtf(1.0f + r1 * r1);
            }
            {
                const float r = x1v - x0v;
                V += 1.0f / sqrtf(1.0f + r * r);
            }
            // a packed weight outside the fp16 range (k_fold_bias): NaN instead of whatever inf operands made of the walker
            hpsi[w] = f16_bad ? __builtin_nanf("") : -0.5f * lap + V * psi.v;
            if (psi_out) psi_out[w] = f16_bad ? __builtin_nanf("") : psi.v;
            if (lap_out) lap_out[w] = f16_bad ? __builtin_nanf("") : lap;
        }
    }
}

// ============================================================================ parameter gradients on the matrix cores (vqmc.py:193-221)
// grad[p] = sum_b ( w_psi[b] d psi_b / d theta_p + w_lap[b] d laplacian_b / d theta_p ) for the two-particle family (<= 64 bases: NBK = 1 or 2 row blocks per dimension), batch by batch:
//   k_efused (st_out)   forward, leaves the (u_0, u_1, log det) jets at the input of every net
//   k_ebwd<PRIOR>       one launch per net, last net first: recomputes the net's forward from its input jets, pulls the adjoint of its output
//                       jets back through the head algebra (wf_etile_adjoint.h) to adjoint head triples, through the conditioner with TRANSPOSED
//                       operand images on the matrix cores (three channels, like the forward), writes the adjoint of the net's input jets for the
//                       next launch -- and, since round 4, forms the weight-gradient products dW[k][u] = sum_walkers sum_channels X_c[k][w] Y_c[u][w]
//                       itself.  The walker axis is the K of that product, while every tensor of the sweep has the walker on the LANE (accumulator
//                       layout): the operands are transposed ON THE MATRIX CORES -- an fp16 fragment times a 0/1 permutation operand is an exact
//                       transposition, one v_mfma per K step (tr_frag) -- and the six 32 x 32 blocks of (dW1, dW2) accumulate in LDS, one private set
//                       per wave (24 KB; the four sets + the operand images fill the 160 KB), summed over the workgroup's waves in wave order at the
//                       end: one 25.6 KB block of partial sums per workgroup.  Tiles are dealt to the waves statically, so the sums -- and whole
//                       training runs -- stay bitwise reproducible.  Rounds 2 - 3 dumped the operands per tile (66 KB: 270 MB per net and 2^17
//                       walkers) for a second kernel (k_ewgrad) that read them back: ~2 GB of HBM traffic per call against ~27 MB algorithmic.
//                       One wave per SIMD (512 registers).
//   k_egrad_reduce, k_egrad_scatter   reduction over the workgroups' blocks (fixed order); scales and folds back to the flat leaf order
__device__ __forceinline__ float wave_max(float v) {   // v >= 0
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
    const unsigned u = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return xhalf_max(fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
}
__device__ __forceinline__ int exponent_of(float amax) { return amax > 0.0f ? __builtin_amdgcn_frexp_expf(amax) : 0; }
using JA = adj::Jt<float>;
using TA = adj::T2t<float>;
#ifndef WF_BWD_WAVES
#define WF_BWD_WAVES 4
#endif
#define WF_BWD_WAVES_ WF_BWD_WAVES
constexpr int kBwdWaves = WF_BWD_WAVES;
// Gradient block of a net (floats, in the units of the MFMA image; NBK = 32-row blocks of the head per dimension):
//   GW0 [64] (d / d W0'[0][u]), Gb0 [64], GW1 [64][64] (k, u), Gb1 [64], GW2 [64][32 NBK] (k, row), Gb2 of dimension 1 [32 NBK], of dimension 0 [32 NBK]
template <int NBK>
struct GL {
    static constexpr int W0 = 0, b0 = 64, W1 = 128, b1 = 4224, W2 = 4288, b21 = W2 + 2048 * NBK, b20 = b21 + 32 * NBK, floats = b20 + 32 * NBK;
};
constexpr int g_floats(int nbk) { return 4288 + 2048 * nbk + 64 * nbk; }
static_assert(GL<1>::floats == g_floats(1) && GL<2>::floats == g_floats(2), "gradient block layout");
constexpr int kESplit = 256;        // partial blocks per net: one per workgroup of k_ebwd (grid <= 256), summed in block order by k_egrad_reduce
// LDS accumulators of a workgroup: blocks 0..3 = dW1 (k block mb = b >> 1, u block nb = b & 1), 4.. = dW2 (k block (b - 4) / NBK, row block (b - 4) % NBK),
// each [4 q][64 lanes][4] (register 4 q + e of the lane: one conflict-free ds_read_b128 per q)
constexpr int acc_blocks(int nbk) { return 4 + 2 * nbk; }
// sets of accumulator blocks per workgroup.  One row block: a private set per wave (4 x 24 KB beside 66 KB of images), summed in wave order at the end.
// Two row blocks: the four private sets (128 KB) do not fit beside 103 KB of images -- ONE shared set filled in tile order (acc_add).  -DWF_ACC_SHARED
// (experiment) shares the set for one row block too: 1.040 ms per loss + gradient of 2^17 walkers against 0.985 ms (the waves move in step, one add apart:
// any jitter of one holds up the other three); profiles/r04_grad33_times.txt
#ifdef WF_ACC_SHARED
constexpr int acc_sets(int) { return 1; }
#else
constexpr int acc_sets(int nbk) { return nbk == 1 ? WF_BWD_WAVES_ : 1; }
#endif
__device__ __forceinline__ int acc_rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // row of register r in lane half h (accumulator layout)
__device__ __forceinline__ f32x16 acc_load(const float* aw, int b, int lane) {
    f32x16 a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(aw + ((b * 4 + q) * 64 + lane) * 4);
        a[4 * q] = v[0]; a[4 * q + 1] = v[1]; a[4 * q + 2] = v[2]; a[4 * q + 3] = v[3];
    }
    return a;
}
__device__ __forceinline__ void acc_store(float* aw, int b, int lane, const f32x16& a) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(aw + ((b * 4 + q) * 64 + lane) * 4) = f32x4{a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
}
// B operand of the transposition product: P[K = (s, khalf, i)][n] = 1 where the K slot holds row n (K slot (s, h, i) of a fragment = register 8 s + i of half h)
__device__ __forceinline__ void make_perm(int lane, f16x8 (&pm)[2]) {
    const int n = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) pm[s][i] = acc_rho(8 * s + i, h) == n ? (_Float16)1.0f : (_Float16)0.0f;
}
__device__ __forceinline__ f16x8 cvt8(const f32x16& d, int s) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    f16x8 o;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f16x2 pr = __builtin_convertvector((f32x2){d[8 * s + i], d[8 * s + i + 1]}, f16x2);
        o[i] = pr[0]; o[i + 1] = pr[1];
    }
    return o;
}
// One 32-row block X (accumulator layout: lane = walker, registers = rows) given as fp16 fragments (hi, lo) -> X^T as fp16 fragments with the ROW on the
// lane and the walkers in the registers (walker acc_rho(r, half) in register r): D[walker][row] = sum_K frag[walker][K] P[K][row] has one non-zero term
// per entry, so hi and lo come through exactly.  rowsum (may be null): += the sum over the lane's 16 walkers of hi + lo (both halves: xhalf at the end).
__device__ __forceinline__ void tr_frag(const Frag& f, const f16x8 (&pm)[2], Frag& t, float* rowsum = nullptr) {
    f32x16 dh = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dl = dh;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dh = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.hi[s], pm[s], dh, 0, 0, 0);
        dl = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.lo[s], pm[s], dl, 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) { t.hi[s] = cvt8(dh, s); t.lo[s] = cvt8(dl, s); }
    if (rowsum) {
        float a = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += dh[r] + dl[r];
        *rowsum += a;
    }
}
__device__ __forceinline__ float half32_sum(float v) {   // sum over the 32 lanes of this lane's half, in every lane of it
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    const unsigned u = __float_as_uint(v);
    const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
__device__ __forceinline__ JA ja_load(const float* __restrict__ st, int slot, int64_t B, int64_t w) {
    const float* p = st + (int64_t)slot * 4 * B + w;
    return JA{p[0], p[B], p[2 * B], p[3 * B]};
}
__device__ __forceinline__ void ja_store(float* __restrict__ st, int slot, int64_t B, int64_t w, JA x) {
    float* p = st + (int64_t)slot * 4 * B + w;
    p[0] = x.v; p[B] = x.a; p[2 * B] = x.b; p[3 * B] = x.h;
}
__device__ __forceinline__ float r_of(float x) { return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x) + 1.0f); }
// Fragments of ADJOINT tensors.  UNI: one power of two per (TILE, channel) instead of per (walker, channel) -- the wave's largest.  The adjoint tensors
// of the reverse kernel take it: what they feed are sums over walkers (the weight gradients; the input adjoints, which the next net's reverse again only
// sums), so a walker far below the tile's largest loses bits that do not show in any sum, and the same fragments serve as operands of the products over
// the walker axis, which need one scale per tile.
// two blocks, every channel scaled (adjoints are unbounded in every channel)
template <bool UNI = false>
__device__ __forceinline__ void to_frags_all(const f32x16 (&blk0)[NCH], const f32x16 (&blk1)[NCH], Frag (&f)[NCH][2], int (&e)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float amax = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fmaxf(fabsf(blk0[c][r]), fabsf(blk1[c][r])));
        e[c] = UNI ? exponent_of(wave_max(amax)) : col_exponent(amax);
        const float sc = __builtin_amdgcn_ldexpf(1.0f, -e[c]);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float r8[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) r8[jj] = (ob == 0 ? blk0[c][8 * s + jj] : blk1[c][8 * s + jj]) * sc;
                split8(r8, f[c][ob].hi[s], f[c][ob].lo[s]);
            }
    }
}
// the NBK row blocks of the adjoint head triples -> fragments [channel][row block] (the K steps of the product with W2'), one power of two per (tile, channel)
template <int NBK>
__device__ __forceinline__ void to_frags_kb(const f32x16 (&blk)[NBK][NCH], Frag (&f)[NCH][2], int (&e)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float amax = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(blk[kb][c][r]));
        e[c] = exponent_of(wave_max(amax));
        const float sc = __builtin_amdgcn_ldexpf(1.0f, -e[c]);
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float r8[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) r8[jj] = blk[kb][c][8 * s + jj] * sc;
                split8(r8, f[c][kb].hi[s], f[c][kb].lo[s]);
            }
    }
}
__device__ __forceinline__ void unscale_all(f32x16 (&acc)[NCH], const int (&e)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float sc = __builtin_amdgcn_ldexpf(1.0f, e[c]);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = acc[c][r] * sc;
    }
}
// (x, x', x'') and the adjoint of (r, r' x', r' x'' + r'' x'^2) -> adjoint of (x, x', x''), in place in g
__device__ __forceinline__ void act_block_bwd(const f32x16 (&x)[NCH], f32x16 (&g)[NCH]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a, b, c;
        adj::r_triple_bwd(r_of(x[0][r]), x[1][r], x[2][r], g[0][r], g[1][r], g[2][r], a, b, c);
        g[0][r] = a; g[1][r] = b; g[2][r] = c;
    }
}
// extended row sums of a flow head over the lane's 16 rows (dimension 1: triples from the conditioner; CONST: dimension 0, (bias, 0, 0))
template <bool CONST>
__device__ __forceinline__ void flow_rows_ext(adj::FlowSumsT<float>& a, const f32x16 (&o)[NCH], const f32x16& g16, const float* __restrict__ tabI, int mesh_stride,
                                              const int* bnd, const LerpN& L, int kb, int h) {
    // the records of chunk q + 1 are requested before the rows of chunk q are worked on (two sets of 4 + 4 records in flight: the compiler's own order
    // waited for every set right behind its request -- one exposed L2 round trip per chunk)
    float4_t tq[2][2][4];
    chunk_rows<4>(tabI, mesh_stride, bnd, L, 8 * kb + h, tq[0][0], tq[0][1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < 3) chunk_rows<4>(tabI, mesh_stride, bnd, L, 8 * kb + 2 * (q + 1) + h, tq[(q + 1) & 1][0], tq[(q + 1) & 1][1]);
        const float4_t (&ta)[4] = tq[q & 1][0], (&tb)[4] = tq[q & 1][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            float v0, v1 = 0.0f, v2 = 0.0f;
            if (CONST) v0 = r_of(o[0][r]);
            else adj::r_triple(r_of(o[0][r]), o[1][r], o[2][r], v0, v1, v2);
            const float g = g16[r];
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_fmaf(tb[k][e] - ta[k][e], L.t, ta[k][e]) * g;
            a.s[0] += v0; a.qv[0] = __builtin_fmaf(v0, g, a.qv[0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a.v0[k] = __builtin_fmaf(v0, t[k], a.v0[k]); a.r[k] += t[k]; }
            if (!CONST) {
                a.s[1] += v1; a.s[2] += v2;
                a.qv[1] = __builtin_fmaf(v1, g, a.qv[1]); a.qv[2] = __builtin_fmaf(v2, g, a.qv[2]);
#pragma unroll
                for (int k = 0; k < 4; ++k) a.v1[k] = __builtin_fmaf(v1, t[k], a.v1[k]);
#pragma unroll
                for (int k = 0; k < 3; ++k) a.v2[k] = __builtin_fmaf(v2, t[k], a.v2[k]);
            }
        }
    }
}
__device__ __forceinline__ void flow_sums_xhalf(adj::FlowSumsT<float>& a) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { a.s[k] = xhalf_sum(a.s[k]); a.qv[k] = xhalf_sum(a.qv[k]); a.v2[k] = xhalf_sum(a.v2[k]); }
#pragma unroll
    for (int k = 0; k < 4; ++k) { a.r[k] = xhalf_sum(a.r[k]); a.v0[k] = xhalf_sum(a.v0[k]); a.v1[k] = xhalf_sum(a.v1[k]); }
}
// adjoint head triples of the lane's rows from the adjoints of the row sums (ab: summed over the halves already, the same in both)
template <bool CONST>
__device__ __forceinline__ void flow_rows_bwd(const adj::FlowSumsT<float>& ab, const f32x16 (&o)[NCH], const f32x16& g16, const float* __restrict__ tabI,
                                              int mesh_stride, const int* bnd, const LerpN& L, int kb, int h, f32x16 (&ob)[NCH]) {
    // the records of chunk q + 1 are requested before the rows of chunk q are worked on (two sets of 4 + 4 records in flight: the compiler's own order
    // waited for every set right behind its request -- one exposed L2 round trip per chunk)
    float4_t tq[2][2][4];
    chunk_rows<4>(tabI, mesh_stride, bnd, L, 8 * kb + h, tq[0][0], tq[0][1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < 3) chunk_rows<4>(tabI, mesh_stride, bnd, L, 8 * kb + 2 * (q + 1) + h, tq[(q + 1) & 1][0], tq[(q + 1) & 1][1]);
        const float4_t (&ta)[4] = tq[q & 1][0], (&tb)[4] = tq[q & 1][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            const float g = g16[r];
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_fmaf(tb[k][e] - ta[k][e], L.t, ta[k][e]) * g;
            float vb0 = __builtin_fmaf(g, ab.qv[0], ab.s[0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) vb0 = __builtin_fmaf(ab.v0[k], t[k], vb0);
            const float rr = r_of(o[0][r]);
            if (CONST) {
                ob[0][r] = vb0 * adj::r_derivs(rr).r1;
                ob[1][r] = 0.0f; ob[2][r] = 0.0f;
            } else {
                float vb1 = __builtin_fmaf(g, ab.qv[1], ab.s[1]), vb2 = __builtin_fmaf(g, ab.qv[2], ab.s[2]);
#pragma unroll
                for (int k = 0; k < 3; ++k) vb1 = __builtin_fmaf(ab.v1[k], t[k], vb1);
#pragma unroll
                for (int k = 0; k < 2; ++k) vb2 = __builtin_fmaf(ab.v2[k], t[k], vb2);
                float x0b, x1b, x2b;
                adj::r_triple_bwd(rr, o[1][r], o[2][r], vb0, vb1, vb2, x0b, x1b, x2b);
                ob[0][r] = x0b; ob[1][r] = x1b; ob[2][r] = x2b;
            }
        }
    }
}
// the prior's rows: extended sums from the triples of c (CONST: channel 0 only), and back
template <bool CONST>
__device__ __forceinline__ void prior_rows_ext(adj::PriorSumsT<float>& a, const f32x16 (&c)[NCH], const float* __restrict__ tabP, int mesh_stride, const int* bnd,
                                               const LerpN& L, int kb, int h) {
    // the records of chunk q + 1 are requested before the rows of chunk q are worked on (two sets of 4 + 4 records in flight: the compiler's own order
    // waited for every set right behind its request -- one exposed L2 round trip per chunk)
    float4_t tq[2][2][4];
    chunk_rows<4>(tabP, mesh_stride, bnd, L, 8 * kb + h, tq[0][0], tq[0][1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < 3) chunk_rows<4>(tabP, mesh_stride, bnd, L, 8 * kb + 2 * (q + 1) + h, tq[(q + 1) & 1][0], tq[(q + 1) & 1][1]);
        const float4_t (&ta)[4] = tq[q & 1][0], (&tb)[4] = tq[q & 1][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            const float c0 = c[0][r];
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = __builtin_fmaf(tb[k][e] - ta[k][e], L.t, ta[k][e]);
#pragma unroll
            for (int k = 0; k < 4; ++k) a.d0[k] = __builtin_fmaf(c0, t[k], a.d0[k]);
            a.cc = __builtin_fmaf(c0, c0, a.cc);
            if (!CONST) {
                const float c1 = c[1][r], c2 = c[2][r];
#pragma unroll
                for (int k = 0; k < 3; ++k) a.d1[k] = __builtin_fmaf(c1, t[k], a.d1[k]);
                a.d2[0] = __builtin_fmaf(c2, t[0], a.d2[0]); a.d2[1] = __builtin_fmaf(c2, t[1], a.d2[1]);
                a.cc1 = __builtin_fmaf(c0, c1, a.cc1); a.c1c1 = __builtin_fmaf(c1, c1, a.c1c1); a.cc2 = __builtin_fmaf(c0, c2, a.cc2);
            }
        }
    }
}
__device__ __forceinline__ void prior_sums_xhalf(adj::PriorSumsT<float>& a) {
#pragma unroll
    for (int k = 0; k < 4; ++k) a.d0[k] = xhalf_sum(a.d0[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) a.d1[k] = xhalf_sum(a.d1[k]);
    a.d2[0] = xhalf_sum(a.d2[0]); a.d2[1] = xhalf_sum(a.d2[1]);
    a.cc = xhalf_sum(a.cc); a.cc1 = xhalf_sum(a.cc1); a.c1c1 = xhalf_sum(a.c1c1); a.cc2 = xhalf_sum(a.cc2);
}
template <bool CONST>
__device__ __forceinline__ void prior_rows_bwd(const adj::PriorSumsT<float>& ab, const f32x16 (&c)[NCH], const float* __restrict__ tabP, int mesh_stride,
                                               const int* bnd, const LerpN& L, int kb, int h, f32x16 (&cb)[NCH]) {
    // the records of chunk q + 1 are requested before the rows of chunk q are worked on (two sets of 3 + 3 records in flight: the compiler's own order
    // waited for every set right behind its request -- one exposed L2 round trip per chunk)
    float4_t tq[2][2][3];
    chunk_rows<3>(tabP, mesh_stride, bnd, L, 8 * kb + h, tq[0][0], tq[0][1]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q < 3) chunk_rows<3>(tabP, mesh_stride, bnd, L, 8 * kb + 2 * (q + 1) + h, tq[(q + 1) & 1][0], tq[(q + 1) & 1][1]);
        const float4_t (&ta)[3] = tq[q & 1][0], (&tb)[3] = tq[q & 1][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 4 * q + e;
            float t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) t[k] = __builtin_fmaf(tb[k][e] - ta[k][e], L.t, ta[k][e]);
            const float c0 = c[0][r];
            float b0 = 2.0f * ab.cc * c0;
#pragma unroll
            for (int k = 0; k < 3; ++k) b0 = __builtin_fmaf(ab.d0[k], t[k], b0);
            if (CONST) {
                cb[0][r] = b0; cb[1][r] = 0.0f; cb[2][r] = 0.0f;
            } else {
                const float c1 = c[1][r], c2 = c[2][r];
                cb[0][r] = b0 + ab.cc1 * c1 + ab.cc2 * c2;
                cb[1][r] = ab.d1[0] * t[0] + ab.d1[1] * t[1] + ab.cc1 * c0 + 2.0f * ab.c1c1 * c1;
                cb[2][r] = ab.d2[0] * t[0] + ab.cc2 * c0;
            }
        }
    }
}
// NB blocks of triples -> fragments [block][channel], one power of two per (walker, channel) over the NB blocks (the operand of a product over the ROWS)
template <int NB>
__device__ __forceinline__ void to_frags_n(const f32x16 (&blk)[NB][NCH], Frag (&f)[NB][NCH], int (&e)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float amax = 0.0f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(blk[b][c][r]));
        e[c] = col_exponent(amax);
        const float sc = __builtin_amdgcn_ldexpf(1.0f, -e[c]);
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float r8[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) r8[jj] = blk[b][c][8 * s + jj] * sc;
                split8(r8, f[b][c].hi[s], f[b][c].lo[s]);
            }
    }
}
// ... of one channel
template <int NB>
__device__ __forceinline__ void to_frags_n1(const f32x16 (&blk)[NB], Frag (&f)[NB], int& e) {
    float amax = 0.0f;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(blk[b][r]));
    e = col_exponent(amax);
    const float sc = __builtin_amdgcn_ldexpf(1.0f, -e);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float r8[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) r8[jj] = blk[b][8 * s + jj] * sc;
            split8(r8, f[b].hi[s], f[b].lo[s]);
        }
}
// block ko of w @ M for ONE channel (M's image as prior_c_block takes it: [ko][ki]{hi 1024, lo 1024})
template <int NBK>
__device__ __forceinline__ void c_block1(const _Float16* obh, const Frag (&wf)[NBK], int e, int ko, int lane, f32x16& out) {
    f32x16 acc = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int ki = 0; ki < NBK; ++ki)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const _Float16* blk = obh + (size_t)(ko * NBK + ki) * 2048;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(blk + (s * 64 + lane) * 8);
            const f16x8 al = *reinterpret_cast<const f16x8*>(blk + 1024 + (s * 64 + lane) * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wf[ki].hi[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wf[ki].lo[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wf[ki].hi[s], acc, 0, 0, 0);
        }
    const float sc = __builtin_amdgcn_ldexpf(1.0f, e);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r] = acc[r] * sc;
}

// forward of one conditioner from its input (u0, u1) to the second hidden layer's pre-activation triples z2 (two 32-unit blocks).  HEAD: on to the head
// triples o (NBK row blocks).
template <bool HEAD, int NBK>
__device__ __forceinline__ void cond_fwd(const float* net, float u0v, float u1v, int lane, f32x16 (&z2a)[NCH], f32x16 (&z2b)[NCH], f32x16 (&o)[NBK][NCH]) {
    Frag f2[NCH][2];
    int e2[NCH];
    using O = NetOff<2, NBK>;
    const int h = lane >> 5;
    const float in0[2] = {u0v, 1.0f}, in1[2] = {u1v, 0.0f};
    f32x16 a0[NCH], a1[NCH];
    init_acc(a0, net + O::b0 + (0 * 2 + h) * 16);
    init_acc(a1, net + O::b0 + (1 * 2 + h) * 16);
    const float w0 = net[O::W0 + 0 * 64 + lane], w1 = net[O::W0 + 1 * 64 + lane];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        a0[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, h ? in1[c] : in0[c], a0[c], 0, 0, 0);
        a1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, h ? in1[c] : in0[c], a1[c], 0, 0, 0);
    }
    act_block(a0);
    act_block(a1);
    to_frags(a0, a1, f2, e2);
    const _Float16* W1h = reinterpret_cast<const _Float16*>(net + O::W1h);
    const _Float16* W1l = reinterpret_cast<const _Float16*>(net + O::W1l);
    init_acc(z2a, net + O::b1 + (0 + h) * 16);
    init_acc(z2b, net + O::b1 + (2 + h) * 16);
    dense64_block<NCH>(W1h, W1l, f2, z2a, lane);
    dense64_block<NCH>(W1h + 2048, W1l + 2048, f2, z2b, lane);
    unscale(z2a, e2);
    unscale(z2b, e2);
    if (HEAD) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) { a0[c] = z2a[c]; a1[c] = z2b[c]; }
        act_block(a0);
        act_block(a1);
        to_frags(a0, a1, f2, e2);
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb) cond_out<NBK>(net, f2, e2, kb, lane, o[kb]);
    }
}
// X operand of a product over the walker axis: one 32-unit block of ACTIVATION jets (channel 0 = r in (0, 1)) -> fragments scaled by 2^-ex[c], one power
// of two per (tile, channel).  ey[c]: the exponents of the other operand's channels.  All three channels' products are to land in ONE accumulator
// chain, so the channels share the product's exponent E = max_c (natural exponent of X_c + ey[c]) and X_c is scaled by 2^-(E - ey[c]) -- at most its
// natural scale; a channel whose product lies below the largest one's loses bits that the sum does not see.  Returns E.
__device__ __forceinline__ int block_frags_x(const f32x16 (&blk)[NCH], const int (&ey)[NCH], Frag (&f)[NCH]) {
    int en[NCH];
    en[0] = 0;
#pragma unroll
    for (int c = 1; c < NCH; ++c) {
        float amax = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(blk[c][r]));
        en[c] = exponent_of(wave_max(amax));
    }
    const int E = max(en[0] + ey[0], max(en[1] + ey[1], en[2] + ey[2]));
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float sc = __builtin_amdgcn_ldexpf(1.0f, ey[c] - E);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float r8[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) r8[jj] = blk[c][8 * s + jj] * sc;
            split8(r8, f[c].hi[s], f[c].lo[s]);
        }
    }
    return E;
}
__device__ __forceinline__ void mfma3(f32x16& p, const f16x8& xh, const f16x8& xl, const f16x8& yh, const f16x8& yl) {
    p = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, yh, p, 0, 0, 0);
    p = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yl, p, 0, 0, 0);
    p = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, p, 0, 0, 0);
}
// one channel's product block over the 32 walkers of the tile: p += A (x) B (both K steps, three split products each)
__device__ __forceinline__ void wgrad_block(f32x16& p, const Frag& xt, const Frag& yt) {
#pragma unroll
    for (int s = 0; s < 2; ++s) mfma3(p, xt.hi[s], xt.lo[s], yt.hi[s], yt.lo[s]);
}
// block b of the workgroup's accumulators += p * un, in TILE ORDER: the tiles of a workgroup are numbered k = 0, 1, .. (tile = block + k * grid, wave
// k % waves), ticket[b] counts the tiles whose product has been added to block b, and tile k's wave adds when the count stands at k.  The sum of every
// block is therefore formed in the same order whatever the timing (bitwise reproducible gradients) although the waves share ONE set of blocks.  Progress:
// tile k waits only for tile k - 1's wave to pass the same point, tile 0 for nobody; the waves of a workgroup are resident together and each works
// through its tiles in increasing k, so every wait ends (the waves fall into step one add apart: ~200 cycles in a tile of ~10^5).
template <bool SHARED_>
__device__ __forceinline__ void acc_add(float* acc, int* ticket, int b, int k, int lane, const f32x16& p, float un) {
#ifdef WF_ACC_NOTICKET   // timing experiment only (racy sums): what the tile order costs
    constexpr bool SHARED = false;
#else
    constexpr bool SHARED = SHARED_;
#endif
    if (SHARED) {
        // Nothing of the matrix pipe may be in flight across the branch of the wait below.  hipcc (ROCm 7.2) counts the wait states between an MFMA and a
        // vector read of its result correctly in straight-line code, but at the join behind this loop it let v_accvgpr_read follow the product's last
        // MFMA by four instructions where eleven are due (ISA of k_ebwd<true, 2>, seventh wait): when the ticket was already there the last rows of
        // the product (registers 12 .. 15) were read before the pipe had written them -- 128 entries of one gradient block changed from run to run
        // by 4e-5 relative (scratch/r04_repro_diag.py).  24 idle issue slots in front of the branch, fenced against the scheduler, retire every MFMA.
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        while (__hip_atomic_load(ticket + b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != k) __builtin_amdgcn_s_sleep(1);
    }
    f32x16 a = acc_load(acc, b, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = __builtin_fmaf(p[r], un, a[r]);
    acc_store(acc, b, lane, a);
    if (SHARED) __hip_atomic_store(ticket + b, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

}  // namespace
// (k_ebwd has external linkage: its two-row-block instantiations are compiled in a translation unit of their own, wf_etile_bwd_k2.hip -- this file again with
// WF_ETILE_ONLY_K2 -- under the max-ilp scheduling strategy, which is worth 6 % to them and costs the one-row-block form 1 %: DESIGN 4.9)
// -DWF_MARKS: comment lines in the assembly at the phase boundaries of k_ebwd (scratch/r04_spill_phases.py counts the spill traffic per phase)
#ifdef WF_MARKS
#define WF_MARK(name) asm volatile("; WF_MARK " name)
#else
#define WF_MARK(name)
#endif
template <bool PRIOR, int NBK>
__global__ __launch_bounds__(kBwdWaves * 64) void k_ebwd(const MfmaDev mm, int net_index, const float* __restrict__ tabI, const float* __restrict__ tabP,
                                                          const float* __restrict__ st_in, float* __restrict__ adjb, const float* __restrict__ w_psi,
                                                          const float* __restrict__ w_lap, int64_t B, float* __restrict__ partial) {
    // partial: [gridDim.x][GL<NBK>::floats] -- this workgroup's block of the net's gradient (image units), written once at the end
    using O = NetOff<2, NBK>;
    using G = GL<NBK>;
    constexpr int kThreads = kBwdWaves * 64;
    constexpr int kMeshStride = 128 * NBK;   // floats per mesh point of the regrouped tables (k_efused)
    constexpr int kAcc = acc_blocks(NBK);
    constexpr int kSets = acc_sets(NBK);
    constexpr bool kShared = kSets == 1;
    constexpr int kKinds = 4 + 2 * NBK;      // per-lane sums: Gb1 (two unit blocks), Gb2 of dimension 1 (NBK), of dimension 0 (NBK), Gb0, GW0
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int bnd_s[32 * NBK];
    __shared__ int ticket[kAcc];
    __shared__ __attribute__((aligned(16))) float c0s[32 * NBK + 16];   // prior: c of dimension 0 ([NBK][2][16], the same for every walker), + the sum of its raw outputs
    if (threadIdx.x < 16 * NBK) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabI + (size_t)mm.n_mesh * kMeshStride)[threadIdx.x];
    else if (threadIdx.x < 32 * NBK) bnd_s[threadIdx.x] = reinterpret_cast<const int*>(tabP + (size_t)mm.n_mesh * kMeshStride)[threadIdx.x - 16 * NBK];
    if (threadIdx.x >= 64 && threadIdx.x < 64 + kAcc) ticket[threadIdx.x - 64] = 0;
    float* net_l = lds + mm.const_floats;
    float* tnet_l = net_l + mm.net_floats;
    float* tcon_l = tnet_l + mm.tnet_floats;
    stage_floats<kThreads>(mm.image + mm.const_img_off, lds, mm.const_floats);
    stage_floats<kThreads>(mm.image + (size_t)net_index * mm.net_floats, net_l, mm.net_floats);
    stage_floats<kThreads>(mm.image + mm.timg_off + (size_t)net_index * mm.tnet_floats, tnet_l, mm.tnet_floats);
    stage_floats<kThreads>(mm.image + mm.tconst_off, tcon_l, NBK * NBK * 1024);
    // the workgroup's accumulators of (dW1, dW2) behind the images
    float* acc_all = tcon_l + NBK * NBK * 1024;
    for (int i = threadIdx.x; i < kSets * kAcc * 1024; i += kThreads) acc_all[i] = 0.0f;
    __syncthreads();
    float* acc = acc_all + (kShared ? 0 : (threadIdx.x >> 6) * kAcc * 1024);
    // bias / input-layer sums of this lane over its wave's tiles: Gb1 and Gb2 (dimension 1) of unit / row (lane & 31) of its block (transposed operands:
    // partial over the lane half's 16 walkers), Gb2 of dimension 0, Gb0, GW0 in the lane assignment of the DPP sums below
    float gb1[2] = {0.0f, 0.0f}, gb21[NBK], gb20[NBK], gb0s = 0.0f, gw0s = 0.0f;
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) { gb21[kb] = 0.0f; gb20[kb] = 0.0f; }
    f16x8 pm[2];
    make_perm(threadIdx.x & 63, pm);
    const float* net = net_l;
    const float* fkI = lds;
    const float* fkP = lds + 32 * NBK;
    const _Float16* obh = reinterpret_cast<const _Float16*>(lds + 64 * NBK);
    const float* cbP = lds + 64 * NBK + NBK * NBK * 1024 + 64 * NBK;   // [NBK][2][16] constant term of the B prior's boundary map times ob_to_b (mm.p_bias)
    const _Float16* TW1h = reinterpret_cast<const _Float16*>(tnet_l);
    const _Float16* TW1l = reinterpret_cast<const _Float16*>(tnet_l + 2048);
    const _Float16* TW2h = reinterpret_cast<const _Float16*>(tnet_l + 4096);
    const _Float16* TW2l = reinterpret_cast<const _Float16*>(tnet_l + 4096 + 1024 * NBK);
    const float* TW0 = tnet_l + 4096 + 2048 * NBK;
    const _Float16* obT = reinterpret_cast<const _Float16*>(tcon_l);
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const int n_mesh = mm.n_mesh;
    const int64_t n_tiles = (B + 31) >> 5;
    if (PRIOR) {
        // dimension 0 of the prior sees the bias alone (empty mask): c = (b2 * keep) @ ob_to_b (+ the constant term) is the same for every walker
        if (threadIdx.x < 64) {
            f32x16 w0[NBK];
            float s0 = 0.0f;
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb) {
                const f32x16 b20 = load16(net + O::b2 + ((0 * NBK + kb) * 2 + h) * 16), keep = load16(fkP + (kb * 2 + h) * 16);
#pragma unroll
                for (int r = 0; r < 16; ++r) { s0 += b20[r]; w0[kb][r] = b20[r] * keep[r]; }
            }
            s0 = xhalf_sum(s0);
            Frag wf[NBK];
            int e0;
            to_frags_n1<NBK>(w0, wf, e0);
#pragma unroll
            for (int ko = 0; ko < NBK; ++ko) {
                f32x16 c;
                c_block1<NBK>(obh, wf, e0, ko, lane, c);
                if (mm.p_bias) {
                    const f32x16 cbv = load16(cbP + (ko * 2 + h) * 16);
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = __builtin_fmaf(s0, cbv[r], c[r]);
                }
                if (j == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c0s[(ko * 2 + h) * 16 + r] = c[r];
                }
            }
            if (lane == 0) c0s[32 * NBK] = s0;
        }
        __syncthreads();
    }
    // tiles are dealt statically (tile = block + k * grid, k = round * waves + wave): which wave sums which tiles does not depend on timing, and the
    // shared accumulator blocks take the tiles' products in the order of k (acc_add)
    int k = (int)(threadIdx.x >> 6);
    for (int64_t tile = (int64_t)blockIdx.x + (int64_t)(threadIdx.x >> 6) * gridDim.x; tile < n_tiles; tile += (int64_t)kBwdWaves * gridDim.x, k += kBwdWaves) {
        const int64_t w = tile * 32 + j;
        const bool valid = w < B;
        const int64_t wl = valid ? w : B - 1;
        // (padding lanes of the last tile repeat walker B - 1 with zero adjoints: their columns add nothing to the sums over walkers)
        const JA u0 = ja_load(st_in, 0, B, wl), u1 = ja_load(st_in, 1, B, wl);
        WF_MARK("tile_start");
        // ---- the net's forward to the head triples o.  The second hidden layer's pre-activations z2, which the reverse needs, are computed again
        // behind the head: 96 registers less across the head algebra
        f32x16 o[NBK][NCH];
        {
            f32x16 z2a[NCH], z2b[NCH];
            cond_fwd<true, NBK>(net, u0.v, u1.v, lane, z2a, z2b, o);
        }
        WF_MARK("fwd_done");
        // ---- head: forward sums, pullback to adjoint head triples ob (dimension 1) and ob0 (dimension 0, channel 0)
        f32x16 ob[NBK][NCH], ob0[NBK];
        JA u0b = adj::jzero<float>(), u1b = adj::jzero<float>(), ldb = adj::jzero<float>();
        if (!PRIOR) {
            const JA y1b = valid ? ja_load(adjb, 0, B, wl) : adj::jzero<float>(), y0b = valid ? ja_load(adjb, 1, B, wl) : adj::jzero<float>();
            ldb = valid ? ja_load(adjb, 2, B, wl) : adj::jzero<float>();
            const LerpN L1 = nlerp(u1.v, n_mesh), L0 = nlerp(u0.v, n_mesh);
            // the row factors and the biases of dimension 0 (its head sees the bias alone: empty mask).  One row block: loaded once per tile and held (the
            // compiler then also hoists what depends on them alone); two row blocks: loaded where they are used -- holding 64 registers of them across the
            // head costs more than it saves there (1.550 against 1.575 ms per loss + gradient of 2^17 walkers; one block: 0.937 against 0.973 the other way)
            f32x16 g16h[NBK], o0h[NBK];
            if (NBK == 1) {
                g16h[0] = load16(fkI + h * 16);
                o0h[0] = load16(net + O::b2 + h * 16);
            }
            auto g16 = [&](int kb) { return NBK == 1 ? g16h[0] : load16(fkI + (kb * 2 + h) * 16); };
            auto bias0 = [&](int kb) { return NBK == 1 ? o0h[0] : load16(net + O::b2 + ((0 * NBK + kb) * 2 + h) * 16); };
            // dimension 0 first, then dimension 1: the two heads share nothing but the incoming adjoints, and their sums / intermediates need not be live together
            JA sb0 = adj::jzero<float>(), tb0 = adj::jzero<float>();
            float tv0 = 0.0f;
            {
                adj::FlowSumsT<float> s0 = adj::flow_sums_zero<float>();
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) {
                    f32x16 o0[NCH];
                    o0[0] = bias0(kb);
                    flow_rows_ext<true>(s0, o0, g16(kb), tabI, kMeshStride, bnd_s, L0, kb, h);
                }
                flow_sums_xhalf(s0);
                JA y0, dl0;
                const adj::FlowHeadFwd<float> f0 = adj::flow_head_fwd(s0, mm.F_I, mm.i_reg, u0, u0, y0, dl0);
                adj::FlowSumsT<float> ab0 = adj::flow_sums_zero<float>();
                adj::flow_head_bwd(s0, f0, mm.F_I, mm.i_reg, u0, u0, y0b, ldb, ab0, sb0, tb0, tv0);
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) {
                    f32x16 o0[NCH], t0[NCH];
                    o0[0] = bias0(kb);
                    flow_rows_bwd<true>(ab0, o0, g16(kb), tabI, kMeshStride, bnd_s, L0, kb, h, t0);
                    ob0[kb] = t0[0];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            JA sb = adj::jzero<float>(), tb = adj::jzero<float>();
            float tv1 = 0.0f;
            {
                adj::FlowSumsT<float> s1 = adj::flow_sums_zero<float>();
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) flow_rows_ext<false>(s1, o[kb], g16(kb), tabI, kMeshStride, bnd_s, L1, kb, h);
                flow_sums_xhalf(s1);
                JA y1, dl1;
                const adj::FlowHeadFwd<float> f1 = adj::flow_head_fwd(s1, mm.F_I, mm.i_reg, u0, u1, y1, dl1);
                adj::FlowSumsT<float> ab1 = adj::flow_sums_zero<float>();
                adj::flow_head_bwd(s1, f1, mm.F_I, mm.i_reg, u0, u1, y1b, ldb, ab1, sb, tb, tv1);
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) flow_rows_bwd<false>(ab1, o[kb], g16(kb), tabI, kMeshStride, bnd_s, L1, kb, h, ob[kb]);
            }
            u0b = JA{tv0, sb.a + sb0.a + tb0.a, sb.b + sb0.b + tb0.b, sb.h + sb0.h + tb0.h};
            u1b = JA{tv1, tb.a, tb.b, tb.h};
        } else {
            const float wp = valid ? w_psi[wl] : 0.0f, wlp = valid ? w_lap[wl] : 0.0f;
            const JA ld = ja_load(st_in, 2, B, wl);
            const JA psib = JA{wp, 0.0f, 0.0f, 2.0f * wlp};
            // dimension 1: c = (o keep) @ ob_to_b as triples (+ the constant term of a boundary constraint with a non-zero value, mm.p_bias: c += (sum o) *
            // (b @ ob_to_b), channel by channel, as k_efused<.., true>); dimension 0: c0s
            float s1 = 0.0f, sder[2] = {0.0f, 0.0f};
            f32x16 c1[NBK][NCH];
            {
                Frag of[NBK][NCH];
                int eo[NCH];
                prior_frags<NBK>(o, fkP, lane, of, eo, s1, sder);
#pragma unroll
                for (int ko = 0; ko < NBK; ++ko) {
                    prior_c_block<NBK>(obh, of, eo, ko, lane, c1[ko]);
                    if (mm.p_bias) {
                        const f32x16 cbv = load16(cbP + (ko * 2 + h) * 16);
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            c1[ko][0][r] = __builtin_fmaf(s1, cbv[r], c1[ko][0][r]);
                            c1[ko][1][r] = __builtin_fmaf(sder[0], cbv[r], c1[ko][1][r]);
                            c1[ko][2][r] = __builtin_fmaf(sder[1], cbv[r], c1[ko][2][r]);
                        }
                    }
                }
            }
            const float s0 = c0s[32 * NBK];
            const float sg1 = s1 < 0.0f ? -1.0f : 1.0f, sg0 = s0 < 0.0f ? -1.0f : 1.0f;
            const bool in0 = u0.v >= 0.0f && u0.v <= 1.0f, in1 = u1.v >= 0.0f && u1.v <= 1.0f;
            const JA uc0 = in0 ? u0 : JA{u0.v < 0.0f ? 0.0f : 1.0f, 0.0f, 0.0f, 0.0f}, uc1 = in1 ? u1 : JA{u1.v < 0.0f ? 0.0f : 1.0f, 0.0f, 0.0f, 0.0f};
            const LerpN L1 = nlerp(uc1.v, n_mesh), L0 = nlerp(uc0.v, n_mesh);
            adj::PriorSumsT<float> p1 = adj::prior_sums_zero<float>(), p0 = adj::prior_sums_zero<float>();
#pragma unroll
            for (int ko = 0; ko < NBK; ++ko) {
                prior_rows_ext<false>(p1, c1[ko], tabP, kMeshStride, bnd_s + 16 * NBK, L1, ko, h);
                f32x16 c0[NCH];
                c0[0] = load16(c0s + (ko * 2 + h) * 16);
                prior_rows_ext<true>(p0, c0, tabP, kMeshStride, bnd_s + 16 * NBK, L0, ko, h);
            }
            prior_sums_xhalf(p1);
            prior_sums_xhalf(p0);
            JA val1, val0;
            const adj::PriorHeadFwd<float> f1 = adj::prior_head_fwd(p1, sg1, u0, uc1, val1);
            const adj::PriorHeadFwd<float> f0 = adj::prior_head_fwd(p0, sg0, u0, uc0, val0);
            const float sc0 = (mm.constrained_mask & 1u) ? 0.70710678118654752f : 1.0f, sc1 = (mm.constrained_mask & 2u) ? 0.70710678118654752f : 1.0f;
            const float ev = __expf(0.5f * ld.v);
            const JA E = adj::japply(ld, ev, 0.5f * ev, 0.25f * ev);
            const JA A = val0 * sc0, Bv = val1 * sc1, P = adj::jmul(A, Bv);
            JA Pb = adj::jzero<float>(), Eb = adj::jzero<float>(), Ab = adj::jzero<float>(), Bb = adj::jzero<float>();
            adj::jmul_bwd(E, psib, Pb);
            adj::jmul_bwd(P, psib, Eb);
            adj::jfun_bwd(ld, 0.5f * ev, 0.25f * ev, 0.125f * ev, Eb, ldb);
            adj::jmul_bwd(Bv, Pb, Ab);
            adj::jmul_bwd(A, Pb, Bb);
            adj::PriorSumsT<float> ab1 = adj::prior_sums_zero<float>(), ab0 = adj::prior_sums_zero<float>();
            JA sb = adj::jzero<float>(), tb1 = adj::jzero<float>(), sb0 = adj::jzero<float>(), tb0 = adj::jzero<float>();
            float tv1 = 0.0f, tv0 = 0.0f;
            adj::prior_head_bwd(p1, f1, sg1, u0, uc1, Bb * sc1, ab1, sb, tb1, tv1);
            adj::prior_head_bwd(p0, f0, sg0, u0, uc0, Ab * sc0, ab0, sb0, tb0, tv0);
            u0b = JA{in0 ? tv0 : 0.0f, sb.a + sb0.a + (in0 ? tb0.a : 0.0f), sb.b + sb0.b + (in0 ? tb0.b : 0.0f), sb.h + sb0.h + (in0 ? tb0.h : 0.0f)};
            u1b = in1 ? JA{tv1, tb1.a, tb1.b, tb1.h} : adj::jzero<float>();
            // rows back: adjoint of c -> through ob_to_b transposed -> adjoint of the raw outputs (the constant term reaches every one of a channel's
            // raw outputs through their sum: sbar)
            {
                f32x16 cb[NBK][NCH];
                float sbar[NCH] = {0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int ko = 0; ko < NBK; ++ko) {
                    prior_rows_bwd<false>(ab1, c1[ko], tabP, kMeshStride, bnd_s + 16 * NBK, L1, ko, h, cb[ko]);
                    if (mm.p_bias) {
                        const f32x16 cbv = load16(cbP + (ko * 2 + h) * 16);
#pragma unroll
                        for (int c = 0; c < NCH; ++c)
#pragma unroll
                            for (int r = 0; r < 16; ++r) sbar[c] = __builtin_fmaf(cb[ko][c][r], cbv[r], sbar[c]);
                    }
                }
                if (mm.p_bias) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) sbar[c] = xhalf_sum(sbar[c]);
                }
                Frag fcb[NBK][NCH];
                int ecb[NCH];
                to_frags_n<NBK>(cb, fcb, ecb);
#pragma unroll
                for (int ki = 0; ki < NBK; ++ki) {
                    f32x16 wb[NCH];
                    prior_c_block<NBK>(obT, fcb, ecb, ki, lane, wb);
                    const f32x16 keep = load16(fkP + (ki * 2 + h) * 16);
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ob[ki][c][r] = __builtin_fmaf(wb[c][r], keep[r], sbar[c]);
                }
            }
            {
                f32x16 cb0[NBK];
                float sbar = 0.0f;
#pragma unroll
                for (int ko = 0; ko < NBK; ++ko) {
                    f32x16 c0[NCH], t[NCH];
                    c0[0] = load16(c0s + (ko * 2 + h) * 16);
                    prior_rows_bwd<true>(ab0, c0, tabP, kMeshStride, bnd_s + 16 * NBK, L0, ko, h, t);
                    cb0[ko] = t[0];
                    if (mm.p_bias) {
                        const f32x16 cbv = load16(cbP + (ko * 2 + h) * 16);
#pragma unroll
                        for (int r = 0; r < 16; ++r) sbar = __builtin_fmaf(cb0[ko][r], cbv[r], sbar);
                    }
                }
                if (mm.p_bias) sbar = xhalf_sum(sbar);
                Frag f0b[NBK];
                int e0b;
                to_frags_n1<NBK>(cb0, f0b, e0b);
#pragma unroll
                for (int ki = 0; ki < NBK; ++ki) {
                    f32x16 wb0;
                    c_block1<NBK>(obT, f0b, e0b, ki, lane, wb0);
                    const f32x16 keep = load16(fkP + (ki * 2 + h) * 16);
#pragma unroll
                    for (int r = 0; r < 16; ++r) ob0[ki][r] = __builtin_fmaf(wb0[r], keep[r], sbar);
                }
            }
        }
        WF_MARK("head_done");
        // Gb2 of dimension 0: sum over the tile's walkers of obar0 (16 registers per half: DPP sums; lane (j, h) keeps register j & 15 where j < 16)
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb) {
            float sb = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a = half32_sum(ob0[kb][r]);
                sb = (j & 15) == r ? a : sb;
            }
            gb20[kb] += sb;
        }
        // ---- conditioner, reverse: hbar2 = W2' obar, zbar2 = act'(z2) hbar2, hbar1 = W1' zbar2, zbar1 = act'(z1) hbar1
        __builtin_amdgcn_sched_barrier(0);
        f32x16 z2a[NCH], z2b[NCH];
        Frag f[NCH][2];       // fragments of the tensor the next product contracts: obar ([channel][row block]), then zbar2 ([channel][unit block])
        int e[NCH];
        to_frags_kb<NBK>(ob, f, e);
        {
            f32x16 o2[NBK][NCH];
            cond_fwd<false, NBK>(net, u0.v, u1.v, lane, z2a, z2b, o2);
        }
        WF_MARK("refwd_done");
        // dW2[k][row] = sum_c sum_w X2_c[k][w] obar_c[row][w]: X2 = act(z2), block by block (32 units: 48 registers of fragments at a time); both operands
        // transposed on the matrix cores; the 2 x NBK blocks of the product go to the accumulator blocks 4 + (k block) NBK + (row block).  Gb2 rides on
        // obar's transposes.
        {
            Frag yt[NBK][NCH];      // obar^T, once for both k blocks
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb) {
                float rs = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) tr_frag(f[c][kb], pm, yt[kb][c], c == 0 ? &rs : nullptr);
                gb21[kb] = __builtin_fmaf(rs, __builtin_amdgcn_ldexpf(1.0f, e[0]), gb21[kb]);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                f32x16 t[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) t[c] = mb ? z2b[c] : z2a[c];
                act_block(t);
                Frag fx[NCH];
                const int E = block_frags_x(t, e, fx);
                f32x16 p[NBK];
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) p[kb] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    Frag xt;
                    tr_frag(fx[c], pm, xt);
#pragma unroll
                    for (int kb = 0; kb < NBK; ++kb) wgrad_block(p[kb], xt, yt[kb][c]);
                }
#pragma unroll
                for (int kb = 0; kb < NBK; ++kb) acc_add<kShared>(acc, ticket, 4 + mb * NBK + kb, k, lane, p[kb], __builtin_amdgcn_ldexpf(1.0f, E));
            }
        }
        WF_MARK("dW2_done");
        f32x16 g0[NCH], g1[NCH];
        {
#pragma unroll
            for (int c = 0; c < NCH; ++c) { g0[c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; g1[c] = g0[c]; }
#pragma unroll
            for (int kt = 0; kt < NBK; ++kt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mfma_step<NCH>(TW2h, TW2l, kt, s, f, g0, lane);
                    mfma_step<NCH>(TW2h + NBK * 1024, TW2l + NBK * 1024, kt, s, f, g1, lane);
                }
            unscale_all(g0, e);
            unscale_all(g1, e);
            act_block_bwd(z2a, g0);
            act_block_bwd(z2b, g1);
            to_frags_all<true>(g0, g1, f, e);
            WF_MARK("zbar2_done");
            // dW1[k][u] = sum_c sum_w X1_c[k][w] zbar2_c[u][w] while the fragments of zbar2 (f, exponents e) are at hand and before the product that
            // consumes them: X1, the first hidden layer's activation triples, is recomputed block by block from (s, 1, 0) (two f32 MFMAs and 16
            // activations per lane and block).  Accumulator blocks 0 .. 3 = (k block mb, u block nb); Gb1 rides on the transposes of zbar2.
            {
                const float in0[2] = {u0.v, 1.0f}, in1[2] = {u1.v, 0.0f};
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    f32x16 t[NCH];
                    init_acc(t, net + O::b0 + (mb * 2 + h) * 16);
                    const float w0 = net[O::W0 + mb * 64 + lane];
#pragma unroll
                    for (int c = 0; c < 2; ++c) t[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, h ? in1[c] : in0[c], t[c], 0, 0, 0);
                    act_block(t);
                    Frag fx[NCH];
                    const int E = block_frags_x(t, e, fx);
                    Frag xt[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) tr_frag(fx[c], pm, xt[c]);
                    const float un = __builtin_amdgcn_ldexpf(1.0f, E);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        f32x16 p = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            Frag yt;
                            float rs = 0.0f;
                            const bool bias = c == 0 && mb == 0;
                            tr_frag(f[c][nb], pm, yt, bias ? &rs : nullptr);
                            wgrad_block(p, xt[c], yt);
                            if (bias) gb1[nb] = __builtin_fmaf(rs, __builtin_amdgcn_ldexpf(1.0f, e[0]), gb1[nb]);
                        }
                        acc_add<kShared>(acc, ticket, 2 * mb + nb, k, lane, p, un);
                    }
                }
            }
            WF_MARK("dW1_done");
#pragma unroll
            for (int c = 0; c < NCH; ++c) { g0[c] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; g1[c] = g0[c]; }
            dense64_block<NCH>(TW1h, TW1l, f, g0, lane);
            dense64_block<NCH>(TW1h + 2048, TW1l + 2048, f, g1, lane);
            unscale_all(g0, e);
            unscale_all(g1, e);
        }
        {
            const float in0[2] = {u0.v, 1.0f}, in1[2] = {u1.v, 0.0f};
            f32x16 a0[NCH], a1[NCH];
            init_acc(a0, net + O::b0 + (0 * 2 + h) * 16);
            init_acc(a1, net + O::b0 + (1 * 2 + h) * 16);
            const float w0 = net[O::W0 + 0 * 64 + lane], w1 = net[O::W0 + 1 * 64 + lane];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                a0[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, h ? in1[c] : in0[c], a0[c], 0, 0, 0);
                a1[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, h ? in1[c] : in0[c], a1[c], 0, 0, 0);
            }
            act_block_bwd(a0, g0);
            act_block_bwd(a1, g1);
        }
        // input layer: Gb0[u] = sum_w zbar1_0[u][w], GW0[u] = sum_w zbar1_0[u][w] s_w + zbar1_1[u][w] (seed of the conditioner's input: (s, 1, 0)), summed
        // over the tile's walkers here (the lanes of a half); lane (j, h) keeps the sums of register j & 15 of block j >> 4
        {
            float sb = 0.0f, sw = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float a0 = half32_sum(g0[0][r]), a1 = half32_sum(g1[0][r]);
                const float b0 = half32_sum(__builtin_fmaf(g0[0][r], u0.v, g0[1][r])), b1 = half32_sum(__builtin_fmaf(g1[0][r], u0.v, g1[1][r]));
                const bool mine = (j & 15) == r;
                sb = mine ? (j < 16 ? a0 : a1) : sb;
                sw = mine ? (j < 16 ? b0 : b1) : sw;
            }
            gb0s += sb;
            gw0s += sw;
        }
        {
            const f32x16 wa = load16(TW0 + (0 * 2 + h) * 16), wb2 = load16(TW0 + (1 * 2 + h) * 16);
            float sbar = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sbar = __builtin_fmaf(wa[r], g0[0][r], __builtin_fmaf(wb2[r], g1[0][r], sbar));
            u0b.v += xhalf_sum(sbar);
        }
        WF_MARK("tile_end");
        if (valid && h == 0) {
            ja_store(adjb, 0, B, w, u0b);
            ja_store(adjb, 1, B, w, u1b);
            ja_store(adjb, 2, B, w, ldb);
        }
    }
    // ---- this workgroup's block of the net's gradient: the accumulator blocks as they stand, the per-lane sums added over the waves in wave order
    __syncthreads();                       // every wave is done with its tiles: the operand images are dead, the accumulators complete
    float* red = lds;                      // [wave][kKinds][64 lanes] over the image area
    {
        float* mine = red + (threadIdx.x >> 6) * kKinds * 64 + lane;
        mine[0] = gb1[0]; mine[64] = gb1[1];
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb) { mine[(2 + kb) * 64] = gb21[kb]; mine[(2 + NBK + kb) * 64] = gb20[kb]; }
        mine[(2 + 2 * NBK) * 64] = gb0s; mine[(3 + 2 * NBK) * 64] = gw0s;
    }
    __syncthreads();
    float* g = partial + (size_t)blockIdx.x * G::floats;
    auto wsum = [&](int kind, int ln) {    // sum over the waves of a lane's value
        float a = 0.0f;
#pragma unroll
        for (int wv = 0; wv < kBwdWaves; ++wv) a += red[(wv * kKinds + kind) * 64 + ln];
        return a;
    };
    for (int i = threadIdx.x; i < kAcc * 1024; i += kThreads) {
        const int b = i >> 10, q = (i >> 8) & 3, ln = (i >> 2) & 63, r = 4 * q + (i & 3);
        const int row = acc_rho(r, ln >> 5), n = ln & 31;
        float a = 0.0f;
#pragma unroll
        for (int st = 0; st < kSets; ++st) a += acc_all[st * kAcc * 1024 + i];
        if (b < 4) g[G::W1 + (32 * (b >> 1) + row) * 64 + 32 * (b & 1) + n] = a;
        else g[G::W2 + (32 * ((b - 4) / NBK) + row) * (32 * NBK) + 32 * ((b - 4) % NBK) + n] = a;
    }
    for (int i = threadIdx.x; i < 128 + 64 * NBK; i += kThreads) {
        if (i < 64) {                      // Gb1[u]: u block = i >> 5; the two lane halves hold the two halves of the tile's walkers
            const int nb = i >> 5, n = i & 31;
            g[G::b1 + i] = wsum(nb, n) + wsum(nb, n + 32);
        } else if (i < 64 + 32 * NBK) {    // Gb2 of dimension 1
            const int t = i - 64, kb = t >> 5, n = t & 31;
            g[G::b21 + t] = wsum(2 + kb, n) + wsum(2 + kb, n + 32);
        } else if (i < 64 + 64 * NBK) {    // Gb2 of dimension 0: lane (j < 16, h) keeps register j of half h
            const int t = i - 64 - 32 * NBK, kb = t >> 5, jj = t & 15, hh = (t >> 4) & 1;
            g[G::b20 + 32 * kb + acc_rho(jj, hh)] = wsum(2 + NBK + kb, jj + 32 * hh);
        } else {                           // Gb0 / GW0: lane (j, h) keeps register j & 15 of block j >> 4
            const int ln = i - 64 - 64 * NBK, jj = ln & 31, hh = ln >> 5;
            const int u = 32 * (jj >> 4) + acc_rho(jj & 15, hh);
            g[G::b0 + u] = wsum(2 + 2 * NBK, ln);
            g[G::W0 + u] = wsum(3 + 2 * NBK, ln);
        }
    }
}

#ifdef WF_ETILE_ONLY_K2
template __global__ void k_ebwd<true, 2>(const MfmaDev, int, const float*, const float*, const float*, float*, const float*, const float*, int64_t, float*);
template __global__ void k_ebwd<false, 2>(const MfmaDev, int, const float*, const float*, const float*, float*, const float*, const float*, int64_t, float*);
#else
extern template __global__ void k_ebwd<true, 2>(const MfmaDev, int, const float*, const float*, const float*, float*, const float*, const float*, int64_t, float*);
extern template __global__ void k_ebwd<false, 2>(const MfmaDev, int, const float*, const float*, const float*, float*, const float*, const float*, int64_t, float*);
#endif
namespace {
// (one launch for the nets of a chunk: blockIdx.y = net; partial [n_nets][kESplit][gf] of which the first n_part blocks are live, gacc [n_nets][gf];
// gf = g_floats(row blocks of the model))
__global__ void k_egrad_reduce(const float* __restrict__ partial, int n_part, int accumulate, float* __restrict__ gacc, int gf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= gf) return;
    const float* pn = partial + (size_t)blockIdx.y * kESplit * gf;
    float* gn = gacc + (size_t)blockIdx.y * gf;
    float sacc = accumulate ? gn[i] : 0.0f;
    int p = 0;
    for (; p + 8 <= n_part; p += 8) {   // eight loads in flight, added in block order (a runtime trip count alone left one dependent load per ~230 ns: 60 us)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = pn[(size_t)(p + k) * gf + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc += v[k];
    }
    for (; p < n_part; ++p) sacc += pn[(size_t)p * gf + i];
    gn[i] = sacc;
}
struct ENetOff {
    int W0, b0, W1, b1, W2, b2, NO, n_out;
    float c2;   // scale of the head's pre-activation: -log2(e) under a sigmoid head, 1 otherwise
};
struct ENetOffs {
    ENetOff n[8];
};
__global__ void k_fill_zero(float* __restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}
// image units -> the reference's leaves: scales of describe_mfma_image, and the column sums folded into the biases behind a tanh (k_fold_bias)
__global__ void k_egrad_scatter(const float* __restrict__ gacc, int n_nets, const ENetOffs offs, float* __restrict__ flat, int nbk) {
    const int gf = g_floats(nbk), rows = 32 * nbk;
    const int W1 = 128, b1 = 4224, W2 = 4288, b21 = W2 + 64 * rows, b20 = b21 + rows;   // (GL<nbk>)
    const int net = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (net >= n_nets || i >= gf) return;
    const ENetOff q = offs.n[net];
    const float* g = gacc + (size_t)net * gf;
    const float c1 = 2.8853900817779268f;
    if (i < 64) flat[q.W0 + i] = c1 * g[i];                                          // W0[0][u]
    else if (i < 128) flat[q.b0 + (i - 64)] = c1 * g[i];
    else if (i < b1) { const int e = i - W1, u = e & 63; flat[q.W1 + e] = -2.0f * c1 * g[i] + c1 * g[b1 + u]; }
    else if (i < W2) flat[q.b1 + (i - b1)] = c1 * g[i];
    else if (i < b21) {
        const int e = i - W2, k = e / rows, jb = e % rows;
        if (jb < q.n_out) flat[q.W2 + k * q.NO + (jb * 2 + 1)] = -2.0f * q.c2 * g[i] + q.c2 * g[b21 + jb];
    } else if (i < b20) { const int jb = i - b21; if (jb < q.n_out) flat[q.b2 + jb * 2 + 1] = q.c2 * g[i]; }
    else { const int jb = i - b20; if (jb < q.n_out) flat[q.b2 + jb * 2 + 0] = q.c2 * g[i]; }
}

// ============================================================================ inverse / sampler of large batches (two-particle family)
// Serial.inverse_fun / the Waveflow prior's sample_fun (made.py:85-100, bsplines_jax.py:144-171) for batches the one-walker-per-wave kernel
// (wf_kernels_wave.hip: 6.8e7 walkers/s, a chain of dependent reads per walker) is too slow for.  Staged: the conditioner of a net runs on the
// matrix cores for the whole batch (k_etile_cond: head outputs to HBM, 384 B per walker), everything else is one lane per walker (k_tsample):
//   dimension 0 of a net does not depend on the walker: its spline is the composite table comp[net] (k_prepare_dim0), inverted by a binary
//     search over the mesh; the prior's first column is drawn by rejection under the table's own maximum (tight: the lerp of P is piecewise
//     linear, so P^2 peaks at a mesh point)
//   dimension 1: coefficients c_j = g_j (v_j / S0 + reg) / Q from the head outputs (as k_etile_flow), the spline sum_j c_j I_j inverted by the
//     same search with 32-term row sums; the prior's second column by rejection from a piecewise-constant envelope over the knot intervals
//     (the largest of the k + 1 B-spline coefficients alive on an interval: k_tsample, phase 1) -- the reference proposes uniformly under the
//     global bound max_i ((e @ b_to_ob)_i)^2: the same law at 2.5 x the acceptance rate
//   the root of a search is rounded to the reference's halving grid exactly as wf_kernels_wave.hip: ispline_inverse does (largest grid point
//     whose table-lerp value does not exceed y)
// Streams: Philox4x32-10 keyed by (seed, walker), proposal n of column col uses counter (n, col + 1) -- as the wave sampler; the draws differ
// from that kernel's (other proposal sequences), their law does not.
struct Philox4 {   // (as scalar::Philox, wf_scalar_impl.h)
    unsigned key0, key1, c0, c1, c2, c3;
    unsigned out[4];
    int have;
    __device__ Philox4(unsigned long long seed, unsigned long long stream) : key0((unsigned)seed), key1((unsigned)(seed >> 32)), c0(0), c1(0), c2((unsigned)stream), c3((unsigned)(stream >> 32)), have(0) {}
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
    __device__ float uniform() {
        if (!have) refill();
        return (float)(out[--have] >> 8) * (1.0f / 16777216.0f);
    }
};
struct TsArgs {
    const float4_t* comp;      // [n_nets][n_mesh] composite tables of dimension 0 (flow nets: Y; prior: P with sign and norm)
    const float* tabI0;        // order-0 rows of the I-spline table [n_mesh][32]
    const float* tabP0;        // order-0 rows of the prior's table [n_mesh][32]
    const float* gI;           // [32] row factors of the flow heads (boundary map; 0 beyond the bases)
    const float* b_to_ob;      // [32][32]
    const float* tabB0;        // plain B-splines of the prior, order 0 [n_mesh][NB] (with ow: the band-limited evaluation of a proposal), or null
    const float* ow;           // the prior's o * keep of the conditioner launch ([tile][row][32 walkers]) where they are the plain B-spline coefficients of c, or null
    int n_mesh, nbI, nbP, n_layers, degP;
    int i_band_int;            // > 0: knot intervals of the I-splines, and their rows are plain (exactly 1 left of a band of k + 1 <= 8 rows, 0 right of it): the band form of phase 2
    float i_reg, tol, box_L;
    unsigned long long seed;
    const unsigned long long* seed_offset_dev;
    int exact;
    int64_t b0;                // index of the chunk's first walker in the batch (the key of a walker's stream is its index in the batch)
};
// largest mesh point m with F(m) <= y (0 if there is none), F monotone on the mesh; yl = F(m), yr = F(m + 1) (yr = yl at the last point)
template <class F>
__device__ __forceinline__ void mesh_search(F f, int last, float y, int& m, float& yl, float& yr) {
    int lo = 0, hi = last;
    float flo = f(0), fhi = f(last);
    const bool beyond = fhi <= y;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const float fm = f(mid);
        if (fm <= y) { lo = mid; flo = fm; } else { hi = mid; fhi = fm; }
    }
    m = beyond ? last : lo;
    yl = beyond ? fhi : flo;
    yr = fhi;
}
// the root on the line through the two mesh values, rounded down to the halving grid 2^-K of helpers.binary_search; the table lerp itself decides
// between the neighbouring grid points (wf_kernels_wave.hip: ispline_inverse)
template <class FL>
__device__ __forceinline__ float grid_root(FL flerp, int m, float yl, float yr, float y, int last, float tol) {
    const float n = (float)last;
    float xs = (float)m / n;
    if (yr > yl) xs = xs + (y - yl) / ((yr - yl) * n);
    int K = 0;
    float w = 1.0f;
    while (K < 64 && w * 0.5f > tol * 0.5f) { w *= 0.5f; ++K; }
    const float scale = ldexpf(1.0f, K);
    float q = floorf(xs * scale);
    q = fminf(fmaxf(q, 0.0f), scale - 1.0f);
    const float f_lo = flerp(q / scale) - y, f_hi = flerp(fminf(q + 1.0f, scale - 1.0f) / scale) - y;
    if (f_hi <= 0.0f && q + 1.0f <= scale - 1.0f) q = q + 1.0f;
    else if (f_lo > 0.0f && q >= 1.0f) q = q - 1.0f;
    return q / scale;
}
__device__ __forceinline__ float comp_lerp_x(const float4_t* __restrict__ comp, float x, int n_mesh) {
    const LerpN L = nlerp(x, n_mesh);
    const float a = comp[L.il].x, b = comp[L.ir].x;
    return __builtin_fmaf(b - a, L.t, a);
}
__device__ __forceinline__ float inv_comp(const float4_t* __restrict__ comp, int n_mesh, float y, float tol) {
    int m;
    float yl, yr;
    mesh_search([&](int i) { return comp[i].x; }, n_mesh - 1, y, m, yl, yr);
    return grid_root([&](float x) { return comp_lerp_x(comp, x, n_mesh); }, m, yl, yr, y, n_mesh - 1, tol);
}
template <int NB>
__device__ __forceinline__ float rows_dot(const float* __restrict__ row, const float (&c)[NB]) {   // sum_j c_j row[j], j ascending
    const float4_t* r4 = reinterpret_cast<const float4_t*>(row);
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < NB / 4; ++q) {
        const float4_t t = r4[q];
        acc = __builtin_fmaf(c[4 * q], t.x, acc);
        acc = __builtin_fmaf(c[4 * q + 1], t.y, acc);
        acc = __builtin_fmaf(c[4 * q + 2], t.z, acc);
        acc = __builtin_fmaf(c[4 * q + 3], t.w, acc);
    }
    return acc;
}
template <int NB>
__device__ __forceinline__ float rows_lerp(const float* __restrict__ tab0, const float (&c)[NB], float x, int n_mesh) {
    const LerpN L = nlerp(x, n_mesh);
    const float4_t* ra = reinterpret_cast<const float4_t*>(tab0 + (size_t)L.il * NB);
    const float4_t* rb = reinterpret_cast<const float4_t*>(tab0 + (size_t)L.ir * NB);
    float acc = 0.0f;
#ifdef WF_TS_FAKE_BAND   // timing experiment only (wrong values): three of the NB / 4 records per row, as a band-limited evaluation would read
    constexpr int kQ = 3;
#else
    constexpr int kQ = NB / 4;
#endif
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
        const float4_t a = ra[q], b = rb[q];
        acc = __builtin_fmaf(c[4 * q], __builtin_fmaf(b.x - a.x, L.t, a.x), acc);
        acc = __builtin_fmaf(c[4 * q + 1], __builtin_fmaf(b.y - a.y, L.t, a.y), acc);
        acc = __builtin_fmaf(c[4 * q + 2], __builtin_fmaf(b.z - a.z, L.t, a.z), acc);
        acc = __builtin_fmaf(c[4 * q + 3], __builtin_fmaf(b.w - a.w, L.t, a.w), acc);
    }
    return acc;
}
template <int NB>
__device__ __forceinline__ float inv_rows(const float* __restrict__ tab0, const float (&c)[NB], int n_mesh, float y, float tol) {
    int m;
    float yl, yr;
    mesh_search([&](int i) { return rows_dot<NB>(tab0 + (size_t)i * NB, c); }, n_mesh - 1, y, m, yl, yr);
    return grid_root([&](float x) { return rows_lerp<NB>(tab0, c, x, n_mesh); }, m, yl, yr, y, n_mesh - 1, tol);
}
// ... with plain I-spline rows (TsArgs::i_band_int): at mesh point i only the rows s .. s + k of knot interval s = floor(x_i n_int) are neither 1 nor 0, so
// sum_j c_j T[i][j] = (sum of the c_j left of a window of 12 rows from a multiple of four) + (the window's terms): three 16-byte records per row instead of
// NB / 4, in the order of the full sum -- the same bits (a row of ones adds c_j, a row of zeros nothing).  cs: the walker's coefficients, p4: their prefix
// sums at the multiples of four, both in LDS (the window moves with the mesh point).  A lerp between neighbouring mesh points needs s .. s + k + 1: k <= 7.
template <int NB>
__device__ __forceinline__ float inv_rows_band(const float* __restrict__ tab0, const float* cs, const float* p4, int n_int, int n_mesh, float y, float tol) {
    auto window = [&](int i) { return min(min((i * n_int) / (n_mesh - 1), n_int - 1) & ~3, NB - 12); };
    auto dot_at = [&](int i) {
        const int a0 = window(i);
        const float4_t* r = reinterpret_cast<const float4_t*>(tab0 + (size_t)i * NB + a0);
        const float4_t* cq = reinterpret_cast<const float4_t*>(cs + a0);
        float acc = p4[a0 >> 2];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4_t t = r[q], c = cq[q];
            acc = __builtin_fmaf(c.x, t.x, acc);
            acc = __builtin_fmaf(c.y, t.y, acc);
            acc = __builtin_fmaf(c.z, t.z, acc);
            acc = __builtin_fmaf(c.w, t.w, acc);
        }
        return acc;
    };
    auto lerp_at = [&](float x) {
        const LerpN L = nlerp(x, n_mesh);
        const int a0 = window(min(L.il, L.ir));
        const float4_t* ra = reinterpret_cast<const float4_t*>(tab0 + (size_t)L.il * NB + a0);
        const float4_t* rb = reinterpret_cast<const float4_t*>(tab0 + (size_t)L.ir * NB + a0);
        const float4_t* cq = reinterpret_cast<const float4_t*>(cs + a0);
        float acc = p4[a0 >> 2];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4_t ta = ra[q], tb = rb[q], c = cq[q];
            acc = __builtin_fmaf(c.x, __builtin_fmaf(tb.x - ta.x, L.t, ta.x), acc);
            acc = __builtin_fmaf(c.y, __builtin_fmaf(tb.y - ta.y, L.t, ta.y), acc);
            acc = __builtin_fmaf(c.z, __builtin_fmaf(tb.z - ta.z, L.t, ta.z), acc);
            acc = __builtin_fmaf(c.w, __builtin_fmaf(tb.w - ta.w, L.t, ta.w), acc);
        }
        return acc;
    };
    int m;
    float yl, yr;
    mesh_search(dot_at, n_mesh - 1, y, m, yl, yr);
    return grid_root(lerp_at, m, yl, yr, y, n_mesh - 1, tol);
}
// channel 0 of the head outputs of walker b: oj[tile][row 0 .. NB)[channel][32 walkers]
template <int NB>
__device__ __forceinline__ float oj0(const float* __restrict__ oj, int64_t b, int row) { return oj[((b >> 5) * NB + row) * 32 + (b & 31)]; }   // (k_etile_cond<., ., 1>: the value channel alone)

// phase 0: prior column 0;  1: prior column 1, then the last layer's dimension 0;  2: dimension 1 of layer `layer`, then dimension 0 of the layer before it
// (layer 0: the box reverse and the result);  3: entry of a plain inverse (latent given): the last layer's dimension 0.
// Between the phases: cur0 = the inverted dimension 0, cur1 = the value waiting for dimension 1, cin = what the conditioner of the next launch sees
// (exact: the inverted prefix; reference mode, made.py:88: the value being inverted).
// NB: padded bases per dimension (32, or 64: two row blocks -- there the rejection loop of the second column stays on the walker's own lane)
template <int PHASE, int NB>
__global__ __launch_bounds__(256, 2) void k_tsample(const TsArgs a, int layer, const float* __restrict__ oj, const float* __restrict__ ug, int64_t B,
                                                 float* __restrict__ cur0, float* __restrict__ cur1, float* __restrict__ cin, float* __restrict__ lat,
                                                 float* __restrict__ latent_out, float* __restrict__ xg) {
    __shared__ float red[256];
    // phase 1, band form: the plain B-spline coefficients q of every walker of the workgroup, one row per lane (+ 4: rows stay 16-byte aligned and
    // fall on different banks)
    constexpr int kQStride = NB + 4;
    constexpr bool kBand2 = PHASE == 2;   // phase 2, band form (inv_rows_band): the walker's spline coefficients and their prefix sums (two row blocks: 90 KB, one
                                                      // workgroup per CU instead of two -- and still 0.537 -> 0.450 ms per 2^17 draws of the 33-knot model)
    __shared__ __attribute__((aligned(16))) float qs[(PHASE == 1 || kBand2) ? 256 * kQStride : 4];
    __shared__ __attribute__((aligned(16))) float p4s[kBand2 ? 256 * (NB / 4 + 4) : 4];
    constexpr int phase = PHASE;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int n_mesh = a.n_mesh;
    unsigned long long seed = a.seed;
    if (a.seed_offset_dev) seed += *a.seed_offset_dev * 0x9E3779B97F4A7C15ull;
    // phase 0: piecewise-constant envelope of the first column's density P^2 (the same for every walker): kBins equal bins of [0, 1], M_j = the largest P^2 at
    // the mesh points of the cells that meet bin j (the lerp of P is piecewise linear: P^2 peaks at a mesh point), red[j] = M_j, red[kBins + j] = sum_{i < j} M_i.
    // Round 3 proposed uniformly under the global maximum: the same law at a fifth of the acceptance rate (34 us of the sampler's 228 at 2^17 walkers).
    constexpr int kBins = 64;
    if (phase == 0) {
        const float4_t* cp = a.comp + (size_t)a.n_layers * n_mesh;
        if (threadIdx.x < kBins) {
            const int j = threadIdx.x;
            const int m0 = max((int)floorf((float)j / kBins * (float)(n_mesh - 1)) - 1, 0), m1 = min((int)ceilf((float)(j + 1) / kBins * (float)(n_mesh - 1)) + 1, n_mesh - 1);
            float mx = 0.0f;
            for (int i = m0; i <= m1; ++i) { const float pv = cp[i].x; mx = fmaxf(mx, pv * pv); }
            red[j] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float run = 0.0f;
            for (int j = 0; j < kBins; ++j) { red[kBins + j] = run; run += red[j]; }
            red[2 * kBins] = run;
        }
        __syncthreads();
    }
    if (b >= B) return;
    // the next layer's dimension 0 from the pair (va, vb) that leaves a layer (or the prior): Reverse.inverse_fun, then the composite table
    auto start_layer = [&](int l, float va, float vb) {
        const float n0 = vb, n1 = va;
        const float o0 = inv_comp(a.comp + (size_t)l * n_mesh, n_mesh, n0, a.tol);
        cur0[b] = o0;
        cur1[b] = n1;
        cin[b] = a.exact ? o0 : n0;
    };
    if (phase == 0) {
        const float4_t* cp = a.comp + (size_t)a.n_layers * n_mesh;
        const float tot = red[2 * kBins];
        float xs = __builtin_nanf("");
        for (int n = 0; n < 100000; ++n) {
            Philox4 prop(seed, (unsigned long long)(a.b0 + b));
            prop.c0 = (unsigned)n;
            prop.c1 = 1u;
            const float t = prop.uniform() * tot, u2 = prop.uniform();
            int j = 0;      // the last bin whose prefix sum does not exceed t
#pragma unroll
            for (int step = kBins / 2; step > 0; step >>= 1) j = red[kBins + j + step] <= t ? j + step : j;
            const float mj = red[j];
            if (!(mj > 0.0f)) continue;
            const float xc = fminf(((float)j + fminf((t - red[kBins + j]) / mj, 1.0f)) * (1.0f / kBins), 0.99999994f);
            const float p = comp_lerp_x(cp, xc, n_mesh);
            if (u2 * mj < p * p) { xs = xc; break; }
        }
        lat[b] = xs;
        cin[b] = xs;
        cin[4 * B + b] = 0.0f;
        return;
    }
    if (phase == 3) {
        cin[4 * B + b] = 0.0f;
        start_layer(a.n_layers - 1, ug[b * 2], ug[b * 2 + 1]);
        return;
    }
    if (phase == 1) {
        // e = c / |c| with c = (o keep) @ ob_to_b from the conditioner launch; bound max_i ((e @ b_to_ob)_i)^2 (bsplines_jax.py:164-166)
        float e[NB];
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < NB; ++j) { e[j] = j < a.nbP ? oj0<NB>(oj, b, j) : 0.0f; ss = __builtin_fmaf(e[j], e[j], ss); }
        const float rn = 1.0f / sqrtf(ss);
#pragma unroll
        for (int j = 0; j < NB; ++j) e[j] = e[j] * rn;
        // q = e @ b_to_ob are the coefficients of this column's factor f = sum_i q_i b_i in the plain B-splines (non-negative, summing to one), so
        // |f| <= max_i |q_i| (the reference's bound, bsplines_jax.py:164-166) and, on the knot interval s where only b_s .. b_{s + k} live,
        // |f| <= M_s = max(|q_s| .. |q_{s + k}|).  Proposals are drawn from the piecewise-constant envelope M_s^2 (one uniform picks the interval and the
        // point in it) and accepted against M_s^2: the same law as the reference's uniform proposals under the global bound, at 5 - 8 x its acceptance rate.
        float aq[NB];
        const bool band = a.ow != nullptr && a.tabB0 != nullptr;
        if (a.ow) {   // (the boundary map only zeroes coefficients: q = e @ b_to_ob = (o keep) / |c|, the product is the identity; round 4)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float qi = i < a.nbP ? oj0<NB>(a.ow, b, i) * rn : 0.0f;
                aq[i] = qi * qi;
                if (PHASE == 1) qs[threadIdx.x * kQStride + i] = qi;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < NB; ++j) acc = __builtin_fmaf(e[j], a.b_to_ob[j * NB + i], acc);
                aq[i] = i < a.nbP ? acc * acc : 0.0f;
            }
        }
        const int n_int = a.nbP - a.degP;       // knot intervals of equal width on [0, 1] (knots: linspace, the end knots (k + 1)-fold)
        float msq[NB], tot = 0.0f;
#pragma unroll
        for (int sI = 0; sI < NB; ++sI) {
            float mx = 0.0f;
#pragma unroll
            for (int d = 0; d <= 8; ++d)
                if (sI + d < NB && d <= a.degP) mx = fmaxf(mx, aq[sI + d]);
            msq[sI] = sI < n_int ? mx : 0.0f;
            tot += msq[sI];
        }
        const float wI = 1.0f / (float)n_int;
        // one proposal (number n of walker wb's sequence) against the envelope (mq, mtot) of the factor with coefficients ec
        auto propose = [&](unsigned long long wb, int n, const float (&ec)[NB], const float (&mq)[NB], float mtot, float& xc, int qrow) {
            Philox4 prop(seed, wb);
            prop.c0 = (unsigned)n;
            prop.c1 = 2u;
            // (__fmul_rn: the product must not contract into the subtraction t - base below -- k_tsample_p1g draws the same numbers only if both round it)
            const float t = __fmul_rn(prop.uniform(), mtot), u2 = prop.uniform();
            float run = 0.0f, base = 0.0f, msel = mq[0];
            int ssel = 0;
#pragma unroll
            for (int sI = 0; sI < NB; ++sI) {   // (the last interval with a positive bound catches t == mtot)
                const bool hit = t >= run && mq[sI] > 0.0f;
                ssel = hit ? sI : ssel;
                base = hit ? run : base;
                msel = hit ? mq[sI] : msel;
                run += mq[sI];
            }
            xc = fminf(((float)ssel + fminf((t - base) / msel, 1.0f)) * wI, 0.99999994f);
            float v;
            if (band) {
                // f(x) = sum_i q_i b_i(x) over the k + 1 plain B-splines alive on knot interval ssel (b_ssel .. b_{ssel + k}, k <= 8): a window of 12
                // coefficients from a multiple of four (the rest of the window multiplies zeros of the table) -- three 16-byte records per table row
                // and lerp end instead of NB / 4, three of the walker's row in LDS (qrow: its own lane's, or the lane's it is served by)
                const int a0 = min(ssel & ~3, NB - 12);
                const LerpN Lx = nlerp(xc, n_mesh);
                const float4_t* ra = reinterpret_cast<const float4_t*>(a.tabB0 + (size_t)Lx.il * NB + a0);
                const float4_t* rb = reinterpret_cast<const float4_t*>(a.tabB0 + (size_t)Lx.ir * NB + a0);
                const float4_t* qr = reinterpret_cast<const float4_t*>(qs + qrow * kQStride + a0);
                v = 0.0f;
#pragma unroll
                for (int qq = 0; qq < 3; ++qq) {
                    const float4_t ta = ra[qq], tb = rb[qq], qv = qr[qq];
                    v = __builtin_fmaf(qv.x, __builtin_fmaf(tb.x - ta.x, Lx.t, ta.x), v);
                    v = __builtin_fmaf(qv.y, __builtin_fmaf(tb.y - ta.y, Lx.t, ta.y), v);
                    v = __builtin_fmaf(qv.z, __builtin_fmaf(tb.z - ta.z, Lx.t, ta.z), v);
                    v = __builtin_fmaf(qv.w, __builtin_fmaf(tb.w - ta.w, Lx.t, ta.w), v);
                }
            } else {
                v = rows_lerp<NB>(a.tabP0, ec, xc, n_mesh);
            }
            return u2 * msel < v * v;
        };
        // Stage A: every lane proposes for its own walker, kTsOwn times at most (88 % of the walkers are done by then).  Stage B: the wave's remaining
        // walkers get eight lanes each, eight consecutive proposals of a walker's sequence per round, the first accepted one in sequence order taken --
        // the same draws as one lane proposing on alone, without the wave waiting 80 rounds for its unluckiest lane.
#ifndef WF_TS_OWN   // (experiment switch; the draws do not depend on it.  2^17 draws, round 4: 16 own proposals 0.281 ms, 12: 0.283, 8: 0.287, 4: 0.294)
#define WF_TS_OWN 16
#endif
        constexpr int kTsOwn = WF_TS_OWN;
        int n_prop = 0;
        float xs = __builtin_nanf("");
        bool done = false;
        const unsigned long long wb_own = (unsigned long long)(a.b0 + b);
        for (int n = 0; n < kTsOwn; ++n) {
            float xc;
            n_prop = n + 1;
            if (propose(wb_own, n, e, msq, tot, xc, (int)threadIdx.x)) { xs = xc; done = true; break; }
        }
        if (NB > 32 || __ballot(true) != ~0ull) {
            // the batch's last, partial wave: lanes are missing from the groups, every walker keeps its own lane (and with two row blocks the
            // coefficients of a walker are too many to hand to other lanes)
            for (int n = kTsOwn; n < 100000 && !done; ++n) {
                float xc;
                n_prop = n + 1;
                if (propose(wb_own, n, e, msq, tot, xc, (int)threadIdx.x)) { xs = xc; done = true; }
            }
        } else {
            const int lane = threadIdx.x & 63, g = lane >> 3, r = lane & 7;
            unsigned long long rem = __ballot(!done);
            for (int pass = 0; pass < 64 && rem; ++pass) {
                // group g serves the g-th walker of `rem`
                unsigned long long mm = rem;
                for (int i = 0; i < g; ++i) mm &= mm - 1;
                const bool has = mm != 0;
                const int src = has ? __ffsll((long long)mm) - 1 : lane;
                float ew[NB], mw[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) { ew[j] = band ? 0.0f : __shfl(e[j], src); mw[j] = __shfl(msq[j], src); }   // (band form: the served walker's coefficients are read from its LDS row)
                const float totw = __shfl(tot, src);
                const unsigned wlo = __shfl((unsigned)(wb_own & 0xFFFFFFFFull), src), whi = __shfl((unsigned)(wb_own >> 32), src);
                const unsigned long long wbw = ((unsigned long long)whi << 32) | wlo;
                float xw = __builtin_nanf("");
                bool found = !has;
                int rounds = 0;
                for (int round = 0; round < (100000 - kTsOwn) / 8; ++round) {
                    float xc = 0.0f;
                    const bool acc = !found && propose(wbw, kTsOwn + round * 8 + r, ew, mw, totw, xc, (int)(threadIdx.x & ~63u) + src);
                    const unsigned long long hits = __ballot(acc);
                    const unsigned gh = (unsigned)(hits >> (8 * g)) & 0xFFu;
                    const float xfirst = __shfl(xc, 8 * g + (gh ? __ffs((int)gh) - 1 : 0));
                    if (!found) rounds = round + 1;
                    if (!found && gh) { xw = xfirst; found = true; }
                    if (__ballot(!found) == 0ull) break;
                }
                // the walkers served in this pass take their draws from the first lane of their group
                const int rank = __popcll(rem & ((1ull << lane) - 1ull));
                const float xmine = __shfl(xw, 8 * (rank & 7));
                const int rmine = __shfl(rounds, 8 * (rank & 7));
                if (!done && rank < 8) { xs = xmine; done = true; n_prop = kTsOwn + 8 * rmine; }
                rem = __ballot(!done);
            }
        }
        const float l0 = lat[b];
#ifdef WF_TS_COUNT   // diagnostics build: the number of proposals of column 1 instead of its draw in the reported latent
        if (latent_out) { latent_out[b * 2] = l0; latent_out[b * 2 + 1] = (float)n_prop; }
#else
        if (latent_out) { latent_out[b * 2] = l0; latent_out[b * 2 + 1] = xs; }
#endif
#ifdef WF_TS_DEBUG   // (diagnostics: an intermediate instead of the first column in the reported latent)
        if (latent_out) latent_out[b * 2] = WF_TS_DEBUG == 1 ? tot : (WF_TS_DEBUG == 2 ? rn : msq[WF_TS_DEBUG - 3]);
#endif
        start_layer(a.n_layers - 1, l0, xs);
        return;
    }
    // phase 2: c_j = g_j (v_j / S0 + reg) / Q (calculate_bijection_params + reg, remove_bias, boundary map), v_j = 1 / (2^o_j + 1)
    float c[NB];
    float S0 = 0.0f, Qv = 0.0f, G = 0.0f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float g = a.gI[j];
        const float v = j < a.nbI ? r_of(oj0<NB>(oj, b, j)) : 0.0f;
        c[j] = v;
        S0 += v;
        Qv = __builtin_fmaf(v, g, Qv);
        G += j < a.nbI ? g : 0.0f;
    }
    const float rS = 1.0f / S0, rQ = 1.0f / __builtin_fmaf(Qv, rS, a.i_reg * G);
#pragma unroll
    for (int j = 0; j < NB; ++j) c[j] = j < a.nbI ? (a.gI[j] * __builtin_fmaf(c[j], rS, a.i_reg)) * rQ : 0.0f;
    const float o0 = cur0[b];
    float o1;
    if (kBand2 && a.i_band_int > 0) {
        float* cs = qs + threadIdx.x * kQStride;
        float* p4 = p4s + threadIdx.x * (NB / 4 + 4);
        float run = 0.0f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if ((j & 3) == 0) p4[j >> 2] = run;
            cs[j] = c[j];
            run = __builtin_fmaf(c[j], 1.0f, run);     // (as the full sum meets a row of ones)
        }
        o1 = inv_rows_band<NB>(a.tabI0, cs, p4, a.i_band_int, n_mesh, cur1[b], a.tol);
    } else {
        o1 = inv_rows<NB>(a.tabI0, c, n_mesh, cur1[b], a.tol);
    }
    if (layer > 0) {
        start_layer(layer - 1, o0, o1);
        return;
    }
    // BoxTransformLayer.reverse_fun_mean (made.py:186-197), two particles
    const float mean = 0.5f * o0, pm = o1 * (1.0f - o0) - (0.5f - mean);
    xg[b * 2] = ((0.0f - mean) + pm) * 2.0f * a.box_L;
    xg[b * 2 + 1] = ((o0 - mean) + pm) * 2.0f * a.box_L;
}

// Phase 1 of the staged sampler with EIGHT LANES PER WALKER from the start (band form only: TsArgs::ow and ::tabB0 set).  k_tsample<1> walks a walker's
// proposals one after the other on its own lane -- ~20 dependent table round trips per wave at two waves per SIMD (64 % of its cycles wait on memory).  Here lane
// r of a walker's group tests proposal 8 * round + r; the first accepted one in sequence order is taken: the same draws, ~3 round trips, sixteen waves per SIMD.
// Everything whose rounding depends on the order of a sum (|c|^2, the prefix sums of the envelope) is summed by the group's first lane in k_tsample<1>'s order.
template <int NB>
__global__ __launch_bounds__(256) void k_tsample_p1g(const TsArgs a, const float* __restrict__ oj, int64_t B, float* __restrict__ cur0, float* __restrict__ cur1,
                                                    float* __restrict__ cin, const float* __restrict__ lat, float* __restrict__ latent_out) {
    constexpr int kStride = NB + 12;     // (+ 8: the envelope reads aq[s .. s + 8]; rows stay 16-byte aligned)
    __shared__ __attribute__((aligned(16))) float qs[32 * kStride], aqs[32 * kStride], mqs[32 * kStride], cums[32 * kStride];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = t >> 3;
    const int r = threadIdx.x & 7, gl = (threadIdx.x & 63) >> 3, wrow = threadIdx.x >> 3, lane = threadIdx.x & 63;
    const bool valid = b < B;
    const int64_t bl = valid ? b : B - 1;
    const int n_mesh = a.n_mesh;
    unsigned long long seed = a.seed;
    if (a.seed_offset_dev) seed += *a.seed_offset_dev * 0x9E3779B97F4A7C15ull;
    float* q = qs + wrow * kStride;
    float* aq = aqs + wrow * kStride;
    float* mq = mqs + wrow * kStride;
    float* cum = cums + wrow * kStride;
    // |c|^2 in k_tsample<1>'s order (j ascending, fused multiply-adds) by the group's first lane
    float rn = 0.0f;
    if (r == 0) {
        float ss = 0.0f;
#pragma unroll 8
        for (int j = 0; j < NB; ++j) { const float cj = j < a.nbP ? oj0<NB>(oj, bl, j) : 0.0f; ss = __builtin_fmaf(cj, cj, ss); }
        rn = 1.0f / sqrtf(ss);
    }
    rn = __shfl(rn, lane & ~7);
#pragma unroll
    for (int jj = 0; jj < NB / 8 + 1; ++jj) {       // (+ 1: the eight slots behind the row, zeros for the envelope's look-ahead)
        const int j = r + 8 * jj;
        const float qi = (j < a.nbP) ? oj0<NB>(a.ow, bl, j) * rn : 0.0f;
        q[j] = qi;
        aq[j] = qi * qi;
    }
    const int n_int = a.nbP - a.degP;
#pragma unroll
    for (int jj = 0; jj < NB / 8; ++jj) {
        const int sI = r + 8 * jj;
        float mx = 0.0f;
#pragma unroll
        for (int d = 0; d <= 8; ++d)
            if (sI + d < NB && d <= a.degP) mx = fmaxf(mx, aq[sI + d]);
        mq[sI] = sI < n_int ? mx : 0.0f;
    }
    float tot = 0.0f;
    if (r == 0) {
#pragma unroll 8
        for (int sI = 0; sI < NB; ++sI) { cum[sI] = tot; tot += mq[sI]; }
    }
    tot = __shfl(tot, lane & ~7);
    const float wI = 1.0f / (float)n_int;
    const unsigned long long wb = (unsigned long long)(a.b0 + bl);
    bool found = !valid;
    float xw = __builtin_nanf("");
    for (int round = 0; round < 12500; ++round) {
        bool acc = false;
        float xc = 0.0f;
        if (!found) {
            Philox4 prop(seed, wb);
            prop.c0 = (unsigned)(round * 8 + r);
            prop.c1 = 2u;
            const float tt = __fmul_rn(prop.uniform(), tot), u2 = prop.uniform();
            // the last interval with cum <= tt and a positive bound (k_tsample<1>'s scan)
            int sI = 0;
#pragma unroll
            for (int step = NB / 2; step > 0; step >>= 1) sI = cum[sI + step] <= tt ? sI + step : sI;
            while (sI > 0 && !(mq[sI] > 0.0f)) --sI;
            const float msel = mq[sI], base = cum[sI];
            xc = fminf(((float)sI + fminf((tt - base) / msel, 1.0f)) * wI, 0.99999994f);
            const int a0 = min(sI & ~3, NB - 12);
            const LerpN Lx = nlerp(xc, n_mesh);
            const float4_t* ra = reinterpret_cast<const float4_t*>(a.tabB0 + (size_t)Lx.il * NB + a0);
            const float4_t* rb = reinterpret_cast<const float4_t*>(a.tabB0 + (size_t)Lx.ir * NB + a0);
            const float4_t* qr = reinterpret_cast<const float4_t*>(q + a0);
            float v = 0.0f;
#pragma unroll
            for (int qq = 0; qq < 3; ++qq) {
                const float4_t ta = ra[qq], tb = rb[qq], qv = qr[qq];
                v = __builtin_fmaf(qv.x, __builtin_fmaf(tb.x - ta.x, Lx.t, ta.x), v);
                v = __builtin_fmaf(qv.y, __builtin_fmaf(tb.y - ta.y, Lx.t, ta.y), v);
                v = __builtin_fmaf(qv.z, __builtin_fmaf(tb.z - ta.z, Lx.t, ta.z), v);
                v = __builtin_fmaf(qv.w, __builtin_fmaf(tb.w - ta.w, Lx.t, ta.w), v);
            }
            acc = u2 * msel < v * v;
        }
        const unsigned long long hits = __ballot(acc);
        const unsigned gh = (unsigned)(hits >> (8 * gl)) & 0xFFu;
        const float xfirst = __shfl(xc, 8 * gl + (gh ? __ffs((int)gh) - 1 : 0));
        if (!found && gh) { xw = xfirst; found = true; }
        if (__ballot(!found) == 0ull) break;
    }
    if (valid && r == 0) {
        const float l0 = lat[b];
        if (latent_out) { latent_out[b * 2] = l0; latent_out[b * 2 + 1] = xw; }
#ifdef WF_TS_DEBUG
        if (latent_out) latent_out[b * 2] = WF_TS_DEBUG == 1 ? tot : (WF_TS_DEBUG == 2 ? rn : mq[WF_TS_DEBUG - 3]);
#endif
        // the last layer's dimension 0 from the pair that leaves the prior (k_tsample: start_layer)
        const int l = a.n_layers - 1;
        const float o0 = inv_comp(a.comp + (size_t)l * n_mesh, n_mesh, xw, a.tol);
        cur0[b] = o0;
        cur1[b] = l0;
        cin[b] = a.exact ? o0 : xw;
    }
}

// mesh_search with the eight lanes of a walker's group (lane r of the group; all eight call it together): eight probes per round between lo and hi instead of the
// midpoint -- four rounds for 2 000 mesh points instead of eleven.  F is monotone on the mesh, so the result (the largest m with F(m) <= y, F(m), F(m + 1)) is the
// one mesh_search finds, bit for bit.
template <class F>
__device__ __forceinline__ void group_mesh_search(F f, int last, float y, int r, int gbase, int& m, float& yl, float& yr) {
    int lo = 0, hi = last;
    const float fe = r == 0 ? f(0) : (r == 1 ? f(last) : 0.0f);
    float flo = __shfl(fe, gbase), fhi = __shfl(fe, gbase + 1);
    const bool beyond = fhi <= y;
    while (hi - lo > 1) {
        const int span = hi - lo;
        const int p = span > 8 ? lo + (int)(((long long)span * (r + 1)) / 9) : lo + 1 + r;      // (distinct, ascending in r, strictly between lo and hi where used)
        const bool use = p < hi;
        const float fp = use ? f(p) : 0.0f;
        const unsigned le = (unsigned)(__ballot(use && fp <= y) >> (gbase & 63)) & 0xFFu;         // monotone: the lanes with F <= y are the first few
        const unsigned usem = (unsigned)(__ballot(use) >> (gbase & 63)) & 0xFFu;
        const int k = __popc(le);                                                                  // probes 0 .. k - 1 lie at or below y
        const int n_use = __popc(usem);
        const int plo = __shfl(p, gbase + (k > 0 ? k - 1 : 0)), phi = __shfl(p, gbase + (k < 8 ? k : 7));
        const float vlo = __shfl(fp, gbase + (k > 0 ? k - 1 : 0)), vhi = __shfl(fp, gbase + (k < 8 ? k : 7));
        if (k > 0) { lo = plo; flo = vlo; }
        if (k < n_use) { hi = phi; fhi = vhi; }
    }
    m = beyond ? last : lo;
    yl = beyond ? fhi : flo;
    yr = fhi;
}
// grid_root with the group: the two lerp values on lanes 0 and 1
template <class FL>
__device__ __forceinline__ float group_grid_root(FL flerp, int m, float yl, float yr, float y, int last, float tol, int r, int gbase) {
    const float n = (float)last;
    float xs = (float)m / n;
    if (yr > yl) xs = xs + (y - yl) / ((yr - yl) * n);
    int K = 0;
    float w = 1.0f;
    while (K < 64 && w * 0.5f > tol * 0.5f) { w *= 0.5f; ++K; }
    const float scale = ldexpf(1.0f, K);
    float q = floorf(xs * scale);
    q = fminf(fmaxf(q, 0.0f), scale - 1.0f);
    const float fv = r == 0 ? flerp(q / scale) - y : (r == 1 ? flerp(fminf(q + 1.0f, scale - 1.0f) / scale) - y : 0.0f);
    const float f_lo = __shfl(fv, gbase), f_hi = __shfl(fv, gbase + 1);
    if (f_hi <= 0.0f && q + 1.0f <= scale - 1.0f) q = q + 1.0f;
    else if (f_lo > 0.0f && q >= 1.0f) q = q - 1.0f;
    return q / scale;
}
// Phase 2 of the staged sampler / inverse with eight lanes per walker (band form of the spline sums: TsArgs::i_band_int > 0): the coefficients of dimension 1 in
// k_tsample<2>'s arithmetic (order-dependent sums by the group's first lane), both mesh searches of the phase as eight-way searches: 12 table round trips instead of 30,
// sixteen waves per SIMD instead of two; the same bits.
template <int NB>
__global__ __launch_bounds__(256) void k_tsample_p2g(const TsArgs a, int layer, const float* __restrict__ oj, int64_t B, float* __restrict__ cur0, float* __restrict__ cur1,
                                                    float* __restrict__ cin, float* __restrict__ xg) {
    constexpr int kStride = NB + 4;
    __shared__ __attribute__((aligned(16))) float cs_all[32 * kStride], p4_all[32 * (NB / 4 + 4)];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = t >> 3;
    const int r = threadIdx.x & 7, lane = threadIdx.x & 63, gbase = lane & ~7, wrow = threadIdx.x >> 3;
    const bool valid = b < B;
    const int64_t bl = valid ? b : B - 1;
    const int n_mesh = a.n_mesh;
    float* cs = cs_all + wrow * kStride;
    float* p4 = p4_all + wrow * (NB / 4 + 4);
    // v_j = 1 / (2^o_j + 1) on the group's lanes, the sums S0, Qv, G in j order by its first lane (k_tsample<2>)
#pragma unroll
    for (int jj = 0; jj < NB / 8; ++jj) {
        const int j = r + 8 * jj;
        cs[j] = j < a.nbI ? r_of(oj0<NB>(oj, bl, j)) : 0.0f;
    }
    float rS = 0.0f, rQ = 0.0f;
    if (r == 0) {
        float S0 = 0.0f, Qv = 0.0f, G = 0.0f;
#pragma unroll 8
        for (int j = 0; j < NB; ++j) {
            const float g = a.gI[j], v = cs[j];
            S0 += v;
            Qv = __builtin_fmaf(v, g, Qv);
            G += j < a.nbI ? g : 0.0f;
        }
        rS = 1.0f / S0;
        rQ = 1.0f / __builtin_fmaf(Qv, rS, a.i_reg * G);
    }
    rS = __shfl(rS, gbase);
    rQ = __shfl(rQ, gbase);
#pragma unroll
    for (int jj = 0; jj < NB / 8; ++jj) {
        const int j = r + 8 * jj;
        cs[j] = j < a.nbI ? (a.gI[j] * __builtin_fmaf(cs[j], rS, a.i_reg)) * rQ : 0.0f;
    }
    if (r == 0) {
        float run = 0.0f;
#pragma unroll 8
        for (int j = 0; j < NB; ++j) {
            if ((j & 3) == 0) p4[j >> 2] = run;
            run = __builtin_fmaf(cs[j], 1.0f, run);
        }
    }
    const int n_int = a.i_band_int;
    auto window = [&](int i) { return min(min((i * n_int) / (n_mesh - 1), n_int - 1) & ~3, NB - 12); };
    auto dot_at = [&](int i) {
        const int a0 = window(i);
        const float4_t* rr = reinterpret_cast<const float4_t*>(a.tabI0 + (size_t)i * NB + a0);
        const float4_t* cq = reinterpret_cast<const float4_t*>(cs + a0);
        float acc = p4[a0 >> 2];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4_t tv = rr[q], c = cq[q];
            acc = __builtin_fmaf(c.x, tv.x, acc);
            acc = __builtin_fmaf(c.y, tv.y, acc);
            acc = __builtin_fmaf(c.z, tv.z, acc);
            acc = __builtin_fmaf(c.w, tv.w, acc);
        }
        return acc;
    };
    auto lerp_at = [&](float x) {
        const LerpN L = nlerp(x, n_mesh);
        const int a0 = window(min(L.il, L.ir));
        const float4_t* ra = reinterpret_cast<const float4_t*>(a.tabI0 + (size_t)L.il * NB + a0);
        const float4_t* rb = reinterpret_cast<const float4_t*>(a.tabI0 + (size_t)L.ir * NB + a0);
        const float4_t* cq = reinterpret_cast<const float4_t*>(cs + a0);
        float acc = p4[a0 >> 2];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4_t ta = ra[q], tb = rb[q], c = cq[q];
            acc = __builtin_fmaf(c.x, __builtin_fmaf(tb.x - ta.x, L.t, ta.x), acc);
            acc = __builtin_fmaf(c.y, __builtin_fmaf(tb.y - ta.y, L.t, ta.y), acc);
            acc = __builtin_fmaf(c.z, __builtin_fmaf(tb.z - ta.z, L.t, ta.z), acc);
            acc = __builtin_fmaf(c.w, __builtin_fmaf(tb.w - ta.w, L.t, ta.w), acc);
        }
        return acc;
    };
    const float o0 = cur0[bl], y1 = cur1[bl];
    int m;
    float yl, yr;
    group_mesh_search(dot_at, n_mesh - 1, y1, r, gbase, m, yl, yr);
    const float o1 = group_grid_root(lerp_at, m, yl, yr, y1, n_mesh - 1, a.tol, r, gbase);
    if (layer > 0) {
        // the layer below: Reverse.inverse_fun, then its dimension 0 through the composite table (k_tsample: start_layer(layer - 1, o0, o1))
        const float4_t* comp = a.comp + (size_t)(layer - 1) * n_mesh;
        const float n0 = o1, n1 = o0;
        group_mesh_search([&](int i) { return comp[i].x; }, n_mesh - 1, n0, r, gbase, m, yl, yr);
        const float od = group_grid_root([&](float x) { return comp_lerp_x(comp, x, n_mesh); }, m, yl, yr, n0, n_mesh - 1, a.tol, r, gbase);
        if (valid && r == 0) {
            cur0[b] = od;
            cur1[b] = n1;
            cin[b] = a.exact ? od : n0;
        }
        return;
    }
    if (valid && r == 0) {
        // BoxTransformLayer.reverse_fun_mean (made.py:186-197), two particles
        const float mean = 0.5f * o0, pm = o1 * (1.0f - o0) - (0.5f - mean);
        xg[b * 2] = ((0.0f - mean) + pm) * 2.0f * a.box_L;
        xg[b * 2 + 1] = ((o0 - mean) + pm) * 2.0f * a.box_L;
    }
}

int check() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    return WF_OK;
}

}  // namespace
#ifndef WF_ETILE_ONLY_K2   // (the host side: in the main translation unit only)

bool energy_tile_fused(const MfmaDev* mdev) {
    const char* e = getenv("WF_ENERGY_FUSED");
    // (two row blocks per dimension: the static LDS of k_efused<2> -- 256 B of chunk bounds -- has to fit beside the resident nets too)
    const bool fits = (mdev->const_floats + mdev->net_floats * mdev->n_nets) * 4 + 512 <= 160 * 1024;
    return !mdev->staged && fits && !(e && atoi(e) == 0);
}

// workspace: state (12 floats), head triples (96 floats), the sign sum (1 float) per walker
int64_t energy_tile_floats(int64_t B) { return B * (12 + 1) + ((B + 31) / 32) * 32 * (32 * NCH); }   // state, s1, head triples of whole tiles

// mdev: the model's MFMA description (resident or not: one net is staged per launch); md: ModelDev on the host (spline sizes, masks)
int launch_energy_tile(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* fk_nat, const float* x, int64_t B,
                       const Protons& pr, float* hpsi, float* psi, float* lap, float* ws, void* stream, float* st_out) {
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return WF_OK;
    // every net resident in LDS (the shipped shapes): the whole of H psi in one launch, nothing through HBM but the walkers and the results.
    // WF_ENERGY_FUSED=0 (read per call) keeps the launch-per-net path below (A/B tests; models whose nets do not fit together take it anyway).
    {
        if (energy_tile_fused(mdev)) {
            const int lds_all = (mdev->const_floats + mdev->net_floats * mdev->n_nets) * (int)sizeof(float);
            const int64_t n_tiles = (B + 31) / 32;
            const unsigned blocks = (unsigned)std::min<int64_t>((n_tiles + kFusedWaves - 1) / kFusedWaves, 256);
#define WF_EFUSED(NBK_, PB_)                                                                                                                   \
    {                                                                                                                                          \
        static DynLdsSlots cfg{};                                                                                                              \
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_efused<NBK_, PB_>), lds_all, &cfg)) return rc;                          \
        hipLaunchKernelGGL((k_efused<NBK_, PB_>), dim3(blocks), dim3(kFusedWaves * 64), lds_all, s, *mdev, tabI4, tabP4, x, B, pr, hpsi, psi, lap, st_out); \
    }
            if (mdev->nbk == 1) {
                if (mdev->p_bias) WF_EFUSED(1, true) else WF_EFUSED(1, false)
            } else {
                if (mdev->p_bias) WF_EFUSED(2, true) else WF_EFUSED(2, false)
            }
#undef WF_EFUSED
            return check();
        }
    }
    float* st = ws;
    float* s1 = st + 12 * B;
    float* oj = s1 + B;
    const unsigned lane_blocks = (unsigned)((B + 255) / 256);
    const int lds_bytes = (mdev->const_floats + mdev->net_floats) * (int)sizeof(float);
    static DynLdsSlots cfg_flow{}, cfg_prior{};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_etile_cond<false>), lds_bytes, &cfg_flow)) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_etile_cond<true>), lds_bytes, &cfg_prior)) return rc;
    const int64_t n_tiles = (B + 31) / 32;
    const unsigned cond_blocks = (unsigned)std::min<int64_t>((n_tiles + kCondWaves - 1) / kCondWaves, 256 * 4);
    hipLaunchKernelGGL(k_etile_box, dim3(lane_blocks), dim3(256), 0, s, x, B, md.box_L, st);
    for (int l = 0; l < md.n_layers; ++l) {
        hipLaunchKernelGGL(k_etile_cond<false>, dim3(cond_blocks), dim3(kCondWaves * 64), lds_bytes, s, *mdev, l, (const float*)st, B, oj, s1);
        hipLaunchKernelGGL(k_etile_flow, dim3(lane_blocks), dim3(256), 0, s, mdev->comp + (size_t)l * mdev->n_mesh, tabI4, fk_nat, md.isp.nb,
                           md.isp.n_mesh, md.i_reg, (const float*)oj, B, st);
    }
    hipLaunchKernelGGL(k_etile_cond<true>, dim3(cond_blocks), dim3(kCondWaves * 64), lds_bytes, s, *mdev, md.n_layers, (const float*)st, B, oj, s1);
    hipLaunchKernelGGL(k_etile_prior, dim3(lane_blocks), dim3(256), 0, s, mdev->comp + (size_t)md.n_layers * mdev->n_mesh, tabP4, md.psp.nb,
                       md.psp.n_mesh, md.constrained_mask, (const float*)oj, (const float*)s1, (const float*)st, x, B, pr, hpsi, psi, lap);
    return check();
}


// ---- host side of the matrix-core gradient path
static int64_t ebwd_lds_floats(const MfmaDev* mdev) {
    // the reverse kernel's LDS: constants, one net's forward and transposed images, the transposed ob_to_b, the workgroup's accumulators of (dW1, dW2)
    return (int64_t)mdev->const_floats + mdev->net_floats + mdev->tnet_floats + mdev->nbk * mdev->nbk * 1024 + acc_sets(mdev->nbk) * acc_blocks(mdev->nbk) * 1024;
}
bool energy_vjp_capable(const MfmaDev* mdev) {
    return mdev->timg_off >= 0 && (mdev->nbk == 1 || mdev->nbk == 2) && energy_tile_fused(mdev) && !mdev->i_gate && !mdev->p_gate &&
           ebwd_lds_floats(mdev) * (int64_t)sizeof(float) <= 160 * 1024 - 1024;
}
// floats of workspace per walker of a chunk (whole tiles), + the fixed part
int64_t energy_vjp_floats_per_walker(int n_nets) { return (int64_t)n_nets * 12 + 12 + 4; }   // per-net input jets, adjoint jets, H psi / psi / seeds
int64_t energy_vjp_fixed_floats(int n_nets, int nbk) { return (int64_t)n_nets * kESplit * g_floats(nbk) + 128; }   // the workgroups' gradient blocks
int energy_vjp_gacc_floats(int n_nets, int nbk) { return n_nets * g_floats(nbk); }

template <int NBK>
static int launch_ebwd_t(const MfmaDev* mdev, const float* tabI4, const float* tabP4, const float* st, float* adjb, const float* w_psi, const float* w_lap, int64_t B,
                         float* partial, unsigned blocks, hipStream_t s) {
    const int n_nets = mdev->n_nets;
    const int lds_bytes = (int)(ebwd_lds_floats(mdev) * (int64_t)sizeof(float));
    static DynLdsSlots cfg_p{}, cfg_f{};
    if (int r2 = ensure_dynamic_lds(reinterpret_cast<const void*>(k_ebwd<true, NBK>), lds_bytes, &cfg_p)) return r2;
    if (int r2 = ensure_dynamic_lds(reinterpret_cast<const void*>(k_ebwd<false, NBK>), lds_bytes, &cfg_f)) return r2;
    for (int n = n_nets - 1; n >= 0; --n) {
        const float* st_n = st + (size_t)n * 12 * B;
        float* part_n = partial + (size_t)n * kESplit * GL<NBK>::floats;
        if (n == n_nets - 1)
            hipLaunchKernelGGL((k_ebwd<true, NBK>), dim3(blocks), dim3(kBwdWaves * 64), lds_bytes, s, *mdev, n, tabI4, tabP4, st_n, adjb, w_psi, w_lap, B, part_n);
        else
            hipLaunchKernelGGL((k_ebwd<false, NBK>), dim3(blocks), dim3(kBwdWaves * 64), lds_bytes, s, *mdev, n, tabI4, tabP4, st_n, adjb, w_psi, w_lap, B, part_n);
    }
    return WF_OK;
}

// One chunk of walkers (B a multiple of 32 except for the last chunk of a batch): forward with the per-net input jets, seeds (mode 2: from H psi of
// this very sweep, e_loc is written; mode 1: w_psi / w_lap given), reverse net by net with the weight-gradient products behind each net.
// gacc [n_nets][g_floats(nbk)]: accumulated over the chunks of a batch (accumulate = 0 for the first one).
int launch_energy_vjp(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* x, int64_t B, int mode, const float* w_psi,
                      const float* w_lap, const Protons& pr, float running_avg, const float* running_avg_dev, float inv_count, float* e_loc, float* ws,
                      float* gacc, int accumulate, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return WF_OK;
    const int n_nets = mdev->n_nets, gf = g_floats(mdev->nbk);
    const int64_t n_tiles = (B + 31) / 32;
    float* st = ws;                                  // [n_nets][12][B]
    float* adjb = st + (size_t)n_nets * 12 * B;      // [12][B]
    float* hpsi = adjb + 12 * B;
    float* psi = hpsi + B;
    float* wp = psi + B;
    float* wl = wp + B;
    float* partial = ws + (((size_t)(n_nets * 12 + 12 + 4) * B + 63) / 64) * 64;   // [n_nets][kESplit][gf]
    int rc = launch_energy_tile(mdev, md, tabI4, tabP4, nullptr, x, B, pr, hpsi, psi, nullptr, nullptr, stream, st);
    if (rc) return rc;
    if (mode == 2) {
        rc = launch_vqmc_seeds(x, B, 2, pr, hpsi, psi, running_avg, inv_count, e_loc, wp, wl, running_avg_dev, stream);
        if (rc) return rc;
        w_psi = wp;
        w_lap = wl;
    }
    const unsigned blocks = (unsigned)std::min<int64_t>((n_tiles + kBwdWaves - 1) / kBwdWaves, 256);
    rc = mdev->nbk == 1 ? launch_ebwd_t<1>(mdev, tabI4, tabP4, st, adjb, w_psi, w_lap, B, partial, blocks, s)
                        : launch_ebwd_t<2>(mdev, tabI4, tabP4, st, adjb, w_psi, w_lap, B, partial, blocks, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_egrad_reduce, dim3((gf + 255) / 256, n_nets), dim3(256), 0, s, (const float*)partial, (int)blocks, accumulate, gacc, gf);
    return check();
}

// gacc -> flat gradient in the reference's leaf order (every entry written: zero first, then the live leaves)
int launch_energy_vjp_finish(const float* gacc, int n_nets, int nbk, const int* offs /* [n_nets][8]: W0, b0, W1, b1, W2, b2, NO, n_out */, const float* c2, float* flat,
                             int64_t n_params, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    ENetOffs o{};
    for (int n = 0; n < n_nets && n < 8; ++n) {
        const int* q = offs + 8 * n;
        o.n[n] = ENetOff{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], c2[n]};
    }
    hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)((n_params + 255) / 256)), dim3(256), 0, s, flat, n_params);
    hipLaunchKernelGGL(k_egrad_scatter, dim3((g_floats(nbk) + 255) / 256, n_nets), dim3(256), 0, s, gacc, n_nets, o, flat, nbk);
    return check();
}

// ---- host side of the staged inverse / sampler
bool tile_sample_capable(const MfmaDev* mdev) {
    return mdev->D == 2 && (mdev->nbk == 1 || mdev->nbk == 2) && mdev->n_layers > 0 && mdev->n_layers < 8 && !mdev->i_gate && !mdev->p_gate &&
           mdev->comp != nullptr && (mdev->const_floats + mdev->net_floats) * 4 <= 160 * 1024 - 64;
}
// floats of workspace: conditioner input (5 B: the slot of the second input sits 4 B behind the first), cur0, cur1, the latent pair, the prior's sign sums,
// the head outputs of whole tiles (32 nbk rows; sized for three channels -- the conditioner launches have written the value channel alone since round 4)
int64_t tile_sample_floats(int64_t B, int nbk) { return B * 10 + ((B + 31) / 32) * 32 * (32 * nbk * NCH) + 64; }

#endif   // WF_ETILE_ONLY_K2
namespace {
template <int NBK>
int launch_tile_sample_t(const MfmaDev* mdev, const ModelDev& md, const TsArgs& a_in, int draw, const float* u, int64_t B, float* x, float* latent, float* ws, hipStream_t s) {
    constexpr int NB = 32 * NBK;
    float* cin = ws;                 // [5][B]
    float* cur0 = cin + 5 * B;
    float* cur1 = cur0 + B;
    float* lat = cur1 + B;           // [B] (column 0 between the two prior phases)
    float* s1 = lat + 2 * B;
    float* oj = ws + (((size_t)10 * B + 63) / 64) * 64;
    float* ow = (mdev->p_plain_bc && !getenv("WF_SAMPLE_DENSE_ENVELOPE")) ? oj + (size_t)((B + 31) / 32) * 32 * NB : nullptr;   // (behind the one channel oj holds: sized for three)
    TsArgs a = a_in;
    a.ow = ow;
    a.tabB0 = (ow && !getenv("WF_SAMPLE_FULL_ROWS")) ? mdev->tabB0 : nullptr;
    const unsigned lane_blocks = (unsigned)((B + 255) / 256);
    const int lds_bytes = (mdev->const_floats + mdev->net_floats) * (int)sizeof(float);
    static DynLdsSlots cfg_flow{}, cfg_prior{};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_etile_cond<false, NBK, 1>), lds_bytes, &cfg_flow)) return rc;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(k_etile_cond<true, NBK, 1>), lds_bytes, &cfg_prior)) return rc;
    const int64_t n_tiles = (B + 31) / 32;
    const unsigned cond_blocks = (unsigned)std::min<int64_t>((n_tiles + kCondWaves - 1) / kCondWaves, 256 * 4);
    const int L = md.n_layers;
    if (draw) {
        hipLaunchKernelGGL((k_tsample<0, NB>), dim3(lane_blocks), dim3(256), 0, s, a, 0, (const float*)oj, u, B, cur0, cur1, cin, lat, latent, x);
        hipLaunchKernelGGL((k_etile_cond<true, NBK, 1>), dim3(cond_blocks), dim3(kCondWaves * 64), lds_bytes, s, *mdev, L, (const float*)cin, B, oj, s1, ow);
        if (a.ow && a.tabB0 && !getenv("WF_SAMPLE_ONE_LANE"))   // (the band form: eight lanes per walker)
            hipLaunchKernelGGL((k_tsample_p1g<NB>), dim3((unsigned)((B * 8 + 255) / 256)), dim3(256), 0, s, a, (const float*)oj, B, cur0, cur1, cin, (const float*)lat, latent);
        else
            hipLaunchKernelGGL((k_tsample<1, NB>), dim3(lane_blocks), dim3(256), 0, s, a, 0, (const float*)oj, u, B, cur0, cur1, cin, lat, latent, x);
    } else {
        hipLaunchKernelGGL((k_tsample<3, NB>), dim3(lane_blocks), dim3(256), 0, s, a, 0, (const float*)oj, u, B, cur0, cur1, cin, lat, latent, x);
    }
    for (int l = L - 1; l >= 0; --l) {
        hipLaunchKernelGGL((k_etile_cond<false, NBK, 1>), dim3(cond_blocks), dim3(kCondWaves * 64), lds_bytes, s, *mdev, l, (const float*)cin, B, oj, s1);
        // (the band form with eight lanes per walker: two row blocks only -- 2^17 draws 0.320 -> 0.290 ms; with one row block the walker's own lane is faster,
        // 0.201 against 0.225: WF_SAMPLE_GROUP_PHASE2 forces it, WF_SAMPLE_ONE_LANE the other form)
        if (a.i_band_int > 0 && !getenv("WF_SAMPLE_ONE_LANE") && (NB > 32 || getenv("WF_SAMPLE_GROUP_PHASE2")))
            hipLaunchKernelGGL((k_tsample_p2g<NB>), dim3((unsigned)((B * 8 + 255) / 256)), dim3(256), 0, s, a, l, (const float*)oj, B, cur0, cur1, cin, x);
        else
            hipLaunchKernelGGL((k_tsample<2, NB>), dim3(lane_blocks), dim3(256), 0, s, a, l, (const float*)oj, u, B, cur0, cur1, cin, lat, latent, x);
    }
    return check();
}
}  // namespace
#ifndef WF_ETILE_ONLY_K2

// draw == 0: x = inverse(u);  draw == 1: latent ~ prior (reported in `latent` if given), x = inverse(latent)
int launch_tile_sample(const MfmaDev* mdev, const ModelDev& md, const float* tabI0, const float* tabP0, const float* fk_nat, int draw, unsigned long long seed,
                       const float* u, int64_t B, float* x, float* latent, int exact, const unsigned long long* seed_offset_dev, int64_t b0, float* ws, void* stream) {
    if (B == 0) return WF_OK;
    TsArgs a{};
    a.comp = mdev->comp;
    a.tabI0 = tabI0;
    a.tabP0 = tabP0;
    a.gI = fk_nat;
    a.b_to_ob = md.b_to_ob;
    a.n_mesh = mdev->n_mesh;
    a.nbI = md.isp.nb;
    a.nbP = md.psp.nb;
    a.degP = md.psp.degree;
    a.i_band_int = (mdev->i_plain_bc && md.isp.degree <= 7 && !getenv("WF_SAMPLE_FULL_ROWS")) ? md.isp.nb - md.isp.degree : 0;
    a.n_layers = md.n_layers;
    a.i_reg = md.i_reg;
    a.tol = md.reverse_tol;
    a.box_L = md.box_L;
    a.seed = seed;
    a.seed_offset_dev = seed_offset_dev;
    a.exact = exact;
    a.b0 = b0;
    return mdev->nbk == 1 ? launch_tile_sample_t<1>(mdev, md, a, draw, u, B, x, latent, ws, (hipStream_t)stream)
                          : launch_tile_sample_t<2>(mdev, md, a, draw, u, B, x, latent, ws, (hipStream_t)stream);
}

#endif   // WF_ETILE_ONLY_K2
}  // namespace wf
