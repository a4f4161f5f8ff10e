// wf_internal.h -- declarations shared by the translation units of libwaveflow_hip.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/waveflow_hip.h"

namespace wf {

// ---- wf_tables.cpp (host, fp64)
std::vector<double> make_knots(int kind, int k, int n_internal);
int n_bases_of(int kind, int k, int n_internal);
int build_raw_table(int kind, int k, int n_internal, int n_mesh, double* out);
int build_ortho_b(int k, int n_internal, int n_mesh, const double* Bt, double* ob, double* b_to_ob, double* ob_to_b);

// ---- device-side model image (wf_model.cpp fills it, kernels read it)
constexpr int kHidden = 64;      // MaskedDense width, model_factory.py:72
constexpr int kMaxLayers = 16;   // flow layers
constexpr int kMaxNets = kMaxLayers + 1;

// One conditioner net (masked weights, reference layout; scalar kernel)
struct NetPlain {
    const float* W0;  // [D][64]   W0 * mask0
    const float* b0;  // [64]
    const float* W1t; // [64 out][64 in]  (W1 * mask1) transposed: row j = weights into hidden unit j
    const float* b1;  // [64]
    const float* W2t; // [D][NBP][64]  (W2 * mask2) transposed and regrouped: row (d, j) = weights of output column j*D+d
    const float* b2;  // [D][NBP]
    // the same masked weights in the orientation the reverse pass contracts over (wf_kernels_grad.hip)
    const float* W1n; // [64 in][64 out]   row a = weights out of hidden unit a
    const float* W2n; // [64 in][D][NBP]   row a = weights out of hidden unit a
    // zero_params of the gated head (model_factory.py:64-67, 84): z[d][j], already |z| under a sigmoid head; zeros without the leaf
    const float* zero; // [D][NBP]
    const float* zero_raw; // [D][NBP]: the leaf as it is (its sign scales the gradient of a sigmoid head's |z|)
};

// One conditioner net in MFMA operand order (see wf_kernels_mfma.hip)
struct NetMfma {
    const float* image;  // LDS image of this net, `image_floats` floats
    int image_floats;
};

using float4_t = __attribute__((ext_vector_type(4))) float;

// One conditioner net for the wave-cooperative kernels (wf_kernels_wave.hip): lane = hidden unit / output row, every
// matrix in "lane-major float4 groups" [16 groups of 4 contracted indices][64 lanes] so that one load instruction of the
// wave reads 1 KB contiguous.  P = ceil(D / 2) output passes; lane (dl, j) of pass p is basis row j of dimension 2p + dl.
struct NetWave {
    const float* W0;        // [D][64]  W0 * mask0
    const float* b0;        // [64]
    const float* b1;        // [64]
    const float* b2;        // [P][64]
    const float4_t* W1f;    // [16][64]: lane j,  group g: (W1*mask1)[a = 4g..4g+3][j]
    const float4_t* W1b;    // [16][64]: lane a,  group g: (W1*mask1)[a][j = 4g..4g+3]
    const float4_t* W2f;    // [P][16][64]: lane (dl, j), group g: (W2*mask2)[a = 4g..][column of (2p+dl, j)]
    const float4_t* W2b;    // [P][16][64]: lane a, group g: (W2*mask2)[a][columns of lanes c = 4g..4g+3 of pass p]
    const float* z;         // [P][64]: zero_params of a gated head in the lane order of b2 (|z| under a sigmoid head); zeros otherwise
};

struct SplineDev {
    const float* tab;    // [2 or 4][n_mesh][NBP] fp32, mesh-major ("dense rows"); order nd, then mesh point, then basis
    int nb;              // real number of bases
    int nbp;             // padded row length (32 or 64)
    int n_mesh;
    int degree;
    // boundary-condition constants: for constraint p, prev[p][j] = T[nd_p][j or nb-1-j][end], value[p]
    int n_left, n_right;
    int left_nd[WF_MAX_BC], right_nd[WF_MAX_BC];
    float left_val[WF_MAX_BC], right_val[WF_MAX_BC];
    float left_prev[WF_MAX_BC][WF_MAX_BC], right_prev[WF_MAX_BC][WF_MAX_BC];
    float left_value[WF_MAX_BC], right_value[WF_MAX_BC];
};

struct ModelDev {
    int D;
    int nbp;                      // padded bases per dimension of every table / weight image (32 or 64)
    int n_layers;
    int layer_kind;
    int box_kind;
    float box_L;
    float i_reg;
    int prior_kind;
    float normal_offset;
    unsigned constrained_mask;    // bit d set: d in constrained_dimension_indices_left
    SplineDev isp;                // flow-layer I-spline (IMADE)
    SplineDev psp;                // prior spline: orthogonal-B (WAVEFLOW) or M (MFLOW)
    const float* ob_to_b;         // [nbp][nbp] fp32, zero beyond nb (WAVEFLOW): row a = ob_to_b[a][:]
    const float* ob_to_b_t;       // the same with the boundary map folded into its rows (wf_model.cpp: bc_map): table-driven kernels
    const float* b_to_ob;         // [nbp][nbp] fp32, zero beyond nb (WAVEFLOW): the sampler's bound (bsplines_jax.py:164-166)
    const float* p_cb;            // [nbp] or null: constant term of the B prior's boundary map, b @ ob_to_b (a constraint with a non-zero value,
                                  // bsplines_jax.py:173-199): c = (A o) @ ob_to_b + (sum o) * p_cb -- the net's outputs reach the constraints divided by their sum
    float reverse_tol;            // IMADE reverse_fun_tol
    int i_gate, p_gate;           // set_nn_output_grad_to_zero of the layers' / the prior's conditioner (wf_model_desc)
    NetPlain nets[kMaxNets];      // flow layers 0..n_layers-1, then the prior net
    NetMfma mnets[kMaxNets];
    NetWave wnets[kMaxNets];
};

// MFMA kernel's view of the model (wf_kernels_mfma.hip)
struct MfmaDev {
    int D, n_layers, layer_kind, box_kind, prior_kind;
    float box_L, i_reg, normal_offset;
    unsigned constrained_mask;
    int i_nb, p_nb, n_mesh;
    int nbk;                   // 32-row blocks per dimension (1: <= 32 bases, 2: <= 64)
    float F_I, F_P;            // sum of the row factors fk (flow-layer spline / prior spline)
    const float* image;        // global image: n_nets net images (net_floats each), then the constants block
    int n_nets, net_floats;
    int const_img_off;         // float offset of the constants block inside the image
    int const_floats;          // fkI[nbk][2][16], fkP[nbk][2][16], ob_to_b image [nbk][nbk]{hi, lo}[2 K steps][64 lanes][8 halves], piece bounds int32 [2 tables][nbk][2][16]
    int staged;                // 0: every net resident in LDS; 1: one LDS slot, nets re-staged per super-chunk of tiles
    int staged_groups;         // staged mode: tile groups per wave and super-chunk (set per launch, 1 .. kStagedGroups)
    const float* tabI;         // [n_mesh][8 nbk pieces][nd 0..1][side: m, m + 1][4 rows] fp32: fk_row * I_row (wf_model.cpp: pack_rows_pairs)
    const float* rsI;          // [n_mesh]{R0_m, R1_m, R0_{m+1}, R1_{m+1}}: sum over rows of tabI (orders 0, 1), both lerp ends in one 16-byte record
    const float* tabP;         // [n_mesh][8 nbk pieces][side][4 rows] fp32, prior rows (B as is; M: fk_row * M_row), nd 0
    const float4_t* comp;      // [n_nets][n_mesh] composite tables of output dimension 0 (k_prepare_dim0)
    const float4_t* comp2;     // [n_nets][n_mesh]{comp[m].xy, comp[m + 1].xy}: both lerp ends of what k_mfma reads, one 16-byte record (k_pair_dim0)
    float* dbg;                // diagnostics builds only (WF_DEBUG / WF_STAMP)
    int exact_div;             // 1: x_l / n by IEEE division (set when the multiply-and-correct form is not bit-identical for this n_mesh)
    int prior_quotient;        // debug (env WF_PRIOR_QUOTIENT=1 at model creation): Waveflow prior head in the reference's quotient form
    int i_gate, p_gate;        // gated heads (wf_model_desc.i_gate / p_gate): zero_params blocks of the net images are live
    int timg_off, tnet_floats, tconst_off;   // transposed operand images of the gradient path behind the constants block (float offsets in `image`; -1: not built)
int i_plain_bc;            // I layers: the boundary map only zeroes coefficients (the staged sampler's band form of the spline sums)
        int p_plain_bc;            // B prior: the boundary map only zeroes coefficients (the staged sampler then reads the plain B-spline coefficients of the prior's second factor off the conditioner launch)
    const float* tabB0;        // [n_mesh][32 nbk] plain B-splines of the prior, order 0 (the staged sampler: the k + 1 of them alive on a knot interval give a proposal's value)
    int p_bias;                // the B prior's boundary map has a constant term: cbP[nbk][2][16] (accumulator layout) sits at the end of the constants block
    const int* f16_ovf;        // [n_nets] 1 = a packed weight of that net is outside the fp16 range (k_fold_bias, rewritten at every upload): outputs are poisoned with NaN
};

constexpr int kStagedGroups = 4;   // staged mode: a wave's tile groups whose state waits in LDS between two nets (LDS: waves x groups x T x (D + 1) x 128 B)
bool mfma_div_ok(int n_mesh);   // host check of div_by_n (wf_mfma_impl.h) against the division for every x_l in [-1, n_mesh]

int launch_mfma(int D, int nbk, const MfmaDev* mdev, int lds_bytes, int mode, const float* x, int64_t B, float* out, float* u,
                int32_t* idx, void* stream);
bool mfma_shape_built(int D, int nbk);
int mfma_extra_lds_floats(int n_nets);
int dim0_coef_floats(int n_nets);
int launch_fold_bias(float* image_dev, int n_nets, int net_floats, int D, int nbk, int* ovf_dev, void* stream);
int launch_prepare_dim0(const ModelDev* md_dev, int n_nets, int n_mesh, const float* fk_nat_dev, float F_I, float F_P, const float* tab_i_dev,
                        const float* tab_p_dev, void* comp_dev,
                        void* stream);

// ---- kernel launchers (wf_kernels_*.hip).  mode: 0 = log_pdf, 1 = psi, 2 = flow only (u, logdet)
int launch_scalar(const ModelDev& md, const ModelDev* md_dev, int mode, const float* x, int64_t B, float* out, float* u,
                  int32_t* idx, void* stream);
int launch_scalar_layer(const ModelDev& md, const ModelDev* md_dev, int layer, const float* u_in, int64_t B, float* y,
                        float* logdet, int32_t* idx, void* stream);
int launch_scalar_inverse(const ModelDev& md, const ModelDev* md_dev, const float* u, int64_t B, float* x, int exact, void* stream);
int launch_scalar_sample(const ModelDev& md, const ModelDev* md_dev, unsigned long long seed, int64_t B, float* x, float* latent,
                         int exact, void* stream);
struct Protons {
    float pos[8];
    int n;
};
// local energy of large batches on the matrix cores (wf_kernels_etile.hip): D = 2, <= 64 bases, mean box, IMADE + Waveflow prior, ungated
int64_t energy_tile_floats(int64_t B);
bool energy_tile_fused(const MfmaDev* mdev);
// local energy of large batches beyond two particles (wf_kernels_etile_dir.hip): D = 3 .. 8, <= 32 bases, mean box, IMADE + Waveflow prior, ungated;
// one coordinate direction at a time, Taylor triples on the matrix cores
bool energy_dir_capable(const MfmaDev* mdev);
int64_t energy_dir_floats(int64_t B, int D);
int launch_energy_dir(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* x, int64_t B, const Protons& pr,
                      float* hpsi, float* psi, float* lap, float* ws, void* stream);
// parameter gradients of psi and its Laplacian on the matrix cores (two-particle family, <= 64 bases; wf_kernels_etile.hip: k_efused, k_ebwd per net, k_egrad_reduce)
bool energy_vjp_capable(const MfmaDev* mdev);
int64_t energy_vjp_floats_per_walker(int n_nets);
int64_t energy_vjp_fixed_floats(int n_nets, int nbk);
int energy_vjp_gacc_floats(int n_nets, int nbk);
int launch_energy_vjp(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* x, int64_t B, int mode, const float* w_psi,
                      const float* w_lap, const Protons& pr, float running_avg, const float* running_avg_dev, float inv_count, float* e_loc, float* ws,
                      float* gacc, int accumulate, void* stream);
int launch_energy_vjp_finish(const float* gacc, int n_nets, int nbk, const int* offs, const float* c2, float* flat, int64_t n_params, void* stream);   // the one-kernel form applies (nets resident in LDS; WF_ENERGY_FUSED=0 switches it off per call)
int launch_energy_tile(const MfmaDev* mdev, const ModelDev& md, const float* tabI4, const float* tabP4, const float* fk_nat, const float* x, int64_t B,
                       const Protons& pr, float* hpsi, float* psi, float* lap, float* ws, void* stream, float* st_out = nullptr);
// The sweeps run over a coefficient ring (wf_ring.h).  kind 0: R1 (first order); 1: R3 (one sample per walker and direction,
// 3 coefficients); 2: RF<K> (one sample per walker and block of K directions, K + 2 coefficients); 3: RF<D> (one sample per walker).
// The taped sweeps use kind 2 with K = D up to 5 coordinates; beyond, the 8..10 live floats per value of RF<D> spill hundreds of registers
// in the reverse sweep and two blocks of 3 or 4 directions are ~20 % faster.  The untaped forward sweep of wf_hamiltonian_fwd is fastest
// with the whole walker in one sample (kind 3) for every D (scratch/energy_ab.py, scratch/grad_ab.py).
#ifndef WF_RF_BLOCK_6
#define WF_RF_BLOCK_6 3
#endif
#ifndef WF_RF_BLOCK_78
#define WF_RF_BLOCK_78 4
#endif
constexpr int rf_block(int D) { return D <= 5 ? D : (D == 6 ? WF_RF_BLOCK_6 : WF_RF_BLOCK_78); }
inline int ring_coefs(int D, int kind) { return kind == 0 ? 1 : (kind == 1 ? 3 : (kind == 2 ? rf_block(D) + 2 : D + 2)); }
inline int ring_samples(int D, int kind) { return (kind == 0 || kind == 3) ? 1 : (kind == 1 ? D : (D + rf_block(D) - 1) / rf_block(D)); }
// reverse pass (wf_kernels_grad.hip)
int grad_ws_rows(int D, int nbp);
int wgrad_partial_floats(int n_nets, int64_t net_img_floats);
int launch_wgrad(int D, int nbp, int ring_kind, int n_nets, int64_t n_samples, const float* ws, float* partial, int accumulate, float* grad_img,
                 int64_t net_img_floats, int* split_out, void* stream);
int launch_wave_fwd(const ModelDev& md, const ModelDev* md_dev, int ring_kind, const float* tabI4, const float* tabP4, const float* fk_nat,
                    const float* x, int64_t B, float* ws, float* tails, int taped, void* stream);
int launch_wave_bwd(const ModelDev& md, const ModelDev* md_dev, int mode, int ring_kind, const float* tabI4, const float* tabP4,
                    const float* fk_nat, int64_t B, const float* w1, const float* w2, float* ws, const float* tails, float* zws, void* stream);
// gated heads: zero_params gradient.  zws [n_samples][n_rows] (k_wave_bwd) -> zgrad [n_rows] (+)=, in a fixed order; then into the flat gradient
int launch_zgrad_reduce(const float* zws, int64_t n_samples, int n_rows, int accumulate, float* zpart, float* zgrad, void* stream);
int launch_zgrad_scatter(const float* zgrad, int n_rows, const int32_t* zmap, const int32_t* zraw_off, const float* plain, float* grad_flat, void* stream);
int launch_wave_energy(const ModelDev& md, const ModelDev* md_dev, const float* tabI4, const float* tabP4, const float* fk_nat, const float* x,
                       int64_t B, const Protons& pr, float* hpsi, float* psi, float* lap, float* tail_ws, void* stream);
// staged inverse / sampler of large two-particle batches (wf_kernels_etile.hip: conditioners on the matrix cores, one lane per walker elsewhere)
bool tile_sample_capable(const MfmaDev* mdev);
int64_t tile_sample_floats(int64_t B, int nbk);
int launch_tile_sample(const MfmaDev* mdev, const ModelDev& md, const float* tabI0, const float* tabP0, const float* fk_nat, int draw, unsigned long long seed,
                       const float* u, int64_t B, float* x, float* latent, int exact, const unsigned long long* seed_offset_dev, int64_t b0, float* ws, void* stream);
int launch_wave_sample(const ModelDev& md, const ModelDev* md_dev, const float* tabI4, const float* tabP4, const float* fk_nat, int draw,
                       unsigned long long seed, const float* u, int64_t B, float* x, float* latent, int exact,
                       const unsigned long long* seed_offset_dev, void* stream);
int launch_tail_out(const ModelDev& md, int mode, const float* tails, int64_t B, float* out, float* u, void* stream, float* w_out = nullptr,
                    float w_value = 0.0f);
int launch_wave_eval(const ModelDev& md, const ModelDev* md_dev, const float* tabI4, const float* tabP4, const float* fk_nat, int mode,
                     const float* x, int64_t B, float* out, float* u, float* tail_ws, void* stream);
int64_t wave_tail_floats(int D, int ring_kind);   // per walker
int launch_energy_seeds(int D, int ring_kind, const float* tails, const float* x, int64_t B, unsigned constrained_mask, const Protons& pr, float running_avg,
                        const float* running_avg_dev, float inv_count, float* e_loc, float* w_psi, float* w_lap, void* stream);
int launch_energy_out(int D, int ring_kind, const float* tails, const float* x, int64_t B, unsigned constrained_mask, const Protons& pr, float* hpsi, float* psi,
                      float* lap, void* stream);
// One entry of a device weight image as a function of the flat parameter vector:
//   kind & 0x0F == 0: image float [dst]  = src >= 0 ? (float)(scale * flat[src]) : (float)scale
//   kind & 0x0F == 1: image halves [dst], [dst_lo] = fp16 pair (hi, lo) of that value, hi + lo = value to 2^-25
//   kind & 0x10: |flat[src]| instead of flat[src]
//   kind >> 8: which image (0 plain, 1 wave, 2 mfma)
struct PackRec {
    int32_t src, kind;
    uint32_t dst, dst_lo;
    double scale;
};
int launch_pack(const float* flat_dev, const PackRec* recs, int64_t n, void* plain, void* wave, void* mfma, void* stream);
int launch_adam(float* params, const float* grad, float* m, float* v, int64_t n, int64_t step, float step_size, float b1, float b2, float eps,
                const unsigned long long* step_dev, void* stream);
// distributed training step (wf_vqmc_train_step_local / _apply): packed fp64 all-reduce buffer, Adam from it, loss-ring push
int launch_pack_reduce_buffer(const float* partial, int split, int64_t n_img, const int32_t* inv, const float* grad, int64_t n_params, double* red,
                              void* stream);
int launch_adam_reduced(float* params, const double* red, float* m, float* v, int64_t n, float step_size, float b1, float b2, float eps,
                        const unsigned long long* step_dev, void* stream);
int launch_ring_push(const double* sums, double* ring, int ring_len, unsigned long long* counter, void* stream);
// Adam with the gradient read straight from k_wgrad's per-split partial images (the gather of launch_grad_gather_partials inlined)
int launch_adam_partials(float* params, const float* partial, int split, int64_t n_img, const int32_t* inv, float* m, float* v, int64_t n,
                         float step_size, float b1, float b2, float eps, const unsigned long long* step_dev, void* stream);
int launch_grad_gather(const float* grad_img, const int32_t* inv, int64_t n_params, float* grad_flat, void* stream);
int launch_grad_gather_partials(const float* partial, int split, int64_t n_img, const int32_t* inv, int64_t n_params, float* grad_flat, void* stream);
int launch_vqmc_seeds(const float* x, int64_t B, int D, const Protons& pr, const float* hpsi, const float* psi, float running_avg,
                      float inv_count, float* e_loc, float* w_psi, float* w_lap, const float* running_avg_dev, void* stream);
int launch_rqs(const float* x, const float* uw, const float* uh, const float* ud, int64_t N, int K, int n_deriv, int inverse,
               float left, float right, float bottom, float top, float* y, float* ld, int32_t* bin, void* stream);
// the coupling stack as a model (wf_kernels_rqs.hip: k_nsc_model): L layers of (f1, f2) conditioner parameters in stax.Dense leaf order
struct NscModelDev {
    int D, L, K, hidden, prior_kind, reverse;
    float tail, normal_offset;
    const float* params;     // device, [L][2][net_floats]
    int64_t net_floats;
};
bool nsc_model_built(int D, int K, int hidden);
int launch_nsc_model(const NscModelDev& md, int mode, const float* x, int64_t B, float* out, float* u, void* stream);
int launch_nsc_latent(int prior_kind, int D, unsigned long long seed, int64_t B, float* z, void* stream);
int64_t nsc_workspace_floats(int64_t B, int dim, int K);
int launch_nsc(const float* x, int64_t B, int dim, int K, float tail, int hidden, const float* params, int inverse, float* y, float* logdet,
               float* ws, void* stream);
int launch_block_sums(const float* v, int64_t B, double* out, void* ws, int64_t ws_bytes, void* stream, double* ring = nullptr, int ring_len = 0,
                      unsigned long long* counter = nullptr);
int64_t block_sums_ws_bytes(int64_t B);
// walkers in any order: rows sorted ascending (xs may be null) and / or their inversion counts (inv may be null); psi *= (-1)^inv
constexpr int kModePresort = 4;   // bit of launch_mfma's mode: sort each row in registers, psi gets (-1)^inversions (wf_mfma_impl.h)
int launch_sort_rows(const float* x, int64_t B, int D, float* xs, int32_t* inv, void* stream);
int launch_apply_sign(float* v, const int32_t* inv, int64_t B, void* stream);

void set_hip_error(int e);

// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel, remembered per device: `slots` is that kernel's own table (zero-initialised
// static, one entry per device ordinal).  The attribute is per device and the call is idempotent, so two threads racing on one entry
// both set a value >= what they need; a model on a second GPU of the process gets its own call.
constexpr int kMaxDevices = 64;
struct DynLdsSlots { int bytes[kMaxDevices]; };
int ensure_dynamic_lds(const void* kernel, int lds_bytes, DynLdsSlots* slots);

}  // namespace wf
