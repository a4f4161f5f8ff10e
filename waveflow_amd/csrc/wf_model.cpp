// wf_model.cpp -- wf_model: host-side model build (tables, masks, device images) and the C ABI.
//
// Reference behaviour mirrored here (paths relative to /root/reference/waveflow):
//   masks / MaskedDense / tiling ........ model_factory.py:8-35, 72-82
//   parameter pytree order .............. wavefunctions.py:110, distributions.py:192, made.py:38,102
//   table dtype ......................... jnp.array(np.load(...)) => fp32 (isplines_jax.py:131)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "wf_internal.h"

namespace wf {

static thread_local int g_last_hip = 0;
void set_hip_error(int e) { g_last_hip = e; }

int ensure_dynamic_lds(const void* kernel, int lds_bytes, DynLdsSlots* slots) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = -1;
    int* slot = dev >= 0 ? &slots->bytes[dev] : nullptr;
    const int have = slot ? __atomic_load_n(slot, __ATOMIC_RELAXED) : 0;
    if (slot && lds_bytes <= have) return WF_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
        set_hip_error((int)e);
        return WF_ERR_HIP;
    }
    if (slot) {   // keep the maximum (another thread may have stored a larger request meanwhile)
        int cur = have;
        while (cur < lds_bytes && !__atomic_compare_exchange_n(slot, &cur, lds_bytes, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    }
    return WF_OK;
}

#define WF_HIP(call)                                   \
    do {                                               \
        hipError_t e_ = (call);                        \
        if (e_ != hipSuccess) {                        \
            wf::set_hip_error((int)e_);                \
            return WF_ERR_HIP;                         \
        }                                              \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// get_masks, model_factory.py:8-19
static inline int deg_in(int a) { return a; }
static inline int deg_hidden(int a, int D) { return a % (D - 1); }
static inline int deg_out(int d) { return d - 1; }

struct NetLayout {
    int n_out;       // bases per dimension (IMADE/prior) or 2 (MADE)
    bool has_zero;   // trailing zero_params[D][n_out] leaf (model_factory.py:84-87)
    int64_t offset;  // offset of W0 in the flat parameter vector
    int64_t count;
};

}  // namespace wf

struct wf_model {
    wf_model_desc desc{};
    int device = 0;
    int kernel_kind = WF_KERNEL_AUTO;
    int i_nb = 0, p_nb = 0;
    int nbp = 32;
    std::vector<wf::NetLayout> nets;  // flow layers then (optionally) the prior net
    int64_t n_params = 0;
    bool params_set = false;
    // a deferred training step (wf_train_state.defer_eval_tables) refreshed the weight images only: the MFMA image then holds unfolded
    // biases and the composite dimension-0 tables are those of older parameters.  Cleared by the next full refresh; while it is set,
    // wf_hamiltonian_fwd stays on the wave sweeps (which read neither) -- include/waveflow_hip.h promises it needs no refresh.
    bool eval_tables_stale = false;
    // fp16 range of the matrix-core operand images: k_fold_bias leaves one flag per net in d_f16_ovf at every upload; the host copy arrives through
    // pinned memory behind ovf_event (f16_overflow() below waits for it when an entry point needs the answer before the upload has finished).
    int* d_f16_ovf = nullptr;
    int* h_f16_ovf = nullptr;
    hipEvent_t ovf_event = nullptr;
    bool ovf_pending = false;
    bool f16_overflow = false;
    bool local_step_tile = false;    // the last wf_vqmc_train_step_local ran at a batch size of the matrix-core sampler / gradient: _apply refreshes every table
    wf::ModelDev dev{};
    std::vector<void*> allocs;
    // device images
    float* d_plain = nullptr;  // all NetPlain arrays, one allocation
    int64_t plain_floats = 0;
    std::vector<int64_t> plain_off;  // per net: offset of W0 in d_plain
    float* d_mfma = nullptr;
    int64_t mfma_floats = 0;
    wf::ModelDev* d_dev = nullptr;  // device copy of `dev`
    bool mfma_ok = false;           // the MFMA kernel covers this configuration
    wf::MfmaDev mdev{};
    wf::MfmaDev* d_mdev = nullptr;
    std::vector<float> mfma_consts;  // constants block of the LDS image (host copy)
    int64_t mfma_lds_floats = 0;
    float* d_tabI = nullptr;
    float* d_tabP = nullptr;
    float* d_fk_nat = nullptr;       // [2][32] natural-order row factors (I layers, prior) for k_prepare_dim0
    void* d_comp = nullptr;          // composite tables [n_nets][n_mesh] float4
    const float* d_tabI4 = nullptr;  // [4][n_mesh][nbp]: I-spline derivative orders 0..3 (local energy)
    const float* d_tabP3 = nullptr;  // [4][n_mesh][nbp]: orthogonal-B derivative orders 0..3 (the energy uses 0..2)
    const float* d_tabB0 = nullptr;  // [n_mesh][nbp]: the PLAIN B-splines (order 0): the staged sampler's band-limited evaluation of a proposal
    // the same two tables regrouped for the lane-per-walker heads of wf_kernels_etile.hip (nbp == 32 only): [n_mesh][8 row chunks][4 orders][4 rows],
    // so that the four orders of four rows of one mesh point are one 64-byte segment
    const float* d_tabI4c = nullptr;
    const float* d_tabP4c = nullptr;
    float* d_flat = nullptr;         // staging copy of a host parameter vector (wf_model_set_params)
    wf::PackRec* d_pack = nullptr;   // descriptions of every entry of the plain, wave and mfma images
    int64_t n_pack = 0;
    float* d_scratch = nullptr;      // private scratch of wf_hamiltonian_fwd (grown on demand)
    int64_t scratch_floats = 0;
    float* d_wave = nullptr;         // NetWave images
    float* d_grad_fk = nullptr;      // [2][64] natural-order row factors for the reverse pass (flow rows, prior rows)
    float* d_egacc = nullptr;        // [n_nets][6400] gradient blocks of the matrix-core gradient path, accumulated over the chunks of a batch
    bool wave_ok = false;            // the wave-cooperative sweeps and sampler cover this model (homogeneous constraints, gated heads included; > 32 bases: D <= 4)
    // boundary conditions as a linear map on the coefficient vector (bc_map below): column sums a~ of A, per spline (I layers / prior);
    // bc_*_ok: homogeneous (no constant term) and every column with a~_j == 0 is entirely zero -> the table-driven kernels apply
    bool is_nsc = false;             // layer_kind WF_LAYER_NSC: the coupling stack (k_nsc_model), none of the conditioner-net machinery
    float* d_nsc = nullptr;          // its parameters on the device (the model's own copy)
    wf::NscModelDev nsc{};
    std::vector<double> bc_i_colsum, bc_p_colsum;
    std::vector<float> p_cb;          // constant term of the B prior's boundary map times ob_to_b, natural order [nbp] (empty: homogeneous constraints)
    bool bc_i_ok = true, bc_p_ok = true;
    bool bc_i_plain = false;   // I layers: the same of their boundary map (the rows of the evaluation table are then the plain I-splines: exactly 1 left of a band of k + 1, 0 right of it)
    bool bc_p_plain = false;   // B prior: the boundary map only zeroes coefficients (a masked identity, no constant term): (o keep) ARE the plain B-spline coefficients of c
    bool grad_psi_ok = false;        // wf_psi_vjp (Waveflow prior, IMADE layers)
    int ring2 = 2;                   // coefficient ring of the second-order sweeps (ring_coefs, wf_internal.h): 2 = RF, 1 = R3
    int32_t* d_grad_map = nullptr;   // [n_params]: forward-image entry (over all nets) that holds each parameter, -1 = none
    float* d_grad_partial = nullptr; // per-split partial gradient images of k_wgrad
    float* d_grad_img = nullptr;     // [n_nets * fwd image floats]: gradient accumulator in forward-image layout
    // gated heads: gradient of the zero_params leaves (rows = n_nets * passes * 64 head lanes of the wave layout)
    int z_rows = 0;
    float* d_zpart = nullptr;        // [64 splits][z_rows]
    float* d_zgrad = nullptr;        // [z_rows]
    int32_t* d_zmap = nullptr;       // row -> index of its leaf entry in the flat vector (-1: padding lane / ungated net)
    int32_t* d_zraw_off = nullptr;   // row -> offset of the raw leaf value in the plain image for |z| heads (-1: signed head)
};

namespace wf {

// forward-orientation part of a net's image (W0, b0, W1t, b1, W2t, b2): also the layout of the gradient accumulator
static int64_t plain_fwd_floats(int D, int nbp) {
    return (int64_t)D * kHidden + kHidden + (int64_t)kHidden * kHidden + kHidden + (int64_t)D * nbp * kHidden + (int64_t)D * nbp;
}
// ... followed by W1n, W2n, zero
static int64_t plain_net_floats(int D, int nbp) {
    return plain_fwd_floats(D, nbp) + (int64_t)kHidden * kHidden + (int64_t)kHidden * D * nbp + 2 * (int64_t)D * nbp;   // ..., zero, zero_raw
}

static int check_bc(const wf_bc& bc, int nb) {
    if (bc.n < 0 || bc.n > WF_MAX_BC) return WF_ERR_INVALID;
    for (int i = 0; i < bc.n; ++i)
        if (bc.n_derivative[i] < 0 || bc.n_derivative[i] > 3 || bc.n_derivative[i] >= nb) return WF_ERR_INVALID;
    return WF_OK;
}

// Dense-row device table: [n_orders][n_mesh][nbp], fp32 cast of the fp64 table.
static void pack_rows(const std::vector<double>& t64, int nb, int n_mesh, int n_orders, int nbp, std::vector<float>& out) {
    out.assign((size_t)n_orders * n_mesh * nbp, 0.0f);
    for (int nd = 0; nd < n_orders; ++nd)
        for (int i = 0; i < nb; ++i)
            for (int m = 0; m < n_mesh; ++m)
                out[((size_t)nd * n_mesh + m) * nbp + i] = (float)t64[((size_t)nd * nb + i) * n_mesh + m];
}

// Boundary-condition constants (enforce_boundary_conditions: isplines_jax.py:158-194,
// bsplines_jax.py:173-199, msplines_jax.py:156-184): X_cached(0.0, j, nd) == T[nd][j][0] and
// X_cached(1.0, j, nd) == T[nd][j][n_mesh-1] in fp32.
static void fill_bc(SplineDev& s, const wf_bc& left, const wf_bc& right, const std::vector<double>& t64, int nb, int n_mesh) {
    auto T = [&](int nd, int j, int m) { return (float)t64[((size_t)nd * nb + j) * n_mesh + m]; };
    s.n_left = left.n;
    s.n_right = right.n;
    for (int p = 0; p < left.n; ++p) {
        const int nd = left.n_derivative[p];
        s.left_nd[p] = nd;
        s.left_val[p] = left.value[p];
        for (int j = 0; j < nd; ++j) s.left_prev[p][j] = T(nd, j, 0);
        s.left_value[p] = T(nd, nd, 0);
    }
    for (int p = 0; p < right.n; ++p) {
        const int nd = right.n_derivative[p];
        s.right_nd[p] = nd;
        s.right_val[p] = right.value[p];
        for (int j = 0; j < nd; ++j) s.right_prev[p][j] = T(nd, nb - j - 1, n_mesh - 1);
        s.right_value[p] = T(nd, nb - nd - 1, n_mesh - 1);
    }
}

// ---- boundary conditions as a linear map.  enforce_boundary_conditions (isplines_jax.py:166-190, msplines_jax.py:155-180,
// bsplines_jax.py:176-189) overwrites coefficient nd (left) / nb-1-nd (right) of every constraint {nd: value} with
// (value - sum_{j<nd} T^(nd)_j(end) c_j) / T^(nd)_nd(end), in dictionary order, before the final normalisation: c' = A c + b with A, b
// fixed per model.  With b == 0 (every value 0; the I-spline's right {0: 1} zeroes the last coefficient, isplines_jax.py:174-179) the
// normalised spline  sum_j c'_j T_j(x) / sum_j c'_j  equals  sum_j (c_j a~_j) T^_j(x) / sum_j (c_j a~_j)  with a~ = A^T 1 (column sums)
// and T^_j = (A^T T)_j / a~_j: the same expression the kernels evaluate for "zero the first / last coefficient" (a~ in {0, 1}, T^ = T),
// so the table-driven kernels (MFMA, wave sweeps, gradients) cover every homogeneous dictionary through their tables and row factors
// alone.  The per-walker scalar kernel keeps the literal sequence (enforce_bc, wf_scalar_impl.h) and also covers the B-spline prior with b != 0.
static void bc_apply(const SplineDev& s, int kind, int nb, std::vector<double>& c) {
    for (int p = 0; p < s.n_left; ++p) {
        const int nd = s.left_nd[p];
        double sum = 0;
        for (int j = 0; j < nd; ++j) sum += (double)s.left_prev[p][j] * c[j];
        c[nd] = ((double)s.left_val[p] - sum) / (double)s.left_value[p];
    }
    for (int p = 0; p < s.n_right; ++p) {
        const int nd = s.right_nd[p];
        if (kind == WF_SPLINE_I && nd == 0) { c[nb - 1] = 0.0; continue; }
        double sum = 0;
        for (int j = 0; j < nd; ++j) sum += (double)s.right_prev[p][j] * c[nb - 1 - j];
        c[nb - nd - 1] = ((double)s.right_val[p] - sum) / (double)s.right_value[p];
    }
}
// -> A [nb][nb] (c' = A c), column sums; false when the map keeps a constant term or has a column that sums to zero without being zero
static bool bc_map(const SplineDev& s, int kind, int nb, std::vector<double>& A, std::vector<double>& colsum, std::vector<double>* bconst = nullptr) {
    A.assign((size_t)nb * nb, 0.0);
    colsum.assign(nb, 0.0);
    std::vector<double> c(nb, 0.0);
    bc_apply(s, kind, nb, c);
    bool ok = true, constant = false;
    for (int i = 0; i < nb; ++i) constant = constant || c[i] != 0.0;
    // A constant term b (a constraint with a non-zero value).  The I- and M-spline coefficients enter the constraints normalised
    // (remove_bias ends with p / sum p: isplines_jax.py:196-202, msplines_jax.py:186-192), so b = b (1^T c) and the map is the linear
    // A + b 1^T on them.  The B-spline prior's weights reach the constraints divided by their signed sum S (model_factory.py:69) and are
    // normalised only afterwards: w' = (A o + S b) / S, so the kernels carry b as a separate term (bconst; round 3) scaled by S = sum o.
    const bool fold = constant && (kind == WF_SPLINE_I || kind == WF_SPLINE_M);
    if (constant && !fold) {
        if (bconst) *bconst = c;
        else ok = false;
    }
    for (int j = 0; j < nb; ++j) {
        std::vector<double> e(nb, 0.0);
        e[j] = 1.0;
        bc_apply(s, kind, nb, e);
        bool all_zero = true;
        for (int i = 0; i < nb; ++i) {
            const double a = e[i] - c[i] + (fold ? c[i] : 0.0);
            A[(size_t)i * nb + j] = a;
            colsum[j] += a;
            all_zero = all_zero && a == 0.0;
        }
        if (all_zero) colsum[j] = 0.0;
        else if (std::fabs(colsum[j]) < 1e-9 || (fold && colsum[j] < 0.0)) ok = false;   // (the kernels' row factors of a folded map stay positive)
    }
    return ok;
}
// rows of a table indexed by the coefficient ([orders][nb][n_mesh] fp64, or [nb][cols] with n_mesh := cols, orders := 1): X^_j = (A^T X)_j / a~_j
static void bc_transform_rows(const std::vector<double>& A, const std::vector<double>& colsum, int nb, int orders, int n_mesh, std::vector<double>& t) {
    std::vector<double> out(t.size(), 0.0);
    for (int nd = 0; nd < orders; ++nd)
        for (int j = 0; j < nb; ++j) {
            if (colsum[j] == 0.0) continue;
            double* o = &out[((size_t)nd * nb + j) * n_mesh];
            for (int i = 0; i < nb; ++i) {
                const double a = A[(size_t)i * nb + j];
                if (a == 0.0) continue;
                const double* src = &t[((size_t)nd * nb + i) * n_mesh];
                for (int m = 0; m < n_mesh; ++m) o[m] += a * src[m];
            }
            for (int m = 0; m < n_mesh; ++m) o[m] /= colsum[j];
        }
    t.swap(out);
}

template <class T>
static int dev_alloc(wf_model* m, T** p, size_t count) {
    void* q = nullptr;
    WF_HIP(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    // WF_POISON=1 (the test suite sets it): fresh device memory starts as NaN patterns, so that a kernel reading anything
    // it or the host has not written shows up as NaN instead of passing by luck on zero-filled pages
    if (getenv("WF_POISON")) (void)hipMemset(q, 0xFF, std::max<size_t>(count, 1) * sizeof(T));
    m->allocs.push_back(q);
    *p = (T*)q;
    return WF_OK;
}

static int upload_table(wf_model* m, const std::vector<float>& h, const float** out) {
    float* d = nullptr;
    int rc = dev_alloc(m, &d, h.size());
    if (rc) return rc;
    WF_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = d;
    return WF_OK;
}

static int mfma_prepare(wf_model* m, const std::vector<double>& i64, const std::vector<double>& p64, const std::vector<double>& o2b);
// [4][n_mesh][nbp] -> [n_mesh][nbp / 4 chunks][4 orders][4 rows] (see d_tabI4c), + the chunks' support bounds
static int upload_chunked(wf_model* m, const std::vector<float>& rows4, int n_mesh, int nbp, const float** out) {
    const int chunks = nbp / 4, stride = nbp * 4;   // floats per mesh point
    std::vector<float> c((size_t)n_mesh * stride);
    for (int mm = 0; mm < n_mesh; ++mm)
        for (int ch = 0; ch < chunks; ++ch)
            for (int k = 0; k < 4; ++k)
                for (int q = 0; q < 4; ++q) c[(((size_t)mm * chunks + ch) * 4 + k) * 4 + q] = rows4[((size_t)k * n_mesh + mm) * nbp + 4 * ch + q];
    // behind the table: int32 [chunks][lo, hi], the support bounds of the chunks (as piece_bounds below: a chunk read at clamp(m, lo, hi)
    // returns the bits of the chunk at m; the head code of the energy path clamps, and walkers outside a chunk's support share lines)
    std::vector<int32_t> bnd(2 * chunks);
    for (int ch = 0; ch < chunks; ++ch) {
        auto same = [&](int a, int b) { return memcmp(&c[((size_t)a * chunks + ch) * 16], &c[((size_t)b * chunks + ch) * 16], 16 * sizeof(float)) == 0; };
        int lo = 0, hi = n_mesh - 1;
        if (!getenv("WF_MFMA_NO_BAND")) {
            while (lo + 1 < n_mesh && same(lo + 1, 0)) ++lo;
            while (hi - 1 >= 0 && same(hi - 1, n_mesh - 1)) --hi;
        }
        bnd[2 * ch] = lo;
        bnd[2 * ch + 1] = hi;
    }
    c.resize(c.size() + 2 * chunks);
    memcpy(&c[(size_t)n_mesh * stride], bnd.data(), 2 * chunks * sizeof(int32_t));
    return upload_table(m, c, out);
}
static int grad_prepare(wf_model* m);
static int64_t wave_net_floats(int D, int nbp);
static int wave_passes(int D, int nbp) { return nbp == 32 ? (D + 1) / 2 : D; }   // output passes: 2 dimensions x 32 rows, or 1 x 64
}  // namespace wf
static int ensure_scratch(const wf_model* cm, int64_t floats);
static constexpr int kTapedLaplacianMaxD = 8;    // largest D whose reverse sweep runs in RF (measured, scratch/grad_ab.py)
static constexpr int64_t kWaveEvalMax = 6144;   // measured crossover ~7000 walkers (scratch/crossover.py)
static constexpr int64_t kGradTileMin = 16384;    // psi / Laplacian gradients: the matrix-core path (k_efused, k_ebwd) from here on
static constexpr int64_t kEnergyTileMin = 16384; // H psi: the tile path (8 launches, staged weight images) from here on
static constexpr int64_t kEnergyTileChunk = (int64_t)1 << 19;   // walkers per pass of the tile path (WF_ENERGY_TILE_CHUNK; 2^20 walkers: 1.20 ms in two passes, 1.30 in one, 1.34 in four)
namespace wf {

// layer_kind WF_LAYER_NSC: Flow(Serial((NeuralSplineCoupling [, Reverse]) x L), Normal | Uniform)
static int nsc_build(wf_model* m) {
    const wf_model_desc& d = m->desc;
    const int D = d.n_dim, K = d.nsc_bins, h = d.nsc_hidden;
    if (D < 2 || D > WF_MAX_DIM || (D % 2) != 0) return WF_ERR_INVALID;          // (dim // 2 coordinates per half, neural_splines.py:256)
    if (d.n_flow_layers < 1 || d.n_flow_layers > kMaxLayers) return WF_ERR_INVALID;
    if (K < 2 || h < 1 || !(d.nsc_tail_bound > 0.0f)) return WF_ERR_INVALID;
    if (d.prior_kind != WF_PRIOR_NORMAL && d.prior_kind != WF_PRIOR_UNIFORM) return WF_ERR_UNSUPPORTED;
    if (!nsc_model_built(D, K, h)) return WF_ERR_UNSUPPORTED;
    const int dh = D / 2, per = 3 * K - 1;
    const int64_t net_floats = (int64_t)dh * h + h + (int64_t)h * h + h + (int64_t)h * per * dh + (int64_t)per * dh;
    m->n_params = net_floats * 2 * d.n_flow_layers;
    int rc = dev_alloc(m, &m->d_flat, (size_t)m->n_params);
    if (rc) return rc;
    rc = dev_alloc(m, &m->d_nsc, (size_t)m->n_params);
    if (rc) return rc;
    m->dev = ModelDev{};
    m->dev.D = D;
    m->dev.prior_kind = d.prior_kind;
    m->nbp = 32;
    m->nsc = NscModelDev{D, d.n_flow_layers, K, h, d.prior_kind, d.nsc_reverse != 0 ? 1 : 0, d.nsc_tail_bound, d.normal_offset, m->d_nsc, net_floats};
    m->is_nsc = true;
    return WF_OK;
}

static int model_build(wf_model* m) {
    const wf_model_desc& d = m->desc;
    const int D = d.n_dim;
    if (d.layer_kind == WF_LAYER_NSC) return nsc_build(m);
    if (D < 2 || D > WF_MAX_DIM) return WF_ERR_INVALID;
    if (d.hidden != kHidden) return WF_ERR_UNSUPPORTED;
    if (d.n_flow_layers < 0 || d.n_flow_layers > kMaxLayers) return WF_ERR_INVALID;
    if (d.layer_kind != WF_LAYER_IMADE && d.layer_kind != WF_LAYER_MADE) return WF_ERR_INVALID;
    if (d.box_kind < WF_BOX_NONE || d.box_kind > WF_BOX_FIRST) return WF_ERR_INVALID;
    if (d.prior_kind < WF_PRIOR_WAVEFLOW || d.prior_kind > WF_PRIOR_NORMAL) return WF_ERR_INVALID;
    if (d.n_mesh < 2) return WF_ERR_INVALID;
    if (d.n_constrained_left < 0 || d.n_constrained_left > WF_MAX_DIM) return WF_ERR_INVALID;

    ModelDev& md = m->dev;
    md = ModelDev{};
    md.D = D;
    md.n_layers = d.n_flow_layers;
    md.layer_kind = d.layer_kind;
    md.box_kind = d.box_kind;
    md.box_L = d.box_size;
    md.i_reg = d.i_reg;
    md.prior_kind = d.prior_kind;
    md.normal_offset = d.normal_offset;
    md.reverse_tol = d.i_reverse_tol > 0.0f ? d.i_reverse_tol : 1.0f / (float)d.n_mesh;   // isplines_jax.py:89-90
    md.i_gate = (d.i_gate != 0 && d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0) ? 1 : 0;
    md.p_gate = (d.p_gate != 0 && (d.prior_kind == WF_PRIOR_WAVEFLOW || d.prior_kind == WF_PRIOR_MFLOW)) ? 1 : 0;
    for (int i = 0; i < d.n_constrained_left; ++i) {
        if (d.constrained_left[i] < 0 || d.constrained_left[i] >= D) return WF_ERR_INVALID;
        md.constrained_mask |= 1u << d.constrained_left[i];
    }

    // padded bases per dimension: 32 covers every shipped configuration; 64 e.g. the 33-knot ("32-bin") variant
    {
        int nb_max = 2;
        if (d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0) nb_max = std::max(nb_max, n_bases_of(WF_SPLINE_I, d.i_degree, d.i_knots));
        if (d.prior_kind == WF_PRIOR_WAVEFLOW) nb_max = std::max(nb_max, n_bases_of(WF_SPLINE_B, d.p_degree, d.p_knots));
        if (d.prior_kind == WF_PRIOR_MFLOW) nb_max = std::max(nb_max, n_bases_of(WF_SPLINE_M, d.p_degree, d.p_knots));
        m->nbp = nb_max <= 32 ? 32 : 64;
        if (nb_max > 64) return WF_ERR_UNSUPPORTED;
        if (m->nbp == 64 && D > 4) return WF_ERR_UNSUPPORTED;
    }
    md.nbp = m->nbp;
    std::vector<double> keep_i64, keep_p64, keep_o2b;
    // ---- tables
    if (d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0) {
        const int nb = n_bases_of(WF_SPLINE_I, d.i_degree, d.i_knots);
        if (d.i_degree < 1 || d.i_knots < 2 || nb < 2) return WF_ERR_INVALID;
        if (nb > m->nbp) return WF_ERR_UNSUPPORTED;
        int rc = check_bc(d.i_left, nb);
        if (rc) return rc;
        rc = check_bc(d.i_right, nb);
        if (rc) return rc;
        // right constraint {0: v}: the reference supports v == 1 only (isplines_jax.py:174-179)
        for (int p = 0; p < d.i_right.n; ++p)
            if (d.i_right.n_derivative[p] == 0 && d.i_right.value[p] != 1.0f) return WF_ERR_INVALID;
        std::vector<double> t64((size_t)4 * nb * d.n_mesh);
        rc = build_raw_table(WF_SPLINE_I, d.i_degree, d.i_knots, d.n_mesh, t64.data());
        if (rc < 0) return rc;
        std::vector<float> rows;
        pack_rows(t64, nb, d.n_mesh, 2, m->nbp, rows);
        rc = upload_table(m, rows, &md.isp.tab);
        if (rc) return rc;
        md.isp.nb = nb; md.isp.nbp = m->nbp; md.isp.n_mesh = d.n_mesh; md.isp.degree = d.i_degree;
        fill_bc(md.isp, d.i_left, d.i_right, t64, nb, d.n_mesh);
        {   // the table-driven kernels read the rows with the boundary map folded in (identical rows for zero-only constraints)
            std::vector<double> A;
            m->bc_i_ok = bc_map(md.isp, WF_SPLINE_I, nb, A, m->bc_i_colsum);
            m->bc_i_plain = m->bc_i_ok;
            for (int i = 0; i < nb && m->bc_i_plain; ++i)
                for (int j = 0; j < nb; ++j)
                    if (A[(size_t)i * nb + j] != ((i == j && m->bc_i_colsum[j] != 0.0) ? 1.0 : 0.0)) { m->bc_i_plain = false; break; }
            if (m->bc_i_ok) bc_transform_rows(A, m->bc_i_colsum, nb, 4, d.n_mesh, t64);
        }
        {   // derivative orders 0..3 for the wave kernels
            std::vector<float> rows4;
            pack_rows(t64, nb, d.n_mesh, 4, m->nbp, rows4);
            rc = upload_table(m, rows4, &m->d_tabI4);
            if (rc) return rc;
            rc = upload_chunked(m, rows4, d.n_mesh, m->nbp, &m->d_tabI4c);   // (the matrix-core energy path: D = 2 only, but the tables are small)
            if (rc) return rc;
        }
        m->i_nb = nb;
        keep_i64.swap(t64);
    }
    if (d.prior_kind == WF_PRIOR_WAVEFLOW) {
        const int nb = n_bases_of(WF_SPLINE_B, d.p_degree, d.p_knots);
        if (d.p_degree < 1 || d.p_knots < 2 || nb < 2) return WF_ERR_INVALID;
        if (nb > m->nbp) return WF_ERR_UNSUPPORTED;
        int rc = check_bc(d.p_left, nb);
        if (rc) return rc;
        rc = check_bc(d.p_right, nb);
        if (rc) return rc;
        std::vector<double> b64((size_t)4 * nb * d.n_mesh), ob64((size_t)4 * nb * d.n_mesh), o2b((size_t)nb * nb), b2o((size_t)nb * nb);
        rc = build_raw_table(WF_SPLINE_B, d.p_degree, d.p_knots, d.n_mesh, b64.data());
        if (rc < 0) return rc;
        rc = build_ortho_b(d.p_degree, d.p_knots, d.n_mesh, b64.data(), ob64.data(), b2o.data(), o2b.data());
        if (rc < 0) return rc;
        std::vector<float> rows;
        pack_rows(ob64, nb, d.n_mesh, 1, m->nbp, rows);
        rc = upload_table(m, rows, &md.psp.tab);
        if (rc) return rc;
        md.psp.nb = nb; md.psp.nbp = m->nbp; md.psp.n_mesh = d.n_mesh; md.psp.degree = d.p_degree;
        {
            std::vector<float> rows3;
            pack_rows(ob64, nb, d.n_mesh, 4, m->nbp, rows3);
            rc = upload_table(m, rows3, &m->d_tabP3);
            if (rc) return rc;
            rc = upload_chunked(m, rows3, d.n_mesh, m->nbp, &m->d_tabP4c);
            if (rc) return rc;
        }
        {
            std::vector<float> rowsB;
            pack_rows(b64, nb, d.n_mesh, 1, m->nbp, rowsB);
            rc = upload_table(m, rowsB, &m->d_tabB0);
            if (rc) return rc;
        }
        fill_bc(md.psp, d.p_left, d.p_right, b64, nb, d.n_mesh);  // BCs use the plain-B table, bsplines_jax.py:176-189
        std::vector<float> o2b32((size_t)m->nbp * m->nbp, 0.0f);   // full [nbp][nbp]: the wave kernels contract over all 32 rows
        for (int a = 0; a < nb; ++a)
            for (int j = 0; j < nb; ++j) o2b32[(size_t)a * m->nbp + j] = (float)o2b[(size_t)a * nb + j];
        rc = upload_table(m, o2b32, &md.ob_to_b);
        if (rc) return rc;
        {   // the constraints act on the net's outputs w before c = w @ ob_to_b: fold the map into the matrix's rows (row a = coefficient a)
            std::vector<double> A, bconst;
            m->bc_p_ok = bc_map(md.psp, WF_SPLINE_B, nb, A, m->bc_p_colsum, &bconst);
            m->bc_p_plain = m->bc_p_ok && bconst.empty();
            for (int i = 0; i < nb && m->bc_p_plain; ++i)
                for (int j = 0; j < nb; ++j)
                    if (A[(size_t)i * nb + j] != ((i == j && m->bc_p_colsum[j] != 0.0) ? 1.0 : 0.0)) { m->bc_p_plain = false; break; }
            if (m->bc_p_ok && !bconst.empty()) {   // constant term: cb = b @ ob_to_b (the matrix as it is, before the map is folded into its rows)
                m->p_cb.assign(m->nbp, 0.0f);
                for (int i = 0; i < nb; ++i) {
                    double acc = 0;
                    for (int a = 0; a < nb; ++a) acc += bconst[a] * o2b[(size_t)a * nb + i];
                    m->p_cb[i] = (float)acc;
                }
                rc = upload_table(m, m->p_cb, &md.p_cb);
                if (rc) return rc;
            }
            if (m->bc_p_ok) bc_transform_rows(A, m->bc_p_colsum, nb, 1, nb, o2b);
            for (int a = 0; a < nb; ++a)
                for (int j = 0; j < nb; ++j) o2b32[(size_t)a * m->nbp + j] = (float)o2b[(size_t)a * nb + j];
            rc = upload_table(m, o2b32, &md.ob_to_b_t);
            if (rc) return rc;
        }
        std::vector<float> b2o32((size_t)m->nbp * m->nbp, 0.0f);
        for (int a = 0; a < nb; ++a)
            for (int j = 0; j < nb; ++j) b2o32[(size_t)a * m->nbp + j] = (float)b2o[(size_t)a * nb + j];
        rc = upload_table(m, b2o32, &md.b_to_ob);
        if (rc) return rc;
        m->p_nb = nb;
        keep_p64.swap(ob64);
        keep_o2b.swap(o2b);
    } else if (d.prior_kind == WF_PRIOR_MFLOW) {
        const int nb = n_bases_of(WF_SPLINE_M, d.p_degree, d.p_knots);
        if (d.p_degree < 2 || d.p_knots < 2 || nb < 2) return WF_ERR_INVALID;
        if (nb > m->nbp) return WF_ERR_UNSUPPORTED;
        int rc = check_bc(d.p_left, nb);
        if (rc) return rc;
        rc = check_bc(d.p_right, nb);
        if (rc) return rc;
        std::vector<double> t64((size_t)4 * nb * d.n_mesh);
        rc = build_raw_table(WF_SPLINE_M, d.p_degree, d.p_knots, d.n_mesh, t64.data());
        if (rc < 0) return rc;
        std::vector<float> rows;
        pack_rows(t64, nb, d.n_mesh, 1, m->nbp, rows);
        rc = upload_table(m, rows, &md.psp.tab);
        if (rc) return rc;
        md.psp.nb = nb; md.psp.nbp = m->nbp; md.psp.n_mesh = d.n_mesh; md.psp.degree = d.p_degree;
        fill_bc(md.psp, d.p_left, d.p_right, t64, nb, d.n_mesh);
        {
            std::vector<double> A;
            m->bc_p_ok = bc_map(md.psp, WF_SPLINE_M, nb, A, m->bc_p_colsum);
            if (m->bc_p_ok) bc_transform_rows(A, m->bc_p_colsum, nb, 4, d.n_mesh, t64);
        }
        {
            std::vector<float> rows4;
            pack_rows(t64, nb, d.n_mesh, 4, m->nbp, rows4);
            rc = upload_table(m, rows4, &m->d_tabP3);
            if (rc) return rc;
        }
        m->p_nb = nb;
        keep_p64.swap(t64);
    }

    // ---- parameter layout (pytree leaf order)
    m->nets.clear();
    int64_t off = 0;
    auto add_net = [&](int n_out, bool has_zero) {
        NetLayout nl;
        nl.n_out = n_out;
        nl.has_zero = has_zero;
        nl.offset = off;
        nl.count = (int64_t)D * kHidden + kHidden + (int64_t)kHidden * kHidden + kHidden + (int64_t)kHidden * n_out * D +
                   (int64_t)n_out * D + (has_zero ? (int64_t)D * n_out : 0);
        off += nl.count;
        m->nets.push_back(nl);
    };
    for (int l = 0; l < d.n_flow_layers; ++l) {
        if (d.layer_kind == WF_LAYER_IMADE) add_net(m->i_nb, true);
        else add_net(2, false);
    }
    if (d.prior_kind == WF_PRIOR_WAVEFLOW || d.prior_kind == WF_PRIOR_MFLOW) add_net(m->p_nb, true);
    m->n_params = off;

    // ---- device weight images
    const int n_nets = (int)m->nets.size();
    m->plain_floats = plain_net_floats(D, m->nbp) * n_nets;
    int rc = dev_alloc(m, &m->d_plain, (size_t)m->plain_floats);
    if (rc) return rc;
    m->plain_off.assign(n_nets, 0);
    for (int n = 0; n < n_nets; ++n) {
        const int64_t base = plain_net_floats(D, m->nbp) * n;
        m->plain_off[n] = base;
        float* p = m->d_plain + base;
        NetPlain& np = md.nets[n];
        np.W0 = p; p += (int64_t)D * kHidden;
        np.b0 = p; p += kHidden;
        np.W1t = p; p += (int64_t)kHidden * kHidden;
        np.b1 = p; p += kHidden;
        np.W2t = p; p += (int64_t)D * m->nbp * kHidden;
        np.b2 = p; p += (int64_t)D * m->nbp;
        np.W1n = p; p += (int64_t)kHidden * kHidden;
        np.W2n = p; p += (int64_t)kHidden * D * m->nbp;
        np.zero = p; p += (int64_t)D * m->nbp;
        np.zero_raw = p;
    }
    {
        const int P = wave_passes(D, m->nbp);
        rc = dev_alloc(m, &m->d_wave, (size_t)(wave_net_floats(D, m->nbp) * n_nets));
        if (rc) return rc;
        for (int n = 0; n < n_nets; ++n) {
            float* p = m->d_wave + wave_net_floats(D, m->nbp) * n;
            NetWave& nw = md.wnets[n];
            nw.W0 = p; p += (int64_t)D * kHidden;
            nw.b0 = p; p += kHidden;
            nw.b1 = p; p += kHidden;
            nw.b2 = p; p += (int64_t)P * 64;
            nw.W1f = reinterpret_cast<const float4_t*>(p); p += 4096;
            nw.W1b = reinterpret_cast<const float4_t*>(p); p += 4096;
            nw.W2f = reinterpret_cast<const float4_t*>(p); p += (int64_t)P * 4096;
            nw.W2b = reinterpret_cast<const float4_t*>(p); p += (int64_t)P * 4096;
            nw.z = p;
        }
    }
    rc = dev_alloc(m, &m->d_dev, 1);
    if (rc) return rc;
    WF_HIP(hipMemcpy(m->d_dev, &md, sizeof(ModelDev), hipMemcpyHostToDevice));
    rc = mfma_prepare(m, keep_i64, keep_p64, keep_o2b);
    if (rc) return rc;
    return grad_prepare(m);
}

// Every entry of a device weight image is scale * flat[src] (or a constant): the images are described once per model as
// PackRec lists and filled on the device by k_pack (wf_kernels_grad.hip) whenever the parameters change.
struct ImageWriter {
    std::vector<PackRec>& out;
    uint32_t o;   // running float offset inside the image
    void f32(int64_t src, double scale = 1.0) { out.push_back(PackRec{(int32_t)src, 0, o++, 0u, src >= 0 ? scale : 0.0}); }
    void f32_abs(int64_t src) { out.push_back(PackRec{(int32_t)src, 0x10, o++, 0u, src >= 0 ? 1.0 : 0.0}); }
    void cst(double value) { out.push_back(PackRec{-1, 0, o++, 0u, value}); }
};
struct NetOffsets {   // flat-vector offsets of the leaves of one conditioner (model_factory.py:72-87 leaf order)
    int64_t W0, b0, W1, b1, W2, b2;
    int NO;
};
static NetOffsets net_offsets(const wf_model* m, int n) {
    const int D = m->desc.n_dim, H = kHidden;
    const NetLayout& nl = m->nets[n];
    NetOffsets q;
    q.NO = nl.n_out * D;
    q.W0 = nl.offset;
    q.b0 = q.W0 + (int64_t)D * H;
    q.W1 = q.b0 + H;
    q.b1 = q.W1 + (int64_t)H * H;
    q.W2 = q.b1 + H;
    q.b2 = q.W2 + (int64_t)H * q.NO;
    return q;
}

static bool net_has_sigmoid_head(const wf_model* m, int n);
static bool net_is_gated(const wf_model* m, int n) { return n == m->desc.n_flow_layers ? m->desc.p_gate != 0 : m->desc.i_gate != 0; }

// Masked, transposed weight image of net n (NetPlain), float offset `base` inside d_plain.
static void describe_plain_image(const wf_model* m, int n, uint32_t base, std::vector<PackRec>& out) {
    const int D = m->desc.n_dim, H = kHidden, nbp = m->nbp;
    const NetLayout& nl = m->nets[n];
    const NetOffsets q = net_offsets(m, n);
    ImageWriter w{out, base};
    // W0 * mask0: [D][H]
    for (int a = 0; a < D; ++a)
        for (int j = 0; j < H; ++j) w.f32(deg_hidden(j, D) >= deg_in(a) ? q.W0 + (int64_t)a * H + j : -1);
    for (int j = 0; j < H; ++j) w.f32(q.b0 + j);
    // (W1 * mask1)^T: [j out][a in]
    for (int j = 0; j < H; ++j)
        for (int a = 0; a < H; ++a) w.f32(deg_hidden(j, D) >= deg_hidden(a, D) ? q.W1 + (int64_t)a * H + j : -1);
    for (int j = 0; j < H; ++j) w.f32(q.b1 + j);
    // (W2 * tile(mask2))^T regrouped: [d][jb][a], reference output column c = jb*D + d (model_factory.py:59-60,81)
    for (int dd = 0; dd < D; ++dd)
        for (int jb = 0; jb < nbp; ++jb)
            for (int a = 0; a < H; ++a)
                w.f32((jb < nl.n_out && deg_out(dd) >= deg_hidden(a, D)) ? q.W2 + (int64_t)a * q.NO + (jb * D + dd) : -1);
    for (int dd = 0; dd < D; ++dd)
        for (int jb = 0; jb < nbp; ++jb) w.f32(jb < nl.n_out ? q.b2 + jb * D + dd : -1);
    // reverse-pass orientation: W1 * mask1 [a in][j out], W2 * mask2 [a in][d][jb]
    for (int a = 0; a < H; ++a)
        for (int j = 0; j < H; ++j) w.f32(deg_hidden(j, D) >= deg_hidden(a, D) ? q.W1 + (int64_t)a * H + j : -1);
    for (int a = 0; a < H; ++a)
        for (int dd = 0; dd < D; ++dd)
            for (int jb = 0; jb < nbp; ++jb)
                w.f32((jb < nl.n_out && deg_out(dd) >= deg_hidden(a, D)) ? q.W2 + (int64_t)a * q.NO + (jb * D + dd) : -1);
    // zero_params[d][j] of a gated head (the leaf follows b2; model_factory.py:84), |z| under a sigmoid head (:62-63); zeros otherwise
    const bool gated = nl.has_zero && net_is_gated(m, n), sig = net_has_sigmoid_head(m, n);
    for (int dd = 0; dd < D; ++dd)
        for (int jb = 0; jb < nbp; ++jb) {
            const int64_t src = (gated && jb < nl.n_out) ? q.b2 + q.NO + (int64_t)dd * nl.n_out + jb : -1;
            if (sig) w.f32_abs(src);
            else w.f32(src);
        }
    for (int dd = 0; dd < D; ++dd)   // the same leaf without the |.| (zero_raw)
        for (int jb = 0; jb < nbp; ++jb) w.f32((gated && jb < nl.n_out) ? q.b2 + q.NO + (int64_t)dd * nl.n_out + jb : -1);
}


// Wave-kernel image of net n (NetWave): W0 [D][64], b0, b1, b2 [P][64], W1f, W1b [16][64][4], W2f, W2b [P][16][64][4]
static int64_t wave_net_floats(int D, int nbp) {
    const int P = wave_passes(D, nbp);
    return (int64_t)D * kHidden + 2 * kHidden + (int64_t)P * 64 + 2 * 4096 + (int64_t)P * 2 * 4096 + (int64_t)P * 64;   // ..., z
}

static void describe_wave_image(const wf_model* m, int n, uint32_t base, std::vector<PackRec>& out) {
    const int D = m->desc.n_dim, H = kHidden, P = wave_passes(D, m->nbp);
    const bool wide = m->nbp == 64;
    const NetLayout& nl = m->nets[n];
    const NetOffsets q = net_offsets(m, n);
    auto w1m = [&](int a, int j) -> int64_t { return deg_hidden(j, D) >= deg_hidden(a, D) ? q.W1 + (int64_t)a * H + j : -1; };
    // column of output lane c of pass p: (d, jb) = (2p + (c >> 5), c & 31), or (p, c) in the 64-row layout
    auto w2m = [&](int a, int p, int c) -> int64_t {
        const int d = wide ? p : 2 * p + (c >> 5), jb = wide ? c : (c & 31);
        if (d >= D || jb >= nl.n_out || deg_out(d) < deg_hidden(a, D)) return -1;
        return q.W2 + (int64_t)a * q.NO + (jb * D + d);
    };
    ImageWriter w{out, base};
    for (int a = 0; a < D; ++a)
        for (int j = 0; j < H; ++j) w.f32(deg_hidden(j, D) >= deg_in(a) ? q.W0 + (int64_t)a * H + j : -1);
    for (int j = 0; j < H; ++j) w.f32(q.b0 + j);
    for (int j = 0; j < H; ++j) w.f32(q.b1 + j);
    for (int p = 0; p < P; ++p)
        for (int c = 0; c < 64; ++c) {
            const int d = wide ? p : 2 * p + (c >> 5), jb = wide ? c : (c & 31);
            w.f32((d < D && jb < nl.n_out) ? q.b2 + jb * D + d : -1);
        }
    for (int g = 0; g < 16; ++g)
        for (int j = 0; j < 64; ++j)
            for (int e = 0; e < 4; ++e) w.f32(w1m(4 * g + e, j));
    for (int g = 0; g < 16; ++g)
        for (int a = 0; a < 64; ++a)
            for (int e = 0; e < 4; ++e) w.f32(w1m(a, 4 * g + e));
    for (int p = 0; p < P; ++p)
        for (int g = 0; g < 16; ++g)
            for (int c = 0; c < 64; ++c)
                for (int e = 0; e < 4; ++e) w.f32(w2m(4 * g + e, p, c));
    for (int p = 0; p < P; ++p)
        for (int g = 0; g < 16; ++g)
            for (int a = 0; a < 64; ++a)
                for (int e = 0; e < 4; ++e) w.f32(w2m(a, p, 4 * g + e));
    // zero_params of a gated head in the lane order of b2 (|z| under a sigmoid head); zeros otherwise
    const bool gated = nl.has_zero && net_is_gated(m, n), sig = net_has_sigmoid_head(m, n);
    for (int p = 0; p < P; ++p)
        for (int c = 0; c < 64; ++c) {
            const int d = wide ? p : 2 * p + (c >> 5), jb = wide ? c : (c & 31);
            const int64_t src = (gated && d < D && jb < nl.n_out) ? q.b2 + q.NO + (int64_t)d * nl.n_out + jb : -1;
            if (sig) w.f32_abs(src);
            else w.f32(src);
        }
}

// ---------------------------------------------------------------------------- MFMA kernel images
static inline int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

static int mfma_net_floats(int D, int nbk) {
    const int S0 = (D + 1) / 2;
    return 128 * S0 + 64 + 4096 + 64 + (D - 1) * nbk * 2048 + 32 * D * nbk + 32 * D * nbk + 64;   // ..., biases, zero_params, unfolded b1 (NetOff::b1c)
}

// per-row factor: remove_bias scaling (isplines_jax.py:196-202 / msplines_jax.py:186-192) times the
// 0/1 "kept by the boundary conditions" mask; 0 beyond the real bases.  Layout [half][16] in accumulator order.
static void row_factors(int kind, bool with_remove_bias, int k, int nb, int nbk, const std::vector<double>& bc_colsum, float* out_acc,
                        float* natural64 = nullptr) {
    std::vector<float> f(64, 0.0f);
    for (int j = 0; j < nb; ++j) f[j] = 1.0f;
    if (with_remove_bias)
        for (int i = 0; i < k; ++i) {
            const int a = kind == WF_SPLINE_I ? i + 1 : i;
            const int b = kind == WF_SPLINE_I ? nb - (i + 2) : nb - (i + 1);
            const float fac = (float)(i + 1) / (float)k;
            f[a] *= fac;
            f[b] *= fac;
        }
    for (int j = 0; j < nb; ++j) f[j] = (float)((double)f[j] * bc_colsum[j]);   // a~ of bc_map: 0 / 1 for zero-only constraints
    for (int kb = 0; kb < nbk; ++kb)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) out_acc[(kb * 2 + h) * 16 + r] = f[32 * kb + acc_row(r, h)];
    if (natural64)
        for (int j = 0; j < 64; ++j) natural64[j] = f[j];
}

// Spline table of the MFMA kernel: [n_mesh][8 * nbk pieces][n_orders][2 sides][4 rows].  Piece p = natural rows 4p .. 4p+3 (a lane of
// walker half h holds the pieces 8 kb + 2q + h, q = 0..3, of every 32-row block in its accumulator registers); side 0 = mesh point m,
// side 1 = mesh point min(m + 1, n_mesh - 1): the two ends of the reference's lerp (isplines_jax.py:45-56) and both derivative orders of
// a piece are one 64-byte record (32 bytes for the one-order prior table), so a lane's four records are all it reads for a spline
// evaluation.  Rows scaled by fk (acc layout [kb][h][16], may be null); rowsum (may be null): [n_mesh][n_orders].
static void pack_rows_pairs(const std::vector<double>& t64, int nb, int n_mesh, int n_orders, int nbk, const float* fk_acc,
                            std::vector<float>& out, std::vector<float>* rowsum) {
    const int n_pieces = 8 * nbk;
    out.assign((size_t)n_mesh * n_pieces * n_orders * 8, 0.0f);
    if (rowsum) rowsum->assign((size_t)n_mesh * n_orders, 0.0f);
    auto entry = [&](int nd, int row, int m) {
        const float t = (float)t64[((size_t)nd * nb + row) * n_mesh + m];   // the reference's fp32 table entry
        if (!fk_acc) return t;
        const int kb = row >> 5, w = row & 31, h = (w >> 2) & 1, r = (w & 3) + 4 * (w >> 3);   // acc_row(r, h) == w
        return (float)((double)fk_acc[(kb * 2 + h) * 16 + r] * (double)t);
    };
    for (int m = 0; m < n_mesh; ++m)
        for (int nd = 0; nd < n_orders; ++nd) {
            double rs = 0.0;
            for (int row = 0; row < nb; ++row) {
                const int pc = row >> 2, e = row & 3;
                for (int sd = 0; sd < 2; ++sd)
                    out[((((size_t)m * n_pieces + pc) * n_orders + nd) * 2 + sd) * 4 + e] = entry(nd, row, std::min(m + sd, n_mesh - 1));
                rs += (double)entry(nd, row, m);
            }
            if (rowsum) (*rowsum)[(size_t)m * n_orders + nd] = (float)rs;
        }
}

// Support bounds of the records of a pack_rows_pairs table: for piece 8 kb + 2q + h, bnd[(kb*2+h)*16+q*2+0] = the last mesh index up
// to which the piece's record equals the one at mesh point 0, bnd[(kb*2+h)*16+q*2+1] = the first one from which it equals the one
// at the last mesh point.  Spline bases have local support (I-splines: 0 below it, their full value above), so a read at
// clamp(m, lo, hi) returns the bits of the read at m, and the walkers outside a piece's support share two records instead of
// touching their own: found by comparing the table's actual fp32 entries, whatever the boundary map or the row factors made of them.
static void piece_bounds(const std::vector<float>& rows, int n_mesh, int n_orders, int nbk, int32_t* bnd) {
    const int n_pieces = 8 * nbk, rec = n_orders * 8;
    for (int kb = 0; kb < nbk; ++kb)
        for (int h = 0; h < 2; ++h)
            for (int q = 0; q < 4; ++q) {
                const int pc = 8 * kb + 2 * q + h;
                auto same = [&](int m, int ref) {
                    return memcmp(&rows[((size_t)m * n_pieces + pc) * rec], &rows[((size_t)ref * n_pieces + pc) * rec], rec * sizeof(float)) == 0;
                };
                int lo = 0, hi = n_mesh - 1;
                while (lo + 1 < n_mesh && same(lo + 1, 0)) ++lo;
                while (hi - 1 >= 0 && same(hi - 1, n_mesh - 1)) --hi;
                bnd[(kb * 2 + h) * 16 + q * 2 + 0] = lo;
                bnd[(kb * 2 + h) * 16 + q * 2 + 1] = hi;
            }
}

static bool net_has_sigmoid_head(const wf_model* m, int n) {
    const bool is_prior = n == m->desc.n_flow_layers;
    if (is_prior) return m->desc.prior_kind == WF_PRIOR_MFLOW;
    return m->desc.layer_kind == WF_LAYER_IMADE;
}

// LDS image of net n in MFMA operand order (wf_kernels_mfma.hip: NetOff<D>), float offset `base` inside d_mfma.  fp16 operand
// pairs: x = hi + lo with hi, lo in fp16 (round to nearest; lo may be subnormal: absolute precision 2^-25), k_pack splits them.
static void describe_mfma_image(const wf_model* m, int n, uint32_t base, std::vector<PackRec>& out) {
    const int D = m->desc.n_dim, H = kHidden, nbk = m->mdev.nbk;
    const int S0 = (D + 1) / 2;
    const NetLayout& nl = m->nets[n];
    const NetOffsets q = net_offsets(m, n);
    // folded activation scales: tanh(x) = 1 - 2/(2^(c1 x) + 1), sigmoid(x) = 1/(1 + 2^(c2 x)).  The kernel feeds the layers behind a
    // tanh with r = 1/(2^(c1 x) + 1) instead of tanh = 1 - 2r: their weights carry the factor -2 here, their biases get the column
    // sums of the weights from k_fold_bias (wf_kernels_mfma.hip) after every k_pack.
    const double c1 = 2.0 * 1.4426950408889634074;
    const bool sig = net_has_sigmoid_head(m, n);
    const double c2 = sig ? -1.4426950408889634074 : 1.0;
    ImageWriter w{out, base};
    auto f16_block = [&](uint32_t n_pairs, auto&& src_of) {   // hi halves then lo halves; returns nothing, advances w.o
        uint32_t hi = 2 * w.o, lo = hi + n_pairs;
        for (uint32_t e = 0; e < n_pairs; ++e) {
            const std::pair<int64_t, double> sv = src_of(e);
            out.push_back(PackRec{(int32_t)sv.first, 1, hi++, lo++, sv.first >= 0 ? sv.second : 0.0});
        }
        w.o += n_pairs;
    };
    // layer 0 (f32 MFMA): A[i = unit 32*ob + (lane&31)][k = 2s + (lane>>5)]
    for (int ob = 0; ob < 2; ++ob)
        for (int s = 0; s < S0; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int unit = 32 * ob + (lane & 31), k = 2 * s + (lane >> 5);
                w.f32((k < D && deg_hidden(unit, D) >= deg_in(k)) ? q.W0 + (int64_t)k * H + unit : -1, c1);
            }
    for (int ob = 0; ob < 2; ++ob)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) w.f32(q.b0 + 32 * ob + acc_row(r, h), c1);
    // layer 1 (f16 MFMA, K = 16 per step): step (t, s), element j of lane half h contracts hidden unit
    // kk = 32t + acc_row(8s + j, h);  images [ob][t][s][lane][8] for hi then lo
    f16_block(4096, [&](uint32_t e) {
        const int j = e & 7, lane = (e >> 3) & 63, s_ = (e >> 9) & 1, t = (e >> 10) & 1, ob = (e >> 11) & 1;
        const int unit = 32 * ob + (lane & 31), kk = 32 * t + acc_row(8 * s_ + j, lane >> 5);
        return std::make_pair(deg_hidden(unit, D) >= deg_hidden(kk, D) ? q.W1 + (int64_t)kk * H + unit : (int64_t)-1, -2.0 * c1);
    });
    for (int ob = 0; ob < 2; ++ob)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) w.f32(q.b1 + 32 * ob + acc_row(r, h), c1);
    // output layer, dimensions 1..D-1, row blocks kb: A[i = basis 32*kb + (lane&31)][k = kk]
    f16_block((uint32_t)((D - 1) * nbk * 2048), [&](uint32_t e) {
        const int j = e & 7, lane = (e >> 3) & 63, s_ = (e >> 9) & 1, t = (e >> 10) & 1;
        const int blk = e >> 11, kb = blk % nbk, d = 1 + blk / nbk;
        const int jb = 32 * kb + (lane & 31), kk = 32 * t + acc_row(8 * s_ + j, lane >> 5);
        const bool live = jb < nl.n_out && deg_out(d) >= deg_hidden(kk, D);
        return std::make_pair(live ? q.W2 + (int64_t)kk * q.NO + (jb * D + d) : (int64_t)-1, -2.0 * c2);
    });
    // biases; padding rows of sigmoid heads get +1e30 so that sigmoid(-x) -> 0 exactly
    for (int d = 0; d < D; ++d)
        for (int kb = 0; kb < nbk; ++kb)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const int jb = 32 * kb + acc_row(r, h);
                    if (jb < nl.n_out) w.f32(q.b2 + jb * D + d, c2);
                    else w.cst(sig ? 1e30 : 0.0);
                }
    // zero_params of a gated head, accumulator layout like the biases (|z| under a sigmoid head); zeros otherwise
    const bool gated = nl.has_zero && net_is_gated(m, n);
    for (int d = 0; d < D; ++d)
        for (int kb = 0; kb < nbk; ++kb)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) {
                    const int jb = 32 * kb + acc_row(r, h);
                    const int64_t src = (gated && jb < nl.n_out) ? q.b2 + q.NO + (int64_t)d * nl.n_out + jb : -1;
                    if (sig) w.f32_abs(src);
                    else w.f32(src);
                }
    // NetOff::b1c: the second hidden layer's bias as c1 * b1, which k_fold_bias leaves alone -- for the centred first-layer activations
    // (r - 1/2) of k_mfma's flow nets (wf_mfma_impl.h: hidden_layers<..., CENTER>)
    for (int ob = 0; ob < 2; ++ob)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) w.f32(q.b1 + 32 * ob + acc_row(r, h), c1);
}

// Decides whether the MFMA kernel covers this model and builds its parameter-independent parts.
// i64 / p64: the fp64 tables already built by model_build (I: [4][nb][n_mesh]; prior: OB or M), o2b: [nb][nb].
// Transposed operand images of net n for the reverse sweep of the matrix-core gradient path (k_ebwd, wf_kernels_etile.hip; D = 2, <= 64 bases):
// hbar_1[k] = sum_u W1'[k][u] zbar_2[u] and hbar_2[k] = sum_j W2'[k][j] obar[j] are MFMA products whose A operand is the weight matrix with the
// INPUT unit on the row, same entries and scales as the forward image.  Layout (floats, base = float offset inside d_mfma; nbk = 32-row blocks of the head):
//   TW1 hi [ob 2][t 2][s 2][lane 64][8 halves] (2048 floats), TW1 lo (2048), TW2 hi [ob 2][kb nbk][s 2][64][8] (1024 nbk), TW2 lo (1024 nbk), W0'[0][unit] in
//   accumulator layout [ob][h][16] (64): the adjoint of the conditioner's input s is sum_u W0'[0][u] zbar_1[u].
static int tnet_floats_of(int nbk) { return 2048 + 2048 + 2048 * nbk + 64; }
static void describe_mfma_image_t(const wf_model* m, int n, uint32_t base, std::vector<PackRec>& out) {
    const int D = m->desc.n_dim, H = kHidden, nbk = m->mdev.nbk;
    const NetLayout& nl = m->nets[n];
    const NetOffsets q = net_offsets(m, n);
    const double c1 = 2.0 * 1.4426950408889634074;
    const double c2 = net_has_sigmoid_head(m, n) ? -1.4426950408889634074 : 1.0;
    ImageWriter w{out, base};
    auto f16_block = [&](uint32_t n_pairs, auto&& src_of) {
        uint32_t hi = 2 * w.o, lo = hi + n_pairs;
        for (uint32_t e = 0; e < n_pairs; ++e) {
            const std::pair<int64_t, double> sv = src_of(e);
            out.push_back(PackRec{(int32_t)sv.first, 1, hi++, lo++, sv.first >= 0 ? sv.second : 0.0});
        }
        w.o += n_pairs;
    };
    // A[m = input unit k = 32 ob + (lane & 31)][kk = output unit u = 32 t + acc_row(8 s + j, lane >> 5)] = W1'[k][u]
    f16_block(4096, [&](uint32_t e) {
        const int j = e & 7, lane = (e >> 3) & 63, s_ = (e >> 9) & 1, t = (e >> 10) & 1, ob = (e >> 11) & 1;
        const int k = 32 * ob + (lane & 31), u = 32 * t + acc_row(8 * s_ + j, lane >> 5);
        return std::make_pair(deg_hidden(u, D) >= deg_hidden(k, D) ? q.W1 + (int64_t)k * H + u : (int64_t)-1, -2.0 * c1);
    });
    // A[m = hidden unit k][kk = basis row jb = 32 kb + acc_row(8 s + j, lane >> 5)] = W2'[k][(jb, d = 1)]
    f16_block((uint32_t)(2048 * nbk), [&](uint32_t e) {
        const int j = e & 7, lane = (e >> 3) & 63, s_ = (e >> 9) & 1, blk = e >> 10, kb = blk % nbk, ob = blk / nbk;
        const int k = 32 * ob + (lane & 31), jb = 32 * kb + acc_row(8 * s_ + j, lane >> 5);
        const bool live = jb < nl.n_out && deg_out(1) >= deg_hidden(k, D);
        return std::make_pair(live ? q.W2 + (int64_t)k * q.NO + (jb * D + 1) : (int64_t)-1, -2.0 * c2);
    });
    for (int ob = 0; ob < 2; ++ob)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) w.f32(q.W0 + 32 * ob + acc_row(r, h), c1);   // W0[k = 0][unit]
}

static int mfma_prepare(wf_model* m, const std::vector<double>& i64, const std::vector<double>& p64, const std::vector<double>& o2b) {
    const wf_model_desc& d = m->desc;
    const int D = d.n_dim;
    m->mfma_ok = false;
    const int nbk = m->nbp / 32;
    if (!mfma_shape_built(D, nbk)) return WF_OK;
    const bool imade = d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0;
    if (imade && !m->bc_i_ok) return WF_OK;
    const bool spline_prior = d.prior_kind == WF_PRIOR_WAVEFLOW || d.prior_kind == WF_PRIOR_MFLOW;
    if (spline_prior && !m->bc_p_ok) return WF_OK;
    const int n_nets = (int)m->nets.size();
    const int consts = 64 * nbk + nbk * nbk * 1024 + 64 * nbk + 32 * nbk;   // fkI, fkP, ob_to_b image, piece bounds (flow table, prior table), cbP (constant term of the B prior's boundary map)
    const int net_floats = mfma_net_floats(D, nbk);
    const int64_t lds_cap = 160 * 1024 / 4 - 64;   // floats (the kernel also holds a few bytes of static LDS: its tile counter)
    int staged;
    if ((int64_t)consts + (int64_t)net_floats * n_nets <= lds_cap) staged = 0;        // every net resident
    else if ((int64_t)consts + net_floats + 16 * kStagedGroups * (D + 1) * 32 <= lds_cap) staged = 1;   // one slot + the state area, re-staged per super-chunk
    else return WF_OK;
    // the matrix-core gradient path (two particles, <= 64 bases, Waveflow prior, IMADE layers): transposed operand images behind the constants block
    const bool timg = D == 2 && (nbk == 1 || nbk == 2) && d.prior_kind == WF_PRIOR_WAVEFLOW && (d.layer_kind == WF_LAYER_IMADE || d.n_flow_layers == 0);
    const int tconsts = nbk * nbk * 1024;   // ob_to_b transposed: blocks [ka][ki]{hi [s 2][lane 64][8 halves] (512 floats), lo (512)}
    const int tnet_floats = tnet_floats_of(nbk);
    const int64_t total = (int64_t)net_floats * n_nets + consts + (timg ? (int64_t)tnet_floats * n_nets + tconsts : 0);

    MfmaDev& md = m->mdev;
    md = MfmaDev{};
    md.D = D; md.n_layers = d.n_flow_layers; md.layer_kind = d.layer_kind; md.box_kind = d.box_kind; md.prior_kind = d.prior_kind;
    md.box_L = d.box_size; md.i_reg = d.i_reg; md.normal_offset = d.normal_offset; md.constrained_mask = m->dev.constrained_mask;
    md.i_nb = m->i_nb; md.p_nb = m->p_nb; md.n_mesh = d.n_mesh; md.nbk = nbk;
    md.n_nets = n_nets; md.net_floats = net_floats; md.const_img_off = net_floats * n_nets; md.const_floats = consts; md.staged = staged;
    md.exact_div = mfma_div_ok(md.n_mesh) ? 0 : 1;
    md.prior_quotient = (getenv("WF_PRIOR_QUOTIENT") && atoi(getenv("WF_PRIOR_QUOTIENT")) != 0) ? 1 : 0;
    md.i_gate = m->dev.i_gate; md.p_gate = m->dev.p_gate;
    md.p_bias = (d.prior_kind == WF_PRIOR_WAVEFLOW && !m->p_cb.empty()) ? 1 : 0;
    md.p_plain_bc = (d.prior_kind == WF_PRIOR_WAVEFLOW && m->bc_p_plain) ? 1 : 0;
    md.tabB0 = m->d_tabB0;
    md.i_plain_bc = (imade && m->bc_i_plain) ? 1 : 0;
    md.timg_off = timg ? net_floats * n_nets + consts : -1;
    md.tnet_floats = tnet_floats;
    md.tconst_off = timg ? md.timg_off + tnet_floats * n_nets : -1;
    // staged mode: one net slot + the state area of the super-chunk (16 waves x kStagedGroups tile groups x (D + 1) x 32 floats: the
    // built staged shapes run 8 waves of one tile; sized for the largest workgroup)
    m->mfma_lds_floats = consts + (staged ? net_floats + 16 * kStagedGroups * (D + 1) * 32 : net_floats * n_nets);

    m->mfma_consts.assign(consts, 0.0f);
    int32_t* bnd = reinterpret_cast<int32_t*>(m->mfma_consts.data() + 64 * nbk + nbk * nbk * 1024);   // [2 tables][nbk][2 halves][16: 4 pieces x (lo, hi), 8 unused -- the lane stride of the fk blocks]
    for (int i = 0; i < 64 * nbk; ++i) bnd[i] = (i & 1) ? d.n_mesh - 1 : 0;   // (no clamp until a table says otherwise)
    if (md.p_bias) {   // cbP[kb][h][r] = p_cb[row of register r in lane half h of block kb]
        float* cb = m->mfma_consts.data() + 64 * nbk + nbk * nbk * 1024 + 64 * nbk;
        for (int kb = 0; kb < nbk; ++kb)
            for (int hh = 0; hh < 2; ++hh)
                for (int r = 0; r < 16; ++r) cb[(kb * 2 + hh) * 16 + r] = m->p_cb[32 * kb + acc_row(r, hh)];
    }
    std::vector<float> fk_nat(128, 0.0f);
    if (imade) {
        float* fk = m->mfma_consts.data();
        row_factors(WF_SPLINE_I, true, d.i_degree, m->i_nb, nbk, m->bc_i_colsum, fk, fk_nat.data());
        double F = 0;
        for (int i = 0; i < 32 * nbk; ++i) F += fk[i];
        md.F_I = (float)F;
        std::vector<float> rows, rowsum;
        pack_rows_pairs(i64, m->i_nb, d.n_mesh, 2, nbk, fk, rows, &rowsum);
        if (!getenv("WF_MFMA_NO_BAND")) piece_bounds(rows, d.n_mesh, 2, nbk, bnd);   // (the switch: tests compare both, bit for bit)
        int rc = upload_table(m, rows, &md.tabI);
        if (rc) return rc;
        {   // [mesh][2] -> [mesh]{R0_m, R1_m, R0_{m+1}, R1_{m+1}} (the last mesh point repeats itself: it is a right end only)
            std::vector<float> pairs((size_t)d.n_mesh * 4);
            for (int mi = 0; mi < d.n_mesh; ++mi) {
                const int mr = std::min(mi + 1, d.n_mesh - 1);
                pairs[(size_t)mi * 4 + 0] = rowsum[(size_t)mi * 2]; pairs[(size_t)mi * 4 + 1] = rowsum[(size_t)mi * 2 + 1];
                pairs[(size_t)mi * 4 + 2] = rowsum[(size_t)mr * 2]; pairs[(size_t)mi * 4 + 3] = rowsum[(size_t)mr * 2 + 1];
            }
            rc = upload_table(m, pairs, &md.rsI);
            if (rc) return rc;
        }
    }
    if (spline_prior) {
        const bool mflow = d.prior_kind == WF_PRIOR_MFLOW;
        float* fk = m->mfma_consts.data() + 32 * nbk;
        row_factors(mflow ? WF_SPLINE_M : WF_SPLINE_B, mflow, d.p_degree, m->p_nb, nbk, m->bc_p_colsum, fk, fk_nat.data() + 64);
        double F = 0;
        for (int i = 0; i < 32 * nbk; ++i) F += fk[i];
        md.F_P = (float)F;
        std::vector<float> rows;
        // M prior: the row factors are folded into the table; B prior: they act on the weights before ob_to_b
        pack_rows_pairs(p64, m->p_nb, d.n_mesh, 1, nbk, mflow ? fk : nullptr, rows, nullptr);
        if (!getenv("WF_MFMA_NO_BAND")) piece_bounds(rows, d.n_mesh, 1, nbk, bnd + 32 * nbk);
        int rc = upload_table(m, rows, &md.tabP);
        if (rc) return rc;
        if (!mflow) {
            // c[i] = sum_a w[a] * ob_to_b[a][i] as split-fp16 MFMA products: block (ko, ki), K step s:
            // A[i = 32*ko + (lane&31)][k = a = 32*ki + acc_row(8*s + j, lane>>5)], hi halves [s][lane][8] then lo halves (1024 each)
            _Float16* o = reinterpret_cast<_Float16*>(m->mfma_consts.data() + 64 * nbk);
            const int nb = m->p_nb;
            for (int ko = 0; ko < nbk; ++ko)
                for (int ki = 0; ki < nbk; ++ki) {
                    _Float16* blk = o + (size_t)(ko * nbk + ki) * 2048;
                    for (int s_ = 0; s_ < 2; ++s_)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int i = 32 * ko + (lane & 31), a = 32 * ki + acc_row(8 * s_ + j, lane >> 5);
                                const float v = (i < nb && a < nb) ? (float)o2b[(size_t)a * nb + i] : 0.0f;
                                const _Float16 hi = (_Float16)v;
                                blk[(s_ * 64 + lane) * 8 + j] = hi;
                                blk[1024 + (s_ * 64 + lane) * 8 + j] = (_Float16)(v - (float)hi);
                            }
                }
        }
    }
    int rc = dev_alloc(m, &m->d_mfma, (size_t)total);
    if (rc) return rc;
    md.image = m->d_mfma;
    rc = dev_alloc(m, &m->d_fk_nat, 128);
    if (rc) return rc;
    WF_HIP(hipMemcpy(m->d_fk_nat, fk_nat.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
    {
        float* comp = nullptr;
        // [comp][coefficients of k_dim0_coeffs][comp2]
        const size_t comp_floats = (size_t)std::max(n_nets, 1) * d.n_mesh * 4, coef_floats = (size_t)dim0_coef_floats(std::max(n_nets, 1));
        rc = dev_alloc(m, &comp, comp_floats + ((coef_floats + 3) & ~(size_t)3) + comp_floats);
        if (rc) return rc;
        m->d_comp = comp;
        md.comp = reinterpret_cast<const float4_t*>(comp);
        md.comp2 = reinterpret_cast<const float4_t*>(comp + comp_floats + ((coef_floats + 3) & ~(size_t)3));
    }
    {   // fp16-range flags of the operand images (one per net), their pinned host copy and the event that says it has arrived
        rc = dev_alloc(m, &m->d_f16_ovf, (size_t)kMaxNets);
        if (rc) return rc;
        WF_HIP(hipMemset(m->d_f16_ovf, 0, kMaxNets * sizeof(int)));
        WF_HIP(hipHostMalloc(reinterpret_cast<void**>(&m->h_f16_ovf), kMaxNets * sizeof(int), hipHostMallocDefault));
        memset(m->h_f16_ovf, 0, kMaxNets * sizeof(int));
        WF_HIP(hipEventCreateWithFlags(&m->ovf_event, hipEventDisableTiming));
        md.f16_ovf = m->d_f16_ovf;
    }
    m->mfma_floats = total;
    m->mfma_ok = true;
    // the constants block does not depend on the parameters: upload it once
    WF_HIP(hipMemset(m->d_mfma, 0, (size_t)total * sizeof(float)));
    WF_HIP(hipMemcpy(m->d_mfma + md.const_img_off, m->mfma_consts.data(), m->mfma_consts.size() * sizeof(float), hipMemcpyHostToDevice));
    if (timg && spline_prior) {
        // wbar[a] = sum_i M[a][i] cbar[i] (M = ob_to_b with the boundary map folded in, as the forward image holds it): block (ka, ki) in the order
        // prior_c_block reads (output block first): A[m = a = 32 ka + (lane & 31)][kk = i = 32 ki + acc_row(8 s + j, lane >> 5)]
        std::vector<float> tc(tconsts, 0.0f);
        _Float16* o = reinterpret_cast<_Float16*>(tc.data());
        const int nb = m->p_nb;
        for (int ka = 0; ka < nbk; ++ka)
            for (int ki = 0; ki < nbk; ++ki) {
                _Float16* blk = o + (size_t)(ka * nbk + ki) * 2048;
                for (int s_ = 0; s_ < 2; ++s_)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int a = 32 * ka + (lane & 31), i = 32 * ki + acc_row(8 * s_ + j, lane >> 5);
                            const float v = (i < nb && a < nb) ? (float)o2b[(size_t)a * nb + i] : 0.0f;
                            const _Float16 hi = (_Float16)v;
                            blk[(s_ * 64 + lane) * 8 + j] = hi;
                            blk[1024 + (s_ * 64 + lane) * 8 + j] = (_Float16)(v - (float)hi);
                        }
            }
        WF_HIP(hipMemcpy(m->d_mfma + md.tconst_off, tc.data(), tc.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return WF_OK;
}

// The wave-cooperative kernels (wf_kernels_wave.hip): <= 32 bases, constraints that only zero the end weights.
static bool wave_capable(const wf_model* m) {
    const wf_model_desc& d = m->desc;
    if (!m->d_wave || (m->nbp == 64 && d.n_dim > 4)) return false;   // (the 64-row sweeps are built for D <= 4)
    const bool imade = d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0;
    if (imade && (!m->d_tabI4 || !m->bc_i_ok)) return false;
    const bool spline_prior = d.prior_kind == WF_PRIOR_WAVEFLOW || d.prior_kind == WF_PRIOR_MFLOW;
    if (spline_prior && (!m->d_tabP3 || !m->bc_p_ok)) return false;
    return true;
}
// ... which is also what the reverse pass and the local energy need (every D the library supports, 2..8, is instantiated)
static bool grad_capable(const wf_model* m) { return m->wave_ok && !m->nets.empty(); }   // (gated heads included: run_vjp_chunks)

// Describes every weight image (PackRec lists on the device) and derives the gradient scatter map: forward-image entry ->
// flat parameter (masked and padding entries have no source: no gradient).
static int pack_prepare(wf_model* m, std::vector<PackRec>& plain) {
    const int D = m->desc.n_dim;
    const int n_nets = (int)m->nets.size();
    std::vector<PackRec> wave, mfma;
    for (int n = 0; n < n_nets; ++n) {
        describe_plain_image(m, n, (uint32_t)m->plain_off[n], plain);
        if (m->d_wave) describe_wave_image(m, n, (uint32_t)(wave_net_floats(D, m->nbp) * n), wave);
        if (m->mfma_ok) describe_mfma_image(m, n, (uint32_t)((int64_t)m->mdev.net_floats * n), mfma);
        if (m->mfma_ok && m->mdev.timg_off >= 0) describe_mfma_image_t(m, n, (uint32_t)(m->mdev.timg_off + (int64_t)m->mdev.tnet_floats * n), mfma);
    }
    std::vector<PackRec> all;
    all.reserve(plain.size() + wave.size() + mfma.size());
    std::vector<PackRec>* lists[3] = {&plain, &wave, &mfma};
    for (int i = 0; i < 3; ++i)
        for (PackRec r : *lists[i]) {
            r.kind |= i << 8;
            all.push_back(r);
        }
    m->n_pack = (int64_t)all.size();
    if (!all.empty()) {
        int rc = dev_alloc(m, &m->d_pack, all.size());
        if (rc) return rc;
        WF_HIP(hipMemcpy(m->d_pack, all.data(), all.size() * sizeof(PackRec), hipMemcpyHostToDevice));
    }
    return dev_alloc(m, &m->d_flat, (size_t)std::max<int64_t>(m->n_params, 1));
}

static int grad_prepare(wf_model* m) {
    const wf_model_desc& d = m->desc;
    const int D = d.n_dim;
    std::vector<PackRec> plain;
    {
        int rc = pack_prepare(m, plain);
        if (rc) return rc;
    }
    m->wave_ok = wave_capable(m);
    if (m->wave_ok) {
        std::vector<float> fk(128, 0.0f), acc(64);
        if (d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0)
            row_factors(WF_SPLINE_I, true, d.i_degree, m->i_nb, m->nbp / 32, m->bc_i_colsum, acc.data(), fk.data());
        if (d.prior_kind == WF_PRIOR_WAVEFLOW) row_factors(WF_SPLINE_B, false, d.p_degree, m->p_nb, m->nbp / 32, m->bc_p_colsum, acc.data(), fk.data() + 64);
        if (d.prior_kind == WF_PRIOR_MFLOW) row_factors(WF_SPLINE_M, true, d.p_degree, m->p_nb, m->nbp / 32, m->bc_p_colsum, acc.data(), fk.data() + 64);
        int rc = dev_alloc(m, &m->d_grad_fk, fk.size());
        if (rc) return rc;
        WF_HIP(hipMemcpy(m->d_grad_fk, fk.data(), fk.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (m->wave_ok) {
        // scratch for the small-batch wave path (tails of up to kWaveEvalMax walkers, first or second order) is reserved here
        // so that those calls never allocate: they can be captured in a hipGraph
        int rc = ensure_scratch(m, kWaveEvalMax * std::max(wave_tail_floats(D, 0), wave_tail_floats(D, 1)));
        if (rc) return rc;
    }
    if (!grad_capable(m) || m->n_params >= (1 << 24)) return WF_OK;
    m->grad_psi_ok = d.prior_kind == WF_PRIOR_WAVEFLOW && (d.layer_kind == WF_LAYER_IMADE || d.n_flow_layers == 0);
    // One taped sample per walker in RF (value, gradient, Laplacian / 2: D + 2 channels), or D samples in R3 (3 channels each).
    // Fixed per model, because workspace sizes depend on it; WF_GRAD_R3 (read here) selects R3 for A/B tests.
    m->ring2 = (getenv("WF_GRAD_R3") || D > kTapedLaplacianMaxD) ? 1 : 2;
    if (m->grad_psi_ok && m->mfma_ok && energy_vjp_capable(&m->mdev)) {
        int rc = dev_alloc(m, &m->d_egacc, (size_t)energy_vjp_gacc_floats((int)m->nets.size(), m->mdev.nbk));
        if (rc) return rc;
    }
    const int n_nets = (int)m->nets.size();
    const int64_t fwd = plain_fwd_floats(D, m->nbp);
    // the plain description lists net n's forward-orientation entries first (plain_net_floats per net)
    std::vector<int32_t> map((size_t)(fwd * n_nets));
    const int64_t per_net = plain_net_floats(D, m->nbp);
    for (int n = 0; n < n_nets; ++n)
        for (int64_t i = 0; i < fwd; ++i) map[(size_t)(fwd * n + i)] = plain[(size_t)(per_net * n + i)].src;
    // inverse: parameter -> its (unique) forward-image entry, -1 for parameters that reach none
    std::vector<int32_t> inv((size_t)std::max<int64_t>(m->n_params, 1), -1);
    for (size_t i = 0; i < map.size(); ++i)
        if (map[i] >= 0) inv[(size_t)map[i]] = (int32_t)i;
    int rc = dev_alloc(m, &m->d_grad_map, inv.size());
    if (rc) return rc;
    WF_HIP(hipMemcpy(m->d_grad_map, inv.data(), inv.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (m->dev.i_gate || m->dev.p_gate) {
        // zero_params gradient rows in the wave layout: row = (net * P + p) * 64 + lane  <->  leaf entry (d, jb) of that net
        const int P = wave_passes(D, m->nbp);
        const bool wide = m->nbp == 64;
        m->z_rows = n_nets * P * 64;
        std::vector<int32_t> zmap((size_t)m->z_rows, -1), zoff((size_t)m->z_rows, -1);
        for (int n = 0; n < n_nets; ++n) {
            const NetLayout& nl = m->nets[n];
            if (!nl.has_zero || !net_is_gated(m, n)) continue;
            const NetOffsets q = net_offsets(m, n);
            const bool sig = net_has_sigmoid_head(m, n);
            for (int p = 0; p < P; ++p)
                for (int c = 0; c < 64; ++c) {
                    const int dd = wide ? p : 2 * p + (c >> 5), jb = wide ? c : (c & 31);
                    if (dd >= D || jb >= nl.n_out) continue;
                    const size_t r = ((size_t)n * P + p) * 64 + c;
                    zmap[r] = (int32_t)(q.b2 + q.NO + (int64_t)dd * nl.n_out + jb);
                    if (sig) zoff[r] = (int32_t)(m->plain_off[n] + (m->dev.nets[n].zero_raw - m->dev.nets[n].W0) + (int64_t)dd * m->nbp + jb);
                }
        }
        rc = dev_alloc(m, &m->d_zmap, zmap.size());
        if (rc) return rc;
        WF_HIP(hipMemcpy(m->d_zmap, zmap.data(), zmap.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        rc = dev_alloc(m, &m->d_zraw_off, zoff.size());
        if (rc) return rc;
        WF_HIP(hipMemcpy(m->d_zraw_off, zoff.data(), zoff.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        rc = dev_alloc(m, &m->d_zpart, (size_t)64 * m->z_rows);
        if (rc) return rc;
        rc = dev_alloc(m, &m->d_zgrad, (size_t)m->z_rows);
        if (rc) return rc;
    }
    rc = dev_alloc(m, &m->d_grad_partial, (size_t)wgrad_partial_floats(n_nets, fwd));
    if (rc) return rc;
    return dev_alloc(m, &m->d_grad_img, map.size());
}

}  // namespace wf

// ------------------------------------------------------------------------------------------ C ABI
using namespace wf;

extern "C" {

int wf_abi_version(void) { return WF_ABI_VERSION; }

const char* wf_strerror(int status) {
    switch (status) {
        case WF_OK: return "ok";
        case WF_ERR_INVALID: return "invalid argument";
        case WF_ERR_UNSUPPORTED: return "configuration not supported by this build";
        case WF_ERR_HIP: return "HIP runtime error (see wf_last_hip_error_string)";
        case WF_ERR_NO_DEVICE: return "no gfx950 device available (there is no CPU fallback)";
        case WF_ERR_NOMEM: return "out of memory";
        case WF_ERR_NUMERIC: return "numerical failure while building tables";
        default: return "unknown status";
    }
}

int wf_last_hip_error(void) { return g_last_hip; }
const char* wf_last_hip_error_string(void) { return hipGetErrorString((hipError_t)g_last_hip); }

int wf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int good = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && std::string(p.gcnArchName).rfind("gfx950", 0) == 0) ++good;
    }
    return good;
}

int wf_tables_build(int kind, int degree, int n_internal_knots, int n_mesh, double* out, double* b_to_ob, double* ob_to_b) {
    if (kind < WF_SPLINE_M || kind > WF_SPLINE_OB) return WF_ERR_INVALID;
    if (degree < 1 || n_internal_knots < 2 || n_mesh < 2) return WF_ERR_INVALID;
    const int nb = n_bases_of(kind, degree, n_internal_knots);
    if (!out) return nb;
    if (kind != WF_SPLINE_OB) return build_raw_table(kind, degree, n_internal_knots, n_mesh, out);
    std::vector<double> b64;
    try {
        b64.resize((size_t)4 * nb * n_mesh);
    } catch (const std::bad_alloc&) {
        return WF_ERR_NOMEM;
    }
    int rc = build_raw_table(WF_SPLINE_B, degree, n_internal_knots, n_mesh, b64.data());
    if (rc < 0) return rc;
    return build_ortho_b(degree, n_internal_knots, n_mesh, b64.data(), out, b_to_ob, ob_to_b);
}

int wf_model_create(const wf_model_desc* desc, int device, wf_model** out) {
    if (!desc || !out) return WF_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return WF_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return WF_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return WF_ERR_NO_DEVICE;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return WF_ERR_NO_DEVICE;
    DeviceGuard g(device);
    if (!g.ok) return WF_ERR_NO_DEVICE;
    wf_model* m = new (std::nothrow) wf_model();
    if (!m) return WF_ERR_NOMEM;
    m->desc = *desc;
    m->device = device;
    int rc;
    try {
        rc = model_build(m);
    } catch (const std::bad_alloc&) {
        rc = WF_ERR_NOMEM;
    }
    if (rc != WF_OK) {
        wf_model_destroy(m);
        return rc;
    }
    *out = m;
    return WF_OK;
}

void wf_model_destroy(wf_model* m) {
    if (!m) return;
    DeviceGuard g(m->device);
    for (void* p : m->allocs) (void)hipFree(p);
    if (m->ovf_event) (void)hipEventDestroy(m->ovf_event);
    if (m->h_f16_ovf) (void)hipHostFree(m->h_f16_ovf);
    delete m;
}

int64_t wf_model_param_count(const wf_model* m) { return m ? m->n_params : WF_ERR_INVALID; }

int wf_model_n_bases(const wf_model* m, int which) {
    if (!m) return WF_ERR_INVALID;
    return which == 0 ? m->i_nb : m->p_nb;
}

int wf_model_set_kernel(wf_model* m, int kernel_kind) {
    if (!m || kernel_kind < WF_KERNEL_AUTO || kernel_kind > WF_KERNEL_WAVE) return WF_ERR_INVALID;
    if (kernel_kind == WF_KERNEL_MFMA && !m->mfma_ok) return WF_ERR_UNSUPPORTED;
    if (kernel_kind == WF_KERNEL_WAVE && !m->wave_ok) return WF_ERR_UNSUPPORTED;
    m->kernel_kind = kernel_kind;
    return WF_OK;
}

// fills every weight image from a device-resident flat vector (asynchronous on `stream`)
static int apply_params(wf_model* m, const float* flat_dev, void* stream, bool eval_tables = true) {
    if (m->is_nsc) {   // the kernel reads the Dense leaves as they are: keep the model's own copy
        if (m->n_params > 0 && flat_dev != m->d_nsc)
            WF_HIP(hipMemcpyAsync(m->d_nsc, flat_dev, (size_t)m->n_params * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        m->params_set = true;
        return WF_OK;
    }
    {
        int rc = launch_pack(flat_dev, m->d_pack, m->n_pack, m->d_plain, m->d_wave, m->d_mfma, stream);
        if (rc) return rc;
    }
    if (m->mfma_ok && eval_tables) {
        // biases of the layers behind a tanh: + column sums of their weights (the kernel's activations are r, tanh = 1 - 2r)
        int rc0 = launch_fold_bias(m->d_mfma, (int)m->nets.size(), m->mdev.net_floats, m->desc.n_dim, m->mdev.nbk, m->d_f16_ovf, stream);
        if (rc0) return rc0;
        // the flags follow the images to the host (not while the stream is being captured: a replayed step keeps the answer of its capture
        // and the kernels' own NaN poisoning is what shows)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
            WF_HIP(hipMemcpyAsync(m->h_f16_ovf, m->d_f16_ovf, m->nets.size() * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
            WF_HIP(hipEventRecord(m->ovf_event, (hipStream_t)stream));
            m->ovf_pending = true;
        }
        // composite tables of output dimension 0 (reads the plain image filled above)
        int rc = launch_prepare_dim0(m->d_dev, (int)m->nets.size(), m->desc.n_mesh, m->d_fk_nat, m->mdev.F_I, m->mdev.F_P, m->d_tabI4, m->d_tabP3, m->d_comp, stream);
        if (rc) return rc;
    }
    if (m->mfma_ok) m->eval_tables_stale = !eval_tables;
    m->params_set = true;
    return WF_OK;
}

int wf_model_set_params(wf_model* m, const float* flat_host, int64_t n, void* stream) {
    if (!m || !flat_host) return WF_ERR_INVALID;
    if (n != m->n_params) return WF_ERR_INVALID;
    DeviceGuard g(m->device);
    hipStream_t s = (hipStream_t)stream;
    if (n > 0) WF_HIP(hipMemcpyAsync(m->d_flat, flat_host, (size_t)n * sizeof(float), hipMemcpyHostToDevice, s));
    int rc = apply_params(m, m->d_flat, stream);
    if (rc) return rc;
    WF_HIP(hipStreamSynchronize(s));   // the caller may reuse flat_host
    return WF_OK;
}

int wf_model_set_params_device(wf_model* m, const float* flat_dev, int64_t n, void* stream) {
    if (!m || !flat_dev) return WF_ERR_INVALID;
    if (n != m->n_params) return WF_ERR_INVALID;
    DeviceGuard g(m->device);
    return apply_params(m, flat_dev, stream);   // the images are packed straight from the caller's vector (m->d_flat only stages host uploads)
}

int wf_adam_step(float* params_dev, const float* grad_dev, float* m_dev, float* v_dev, int64_t n, int64_t step, float step_size, float b1,
                 float b2, float eps, void* stream) {
    if (n < 0 || step < 0 || (n > 0 && (!params_dev || !grad_dev || !m_dev || !v_dev))) return WF_ERR_INVALID;
    if (n == 0) return WF_OK;
    return launch_adam(params_dev, grad_dev, m_dev, v_dev, n, step, step_size, b1, b2, eps, nullptr, stream);
}

// Is a packed weight of the current parameters outside the fp16 range?  Then every path that feeds fp16 operand images to the matrix cores
// (k_mfma, the tile kernels of wf_kernels_etile.hip) is off: `auto` takes the fp32 scalar / wave kernels, an explicit request for the MFMA
// kernel returns WF_ERR_UNSUPPORTED.  Waits for the upload's flag copy if it is still in flight (a host wait of the pack kernels, ~50 us,
// only in the call that follows an asynchronous upload).
static bool f16_overflow(const wf_model* cm) {
    wf_model* m = const_cast<wf_model*>(cm);
    if (!m->mfma_ok) return false;
    if (m->ovf_pending) {
        if (hipEventSynchronize(m->ovf_event) != hipSuccess) return true;
        m->ovf_pending = false;
        bool any = false;
        for (size_t n = 0; n < m->nets.size(); ++n) any |= m->h_f16_ovf[n] != 0;
        m->f16_overflow = any;
    }
    return m->f16_overflow;
}

static int check_fwd(const wf_model* m, const void* x, int64_t B, const void* out) {
    if (!m || B < 0) return WF_ERR_INVALID;
    if (B > 0 && (!x || !out)) return WF_ERR_INVALID;
    if (!m->params_set && m->n_params > 0) return WF_ERR_INVALID;
    return WF_OK;
}


// presort: the rows of x arrive in any order -- evaluate on the ascending sort of each row, psi (mode 1) times (-1)^inversions (helpers.py:55-58).
// The MFMA kernel sorts in registers; the wave and the scalar kernel read sorted rows from the model's scratch (one small launch in front,
// the sign behind).
static int dispatch(const wf_model* m, int mode, const float* x, int64_t B, float* out, float* u, int32_t* idx, void* stream, bool presort = false) {
    DeviceGuard g(m->device);
    if (B == 0) return WF_OK;
    if (m->is_nsc) {
        if (mode == 1 || idx || presort) return WF_ERR_UNSUPPORTED;
        return launch_nsc_model(m->nsc, mode, x, B, out, u, stream);
    }
    const int Dm = m->desc.n_dim;
    // sorted rows + inversion counts for the kernels that do not sort themselves: behind the first `front` floats of the scratch
    auto presorted = [&](int64_t front, const float** xs, int32_t** inv) -> int {
        int rc = ensure_scratch(m, front + B * (Dm + 1));
        if (rc) return rc;
        float* s = m->d_scratch + front;
        *inv = reinterpret_cast<int32_t*>(s + B * Dm);
        rc = launch_sort_rows(*xs, B, Dm, s, *inv, stream);   // (*xs: the caller's rows -- replaced by the sorted copy only now)
        *xs = s;
        return rc;
    };
    // Small batches: one wave per walker (wf_kernels_wave.hip) takes 14 us for up to ~1000 walkers where the MFMA kernel,
    // which first stages its weight images into LDS, takes 38-42 us whatever the batch; from ~7000 walkers on the MFMA
    // kernel's throughput wins (4096: 30 vs 42 us, 8192: 46 vs 42 us).  The wave kernel does
    // not report bin indices.
    const bool wave_fits = m->wave_ok && !idx;
    const bool use_wave = wave_fits && (m->kernel_kind == WF_KERNEL_WAVE || (m->kernel_kind == WF_KERNEL_AUTO && B <= kWaveEvalMax));
    if (m->kernel_kind == WF_KERNEL_WAVE && !use_wave) return WF_ERR_UNSUPPORTED;
    if (use_wave) {
        const int D = m->desc.n_dim;
        const int64_t chunk = std::min<int64_t>(B, (int64_t)1 << 20);
        int rc = ensure_scratch(m, chunk * wave_tail_floats(D, 0) + (presort ? B * (D + 1) : 0));
        if (rc) return rc;
        int32_t* inv = nullptr;
        if (presort) {
            rc = presorted(chunk * wave_tail_floats(D, 0), &x, &inv);
            if (rc) return rc;
        }
        for (int64_t c0 = 0; c0 < B; c0 += chunk) {
            const int64_t bc = std::min(chunk, B - c0);
            rc = launch_wave_eval(m->dev, m->d_dev, m->d_tabI4, m->d_tabP3, m->d_grad_fk, mode, x + c0 * D, bc, out + c0, u ? u + c0 * D : nullptr,
                                  m->d_scratch, stream);
            if (rc) return rc;
        }
        return (presort && mode == 1) ? launch_apply_sign(out, inv, B, stream) : WF_OK;
    }
    bool use_mfma = m->mfma_ok && m->kernel_kind != WF_KERNEL_SCALAR;
    if (use_mfma && f16_overflow(m)) {   // a weight outside the fp16 range: the fp32 kernels only
        if (m->kernel_kind == WF_KERNEL_MFMA) return WF_ERR_UNSUPPORTED;
        use_mfma = false;
    }
#if defined(WF_DEBUG) || defined(WF_STAMP)
    if (use_mfma && getenv("WF_DBG_PTR")) const_cast<wf_model*>(m)->mdev.dbg = (float*)strtoull(getenv("WF_DBG_PTR"), nullptr, 0);
#endif
    if (use_mfma)
        return launch_mfma(m->dev.D, m->mdev.nbk, &m->mdev, (int)(m->mfma_lds_floats * sizeof(float)), mode | (presort ? kModePresort : 0), x, B, out, u, idx, stream);
    int32_t* inv = nullptr;
    if (presort) {
        int rc = presorted(0, &x, &inv);
        if (rc) return rc;
    }
    int rc = launch_scalar(m->dev, m->d_dev, mode, x, B, out, u, idx, stream);
    if (rc || !(presort && mode == 1)) return rc;
    return launch_apply_sign(out, inv, B, stream);
}

int wf_logpdf_fwd(const wf_model* m, const float* x_dev, int64_t B, float* logp_dev, float* u_dev, int32_t* bin_idx_dev,
                  void* stream) {
    int rc = check_fwd(m, x_dev, B, logp_dev);
    if (rc) return rc;
    return dispatch(m, 0, x_dev, B, logp_dev, u_dev, bin_idx_dev, stream);
}

int wf_psi_fwd(const wf_model* m, const float* x_dev, int64_t B, float* psi_dev, float* u_dev, int32_t* bin_idx_dev,
               void* stream) {
    int rc = check_fwd(m, x_dev, B, psi_dev);
    if (rc) return rc;
    if (m->desc.prior_kind != WF_PRIOR_WAVEFLOW) return WF_ERR_INVALID;
    return dispatch(m, 1, x_dev, B, psi_dev, u_dev, bin_idx_dev, stream);
}

int wf_psi_antisym_fwd(const wf_model* m, const float* x_dev, int64_t B, float* psi_dev, int32_t* inversions_dev, void* stream) {
    int rc = check_fwd(m, x_dev, B, psi_dev);
    if (rc) return rc;
    if (m->desc.prior_kind != WF_PRIOR_WAVEFLOW) return WF_ERR_INVALID;
    rc = dispatch(m, 1, x_dev, B, psi_dev, nullptr, nullptr, stream, true);
    if (rc || !inversions_dev || B == 0) return rc;
    DeviceGuard g(m->device);
    return launch_sort_rows(x_dev, B, m->desc.n_dim, nullptr, inversions_dev, stream);
}

int wf_logpdf_unsorted_fwd(const wf_model* m, const float* x_dev, int64_t B, float* logp_dev, void* stream) {
    int rc = check_fwd(m, x_dev, B, logp_dev);
    if (rc) return rc;
    return dispatch(m, 0, x_dev, B, logp_dev, nullptr, nullptr, stream, true);
}

int wf_inversion_count(const float* x_dev, int64_t B, int32_t n_dim, int32_t* count_dev, void* stream) {
    if (B < 0 || n_dim < 1 || n_dim > WF_MAX_DIM || (B > 0 && (!x_dev || !count_dev))) return WF_ERR_INVALID;
    if (wf_device_count() <= 0) return WF_ERR_NO_DEVICE;
    return launch_sort_rows(x_dev, B, n_dim, nullptr, count_dev, stream);
}

int wf_flow_fwd(const wf_model* m, const float* x_dev, int64_t B, float* u_dev, float* logdet_dev, void* stream) {
    int rc = check_fwd(m, x_dev, B, logdet_dev);
    if (rc) return rc;
    if (B > 0 && !u_dev) return WF_ERR_INVALID;
    return dispatch(m, 2, x_dev, B, logdet_dev, u_dev, nullptr, stream);
}

int wf_layer_fwd(const wf_model* m, int layer, const float* u_in_dev, int64_t B, float* y_dev, float* logdet_dev,
                 int32_t* bin_idx_dev, void* stream) {
    int rc = check_fwd(m, u_in_dev, B, y_dev);
    if (rc) return rc;
    if (layer < 0 || layer >= m->desc.n_flow_layers || (B > 0 && !logdet_dev)) return WF_ERR_INVALID;
    if (m->is_nsc) return WF_ERR_UNSUPPORTED;
    DeviceGuard g(m->device);
    if (B == 0) return WF_OK;
    return launch_scalar_layer(m->dev, m->d_dev, layer, u_in_dev, B, y_dev, logdet_dev, bin_idx_dev, stream);
}

// Inverse / sampler.  One wave per walker (wf_kernels_wave.hip: 64-way mesh search instead of the halving loop, 64 rejection
// proposals per round) finishes 128 walkers in 29 us where one lane per walker (wf_kernels_scalar.hip, the reference-order
// loops) needs 1.4 ms, and stays ahead up to ~2^18 walkers (65536: 1.1 vs 1.9 ms; 2^18: 4.4 vs 4.2-4.6 ms; 2^20: 17.3 vs
// 15.9-17.3 ms, scratch/sampler_crossover.py): the switch sits at 2^17.
static constexpr int64_t kWaveSampleMax = 131072;
static int64_t wave_sample_max() {   // tuning knob (read at every call): WF_WAVE_SAMPLE_MAX overrides the switch point
    const char* e = getenv("WF_WAVE_SAMPLE_MAX");
    return e ? atoll(e) : kWaveSampleMax;
}

// Large batches of the two-particle family (the family of the matrix-core local energy, <= 64 bases): the staged inverse / sampler of
// wf_kernels_etile.hip (conditioners on the matrix cores, one lane per walker for the searches).  WF_SAMPLE_TILE_MIN (read per call) moves the switch
// point; 0 disables the path.  It reads the MFMA image and the composite dimension-0 tables: not while they are stale (deferred training steps).
static constexpr int64_t kTileSampleMin = 16384;
static constexpr int64_t kTileSampleChunk = 1 << 18;   // walkers per pass of a call without a caller's workspace (the model's scratch: 111 MB)
// capability at this batch size (the model, the knob): what workspace queries and the training steps' refresh decisions go by -- independent of
// transient state
static bool tile_sample_capable_at(const wf_model* m, int64_t B) {
    const char* e = getenv("WF_SAMPLE_TILE_MIN");
    const int64_t mn = e ? atoll(e) : kTileSampleMin;
    const wf_model_desc& d = m->desc;
    // (the piecewise-constant envelope of k_tsample's second prior column takes the maximum over at most 9 coefficients per knot interval:
    // prior degrees above 8 keep the wave / one-lane samplers, whose bound is the global one)
    return mn > 0 && B >= mn && d.n_dim == 2 && m->nbp == 32 * m->mdev.nbk && m->mfma_ok && d.box_kind == WF_BOX_MEAN && d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0 &&
           d.prior_kind == WF_PRIOR_WAVEFLOW && d.p_degree <= 8 && m->d_tabI4 && m->d_tabP3 && m->dev.b_to_ob && m->d_grad_fk && tile_sample_capable(&m->mdev);
}
// ... and right now: the MFMA image and the composite tables are fresh, every packed weight inside the fp16 range
static bool tile_sample_ok(const wf_model* m, int64_t B) { return tile_sample_capable_at(m, B) && !m->eval_tables_stale && !f16_overflow(m); }
// in passes of what the workspace holds; a walker's stream is keyed by its index in the batch, whatever the passes
static int run_tile_sample(const wf_model* m, int draw, uint64_t seed, const float* u_dev, int64_t B, float* x_dev, float* latent_dev, int exact,
                           const unsigned long long* counter_dev, float* ws, int64_t ws_floats, void* stream) {
    int64_t chunk = B;
    while (chunk > 32 && tile_sample_floats(chunk, m->mdev.nbk) > ws_floats) chunk = ((chunk / 2) + 31) / 32 * 32;
    if (tile_sample_floats(chunk, m->mdev.nbk) > ws_floats) return WF_ERR_INVALID;
    for (int64_t c0 = 0; c0 < B; c0 += chunk) {
        const int64_t bc = std::min(chunk, B - c0);
        int rc = launch_tile_sample(&m->mdev, m->dev, m->d_tabI4, m->d_tabP3, m->d_grad_fk, draw, (unsigned long long)seed, u_dev ? u_dev + c0 * 2 : nullptr, bc,
                                    x_dev + c0 * 2, latent_dev ? latent_dev + c0 * 2 : nullptr, exact, counter_dev, c0, ws, stream);
        if (rc) return rc;
    }
    return WF_OK;
}

int wf_inverse_fwd(const wf_model* m, const float* u_dev, int64_t B, float* x_dev, int32_t exact, void* stream) {
    int rc = check_fwd(m, u_dev, B, x_dev);
    if (rc) return rc;
    DeviceGuard g(m->device);
    if (B == 0) return WF_OK;
    if (m->is_nsc) {   // a coupling layer's inverse is exact either way; the log-det of the inverse goes to the model's scratch
        rc = ensure_scratch(m, B);
        if (rc) return rc;
        return launch_nsc_model(m->nsc, 3, u_dev, B, m->d_scratch, x_dev, stream);
    }
    if (tile_sample_ok(m, B)) {
        const int64_t fl = tile_sample_floats(std::min(B, kTileSampleChunk), m->mdev.nbk);
        rc = ensure_scratch(m, fl);
        if (rc) return rc;
        return run_tile_sample(m, 0, 0, u_dev, B, x_dev, nullptr, exact, nullptr, m->d_scratch, fl, stream);
    }
    if (m->wave_ok && B <= wave_sample_max())
        return launch_wave_sample(m->dev, m->d_dev, m->d_tabI4, m->d_tabP3, m->d_grad_fk, 0, 0ull, u_dev, B, x_dev, nullptr, exact, nullptr, stream);
    return launch_scalar_inverse(m->dev, m->d_dev, u_dev, B, x_dev, exact, stream);
}

int wf_sample(const wf_model* m, uint64_t seed, int64_t B, float* x_dev, float* latent_dev, int32_t exact, void* stream) {
    int rc = check_fwd(m, x_dev, B, x_dev);
    if (rc) return rc;
    DeviceGuard g(m->device);
    if (B == 0) return WF_OK;
    if (m->is_nsc) {   // z ~ prior (Philox, keyed like the other samplers), x = inverse(z)
        rc = ensure_scratch(m, B * (m->desc.n_dim + 1));
        if (rc) return rc;
        float* z = latent_dev ? latent_dev : m->d_scratch + B;
        rc = launch_nsc_latent(m->desc.prior_kind, m->desc.n_dim, (unsigned long long)seed, B, z, stream);
        if (rc) return rc;
        return launch_nsc_model(m->nsc, 3, z, B, m->d_scratch, x_dev, stream);
    }
    if (tile_sample_ok(m, B)) {
        const int64_t fl = tile_sample_floats(std::min(B, kTileSampleChunk), m->mdev.nbk);
        rc = ensure_scratch(m, fl);
        if (rc) return rc;
        return run_tile_sample(m, 1, seed, nullptr, B, x_dev, latent_dev, exact, nullptr, m->d_scratch, fl, stream);
    }
    if (m->wave_ok && B <= wave_sample_max())
        return launch_wave_sample(m->dev, m->d_dev, m->d_tabI4, m->d_tabP3, m->d_grad_fk, 1, (unsigned long long)seed, nullptr, B, x_dev, latent_dev,
                                  exact, nullptr, stream);
    return launch_scalar_sample(m->dev, m->d_dev, (unsigned long long)seed, B, x_dev, latent_dev, exact, stream);
}

// grows the model's private device scratch (tails of the wave kernels when the caller passes no workspace)
static int ensure_scratch(const wf_model* cm, int64_t floats) {
    wf_model* m = const_cast<wf_model*>(cm);
    if (m->scratch_floats >= floats) return WF_OK;
    if (m->d_scratch) {
        WF_HIP(hipDeviceSynchronize());
        (void)hipFree(m->d_scratch);
        m->allocs.erase(std::remove(m->allocs.begin(), m->allocs.end(), (void*)m->d_scratch), m->allocs.end());
        m->d_scratch = nullptr;
        m->scratch_floats = 0;
    }
    int rc = dev_alloc(m, &m->d_scratch, (size_t)floats);
    if (rc) return rc;
    m->scratch_floats = floats;
    return WF_OK;
}

int wf_hamiltonian_fwd(const wf_model* m, const float* x_dev, int64_t B, const float* protons_host, int32_t n_protons, float* hpsi_dev,
                       float* psi_dev, float* laplacian_dev, void* stream) {
    int rc = check_fwd(m, x_dev, B, hpsi_dev);
    if (rc) return rc;
    if (n_protons < 0 || n_protons > 8 || (n_protons > 0 && !protons_host)) return WF_ERR_INVALID;
    if (m->desc.prior_kind != WF_PRIOR_WAVEFLOW) return WF_ERR_INVALID;
    if (!m->wave_ok || !m->d_tabP3 || !m->d_grad_fk) return WF_ERR_UNSUPPORTED;
    if (m->desc.n_flow_layers > 0 && m->desc.layer_kind != WF_LAYER_IMADE) return WF_ERR_UNSUPPORTED;
    Protons pr{};
    pr.n = n_protons;
    for (int i = 0; i < n_protons; ++i) pr.pos[i] = protons_host[i];
    DeviceGuard g(m->device);
    if (B == 0) return WF_OK;
    const int D = m->desc.n_dim;
    const int64_t chunk = std::min<int64_t>(B, (int64_t)1 << 20);
    // Large batches of the two-particle family: conditioner jets on the matrix cores + lane-per-walker heads (wf_kernels_etile.hip).
    // WF_ENERGY_TILE_MIN (read per call) moves the switch point; 0 disables the path.
    {
        const char* e = getenv("WF_ENERGY_TILE_MIN");
        const int64_t tile_min = e ? atoll(e) : kEnergyTileMin;
        const wf_model_desc& d = m->desc;
        // (<= 32 bases: the one-kernel form or, if the nets do not fit LDS together, the launch-per-net form; 33 .. 64 bases: the one-kernel form only)
        const bool family = D == 2 && (m->nbp == 32 || (m->nbp == 64 && m->mfma_ok && energy_tile_fused(&m->mdev))) && m->mfma_ok && d.box_kind == WF_BOX_MEAN && d.layer_kind == WF_LAYER_IMADE &&
                            d.n_flow_layers > 0 && !m->dev.i_gate && !m->dev.p_gate && m->d_tabI4c && m->d_tabP4c && (!m->mdev.p_bias || energy_tile_fused(&m->mdev)) && !getenv("WF_ENERGY_R3");   // (a constant term of the prior's boundary map: the one-kernel form only)
        if (family && tile_min > 0 && B >= tile_min && !m->eval_tables_stale && !f16_overflow(m)) {
            // the conditioner and the head kernels exchange 384 B per walker and net through the scratch buffer: chunks that keep it
            // (and its re-use by the next net and the next chunk) inside the 256 MB memory-side cache instead of HBM
            if (energy_tile_fused(&m->mdev))   // every net resident in LDS: one launch for the whole batch, no exchange buffer (k_efused)
                return launch_energy_tile(&m->mdev, m->dev, m->d_tabI4c, m->d_tabP4c, m->d_grad_fk, x_dev, B, pr, hpsi_dev, psi_dev, laplacian_dev, nullptr, stream);
            const char* ec = getenv("WF_ENERGY_TILE_CHUNK");
            const int64_t tchunk = std::min<int64_t>(B, std::max<int64_t>(ec ? atoll(ec) : kEnergyTileChunk, 1024));
            rc = ensure_scratch(m, energy_tile_floats(tchunk));
            if (rc) return rc;
            for (int64_t c0 = 0; c0 < B; c0 += tchunk) {
                const int64_t bc = std::min(tchunk, B - c0);
                rc = launch_energy_tile(&m->mdev, m->dev, m->d_tabI4c, m->d_tabP4c, m->d_grad_fk, x_dev + c0 * D, bc, pr, hpsi_dev + c0,
                                        psi_dev ? psi_dev + c0 : nullptr, laplacian_dev ? laplacian_dev + c0 : nullptr, m->d_scratch, stream);
                if (rc) return rc;
            }
            return WF_OK;
        }
    }
    // Large batches beyond two particles: one coordinate direction at a time with Taylor triples on the matrix cores (wf_kernels_etile_dir.hip), the same
    // switch point and knobs as the two-particle tile path
    {
        const char* e = getenv("WF_ENERGY_TILE_MIN");
        const int64_t tile_min = e ? atoll(e) : kEnergyTileMin;
        const wf_model_desc& d = m->desc;
        const bool family = D >= 3 && m->nbp == 32 && m->mfma_ok && d.box_kind == WF_BOX_MEAN && d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0 && !m->dev.i_gate &&
                            !m->dev.p_gate && m->d_tabI4c && m->d_tabP4c && energy_dir_capable(&m->mdev) && !getenv("WF_ENERGY_R3");
        if (family && tile_min > 0 && B >= tile_min && !m->eval_tables_stale && !f16_overflow(m)) {
            const int64_t dchunk = std::min<int64_t>(B, (int64_t)1 << 18);   // 12 D (D + 1) bytes of jets per walker: 226 MB at D = 8
            rc = ensure_scratch(m, energy_dir_floats(dchunk, D));
            if (rc) return rc;
            for (int64_t c0 = 0; c0 < B; c0 += dchunk) {
                const int64_t bc = std::min(dchunk, B - c0);
                rc = launch_energy_dir(&m->mdev, m->dev, m->d_tabI4c, m->d_tabP4c, x_dev + c0 * D, bc, pr, hpsi_dev + c0, psi_dev ? psi_dev + c0 : nullptr,
                                       laplacian_dev ? laplacian_dev + c0 : nullptr, m->d_scratch, stream);
                if (rc) return rc;
            }
            return WF_OK;
        }
    }
    rc = ensure_scratch(m, chunk * wave_tail_floats(D, 1));
    if (rc) return rc;
    for (int64_t c0 = 0; c0 < B; c0 += chunk) {
        const int64_t bc = std::min(chunk, B - c0);
        rc = launch_wave_energy(m->dev, m->d_dev, m->d_tabI4, m->d_tabP3, m->d_grad_fk, x_dev + c0 * D, bc, pr, hpsi_dev + c0,
                                psi_dev ? psi_dev + c0 : nullptr, laplacian_dev ? laplacian_dev + c0 : nullptr, m->d_scratch, stream);
        if (rc) return rc;
    }
    return WF_OK;
}

// workspace of the reverse pass per walker: tape + tails of its samples, plus 4 floats (H psi, psi, w_psi, w_lap)
static int64_t vjp_bytes_per_walker(const wf_model* m, bool second_order) {
    const int D = m->desc.n_dim;
    const int kind = second_order ? m->ring2 : 0;
    const int64_t samples = ring_samples(D, kind), nc = ring_coefs(D, kind);
    const int64_t zrows = m->z_rows;   // gated heads: one zero_params adjoint per (sample, head lane)
    return (samples * ((int64_t)m->nets.size() * grad_ws_rows(D, m->nbp) * nc + zrows) + wave_tail_floats(D, kind) + 4) * (int64_t)sizeof(float);
}

// the matrix-core gradient path (wf_kernels_etile.hip) applies to this model at this batch size: capability + the knob WF_GRAD_TILE_MIN (read per call),
// independent of transient state (stale evaluation tables, fp16 range of the current parameters)
static bool grad_tile_capable_at(const wf_model* m, int64_t B) {
    if (!m->d_egacc) return false;
    const char* e = getenv("WF_GRAD_TILE_MIN");
    const int64_t tile_min = e ? atoll(e) : kGradTileMin;
    const wf_model_desc& d = m->desc;
    return tile_min > 0 && B >= tile_min && d.n_dim == 2 && (m->nbp == 32 || m->nbp == 64) && m->mfma_ok && d.box_kind == WF_BOX_MEAN && d.layer_kind == WF_LAYER_IMADE && d.n_flow_layers > 0 &&
           d.prior_kind == WF_PRIOR_WAVEFLOW && m->d_tabI4c && m->d_tabP4c && energy_vjp_capable(&m->mdev);
}

static int64_t vjp_ws_bytes(const wf_model* m, int64_t B, bool second_order) {
    if (!m || B < 0) return WF_ERR_INVALID;
    if (!m->d_grad_map || (second_order && !m->grad_psi_ok)) return WF_ERR_UNSUPPORTED;
    const int64_t chunk = std::min<int64_t>(std::max<int64_t>(B, 1), 32768);
    int64_t bytes = chunk * vjp_bytes_per_walker(m, second_order);
    if (second_order && grad_tile_capable_at(m, B)) {   // the matrix-core gradient path has a fixed part (the partial gradient blocks of every net): only where the path applies (a smaller WF_GRAD_TILE_MIN at query time moves it)
        const int n_nets = (int)m->nets.size();
        bytes = std::max<int64_t>(bytes, (energy_vjp_fixed_floats(n_nets, m->mdev.nbk) + ((chunk + 31) / 32 * 32) * energy_vjp_floats_per_walker(n_nets)) * (int64_t)sizeof(float));
    }
    return bytes;
}

// mode 0: log_pdf, w1 only;  mode 1: psi (w1) and, with second_order, its Laplacian (w2);
// mode 2: loss_fn_efficient (vqmc.py:193-212): the weights come from H psi of the same forward sweep, e_loc_dev is written
// mode 3: maximum likelihood: every walker carries the weight inv_count (signed), e_loc_dev receives log_pdf of the same sweep
static int run_vjp_chunks(const wf_model* m, int mode, bool second_order, const float* x_dev, int64_t B, const float* w1, const float* w2,
                          const Protons* pr, float running_average, float inv_count, float* e_loc_dev, float* grad_dev, void* workspace_dev,
                          int64_t workspace_bytes, void* stream, const float* running_average_dev = nullptr, int* defer_gather_split = nullptr) {
    const int D = m->desc.n_dim;
    const int64_t chunk = workspace_bytes / vjp_bytes_per_walker(m, second_order);
    if (B > 0 && chunk < 1) return WF_ERR_INVALID;
    DeviceGuard g(m->device);
    hipStream_t s = (hipStream_t)stream;
    const int n_nets = (int)m->nets.size();
    const int64_t fwd = plain_fwd_floats(D, m->nbp);
    const int64_t n_img = fwd * n_nets;
    const int kind = second_order ? m->ring2 : 0;
    const int64_t samples_per = ring_samples(D, kind), nc = ring_coefs(D, kind);
    float* tape = (float*)workspace_dev;
    float* tails = tape + chunk * samples_per * n_nets * grad_ws_rows(D, m->nbp) * nc;
    float* per_walker = tails + chunk * wave_tail_floats(D, kind);   // [4][chunk]
    float* zws = m->z_rows ? per_walker + 4 * chunk : nullptr;        // [chunk * samples_per][z_rows]
    if (B == 0) {   // the gradient of an empty batch is zero
        WF_HIP(hipMemsetAsync(grad_dev, 0, (size_t)m->n_params * sizeof(float), s));
        return WF_OK;
    }
    // Large batches of the two-particle family (the family of the one-kernel H psi, <= 64 bases): forward, reverse and weight-gradient products on the
    // matrix cores (wf_kernels_etile.hip: k_efused with the per-net input jets, k_ebwd per net with the weight-gradient products inside).  WF_GRAD_TILE_MIN (read per call) moves the
    // switch point; 0 disables the path.
    if ((mode == 1 || mode == 2) && second_order && m->d_egacc) {
        const int64_t tile_min = 1;
        const bool family = grad_tile_capable_at(m, B) && !m->eval_tables_stale && !f16_overflow(m);
        const int64_t per = energy_vjp_floats_per_walker(n_nets) * (int64_t)sizeof(float), fixed = energy_vjp_fixed_floats(n_nets, m->mdev.nbk) * (int64_t)sizeof(float);
        const int64_t tchunk = workspace_bytes > fixed ? ((workspace_bytes - fixed) / per) / 32 * 32 : 0;
        if (family && tile_min > 0 && B >= tile_min && tchunk >= 32) {
            Protons none{};
            for (int64_t c0 = 0; c0 < B; c0 += tchunk) {
                const int64_t bc = std::min(tchunk, B - c0);
                int rc = launch_energy_vjp(&m->mdev, m->dev, m->d_tabI4c, m->d_tabP4c, x_dev + c0 * D, bc, mode, w1 ? w1 + c0 : nullptr, w2 ? w2 + c0 : nullptr,
                                           pr ? *pr : none, running_average, running_average_dev, inv_count, e_loc_dev ? e_loc_dev + c0 : nullptr,
                                           (float*)workspace_dev, m->d_egacc, c0 > 0, stream);
                if (rc) return rc;
            }
            std::vector<int> offs((size_t)n_nets * 8);
            std::vector<float> c2((size_t)n_nets);
            for (int n = 0; n < n_nets; ++n) {
                const NetOffsets q = net_offsets(m, n);
                int* o = &offs[(size_t)n * 8];
                o[0] = (int)q.W0; o[1] = (int)q.b0; o[2] = (int)q.W1; o[3] = (int)q.b1; o[4] = (int)q.W2; o[5] = (int)q.b2; o[6] = q.NO; o[7] = m->nets[n].n_out;
                c2[n] = net_has_sigmoid_head(m, n) ? -1.4426950408889634f : 1.0f;
            }
            if (defer_gather_split) *defer_gather_split = 0;   // the gradient is in grad_dev
            return launch_energy_vjp_finish(m->d_egacc, n_nets, m->mdev.nbk, offs.data(), c2.data(), grad_dev, m->n_params, stream);
        }
    }
    const bool single = B <= chunk;   // one chunk: the partial images are summed by the gather itself (one launch less)
    int split = 0;
    for (int64_t c0 = 0; c0 < B; c0 += chunk) {
        const int64_t bc = std::min(chunk, B - c0);
        const float* x = x_dev + c0 * D;
        float *wp = per_walker + 2 * chunk, *wl = per_walker + 3 * chunk;
        int rc = launch_wave_fwd(m->dev, m->d_dev, kind, m->d_tabI4, m->d_tabP3, m->d_grad_fk, x, bc, tape, tails, 1, stream);
        if (rc) return rc;
        const float *cw1 = w1 ? w1 + c0 : nullptr, *cw2 = w2 ? w2 + c0 : nullptr;
        if (mode == 2) {
            // (writing E_L and the weights from inside the forward kernel -- possible in RF, where a sample is a whole walker -- was
            // measured slower: the kernel grows by more than the 4.6 us launch it saves)
            rc = launch_energy_seeds(D, kind, tails, x, bc, m->dev.constrained_mask, *pr, running_average, running_average_dev, inv_count,
                                     e_loc_dev + c0, wp, wl, stream);
            if (rc) return rc;
            cw1 = wp;
            cw2 = wl;
        }
        if (mode == 3) {
            rc = launch_tail_out(m->dev, 0, tails, bc, e_loc_dev + c0, nullptr, stream, wp, inv_count);   // log_pdf values + the constant weights
            if (rc) return rc;
            cw1 = wp;
        }
        rc = launch_wave_bwd(m->dev, m->d_dev, (mode == 0 || mode == 3) ? 0 : 1, kind, m->d_tabI4, m->d_tabP3, m->d_grad_fk, bc, cw1, cw2, tape,
                             tails, zws, stream);
        if (rc) return rc;
        if (zws) {
            rc = launch_zgrad_reduce(zws, bc * samples_per, m->z_rows, c0 > 0, m->d_zpart, m->d_zgrad, stream);
            if (rc) return rc;
        }
        rc = launch_wgrad(D, m->nbp, kind, n_nets, bc * samples_per, tape, m->d_grad_partial, c0 > 0, m->d_grad_img, fwd,
                          single ? &split : nullptr, stream);
        if (rc) return rc;
    }
    if (defer_gather_split) *defer_gather_split = single ? split : 0;
    if (single && defer_gather_split) return WF_OK;   // the caller reads m->d_grad_partial itself (launch_adam_partials; ungated models only)
    int rc = single ? launch_grad_gather_partials(m->d_grad_partial, split, n_img, m->d_grad_map, m->n_params, grad_dev, stream)
                    : launch_grad_gather(m->d_grad_img, m->d_grad_map, m->n_params, grad_dev, stream);
    if (rc || !zws) return rc;
    // the zero_params leaves (the gather wrote 0 there: they reach no weight-image entry)
    return launch_zgrad_scatter(m->d_zgrad, m->z_rows, m->d_zmap, m->d_zraw_off, m->d_plain, grad_dev, stream);
}

int64_t wf_psi_vjp_workspace_bytes(const wf_model* m, int64_t B) { return vjp_ws_bytes(m, B, true); }
int64_t wf_logpdf_vjp_workspace_bytes(const wf_model* m, int64_t B) { return vjp_ws_bytes(m, B, false); }

int wf_psi_vjp(const wf_model* m, const float* x_dev, int64_t B, const float* w_psi_dev, const float* w_lap_dev, float* grad_dev,
               void* workspace_dev, int64_t workspace_bytes, void* stream) {
    int rc = check_fwd(m, x_dev, B, grad_dev);
    if (rc) return rc;
    if (!grad_dev) return WF_ERR_INVALID;
    if (!m->d_grad_map || !m->grad_psi_ok) return WF_ERR_UNSUPPORTED;
    if (B > 0 && (!w_psi_dev || !w_lap_dev || !workspace_dev)) return WF_ERR_INVALID;
    return run_vjp_chunks(m, 1, true, x_dev, B, w_psi_dev, w_lap_dev, nullptr, 0.0f, 0.0f, nullptr, grad_dev, workspace_dev, workspace_bytes, stream);
}

int wf_logpdf_vjp(const wf_model* m, const float* x_dev, int64_t B, const float* w_dev, float* grad_dev, void* workspace_dev,
                  int64_t workspace_bytes, void* stream) {
    int rc = check_fwd(m, x_dev, B, grad_dev);
    if (rc) return rc;
    if (!grad_dev) return WF_ERR_INVALID;
    if (!m->d_grad_map) return WF_ERR_UNSUPPORTED;
    if (B > 0 && (!w_dev || !workspace_dev)) return WF_ERR_INVALID;
    return run_vjp_chunks(m, 0, false, x_dev, B, w_dev, nullptr, nullptr, 0.0f, 0.0f, nullptr, grad_dev, workspace_dev, workspace_bytes, stream);
}

int wf_logpdf_loss_grad(const wf_model* m, const float* x_dev, int64_t B, float weight, float* logp_dev, float* grad_dev, void* workspace_dev,
                        int64_t workspace_bytes, void* stream) {
    int rc = check_fwd(m, x_dev, B, grad_dev);
    if (rc) return rc;
    if (!grad_dev) return WF_ERR_INVALID;
    if (!m->d_grad_map) return WF_ERR_UNSUPPORTED;
    if (B > 0 && (!logp_dev || !workspace_dev)) return WF_ERR_INVALID;
    return run_vjp_chunks(m, 3, false, x_dev, B, nullptr, nullptr, nullptr, 0.0f, weight, logp_dev, grad_dev, workspace_dev, workspace_bytes, stream);
}

int wf_vqmc_loss_grad(const wf_model* m, const float* x_dev, int64_t B, const float* protons_host, int32_t n_protons, float running_average,
                      float inv_count, float* e_loc_dev, float* grad_dev, void* workspace_dev, int64_t workspace_bytes, void* stream) {
    int rc = check_fwd(m, x_dev, B, grad_dev);
    if (rc) return rc;
    if (!grad_dev) return WF_ERR_INVALID;
    if (n_protons < 0 || n_protons > 8 || (n_protons > 0 && !protons_host)) return WF_ERR_INVALID;
    if (!m->d_grad_map || !m->grad_psi_ok) return WF_ERR_UNSUPPORTED;
    if (B > 0 && (!e_loc_dev || !workspace_dev)) return WF_ERR_INVALID;
    Protons pr{};
    pr.n = n_protons;
    for (int i = 0; i < n_protons; ++i) pr.pos[i] = protons_host[i];
    return run_vjp_chunks(m, 2, true, x_dev, B, nullptr, nullptr, &pr, running_average, inv_count, e_loc_dev, grad_dev, workspace_dev, workspace_bytes,
                          stream);
}

// ---- one whole training step on the device (see include/waveflow_hip.h)
static int64_t align256(int64_t v) { return (v + 255) / 256 * 256; }

// Adam step of a captured training step: the gradient is either in `grad` (several chunks) or still in the per-split partial images
// of the single chunk (split > 0), in which case the gather is part of the update kernel
static int adam_from_sweep(wf_model* m, const wf_train_state* st, const float* grad, int split, float step_size, float b1, float b2, float eps,
                           void* stream) {
    const unsigned long long* counter = (const unsigned long long*)st->counter_dev;
    if (split > 0) {
        const int64_t n_img = plain_fwd_floats(m->desc.n_dim, m->nbp) * (int64_t)m->nets.size();
        return launch_adam_partials(st->params_dev, m->d_grad_partial, split, n_img, m->d_grad_map, st->m_dev, st->v_dev, m->n_params, step_size, b1,
                                    b2, eps, counter, stream);
    }
    return launch_adam(st->params_dev, grad, st->m_dev, st->v_dev, m->n_params, 0, step_size, b1, b2, eps, counter, stream);
}

int64_t wf_vqmc_train_step_workspace_bytes(const wf_model* m, int64_t batch) {
    if (!m || batch < 1) return WF_ERR_INVALID;
    // (capability, not the current state: a size queried while the evaluation tables are stale holds after the next full refresh as well)
    if (!m->d_grad_map || !m->grad_psi_ok || !m->wave_ok || (batch > kWaveSampleMax && !tile_sample_capable_at(m, batch))) return WF_ERR_UNSUPPORTED;
    // (the staged sampler of large batches works in the gradient's workspace before the gradient needs it)
    return align256(batch * m->desc.n_dim * 4) + align256(batch * 4) + align256(m->n_params * 4) + 256 + align256(block_sums_ws_bytes(batch)) +
           std::max<int64_t>(vjp_ws_bytes(m, batch, true), tile_sample_capable_at(m, batch) ? align256(tile_sample_floats(std::min(batch, kTileSampleChunk), m->mdev.nbk) * 4) : 0);
}

int wf_vqmc_train_step(wf_model* m, const wf_train_state* st, uint64_t seed, int64_t batch, const float* protons_host, int32_t n_protons,
                       float step_size, float b1, float b2, float eps, int32_t exact_sampler, void* workspace_dev, int64_t workspace_bytes,
                       void* stream) {
    if (!m || !st || batch < 1 || n_protons < 0 || n_protons > 8 || (n_protons > 0 && !protons_host)) return WF_ERR_INVALID;
    if (!st->params_dev || !st->m_dev || !st->v_dev || !st->counter_dev || !st->running_average_dev || !st->loss_ring_dev || st->ring_len < 1)
        return WF_ERR_INVALID;
    if (!m->d_grad_map || !m->grad_psi_ok || !m->wave_ok || (batch > kWaveSampleMax && !tile_sample_ok(m, batch))) return WF_ERR_UNSUPPORTED;
    if (!m->params_set || !workspace_dev || workspace_bytes < wf_vqmc_train_step_workspace_bytes(m, batch)) return WF_ERR_INVALID;
    DeviceGuard g(m->device);
    const int D = m->desc.n_dim;
    char* p = (char*)workspace_dev;
    float* x = (float*)p; p += align256(batch * D * 4);
    float* e_loc = (float*)p; p += align256(batch * 4);
    float* grad = (float*)p; p += align256(m->n_params * 4);
    double* sums = (double*)p; p += 256;
    void* sums_ws = p; p += align256(block_sums_ws_bytes(batch));
    const int64_t vjp_bytes = workspace_bytes - (p - (char*)workspace_dev);
    Protons pr{};
    pr.n = n_protons;
    for (int i = 0; i < n_protons; ++i) pr.pos[i] = protons_host[i];
    const unsigned long long* counter = (const unsigned long long*)st->counter_dev;
    // walkers ~ the sampler, stream advanced by the device counter
    int rc = tile_sample_ok(m, batch)
                 ? run_tile_sample(m, 1, seed, nullptr, batch, x, nullptr, exact_sampler, counter, (float*)p, vjp_bytes / 4, stream)
                 : launch_wave_sample(m->dev, m->d_dev, m->d_tabI4, m->d_tabP3, m->d_grad_fk, 1, (unsigned long long)seed, nullptr, batch, x, nullptr,
                                      exact_sampler, counter, stream);
    if (rc) return rc;
    // mean local energy and its gradient under the custom tangent rule, running average from the device scalar
    int split = 0;   // (gated heads: no deferred gather -- the flat gradient gets its zero_params entries, Adam reads it)
    rc = run_vjp_chunks(m, 2, true, x, batch, nullptr, nullptr, &pr, 0.0f, 1.0f / (float)batch, e_loc, grad, p, vjp_bytes, stream,
                        st->running_average_dev, m->z_rows ? nullptr : &split);
    if (rc) return rc;
    rc = adam_from_sweep(m, st, grad, split, step_size, b1, b2, eps, stream);
    if (rc) return rc;
    // A step whose batch size puts it on the matrix-core sampler / gradient reads the MFMA image and the composite tables: it refreshes them
    // whatever defer_eval_tables says -- a hipGraph of this step replays the kernels chosen at capture, and a deferred refresh would leave
    // them on stale tables from the second replay on (the selection above does not depend on the deferral either: with stale tables at
    // call time the step takes the wave sweeps, which are valid in every replay).
    rc = apply_params(m, st->params_dev, stream, !st->defer_eval_tables || tile_sample_capable_at(m, batch) || grad_tile_capable_at(m, batch));
    if (rc) return rc;
    // batch sums of the local energies -> loss ring, step counter + 1 (after Adam, which reads the counter as its step index)
    return launch_block_sums(e_loc, batch, sums, sums_ws, block_sums_ws_bytes(batch), stream, st->loss_ring_dev, st->ring_len,
                             (unsigned long long*)st->counter_dev);
}

int wf_vqmc_train_step_local(wf_model* m, const wf_train_state* st, uint64_t seed, int64_t batch_local, const float* protons_host, int32_t n_protons,
                             float inv_global_batch, int32_t exact_sampler, double* reduce_dev, void* workspace_dev, int64_t workspace_bytes,
                             void* stream) {
    if (!m || !st || !reduce_dev || batch_local < 1 || n_protons < 0 || n_protons > 8 || (n_protons > 0 && !protons_host)) return WF_ERR_INVALID;
    if (!st->counter_dev || !st->running_average_dev) return WF_ERR_INVALID;
    if (!m->d_grad_map || !m->grad_psi_ok || !m->wave_ok || (batch_local > kWaveSampleMax && !tile_sample_ok(m, batch_local))) return WF_ERR_UNSUPPORTED;
    if (!m->params_set || !workspace_dev || workspace_bytes < wf_vqmc_train_step_workspace_bytes(m, batch_local)) return WF_ERR_INVALID;
    DeviceGuard g(m->device);
    const int D = m->desc.n_dim;
    char* p = (char*)workspace_dev;
    float* x = (float*)p; p += align256(batch_local * D * 4);
    float* e_loc = (float*)p; p += align256(batch_local * 4);
    float* grad = (float*)p; p += align256(m->n_params * 4);
    p += 256;
    void* sums_ws = p; p += align256(block_sums_ws_bytes(batch_local));
    const int64_t vjp_bytes = workspace_bytes - (p - (char*)workspace_dev);
    Protons pr{};
    pr.n = n_protons;
    for (int i = 0; i < n_protons; ++i) pr.pos[i] = protons_host[i];
    int rc = tile_sample_ok(m, batch_local)
                 ? run_tile_sample(m, 1, seed, nullptr, batch_local, x, nullptr, exact_sampler, (const unsigned long long*)st->counter_dev, (float*)p,
                                   vjp_bytes / 4, stream)
                 : launch_wave_sample(m->dev, m->d_dev, m->d_tabI4, m->d_tabP3, m->d_grad_fk, 1, (unsigned long long)seed, nullptr, batch_local, x,
                                      nullptr, exact_sampler, (const unsigned long long*)st->counter_dev, stream);
    if (rc) return rc;
    int split = 0;
    rc = run_vjp_chunks(m, 2, true, x, batch_local, nullptr, nullptr, &pr, 0.0f, inv_global_batch, e_loc, grad, p, vjp_bytes, stream,
                        st->running_average_dev, m->z_rows ? nullptr : &split);
    if (rc) return rc;
    const int64_t n_img = plain_fwd_floats(D, m->nbp) * (int64_t)m->nets.size();
    rc = launch_pack_reduce_buffer(m->d_grad_partial, split, n_img, m->d_grad_map, grad, m->n_params, reduce_dev, stream);
    if (rc) return rc;
    m->local_step_tile = tile_sample_capable_at(m, batch_local) || grad_tile_capable_at(m, batch_local);   // -> wf_vqmc_train_step_apply refreshes everything
    return launch_block_sums(e_loc, batch_local, reduce_dev + m->n_params, sums_ws, block_sums_ws_bytes(batch_local), stream);
}

int wf_vqmc_train_step_apply(wf_model* m, const wf_train_state* st, const double* reduce_dev, float step_size, float b1, float b2, float eps,
                             void* stream) {
    if (!m || !st || !reduce_dev) return WF_ERR_INVALID;
    if (!st->params_dev || !st->m_dev || !st->v_dev || !st->counter_dev || !st->loss_ring_dev || st->ring_len < 1) return WF_ERR_INVALID;
    if (!m->d_grad_map) return WF_ERR_UNSUPPORTED;
    DeviceGuard g(m->device);
    int rc = launch_adam_reduced(st->params_dev, reduce_dev, st->m_dev, st->v_dev, m->n_params, step_size, b1, b2, eps,
                                 (const unsigned long long*)st->counter_dev, stream);
    if (rc) return rc;
    rc = apply_params(m, st->params_dev, stream, !st->defer_eval_tables || m->local_step_tile);   // (see wf_vqmc_train_step)
    if (rc) return rc;
    return launch_ring_push(reduce_dev + m->n_params, st->loss_ring_dev, st->ring_len, (unsigned long long*)st->counter_dev, stream);
}

int64_t wf_mle_train_step_workspace_bytes(const wf_model* m, int64_t N) {
    if (!m || N < 1) return WF_ERR_INVALID;
    if (!m->d_grad_map) return WF_ERR_UNSUPPORTED;
    return align256(N * 4) + align256(m->n_params * 4) + 256 + align256(block_sums_ws_bytes(N)) + vjp_ws_bytes(m, N, false);
}

int wf_mle_train_step(wf_model* m, const wf_train_state* st, const float* x_dev, int64_t N, float step_size, float b1, float b2, float eps,
                      void* workspace_dev, int64_t workspace_bytes, void* stream) {
    if (!m || !st || !x_dev || N < 1) return WF_ERR_INVALID;
    if (!st->params_dev || !st->m_dev || !st->v_dev || !st->counter_dev || !st->loss_ring_dev || st->ring_len < 1) return WF_ERR_INVALID;
    if (!m->d_grad_map) return WF_ERR_UNSUPPORTED;
    if (!m->params_set || !workspace_dev || workspace_bytes < wf_mle_train_step_workspace_bytes(m, N)) return WF_ERR_INVALID;
    DeviceGuard g(m->device);
    char* p = (char*)workspace_dev;
    float* lp = (float*)p; p += align256(N * 4);
    float* grad = (float*)p; p += align256(m->n_params * 4);
    double* sums = (double*)p; p += 256;
    void* sums_ws = p; p += align256(block_sums_ws_bytes(N));
    const int64_t vjp_bytes = workspace_bytes - (p - (char*)workspace_dev);
    // loss = -mean log_pdf (benchmark_tests.py:84-87): value from the forward sweep, gradient from the reverse sweep
    int split = 0;
    int rc = run_vjp_chunks(m, 3, false, x_dev, N, nullptr, nullptr, nullptr, 0.0f, -1.0f / (float)N, lp, grad, p, vjp_bytes, stream, nullptr,
                            m->z_rows ? nullptr : &split);
    if (rc) return rc;
    rc = adam_from_sweep(m, st, grad, split, step_size, b1, b2, eps, stream);
    if (rc) return rc;
    rc = apply_params(m, st->params_dev, stream, !st->defer_eval_tables);
    if (rc) return rc;
    return launch_block_sums(lp, N, sums, sums_ws, block_sums_ws_bytes(N), stream, st->loss_ring_dev, st->ring_len,
                             (unsigned long long*)st->counter_dev);
}

int wf_vqmc_seeds(const float* x_dev, int64_t B, int32_t n_dim, const float* protons_host, int32_t n_protons, const float* hpsi_dev,
                  const float* psi_dev, float running_average, float inv_count, float* e_loc_dev, float* w_psi_dev, float* w_lap_dev,
                  void* stream) {
    if (B < 0 || n_dim < 1 || n_dim > WF_MAX_DIM || n_protons < 0 || n_protons > 8 || (n_protons > 0 && !protons_host)) return WF_ERR_INVALID;
    if (B > 0 && (!x_dev || !hpsi_dev || !psi_dev || !e_loc_dev || !w_psi_dev || !w_lap_dev)) return WF_ERR_INVALID;
    if (B == 0) return WF_OK;
    Protons pr{};
    pr.n = n_protons;
    for (int i = 0; i < n_protons; ++i) pr.pos[i] = protons_host[i];
    return launch_vqmc_seeds(x_dev, B, n_dim, pr, hpsi_dev, psi_dev, running_average, inv_count, e_loc_dev, w_psi_dev, w_lap_dev, nullptr, stream);
}

int wf_rqs_fwd(const float* x_dev, const float* uw_dev, const float* uh_dev, const float* ud_dev, int64_t N, int32_t K, int32_t n_deriv,
               int32_t inverse, float left, float right, float bottom, float top, float* y_dev, float* logabsdet_dev, int32_t* bin_dev,
               void* stream) {
    if (N < 0 || K < 1 || K > 256) return WF_ERR_INVALID;
    if (n_deriv != K - 1 && n_deriv != K + 1) return WF_ERR_INVALID;
    if (!(right > left) || !(top > bottom)) return WF_ERR_INVALID;
    if (1e-3f * K > 1.0f) return WF_ERR_INVALID;   // "Minimal bin width too large for the number of bins" (neural_splines.py:91-94)
    if (N > 0 && (!x_dev || !uw_dev || !uh_dev || (!ud_dev && n_deriv > 0) || !y_dev || !logabsdet_dev)) return WF_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return WF_ERR_NO_DEVICE;
    return launch_rqs(x_dev, uw_dev, uh_dev, ud_dev, N, K, n_deriv, inverse, left, right, bottom, top, y_dev, logabsdet_dev, bin_dev, stream);
}

int64_t wf_nsc_workspace_bytes(int64_t B, int32_t dim, int32_t K) {
    if (B < 0 || dim < 2 || dim > WF_MAX_DIM || (dim & 1) || K < 2 || K > 32) return WF_ERR_INVALID;
    return nsc_workspace_floats(B, dim, K) * (int64_t)sizeof(float);
}

int wf_nsc_fwd(const float* x_dev, int64_t B, int32_t dim, int32_t K, float tail_bound, int32_t hidden, const float* params_dev, int32_t inverse,
               float* y_dev, float* logdet_dev, void* workspace_dev, int64_t workspace_bytes, void* stream) {
    if (B < 0 || dim < 2 || dim > WF_MAX_DIM || (dim & 1) || K < 2 || K > 32 || hidden < 1 || hidden > 64 || !(tail_bound > 0.0f)) return WF_ERR_INVALID;
    if (1e-3f * K > 1.0f) return WF_ERR_INVALID;
    if (B > 0 && (!x_dev || !params_dev || !y_dev || !logdet_dev || !workspace_dev)) return WF_ERR_INVALID;
    if (workspace_bytes < wf_nsc_workspace_bytes(B, dim, K)) return WF_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return WF_ERR_NO_DEVICE;
    // the shapes the one-kernel stack is built for (every reference default) never touch the workspace; WF_NSC_STAGED=1 forces the
    // launch-per-half-step path (the only one for other widths / bin counts)
    if (nsc_model_built(dim, K, hidden) && !getenv("WF_NSC_STAGED")) {
        const int dh = dim / 2, per = 3 * K - 1;
        const int64_t net_floats = (int64_t)dh * hidden + hidden + (int64_t)hidden * hidden + hidden + (int64_t)hidden * per * dh + (int64_t)per * dh;
        const NscModelDev md{dim, 1, K, hidden, WF_PRIOR_NORMAL, 0, tail_bound, 0.0f, params_dev, net_floats};
        return launch_nsc_model(md, inverse ? 3 : 2, x_dev, B, logdet_dev, y_dev, stream);
    }
    return launch_nsc(x_dev, B, dim, K, tail_bound, hidden, params_dev, inverse, y_dev, logdet_dev, (float*)workspace_dev, stream);
}

int64_t wf_block_sums_workspace_bytes(int64_t B) { return block_sums_ws_bytes(B); }

int wf_block_sums(const float* v_dev, int64_t B, double* out_dev, void* workspace_dev, int64_t workspace_bytes, void* stream) {
    if (B < 0 || !out_dev || (B > 0 && !v_dev)) return WF_ERR_INVALID;
    if (workspace_bytes < block_sums_ws_bytes(B) || !workspace_dev) return WF_ERR_INVALID;
    return launch_block_sums(v_dev, B, out_dev, workspace_dev, workspace_bytes, stream);
}

}  // extern "C"
