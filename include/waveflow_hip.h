/*
 * waveflow_hip.h -- C ABI of libwaveflow_hip.so (MI355X / gfx950).
 *
 * The reference (aspuru-guzik-group/waveflow) is pure Python/JAX and has no FFI;
 * its boundary for the flow-density hot path is three closure protocols
 * (SURVEY.md §8b).  Each entry point below names the reference closure(s) it
 * replaces (paths relative to /root/reference/waveflow).  The Python package
 * waveflow_amd re-creates those closures on top of this ABI (ctypes binding:
 * waveflow_amd/_lib.py; the stub a reference maintainer would add is shown in
 * INTEGRATION.md).
 *
 * Conventions
 *   - plain C types only; every function returns 0 (WF_OK) or a negative
 *     wf_status, never throws;  wf_strerror() gives the text;
 *   - the caller owns every buffer; `*_dev` pointers are device (HBM) pointers,
 *     row-major, fp32 unless stated; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream); launches are asynchronous on that stream;
 *   - the library owns wf_model (wf_model_create / wf_model_destroy).  The
 *     large-batch forward entry points (wf_logpdf_fwd / wf_psi_fwd / wf_flow_fwd
 *     / wf_layer_fwd above 6144 rows) only read the model: concurrent launches
 *     on different streams are safe.  The small-batch forward path,
 *     wf_inverse_fwd and wf_sample (from 16 384 rows on they stage conditioner
 *     outputs in the model's scratch buffer, which may be re-allocated -- with a
 *     device synchronisation -- when a call needs more than any before: not inside
 *     a stream capture unless a call of that size has run before),
 *     wf_hamiltonian_fwd and the gradient / training entry points use per-model
 *     scratch: issue those for one model on one stream at a time (or serialise
 *     them with events).  wf_model_set_params(_device) must not run concurrently
 *     with launches that use the model;
 *   - fp16 range: the matrix-core kernels read the weights behind a tanh as fp16
 *     pairs of scale * W (|scale| <= 5.8).  A parameter vector with such a product
 *     at or beyond 65 520 is detected at upload: while it is loaded WF_KERNEL_AUTO
 *     takes the fp32 kernels (scalar / wave sweeps), WF_KERNEL_MFMA returns
 *     WF_ERR_UNSUPPORTED, and a matrix-core kernel that runs anyway (a replayed
 *     hipGraph captured with other parameters) writes NaN, never a finite wrong
 *     value;
 *   - there is no CPU fallback: on a machine without a gfx950 device every
 *     device entry point returns WF_ERR_NO_DEVICE.
 */
#ifndef WAVEFLOW_HIP_H
#define WAVEFLOW_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WF_ABI_VERSION 2
#define WF_MAX_DIM 16
#define WF_MAX_BC 4

typedef enum {
    WF_OK = 0,
    WF_ERR_INVALID = -1,      /* bad argument / inconsistent descriptor                */
    WF_ERR_UNSUPPORTED = -2,  /* valid in the reference, not built here */
    WF_ERR_HIP = -3,          /* a HIP runtime call failed (wf_last_hip_error())         */
    WF_ERR_NO_DEVICE = -4,    /* no gfx950 device visible                               */
    WF_ERR_NOMEM = -5,
    WF_ERR_NUMERIC = -6       /* table build: singular Gram matrix, odd basis count ...  */
} wf_status;

/* spline families: splines/msplines_jax.py, isplines_jax.py, bsplines_jax.py */
typedef enum { WF_SPLINE_M = 0, WF_SPLINE_I = 1, WF_SPLINE_B = 2, WF_SPLINE_OB = 3 } wf_spline_kind;
/* bijector layers: flows/bijections/made.py:44-105 (IMADE), :7-41 (MADE) */
typedef enum { WF_LAYER_IMADE = 0, WF_LAYER_MADE = 1, WF_LAYER_NSC = 2 } wf_layer_kind;
/* BoxTransformLayer xu_coord_type: made.py:156-183 ('mean'), :118-137 ('first') */
typedef enum { WF_BOX_NONE = 0, WF_BOX_MEAN = 1, WF_BOX_FIRST = 2 } wf_box_kind;
/* density heads: wavefunctions.py:9-112 (Waveflow), flows/distributions.py:116-194 (MFlow),
 * :67-112 (Flow) with Uniform (:26-41, prior_support=(0,1)) or Normal(offset) (:8-23) priors */
typedef enum { WF_PRIOR_WAVEFLOW = 0, WF_PRIOR_MFLOW = 1, WF_PRIOR_UNIFORM = 2, WF_PRIOR_NORMAL = 3 } wf_prior_kind;
/* which device kernel evaluates the model */
typedef enum { WF_KERNEL_AUTO = 0, WF_KERNEL_SCALAR = 1, WF_KERNEL_MFMA = 2, WF_KERNEL_WAVE = 3 } wf_kernel_kind;

/* constraints_dict_{left,right}: {n_derivative: value} in insertion order
 * (isplines_jax.py:158-194, bsplines_jax.py:173-199, msplines_jax.py:156-184) */
typedef struct {
    int32_t n;
    int32_t n_derivative[WF_MAX_BC];
    float value[WF_MAX_BC];
} wf_bc;

/* Everything model_factory.get_waveflow_model (model_factory.py:121-146), get_model (:96-116) and
 * benchmark_tests.get_model (benchmark_tests.py:50-78) close over. */
typedef struct {
    int32_t n_dim;              /* input_dim D (>= 2: hidden degrees are arange(64) % (D-1), model_factory.py:15) */
    int32_t hidden;             /* 64, model_factory.py:72 */
    int32_t n_flow_layers;      /* (IMADE|MADE, Reverse) pairs */
    int32_t layer_kind;         /* wf_layer_kind */
    int32_t box_kind;           /* wf_box_kind */
    float box_size;             /* box_side */
    int32_t i_degree;           /* IMADE spline_degree */
    int32_t i_knots;            /* IMADE n_internal_knots */
    float i_reg;                /* spline_regularization, made.py:68 */
    wf_bc i_left, i_right;      /* IMADE constraints */
    int32_t prior_kind;         /* wf_prior_kind */
    int32_t p_degree, p_knots;  /* prior spline degree / n_internal_knots */
    wf_bc p_left, p_right;      /* prior constraints */
    float normal_offset;        /* Normal(offset) */
    int32_t n_constrained_left; /* constrained_dimension_indices_left, model_factory.py:124-129 */
    int32_t constrained_left[WF_MAX_DIM];
    int32_t n_mesh;             /* n_spline_base_mesh_points (2000) */
    float i_reverse_tol;        /* IMADE reverse_fun_tol (bisection tolerance of the inverse, isplines_jax.py:153-156) */
    /* set_nn_output_grad_to_zero of the layers' / the prior's conditioner (model_factory.py:55-67, ABI 2):
     *   bij[d][j] = g_d(x) * head(o[d][j]) + z[d][j],   g_0 = 1, g_d = prod_{i<d} x_i^3  (x: the conditioner's input),
     * z = zero_params[d][j] (its absolute value under a sigmoid head), before the division by sum_j bij[d][j].
     * Built everywhere: the three forward kernels (in the wave sweep the gate travels as a jet: wf_hamiltonian_fwd), inverse and both
     * samplers, the gradient entry points (zero_params leaves included; d|z| = sign(z) under a sigmoid head) and the fused training
     * steps.  The oracle for this branch restates the four reference lines; the reference ships no output of a gated model (unpinned). */
    int32_t i_gate, p_gate;
    /* layer_kind WF_LAYER_NSC (ABI 2): n_flow_layers NeuralSplineCoupling layers (flows/bijections/neural_splines.py:244-296; K bins,
     * tail bound B, FCNN conditioners of width hidden_dim), each followed by flows.Reverse when nsc_reverse != 0, under a Normal or
     * Uniform prior: Flow(Serial(...), Normal()) in one launch per call.  n_dim even, 2..8; K 2..16; hidden 8 or 32.  Parameters:
     * per layer f1 then f2, each in stax.Dense leaf order W1 [dh][h], b1, W2 [h][h], b2, W3 [h][(3K-1) dh], b3.
     * wf_logpdf_fwd, wf_flow_fwd, wf_inverse_fwd, wf_sample; everything else returns WF_ERR_UNSUPPORTED for this kind. */
    int32_t nsc_bins;
    float nsc_tail_bound;
    int32_t nsc_hidden;
    int32_t nsc_reverse;
} wf_model_desc;

typedef struct wf_model wf_model;

/* -- host-only ----------------------------------------------------------------------------- */

/* Basis tables, replaces the one-off table build in ISpline_fun/BSpline_fun/MSpline_fun.init_fun
 * (isplines_jax.py:106-131, bsplines_jax.py:68-116, msplines_jax.py:84-108 -> splines_np.py:42-137,
 * ortho_splines.py:43-161).  out: fp64 [4][n_bases][n_mesh] (derivative orders 0..3).  For
 * WF_SPLINE_OB, b_to_ob / ob_to_b ([n_bases][n_bases], may be NULL) receive the change-of-basis
 * matrices.  Returns n_bases (> 0) or a negative wf_status; with out == NULL only returns n_bases. */
int wf_tables_build(int kind, int degree, int n_internal_knots, int n_mesh, double* out, double* b_to_ob,
                    double* ob_to_b);

const char* wf_strerror(int status);
int wf_abi_version(void);
/* last hipError_t seen by this thread (as int) and its string */
int wf_last_hip_error(void);
const char* wf_last_hip_error_string(void);
/* number of visible gfx950 devices (0 if none / no driver) */
int wf_device_count(void);

/* -- model ---------------------------------------------------------------------------------- */

/* Builds tables, masks and device buffers on `device`.  Replaces init_fun(rng, input_dim) of
 * Waveflow / MFlow / Flow (wavefunctions.py:13-31, distributions.py:118-137, :89-93) minus the
 * random parameter draw. */
int wf_model_create(const wf_model_desc* desc, int device, wf_model** out);
void wf_model_destroy(wf_model* m);

/* number of fp32 parameters = pytree leaves of the reference's `params`, flattened in leaf order:
 * per flow layer W0[D,64] b0[64] W1[64,64] b1[64] W2[64,D*nb] b2[D*nb] zero[D,nb] (MADE: no zero,
 * nb = 2), then the prior net in the same order (WAVEFLOW / MFLOW priors only). */
int64_t wf_model_param_count(const wf_model* m);
int wf_model_n_bases(const wf_model* m, int which /* 0 = flow-layer spline, 1 = prior spline */);

/* Uploads parameters (host pointer, flat leaf order) and re-derives the device-side masked /
 * MFMA-permuted weight images.  Synchronous with respect to `stream`. */
int wf_model_set_params(wf_model* m, const float* flat_host, int64_t n, void* stream);

/* The same from a device-resident flat vector (fp32, same leaf order): asynchronous on `stream`, no host work -- the images
 * are filled by a device kernel from per-model descriptions, so a training loop can keep parameters, gradient and optimiser
 * state on the GPU (and the call can be captured in a hipGraph). */
int wf_model_set_params_device(wf_model* m, const float* flat_dev, int64_t n, void* stream);

/* One Adam step on device vectors, the rule of jax.example_libraries.optimizers.adam (vqmc.py:136; `step` is the index passed
 * to opt_update): m <- (1-b1) g + b1 m; v <- (1-b2) g^2 + b2 v; x <- x - step_size * m/(1-b1^(step+1)) / (sqrt(v/(1-b2^(step+1))) + eps). */
int wf_adam_step(float* params_dev, const float* grad_dev, float* m_dev, float* v_dev, int64_t n, int64_t step, float step_size, float b1,
                 float b2, float eps, void* stream);

/* select the kernel used by the *_fwd entry points (default WF_KERNEL_AUTO) */
int wf_model_set_kernel(wf_model* m, int kernel_kind);

/* -- device entry points (the hot path) ----------------------------------------------------- */

/* log_pdf(params, inputs[, return_sample]) of Waveflow (wavefunctions.py:33-52), MFlow
 * (distributions.py:139-163), Flow (distributions.py:95-102).
 *   x_dev   [B][D]   walker coordinates
 *   logp_dev[B]
 *   u_dev   [B][D] or NULL: the latent sample (`return_sample=True`), after the prior's clip
 *   bin_idx_dev [B][n_flow_layers+1][D][2] int32 or NULL: (x_l, x_r) = (floor, ceil)(u*(n_mesh-1))
 *           of every table lookup (isplines_jax.py:47-48), debug / parity output */
int wf_logpdf_fwd(const wf_model* m, const float* x_dev, int64_t B, float* logp_dev, float* u_dev,
                  int32_t* bin_idx_dev, void* stream);

/* psi(params, inputs) of Waveflow (wavefunctions.py:54-71); WF_PRIOR_WAVEFLOW models only. */
int wf_psi_fwd(const wf_model* m, const float* x_dev, int64_t B, float* psi_dev, float* u_dev,
               int32_t* bin_idx_dev, void* stream);

/* The antisymmetrised wavefunction of walkers in ANY particle order: psi(params, sort(x)) * (-1)^inversions(x) -- what the
 * reference's callers compute on the host around psi (utils/helpers.py:55-58: `inversion_count = get_num_inversion_count(c);
 * z = psi(params, np.sort(c, -1)) * (-1) ** inversion_count`; utils/coordinates.py:41-51; tests/test_waveflow.py:39-43).  Each
 * row is sorted ascending on the device (the matrix-core kernel: in registers, a network of adjacent exchanges whose count is
 * the inversion count, ties exchanging nothing -- `c[i] > c[j]` strictly, as the reference counts) and psi gets the sign.
 *   inversions_dev [B] int32 or NULL: get_num_inversion_count of every row
 * wf_logpdf_unsorted_fwd: log_pdf(params, sort(x)), the density of the unordered configuration's ordered image (no sign).
 * wf_inversion_count: coordinates.get_num_inversion_count alone, no model. */
int wf_psi_antisym_fwd(const wf_model* m, const float* x_dev, int64_t B, float* psi_dev, int32_t* inversions_dev, void* stream);
int wf_logpdf_unsorted_fwd(const wf_model* m, const float* x_dev, int64_t B, float* logp_dev, void* stream);
int wf_inversion_count(const float* x_dev, int64_t B, int32_t n_dim, int32_t* count_dev, void* stream);

/* Serial(...).direct_fun (bijections.py:452-460): u[B][D], logdet[B] of the whole bijector stack. */
int wf_flow_fwd(const wf_model* m, const float* x_dev, int64_t B, float* u_dev, float* logdet_dev, void* stream);

/* One bijector layer's direct_fun (IMADE made.py:66-81 / MADE made.py:21-27), without the
 * following Reverse: y[B][D], logdet[B]; bin_idx_dev [B][D][2] or NULL (IMADE only). */
int wf_layer_fwd(const wf_model* m, int layer, const float* u_in_dev, int64_t B, float* y_dev, float* logdet_dev,
                 int32_t* bin_idx_dev, void* stream);

/* Serial(...).inverse_fun (bijections.py:462-463): x[B][D] from u[B][D] through the reversed layer stack
 * (Reverse, IMADE.inverse_fun made.py:85-100 by bisection helpers.py:150-166 / MADE.inverse_fun made.py:29-37,
 * BoxTransformLayer.reverse_fun_* made.py:139-154,186-197).  exact == 0 reproduces the reference's IMADE.inverse_fun, which
 * evaluates the conditioner on its inputs (made.py:88) and is therefore not the inverse of direct_fun for columns > 0;
 * exact != 0 conditions on the reconstructed prefix (the true inverse). */
int wf_inverse_fwd(const wf_model* m, const float* u_dev, int64_t B, float* x_dev, int32_t exact, void* stream);

/* sample(rng, params, num_samples) of Waveflow (wavefunctions.py:74-107), MFlow (distributions.py:165-190) and Flow
 * (distributions.py:104-108): column-by-column rejection sampling of the prior, then wf_inverse_fwd.  x_dev[B][D];
 * latent_dev[B][D] or NULL (`return_original_samples`).  Counter-based Philox4x32-10 keyed by (seed, walker index):
 * results are reproducible for a given seed but do not follow JAX's threefry stream (parity unpinned).  The rejection loop of a
 * column is bounded (~1e5 proposals, where the reference's while_loop is not): a walker that exhausts it is written as NaN.
 * Three kernels share the work: one walker per wave (small and medium batches), one lane per walker (other large batches), and for
 * batches >= 16384 of two-particle models with <= 64 bases the staged form (wf_kernels_etile.hip: conditioners on the matrix cores, the
 * mesh searches one lane per walker; WF_SAMPLE_TILE_MIN moves the switch); the same holds for wf_inverse_fwd.  The kernels draw from the
 * same law with their own proposal sequences, so the walkers of a seed change across the switch points. */
int wf_sample(const wf_model* m, uint64_t seed, int64_t B, float* x_dev, float* latent_dev, int32_t exact, void* stream);

/* H psi = -1/2 laplacian(psi) + V psi of the Waveflow wavefunction: physics.construct_hamiltonian_function (utils/physics.py:79-93)
 * with physics.laplacian (:50-52, trace of jax.hessian -- the table lerp differentiates to the next cached derivative table,
 * isplines_jax.py:60-66) and the one-dimensional soft-Coulomb physics.get_potential (:60-76) for `n_protons` <= 8 protons at
 * `protons_host`.  hpsi_dev[B]; psi_dev[B] and laplacian_dev[B] may be NULL.  The local energy of vqmc.loss_fn_efficient
 * (vqmc.py:193-200) is hpsi / (psi + 1e-8).  Batches of >= 16384 walkers of two-particle models (<= 64 bases, mean-type box, ungated) run on the
 * matrix cores -- one kernel for the whole of H psi where the nets fit LDS together (the shipped shapes), else launch by launch in passes of 2^19
 * walkers (wf_kernels_etile.hip; WF_ENERGY_TILE_MIN moves the switch, WF_ENERGY_TILE_CHUNK the pass size) --,
 * everything else on the wave-cooperative kernel: same function.
 * Coverage: WF_PRIOR_WAVEFLOW models with IMADE layers, boundary constraints the spline tables carry (any value on the I layers and, since
 * round 3, on the B-spline prior), gated heads included;
 * D = 2..8 with <= 32 bases per dimension, D = 2..4 with 33..64. */
int wf_hamiltonian_fwd(const wf_model* m, const float* x_dev, int64_t B, const float* protons_host, int32_t n_protons,
                       float* hpsi_dev, float* psi_dev, float* laplacian_dev, void* stream);

/* Parameter gradient of psi and of its Laplacian -- the vector-Jacobian product behind vqmc.train_step_efficient
 * (vqmc.py:193-221: value_and_grad of loss_fn_efficient through physics.laplacian, utils/physics.py:50-52):
 *     grad_dev[p] = sum_b ( w_psi_dev[b] * d psi_b / d theta_p + w_lap_dev[b] * d laplacian(psi)_b / d theta_p )
 * in the flat leaf order of wf_model_set_params (masked-out weight entries and the zero_params leaves get 0, as in the
 * reference where they only enter multiplied by their mask).  The table lerp differentiates to the next cached table and
 * the order beyond the last cached one clamps to it (JAX's out-of-range index semantics at isplines_jax.py:65).
 * workspace: wf_psi_vjp_workspace_bytes(m, B) bytes suffice for any B (larger batches are processed in chunks of what the
 * workspace holds).  Same model coverage as wf_hamiltonian_fwd.  Batches >= 16384 (WF_GRAD_TILE_MIN, read per call) of ungated two-particle
 * models with <= 64 bases take the matrix-core path (wf_kernels_etile.hip: k_efused, k_ebwd per net; <= 32 bases since round 3, 33..64 since
 * round 4); everything else the reverse wave sweeps.  Bitwise reproducible on either path. */
int64_t wf_psi_vjp_workspace_bytes(const wf_model* m, int64_t B);
int wf_psi_vjp(const wf_model* m, const float* x_dev, int64_t B, const float* w_psi_dev, const float* w_lap_dev, float* grad_dev,
               void* workspace_dev, int64_t workspace_bytes, void* stream);

/* loss_fn_efficient and its gradient in one pass (vqmc.py:193-221): one forward sweep gives H psi and psi of every walker,
 * the tangent-rule weights of wf_vqmc_seeds follow on the device, the reverse sweep and the contraction give
 * grad_dev[n_params] (scaled by inv_count = 1 / global batch); e_loc_dev[B] = hpsi / (psi + 1e-8) for the caller's
 * batch statistics (wf_block_sums).  Workspace: wf_psi_vjp_workspace_bytes. */
int wf_vqmc_loss_grad(const wf_model* m, const float* x_dev, int64_t B, const float* protons_host, int32_t n_protons, float running_average,
                      float inv_count, float* e_loc_dev, float* grad_dev, void* workspace_dev, int64_t workspace_bytes, void* stream);

/* One whole VQMC training step without the host (the body of vqmc.ModelTrainer's loop, vqmc.py:102-117: sample -> loss and
 * gradient -> Adam -> parameters): every per-step scalar lives on the device, so the sequence of launches is identical from
 * step to step and can be captured once in a hipGraph and replayed.
 *   counter_dev          step index i of this step (the value passed to opt_update); seeds the sampler together with `seed`,
 *                        gives Adam's bias corrections, selects slot i mod ring_len of the loss ring; incremented at the end
 *   running_average_dev  the `running_average` of loss_fn_efficient (the caller refreshes it every 100 steps, vqmc.py:112-113)
 *   loss_ring_dev        [ring_len][3] doubles: [sum E_L, sum E_L^2, batch] of each step
 *   defer_eval_tables    0: every image of the model holds the updated parameters on exit.  1: the step refreshes the weight
 *                        images only and leaves out the composite tables that just the large-batch evaluation kernel reads
 *                        (13 % of a 128-walker step); the caller then calls wf_model_set_params_device(m, params_dev, ..)
 *                        before wf_logpdf_fwd / wf_psi_fwd / wf_flow_fwd / wf_layer_fwd -- further training steps, wf_sample,
 *                        wf_hamiltonian_fwd and the gradient entry points need nothing
 * The model's weight images must hold params_dev on entry (wf_model_set_params_device); they hold the updated parameters on
 * exit.  Single process; batch <= 131072 where the step samples with the wave kernel (beyond: WF_ERR_UNSUPPORTED, step from the host
 * with wf_sample, wf_vqmc_loss_grad, wf_adam_step), any batch where the staged large-batch sampler applies (see wf_sample; it and the
 * matrix-core gradient path read the tables a deferred step leaves stale: from 16384 walkers per step on -- wherever one of them applies at the
 * step's batch size -- the step refreshes every table whatever defer_eval_tables says, so that a hipGraph of it replays on fresh tables).
 * Workspace: wf_vqmc_train_step_workspace_bytes (by capability at that batch size, independent of transient state). */
typedef struct wf_train_state {
    float* params_dev;
    float* m_dev;
    float* v_dev;
    uint64_t* counter_dev;
    float* running_average_dev;
    double* loss_ring_dev;
    int32_t ring_len;
    int32_t defer_eval_tables;
} wf_train_state;
int64_t wf_vqmc_train_step_workspace_bytes(const wf_model* m, int64_t batch);
int wf_vqmc_train_step(wf_model* m, const wf_train_state* st, uint64_t seed, int64_t batch, const float* protons_host, int32_t n_protons,
                       float step_size, float b1, float b2, float eps, int32_t exact_sampler, void* workspace_dev, int64_t workspace_bytes,
                       void* stream);

/* The same step for walkers sharded over several GPUs (one process per GPU), in two halves around the caller's collective:
 *   wf_vqmc_train_step_local   this rank's walkers (sampler stream `seed`, advanced by the device counter) -> their contribution to
 *                              the gradient of loss_fn_efficient, the tangent rule scaled by inv_global_batch = 1 / (walkers of ALL
 *                              ranks) -> reduce_dev[n_params + 3] doubles = [gradient, sum E_L, sum E_L^2, local walkers]
 *   (caller)                   SUM all-reduce of reduce_dev over the ranks: the step's one collective (RCCL), on `stream`
 *   wf_vqmc_train_step_apply   Adam with the reduced gradient, image refill, reduced sums to the loss ring, counter + 1
 * Every rank ends the step with the same parameters.  Same workspace size as wf_vqmc_train_step (with the local batch); both halves
 * are asynchronous and hipGraph-capturable. */
int wf_vqmc_train_step_local(wf_model* m, const wf_train_state* st, uint64_t seed, int64_t batch_local, const float* protons_host, int32_t n_protons,
                             float inv_global_batch, int32_t exact_sampler, double* reduce_dev, void* workspace_dev, int64_t workspace_bytes,
                             void* stream);
int wf_vqmc_train_step_apply(wf_model* m, const wf_train_state* st, const double* reduce_dev, float step_size, float b1, float b2, float eps,
                             void* stream);

/* One epoch of benchmark_tests.train_model (benchmark_tests.py:98-101, 138-144) without the host: -mean log_pdf over the N
 * resident rows x_dev[N][D], its gradient, Adam, image refill; [sum log_pdf, sum log_pdf^2, N] of the epoch goes to the loss ring
 * (the loss is -sum / N).  Same state conventions as wf_vqmc_train_step (running_average_dev is unused).  hipGraph-capturable. */
int64_t wf_mle_train_step_workspace_bytes(const wf_model* m, int64_t N);
int wf_mle_train_step(wf_model* m, const wf_train_state* st, const float* x_dev, int64_t N, float step_size, float b1, float b2, float eps,
                      void* workspace_dev, int64_t workspace_bytes, void* stream);

/* Parameter gradient of the log-density: grad_dev[p] = sum_b w_dev[b] * d log_pdf_b / d theta_p for every model wf_logpdf_fwd
 * evaluates with boundary constraints the spline tables carry (any value), gated heads included, and <= 32 bases per dimension (or <= 64 for D <= 4) (IMADE or MADE layers; Waveflow, M-spline, Normal or Uniform
 * prior).  With w = -1/B this is the gradient of benchmark_tests.loss (benchmark_tests.py:84-87, 98-101); with per-walker
 * weights it is the jacrev(log_pdf) contraction of vqmc.train_step (vqmc.py:175-180). */
int64_t wf_logpdf_vjp_workspace_bytes(const wf_model* m, int64_t B);
int wf_logpdf_vjp(const wf_model* m, const float* x_dev, int64_t B, const float* w_dev, float* grad_dev, void* workspace_dev,
                  int64_t workspace_bytes, void* stream);

/* Maximum-likelihood value and gradient in one pass (benchmark_tests.loss + grad(loss), benchmark_tests.py:84-101): one forward
 * sweep gives logp_dev[B] = log_pdf of every row and the tape, the reverse sweep gives grad_dev = weight * sum_b d log_pdf_b / d theta
 * (weight = -1 / N for the mean negative log-likelihood).  Workspace: wf_logpdf_vjp_workspace_bytes. */
int wf_logpdf_loss_grad(const wf_model* m, const float* x_dev, int64_t B, float weight, float* logp_dev, float* grad_dev, void* workspace_dev,
                        int64_t workspace_bytes, void* stream);

/* Per-walker weights of loss_fn_efficient's tangent rule (vqmc.py:198-212) for wf_psi_vjp, from wf_hamiltonian_fwd's outputs:
 *     e_loc = hpsi / (psi + 1e-8);   d loss = [2 (e_loc - running_average)/psi - hpsi/psi^2] d psi + (1/psi) d hpsi,
 *     d hpsi = -1/2 d laplacian + V d psi   =>   w_psi = (... + V/psi) * inv_count,  w_lap = -1/(2 psi) * inv_count
 * with V the soft-Coulomb potential of wf_hamiltonian_fwd and inv_count = 1 / (global batch size). */
int wf_vqmc_seeds(const float* x_dev, int64_t B, int32_t n_dim, const float* protons_host, int32_t n_protons, const float* hpsi_dev,
                  const float* psi_dev, float running_average, float inv_count, float* e_loc_dev, float* w_psi_dev, float* w_lap_dev,
                  void* stream);

/* Rational-quadratic spline bijector, elementwise (flows/bijections/neural_splines.py:16-184; dead code in the reference,
 * parity unpinned).  x[N]; uw, uh [N][K] unnormalised widths / heights; ud [N][n_deriv] unnormalised derivatives with
 * n_deriv == K-1 (unconstrained_RQS: identity outside [left, right], boundary derivatives 1) or K+1 (RQS: explicit).
 * inverse != 0 applies the inverse map.  y[N], logabsdet[N]; bin_dev[N] (may be NULL) = selected bin, -1 in the tails.
 * min_bin_width = min_bin_height = min_derivative = 1e-3 as in the reference. */
int wf_rqs_fwd(const float* x_dev, const float* uw_dev, const float* uh_dev, const float* ud_dev, int64_t N, int32_t K,
               int32_t n_deriv, int32_t inverse, float left, float right, float bottom, float top, float* y_dev,
               float* logabsdet_dev, int32_t* bin_dev, void* stream);

/* NeuralSplineCoupling.direct_fun / inverse_fun (flows/bijections/neural_splines.py:244-300): x[B][dim] (dim even) ->
 * y[B][dim], logdet[B].  Two half-steps, each: FCNN (Dense(hidden) Tanh Dense(hidden) Tanh Dense((3K-1) dim/2), :187-188) on one
 * half -> per coordinate K widths and K heights (softmax * 2 tail_bound) and K-1 derivatives (softplus) -> unconstrained_RQS of
 * the other half with those values as its unnormalised parameters (the reference's double normalisation is kept).
 * params_dev: f1 then f2, each W1 [dim/2][hidden], b1, W2 [hidden][hidden], b2, W3 [hidden][(3K-1) dim/2], b3 (stax.Dense leaf
 * order).  The layer is dead code in the reference (only its invertibility is exercised, tests/test_bijections.py:138): parity
 * unpinned.  hidden <= 64, K <= 32. */
int64_t wf_nsc_workspace_bytes(int64_t B, int32_t dim, int32_t K);
int wf_nsc_fwd(const float* x_dev, int64_t B, int32_t dim, int32_t K, float tail_bound, int32_t hidden, const float* params_dev, int32_t inverse,
               float* y_dev, float* logdet_dev, void* workspace_dev, int64_t workspace_bytes, void* stream);

/* Local block sums for the VQMC expectation (vqmc.py:196: the batch mean is the only reduction over
 * walkers): out_dev[3] (fp64) = { sum v, sum v^2, count } over v[B]; deterministic (fixed-order) reduction.
 * The caller all-reduces these three doubles across ranks (RCCL) -- see waveflow_amd/distributed.py. */
int wf_block_sums(const float* v_dev, int64_t B, double* out_dev, void* workspace_dev, int64_t workspace_bytes,
                  void* stream);
int64_t wf_block_sums_workspace_bytes(int64_t B);

#ifdef __cplusplus
}
#endif
#endif /* WAVEFLOW_HIP_H */
