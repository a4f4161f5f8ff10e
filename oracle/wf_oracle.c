/*
 * wf_oracle.c -- CPU restatement of waveflow's flow-density hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under waveflow_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it (as the checker / the reported CPU baseline).
 *
 * It follows the reference line by line, in the reference's operation order,
 * in the reference's precision (tables fp64 -> stored fp32; evaluation fp32).
 * Compile with -ffp-contract=off so that no multiply-add is fused.
 *
 * Parity pin (see tests/test_oracle_*.py):
 *   - tables: bit-equal to the reference's own fixture tables
 *     waveflow/tests/splines/cached_bases/{I,B}/ (k=5, 16 knots) and to probes
 *     produced by importing the reference's splines_np (tests/golden/ref_probes.npz);
 *   - evaluation: the He checkpoint's psi grids shipped in
 *     data_submission_apl_ml/He_1d_L10box_batch256 (tests/golden/he_golden.npz).
 *   - MFlow/Flow log_pdf and RQS: the reference ships no parameters / no
 *     runnable caller for them => those entry points are "parity unpinned"
 *     (formula restatements, self-consistency tests only).
 *
 * Reference citations are "file:line" relative to /root/reference/waveflow.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Part 1: basis tables (fp64), splines/splines_np.py                  */
/* ------------------------------------------------------------------ */

/* numpy.linspace(0, 1, n)[i]: arange(n) * (1/(n-1)), last element forced to 1. */
static double linspace01(int i, int n) {
    if (i == n - 1) return 1.0;
    double step = 1.0 / (double)(n - 1);
    return (double)i * step + 0.0;
}

/* knots: isplines_jax.py:91-93 (rep = k+1), bsplines_jax.py:58-60 (rep = k+1),
 * msplines_jax.py:72-74 (rep = k): first and last internal knot repeated `rep` times. */
static int make_knots(int n_internal, int rep, double* t) {
    int n = 0;
    for (int r = 0; r < rep; ++r) t[n++] = linspace01(0, n_internal);
    for (int i = 1; i < n_internal - 1; ++i) t[n++] = linspace01(i, n_internal);
    for (int r = 0; r < rep; ++r) t[n++] = linspace01(n_internal - 1, n_internal);
    return n;
}

/* splines_np.py:42-62 */
static double M_np(double x, int k, int i, const double* t, int nt, int max_k, int nd) {
    if (k == 1) {
        if ((x >= t[i] && x < t[i + 1]) || (i >= nt - (max_k + 1) && x >= t[i] && x <= t[i + 1])) {
            if (t[i + 1] - t[i] == 0) return 0;
            if (nd == 0) return 1 / (t[i + 1] - t[i]);
            return 0;
        }
        return 0;
    }
    if (t[i + k] - t[i] == 0) return 0;
    if (nd == 0) {
        double a = (x - t[i]) * M_np(x, k - 1, i, t, nt, max_k, 0);
        double b = (t[i + k] - x) * M_np(x, k - 1, i + 1, t, nt, max_k, 0);
        return (double)k * (a + b) / ((double)(k - 1) * (t[i + k] - t[i]));
    } else if (nd == 1) {
        double pre = (double)k / ((double)(k - 1) * (t[i + k] - t[i]));
        double a = (x - t[i]) * M_np(x, k - 1, i, t, nt, max_k, nd);
        double b = (t[i + k] - x) * M_np(x, k - 1, i + 1, t, nt, max_k, nd);
        double c = M_np(x, k - 1, i, t, nt, max_k, 0);
        double d = M_np(x, k - 1, i + 1, t, nt, max_k, 0);
        return pre * (((a + b) + c) - d);
    } else {
        double pre = (double)k / ((double)(k - 1) * (t[i + k] - t[i]));
        double a = (x - t[i]) * M_np(x, k - 1, i, t, nt, max_k, nd);
        double b = (t[i + k] - x) * M_np(x, k - 1, i + 1, t, nt, max_k, nd);
        double c = M_np(x, k - 1, i, t, nt, max_k, nd - 1);
        double d = M_np(x, k - 1, i + 1, t, nt, max_k, nd - 1);
        return pre * ((a + b) + (double)nd * (c - d));
    }
}

/* numpy add.reduce over a contiguous 1-D double array (pairwise_sum in
 * numpy/core/src/umath/loops_utils.h.src): plain loop for n < 8, eight
 * accumulators for 8 <= n <= 128. */
static double np_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

/* np.searchsorted(t, x, 'left') */
static int searchsorted_left(const double* t, int nt, double x) {
    int lo = 0, hi = nt;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (t[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* splines_np.py:79-93 */
static double I_np(double x, int k, int i, const double* t, int nt, int max_k, int nd) {
    int j;
    if (x == 0.0) j = k; else j = searchsorted_left(t, nt, x) - 1;
    if (i > j || i == nt - (k + 1)) return 0;
    if (i <= j - k) return nd == 0 ? 1 : 0;
    double terms[64];
    int n = 0;
    for (int m = i; m <= j; ++m)
        terms[n++] = (t[m + k + 1] - t[m]) * M_np(x, k + 1, m, t, nt, max_k, nd) / (double)(k + 1);
    return np_sum(terms, n);
}

static double dB_np(double x, int k, int i, const double* t, int nt, int max_k, int nd);

/* splines_np.py:101-120 */
static double B_np(double x, int k, int i, const double* t, int nt, int max_k, int nd) {
    if (nd == 0) {
        if (k == 0) {
            if ((t[i] <= x && x < t[i + 1]) || (i >= nt - (max_k + 2) && x >= t[i] && x <= t[i + 1])) return 1.0;
            return 0.0;
        }
        double c1, c2;
        if (t[i + k] == t[i]) c1 = 0.0;
        else c1 = (x - t[i]) / (t[i + k] - t[i]) * B_np(x, k - 1, i, t, nt, max_k, 0);
        if (t[i + k + 1] == t[i + 1]) c2 = 0.0;
        else c2 = (t[i + k + 1] - x) / (t[i + k + 1] - t[i + 1]) * B_np(x, k - 1, i + 1, t, nt, max_k, 0);
        return c1 + c2;
    }
    return dB_np(x, k, i, t, nt, max_k, nd);
}

/* splines_np.py:127-137 */
static double dB_np(double x, int k, int i, const double* t, int nt, int max_k, int nd) {
    double c1, c2;
    if (t[i + k] - t[i] == 0) c1 = 0;
    else c1 = B_np(x, k - 1, i, t, nt, max_k, nd - 1) / (t[i + k] - t[i]);
    if (t[i + k + 1] - t[i + 1] == 0) c2 = 0;
    else c2 = B_np(x, k - 1, i + 1, t, nt, max_k, nd - 1) / (t[i + k + 1] - t[i + 1]);
    return (double)k * (c1 - c2);
}

/* kind: 0 = M (msplines_jax.py:84-105), 1 = I (isplines_jax.py:106-128),
 * 2 = plain B (bsplines_jax.py:88-95).  out: [n_bases][n_mesh] fp64.
 * Returns n_bases; with out == NULL only returns n_bases. */
int wfo_table(int kind, int k, int n_internal, int n_mesh, int nd, double* out) {
    double t[512];
    int nt, nb;
    if (kind == 0) { nt = make_knots(n_internal, k, t); nb = nt - k; }
    else if (kind == 1) { nt = make_knots(n_internal, k + 1, t); nb = nt - k; }
    else { nt = make_knots(n_internal, k + 1, t); nb = nt - k - 1; }
    if (!out) return nb;
    for (int i = 0; i < nb; ++i)
        for (int m = 0; m < n_mesh; ++m) {
            double x = linspace01(m, n_mesh);
            double v;
            if (kind == 0) v = M_np(x, k, i, t, nt, k, nd);
            else if (kind == 1) v = I_np(x, k, i, t, nt, k + 1, nd);
            else v = B_np(x, k, i, t, nt, k, nd);
            out[(size_t)i * n_mesh + m] = v;
        }
    return nb;
}

int wfo_knots(int kind, int k, int n_internal, double* t) {
    return make_knots(n_internal, kind == 0 ? k : k + 1, t);
}

/* ------------------------------------------------------------------ */
/* Part 2: evaluation (fp32; -DWFO_F64 builds the same code in fp64 = "exact arithmetic" yardstick) */
#ifdef WFO_F64
typedef double real;
#define R_FLOOR floor
#define R_CEIL ceil
#define R_TANH tanh
#define R_EXP exp
#define R_LOG log
#define R_SQRT sqrt
#else
typedef float real;
#define R_FLOOR floorf
#define R_CEIL ceilf
#define R_TANH tanhf
#define R_EXP expf
#define R_LOG logf
#define R_SQRT sqrtf
#endif
/* constants keep their fp32 VALUES in both builds (JAX weak-typed Python scalars become fp32) */
#define C(x) ((real)(x##f))

/* ------------------------------------------------------------------ */

#define WFO_MAX_D 16
#define WFO_MAX_NB 96
#define WFO_MAX_H 64
#define WFO_MAX_BC 4

typedef struct {
    int n;                      /* number of {n_derivative: value} entries, insertion order */
    int nd[WFO_MAX_BC];
    float val[WFO_MAX_BC];
} wfo_bc;

typedef struct {
    int k, nb, n_mesh;
    const float* tab;           /* [4][nb][n_mesh] fp32 (jnp.array(np.load(...)) => fp32) */
    wfo_bc left, right;
} wfo_spline;

typedef struct {
    int D;
    int hidden;                 /* 64: model_factory.py:72 */
    int n_layers;               /* flow layers (each followed by Reverse) */
    int layer_kind;             /* 0 = IMADE (made.py:44-105), 1 = MADE affine (made.py:7-41) */
    int box_kind;               /* 0 = none, 1 = 'mean' (made.py:156-183), 2 = 'first' (made.py:118-137) */
    float box_L;
    float i_reg;                /* spline_regularization, made.py:68 */
    wfo_spline isp;             /* I-spline of the IMADE layers */
    int prior_kind;             /* 0 = Waveflow B^2 (wavefunctions.py:33-71), 1 = MFlow M (distributions.py:139-163),
                                   2 = Uniform with prior_support (0,1) (distributions.py:26-41, benchmark_tests.py:59-63),
                                   3 = Normal(offset), no support clip (distributions.py:8-23) */
    float normal_offset;
    wfo_spline psp;             /* prior spline: orthogonal-B table (kind 0) or M table (kind 1) */
    const float* psp_plain;     /* kind 0: plain-B table [4][nb][n_mesh] used by enforce_boundary_conditions */
    const float* ob_to_b;       /* kind 0: [nb][nb] fp32, bsplines_jax.py:134 */
    int n_constr_left;          /* constrained_dimension_indices_left, model_factory.py:124-129 */
    int constr_left[WFO_MAX_D];
    float reverse_tol;          /* IMADE reverse_fun_tol (made.py:44, isplines_jax.py:153-156) */
    const float* b_to_ob;       /* kind 0 prior: [nb][nb], used by the rejection sampler's bound (bsplines_jax.py:164-166) */
    int i_gate, p_gate;         /* set_nn_output_grad_to_zero of the layers' / the prior's conditioner (model_factory.py:64-67) */
} wfo_model;

/* X_cached, isplines_jax.py:45-56 / msplines_jax.py:30-41 / bsplines_jax.py:19-30.
 * jnp indexing semantics for the two gathers: negative indices wrap once,
 * then out-of-bounds indices clamp.  idx (optional) receives (x_l, x_r) as computed. */
static inline int wrap_clamp(int i, int n) {
    if (i < 0) i += n;
    if (i < 0) i = 0;
    if (i > n - 1) i = n - 1;
    return i;
}

static inline real x_cached(const wfo_spline* s, int nd, int i, real x, int* idx) {
    int n_points = s->n_mesh - 1;
    real xs = x * (real)n_points;
    int x_l = (int)R_FLOOR(xs);
    int x_r = (int)R_CEIL(xs);
    if (idx) { idx[0] = x_l; idx[1] = x_r; }
    const float* row = s->tab + ((size_t)nd * s->nb + i) * s->n_mesh;
    real y_l = row[wrap_clamp(x_l, s->n_mesh)];
    real y_r = row[wrap_clamp(x_r, s->n_mesh)];
    real dx = x - (real)x_l / (real)n_points;
    real slope = (y_r - y_l) * (real)n_points;
    return y_l + slope * dx;
}

/* ispline/mspline/bspline with zero_border=False: sum(c[i] * X_cached(x, i)), isplines_jax.py:78 */
static real spline_apply(const wfo_spline* s, int nd, const real* c, real x, int* idx) {
    real acc = C(0.0);
    for (int i = 0; i < s->nb; ++i) acc = acc + c[i] * x_cached(s, nd, i, x, i == 0 ? idx : 0);
    return acc;
}

static real tab_at(const float* tab, const wfo_spline* s, int nd, int i, int m) {
    return tab[((size_t)nd * s->nb + i) * s->n_mesh + m];
}

/* enforce_boundary_conditions: isplines_jax.py:158-194 (kind 1), bsplines_jax.py:173-199 (kind 2),
 * msplines_jax.py:156-184 (kind 0).  tab = the table the reference evaluates at 0.0 / 1.0
 * (plain-B table for kind 2).  X_cached(0.0, j) == T[j][0], X_cached(1.0, j) == T[j][n_mesh-1]. */
static void enforce_bc(const wfo_spline* s, const float* tab, int kind, real* w) {
    int nb = s->nb, last = s->n_mesh - 1;
    for (int p = 0; p < s->left.n; ++p) {
        int nd = s->left.nd[p];
        real sum = C(0.0);
        for (int j = 0; j < nd; ++j) sum = sum + tab_at(tab, s, nd, j, 0) * w[j];
        real value = tab_at(tab, s, nd, nd, 0);
        w[nd] = (s->left.val[p] - sum) / value;
    }
    for (int p = 0; p < s->right.n; ++p) {
        int nd = s->right.nd[p];
        if (kind == 1 && nd == 0) { w[nb - nd - 1] = C(0.0); continue; }   /* isplines_jax.py:174-176 */
        real sum = C(0.0);
        for (int j = 0; j < nd; ++j) sum = sum + tab_at(tab, s, nd, nb - j - 1, last) * w[nb - 1 - j];
        real value = tab_at(tab, s, nd, nb - nd - 1, last);
        w[nb - nd - 1] = (s->right.val[p] - sum) / value;
    }
    if (kind == 2) {
        real ss = C(0.0);
        for (int j = 0; j < nb; ++j) ss = ss + w[j] * w[j];
        real nrm = R_SQRT(ss);
        for (int j = 0; j < nb; ++j) w[j] = w[j] / nrm;
    } else {
        real ss = C(0.0);
        for (int j = 0; j < nb; ++j) ss = ss + w[j];
        for (int j = 0; j < nb; ++j) w[j] = w[j] / ss;
    }
}

/* remove_bias: isplines_jax.py:196-202 (kind 1), msplines_jax.py:186-192 (kind 0) */
static void remove_bias(int kind, int k, int nb, real* p) {
    for (int i = 0; i < k; ++i) {
        int a = kind == 1 ? i + 1 : i;
        int b = kind == 1 ? nb - (i + 2) : nb - (i + 1);
        p[a] = p[a] * (real)(i + 1) / (real)k;
        p[b] = p[b] * (real)(i + 1) / (real)k;
    }
    real ss = C(0.0);
    for (int j = 0; j < nb; ++j) ss = ss + p[j];
    for (int j = 0; j < nb; ++j) p[j] = p[j] / ss;
}

/* get_masks, model_factory.py:8-19: degrees in = arange(D); hidden = arange(H) % (D-1); out = arange(D) - 1 */
static inline int deg_in(int a) { return a; }
static inline int deg_hid(int a, int D) { return a % (D - 1); }
static inline int deg_out(int a) { return a - 1; }

/* MaskedDense stack, model_factory.py:21-35,72-82: returns raw net output o[n_out*D] (column c = j*D + d).
 * params: W0[D][H], b0[H], W1[H][H], b1[H], W2[H][n_out*D], b2[n_out*D]. */
static const float* conditioner(const float* p, int D, int H, int n_out, const real* x, real* o) {
    const float *W0 = p, *b0 = W0 + D * H, *W1 = b0 + H, *b1 = W1 + H * H, *W2 = b1 + H, *b2 = W2 + (size_t)H * n_out * D;
    real h1[WFO_MAX_H], h2[WFO_MAX_H];
    for (int j = 0; j < H; ++j) {
        real acc = C(0.0);
        for (int a = 0; a < D; ++a) {
            real m = deg_hid(j, D) >= deg_in(a) ? C(1.0) : C(0.0);
            acc = acc + x[a] * (W0[a * H + j] * m);
        }
        h1[j] = R_TANH(acc + b0[j]);
    }
    for (int j = 0; j < H; ++j) {
        real acc = C(0.0);
        for (int a = 0; a < H; ++a) {
            real m = deg_hid(j, D) >= deg_hid(a, D) ? C(1.0) : C(0.0);
            acc = acc + h1[a] * (W1[a * H + j] * m);
        }
        h2[j] = R_TANH(acc + b1[j]);
    }
    int NO = n_out * D;
    for (int c = 0; c < NO; ++c) {
        int d = c % D;                                   /* jnp.tile(masks[-1], output_shape) */
        real acc = C(0.0);
        for (int a = 0; a < H; ++a) {
            real m = deg_out(d) >= deg_hid(a, D) ? C(1.0) : C(0.0);
            acc = acc + h2[a] * (W2[(size_t)a * NO + c] * m);
        }
        o[c] = acc + b2[c];
    }
    return b2 + NO;
}

/* Conditioning probes (test infrastructure only).  When set, for the walker being evaluated:
 *   g_cond   the smallest spline derivative dy of any layer / dimension and the smallest per-dimension prior factor (psi_d^2 or the
 *            M-spline density): log_pdf sums log(. + 1e-7) of these;
 *   g_cond2  the smallest |sum_j o_j| / sum_j |o_j| of the Waveflow prior head, whose raw (signed) outputs the reference divides by
 *            their sum (model_factory.py:69): in fp32 that quotient loses all accuracy as the sum passes through zero.
 * A relative tolerance on log_pdf is meaningful only where both stay away from 0. */
static __thread real* g_cond = 0;
static __thread real* g_cond2 = 0;
static inline void cond_note(real v) { if (g_cond && v < *g_cond) *g_cond = v; }

/* calculate_bijection_params, model_factory.py:56-70:
 * bij[d][j] = o[j*D + d]; optional sigmoid (then zero_params -> |zero_params|); with set_nn_output_grad_to_zero (`gate`)
 * bij = cubed_input_product * bij + zero_params, cubed_input_product[d] = prod_{i<d} x_i^3 (roll of the cumprod, entry 0 set to 1);
 * bij /= bij.sum(-1).  Returns pointer past (net, zero_params). */
static const float* bijection_params(const float* p, int D, int H, int nb, int allow_negative, int gate, const real* x,
                                     real* bij /* [D][nb] */) {
    real o[WFO_MAX_D * WFO_MAX_NB];
    const float* next = conditioner(p, D, H, nb, x, o);   /* next: zero_params[D][nb] */
    real g = C(1.0);
    for (int d = 0; d < D; ++d) {
        real ss = C(0.0), sa = C(0.0);
        if (d > 0) g = g * (x[d - 1] * x[d - 1] * x[d - 1]);
        for (int j = 0; j < nb; ++j) {
            real v = o[j * D + d];
            if (!allow_negative) v = C(1.0) / (C(1.0) + R_EXP(-v));     /* jax.nn.sigmoid */
            if (gate) {
                real z = (real)next[d * nb + j];
                if (!allow_negative && z < 0) z = -z;
                v = g * v + z;
            }
            bij[d * nb + j] = v;
            ss = ss + v;
            sa = sa + (v < 0 ? -v : v);
        }
        if (g_cond2 && allow_negative) { real q = (ss < 0 ? -ss : ss) / sa; if (q < *g_cond2) *g_cond2 = q; }
        for (int j = 0; j < nb; ++j) bij[d * nb + j] = bij[d * nb + j] / ss;
    }
    return next + D * nb;  /* skip zero_params[D][nb] */
}

/* IMADE.direct_fun, made.py:66-81.  idx (optional): [D][2] bin indices of this layer. */
static const float* imade_direct(const wfo_model* m, const float* p, const real* x, real* y, real* logdet, int* idx) {
    int D = m->D, nb = m->isp.nb;
    real bij[WFO_MAX_D * WFO_MAX_NB];
    const float* next = bijection_params(p, D, m->hidden, nb, 0, m->i_gate, x, bij);
    real ld = C(0.0);
    for (int d = 0; d < D; ++d) {
        real* w = bij + d * nb;
        for (int j = 0; j < nb; ++j) w[j] = w[j] + m->i_reg;
        remove_bias(1, m->isp.k, nb, w);
        enforce_bc(&m->isp, m->isp.tab, 1, w);
        y[d] = spline_apply(&m->isp, 0, w, x[d], idx ? idx + 2 * d : 0);
        real dy = spline_apply(&m->isp, 1, w, x[d], 0);  /* grad via defjvp -> table nd+1, isplines_jax.py:60-66 */
        cond_note(dy);
        ld = ld + R_LOG(dy + C(1e-7));
    }
    *logdet = ld;
    return next;
}

/* MADE.direct_fun, made.py:21-27 with simple_masked_transform (model_factory.py:37-51, output_shape=2) */
static const float* made_direct(const wfo_model* m, const float* p, const real* x, real* y, real* logdet) {
    int D = m->D;
    real o[2 * WFO_MAX_D];
    const float* next = conditioner(p, D, m->hidden, 2, x, o);
    real ls = C(0.0);
    for (int d = 0; d < D; ++d) {
        real lw = o[d], bias = o[D + d];              /* jnp.split(..., 2, axis=1) */
        y[d] = (x[d] - bias) * R_EXP(-lw);
        ls = ls + lw;
    }
    *logdet = -ls;
    return next;
}

/* BoxTransformLayer, made.py:118-137 ('first') and :156-183 ('mean') */
static void box_direct(const wfo_model* m, const real* x, real* u, real* logdet) {
    int D = m->D;
    real L = m->box_L, tol = C(1e-7);
    if (m->box_kind == 1) {
        real s = C(0.0);
        for (int d = 0; d < D; ++d) s = s + x[d];
        real mean = s / (real)D;
        real l = mean - x[0];
        real w = x[D - 1] - x[0];
        real space_left = 2 * L;
        real ld = C(0.0);
        for (int i = 0; i < D - 1; ++i) {
            real diff = x[i + 1] - x[i];
            u[i] = diff / (space_left + tol);
            ld = ld - R_LOG(space_left + tol);
            space_left = space_left - diff;
        }
        u[D - 1] = (mean + L - l) / (2 * L - w + tol);
        ld = ld - R_LOG(2 * L - w + tol);
        *logdet = ld;
    } else {
        u[0] = (x[0] + L) / (2 * L);
        real ls = C(0.0);
        for (int i = 1; i < D; ++i) u[i] = (x[i] - x[i - 1]) / (L - x[i - 1] + tol);
        for (int i = 0; i < D - 1; ++i) ls = ls + R_LOG(L - x[i] + tol);
        *logdet = -R_LOG(2 * L) - ls;
    }
}

/* Serial.feed_forward (bijections.py:452-457) over [Box], (layer, Reverse) * n_layers.
 * Returns pointer to the prior net's params.  idx: [n_layers][D][2] or NULL. */
static const float* flow_direct(const wfo_model* m, const float* params, const real* x, real* u, real* logdet, int* idx) {
    int D = m->D;
    real cur[WFO_MAX_D], nxt[WFO_MAX_D];
    real ld_total = C(0.0), ld;
    if (m->box_kind) { box_direct(m, x, cur, &ld); ld_total = ld_total + ld; }
    else memcpy(cur, x, sizeof(real) * D);
    const float* p = params;
    for (int l = 0; l < m->n_layers; ++l) {
        if (m->layer_kind == 0) p = imade_direct(m, p, cur, nxt, &ld, idx ? idx + (size_t)l * D * 2 : 0);
        else p = made_direct(m, p, cur, nxt, &ld);
        ld_total = ld_total + ld;
        for (int d = 0; d < D; ++d) cur[d] = nxt[D - 1 - d];      /* Reverse, bijections.py:337-340 */
    }
    memcpy(u, cur, sizeof(real) * D);
    *logdet = ld_total;
    return p;
}

static inline real clip01(real v) { return v < C(0.0) ? C(0.0) : (v > C(1.0) ? C(1.0) : v); }

/* One walker: mode 0 = log_pdf, 1 = psi (Waveflow only). */
static real eval_one(const wfo_model* m, const float* params, const float* xin, int mode, float* u_out, int* idx) {
    int D = m->D;
    real u[WFO_MAX_D], logdet;
    real x[WFO_MAX_D];
    for (int d = 0; d < D; ++d) x[d] = (real)xin[d];
    const float* pp = flow_direct(m, params, x, u, &logdet, idx);
    real result;
    if (m->prior_kind == 0) {
        /* wavefunctions.py:33-71 */
        int nb = m->psp.nb;
        real bij[WFO_MAX_D * WFO_MAX_NB];
        bijection_params(pp, D, m->hidden, nb, 1, m->p_gate, u, bij);
        real lp = C(0.0), prod = C(1.0);
        int* pidx = idx ? idx + (size_t)m->n_layers * D * 2 : 0;
        for (int d = 0; d < D; ++d) {
            real* w = bij + d * nb;
            enforce_bc(&m->psp, m->psp_plain, 2, w);
            u[d] = clip01(u[d]);
            /* BSpline_fun.apply_fun, bsplines_jax.py:127-137 */
            real c[WFO_MAX_NB];
            real ss = C(0.0);
            for (int j = 0; j < nb; ++j) {
                real acc = C(0.0);
                for (int a = 0; a < nb; ++a) acc = acc + w[a] * m->ob_to_b[a * nb + j];
                c[j] = acc;
                ss = ss + acc * acc;
            }
            real nrm = R_SQRT(ss);
            for (int j = 0; j < nb; ++j) c[j] = c[j] / nrm;
            real v = spline_apply(&m->psp, 0, c, u[d], pidx ? pidx + 2 * d : 0);
            cond_note(v * v);
            int constrained = 0;
            for (int q = 0; q < m->n_constr_left; ++q) if (m->constr_left[q] == d) constrained = 1;
            if (mode == 0) {
                real pr = v * v;
                if (constrained) pr = pr / 2;
                lp = lp + R_LOG(pr + C(1e-7));
            } else {
                if (constrained) v = v / R_SQRT(C(2.0));
                prod = prod * v;
            }
        }
        result = mode == 0 ? lp + logdet : prod * R_EXP(C(0.5) * logdet);
    } else if (m->prior_kind == 1) {
        /* distributions.py:139-163 */
        int nb = m->psp.nb;
        real bij[WFO_MAX_D * WFO_MAX_NB];
        bijection_params(pp, D, m->hidden, nb, 0, m->p_gate, u, bij);
        real lp = C(0.0);
        int* pidx = idx ? idx + (size_t)m->n_layers * D * 2 : 0;
        for (int d = 0; d < D; ++d) {
            real* w = bij + d * nb;
            remove_bias(0, m->psp.k, nb, w);
            enforce_bc(&m->psp, m->psp.tab, 0, w);
            u[d] = clip01(u[d]);
            real v = spline_apply(&m->psp, 0, w, u[d], pidx ? pidx + 2 * d : 0);
            cond_note(v);
            lp = lp + R_LOG(v + C(1e-7));
        }
        result = lp + logdet;
    } else if (m->prior_kind == 2) {
        /* Flow.log_pdf distributions.py:95-102 with Uniform + prior_support=(0,1): clip => logpdf 0 */
        for (int d = 0; d < D; ++d) u[d] = clip01(u[d]);
        result = C(0.0) + logdet;
    } else {
        /* Normal(offset): norm.logpdf(u + offset).sum(1), distributions.py:14-15 */
        real lp = C(0.0);
        for (int d = 0; d < D; ++d) {
            /* jax.scipy.stats.norm.logpdf: (log(2*pi*scale^2) + ((x-loc)/scale)^2) / -2 */
            real z = u[d] + m->normal_offset;
            lp = lp + (C(1.8378770664093453) + z * z) / -C(2.0);
        }
        result = lp + logdet;
    }
    if (u_out) for (int d = 0; d < D; ++d) u_out[d] = (float)u[d];
    return result;
}

/* mode 0: log_pdf, 1: psi.  idx_out: [B][n_layers+1][D][2] int32 or NULL.  threads <= 1: serial. */
int wfo_eval(const wfo_model* m, const float* params, const float* x, int64_t B, int mode, float* out, float* u_out,
             int32_t* idx_out, int threads) {
    if (m->D > WFO_MAX_D || m->D < 2 || m->hidden > WFO_MAX_H) return -1;
    if (m->layer_kind == 0 && m->isp.nb > WFO_MAX_NB) return -1;
    if (mode == 1 && m->prior_kind != 0) return -2;
    int D = m->D;
    size_t istride = (size_t)(m->n_layers + 1) * D * 2;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
    for (int64_t b = 0; b < B; ++b) {
        int idx_local[(8 + 1) * WFO_MAX_D * 2];
        int* idx = 0;
        if (idx_out && m->n_layers <= 8) { idx = idx_local; memset(idx, 0, sizeof(idx_local)); }
        out[b] = (float)eval_one(m, params, x + b * D, mode, u_out ? u_out + b * D : 0, idx);
        if (idx) for (size_t q = 0; q < istride; ++q) idx_out[b * istride + q] = idx[q];
    }
    return 0;
}

/* wfo_eval plus cond_out[B]: the conditioning probe of each walker (see g_cond). */
int wfo_eval_cond(const wfo_model* m, const float* params, const float* x, int64_t B, int mode, float* out, float* cond_out,
                  float* cond2_out, int threads) {
    if (m->D > WFO_MAX_D || m->D < 2 || m->hidden > WFO_MAX_H) return -1;
    if (m->layer_kind == 0 && m->isp.nb > WFO_MAX_NB) return -1;
    if (mode == 1 && m->prior_kind != 0) return -2;
    int D = m->D;
#pragma omp parallel for schedule(static) num_threads(threads > 1 ? threads : 1)
    for (int64_t b = 0; b < B; ++b) {
        real c = C(1e30), c2 = C(1.0);
        g_cond = &c;
        g_cond2 = &c2;
        out[b] = (float)eval_one(m, params, x + b * D, mode, 0, 0);
        g_cond = 0;
        g_cond2 = 0;
        cond_out[b] = (float)c;
        cond2_out[b] = (float)c2;
    }
    return 0;
}

/* One IMADE layer, direct: (y[B][D], logdet[B]) from u[B][D]; layer params start at `params`. */
int wfo_imade_direct(const wfo_model* m, const float* params, const float* u, int64_t B, float* y, float* logdet,
                     int32_t* idx_out) {
    int D = m->D;
    for (int64_t b = 0; b < B; ++b) {
        int idx[WFO_MAX_D * 2];
        real xin[WFO_MAX_D], yo[WFO_MAX_D], ld;
        for (int d = 0; d < D; ++d) xin[d] = (real)u[b * D + d];
        imade_direct(m, params, xin, yo, &ld, idx);
        for (int d = 0; d < D; ++d) y[b * D + d] = (float)yo[d];
        logdet[b] = (float)ld;
        if (idx_out) for (int q = 0; q < 2 * D; ++q) idx_out[b * 2 * D + q] = idx[q];
    }
    return 0;
}

/* Serial direct only (no prior): u[B][D], logdet[B]. */
int wfo_flow_direct(const wfo_model* m, const float* params, const float* x, int64_t B, float* u, float* logdet) {
    int D = m->D;
    for (int64_t b = 0; b < B; ++b) {
        real xin[WFO_MAX_D], uo[WFO_MAX_D], ld;
        for (int d = 0; d < D; ++d) xin[d] = (real)x[b * D + d];
        flow_direct(m, params, xin, uo, &ld, 0);
        for (int d = 0; d < D; ++d) u[b * D + d] = (float)uo[d];
        logdet[b] = (float)ld;
    }
    return 0;
}


/* ------------------------------------------------------------------ */
/* Part 2b: inverse direction                      */
/* ------------------------------------------------------------------ */

/* helpers.binary_search (utils/helpers.py:150-166) on func(x) = spline(w, x) - y over [0, 1] */
static real ispline_reverse(const wfo_model* m, const real* w, real y, real tol) {
    real low = C(0.0), high = C(1.0);
    for (;;) {
        real mid = C(0.5) * (low + high);
        if (!((low + tol / 2 < mid) && (mid < high - tol / 2))) break;
        real f = spline_apply(&m->isp, 0, w, mid, 0) - y;
        if (f > 0) high = mid; else low = mid;
    }
    return low;
}

/* IMADE.inverse_fun (made.py:85-100).  NOTE the reference evaluates the conditioner on `inputs` (the values being
 * inverted), not on the partially reconstructed outputs (contrast MADE.inverse_fun, made.py:29-37): reproduced. */
static const float* imade_inverse(const wfo_model* m, const float* p, const real* in, real* out, int exact) {
    int D = m->D, nb = m->isp.nb;
    real bij[WFO_MAX_D * WFO_MAX_NB];
    const float* next = p;
    for (int d = 0; d < D; ++d) out[d] = C(0.0);
    for (int d = 0; d < D; ++d) {
        /* exact != 0: the true autoregressive inverse (conditioner on the reconstructed prefix) */
        if (d == 0 || exact) next = bijection_params(p, D, m->hidden, nb, 0, m->i_gate, exact ? out : in, bij);
        real* w = bij + d * nb;
        for (int j = 0; j < nb; ++j) w[j] = w[j] + m->i_reg;
        remove_bias(1, m->isp.k, nb, w);
        enforce_bc(&m->isp, m->isp.tab, 1, w);
        out[d] = ispline_reverse(m, w, in[d], (real)m->reverse_tol);
    }
    return next;
}

/* MADE.inverse_fun (made.py:29-37): sequential, conditioner on the partially reconstructed outputs */
static const float* made_inverse(const wfo_model* m, const float* p, const real* in, real* out) {
    int D = m->D;
    const float* next = p;
    for (int d = 0; d < D; ++d) out[d] = C(0.0);
    for (int c = 0; c < D; ++c) {
        real o[2 * WFO_MAX_D];
        next = conditioner(p, D, m->hidden, 2, out, o);
        out[c] = in[c] * R_EXP(o[c]) + o[D + c];
    }
    return next;
}

/* BoxTransformLayer.reverse_fun_mean (made.py:186-197) / reverse_fun_first (made.py:139-154) */
static void box_reverse(const wfo_model* m, const real* u, real* x) {
    int D = m->D;
    real L = m->box_L;
    if (m->box_kind == 1) {
        real out[WFO_MAX_D];
        real c = C(0.0), s = C(0.0);
        out[0] = C(0.0);
        for (int i = 0; i < D - 1; ++i) { c = c + u[i]; out[i + 1] = c; }
        for (int i = 0; i < D; ++i) s = s + out[i];
        real mean = s / (real)D;
        real w = out[D - 1];
        real pm = u[D - 1] * (1 - w) - (C(0.5) - mean);
        for (int i = 0; i < D; ++i) x[i] = (out[i] - mean + pm) * 2 * L;
    } else {
        x[0] = (u[0] - C(0.5)) * 2 * L;
        for (int i = 1; i < D; ++i) x[i] = u[i] * (L - x[i - 1]) + x[i - 1];
    }
}

/* Serial.inverse_fun (bijections.py:462-463): layers in reverse order.  layer_off[l] = offset of layer l's params. */
int wfo_inverse(const wfo_model* m, const float* params, const float* u, int64_t B, float* x_out, int exact) {
    int D = m->D;
    /* parameter offset of each flow layer */
    int64_t per_layer;
    {
        int H = m->hidden, n_out = m->layer_kind == 0 ? m->isp.nb : 2;
        per_layer = (int64_t)D * H + H + (int64_t)H * H + H + (int64_t)H * n_out * D + (int64_t)n_out * D + (m->layer_kind == 0 ? (int64_t)D * n_out : 0);
    }
    for (int64_t b = 0; b < B; ++b) {
        real cur[WFO_MAX_D], nxt[WFO_MAX_D];
        for (int d = 0; d < D; ++d) cur[d] = (real)u[b * D + d];
        for (int l = m->n_layers - 1; l >= 0; --l) {
            for (int d = 0; d < D; ++d) nxt[d] = cur[D - 1 - d];          /* Reverse.inverse_fun */
            const float* p = params + per_layer * l;
            if (m->layer_kind == 0) imade_inverse(m, p, nxt, cur, exact); else made_inverse(m, p, nxt, cur);
        }
        if (m->box_kind) { box_reverse(m, cur, nxt); for (int d = 0; d < D; ++d) cur[d] = nxt[d]; }
        for (int d = 0; d < D; ++d) x_out[b * D + d] = (float)cur[d];
    }
    return 0;
}

/* The density the reference's rejection sampler draws column `col` from, given the already drawn columns
 * (wavefunctions.py:89-104 / distributions.py:170-186): value at x, and the sampler's upper bound ymax.
 * outputs: [D] with zeros in the columns not drawn yet. */
int wfo_prior_column_density(const wfo_model* m, const float* params, const float* outputs, int col, const float* xs, int n,
                             float* dens, float* ymax_out) {
    int D = m->D, nb = m->psp.nb, H = m->hidden;
    int n_out_l = m->layer_kind == 0 ? m->isp.nb : 2;
    int64_t per_layer = (int64_t)D * H + H + (int64_t)H * H + H + (int64_t)H * n_out_l * D + (int64_t)n_out_l * D + (m->layer_kind == 0 ? (int64_t)D * n_out_l : 0);
    const float* pp = params + per_layer * m->n_layers;
    real o[WFO_MAX_D], bij[WFO_MAX_D * WFO_MAX_NB];
    for (int d = 0; d < D; ++d) o[d] = (real)outputs[d];
    if (m->prior_kind == 0) {
        bijection_params(pp, D, H, nb, 1, m->p_gate, o, bij);
        real* w = bij + col * nb;
        enforce_bc(&m->psp, m->psp_plain, 2, w);
        real c[WFO_MAX_NB], ss = C(0.0);
        for (int j = 0; j < nb; ++j) {
            real acc = C(0.0);
            for (int a = 0; a < nb; ++a) acc = acc + w[a] * m->ob_to_b[a * nb + j];
            c[j] = acc; ss = ss + acc * acc;
        }
        real nrm = R_SQRT(ss);
        for (int j = 0; j < nb; ++j) c[j] = c[j] / nrm;
        real ymax = C(0.0);
        for (int j = 0; j < nb; ++j) {
            real acc = C(0.0);
            for (int a = 0; a < nb; ++a) acc = acc + c[a] * m->b_to_ob[a * nb + j];
            if (acc * acc > ymax) ymax = acc * acc;
        }
        *ymax_out = (float)ymax;
        for (int i = 0; i < n; ++i) { real v = spline_apply(&m->psp, 0, c, (real)xs[i], 0); dens[i] = (float)(v * v); }
    } else if (m->prior_kind == 1) {
        bijection_params(pp, D, H, nb, 0, m->p_gate, o, bij);
        real* w = bij + col * nb;
        remove_bias(0, m->psp.k, nb, w);
        enforce_bc(&m->psp, m->psp.tab, 0, w);
        real mx = w[0];
        for (int j = 1; j < nb; ++j) if (w[j] > mx) mx = w[j];
        *ymax_out = (float)(mx * (real)(nb + m->psp.k));   /* params.max() * n_knots, msplines_jax.py:147-150 */
        for (int i = 0; i < n; ++i) dens[i] = (float)spline_apply(&m->psp, 0, w, (real)xs[i], 0);
    } else return -1;
    return 0;
}

/* ------------------------------------------------------------------ */
/* Part 3: rational-quadratic spline (neural_splines.py:74-184).       */
/* PARITY UNPINNED: dead code in the reference (removed jax.ops API).  */
/* ------------------------------------------------------------------ */
#define WFO_RQS_MAXK 64
/* x in [left,right]; uw, uh: [K]; ud: [K+1] unnormalised derivatives (already padded). */
int wfo_rqs(float x, const float* uw, const float* uh, const float* ud, int K, int inverse, float left, float right,
            float bottom, float top, float* out, float* logabsdet, int* bin) {
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    float cw[WFO_RQS_MAXK + 1], ch[WFO_RQS_MAXK + 1], w[WFO_RQS_MAXK], h[WFO_RQS_MAXK], der[WFO_RQS_MAXK + 1];
    if (K > WFO_RQS_MAXK) return -1;
    /* softmax */
    float mx = uw[0];
    for (int i = 1; i < K; ++i) mx = uw[i] > mx ? uw[i] : mx;
    float s = 0.0f;
    for (int i = 0; i < K; ++i) { w[i] = expf(uw[i] - mx); s = s + w[i]; }
    cw[0] = 0.0f;
    float c = 0.0f;
    for (int i = 0; i < K; ++i) { w[i] = min_w + (1 - min_w * K) * (w[i] / s); c = c + w[i]; cw[i + 1] = c; }
    for (int i = 0; i <= K; ++i) cw[i] = (right - left) * cw[i] + left;
    cw[0] = left; cw[K] = right;
    for (int i = 0; i < K; ++i) w[i] = cw[i + 1] - cw[i];
    for (int i = 0; i <= K; ++i) der[i] = min_d + (ud[i] > 20.0f ? ud[i] : log1pf(expf(ud[i])));
    mx = uh[0];
    for (int i = 1; i < K; ++i) mx = uh[i] > mx ? uh[i] : mx;
    s = 0.0f;
    for (int i = 0; i < K; ++i) { h[i] = expf(uh[i] - mx); s = s + h[i]; }
    ch[0] = 0.0f; c = 0.0f;
    for (int i = 0; i < K; ++i) { h[i] = min_h + (1 - min_h * K) * (h[i] / s); c = c + h[i]; ch[i + 1] = c; }
    for (int i = 0; i <= K; ++i) ch[i] = (top - bottom) * ch[i] + bottom;
    ch[0] = bottom; ch[K] = top;
    for (int i = 0; i < K; ++i) h[i] = ch[i + 1] - ch[i];
    /* searchsorted, neural_splines.py:11-13: last location += eps; sum(x >= loc) - 1 */
    const float* loc = inverse ? ch : cw;
    int b = -1;
    for (int i = 0; i <= K; ++i) { float l = loc[i]; if (i == K) l = l + 1e-6f; if (x >= l) b++; }
    if (b < 0) b = 0;
    if (b > K - 1) b = K - 1;
    if (bin) *bin = b;
    float in_cw = cw[b], in_w = w[b], in_ch = ch[b], in_h = h[b];
    float delta = h[b] / w[b];
    float d0 = der[b], d1 = der[b + 1];
    if (inverse) {
        float a = (x - in_ch) * (d0 + d1 - 2 * delta) + in_h * (delta - d0);
        float bq = in_h * d0 - (x - in_ch) * (d0 + d1 - 2 * delta);
        float cq = -delta * (x - in_ch);
        float disc = bq * bq - 4 * a * cq;
        float root = (2 * cq) / (-bq - sqrtf(disc));
        *out = root * in_w + in_cw;
        float t1 = root * (1 - root);
        float den = delta + ((d0 + d1 - 2 * delta) * t1);
        float num = delta * delta * (d1 * root * root + 2 * delta * t1 + d0 * (1 - root) * (1 - root));
        *logabsdet = -(logf(num) - 2 * logf(den));
    } else {
        float th = (x - in_cw) / in_w;
        float t1 = th * (1 - th);
        float num = in_h * (delta * th * th + d0 * t1);
        float den = delta + ((d0 + d1 - 2 * delta) * t1);
        *out = in_ch + num / den;
        float dnum = delta * delta * (d1 * th * th + 2 * delta * t1 + d0 * (1 - th) * (1 - th));
        *logabsdet = logf(dnum) - 2 * logf(den);
    }
    return 0;
}

/* Batched form.  n_deriv == K+1: RQS (neural_splines.py:74-184).  n_deriv == K-1: unconstrained_RQS (:16-71):
 * derivatives padded with the constant log(exp(1 - 1e-3) - 1) at both ends, identity outside [left, right]. */
int wfo_rqs_batch(const float* x, const float* uw, const float* uh, const float* ud, int64_t N, int K, int n_deriv, int inverse,
                  float left, float right, float bottom, float top, float* out, float* logabsdet, int32_t* bin) {
    if (K > WFO_RQS_MAXK) return -1;
    const float edge = logf(expf(1 - 1e-3f) - 1);
    for (int64_t e = 0; e < N; ++e) {
        float d[WFO_RQS_MAXK + 1];
        if (n_deriv == K - 1) {
            d[0] = edge; d[K] = edge;
            for (int i = 0; i < K - 1; ++i) d[i + 1] = ud[e * n_deriv + i];
            if (!(x[e] >= left && x[e] <= right)) { out[e] = x[e]; logabsdet[e] = 0.0f; if (bin) bin[e] = -1; continue; }
        } else {
            for (int i = 0; i <= K; ++i) d[i] = ud[e * n_deriv + i];
        }
        int b;
        int rc = wfo_rqs(x[e], uw + e * K, uh + e * K, d, K, inverse, left, right, bottom, top, out + e, logabsdet + e, &b);
        if (rc) return rc;
        if (bin) bin[e] = b;
    }
    return 0;
}
