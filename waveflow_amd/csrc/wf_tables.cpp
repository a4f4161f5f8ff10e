// wf_tables.cpp -- host-side basis tables for libwaveflow_hip (fp64, one-off at model build).
//
// What it produces is what the reference caches under ./cached_splines_bases/{M,I,B}/ at the
// first init_fun (waveflow/splines/isplines_jax.py:106-131, bsplines_jax.py:68-116,
// msplines_jax.py:84-108): for derivative orders 0..3, every basis function sampled on
// linspace(0, 1, n_mesh).  The reference evaluates each sample with a Python recursion
// (splines_np.py:42-137); here each mesh point is evaluated once for ALL bases and ALL
// derivative orders with a bottom-up Cox-de Boor triangle.  Every node of the triangle is the
// same floating-point expression as the corresponding recursive call, so the tables are
// bit-identical to the reference's fixtures (tests/test_tables.py).
//
// The orthogonalised B basis (bsplines_jax.py:98-106, ortho_splines.py:43-161) is built with
// coefficient tracking: the symmetric Gram-Schmidt runs on the 2000-sample vectors while the
// same row operations are applied to an identity matrix, which yields b_to_ob directly (no
// pseudo-inverse); ob_to_b is its inverse.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

#include "wf_internal.h"

namespace wf {

// numpy.linspace(0, 1, n)[i]
static inline double linspace01(int i, int n) {
    if (i == n - 1) return 1.0;
    const double step = 1.0 / (double)(n - 1);
    return (double)i * step + 0.0;
}

std::vector<double> make_knots(int kind, int k, int n_internal) {
    // M: first/last knot k-fold (msplines_jax.py:72-74); I, B: (k+1)-fold (isplines_jax.py:91-93,
    // bsplines_jax.py:58-60)
    const int rep = std::max(1, kind == WF_SPLINE_M ? k : k + 1);
    std::vector<double> t;
    for (int r = 0; r < rep; ++r) t.push_back(linspace01(0, n_internal));
    for (int i = 1; i < n_internal - 1; ++i) t.push_back(linspace01(i, n_internal));
    for (int r = 0; r < rep; ++r) t.push_back(linspace01(n_internal - 1, n_internal));
    return t;
}

int n_bases_of(int kind, int k, int n_internal) {
    const int nt = (int)make_knots(kind, k, n_internal).size();
    return (kind == WF_SPLINE_B || kind == WF_SPLINE_OB) ? nt - k - 1 : nt - k;
}

namespace {

constexpr int ND = 4;

// M-spline triangle at one abscissa: v[nd][order][i], order = 1..K (splines_np.py:42-62).
struct MTriangle {
    int K, nt;
    std::vector<double> v;
    MTriangle(int K_, int nt_) : K(K_), nt(nt_), v((size_t)ND * (K_ + 1) * nt_, 0.0) {}
    double& at(int nd, int k, int i) { return v[((size_t)nd * (K + 1) + k) * nt + i]; }

    void eval(double x, const double* t, int max_k) {
        std::fill(v.begin(), v.end(), 0.0);
        for (int i = 0; i + 1 < nt; ++i) {
            const bool inside = (x >= t[i] && x < t[i + 1]) || (i >= nt - (max_k + 1) && x >= t[i] && x <= t[i + 1]);
            if (inside && t[i + 1] - t[i] != 0) at(0, 1, i) = 1 / (t[i + 1] - t[i]);
        }
        for (int k = 2; k <= K; ++k)
            for (int i = 0; i + k < nt; ++i) {
                const double span = t[i + k] - t[i];
                if (span == 0) continue;
                const double xl = x - t[i], xr = t[i + k] - x;
                at(0, k, i) = (double)k * (xl * at(0, k - 1, i) + xr * at(0, k - 1, i + 1)) / ((double)(k - 1) * span);
                const double pre = (double)k / ((double)(k - 1) * span);
                at(1, k, i) = pre * (((xl * at(1, k - 1, i) + xr * at(1, k - 1, i + 1)) + at(0, k - 1, i)) - at(0, k - 1, i + 1));
                for (int nd = 2; nd < ND; ++nd)
                    at(nd, k, i) = pre * ((xl * at(nd, k - 1, i) + xr * at(nd, k - 1, i + 1)) +
                                          (double)nd * (at(nd - 1, k - 1, i) - at(nd - 1, k - 1, i + 1)));
            }
    }
};

// numpy's add.reduce on a short contiguous double vector (the reference sums the I-spline terms
// with np.array([...]).sum(), splines_np.py:93): straight loop below 8 terms, 8 lanes above.
double numpy_sum(const double* a, int n) {
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// B-spline triangle: v[nd][degree][i], degree = 0..K (splines_np.py:101-137).
struct BTriangle {
    int K, nt;
    std::vector<double> v;
    BTriangle(int K_, int nt_) : K(K_), nt(nt_), v((size_t)ND * (K_ + 1) * nt_, 0.0) {}
    double& at(int nd, int k, int i) { return v[((size_t)nd * (K + 1) + k) * nt + i]; }

    void eval(double x, const double* t, int max_k) {
        std::fill(v.begin(), v.end(), 0.0);
        for (int i = 0; i + 1 < nt; ++i) {
            const bool inside = (t[i] <= x && x < t[i + 1]) || (i >= nt - (max_k + 2) && x >= t[i] && x <= t[i + 1]);
            at(0, 0, i) = inside ? 1.0 : 0.0;
        }
        for (int k = 1; k <= K; ++k)
            for (int i = 0; i + k + 1 < nt; ++i) {
                const double c1 = t[i + k] == t[i] ? 0.0 : (x - t[i]) / (t[i + k] - t[i]) * at(0, k - 1, i);
                const double c2 =
                    t[i + k + 1] == t[i + 1] ? 0.0 : (t[i + k + 1] - x) / (t[i + k + 1] - t[i + 1]) * at(0, k - 1, i + 1);
                at(0, k, i) = c1 + c2;
                for (int nd = 1; nd < ND; ++nd) {
                    const double d1 = (t[i + k] - t[i] == 0) ? 0 : at(nd - 1, k - 1, i) / (t[i + k] - t[i]);
                    const double d2 = (t[i + k + 1] - t[i + 1] == 0) ? 0 : at(nd - 1, k - 1, i + 1) / (t[i + k + 1] - t[i + 1]);
                    at(nd, k, i) = (double)k * (d1 - d2);
                }
            }
    }
};

// np.searchsorted(t, x, 'left')
int lower_bound_idx(const std::vector<double>& t, double x) {
    return (int)(std::lower_bound(t.begin(), t.end(), x) - t.begin());
}

}  // namespace

// out: [4][nb][n_mesh]
int build_raw_table(int kind, int k, int n_internal, int n_mesh, double* out) {
    if (k < 1 || n_internal < 2 || n_mesh < 2) return WF_ERR_INVALID;
    const std::vector<double> t = make_knots(kind, k, n_internal);
    const int nt = (int)t.size();
    const int nb = n_bases_of(kind, k, n_internal);
    if (nb < 1) return WF_ERR_INVALID;
    const size_t plane = (size_t)nb * n_mesh;
    if (kind == WF_SPLINE_M) {
        if (k < 2) return WF_ERR_UNSUPPORTED;
        MTriangle tri(k, nt);
        for (int m = 0; m < n_mesh; ++m) {
            tri.eval(linspace01(m, n_mesh), t.data(), k);
            for (int nd = 0; nd < ND; ++nd)
                for (int i = 0; i < nb; ++i) out[nd * plane + (size_t)i * n_mesh + m] = tri.at(nd, k, i);
        }
    } else if (kind == WF_SPLINE_I) {
        // I(x, k, i) = sum_{m=i..j} (t[m+k+1]-t[m]) * M(x, k+1, m) / (k+1), splines_np.py:79-93
        MTriangle tri(k + 1, nt);
        std::vector<double> terms(nt);
        for (int m = 0; m < n_mesh; ++m) {
            const double x = linspace01(m, n_mesh);
            tri.eval(x, t.data(), k + 1);
            const int j = (x == 0.0) ? k : lower_bound_idx(t, x) - 1;
            for (int nd = 0; nd < ND; ++nd)
                for (int i = 0; i < nb; ++i) {
                    double v;
                    if (i > j || i == nt - (k + 1)) v = 0;
                    else if (i <= j - k) v = nd == 0 ? 1 : 0;
                    else {
                        int n = 0;
                        for (int q = i; q <= j; ++q) terms[n++] = (t[q + k + 1] - t[q]) * tri.at(nd, k + 1, q) / (double)(k + 1);
                        v = numpy_sum(terms.data(), n);
                    }
                    out[nd * plane + (size_t)i * n_mesh + m] = v;
                }
        }
    } else {
        BTriangle tri(k, nt);
        for (int m = 0; m < n_mesh; ++m) {
            tri.eval(linspace01(m, n_mesh), t.data(), k);
            for (int nd = 0; nd < ND; ++nd)
                for (int i = 0; i < nb; ++i) out[nd * plane + (size_t)i * n_mesh + m] = tri.at(nd, k, i);
        }
    }
    return nb;
}

namespace {

// Left-to-right Gram-Schmidt on the columns listed in `order` (ortho_splines.py:140-161), carried
// out on coefficient vectors: q_c = sum_j C[c][j] * b_j.  G is the Gram matrix of the b_j.
// Returns C (rows = orthonormal vectors in processing order).
std::vector<std::vector<long double>> gs_l2r(const std::vector<std::vector<long double>>& G, const std::vector<int>& order) {
    const int M = (int)order.size(), nb = (int)G.size();
    std::vector<std::vector<long double>> C(M, std::vector<long double>(nb, 0.0L));
    auto dot = [&](const std::vector<long double>& a, const std::vector<long double>& b) {
        long double s = 0;
        for (int i = 0; i < nb; ++i) {
            long double r = 0;
            for (int j = 0; j < nb; ++j) r += G[i][j] * b[j];
            s += a[i] * r;
        }
        return s;
    };
    for (int c = 0; c < M; ++c) {
        std::vector<long double> v(nb, 0.0L);
        v[order[c]] = 1.0L;
        // modified Gram-Schmidt, two passes for orthogonality at working precision
        for (int pass = 0; pass < 2; ++pass)
            for (int p = 0; p < c; ++p) {
                const long double h = dot(C[p], v);
                for (int i = 0; i < nb; ++i) v[i] -= h * C[p][i];
            }
        const long double nrm = std::sqrt(dot(v, v));
        for (int i = 0; i < nb; ++i) C[c][i] = v[i] / nrm;
    }
    return C;
}

bool invert(std::vector<std::vector<long double>> a, std::vector<std::vector<long double>>& inv) {
    const int n = (int)a.size();
    inv.assign(n, std::vector<long double>(n, 0.0L));
    for (int i = 0; i < n; ++i) inv[i][i] = 1.0L;
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0L) return false;
        std::swap(a[p], a[c]);
        std::swap(inv[p], inv[c]);
        const long double d = a[c][c];
        for (int j = 0; j < n; ++j) { a[c][j] /= d; inv[c][j] /= d; }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const long double f = a[r][c];
            if (f == 0.0L) continue;
            for (int j = 0; j < n; ++j) { a[r][j] -= f * a[c][j]; inv[r][j] -= f * inv[c][j]; }
        }
    }
    return true;
}

}  // namespace

// Orthogonalised B tables.  ob: [4][nb][n_mesh]; b_to_ob, ob_to_b: [nb][nb] (either may be null).
int build_ortho_b(int k, int n_internal, int n_mesh, const double* Bt /* [4][nb][n_mesh] */, double* ob, double* b_to_ob,
                  double* ob_to_b) {
    const int nb = n_bases_of(WF_SPLINE_B, k, n_internal);
    if (nb % 2) return WF_ERR_NUMERIC;  // the reference exits on an odd basis count, ortho_splines.py:58-63
    const int npair = nb / 2;
    const size_t plane = (size_t)nb * n_mesh;
    // Gram matrix of the sampled basis vectors (ovlp = mat.T @ mat, ortho_splines.py:65)
    std::vector<std::vector<long double>> G(nb, std::vector<long double>(nb, 0.0L));
    for (int i = 0; i < nb; ++i)
        for (int j = i; j < nb; ++j) {
            long double s = 0;
            for (int m = 0; m < n_mesh; ++m) s += (long double)Bt[(size_t)i * n_mesh + m] * Bt[(size_t)j * n_mesh + m];
            G[i][j] = G[j][i] = s;
        }
    // processing orders of the two sweeps (ind_j / ind_k shuffles, ortho_splines.py:72-89):
    // left sweep visits 0, M-1, 1, M-2, ... ; right sweep visits M-1, 0, M-2, 1, ...
    std::vector<int> ordL(nb), ordR(nb);
    for (int i = 0; i < npair; ++i) {
        ordL[2 * i] = i;          ordL[2 * i + 1] = nb - 1 - i;
        ordR[2 * i] = nb - 1 - i; ordR[2 * i + 1] = i;
    }
    const auto CL = gs_l2r(G, ordL), CR = gs_l2r(G, ordR);
    // symmetrisation of the pair (left vector 2i, right vector 2i), ortho_splines.py:97-103,115-137
    std::vector<std::vector<long double>> T(nb, std::vector<long double>(nb, 0.0L));  // rows: ob_i in terms of b_j
    for (int i = 0; i < npair; ++i) {
        const auto &v1 = CL[2 * i], &v2 = CR[2 * i];
        long double ov = 0;
        for (int a = 0; a < nb; ++a) {
            long double r = 0;
            for (int b = 0; b < nb; ++b) r += G[a][b] * v2[b];
            ov += v1[a] * r;
        }
        if (!(ov >= 0 && ov <= 1)) return WF_ERR_NUMERIC;  // the reference asserts this, ortho_splines.py:128
        const long double s1 = 1.0L / std::sqrt(1 + ov), s2 = 1.0L / std::sqrt(1 - ov);
        const long double a1 = 0.5L * (s1 + s2), a2 = 0.5L * (s1 - s2);
        for (int a = 0; a < nb; ++a) {
            T[i][a] = a1 * v1[a] + a2 * v2[a];
            T[nb - 1 - i][a] = a2 * v1[a] + a1 * v2[a];
        }
    }
    // scaling: * sqrt(N) (ortho_splines.py:105-107), then / sqrt(sum(ob_0^2)/n_mesh) (bsplines_jax.py:99)
    {
        long double n0 = 0;
        for (int a = 0; a < nb; ++a) {
            long double r = 0;
            for (int b = 0; b < nb; ++b) r += G[a][b] * T[0][b];
            n0 += T[0][a] * r;
        }
        const long double sq = std::sqrt((long double)n_mesh);
        const long double scale = sq / std::sqrt((n0 * sq * sq) / (long double)n_mesh);
        for (auto& row : T)
            for (auto& v : row) v *= scale;
    }
    std::vector<std::vector<long double>> Tinv;
    if (!invert(T, Tinv)) return WF_ERR_NUMERIC;
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
            if (b_to_ob) b_to_ob[(size_t)i * nb + j] = (double)T[i][j];
            if (ob_to_b) ob_to_b[(size_t)i * nb + j] = (double)Tinv[i][j];
        }
    if (ob)
        for (int nd = 0; nd < ND; ++nd)
            for (int i = 0; i < nb; ++i)
                for (int m = 0; m < n_mesh; ++m) {
                    long double s = 0;
                    for (int j = 0; j < nb; ++j) s += T[i][j] * (long double)Bt[nd * plane + (size_t)j * n_mesh + m];
                    ob[nd * plane + (size_t)i * n_mesh + m] = (double)s;
                }
    return nb;
}

}  // namespace wf
