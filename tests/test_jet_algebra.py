"""The coefficient algebras of the derivative sweeps (waveflow_amd/csrc/wf_ring.h), restated in NumPy: R3 = IR[t]/t^3 and
RF<D> = (value, gradient, Laplacian / 2).  What the kernels rely on: both are commutative rings, function composition lifts by
the chain rule, and both are Frobenius algebras -- under the pairing <a, b> = top coefficient of a b the adjoint of "multiply by
b" is "multiply by b", which is what lets the reverse sweep reuse the forward arithmetic (adjoints stored with value and top slot
exchanged).  CPU-only; the kernels themselves are checked in test_gpu_grad.py / test_gpu_energy.py."""
import numpy as np
import pytest


class R3:
    n = 3

    @staticmethod
    def mul(a, b):
        return np.array([a[0] * b[0], a[1] * b[0] + a[0] * b[1], a[2] * b[0] + a[1] * b[1] + a[0] * b[2]])

    @staticmethod
    def lift(a, f, f1, f2):
        return np.array([f, f1 * a[1], f1 * a[2] + 0.5 * f2 * a[1] ** 2])


def RF(D):
    class _RF:
        n = D + 2

        @staticmethod
        def mul(a, b):
            return np.concatenate([[a[0] * b[0]], a[1:-1] * b[0] + a[0] * b[1:-1], [a[-1] * b[0] + a[0] * b[-1] + a[1:-1] @ b[1:-1]]])

        @staticmethod
        def lift(a, f, f1, f2):
            return np.concatenate([[f], f1 * a[1:-1], [f1 * a[-1] + 0.5 * f2 * (a[1:-1] @ a[1:-1])]])
    return _RF


def partner(n, k):          # ring_partner in wf_kernels_grad.hip
    return n - 1 if k == 0 else (0 if k == n - 1 else k)


def pair(A, a, b):          # <a, b> = top coefficient of a b
    return A.mul(a, b)[-1]


@pytest.mark.parametrize("A", [R3, RF(2), RF(5)])
def test_commutative_ring_and_frobenius_pairing(A):
    g = np.random.default_rng(0)
    a, b, c = (g.normal(size=A.n) for _ in range(3))
    np.testing.assert_allclose(A.mul(a, b), A.mul(b, a), atol=1e-13)
    np.testing.assert_allclose(A.mul(A.mul(a, b), c), A.mul(a, A.mul(b, c)), atol=1e-12)
    np.testing.assert_allclose(A.mul(a, b + c), A.mul(a, b) + A.mul(a, c), atol=1e-12)
    # the pairing couples slot k with slot partner(k), weight 1: that is the contraction k_wgrad performs
    want = sum(a[k] * b[partner(A.n, k)] for k in range(A.n))
    assert abs(pair(A, a, b) - want) < 1e-12
    # non-degenerate, and multiplication is self-adjoint under it: <a b, c> = <a, b c>
    G = np.array([[pair(A, np.eye(A.n)[i], np.eye(A.n)[j]) for j in range(A.n)] for i in range(A.n)])
    assert abs(np.linalg.det(G)) > 0.5
    assert abs(pair(A, A.mul(a, b), c) - pair(A, a, A.mul(b, c))) < 1e-12
    # hence, for y = a b and an adjoint ybar (a ring element under the pairing), abar = ybar b: d<ybar, y> = <ybar b, da>
    da, ybar = g.normal(size=A.n), g.normal(size=A.n)
    assert abs(pair(A, ybar, A.mul(da, b)) - pair(A, A.mul(ybar, b), da)) < 1e-12


@pytest.mark.parametrize("D", [2, 4])
def test_rf_carries_value_gradient_and_half_laplacian(D):
    """F(x) = tanh(u(x)) * exp(v(x)) with quadratic u, v: RF arithmetic on the jets of u and v gives the jet of F."""
    A = RF(D)
    g = np.random.default_rng(1)
    Qu, Qv, bu, bv = g.normal(size=(D, D)), g.normal(size=(D, D)), g.normal(size=D), g.normal(size=D)
    Qu, Qv = Qu + Qu.T, Qv + Qv.T
    x = g.normal(size=D) * 0.3

    def jet(Q, b):   # value, gradient, Laplacian / 2 of 1/2 x Q x + b x
        return np.concatenate([[0.5 * x @ Q @ x + b @ x], Q @ x + b, [0.5 * np.trace(Q)]])

    ju, jv = jet(Qu, bu), jet(Qv, bv)
    t = np.tanh(ju[0])
    e = np.exp(jv[0])
    F = A.mul(A.lift(ju, t, 1 - t * t, -2 * t * (1 - t * t)), A.lift(jv, e, e, e))

    def f(y):
        return np.tanh(0.5 * y @ Qu @ y + bu @ y) * np.exp(0.5 * y @ Qv @ y + bv @ y)

    h = 1e-4
    grad = np.array([(f(x + h * np.eye(D)[i]) - f(x - h * np.eye(D)[i])) / (2 * h) for i in range(D)])
    lap = sum((f(x + h * np.eye(D)[i]) - 2 * f(x) + f(x - h * np.eye(D)[i])) / h ** 2 for i in range(D))
    assert abs(F[0] - f(x)) < 1e-12
    np.testing.assert_allclose(F[1:-1], grad, rtol=1e-6, atol=1e-8)
    assert abs(2 * F[-1] - lap) < 1e-5 * max(1.0, abs(lap))
    # the same Laplacian from D directional second-order jets in R3 (what WF_GRAD_R3 / WF_ENERGY_R3 run)
    tot = 0.0
    for i in range(D):
        ui = np.array([ju[0], ju[1 + i], 0.5 * Qu[i, i]])
        vi = np.array([jv[0], jv[1 + i], 0.5 * Qv[i, i]])
        Fi = R3.mul(R3.lift(ui, t, 1 - t * t, -2 * t * (1 - t * t)), R3.lift(vi, e, e, e))
        tot += 2 * Fi[2]
    assert abs(tot - 2 * F[-1]) < 1e-10 * max(1.0, abs(tot))
