"""Host logic of the trainer (waveflow_amd/vqmc.py): the Adam rule of jax.example_libraries.optimizers.adam (vqmc.py:136)."""
import numpy as np

from waveflow_amd import vqmc


def test_adam_follows_the_jax_example_library_rule():
    g = np.random.default_rng(0)
    params = ([(g.normal(size=(2, 3)).astype(np.float32), g.normal(size=3).astype(np.float32))], (g.normal(size=4).astype(np.float32),))
    opt_init, opt_update, get_params = vqmc.adam(step_size=1e-2)
    st = opt_init(params)
    x = np.concatenate([params[0][0][0].ravel(), params[0][0][1], params[1][0]]).astype(np.float64)
    m, v = np.zeros_like(x), np.zeros_like(x)
    for i in range(25):
        grad_tree = ([(g.normal(size=(2, 3)).astype(np.float32), g.normal(size=3).astype(np.float32))], (g.normal(size=4).astype(np.float32),))
        gr = np.concatenate([grad_tree[0][0][0].ravel(), grad_tree[0][0][1], grad_tree[1][0]]).astype(np.float64)
        st = opt_update(i, grad_tree if i % 2 else gr.astype(np.float32), st)     # pytree or flat gradient
        m = 0.1 * gr + 0.9 * m
        v = 0.001 * gr ** 2 + 0.999 * v
        x = x - 1e-2 * (m / (1 - 0.9 ** (i + 1))) / (np.sqrt(v / (1 - 0.999 ** (i + 1))) + 1e-8)
    out = get_params(st)
    assert out[0][0][0].shape == (2, 3) and out[1][0].shape == (4,)
    got = np.concatenate([out[0][0][0].ravel(), out[0][0][1], out[1][0]])
    np.testing.assert_allclose(got, x, rtol=0, atol=2e-5)


def test_step_size_schedule_and_first_step():
    opt_init, opt_update, get_params = vqmc.adam(step_size=lambda i: 0.5 if i == 0 else 0.0)
    st = opt_init((np.zeros(3, np.float32),))
    st = opt_update(0, np.array([1.0, -2.0, 0.0], np.float32), st)
    # first Adam step moves every coordinate with a gradient by step_size against its sign
    np.testing.assert_allclose(get_params(st)[0], [-0.5, 0.5, 0.0], atol=1e-6)
    st2 = opt_update(1, np.array([1.0, 1.0, 1.0], np.float32), st)
    np.testing.assert_allclose(get_params(st2)[0], get_params(st)[0])


def test_sliding_statistics_and_bisection_helpers():
    from waveflow_amd.utils import helpers
    x = np.arange(10.0)
    a = helpers.uniform_sliding_average(x, 4)
    assert a.shape == x.shape and abs(a[-1] - np.mean(x[-4:])) < 1e-12 and abs(a[0] - x[0]) < 1e-12
    s = helpers.uniform_sliding_stdev(x, 4)
    assert s.shape == x.shape and abs(s[-1] - np.std(x[-4:])) < 1e-12
    assert abs(helpers.moving_average(1.0, 3.0, 0.25) - 1.5) < 1e-12
    r = helpers.binary_search(lambda t: t * t - 0.25, 0.0, 1.0, tol=1e-6)
    assert abs(r - 0.5) < 2e-6 and r <= 0.5
