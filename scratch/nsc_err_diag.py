"""Error of the coupling-stack kernel's log_pdf against the layer-by-layer restatement, next to the restatement's own sensitivity to a
1-ulp-level perturbation of the input (the conditioning of log_pdf at x): calibrates the per-element bound of
tests/test_gpu_rqs.py::test_neural_spline_coupling_stack_as_a_model."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_rqs as T
from waveflow_amd import flows

for dim, K, hidden, reverse, prior in [(2, 5, 8, True, "normal"), (4, 5, 8, True, "normal"), (2, 8, 32, False, "uniform"), (3, 5, 8, True, "uniform")]:
    g = np.random.default_rng(100 + dim)
    items = []
    for _ in range(3):
        items += [flows.NeuralSplineCoupling(K=K, B=3, hidden_dim=hidden), flows.Reverse()] if reverse else [flows.NeuralSplineCoupling(K=K, B=3, hidden_dim=hidden)]
    pr = flows.Normal(-0.25) if prior == "normal" else flows.Uniform()
    try:
        init = flows.Flow(flows.Serial(*items), pr, prior_support=None if prior == "normal" else (0.0, 1.0))
        params, log_pdf, sample = init(7, dim)
    except Exception as e:
        print(dim, K, hidden, "skipped:", e); continue
    scale = lambda net: [tuple(a * (1.5 if a.ndim == 2 else 3e4) for a in l) if l else () for l in net]
    params = [tuple(scale(net) for net in p) if p else () for p in params]
    x = g.uniform(-3.4, 3.4, size=(4099, dim)).astype(np.float32)

    def want_of(xx):
        z, ld = xx.astype(np.float32), np.zeros(len(xx))
        for p in params:
            if p:
                z, l = T._nsc_oracle(p, z, K, 3.0, False); z = z.astype(np.float32); ld = ld + l
            else:
                z = z[:, ::-1]
        return ld + (-0.5 * (np.log(2 * np.pi) + (z.astype(np.float64) - 0.25) ** 2)).sum(1) if prior == "normal" else ld

    lp = np.asarray(log_pdf(params, x))
    want = want_of(x)
    sens = np.zeros(len(x))
    for d in range(dim):
        for sgn in (1, -1):
            xp = x.copy(); xp[:, d] = np.nextafter(xp[:, d], np.float32(sgn * 10.0))
            sens = np.maximum(sens, np.abs(want_of(xp) - want))
    err = np.abs(lp - want)
    r = err / (1e-5 + sens)
    print(f"dim {dim} K {K} hidden {hidden} {prior}: err median {np.median(err):.2e} p99 {np.quantile(err, 0.99):.2e} p99.9 {np.quantile(err, 0.999):.2e} max {err.max():.2e}; "
          f"1-ulp sensitivity median {np.median(sens):.2e} p99 {np.quantile(sens, 0.99):.2e} max {sens.max():.2e}; err / (1e-5 + sens): p99 {np.quantile(r, 0.99):.1f} max {r.max():.1f}; "
          f"n(err > 1e-3) {(err > 1e-3).sum()}, of which sens > 1e-4: {((err > 1e-3) & (sens > 1e-4)).sum()}")
