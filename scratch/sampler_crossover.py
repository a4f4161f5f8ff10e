import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import model_factory
flat = np.load('tests/golden/he_checkpoint.npz')['flat']
init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23, n_i_internal_knots=23, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6, n_flow_layers=3, box_size=10)
params, psi, log_pdf, sample = init_fun(0, 2)
m = psi.model; m.set_params(flat)
def T(B, exact):
    for _ in range(2): m.sample(1, B, exact=exact)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): m.sample(1, B, exact=exact)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / 5 * 1e3
for B in (4096, 16384, 65536, 262144, 1048576):
    row = []
    for mx in ("100000000", "0"):
        os.environ["WF_WAVE_SAMPLE_MAX"] = mx
        row.append((T(B, True), T(B, False)))
    print(f"B={B}: wave exact {row[0][0]:.2f} ms ref {row[0][1]:.2f} ms | one-lane exact {row[1][0]:.2f} ms ref {row[1][1]:.2f} ms")
