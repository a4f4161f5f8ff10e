import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from waveflow_amd import vqmc
from waveflow_amd.utils import physics
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
psi, log_pdf, sample, st, opt_update, get_params = vqmc.create_train_state(10, 1e-3, 2, rng=0)
h_fn = physics.construct_hamiltonian_function(psi, protons=physics.system_catalogue[1]['He'][0], n_space_dimensions=1, eps=0.0)
params = get_params(st)
m = psi.model
def T(f, n=50):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
flat = st.x.cpu().numpy().copy()
i = [0]
def setp():
    i[0] += 1
    f = flat.copy(); f[0] += 1e-6 * i[0]
    m.set_params(f)
print(f"B={B}")
print("set_params          %.3f ms" % T(setp))
print("sample (reference)  %.3f ms" % T(lambda: sample(1, params, B)))
print("sample (exact)      %.3f ms" % T(lambda: sample(1, params, B, exact_inverse=True)))
x = sample(1, params, B, exact_inverse=True)
print("loss_grad (device)  %.3f ms" % T(lambda: m.vqmc_loss_grad(x, h_fn.protons, 0.0)))
print("loss_and_grad+host  %.3f ms" % T(lambda: vqmc.loss_and_grad_efficient(params, psi, h_fn, x, 0.0)))
g = np.zeros_like(flat)
print("adam (host)         %.3f ms" % T(lambda: opt_update(1, g, st)))
print("get_params          %.3f ms" % T(lambda: get_params(st)))
from waveflow_amd.core import flatten_params
print("flatten_params      %.3f ms" % T(lambda: flatten_params(params)))
