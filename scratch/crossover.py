"""Where do the matrix-core paths overtake the one-walker-per-wave kernels?  sample / loss + gradient / H psi of the He model at small batches, both ways."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
from waveflow_amd.utils import physics
m, flat = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("%8s | %-23s | %-23s | %-23s" % ("walkers", "sample ms  tile / wave", "loss+grad ms tile / wave", "H psi ms   tile / wave"))
for B in (1024, 2048, 4096, 8192, 16384, 32768):
    x = m.sample(3, B, exact=True)
    row = []
    for env in ("WF_SAMPLE_TILE_MIN", "WF_GRAD_TILE_MIN", "WF_ENERGY_TILE_MIN"):
        pair = []
        for v in ("1", "0"):
            os.environ[env] = v
            if env == "WF_SAMPLE_TILE_MIN": pair.append(t(lambda: m.sample(5, B, exact=True)))
            elif env == "WF_GRAD_TILE_MIN": pair.append(t(lambda: m.vqmc_loss_grad(x, protons, -1.8)))
            else: pair.append(t(lambda: m.hamiltonian(x, protons)))
        del os.environ[env]
        row.append("%9.3f / %9.3f" % tuple(pair))
    print("%8d | %-23s | %-23s | %-23s" % (B, *row), flush=True)
