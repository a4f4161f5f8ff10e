"""psi / Laplacian parameter gradients of the He checkpoint: the matrix-core path (k_ebwd + k_ewgrad) against the wave sweeps, per leaf block; timing."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import sorted_walkers
import bench
from waveflow_amd.utils import physics
m, flat = bench.he_model("auto")
protons = physics.system_catalogue[1]["He"][0].reshape(-1)
B = int(os.environ.get("B", 32768))
x = torch.as_tensor(sorted_walkers(B, 2, 9.5, 7)).cuda()
g = np.random.default_rng(3)
w1 = g.normal(size=B).astype(np.float32); w2 = (0.1 * g.normal(size=B)).astype(np.float32)

def run(tile):
    os.environ["WF_GRAD_TILE_MIN"] = "1" if tile else "0"
    out = m.psi_vjp(x, w1, w2).cpu().numpy().astype(np.float64)
    del os.environ["WF_GRAD_TILE_MIN"]
    return out

gt, gw = run(True), run(False)
print("finite:", np.isfinite(gt).all(), " overall rel l2 (tile vs wave): %.3e   |wave| %.3e" % (np.linalg.norm(gt - gw) / np.linalg.norm(gw), np.linalg.norm(gw)))
# leaf blocks: per net W0[2][64], b0[64], W1[64][64], b1[64], W2[64][NO], b2[NO], zero[2][nb]
off = 0
for n, nb in enumerate((29, 29, 29, 28)):
    NO = nb * 2
    for name, size in (("W0", 128), ("b0", 64), ("W1", 4096), ("b1", 64), ("W2", 64 * NO), ("b2", NO), ("zero", NO)):
        a, b = gt[off:off + size], gw[off:off + size]
        nrm = np.linalg.norm(b)
        print("net %d %-4s rel l2 %.3e  (|wave| %.3e, |tile| %.3e)" % (n, name, np.linalg.norm(a - b) / max(nrm, 1e-30), nrm, np.linalg.norm(a)))
        off += size
assert off == gt.size, (off, gt.size)
# loss + gradient entry point (seeds from H psi of the same sweep): walkers from the model's own |psi|^2 sampler, where E_L is well conditioned
x = m.sample(11, B, exact=True)
for tile in (True, False):
    os.environ["WF_GRAD_TILE_MIN"] = "1" if tile else "0"
    sums, grad = m.vqmc_loss_grad(x, protons, -1.8)
    ms = bench.event_ms(lambda: m.vqmc_loss_grad(x, protons, -1.8), 5, 2)
    print("vqmc_loss_grad %s: sums %s  |grad| %.6e   %.3f ms per %d walkers = %.3e walkers/s" % ("tile" if tile else "wave", sums.cpu().numpy(), float(grad.double().norm()), ms, B, B / ms * 1e3))
    if tile: g_t = grad.double().cpu().numpy()
    else: print("loss-grad rel l2 (tile vs wave): %.3e" % (np.linalg.norm(g_t - grad.double().cpu().numpy()) / np.linalg.norm(grad.double().cpu().numpy())))
    del os.environ["WF_GRAD_TILE_MIN"]
