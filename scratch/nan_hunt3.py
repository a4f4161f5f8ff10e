"""The graphed trainer with the graph replaced by eager calls of the recorded step; stops at the first non-finite loss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import waveflow_amd.core as core
from waveflow_amd import vqmc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
orig = core.DeviceModel.train_step
rec = {}
al = lambda n: (n + 255) // 256 * 256
def rec_step(self, *a, **k):
    rec['call'] = (self, a, k)
class Stop(Exception): pass
class FakeGraph:
    def replay(self):
        m, a, k = rec['call']
        st = a[0]
        prev = st["x"].clone()
        c = int(st["counter"].item())
        orig(m, *a, **k)
        r = st["ring"].cpu().numpy()[c % st["ring_len"]]
        if not np.isfinite(r).all():
            ws = st["ws"]
            x = ws[:B * 2 * 4].view(torch.float32).view(B, 2).clone()
            el = ws[al(B * 2 * 4):al(B * 2 * 4) + B * 4].view(torch.float32).clone()
            g = ws[al(B * 2 * 4) + al(B * 4):al(B * 2 * 4) + al(B * 4) + m.n_params * 4].view(torch.float32).clone()
            bad = torch.nonzero(~torch.isfinite(el)).flatten().tolist()
            print("step", c, "ring", r, "bad e_loc walkers", bad[:8], "bad grad entries", int((~torch.isfinite(g)).sum()), "prev params finite", bool(torch.isfinite(prev).all()))
            print("x of bad", x[bad[:8]].cpu().numpy(), "x finite", bool(torch.isfinite(x).all()), "x range", float(x.min()), float(x.max()))
            np.savez("gpurun_out/nan_case.npz", x=x.cpu().numpy(), flat=prev.cpu().numpy(), el=el.cpu().numpy())
            m.set_params_device(prev)
            pr = a[3]
            h, ps, lap = m.hamiltonian(x, pr, return_psi=True, return_laplacian=True)
            print("same walkers, wf_hamiltonian_fwd (RF untaped): non-finite", int((~torch.isfinite(h)).sum()), "min |psi|", float(ps.abs().min()))
            os.environ["WF_ENERGY_R3"] = "1"
            h3, ps3, lap3 = m.hamiltonian(x, pr, return_psi=True, return_laplacian=True)
            print("R3 untaped: non-finite", int((~torch.isfinite(h3)).sum()))
            for i in bad[:4]:
                print("walker", i, x[i].cpu().numpy(), "RF: Hpsi", float(h[i]), "psi", float(ps[i]), "lap", float(lap[i]), " R3: Hpsi", float(h3[i]), "lap", float(lap3[i]))
            raise Stop()
class fake_ctx:
    def __init__(self, g, stream=None): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False
torch.cuda.CUDAGraph = FakeGraph
torch.cuda.graph = fake_ctx
core.DeviceModel.train_step = rec_step
t = vqmc.ModelTrainer(system_name='He', learning_rate=1e-4, box_length=10, num_epochs=150, batch_size=B, log_every=10**9)
t.save_dir = '/tmp/wf_nan3'
t.exact_sampler = True
try:
    t.start_training(verbose=False)
    print("no NaN")
except Stop:
    pass
