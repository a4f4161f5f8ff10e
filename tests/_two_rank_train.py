"""Helper of tests/test_gpu_grad.py: one rank of a 2-rank training run that shares cuda:0 (gloo rendezvous on 127.0.0.1)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from waveflow_amd import vqmc  # noqa: E402

out_dir, steps, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
t = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=steps, batch_size=batch, log_every=10 ** 9)
t.save_dir = os.path.join(out_dir, "run")
t.exact_sampler = True
params, loss = t.start_training(verbose=False)
np.save(os.path.join(out_dir, f"params_rank{rank}.npy"), params.flat.cpu().numpy())
np.save(os.path.join(out_dir, f"loss_rank{rank}.npy"), np.asarray(loss[1:], dtype=np.float64))
dist.destroy_process_group()
