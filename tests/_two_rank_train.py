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
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("WF_TEST_BACKEND", "gloo")   # "none": no process group at all (single-process reference run)
if backend == "none":
    rank, world = 0, 1   # "nccl" = RCCL (one rank per GPU; the shared-GPU tests use gloo)
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0)
if backend == "none":
    pass
elif backend == "nccl":
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", torch.cuda.current_device()))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
t = vqmc.ModelTrainer(system_name="He", learning_rate=1e-3, box_length=10, num_epochs=steps, batch_size=batch, log_every=10 ** 9)
t.save_dir = os.path.join(out_dir, "run")
t.exact_sampler = True
params, loss = t.start_training(verbose=False)
np.save(os.path.join(out_dir, f"params_rank{rank}.npy"), params.flat.cpu().numpy())
np.save(os.path.join(out_dir, f"loss_rank{rank}.npy"), np.asarray(loss[1:], dtype=np.float64))
if backend != "none":
    dist.destroy_process_group()
