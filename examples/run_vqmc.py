"""VQMC training of the 1-D He model on one MI355X -- the drop-in of the reference's examples/run_vqmc.py
(only the import changes; optional overrides on the command line for a short run):

    python examples/run_vqmc.py [--epochs N] [--batch B] [--lr LR] [--exact-sampler] [--save-dir DIR]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/run_vqmc.py --batch 8192 ...
        (one process per GPU: the walkers of a step are split over the ranks, one RCCL all-reduce per step)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from waveflow_amd import vqmc  # noqa: E402   (reference: from waveflow import vqmc)

if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("WF_FORCE_DIST") == "1":
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=int(os.environ.get("RANK", "0")), world_size=int(os.environ.get("WORLD_SIZE", "1")),
                            device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=80000)
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--log-every", type=int, default=10000)
ap.add_argument("--exact-sampler", action="store_true", help="draw walkers from |psi|^2 instead of the reference's sampler")
ap.add_argument("--save-dir", default=None)
args = ap.parse_args()

box_length = 12
n_knots = 23
n_layer = 3
spline_degree = 6
trainer = vqmc.ModelTrainer(num_epochs=args.epochs, box_length=box_length, batch_size=args.batch, log_every=args.log_every,
                            learning_rate=args.lr)
trainer.num_knots = n_knots
trainer.n_flow_layer = n_layer
trainer.spline_degree = spline_degree
trainer.exact_sampler = args.exact_sampler
if args.save_dir:
    trainer.save_dir = args.save_dir
trainer.start_training()

if "dist" in globals() and dist.is_initialized():
    dist.destroy_process_group()
