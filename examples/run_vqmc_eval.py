#!/usr/bin/env python3
"""Evaluation half of examples/run_vqmc.py on MI355X: same knobs (vqmc.ModelTrainer attributes, vqmc.py:20-51), but
instead of training it loads a checkpoint (reference pickle or flat npz), draws walkers with the model's sampler and
reports the VQMC energy <E_L> = <H psi / (psi + 1e-8)> (vqmc.py:193-200) over all ranks with ONE all-reduce of three
fp64 numbers (RCCL over xGMI), then writes the reference's checkpoint artefacts (helpers.create_checkpoint_wavefunc).

    python examples/run_vqmc_eval.py [--checkpoint PATH] [--batch 65536] [--save-dir ./results/He_1d_L10box]
    python -m torch.distributed.run --nproc-per-node 8 examples/run_vqmc_eval.py ...
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--checkpoint", default=None, help="reference `checkpoints` pickle; default: the shipped He parameters")
ap.add_argument("--batch", type=int, default=1 << 16, help="walkers per GPU")
ap.add_argument("--box-length", type=float, default=10)
ap.add_argument("--save-dir", default=None)
ap.add_argument("--exact-inverse", action="store_true", help="sample from |psi|^2 exactly instead of reproducing made.py:88")
args = ap.parse_args()

import torch
import torch.distributed as dist
from waveflow_amd import checkpoint, distributed as wfd
from waveflow_amd.model_factory import get_waveflow_model
from waveflow_amd.utils import helpers, physics

rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl")

# vqmc.ModelTrainer defaults (vqmc.py:26-34) + examples/run_vqmc.py:3-13
system_name, n_space_dimension = "He", 1
protons, n_particle = physics.system_catalogue[n_space_dimension][system_name]
spline_degree, num_knots, n_flow_layer = 6, 23, 3
init_fun = get_waveflow_model(n_particle, base_spline_degree=spline_degree, i_spline_degree=spline_degree, n_prior_internal_knots=num_knots,
                              n_i_internal_knots=num_knots, i_spline_reg=0.05, i_spline_reverse_fun_tol=0.000001,
                              n_flow_layers=n_flow_layer, box_size=args.box_length, xu_coord_type="mean")
params, psi, log_pdf, sample = init_fun(2, n_particle)
epoch = 0
if args.checkpoint:
    loaded, epoch = checkpoint.load_reference_checkpoint(args.checkpoint)
    params = loaded
else:
    flat = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "he_checkpoint.npz"))
    params, epoch = checkpoint.unflatten_like(params, flat["flat"]), int(flat["epoch"])

h_fn = physics.construct_hamiltonian_function(psi, protons=protons, n_space_dimensions=n_space_dimension, eps=0.0)
batch = sample(1234 + rank, params, args.batch, exact_inverse=args.exact_inverse)
psi_val = psi(params, batch)
local_energy = h_fn(params, batch)[:, 0] / (psi_val + 1e-8)
mean, var, stderr = wfd.ShardedDensity(psi.model).expectation(local_energy)
if rank == 0:
    print(f"epoch {epoch} | walkers {args.batch * world} on {world} GPU(s) | <E_L> = {mean:.4f} +- {stderr:.4f} (sample variance {var:.3e})")
    if args.save_dir:
        system_dict = {"system_name": system_name, "box_length": args.box_length, "n_particle": n_particle,
                       "n_space_dimension": n_space_dimension, "window": 100, "n_plotting": 200}
        helpers.create_checkpoint_wavefunc(7, args.save_dir, psi, sample, params, epoch, [mean], [[mean]], system_dict)
        print("wrote", args.save_dir)
if world > 1:
    dist.destroy_process_group()
