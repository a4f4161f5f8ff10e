#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b52
timeout 1500 python3 -m pytest tests/test_gpu_grad.py tests/test_gpu_energy.py -m gpu -x -q -k "nonzero_boundary or local_energy or hamiltonian or matrix_cores or tile_path or psi_vjp" > gpurun_out/b52/tests.txt 2>&1
tail -12 gpurun_out/b52/tests.txt
