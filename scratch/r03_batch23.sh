#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b23
python3 scratch/sample_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/b23/sample_prof.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q -k "sampl or invers or training" > gpurun_out/b23/tests.txt 2>&1
tail -8 gpurun_out/b23/tests.txt
