#!/bin/bash
# the whole GPU suite, the smoke entry and the default bench line, as the driver runs them at the end of a round
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/full
timeout 2400 python3 -m pytest tests -m gpu -x -q > gpurun_out/full/tests.txt 2>&1
tail -5 gpurun_out/full/tests.txt
timeout 900 python3 __graft_entry__.py --smoke > gpurun_out/full/smoke.txt 2>&1; tail -8 gpurun_out/full/smoke.txt
timeout 900 python3 bench.py > gpurun_out/full/bench.txt 2>&1; tail -2 gpurun_out/full/bench.txt
