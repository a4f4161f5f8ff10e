#!/bin/bash
# rebuild the wave kernels with another register-allocation target per kernel family and time what each family drives
bench() {
  python scratch/grad_ab.py 2 4 8 2>&1 | grep "^D=" | cut -c1-75
  python scratch/step_breakdown.py 128 2>&1 | grep -E "loss_grad|sample \(exact"
  python scratch/bench_energy.py 2>&1 | grep hamiltonian
  python scratch/bench_mle.py 2>&1 | grep -E "IFlow|MFlow" | cut -c1-80
  python scratch/crossover.py 2>&1 | grep -E "B=256|B=4096" | cut -c1-30
}
for cfg in "" "-DWF_OCC_FWD1=2" "-DWF_OCC_BWD1=2" "-DWF_OCC_FWD2=2" "-DWF_OCC_BWD2=2" "-DWF_OCC_SAMPLE=2" "-DWF_OCC_BWD2=2 -DWF_OCC_FWD2=2"; do
  touch waveflow_amd/csrc/wf_kernels_wave.hip
  WF_CXXFLAGS="$cfg" python -m waveflow_amd.build > /dev/null 2>&1
  echo "=== flags: $cfg"
  bench
done
