// wf_etile_bwd_k2.hip -- the two-row-block instantiations of the matrix-core reverse kernel (k_ebwd<., 2>, wf_kernels_etile.hip) as a translation unit of their own,
// compiled under -mllvm -amdgpu-sched-strategy=max-ilp (waveflow_amd/build.py): 1.532 -> 1.444 ms per loss + gradient of 2^17 walkers of the 33-knot model; the same
// strategy costs the one-row-block form 1 %.  The file is wf_kernels_etile.hip with its host side switched off.
#define WF_ETILE_ONLY_K2 1
#include "wf_kernels_etile.hip"
