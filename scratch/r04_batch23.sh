#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b23; mkdir -p $O
stats() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -- "$@" > $O/$name.log 2>&1; find $O/tmp_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv; rm -rf $O/tmp_$name; }
export KN=23
stats new python3 scratch/r04_grad33_prof.py
export WF_LIB=$PWD/scratch/variants/libwf_e1swap.so WF_LIB_EXPERIMENT=1
stats e1swap python3 scratch/r04_grad33_prof.py
unset WF_LIB WF_LIB_EXPERIMENT
stats new2 python3 scratch/r04_grad33_prof.py
for n in new e1swap new2; do echo "== $n"; head -5 $O/${n}_kernel_stats.csv | cut -c1-60,140-240; done
