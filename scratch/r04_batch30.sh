#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b30; mkdir -p $O
for w in 16 12 8; do for B in 131072 1048576; do ./scratch/ubench3/cond16_w$w $B; echo "exit $?"; done; done 2>&1 | tee $O/cond16.txt
for B in 131072 1048576; do
  export B
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$B -- python3 scratch/r04_cond_baseline.py > $O/base_$B.log 2>&1
  find $O/tmp_$B -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/base_${B}_kernel_stats.csv; rm -rf $O/tmp_$B
  echo "== library, B=$B"; grep "k_etile_cond" $O/base_${B}_kernel_stats.csv | cut -c1-80,200-290
done | tee $O/baseline.txt
