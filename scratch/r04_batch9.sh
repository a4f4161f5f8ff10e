#!/bin/bash
# gradient path: tests of the two-row-block form, A/B of the shared accumulators against the previous commit (23 knots), kernel stats + scratch traffic of both models
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b9; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_grad.py -q -k "matrix_cores or tile_path or large_batch_training or captured_large" > $O/tests.txt 2>&1; echo "exit $?" >> $O/tests.txt
tail -5 $O/tests.txt
for r in 1 2 3; do
  timeout 300 python3 scratch/r04_grad33_time.py 2>/dev/null | grep "matrix" >> $O/time_new.txt
  WF_LIB=$PWD/scratch/variants/libwf_prev_acc.so WF_LIB_EXPERIMENT=1 timeout 300 python3 scratch/r04_grad33_time.py 2>/dev/null | grep "23 knots matrix" >> $O/time_prev.txt
done
echo new; cat $O/time_new.txt; echo prev; cat $O/time_prev.txt
stats() { name=$1; shift; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tmp_$name -- "$@" > $O/$name.log 2>&1; find $O/tmp_$name -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${name}_kernel_stats.csv; rm -rf $O/tmp_$name; }
export KN=33; stats grad33 python3 scratch/r04_grad33_prof.py
export KN=23; stats grad23 python3 scratch/r04_grad33_prof.py
head -8 $O/grad33_kernel_stats.csv | cut -c1-220; head -8 $O/grad23_kernel_stats.csv | cut -c1-220
pmc() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- python3 scratch/r04_grad33_prof.py > $O/pmc_$name.log 2>&1; }
export KN=33; pmc w33 WRITE_SIZE; pmc f33 FETCH_SIZE
python3 - <<'PY'
import csv, glob, collections
for tag in ("w33", "f33"):
    for f in glob.glob(f"gpurun_out/r04b9/pmc_{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            if "ebwd" in k[0] or "efused" in k[0]: print(tag, k, "mean per launch (KB)", sum(v) / len(v))
PY
rm -rf $O/pmc_w33 $O/pmc_f33
