#!/bin/bash
# per-launch durations of one staged sample() call (kernel trace in issue order)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b29
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/b29/tmp -- python3 scratch/sample_prof.py > gpurun_out/b29/prof.log 2>&1
f=$(find gpurun_out/b29/tmp -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])) for r in rows]
# the last sample() call: find the last run of 9 launches starting with k_tsample, cond<true>
idx = [i for i, (n, d, s) in enumerate(names) if "k_etile_cond<true" in n]
i0 = idx[-1] - 1
t0 = names[i0][2]
for n, d, s in names[i0:i0 + 9]:
    print(f"{(s - t0) / 1e3:8.1f} us  {d:7.1f} us  {n[:60]}")
PY
rm -rf gpurun_out/b29/tmp
