#!/bin/bash
# PMC pass over scratch/bench_energy.py (kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_energy
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/a -- python3 scratch/bench_energy.py > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $out/b -- python3 scratch/bench_energy.py > $out/b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob(f"gpurun_out/pmc_energy/{d}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "k_wave_fwd" in k:
                acc[(k, r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in acc.items():
            print(k)
            for n, v in sorted(c.items()):
                print(f"   {n:28s} {sum(v)/len(v):14.0f}  (n={len(v)})")
PY
