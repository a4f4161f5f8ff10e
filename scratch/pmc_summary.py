import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for f in glob.glob(os.path.join(root, "*", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in agg:
    if not any(t in k for t in os.environ.get("WF_PMC_KERNELS", "k_mfma,k_eval").split(",")):
        continue
    print("==", k[:90])
    if dur[k]:
        d = sorted(dur[k]); print("   duration us (profiled): median %.1f min %.1f n=%d" % (d[len(d)//2], d[0], len(d)))
    for c in sorted(agg[k]):
        # sum over dimension instances per dispatch, then average over dispatches
        per = defaultdict(float)
        for did, v in agg[k][c]:
            per[did] += v
        vals = list(per.values())
        print("   %-32s %16.0f  (avg per dispatch, %d dispatches)" % (c, sum(vals) / len(vals), len(vals)))
