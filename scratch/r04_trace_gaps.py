"""kernel_trace.csv -> per-step timeline: for the last steps of the run, kernel durations, gaps between consecutive kernels, step period"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# a step starts at every k_tsample<0...> (first kernel of the sampler)
starts = [i for i, r in enumerate(rows) if "k_tsample<0" in r[2]]
if len(starts) < 12: print("few steps", len(starts)); sys.exit()
sel = starts[-11:]
per = []
for a, b in zip(sel[:-1], sel[1:]):
    ks = rows[a:b]
    busy = sum(e - s for s, e, _ in ks)
    period = rows[b][0] - rows[a][0]
    per.append((period, busy, len(ks)))
print("steps analysed", len(per))
print("period us  median %.1f" % (sorted(p for p, _, _ in per)[len(per) // 2] / 1e3), " busy us median %.1f" % (sorted(b for _, b, _ in per)[len(per) // 2] / 1e3), " kernels per step", per[0][2])
a, b = sel[-2], sel[-1]
prev_end = None
for s, e, n in rows[a:b]:
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("  %8.1f us  gap %6.1f  %s" % ((e - s) / 1e3, gap, n[:90]))
    prev_end = e
