import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
import bench
m, flat = bench.he_model("scalar")
B = 1 << 20
x = bench.walkers(B, 1234).cuda()
ls, us, idxs = m.log_pdf(x, return_sample=True, return_bin_idx=True)
m.set_kernel("mfma")
from collections import Counter
cnt = Counter(); lane_cnt = Counter()
for run in range(6):
    lm, um, idxm = m.log_pdf(x, return_sample=True, return_bin_idx=True)
    bad = ((lm - ls).abs() > 0.05).nonzero().flatten()
    tiles = sorted(set((bad // 32).tolist()))
    for t in tiles:
        sl = slice(t * 32, t * 32 + 32)
        di = (idxm[sl] - idxs[sl]).abs().cpu().numpy()   # [32, 4, 2, 2]
        first = None
        for l in range(4):
            for d in range(2):
                if (di[:, l, d, 0] > 1).any() and first is None:
                    first = (l, d)
        nbadl = int(((lm[sl] - ls[sl]).abs() > 0.05).sum())
        cnt[first] += 1
        lanes = tuple(np.nonzero(di[:, first[0], first[1], 0] > 1)[0].tolist()) if first else ()
        lane_cnt[len(lanes)] += 1
        if run == 0 and len(tiles) < 40: print("tile", t, "first deviating input (layer, dim):", first, "bad walkers in tile", nbadl, "lanes", lanes[:8], "wave-in-wg", t % 8 if False else None)
print("first-deviation histogram:", dict(cnt)); print("n lanes affected hist:", dict(lane_cnt))
