import numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
np.set_printoptions(linewidth=220, precision=6, suppress=True)
B = 1 << 20
dbg = torch.zeros(B * 2 * 64, device="cuda")
os.environ["WF_DBG_PTR"] = str(dbg.data_ptr())
import bench
m, flat = bench.he_model("scalar")
x = bench.walkers(B, 1234).cuda()
ls = m.log_pdf(x)
m.set_kernel("mfma")
shown = 0
for run in range(12):
    dbg.zero_()
    lm = m.log_pdf(x)
    bad = ((lm - ls).abs() > 0.05).nonzero().flatten()
    tiles = sorted(set((bad // 32).tolist()))
    for t in tiles:
        if shown >= 4: break
        g = dbg.view(B, 2, 64)[t * 32:(t + 1) * 32].cpu().numpy().astype(np.float64)   # [32 lanes][2 halves][64]
        a, b, v = g[:, :, 0:16], g[:, :, 16:32], g[:, :, 32:48]
        A, Bv, part, res, tt = g[:, :, 48], g[:, :, 49], g[:, :, 50], g[:, :, 51], g[:, :, 52]
        A_chk = (v * a).sum(-1); B_chk = (v * b).sum(-1)
        part_chk = A + (Bv - A) * tt
        res_chk = part[:, 0] + part[:, 1]
        badl = ((lm[t*32:(t+1)*32] - ls[t*32:(t+1)*32]).abs() > 0.05).nonzero().flatten().tolist()
        print("tile", t, "bad walkers", badl)
        print("  |A - sum v*a| h0", np.abs(A - A_chk)[:, 0].round(5)); print("  |A - sum v*a| h1", np.abs(A - A_chk)[:, 1].round(5))
        print("  |B - sum v*b| h0", np.abs(Bv - B_chk)[:, 0].round(5)); print("  |B - sum v*b| h1", np.abs(Bv - B_chk)[:, 1].round(5))
        print("  |part - chk| max", np.abs(part - part_chk).max(), " |res - (p0+p1)| h0", np.abs(res[:, 0] - res_chk).round(5), "h1", np.abs(res[:, 1] - res_chk).round(5))
        # v constant across lanes?
        print("  v spread over lanes (max-min) h0/h1:", (v[:, 0].max(0) - v[:, 0].min(0)).max(), (v[:, 1].max(0) - v[:, 1].min(0)).max())
        # a row sanity: compare lane's a with another lane having same il? skip; print a for one bad & one good lane
        if badl:
            print("  a bad lane", badl[0], a[badl[0], 0].round(4), a[badl[0], 1].round(4))
        shown += 1
