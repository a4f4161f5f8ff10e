#!/usr/bin/env python3
"""Round 4, VERDICT item 1a: which product of k_mfma moves its agreement with the fp32 oracle?  CPU only.

The He model (shipped checkpoint) is evaluated in fp64 NumPy with ONE site at a time replaced by an emulation of what the matrix-core
kernel computes there (wf_mfma_impl.h, wf_model.cpp: describe_mfma_image, k_fold_bias):

  H2   pre-activation of the second hidden layer:  b1' + sum_k W1'_k r1_k   (W' = -2 c W: tanh = 1 - 2 r folded into the weights,
       b' = c b + c sum_k W_k; operands as fp16 pairs hi + lo; products hi*hi + hi*lo + lo*hi; fp32 accumulation per MFMA K step)
  O    raw head outputs:                           b2' + sum_k W2'_k r2_k   (same)
  OB   the prior's change of basis c = (o * keep) @ ob_to_b                 (per-walker power-of-two scale, same three products)

and, next to it, by the reference's own arithmetic at that site (fp32 tanh values, sequential fp32 dot product, as jnp / the C oracle).
Reported per site and variant, on the well-conditioned subset of C3's walkers (parity_stats: cond > 0.05, |log_pdf| > 1):
  * the share of walkers whose log_pdf moves by more than 1e-5 relative against the all-fp64 evaluation,
  * the 99th percentile and the maximum of that relative change.
Variants of the kernel emulation: '3prod' (shipped), '4prod' (+ lo*lo), 'f32ops' (unsplit fp32 operands in the folded form: what the fold
alone costs), 'unfold' (split operands, but W tanh with tanh = 1 - 2r formed in fp32 before the split: no cancellation against the bias).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import parity_stats  # noqa: E402

LOG2E = 1.4426950408889634074
f32 = np.float32


def split16(x32):
    hi = x32.astype(np.float16)
    lo = (x32 - hi.astype(f32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def mfma_chain(acc32, A_hi, A_lo, B_hi, B_lo, variant):
    """acc32 [B, N] fp32; A_* [K, N] (weights), B_* [B, K] (activations), fp64 holding fp16 values; K steps of 16; fp32 rounding after every
    MFMA instruction (exact products, exact sum inside one instruction: the most favourable model of the hardware's accumulation)."""
    K = A_hi.shape[0]
    acc = acc32.astype(f32)
    for k0 in range(0, K, 16):
        sl = slice(k0, k0 + 16)
        prods = []
        if variant == "4prod":
            prods.append(B_lo[:, sl] @ A_lo[sl])
        prods += [B_hi[:, sl] @ A_lo[sl], B_lo[:, sl] @ A_hi[sl], B_hi[:, sl] @ A_hi[sl]]
        for p in prods:
            acc = (acc.astype(np.float64) + p).astype(f32)
    return acc


def kernel_dense(r32, W, b, c_in, variant):
    """pre-activation of a layer behind a tanh layer as k_mfma computes it.  r32 [B, K] fp32 = 1 / (2^xs + 1); W [K, N] fp32 (masked), b [N]."""
    Wp = (np.float64(-2.0 * c_in) * W.astype(np.float64)).astype(f32)          # k_pack: (float)(scale * (double)flat)
    Wh, Wl = split16(Wp)
    colsum = np.zeros(W.shape[1], f32)
    for k in range(W.shape[0]):                                               # k_fold_bias: fp32 sum of (hi + lo) in k order
        colsum = (colsum + (Wh[k] + Wl[k]).astype(f32)).astype(f32)
    bp = ((np.float64(c_in) * b.astype(np.float64)).astype(f32) + f32(-0.5) * colsum).astype(f32)
    acc0 = np.broadcast_to(bp, (r32.shape[0], W.shape[1])).astype(f32)
    if variant == "f32ops":       # unsplit operands, exact products, one rounding per K step of 16 (fold cancellation only)
        acc = acc0
        for k0 in range(0, W.shape[0], 16):
            acc = (acc.astype(np.float64) + r32[:, k0:k0 + 16].astype(np.float64) @ Wp[k0:k0 + 16].astype(np.float64)).astype(f32)
        return acc
    if variant == "unfold":       # tanh = 1 - 2r in fp32, weights c W, bias c b: split operands, three products
        t32 = (f32(1.0) - f32(2.0) * r32).astype(f32)
        Wu = (np.float64(c_in) * W.astype(np.float64)).astype(f32)
        Wuh, Wul = split16(Wu)
        th, tl = split16(t32)
        bu = (np.float64(c_in) * b.astype(np.float64)).astype(f32)
        return mfma_chain(np.broadcast_to(bu, acc0.shape), Wuh, Wul, th, tl, "3prod")
    if variant in ("center", "center4"):   # r' = r - 0.5 in fp32 (tanh = -2 r'): no constant to cancel; center4: weights as three fp16 pieces
        rc = (r32 - f32(0.5)).astype(f32)
        ch, cl = split16(rc)
        bu = (np.float64(c_in) * b.astype(np.float64)).astype(f32)
        acc = mfma_chain(np.broadcast_to(bu, acc0.shape), Wh, Wl, ch, cl, "3prod")
        if variant == "center4":
            Wm = (Wp - Wh.astype(f32) - Wl.astype(f32)).astype(np.float16).astype(np.float64)
            for k0 in range(0, W.shape[0], 16):
                acc = (acc.astype(np.float64) + ch[:, k0:k0 + 16] @ Wm[k0:k0 + 16]).astype(f32)
        return acc
    rh, rl = split16(r32)
    return mfma_chain(acc0, Wh, Wl, rh, rl, variant)


def ref32_dense(t32, W, b):
    """the reference at that site: fp32 tanh values, fp32 dot product (sequential accumulation), + bias"""
    acc = np.zeros((t32.shape[0], W.shape[1]), f32)
    for k in range(W.shape[0]):
        acc = (acc + (t32[:, k:k + 1] * W[k][None, :]).astype(f32)).astype(f32)
    return (acc + b[None, :]).astype(f32)


class He:
    def __init__(self):
        self.D, self.H, self.k, self.L, self.reg = 2, 64, 6, 10.0, 0.05
        self.I = np.asarray(oracle.table(oracle.KIND_I, 6, 23), dtype=f32).astype(np.float64)
        Bt, OB, b2o, o2b = oracle.ortho_b(6, 23)
        self.OB = np.asarray(OB, dtype=f32).astype(np.float64)
        self.o2b = np.asarray(o2b, dtype=f32)
        self.nbi, self.nbp = self.I.shape[1], self.OB.shape[1]

    def nets(self, flat):
        out, off = [], 0
        for n_out in [self.nbi] * 3 + [self.nbp]:
            D, H = self.D, self.H
            sizes = [D * H, H, H * H, H, H * n_out * D, n_out * D, D * n_out]
            parts = []
            for s in sizes:
                parts.append(flat[off:off + s]); off += s
            out.append(dict(W0=parts[0].reshape(D, H), b0=parts[1], W1=parts[2].reshape(H, H), b1=parts[3],
                            W2=parts[4].reshape(H, n_out * D), b2=parts[5], n_out=n_out))
        assert off == flat.size
        return out

    @staticmethod
    def lerp(tab, x):
        n = tab.shape[-1] - 1
        xs = x * n
        il = np.floor(xs).astype(np.int64); ir = np.ceil(xs).astype(np.int64)
        ilc = np.clip(np.where(il < 0, il + n + 1, il), 0, n); irc = np.clip(np.where(ir < 0, ir + n + 1, ir), 0, n)
        yl, yr = tab[:, ilc].T, tab[:, irc].T
        return yl + (yr - yl) * n * (x - il / n)[:, None]

    def conditioner(self, net, s, site, mode, head_c, tag):
        """-> raw outputs of dimension 1 [B, n_out] in fp64.  s: the conditioner's only live input (dimension 0), fp64.
        site in {None, 'H2', 'O'}: which product is replaced; mode: 'ref32' or a kernel variant."""
        W0, b0, W1, b1 = (net[k] for k in ("W0", "b0", "W1", "b1"))
        W2 = net["W2"][:, 1::2]            # columns j * D + 1
        b2 = net["b2"][1::2]
        z1 = s[:, None] * W0[0].astype(np.float64)[None, :] + b0.astype(np.float64)[None, :]
        t1 = np.tanh(z1)
        c1 = 2.0 * LOG2E
        site = site or ()
        if "H2" in site and mode == "ref32":
            z2 = ref32_dense(t1.astype(f32), W1, b1).astype(np.float64)
        elif "H2" in site:
            r1 = (1.0 / (np.exp2(c1 * z1) + 1.0)).astype(f32)
            z2 = kernel_dense(r1, W1, b1, c1, self._h2_mode or mode).astype(np.float64) / c1
        else:
            z2 = t1 @ W1.astype(np.float64) + b1.astype(np.float64)
        t2 = np.tanh(z2)
        if "O" in site and mode == "ref32":
            o = ref32_dense(t2.astype(f32), W2, b2).astype(np.float64)
        elif "O" in site:
            r2 = (1.0 / (np.exp2(c1 * z2) + 1.0)).astype(f32)
            o = kernel_dense(r2, W2, b2, head_c, mode).astype(np.float64) / head_c
        else:
            o = t2 @ W2.astype(np.float64) + b2.astype(np.float64)
        return o

    def log_pdf(self, flat, x, site=(), mode=None, which_nets=(0, 1, 2, 3)):
        nets = self.nets(flat)
        L, tol, k = self.L, 1e-7, self.k
        x = x.astype(np.float64)
        mean = x.mean(-1)
        l = mean - x[:, 0]; w = x[:, 1] - x[:, 0]
        u = np.stack([(x[:, 1] - x[:, 0]) / (2 * L + tol), (mean + L - l) / (2 * L - w + tol)], -1)
        ld = -np.log(2 * L + tol) - np.log(2 * L - w + tol)
        nb = self.nbi
        scale = np.ones(nb)
        for i in range(k):
            scale[i + 1] *= (i + 1) / k
            scale[nb - (i + 2)] *= (i + 1) / k
        keep = np.ones(nb); keep[0] = 0; keep[-1] = 0
        for n in range(3):
            net = nets[n]
            st = site if n in which_nets else ()
            # dimension 0: bias only
            o0 = net["b2"][0::2].astype(np.float64)[None, :].repeat(x.shape[0], 0)
            o1 = self.conditioner(net, u[:, 0], st, mode, -LOG2E, f"flow{n}")
            ys, dys = [], []
            for d, o in ((0, o0), (1, o1)):
                p = 1.0 / (1.0 + np.exp(-o))
                p = p / p.sum(-1, keepdims=True) + self.reg
                p = p * scale
                p = p / p.sum(-1, keepdims=True)
                p = p * keep
                p = p / p.sum(-1, keepdims=True)
                ys.append((p * self.lerp(self.I[0], u[:, d])).sum(-1))
                dys.append((p * self.lerp(self.I[1], u[:, d])).sum(-1))
            ld = ld + np.log(dys[0] + 1e-7) + np.log(dys[1] + 1e-7)
            u = np.stack([ys[1], ys[0]], -1)
        net = nets[3]
        st = site if 3 in which_nets else ()
        nbp = self.nbp
        keepp = np.ones(nbp); keepp[0] = 0; keepp[-1] = 0
        o0 = net["b2"][0::2].astype(np.float64)[None, :].repeat(x.shape[0], 0)
        o1 = self.conditioner(net, u[:, 0], st, mode, 1.0, "prior")
        uc = np.clip(u, 0.0, 1.0)
        lp = ld
        for d, o in ((0, o0), (1, o1)):
            wv = o / o.sum(-1, keepdims=True)
            wv = wv * keepp
            wv = wv / np.sqrt((wv ** 2).sum(-1, keepdims=True))
            if "OB" in st and d == 1 and mode == "ref32":
                c = ref32_dense(wv.astype(f32), self.o2b, np.zeros(nbp, f32)).astype(np.float64)
            elif "OB" in st and d == 1:
                ok = (o * keepp).astype(f32)
                amax = np.abs(ok).max(-1)
                ex = np.where(amax > 0, np.frexp(amax)[1], 0)
                oks = np.ldexp(ok, -ex[:, None]).astype(f32)
                oh, ol = split16(oks)
                Ah, Al = split16(self.o2b)
                c = mfma_chain(np.zeros((x.shape[0], nbp), f32), Ah, Al, oh, ol, mode).astype(np.float64) * np.sign(o.sum(-1))[:, None]
            else:
                c = wv @ self.o2b.astype(np.float64)
            c = c / np.sqrt((c ** 2).sum(-1, keepdims=True))
            ps = (c * self.lerp(self.OB[0], uc[:, d])).sum(-1)
            lp = lp + np.log(ps ** 2 * (0.5 if d == 0 else 1.0) + 1e-7)
        return lp


def _mixed(self, flat, x, h2_mode):
    orig = self.conditioner

    def cond(net, s, site, mode, head_c, tag):
        if tag.startswith("flow"):
            # two passes are not needed: kernel_dense is called per site with its own variant
            self._h2_mode = h2_mode
        else:
            self._h2_mode = "3prod"
        return orig(net, s, site, mode, head_c, tag)
    self.conditioner = cond
    try:
        return self.log_pdf(flat, x, ("H2", "O", "OB"), "3prod")
    finally:
        self.conditioner = orig
        self._h2_mode = None


He.log_pdf_mixed = _mixed
He._h2_mode = None


def main():
    import torch
    B = int(os.environ.get("WF_ATTR_B", 1 << 18))
    g = torch.Generator().manual_seed(1234)
    x = (torch.rand(1 << 20, 2, generator=g) * 2 - 1) * 10.0
    x = torch.sort(x, dim=-1).values.numpy()[:B]
    flat = np.load(os.path.join(ROOT, "tests", "golden", "he_checkpoint.npz"))["flat"]
    om = oracle.he_model(10.0)
    truth, cond, _ = om.log_pdf_cond(flat, x, threads=8, f64=True)
    o32 = om.log_pdf(flat, x, threads=8)
    sel = (cond > parity_stats.COND_MIN) & (np.abs(truth) > parity_stats.LOGP_MIN)
    xs = x[sel]
    print(f"{B} walkers, {int(sel.sum())} well conditioned; fp32 oracle vs fp64 oracle on them: outside 1e-5 rel "
          f"{float((np.abs(o32[sel] - truth[sel]) > 1e-5 * np.abs(truth[sel])).mean()):.4f}")
    he = He()
    base = he.log_pdf(flat, xs)
    chk = np.abs(base - truth[sel].astype(np.float64)) / np.abs(truth[sel])
    print(f"fp64 NumPy restatement vs fp64 C oracle (rounded to fp32): max rel {chk.max():.2e}")
    rows = []
    t32, o32s = truth[sel].astype(np.float64), o32[sel].astype(np.float64)
    if os.environ.get("WF_ATTR_ALL", "1") == "1":
        # every site at once: the kernel's matrix arithmetic (all else fp64) against the fp32 oracle -- what the direct rate sees of it
        for mode in ("3prod", "4prod", "center", "center4"):
            v = he.log_pdf(flat, xs, ("H2", "O", "OB"), mode if mode in ("3prod", "4prod") else "3prod")
            if mode in ("center", "center4"):   # H2 of the flow nets in the centred form, everything else as shipped
                class _M(He):
                    pass
                v = he.log_pdf_mixed(flat, xs, mode)
            print(f"  all sites, {mode:8s}: outside 1e-5 rel of fp64 {float((np.abs(v - base) > 1e-5 * np.abs(base)).mean()):.4f}; "
                  f"direct vs fp32 oracle {float((np.abs(v - o32s) <= 1e-5 * np.abs(o32s)).mean()):.4f}  (fp32 oracle vs fp64: "
                  f"{float((np.abs(o32s - t32) <= 1e-5 * np.abs(t32)).mean()):.4f})", flush=True)
    for site, nets, label in (("H2", (0, 1, 2), "H2 flow nets"), ("O", (0, 1, 2), "O  flow nets"), ("H2", (3,), "H2 prior"), ("O", (3,), "O  prior"),
                              ("OB", (3,), "ob_to_b")):
        for mode in ("ref32", "3prod", "4prod") + (("f32ops", "unfold", "center", "center4") if site != "OB" else ()):
            v = he.log_pdf(flat, xs, (site,), mode, nets)
            rel = np.abs(v - base) / np.abs(base)
            rows.append((label, mode, float((rel > 1e-5).mean()), float(np.quantile(rel, 0.99)), float(rel.max()), float(np.median(rel))))
            print(f"  {label:14s} {mode:7s}: moved > 1e-5 rel {rows[-1][2]:.4f}   median {rows[-1][5]:.2e}  p99 {rows[-1][3]:.2e}   max {rows[-1][4]:.2e}", flush=True)


if __name__ == "__main__":
    main()
