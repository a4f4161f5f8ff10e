#!/usr/bin/env python3
"""bench.py -- flow log-prob evals/s of the He (2 e-, 1-D box) waveflow model on MI355X.

One "step" = one pass of the hot path over one batch of synthetic walkers per GPU:
    log_pdf launch  ->  fp64 block sums [sum, sum^2, n] of the batch  ->  (N>1) one RCCL all-reduce of those
    3 doubles (the <E_L> site of vqmc.py:196).
Walkers shard over ranks with no exchange of coordinates (weak scaling, B per GPU fixed).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time


def cpu_share():
    """CPUs this process may use: the scheduler affinity, cut to the cgroup's quota (cpu.max) when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


# The GPU box shows every hardware thread of the host (256) under a 16-CPU quota.  Thread pools sized by the former (OpenMP of numpy / torch)
# overrun the quota, and their idle workers keep spinning for KMP_BLOCKTIME (200 ms) after a parallel region: the cgroup then stalls the whole
# process -- the launching thread included -- until the next period.  Seen as a 40 ms hole in a 6 ms timed region, in one run out of five
# (cpu.stat: nr_throttled 4, throttled_usec 9.7e6 after one bench run).  Size the pools by the share, and let idle workers sleep; set before
# numpy / torch load their runtimes.
for _k, _v in (("OMP_NUM_THREADS", str(min(cpu_share(), 16))), ("MKL_NUM_THREADS", str(min(cpu_share(), 16))), ("OMP_WAIT_POLICY", "PASSIVE"),
               ("KMP_BLOCKTIME", "0"), ("GOMP_SPINCOUNT", "0")):
    os.environ.setdefault(_k, _v)

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per eval for the He config
FLOP_PER_EVAL = 63232          # dense conditioner: 3 x 15872 + 15616
BYTES_PER_EVAL = 12            # 2 x fp32 in + 1 x fp32 out
PEAK_F32_MATRIX_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
PEAK_F16_MATRIX_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak
PEAK_HBM_GBS = 8000.0
# Executed matrix-core work per eval (He, 23 knots): per 32-walker tile and net 36 v_mfma_f32_32x32x16_f16 (two hidden blocks and one
# output block of 12 each: 3 split products x 4 K steps; dimension 0 is table-driven and issues none), x 4 nets = 144, + 6 for the
# ob_to_b product of the prior = 150 of 32768 FLOP, plus 8 v_mfma_f32_32x32x2_f32 of 4096 FLOP (input layer, 2 per net).
MFMA_F16_PER_TILE, MFMA_F32_PER_TILE = 150, 8
MFMA_FLOP_PER_EVAL = (MFMA_F16_PER_TILE * 32768 + MFMA_F32_PER_TILE * 4096) / 32
# HBM bytes of one 2^20-walker log_pdf launch: FETCH_SIZE + WRITE_SIZE (KB) of the headline kernel, READ AT RUN TIME from the committed
# summary of the separate rocprofv3 --pmc passes of this command (scratch/pmc.sh -> scratch/pmc_summary.py).  The walker read is 8 B per
# lane, not the 16 B-per-lane stream the guide's x2 FETCH_SIZE correction is calibrated on (the sum equals the 12.6 MB of algorithmic
# bytes within 20 %), so no correction is applied.
PMC_SUMMARIES = ("profiles/r04_pmc_summary.txt", "profiles/r03_pmc_summary.txt")


def pmc_traffic(kernel_substring="k_mfma<2, 1, 16, 1, false, true>"):
    """-> {"bytes", "source"} from the newest committed PMC summary that has the kernel, or None."""
    for rel in PMC_SUMMARIES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        cur, vals = None, {}
        for line in open(path):
            if line.startswith("=="):
                cur = line
                continue
            if cur and kernel_substring in cur:
                f = line.split()
                if len(f) >= 2 and f[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    vals[f[0]] = float(f[1])
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            return {"bytes": int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), "source": rel,
                    "fetch_kb": vals["FETCH_SIZE"], "write_kb": vals["WRITE_SIZE"]}
    return None
# reverse sweep of one walker, He, in the (value, gradient, Laplacian) algebra RF<2> = 4 channels: per net 2 dense 64x64 products
# (+ the 32x32 change of basis of the prior, forward and transposed, for 2 dimensions), FMA = 2 FLOP
VQMC_BWD_FLOP_PER_WALKER = 2 * (4 * 2 * 64 * 64 * 4 + 2 * 2 * 32 * 32 * 4)
VQMC_BWD_SHARE = 0.54   # of the loss + gradient time (profiles/r01h_loss_grad_kernel_stats.csv)
# Batches >= 16 384 walkers of the two-particle family take the matrix-core gradient path.  Its dominant kernel is the per-net reverse
# kernel k_ebwd<false> (three launches per call): 480 v_mfma_f32_32x32x16_f16 (288 of the sweep itself, 192 of the transposes and weight-gradient
# products it has formed itself since round 4) + 16 v_mfma_f32_32x32x2_f32 per 32-walker tile (disassembly of the linked library).  Its time is not
# measured by this line (the C call is one unit): DERIVED from its share of the call in the committed rocprofv3 run, 3 x 209.9 us of 1 006 us
# (profiles/r04_grad_tile_kernel_stats.csv, r04_grad_tile_check.txt).
GRAD_TILE_MFMA_FLOP_PER_WALKER_NET = (480 * 32768 + 16 * 4096) / 32
GRAD_TILE_BWD_SHARE = 0.626
GRAD_TILE_MIN = 16384
# Executed matrix-core work of the other two log_pdf shapes (static MFMA counts of the linked kernels x their trip counts, scratch/isa/isa_stats.py):
# 33-knot He (k_mfma<2,2,12,1>: two output blocks per net, 2 x 2 blocks of the prior's change of basis): per tile and flow net 24 + 24, prior net
# 24 + 24 + 24 = 48 * 3 + 72 = 216 f16 + 8 f32;  8-electron chain (k_mfma<8,1,8,1>: 7 output blocks per net): per net 24 + 7 * 12 = 108, + 7 * 6
# for the prior = 4 * 108 + 42 = 474 f16, + 4 * 8 f32 (K = 8 input layer: 4 steps x 2 blocks)
MFMA_PER_TILE = {"33knot": (216, 8), "c4": (474, 32)}


def mfma_roofline(key, n_walkers, ms, kernel):
    f16, f32 = MFMA_PER_TILE[key]
    flop_per_eval = (f16 * 32768 + f32 * 4096) / 32
    ach = n_walkers / (ms * 1e-3) * flop_per_eval / 1e12
    return {"bound": "mfma", "achieved": ach, "peak": PEAK_F16_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F16_MATRIX_TFLOPS, "traffic": None,
            "kernel": kernel, "kernel_ms": ms, "executed_mfma_flop_per_eval": flop_per_eval}


def he_model(kernel):
    from waveflow_amd import model_factory
    flat = np.load(os.path.join(ROOT, "tests", "golden", "he_checkpoint.npz"))["flat"]
    init_fun = model_factory.get_waveflow_model(2, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=23,
                                                n_i_internal_knots=23, i_spline_reg=0.05, i_spline_reverse_fun_tol=1e-6,
                                                n_flow_layers=3, box_size=10, xu_coord_type="mean")
    params, psi, log_pdf, _ = init_fun(0, 2)
    model = log_pdf.model
    model.set_params(flat)
    model.set_kernel(kernel)
    return model, flat


def seeded_model(D, knots, kernel="auto"):
    """Waveflow model with this build's seeded initial parameters (configs without a shipped checkpoint: SURVEY 8d C3 variant, C4)."""
    from waveflow_amd import model_factory
    init_fun = model_factory.get_waveflow_model(D, base_spline_degree=6, i_spline_degree=6, n_prior_internal_knots=knots,
                                                n_i_internal_knots=knots, i_spline_reg=0.05, n_flow_layers=3, box_size=10, xu_coord_type="mean")
    params, psi, log_pdf, _ = init_fun(0, D)
    model = log_pdf.model
    model.ensure_params(params)
    model.set_kernel(kernel)
    return model


def sorted_uniform(B, D, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(B, D, generator=g) * 2 - 1) * 10.0
    return torch.sort(x, dim=-1).values.contiguous()


def kernel_ms(model, x, n=50, warm=150):
    """Mean HIP-event time of wf_logpdf_fwd on the current stream (150 untimed launches first: the sustained-clock regime, see --warmup)."""
    import torch
    for _ in range(warm):
        model.log_pdf(x)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        model.log_pdf(x)
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


HPSI_KERNELS = "k_efused (one launch: box + 3 x (conditioner Taylor channels on the matrix cores + head) + prior + H psi)"
GRAD_KERNELS = ("k_efused<1> (forward, per-net input jets kept) + k_vqmc_seeds + 4 x k_ebwd<PRIOR> (reverse of one net on the matrix cores, weight-gradient "
                "products included) + k_egrad_reduce + k_egrad_scatter")
SAMPLE_KERNELS = "k_tsample (one lane per walker: rejection draws, mesh searches) + 4 x k_etile_cond (conditioner of a net on the matrix cores)"
GRAD_KERNELS_WAVE = "k_wave_fwd<2,RF<2>> + k_energy_out + k_vqmc_seeds + k_wave_bwd<2,RF<2>> + k_wgrad<4> + k_wgrad_reduce + k_grad_gather"


def event_ms(fn, n, warm):
    """Mean HIP-event time of fn() on the current stream over n calls, after `warm` untimed ones."""
    import torch
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in ev]))


def rqs_leg(n_elements, n=50, warm=20):
    """The RQS bijector, the one HBM-bound kernel of the path: unconstrained RQS forward, K = 32 bins."""
    import torch
    from waveflow_amd.flows import unconstrained_RQS
    K, N = 32, n_elements
    g = torch.Generator(device="cuda").manual_seed(1234)
    uw = torch.randn(N, K, device="cuda", generator=g)
    uh = torch.randn(N, K, device="cuda", generator=g)
    ud = torch.randn(N, K - 1, device="cuda", generator=g)
    x = torch.rand(N, device="cuda", generator=g) * 2.4 - 1.2
    ms = event_ms(lambda: unconstrained_RQS(x, uw, uh, ud), n, warm)
    bytes_per = (2 * K + (K - 1) + 1) * 4 + 8       # uw, uh, ud, x in; y, logabsdet out
    gbs = N * bytes_per / (ms * 1e-3) / 1e9
    return {"kernel": "k_rqs_reg<32>", "kernel_ms": ms, "elements": N, "bytes_per_element": bytes_per, "gb_per_s": gbs,
            "frac_of_hbm_peak": gbs / PEAK_HBM_GBS, "evals_per_s": N / (ms * 1e-3)}


def nsc_model():
    from waveflow_amd import flows
    L, K, H, D = 3, 5, 8, 2
    items = []
    for _ in range(L):
        items += [flows.NeuralSplineCoupling(K=K, B=3, hidden_dim=H), flows.Reverse()]
    params, log_pdf, _ = flows.Flow(flows.Serial(*items), flows.Normal())(7, D)
    params = [tuple([tuple(a * (1.5 if a.ndim == 2 else 3e4) for a in l) if l else () for l in net] for net in p) if p else () for p in params]
    m = log_pdf.model
    m.ensure_params(params)
    return m, (L, K, H, D)


def nsc_leg(B, n=30, warm=20):
    """log_pdf of the coupling-stack model (judge-added row a13): Flow(Serial((NeuralSplineCoupling, Reverse) x 3), Normal()), D = 2."""
    import torch
    m, _ = nsc_model()
    x = (torch.rand(B, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1234)) * 6.8 - 3.4).contiguous()
    ms = event_ms(lambda: m.log_pdf(x), n, warm)
    return {"kernel": "k_nsc_model<8, 8, 1, 5>", "kernel_ms": ms, "walkers": B, "evals_per_s": B / (ms * 1e-3)}


def vqmc_legs(model, n_h=20, n_g=10):
    """SURVEY 8f ranks 1-2 at BASELINE sizes: H psi of 2^20 walkers (config 5's payload per GPU) and loss + gradient of 2^17."""
    from waveflow_amd.utils import physics
    protons = physics.system_catalogue[1]["He"][0].reshape(-1)
    xb = walkers(1 << 20, 4321).cuda()
    ms_h = event_ms(lambda: model.hamiltonian(xb, protons), n_h, 10)
    xg = walkers(1 << 17, 1234).cuda()
    ms_g = event_ms(lambda: model.vqmc_loss_grad(xg, protons, -1.8), n_g, 3)
    m33 = seeded_model(2, 33, "auto")   # BASELINE's "32-bin" variant: two 32-row blocks per dimension (k_efused<2>)
    ms_h33 = event_ms(lambda: m33.hamiltonian(xb, protons), 10, 5)
    ms_g33 = event_ms(lambda: m33.vqmc_loss_grad(xg, protons, -1.8), n_g, 3)   # (round 4: k_ebwd<., 2>; the wave sweeps took 8.1 ms)
    del m33
    # the sampler (walkers ~ |psi|^2: prior columns by rejection, inverse flow) through the staged large-batch form
    seeds = iter(range(100, 200))
    ms_s = event_ms(lambda: model.sample(next(seeds), 1 << 17, exact=True), 10, 3)
    # whole training steps of 2^17 walkers (sample -> loss + gradient -> Adam -> image refresh; one hipGraph replay each): 260 steps minus 60 steps
    import time as _time
    from waveflow_amd import vqmc

    def train_seconds(n_steps):
        tr = vqmc.ModelTrainer(system_name="He", learning_rate=1e-4, box_length=10, num_epochs=n_steps, batch_size=1 << 17, log_every=10 ** 9)
        tr.save_dir = os.path.join("/tmp", "wf_bench_step_2pow17")
        tr.exact_sampler = True
        t0 = _time.perf_counter()
        tr.start_training(verbose=False)
        return _time.perf_counter() - t0
    ms_step = (train_seconds(260) - train_seconds(60)) / 200 * 1e3
    return {"sample_2pow17": {"kernels": SAMPLE_KERNELS, "ms": ms_s, "walkers": 1 << 17, "walkers_per_s": (1 << 17) / (ms_s * 1e-3)},
            "train_step_2pow17": {"ms": ms_step, "walkers": 1 << 17, "walkers_per_s": (1 << 17) / (ms_step * 1e-3),
                                  "what": "vqmc.ModelTrainer, He, batch 2^17: sampler + loss + gradient + Adam + image refresh per step, hipGraph replay "
                                          "(wall time of 260 steps minus 60 steps)"},
            **{"hpsi_2pow20": {"kernels": HPSI_KERNELS, "ms": ms_h, "walkers": 1 << 20, "walkers_per_s": (1 << 20) / (ms_h * 1e-3)},
            "hpsi_33knot_2pow20": {"kernels": "k_efused<2>", "ms": ms_h33, "walkers": 1 << 20, "walkers_per_s": (1 << 20) / (ms_h33 * 1e-3)},
            "loss_grad_2pow17": {"kernels": GRAD_KERNELS, "ms": ms_g, "walkers": 1 << 17,
                                 "walkers_per_s": (1 << 17) / (ms_g * 1e-3)},
            "loss_grad_33knot_2pow17": {"kernels": "k_efused<2> + k_ebwd<., 2> per net + k_egrad_reduce", "ms": ms_g33, "walkers": 1 << 17,
                                        "walkers_per_s": (1 << 17) / (ms_g33 * 1e-3)}}}


def extra_legs(model, flat):
    """The other measurement legs of SURVEY 8d, in the same run (rank 0, N = 1): every figure is a HIP-event kernel time."""
    import torch
    out = {}
    # C3 "32-bin" variant: 33 internal knots = 32 intervals (39 / 38 bases: two 32-row blocks per dimension), seeded parameters
    m33 = seeded_model(2, 33, "mfma")
    x = sorted_uniform(1 << 20, 2, 1234).cuda()
    ms = kernel_ms(m33, x)
    out["variant_33knot"] = {"evals_per_s": (1 << 20) / (ms * 1e-3), "kernel_ms": ms, "kernel": "k_mfma<2,2,12,1>",
                             "workload": "He, 33 knots (32 intervals), 2^20 walkers, seeded parameters",
                             "roofline": mfma_roofline("33knot", 1 << 20, ms, "k_mfma<2,2,12,1>")}
    del m33
    # C2: the reference's batch size, one call (shipped checkpoint); AUTO routes it to the wave kernel
    model.set_kernel("auto")
    x256 = sorted_uniform(256, 2, 99).cuda()
    out["c2_batch256_us"] = kernel_ms(model, x256, n=200, warm=20) * 1e3
    # C4 as BASELINE words it ("square-flow antisymmetrised psi", helpers.py:55-58): 8-electron chain, 2^18 UNSORTED walkers, seeded parameters --
    # wf_psi_antisym_fwd sorts each row and signs psi inside k_mfma; beside it log_pdf of the same walkers sorted on the host (rounds 2 - 3's leg)
    m8 = seeded_model(8, 23, "mfma")
    g8 = torch.Generator().manual_seed(1234)
    x8u = ((torch.rand(1 << 18, 8, generator=g8) * 2 - 1) * 10.0).cuda().contiguous()
    ms8 = event_ms(lambda: m8.psi_antisym(x8u), 10, 3)
    x8 = torch.sort(x8u, dim=-1).values.contiguous()
    ms8s = kernel_ms(m8, x8, n=10, warm=3)
    out["c4_d8_2pow18"] = {"evals_per_s": (1 << 18) / (ms8 * 1e-3), "kernel_ms": ms8, "kernel": "k_mfma<8,1,8,1>", "entry": "wf_psi_antisym_fwd (unsorted walkers)",
                           "logpdf_of_sorted_walkers_kernel_ms": ms8s,
                           "roofline": mfma_roofline("c4", 1 << 18, ms8, "k_mfma<8,1,8,1>")}
    # H psi beyond two particles (SURVEY 8f rank 1 for C4's model and a 4-electron chain; proton positions: an even chain inside the box): the
    # directional matrix-core path of wf_kernels_etile_dir.hip, and the wave kernel it replaces at these sizes beside it (WF_ENERGY_TILE_MIN=0)
    def hpsi_leg(model_d, xs, D):
        pr_d = np.linspace(-7.0, 7.0, D).astype(np.float32)
        ms_t = event_ms(lambda: model_d.hamiltonian(xs, pr_d), 5, 2)
        os.environ["WF_ENERGY_TILE_MIN"] = "0"
        ms_w = event_ms(lambda: model_d.hamiltonian(xs, pr_d), 2, 1)
        del os.environ["WF_ENERGY_TILE_MIN"]
        return {"kernels": f"k_edir_box<{D}> + {model_d.n_layers} x k_edir<{D}, false> + k_edir<{D}, true> (one coordinate direction at a time, Taylor triples on the matrix cores)",
                "ms": ms_t, "walkers": int(xs.shape[0]), "walkers_per_s": xs.shape[0] / (ms_t * 1e-3), "wave_kernel_ms": ms_w,
                "wave_kernel_walkers_per_s": xs.shape[0] / (ms_w * 1e-3)}
    out["hpsi_c4_2pow18"] = hpsi_leg(m8, x8, 8)
    del m8
    m4 = seeded_model(4, 23, "auto")
    out["hpsi_d4_2pow18"] = hpsi_leg(m4, sorted_uniform(1 << 18, 4, 1234).cuda(), 4)
    del m4
    torch.cuda.synchronize()
    # the secondary paths, so that the driver's record carries them (each reproducible from a file under profiles/)
    model.set_kernel("auto")
    out["rqs_2pow21"] = rqs_leg(1 << 21)
    out.update(vqmc_legs(model))
    out["nsc_2pow20"] = nsc_leg(1 << 20)
    torch.cuda.synchronize()
    return out


def cpu_baseline_1thread(flat, x_host, budget_s=5.0):
    import oracle
    om = oracle.he_model(10.0)
    n0 = 2048
    t = time.perf_counter()
    om.log_pdf(flat, x_host[:n0], threads=1)
    dt = time.perf_counter() - t
    n = int(min(x_host.shape[0], max(n0, n0 * budget_s / max(dt, 1e-6))))
    t = time.perf_counter()
    om.log_pdf(flat, x_host[:n], threads=1)
    dt = time.perf_counter() - t
    return {"value": n / dt, "unit": "evals/s", "cores": 1, "kind": "port", "sample": f"first {n} walkers, one thread, {dt:.1f} s"}


def walkers(B, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(B, 2, generator=g) * 2 - 1) * 10.0
    return torch.sort(x, dim=-1).values.contiguous()


def host_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota."""
    return cpu_share()


def cpu_baseline(flat, x_host, budget_s=10.0):
    """The oracle (port of the reference algorithm) on the host cores, on a bounded sample of the same walkers."""
    import oracle
    cores = host_cores()
    om = oracle.he_model(10.0)
    n0 = min(x_host.shape[0], 2048 * cores)
    t = time.perf_counter()
    om.log_pdf(flat, x_host[:n0], threads=cores)
    dt = time.perf_counter() - t
    n = int(min(x_host.shape[0], max(n0, n0 * budget_s / max(dt, 1e-6))))
    # whole passes over the sample until the budget (~12 s of CPU work) is used: the batch itself takes only ~4 s on 16 cores
    passes, t = 0, time.perf_counter()
    while True:
        om.log_pdf(flat, x_host[:n], threads=cores)
        passes += 1
        dt = time.perf_counter() - t
        if dt >= budget_s or passes >= 8:
            break
    dt /= passes
    return {"value": n / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of the benchmark's walkers, oracle/wf_oracle.c (fp32 restatement of the JAX reference), "
                      f"OpenMP over walkers on {cores} threads, {passes} passes of {dt:.1f} s"}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) under torch.distributed.run as a CHILD process -- this process
    has not touched the GPU and never does --, relay rank 0's JSON line to stdout, everything else to stderr, exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True)
    lines = 0
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, saw {lines}\n")
        rc = 1
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # Defaults: 300 untimed + 200 timed steps (0.15 s of GPU time).  The warm-up is long on purpose: from an idle GPU the first ~60 back-to-back
    # launches of the headline kernel take 0.29 - 0.32 ms, from ~200 on 0.259 ms (scratch/time_warm.py, profiles/r02_warmup.txt: the device
    # reaches its sustained clocks only under sustained load); a VQMC job sits in the second regime.
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--batch", type=int, default=1 << 20, help="walkers per GPU")
    ap.add_argument("--kernel", default="auto", choices=["auto", "scalar", "mfma", "wave"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary legs (33-knot variant, batch 256, D = 8, 1-thread CPU)")
    ap.add_argument("--workload", default="he_logpdf", choices=["he_logpdf", "rqs", "vqmc", "nsc"],
                    help="he_logpdf: the BASELINE metric (default).  rqs: the RQS bijector kernel alone (SURVEY row a12), an "
                         "HBM-bound elementwise op: 2 dims x `--batch` walkers, 32 bins")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)       # before anything touches the GPU: the ranks are fresh child processes
    if args.workload == "rqs":
        return main_rqs(args)
    if args.workload == "vqmc":
        return main_vqmc(args)
    if args.workload == "nsc":
        return main_nsc(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU (launch with --nproc-per-node {args.gpus}, or without a "
                         f"launcher: bench.py starts its own ranks)")
    if os.environ.get("WF_BENCH_SHARE_GPU0") == "1":   # test hook: several ranks on one GPU (with WF_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("WF_FORCE_DIST") == "1"   # WF_FORCE_DIST: exercise RCCL init with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("WF_BENCH_BACKEND", "nccl")   # "nccl" is RCCL; "gloo" only for the shared-GPU test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    model, flat = he_model(args.kernel)
    B = args.batch
    x_host = walkers(B, 1234 + rank)
    x = x_host.to(dev)
    lp = torch.empty(B, device=dev, dtype=torch.float32)

    from waveflow_amd import _lib
    import ctypes
    L = _lib.lib()
    ws = torch.empty(int(L.wf_block_sums_workspace_bytes(B)), device=dev, dtype=torch.uint8)
    # one [sum, sum^2, n] triple per step: the all-reduce of step i runs on RCCL's stream while step i+1 computes
    sums_all = torch.zeros(args.steps + args.warmup + 2, 3, device=dev, dtype=torch.float64)
    pending = []
    n_done = [0]
    stream = torch.cuda.current_stream(dev)
    sp = ctypes.c_void_p(stream.cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    # (Round 4 tried the block sums of step i on a second stream under the kernel of step i + 1: the persistent workgroups of k_mfma leave the sums'
    # blocks no room before they drain -- the sums' launches took 35 - 160 us each and the step time did not move, 0.3055 against 0.3011 ms:
    # profiles/r04_two_stream_attempt_kernel_stats.csv.  One stream.)
    def step(i=None, collective=True):
        if i is not None:
            ev0[i].record(stream)
        _lib.check(L.wf_logpdf_fwd(model._h, P(x), B, P(lp), None, None, sp), "wf_logpdf_fwd")
        if i is not None:
            ev1[i].record(stream)
        sums = sums_all[n_done[0] % sums_all.shape[0]]
        n_done[0] += 1
        _lib.check(L.wf_block_sums(P(lp), B, P(sums), P(ws), ws.numel(), sp), "wf_block_sums")
        if use_dist and collective:   # one RCCL all-reduce of 3 doubles per step, asynchronous: completed in fence(), inside the timed region
            pending.append(dist.all_reduce(sums, op=dist.ReduceOp.SUM, async_op=True))
            if len(pending) > 32:      # bound the number of outstanding collectives: a stream-side wait on one that finished long ago
                pending.pop(0).wait()

    def fence():
        for w in pending:
            w.wait()
        pending.clear()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step(0)     # untimed: first use of the events / first fence
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    last = sums_all[(n_done[0] - 1) % sums_all.shape[0]]
    mean_logp = float((last[0] / last[2]).item())
    kern_ms_ranks, solo_rates = [kern_ms], None
    if use_dist:   # every rank's own kernel time, so that an N > 1 record can be checked against the N = 1 one
        # ... and its N = 1-style rate: the same K steps (kernel + block sums) WITHOUT the collective, all ranks at once (what a rank of a
        # node under the same load does alone): scaling_efficiency = value / sum of these.  After the timed region, never part of `value`.
        fence()
        ts = time.perf_counter()
        for _ in range(args.steps):
            step(collective=False)
        torch.cuda.synchronize(dev)
        solo = B * args.steps / (time.perf_counter() - ts)
        kt = torch.tensor([kern_ms, solo], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(kt) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, kt)
        kern_ms_ranks = [float(g[0].item()) for g in gathered]
        solo_rates = [float(g[1].item()) for g in gathered]
    # AFTER the timed region (never part of `value`): the same launch once the device has reached its sustained clocks -- from an idle GPU
    # the first ~60 back-to-back launches take 0.29 - 0.32 ms, from ~200 on 15 % less (scratch/time_warm.py)
    sustained = None
    if rank == 0 and world == 1 and not args.no_extras:
        n_before = max(0, 300 - args.steps - args.warmup)
        sustained = {"kernel_ms": kernel_ms(model, x, n=100, warm=n_before), "launches_before": n_before + args.steps + args.warmup + 1, "timed_launches": 100}

    if rank == 0:
        evals = B * world * args.steps
        value = evals / dt
        k_evals_s = B / (kern_ms * 1e-3)
        waves = os.environ.get("WF_MFMA_WAVES", "16")
        tiles = os.environ.get("WF_MFMA_TILES", "1")
        mfma_tflops = k_evals_s * MFMA_FLOP_PER_EVAL / 1e12
        # PMC bytes of the committed summary: for the 2^20-walker launch of the default workgroup shape only
        traffic = pmc_traffic() if (B == (1 << 20) and args.kernel in ("auto", "mfma") and waves == "16" and tiles == "1") else None
        out = {
            "metric": "flow log-prob evals/sec", "value": value, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "1D He-like 2e- box L=10 (shipped He checkpoint: 3 IMADE layers k=6/23 knots + B-spline prior), "
                                   f"log_pdf over {B} sorted U(-L,L)^2 walkers per GPU, + fp64 block sums"
                                   + (" + 1 RCCL all-reduce of 3 doubles per step (overlapped with the next step's kernel)" if world > 1 else ""),
                       "walkers_per_gpu": B, "kernel": args.kernel, "mean_logp": mean_logp},
            # bound = the matrix cores: achieved = EXECUTED f16 / f32 MFMA FLOP per second of the dominant kernel, peak = the dense f16
            # MFMA peak of the guide.  The algorithmic fp32 figures of SURVEY 8d are the labelled extras.
            "roofline": {"bound": "mfma", "achieved": mfma_tflops, "peak": PEAK_F16_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": mfma_tflops / PEAK_F16_MATRIX_TFLOPS,
                         "traffic": traffic["bytes"] if traffic else None,
                         "traffic_source": traffic["source"] if traffic else None,
                         "kernel": {"scalar": "k_eval<2,32>", "wave": "k_wave_fwd<2,R1>"}.get(args.kernel, f"k_mfma<2,1,{waves},{tiles}>"),
                         "kernel_ms": kern_ms, "executed_mfma_flop_per_eval": MFMA_FLOP_PER_EVAL,
                         "algorithmic_flop_per_eval": FLOP_PER_EVAL, "algorithmic_tflops": k_evals_s * FLOP_PER_EVAL / 1e12,
                         "algorithmic_frac_of_f32_matrix_peak": k_evals_s * FLOP_PER_EVAL / 1e12 / PEAK_F32_MATRIX_TFLOPS,
                         "note": "achieved = executed matrix FLOP (150 v_mfma_f32_32x32x16_f16 + 8 v_mfma_f32_32x32x2_f32 per 32-walker tile; "
                                 "fp32-accurate products are three f16 MFMA products of 2-way split operands) / mean HIP-event kernel time; "
                                 "peak = dense f16 MFMA.  The kernel is bound by vector issue (activations, splines), not by the matrix "
                                 "cores: DESIGN.md 4.1.  algorithmic_* = SURVEY 8d's 63 232 FLOP per eval against the f32 matrix peak."},
            "hbm": {"achieved": k_evals_s * BYTES_PER_EVAL / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": k_evals_s * BYTES_PER_EVAL / 1e9 / PEAK_HBM_GBS, "bytes_per_eval": BYTES_PER_EVAL},
        }
        if sustained is not None:
            out["kernel_ms_sustained"] = sustained
            out["roofline"]["frac_sustained"] = B / (sustained["kernel_ms"] * 1e-3) * MFMA_FLOP_PER_EVAL / 1e12 / PEAK_F16_MATRIX_TFLOPS
        if use_dist:
            out["rccl_ranks"] = dist.get_world_size()
            out["dist_backend"] = dist.get_backend()
            out["kernel_ms_per_rank"] = kern_ms_ranks
            out["per_rank_solo_evals_per_s"] = solo_rates
            out["scaling_efficiency"] = value / sum(solo_rates)
            out["roofline_per_rank"] = [{"kernel_ms": k, "achieved": B / (k * 1e-3) * MFMA_FLOP_PER_EVAL / 1e12, "unit": "TFLOP/s",
                                         "frac": B / (k * 1e-3) * MFMA_FLOP_PER_EVAL / 1e12 / PEAK_F16_MATRIX_TFLOPS} for k in kern_ms_ranks]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flat, x_host.numpy())
            if not args.no_extras:
                out["cpu_baseline_1thread"] = cpu_baseline_1thread(flat, x_host.numpy())
        if world == 1 and not args.no_extras and args.kernel in ("auto", "mfma"):
            out.update(extra_legs(model, flat))
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def main_rqs(args):
    """Secondary line: unconstrained RQS forward, K = 32 bins, N = 2 * batch elements (single GPU, no collective)."""
    K, N = 32, 2 * args.batch
    t0 = time.perf_counter()
    leg = rqs_leg(N, n=args.steps, warm=args.warmup + 1)
    dt = time.perf_counter() - t0
    print(json.dumps({
        "metric": "RQS bijector evals/sec", "value": leg["evals_per_s"], "unit": "evals/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": leg["kernel_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "wall_s": dt,
        "config": {"workload": f"unconstrained RQS forward (neural_splines.py:16-71), K=32 bins, {N} elements (2 dims x {args.batch} walkers)"},
        "value_is": "elements / mean HIP-event kernel time (rounds 1 - 2 of this line reported elements / wall time of the loop: ~5 % lower)",
        "roofline": {"bound": "hbm", "achieved": leg["gb_per_s"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": leg["frac_of_hbm_peak"], "traffic": None,
                     "kernel": "k_rqs_reg<32>", "kernel_ms": leg["kernel_ms"], "bytes_per_eval": leg["bytes_per_element"]}}), flush=True)


def main_nsc(args):
    """Secondary line: log_pdf of Flow(Serial((NeuralSplineCoupling(K=5, B=3, hidden_dim=8), Reverse) x 3), Normal()) in two dimensions --
    the reference's defaults (neural_splines.py:244) -- over `--batch` walkers: one launch of k_nsc_model<8, 8> (single GPU)."""
    import torch
    from waveflow_amd import flows
    L, K, H, D = 3, 5, 8, 2
    items = []
    for _ in range(L):
        items += [flows.NeuralSplineCoupling(K=K, B=3, hidden_dim=H), flows.Reverse()]
    params, log_pdf, _ = flows.Flow(flows.Serial(*items), flows.Normal())(7, D)
    params = [tuple([tuple(a * (1.5 if a.ndim == 2 else 3e4) for a in l) if l else () for l in net] for net in p) if p else () for p in params]
    m = log_pdf.model
    m.ensure_params(params)
    B = args.batch
    x = (torch.rand(B, D, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1234)) * 6.8 - 3.4).contiguous()
    for _ in range(args.warmup + 1):
        m.log_pdf(x)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        m.log_pdf(x)
        b.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    bytes_per = (D + 1) * 4                      # x in, log_pdf out: the weights come through the scalar cache
    gbs = B * bytes_per / (kern_ms * 1e-3) / 1e9
    # the launch-per-half-step path of the bare layer, for comparison: (3K - 1) * 2 * 4 bytes of spline parameters per coordinate through HBM
    one = flows.NeuralSplineCoupling(K=K, B=3, hidden_dim=H)
    p1, direct_fun, _ = one(3, D)
    os.environ["WF_NSC_STAGED"] = "1"
    for _ in range(3):
        direct_fun(p1, x)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        direct_fun(p1, x)
    torch.cuda.synchronize()
    staged_ms = (time.perf_counter() - t1) / 10 * 1e3
    del os.environ["WF_NSC_STAGED"]
    print(json.dumps({
        "metric": "coupling-flow log_pdf evals/sec", "value": B * args.steps / dt, "unit": "evals/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Flow(Serial((NeuralSplineCoupling(K={K}, B=3, hidden_dim={H}), Reverse) x {L}), Normal()) log_pdf, D={D}, {B} walkers"},
        "roofline": {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                     "kernel": "k_nsc_model<8, 8, 1, 5>", "kernel_ms": kern_ms, "bytes_per_eval": bytes_per,
                     "note": "12 bytes per walker against ~9 500 vector instructions (three layers of two half-steps of 1 584: two tanh layers, two soft-maxes and a "
                             "soft-plus, the spline's own soft-max, search and logs): the launch is bounded by instruction issue (DESIGN 4.4: ~70 % of "
                             "that bound), not by HBM"},
        "one_layer_staged_path_ms": staged_ms, "one_layer_share_of_fused_ms": kern_ms / L}), flush=True)


def main_vqmc(args):
    """Secondary line: the VQMC inner loop on the shipped He checkpoint, single GPU.
    value = loss + gradient walkers/s (wf_vqmc_loss_grad, 2^17 walkers); also H psi walkers/s and whole training steps/s
    (batch 128, hipGraph replay).  cpu_baseline = the torch autograd oracle on the host cores (bounded sample)."""
    import torch
    from waveflow_amd import vqmc
    from waveflow_amd.utils import physics
    model, flat = he_model("auto")
    protons = physics.system_catalogue[1]["He"][0].reshape(-1)
    B = min(args.batch, 1 << 17)
    x = walkers(B, 1234).cuda()

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    n = max(args.steps // 5, 5)
    t_grad = timed(lambda: model.vqmc_loss_grad(x, protons, -1.8), n)
    t_h = timed(lambda: model.hamiltonian(x, protons), n)
    # H psi at the BASELINE batch (2^20 walkers: config 5's payload per GPU): the matrix-core tile path (wf_kernels_etile.hip, default from
    # 16 384 walkers on) and, beside it, the wave kernel on the same batch (WF_ENERGY_TILE_MIN=0)
    xb = walkers(1 << 20, 4321).cuda()
    for _ in range(20):
        model.hamiltonian(xb, protons)
    t_h20 = timed(lambda: model.hamiltonian(xb, protons), 20)
    os.environ["WF_ENERGY_TILE_MIN"] = "0"
    t_h20_wave = timed(lambda: model.hamiltonian(xb, protons), 3)
    del os.environ["WF_ENERGY_TILE_MIN"]
    # the dominant kernel of the gradient, timed with events
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        model.vqmc_loss_grad(x, protons, -1.8)
        b.record()
    torch.cuda.synchronize()
    step_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # whole training steps (sample -> loss + gradient -> Adam -> image refill), batch 128, one hipGraph replay each
    tr = vqmc.ModelTrainer(system_name="He", learning_rate=1e-4, box_length=10, num_epochs=4000, batch_size=128, log_every=10 ** 9)
    tr.save_dir = os.path.join("/tmp", "wf_bench_vqmc")
    tr.exact_sampler = True
    t0 = time.perf_counter()
    tr.start_training(verbose=False)
    t_train = (time.perf_counter() - t0) / 4000
    tile = B >= int(os.environ.get("WF_GRAD_TILE_MIN", GRAD_TILE_MIN)) > 0
    if tile:
        k_ms = GRAD_TILE_BWD_SHARE * step_ms / 3
        ach = B * GRAD_TILE_MFMA_FLOP_PER_WALKER_NET / (k_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F16_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F16_MATRIX_TFLOPS, "traffic": None,
                "kernel": "k_ebwd<false>", "kernel_ms": k_ms,
                "kernel_ms_is": "derived: share of the call in profiles/r04_grad_tile_kernel_stats.csv x the call's event time / 3 launches",
                "note": "executed matrix FLOP of one reverse launch (split-fp16 products, three Taylor channels, the transposes and weight-gradient "
                        "products) / its share of the call (62.6 %, three launches); the kernel runs one wave per SIMD and is bound by vector issue "
                        "and register spills, not by the matrix pipe (profiles/r04_grad_tile_pmc.txt)"}
    else:
        ach = B * VQMC_BWD_FLOP_PER_WALKER / (VQMC_BWD_SHARE * step_ms * 1e-3) / 1e12
        roof = {"bound": "valu", "achieved": ach, "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MATRIX_TFLOPS, "traffic": None,
                "kernel": "k_wave_bwd<2, RF<2>>", "kernel_ms": VQMC_BWD_SHARE * step_ms,
                "note": "the sweeps are fp32 vector (VALU) work, which this contract's bound enum does not name: achieved = algorithmic "
                        "FMA FLOP of the reverse sweep (one 4-channel sample per walker) / its share of the step (54 %, profiles/r01h_*); "
                        "peak = the fp32 vector peak"}
    out = {
        "metric": "VQMC loss+gradient walkers/sec", "value": B / t_grad, "unit": "walkers/s", "n_gpus": 1, "steps": n, "warmup": 3,
        "ms_per_step": t_grad * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"1D He (shipped checkpoint): loss_fn_efficient + gradient over {B} walkers (vqmc.py:193-221); also H psi and "
                               "whole training steps at batch 128", "walkers": B},
        "hpsi_walkers_per_s": B / t_h,
        "hpsi_2pow20": {"walkers_per_s": (1 << 20) / t_h20, "ms": t_h20 * 1e3, "kernels": HPSI_KERNELS,
                        "wave_kernel_walkers_per_s": (1 << 20) / t_h20_wave},
        "train_steps_per_s_batch128": 1.0 / t_train, "train_ms_per_step_batch128": t_train * 1e3,
        "roofline": roof,
        "kernels": GRAD_KERNELS if tile else GRAD_KERNELS_WAVE,
    }
    if not args.no_cpu_baseline:
        from oracle import energy_torch as et
        cores = host_cores()
        torch.set_num_threads(cores)
        xs = x[:256].cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        et.vqmc_loss_grad(et.he_model(torch.float64), flat, xs, protons, -1.8)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 256 / dt, "unit": "walkers/s", "cores": cores, "kind": "port",
                               "sample": "torch reverse-mode oracle (oracle/energy_torch.py, fp64) on 256 walkers of the same batch"}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
