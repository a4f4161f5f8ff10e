"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding arithmetic and the single all-reduce."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from waveflow_amd import distributed as wfd


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 256, 1 << 20, (1 << 23) + 3):
        for world in (1, 2, 3, 8):
            rows = [wfd.shard_bounds(n, r, world) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
            sizes = [hi - lo for lo, hi in rows]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        wfd.shard_bounds(10, 2, 2)


def test_moments_to_stats():
    v = np.random.default_rng(0).normal(3.0, 2.0, size=10000)
    mean, var, se = wfd.moments_to_stats([v.sum(), (v ** 2).sum(), v.size])
    assert abs(mean - v.mean()) < 1e-12 and abs(var - v.var()) < 1e-9 and abs(se - v.std() / 100) < 1e-9
    assert all(np.isnan(wfd.moments_to_stats([0, 0, 0])))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank generates the same global vector and keeps only its shard (stand-in for its walkers' log_pdf)
        v = np.random.default_rng(123).normal(-16.0, 5.0, size=n_total).astype(np.float32)
        lo, hi = wfd.shard_bounds(n_total, rank, world)
        mine = v[lo:hi].astype(np.float64)
        sums = torch.tensor([mine.sum(), (mine ** 2).sum(), float(mine.size)], dtype=torch.float64)
        wfd.all_reduce_moments(sums)
        q.put((rank, sums.tolist(), (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_expectation_matches_single_rank(world):
    n_total = 100003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    v = np.random.default_rng(123).normal(-16.0, 5.0, size=n_total).astype(np.float32).astype(np.float64)
    ref = [v.sum(), (v ** 2).sum(), float(n_total)]
    for rank, sums, rows in res:
        assert sums[2] == n_total
        assert abs(sums[0] - ref[0]) <= 1e-12 * abs(ref[0]) and abs(sums[1] - ref[1]) <= 1e-12 * ref[1]
    # every rank holds the same reduced triple (bitwise: gloo reduces in a fixed order)
    assert all(r[1] == res[0][1] for r in res)
    mean, var, se = wfd.moments_to_stats(res[0][1])
    assert abs(mean - v.mean()) < 1e-9


def test_all_reduce_is_noop_without_group():
    s = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    assert wfd.all_reduce_moments(s).tolist() == [1.0, 2.0, 3.0]


def _grad_worker(rank, world, port, n_params, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ragged shards: rank r holds 100 + 7 r walkers; its gradient is already scaled by 1 / global count
        n_local = 100 + 7 * rank
        n_global = wfd.global_count(n_local, "cpu")
        g = np.random.default_rng(50 + rank)
        grad = torch.tensor(g.normal(size=n_params).astype(np.float32) / n_global)
        e = g.normal(-2.0, 1.0, size=n_local)
        sums = torch.tensor([e.sum(), (e ** 2).sum(), float(n_local)], dtype=torch.float64)
        grad2, sums2 = wfd.all_reduce_gradient_and_moments(grad, sums)
        q.put((rank, n_global, grad2.numpy(), sums2.tolist()))
    finally:
        dist.destroy_process_group()


def test_training_step_collective_packs_gradient_and_moments():
    """the training step's one all-reduce carries [gradient, sum E, sum E^2, n]."""
    world, n_params = 2, 32588
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, n_params, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_global = sum(100 + 7 * r for r in range(world))
    want_g = np.zeros(n_params)
    want_s = np.zeros(3)
    for r in range(world):
        g = np.random.default_rng(50 + r)
        want_g += (g.normal(size=n_params).astype(np.float32) / n_global).astype(np.float64)
        e = g.normal(-2.0, 1.0, size=100 + 7 * r)
        want_s += [e.sum(), (e ** 2).sum(), e.size]
    for rank, ng, grad, sums in res:
        assert ng == n_global and grad.dtype == np.float32 and grad.shape == (n_params,)
        np.testing.assert_allclose(grad, want_g, rtol=0, atol=1e-7)
        np.testing.assert_allclose(sums, want_s, rtol=1e-12)
    assert np.array_equal(res[0][2], res[1][2])


def test_gradient_collective_is_noop_without_group():
    g, s = torch.ones(5), torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64)
    g2, s2 = wfd.all_reduce_gradient_and_moments(g, s)
    assert g2 is g and s2 is s and wfd.global_count(17, "cpu") == 17


def _c5_worker(rank, world, port, n_total, n_params, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = wfd.shard_bounds(n_total, rank, world)
        # the walkers' stand-in payload is a function of the global row index: every rank can form its shard without the others
        idx = np.arange(lo, hi, dtype=np.float64)
        e = np.sin(idx * 1e-3) - 1.8
        sums = torch.tensor([e.sum(), (e ** 2).sum(), float(hi - lo)], dtype=torch.float64)
        n_global = wfd.global_count(hi - lo, "cpu")
        grad = torch.full((n_params,), float(hi - lo) / n_global, dtype=torch.float32)   # shard weight: sums to 1 over the ranks
        grad2, sums2 = wfd.all_reduce_gradient_and_moments(grad, sums)
        q.put((rank, (lo, hi), n_global, float(grad2.min()), float(grad2.max()), sums2.tolist()))
    finally:
        dist.destroy_process_group()


def test_config5_eight_ranks_ragged_shards_of_2pow23_rows():
    """BASELINE configs[4] on CPU (gloo, world 8): 2^23 + 3 rows cut into ragged contiguous shards, one packed all-reduce of
    [gradient, sum E, sum E^2, n]; every rank ends with the global triple, the shard weights add up to 1."""
    world, n_total, n_params = 8, (1 << 23) + 3, 32588
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c5_worker, args=(r, world, port, n_total, n_params, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = [r[1] for r in res]
    assert rows[0][0] == 0 and rows[-1][1] == n_total and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    sizes = [hi - lo for lo, hi in rows]
    assert max(sizes) - min(sizes) == 1          # ragged by one row
    idx = np.arange(n_total, dtype=np.float64)
    e = np.sin(idx * 1e-3) - 1.8
    want = [e.sum(), (e ** 2).sum(), float(n_total)]
    for rank, _, n_global, gmin, gmax, sums in res:
        assert n_global == n_total and sums[2] == n_total
        assert abs(sums[0] - want[0]) <= 1e-11 * abs(want[0]) and abs(sums[1] - want[1]) <= 1e-11 * want[1]
        assert abs(gmin - 1.0) < 1e-6 and abs(gmax - 1.0) < 1e-6
    assert all(r[5] == res[0][5] for r in res)    # identical on every rank
