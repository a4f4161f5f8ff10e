"""Host-side helpers that need no GPU: the inversion count of coordinates.py:41-51, bench.py's PMC-summary reader and its rank bookkeeping."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def test_inversion_count_matches_the_pairwise_definition():
    """get_num_inversion_count (coordinates.py:17-51: insertion count per row) == #{i < j : c_i > c_j}, ties counting nothing."""
    from waveflow_amd.utils.coordinates import get_num_inversion_count
    g = np.random.default_rng(5)
    for D in (2, 3, 8):
        c = g.normal(size=(500, D)).astype(np.float32)
        c[::7, 0] = c[::7, D - 1]                       # ties
        want = np.zeros(500, np.int64)
        for i in range(D):
            for j in range(i + 1, D):
                want += c[:, i] > c[:, j]
        got = get_num_inversion_count(c)
        assert got.dtype == np.int64 and np.array_equal(got, want)
    assert get_num_inversion_count(np.array([[3.0, 2.0, 1.0]]))[0] == 3 and get_num_inversion_count(np.array([[1.0, 1.0, 1.0]]))[0] == 0


def test_bench_reads_the_traffic_from_the_committed_pmc_summary(tmp_path, monkeypatch):
    """roofline.traffic is FETCH_SIZE + WRITE_SIZE (KB) of the headline kernel in the newest committed summary, read at run time."""
    sys.path.insert(0, ROOT)
    import bench
    t = bench.pmc_traffic()
    assert t is not None and t["source"] in bench.PMC_SUMMARIES and os.path.exists(os.path.join(ROOT, t["source"]))
    assert t["bytes"] == int((t["fetch_kb"] + t["write_kb"]) * 1024) and 10e6 < t["bytes"] < 30e6      # ~12.6 MB algorithmic at 2^20 walkers
    # a synthetic summary: the numbers of the matching kernel's section, nothing else
    p = tmp_path / "profiles"
    p.mkdir()
    (p / "s.txt").write_text("== void wf::other<1>(int)\n   FETCH_SIZE   5  (avg)\n   WRITE_SIZE   6  (avg)\n"
                             "== void wf::mfma::k_mfma<2, 1, 16, 1, false, true>(wf::MfmaDev, int)\n   FETCH_SIZE   100  (avg)\n   SQ_WAVES 4096 (avg)\n   WRITE_SIZE   28  (avg)\n")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "PMC_SUMMARIES", ("profiles/missing.txt", "profiles/s.txt"))
    t = bench.pmc_traffic()
    assert t == {"bytes": 128 * 1024, "source": "profiles/s.txt", "fetch_kb": 100.0, "write_kb": 28.0}


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py --gpus 1 inside a WORLD_SIZE=4 environment exits before it touches a GPU (one rank per GPU: the line's n_gpus is never a guess)."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stdout + r.stderr) and not any(ln.startswith('{"metric"') for ln in r.stdout.splitlines())
