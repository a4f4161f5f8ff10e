import os
import sys


def _cpu_share():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


# (bench.py, top: thread pools sized by the host's 256 hardware threads overrun the GPU box's 16-CPU quota and the cgroup stalls the process)
# (the pools only: a passive wait policy triples the CPU suite's time -- the torch autograd oracle is thousands of small parallel regions)
for _k, _v in (("OMP_NUM_THREADS", str(min(_cpu_share(), 16))), ("MKL_NUM_THREADS", str(min(_cpu_share(), 16)))):
    os.environ.setdefault(_k, _v)

import numpy as np  # noqa: E402
import pytest  # noqa: E402

os.environ.setdefault("WF_POISON", "1")   # NaN-fill fresh device allocations of the library: unwritten reads cannot pass by luck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library():
    """Built files are git-ignored but travel to the GPU box: always run the (incremental, mtime + flag aware) build so that the
    suite never runs against a library older than the checked-in sources.  hipcc cross-compiles without a GPU."""
    from waveflow_amd import build as wf_build
    wf_build.build()


@pytest.fixture(scope="session")
def golden():
    return {n[:-4]: np.load(os.path.join(GOLDEN, n)) for n in os.listdir(GOLDEN) if n.endswith(".npz")}


@pytest.fixture(scope="session")
def he_flat(golden):
    return golden["he_checkpoint"]["flat"]


def he_grid(L=10.0, ngrid=100):
    """helpers.py:52-58: the 100x100 grid the reference evaluates psi on."""
    y, x = np.meshgrid(np.linspace(-L, L, ngrid), np.linspace(-L, L, ngrid))
    coords = np.stack([x, y], axis=-1).reshape(-1, 2)
    inv = (coords[:, 0] > coords[:, 1]).astype(np.int64)
    return coords, np.sort(coords, axis=-1), (-1.0) ** inv


def sorted_walkers(B, D, L, seed):
    """SURVEY §8d C3/C4 inputs: U(-L, L)^D sorted ascending per row, fp32."""
    g = np.random.default_rng(seed)
    return np.sort(g.uniform(-L, L, size=(B, D)).astype(np.float32), axis=-1)
