import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


def _graph(oracle, name, dangling):
    from fora_amd import synth
    n, m, seed = synth.PRESETS[name]
    src, dst = synth.rmat_graph(n, m, seed, dangling)
    return oracle.Graph.from_edges(n, m, src, dst)


@pytest.fixture(scope="session")
def tiny(oracle):
    return _graph(oracle, "tiny", "none")


@pytest.fixture(scope="session")
def tiny_dangling(oracle):
    return _graph(oracle, "tiny", "rmat")


@pytest.fixture(scope="session")
def small(oracle):
    return _graph(oracle, "small", "none")


@pytest.fixture(scope="session")
def small_dangling(oracle):
    return _graph(oracle, "small", "rmat")


@pytest.fixture(scope="session")
def engine():
    """The HIP engine through the C ABI.  No fallback: fails if the library or GPU is missing."""
    import fora_amd
    e = fora_amd.Engine(0)
    arch, cus, hbm = e.device_info()
    assert arch.startswith("gfx950"), arch
    yield e
    e.close()


@pytest.fixture(scope="session")
def engine_test():
    """The build with the schedule experiments compiled in (libfora_hip_test.so, -DFORA_TEST_PATHS=1): threshold rounds
    and bounded deferral lost to the plain schedule and are not in the product library; their
    equivalence with the twin is still checked, against this build."""
    import fora_amd
    from fora_amd import capi
    e = fora_amd.Engine(0, lib=capi.TEST_LIB)
    assert e.get_option("test_paths") == 1
    yield e
    e.close()


def pick_sources(g, count, seed, want_dangling=False):
    rng = np.random.Generator(np.random.PCG64(seed))
    deg = g.deg
    pool = np.flatnonzero(deg == 0) if want_dangling else np.flatnonzero(deg > 0)
    if pool.size == 0:
        return np.zeros(0, dtype=np.int32)
    return rng.choice(pool, size=min(count, pool.size), replace=False).astype(np.int32)
