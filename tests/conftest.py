import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "vulkan-pbr-renderer_amd", "python")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
# the region kernel's self-check counters (wave-slices that had to be recomputed with direct loads: must stay 0) are on for the tests
os.environ.setdefault("PBR_MC_STATS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Large synthetic inputs are built by forked worker processes.  Forking a process that has initialised the HIP runtime is
# unsupported (and hangs under a preloaded profiler), so the `gpu` fixture builds every big input the selected tests ask for
# BEFORE it initialises the backend; the fixtures below only hand the cached arrays out.
_BIG = {}
_BIG_BUILDERS = {
    "c2_env": lambda synth, nw: synth.synth_env(1024, seed=0x5EED0001, workers=min(6, nw)),
    "c4_env": lambda synth, nw: synth.synth_env(2048, seed=0x5EED0004, workers=min(6, nw)),          # C4
    "c5_gbuffer": lambda synth, nw: synth.synth_gbuffer_temple(7680, 4320, workers=nw),              # C5
}


def _big(name, workers=1):
    if name not in _BIG:
        from pbrhip import synth
        _BIG[name] = _BIG_BUILDERS[name](synth, workers)
    return _BIG[name]


@pytest.fixture(scope="session")
def gpu(request):
    """Initialised HIP backend (libgpu_hip.so); fails loudly if the library is missing."""
    import pbrhip
    wanted = {n for item in request.session.items for n in getattr(item, "fixturenames", ()) if n in _BIG_BUILDERS}
    nw = max(1, min(16, os.cpu_count() or 1))
    for name in sorted(wanted):
        _big(name, nw)
    L = pbrhip.init()
    yield L
    L.GPU_WaitUntilIdle()
    L.GPU_Deinit()


@pytest.fixture(scope="session")
def c2_env(gpu):
    return _big("c2_env")


@pytest.fixture(scope="session")
def c4_env(gpu):
    return _big("c4_env")


@pytest.fixture(scope="session")
def c5_gbuffer(gpu):
    return _big("c5_gbuffer")
