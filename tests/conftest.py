import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def kopt():
    """The calling thread's kernel switches (include/dlsa_hip.h dlsa_kernel_options) for the length of a test: kopt.set(lars_q=0),
    kopt.clear("lars_q_wgs").  What `monkeypatch.setenv("DLSA_LARS_Q", "0")` was before round 5 -- the library no longer reads the
    environment for these."""
    import ctypes
    from dlsa_amd import _lib, engine

    class Switches:
        def __init__(self):
            self.fields = {}

        def _apply(self):
            lib = _lib.load()
            if self.fields:
                c = engine.KernelOptions(**self.fields).as_c()
                _lib.check(lib.dlsa_kernel_set_options(ctypes.byref(c)))
            else:
                lib.dlsa_kernel_set_options(None)

        def set(self, **kw):
            self.fields.update(kw)
            self._apply()

        def clear(self, *names):
            for n in names:
                self.fields.pop(n, None)
            self._apply()

    k = Switches()
    yield k
    k.fields = {}
    k._apply()


def _free_device_cache():
    """Hand torch's cached blocks and the engine's per-stream scratch back to the driver.  torch.cuda.mem_get_info()
    does not count cached blocks as free, so a 100 GB test that follows another one in the same process would otherwise
    see a 'full' device (round 2: the config-3 no-copy test was skipped on the driver's box for exactly that reason)."""
    import gc
    import torch
    gc.collect()
    try:
        from dlsa_amd import engine
        engine.release_workspace()
    except Exception:
        pass
    torch.cuda.empty_cache()


@pytest.fixture(autouse=True)
def _gpu_memory_hygiene(request):
    """After every GPU test: drop what it left cached when that is more than 2 GB."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import torch
    if torch.cuda.is_available() and torch.cuda.memory_reserved() > 2e9:
        torch.cuda.synchronize()
        _free_device_cache()


def need_hbm(nbytes):
    """For the at-scale tests: free the caches, then REQUIRE the memory.  Skips only on a device that is physically too
    small (a whole MI355X has 288 GB); a full-size device that cannot provide the bytes is a failure, never a silent
    shrink of the test's size."""
    import torch
    torch.cuda.synchronize()
    _free_device_cache()
    total = torch.cuda.get_device_properties(0).total_memory
    if total < 250e9:
        pytest.skip("needs a whole MI355X (288 GB); this device has %.0f GB" % (total / 1e9))
    free, _ = torch.cuda.mem_get_info()
    assert free >= nbytes, "only %.1f GB free of %.1f GB: %.1f GB needed (memory_allocated %.1f GB)" % (
        free / 1e9, total / 1e9, nbytes / 1e9, torch.cuda.memory_allocated() / 1e9)
