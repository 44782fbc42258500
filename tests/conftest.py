import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "audio-diffusion_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    try:    # no tqdm monitor thread in the test process: ctypes releases the GIL during native calls, and a second Python thread could run the
            # cyclic collector — i.e. other models' destructors — while the main thread is inside a stream capture
        import tqdm
        tqdm.tqdm.monitor_interval = 0
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="module", autouse=True)
def _collect_models_at_module_end():
    """Native models (hundreds of MB of device memory each at full size, a side stream and a split-K slab buffer each) die in `__del__`; the ones
    caught in reference cycles wait for the cyclic collector, which would otherwise run at an arbitrary point of a LATER module — in the middle
    of its capture loops. Collect them where they were made."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except Exception:
        pass
