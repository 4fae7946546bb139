import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libqr_ref.so (partial reference build)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build(ref=os.path.isdir("/root/reference/src"))
    return oracle


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_runtime_first():
    """torch bundles its own HIP runtime.  When libqr_hip.so (linked against
    /opt/rocm's) initialises first, a later torch.cuda initialisation in the same
    process can report "No HIP GPUs are available"; initialising torch first is
    harmless on a CPU-only box."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session", autouse=True)
def _no_readback_was_read_twice():
    """the polled read-backs validate themselves and re-read what does not fit -- a
    retry must not be absorbed silently.  Every context's count is added up when it is closed
    (quickrank_amd/_capi.py); the session fails if any read-back was ever read twice."""
    yield
    mod = sys.modules.get("quickrank_amd._capi")
    if mod is not None:
        for ctx in list(mod._LIVE or ()):
            ctx.close()
        assert mod.READBACK_RETRIES == 0, f"{mod.READBACK_RETRIES} re-reads of polled read-backs (qr_readback_retries)"
