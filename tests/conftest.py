import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """make sure libhwgat_hip.so matches the sources: build.py is a no-op when the digest stamp is
    current (the normal case: the .so travels with the tree) and recompiles with hipcc otherwise.  If that
    fails, `_lib.lib()` refuses the stale library (digest stamp check), so no test can pass on old code."""
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("hwgat_build", os.path.join(ROOT, "sl-hwgat_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    except Exception as exc:                                   # noqa: BLE001
        sys.stderr.write(f"[conftest] could not (re)build libhwgat_hip.so: {exc}\n")
