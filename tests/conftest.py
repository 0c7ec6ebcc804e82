import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


# Tuning / debugging knobs of the library (csrc/host/hip_backend.cpp reads them at upload and render time).  A value
# left in the caller's environment (a sweep, an A/B run) must not reach the tests: they pin the library's defaults.
# PTR_TEST_VARIANT=<name> is the one deliberate exception: it runs the suite against variants/libptr_<name>.so (an A/B build of
# tools/build_variant.sh), so a kernel variant is parity-checked before it is timed.
_KEEP = {"PTR_TEST_VERBOSE", "PTR_TEST_VARIANT"}
for _name in [k for k in os.environ if k.startswith("PTR_") and k not in _KEEP]:
    del os.environ[_name]
if os.environ.get("PTR_TEST_VARIANT"):
    os.environ["PTR_HIP_LIBRARY"] = os.path.join(ROOT, "variants", "libptr_%s.so" % os.environ["PTR_TEST_VARIANT"])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """Bring the HIP library, the CLI and the oracle up to date once per session (incremental make; no GPU needed), so
    the tests never run against a stale binary."""
    import __graft_entry__ as entry

    entry.ensure_built()
    from scenes.gen_assets import ensure_assets

    ensure_assets()
