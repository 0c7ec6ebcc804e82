import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """Build the HIP library, the CLI and the oracle once per session if they are missing (no GPU needed)."""
    import __graft_entry__ as entry

    entry.ensure_built()
    from scenes.gen_assets import ensure_assets

    ensure_assets()
