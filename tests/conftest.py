import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the round-end driver)")


@pytest.fixture(scope="session")
def sandbox(tmp_path_factory):
    """assets/ tree (synthetic FLAME pkl + public topology/embeddings) built from tests/golden/assets_bundle.npz."""
    from oracle import assets as A
    d = tmp_path_factory.mktemp("smirk_assets")
    A.write_sandbox(str(d))
    return str(d)


@pytest.fixture()
def in_sandbox(sandbox, monkeypatch):
    """cwd = sandbox, because the reference (and the drop-in) read assets by cwd-relative path."""
    monkeypatch.chdir(sandbox)
    return sandbox


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
