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


@pytest.fixture(autouse=True)
def _no_silent_split_fp16_overflow(request):
    """Every GPU test doubles as a check of the library's always-on range flag (include/smirk_hip.h smirk_range_flag_peek): after the test, with the device
    synchronised, no kernel may have stored a value the split-fp16 format cannot carry.  Tests that overflow on purpose clear the flag themselves."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import torch
    if not torch.cuda.is_available():
        return
    from smirk_amd import _lib as L
    if L._LIB is None:
        return
    torch.cuda.synchronize()
    tripped = L._LIB.smirk_range_flag_peek()
    L._LIB.smirk_range_flag_clear()
    assert not tripped, "a kernel stored a value outside the split-fp16 range (|x| >= 65520 / non-finite) during this test"
