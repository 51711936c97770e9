import pathlib
import sys

import pytest

try:  # torch bundles its own HIP runtime under the same soname as /opt/rocm's: whichever loads first serves the
    import torch  # noqa: F401  whole process, and torch refuses to start on the other one -- so let torch go first
except ImportError:  # pragma: no cover
    torch = None

ROOT = pathlib.Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "oracle", ROOT / "audio-forge_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no binaries (they are git-ignored): build the HIP library (hipcc cross-compiles
    without a GPU) and the oracle once, exactly as `__graft_entry__.build()` does."""
    lib = ROOT / "audio-forge_amd" / "libaudioforge_mi.so"
    oracle_lib = ROOT / "oracle" / "libaf_oracle.so"
    if lib.exists() and oracle_lib.exists():
        return
    import __graft_entry__

    __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    import af_oracle_py

    af_oracle_py.lib()
    return af_oracle_py
