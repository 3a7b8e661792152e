import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fast-artistic-videos_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    O.set_threads(min(16, len(os.sched_getaffinity(0))))      # small problems: a 256-thread team costs more than it computes
    return O


@pytest.fixture(scope="session")
def favlib():
    """libfav must exist (built by __graft_entry__.build()); there is no fallback."""
    import fav_amd
    if not os.path.exists(fav_amd.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return fav_amd


@pytest.fixture(scope="session")
def cuda():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def poison(tmp_path_factory):
    """NaN-poison every CU's LDS and 1 GB of device memory (tests/util/lds_poison.hip, compiled once per session with hipcc): the
    state a GPU may be handed over in.  Kernels must not depend on anything they did not write (0 x NaN = NaN)."""
    import ctypes, subprocess
    src = os.path.join(ROOT, "tests", "util", "lds_poison.hip")
    so = str(tmp_path_factory.mktemp("poison") / "liblds_poison.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, src])
    lib = ctypes.CDLL(so)

    def run(repeats=3):
        assert lib.poison_lds(repeats) == 0
    return run
