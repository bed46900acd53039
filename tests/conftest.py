import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The CPU oracle is test infrastructure: (re)build it once per session."""
    import _oracle
    _oracle.build()
    yield


@pytest.fixture(scope="session", autouse=True)
def _warm_page_cache():
    """On a GPU box whose image has just been pulled, the first touch of the big ROCm files goes page fault by page fault
    (librccl.so alone is 570 MB of code objects: a process that links it spent minutes in its first ncclCommInitRank on a
    cold box in round 3, looking like a hang).  Read them once, sequentially, in the background while the first tests
    run -- a plain read streams at the disk's rate."""
    import threading
    if not os.path.exists("/dev/kfd"):
        yield
        return
    files = ["/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libhsa-runtime64.so.1",
             "/opt/rocm/lib/llvm/bin/clang-22", "/opt/rocm/lib/llvm/bin/lld", "/opt/rocm/lib/libamd_comgr.so.3"]

    def read_all():
        for f in files:
            try:
                with open(os.path.realpath(f), "rb") as fh:
                    while fh.read(1 << 24):
                        pass
            except OSError:
                pass

    t = threading.Thread(target=read_all, daemon=True)
    t.start()
    yield
