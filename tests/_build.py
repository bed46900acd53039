"""Native build products of a test session, each built ONCE per pytest process (test infrastructure).

Round 3's suite ran `make -C csrc` + `make -C oracle` + a g++ / hipcc link inside every C++ test: on a cold box (fresh
snapshot, compilers and ROCm libraries not yet paged in) that was minutes of the driver's 20.  Here: `native()` runs the
two makes once, `cpp(name)` builds a test binary the first time a test asks for it and hands the same file to every
later test."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ros_navigation_amd", "csrc")
LIB_DIR = os.path.join(ROOT, "ros_navigation_amd")
ORACLE_DIR = os.path.join(ROOT, "oracle")
HIPCC = "/opt/rocm/bin/hipcc"
BIN_DIR = os.path.join(tempfile.gettempdir(), "rna_test_bin_%d" % os.getuid())

_done = {}


def native():
    """librna.so, librna_rccl.so (HIP, gfx950) and the oracle, once per session"""
    if "native" not in _done:
        subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL)
        _done["native"] = True


# name -> (source relative to the repo root, kind): "host" = g++ against librna + the oracle (the C++ host mirrors check
# themselves against it), "plain" = g++ alone, "rccl" = hipcc against librna_rccl + librccl
SOURCES = {
    "host_mirror_test": ("tests/cpp/host_mirror_test.cpp", "host"),
    "nav_graph_node_shaped": ("tests/cpp/nav_graph_node_shaped.cpp", "host"),
    "nav_node_shaped": ("tests/cpp/nav_node_shaped.cpp", "host"),
    "rate_loop_test": ("tests/cpp/rate_loop_test.cpp", "plain"),
    "id_bootstrap_test": ("tests/cpp/id_bootstrap_test.cpp", "plain"),
    "tiled_host": ("examples/tiled_host.cpp", "rccl"),
}


def cpp(name):
    """path of the test binary `name`, built on first use"""
    if name in _done:
        return _done[name]
    src, kind = SOURCES[name]
    src = os.path.join(ROOT, src)
    os.makedirs(BIN_DIR, exist_ok=True)
    exe = os.path.join(BIN_DIR, name)
    if kind == "plain":
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", src, "-o", exe])
    elif kind == "host":
        native()
        subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", "-Wno-reorder", src, "-o", exe, "-L" + LIB_DIR, "-lrna", "-L" + ORACLE_DIR,
                               "-lrna_oracle", "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath," + ORACLE_DIR, "-lm", "-lpthread"])
    elif kind == "rccl":
        native()
        subprocess.check_call([HIPCC, "-O1", "-std=c++17", src, "-o", exe, "-L" + LIB_DIR, "-lrna_rccl", "-lrna", "-L/opt/rocm/lib", "-lrccl",
                               "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    else:
        raise ValueError(kind)
    _done[name] = exe
    return exe
