"""C++ host mirror (reference class names over the C ABI): compiles everywhere, runs on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp")
EXE = "/tmp/rna_host_mirror_test"


def build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ros_navigation_amd", "csrc"), "-j4", "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "librna_oracle.so"])
    lib = os.path.join(ROOT, "ros_navigation_amd")
    orc = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-Wall", SRC, "-o", EXE, "-L" + lib, "-lrna", "-L" + orc,
                           "-lrna_oracle", "-Wl,-rpath," + lib, "-Wl,-rpath," + orc, "-lm"])


def test_host_mirror_compiles_against_the_c_abi():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_host_mirror_matches_oracle_on_gpu():
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "host mirror OK" in out.stdout, out.stdout + out.stderr
