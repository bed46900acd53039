"""CPU-only: librna.so loads and exports every symbol include/rna.h declares; struct layouts of the
ctypes binding match the header; the product refuses to run without a device instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "ros_navigation_amd", "csrc"), "-j4", "-s"])
    from ros_navigation_amd import capi
    return capi


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rna.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rna_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(capi):
    L = capi.lib()
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), "librna.so does not export %s" % s
    assert sorted(capi.SYMBOLS) == syms
    assert L.rna_abi_version() == 1


def test_struct_layouts_match_header(capi):
    src = open(os.path.join(ROOT, "include", "rna.h")).read()
    prog = r"""
    #include <stdio.h>
    #include "rna.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(rna_geometry), sizeof(rna_ray), sizeof(rna_vfh_params),
             sizeof(rna_pose), sizeof(rna_vfh_out), sizeof(rna_astar_query), sizeof(rna_astar_result),
             sizeof(rna_rrt_query), sizeof(rna_rrt_result), (size_t)RNA_K_COUNT, sizeof(rna_laser_scan));
      return 0;
    }"""
    exe = "/tmp/rna_layout_check"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=prog.encode(), check=True)
    sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [C.sizeof(capi.Geometry), capi.RAY_DTYPE.itemsize, C.sizeof(capi.VfhParams),
                     capi.POSE_DTYPE.itemsize, capi.VFH_OUT_DTYPE.itemsize, capi.ASTAR_QUERY_DTYPE.itemsize,
                     capi.ASTAR_RESULT_DTYPE.itemsize, capi.RRT_QUERY_DTYPE.itemsize, capi.RRT_RESULT_DTYPE.itemsize,
                     len(capi.KERNELS), capi.SCAN_DTYPE.itemsize]
    assert src.count("extern \"C\"") == 1


def test_no_device_means_error_not_fallback(capi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    rc = capi.lib().rna_create(C.byref(h), 1.0, 1.0, 0.05, 0.0, 0.0, 0)
    assert rc == -6 and not h.value  # RNA_ENODEVICE
    with pytest.raises(capi.RnaError):
        capi.Engine(1.0, 1.0, 0.05)


def test_tailor_plan_host_entry_point(capi):
    """rna_tailor_plan (Nav::taileredPlan, mc/src/nav_node.cpp:192-204) is host-only: checked here against the oracle"""
    import numpy as np
    import _oracle as O
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 5, 6, 11, 57):
        plan = rng.normal(size=(n, 2))
        for stride in (1, 2, 5):
            assert np.array_equal(capi.tailor_plan(plan, stride), O.tailor_plan(plan, stride))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ros_navigation_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "rna_oracle" not in txt and "_oracle" not in txt and "librna_oracle" not in txt, f
