"""CPU-only (hipcc cross-compiles): the register and LDS budgets that let the engine-stream kernels (map update, VFH+,
field reset) run NEXT TO the A* search workgroups that fill every CU when batches are pipelined.  One search workgroup
per CU holds 16 wavefronts (4 per SIMD) and most of the LDS; a map-update / VFH+ workgroup only gets onto that CU if
its LDS still fits into the 160 KB and one of its wavefronts fits into the VGPRs four search wavefronts leave free.
Measured when either budget was broken: vfh_step 0.13 ms -> 5.5 ms per step (LDS, round 2), himm_prep 0.4 -> 6.7 ms
(VGPRs, round 1) -- the step rate then hangs on the engine stream instead of the search capacity."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ros_navigation_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
LDS_PER_CU = 160 * 1024
VGPRS_PER_SIMD = 512


def resources(src):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-c",
                          os.path.join(CSRC, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
        for key in ("VGPRs", "AGPRs", r"LDS Size \[bytes/block\]", r"ScratchSize \[bytes/lane\]"):
            m = re.search(r"remark:\s+" + key + r": (\d+)", line)
            if m and name:
                res[name][key.split(" ")[0]] = int(m.group(1))
    return res


def alloc(vgprs):
    return (vgprs + 7) // 8 * 8       # gfx950 allocates VGPRs in blocks of 8


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_engine_stream_kernels_fit_next_to_the_search_workgroups():
    tile = resources("astar_tile.hip")
    search = next(v for k, v in tile.items() if "tsa_search_kernel" in k)
    assert search["ScratchSize"] == 0                                   # no spills in the relaxation loop
    search_lds = search["LDS"] + 2 * 4 * ((128 * 128 + 31) // 32)       # + the two tile bitsets of a 4096 x 4096 map
    free_vgprs = VGPRS_PER_SIMD - 4 * alloc(search["VGPRs"])
    free_lds = LDS_PER_CU - search_lds
    side = {}
    for src in ("vfh.hip", "himm.hip", "engine.hip"):
        side.update(resources(src))
    side.update({k: v for k, v in tile.items() if "tsa_search_kernel" not in k})
    hot = ("vfh_step_kernel", "himm_prep_kernel", "himm_tile_raster_kernel", "himm_bin_count_kernel", "himm_bin_scan_kernel",
           "himm_bin_fill_kernel", "himm_apply_kernel", "himm_collect_kernel",
           "compose_dirty_tiles_kernel", "nbr_mask_tiles_kernel", "tsa_reset_kernel", "tsa_snapshot_kernel", "tsa_backtrace_kernel")
    seen = 0
    for name, r in side.items():
        if not any(h in name for h in hot):
            continue
        seen += 1
        assert alloc(r["VGPRs"]) <= free_vgprs, "%s: %d VGPRs, %d left beside four search wavefronts" % (name, r["VGPRs"], free_vgprs)
        assert r["LDS"] <= free_lds, "%s: %d B of LDS, %d B left beside a search workgroup" % (name, r["LDS"], free_lds)
    assert seen >= 12
