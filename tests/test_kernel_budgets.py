"""CPU-only (hipcc cross-compiles): the resource budgets of the grid-A* search kernel.  The tile job keeps a 64 x 16
tile in registers (2 x 16 rows) and is bound by what one wavefront can issue, so throughput comes from EIGHT wavefronts
per SIMD -- four workgroups (queries) of 8 wavefronts per CU: the kernel has to fit 64 VGPRs, must not spill in the
tile job, and four workgroups' LDS (scratch per wavefront, the open list's 3072 nodes / 256 class heads / tables, four tile bit sets) has to fit a CU
for the bench's map.  (astar.hip keeps 32 of the 256 CUs out of the search streams' CU mask for the engine stream's
short kernels.  Measured when the kernel needed 116 VGPRs -- four wavefronts per SIMD: 22 k instead of 36 k cycles/s;
with the job's lane constants spilled to scratch: 33 k.)  The path backtrace runs inside the search kernel since round 3
(first wavefront, in the open list's LDS): its budget is the search kernel's."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ros_navigation_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
TILE_FLAGS = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]   # FLAGS_astar_tile and FLAGS_vfh of ros_navigation_amd/csrc/Makefile
LDS_PER_CU = 160 * 1024
VGPRS_PER_SIMD = 512


def resources(src):
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", *(TILE_FLAGS if src in ("astar_tile.hip", "vfh.hip") else []), "-c",
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
def test_eight_search_wavefronts_fit_a_simd():
    tile = resources("astar_tile.hip")
    search = next(v for k, v in tile.items() if "tsa_search_kernelILi8ELb0E" in k)     # the pipelined instantiation: 8 wavefronts per workgroup
    single = next(v for k, v in tile.items() if "tsa_search_kernelILi16ELb0E" in k)    # one batch at a time: 16
    assert alloc(single["VGPRs"]) * 8 <= VGPRS_PER_SIMD and single["ScratchSize"] <= 32, single
    assert alloc(search["VGPRs"]) * 8 <= VGPRS_PER_SIMD, search        # 8 wavefronts per SIMD = 4 workgroups of 8 per CU
    assert search["ScratchSize"] <= 32, search                          # a few kernel-level values may live in scratch ...
    # ... but nothing is spilled or reloaded INSIDE the tile job (between its first and last marker in the assembly): a scratch
    # reload there waits for every store of the job that is still in flight.  Round 5's first sticky-tile build had four (tile
    # numbers and bit masks hoisted out of the new job loop as loop invariants) and ran 11 % slower than the kernel it replaced.
    asm = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", *TILE_FLAGS, "-S", "--cuda-device-only",
                          os.path.join(CSRC, "astar_tile.hip"), "-o", "-"], capture_output=True, text=True, timeout=600)
    assert asm.returncode == 0, asm.stderr[-2000:]
    lines = asm.stdout.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3rna17tsa_search_kernelILi8ELb0EEEvNS_9TsaLaunchE:"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    inside, in_job = [], False
    for l in lines[start:end]:
        if "TSA_MARK job_begin" in l:
            in_job = True
        elif "TSA_MARK job_end" in l:
            in_job = False
        elif in_job and l.strip().startswith("scratch_"):
            inside.append(l.strip())
    assert not inside, inside
    bench_lds = search["LDS"] + 4 * 4 * ((64 * 256 + 31) // 32)         # + the tile bit sets (pending / running pairs, open, far) of a 4096 x 4096 map (64 x 256 tiles)
    assert 4 * bench_lds <= LDS_PER_CU, bench_lds
    largest_lds = search["LDS"] + 4 * 4 * 2048                          # the largest supported map (65536 tiles): at least two per CU
    assert 2 * largest_lds <= LDS_PER_CU, largest_lds
    assert not any("tsa_backtrace_kernel" in k or "tsa_reset_kernel" in k for k in tile)   # both live inside the search kernel now


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_rasteriser_fits_a_cu_that_has_lost_one_search_workgroup():
    """Outside its reserve of 32 CUs the map update can only use what the searches leave: a CU whose four search
    workgroups are all resident has no wave slot free, one that has lost a workgroup has eight -- exactly a rasteriser
    workgroup (8 wavefronts, half-tile jobs with 8 KiB of clear counters).  Its LDS and registers must fit there."""
    himm = resources("himm.hip")
    raster = next(v for k, v in himm.items() if "himm_tile_raster_kernel" in k)
    tile = resources("astar_tile.hip")
    search = next(v for k, v in tile.items() if "tsa_search_kernelILi8ELb0E" in k)
    bench_search_lds = search["LDS"] + 4 * 4 * ((64 * 256 + 31) // 32)
    assert raster["LDS"] <= 16 * 1024, raster
    assert 3 * bench_search_lds + raster["LDS"] <= LDS_PER_CU, (bench_search_lds, raster["LDS"])
    assert alloc(raster["VGPRs"]) * 8 <= VGPRS_PER_SIMD and raster["ScratchSize"] == 0, raster     # 8 wavefronts per SIMD


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_six_vfh_wavefronts_fit_a_simd_without_a_spill():
    """vfh_step_kernel is bound by instruction issue at thousands of poses: what round 6 measured (78 us for 16 384 poses, DESIGN.md 5
    "Round 6" 6) was measured at six wavefronts per SIMD with nothing in scratch -- five cost 5 %, seven spill.  One workgroup
    (two wavefronts) per pose: its static LDS plus the Steerer window's dynamic share (480 magnitudes + 15 words) leaves the wave slots,
    not the LDS, as the limit of a CU."""
    vfh = resources("vfh.hip")
    step = next(v for k, v in vfh.items() if "vfh_step_kernel" in k)
    assert step["ScratchSize"] == 0, step
    assert alloc(step["VGPRs"]) * 6 <= VGPRS_PER_SIMD, step
    dynamic = (((480 + 31) // 32) * 32 + 15) * 4
    assert 12 * (step["LDS"] + dynamic) <= LDS_PER_CU, (step["LDS"], dynamic)     # 24 wavefronts = 6 per SIMD are 12 workgroups
