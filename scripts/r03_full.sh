#!/bin/bash
mkdir -p gpurun_out/r03
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 200 python scripts/fuzz_astar.py 120 71 2>&1 | tail -2
timeout 200 python scripts/fuzz_astar.py 120 72 2>&1 | tail -2
timeout 150 python scripts/fuzz_himm_vfh.py 90 73 2>&1 | tail -2
timeout 150 python scripts/fuzz_tiled.py 60 74 2>&1 | tail -2
