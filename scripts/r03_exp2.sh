#!/bin/bash
mkdir -p gpurun_out/r03
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/ubench_valu.hip -o /tmp/ubench_valu && timeout 120 /tmp/ubench_valu > gpurun_out/r03/ubench_valu2.txt 2>&1
bash scripts/pmc_astar_sq.sh 96000 > gpurun_out/r03/pmc_sq_async.txt 2>&1
grep "W=8" gpurun_out/r03/ubench_valu2.txt
