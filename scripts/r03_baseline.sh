#!/bin/bash
# round-3 baseline on a fresh box: VALU issue-rate microbenchmark, default bench, 20-step bench, phase timers
mkdir -p gpurun_out/r03
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu > gpurun_out/r03/ubench_valu.txt 2>&1
python bench.py --no-cpu > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err
python bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/r03/bench_20.json 2> gpurun_out/r03/bench_20.err
RNA_LIB=librna_stats.so python bench.py --no-cpu > gpurun_out/r03/bench_stats.json 2> gpurun_out/r03/bench_stats.err
REPS=3 RNA_LIB=librna_stats.so python scripts/astar_stats.py 4096 256 24000 > gpurun_out/r03/astar_stats.txt 2>&1
tail -3 gpurun_out/r03/ubench_valu.txt
