#!/bin/bash
# first run of the asynchronous scheduler: small parity first (bounded by timeouts: a scheduler bug would spin)
mkdir -p gpurun_out/r03
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/r03/smoke.txt
timeout 600 python -m pytest tests -m gpu -x -q -k "astar" > gpurun_out/r03/pytest_astar.txt 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/pytest_astar.txt
timeout 300 python bench.py --no-cpu > gpurun_out/r03/bench_async.json 2> gpurun_out/r03/bench_async.err; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 > gpurun_out/r03/bench_async20.json 2> gpurun_out/r03/bench_async20.err
RNA_LIB=librna_stats.so timeout 300 python bench.py --no-cpu > gpurun_out/r03/bench_async_stats.json 2> gpurun_out/r03/bench_async_stats.err
REPS=3 RNA_LIB=librna_stats.so timeout 300 python scripts/astar_stats.py 4096 256 24000,48000,96000,192000 > gpurun_out/r03/astar_stats_async.txt 2>&1
tail -3 gpurun_out/r03/pytest_astar.txt
