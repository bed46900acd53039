#!/bin/bash
# developer run: the phase timers of the search kernel (a -DRNA_TSA_STATS build) under the default bench load and alone
cd ${GRAFT_REPO_ROOT:-/root/repo}
RNA_LIB=librna_stats.so RNA_BENCH_ONLY_ASTAR=1 timeout 300 python bench.py --no-cpu --steps 20 2>&1 | grep "tsa stats\|\"value\"" | cut -c1-700
REPS=2 RNA_LIB=librna_stats.so timeout 300 python scripts/astar_stats.py 4096 256 96000 2>&1 | grep "tsa stats\|search ms" | cut -c1-700
