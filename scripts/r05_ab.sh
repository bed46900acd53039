#!/bin/bash
# developer run (round 5): A* parity tests of the product library, then the default bench for several builds of the library in
# turn (twice around, scripts/r04_ab.sh), then the job statistics of the -DRNA_TSA_STATS build under the bench's load
# usage: bash scripts/r05_ab.sh out_name lib1.so lib2.so ...      (R05_TESTS=0 skips the parity tests, R05_STATS=0 the stats run)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05
NAME=$1; shift
if [ "${R05_TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "astar" > gpurun_out/r05/${NAME}_tests.txt 2>&1
  tail -n 4 gpurun_out/r05/${NAME}_tests.txt
fi
bash scripts/r04_ab.sh r05/${NAME}.txt "$@"
if [ "${R05_STATS:-1}" = "1" ]; then
  RNA_LIB=librna_stats.so timeout 300 python bench.py --no-cpu --steps 20 2>&1 | grep "tsa stats\|\"value\"" | cut -c1-900 > gpurun_out/r05/${NAME}_job_stats.txt
  cat gpurun_out/r05/${NAME}_job_stats.txt | cut -c1-600
fi
