#!/bin/bash
# developer run after a change of the search kernel: A* parity tests, the bench (full loop and searches only, twice),
# the phase timers of the -DRNA_TSA_STATS build (make variant NAME=stats EXTRA=-DRNA_TSA_STATS) and the SQ counters
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "astar or loop or retry" 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'frac_wall', round(d['roofline']['frac_wall'],4))"
RNA_BENCH_ONLY_ASTAR=1 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('only', round(d['value']), 'ms/step', round(d['ms_per_step'],3))"
done
[ -f ros_navigation_amd/librna_stats.so ] && bash scripts/r03_stats.sh | grep -v '^{'
bash scripts/pmc_astar_sq.sh 96000 2>&1 | grep -E "INSTS_VALU|INSTS_SALU|WAIT_ANY|WAVE_CYCLES|GUI"
