#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "astar or loop or retry" 2>&1 | tail -3
for i in 1 2; do
python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', round(d['value']), 'ms/step', round(d['ms_per_step'],3))"
RNA_BENCH_ONLY_ASTAR=1 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('only', round(d['value']), 'ms/step', round(d['ms_per_step'],3))"
done
bash scripts/r03_stats.sh | grep -v '^{'
bash scripts/pmc_astar_sq.sh 96000 2>&1 | grep -E "INSTS_VALU|INSTS_SALU|WAIT_ANY|WAVE_CYCLES|GUI"
