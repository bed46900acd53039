#!/bin/bash
# developer run (round 5): wavefronts per query x CUs the searches may use.  Seven wavefronts per query leave one wave slot per
# SIMD (and 10 KB of LDS per CU) to the engine stream's kernels on EVERY CU instead of 32 CUs kept back for them.
# usage: bash scripts/r05_waves.sh out_name
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/$1.txt
: > $OUT
run() {   # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --steps 20 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']
    print('$label', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3))
except Exception as ex:
    print('$label FAILED', ex)
" >> $OUT
  tail -n 2 /tmp/err.txt | grep -v amdgpu.ids >> $OUT
}
for rep in 1 2; do
run "w8 skip32 astar-only" RNA_BENCH_ONLY_ASTAR=1
run "w8 skip0  astar-only" RNA_BENCH_ONLY_ASTAR=1 RNA_SEARCH_CU_SKIP=0
run "w7 skip0  astar-only" RNA_BENCH_ONLY_ASTAR=1 RNA_SEARCH_CU_SKIP=0 RNA_LIB=librna_w7.so
run "w7 skip32 astar-only" RNA_BENCH_ONLY_ASTAR=1 RNA_LIB=librna_w7.so
run "w6 skip0  astar-only" RNA_BENCH_ONLY_ASTAR=1 RNA_SEARCH_CU_SKIP=0 RNA_LIB=librna_w6.so
run "w8 skip32 full-loop " A=1
run "w7 skip0  full-loop " RNA_SEARCH_CU_SKIP=0 RNA_LIB=librna_w7.so
run "w7 skip8  full-loop " RNA_SEARCH_CU_SKIP=8 RNA_LIB=librna_w7.so
run "w7 skip16 full-loop " RNA_SEARCH_CU_SKIP=16 RNA_LIB=librna_w7.so
done
cat $OUT
