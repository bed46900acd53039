run() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value %.0f ms/step %.2f search %.2f'%(d['value'],d['ms_per_step'],d['kernel_ms_per_pass']['astar_search']))"; }
run RNA_LIB=librna.so
for m in 16 24 32 48; do run RNA_LIB=librna_m$m.so; done
RNA_LIB=librna_m24.so timeout 300 python -m pytest tests -m gpu -x -q -k astar 2>&1 | tail -1
