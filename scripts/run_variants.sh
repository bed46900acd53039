export RNA_ASTAR_KERNEL=tile
run() { echo "== P=$P $*"; env "$@" timeout 300 python bench.py --steps 16 --warmup 6 --no-cpu --bucket-width 16000 --pipeline $P 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value %.0f ms/step %.2f search %.2f init %.2f'%(d['value'],d['ms_per_step'],d['kernel_ms_per_step']['astar_search'],d['kernel_ms_per_step']['astar_init']))"; }
for P in 4 8; do run RNA_LIB=librna.so; run RNA_LIB=librna_w8.so; run RNA_LIB=librna_w4.so; done
RNA_ASTAR_KERNEL=tile timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "astar" 2>&1 | tail -1
