export RNA_ASTAR_KERNEL=persist
run() { echo "== P=$P $*"; env "$@" timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu --pipeline $P 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value %.0f ms/step %.2f search %.2f'%(d['value'],d['ms_per_step'],d['kernel_ms_per_step']['astar_search']))"; }
P=1 run RNA_LIB=librna.so
P=2 run RNA_LIB=librna_w8.so
P=2 run RNA_LIB=librna_w8.so RNA_TSA_BLOCKS_PER_CU=2
P=4 run RNA_LIB=librna_w4.so
P=4 run RNA_LIB=librna_w4.so RNA_TSA_BLOCKS_PER_CU=2
P=4 run RNA_LIB=librna_w4.so RNA_TSA_BLOCKS_PER_CU=4
