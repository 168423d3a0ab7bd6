r1() { python bench.py --workload s1 --no-extras --steps 12 --warmup 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
r1 s1_default
EVT_WGRAD_GEMM_BLOCKS=1024 r1 s1_wg1024
r1 s1_default
EVT_WGRAD_GEMM_BLOCKS=1024 r1 s1_wg1024
EVT_WGRAD_GEMM_BLOCKS=2048 r1 s1_wg2048
