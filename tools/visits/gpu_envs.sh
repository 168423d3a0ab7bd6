#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04ai}
mkdir -p $O
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "warm            "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "default         "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
for b in 16 64 1024 4096; do
echo -n "GRAPH_BATCH=$b  "; DEBUG_HIP_GRAPH_BATCH_SIZE=$b run timeout 300 python bench.py --workload s2 --no-extras --steps 20
done
echo -n "default         "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "MAX_BATCH=4096  "; DEBUG_CLR_MAX_BATCH_SIZE=4096 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "GPU_MAX_HW_QUEUES=8 "; GPU_MAX_HW_QUEUES=8 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
} | tee $O/envs.txt
grep -v amdgpu.ids $O/err.txt | tail -3
