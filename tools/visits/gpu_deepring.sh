#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04am}
mkdir -p $O
for v in 0 3 4; do
echo "== EVT_DEEP_RING=$v"
EVT_DEEP_RING=$v timeout 300 python tools/bench_conv.py --only "dP2,dP11" --iters 20 2>&1 | grep -v amdgpu.ids | grep -i "dP\|name" | cut -c1-160
done | tee $O/deep_ring.txt
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "ring0 "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "ring3 "; EVT_DEEP_RING=3 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "ring4 "; EVT_DEEP_RING=4 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "ring0 "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
} | tee $O/steps.txt
