#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04ar}
mkdir -p $O
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "warm      "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
for mb in 48 24 12 6 48 24 12; do
echo -n "slab_mb=$mb "; EVT_WGRAD_SLAB_MB=$mb run timeout 300 python bench.py --workload s2 --no-extras --steps 20
done
} | tee $O/slabmb.txt
