#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04ag}
mkdir -p $O
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
for cap in 32 16 12 24 32 16; do
echo -n "cap=$cap "; EVT_WGRAD_PARTS_CAP=$cap run timeout 300 python bench.py --workload s2 --no-extras --steps 20
done
} | tee $O/cap.txt
