#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r04r}
mkdir -p $O
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "normal      "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "no book     "; run timeout 300 python tools/exp/nobook.py --workload s2 --no-extras --steps 20
echo -n "no grad     "; NOBOOK=grad run timeout 300 python tools/exp/nobook.py --workload s2 --no-extras --steps 20
echo -n "no adamw    "; NOBOOK=adamw run timeout 300 python tools/exp/nobook.py --workload s2 --no-extras --steps 20
echo -n "no fold     "; NOBOOK=fold run timeout 300 python tools/exp/nobook.py --workload s2 --no-extras --steps 20
echo -n "normal      "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
} | tee $O/nobook.txt
tail -3 $O/err.txt
