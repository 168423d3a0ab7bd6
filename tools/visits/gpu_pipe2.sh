#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r04v}
mkdir -p $O
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
for q in 1 2 8; do
echo -n "queues=$q pipe   "; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "queues=$q serial "; DEBUG_HIP_FORCE_GRAPH_QUEUES=$q EVT_BOOK_PIPE=0 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
done
echo -n "default  pipe   "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "default  serial "; EVT_BOOK_PIPE=0 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
} | tee $O/pipe_queues.txt
tail -3 $O/err.txt
