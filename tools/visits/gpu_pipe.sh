#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r04s}
mkdir -p $O
timeout 900 python -m pytest tests/test_s2_book_pipe_gpu.py tests/test_s2_graph_gpu.py -x -q 2>&1 | tail -30 > $O/tests.txt
tail -12 $O/tests.txt
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "pipe   "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "serial "; EVT_BOOK_PIPE=0 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "pipe   "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "serial "; EVT_BOOK_PIPE=0 run timeout 300 python bench.py --workload s2 --no-extras --steps 20
} | tee $O/pipe.txt
tail -3 $O/err.txt
