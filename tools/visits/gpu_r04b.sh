#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r04b}
mkdir -p $O
timeout 600 python -m pytest tests/test_resunit_bwd_gpu.py tests/test_resunit_gpu.py -q 2>&1 | tail -40 > $O/tests.txt
tail -5 $O/tests.txt
timeout 300 python tools/bench_resunit.py > $O/bench_resunit.txt 2>&1
cat $O/bench_resunit.txt
timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>$O/bench_s2.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', round(d['ms_per_step'],3), 'ms/step')" | tee $O/bench_s2.txt
(cd ab_base && timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', round(d['ms_per_step'],3), 'ms/step')") | tee -a $O/bench_s2.txt
timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>>$O/bench_s2.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', round(d['ms_per_step'],3), 'ms/step')" | tee -a $O/bench_s2.txt
tail -5 $O/bench_s2.err
