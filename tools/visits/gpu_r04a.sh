#!/bin/bash
# round 4, first GPU visit: parity of the fused ResBlock backward, its timings, same-box A/B of the s2 step
export TMPDIR=/tmp
O=gpurun_out/${1:-r04a}
mkdir -p $O
timeout 600 python -m pytest tests/test_resunit_bwd_gpu.py tests/test_resunit_gpu.py -q -x 2>&1 | tail -40 > $O/tests.txt
tail -5 $O/tests.txt
timeout 300 python tools/bench_resunit.py > $O/bench_resunit.txt 2>&1
cat $O/bench_resunit.txt
if [ -d ab_base ]; then timeout 600 bash tools/ab_bench.sh s2 > $O/ab_s2.txt 2>&1; cat $O/ab_s2.txt; fi
