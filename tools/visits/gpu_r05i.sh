#!/bin/bash
# round 5, visit i: weight images in fragment order for the wide ResBlock kernels -- parity, launch times, step
export TMPDIR=/tmp
O=gpurun_out/r05i; mkdir -p $O; rm -f $O/*.txt
timeout 600 python -m pytest tests/test_resunit_bwd_gpu.py tests/test_resunit_gpu.py tests/test_wn_layer_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -8 $O/tests.log
timeout 200 python tools/bench_resunit.py --wide-fwd 2>&1 | grep -v amdgpu.ids > $O/bench_resunit_wide_fwd.txt; cat $O/bench_resunit_wide_fwd.txt
for i in 1 2; do
  for v in "EVT_WN_NT=1" "EVT_NO_WN_LAYER=1"; do
    env $v timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms/step')" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
