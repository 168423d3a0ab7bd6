#!/bin/bash
# round 5, visit j: one-launch data half of the WN layer backward -- parity, stack timing, step A/B
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O; rm -f $O/*.txt
timeout 400 python -m pytest tests/test_wn_layer_gpu.py tests/test_enc_ops_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
timeout 120 python tools/bench_wn.py --flush --backward 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/bench_wn.txt
EVT_WN_BWD_NT=2 timeout 120 python tools/bench_wn.py --flush --backward 2>&1 | grep -v amdgpu.ids | tail -4 >> $O/bench_wn.txt
cat $O/bench_wn.txt
for i in 1 2; do
  for v in "EVT_WN_BWD_NT=1" "EVT_WN_BWD_NT=2" "EVT_NO_WN_LAYER=1"; do
    env $v timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms/step')" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
