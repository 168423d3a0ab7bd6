#!/bin/bash
# round 5, visit h: one-launch WN layer forward -- parity against the four launches, tile / ring variants, step A/B
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O; rm -f $O/bench_wn.txt $O/step_ab.txt
timeout 400 python -m pytest tests/test_wn_layer_gpu.py tests/test_enc_ops_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
for nt in 1 2; do for ring in 8 4; do
  EVT_WN_NT=$nt EVT_WN_RING=$ring timeout 120 python tools/bench_wn.py --flush 2>&1 | grep -v amdgpu.ids >> $O/bench_wn.txt
done; done
cat $O/bench_wn.txt
for i in 1 2; do
  for v in "EVT_WN_NT=1" "EVT_WN_NT=2" "EVT_NO_WN_LAYER=1"; do
    env $v timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms/step')" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
