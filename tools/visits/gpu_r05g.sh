#!/bin/bash
# round 5, visit 7: the cut program with the merged generator pieces no longer joining the weight-gradient side stream early
export TMPDIR=/tmp
O=gpurun_out/${1:-r05g}
mkdir -p $O
for g in 2 1; do
  EVT_DP_G_PIECES=$g timeout 600 python -m pytest tests/test_zz_dp_overlap_gpu.py tests/test_zz_rccl_selftest_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids\|MIOpen\|socket.cpp\|RCCL\|HIP version\|ROCm\|Hostname\|Librccl" | tail -2
done | tee $O/tests.txt
run() {
  EVT_DP_D_PIECES=$2 EVT_DP_G_PIECES=$3 timeout 300 python bench.py --workload s2 --steps 30 --warmup 5 --no-extras --dp-program $1 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('dp-program $1  D $2  G $3:', round(d['ms_per_step'],3), 'ms', d['config']['launch'], d.get('comm'))"
}
{ run 0 6 3; run 1 6 3; run 1 6 2; run 1 6 1; run 1 1 1; run 2 6 2; run 0 6 3; run 1 6 2; run 1 6 1; } | tee $O/dp_program.txt
