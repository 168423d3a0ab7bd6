#!/bin/bash
# round 5, visit 2: the IEEE-half build (op parity cases, the fp16_run whole-step golden, scaler kernels), a bf16 regression
# subset after the h16_t rename, the piece-count sweep of the data-parallel program on one GPU, bench --dtype f16
export TMPDIR=/tmp
O=gpurun_out/${1:-r05b}
mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_s2_fp16_gpu.py tests/test_fp16_ops_gpu.py -q -s > $O/fp16.log 2>&1
echo "fp16 rc=$? $(( $(date +%s)-t0 ))s" | tee $O/times.txt
grep -v "amdgpu.ids\|MIOpen" $O/fp16.log | tail -60
t1=$(date +%s)
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_resunit_gpu.py tests/test_resunit_bwd_gpu.py tests/test_mha_gpu.py tests/test_s2_model_gpu.py tests/test_s1_gpu.py tests/test_zz_rccl_selftest_gpu.py -q -x > $O/bf16_subset.log 2>&1
echo "bf16 subset rc=$? $(( $(date +%s)-t1 ))s" | tee -a $O/times.txt
tail -3 $O/bf16_subset.log
run() {  # dp-program, D pieces, G pieces
  EVT_DP_D_PIECES=$2 EVT_DP_G_PIECES=$3 timeout 300 python bench.py --workload s2 --steps 30 --warmup 5 --no-extras --dp-program $1 2> $O/dp.err | grep '^{' | tail -1 > $O/dp_$1_$2_$3.json
  python - <<PY
import json
try:
    d = json.load(open("$O/dp_$1_$2_$3.json"))
    print("dp-program $1  D pieces $2  G pieces $3:", round(d["ms_per_step"], 3), "ms", d["config"]["launch"], d.get("comm"))
except Exception as e:
    print("dp-program $1 $2 $3 failed", e)
PY
}
{ run 0 6 3; run 1 6 3; run 1 3 3; run 1 2 3; run 1 1 3; run 1 2 2; run 1 1 2; run 1 1 1; run 2 6 3; run 2 2 2; run 2 1 2; run 0 6 3; } | tee $O/dp_program.txt
timeout 300 python bench.py --workload s2 --steps 20 --warmup 5 --no-extras --dtype f16 2> $O/f16.err | grep '^{' | tail -1 > $O/bench_f16.json
head -c 900 $O/bench_f16.json; echo
echo "all $(( $(date +%s)-t0 ))s" | tee -a $O/times.txt
