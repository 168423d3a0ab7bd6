#!/bin/bash
# round 5, visit 1: the new parity pins at the bench shapes, the one-rank RCCL self-test, and the cost of the data-parallel
# (cut, eleven-graph) program next to the three-phase program on one GPU (bench.py --dp-program 0 | 1 | 2, alternated)
export TMPDIR=/tmp
O=gpurun_out/${1:-r05a}
mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_s1_c3_gpu.py tests/test_zz_bf16_cosine_gpu.py tests/test_zz_rccl_selftest_gpu.py -q -s > $O/pins.log 2>&1
echo "pins rc=$? $(( $(date +%s)-t0 ))s" | tee $O/times.txt
grep -v amdgpu.ids $O/pins.log | tail -40
for r in 1 2; do
  for m in 0 1 2; do
    timeout 300 python bench.py --workload s2 --steps 30 --warmup 5 --no-extras --dp-program $m 2> $O/dp$m.err | tail -1 > $O/dp_program_${m}_run$r.json
    python - <<PY
import json
d = json.load(open("$O/dp_program_${m}_run$r.json"))
print("dp-program $m run $r:", round(d["ms_per_step"], 3), "ms", d["config"]["launch"], d.get("comm"))
PY
  done
done | tee $O/dp_program.txt
echo "all $(( $(date +%s)-t0 ))s" | tee -a $O/times.txt
