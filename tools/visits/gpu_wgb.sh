#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04aa}
mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_s1_gpu.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
{
echo "== with bias";  timeout 200 python tools/bench_wgrad_gemm.py 2>&1 | grep -v amdgpu.ids
echo "== no bias"; timeout 200 python tools/bench_wgrad_gemm.py --no-bias 2>&1 | grep -v amdgpu.ids
} | tee $O/wgrad_gemm_bias_mma.txt
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step', d['value'])"; }
{
echo -n "s1 new  "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 base "; (cd ab_base && run timeout 300 python bench.py --workload s1 --no-extras --steps 8)
echo -n "s1 new  "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 base "; (cd ab_base && run timeout 300 python bench.py --workload s1 --no-extras --steps 8)
} | tee $O/steps.txt
grep -v amdgpu.ids $O/err.txt | tail -5
