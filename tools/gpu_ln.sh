#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04au}
mkdir -p $O
timeout 900 python -m pytest tests/test_enc_ops_gpu.py tests/test_gemm_gpu.py tests/test_s1_gpu.py tests/test_s1_c3_gpu.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "s1 warm   "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 waves8 "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 waves4 "; EVT_LN_BWD_WAVES4=1 run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 waves8 "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 waves4 "; EVT_LN_BWD_WAVES4=1 run timeout 300 python bench.py --workload s1 --no-extras --steps 8
} | tee $O/steps.txt
grep -v amdgpu.ids $O/err.txt | tail -3
