#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04aj}
mkdir -p $O
timeout 1200 python -m pytest tests/test_enc_ops_gpu.py tests/test_s2_graph_gpu.py tests/test_s2_parity_r2_gpu.py tests/test_s2_model_gpu.py tests/test_zz_bf16_cosine_gpu.py -q 2>&1 | tail -25 | tee $O/tests.txt
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "warm "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "new  "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "base "; (cd ab_base && run timeout 300 python bench.py --workload s2 --no-extras --steps 20)
echo -n "new  "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
} | tee $O/steps.txt
grep -v amdgpu.ids $O/err.txt | tail -3
