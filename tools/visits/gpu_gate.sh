#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04aw}
mkdir -p $O
timeout 900 python -m pytest tests/test_enc_ops_gpu.py tests/test_s2_parity_r2_gpu.py tests/test_s2_model_gpu.py tests/test_conv_gpu.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "s2 warm "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "s2 new  "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
echo -n "s2 base "; (cd ab_base && run timeout 300 python bench.py --workload s2 --no-extras --steps 20)
echo -n "s2 new  "; run timeout 300 python bench.py --workload s2 --no-extras --steps 20
} | tee $O/steps.txt
bash tools/visits/gpu_s2_graphstats.sh $(basename $O) > /dev/null 2>&1
grep "gated\|window" $O/replay_kernels.txt | cut -c1-120
grep -v amdgpu.ids $O/err.txt | tail -3
