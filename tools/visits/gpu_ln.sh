#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04au}
mkdir -p $O
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_s1_gpu.py tests/test_s1_c3_gpu.py -x -q 2>&1 | tail -4 | tee $O/tests.txt
run() { "$@" 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), 'ms/step')"; }
{
echo -n "s1 warm "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 new  "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
echo -n "s1 base "; (cd ab_base && run timeout 300 python bench.py --workload s1 --no-extras --steps 8)
echo -n "s1 new  "; run timeout 300 python bench.py --workload s1 --no-extras --steps 8
} | tee $O/steps.txt
mkdir -p $O/p2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python bench.py --workload s1 --steps 4 --warmup 2 --no-extras > $O/p2.log 2>&1
find $O/p2 -name '*kernel_stats.csv' -exec cp {} $O/s1_kernel_stats.csv \;
rm -rf $O/p2
grep "ce_sum\|res_drop_ln" $O/s1_kernel_stats.csv | cut -c1-60,150-260
grep -v amdgpu.ids $O/err.txt | tail -3
