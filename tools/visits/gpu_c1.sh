#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/${1:-r04ax}
mkdir -p $O
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_bank_gpu.py tests/test_disc_gen_gpu.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
b() { timeout 200 python tools/bench_conv.py --wonly --only "dP2 last,dP7 last,conv_post" --iters 30 2>&1 | grep -v amdgpu.ids | grep "dP\|conv_post" | cut -c1-60,140-175; }
{
echo "== x-stationary"; b
echo "== old (EVT_NO_COUT1_XS)"; EVT_NO_COUT1_XS=1 b
} | tee $O/c1_xs.txt
