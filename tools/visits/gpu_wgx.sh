#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r04w}
mkdir -p $O
{
for ns in 2 4; do
  echo "== NS=$ns with bias";  EVT_WGRAD_GEMM_NS=$ns timeout 200 python tools/bench_wgrad_gemm.py 2>&1 | grep -v amdgpu.ids
  echo "== NS=$ns no bias"; EVT_WGRAD_GEMM_NS=$ns timeout 200 python tools/bench_wgrad_gemm.py --no-bias 2>&1 | grep -v amdgpu.ids
done
echo "== NS=4 no bias 256 blocks"; EVT_WGRAD_GEMM_BLOCKS=256 EVT_WGRAD_GEMM_NS=4 timeout 200 python tools/bench_wgrad_gemm.py --no-bias 2>&1 | grep -v amdgpu.ids
} | tee $O/wgrad_gemm_bias.txt
