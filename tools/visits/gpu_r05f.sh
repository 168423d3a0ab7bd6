#!/bin/bash
# round 5, visit 6: the tiled weight fold (wn_fold_tile_kernel) -- parity, kernel time, step time against EVT_FOLD_ROWWISE=1
export TMPDIR=/tmp
O=gpurun_out/${1:-r05f}
mkdir -p $O
timeout 600 python -m pytest tests/test_bank_gpu.py tests/test_conv_gpu.py tests/test_resunit_gpu.py tests/test_s2_model_gpu.py tests/test_gemm_gpu.py tests/test_zz_readers_train_gpu.py -q -x 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -6 | tee $O/tests.txt
for r in 1 2; do for v in tiled rowwise; do
  if [ $v = rowwise ]; then export EVT_FOLD_ROWWISE=1; else unset EVT_FOLD_ROWWISE; fi
  timeout 300 python bench.py --workload s2 --steps 30 --warmup 5 --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); print('fold $v run $r:', round(d['ms_per_step'],3), 'ms')"
done; done | tee $O/fold_step.txt
unset EVT_FOLD_ROWWISE
mkdir -p $O/p
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python bench.py --workload s2 --steps 6 --warmup 3 --no-extras --graphs 0 > $O/p.log 2>&1
f=$(find $O/p -name '*kernel_stats.csv' | head -1); grep -i "wn_fold\|wn_grad\|adamw" $f | cut -c1-160 | tee $O/fold_kernels.txt
rm -rf $O/p
