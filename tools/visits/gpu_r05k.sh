#!/bin/bash
# round 5, visit k: WN layer kernels with the conflict-free LDS swizzles -- parity, counters, stack timing, step
export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O; rm -rf $O/*
timeout 400 python -m pytest tests/test_wn_layer_gpu.py tests/test_enc_ops_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
mkdir -p $O/p6
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p6 -- python -m pytest tests/test_wn_layer_gpu.py -q -k "16-200 and bf16" > $O/p6.log 2>&1
Wn=$(find $O/p6 -name '*counter_collection.csv' | head -1)
if [ -n "$Wn" ]; then timeout 60 python tools/pmc_summary.py $Wn wn_layer > $O/wn_layer_pmc.txt 2>&1; fi
rm -rf $O/p6; cat $O/wn_layer_pmc.txt
timeout 120 python tools/bench_wn.py --flush 2>&1 | grep "per layer" > $O/bench_wn.txt
timeout 120 python tools/bench_wn.py --flush --backward 2>&1 | grep "per layer" >> $O/bench_wn.txt
cat $O/bench_wn.txt
for i in 1 2; do
  timeout 300 python bench.py --workload s2 --no-extras --steps 20 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step')" >> $O/step.txt
done
cat $O/step.txt
