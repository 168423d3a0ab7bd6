#!/bin/bash
# attention kernels: parity tests, variant timings, two PMC passes (rocprofv3 --pmc, no other trace domains)
export TMPDIR=/tmp
O=gpurun_out/${1:-r02c}
mkdir -p $O
timeout 300 python -m pytest tests/test_s1_gpu.py -m gpu -x -q -k attention > $O/attn_tests.log 2>&1; tail -2 $O/attn_tests.log
timeout 200 python tools/bench_attn.py 2>/dev/null | tee $O/bench_attn.json
timeout 200 python tools/bench_attn.py --batch 16 2>/dev/null | tee $O/bench_attn_b16.json
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -- python tools/bench_attn.py --iters 2 > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc2 -- python tools/bench_attn.py --iters 2 > $O/pmc2.log 2>&1
for d in pmc1 pmc2; do f=$(find $O/$d -name '*counter_collection.csv' | head -1); echo "== $d $f"; [ -n "$f" ] && python tools/pmc_summary.py $f attn_ | tee $O/${d}_summary.txt; find $O/$d -type f ! -name '*counter_collection.csv' -delete; done
