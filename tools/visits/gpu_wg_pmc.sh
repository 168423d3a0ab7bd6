#!/bin/bash
# LDS behaviour of the weight-gradient GEMM: bank conflicts / LDS-array cycles / MFMA busy, per kernel
export TMPDIR=/tmp
O=gpurun_out/${1:-r04y}
mkdir -p $O
for ns in 2 4; do
EVT_WGRAD_GEMM_NS=$ns timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $O/pmc$ns -- python tools/bench_wgrad_gemm.py --iters 2 > $O/pmc$ns.log 2>&1
f=$(find $O/pmc$ns -name '*counter_collection.csv' | head -1); echo "== ns=$ns $f"
[ -n "$f" ] && python tools/pmc_summary.py $f wgrad_gemm | tee $O/pmc${ns}_summary.txt
rm -rf $O/pmc$ns
done
tail -5 $O/pmc2.log
