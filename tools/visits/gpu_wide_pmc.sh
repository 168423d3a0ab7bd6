#!/bin/bash
# LDS / MFMA counters of the fused ResBlock kernels (wide forward, narrow grouped forward / backward)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04bc}
mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p -- python tools/bench_resunit.py > $O/p.log 2>&1
f=$(find $O/p -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python tools/pmc_summary.py $f resunit | tee $O/resunit_pmc.txt
rm -rf $O/p
