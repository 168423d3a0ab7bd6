#!/bin/bash
# Round 5's closing GPU visit: whole -m gpu suite, smoke, the driver's bench line (+ the fp16_run line), rocprofv3 kernel stats
# of both steps, HBM traffic counters (two PMC passes), per-shape conv table, attention kernel timings + MFMA/VALU counters,
# kernel list of the replayed step.
export TMPDIR=/tmp
O=gpurun_out/${1:-r05z}
mkdir -p $O
t0=$(date +%s)
timeout 1300 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$? $(( $(date +%s)-t0 ))s" > $O/times.txt
tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/times.txt; tail -1 $O/smoke.log
t1=$(date +%s)
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s)-t1 ))s" >> $O/times.txt
timeout 300 python bench.py --workload s2 --dtype f16 --no-extras 2>/dev/null | grep '^{' | tail -1 > $O/bench_line_f16.json
mkdir -p $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python bench.py --workload s2 --steps 6 --warmup 3 --no-extras --graphs 0 > $O/p1.log 2>&1
find $O/p1 -name '*kernel_stats.csv' -exec cp {} $O/s2_kernel_stats_eager.csv \;
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python bench.py --workload s1 --steps 4 --warmup 2 --no-extras > $O/p2.log 2>&1
find $O/p2 -name '*kernel_stats.csv' -exec cp {} $O/s1_kernel_stats.csv \;
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- python bench.py --workload s2 --steps 2 --warmup 2 --no-extras --graphs 0 > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- python bench.py --workload s2 --steps 2 --warmup 2 --no-extras --graphs 0 > $O/p4.log 2>&1
F=$(find $O/p3 -name '*counter_collection.csv' | head -1); W=$(find $O/p4 -name '*counter_collection.csv' | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then timeout 60 python tools/pmc_traffic.py $F $W > $O/pmc_traffic.json 2> $O/pmc_traffic.err; fi
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p5 -- python tools/bench_attn.py --iters 2 > $O/p5.log 2>&1
A=$(find $O/p5 -name '*counter_collection.csv' | head -1)
if [ -n "$A" ]; then timeout 60 python tools/pmc_summary.py $A attn_ --json $O/attn_pmc.json > $O/attn_pmc.txt 2>&1; fi
mkdir -p $O/p6
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p6 -- python -m pytest tests/test_wn_layer_gpu.py -q -k "16-200 and bf16" > $O/p6.log 2>&1
Wn=$(find $O/p6 -name '*counter_collection.csv' | head -1)
if [ -n "$Wn" ]; then timeout 60 python tools/pmc_summary.py $Wn wn_layer > $O/wn_layer_pmc.txt 2>&1; fi
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6
timeout 120 python tools/bench_wn.py --flush > $O/bench_wn.txt 2>&1
timeout 120 python tools/bench_wn.py --flush --backward >> $O/bench_wn.txt 2>&1
timeout 200 python tools/trace_shapes.py --top 400 > $O/conv_time_by_shape.txt 2>&1
timeout 100 python tools/bench_attn.py > $O/bench_attn_b32.json 2>/dev/null
bash tools/visits/graphstats.sh $O/replay_kernels.txt > /dev/null 2>&1
echo "all $(( $(date +%s)-t0 ))s" >> $O/times.txt
cat $O/times.txt; head -c 700 $O/bench_line.json; echo; head -c 400 $O/bench_line_f16.json; echo; head -12 $O/pmc_traffic.json
