#!/bin/bash
# The round's closing GPU visit: whole -m gpu suite, the driver's bench line, rocprofv3 kernel stats of both steps,
# HBM traffic counters (two PMC passes), per-shape conv table, attention kernel timings + MFMA/VALU counters.
export TMPDIR=/tmp
O=gpurun_out/${1:-r04z}
mkdir -p $O
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$? $(( $(date +%s)-t0 ))s" > $O/times.txt
tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/times.txt; tail -1 $O/smoke.log
t1=$(date +%s)
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s)-t1 ))s" >> $O/times.txt
mkdir -p $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python bench.py --workload s2 --steps 6 --warmup 3 --no-extras --graphs 0 > $O/p1.log 2>&1
find $O/p1 -name '*kernel_stats.csv' -exec cp {} $O/s2_kernel_stats_eager.csv \;
EVT_S1_WGRAD_SIDE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python bench.py --workload s1 --steps 4 --warmup 2 --no-extras > $O/p2.log 2>&1   # side stream off: per-kernel durations stand alone
find $O/p2 -name '*kernel_stats.csv' -exec cp {} $O/s1_kernel_stats.csv \;
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- python bench.py --workload s2 --steps 2 --warmup 2 --no-extras --graphs 0 > $O/p3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- python bench.py --workload s2 --steps 2 --warmup 2 --no-extras --graphs 0 > $O/p4.log 2>&1
F=$(find $O/p3 -name '*counter_collection.csv' | head -1); W=$(find $O/p4 -name '*counter_collection.csv' | head -1)
if [ -n "$F" ] && [ -n "$W" ]; then timeout 60 python tools/pmc_traffic.py $F $W > $O/pmc_traffic.json 2> $O/pmc_traffic.err; fi
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/p5 -- python tools/bench_attn.py --iters 2 > $O/p5.log 2>&1
A=$(find $O/p5 -name '*counter_collection.csv' | head -1)
if [ -n "$A" ]; then timeout 60 python tools/pmc_summary.py $A attn_ --json $O/attn_pmc.json > $O/attn_pmc.txt 2>&1; fi
rm -rf $O/p1 $O/p2 $O/p3 $O/p4 $O/p5
timeout 200 python tools/trace_shapes.py --top 400 > $O/conv_time_by_shape.txt 2>&1
timeout 100 python tools/bench_attn.py > $O/bench_attn_b32.json 2>/dev/null
timeout 100 python tools/bench_conv.py --wonly --only "plain C,dP,dS 1024,WN in,FFN 192->768 k3 T200" --iters 20 > $O/bench_wgrad.txt 2>&1
timeout 300 bash tools/visits/gpu_s2_graphstats.sh $(basename $O) > /dev/null 2>&1          # kernel list of the replayed step
timeout 300 python tools/glue_lines.py --top 120 2>&1 | grep -v amdgpu.ids > $O/glue_lines.txt
timeout 120 python tools/bench_wgrad_gemm.py 2>&1 | grep -v amdgpu.ids > $O/bench_wgrad_gemm.txt
echo "all $(( $(date +%s)-t0 ))s" >> $O/times.txt
cat $O/times.txt; head -c 600 $O/bench_line.json; echo; cat $O/pmc_traffic.json | head -20
