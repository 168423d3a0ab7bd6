#!/bin/bash
# One GPU-box visit: whole -m gpu suite, the driver's bench line, rocprofv3 kernel stats of the s2 step, per-shape table.
export TMPDIR=/tmp
O=gpurun_out/r02a
mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$? $(( $(date +%s)-t0 ))s" > $O/times.txt
t1=$(date +%s)
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$? $(( $(date +%s)-t1 ))s" >> $O/times.txt
t2=$(date +%s)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -- python bench.py --workload s2 --steps 6 --warmup 3 --no-extras --graphs 0 > $O/prof_s2.log 2>&1
echo "rocprof s2 rc=$? $(( $(date +%s)-t2 ))s" >> $O/times.txt
find $O/prof_s2 -name '*kernel_stats.csv' -exec cp {} $O/s2_kernel_stats_eager.csv \;
find $O/prof_s2 -type f ! -name '*stats.csv' -delete
t3=$(date +%s)
timeout 300 python tools/trace_shapes.py --top 400 > $O/conv_time_by_shape.txt 2>&1; echo "trace rc=$? $(( $(date +%s)-t3 ))s" >> $O/times.txt
cat $O/times.txt; tail -3 $O/gpu_tests.log; cat $O/bench_line.json | head -c 3000
