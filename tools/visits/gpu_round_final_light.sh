#!/bin/bash
# after a closing visit: smoke, the driver's bench line and the kernel statistics again on a later tree (no counters, no test tier)
export TMPDIR=/tmp
O=gpurun_out/${1:-r06light}
mkdir -p $O/p1 $O/p2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p1 -- python bench.py --workload s2 --steps 6 --warmup 3 --no-extras --graphs 0 > $O/p1.log 2>&1
find $O/p1 -name '*kernel_stats.csv' -exec cp {} $O/s2_kernel_stats_eager.csv \;
EVT_S1_WGRAD_SIDE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python bench.py --workload s1 --steps 4 --warmup 2 --no-extras > $O/p2.log 2>&1
find $O/p2 -name '*kernel_stats.csv' -exec cp {} $O/s1_kernel_stats.csv \;
rm -rf $O/p1 $O/p2
timeout 200 python tools/trace_shapes.py --top 400 > $O/conv_time_by_shape.txt 2>&1
timeout 300 bash tools/visits/gpu_s2_graphstats.sh $(basename $O) > /dev/null 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["s1"]["value"], d["s1"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["hifigan_dec"]["frac"], d["roofline"]["hifigan_dec"]["ms_per_step"], d["s1"]["roofline"]["frac"])
PY
