#!/bin/bash
# s2 step: bench line with the roofline objects, per-shape table of the conv entry points, rocprofv3 kernel stats (eager)
export TMPDIR=/tmp
O=gpurun_out/${1:-r04d}
mkdir -p $O
timeout 300 python -m pytest tests/test_bank_gpu.py tests/test_resunit_bwd_gpu.py -q 2>&1 | tail -5
timeout 400 python bench.py --workload s2 > $O/bench_line.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
d=json.load(open('$O/bench_line.json'))
r=d.get('roofline',{})
print('ms/step',d['ms_per_step'],'value',d['value'])
print('dominant',r.get('kernel'),r.get('frac'),r.get('ms_per_step'))
print('dec',json.dumps(r.get('hifigan_dec'),indent=None)[:900])
PY
timeout 300 python tools/trace_shapes.py --top 400 > $O/conv_time_by_shape.txt 2>&1
head -40 $O/conv_time_by_shape.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -- python bench.py --workload s2 --steps 6 --warmup 3 --no-extras --graphs 0 > $O/prof_s2.log 2>&1
find $O/prof_s2 -name '*kernel_stats.csv' -exec cp {} $O/s2_kernel_stats_eager.csv \;
rm -rf $O/prof_s2
head -30 $O/s2_kernel_stats_eager.csv | cut -c1-200
