#!/bin/bash
# kernel trace (with queues and grids) of the graph-replayed s2 step, kept gzipped for tools/timeline_fill.py
export TMPDIR=/tmp
O=gpurun_out/${1:-r06t}
mkdir -p $O/p
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/p -- python bench.py --workload s2 --steps 6 --warmup 4 --no-extras > $O/p.log 2>&1
f=$(find $O/p -name '*kernel_trace.csv' | head -1)
python tools/timeline_fill.py "$f" --steps 6 > $O/timeline_fill.txt 2>&1
gzip -c "$f" > $O/kernel_trace.csv.gz
rm -rf $O/p
cat $O/timeline_fill.txt | head -70
