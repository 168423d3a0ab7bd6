#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ac}
mkdir -p $O/p2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2 -- python bench.py --workload s1 --steps 4 --warmup 2 --no-extras > $O/p2.log 2>&1
find $O/p2 -name '*kernel_stats.csv' -exec cp {} $O/s1_kernel_stats.csv \;
rm -rf $O/p2
head -30 $O/s1_kernel_stats.csv | cut -c1-180
