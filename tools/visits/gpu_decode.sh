#!/bin/bash
# s1 decoding: token-exact tests, throughput, rocprofv3 kernel stats of the replayed step
export TMPDIR=/tmp
O=gpurun_out/${1:-r04m}
mkdir -p $O


timeout 300 python tools/bench_s1_decode.py --rows 4 > $O/s1_decode_bench.json 2> $O/bench.err; cat $O/s1_decode_bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/bench_s1_decode.py --reps 1 > $O/prof.log 2>&1
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/s1_decode_kernel_stats.csv \;
rm -rf $O/prof
head -14 $O/s1_decode_kernel_stats.csv | cut -c1-170
