#!/bin/bash
# round 6, visit d: s1 fp16 fixture test; PMC traffic passes (FETCH_SIZE / WRITE_SIZE) of the s2 step for pmc_traffic.json
mkdir -p gpurun_out/r06d
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_s1_fp16_gpu.py -x -q -m gpu -s 2>&1 | tail -15 > gpurun_out/r06d/s1_fp16_test.txt
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python bench.py --workload s2 --steps 2 --warmup 2 --no-extras --graphs 0 > gpurun_out/r06d/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  cp "$f" /tmp/$c.csv; gzip -c "$f" > gpurun_out/r06d/$c.csv.gz
done
python tools/pmc_traffic.py /tmp/FETCH_SIZE.csv /tmp/WRITE_SIZE.csv 4 > gpurun_out/r06d/pmc_traffic.json 2> gpurun_out/r06d/pmc_traffic.err
cat gpurun_out/r06d/s1_fp16_test.txt; head -c 1500 gpurun_out/r06d/pmc_traffic.json; tail -3 gpurun_out/r06d/pmc_traffic.err
