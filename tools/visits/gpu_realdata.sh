#!/bin/bash
# real-data step time of the s2 trainer: bucketed ragged batches from a synthetic feature directory (2-10 s clips),
# EVT_PAD_FRAMES in {0, 8, 16, 32}: audio-s/s, captured shapes, share of eager steps
export TMPDIR=/tmp
O=gpurun_out/${1:-r04p}
mkdir -p $O
for pad in 0 8 16 32; do
  EVT_PAD_FRAMES=$pad timeout 500 python tools/bench_reader.py --items 1024 --train-steps 240 2>$O/err_$pad.txt | tail -1 > $O/realdata_pad$pad.json
  cat $O/realdata_pad$pad.json
done
