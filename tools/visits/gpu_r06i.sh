#!/bin/bash
# round 6, visit i: the reader processes on the GPU box's host (items/s) and a real-data training run with / without them
mkdir -p gpurun_out/r06i
cd $GRAFT_REPO_ROOT
nproc > gpurun_out/r06i/nproc.txt
for w in 4 6 8; do python tools/bench_reader.py --items 1024 --workers $w 2>/dev/null | tail -1 > gpurun_out/r06i/host_w$w.json; done
EVT_PAD_FRAMES=16 python tools/bench_reader.py --items 1024 --train-steps 400 --workers 0 2>/dev/null | tail -1 > gpurun_out/r06i/train_w0.json
EVT_PAD_FRAMES=16 python tools/bench_reader.py --items 1024 --train-steps 400 --workers 6 2>/dev/null | tail -1 > gpurun_out/r06i/train_w6.json
cat gpurun_out/r06i/nproc.txt; for f in gpurun_out/r06i/*.json; do echo $f; python -c "
import json,sys; d=json.load(open('$f')); print({k:d[k] for k in d if k in ('host','host_workers')}); t=d.get('train'); print(t and {k:t[k] for k in ('ms_per_step','reader_wait_ms_per_step','audio_seconds_per_s','eager_share','last_third')})"; done
