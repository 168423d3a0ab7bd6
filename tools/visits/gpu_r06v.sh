#!/bin/bash
out=gpurun_out/${1:-r06v}
mkdir -p $out
T="tests/test_zz_bf16_cosine_gpu.py::test_s2_every_parameter_bf16_vs_fp32"
run() { echo "== $1"; env $1 python -m pytest "$T" -q -s -k c2_bench_shape 2>&1 | grep -E "s2 G|passed|failed" | cut -c1-420; }
run "EVT_MPD_STREAMS=1 EVT_ENC_STREAM=0 EVT_DEC_STREAM=0 EVT_CONV_PLAIN_X=0" | tee -a $out/cos.txt
run "EVT_MPD_STREAMS=2 EVT_ENC_STREAM=1 EVT_DEC_STREAM=1 EVT_CONV_PLAIN_X=0" | tee -a $out/cos.txt
run "EVT_MPD_STREAMS=1 EVT_ENC_STREAM=0 EVT_DEC_STREAM=0 EVT_CONV_PLAIN_X=1" | tee -a $out/cos.txt
run "EVT_MPD_STREAMS=2 EVT_ENC_STREAM=1 EVT_DEC_STREAM=1 EVT_CONV_PLAIN_X=1" | tee -a $out/cos.txt
