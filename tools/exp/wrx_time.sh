export TMPDIR=/tmp
mkdir -p gpurun_out/r04ba
for v in 1 0 1 0; do echo "== EVT_WGRAD_RING_XCD=$v"; EVT_WGRAD_RING_XCD=$v timeout 200 python tools/bench_conv.py --wonly --only "WN in,WN rs,FFN 192->768 k3 T200,FFN 768->192 k3 T200" --iters 40 2>&1 | grep -v amdgpu.ids | grep "WN\|FFN" | cut -c1-45,140-175; done | tee gpurun_out/r04ba/wrx_time.txt
