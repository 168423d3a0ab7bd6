export TMPDIR=/tmp
O=gpurun_out/r04bb; mkdir -p $O
EVT_PAD_FRAMES=16 timeout 500 python tools/bench_reader.py --items 1024 --train-steps 600 2>$O/err.txt | tail -1 > $O/realdata_pad16_shapes64.json
cat $O/realdata_pad16_shapes64.json
EVT_PAD_FRAMES=16 EVT_GRAPH_SHAPES=16 timeout 500 python tools/bench_reader.py --items 1024 --train-steps 600 2>>$O/err.txt | tail -1 > $O/realdata_pad16_shapes16.json
cat $O/realdata_pad16_shapes16.json
tail -2 $O/err.txt | grep -v amdgpu
