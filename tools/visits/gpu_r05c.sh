#!/bin/bash
# round 5, visit 3: feature extractors (N2) against transformers on the host CPU; where the f16 conv_ring backward-data error
# sits; kernel lists of the replayed step under the three-phase program and the cut (data-parallel) program
export TMPDIR=/tmp
O=gpurun_out/${1:-r05c}
mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_feature_extractors_gpu.py -q -x -s > $O/feat.log 2>&1
echo "feat rc=$? $(( $(date +%s)-t0 ))s" | tee $O/times.txt
grep -v "amdgpu.ids\|MIOpen" $O/feat.log | tail -40
timeout 300 python tools/exp/diag_ring_f16.py > $O/diag_ring.txt 2>&1; tail -25 $O/diag_ring.txt
bash tools/visits/graphstats.sh $O/replay_dp0.txt --dp-program 0
EVT_DP_D_PIECES=1 EVT_DP_G_PIECES=1 bash tools/visits/graphstats.sh $O/replay_dp1_1_1.txt --dp-program 1
bash tools/visits/graphstats.sh $O/replay_dp1_6_3.txt --dp-program 1
echo "all $(( $(date +%s)-t0 ))s" | tee -a $O/times.txt
