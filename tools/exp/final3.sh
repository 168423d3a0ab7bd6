export TMPDIR=/tmp
O=gpurun_out/r04zzz; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_gpu.py tests/test_bank_gpu.py tests/test_s2_graph_gpu.py -x -q 2>&1 | tail -3 | tee $O/tests.txt
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04zzz/bench_line.json').read().strip().splitlines()[-1])
print('s2', d['ms_per_step'], d['value']); print('s1', d['s1']['ms_per_step'], d['s1']['value'])
h=d['roofline']['hifigan_dec']; print('dec', h['ms_per_step'], h['launches_per_step'], h['frac'])
PY
