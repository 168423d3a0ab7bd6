#!/bin/bash
# rocprofv3 kernel stats of the eager s2 step + summary of the torch-native / vendor share
export TMPDIR=/tmp
O=gpurun_out/${1:-r02d}
mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s2 -- python bench.py --workload s2 --steps 6 --warmup 3 --no-extras --graphs 0 > $O/prof_s2.log 2>&1
find $O/prof_s2 -name '*kernel_stats.csv' -exec cp {} $O/s2_kernel_stats_eager.csv \;
rm -rf $O/prof_s2
python - <<PY
import csv
rows=list(csv.DictReader(open('$O/s2_kernel_stats_eager.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
nat=[r for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'] or 'Cijk' in r['Name']]
print('GPU ms/step',tot/9/1e6,'native+vendor share',sum(float(r['TotalDurationNs']) for r in nat)/tot, 'launches/step', sum(int(r['Calls']) for r in nat)/9, 'of', sum(int(r['Calls']) for r in rows)/9)
for r in nat[:14]:
    print(f"{float(r['TotalDurationNs'])/9/1e3:8.1f}us/step {int(r['Calls'])/9:7.1f}/step {float(r['AverageNs'])/1e3:7.1f}us  {r['Name'][:110]}")
PY
