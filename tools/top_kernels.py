"""print the top rows of a rocprofv3 kernel_stats.csv as per-step figures: python tools/top_kernels.py file.csv [steps=9] [n=40]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 9.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU time {tot / steps / 1e6:.2f} ms/step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step")
for r in rows[:n]:
    print(f"{float(r['TotalDurationNs']) / tot * 100:5.1f}% {float(r['TotalDurationNs']) / steps / 1e3:8.1f}us/step "
          f"{int(r['Calls']) / steps:7.1f}/step {float(r['AverageNs']) / 1e3:8.1f}us  {r['Name'][:105]}")
