#!/bin/bash
# same-box A/B of the s2 (and optionally s1) step: ab_base/ (a worktree of the last commit with its own build) vs this tree
W=${1:-s2}
for i in 1 2; do
  for d in ab_base .; do
    (cd $d && timeout 300 python bench.py --workload $W --no-extras --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', round(d['ms_per_step'],3), 'ms/step')")
  done
done
