#!/bin/bash
# Sweep-kernel variants on the headline step (single GPU) and as rank 0..7 of 8 (strong-scaling projection).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for S in 1 2 4 8; do
  echo "== HV_TSDF_BATCH_SPLIT=$S"
  HV_TSDF_BATCH_SPLIT=$S python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('N=1 fps', d['value'], 'ms/step', d['ms_per_step'])"
  HV_TSDF_BATCH_SPLIT=$S python tools/simulate_ranks.py --worlds 8 --steps 10 2>/dev/null | tail -1
done
