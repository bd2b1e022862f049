#!/bin/bash
# Sweep-kernel variants (HV_TSDF_BATCH_SPLIT) on the headline step: single GPU and the slowest rank of 2/4/8
# (strong-scaling projection on one GPU, tools/simulate_ranks.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for S in ${SPLITS:-2 4 8}; do
  echo "== HV_TSDF_BATCH_SPLIT=$S"
  HV_TSDF_BATCH_SPLIT=$S python tools/simulate_ranks.py --worlds ${WORLDS:-1,2,4,8} --steps 10 2>/dev/null | cut -c1-120
done
