#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
VARIANTS=${VARIANTS:-0:2 4:4}
for CS in $VARIANTS; do
  C=${CS%%:*}; S=${CS##*:}
  echo -n "COL=$C SPLIT=$S  "
  HV_TSDF_BATCH_COL=$C HV_TSDF_BATCH_SPLIT=$S python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fps', d['value'])"
done
