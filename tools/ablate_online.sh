#!/bin/bash
# usage: tools/ablate_online.sh "0 9" "4096 8192"   -- online-mode frames/s per kernel variant x grid size
for V in $1; do for G in $2; do
  HV_TSDF_DEBUG_VARIANT=$V HV_TSDF_GRID=$G timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --mode online 2>&1 | tail -1 > /tmp/abo.json
  python -c "import json; d=json.load(open('/tmp/abo.json')); print('variant $V grid $G', d['value'], 'fps', d['ms_per_step'], 'ms/step')"
done; done
