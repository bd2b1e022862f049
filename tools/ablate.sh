#!/bin/bash
# usage: tools/ablate.sh "0 1 2 8 9" [extra bench args]  -- prints frames/s and ms/step per online-kernel variant (HV_TSDF_DEBUG_VARIANT)
VARS="$1"; shift
for V in $VARS; do
  HV_TSDF_DEBUG_VARIANT=$V timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>&1 | tail -1 > /tmp/ab_$V.json
  python -c "import json; d=json.load(open('/tmp/ab_$V.json')); print('variant $V', d['value'], 'fps', d['ms_per_step'], 'ms/step')"
done
