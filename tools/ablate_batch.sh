#!/bin/bash
# usage: tools/ablate_batch.sh "4 2 1"  -- multi-frame sweep with ZPW z-slabs per wave
for Z in $1; do
  HV_TSDF_BATCH_ZPW=$Z timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > /tmp/abz_$Z.json
  python -c "import json; d=json.load(open('/tmp/abz_$Z.json')); print('ZPW $Z', d['value'], 'fps', d['ms_per_step'], 'ms/step')"
done
