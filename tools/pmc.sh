#!/bin/bash
# usage: tools/pmc.sh <variant> <tag> COUNTER [COUNTER...]   (one rocprofv3 --pmc pass, kernel-trace only)
V=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
HV_TSDF_DEBUG_VARIANT=$V timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o pmc -- python $R/bench.py --steps 2 --warmup 1 --frames-per-step 8 --no-cpu-baseline > $R/gpurun_out/pmc_$TAG.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pmc_$TAG
