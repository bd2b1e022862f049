#!/bin/bash
# Produces the rocprofv3 evidence for one round under gpurun_out/ (copy the summaries to profiles/rNN/):
#   kernel-trace + stats (timing), then FETCH_SIZE and WRITE_SIZE in separate --pmc passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
cat $R/gpurun_out/prof_kt/bench_kernel_stats.csv | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$C.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_$C
done
