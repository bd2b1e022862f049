#!/bin/bash
# Produces the rocprofv3 evidence for one round under gpurun_out/ (copy the summaries to profiles/rNN/):
#   kernel-trace + stats (timing), then FETCH_SIZE and WRITE_SIZE in separate --pmc passes, then one SQ pass and one
#   TCP pass (counters only, no trace domains besides --kernel-trace).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
cat $R/gpurun_out/prof_kt/bench_kernel_stats.csv | cut -c1-200
pmc_pass() { # name, counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$name.log 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_$name
}
pmc_pass FETCH_SIZE FETCH_SIZE
pmc_pass WRITE_SIZE WRITE_SIZE
pmc_pass sq SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM GRBM_GUI_ACTIVE
pmc_pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
