#!/bin/bash
# Round evidence in one GPU call: GPU test suite, the default bench line, rocprofv3 kernel stats + PMC passes of the same
# command (tools/profile_round.sh), the ownership-sharding projection, the VOXEL_GRID kernel table.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_GIT_HEAD=29b4d81
cd $R
mkdir -p gpurun_out
( time python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -12 gpurun_out/pytest_gpu.log
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log | cut -c1-300
mkdir -p profiles/r02 && cp gpurun_out/prof_round/pmc_summary.json profiles/r02/pmc_summary.json
python bench.py > gpurun_out/bench_n1.log 2>&1
grep '^{"metric"' gpurun_out/bench_n1.log > gpurun_out/bench_n1.json; cut -c1-1500 gpurun_out/bench_n1.json
python tools/simulate_ranks.py --worlds 1,2,4,8 --steps 10 > gpurun_out/simulate_ranks.jsonl 2>/dev/null; cat gpurun_out/simulate_ranks.jsonl | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vg_kt -o vg -- python $R/tools/bench_voxel_grid.py --steps 3 > $R/gpurun_out/bench_voxel_grid.json 2>/dev/null
find $R/gpurun_out/vg_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/vg_kernel_stats.csv \;
cut -c1-120 $R/gpurun_out/vg_kernel_stats.csv | head -12
