#!/bin/bash
# Round evidence in one GPU call: GPU test suite, rocprofv3 kernel stats + PMC passes of the default bench command
# (tools/profile_round.sh), the VOXEL_GRID kernel table + traffic passes, then the default bench line (which parses both
# summaries), the ownership-sharding projection, extraction on its own.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_GIT_HEAD=$(cat $R/.final_head 2>/dev/null)
cd $R
mkdir -p gpurun_out profiles/r02
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -12 gpurun_out/pytest_gpu.log
# frame caches first: the generator's worker processes must not start under the profiler
timeout 300 python -c "import bench; bench.load_frames('synthetic_640x480_5mm', 640); bench.load_frames('synthetic_640x480_5mm', 32); bench.load_frames('synthetic_640x480_5mm', 320)" > /dev/null 2>&1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log | cut -c1-300
cp gpurun_out/prof_round/pmc_summary.json profiles/r02/pmc_summary.json
pushd /tmp > /dev/null; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vg_kt -o vg -- python $R/tools/bench_voxel_grid.py --steps 3 > $R/gpurun_out/bench_voxel_grid_profiled.json 2>/dev/null
find $R/gpurun_out/vg_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/vg_kernel_stats.csv \;
cut -c1-120 $R/gpurun_out/vg_kernel_stats.csv | head -12
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/vg_pmc_$C -o pmc -- python $R/tools/bench_voxel_grid.py --steps 3 > /dev/null 2>&1
done
popd > /dev/null
python tools/pmc_summary.py --json gpurun_out/pmc_voxel_grid.json --command-key "tools/bench_voxel_grid.py --steps 3" gpurun_out/vg_pmc_FETCH_SIZE gpurun_out/vg_pmc_WRITE_SIZE > gpurun_out/vg_pmc_summary.txt 2>&1; grep -E "vgb|vg_" gpurun_out/vg_pmc_summary.txt | cut -c1-200
cp gpurun_out/pmc_voxel_grid.json profiles/r02/pmc_voxel_grid.json
timeout 300 python tools/bench_voxel_grid.py --steps 3 > gpurun_out/bench_voxel_grid.json 2>/dev/null; cut -c1-600 gpurun_out/bench_voxel_grid.json
timeout 600 python bench.py > gpurun_out/bench_n1.log 2>&1
grep '^{"metric"' gpurun_out/bench_n1.log > gpurun_out/bench_n1.json; cut -c1-1500 gpurun_out/bench_n1.json
timeout 300 python tools/simulate_ranks.py --worlds 1,2,4,8 --steps 10 > gpurun_out/simulate_ranks.jsonl 2>/dev/null; cat gpurun_out/simulate_ranks.jsonl | cut -c1-200
timeout 200 python tools/bench_extraction.py > gpurun_out/bench_extraction.txt 2>&1; grep -E "^units|^rep" gpurun_out/bench_extraction.txt
