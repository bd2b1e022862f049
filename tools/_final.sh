#!/bin/bash
# Round evidence in one GPU call: GPU test suite, rocprofv3 kernel stats + PMC passes of the default bench command
# (tools/profile_round.sh), the VOXEL_GRID and semantic kernel tables + traffic passes, then the default bench line (which parses
# the summaries), the two sweep forms, the N = 8 line on the one GPU, the ownership-sharding projection.  Every profiler run sits
# under its own `timeout`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r06}
export GRAFT_GIT_HEAD=$(cat $R/.final_head 2>/dev/null)
cd $R
mkdir -p gpurun_out profiles/$ROUND
( time timeout 1200 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -12 gpurun_out/pytest_gpu.log
# frame caches first: the generator's worker processes must not start under the profiler
timeout 400 python -c "import bench; bench.load_frames('synthetic_640x480_5mm', 600); bench.load_frames('synthetic_640x480_5mm', 32); bench.load_frames('synthetic_640x480_5mm', 64); bench.load_frames('synthetic_640x480_5mm', 192); from tools import bench_tum; bench_tum.tum_frames(192)" > /dev/null 2>&1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log | cut -c1-300
cp gpurun_out/prof_round/pmc_summary.json profiles/$ROUND/pmc_summary.json
pushd /tmp > /dev/null; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vg_kt -o vg -- python $R/tools/bench_voxel_grid.py --steps 3 > $R/gpurun_out/bench_voxel_grid_profiled.json 2>/dev/null
find $R/gpurun_out/vg_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/vg_kernel_stats.csv \;
cut -c1-120 $R/gpurun_out/vg_kernel_stats.csv | head -8
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/vg_pmc_$C -o pmc -- python $R/tools/bench_voxel_grid.py --steps 3 > /dev/null 2>&1
done
# semantic flow: kernel table at 1 cm / 640x480 and at the ScanNet shape (1296x968, 2 mm)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sem_kt -o sem -- python $R/tools/bench_semantic.py --frames 10 --cpu-frames 0 > $R/gpurun_out/bench_semantic_profiled.json 2>/dev/null
find $R/gpurun_out/sem_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/sem_kernel_stats.csv \;
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sem2_kt -o sem -- python $R/tools/bench_semantic.py --frames 12 --cpu-frames 0 --voxel 0.002 --config scannet_1296x968_2mm --stride 2 > $R/gpurun_out/bench_semantic_scannet_profiled.json 2>/dev/null
find $R/gpurun_out/sem2_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/sem_scannet_kernel_stats.csv \;
cut -c1-120 $R/gpurun_out/sem_scannet_kernel_stats.csv | head -8
python $R/tools/memset_split.py $R/gpurun_out/sem_kt > $R/gpurun_out/memset_split_semantic.json 2>/dev/null
python $R/tools/memset_split.py $R/gpurun_out/sem2_kt > $R/gpurun_out/memset_split_semantic_scannet_2mm.json 2>/dev/null
# semantic flow: HBM traffic per kernel (two --pmc passes each, counters only)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sem_pmc_$C -o pmc -- python $R/tools/bench_semantic.py --frames 10 --cpu-frames 0 > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sem2_pmc_$C -o pmc -- python $R/tools/bench_semantic.py --frames 12 --cpu-frames 0 --voxel 0.002 --config scannet_1296x968_2mm --stride 2 > /dev/null 2>&1
done
popd > /dev/null
python tools/pmc_summary.py --json gpurun_out/pmc_voxel_grid.json --command-key "tools/bench_voxel_grid.py --steps 3" gpurun_out/vg_pmc_FETCH_SIZE gpurun_out/vg_pmc_WRITE_SIZE > gpurun_out/vg_pmc_summary.txt 2>&1; grep -E "vgb|vg_" gpurun_out/vg_pmc_summary.txt | cut -c1-200
cp gpurun_out/pmc_voxel_grid.json profiles/$ROUND/pmc_voxel_grid.json
python tools/pmc_summary.py --json gpurun_out/pmc_semantic.json --command-key "tools/bench_semantic.py --frames 10 --cpu-frames 0" gpurun_out/sem_pmc_FETCH_SIZE gpurun_out/sem_pmc_WRITE_SIZE > gpurun_out/sem_pmc_summary.txt 2>&1
python tools/pmc_summary.py --json gpurun_out/pmc_semantic_scannet_2mm.json --command-key "tools/bench_semantic.py --frames 12 --cpu-frames 0 --voxel 0.002 --config scannet_1296x968_2mm --stride 2" gpurun_out/sem2_pmc_FETCH_SIZE gpurun_out/sem2_pmc_WRITE_SIZE > gpurun_out/sem2_pmc_summary.txt 2>&1
cp gpurun_out/pmc_semantic.json gpurun_out/pmc_semantic_scannet_2mm.json profiles/$ROUND/
grep -E "sem|shadow" gpurun_out/sem2_pmc_summary.txt | cut -c1-200 | head -8
timeout 200 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_n1.log 2>&1
grep '^{"metric"' gpurun_out/bench_n1.log > gpurun_out/bench_n1.json; cut -c1-1500 gpurun_out/bench_n1.json
timeout 250 python tools/sweep_variants.py --steps 9 --warmup 3 --repeat 3 "HV_TSDF_SWEEP=4" "HV_TSDF_SWEEP=4 HV_TSDF_SWEEP_ZS=2" "HV_TSDF_SWEEP=4 HV_TSDF_PIPELINE=0" "HV_TSDF_SWEEP=2" 2>/dev/null | tail -4 > gpurun_out/sweep_forms.jsonl; cut -c1-200 gpurun_out/sweep_forms.jsonl
for SH in owner tile; do
  BENCH_LIVE_PMC=0 timeout 300 python bench.py --force-dist --sharding $SH --steps 6 --warmup 2 --clock-ramp-steps 4 --no-secondary --no-cpu-baseline > gpurun_out/bench_nccl1_$SH.log 2>&1
  grep '^{"metric"' gpurun_out/bench_nccl1_$SH.log > gpurun_out/bench_nccl1_$SH.json; python -c "
import json; z=json.load(open('gpurun_out/bench_nccl1_$SH.json')); print('nccl world 1 $SH', z['value'], z.get('merge'))" 2>&1 | tail -1 | cut -c1-300
done
timeout 250 python tools/simulate_ranks.py --worlds 1,2,4,8 --steps 12 > gpurun_out/simulate_ranks.jsonl 2>/dev/null; cut -c1-200 gpurun_out/simulate_ranks.jsonl
# the driver's own N = 8 command shape on this one GPU (eight ranks share device 0, gloo transport: rendezvous, per-rank load_frames,
# eight pools in one HBM, the per_rank block; NOT a scaling number)
for SH in owner tile; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 8 --steps 6 --warmup 2 --clock-ramp-steps 4 --all-on-device0 --backend gloo --sharding $SH > gpurun_out/bench_n8_$SH.log 2>&1
  grep '^{"metric"' gpurun_out/bench_n8_$SH.log > gpurun_out/bench_n8_$SH.json; python -c "
import json; z=json.load(open('gpurun_out/bench_n8_$SH.json')); print('n8 $SH', z['value'], z['n_gpus'], z.get('per_rank',{}).get('units_held'), z.get('merge'))" 2>&1 | tail -1 | cut -c1-400
done
