#!/bin/bash
# Round evidence in one GPU call: GPU test suite, rocprofv3 kernel stats + PMC passes of the default bench command
# (tools/profile_round.sh), the VOXEL_GRID and semantic kernel tables + traffic passes, then the default bench line (which parses
# the summaries), the sweep A/B table, the ownership-sharding projection.  Every profiler run sits under its own `timeout`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r05}
export GRAFT_GIT_HEAD=$(cat $R/.final_head 2>/dev/null)
cd $R
mkdir -p gpurun_out profiles/$ROUND
( time timeout 900 python -m pytest tests -m gpu -q --durations=5 ) > gpurun_out/pytest_gpu.log 2>&1
tail -12 gpurun_out/pytest_gpu.log
# frame caches first: the generator's worker processes must not start under the profiler
timeout 300 python -c "import bench; bench.load_frames('synthetic_640x480_5mm', 600); bench.load_frames('synthetic_640x480_5mm', 32); bench.load_frames('synthetic_640x480_5mm', 192)" > /dev/null 2>&1
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -5 gpurun_out/profile_round.log | cut -c1-300
cp gpurun_out/prof_round/pmc_summary.json profiles/$ROUND/pmc_summary.json
pushd /tmp > /dev/null; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/vg_kt -o vg -- python $R/tools/bench_voxel_grid.py --steps 3 > $R/gpurun_out/bench_voxel_grid_profiled.json 2>/dev/null
find $R/gpurun_out/vg_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/vg_kernel_stats.csv \;
cut -c1-120 $R/gpurun_out/vg_kernel_stats.csv | head -12
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/vg_pmc_$C -o pmc -- python $R/tools/bench_voxel_grid.py --steps 3 > /dev/null 2>&1
done
# semantic flow: kernel table at 1 cm / 640x480 and at the ScanNet shape (1296x968, 2 mm)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sem_kt -o sem -- python $R/tools/bench_semantic.py --frames 10 --cpu-frames 0 > $R/gpurun_out/bench_semantic_profiled.json 2>/dev/null
find $R/gpurun_out/sem_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/sem_kernel_stats.csv \;
cut -c1-120 $R/gpurun_out/sem_kernel_stats.csv | head -14
timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sem2_kt -o sem -- python $R/tools/bench_semantic.py --frames 5 --cpu-frames 0 --voxel 0.002 --config scannet_1296x968_2mm --stride 2 > $R/gpurun_out/bench_semantic_scannet_profiled.json 2>/dev/null
find $R/gpurun_out/sem2_kt -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/sem_scannet_kernel_stats.csv \;
cut -c1-120 $R/gpurun_out/sem_scannet_kernel_stats.csv | head -10
# the fills of those two profiles, split into per-keyframe ones and one-off set-up
python $R/tools/memset_split.py $R/gpurun_out/sem_kt > $R/gpurun_out/memset_split_semantic.json 2>/dev/null
python $R/tools/memset_split.py $R/gpurun_out/sem2_kt > $R/gpurun_out/memset_split_semantic_scannet_2mm.json 2>/dev/null
cut -c1-400 $R/gpurun_out/memset_split_semantic_scannet_2mm.json
# semantic flow: HBM traffic per kernel (two --pmc passes each, counters only)
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sem_pmc_$C -o pmc -- python $R/tools/bench_semantic.py --frames 10 --cpu-frames 0 > /dev/null 2>&1
  timeout 250 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sem2_pmc_$C -o pmc -- python $R/tools/bench_semantic.py --frames 5 --cpu-frames 0 --voxel 0.002 --config scannet_1296x968_2mm --stride 2 > /dev/null 2>&1
done
# one rank's share at 8 ranks: what overlaps what (timeline of the last launches)
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rank8_kt -o t -- python $R/tools/sweep_variants.py --steps 18 --warmup 3 --repeat 2 --owner 3/8 HV_TSDF_SWEEP=4 > /dev/null 2>&1
python $R/tools/timeline.py $R/gpurun_out/rank8_kt 16 > $R/gpurun_out/rank8_timeline.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rank1_kt -o t -- python $R/tools/sweep_variants.py --steps 18 --warmup 3 --repeat 2 HV_TSDF_SWEEP=4 > /dev/null 2>&1
python $R/tools/timeline.py $R/gpurun_out/rank1_kt 16 > $R/gpurun_out/pipeline_timeline.txt 2>&1
popd > /dev/null
python tools/pmc_summary.py --json gpurun_out/pmc_voxel_grid.json --command-key "tools/bench_voxel_grid.py --steps 3" gpurun_out/vg_pmc_FETCH_SIZE gpurun_out/vg_pmc_WRITE_SIZE > gpurun_out/vg_pmc_summary.txt 2>&1; grep -E "vgb|vg_" gpurun_out/vg_pmc_summary.txt | cut -c1-200
cp gpurun_out/pmc_voxel_grid.json profiles/$ROUND/pmc_voxel_grid.json
python tools/pmc_summary.py --json gpurun_out/pmc_semantic.json --command-key "tools/bench_semantic.py --frames 10 --cpu-frames 0" gpurun_out/sem_pmc_FETCH_SIZE gpurun_out/sem_pmc_WRITE_SIZE > gpurun_out/sem_pmc_summary.txt 2>&1
python tools/pmc_summary.py --json gpurun_out/pmc_semantic_scannet_2mm.json --command-key "tools/bench_semantic.py --frames 5 --cpu-frames 0 --voxel 0.002 --config scannet_1296x968_2mm --stride 2" gpurun_out/sem2_pmc_FETCH_SIZE gpurun_out/sem2_pmc_WRITE_SIZE > gpurun_out/sem2_pmc_summary.txt 2>&1
cp gpurun_out/pmc_semantic.json gpurun_out/pmc_semantic_scannet_2mm.json profiles/$ROUND/
grep -E "sem|shadow" gpurun_out/sem2_pmc_summary.txt | cut -c1-200 | head -12
timeout 200 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_n1.log 2>&1
grep '^{"metric"' gpurun_out/bench_n1.log > gpurun_out/bench_n1.json; cut -c1-1500 gpurun_out/bench_n1.json
timeout 250 python tools/sweep_variants.py --steps 18 --warmup 3 --repeat 3 "HV_TSDF_SWEEP=4" "HV_TSDF_SWEEP=4 HV_TSDF_LPT=2" "HV_TSDF_SWEEP=4 HV_TSDF_FUSED=1" "HV_TSDF_SWEEP=4 HV_TSDF_FINISH=epilogue" "HV_TSDF_SWEEP=4 HV_TSDF_SWEEP_ANYSKIP=0" "HV_TSDF_SWEEP=3" "HV_TSDF_SWEEP=2" "HV_TSDF_SWEEP=1" 2>/dev/null | tail -8 > gpurun_out/sweep_forms.jsonl; cut -c1-200 gpurun_out/sweep_forms.jsonl
# round 5: issue-rate table on measured clocks, NS1 (MFMA unprojection) beside the VALU form, the N > 1 line of bench.py over RCCL with one rank
hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates tools/valu_rates.hip > /dev/null 2>&1 && timeout 120 /tmp/valu_rates > gpurun_out/valu_issue_rates.txt 2>&1; head -4 gpurun_out/valu_issue_rates.txt
pushd /tmp > /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ns1_kt -o ns1 -- python $R/tools/ns1_mfma.py > $R/gpurun_out/ns1_mfma.json 2>/dev/null
popd > /dev/null
find gpurun_out/ns1_kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/ns1_kernel_stats.csv \;; grep -i unproject gpurun_out/ns1_kernel_stats.csv | cut -c1-160
for SH in owner tile; do
  BENCH_LIVE_PMC=0 timeout 300 python bench.py --force-dist --sharding $SH --steps 6 --warmup 2 --clock-ramp-steps 4 --no-secondary --no-cpu-baseline > gpurun_out/bench_nccl1_$SH.log 2>&1
  grep '^{"metric"' gpurun_out/bench_nccl1_$SH.log > gpurun_out/bench_nccl1_$SH.json; python -c "
import json; z=json.load(open('gpurun_out/bench_nccl1_$SH.json')); print('nccl world 1 $SH', z['value'], z.get('merge'))" 2>&1 | tail -1 | cut -c1-300
done
timeout 250 python tools/simulate_ranks.py --worlds 1,2,4,8 --steps 12 > gpurun_out/simulate_ranks.jsonl 2>/dev/null; cut -c1-200 gpurun_out/simulate_ranks.jsonl
timeout 250 python tools/simulate_ranks.py --worlds 1,2,4,8 --steps 12 --sharding coherent > gpurun_out/simulate_ranks_coherent.jsonl 2>/dev/null; cut -c1-200 gpurun_out/simulate_ranks_coherent.jsonl
# the N > 1 code path of bench.py on this one GPU (two ranks share device 0, gloo transport: the kernels, the sharding and the
# merge logic are what an N-GPU RCCL run executes; NOT a scaling number)
for SH in owner coherent tile; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --clock-ramp-steps 4 --all-on-device0 --backend gloo --sharding $SH > gpurun_out/bench_n2_$SH.log 2>&1
  grep '^{"metric"' gpurun_out/bench_n2_$SH.log > gpurun_out/bench_n2_$SH.json; cut -c1-400 gpurun_out/bench_n2_$SH.json; python -c "
import json; z=json.load(open('gpurun_out/bench_n2_$SH.json')); print('$SH', z['value'], z.get('merge'))" 2>&1 | tail -1
done
