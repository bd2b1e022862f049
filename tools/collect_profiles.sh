#!/bin/bash
# After tools/_final.sh has run on the GPU box: copy the summaries it left under gpurun_out/ (scratch) into profiles/$ROUND (tracked).
ROUND=${ROUND:-r06}
cd "$(dirname "$0")/.."
P=profiles/$ROUND
mkdir -p $P
for f in bench_n1.json bench_n8_owner.json bench_n8_tile.json bench_nccl1_owner.json bench_nccl1_tile.json pytest_gpu.log smoke.log sweep_forms.jsonl simulate_ranks.jsonl \
         memset_split_semantic.json memset_split_semantic_scannet_2mm.json pmc_voxel_grid.json pmc_semantic.json pmc_semantic_scannet_2mm.json; do
  cp gpurun_out/$f $P/ 2>/dev/null
done
cp gpurun_out/prof_round/pmc_summary.json $P/pmc_summary.json
cp gpurun_out/prof_round/pmc_summary.txt $P/pmc_summary.txt 2>/dev/null
cp gpurun_out/prof_round/sweep_timed_launches.txt $P/sweep_timed_launches.txt 2>/dev/null
find gpurun_out/prof_round -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/rocprofv3_kernel_stats.csv
cp gpurun_out/vg_kernel_stats.csv $P/kernel_stats_voxel_grid.csv
cp gpurun_out/sem_kernel_stats.csv $P/kernel_stats_semantic.csv
cp gpurun_out/sem_scannet_kernel_stats.csv $P/kernel_stats_semantic_scannet_2mm.csv
cp gpurun_out/vg_pmc_summary.txt $P/pmc_voxel_grid.txt
cp gpurun_out/sem_pmc_summary.txt $P/pmc_semantic.txt
cp gpurun_out/sem2_pmc_summary.txt $P/pmc_semantic_scannet_2mm.txt
cp gpurun_out/bench_semantic_profiled.json $P/bench_semantic.json
cp gpurun_out/bench_semantic_scannet_profiled.json $P/bench_semantic_scannet_2mm.json
ls -la $P
