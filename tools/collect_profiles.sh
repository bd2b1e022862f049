#!/bin/bash
# After tools/_final.sh has run on the GPU box: copy the summaries it left under gpurun_out/ (scratch) into profiles/$ROUND (tracked).
ROUND=${ROUND:-r05}
cd "$(dirname "$0")/.."
P=profiles/$ROUND
mkdir -p $P
cp gpurun_out/bench_n1.json gpurun_out/bench_n2_owner.json gpurun_out/bench_n2_coherent.json gpurun_out/bench_n2_tile.json $P/
cp gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/sweep_forms.jsonl gpurun_out/simulate_ranks.jsonl gpurun_out/simulate_ranks_coherent.jsonl $P/
cp gpurun_out/pipeline_timeline.txt gpurun_out/rank8_timeline.txt $P/
cp gpurun_out/prof_round/pmc_summary.json $P/pmc_summary.json
cp gpurun_out/prof_round/pmc_summary.txt $P/pmc_summary.txt 2>/dev/null
find gpurun_out/prof_round -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $P/rocprofv3_kernel_stats.csv
cp gpurun_out/vg_kernel_stats.csv $P/kernel_stats_voxel_grid.csv
cp gpurun_out/sem_kernel_stats.csv $P/kernel_stats_semantic.csv
cp gpurun_out/sem_scannet_kernel_stats.csv $P/kernel_stats_semantic_scannet_2mm.csv
cp gpurun_out/pmc_voxel_grid.json $P/pmc_voxel_grid.json
cp gpurun_out/vg_pmc_summary.txt $P/pmc_voxel_grid.txt
cp gpurun_out/pmc_semantic.json gpurun_out/pmc_semantic_scannet_2mm.json $P/ 2>/dev/null
cp gpurun_out/sem_pmc_summary.txt $P/pmc_semantic.txt
cp gpurun_out/sem2_pmc_summary.txt $P/pmc_semantic_scannet_2mm.txt
cp gpurun_out/bench_semantic_profiled.json $P/bench_semantic.json
cp gpurun_out/bench_semantic_scannet_profiled.json $P/bench_semantic_scannet_2mm.json
ls -la $P
# round 5 additions
cp gpurun_out/valu_issue_rates.txt $P/valu_issue_rates.txt 2>/dev/null
cp gpurun_out/ns1_mfma.json $P/ns1_mfma.json 2>/dev/null
cp gpurun_out/ns1_kernel_stats.csv $P/kernel_stats_ns1.csv 2>/dev/null
cp gpurun_out/bench_nccl1_owner.json gpurun_out/bench_nccl1_tile.json $P/ 2>/dev/null
cp gpurun_out/memset_split_semantic.json gpurun_out/memset_split_semantic_scannet_2mm.json $P/ 2>/dev/null
ls -la $P
