cd $GRAFT_REPO_ROOT
timeout 250 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -15
for v in 1 0; do echo "MULT $v"; HV_TSDF_BATCH_MULT=$v timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200; done
