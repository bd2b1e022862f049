cd $GRAFT_REPO_ROOT
run() { timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['online_mode']['value'])"; }
echo base; run
echo rtab; HV_TSDF_BATCH_RTAB=1 run
echo base; run
echo rtab; HV_TSDF_BATCH_RTAB=1 run
HV_TSDF_BATCH_RTAB=1 timeout 250 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -3
