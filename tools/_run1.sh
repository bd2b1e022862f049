cd $GRAFT_REPO_ROOT
run() { timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['online_mode']['value'])"; }
echo base; run
echo wpe5; HV_TSDF_BATCH_WPE=5 run
echo col8s4; HV_TSDF_BATCH_COL=8 run
echo col4s8; HV_TSDF_BATCH_SPLIT=8 run
echo col4s2; HV_TSDF_BATCH_SPLIT=2 run
echo col2s8; HV_TSDF_BATCH_COL=2 HV_TSDF_BATCH_SPLIT=8 run
echo grid4096; HV_TSDF_GRID=4096 run
echo grid16384; HV_TSDF_GRID=16384 run
