cd $GRAFT_REPO_ROOT
timeout 40 python -m pytest tests/test_gpu_tsdf.py -x -q 2>&1 | tail -2
run() { timeout 20 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['online_mode']['value'])"; }
echo order1; run
echo order0; HV_TSDF_BATCH_ORDER=0 run
echo order1 w8; timeout 15 python tools/simulate_ranks.py --worlds 8 --steps 10 2>/dev/null | cut -c1-170
echo order0 w8; HV_TSDF_BATCH_ORDER=0 timeout 15 python tools/simulate_ranks.py --worlds 8 --steps 10 2>/dev/null | cut -c1-170
