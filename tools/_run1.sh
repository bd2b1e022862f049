cd $GRAFT_REPO_ROOT
timeout 150 python -m pytest tests/test_gpu_tsdf.py tests/test_gpu_configs.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -8
timeout 60 python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_touch_patch.json
python -c "import json; d=json.load(open('gpurun_out/bench_touch_patch.json')); print(d['value'], d['ms_per_step'], d['online_mode']['value'])"
timeout 60 python tools/simulate_ranks.py --worlds 1,8 --steps 10 2>/dev/null | cut -c1-200
