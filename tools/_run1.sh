cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 170 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/final_pytest.log; cat gpurun_out/final_pytest.log
timeout 90 python bench.py 2> gpurun_out/bench_n1.err | tail -1 > gpurun_out/bench_n1.json
python -c "import json; d=json.load(open('gpurun_out/bench_n1.json')); print(d['value'], d['ms_per_step'], d['online_mode']['value'], d['roofline']['frac'], d['online_mode']['roofline']['frac'], d['cpu_baseline'])"
timeout 60 python tools/simulate_ranks.py --worlds 1,2,4,8 --steps 10 2>/dev/null > gpurun_out/simulate_ranks.jsonl; cut -c1-120 gpurun_out/simulate_ranks.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1
cut -c1-50,190-330 $R/gpurun_out/prof_kt/bench_kernel_stats.csv | head -12
