cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh 2>&1 | tee gpurun_out/profile_round.txt
timeout 250 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; tail -1 gpurun_out/bench_n1.json | cut -c1-1500
