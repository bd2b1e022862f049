cd $GRAFT_REPO_ROOT
timeout 100 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['online_mode']['value'])"
timeout 700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
