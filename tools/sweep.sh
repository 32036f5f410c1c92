cd $GRAFT_REPO_ROOT
for cfg in "0 1 --no-priority-stream" "0 1" "0 2 --no-priority-stream" "0 1 --no-priority-stream"; do
  set -- $cfg
  echo "== reserve=$1 streams=$2 $3"
  NUMPYWREN_AMD_RESERVE_CUS=$1 timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams $2 $3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
done
