for g in 0 1; do
  echo "RTX_GRAPH=$g"
  RTX_GRAPH=$g timeout -k 10 200 python tools/hostcost.py 2>&1 | tail -6
  RTX_GRAPH=$g timeout -k 10 200 python bench.py --workload monkey --cpu-seconds 0 --no-reference-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('monkey', d['ms_per_step'], d['parity'])"
  RTX_GRAPH=$g timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('atrium', d['ms_per_step'], d['parity'])"
  RTX_GRAPH=$g timeout -k 10 200 python bench.py --cpu-seconds 0 --no-extras --simulate-ranks 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shard8', d['ms_per_step'])"
done
