#!/bin/bash
# bench.py (3 frames in flight) with the packet kernels' persistent grids at A blocks per CU (shadow rays) / C (closest hit):
# does leaving SIMD slots free let other frames' latency-bound kernels co-reside?   usage: tools/grid_ab.sh "A:C" ...
for spec in "$@"; do
  a=${spec%%:*}; c=${spec##*:}
  out=$(RTX_PK_GRID_ANY=$a RTX_PK_GRID_CLOSEST=$c python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | tail -1)
  echo "any=$a closest=$c $(echo "$out" | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['value'], {k: round(x['avg_ms'],3) for k,x in j['kernels'].items()})")"
done
