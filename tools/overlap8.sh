#!/bin/bash
# 1/8 tile shard: one stream per frame (merged shadow launch) vs the two-stream overlap, by frames in flight
B=$(dirname $0)/../bench.py
for m in "--serial" "--overlap"; do for f in 1 2 3 4; do timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks 8 --frames-in-flight $f --steps 90 $m 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks 8', '$m', 'fif', $f, j['ms_per_step'], flush=True)"; done; done
