#!/bin/bash
# one-stream mode: one shadow-ray launch per level vs one for all levels, full frame and 1/2, 1/4, 1/8 tile shards
B=$(dirname $0)/../bench.py
for m in 0 1; do for n in 1 2 4 8; do RTX_SERIAL_MERGE_ANY=$m timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks $n --steps 90 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge_any', $m, 'ranks', $n, j['ms_per_step'], flush=True)"; done; done
