#!/bin/bash
# single-GPU rehearsal of the strong-scaling runs: the tile shard rank 0 of N would own (no gather), 3 frames in flight
B=$(dirname $0)/../bench.py
for n in 1 2 4 8; do timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks $n --steps 60 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks', $n, 'ms/step', j['ms_per_step'], {k: round(v['avg_ms'], 4) for k, v in j['kernels'].items()})"; done
