#!/bin/bash
# single-GPU rehearsal of the strong-scaling runs: time the tile shard rank 0 of N would own, for several frames-in-flight settings
B=$(dirname $0)/../bench.py
for n in 1 2 4 8; do for f in 3 4 6 8; do timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks $n --frames-in-flight $f --steps 90 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks', $n, 'fif', $f, j['ms_per_step'])"; done; done
