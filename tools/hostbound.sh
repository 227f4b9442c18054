#!/bin/bash
# is the 1/8 tile shard (strong scaling at 8 GPUs) limited by host-side submission?  with / without the per-launch HIP events
B=$(dirname $0)/../bench.py
for n in 1 8; do for t in "" "--no-kernel-timing"; do for f in 3 6; do timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks $n --frames-in-flight $f --steps 90 $t 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks', $n, 'fif', $f, '$t', j['ms_per_step'], flush=True)"; done; done; done
