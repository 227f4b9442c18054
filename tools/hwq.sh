#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): frames in flight beyond that share a queue and serialise.
# Sweep of hardware queues x frames in flight for the full frame (ranks 1) and the 1/8 tile shard (ranks 8).
B=$(dirname $0)/../bench.py
for q in 4 8; do for n in 1 8; do for f in 3 4 6 8; do GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks $n --frames-in-flight $f --steps 90 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq', $q, 'ranks', $n, 'fif', $f, j['ms_per_step'], flush=True)"; done; done; done
