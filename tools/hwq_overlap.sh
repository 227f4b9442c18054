#!/bin/bash
# two-stream overlap mode with enough hardware queues for 2 streams x 3 frames
B=$(dirname $0)/../bench.py
for q in 4 8; do for n in 1 8; do for m in "--serial" "--overlap"; do for f in 2 3; do GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks $n --frames-in-flight $f --steps 90 $m 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hwq', $q, 'ranks', $n, '$m', 'fif', $f, j['ms_per_step'], flush=True)"; done; done; done; done
