#!/bin/bash
B=$(dirname $0)/../bench.py
for n in 1 8; do for f in 2 3 4 6; do timeout -k 10 200 python $B --cpu-seconds 0 --no-extras --simulate-ranks $n --frames-in-flight $f --steps 60 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ranks', $n, 'fif', $f, 'ms/step', j['ms_per_step'])"; done; done
