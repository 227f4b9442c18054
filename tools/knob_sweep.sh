#!/bin/bash
# Throughput-mode (bench.py, 3 frames in flight) sweep of one environment knob.  usage: tools/knob_sweep.sh KNOB v1 v2 ... [-- extra bench args]
K=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for v in "${VALS[@]}"; do
  printf "%s=%s  " "$K" "$v"
  env $K=$v python bench.py --no-extras --cpu-seconds 0 --steps 60 "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in j['kernels'].items()})"
done
