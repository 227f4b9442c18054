for f in 1 2 3 4; do for m in "" "--serial"; do timeout -k 10 200 python $(dirname $0)/../bench.py --cpu-seconds 0 --no-extras --frames-in-flight $f $m 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fif', $f, '$m', j['ms_per_step'], j['value'])"; done; done
