for sm in 255 64 32 16 8 4; do RTX_PK_ITEM_SMAX=$sm python3 tools/split_ab.py 2048 2>&1 | tail -1 | sed "s/^/smax=$sm /"; done
RTX_PK_ITEM_SMAX=16 python3 tools/split_ab.py 0 2>&1 | tail -1 | sed "s/^/smax=16 nosplit /"
for sm in 16 8; do echo pipelined smax=$sm $(RTX_PK_ITEM_SMAX=$sm RTX_PK_SPLIT=2048 python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"ms_per_step\"], {k: round(x[\"avg_ms\"],3) for k,x in j[\"kernels\"].items()})"); done
