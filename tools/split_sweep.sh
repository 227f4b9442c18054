python3 tools/split_ab.py 2048:8:4 2048:12:4 2048:12:6 2048:16:4 2048:16:8 2048:24:8 2>&1 | grep split= | cut -c1-200
RTX_HIP_LIB=$PWD/cpu-raytracer_amd/csrc/librtx_hip_s6.so python3 tools/split_ab.py 2048:8:4 2>&1 | tail -1 | cut -c1-200
for t in 8 12 16; do echo pipelined T0=$t $(RTX_PK_DEFER=$t python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"ms_per_step\"])"); done
echo pipelined s6 $(RTX_HIP_LIB=$PWD/cpu-raytracer_amd/csrc/librtx_hip_s6.so python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"ms_per_step\"])")
