run() { echo "$1 grid=$2 $(RTX_SHADE_GRID=$2 RTX_HIP_LIB=$PWD/cpu-raytracer_amd/csrc/librtx_hip_$1.so python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"ms_per_step\"], round(j[\"kernels\"][\"k_shade\"][\"avg_ms\"],3))")"; }
echo base $(python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-extras 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j[\"ms_per_step\"], round(j[\"kernels\"][\"k_shade\"][\"avg_ms\"],3))")
for g in 2 3 4 5; do run sb256w3 $g; done
for g in 4 6; do run sb192w3 $g; done
for g in 3 4; do run sb320w3 $g; done
for g in 4 8 12; do run sb128w3 $g; done
for g in 8 16; do run sb64w3 $g; done
run sb256w3 4
