#!/bin/bash
# Everything profiles/rNN_* is made of, in one call on the GPU box: bench line, same command under rocprofv3 --kernel-trace --stats,
# HBM-side PMC traffic, roofline peak, simulated tile shards, cfg2 / cfg5 lines.  usage: tools/profile_round.sh r02x
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
# PMC traffic first: bench.py quotes the newest profiles/r*_hbm_traffic.json as roofline.traffic
bash tools/hbm_traffic.sh $TAG > /dev/null 2>&1 && cp gpurun_out/hbm_traffic_$TAG.json $O/hbm_traffic.json && cp $O/hbm_traffic.json profiles/${TAG}_hbm_traffic.json
[ -x experiments/gather_occupancy ] && experiments/gather_occupancy > $O/gather_occupancy.txt 2>&1 && cp $O/gather_occupancy.txt profiles/${TAG}_gather_occupancy.txt
python3 bench.py > $O/bench_atrium.json 2> $O/bench_atrium.err
bash tools/roofline_peak.sh > /dev/null && cp gpurun_out/roofline_peak.jsonl $O/roofline_peak.jsonl
bash tools/simscale2.sh > $O/simulated_tile_shards.txt 2>&1
python3 bench.py --workload monkey --cpu-seconds 2 --no-reference-baseline > $O/bench_monkey_cfg2.json 2> /dev/null
python3 tools/bench_cfg5.py > $O/bench_cfg5_dynamic.json 2> $O/bench_cfg5.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --cpu-seconds 0 --no-extras > $O/bench_atrium_same_command.json 2> $O/stats.log )
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_atrium_same_command.csv \;
ls $O
