#!/bin/bash
# HBM-side traffic of the frame kernels at HEAD: rocprofv3 PMC passes over the benchmark's own command (FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes: together they exceed the TCC block's 4 counter slots and rocprofv3 aborts with "error code 38: Request exceeds the
# capabilities of the hardware to collect").  Result: gpurun_out/hbm_traffic_TAG.json (copy to profiles/rNN_hbm_traffic.json).
set -e
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/hbm_${TAG}_$C -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --no-extras > $R/gpurun_out/hbm_${TAG}_$C.log 2>&1
done
python3 $R/tools/hbm_traffic_summary.py $R/gpurun_out/hbm_${TAG}_FETCH_SIZE $R/gpurun_out/hbm_${TAG}_WRITE_SIZE > $R/gpurun_out/hbm_traffic_${TAG}.json
cat $R/gpurun_out/hbm_traffic_${TAG}.json
