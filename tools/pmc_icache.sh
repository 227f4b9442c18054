#!/bin/bash
# Instruction-cache and wait counters of one frame's kernels (one --pmc pass), for A/B runs of the fused trace+shade kernel.
# usage: tools/pmc_icache.sh TAG [env assignments for pmc_frame.py are inherited]  -> gpurun_out/pmcic_TAG/, gpurun_out/pmcic_TAG_summary.json
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcic_${TAG} -- python3 $R/tools/pmc_frame.py > $R/gpurun_out/pmcic_${TAG}.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcic_${TAG} > $R/gpurun_out/pmcic_${TAG}_summary.json
