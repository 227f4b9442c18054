#!/bin/bash
# Memory-path counters of one frame's kernels (two --pmc passes): TLB, write latency, wait cycles.  usage: tools/pmc_mem.sh TAG
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcmem_${TAG}_1 -- python3 $R/tools/pmc_frame.py > $R/gpurun_out/pmcmem_${TAG}_1.log 2>&1
rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcmem_${TAG}_2 -- python3 $R/tools/pmc_frame.py > $R/gpurun_out/pmcmem_${TAG}_2.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcmem_${TAG}_1 $R/gpurun_out/pmcmem_${TAG}_2 > $R/gpurun_out/pmcmem_${TAG}_summary.json
