#!/bin/bash
# The roofline `peak` of bench.py: the chip's rate for the traversal ARITHMETIC alone (experiments/slab_rate.hip, built by
# `make -C experiments` / __graft_entry__.build()).  Output -> gpurun_out/roofline_peak.jsonl (copy to profiles/rNN_roofline_peak.jsonl).
set -e
cd "$(dirname "$0")/.."
./experiments/slab_rate | tee gpurun_out/roofline_peak.jsonl
