#!/bin/bash
# k_shade A/B: block size variants x grid sizes; prints the level-0 / total k_shade time of one cfg3 frame (serial mode)
cd $(dirname $0)/..
for lib in default b256 b192 b128 b256w4; do for g in 2 4 8 16; do
  L=""; [ $lib != default ] && L=cpu-raytracer_amd/csrc/librtx_hip_$lib.so
  RTX_HIP_LIB=$L RTX_SHADE_GRID=$g timeout -k 10 120 python tools/perlevel2.py 2>/dev/null | sed -n 1,18p | grep shade | tr '\n' ' ' | awk -v l=$lib -v g=$g '{print l, "grid x" g, $2, $5, $8, $11}'
done; done
