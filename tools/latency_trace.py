#!/usr/bin/env python3
"""One cfg3 frame at a time in the default (two-stream) shape, for a kernel trace: rocprofv3 --kernel-trace ... -- python3 tools/latency_trace.py;
then tools/trace_gaps.py on the CSV shows what the 1.4 ms of a frame consist of (launch durations, gaps per stream, overlap of the two streams)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd"))
from pyrtx import api, host
sc = host.atrium_scene(1920, 1080, 3, detail=1)
r = api.Renderer(sc)
for _ in range(int(os.environ.get("FRAMES", "12"))):
    r.render_async()
    r.synchronize()
