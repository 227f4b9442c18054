"""Renders a few cfg3 frames one at a time (RTX_RENDER_SERIAL, one traversal launch per level) — the workload of tools/pmc.sh."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd"))
from pyrtx import api, host
os.environ.setdefault("RTX_SERIAL_MERGE_ANY", "0")
sc = host.atrium_scene(1920, 1080, int(os.environ.get("RTX_PMC_BOUNCES", "3")))
r = api.Renderer(sc)
lane = os.environ.get("RTX_PMC_LANE", "0") == "1"
for _ in range(int(os.environ.get("RTX_PMC_FRAMES", "3"))):
    r.render_async(serial=True, lane_trace=lane)
r.synchronize()
