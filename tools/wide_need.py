"""Prints the packet-stack need of the 4-wide shadow-ray records (rtx_debug_blas_wide) for the bench scenes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cpu-raytracer_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from pyrtx import api, host
sc = host.atrium_scene(1920, 1080, 3)
r = api.Renderer(sc)
print("atrium", [(len(b.nodes), r.debug_blas_wide(i)) for i, b in enumerate(sc.blas)])
import util
for name in util.GOLDENS:
    sc, g = util.load_golden(name)
    r = api.Renderer(sc)
    print(name, [(len(b.nodes), r.debug_blas_wide(i)) for i, b in enumerate(sc.blas)])
