#!/usr/bin/env python3
"""Which subtree first for shadow rays?  Per-ray any-hit walks of the cfg3 frame's level-0 shadow rays under seven child-order policies
(experiments/packet_sim.c order_sim; CPU only).  Usage: python experiments/order_sim.py [block_step]"""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "cpu-raytracer_amd"))
from pyrtx import host  # noqa: E402
so = os.path.join(HERE, "packet_sim.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "packet_sim.c"), "-lm"])
lib = C.CDLL(so)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = host.atrium_scene(1920, 1080, 3)
b = sc.blas[0]
nodes = np.ascontiguousarray(b.nodes); tris = np.ascontiguousarray(b.tri_hot)
cam = np.concatenate([sc.camera[k][0] for k in ("position", "rotated_top_left_corner", "rotated_x_axis", "rotated_y_axis")]).astype(np.float32)
lights = np.array([[0, *sc.point_lights["position"][0]], [0, *sc.spot_lights["position"][0]], [1, *sc.dir_lights["negative_direction"][0]]], np.float32)
out = (C.c_double * 28)()
lib.order_sim(nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), cam.ctypes.data_as(C.c_void_p), 1920, 1080, lights.ctypes.data_as(C.c_void_p), 3, step, out)
names = ["left first", "larger area first", "front to back", "back to front", "smaller t_near first", "longer overlap first", "area x overlap first"]
for p, n in enumerate(names):
    pairs, tri, rays, occ = out[4 * p:4 * p + 4]
    print("%-22s pair steps/ray %6.2f  triangle tests/ray %5.2f  weighted (2 + 2.5 per) %7.2f  occluded %.3f" % (n, pairs / rays, tri / rays, (2 * pairs + 2.5 * tri) / rays, occ / rays))
