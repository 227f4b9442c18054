#!/usr/bin/env python3
"""Driver of experiments/packet_sim.c: per-lane vs wave-coherent traversal step counts on the cfg3 frame (CPU, no GPU needed).
Usage: python experiments/packet_sim.py [block_step]   (block_step 4 = every 4th 8x8 block in x and y: a 1/16 sample)"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "cpu-raytracer_amd"))
from pyrtx import host  # noqa: E402

so = os.path.join(HERE, "packet_sim.so")
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "packet_sim.c"), "-lm"])
lib = C.CDLL(so)
FIELDS = ["lane_pairs", "lane_tris", "lane_rays", "wave_iters_lockstep", "pk_pairs", "pk_tris", "pk_packets", "pk_pair_lanes", "pk_tri_lanes", "pk_max_stack", "pk_split_packets"]


class Stats(C.Structure):
    _fields_ = [(f, C.c_double) for f in FIELDS]


step = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = host.atrium_scene(1920, 1080, 3)
b = sc.blas[0]
nodes = np.ascontiguousarray(b.nodes); tris = np.ascontiguousarray(b.tri_hot)
cam = np.concatenate([sc.camera[k][0] for k in ("position", "rotated_top_left_corner", "rotated_x_axis", "rotated_y_axis")]).astype(np.float32)
lights = np.array([[0, *sc.point_lights["position"][0]], [0, *sc.spot_lights["position"][0]], [1, *sc.dir_lights["negative_direction"][0]]], np.float32)
prim, shad = Stats(), Stats()
args_tuple = None
for T in (0, 2, 4, 8, 12, 16, 24, 32, 64):
    lib.hybrid_set(T)
    lib.packet_sim(nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), cam.ctypes.data_as(C.c_void_p), 1920, 1080,
               lights.ctypes.data_as(C.c_void_p), 3, step, C.byref(prim), C.byref(shad), None)
    out = (C.c_double * 5)(); lib.hybrid_get(out)
    pk = max(out[3], 1)
    print('hybrid T=%2d: packet steps/packet %.1f  lane lockstep iters/packet %.1f  lane steps/packet %.1f  est cost/packet %.0f ticks  max packet cost %.0f' % (T, out[0] / pk, out[1] / pk, out[2] / pk, (out[0] * 800 + out[1] * 1100) / pk, out[4]), flush=True)
lib.packet_sim(nodes.ctypes.data_as(C.c_void_p), tris.ctypes.data_as(C.c_void_p), cam.ctypes.data_as(C.c_void_p), 1920, 1080,
               lights.ctypes.data_as(C.c_void_p), 3, step, C.byref(prim), C.byref(shad), None)
for name, s in (("primary", prim), ("shadow0", shad)):
    d = {f: getattr(s, f) for f in FIELDS}
    r, p = d["lane_rays"], max(d["pk_packets"], 1)
    d["per_ray_lane_pairs"] = d["lane_pairs"] / r; d["per_ray_lane_tris"] = d["lane_tris"] / r
    d["per_packet_pairs"] = d["pk_pairs"] / p; d["per_packet_tris"] = d["pk_tris"] / p
    d["pair_occupancy"] = d["pk_pair_lanes"] / max(d["pk_pairs"], 1) / 64; d["tri_occupancy"] = d["pk_tri_lanes"] / max(d["pk_tris"], 1) / 64
    d["lane_steps_per_64rays"] = (d["lane_pairs"] + d["lane_tris"]) / r * 64
    d["lockstep_iters_per_packet"] = d["wave_iters_lockstep"] / p
    print(name, json.dumps({k: round(v, 3) for k, v in d.items()}))
