"""ctypes mirrors of the structs in include/rtx.h (the C ABI of the render path)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import scene_io as sio

RTX_MAX_MIP_LEVELS = 16


class RtxConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("bounces", C.c_int32), ("stack_size", C.c_int32),
                ("traversal_strategy", C.c_int32), ("texture_mode", C.c_int32), ("mip_filter", C.c_int32),
                ("max_anisotropy", C.c_float), ("device", C.c_int32), ("heatmap", C.c_int32), ("reserved", C.c_int32 * 6)]


class RtxTextureDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("mipmapped", C.c_int32), ("mip_levels", C.c_int32),
                ("mip_offsets", C.c_int32 * RTX_MAX_MIP_LEVELS)]


class RtxCamera(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("rotated_top_left_corner", C.c_float * 3),
                ("rotated_x_axis", C.c_float * 3), ("rotated_y_axis", C.c_float * 3)]


class RtxFrame(C.Structure):
    _fields_ = [("camera", RtxCamera), ("ambient", C.c_float * 3), ("pad0", C.c_int32),
                ("tlas_nodes", C.c_void_p), ("tlas_node_count", C.c_int32),
                ("tlas_indices", C.c_void_p), ("tlas_index_count", C.c_int32),
                ("instances", C.c_void_p), ("instance_count", C.c_int32),
                ("spheres", C.c_void_p), ("sphere_count", C.c_int32),
                ("planes", C.c_void_p), ("plane_count", C.c_int32),
                ("point_lights", C.c_void_p), ("point_light_count", C.c_int32),
                ("spot_lights", C.c_void_p), ("spot_light_count", C.c_int32),
                ("directional_lights", C.c_void_p), ("directional_light_count", C.c_int32)]


class RtxStats(C.Structure):
    _fields_ = [("num_primary_rays", C.c_uint64), ("num_shadow_rays", C.c_uint64),
                ("num_reflection_rays", C.c_uint64), ("num_refraction_rays", C.c_uint64)]

    def as_dict(self):
        return {"primary": self.num_primary_rays, "shadow": self.num_shadow_rays,
                "reflection": self.num_reflection_rays, "refraction": self.num_refraction_rays}


_WORK_FIELDS = ["closest_rays", "any_rays", "tlas_nodes_closest", "tlas_nodes_any", "blas_nodes_closest", "blas_nodes_any",
                "instances_closest", "instances_any", "tri_tests_closest", "tri_tests_any", "triangle_hits", "shaded_hits",
                "sky_lookups", "texel_fetches", "rays_spawned"]


class RtxWork(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in _WORK_FIELDS] + [("reserved", C.c_uint64 * 5)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in _WORK_FIELDS}
        # RTX_RENDER_PACKET_STATS (csrc/rtx_packet.h): packet-walk figures, closest-hit and shadow-ray launches summed
        for i, n in enumerate(("pk_packets", "pk_pair_steps", "pk_tri_steps", "pk_pair_lanes", "pk_tri_lanes")):
            d[n] = int(self.reserved[i])
        return d


assert C.sizeof(RtxConfig) == 64 and C.sizeof(RtxTextureDesc) == 80 and C.sizeof(RtxCamera) == 48
assert C.sizeof(RtxWork) == 160


def fill_frame(fr: RtxFrame, sc: "sio.Scene", keep: list) -> None:
    """Point an rtx_frame at the scene's numpy arrays (kept alive through `keep`)."""
    C.memmove(C.byref(fr.camera), sc.camera.ctypes.data, C.sizeof(RtxCamera))
    amb = np.asarray(sc.ambient, np.float32)
    for i in range(3):
        fr.ambient[i] = float(amb[i])

    def ptr(arr, dt):
        a = np.ascontiguousarray(arr, dtype=dt)
        keep.append(a)
        return a.ctypes.data if len(a) else None

    fr.tlas_nodes = ptr(sc.tlas_nodes, sio.BVH_NODE); fr.tlas_node_count = len(sc.tlas_nodes)
    fr.tlas_indices = ptr(sc.tlas_indices, np.int32); fr.tlas_index_count = len(sc.tlas_indices)
    fr.instances = ptr(sc.instances, sio.INSTANCE); fr.instance_count = len(sc.instances)
    fr.spheres = ptr(sc.spheres, sio.SPHERE); fr.sphere_count = len(sc.spheres)
    fr.planes = ptr(sc.planes, sio.PLANE); fr.plane_count = len(sc.planes)
    fr.point_lights = ptr(sc.point_lights, sio.POINT_LIGHT); fr.point_light_count = len(sc.point_lights)
    fr.spot_lights = ptr(sc.spot_lights, sio.SPOT_LIGHT); fr.spot_light_count = len(sc.spot_lights)
    fr.directional_lights = ptr(sc.dir_lights, sio.DIR_LIGHT); fr.directional_light_count = len(sc.dir_lights)
